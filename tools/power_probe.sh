#!/bin/bash
# What the board reports while the bench batch runs back to back (run on the GPU box): rocm-smi power / clocks / temperature sampled once a
# second beside ~12 s of forwards, then once more idle -> gpurun_out/${1:-r06}_power_samples.txt. The evidence behind "the pipeline is
# power-limited" other than the kernel's own cycle counter (docs/NOTEBOOK.md 9.10).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r06}_power_samples.txt
cd $R; mkdir -p gpurun_out
sample() { (rocm-smi --showpower --showclocks --showtemp --showperflevel --showmaxpower 2>&1 || amd-smi metric -p -c -t 2>&1) | grep -v "^$" | grep -iv "^=\|WARNING" | head -40; }
{ echo "### idle, before"; sample
  python tools/prof_run.py 4000 > /dev/null 2>&1 &
  PID=$!
  sleep 4
  for k in 1 2 3 4 5 6; do echo "### under load, sample $k"; sample; sleep 1; done
  kill $PID 2>/dev/null; wait $PID 2>/dev/null
  sleep 3; echo "### idle, after"; sample; } > $O 2>&1
tail -60 $O
