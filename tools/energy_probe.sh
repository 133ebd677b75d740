#!/bin/bash
# tools/probe/mfma_energy_probe (one instruction class at a time, all CUs, ~1 s each) with rocm-smi power / sclk sampled beside it
# (run on the GPU box) -> gpurun_out/${1:-r06}_energy_probe.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r06}_energy_probe.txt
cd $R; mkdir -p gpurun_out
( tools/probe/mfma_energy_probe > $O.probe 2>&1 & )
sleep 0.5
for k in $(seq 1 26); do
  echo "t=$(date +%s.%N | cut -c7-14) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|Package Power (W)" | sed 's/.*: //' | tr '\n' ' ')"
  sleep 0.12
done > $O.power
sleep 1
{ echo "# tools/probe/mfma_energy_probe.hip: one instruction class per ~1 s launch, 256 CUs x 8 wavefronts"; cat $O.probe
  echo "# rocm-smi beside it (sclk, package power W), in launch order"; cat $O.power; } > $O
rm -f $O.probe $O.power; cat $O
