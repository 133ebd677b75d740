#!/bin/bash
# One gpurun call that refreshes every artefact under profiles/ for round $1 (e.g. r02). Run on the GPU box:
#   tools/measure_round.sh r03        -> gpurun_out/$1_*  (copy into profiles/ afterwards)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06}; O=$R/gpurun_out
mkdir -p $O
cd $R
# the GPU parity suite FIRST: it writes gpurun_out/parity_worst_errors.json (worst |hip - reference| per precision and tensor), which
# is copied under the round's name right behind it (round 4 committed an empty file: the copy ran before the suite)
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/${TAG}_gpu_parity.log 2>&1; tail -3 $O/${TAG}_gpu_parity.log
cp $O/parity_worst_errors.json $O/${TAG}_parity_worst_errors.json
python tools/isa_counts.py $O/${TAG}_isa_counts.json > /dev/null 2>&1
(tools/probe/valu_cost_probe; tools/probe/hetero_probe; tools/probe/ilv_probe_asm1; tools/probe/spec_probe_pf2; tools/probe/prio_probe; tools/probe/helper_probe; tools/probe/shadow_probe; tools/probe/dep_probe) > $O/${TAG}_probes.txt 2>&1
bash tools/pmc_traffic.sh > $O/${TAG}_pmc_traffic.log 2>&1
cp $O/pmc_traffic/pmc_traffic.json $O/${TAG}_pmc_traffic.json; cp $O/pmc_traffic/FETCH_SIZE_per_kernel.csv $O/${TAG}_pmc_FETCH_SIZE_per_kernel.csv; cp $O/pmc_traffic/WRITE_SIZE_per_kernel.csv $O/${TAG}_pmc_WRITE_SIZE_per_kernel.csv
cp $O/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json   # bench.py reads the stamped file: PMC pass first, so the bench line of THIS call carries traffic
python bench.py > $O/${TAG}_bench_full.json 2> $O/${TAG}_bench_full.err
python tools/run_configs.py > $O/${TAG}_configs_2to5.json 2> $O/${TAG}_configs.err
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/prof_$TAG && \
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o $TAG -- python $R/bench.py --steps 10 --warmup 3 --no-extras > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_rocprof.err )
cp $(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv 2>/dev/null
# the same trace restricted to the 10 timed steps (the --stats CSV also averages the clock warm-up launches)
python tools/rocprof_timed_stats.py $(find $O/prof_$TAG -name "*kernel_trace.csv" | head -1) 10 > $O/${TAG}_bench_kernel_stats_timed.csv
python bench.py --scaling strong --steps 20 --warmup 5 > $O/${TAG}_bench_strong_1gpu.json 2> $O/${TAG}_bench_strong.err
TMPNN_BENCH_ONE_DEVICE=1 TMPNN_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --proteins-per-gpu 8 --no-cpu-baseline > $O/${TAG}_bench_selflaunch_2ranks_one_device.json 2> $O/${TAG}_bench_selflaunch.err
# round 6: the line the driver would run cold on an 8-GPU node, executed once with EIGHT ranks (all on this one device, gloo): collective.ranks_seen == 8
TMPNN_BENCH_ONE_DEVICE=1 TMPNN_BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 5 --warmup 2 --proteins-per-gpu 8 --no-cpu-baseline > $O/${TAG}_bench_selflaunch_8ranks_one_device.json 2> $O/${TAG}_bench_selflaunch8.err
# RCCL itself, on this 1-GPU box: a process group of ONE rank (init_process_group("nccl"), device-buffer all_gather_into_tensor, the async overlap)
TMPNN_BENCH_FORCE_GROUP=1 python bench.py --steps 20 --warmup 5 --no-extras > $O/${TAG}_bench_rccl_1rank.json 2> $O/${TAG}_bench_rccl_1rank.err
bash tools/pmc_sq.sh > $O/${TAG}_pmc_sq.log 2>&1
cp $O/pmc_sq_summary.txt $O/${TAG}_pmc_sq_summary.txt
# round 4: the end-to-end leg on its own, the kernel-boundary probe (+ its rocprofv3 trace), the margin probe
python tools/e2e_bench.py > $O/${TAG}_e2e.json 2> $O/${TAG}_e2e.err
python tools/gap_probe.py > $O/${TAG}_gap_probe.json 2> $O/${TAG}_gap_probe.err
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/gaptrace_$TAG && rocprofv3 --kernel-trace --output-format csv -d $O/gaptrace_$TAG -o gap -- python $R/tools/gap_probe.py --trace > /dev/null 2> $O/${TAG}_gaptrace.err )
python tools/gap_probe.py --gaps $(find $O/gaptrace_$TAG -name "*kernel_trace.csv" | head -1) > $O/${TAG}_gap_trace_summary.json 2>> $O/${TAG}_gaptrace.err
rm -rf $O/gaptrace_$TAG
python tools/margin_probe.py > $O/${TAG}_margin_probe.json 2> $O/${TAG}_margin_probe.err
bash tools/power_probe.sh $TAG > /dev/null 2>&1      # board power / sclk beside the running bench batch -> ${TAG}_power_samples.txt
# round 6: the edge update's ablation ladder (8-wavefront form and the wavefront-per-block experiment) and the message pass's, in step time + cycles
TAG=$TAG tools/ab_edge.sh run > /dev/null 2>&1
TAG=$TAG tools/ab_msg_wave.sh run > /dev/null 2>&1
rm -rf $O/prof_$TAG/*/*.db 2>/dev/null
tail -c 600 $O/${TAG}_bench_full.err; head -c 400 $O/${TAG}_bench_full.json; echo; head -9 $O/${TAG}_bench_kernel_stats_timed.csv; tail -12 $O/${TAG}_pmc_traffic.log
