#!/bin/bash
# A/B harness for one gpurun call: every argument is "name:ENV=VAL,ENV=VAL" (or just "name"); runs the timed bench
# workload once per configuration and leaves one JSON line per configuration in gpurun_out/ab_<name>.json.
#   tools/ab_run.sh base wt0:TMPNN_WT=0 pipe0:TMPNN_LIB=tools/ab/libtmpnn_pipe0.so
mkdir -p gpurun_out
STEPS=${AB_STEPS:-30}
WARM=${AB_WARMUP:-15}
for spec in "$@"; do
  name=${spec%%:*}
  envs=""
  if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  env $envs python bench.py --steps $STEPS --warmup $WARM --no-extras > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1])
    k = {a: round(b["avg_ms"], 4) for a, b in d.get("kernels", {}).items()}
    print(f"{n:12s} {d['value']/1e6:8.2f} M preds/s  {d['ms_per_step']:.3f} ms  {k}")
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/ab_{n}.err").read()[-800:])
PY
done
