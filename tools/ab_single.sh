#!/bin/bash
# single-protein latency A/B in one gpurun call: every argument is "name:ENV=VAL,ENV=VAL" (or just "name").
#   tools/ab_single.sh base deep0:TMPNN_NODE_DEEP=0 d2:TMPNN_LIB=thermompnn_amd/libtmpnn_d2.so
for spec in "$@"; do
  name=${spec%%:*}
  envs=""
  if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  env $envs python bench.py --steps 15 --warmup 8 --no-extras --proteins-per-gpu 1 > /tmp/s1.json 2>/tmp/s1.err
  env $envs python - "$name" <<'PY'
import json, sys, time
import torch
sys.path.insert(0, ".")
import bench
from thermompnn_amd.engine import Engine
from thermompnn_amd.weights import synthetic_state_dict
name = sys.argv[1]
try:
    d = json.loads(open("/tmp/s1.json").read().strip().splitlines()[-1])
    k = {a: round(b["avg_ms"] * 1e3, 1) for a, b in d.get("kernels", {}).items()}
except Exception as e:
    k = {"FAILED": repr(e), "err": open("/tmp/s1.err").read()[-300:]}
dev = torch.device("cuda:0")
eng = Engine(synthetic_state_dict(0), dev)
out = {}
for L in (256, 2048):
    b = bench.build_batch(1, L, 7, dev)
    f = lambda: eng.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=L, check_status=False)
    for _ in range(30): f()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(200): f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200)
    out[L] = round(best * 1e3, 4)
print(f"{name:10s} L256 {out[256]:.4f} ms  L2048 {out[2048]:.4f} ms  per-kernel us (event-timed, L=256): {k}")
PY
done
