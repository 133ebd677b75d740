#!/bin/bash
# per-kernel time vs batch size (prologue / fixed cost = intercept): tools/ab_scale.sh "ENV=VAL ..." 8 16 32 64 128
envs=$1; shift
for B in "$@"; do
  env $envs python bench.py --steps 15 --warmup 8 --no-extras --proteins-per-gpu $B > /tmp/sc.json 2>/tmp/sc.err
  python - $B <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/sc.json").read().strip().splitlines()[-1])
    k = {a: round(b["avg_ms"] * 1e3, 1) for a, b in d.get("kernels", {}).items()}
    print(f"B={sys.argv[1]:>4s} T={int(sys.argv[1])*256:6d} {d['ms_per_step']:.3f} ms  us: {k}")
except Exception as e:
    print("FAILED", e, open("/tmp/sc.err").read()[-500:])
PY
done
