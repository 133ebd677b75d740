"""Per-kernel HIP-event times of the bench batch for whatever library TMPNN_LIB names — NO status check and no JSON contract:
the timing loop for tools/ablate.sh, whose variants compute garbage on purpose (bench.py refuses a run whose status word is set)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from thermompnn_amd import _lib  # noqa: E402
from thermompnn_amd.engine import Engine  # noqa: E402
from thermompnn_amd.weights import synthetic_state_dict  # noqa: E402

steps, warm = int(os.environ.get("AB_STEPS", "40")), int(os.environ.get("AB_WARMUP", "20"))
dev = torch.device("cuda:0")
lib = _lib.load()
eng = Engine(synthetic_state_dict(0), dev, 48)
b = bench.build_batch(64, 256, 0, dev)
out = {"ddg": torch.empty((b["T"], 21), device=dev)}
f = lambda: eng.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=256, out=out, check_status=False)
for _ in range(warm):
    f()
torch.cuda.synchronize()
lib.tmpnn_profile_enable(1)
t0 = time.perf_counter()
for _ in range(steps):
    f()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
k = {n: round(ms / c, 4) for n, (ms, c) in bench.fetch_profile(lib).items()}
print(f"{sys.argv[1] if len(sys.argv) > 1 else '-':12s} {dt * 1e3:.3f} ms/step  {k}")
