"""40 forwards of the bench batch (64 x L=256) for the debug library's phase timers (tools only): clocks are settled by the time the last
lines are printed.   TMPNN_LIB=thermompnn_amd/libtmpnn_debug.so TMPNN_MSG_PROF=1 python tools/prof_run.py 2>&1 | grep phases | tail -1
(the wavefront-per-residue message kernel also prints the shader clock it ran at: cycle counter against the 100 MHz reference)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from thermompnn_amd.engine import Engine  # noqa: E402
from thermompnn_amd.weights import synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
eng = Engine(synthetic_state_dict(0), dev, 48)
b = bench.build_batch(64, 256, 3, dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    eng.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=256, check_status=False)
torch.cuda.synchronize()
