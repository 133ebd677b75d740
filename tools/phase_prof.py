import os as _os
_os.environ.setdefault("TMPNN_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "thermompnn_amd", "libtmpnn_debug.so"))   # TMPNN_*_PROF timers exist only in the debug variant
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from thermompnn_amd.engine import Engine
from thermompnn_amd.weights import synthetic_state_dict
dev = torch.device("cuda:0")
eng = Engine(synthetic_state_dict(0), dev, 48)
b = bench.build_batch(64, 256, 3, dev)
for _ in range(2):
    eng.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=256, check_status=False)
torch.cuda.synchronize()
