"""TMPNN_NODE_PROF=1 python tools/node_prof.py  (GPU box): stage stamps of node_update8_deep_kernel on one L=256 protein"""
import os as _os
_os.environ.setdefault("TMPNN_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "thermompnn_amd", "libtmpnn_debug.so"))   # TMPNN_*_PROF timers exist only in the debug variant
import sys
import torch
sys.path.insert(0, ".")
import bench
from thermompnn_amd.engine import Engine
from thermompnn_amd.weights import synthetic_state_dict
dev = torch.device("cuda:0")
eng = Engine(synthetic_state_dict(0), dev)
b = bench.build_batch(1, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 7, dev)
for _ in range(3):
    eng.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=b["L"])
    torch.cuda.synchronize()
    print("----", file=sys.stderr)
