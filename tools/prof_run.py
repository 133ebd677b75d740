import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from thermompnn_amd.engine import Engine
from thermompnn_amd.weights import synthetic_state_dict
dev = torch.device("cuda:0")
eng = Engine(synthetic_state_dict(0), dev, 48)
b = bench.build_batch(64, 256, 3, dev)
for _ in range(40):
    eng.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=256, check_status=False)
torch.cuda.synchronize()
