"""Workload for the rocprofv3 --pmc passes (HBM traffic per launch): 3 forwards of the bench batch plus two kernels
with KNOWN byte counts to calibrate FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md §HBM):
  gather_rows: reads 3.1 MB idx + 8.4 MB table (L2-resident), writes 402.7 MB
  torch copy : reads 402.7 MB, writes 402.7 MB
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_batch  # noqa: E402
from thermompnn_amd.engine import Engine  # noqa: E402
from thermompnn_amd.weights import synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
eng = Engine(synthetic_state_dict(0), dev, 48)
b = build_batch(64, 256, 0, dev)
out = {"ddg": torch.empty((b["T"], 21), device=dev)}
for _ in range(3):
    eng.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=256, out=out)
g = torch.Generator().manual_seed(0)
nodes = torch.randn(16384, 128, generator=g).to(dev)
idx = ((torch.arange(16384) // 256 * 256).repeat_interleave(48) + torch.randint(0, 256, (16384 * 48,), generator=g)).int().to(dev)
for _ in range(3):
    o = eng.gather_rows(nodes, idx)
dst = torch.empty_like(o)
for _ in range(3):
    dst.copy_(o)
torch.cuda.synchronize()
print("ok", o.numel() * 4)
