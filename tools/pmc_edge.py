"""Run each full enc_edge variant a few times on a fixed input (for rocprofv3 --pmc SQ_* passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermompnn_amd import _lib
from thermompnn_amd.engine import Engine, _ptr, _stream
from thermompnn_amd.weights import synthetic_state_dict
lib = _lib.load(); eng = Engine(synthetic_state_dict(0), "cuda:0")
T = 16384
g = torch.Generator().manual_seed(0)
P = torch.randn(T, 256, generator=g).cuda(); hE0 = torch.randn(T, 48, 128, generator=g).cuda()
E_idx = ((torch.arange(T) // 256 * 256)[:, None] + torch.randint(0, 256, (T, 48), generator=g)).int().cuda()
outs = {}
for abl in (0, 32, 128):
    for rep in range(3):
        x = hE0.clone()
        assert lib.tmpnn_ablate_enc_edge(eng.w.handle, 0, _ptr(P), _ptr(x), _ptr(E_idx), T, abl, _stream()) == 0
    torch.cuda.synchronize()
    outs[abl] = x
print("max diff fp32 8-wave vs split-precision form:", float((outs[32] - outs[128]).abs().max()))
