"""Timing ablation of the encoder edge-update kernel on the GPU box (not a test, not the product):
    python tools/ablate.py
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermompnn_amd import _lib  # noqa: E402
from thermompnn_amd.engine import Engine, _ptr, _stream  # noqa: E402
from thermompnn_amd.weights import synthetic_state_dict  # noqa: E402

lib = _lib.load()
eng = Engine(synthetic_state_dict(0), "cuda:0")
T = 16384
g = torch.Generator().manual_seed(0)
P = torch.randn(T, 256, generator=g).cuda()
hE = torch.randn(T, 48, 128, generator=g).cuda()
base = (torch.arange(T) // 256 * 256)[:, None]
E_idx = (base + torch.randint(0, 256, (T, 48), generator=g)).int().cuda()
names = {0: "classic (4 waves)", 32: "8 waves", 128: "8 waves bf16x3", 7: "MFMA + LDS only", 8: "no MFMA"}
ref = None
hE_fixed = hE.clone()
for abl, name in names.items():
    for _ in range(2):
        lib.tmpnn_ablate_enc_edge(eng.w.handle, 0, _ptr(P), _ptr(hE), _ptr(E_idx), T, abl, _stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        rc = lib.tmpnn_ablate_enc_edge(eng.w.handle, 0, _ptr(P), _ptr(hE), _ptr(E_idx), T, abl, _stream())
        assert rc == 0, lib.tmpnn_last_error()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    if abl in (0, 32, 128):     # the three full variants must agree bit for bit on the same input
        x = hE_fixed.clone()
        lib.tmpnn_ablate_enc_edge(eng.w.handle, 0, _ptr(P), _ptr(x), _ptr(E_idx), T, abl, _stream())
        torch.cuda.synchronize()
        if ref is None:
            ref = x
        else:
            print("   bitwise equal to classic:", bool(torch.equal(ref, x)), float((ref - x).abs().max()))
    print(f"abl={abl:2d} {name:20s} {ms:7.3f} ms  {2.0 * T * 48 * 3 * 128 * 128 / ms / 1e9:6.1f} TFLOP/s-equivalent")
