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
names = {0: "full", 1: "no global loads", 2: "no GELU", 4: "no LN/store", 7: "MFMA + LDS only", 8: "no MFMA"}
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
    print(f"abl={abl:2d} {name:20s} {ms:7.3f} ms  {2.0 * T * 48 * 3 * 128 * 128 / ms / 1e9:6.1f} TFLOP/s-equivalent")
