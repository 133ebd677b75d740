"""Effective shader clock of the MI355X under a saturated fp32-MFMA load (tools only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermompnn_amd import _lib
from thermompnn_amd.engine import _ptr, _stream
lib = _lib.load()
torch.cuda.init()
sink = torch.zeros(256, device="cuda")
for blocks, iters in ((256, 2000), (256, 20000), (32, 20000), (1, 20000)):
    out = torch.zeros(2 * blocks, dtype=torch.int64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.tmpnn_clock_probe(blocks, iters, _ptr(out), _ptr(sink), _stream()) == 0
    e1.record(); torch.cuda.synchronize()
    o = out.cpu().view(-1, 2).double()
    cyc, ticks = o[:, 0].mean().item(), o[:, 1].mean().item()
    ghz = cyc / (ticks / 100e6) / 1e9
    per_mfma = cyc / (iters * 192)
    tf = blocks * 4 * iters * 192 * 2 * 16 * 16 * 4 / (e1.elapsed_time(e0) * -1e-3) / 1e12 if False else blocks * 4 * iters * 192 * 2048 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    print(f"blocks={blocks:4d} iters={iters:6d}: {ghz:.3f} GHz effective, {per_mfma:.2f} shader cycles per MFMA, {e0.elapsed_time(e1):.2f} ms, {tf:.1f} TFLOP/s")
