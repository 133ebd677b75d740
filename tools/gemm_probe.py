"""fp32-MFMA vs split-precision (bf16x3 six-term, f16x2 three-term) GEMM cores: accuracy against float64 and speed (tools only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermompnn_amd import _lib
from thermompnn_amd.engine import _ptr, _stream
lib = _lib.load()
T = 4096
g = torch.Generator().manual_seed(0)
X = torch.randn(T, 48, 128, generator=g).cuda()
W = (torch.rand(128, 128, generator=g) * 0.3 - 0.15).cuda()
ref = (X.double() @ W.double().t())
for mode, name in ((0, "fp32 16x16x4"), (1, "bf16x3 six-term"), (2, "f16x2 three-term")):
    Y = torch.zeros_like(X)
    assert lib.tmpnn_gemm_probe(mode, _ptr(X), _ptr(W), _ptr(Y), T, 1, _stream()) == 0, lib.tmpnn_last_error()
    torch.cuda.synchronize()
    err = (Y.double() - ref).abs().max().item()
    reps = 64
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.tmpnn_gemm_probe(mode, _ptr(X), _ptr(W), _ptr(Y), T, reps, _stream())
    e0.record()
    lib.tmpnn_gemm_probe(mode, _ptr(X), _ptr(W), _ptr(Y), T, reps, _stream())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    tf = 2.0 * T * 48 * 128 * 128 * reps / (ms * 1e-3) / 1e12
    print(f"{name:18s} max |err| vs fp64 = {err:.2e}   {ms:7.3f} ms for {reps} GEMMs/tile  ->  {tf:6.1f} fp32-equivalent TFLOP/s")

# small-magnitude inputs: does the fp16 path keep values below the fp16 normal range (6.1e-5)?
for scale in (1e-3, 1e-5, 1e-7):
    Xs = (X[:64] * scale).contiguous()
    refs = Xs.double() @ W.double().t()
    out = []
    for mode in (0, 1, 2):
        Y = torch.zeros_like(Xs)
        lib.tmpnn_gemm_probe(mode, _ptr(Xs), _ptr(W), _ptr(Y), 64, 1, _stream()); torch.cuda.synchronize()
        out.append(((Y.double() - refs).abs().max() / refs.abs().max()).item())
    print(f"input scale {scale:g}: max err / max|ref|  fp32 {out[0]:.2e}  bf16x3 {out[1]:.2e}  f16x2 {out[2]:.2e}")
