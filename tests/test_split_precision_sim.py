"""CPU emulation of the split-precision GEMM arithmetic (thermompnn_amd/csrc/tmpnn_split.h) pinned to the reference goldens.

Every Linear of the restructured schedule (tests/schedule_model.py) is replaced by the f16x2 three-term product
    x = h + l,  h = fp16(x),  l = fp16(x - h)  (unscaled: an fp16 subnormal for small x);   y = h h + h l + l h      (fp32 accumulation)
and the result must stay inside the parity tolerances against vectors produced by the reference itself. This is the
CPU-side evidence that the default matrix-core mode of the HIP engine is parity-neutral (the GPU tests check the kernels).
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from conftest import load_golden, tol_scale, weights_for_case
from oracle import thermompnn_oracle as orc
from thermompnn_amd.weights import synthetic_state_dict

HERE = os.path.dirname(os.path.abspath(__file__))


def _split2(x):
    h = x.to(torch.float16).float()
    return h, (x - h).to(torch.float16).float()          # torch keeps fp16 subnormals, like v_cvt_pk_f16_f32 and the MFMA


def _linear_f16x2(x, w, b=None):
    xh, xl = _split2(x)
    wh, wl = _split2(w)
    y = xh @ wh.t() + (xh @ wl.t() + xl @ wh.t())
    return y if b is None else y + b


def _linear_bf16_3term(x, w, b=None):
    def s3(v):
        h = v.to(torch.bfloat16).float()
        return h, (v - h).to(torch.bfloat16).float()
    xh, xm = s3(x)
    wh, wm = s3(w)
    y = xh @ wh.t() + (xh @ wm.t() + xm @ wh.t())
    return y if b is None else y + b


def _kernel_gelu_coefficients():
    """The exponent polynomial of the shipped GELU (tmpnn_common.h: gelu2), highest power first, parsed from
    the kernel source so that the emulation below cannot drift from what the GPU runs."""
    import re
    src = open(os.path.join(os.path.dirname(HERE), "thermompnn_amd", "csrc", "tmpnn_common.h")).read()
    a = src.index("__device__ __forceinline__ f2 gelu2(f2 x) {")
    blk = src[src.index("f2 q = ", a):src.index("const f2 e = ", a) + 80]
    c = [float(x) for x in re.findall(r"(-?\d\.\d+e[+-]\d+)f", blk)]
    c = [c[0], c[2]] + c[4:]                 # the first fma lists each of its two constants twice ({c, c})
    assert len(c) == 7, c
    return c


def _gelu_kernel_form(x):
    """gelu2 as the f16x2 kernels run it (TM_GELU_NAN3: tmpnn_edge / _msg / _node.hip), in emulated fp32: t = min(|x|, 4 sqrt2); 6 Horner fmas;
    exp2; max(x, 0) - t h with the CLAMPED t (every step rounded to fp32)."""
    c = _kernel_gelu_coefficients()
    ax = x.abs()
    t = torch.clamp(ax, max=float(np.float32(5.656854249))).double()
    r = torch.full_like(t, float(np.float32(c[0])))
    for k in c[1:]:
        r = (r * t + float(np.float32(k))).float().double()
    h = torch.exp2(r).float().double()
    return (-t * h + torch.clamp(x, min=0).double()).float()


def _run(case, linear, gelu=None):
    g = load_golden(case)
    t = torch.from_numpy
    W = weights_for_case(g)
    mp, hd = orc.split_weights(W)
    S = t(g["S"].astype(np.int64))
    # the schedule model writes some per-edge GEMMs as `a @ W.t()`: route them through F.linear as well
    src = open(os.path.join(HERE, "schedule_model.py")).read()
    for a, b in (("h_E @ W1[:, 128:256].t()", "F.linear(h_E, W1[:, 128:256])"),
                 ("h_E @ W11[:, 128:256].t()", "F.linear(h_E, W11[:, 128:256])"),
                 ("rbf @ We[:, 16:].t()", "F.linear(rbf, We[:, 16:])")):
        assert a in src
        src = src.replace(a, b)
    ns = {}
    exec(compile(src, "schedule_model_split", "exec"), ns)
    orig, orig_gelu = F.linear, F.gelu
    F.linear = linear
    if gelu is not None:
        F.gelu = gelu
    try:
        with torch.no_grad():
            hs, _, _ = ns["mpnn_schedule"](mp, t(g["X"]), t(g["mask"]), S, t(g["residue_idx"].astype(np.int64)),
                                           t(g["chain_enc"].astype(np.int64)))
    finally:
        F.linear, F.gelu = orig, orig_gelu
    with torch.no_grad():
        _, ddg = orc.head_table(hd, [hs[2][None], hs[1][None]], mp["W_s.weight"][S][None], S[None])
    have = ~np.isnan(g["ddg"][:, 0])
    return (float(np.abs(hs[2].numpy() - g["hV_dec3"]).max()),
            float(np.abs(ddg[0].numpy()[have][:, :20] - g["ddg"][have]).max()) / tol_scale(g, g["ddg"]))


def test_f16x2_three_term_split_is_parity_neutral():
    h_err, d_err = _run("2OCJ_A", _linear_f16x2)
    assert h_err < 1e-5 and d_err < 1e-4, (h_err, d_err)


def test_a_three_term_bf16_split_would_not_be():
    """Negative control: the emulation does discriminate — hh + hm + mh in bf16 misses the hidden-state tolerance."""
    h_err, _ = _run("2OCJ_A", _linear_bf16_3term)
    assert h_err > 1e-5, h_err


def test_kernel_gelu_form_is_at_the_fp32_floor():
    """The degree-6 exponent polynomial of gelu2 (fitted to the error of gelu itself, tools/fit_gelu.py) against the exact-erf
    GELU in float64: <= 3e-7 everywhere, i.e. the rounding floor of x Phi(x) in fp32 (2.4e-7 at |x| = 4)."""
    x = torch.cat([torch.linspace(-8, 8, 400001), torch.randn(200000, generator=torch.Generator().manual_seed(0)) * 1.5,
                   torch.linspace(-300, 300, 60001)]).float()          # far beyond the clamp too (hot-draw Linear outputs reach 2e2)
    want = 0.5 * x.double() * (1.0 + torch.erf(x.double() / np.sqrt(2.0)))
    assert float((_gelu_kernel_form(x).double() - want).abs().max()) <= 3.0e-7


def test_f16x2_with_the_kernel_gelu_stays_parity_neutral_on_every_weight_set():
    """Split-precision GEMMs AND the polynomial GELU together, on the Xavier draws and the hot draw: still inside the
    tolerances against the reference's vectors (a degree-4 exponent, 1.5e-6, measured 1.2e-5 on the hidden states: not neutral)."""
    for case in ("2OCJ_A", "2OCJ_A_w1", "2OCJ_A_hot"):
        h_err, d_ratio = _run(case, _linear_f16x2, _gelu_kernel_form)
        assert h_err < 1e-5 and d_ratio < 1e-4, (case, h_err, d_ratio)
