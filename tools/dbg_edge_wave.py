"""debug: the edge update's wavefront-per-block form (tmpnn_edge_wave.hip, large launches) against the 8-wavefront form, bit for bit, on
launches large enough to take it: 16 x L=256, a ragged batch with masked residues and proteins shorter than K (empty neighbour slots),
one L=2048 chain; f16x2. The switch (TMPNN_EDGE_WAVE_MIN=-1: never) lives in the debug variant of the library only.
python tools/dbg_edge_wave.py (GPU box)"""
import os, subprocess, sys
import numpy as np
DEBUG_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "thermompnn_amd", "libtmpnn_debug.so")
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from thermompnn_amd.engine import Engine
    from thermompnn_amd.synthetic import synthetic_backbone
    from thermompnn_amd.weights import synthetic_state_dict
    eng = Engine(synthetic_state_dict(0), torch.device("cuda:0"), retry_precision=None)
    out = {}
    rng = np.random.default_rng(12)
    cases = [[256] * 16, [int(x) for x in rng.integers(20, 400, size=24)] + [30, 5, 1], [2048, 700]]
    for case, lens in enumerate(cases):
        Xs, Ss = [], []
        for k, L in enumerate(lens):
            X, seq = synthetic_backbone(L, 500 + 40 * case + k)
            Xs.append(X.astype(np.float32))
            Ss.append(np.array(["ACDEFGHIKLMNPQRSTVWY".index(c) for c in seq], dtype=np.int32))
        X, S = np.concatenate(Xs), np.concatenate(Ss)
        T = len(S)
        mask = (rng.random(T) > (0.1 if case == 1 else 0.0)).astype(np.float32)
        X[mask == 0] = 0
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        ridx = np.concatenate([np.arange(L) for L in lens]).astype(np.int32)
        r = eng.ssm_forward(X, S, mask, ridx, np.ones(T, np.int32), off, want_hidden=True, want_log_probs=True, want_E_idx=True)
        for k, v in r.items():
            out[f"{k}{case}"] = v.cpu().numpy()
    np.savez(sys.argv[1], **out)
    sys.exit(0)
for v in ("-1", "6"):
    subprocess.run([sys.executable, __file__, f"/tmp/edge_wave{v}.npz"], env=dict(os.environ, TMPNN_EDGE_WAVE_MIN=v, TMPNN_LIB=DEBUG_LIB), check=True)
a, b = np.load("/tmp/edge_wave-1.npz"), np.load("/tmp/edge_wave6.npz")
ok = True
for k in a.files:
    same = np.array_equal(a[k].view(np.int32), b[k].view(np.int32))
    ok &= same
    d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
    print(k, a[k].shape, "identical" if same else f"DIFFERENT: max diff {np.nanmax(d):.3e}, n diff {int((d > 0).sum())}, max |a| {np.nanmax(np.abs(a[k])):.3e}, nan {int(np.isnan(b[k].astype(np.float64)).sum())}")
print("ALL IDENTICAL" if ok else "MISMATCH")
