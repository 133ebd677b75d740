"""debug: the fused small-launch forms (k-NN inside the featurizer launch; edge update of layer l + message pass of the next layer in
ONE launch, edge_msg_fused_kernel) against the separate launches, bit for bit — neighbour lists included: a single L=256 protein, L=64, L=30 (< K: empty neighbour slots), a ragged batch of
three with masked residues, in f16x2. The switch lives in the debug variant of the library only.  python tools/dbg_fused.py (GPU box)"""
import os, subprocess, sys
import numpy as np
DEBUG_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "thermompnn_amd", "libtmpnn_debug.so")
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from thermompnn_amd.engine import Engine
    from thermompnn_amd.synthetic import synthetic_backbone
    from thermompnn_amd.weights import synthetic_state_dict
    eng = Engine(synthetic_state_dict(0), torch.device("cuda:0"))
    out = {}
    rng = np.random.default_rng(11)
    for case, lens in enumerate([[256], [64], [30], [100, 17, 90], [5, 1]]):
        Xs, Ss = [], []
        for k, L in enumerate(lens):
            X, seq = synthetic_backbone(L, 300 + 10 * case + k)
            Xs.append(X.astype(np.float32))
            Ss.append(np.array(["ACDEFGHIKLMNPQRSTVWY".index(c) for c in seq], dtype=np.int32))
        X, S = np.concatenate(Xs), np.concatenate(Ss)
        T = len(S)
        mask = (rng.random(T) > (0.1 if case == 3 else 0.0)).astype(np.float32)
        X[mask == 0] = 0
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        ridx = np.concatenate([np.arange(L) for L in lens]).astype(np.int32)
        r = eng.ssm_forward(X, S, mask, ridx, np.ones(T, np.int32), off, want_hidden=True, want_log_probs=True, want_E_idx=True)
        for k, v in r.items():
            out[f"{k}{case}"] = v.cpu().numpy()
    np.savez(sys.argv[1], **out)
    sys.exit(0)
for v in ("0", "1"):
    subprocess.run([sys.executable, __file__, f"/tmp/fused{v}.npz"], env=dict(os.environ, TMPNN_FUSE_SMALL=v, TMPNN_LIB=DEBUG_LIB), check=True)
a, b = np.load("/tmp/fused0.npz"), np.load("/tmp/fused1.npz")
ok = True
for k in a.files:
    same = np.array_equal(a[k].view(np.int32), b[k].view(np.int32))
    ok &= same
    d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
    print(k, a[k].shape, "identical" if same else f"DIFFERENT: max diff {np.nanmax(d):.3e}, n diff {int((d > 0).sum())}")
print("ALL IDENTICAL" if ok else "MISMATCH")
