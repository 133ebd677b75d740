"""debug: register-resident k-NN rows (TMPNN_KNN_REG=1, default) against the LDS form (=0), bit for bit.  python tools/dbg_knn.py (GPU box)"""
import os, subprocess, sys
import numpy as np
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, ".")
    from thermompnn_amd.engine import Engine
    from thermompnn_amd.weights import synthetic_state_dict
    eng = Engine(synthetic_state_dict(0), torch.device("cuda:0"))
    rng = np.random.default_rng(5)
    out = {}
    for case, lens in enumerate([[32], [100, 17, 256], [300, 512, 1, 64, 65], [256] * 8, [48, 49, 47]]):
        T = sum(lens)
        X = rng.normal(size=(T, 4, 3)).astype(np.float32) * 8.0
        X[:, 1] = np.round(X[:, 1] * 2) / 2             # Ca on a 0.5 A lattice: many EXACT distance ties
        if T > 40:
            X[7] = X[3]                                  # duplicate residues
            X[11] = X[3]
        mask = (rng.random(T) > 0.1).astype(np.float32)  # ~10 % masked residues
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        E, D = eng.knn_topk(torch.tensor(X).cuda(), torch.tensor(mask).cuda(), torch.tensor(off).cuda())
        out[f"E{case}"], out[f"D{case}"] = E.cpu().numpy(), D.cpu().numpy()
    np.savez(sys.argv[1], **out)
    sys.exit(0)
for v in ("0", "1"):
    subprocess.run([sys.executable, __file__, f"/tmp/knn{v}.npz"], env=dict(os.environ, TMPNN_KNN_REG=v), check=True)
a, b = np.load("/tmp/knn0.npz"), np.load("/tmp/knn1.npz")
ok = True
for k in a.files:
    same = np.array_equal(a[k].view(np.int32), b[k].view(np.int32))
    ok &= same
    print(k, a[k].shape, "identical" if same else f"DIFFERENT at {np.argwhere(a[k].view(np.int32) != b[k].view(np.int32))[:5].tolist()}")
print("ALL IDENTICAL" if ok else "MISMATCH")
