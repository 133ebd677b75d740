"""debug: the three k-NN row forms, bit for bit — LDS rows (TMPNN_KNN_REG=0), register rows by extract-min rounds
(TMPNN_KNN_SEL=0) and register rows by threshold + compaction + bitonic sort (the default) — on lattice coordinates with exact
distance ties, duplicated and masked residues, L < K, L = 64 / 65 / 512 and ragged batches. The switches exist only in the debug
variant of the library (thermompnn_amd/libtmpnn_debug.so).  python tools/dbg_knn.py (GPU box)"""
import os, subprocess, sys
import numpy as np
DEBUG_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "thermompnn_amd", "libtmpnn_debug.so")
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
forms = {"lds": dict(TMPNN_KNN_REG="0"), "reg_min": dict(TMPNN_KNN_REG="1", TMPNN_KNN_SEL="0"), "reg_sel": dict(TMPNN_KNN_REG="1", TMPNN_KNN_SEL="1")}
for name, env in forms.items():
    subprocess.run([sys.executable, __file__, f"/tmp/knn_{name}.npz"], env=dict(os.environ, TMPNN_LIB=DEBUG_LIB, **env), check=True)
ref = np.load("/tmp/knn_lds.npz")
ok = True
for name in ("reg_min", "reg_sel"):
    b = np.load(f"/tmp/knn_{name}.npz")
    for k in ref.files:
        same = np.array_equal(ref[k].view(np.int32), b[k].view(np.int32))
        ok &= same
        print(name, k, ref[k].shape, "identical" if same else f"DIFFERENT at {np.argwhere(ref[k].view(np.int32) != b[k].view(np.int32))[:5].tolist()}")
print("ALL IDENTICAL" if ok else "MISMATCH")
