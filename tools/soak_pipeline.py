import os, sys, time, resource, tempfile, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from thermompnn_amd import ssm_scan
from thermompnn_amd.engine import Engine
from thermompnn_amd.synthetic import synthetic_backbone, backbone_pdb_text
from thermompnn_amd.weights import synthetic_state_dict
d = tempfile.mkdtemp(dir="/dev/shm")
rng = np.random.default_rng(4)
paths = []
for i, L in enumerate(rng.integers(30, 300, size=200)):
    X, seq = synthetic_backbone(int(L), 77 + i)
    p = os.path.join(d, f"s{i}.pdb"); open(p, "w").write(backbone_pdb_text(X, seq)); paths.append(p)
eng = Engine(synthetic_state_dict(0), "cuda:0", 48)
ref = None
t0 = time.time()
for it in range(40):
    out = os.path.join(d, "o.npz" if it % 2 else "o.csv")
    n, st = ssm_scan.scan_to_file(eng, paths, ["A"] * len(paths), out, chunk_files=int(rng.integers(3, 60)), centrality=bool(it % 3 == 0))
    if out.endswith(".npz"):
        z = np.load(out)["ddg"]
        if ref is None: ref = z
        assert np.array_equal(ref, z), "results changed between scans"
    if it % 10 == 9:
        print(it + 1, "scans", round(time.time() - t0, 1), "s  maxrss MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024, "cuda MB", torch.cuda.memory_allocated() // 2**20, "threads", len(__import__("threading").enumerate()))
print("soak ok")
