#!/usr/bin/env python3
"""Which op thins the margin of E / h_E0 / h_E_final in f16x2 mode? Featurizer outputs of the golden
structures against the float64 truth and the reference-equal fp32 oracle, for: the f16x2 featurizer (split GEMMs + packed
Gaussians), the fp32 featurizer inside f16x2 mode (debug library, TMPNN_FEAT_SPLIT=0) and the fp32 engine.
    python tools/margin_probe.py        (GPU box; spawns itself once per configuration)"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def worker(precision):
    import numpy as np
    from conftest import load_golden, oracle_trace_f64, weights_for_case
    from test_gpu_parity import align, oracle_trace, packed_inputs
    from thermompnn_amd.engine import Engine
    out = {}
    for case in ("2OCJ_A", "syn_L256_s1"):
        g = load_golden(case)
        eng = Engine(weights_for_case(g), "cuda:0", 48, precision=precision)
        p = packed_inputs(g)
        E_idx, D_nb = eng.knn_topk(p["X"], p["mask"], p["offsets"])
        ei = E_idx.cpu().numpy()
        Keff = min(48, p["L"])
        valid = np.nonzero(g["mask"] > 0)[0]
        tr, t64 = oracle_trace(weights_for_case(g), g, ei[:, :Keff]), oracle_trace_f64(g, ei[:, :Keff])
        h_E, E = eng.edge_featurize(p["X"], p["ridx"], p["cenc"], E_idx, D_nb, want_E=True)
        for name, got in (("E", E), ("h_E0", h_E)):
            a, b = align(got.cpu().numpy(), ei, tr[name], tr["E_idx"], valid)
            c, _ = align(t64[name], t64["E_idx"], tr[name], tr["E_idx"], valid)
            out[f"{case}/{name}"] = {"hip_vs_reference": float(np.abs(a - b).max()), "hip_vs_f64": float(np.abs(a - c).max()),
                                     "reference_vs_f64": float(np.abs(b - c).max())}
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker(sys.argv[1])
        sys.exit(0)
    dbg = os.path.join(REPO, "thermompnn_amd", "libtmpnn_debug.so")
    res = {}
    for name, prec, env in (("f16x2 (shipped: split GEMMs + packed Gaussians)", "f16x2", {}),
                            ("f16x2 engine, fp32 featurizer (debug lib, TMPNN_FEAT_SPLIT=0)", "f16x2", {"TMPNN_LIB": dbg, "TMPNN_FEAT_SPLIT": "0"}),
                            ("bf16x3 engine (fp32 featurizer)", "bf16x3", {}), ("fp32 engine", "fp32", {})):
        r = subprocess.run([sys.executable, __file__, prec], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        res[name] = json.loads(line[0][7:]) if line else {"error": r.stderr[-500:]}
    print(json.dumps(res, indent=1))
