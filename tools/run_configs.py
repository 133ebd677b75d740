"""BASELINE.json configs 2-5 on one MI355X (measurement script; writes one JSON document to stdout).
    python tools/run_configs.py > gpurun_out/configs.json
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import fetch_profile, kernel_bytes, kernel_flops, kernel_roofs  # noqa: E402
from thermompnn_amd import _lib  # noqa: E402
from thermompnn_amd.engine import Engine  # noqa: E402
from thermompnn_amd.synthetic import synthetic_backbone  # noqa: E402
from thermompnn_amd.weights import synthetic_state_dict  # noqa: E402

AA20 = "ACDEFGHIKLMNPQRSTVWY"
dev = torch.device("cuda:0")
lib = _lib.load()
eng = Engine(synthetic_state_dict(0), dev, 48)


def pack(lengths, seeds):
    xs, ss, ri = [], [], []
    for L, sd in zip(lengths, seeds):
        X, seq = synthetic_backbone(int(L), int(sd))
        xs.append(X); ss.append([AA20.index(c) for c in seq]); ri.append(np.arange(L))
    T = int(sum(lengths))
    t = lambda a, dt: torch.tensor(np.concatenate(a), dtype=dt, device=dev)
    return dict(X=t(xs, torch.float32), S=t(ss, torch.int32), mask=torch.ones(T, device=dev), ridx=t(ri, torch.int32),
                cenc=torch.ones(T, dtype=torch.int32, device=dev),
                offsets=torch.tensor(np.concatenate([[0], np.cumsum(lengths)]), dtype=torch.int32, device=dev),
                T=T, max_len=int(max(lengths)))


def run(b, steps, warmup=2, post=None):
    out = {"ddg": torch.empty((b["T"], 21), device=dev)}
    f = lambda: eng.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=b["max_len"], out=out, check_status=False)
    for _ in range(warmup):
        f()
        if post: post(out["ddg"])
    torch.cuda.synchronize()
    lib.tmpnn_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        f()
        if post: post(out["ddg"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = fetch_profile(lib)
    lib.tmpnn_profile_enable(0)
    return dt, {k: ms / n for k, (ms, n) in prof.items()}


res = {}
# config 2: one L=256 protein (latency)
b = pack([256], [0])
dt, k = run(b, 200, 10)
res["engine_precision"] = eng.precision
res["config2_single_L256"] = {"ms": dt * 1e3, "preds_per_s": 5120 / dt, "kernel_avg_ms": k, "gpu_kernel_ms": sum(
    v * {"enc_msg": 3, "dec_msg": 3, "enc_edge": 3, "node_update": 6}.get(n, 1) for n, v in k.items())}
# config 3: 1024 ragged proteins, L ~ U[64, 512]
lens = np.random.default_rng(1).integers(64, 513, size=1024)
t0 = time.perf_counter(); b = pack(lens, 1000 + np.arange(1024)); gen_s = time.perf_counter() - t0
dt, k = run(b, 3, 1)
edges = int(sum(L * min(48, L) for L in lens))
res["config3_ragged_1024"] = {"proteins": 1024, "residues": b["T"], "preds": 20 * b["T"], "ms": dt * 1e3, "preds_per_s": 20 * b["T"] / dt,
                              "host_generation_s": gen_s, "kernel_avg_ms": k,
                              "enc_edge_tflops": kernel_flops("enc_edge", b["T"], edges) / (k["enc_edge"] * 1e-3) / 1e12,
                              "workspace_GB": lib.tmpnn_workspace_bytes(b["T"]) / 1e9}
# config 4: 300 Megascale-like proteins (K_eff = min(48, L) < 48 for many), 200k explicit mutations
rng = np.random.default_rng(2)
lens = rng.integers(40, 73, size=300)
b = pack(lens, 5000 + np.arange(300))
flat = rng.choice(20 * b["T"], size=200000, replace=False)            # (residue, aa) pairs without replacement
sel = torch.tensor(flat // 20 * 21 + flat % 20, device=dev)          # what dist.select_mutations computes from (protein, pos, aa)
picked = {}
def post(ddg):
    picked["v"] = ddg.view(-1)[sel]
dt, k = run(b, 20, 3, post)
res["config4_megascale_like"] = {"proteins": 300, "residues": b["T"], "mutations": 200000, "ms": dt * 1e3,
                                 "listed_preds_per_s": 200000 / dt, "table_preds_per_s": 20 * b["T"] / dt, "kernel_avg_ms": k,
                                 "note": "single GPU; the 8-GPU run adds one all-gather of 0.8 MB (dist.ssm_scan)"}
# config 5: one L=2048 chain
b = pack([2048], [3])
dt, k = run(b, 50, 5)
res["config5_L2048"] = {"ms": dt * 1e3, "preds_per_s": 40960 / dt, "kernel_avg_ms": k,
                        # per kernel: algorithmic HBM bytes / launch time, and the fraction of its binding roof (bench.kernel_roofs)
                        "kernel_hbm_GBps": {n: kernel_bytes(n, 2048, 2048 * 48) / (v * 1e-3) / 1e9 for n, v in k.items() if kernel_bytes(n, 2048, 2048 * 48)},
                        "kernel_roofs": {n: {kk: r[kk] for kk in ("bound", "frac", "t_hbm_us", "t_mfma_us")}
                                         for n, v in k.items() if kernel_bytes(n, 2048, 2048 * 48)
                                         for r in [kernel_roofs(n, 2048, 2048 * 48, v, eng.precision)]},
                        "note": "one 2048-residue chain = 2048 tiles on 256 CUs: 8 tiles per persistent workgroup, so prologues (weight fragments) "
                                "and the tail weigh more than in the 64-protein batch; k-NN rows > 512 use the LDS form",
                        # default (f16x2) kernels, from hipcc -Rpass-analysis=kernel-resource-usage
                        "lds_bytes_per_workgroup": {"knn (4 rows, dynamic)": 4 * (2048 + 33) * 4, "featurize_split": 146368, "msg8_rp": 49920, "msg8_wave (launches of >= 16 residues per CU)": 135680,
                                                    "enc_edge8_rp": 52992, "node_update8 (64 rows)": 126976, "head8": 98432},
                        "waves_per_simd": {"knn": 8, "featurize_split": 2, "msg8_rp": 2, "enc_edge8_rp": 2, "node_update8": 2, "head8": 2}}
print(json.dumps(res, indent=1))
