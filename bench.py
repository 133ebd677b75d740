#!/usr/bin/env python3
"""ThermoMPNN SSM hot-path benchmark on MI355X (contract: python bench.py --gpus N --steps K --warmup W).

A "step" = one pass of the fused HIP path (tmpnn_ssm_forward: kNN -> featurizer -> 3 enc -> 3 dec -> ddG head)
over one batch of B synthetic proteins of L=256 residues (K=48, h=128; BASELINE.json configs[1] replicated B
times so the 256-CU chip is filled) that is already resident in HBM, producing B*L*20 mutant ddG predictions.
With N > 1 every rank owns its own B proteins (weak scaling, proteins are independent) and each step ends with
the path's one exchange step: an RCCL all-gather of the per-rank ddG tables.

`python bench.py --gpus N` with N > 1 and no launcher in the environment starts itself under
`python -m torch.distributed.run --nproc-per-node N` (one process per GPU, RCCL); the line then carries a `collective`
block proving what the process group saw (`ranks_seen` = all_reduce of ones, one device id per rank).
`--scaling strong` runs BASELINE.json configs[3] instead (300 Megascale-like proteins / 200 000 listed mutants, FIXED total
work sharded over the ranks by LPT, timed with and without the all-gather of the ddG tables).

Prints ONE JSON line on rank 0: whole-job preds/s + `roofline` (dominant kernel, HIP-event timed inside the
timed region, BOTH roofs: algorithmic HBM bytes / 8 TB/s and executed MFMA flops / peak, `bound` = the binding one) +
`cpu_baseline` (the CPU oracle on the host cores, bounded sample, rank 0 at N=1 only; `saturated` = P processes x 8
threads filling the host).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from thermompnn_amd import _lib  # noqa: E402
from thermompnn_amd.engine import Engine  # noqa: E402
from thermompnn_amd.synthetic import synthetic_backbone  # noqa: E402
from thermompnn_amd.weights import synthetic_state_dict  # noqa: E402

AA20 = "ACDEFGHIKLMNPQRSTVWY"
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 / f16 MFMA peak
SPLIT_TERMS = {"f16x2": 3, "bf16x3": 6}   # 16-bit MFMAs per fp32-class multiply-accumulate (tmpnn_split.h)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E spec peak (6.29 TB/s measured copy)


def kernel_flops(name, T, edges):
    """Executed (algorithmic, minimal-schedule) flops of ONE launch of each kernel; DESIGN.md §4."""
    H2 = 128 * 128
    return {
        "featurize": 2.0 * edges * (400 * 128 + H2),
        "enc_msg": 2.0 * edges * 2 * H2,
        "dec_msg": 2.0 * edges * 2 * H2,
        "enc_edge": 2.0 * edges * 3 * H2,
        "node_update": 2.0 * T * (H2 + 2 * 512 * 128),
        "node_proj": 2.0 * T * 256 * 128,
        "head": 2.0 * T * (384 * 384 + 384 * 64 + 64 * 32 + 32 * 21),
    }.get(name)


def kernel_bytes(name, T, edges):
    """ALGORITHMIC HBM bytes of ONE launch (E_b = 512 B per edge per pass over h_E; node-sized operands once):
    what a launch must move if every operand crosses HBM exactly once. DESIGN.md §4 table."""
    E_b = 512.0 * edges
    node = 512.0 * T                                   # one [T,128] fp32 state
    return {
        "featurize": E_b + T * (48 + 8 + 2 * 192),                      # writes h_E; reads X, ridx/chain, E_idx + D_nb
        "enc_msg": E_b + 2 * node + T * 192 + node + 4.0 * T,           # reads h_E, P [T,256], E_idx; writes Ssum, cnt
        "dec_msg": E_b + 2 * node + T * 192 + node + 4.0 * T,
        "enc_edge": 2 * E_b + 2 * node + T * 192,                       # reads + rewrites h_E in place; reads P, E_idx
        "node_update": 5 * node + 8.0 * T,                              # Ssum, h_V in; h_V, P [T,256] out (+ cnt, mask)
        "node_proj": 3 * node,
        "head": 3 * node + 84.0 * T,                                    # two decoder states + S in; [T,21] out
        "knn": T * (48 + 4) + T * 2 * 192 + 3 * node,                   # X, mask in; E_idx, D_nb (+ zero state, first projection) out
    }.get(name)


def kernel_roofs(name, T, edges, avg_ms, mode):
    """Both roofs of one kernel launch: t_hbm = algorithmic bytes / 8 TB/s, t_mfma = EXECUTED matrix-core flops / the dense
    peak of the instruction that runs them (f16x2: 3 f16 MFMAs per multiply-accumulate at 2.5 PF; bf16x3: 6; fp32: the
    157.3 TF fp32 matrix pipe). The binding roof is the larger time; frac = that time / the measured launch time."""
    fl, by = kernel_flops(name, T, edges), kernel_bytes(name, T, edges)
    per_edge = ("enc_edge", "enc_msg", "dec_msg")
    if mode == "f16x2" and name in per_edge + ("featurize", "node_update", "head"):
        terms = 3
    elif mode == "bf16x3" and name in per_edge:          # node / head / featurizer run the fp32 MFMA kernels in this mode
        terms = 6
    else:
        terms = 0
    peak_tf = BF16_MFMA_PEAK_TFLOPS / terms if terms else FP32_MFMA_PEAK_TFLOPS
    t_mfma = fl / (peak_tf * 1e12) if fl else 0.0
    t_hbm = by / (HBM_PEAK_GBS * 1e9) if by else 0.0
    t = avg_ms * 1e-3
    bound = "hbm" if t_hbm >= t_mfma else "mfma"
    return {"bound": bound, "t_hbm_us": t_hbm * 1e6, "t_mfma_us": t_mfma * 1e6, "frac": max(t_hbm, t_mfma) / t if t > 0 else None,
            "hbm": {"bytes_per_launch": by, "achieved_GBps": by / t / 1e9 if by else None, "peak_GBps": HBM_PEAK_GBS,
                    "frac": t_hbm / t if t > 0 else None},
            "mfma": {"flops_per_launch": fl, "terms": terms, "achieved_TFLOPs": fl / t / 1e12 if fl else None, "peak_TFLOPs": peak_tf,
                     "frac": t_mfma / t if t > 0 else None,
                     "executed": ({"dtype": "f16" if mode == "f16x2" else "bf16", "achieved_TFLOPs": terms * fl / t / 1e12,
                                   "peak_TFLOPs": BF16_MFMA_PEAK_TFLOPS} if terms and fl else None)}}


def kernel_source_stamp():
    """Hash of the DEVICE sources (csrc/*.hip + the headers they include): a PMC or ISA-count file measured on other kernels must
    not be quoted for these. The host-only translation units (tmpnn_pdb.cpp, tmpnn_csv.cpp) are not part of it (round 4: an edit of
    the CSV writer must not void the kernels' evidence)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(REPO, "thermompnn_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, T):
    """HBM bytes per launch from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes committed under profiles/ (collected
    separately by tools/pmc_traffic.sh at T = 16384, gfx950 FETCH_SIZE correction applied). Every file carries the hash of
    the kernel sources it was measured on; only a file measured on THESE sources is quoted (newest round first) — a stale
    file or another batch size yields None rather than an old number."""
    if T != 16384:
        return None
    import glob
    stamp = None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            stamp = stamp or kernel_source_stamp()
            if d.get("kernel_source_stamp") == stamp:
                return d["kernels"].get(kernel, {}).get("traffic_bytes")
        except Exception:
            continue
    return None


def live_pmc_traffic(patterns, timeout_s=90):
    """HBM bytes per launch measured INSIDE this run (round 6; the round-5 review: the committed file is the builder's claim): two
    child processes `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, kernel trace only — the
    combination MI355X_MICROARCH.md prescribes) over tools/pmc_workload.py: three forwards of the bench batch (T = 16384) and a
    402.7 MB device copy as the calibration. FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads: the copy must come out
    at 2 x 402.7 MB), WRITE_SIZE is exact, unit KB = 1024 B — the arithmetic of tools/pmc_traffic.sh. patterns = {name: substring of the
    kernel symbol}. Returns ({name: bytes per launch}, info) or (None, why). Never raises: the bench line must not depend on it."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    t0 = time.perf_counter()
    try:
        exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
        if exe is None:
            return None, "rocprofv3 not found"
        if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
            return None, "this process is itself running under a profiler"
        per = {}
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            env = dict(os.environ, TMPDIR="/tmp", TMPNN_BENCH_PMC_CHILD="1")
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                out = os.path.join(tmp, c)
                r = subprocess.run([exe, "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", out, "-o", c, "--", sys.executable,
                                    os.path.join(REPO, "tools", "pmc_workload.py")], cwd="/tmp", env=env, capture_output=True, text=True,
                                   timeout=timeout_s)
                acc = {}
                for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if row["Counter_Name"] == c:
                            acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
                if not acc:
                    return None, f"{c} pass produced no counters (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"
                for name, pat in patterns.items():
                    v = [x for k, xs in acc.items() if pat in k for x in xs]
                    if pat == "copyBuffer":
                        v = sorted(v)[-3:]                 # the runtime's copy kernel also moves the weights: the three 402.7 MB copies
                    if v:
                        per.setdefault(name, {})[c] = (sum(v) / len(v), len(v))
        res = {}
        for name, d in per.items():
            if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                res[name] = d["FETCH_SIZE"][0] * 1024 * 2 + d["WRITE_SIZE"][0] * 1024
        cal = res.get("device_copy_calibration")
        info = {"source": "live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes over tools/pmc_workload.py inside this run",
                "seconds": time.perf_counter() - t0, "launches_averaged": {n: d["FETCH_SIZE"][1] for n, d in per.items() if "FETCH_SIZE" in d},
                "calibration_copy_ratio": cal / (2 * 16384 * 48 * 128 * 4) if cal else None}
        if not cal or not 0.95 < info["calibration_copy_ratio"] < 1.05:
            return None, f"calibration copy off: {info['calibration_copy_ratio']}"
        return res, info
    except Exception as e:                                  # noqa: BLE001 — a timeout, a parse error, a missing column: fall back to the file
        return None, f"{type(e).__name__}: {e}"


# Issue cost of one wavefront instruction as ONE SIMD sees it with its two wavefronts issuing dense, independent streams
# (tools/probe/valu_cost_probe.hip, round 5, cycles): plain two-operand VALU 2.5, v_fma_f32 2.7, everything VOP3 / converting / DPP
# 3.6-3.7, packed fp32 3.76, transcendentals / v_permlane*_swap / v_fma_mix* 6.6. (One wavefront alone cannot issue faster than one
# VALU instruction per 5.0-5.6 cycles whatever the class.) MFMA: v_mfma_f32_16x16x32_{f16,bf16} 16 cycles of the SIMD's matrix pipe
# (16.3 measured, tools/probe/hetero_probe.hip), v_mfma_f32_16x16x4_f32 32.
VALU_COST = {"v_add_f32": 2.5, "v_sub_f32": 2.5, "v_mul_f32": 2.5, "v_mov_b32": 2.5, "v_fma_f32": 2.7, "v_fmac_f32": 2.7, "v_fmamk_f32": 2.7,
             "v_fmaak_f32": 2.7, "v_permlane16_swap_b32": 6.6, "v_permlane32_swap_b32": 6.6, "v_fma_mixlo_f16": 6.6, "v_fma_mixhi_f16": 6.6}
VALU_COST_DEFAULT, VALU_COST_PACKED, VALU_COST_TRANS = 3.6, 3.76, 6.6
# what the pure-VALU wavefront of tools/probe/hetero_probe.hip keeps of its rate beside a saturated matrix pipe (4.4 -> 7.5 cycles
# per v_fma_f32, 5.3 -> 7.7 per v_pk_fma_f32: 59-69 %)
HETERO_VALU_RATE = 0.64


def valu_cost(op):
    base = op[:-4] if op.endswith(("_e32", "_e64")) else op
    if base in VALU_COST:
        return VALU_COST[base]
    if base.startswith("v_pk_"):
        return VALU_COST_PACKED
    if base.startswith(("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")):
        return VALU_COST_TRANS
    return VALU_COST_DEFAULT


ISA_SYMBOLS = {   # bench kernel name -> the instantiation the bench batch launches (f16x2, images, 32-bit gather offsets)
    "enc_edge": "enc_edge8_rp_kernelI7SplitH2Lb0ELb1E", "enc_msg": "msg8_wave_kernelILb0ELb1ELb0E",
    "dec_msg": "msg8_wave_kernelILb1ELb1ELb0E"}
# wavefront-trips of the counted loop per tile (= one residue's 48 x 128 edge block): the 8-wavefront kernels run ONE trip in each of a
# workgroup's 8 wavefronts, the wavefront-per-residue message kernel THREE trips (16-row blocks) in one wavefront
ISA_WAVE_TRIPS = {"enc_edge": 8, "enc_msg": 3, "dec_msg": 3}
# (not the featurizer: its tile loop contains run-time loops — 9.4 trips of the Gaussian loop per tile — so static counts of one
#  trip understate what a wavefront issues; the per-edge kernels' tile loops are straight-line code)


def mfma_cycles(op):
    return 32.0 if op.endswith("_f32") and "x4_" in op else 16.0 if "16x16x32" in op else 32.0 if "32x32" in op else 16.0


def isa_counts(kernel):
    """Static per-trip instruction counts of the kernel's tile loop (tools/isa_counts.py -> profiles/r*_isa_counts.json, stamped
    with the kernel-source hash like the PMC files: counts of other sources are not quoted)."""
    import glob
    sym = ISA_SYMBOLS.get(kernel)
    if sym is None:
        return None
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_isa_counts.json")), reverse=True):
        try:
            d = json.load(open(f))
            if d.get("source_stamp") != kernel_source_stamp():
                continue
            for name, e in d["kernels"].items():
                if sym in name and "tile_loop" in e:
                    return dict(e["tile_loop"], file=os.path.basename(f), symbol=name, vgprs=e.get("next_free_vgpr"), lds_bytes=e.get("lds_bytes"),
                                valu_ops=e.get("tile_loop_valu_ops", {}))
        except Exception:
            continue
    return None


def issue_roof(kernel, tiles, n_cus, clock_ghz, avg_ms):
    """The THIRD roof: instruction issue, as a BRACKET. One tile (= one residue's 48 x 128 edge block) is one
    trip of the persistent loop in each of the workgroup's 8 wavefronts, 2 per SIMD. Per SIMD and tile the two wavefronts issue
    t_valu = 2 x sum(count x cost of the opcode) cycles of VALU work and t_mfma = 2 x sum(MFMA x 16) cycles of matrix-pipe work
    (static counts of the shipped code object, costs measured by tools/probe/valu_cost_probe.hip / hetero_probe.hip). The two pipes
    are separate, so the bound lies between
      t_overlap = max(t_valu, t_mfma)   (the streams of the two wavefronts overlap perfectly) and
      t_serial  = t_valu + t_mfma       (they never do — what lock-step GEMM / epilogue phases between barriers amount to),
    with t_hetero = t_mfma + max(0, t_valu - 0.64 t_mfma) in between: the overlap the hetero probe measured for a pure MFMA stream
    beside a pure VALU stream (the VALU stream keeps 59-69 % of its rate). None of them contains barriers, LDS / memory latency,
    s_waitcnt or s_nop. `frac_*` = that time / the measured launch."""
    c = isa_counts(kernel)
    if c is None or not clock_ghz:
        return None
    mf = {k[5:]: v for k, v in c.items() if k.startswith("mfma:")}
    cyc_mfma = sum(mfma_cycles(op) * n for op, n in mf.items())
    ops = c.get("valu_ops") or {}
    n_valu = c.get("valu", 0) + c.get("valu_packed", 0) + c.get("valu_trans", 0)
    if ops and sum(ops.values()) == n_valu:
        cyc_valu = sum(valu_cost(op) * n for op, n in ops.items())
    else:       # a counts file without the opcode histogram: class costs
        cyc_valu = VALU_COST_DEFAULT * c.get("valu", 0) + VALU_COST_PACKED * c.get("valu_packed", 0) + VALU_COST_TRANS * c.get("valu_trans", 0)
    tiles_per_cu = -(-tiles // n_cus)
    waves_per_simd = ISA_WAVE_TRIPS.get(kernel, 8) / 4.0      # wavefront-trips per tile and SIMD
    to_s = lambda cyc: tiles_per_cu * waves_per_simd * cyc / (clock_ghz * 1e9)
    t_valu, t_mfma = to_s(cyc_valu), to_s(cyc_mfma)
    t_serial, t_overlap = t_valu + t_mfma, max(t_valu, t_mfma)
    t_hetero = t_mfma + max(0.0, t_valu - HETERO_VALU_RATE * t_mfma)
    t_meas = avg_ms * 1e-3
    return {"t_serial_us": t_serial * 1e6, "t_hetero_us": t_hetero * 1e6, "t_overlap_us": t_overlap * 1e6,
            "frac_serial": t_serial / t_meas, "frac_hetero": t_hetero / t_meas, "frac_overlap": t_overlap / t_meas,
            "t_valu_us": t_valu * 1e6, "t_mfma_us": t_mfma * 1e6,
            "cycles_per_simd_tile": {"valu": waves_per_simd * cyc_valu, "mfma": waves_per_simd * cyc_mfma,
                                     "measured": avg_ms * 1e-3 * clock_ghz * 1e9 / tiles_per_cu},
            "counts_per_wavefront_trip": {"mfma": sum(mf.values()), "valu": c.get("valu", 0), "valu_packed": c.get("valu_packed", 0),
                                          "valu_trans": c.get("valu_trans", 0), "salu": c.get("salu", 0), "lds": c.get("lds", 0),
                                          "vmem": c.get("vmem", 0), "barriers": c.get("barrier", 0), "waitcnt": c.get("waitcnt", 0)},
            "mean_valu_cost_cycles": cyc_valu / n_valu if n_valu else None,
            "tiles_per_cu": tiles_per_cu, "wavefront_trips_per_tile": ISA_WAVE_TRIPS.get(kernel, 8), "clock_GHz": clock_ghz,
            "vgprs": c.get("vgprs"), "lds_bytes": c.get("lds_bytes"), "counts_from": c.get("file"), "symbol": c.get("symbol"),
            "note": "bracket of the instruction-issue bound: static counts of the shipped code object's tile loop (tools/isa_counts.py) x "
                    "per-opcode issue cost with two wavefronts per SIMD (tools/probe/valu_cost_probe.hip: 2.5-3.8 cycles, transcendental / "
                    "permlane swap 6.6), 16-bit 16x16x32 MFMA 16 cycles; serial = VALU + MFMA, overlap = max, hetero = the overlap "
                    "tools/probe/hetero_probe.hip measured between a pure MFMA and a pure VALU wavefront. Barriers, LDS and memory "
                    "latency, s_waitcnt / s_nop are in none of them"}


def end_to_end_sharded(eng, rank, world, n_files=None):
    """The many-PDB scan on ALL ranks of a multi-GPU run: the same 1 024 files as ``end_to_end``, sharded by
    LPT over the ranks — to ONE CSV through dist.scan_files_to_csv (every rank formats its own shard, byte counts exchanged, text
    placed in the one file) and to the binary tables through dist.scan_files (one gather to rank 0, over RCCL straight from the
    device buffers). Wall time between barriers, median of three runs. Called by every rank; -> the record on rank 0, None elsewhere.
    The loop being sharded: analysis/SSM.py:105-176."""
    import shutil
    import tempfile
    import torch.distributed as dist
    from thermompnn_amd import dist as tdist
    from thermompnn_amd import pipeline, ssm_scan
    from thermompnn_amd.synthetic import backbone_pdb_text
    n_files = int(n_files or os.environ.get("TMPNN_E2E_FILES", "1024"))
    box = [None]
    if rank == 0:
        try:                                                 # (a failure here must still reach the broadcast the others wait in)
            base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
            d = tempfile.mkdtemp(prefix="tmpnn_e2e_", dir=base)
            lens = np.random.default_rng(1).integers(64, 513, size=n_files)
            paths = []
            for i, L in enumerate(lens):
                X, seq = synthetic_backbone(int(L), 1000 + i)
                paths.append(os.path.join(d, f"syn_{i:04d}.pdb"))
                with open(paths[-1], "w") as fh:
                    fh.write(backbone_pdb_text(X, seq))
            box[0] = (d, paths, int(lens.sum()), base is not None)
        except Exception as e:                               # noqa: BLE001
            box[0] = ("error", repr(e)[:300])
    dist.broadcast_object_list(box, src=0)
    if box[0][0] == "error":
        raise RuntimeError("end_to_end_sharded: rank 0 could not write the input files: " + box[0][1])
    d, paths, T, tmpfs = box[0]
    chains = ["A"] * n_files
    try:
        def timed(fn, n=3):
            runs = []
            for _ in range(n):
                dist.barrier()
                t0 = time.perf_counter()
                fn()
                dist.barrier()
                runs.append(time.perf_counter() - t0)
            runs.sort()
            return runs

        csv_path, npz_path = os.path.join(d, "out.csv"), os.path.join(d, "out.npz")
        to_csv = lambda ps=paths: tdist.scan_files_to_csv(eng, ps, chains[:len(ps)], csv_path, include_cys=True)

        def to_npz(ps=paths):
            res = tdist.scan_files(eng, ps, chains[:len(ps)])
            if rank == 0:
                ssm_scan.write_scan_npz(npz_path, res)
        to_csv(paths[:128])                                  # warm-up (pinned slots, first launches), not timed
        to_npz(paths[:128])
        rc, rn = timed(to_csv), timed(to_npz)
        if rank != 0:
            return None
        size = os.path.getsize(csv_path)
        rec = lambda runs, rows: {"wall_s": runs[len(runs) // 2], "wall_s_all_runs": runs, "preds_per_s": 20 * T / runs[len(runs) // 2],
                                  "files_per_s": n_files / runs[len(runs) // 2], "rows": rows, "reported": f"median of {len(runs)} runs"}
        return {"files": n_files, "residues": T, "preds": 20 * T, "ranks": world, "tmpfs": tmpfs, "usable_cpus_per_rank": pipeline.usable_cpus(),
                "to_csv": dict(rec(rc, 20 * T), output_MB=size / 1e6, writer="sharded: every rank formats its LPT shard, one file (dist.scan_files_to_csv)"),
                "to_npz": dict(rec(rn, T), gather="one padded gather to rank 0 (device buffers under RCCL)"),
                "workload": "BASELINE configs[2] protein set as backbone-only PDB files on tmpfs, sharded over the ranks by LPT; wall time "
                            "between barriers from the path list to the closed output file"}
    finally:
        dist.barrier()
        if rank == 0:
            shutil.rmtree(d, ignore_errors=True)


def shader_clock_ghz(lib, device):
    """Effective shader clock right after the timed region (tmpnn_clock_probe: cycle counter against the 100 MHz reference
    under a saturated MFMA stream on all CUs)."""
    try:
        blocks, iters = 256, 4000
        out = torch.zeros(2 * blocks, dtype=torch.int64, device=device)
        sink = torch.zeros(256, device=device)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if lib.tmpnn_clock_probe(blocks, iters, C.c_void_p(out.data_ptr()), C.c_void_p(sink.data_ptr()), st) != 0:
            return None
        torch.cuda.synchronize()
        o = out.cpu().view(-1, 2).double()
        return float(o[:, 0].mean() / (o[:, 1].mean() / 100e6) / 1e9)
    except Exception:
        return None


def clock_under_load_ghz(lib, device, step, ms_per_step):
    """The clock the chip keeps while the timed workload runs (tmpnn_clock_monitor: one sleeping wavefront on a side stream reads the
    shader cycle counter against the 100 MHz reference for ~60 % of a second, un-timed, run of steps). Round 5: 2.03-2.1 GHz under
    the f16x2 pipeline against 2.35-2.4 GHz for the same chip idle or under an fp32-MFMA loop — the forward is power-limited, and
    every cycle figure of the issue bracket has to be priced with THIS clock."""
    try:
        n = max(12, int(60.0 / max(ms_per_step, 1e-3)))                # ~60 ms of forwards
        iters = max(1000, int(0.6 * n * ms_per_step * 1e-3 * 2.0e9 / 8128))
        out = torch.zeros(2, dtype=torch.int64, device=device)
        side = torch.cuda.Stream(device=device)
        for _ in range(4):
            step(False)
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        if lib.tmpnn_clock_monitor(iters, C.c_void_p(out.data_ptr()), C.c_void_p(side.cuda_stream)) != 0:
            return None
        for _ in range(n):
            step(False)
        torch.cuda.synchronize()
        cyc, ticks = (float(v) for v in out.cpu())
        return cyc / (ticks / 100e6) / 1e9 if ticks > 0 else None
    except Exception:
        return None


def build_batch(n_proteins, L, seed0, device):
    xs, ss = [], []
    for i in range(n_proteins):
        X, seq = synthetic_backbone(L, seed0 + i)
        xs.append(X)
        ss.append([AA20.index(c) for c in seq])
    T = n_proteins * L
    t = lambda a, dt: torch.tensor(np.asarray(a), dtype=dt, device=device)
    return dict(X=t(np.concatenate(xs), torch.float32), S=t(np.concatenate(ss), torch.int32),
                mask=torch.ones(T, device=device), ridx=t(np.tile(np.arange(L), n_proteins), torch.int32),
                cenc=torch.ones(T, dtype=torch.int32, device=device),
                offsets=t(np.arange(n_proteins + 1) * L, torch.int32), T=T, L=L, n=n_proteins,
                X_cpu=xs[0], S_cpu=ss[0])


def fetch_profile(lib):
    cap = 64
    names = (C.c_char_p * cap)()
    ms = (C.c_double * cap)()
    cnt = (C.c_int64 * cap)()
    n = lib.tmpnn_profile_fetch(names, ms, cnt, cap)
    if n < 0:
        raise RuntimeError(lib.tmpnn_last_error().decode())
    return {names[i].decode(): (ms[i], cnt[i]) for i in range(n)}


def gather_microbench(eng, device, n_nodes=16384, K=48, C_=128, iters=20):
    """Standalone gather_nodes roofline (the 'gather HBM GB/s' of BASELINE.json:metric): out[r,:] = nodes[idx[r],:].
    Algorithmic bytes per call = idx (int32) + node table once + output  (SURVEY §8d)."""
    g = torch.Generator(device="cpu").manual_seed(0)
    nodes = torch.randn(n_nodes, C_, generator=g).to(device)
    # neighbour-like indices: each residue gathers from its own 256-residue protein
    base = (torch.arange(n_nodes) // 256 * 256).repeat_interleave(K)
    idx = (base + torch.randint(0, 256, (n_nodes * K,), generator=g)).to(device=device, dtype=torch.int32)
    for _ in range(3):
        out = eng.gather_rows(nodes, idx)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(iters):
        out = eng.gather_rows(nodes, idx)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / iters
    nbytes = idx.numel() * 4 + nodes.numel() * 4 + out.numel() * 4
    # The gather is 97 % writes, so the measured HBM ceiling beside it is a write-only fill of the same output (a
    # read+write copy moves twice the bytes per output element and is not a roof for this kernel); the copy is reported too.
    for _ in range(3):
        out.fill_(1.0)
    ev0.record()
    for _ in range(iters):
        out.fill_(1.0)
    ev1.record()
    torch.cuda.synchronize()
    fill_ms = ev0.elapsed_time(ev1) / iters
    fill_gbs = out.numel() * 4 / fill_ms / 1e6
    src = torch.empty_like(out)
    for _ in range(3):
        src.copy_(out)
    ev0.record()
    for _ in range(iters):
        src.copy_(out)
    ev1.record()
    torch.cuda.synchronize()
    copy_ms = ev0.elapsed_time(ev1) / iters
    copy_gbs = 2 * out.numel() * 4 / copy_ms / 1e6
    return {"bound": "hbm", "achieved": nbytes / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": nbytes / ms / 1e6 / HBM_PEAK_GBS, "traffic": None, "kernel": "gather_rows (tmpnn_gather_rows_i32)",
            "bytes_per_launch": nbytes, "ms_per_launch": ms, "rows": idx.numel(), "C": C_,
            "measured_fill_GBps": fill_gbs, "frac_of_measured_fill": nbytes / ms / 1e6 / fill_gbs,
            "measured_copy_GBps": copy_gbs}


def end_to_end(eng, n_files=None, budget_note=None):
    """PDB files -> ddG at engine speed: BASELINE configs[2]'s protein set (1 024
    synthetic proteins, L ~ U[64, 512], rng(1), backbone seed 1000 + i) written as backbone-only PDB files to tmpfs, then the
    whole host path timed wall-clock — native threaded parse -> pinned staging -> async H2D -> fused forward -> async D2H ->
    native columnar CSV writer (or the binary tables), as a three-stage pipeline over chunks of files
    (thermompnn_amd/pipeline.py). File generation and a warm-up scan (pinned allocations, first launches) are outside the
    timed region; everything from the path list to the closed output file is inside. The reference's serial loop:
    analysis/SSM.py:105-176, custom_inference.py:94-111."""
    import shutil
    import tempfile
    from thermompnn_amd import custom_inference, pipeline, ssm_scan
    from thermompnn_amd.synthetic import backbone_pdb_text
    n_files = int(n_files or os.environ.get("TMPNN_E2E_FILES", "1024"))
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="tmpnn_e2e_", dir=base)
    try:
        t0 = time.perf_counter()
        lens = np.random.default_rng(1).integers(64, 513, size=n_files)
        paths = []
        for i, L in enumerate(lens):
            X, seq = synthetic_backbone(int(L), 1000 + i)
            paths.append(os.path.join(d, f"syn_{i:04d}.pdb"))
            with open(paths[-1], "w") as fh:
                fh.write(backbone_pdb_text(X, seq))
        gen_s = time.perf_counter() - t0
        in_bytes = sum(os.path.getsize(p) for p in paths)
        chains = ["A"] * n_files
        T = int(lens.sum())
        cpus = pipeline.usable_cpus()
        ssm_scan.scan_to_file(eng, paths[:128], chains[:128], os.path.join(d, "warm.csv"))          # warm-up, not timed
        ssm_scan.scan_to_file(eng, paths[:128], chains[:128], os.path.join(d, "warm.npz"))
        out = {"files": n_files, "residues": T, "preds": 20 * T, "input_MB": in_bytes / 1e6, "usable_cpus": cpus,
               "tmpfs": base is not None, "generate_files_s": gen_s,
               "workload": "BASELINE configs[2] protein set as backbone-only PDB files (N, CA, C, O records) on tmpfs; full 20 x L SSM "
                           "of every protein; timed from the path list to the closed output file, model already loaded"}

        def leg(name):
            path = os.path.join(d, "out." + name)
            t1 = time.perf_counter()
            rows, st = ssm_scan.scan_to_file(eng, paths, chains, path, include_cys=True)
            wall = time.perf_counter() - t1
            size = os.path.getsize(path)
            os.remove(path)
            return {"wall_s": wall, "files_per_s": n_files / wall, "preds_per_s": 20 * T / wall, "rows": rows, "output_MB": size / 1e6,
                    "chunks": st.chunks, "stage_busy_s": {"parse_and_pack": st.parse_s, "gpu_enqueue": st.gpu_enqueue_s,
                                                          "wait_for_gpu": st.gpu_wait_s, "write": st.sink_s},
                    "note": "stage_busy_s are per-stage busy seconds of concurrently running stages (their sum exceeds wall_s)"}

        def median_of(name, n):
            """n runs; the reported leg is the MEDIAN run (a best-of-three over a 2.4 x spread is not a rate), the best
            and every run ride along."""
            runs = sorted((leg(name) for _ in range(n)), key=lambda r: r["wall_s"])
            med = dict(runs[len(runs) // 2])
            med["wall_s_all_runs"] = [r["wall_s"] for r in runs]
            med["wall_s_best"], med["preds_per_s_best"] = runs[0]["wall_s"], runs[0]["preds_per_s"]
            med["wall_s_spread"] = runs[-1]["wall_s"] / runs[0]["wall_s"]
            med["reported"] = f"median of {n} runs"
            return med

        out["to_csv"] = median_of("csv", 5)
        out["to_npz"] = median_of("npz", 5)

        def tmpfs_ceiling(n_bytes, n_threads, n):
            """What the file system gives a writer that formats NOTHING: `n_threads` threads pwrite ready-made 8 MB blocks into a fresh
            file of the CSV leg's size, n runs (is the run-to-run spread of the CSV leg the writer or tmpfs?)."""
            import threading
            blk = 8 << 20
            buf = bytes(blk)
            offs = list(range(0, n_bytes, blk))
            path = os.path.join(d, "ceiling.bin")
            runs = []
            for _ in range(n):
                fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)

                def w(k):
                    for o in offs[k::n_threads]:
                        os.pwrite(fd, buf[:min(blk, n_bytes - o)], o)
                t1 = time.perf_counter()
                th = [threading.Thread(target=w, args=(k,)) for k in range(n_threads)]
                [x.start() for x in th]
                [x.join() for x in th]
                os.close(fd)
                runs.append(time.perf_counter() - t1)
                os.remove(path)
            runs.sort()
            return {"bytes": n_bytes, "threads": n_threads, "block_MB": 8, "wall_s_all_runs": runs, "wall_s_median": runs[len(runs) // 2],
                    "GB_per_s_median": n_bytes / runs[len(runs) // 2] / 1e9, "wall_s_spread": runs[-1] / runs[0],
                    "note": "threads pwrite zero-filled 8 MB blocks into a fresh file of the CSV leg's size on the same file system: the "
                            "floor and the run-to-run spread the page allocation of the file system itself puts under any writer"}

        out["tmpfs_write_ceiling"] = tmpfs_ceiling(int(out["to_csv"]["output_MB"] * 1e6), max(1, min(16, cpus)), 5)
        # stage ceilings measured alone: the parser over all files, the forward over the same ragged chunks
        import ctypes as C2
        lib = _lib.load()
        cp = (C2.c_char_p * n_files)(*[p.encode() for p in paths])
        cc = (C2.c_char_p * n_files)(*[b"A"] * n_files)
        hs = (C2.c_void_p * n_files)()
        for nt in (1, max(1, cpus - 2)):
            t1 = time.perf_counter()
            lib.tmpnn_pdb_parse_batch(cp, cc, n_files, nt, hs)
            dt = time.perf_counter() - t1
            for h in hs:
                lib.tmpnn_pdb_free(C2.c_void_p(h))
            out.setdefault("parse_alone", {})[f"{nt}_threads"] = {"files_per_s": n_files / dt, "MB_per_s": in_bytes / dt / 1e6,
                                                                  "preds_per_s_equivalent": 20 * T / dt}
        # the single-structure script (BASELINE configs[0] shape): examples-style 2OCJ -> CSV
        pdb = os.path.join(REPO, "tests", "golden", "2OCJ.pdb")
        if os.path.exists(pdb):
            import contextlib
            import io
            t1 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):          # (the script announces its output file on stdout)
                custom_inference.main(["--pdb", pdb, "--chain", "A", "--synthetic_weights", "0", "--out_dir", d])
            cold = time.perf_counter() - t1
            model = custom_inference.load_model("", "", 0, device=eng.device)
            custom_inference.ssm_to_csv(model, pdb, "A", os.path.join(d, "w.csv"))
            warm = 1e9
            for _ in range(7):
                t1 = time.perf_counter()
                custom_inference.ssm_to_csv(model, pdb, "A", os.path.join(d, "w.csv"))
                warm = min(warm, time.perf_counter() - t1)
            shaped = 1e9
            for _ in range(3):                                       # (the first call also pays one-time allocations)
                t1 = time.perf_counter()
                custom_inference.write_csv(custom_inference.ssm_rows(model, pdb, "A"), os.path.join(d, "r.csv"))
                shaped = min(shaped, time.perf_counter() - t1)
            out["custom_inference_2OCJ"] = {"main_s": cold, "pdb_to_csv_warm_s": warm, "reference_shaped_api_s": shaped,
                                            "rows": 3880, "note": "main_s = the whole script in-process (synthetic checkpoint written "
                                            "and loaded, weights repacked, parse, forward, CSV); pdb_to_csv_warm_s = parse + forward + "
                                            "CSV with the model resident; reference_shaped_api_s = TransferModel.forward(pdb, "
                                            "mutations) with one Mutation and one result dict per mutant + csv.writer"}
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def usable_cpus():
    """Logical CPUs this process may actually use: the scheduler affinity mask, capped by the cgroup CPU quota (a container
    that shows 256 CPUs in /proc/cpuinfo may be allowed a fraction of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return n, quota


def host_cpu_info():
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    aff, quota = usable_cpus()
    return {"host_cpus": os.cpu_count(), "cpu_model": model, "affinity_cpus": aff, "cgroup_cpu_quota": quota}


def cpu_baseline_reference_shaped(batch, threads, budget_s=8.0):
    """The reference's own loop shape (transfer_model.py:86-120): ONE encoder/decoder forward, then the head evaluated per
    MUTATION (20 x L times) — oracle.transfer_forward_loop on a bounded sample of positions, scaled to the full scan."""
    from oracle import thermompnn_oracle as orc
    W = synthetic_state_dict(0)
    L = batch["L"]
    X = torch.tensor(batch["X_cpu"], dtype=torch.float32)[None]
    S = torch.tensor(batch["S_cpu"])[None]
    ones, ar = torch.ones(1, L), torch.arange(L)[None]
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    from thermompnn_amd.datasets import ALPHABET, Mutation
    n_pos = 16
    muts = [Mutation(position=p, wildtype=AA20[batch["S_cpu"][p]], mutation=AA20[a], ddG=None, pdb="syn")
            for p in range(n_pos) for a in range(20)]
    with torch.no_grad():
        t0 = time.perf_counter()
        orc.transfer_forward_loop(W, X, S, ones, ones, ar, ones.long(), muts, ALPHABET, 48)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        orc.transfer_forward_loop(W, X, S, ones, ones, ar, ones.long(), muts[:20], ALPHABET, 48)
        dt_small = time.perf_counter() - t1
    torch.set_num_threads(old)
    per_mut = max(dt - dt_small, 1e-9) / (len(muts) - 20)              # head cost per mutation
    body = max(dt_small - 20 * per_mut, 0.0)                            # the shared encoder/decoder forward
    full = body + 20 * L * per_mut
    return {"value": 20 * L / full, "unit": "preds/s", "cores": threads, "kind": "port",
            "sample": f"reference-shaped loop: one forward + the head per mutation, timed on {len(muts)} mutations of one "
                      f"synthetic L={L} protein and scaled to its 20 x L = {20 * L} mutants (head {per_mut * 1e3:.2f} ms/mutation, "
                      f"forward {body:.2f} s)"}


def cpu_baseline(batch, budget_s=15.0):
    """The CPU oracle (oracle/thermompnn_oracle.py, kind 'port') on this box's host cores: full SSM of ONE synthetic
    L=256 protein, vectorised head (the oracle is the checker; here it is only timed, as a baseline). torch's default
    thread count oversubscribes the small ops of a single protein, so a few thread counts share the budget and the best
    one is reported with the number of threads it used."""
    from oracle import thermompnn_oracle as orc
    W = synthetic_state_dict(0)
    L = batch["L"]
    X = torch.tensor(batch["X_cpu"], dtype=torch.float32)[None]
    S = torch.tensor(batch["S_cpu"])[None]
    ones, ar = torch.ones(1, L), torch.arange(L)[None]
    default_threads = torch.get_num_threads()
    cands = sorted({default_threads, min(default_threads, 32), min(default_threads, 8)}, reverse=True)
    tried, best = {}, None
    with torch.no_grad():
        for nt in cands:
            torch.set_num_threads(nt)
            orc.ssm_table(W, X, S, ones, ones, ar, ones.long(), 48)          # warm-up
            t0, reps = time.perf_counter(), 0
            while time.perf_counter() - t0 < budget_s / len(cands) or reps < 2:
                orc.ssm_table(W, X, S, ones, ones, ar, ones.long(), 48)
                reps += 1
            dt = time.perf_counter() - t0
            tried[nt] = reps * L * 20 / dt
            if best is None or tried[nt] > tried[best]:
                best = nt
                best_desc = (reps, dt)
    torch.set_num_threads(default_threads)
    info = host_cpu_info()
    try:
        shaped = cpu_baseline_reference_shaped(batch, best)
    except Exception as e:                                   # a baseline leg must never take the bench line down
        shaped = {"error": repr(e)}
    return {"value": tried[best], "unit": "preds/s", "cores": best, "kind": "port", **info,
            "form": "vectorised (head once per position)", "reference_shaped": shaped,
            "sample": f"{best_desc[0]} x full SSM of one synthetic L={L} protein (5120 preds each), vectorised head, torch CPU "
                      f"fp32, {best_desc[1]:.1f} s at {best} threads (threads tried -> preds/s: "
                      + ", ".join(f"{k}: {v:.0f}" for k, v in tried.items()) + ")"}


def cpu_worker(threads, t_start, t_end):
    """One process of the saturated CPU leg (`bench.py --cpu-worker THREADS T_START T_END`): the CPU oracle's full SSM of one
    synthetic L=256 protein, repeated inside the common wall-clock window; prints the repetitions completed in it."""
    torch.set_num_threads(threads)
    from oracle import thermompnn_oracle as orc
    W = synthetic_state_dict(0)
    L = 256
    Xn, seq = synthetic_backbone(L, 0)
    X = torch.tensor(Xn, dtype=torch.float32)[None]
    S = torch.tensor([AA20.index(c) for c in seq])[None]
    ones, ar = torch.ones(1, L), torch.arange(L)[None]
    reps = 0
    with torch.no_grad():
        orc.ssm_table(W, X, S, ones, ones, ar, ones.long(), 48)            # warm-up
        late = time.time() > t_start                                       # (a late starter under-counts; it is reported)
        while time.time() < t_start:
            time.sleep(0.01)
        while True:
            orc.ssm_table(W, X, S, ones, ones, ar, ones.long(), 48)
            if time.time() > t_end:
                break
            reps += 1
    print(json.dumps({"reps": reps, "late": late}))


def cpu_baseline_saturated(threads_per_proc=8, window_s=10.0, startup_s=25.0):
    """Throughput-fair CPU leg: P = host_cpus / 8 processes x 8 threads, every one running the oracle's vectorised SSM of
    one L=256 protein for the same wall-clock window; aggregate preds/s over the window (repetitions that END inside it)."""
    import subprocess
    aff, quota = usable_cpus()
    ncpu = int(min(aff, quota)) if quota else aff        # what the container may really use
    P = max(1, ncpu // threads_per_proc)
    t_start = time.time() + startup_s                    # the P interpreters import torch and warm up before the window opens
    t_end = t_start + window_s
    env = dict(os.environ, OMP_NUM_THREADS=str(threads_per_proc), MKL_NUM_THREADS=str(threads_per_proc))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(threads_per_proc), repr(t_start), repr(t_end)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for _ in range(P)]
    reps, failed, late = 0, 0, 0
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=startup_s + window_s + 120)
            d = json.loads(out.strip().splitlines()[-1])
            reps += d["reps"]
            late += bool(d["late"])
        except Exception:
            failed += 1
            pr.kill()
    return {"value": reps * 5120 / window_s, "unit": "preds/s", "processes": P, "threads": threads_per_proc,
            "cores": P * threads_per_proc, "usable_cpus": ncpu, "failed_processes": failed, "late_processes": late, "window_s": window_s,
            "sample": f"{P} processes x {threads_per_proc} threads, each repeating the vectorised full SSM of one synthetic L=256 "
                      f"protein (5120 preds) for a common {window_s:.0f} s window: {reps} completed"}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    (one process per GPU; rendezvous on 127.0.0.1 and a free port)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def spread(ev):
    """min / median / max of the per-step durations between consecutive stream events (ms)."""
    d = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(len(ev) - 1))
    if not d:
        return None
    return {"min": d[0], "median": d[len(d) // 2], "max": d[-1], "n": len(d),
            "note": "per-step durations between hipEvents recorded on the launch stream after every step"}


def strong_workload(rank, world, device):
    """BASELINE.json configs[3]: 300 Megascale-like proteins (L in [40, 72]) and 200 000 listed (protein, position, aa) triples
    drawn without replacement (default_rng(2)); proteins sharded over the ranks by LPT on L x min(K, L)."""
    from thermompnn_amd.dist import pack_proteins, partition_proteins
    rng = np.random.default_rng(2)
    lens = rng.integers(40, 73, size=300)
    prots = []
    for i, L in enumerate(lens):
        X, seq = synthetic_backbone(int(L), 5000 + i)
        prots.append(dict(X=X.astype(np.float32), S=np.array([AA20.index(c) for c in seq], dtype=np.int32), mask=np.ones(L, np.float32),
                          residue_idx=np.arange(L, dtype=np.int32), chain_enc=np.ones(L, np.int32)))
    T = int(lens.sum())
    flat = rng.choice(20 * T, size=200000, replace=False)
    shards = partition_proteins([int(x) for x in lens], world, 48)
    rows = [int(sum(lens[i] for i in s)) for s in shards]
    # row of protein i in the gathered [world * max_rows, 21] buffer
    max_rows = max(rows)
    row0 = np.zeros(300, dtype=np.int64)
    for r, s in enumerate(shards):
        pos = r * max_rows
        for i in s:
            row0[i] = pos
            pos += int(lens[i])
    starts = np.concatenate([[0], np.cumsum(lens)])
    res_of = flat // 20
    pid = np.searchsorted(starts, res_of, side="right") - 1
    sel = torch.as_tensor((row0[pid] + (res_of - starts[pid])) * 21 + flat % 20, device=device)
    b = pack_proteins(prots, shards[rank], device)
    b["T"] = rows[rank]
    return dict(batch=b, sel=sel, rows=rows, max_rows=max_rows, lens=lens, total_T=T,
                edges=int(sum(int(L) * min(48, int(L)) for L in lens)),
                my_edges=int(sum(int(lens[i]) * min(48, int(lens[i])) for i in shards[rank])))


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)      # the clocks settle over the first ~15 forwards
    ap.add_argument("--proteins-per-gpu", type=int, default=64)
    ap.add_argument("--length", type=int, default=256)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank owns its own 64 x L=256 proteins; strong: BASELINE configs[3], 300 proteins / "
                         "200 000 listed mutants in total, sharded over the ranks")
    ap.add_argument("--precision", default=None, choices=["f16x2", "bf16x3", "fp32"],
                    help="matrix-core path of the per-edge GEMMs (default: the library default, f16x2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the PDB-files -> CSV / binary wall-clock leg")
    ap.add_argument("--no-live-pmc", action="store_true", help="quote the committed PMC traffic file instead of measuring HBM bytes in two rocprofv3 child passes")
    ap.add_argument("--no-extras", action="store_true",
                    help="timed workload only (no gather microbench / single-protein leg / cpu baseline): use under rocprofv3")
    args = ap.parse_args()

    if os.environ.get("TMPNN_BENCH_WATCHDOG"):     # debugging aid: dump every thread's stack and exit if the run takes longer than N s
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["TMPNN_BENCH_WATCHDOG"]), exit=True)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                           # does not return
    # The contract is ONE line on stdout. The libraries underneath do not know that: RCCL prints a version banner through C stdio
    # (it surfaces when the process exits), gloo announces its peers while connecting. File descriptor 1 therefore points at stderr
    # from here to the end of the process; the JSON line is written to a private duplicate of the original stdout.
    sys.stdout.flush()
    out_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE={world})")
    # TMPNN_BENCH_ONE_DEVICE=1 + TMPNN_BENCH_BACKEND=gloo: smoke-test the N>1 code path on a 1-GPU box (all ranks on
    # cuda:0, collectives staged through gloo). The real multi-GPU run uses one GPU per rank over RCCL.
    one_device = os.environ.get("TMPNN_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("TMPNN_BENCH_BACKEND", "nccl")
    # TMPNN_BENCH_FORCE_GROUP=1: a process group (and the per-step exchange) even with ONE rank — the only way to run RCCL itself
    # on a 1-GPU box: init_process_group("nccl"), all_reduce / all_gather_into_tensor on device buffers, the asynchronous
    # double-buffered overlap (a 1-rank ncclAllGather is a device copy, but every call of the N > 1 path is made)
    grouped = world > 1 or os.environ.get("TMPNN_BENCH_FORCE_GROUP") == "1"
    dev_index = 0 if one_device else local_rank
    if world > 1 and not one_device and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world}: only {torch.cuda.device_count()} GPU(s) visible (one process per GPU)")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    collective = None
    if grouped:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:      # forced 1-rank group without a launcher
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        # proof of what the group is: an all_reduce of ones (on device memory for RCCL) and every rank's device identity
        ones = torch.ones(1, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        props = torch.cuda.get_device_properties(device)
        ident = {"rank": rank, "device_index": dev_index, "name": props.name,
                 "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None)}
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        collective = {"backend": backend + (" (RCCL)" if backend == "nccl" else ""), "world_size": dist.get_world_size(),
                      "ranks_seen": int(ones.item()), "devices": idents,
                      "distinct_devices": len({(d["device_index"], d["uuid"], d["pci_bus_id"]) for d in idents}),
                      "one_device_smoke_mode": one_device}

    lib = _lib.load()
    eng = Engine(synthetic_state_dict(0), device, 48, precision=args.precision)
    B, L = args.proteins_per_gpu, args.length
    strong = args.scaling == "strong"
    if strong:
        sw = strong_workload(rank, world, device)
        batch = sw["batch"]
        T_loc, max_len = batch["T"], batch["max_len"]
        gather_rows_n = sw["max_rows"]                   # every rank's table padded to the largest shard (one all_gather_into_tensor)
    else:
        batch = build_batch(B, L, 100000 * rank, device)
        T_loc, max_len = batch["T"], L
        gather_rows_n = T_loc
    # N > 1: the per-step exchange (all-gather of the ddG tables over RCCL/xGMI) is asynchronous and double-buffered: the
    # collective of step k runs on RCCL's stream under the forward of step k+1; a buffer pair is reused only after its
    # collective has been waited for (stream-side wait, no host sync). Everything is drained inside the timed region.
    nbuf = 2 if grouped else 1
    outs = [{"ddg": torch.zeros((gather_rows_n, 21), dtype=torch.float32, device=device)} for _ in range(nbuf)]
    views = [{"ddg": o["ddg"][:T_loc]} for o in outs]     # the forward writes the first T_loc rows; the padding stays zero
    out = views[0]
    gathered = [torch.empty((world * gather_rows_n, 21), dtype=torch.float32, device=device) for _ in range(2)] if grouped else None
    picked = [None]
    pending = [None, None]
    step_no = [0]

    def step(gather=True):
        k = step_no[0] & 1 if grouped else 0
        if grouped and pending[k] is not None:
            pending[k].wait()
            pending[k] = None
            if strong:
                picked[0] = gathered[k].view(-1)[sw["sel"]]           # the 200 000 listed mutants out of the gathered tables
        # check_status=False: nothing in the step synchronises; the device status word is read once after the timed region
        eng.ssm_forward(batch["X"], batch["S"], batch["mask"], batch["ridx"], batch["cenc"], batch["offsets"],
                        max_len=max_len, out=views[k], check_status=False)
        if not grouped:
            if strong:
                picked[0] = outs[0]["ddg"].view(-1)[sw["sel"]]
        elif not gather:
            pass
        elif backend == "nccl":
            pending[k] = dist.all_gather_into_tensor(gathered[k], outs[k]["ddg"], async_op=True)
        else:                                            # gloo smoke mode (all ranks on one GPU): staged through the host
            host = outs[k]["ddg"].cpu()
            gh = torch.empty((world * gather_rows_n, 21), dtype=torch.float32)
            dist.all_gather_into_tensor(gh, host)
            gathered[k].copy_(gh)
            if strong:
                picked[0] = gathered[k].view(-1)[sw["sel"]]
        step_no[0] += 1

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
                if strong:
                    picked[0] = gathered[k].view(-1)[sw["sel"]]

    def barrier():
        drain()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    # fixed clock warm-up (not counted in --warmup): the shader clock settles over the first ~0.3 s of sustained load.
    # Forward only, NO collective: the loop is bounded by each rank's own clock, so ranks run different numbers of
    # iterations — a collective in here deadlocks intermittently (caught by tests/test_gpu_parity.py::test_two_rank_bench_and_cli).
    # (TMPNN_BENCH_WARMUP_SKEW=1, test hook: rank r warms up (1 + r) x as long, so the iteration counts differ for certain)
    warm_s = float(os.environ.get("TMPNN_BENCH_CLOCK_WARMUP_S", "0.3")) * (1 + rank if os.environ.get("TMPNN_BENCH_WARMUP_SKEW") == "1" else 1)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < warm_s:
        step(gather=False)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()

    def timed(n, gather=True, profile=False):
        """EXACTLY n steps between barrier + synchronize on both sides; -> (seconds, per-kernel profile, step events)."""
        if profile:
            lib.tmpnn_profile_enable(1)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for k in range(n):
            step(gather)
            ev[k + 1].record()
        barrier()
        dt_ = time.perf_counter() - t0
        prof_ = fetch_profile(lib) if profile else {}
        lib.tmpnn_profile_enable(0)
        if grouped:
            tmax = torch.tensor([dt_], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_ = float(tmax.item())
        return dt_, prof_, ev

    # Per-kernel times. Two hipEvent records per launch cost the stream ~2 us each: around all 20 launches of a step that was
    # 2.7 % of `value` (measured: 2.864 vs 2.787 ms). So: (a) an UN-timed pass with every kernel recorded -> the `kernels`
    # table and which kernel is dominant; (b) the timed region with events around the dominant kernel's launches only ->
    # `roofline.achieved` is still measured live inside the timed region, at 0.4 % instead of 2.7 %. (The dominant kernel measures
    # 2-4 % LONGER there than in pass (a) of the same process, spare records in front of its start event or not: presumably the
    # idle gaps that 40 records per step open lower the average power and let the clocks rise, so a kernel timed between gaps is
    # faster than the same kernel in the back-to-back stream that `value` is measured on. Both figures are in `kernels`;
    # rocprofv3's trace opens gaps too.)
    prof_all, n_all, dom_name = {}, 0, None
    if not args.no_profile:
        n_all = max(5, min(args.steps, 20))
        _, prof_all, _ = timed(n_all, True, True)
        if prof_all:
            dom_name = max(prof_all, key=lambda k: prof_all[k][0])
            lib.tmpnn_profile_select(dom_name.encode())
    dt, prof, events = timed(args.steps, True, dom_name is not None)
    lib.tmpnn_profile_select(None)
    eng.check_last_status()                              # a range / max_len problem in the timed work is an error, not a number
    step_spread = spread(events)
    clock_probe_ghz = shader_clock_ghz(lib, device) if rank == 0 else None
    clock_ghz = clock_under_load_ghz(lib, device, step, dt / args.steps * 1e3) if rank == 0 and not grouped else None
    if clock_ghz is None:
        clock_ghz = clock_probe_ghz
    n_cus = torch.cuda.get_device_properties(device).multi_processor_count

    if strong:
        units_per_step, unit_name = 200000, "listed mutant ddG predictions"
        workload = ("BASELINE configs[3]: 300 synthetic Megascale-like proteins (L in [40, 72], K_eff = min(48, L)), full SSM tables "
                    f"on the ranks' LPT shards (FIXED total work: {sw['total_T']} residues = {20 * sw['total_T']} table predictions), "
                    "all-gather of the padded ddG tables, then the 200 000 listed (protein, position, aa) mutants selected out of "
                    "them on every rank; inputs resident in HBM")
    else:
        units_per_step, unit_name = world * B * L * 20, "mutant ddG predictions"
        workload = (f"BASELINE configs[1] x {B}: {B} synthetic L={L} proteins per GPU (K=48, h=128), full 20xL SSM each, inputs "
                    "resident in HBM" + ("; per-step RCCL all-gather of ddG tables (asynchronous, overlapped with the next step)"
                                         if grouped else ""))
    result = {
        "metric": "mutant ddG preds/sec (SSM, L=256, K=48)", "value": units_per_step * args.steps / dt, "unit": "preds/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": {"f16x2": "f32 (per-edge matmuls: f16x2 split on the 16-bit matrix cores, 22 significant bits, fp32 accumulate)",
                  "bf16x3": "f32 (per-edge matmuls: exact bf16x3 split on the 16-bit matrix cores, fp32 accumulate)",
                  "fp32": "f32 (fp32 MFMA throughout)"}[eng.precision],
        "data": "synthetic",
        "config": {"workload": workload, "proteins_per_gpu": (B if not strong else None), "L": (L if not strong else "40..72"),
                   "K": 48, "h": 128, "preds_per_step": units_per_step, "unit_counted": unit_name,
                   "weights": "synthetic_state_dict(seed=0)", "parallelism": f"proteins sharded x{world}",
                   "matmul": eng.precision + " (per-edge GEMMs: fp32 operands as split 16-bit planes, fp32 accumulation, "
                             "fp32-class accuracy; --precision bf16x3|fp32 select the other matrix-core paths, see `modes`)"},
        "ms_per_step_spread": step_spread,
    }
    result["config"]["ms_per_step_spread"] = step_spread
    if collective is not None:
        collective["per_step"] = ("all_gather_into_tensor of [%d, 21] fp32 per rank (%.2f MB gathered), async on RCCL's stream, "
                                  "double-buffered" % (gather_rows_n, world * gather_rows_n * 84 / 1e6))
        result["collective"] = collective
    if strong:
        result["table_preds_per_s"] = 20 * sw["total_T"] * args.steps / dt
        if grouped:                                      # the same steps without the exchange (and without the selection)
            for _ in range(2):
                step(gather=False)
            barrier()
            dt_nc, _, _ = timed(args.steps, False, False)
            result["excl_collective"] = {"value": 200000 * args.steps / dt_nc, "ms_per_step": dt_nc / args.steps * 1e3,
                                         "note": "forward on every rank's shard only: no all-gather, no selection"}
        result["checksum_listed"] = float(picked[0].double().sum().item()) if picked[0] is not None else None
    elif grouped:
        # weak scaling too: the same steps without the per-step exchange, so a scaling curve separates compute from the collective
        for _ in range(2):
            step(gather=False)
        barrier()
        dt_nc, _, _ = timed(args.steps, False, False)
        result["excl_collective"] = {"value": units_per_step * args.steps / dt_nc, "ms_per_step": dt_nc / args.steps * 1e3,
                                     "collective_cost_ms_per_step": (dt - dt_nc) / args.steps * 1e3,
                                     "note": "forward on every rank only: no all-gather of the ddG tables (the timed `value` includes it, "
                                             "asynchronous and overlapped with the next step)"}

    if rank == 0:
        T = T_loc
        edges = sw["my_edges"] if strong else T * min(48, L)
        if prof_all:
            kern = {k: {"avg_ms": ms / n, "launches": int(n), "total_ms": ms, "launches_per_step": n / n_all,
                        "timed": f"un-timed pass of {n_all} steps, every kernel recorded"} for k, (ms, n) in prof_all.items()}
            if dom_name in prof:                             # the dominant kernel: the timed region's own events
                ms, n = prof[dom_name]
                kern[dom_name] = {"avg_ms": ms / n, "launches": int(n), "total_ms": ms, "launches_per_step": n / args.steps,
                                  "timed": "inside the timed region (the only kernel recorded there)",
                                  "avg_ms_untimed_pass": prof_all[dom_name][0] / prof_all[dom_name][1]}
            mode = eng.precision
            for k, v in kern.items():
                f = kernel_flops(k, T, edges)
                if f:
                    v["tflops"] = f / (v["avg_ms"] * 1e-3) / 1e12
                by = kernel_bytes(k, T, edges)
                if by:
                    v["hbm_GBps"] = by / (v["avg_ms"] * 1e-3) / 1e9
                    rf = kernel_roofs(k, T, edges, v["avg_ms"], mode)
                    v["bound"], v["frac_of_binding_roof"] = rf["bound"], rf["frac"]
                    ir = issue_roof(k, T, n_cus, clock_ghz, v["avg_ms"]) if mode == "f16x2" and not strong else None
                    if ir:
                        v["frac_of_issue_serial"], v["frac_of_issue_overlap"] = ir["frac_serial"], ir["frac_overlap"]
                        v["t_issue_serial_us"], v["t_issue_overlap_us"] = ir["t_serial_us"], ir["t_overlap_us"]
            dom = dom_name
            rf = kernel_roofs(dom, T, edges, kern[dom]["avg_ms"], mode)
            side = rf["hbm"] if rf["bound"] == "hbm" else rf["mfma"]
            if rf["bound"] == "hbm":
                achieved, peak, unit = side["achieved_GBps"], HBM_PEAK_GBS, "GB/s"
            else:
                achieved, peak, unit = side["achieved_TFLOPs"], side["peak_TFLOPs"], "TFLOP/s"
            terms = rf["mfma"]["terms"]
            result["roofline"] = {
                "bound": rf["bound"], "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
                "traffic": pmc_traffic(dom, T), "traffic_source": "file: profiles/r*_pmc_traffic.json measured on these kernel sources (hash-stamped)",
                "kernel": dom, "avg_launch_ms": kern[dom]["avg_ms"],
                "issue": issue_roof(dom, T, n_cus, clock_ghz, kern[dom]["avg_ms"]) if mode == "f16x2" and not strong else None,
                "clock_GHz": {"under_this_workload": clock_ghz, "fp32_mfma_probe": clock_probe_ghz,
                              "note": "under_this_workload: tmpnn_clock_monitor on a side stream beside an un-timed run of the same steps — "
                                      "the f16x2 pipeline is power-limited (round 5: 2.03 GHz inside a message kernel, 2.38 GHz with its "
                                      "MFMAs removed); the issue bracket's cycles are priced with it"},
                "algorithmic_bytes_per_launch": rf["hbm"]["bytes_per_launch"], "flops_per_launch": rf["mfma"]["flops_per_launch"],
                "t_hbm_roof_us": rf["t_hbm_us"], "t_mfma_roof_us": rf["t_mfma_us"],
                "hbm": rf["hbm"], "mfma": rf["mfma"],
                "vs_fp32_mfma": {"peak": FP32_MFMA_PEAK_TFLOPS, "frac": rf["mfma"]["achieved_TFLOPs"] / FP32_MFMA_PEAK_TFLOPS},
                "note": ("both roofs of the dominant kernel: hbm = algorithmic bytes (SURVEY §8d: 512 B per edge per pass over h_E, node "
                         "operands once) / 8 TB/s; mfma = algorithmic fp32-class flops against " +
                         (f"{BF16_MFMA_PEAK_TFLOPS:.0f} / {terms} TFLOP/s (the kernel runs them as {terms}-term {mode} split products on the "
                          "16-bit matrix cores)" if terms else "the 157.3 TFLOP/s fp32 matrix pipe") +
                         "; `bound` is the roof with the larger time, `frac` is against it; `issue` is the third roof as a bracket — the "
                         "VALU and matrix-pipe time of the shipped code object's tile loop at the clock measured UNDER THIS WORKLOAD, "
                         "serialised (frac_serial) or perfectly overlapped (frac_overlap); the chip runs this pipeline power-limited "
                         "(clock_GHz), which is why pipe times add whatever the schedule"),
                "traffic_unit": "HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE doubled on gfx950, checked on a 402.7 MB device copy) — measured in two child passes inside this run when this is the default one-GPU line (`traffic_source`), else from profiles/r*_pmc_traffic.json if that file was measured on these kernel sources, else null",
                "timed_with": "hipEvent pairs on the launch stream around this kernel's launches inside the timed region"}
            result["kernels"] = kern
            per_step = lambda k, v: v["launches_per_step"]
            total_fl = sum(kernel_flops(k, T, edges) * per_step(k, v) for k, v in kern.items() if kernel_flops(k, T, edges))
            t_roof = 0.0
            for k, v in kern.items():
                r2 = kernel_roofs(k, T, edges, v["avg_ms"], mode)
                t_roof += max(r2["t_hbm_us"], r2["t_mfma_us"]) * 1e-6 * per_step(k, v)
            result["pipeline"] = {"executed_gflop_per_step": total_fl / 1e9,
                                  "tflops_end_to_end": total_fl / (dt / args.steps) / 1e12,
                                  "frac_of_fp32_mfma_peak": total_fl / (dt / args.steps) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                  "frac_of_binding_roof": t_roof / (dt / args.steps),
                                  "binding_roof_ms_per_step": t_roof * 1e3,
                                  "note": "frac_of_binding_roof = sum over the launches of a step of max(t_hbm, t_mfma) / measured step time; "
                                          "frac_of_fp32_mfma_peak > 1 means faster than the fp32 matrix pipe could run the algorithmic flops",
                                  "gpu_kernel_ms_per_step": sum(v["avg_ms"] * v["launches_per_step"] for v in kern.values())}
    if rank == 0 and not args.no_extras and not strong:
        result["roofline_gather"] = gather_microbench(eng, device)
        result["roofline_gather"]["traffic"] = pmc_traffic("gather_rows", 16384)
        # HBM traffic of the dominant kernel measured in THIS run (two rocprofv3 --pmc child passes, ~1 min; the committed file stays the
        # fallback and the cross-check). Only for the default one-GPU line: T = 16384 is what tools/pmc_workload.py runs.
        if world == 1 and T == 16384 and "roofline" in result and not args.no_live_pmc and eng.precision == "f16x2":
            pats = {"enc_edge": "enc_edge", "gather_rows": "gather_rows_kernel", "device_copy_calibration": "copyBuffer"}
            live, info = live_pmc_traffic(pats)
            if live and live.get(result["roofline"]["kernel"]):
                result["roofline"]["traffic_committed_file"] = result["roofline"]["traffic"]
                result["roofline"]["traffic"] = live[result["roofline"]["kernel"]]
                result["roofline"]["traffic_source"] = info
                result["roofline"]["traffic_over_algorithmic"] = result["roofline"]["traffic"] / result["roofline"]["algorithmic_bytes_per_launch"]
                if live.get("gather_rows"):
                    result["roofline_gather"]["traffic"] = live["gather_rows"]
            else:
                result["roofline"]["traffic_live_pass"] = f"not used: {info}"
        # single-protein latency (the literal configs[1]): B = 1 — stream launches, and the same 20 launches replayed from
        # one captured hipGraph (Engine.capture_graph: the C-ABI never syncs or allocates, so it captures as is)
        one = build_batch(1, L, 0, device)
        o1 = {"ddg": torch.empty((L, 21), dtype=torch.float32, device=device)}
        fwd1 = lambda: eng.ssm_forward(one["X"], one["S"], one["mask"], one["ridx"], one["cenc"], one["offsets"], max_len=L,
                                       out=o1, check_status=False)

        def latency(fn, n=200):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / n

        lat = latency(fwd1)
        result["single_protein"] = {"ms": lat * 1e3, "preds_per_s": L * 20 / lat}
        try:
            if grouped:       # no stream capture next to a live RCCL communicator (its watchdog thread's event queries can invalidate one)
                raise RuntimeError("hipGraph leg skipped in multi-rank runs; measured in the 1-GPU run")
            graph, _ = eng.capture_graph(one["X"], one["S"], one["mask"], one["ridx"], one["cenc"], one["offsets"], max_len=L, out=o1)
            ref = o1["ddg"].clone()
            lat_g = latency(graph.replay)
            result["single_protein"].update({"hipgraph_ms": lat_g * 1e3, "hipgraph_preds_per_s": L * 20 / lat_g,
                                             "hipgraph_bitwise_equal": bool(torch.equal(ref, o1["ddg"]))})
            big = build_batch(1, 2048, 3, device)
            o2 = {"ddg": torch.empty((2048, 21), dtype=torch.float32, device=device)}
            g2, _ = eng.capture_graph(big["X"], big["S"], big["mask"], big["ridx"], big["cenc"], big["offsets"], max_len=2048, out=o2)
            lat2 = latency(g2.replay, 100)
            # the small graph must still replay correctly after a LARGER capture (each graph owns its workspace)
            graph.replay()
            torch.cuda.synchronize()
            result["single_protein_L2048"] = {"hipgraph_ms": lat2 * 1e3, "hipgraph_preds_per_s": 2048 * 20 / lat2}
            result["single_protein"]["hipgraph_bitwise_equal_after_larger_capture"] = bool(torch.equal(ref, o1["ddg"]))
        except Exception as e:                                # graph capture is an extra: report, do not fail the line
            result["single_protein"]["hipgraph_error"] = repr(e)[:300]
        # the other matrix-core paths on the SAME workload, same process (precision is an engine argument)
        if not grouped:
            modes = {}
            for prec in ("f16x2", "bf16x3", "fp32"):
                if prec == eng.precision:
                    continue
                e2 = Engine(synthetic_state_dict(0), device, 48, precision=prec)
                f2 = lambda: e2.ssm_forward(batch["X"], batch["S"], batch["mask"], batch["ridx"], batch["cenc"], batch["offsets"],
                                            max_len=L, out=out, check_status=False)
                for _ in range(5):
                    f2()
                torch.cuda.synchronize()
                lib.tmpnn_profile_enable(1)
                n2 = max(5, args.steps // 2)
                t2 = time.perf_counter()
                for _ in range(n2):
                    f2()
                torch.cuda.synchronize()
                dt2 = (time.perf_counter() - t2) / n2
                pk = fetch_profile(lib)
                lib.tmpnn_profile_enable(0)
                e2.check_last_status()
                ee = pk.get("enc_edge")
                r2 = kernel_roofs("enc_edge", batch["T"], batch["T"] * min(48, L), ee[0] / ee[1], prec) if ee else None
                modes[prec] = {"value": B * L * 20 / dt2, "unit": "preds/s", "ms_per_step": dt2 * 1e3, "steps": n2,
                               "enc_edge_tflops": r2["mfma"]["achieved_TFLOPs"] if r2 else None,
                               "enc_edge_frac_of_fp32_mfma_peak": r2["mfma"]["achieved_TFLOPs"] / FP32_MFMA_PEAK_TFLOPS if r2 else None,
                               "enc_edge_bound": r2["bound"] if r2 else None,
                               "enc_edge_frac_of_binding_roof": r2["frac"] if r2 else None}
            result["modes"] = modes
            # (a reader who takes `dtype` by the letter — fp32 operands, 24 significant bits — finds the exact modes' rates in `config`,
            #  the one object every digest of this line keeps)
            result["config"]["other_precisions"] = ", ".join(f"{k} {v['value'] / 1e6:.1f} M preds/s" for k, v in modes.items())
        if not grouped and not args.no_end_to_end:
            try:
                result["end_to_end"] = end_to_end(eng)
            except Exception as e:                           # an extra leg must never take the bench line down
                result["end_to_end"] = {"error": repr(e)[:400]}
        if not grouped and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(batch)
            try:
                sat = cpu_baseline_saturated()
            except Exception as e:                           # a baseline leg must never take the bench line down
                sat = {"error": repr(e)[:300]}
            result["cpu_baseline"]["saturated"] = sat
            result["cpu_baseline"]["gpu_over_cpu_single_process"] = result["value"] / result["cpu_baseline"]["value"]
            ref_cpu = sat.get("value") or result["cpu_baseline"]["value"]
            result["cpu_baseline"]["gpu_over_cpu"] = result["value"] / ref_cpu
            result["cpu_baseline"]["gpu_over_cpu_note"] = ("against the saturated host (all logical CPUs busy)" if sat.get("value")
                                                           else "against the best single-process thread count")
    if grouped and dist.is_initialized() and not args.no_extras and not args.no_end_to_end and not strong:
        try:                                                 # every rank takes part (the other ranks waited here for rank 0's extras)
            e2e = end_to_end_sharded(eng, rank, world)
        except Exception as e:                               # noqa: BLE001 - an extra leg must never take the bench line down
            e2e = {"error": repr(e)[:400]}
        if rank == 0:
            result["end_to_end"] = e2e
    if rank == 0:
        os.write(out_fd, (json.dumps(result) + "\n").encode())
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
