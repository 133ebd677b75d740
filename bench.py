#!/usr/bin/env python3
"""ThermoMPNN SSM hot-path benchmark on MI355X (contract: python bench.py --gpus N --steps K --warmup W).

A "step" = one pass of the fused HIP path (tmpnn_ssm_forward: kNN -> featurizer -> 3 enc -> 3 dec -> ddG head)
over one batch of B synthetic proteins of L=256 residues (K=48, h=128; BASELINE.json configs[1] replicated B
times so the 256-CU chip is filled) that is already resident in HBM, producing B*L*20 mutant ddG predictions.
With N > 1 every rank owns its own B proteins (weak scaling, proteins are independent) and each step ends with
the path's one exchange step: an RCCL all-gather of the per-rank ddG tables.

Prints ONE JSON line on rank 0: whole-job preds/s + `roofline` (dominant kernel, HIP-event timed inside the
timed region) + `cpu_baseline` (the CPU oracle on the host cores, bounded sample, rank 0 at N=1 only).
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
    """Executed (algorithmic, minimal-schedule) flops of ONE launch of each kernel; DESIGN.md §5."""
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


PMC_TRAFFIC_FILE = os.path.join(REPO, "profiles", "r02_pmc_traffic.json")


def kernel_source_stamp():
    """Hash of the kernel sources: a PMC file measured on other kernels must not be quoted for these."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(REPO, "thermompnn_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, T):
    """HBM bytes per launch from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes committed under profiles/ (collected
    separately by tools/pmc_traffic.sh at T = 16384, gfx950 FETCH_SIZE correction applied). The file carries the hash of
    the kernel sources it was measured on and the kernel symbol of every entry; a stale file (sources changed since the
    PMC pass) or another batch size yields None rather than an old number."""
    if T != 16384 or not os.path.exists(PMC_TRAFFIC_FILE):
        return None
    try:
        d = json.load(open(PMC_TRAFFIC_FILE))
        if d.get("kernel_source_stamp") != kernel_source_stamp():
            return None
        return d["kernels"].get(kernel, {}).get("traffic_bytes")
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


def host_cpu_info():
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"host_cpus": os.cpu_count(), "cpu_model": model}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)      # the clocks settle over the first ~15 forwards
    ap.add_argument("--proteins-per-gpu", type=int, default=64)
    ap.add_argument("--length", type=int, default=256)
    ap.add_argument("--precision", default=None, choices=["f16x2", "bf16x3", "fp32"],
                    help="matrix-core path of the per-edge GEMMs (default: the library default, f16x2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="timed workload only (no gather microbench / single-protein leg / cpu baseline): use under rocprofv3")
    args = ap.parse_args()

    if os.environ.get("TMPNN_BENCH_WATCHDOG"):     # debugging aid: dump every thread's stack and exit if the run takes longer than N s
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["TMPNN_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # TMPNN_BENCH_ONE_DEVICE=1 + TMPNN_BENCH_BACKEND=gloo: smoke-test the N>1 code path on a 1-GPU box (all ranks on
    # cuda:0, collectives staged through gloo). The real multi-GPU run uses one GPU per rank over RCCL.
    one_device = os.environ.get("TMPNN_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("TMPNN_BENCH_BACKEND", "nccl")
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    lib = _lib.load()
    eng = Engine(synthetic_state_dict(0), device, 48, precision=args.precision)
    B, L = args.proteins_per_gpu, args.length
    batch = build_batch(B, L, 100000 * rank, device)
    # N > 1: the per-step exchange (all-gather of the ddG tables over RCCL/xGMI) is asynchronous and double-buffered: the
    # collective of step k runs on RCCL's stream under the forward of step k+1; a buffer pair is reused only after its
    # collective has been waited for (stream-side wait, no host sync). Everything is drained inside the timed region.
    outs = [{"ddg": torch.empty((batch["T"], 21), dtype=torch.float32, device=device)} for _ in range(2 if world > 1 else 1)]
    out = outs[0]
    gathered = [torch.empty((world * batch["T"], 21), dtype=torch.float32, device=device) for _ in range(2)] if world > 1 else None
    pending = [None, None]
    step_no = [0]

    def step(gather=True):
        k = step_no[0] & 1 if world > 1 else 0
        if world > 1 and pending[k] is not None:
            pending[k].wait()
            pending[k] = None
        # check_status=False: nothing in the step synchronises; the device status word is read once after the timed region
        eng.ssm_forward(batch["X"], batch["S"], batch["mask"], batch["ridx"], batch["cenc"], batch["offsets"],
                        max_len=L, out=outs[k], check_status=False)
        if world > 1 and not gather:
            pass
        elif world > 1 and backend == "nccl":
            pending[k] = dist.all_gather_into_tensor(gathered[k], outs[k]["ddg"], async_op=True)
        elif world > 1:                                  # gloo smoke mode (all ranks on one GPU): staged through the host
            host = outs[k]["ddg"].cpu()
            gh = torch.empty((world * batch["T"], 21), dtype=torch.float32)
            dist.all_gather_into_tensor(gh, host)
            gathered[k].copy_(gh)
        step_no[0] += 1

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    def barrier():
        drain()
        if world > 1:
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
    profile = not args.no_profile
    if profile:
        lib.tmpnn_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    prof = fetch_profile(lib) if profile else {}
    lib.tmpnn_profile_enable(0)
    eng.check_last_status()                              # a range / max_len problem in the timed work is an error, not a number
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    preds_per_step = world * B * L * 20
    result = {
        "metric": "mutant ddG preds/sec (SSM, L=256, K=48)", "value": preds_per_step * args.steps / dt, "unit": "preds/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f16x2": "f32 (per-edge matmuls: f16x2 split on the 16-bit matrix cores, 22 significant bits, fp32 accumulate)",
                  "bf16x3": "f32 (per-edge matmuls: exact bf16x3 split on the 16-bit matrix cores, fp32 accumulate)",
                  "fp32": "f32 (fp32 MFMA throughout)"}[eng.precision],
        "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1] x {B}: {B} synthetic L={L} proteins per GPU (K=48, h=128), "
                               "full 20xL SSM each, inputs resident in HBM" +
                               ("; per-step RCCL all-gather of ddG tables (asynchronous, overlapped with the next step)" if world > 1 else ""),
                   "proteins_per_gpu": B, "L": L, "K": 48, "h": 128, "preds_per_step": preds_per_step,
                   "weights": "synthetic_state_dict(seed=0)", "parallelism": f"proteins sharded x{world}",
                   "matmul": eng.precision + " (per-edge GEMMs: fp32 operands as split 16-bit planes, fp32 accumulation, "
                             "fp32-class accuracy; --precision bf16x3|fp32 select the other matrix-core paths, see `modes`)"},
    }

    if rank == 0:
        T, edges = batch["T"], batch["T"] * min(48, L)
        if prof:
            kern = {k: {"avg_ms": ms / n, "launches": int(n), "total_ms": ms} for k, (ms, n) in prof.items()}
            dom = max(kern, key=lambda k: kern[k]["total_ms"])
            fl = kernel_flops(dom, T, edges)
            achieved = fl / (kern[dom]["avg_ms"] * 1e-3) / 1e12
            mode = eng.precision
            split_kernels = ("enc_edge", "enc_msg", "dec_msg") + (("featurize",) if mode == "f16x2" else ())
            terms = SPLIT_TERMS.get(mode, 0) if dom in split_kernels else 0
            # peak for the ALGORITHMIC (fp32-class) flops: the fp32 matrix pipe, or — on the split paths — the 16-bit
            # dense peak divided by the MFMAs each multiply-accumulate costs
            peak = BF16_MFMA_PEAK_TFLOPS / terms if terms else FP32_MFMA_PEAK_TFLOPS
            result["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                                  "frac": achieved / peak, "traffic": pmc_traffic(dom, T), "kernel": dom,
                                  "note": (f"algorithmic fp32-class flops; the kernel runs them as {terms}-term {mode} split "
                                           f"products on the 16-bit matrix cores, so peak = {BF16_MFMA_PEAK_TFLOPS:.0f} / {terms} "
                                           "TFLOP/s ('executed' = the same ratio in executed 16-bit flops; 'vs_fp32_mfma' = "
                                           "against the fp32 matrix pipe the reference arithmetic would use)"
                                           if terms else "exact fp32 MFMA"),
                                  "executed": ({"dtype": "f16" if mode == "f16x2" else "bf16", "flops_per_launch": terms * fl,
                                                "achieved": terms * achieved, "peak": BF16_MFMA_PEAK_TFLOPS,
                                                "frac": terms * achieved / BF16_MFMA_PEAK_TFLOPS} if terms else None),
                                  "vs_fp32_mfma": {"peak": FP32_MFMA_PEAK_TFLOPS, "frac": achieved / FP32_MFMA_PEAK_TFLOPS},
                                  "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/r02_pmc_traffic.json; null when that file was measured on other kernel sources)",
                                  "flops_per_launch": fl, "avg_launch_ms": kern[dom]["avg_ms"],
                                  "timed_with": "hipEvent pairs on the launch stream inside the timed region"}
            for k, v in kern.items():
                f = kernel_flops(k, T, edges)
                if f:
                    v["tflops"] = f / (v["avg_ms"] * 1e-3) / 1e12
            result["kernels"] = kern
            total_fl = sum(kernel_flops(k, T, edges) * v["launches"] / args.steps for k, v in kern.items() if kernel_flops(k, T, edges))
            result["pipeline"] = {"executed_gflop_per_step": total_fl / 1e9,
                                  "tflops_end_to_end": total_fl / (dt / args.steps) / 1e12,
                                  "frac_of_fp32_mfma_peak": total_fl / (dt / args.steps) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                  "note": "algorithmic fp32-class flops of all kernels; > 1 means faster than the fp32 matrix pipe could run them",
                                  "gpu_kernel_ms_per_step": sum(v["total_ms"] for v in kern.values()) / args.steps}
    if rank == 0 and not args.no_extras:
        result["roofline_gather"] = gather_microbench(eng, device)
        result["roofline_gather"]["traffic"] = pmc_traffic("gather_rows", 16384)
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
            if world > 1:     # no stream capture next to a live RCCL communicator (its watchdog thread's event queries can invalidate one)
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
            result["single_protein_L2048"] = {"hipgraph_ms": lat2 * 1e3, "hipgraph_preds_per_s": 2048 * 20 / lat2}
        except Exception as e:                                # graph capture is an extra: report, do not fail the line
            result["single_protein"]["hipgraph_error"] = repr(e)[:300]
        # the other matrix-core paths on the SAME workload, same process (precision is an engine argument)
        if world == 1:
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
                edge_tf = kernel_flops("enc_edge", batch["T"], batch["T"] * min(48, L)) / (ee[0] / ee[1] * 1e-3) / 1e12 if ee else None
                terms = SPLIT_TERMS.get(prec, 0)
                modes[prec] = {"value": B * L * 20 / dt2, "unit": "preds/s", "ms_per_step": dt2 * 1e3, "steps": n2,
                               "enc_edge_tflops": edge_tf,
                               "enc_edge_frac_of_fp32_mfma_peak": edge_tf / FP32_MFMA_PEAK_TFLOPS if edge_tf else None,
                               "enc_edge_frac_of_its_peak": (edge_tf / (BF16_MFMA_PEAK_TFLOPS / terms if terms else FP32_MFMA_PEAK_TFLOPS))
                               if edge_tf else None}
            result["modes"] = modes
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(batch)
            result["cpu_baseline"]["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
