#!/usr/bin/env python3
"""Where does the single-protein forward's time go between kernels? (20 launches with ~5 us end-to-start
hand-overs against the guide's 1.45-1.9 us for a dependent kernel boundary.)
    python tools/gap_probe.py                      host enqueue time vs latency: real forward, trivial chains, torch chain
    rocprofv3 --kernel-trace --output-format csv -d DIR -o gap -- python tools/gap_probe.py --trace
    python tools/gap_probe.py --gaps DIR/.../gap_kernel_trace.csv       per-boundary end-to-start gaps of the traced run"""
import csv
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gaps(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    out = {}
    seq = []
    for a, b in zip(rows, rows[1:]):
        gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        if gap > 200:                                           # a host-side pause between chains
            seq.append(None)
            continue
        name = lambda r: r["Kernel_Name"].split("(")[0].split("<")[0][-40:]
        key = f"{name(a)} -> {name(b)}"
        out.setdefault(key, []).append(gap)
        seq.append(gap)
    dur = {}
    for r in rows:
        dur.setdefault(r["Kernel_Name"].split("(")[0].split("<")[0][-40:], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    print(json.dumps({"gap_us_median_by_boundary": {k: [round(med(v), 2), len(v)] for k, v in sorted(out.items(), key=lambda kv: -len(kv[1]))[:40]},
                      "kernel_us_median": {k: [round(med(v), 2), len(v)] for k, v in dur.items()}}, indent=1))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--gaps":
        return gaps(sys.argv[2])
    import ctypes as C
    import torch
    import bench
    from thermompnn_amd import _lib
    from thermompnn_amd.engine import Engine
    from thermompnn_amd.weights import synthetic_state_dict
    trace = "--trace" in sys.argv
    dev = torch.device("cuda:0")
    lib = _lib.load()
    eng = Engine(synthetic_state_dict(0), dev, 48)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    buf = torch.zeros(1 << 22, device=dev)
    res = {}

    def measure(name, fn, n_launch, reps=200):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        # one call from an idle stream: host time of the enqueue alone, then its completion
        lat = []
        for _ in range(30):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            fn()
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            lat.append((t2 - t1, time.perf_counter() - t1))
        lat.sort()
        res[name] = {"launches": n_launch, "back_to_back_us_per_call": t_all / reps * 1e6, "host_enqueue_us_per_call_back_to_back": t_host / reps * 1e6,
                     "idle_stream_host_enqueue_us": lat[len(lat) // 2][0] * 1e6, "idle_stream_latency_us": sorted(x[1] for x in lat)[len(lat) // 2] * 1e6,
                     "per_launch_us": t_all / reps / n_launch * 1e6}

    for L in (256,) if trace else (256, 64, 2048):
        one = bench.build_batch(1, L, 0, dev)
        o1 = {"ddg": torch.empty((L, 21), dtype=torch.float32, device=dev)}
        fwd = lambda: eng.ssm_forward(one["X"], one["S"], one["mask"], one["ridx"], one["cenc"], one["offsets"], max_len=L, out=o1, check_status=False)
        measure(f"forward_L{L}", fwd, 20, 100 if trace else 200)
    chains = {"trivial_256x512": (256, 512, 0, 0), "trivial_256x512_lds77k": (256, 512, 77 * 1024, 0),
              "trivial_256x512_dirty6MB": (256, 512, 0, 1572864), "trivial_16x512": (16, 512, 0, 0)}
    for name, (g, b, lds, dirty) in chains.items():
        fn = lambda g=g, b=b, lds=lds, dirty=dirty: _lib.check(lib.tmpnn_launch_probe(20, g, b, lds, C.c_void_p(buf.data_ptr()), dirty, st()))
        measure(name, fn, 20, 100 if trace else 200)
    if not trace:
        x = torch.zeros(64, device=dev)

        def torch_chain():
            for _ in range(20):
                x.add_(1.0)
        measure("torch_add_chain", torch_chain, 20)
        # the same forward replayed from a captured graph
        one = bench.build_batch(1, 256, 0, dev)
        o1 = {"ddg": torch.empty((256, 21), dtype=torch.float32, device=dev)}
        g, _ = eng.capture_graph(one["X"], one["S"], one["mask"], one["ridx"], one["cenc"], one["offsets"], max_len=256, out=o1)
        measure("forward_L256_hipgraph", g.replay, 20)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
