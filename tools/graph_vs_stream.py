"""Stream launches vs one captured hipGraph for the bench batch and for a single protein (measurement script)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from thermompnn_amd.engine import Engine
from thermompnn_amd.weights import synthetic_state_dict
dev = torch.device("cuda:0")
eng = Engine(synthetic_state_dict(0), dev, 48)
for B, L, n in ((64, 256, 50), (1, 256, 300), (1, 2048, 100), (8, 256, 200)):
    b = bench.build_batch(B, L, 3, dev)
    o = {"ddg": torch.empty((b["T"], 21), device=dev)}
    f = lambda: eng.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=L, out=o, check_status=False)
    g, _ = eng.capture_graph(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=L, out=o)
    res = {}
    for name, fn in (("stream", f), ("graph", g.replay)):
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.3:
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B} L={L}: stream {res['stream']:.4f} ms  graph {res['graph']:.4f} ms  -> {B * L * 20 / res['graph'] / 1e3:.2f} M preds/s (graph)")
