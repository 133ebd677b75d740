"""do two independent half-batches on two streams overlap the latency-bound kernels (k-NN, node_update,
head) of one half with the per-edge kernels of the other?  One gpurun measurement; prints ms per 64 proteins for
(a) one stream, 64 proteins per forward; (b) two streams, 32 proteins each, launched alternately."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from thermompnn_amd.engine import Engine  # noqa: E402
from thermompnn_amd.weights import synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
W = synthetic_state_dict(0)
engs = [Engine(W, dev, 48) for _ in range(2)]
full = bench.build_batch(64, 256, 0, dev)
halves = [bench.build_batch(32, 256, 0, dev), bench.build_batch(32, 256, 32, dev)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
outs = [{"ddg": torch.empty((h["T"], 21), device=dev)} for h in halves]
out_full = {"ddg": torch.empty((full["T"], 21), device=dev)}


def fwd(e, b, o):
    e.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=256, out=o, check_status=False)


def one():
    fwd(engs[0], full, out_full)


def two():
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            fwd(engs[k], halves[k], outs[k])


def timeit(f, n=40):
    for _ in range(15):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for s in streams:
    s.wait_stream(torch.cuda.current_stream(dev))
a = timeit(one)
b = timeit(two)
a2 = timeit(one)
ref = torch.cat([outs[0]["ddg"], outs[1]["ddg"]])
print(f"one stream, 64 proteins: {a:.3f} ms (again: {a2:.3f}); two streams x 32 proteins: {b:.3f} ms  -> {100 * (a - b) / a:+.1f} %; "
      f"bitwise equal to the single batch: {bool(torch.equal(ref, out_full['ddg']))}")
