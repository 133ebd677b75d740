"""debug: does node_update8_deep_kernel give the bits of node_update8_split_kernel?  python tools/dbg_deep.py (GPU box)"""
import os, subprocess, sys
import numpy as np
DEBUG_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "thermompnn_amd", "libtmpnn_debug.so")   # the switch lives in the debug variant only
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, ".")
    import bench
    from thermompnn_amd.engine import Engine
    from thermompnn_amd.weights import synthetic_state_dict
    dev = torch.device("cuda:0")
    eng = Engine(synthetic_state_dict(0), dev)
    b = bench.build_batch(1, 300, 7, dev)
    out = eng.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=300, want_hidden=True, want_log_probs=True)
    np.savez(sys.argv[1], **{k: v.cpu().numpy() for k, v in out.items() if hasattr(v, "cpu")})
    sys.exit(0)
for v in ("0", "1"):
    subprocess.run([sys.executable, __file__, f"/tmp/deep{v}.npz"], env=dict(os.environ, TMPNN_NODE_DEEP=v, TMPNN_LIB=DEBUG_LIB), check=True)
a, b = np.load("/tmp/deep0.npz"), np.load("/tmp/deep1.npz")
for k in a.files:
    d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
    print(k, a[k].shape, "max diff", d.max(), "n diff", int((d > 0).sum()), "first", np.argwhere(d > 0)[:3].tolist())
