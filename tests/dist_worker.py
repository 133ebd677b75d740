"""Worker for the multi-rank tests (launched by python -m torch.distributed.run): every rank builds the same synthetic
protein set, runs the REAL engine through dist.ssm_scan (LPT shards + one all-gather) and rank 0 saves what it got.
    python -m torch.distributed.run --nproc-per-node 2 tests/dist_worker.py OUT.npz [n_proteins]
All ranks share cuda:0 when TMPNN_ONE_DEVICE=1 (gloo group); one GPU per rank over RCCL otherwise."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermompnn_amd import dist as tdist  # noqa: E402
from thermompnn_amd.engine import Engine  # noqa: E402
from thermompnn_amd.synthetic import synthetic_backbone  # noqa: E402
from thermompnn_amd.weights import synthetic_state_dict  # noqa: E402

AA20 = "ACDEFGHIKLMNPQRSTVWY"


def protein_set(n, seed=2):
    rng = np.random.default_rng(seed)
    prots = []
    for i, L in enumerate(rng.integers(40, 73, size=n)):
        X, seq = synthetic_backbone(int(L), 5000 + i)
        prots.append(dict(X=X.astype(np.float32), S=np.array([AA20.index(c) for c in seq], dtype=np.int32),
                          mask=np.ones(L, np.float32), residue_idx=np.arange(L, dtype=np.int32),
                          chain_enc=np.ones(L, np.int32)))
    return prots


def main():
    out, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24
    rank, world, device = tdist.init_from_env()
    eng = Engine(synthetic_state_dict(0), device, 48)
    prots = protein_set(n)
    with torch.cuda.device(device):
        tables, cen = tdist.ssm_scan(eng, prots, centrality=True)
    shards = tdist.partition_proteins([len(p["S"]) for p in prots], world, 48)
    if rank == 0:
        np.savez(out, world=world, shard_sizes=np.array([len(s) for s in shards]),
                 **{f"t{i}": t.cpu().numpy() for i, t in enumerate(tables)},
                 **{f"c{i}": c.cpu().numpy() for i, c in enumerate(cen)})
    if world > 1:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
