"""Multi-GPU site-saturation scans: proteins are independent, so they are sharded across ranks (one
process per GPU, torch.distributed over RCCL/xGMI) with NO data-path collective; the only exchange is one
all-gather of the per-rank ddG tables at the end of a scan.

The reference has no distributed code to mirror (single process, devices=1 — train_thermompnn.py:171-172);
the serial loop this replaces is analysis/SSM.py:105-126 (one protein per forward).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def partition_proteins(lengths: Sequence[int], world: int, k_neighbors: int = 48) -> List[List[int]]:
    """Longest-processing-time greedy bin packing on the edge count L*min(K, L) (work is proportional to
    edges). Deterministic: ties broken by protein index; each rank's list is in ascending protein order."""
    cost = [int(L) * min(int(k_neighbors), int(L)) for L in lengths]
    order = sorted(range(len(lengths)), key=lambda i: (-cost[i], i))
    load = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        shards[r].append(i)
        load[r] += cost[i]
    return [sorted(s) for s in shards]


def all_gather_tables(local: torch.Tensor, rows_per_rank: Sequence[int], group=None) -> List[torch.Tensor]:
    """All-gather ragged [rows_r, C] tables: pad every shard to max rows, ONE all_gather_into_tensor
    (ncclAllGather on RCCL), slice the padding off. Returns the list of per-rank tables on every rank."""
    world = dist.get_world_size(group)
    assert len(rows_per_rank) == world
    max_rows = max(rows_per_rank) if rows_per_rank else 0
    C = local.shape[1]
    padded = local.new_zeros((max_rows, C))
    padded[: local.shape[0]] = local
    out = local.new_empty((world * max_rows, C))
    dist.all_gather_into_tensor(out, padded, group=group)
    return [out[r * max_rows: r * max_rows + rows_per_rank[r]] for r in range(world)]


def scan_sharded(lengths: Sequence[int], compute_shard: Callable[[List[int]], torch.Tensor], group=None,
                 k_neighbors: int = 48, gather: bool = True) -> Optional[List[torch.Tensor]]:
    """Run a many-protein scan across the process group.

    ``compute_shard(protein_ids) -> [sum(L_i for i in ids), C]`` evaluates this rank's proteins (packed in the
    given order) — on a GPU rank that is ``Engine.ssm_forward`` on the packed shard.
    Returns one [L_i, C] table per protein in the ORIGINAL order on every rank (or only this rank's
    tables, as a dict-free list with ``None`` holes, when ``gather=False``)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    shards = partition_proteins(lengths, world, k_neighbors)
    mine = shards[rank]
    local = compute_shard(mine)
    assert local.shape[0] == sum(lengths[i] for i in mine), "compute_shard returned the wrong number of rows"
    tables: List[Optional[torch.Tensor]] = [None] * len(lengths)
    if world == 1 or not gather:
        per_rank = {rank: local}
    else:
        rows = [sum(lengths[i] for i in s) for s in shards]
        per_rank = dict(enumerate(all_gather_tables(local, rows, group)))
    for r, table in per_rank.items():
        pos = 0
        for i in shards[r]:
            tables[i] = table[pos: pos + lengths[i]]
            pos += lengths[i]
    return tables


def pack_proteins(proteins: Sequence[dict], ids: Sequence[int], device):
    """Pack the selected proteins (dicts with X [L,4,3], S, mask, residue_idx, chain_enc as arrays/tensors)
    into the engine's ragged layout."""
    import numpy as np
    if not ids:
        z = lambda dt: torch.zeros(0, dtype=dt, device=device)
        return dict(X=torch.zeros((0, 4, 3), device=device), S=z(torch.int32), mask=z(torch.float32),
                    ridx=z(torch.int32), cenc=z(torch.int32), offsets=torch.zeros(1, dtype=torch.int32, device=device),
                    max_len=0)
    cat = lambda k, dt: torch.as_tensor(np.concatenate([np.asarray(proteins[i][k]) for i in ids])).to(device=device, dtype=dt)
    lens = [len(proteins[i]["S"]) for i in ids]
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=device)
    return dict(X=cat("X", torch.float32), S=cat("S", torch.int32), mask=cat("mask", torch.float32),
                ridx=cat("residue_idx", torch.int32), cenc=cat("chain_enc", torch.int32), offsets=offsets,
                max_len=max(lens))


def ssm_scan(engine, proteins: Sequence[dict], group=None, gather: bool = True):
    """Full SSM of many proteins, sharded over the group's GPUs: -> list of [L_i, 21] ddG tables."""
    lengths = [len(p["S"]) for p in proteins]

    def compute(ids):
        b = pack_proteins(proteins, ids, engine.device)
        if not ids:
            return torch.zeros((0, 21), dtype=torch.float32, device=engine.device)
        return engine.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=b["max_len"])["ddg"]

    return scan_sharded(lengths, compute, group, engine.K, gather)
