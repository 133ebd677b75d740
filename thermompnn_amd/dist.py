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
    if local.is_cuda and dist.get_backend(group) != "nccl":
        # a gloo group (CPU tests, the one-device smoke mode) exchanges host buffers; the real path is RCCL on device memory
        host = padded.cpu()
        out_h = host.new_empty((world * max_rows, C))
        dist.all_gather_into_tensor(out_h, host, group=group)
        out = out_h.to(local.device)
    else:
        out = local.new_empty((world * max_rows, C))
        dist.all_gather_into_tensor(out, padded, group=group)
    return [out[r * max_rows: r * max_rows + rows_per_rank[r]] for r in range(world)]


def gather_tables_root(local: torch.Tensor, rows_per_rank: Sequence[int], group=None, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Ragged [rows_r, C] tables -> rank ``dst`` only: ONE padded ``dist.gather`` (ncclGather-style send/recv on RCCL) — 1/N of
    the all-gather's traffic, for consumers where a single rank writes the result (the ssm_scan CLI). Returns the list of
    per-rank tables on ``dst`` and None on the other ranks. ``dst`` is a rank of ``group``."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(rows_per_rank) == world
    max_rows = max(rows_per_rank) if rows_per_rank else 0
    C = local.shape[1]
    on_host = local.is_cuda and dist.get_backend(group) != "nccl"       # gloo group (CPU tests / one-device mode): host buffers
    padded = local.new_zeros((max_rows, C))
    padded[: local.shape[0]] = local
    if on_host:
        padded = padded.cpu()
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
    if rank != dst:
        return None
    return [(b.to(local.device) if on_host else b)[: rows_per_rank[r]] for r, b in enumerate(bufs)]


def scan_sharded(lengths: Sequence[int], compute_shard: Callable[[List[int]], torch.Tensor], group=None,
                 k_neighbors: int = 48, gather=True) -> Optional[List[torch.Tensor]]:
    """Run a many-protein scan across the process group.

    ``compute_shard(protein_ids) -> [sum(L_i for i in ids), C]`` evaluates this rank's proteins (packed in the
    given order) — on a GPU rank that is ``Engine.ssm_forward`` on the packed shard.
    ``gather``: True / "all" — one all-gather, every rank returns every protein's [L_i, C] table in the ORIGINAL order;
    "root" — one gather to rank 0, which returns all tables while the other ranks return only their own (``None`` holes);
    False — no collective, every rank returns only its own tables (``None`` holes)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    shards = partition_proteins(lengths, world, k_neighbors)
    mine = shards[rank]
    local = compute_shard(mine)
    assert local.shape[0] == sum(lengths[i] for i in mine), "compute_shard returned the wrong number of rows"
    tables: List[Optional[torch.Tensor]] = [None] * len(lengths)
    rows = [sum(lengths[i] for i in s) for s in shards]
    if world == 1 or not gather:
        per_rank = {rank: local}
    elif gather == "root":
        got = gather_tables_root(local, rows, group, 0)
        per_rank = dict(enumerate(got)) if got is not None else {rank: local}
    else:
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


def ssm_scan(engine, proteins: Sequence[Optional[dict]], group=None, gather=True, centrality: bool = False,
             chunk_residues: int = 1 << 18, radius: float = 10.0, lengths: Optional[Sequence[int]] = None):
    """Full SSM of many proteins, sharded over the group's GPUs (analysis/SSM.py:105-126 runs them one per forward).

    Every rank runs the ragged pipeline on its LPT shard, in chunks of at most ``chunk_residues`` residues (the
    workspace is ~27.5 KB per residue), then ONE padded all-gather exchanges the tables. With ``centrality`` the
    neighbour count (#CA within ``radius``; compute_centrality, thermompnn_benchmarking.py:20-35, masked on the CA
    atom) rides along as a 22nd column, so there is still a single collective.
    ``lengths``: every protein's length when ``proteins`` only holds THIS rank's shard (``None`` elsewhere) — a rank then never
    needs the structures it does not compute (``parse_sharded``: each rank parses ~2/N of the files instead of all).
    ``gather``: True (all-gather: the API default, every rank gets every table), "root" (one gather to rank 0: what a
    driver needs when only rank 0 writes), False (no exchange).
    -> list of [L_i, 21] ddG tables (device tensors) in the original order — or (tables, [L_i] int32 counts)."""
    lengths = [len(p["S"]) for p in proteins] if lengths is None else [int(x) for x in lengths]
    assert len(lengths) == len(proteins)
    C = 22 if centrality else 21

    def compute(ids):
        local = torch.empty((sum(lengths[i] for i in ids), C), dtype=torch.float32, device=engine.device)
        pos, k = 0, 0
        while k < len(ids):
            chunk, tot = [], 0
            while k < len(ids) and (not chunk or tot + lengths[ids[k]] <= chunk_residues):
                chunk.append(ids[k])
                tot += lengths[ids[k]]
                k += 1
            b = pack_proteins(proteins, chunk, engine.device)
            out = engine.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=b["max_len"])
            local[pos:pos + tot, :21] = out["ddg"]
            if centrality:
                import numpy as np
                ca = torch.as_tensor(np.concatenate([np.asarray(proteins[i].get("ca_mask", proteins[i]["mask"])) for i in chunk]))
                local[pos:pos + tot, 21] = engine.centrality(b["X"], ca, b["offsets"], radius).to(torch.float32)
            pos += tot
        return local

    tables = scan_sharded(lengths, compute, group, engine.K, gather)
    if not centrality:
        return tables
    ddg = [None if t is None else t[:, :21] for t in tables]
    cen = [None if t is None else t[:, 21].round().to(torch.int32) for t in tables]
    return ddg, cen


def parse_sharded(paths: Sequence[str], chains: Optional[Sequence] = None, group=None, k_neighbors: int = 48,
                  parse=None):
    """Parse a many-PDB job without every rank reading every file: rank r parses files r, r + N, ... (a cheap, evenly
    spread length pre-pass), the lengths and sequences are exchanged (a few bytes per residue, ``all_gather_object``), the
    LPT partition is computed from the lengths, and each rank parses only the files of its own shard that it has not read yet.
    -> (proteins: list with this rank's shard filled in and ``None`` elsewhere, lengths, seqs, names). World of one: everything."""
    from . import native_pdb
    parse = parse or native_pdb.parse_pdbs
    n = len(paths)
    chains = list(chains) if chains is not None else [None] * n
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = list(range(rank, n, world))
    got, info, err = {}, {}, None
    try:
        got = dict(zip(mine, parse([paths[i] for i in mine], [chains[i] for i in mine])))
        info = {i: (len(p["S"]), p["seq"], p["name"]) for i, p in got.items()}
    except Exception as e:               # noqa: BLE001 - a bad file on ONE rank must fail EVERY rank (the others would wait in
        err = f"rank {rank}: {e}"        # the collective below, or in the table exchange, until the backend's timeout)
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, (err, info), group=group)
        raise_first_error([p[0] for p in parts])
        info = {k: v for part in parts for k, v in part[1].items()}
    elif err:
        raise RuntimeError(err)
    lengths = [info[i][0] for i in range(n)]
    shard = partition_proteins(lengths, world, k_neighbors)[rank]
    missing = [i for i in shard if i not in got]
    err = None
    try:
        got.update(zip(missing, parse([paths[i] for i in missing], [chains[i] for i in missing])))
    except Exception as e:               # noqa: BLE001
        err = f"rank {rank}: {e}"
    agree_or_raise(err, group)
    proteins = [got[i] if i in set(shard) else None for i in range(n)]
    return proteins, lengths, [info[i][1] for i in range(n)], [info[i][2] for i in range(n)]


def raise_first_error(errors: Sequence[Optional[str]]) -> None:
    bad = [e for e in errors if e]
    if bad:
        raise RuntimeError("sharded scan failed on %d rank(s): %s" % (len(bad), "; ".join(bad)))


def agree_or_raise(err: Optional[str], group=None) -> None:
    """Every rank reports its error (or None); if any rank failed, ALL ranks raise the same RuntimeError — so a failure that
    only one rank sees never leaves the others blocked in the next collective."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, err, group=group)
        raise_first_error(parts)
    elif err:
        raise RuntimeError(err)


def scan_files(engine, paths: Sequence[str], chains: Optional[Sequence] = None, group=None, centrality: bool = False,
               k_neighbors: Optional[int] = None, device_tables: Optional[bool] = None, run_pipeline=None, **pipeline_kw):
    """PDB files -> ddG tables on rank 0, sharded over the group's GPUs, each rank running the parse || forward || copy-back
    pipeline (thermompnn_amd.pipeline.scan_files) on its LPT shard and ONE gather to rank 0 at the end.

    Lengths come from a strided pre-pass (rank r parses files r, r + N, ...: a few hundred microseconds per file) whose
    (length, sequence, name) triples are exchanged with ``all_gather_object``; a file that fails to parse on one rank fails
    every rank. -> rank 0: dict(table float32 [T, 21] host array in the ORIGINAL file order, offsets int64 [n+1], seqs, names,
    neighbors int32 [T] or None, stats); other ranks: the same dict with table / neighbors = None.
    ``run_pipeline``: the function used as ``pipeline.scan_files`` (tests inject a CPU stand-in)."""
    import numpy as np
    from . import native_pdb, pipeline
    n = len(paths)
    chains = list(chains) if chains is not None else [None] * n
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    K = int(k_neighbors or engine.K)
    if world == 1:
        shard = list(range(n))
        info = None
    else:
        info = _length_prepass(paths, chains, rank, world, group)
        shard = partition_proteins([info[i][0] for i in range(n)], world, K)[rank]
    # over RCCL the shard's tables never leave the device between the forwards and the gather (round 5; rounds 3-4 copied every
    # chunk back, concatenated on the host and uploaded the result again for the collective)
    # (``device_tables``: None = when the group's backend is nccl; True forces it — the one-device gloo test mode exercises the path)
    on_device = world > 1 and (dist.get_backend(group) == "nccl" if device_tables is None else bool(device_tables))
    dev_table = torch.empty((sum(info[i][0] for i in shard), 21), dtype=torch.float32, device=engine.device) if on_device else None
    C_ = 22 if centrality else 21
    acc, seqs_l, names_l, lens_l = [], [], [], []

    def sink(ch):
        if dev_table is None:
            t = np.empty((ch.T, C_), np.float32)
            t[:, :21] = ch.table
            if centrality:
                t[:, 21] = ch.neighbors
            acc.append(t)
        elif centrality:
            acc.append(ch.neighbors.astype(np.float32))
        seqs_l.extend(ch.seqs())
        names_l.extend(ch.names)
        lens_l.extend(int(x) for x in np.diff(ch.offsets))

    err, stats = None, None
    try:
        stats = (run_pipeline or pipeline.scan_files)(engine, [paths[i] for i in shard], [chains[i] for i in shard], sink,
                                                      centrality=centrality, device_table=dev_table, **pipeline_kw)
    except Exception as e:               # noqa: BLE001
        err = f"rank {rank}: {type(e).__name__}: {e}"
    agree_or_raise(err, group)
    if dev_table is None:
        local = np.concatenate(acc) if acc else np.zeros((0, C_), np.float32)
    if world == 1:
        lengths, seqs, names, table = lens_l, seqs_l, names_l, local
    else:
        lengths = [info[i][0] for i in range(n)]
        seqs, names = [info[i][1] for i in range(n)], [info[i][2] for i in range(n)]
        # (an agreed error, not an assert: a rank that raises alone leaves the others blocked in the gather below)
        agree_or_raise(None if lens_l == [lengths[i] for i in shard] else
                       f"rank {rank}: a file changed between the length pre-pass and the scan", group)
        shards = partition_proteins(lengths, world, K)
        rows = [sum(lengths[i] for i in s) for s in shards]
        if dev_table is not None:
            loc_t = dev_table
            if centrality:                                   # the neighbour counts ride along as the 22nd column: one collective
                cen = torch.from_numpy(np.concatenate(acc) if acc else np.zeros(0, np.float32)).to(engine.device)
                loc_t = torch.cat([dev_table, cen[:, None]], dim=1)
        else:
            loc_t = torch.from_numpy(local)
            if "nccl" in str(dist.get_backend(group)):       # RCCL moves device memory only: a host-side table (device_tables=False,
                loc_t = loc_t.to(engine.device)              # or a mixed "cpu:gloo,cuda:nccl" group) is uploaded for the collective
        got = gather_tables_root(loc_t, rows, group, 0)
        table = None
        if got is not None:                                  # rank 0: per-rank shard tables -> the original file order
            starts = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
            table = np.empty((int(starts[-1]), C_), np.float32)
            for r, g in enumerate(got):
                g = g.cpu().numpy()
                pos = 0
                for i in shards[r]:
                    table[starts[i]:starts[i + 1]] = g[pos:pos + lengths[i]]
                    pos += lengths[i]
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    out = dict(table=None, neighbors=None, offsets=offsets, seqs=seqs, names=names, stats=stats)
    if table is not None:
        out["table"] = np.ascontiguousarray(table[:, :21])
        out["neighbors"] = np.rint(table[:, 21]).astype(np.int32) if centrality else None
    return out


def _length_prepass(paths, chains, rank: int, world: int, group=None) -> dict:
    """Rank r parses files r, r + N, ... and the (length, sequence, name) triples are exchanged: -> {file index: triple} on every
    rank. A file that fails on ONE rank fails EVERY rank through the same collective."""
    from . import native_pdb
    mine, err, part = list(range(rank, len(paths), world)), None, {}
    try:
        part = {i: (len(p["S"]), p["seq"], p["name"]) for i, p in
                zip(mine, native_pdb.parse_pdbs([paths[i] for i in mine], [chains[i] for i in mine]))}
    except Exception as e:           # noqa: BLE001
        err = f"rank {rank}: {e}"
    parts = [None] * world
    dist.all_gather_object(parts, (err, part), group=group)
    raise_first_error([p[0] for p in parts])
    return {k: v for p in parts for k, v in p[1].items()}


def _copy_range(src_fd: int, dst_fd: int, src_off: int, dst_off: int, n: int) -> None:
    """n bytes of src at src_off -> dst at dst_off, inside the kernel where the platform offers it (copy_file_range), else
    through a buffer."""
    import os
    use_cfr = hasattr(os, "copy_file_range")
    while n > 0:
        if use_cfr:
            try:
                k = os.copy_file_range(src_fd, dst_fd, min(n, 1 << 30), src_off, dst_off)
                if k > 0:
                    src_off, dst_off, n = src_off + k, dst_off + k, n - k
                    continue
            except OSError:
                pass
            use_cfr = False
        buf = os.pread(src_fd, min(n, 8 << 20), src_off)
        if not buf:
            raise OSError("part file shorter than its recorded length")
        os.pwrite(dst_fd, buf, dst_off)
        src_off, dst_off, n = src_off + len(buf), dst_off + len(buf), n - len(buf)


def default_memory_budget(local_world: Optional[int] = None) -> int:
    """Bytes of CSV text a rank may keep in memory before it falls back to a part file: a third of the host's MemAvailable divided by the
    ranks that share the host (LOCAL_WORLD_SIZE), at most 16 GiB. The text of a scan grows as L^2 per protein (every row repeats the
    sequence); the mapping is MAP_NORESERVE, so without this bound the kernel would kill the process (or SIGBUS it inside memcpy) instead
    of the scan failing cleanly or spilling to the part file."""
    import os
    avail = 8 << 30
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    avail = int(line.split()[1]) * 1024
                    break
    except OSError:
        pass
    lw = local_world or int(os.environ.get("LOCAL_WORLD_SIZE", "0") or 0) or 1
    return max(64 << 20, min(16 << 30, avail // (3 * lw)))


def _csv_field_bytes(s: str) -> int:
    """Upper bound of the bytes the native writer emits for one text field: UTF-8, embedded quotes doubled, two enclosing quotes."""
    b = s.encode("utf-8", "surrogateescape")
    return len(b) + b.count(b'"') + 2


def _shared_fs_check(out: str, rank: int, world: int, group) -> Optional[str]:
    """Every rank pwrites into ONE file: that needs a file system all ranks share. Rank 0 drops a nonce beside ``out``, everybody looks
    for it. -> None when all ranks see it, else the agreed reason (same on every rank)."""
    import os
    import uuid
    box = [uuid.uuid4().hex if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    probe = f"{out}.{box[0]}.probe"
    err = None
    if rank == 0:
        try:
            with open(probe, "w") as f:
                f.write(box[0])
        except OSError as e:
            err = f"rank 0: cannot write beside {out}: {e}"
    dist.barrier(group=group)
    seen = False
    try:
        with open(probe) as f:
            seen = f.read() == box[0]
    except OSError:
        pass
    flags = [None] * world
    dist.all_gather_object(flags, (seen, err), group=group)
    if rank == 0:
        try:
            os.remove(probe)
        except OSError:
            pass
    errs = [e for _, e in flags if e]
    if errs:
        return errs[0]
    blind = [r for r, (ok, _) in enumerate(flags) if not ok]
    return None if not blind else f"ranks {blind} do not see rank 0's files beside {out} (no shared file system)"


def scan_files_to_csv(engine, paths: Sequence[str], chains: Optional[Sequence], out: str, model_name: str = "ThermoMPNN",
                      dataset: str = "custom", pick_best: bool = False, include_cys: bool = False, centrality: bool = False,
                      group=None, k_neighbors: Optional[int] = None, n_threads: int = 0, run_pipeline=None,
                      max_part_bytes: Optional[int] = None, shared_fs: Optional[bool] = None, **pipeline_kw):
    """PDB files -> ONE CSV in the reference's layout (analysis/SSM.py:102-176), every rank FORMATTING its own shard (round 5; until
    round 4 every table went to rank 0, which formatted the whole listing alone: one writer does 7-17 M predictions/s against
    > 100 M/s per GPU, so an 8-GPU scan to CSV ran at the one-GPU rate).

      1. length pre-pass + LPT shards as in ``scan_files``; every rank knows every sequence, hence every protein's number of rows
         and the running index of its first row in the final listing;
      2. each rank runs the parse || forward || write pipeline on its shard; the writer thread formats chunk after chunk with the
         FINAL running indices into memory (``tmpnn_csv_open_mem``; a part file beside ``out`` when the shard's text would exceed
         ``max_part_bytes`` of address space) and records the bytes of text of every protein;
      3. ONE ``all_gather_object`` of those byte counts -> every protein's byte offset in the final file; rank 0 creates ``out``
         (header, ftruncate to the total), every rank writes its proteins' text at their offsets (pwrite from the buffer /
         copy_file_range from the part file).
    The result is byte-identical to the one-rank file (tests: world-2 gloo with a stand-in pipeline on CPU, two ranks on one GPU
    with the real engine). ``run_pipeline``: the function used as ``pipeline.scan_files`` (tests inject a CPU stand-in).
    -> (rows, stats) on every rank (rows = of the whole file).

    Memory: a rank keeps its shard's text (about 411 bytes per prediction for L = 256: every row repeats the sequence) in an anonymous
    mapping up to ``max_part_bytes`` (default ``default_memory_budget()``: a third of MemAvailable / LOCAL_WORLD_SIZE, at most 16 GiB);
    larger shards — or a mapping the kernel refuses — go to a part file beside ``out``.
    File system: every rank opens ``out``, so all ranks must share a coherent file system (one node, or a POSIX-coherent parallel file
    system; NFS gives weak guarantees for unlocked concurrent range writes). ``shared_fs``: None = probe it (rank 0 drops a nonce
    beside ``out``) and fall back to the gather-to-rank-0 writer (``scan_files`` + one writer) when a rank does not see it; False =
    that fallback at once; True = no probe."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from . import native_csv, pipeline
    run_pipeline = run_pipeline or pipeline.scan_files
    n = len(paths)
    chains = list(chains) if chains is not None else [None] * n
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    K = int(k_neighbors or engine.K)
    nt = n_threads or max(1, pipeline.usable_cpus() - 1)
    if world == 1:
        from . import ssm_scan
        return ssm_scan.scan_to_file(engine, paths, chains, out, model_name, dataset, pick_best, include_cys, centrality,
                                     n_threads=nt, **pipeline_kw)
    why = "shared_fs=False" if shared_fs is False else (None if shared_fs else _shared_fs_check(out, rank, world, group))
    if why is not None:                                      # one writer: every table to rank 0 (what rounds 3-4 did for every scan)
        from . import ssm_scan
        r = scan_files(engine, paths, chains, group=group, centrality=centrality, k_neighbors=k_neighbors, run_pipeline=run_pipeline,
                       device_tables=False if run_pipeline is not pipeline.scan_files else None, **pipeline_kw)
        rows, err = 0, None
        if rank == 0:
            try:
                rows = ssm_scan.write_scan_csv(out, r, model_name, dataset, pick_best, include_cys, None, n_threads=nt)
            except Exception as e:       # noqa: BLE001
                err = f"rank 0: {type(e).__name__}: {e}"
        agree_or_raise(err, group)
        box = [rows]
        dist.broadcast_object_list(box, src=0, group=group)
        return int(box[0]), r["stats"]
    if max_part_bytes is None:
        max_part_bytes = default_memory_budget()
    info = _length_prepass(paths, chains, rank, world, group)
    lengths = [info[i][0] for i in range(n)]
    shard = partition_proteins(lengths, world, K)[rank]
    per = 1 if pick_best else (20 if include_cys else 19)
    rows_of = [sum(c != "-" for c in info[i][1]) * per for i in range(n)]
    first_row = np.concatenate([[0], np.cumsum(rows_of)]).astype(np.int64)
    # this rank's text stays in MEMORY (an anonymous, huge-page-backed mapping) until the byte counts are exchanged: a part file beside
    # ``out`` pays the file system's page allocation twice, and that allocation — not the formatting — is what bounds the CSV sink
    # (bench.py: tmpfs_write_ceiling). Address space for an upper bound of the shard's text; beyond ``max_part_bytes`` a part file.
    # (bytes as the writer emits them: UTF-8, quotes doubled + two enclosing ones; the name appears twice in --pick_best rows)
    fixed = _csv_field_bytes(model_name) + _csv_field_bytes(dataset)
    bound = sum(rows_of[i] * (_csv_field_bytes(info[i][1]) + fixed + 2 * _csv_field_bytes(info[i][2].strip(".pdb")) + 96) for i in shard) + (1 << 16)
    in_memory = bound <= max_part_bytes
    part = f"{out}.part{rank}"
    nbytes = {}
    err, stats = None, None
    w = None
    try:
        if in_memory:
            try:
                w = native_csv.CsvWriter(None, native_csv.SCHEMA_SSM, pick_best=pick_best, memory_capacity=bound)
            except (OSError, MemoryError, RuntimeError):     # the kernel refused the mapping: the part file instead
                in_memory = False
        if not in_memory:
            w = native_csv.CsvWriter(part, native_csv.SCHEMA_SSM, header=False, pick_best=pick_best)
        done = [0]

        def sink(ch):
            ids = shard[done[0]:done[0] + ch.n]
            got = w.write_ssm(ch.table, ch.offsets, ch.seq_ptrs, [x.strip(".pdb") for x in ch.names], neighbors=ch.neighbors,
                              model=model_name, dataset=dataset, pick_best=pick_best, include_cys=include_cys, n_threads=nt,
                              first_rows=[int(first_row[i]) for i in ids], want_bytes=True)
            for i, b in zip(ids, got):
                nbytes[i] = int(b)
            done[0] += ch.n

        stats = run_pipeline(engine, [paths[i] for i in shard], [chains[i] for i in shard], sink, centrality=centrality, **pipeline_kw)
        if done[0] != len(shard):
            raise RuntimeError(f"the pipeline delivered {done[0]} of this rank's {len(shard)} proteins")
    except Exception as e:               # noqa: BLE001
        err = f"rank {rank}: {type(e).__name__}: {e}"
    try:
        if w is not None and not in_memory:
            w.close()
        agree_or_raise(err, group)
        parts = [None] * world
        dist.all_gather_object(parts, nbytes, group=group)
        allb = {k: v for p in parts for k, v in p.items()}
        hdr = native_csv.header_text(native_csv.SCHEMA_SSM, pick_best)
        missing = [i for i in range(n) if i not in allb]
        agree_or_raise(f"no rank reported the text of proteins {missing[:8]}" if missing else None, group)   # (same verdict on every rank)
        offs = np.concatenate([[len(hdr)], len(hdr) + np.cumsum([allb[i] for i in range(n)])]).astype(np.int64)
        err = None
        if rank == 0:
            try:
                fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
                try:
                    os.pwrite(fd, hdr, 0)
                    os.ftruncate(fd, int(offs[-1]))
                finally:
                    os.close(fd)
            except OSError as e:
                err = f"rank 0: cannot create {out}: {e}"
        agree_or_raise(err, group)                           # (also the barrier behind which the file exists)
        err = None
        try:
            jobs, pos = [], 0
            for i in shard:                                  # consecutive proteins of the final file are placed as one range
                if jobs and jobs[-1][1] + jobs[-1][2] == int(offs[i]) and jobs[-1][0] + jobs[-1][2] == pos:
                    jobs[-1][2] += allb[i]
                else:
                    jobs.append([pos, int(offs[i]), allb[i]])
                pos += allb[i]
            blk = 16 << 20                                   # (ranges cut into blocks so that every writer thread has work)
            jobs = [[a + k, b + k, min(blk, m - k)] for a, b, m in jobs for k in range(0, m, blk)]
            dst = os.open(out, os.O_WRONLY)
            src = -1 if in_memory else os.open(part, os.O_RDONLY)
            try:
                if in_memory:
                    text = w.memory()
                    view = memoryview(text)

                    def place(j):
                        a, b, m = j
                        while m > 0:
                            k = os.pwrite(dst, view[a:a + m], b)
                            a, b, m = a + k, b + k, m - k
                else:
                    place = lambda j: _copy_range(src, dst, j[0], j[1], j[2])
                with ThreadPoolExecutor(max_workers=max(1, min(nt, 16))) as ex:
                    list(ex.map(place, jobs))
            finally:
                os.close(dst)
                if src >= 0:
                    os.close(src)
        except Exception as e:           # noqa: BLE001 (anything: a rank that raises alone leaves the others in the collective below)
            err = f"rank {rank}: {type(e).__name__}: {e}"
        agree_or_raise(err, group)
    finally:
        if w is not None and in_memory:
            try:
                view = text = None                           # (the mapping goes away with the writer)
                w.close()
            except Exception:                                # noqa: BLE001
                pass
        if not in_memory:
            try:
                os.remove(part)
            except OSError:
                pass
    return int(first_row[-1]), stats


def select_mutations(tables: Sequence[torch.Tensor], triples) -> torch.Tensor:
    """ddG of an explicit mutation list (BASELINE config 4: 200 k listed mutants over 300 proteins).
    ``triples``: integer array [M, 3] of (protein index, 0-based position, amino-acid index into ALPHABET[:20]);
    ``tables``: the per-protein [L_i, 21] tables of ``ssm_scan``. One gather on the device -> [M] float32."""
    import numpy as np
    tr = np.asarray(triples, dtype=np.int64).reshape(-1, 3)
    if tr.shape[0] == 0:
        return torch.zeros(0, dtype=torch.float32, device=tables[0].device if tables else "cpu")
    lens = np.array([t.shape[0] for t in tables], dtype=np.int64)
    if (tr[:, 0] < 0).any() or (tr[:, 0] >= len(tables)).any():
        raise IndexError("mutation list names a protein outside the scanned set")
    if (tr[:, 1] < 0).any() or (tr[:, 1] >= lens[tr[:, 0]]).any() or (tr[:, 2] < 0).any() or (tr[:, 2] >= 20).any():
        raise IndexError("mutation list has a position / amino-acid index out of range")
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    flat = torch.cat([t.reshape(-1) for t in tables])
    idx = torch.as_tensor((starts[tr[:, 0]] + tr[:, 1]) * tables[0].shape[1] + tr[:, 2], device=flat.device)
    return flat[idx]


def init_from_env(one_device: bool = False, backend: Optional[str] = None):
    """Process-group set-up for a ``torchrun`` / ``python -m torch.distributed.run`` launch (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in the environment): one process per GPU, backend "nccl" (= RCCL over xGMI on ROCm).
    ``one_device`` (or TMPNN_ONE_DEVICE=1) puts every rank on cuda:0 with a gloo group — the N>1 code path on a 1-GPU
    box. -> (rank, world, device); a plain single process gets (0, 1, cuda:0) and no group."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    one_device = one_device or os.environ.get("TMPNN_ONE_DEVICE") == "1"
    index = 0 if one_device else int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", index)
    torch.cuda.set_device(index)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or os.environ.get("TMPNN_DIST_BACKEND") or ("gloo" if one_device else "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return rank, world, device
