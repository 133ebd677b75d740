"""N>1 path on CPU: world_size 2 / 4 / 8 gloo processes exercise the sharding, the ragged gathers, the sharded CSV writer and BASELINE
config 4's listed-mutation scan (no GPU compute)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from thermompnn_amd.dist import all_gather_tables, partition_proteins, scan_sharded


def fake_table(pid, L):
    """Deterministic stand-in for a protein's ddG table (the collective logic does not care about the values)."""
    r = np.random.default_rng(1000 + pid)
    return torch.tensor(r.normal(size=(L, 21)), dtype=torch.float32)


def _worker(rank, world, port, lengths, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seen = []

        def compute(ids):
            seen.extend(ids)
            return torch.cat([fake_table(i, lengths[i]) for i in ids]) if ids else torch.zeros((0, 21))

        tables = scan_sharded(lengths, compute)
        ok = all(torch.equal(t, fake_table(i, lengths[i])) for i, t in enumerate(tables))
        # ragged gather incl. an empty shard
        rows = [3, 0] if world == 2 else [(r * 5) % 4 for r in range(world)]      # uneven, with empty shards
        got = all_gather_tables(torch.full((rows[rank], 2), float(rank)), rows)
        ok = ok and [g.shape[0] for g in got] == rows and all((g == r).all() for r, g in enumerate(got))
        # gather-to-root (what the CLI uses: only rank 0 writes), same ragged case incl. the empty shard
        from thermompnn_amd.dist import gather_tables_root
        root = gather_tables_root(torch.full((rows[rank], 2), float(rank)), rows)
        if rank == 0:
            ok = ok and [g.shape[0] for g in root] == rows and all((g == r).all() for r, g in enumerate(root))
        else:
            ok = ok and root is None
        t_root = scan_sharded(lengths, compute, gather="root")
        mine = partition_proteins(lengths, world)[rank]
        if rank == 0:
            ok = ok and all(torch.equal(t, fake_table(i, lengths[i])) for i, t in enumerate(t_root))
        else:
            ok = ok and all((t is not None) == (i in mine) for i, t in enumerate(t_root))
        seen[:] = seen[:len(seen) // 2]                    # (the second scan computed the same shard again)
        q.put((rank, ok, sorted(seen)))
    finally:
        dist.destroy_process_group()


def _parse_worker(rank, world, port, paths, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from thermompnn_amd import native_pdb
        from thermompnn_amd.dist import parse_sharded
        read = []

        def parse(ps, chains):
            read.extend(ps)
            return native_pdb.parse_pdbs(ps, chains)

        prots, lengths, seqs, names = parse_sharded(paths, ["A"] * len(paths), parse=parse)
        q.put((rank, read, lengths, [p is not None for p in prots], seqs, names,
               [None if p is None else (len(p["S"]), p["seq"]) for p in prots]))
    finally:
        dist.destroy_process_group()


def test_sharded_parse_reads_each_file_where_it_is_needed(tmp_path):
    """dist.parse_sharded on two ranks: every rank learns every length / sequence / name, holds exactly its own LPT shard, and
    no rank reads all the files (a strided pre-pass + the rest of its shard)."""
    import shutil
    from conftest import GOLDEN
    from thermompnn_amd import native_pdb
    from thermompnn_amd.dist import parse_sharded
    src = [os.path.join(GOLDEN, "2OCJ.pdb"), os.path.join(GOLDEN, "2OCJ_gap_chainA.pdb")]
    paths = []
    for k in range(6):
        dst = str(tmp_path / f"p{k}.pdb")
        shutil.copy(src[k % 2], dst)
        paths.append(dst)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_parse_worker, args=(r, 2, port, paths, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = native_pdb.parse_pdbs(paths, ["A"] * 6)
    shards = partition_proteins([len(p["S"]) for p in full], 2)
    for rank, read, lengths, have, seqs, names, mine in results:
        assert lengths == [len(p["S"]) for p in full] and seqs == [p["seq"] for p in full] and names == [p["name"] for p in full]
        assert [i for i, h in enumerate(have) if h] == shards[rank]
        assert all(m == (len(full[i]["S"]), full[i]["seq"]) for i, m in enumerate(mine) if m is not None)
        assert len(read) == len(set(read)) < 6 and set(paths[rank::2]) <= set(read)
    # a world of one parses everything once
    prots, lengths, seqs, names = parse_sharded(paths, ["A"] * 6)
    assert all(p is not None for p in prots) and lengths == [len(p["S"]) for p in full]


def _bad_file_worker(rank, world, port, paths, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from thermompnn_amd.dist import parse_sharded
        try:
            parse_sharded(paths, ["A"] * len(paths))
            q.put((rank, "no error"))
        except RuntimeError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_sharded_parse_fails_on_every_rank_when_one_file_is_corrupt(tmp_path):
    """a malformed file is read by ONE rank only (strided pre-pass); the error must reach every rank through the
    same collective instead of leaving the others blocked until the backend's timeout."""
    import shutil
    from conftest import GOLDEN
    paths = []
    for k in range(4):
        dst = str(tmp_path / f"p{k}.pdb")
        shutil.copy(os.path.join(GOLDEN, "2OCJ.pdb"), dst)
        paths.append(dst)
    (tmp_path / "p1.pdb").write_text("ATOM      1  N   ALA A   1      xx.000   0.000   0.000\n")      # rank 1's stride
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bad_file_worker, args=(r, 2, port, paths, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all("rank 1" in msg and "malformed" in msg for _, msg in results), results


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_scan_gloo(world):
    """R in {2, 4, 8} (R = 1: test_single_process_scan_needs_no_group). 13 proteins over 8 ranks: uneven LPT shards."""
    lengths = [int(x) for x in np.random.default_rng(2).integers(40, 73, size=11)] + [300, 5]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lengths, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results)
    seen = sorted(i for _, _, ids in results for i in ids)
    assert seen == list(range(len(lengths)))                      # every protein computed exactly once
    shards = partition_proteins(lengths, world)
    assert [sorted(ids) for _, _, ids in sorted(results)] == shards


def test_csv_shard_bounds_and_memory_budget(monkeypatch):
    """the memory form of the sharded CSV writer is bounded by the host's memory (a third of MemAvailable per local rank, at
    most 16 GiB, never below 64 MiB), and the text bound counts the bytes the native writer emits: UTF-8, doubled quotes, two enclosing
    quotes — a name of quotes or non-ASCII letters must not overflow a buffer sized by its character count."""
    from thermompnn_amd import dist as tdist
    one = tdist.default_memory_budget(1)
    assert 64 << 20 <= one <= 16 << 30
    assert tdist.default_memory_budget(8) <= one
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    assert tdist.default_memory_budget() == tdist.default_memory_budget(4)
    assert tdist._csv_field_bytes("abc") == 5
    assert tdist._csv_field_bytes('a"b"') == 4 + 2 + 2
    assert tdist._csv_field_bytes("prot\u00e9ine") == len("prot\u00e9ine".encode()) + 2 == 11


def test_partition_is_balanced_and_deterministic():
    lengths = [int(x) for x in np.random.default_rng(1).integers(64, 513, size=1024)]   # BASELINE config 3
    for world in (1, 2, 4, 8):
        shards = partition_proteins(lengths, world)
        assert sorted(i for s in shards for i in s) == list(range(1024))
        load = [sum(lengths[i] * min(48, lengths[i]) for i in s) for s in shards]
        assert max(load) - min(load) <= 512 * 48                  # LPT: within one largest item
        assert shards == partition_proteins(lengths, world)
    short = partition_proteins([10, 20, 30], 8)                   # more ranks than proteins
    assert sum(len(s) for s in short) == 3 and len(short) == 8


def test_single_process_scan_needs_no_group():
    lengths = [7, 3, 9]
    tables = scan_sharded(lengths, lambda ids: torch.cat([fake_table(i, lengths[i]) for i in ids]))
    assert all(torch.equal(t, fake_table(i, L)) for i, (t, L) in enumerate(zip(tables, lengths)))


# ---- the sharded result writer (round 5): every rank formats its shard, byte offsets are exchanged, ONE file comes out ----------
def _seq_table(seq, with_neighbors):
    """A deterministic stand-in for a protein's [L, 21] ddG table (and neighbour counts), a function of the sequence only."""
    r = np.random.default_rng(sum(ord(c) * (k + 1) for k, c in enumerate(seq)))
    t = r.normal(scale=2.0, size=(len(seq), 21)).astype(np.float32)
    return t, (r.integers(0, 40, size=len(seq)).astype(np.int32) if with_neighbors else None)


def _standin_pipeline(engine, paths, chains, sink, centrality=False, chunk_files=2, **_):
    """What pipeline.scan_files does, without a GPU: parse (native, CPU), 'forward' = _seq_table, chunks handed to the sink in
    file order."""
    from thermompnn_amd import native_pdb, pipeline
    pos, index = 0, 0
    while pos < len(paths):
        m = min(chunk_files, len(paths) - pos)
        prots = native_pdb.parse_pdbs(list(paths[pos:pos + m]), list(chains[pos:pos + m]))
        tabs = [_seq_table(p["seq"], centrality) for p in prots]
        off = np.concatenate([[0], np.cumsum([len(p["seq"]) for p in prots])]).astype(np.int32)
        sink(pipeline.Chunk(index=index, first=pos, n=m, T=int(off[-1]), offsets=off, table=np.concatenate([t for t, _ in tabs]),
                            neighbors=np.concatenate([nb for _, nb in tabs]) if centrality else None,
                            seq_ptrs=[p["seq"] for p in prots], names=[os.path.basename(x)[:-4] for x in paths[pos:pos + m]]))
        pos += m
        index += 1
    return pipeline.ScanStats(files=len(paths))


class _NoEngine:
    K = 48


def _csv_worker(rank, world, port, paths, out, pick, cen, q, max_part_bytes=None, shared_fs=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from thermompnn_amd.dist import scan_files_to_csv
        rows, _ = scan_files_to_csv(_NoEngine(), paths, ["A"] * len(paths), out, "ThermoMPNN", "my set", pick_best=pick, include_cys=not pick,
                                    centrality=cen, n_threads=2, run_pipeline=_standin_pipeline, chunk_files=2, max_part_bytes=max_part_bytes,
                                    shared_fs=shared_fs)
        q.put((rank, rows, os.path.exists(f"{out}.part{rank}")))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_files,pick,cen,in_memory,world,shared_fs", [
    (7, False, True, True, 2, None), (7, True, False, True, 2, None), (1, False, False, True, 2, None), (5, False, False, False, 2, None),
    (5, False, True, True, 8, None), (5, True, False, True, 8, None), (11, False, False, False, 8, None), (7, False, True, True, 4, None),
    (7, False, True, True, 2, False), (5, True, False, True, 4, False), (-5, True, False, True, 2, None)])
def test_sharded_csv_is_byte_identical_to_the_one_writer_file(tmp_path, n_files, pick, cen, in_memory, world, shared_fs):
    """dist.scan_files_to_csv on 2 / 4 / 8 gloo ranks (a stand-in pipeline on CPU): the one output file equals what ONE writer makes
    of the same tables in file order — running indices, --pick_best's dupe_detector column, neighbour counts — including the cases
    where ranks' shards are empty (one file on two ranks; five files on eight ranks: three empty shards). A rank keeps its text in
    memory until the byte counts are exchanged; ``in_memory=False`` forces the part-file form used for very large shards (gone
    afterwards). ``shared_fs=False``: the gather-to-rank-0 writer that replaces the sharded one on a file system the ranks do not
    share (None probes: the temporary directory is shared, so the sharded writer runs). ``n_files < 0``: file names full of quotes and
    non-ASCII letters (twice per row with ``--pick_best``): the text bound of the memory form counts encoded, quoted bytes."""
    import shutil
    from conftest import GOLDEN
    from thermompnn_amd import native_csv, native_pdb
    src = [os.path.join(GOLDEN, "2OCJ.pdb"), os.path.join(GOLDEN, "2OCJ_gap_chainA.pdb")]
    paths = []
    odd_names, n_files = n_files < 0, abs(n_files)
    for k in range(n_files):
        dst = str(tmp_path / (f'x"q""\u00e9\u00fc\u4e2d{k}.pdb' if odd_names else f"prot{k}.pdb"))
        shutil.copy(src[k % 2], dst)
        paths.append(dst)
    out = str(tmp_path / "sharded.csv")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_csv_worker, args=(r, world, port, paths, out, pick, cen, q, None if in_memory else 0, shared_fs)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".probe")]       # the shared-file-system nonce is gone
    prots = native_pdb.parse_pdbs(paths, ["A"] * n_files)
    tabs = [_seq_table(p["seq"], cen) for p in prots]
    off = np.concatenate([[0], np.cumsum([len(p["seq"]) for p in prots])]).astype(np.int32)
    with native_csv.CsvWriter(str(tmp_path / "one.csv")) as w:
        w.write_ssm(np.concatenate([t for t, _ in tabs]), off, [p["seq"] for p in prots], [os.path.basename(x)[:-4].strip(".pdb") for x in paths],
                    neighbors=np.concatenate([nb for _, nb in tabs]) if cen else None, dataset="my set", pick_best=pick,
                    include_cys=not pick, n_threads=2)
    want = (tmp_path / "one.csv").read_bytes()
    assert (tmp_path / "sharded.csv").read_bytes() == want
    assert all(rows == w.rows and not part_left for _, rows, part_left in results)


# ---- BASELINE config 4 at the 8-rank scale (300 proteins, 200 000 listed mutants), CPU stand-in for the forward -----------------
def _config4(n=300, m=200_000):
    """BASELINE config 4 as SURVEY 8(d) states it: L ~ UniformInt[40, 72], triples sampled without replacement from the 20 L tables,
    default_rng(2)."""
    r = np.random.default_rng(2)
    lengths = [int(x) for x in r.integers(40, 73, size=n)]
    starts = np.concatenate([[0], np.cumsum(lengths)])
    flat = r.choice(int(starts[-1]) * 20, size=m, replace=False)
    res, aa = flat // 20, flat % 20
    prot = np.searchsorted(starts, res, side="right") - 1
    return lengths, np.stack([prot, res - starts[prot], aa], axis=1)


def _config4_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from thermompnn_amd.dist import select_mutations
        lengths, triples = _config4()
        mine = []

        def compute(ids):
            mine.extend(ids)
            return torch.cat([fake_table(i, lengths[i]) for i in ids]) if ids else torch.zeros((0, 21))

        tables = scan_sharded(lengths, compute)                         # all-gather: every rank holds every table
        got = select_mutations(tables, triples)
        root = scan_sharded(lengths, compute, gather="root")           # what the CLI uses: rank 0 only
        got_root = select_mutations(root, triples) if rank == 0 else None
        q.put((rank, got.numpy(), None if got_root is None else got_root.numpy(), sorted(mine[:len(mine) // 2])))
    finally:
        dist.destroy_process_group()


def test_config4_listed_mutations_world8_equals_world1():
    """300 proteins / 200 000 (protein, position, amino acid) triples through scan_sharded + select_mutations on 8 gloo ranks: every
    rank's listed ddG vector — and rank 0's after the gather-to-root form — equals the one-process result bit for bit, every protein
    is computed by exactly one rank, and the 8 LPT shards are balanced to within one protein."""
    from thermompnn_amd.dist import select_mutations
    lengths, triples = _config4()
    assert triples.shape == (200_000, 3) and len({tuple(t) for t in triples[:5000]}) == 5000
    want = select_mutations([fake_table(i, L) for i, L in enumerate(lengths)], triples).numpy()
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_config4_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=240) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, got_root, _ in results:
        assert np.array_equal(got, want)
        assert (got_root is None) == (rank != 0) and (got_root is None or np.array_equal(got_root, want))
    shards = partition_proteins(lengths, world)
    assert [ids for _, _, _, ids in results] == shards
    load = [sum(lengths[i] * min(48, lengths[i]) for i in s) for s in shards]
    assert max(load) - min(load) <= 72 * 48
