"""Counterpart of /root/reference/analysis/SSM.py: site-saturation scans over MANY proteins.

The reference runs one protein per forward and appends to a pandas frame cell by cell with one device sync per
mutation (SSM.py:105-147). Here a three-stage pipeline over chunks of files (thermompnn_amd/pipeline.py): native threaded
PDB parsing into pinned staging buffers -> one async H2D copy, one fused ragged forward, one async D2H copy -> native
columnar CSV writer (csrc/tmpnn_csv.cpp; ``--out x.npz`` writes the binary tables instead), with parse(k+1) || forward(k) ||
write(k-1). ``rows_for_protein`` / ``write_csv`` below are the reference-shaped per-row form the native writer is tested
against byte for byte. Post-processing options keep the reference semantics:
  --centrality   'neighbors' column = #CA within 10 A (compute_centrality, SSM.py:129-132,144-145)
  --pick_best    keep one row per position carrying best_AA = argmin ddG (retrieve_best_mutants, SSM.py:32-42,153-162)
  --include_cys  otherwise mutations to C are excluded (from the best-pick, or dropped from the listing; :164-166)
Launched under ``python -m torch.distributed.run --nproc-per-node N -m thermompnn_amd.ssm_scan ...`` the proteins
are sharded over the N GPUs (dist.scan_files: LPT partition, the same pipeline per rank, ONE gather of the tables to rank 0,
which writes); a plain ``python -m thermompnn_amd.ssm_scan`` streams chunk after chunk into the output file.
  --mutations FILE   CSV with columns pdb,position,mutation (0-based position into the parsed sequence): only the listed
                     mutants are written (BASELINE config 4: an explicit list over many proteins)
"""
from __future__ import annotations

import argparse
import csv
from typing import List, Optional, Sequence

import numpy as np
import torch

from .datasets import ALPHABET

AA20 = ALPHABET[:-1]
COLUMNS = ["WT Seq", "Model", "Dataset", "ddG_pred", "position", "wildtype", "mutation", "neighbors", "best_AA", "pdb"]
# --pick_best frames carry one more column that the reference adds for drop_duplicates and never drops (SSM.py:161-162)
COLUMNS_PICK_BEST = COLUMNS + ["dupe_detector"]


def retrieve_best_mutants(ddg_table: np.ndarray, allow_cys: bool = True) -> List[str]:
    """Best (lowest ddG) mutant letter at each position of a [L, 20] table; first minimum wins like
    ``idxmin`` (SSM.py:32-42). Cysteine is excluded unless ``allow_cys``."""
    t = np.array(ddg_table[:, :20], dtype=np.float64, copy=True)
    t[np.isnan(t)] = np.inf                                  # idxmin skips missing values
    if not allow_cys:
        t[:, AA20.index("C")] = np.inf
    return [AA20[i] for i in np.argmin(t, axis=1)]


def scan_proteins(engine, proteins: Sequence[Optional[dict]], centrality: bool = False, chunk_residues: int = 1 << 18, group=None,
                  lengths: Optional[Sequence[int]] = None):
    """proteins: dicts from native_pdb.parse_pdb (with ``lengths``: only this rank's shard, ``None`` elsewhere — dist.parse_sharded).
    -> (list of [L,21] ddG arrays, list of neighbour-count arrays or None).
    Sharded over ``group``'s ranks when torch.distributed is initialised (every rank gets every table back); processed in
    ragged chunks of at most ``chunk_residues`` residues. Non-finite results never reach the caller: the engine reruns
    an overflowing f16x2 batch in bf16x3 or raises (Engine.ssm_forward)."""
    from .dist import ssm_scan
    res = ssm_scan(engine, proteins, group=group, centrality=centrality, chunk_residues=chunk_residues, lengths=lengths)
    tables, cen = res if centrality else (res, None)
    flat = torch.cat([t.reshape(-1) for t in tables]).cpu().numpy() if tables else np.zeros(0, np.float32)   # one D2H copy
    out, pos = [], 0
    for t in tables:
        n = t.shape[0] * 21
        out.append(flat[pos:pos + n].reshape(-1, 21))
        pos += n
    neigh = [c.cpu().numpy() for c in cen] if cen is not None else None
    return out, neigh


def read_mutation_list(path: str, proteins: Sequence[dict]):
    """CSV with columns pdb,position,mutation[,wildtype] -> int64 [M,3] triples (protein index, position, aa index).
    ``pdb`` matches the parsed structure's name; a stated wildtype must agree with the structure's residue (the
    reference asserts this too, custom_inference.py:86-87)."""
    by_name = {p["name"]: i for i, p in enumerate(proteins)}          # (only "name" and "seq" of every entry are used)
    out = []
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            if r["pdb"] not in by_name:
                raise KeyError(f"{path}: unknown pdb {r['pdb']!r}")
            i, pos, mut = by_name[r["pdb"]], int(r["position"]), r["mutation"].strip()
            seq = proteins[i]["seq"]
            assert 0 <= pos < len(seq), f"{r['pdb']}: position {pos} outside the {len(seq)}-residue sequence"
            assert mut in AA20, f"{r['pdb']}: unknown amino acid {mut!r}"
            wt = (r.get("wildtype") or "").strip()
            assert not wt or wt == seq[pos], f"{r['pdb']}: wildtype {wt}{pos} does not match the structure ({seq[pos]})"
            out.append((i, pos, AA20.index(mut)))
    return np.asarray(out, dtype=np.int64).reshape(-1, 3)


def rows_for_protein(p: dict, table: np.ndarray, neighbors, model_name: str, dataset: str, pick_best: bool,
                     include_cys: bool):
    """Row dicts for one protein in the reference's column layout."""
    seq = p["seq"]
    name = p["name"].strip(".pdb")
    best = retrieve_best_mutants(table, allow_cys=include_cys) if pick_best else None
    rows = []
    for pos, wt in enumerate(seq):
        if wt == "-":
            continue
        for a, mut in enumerate(AA20):
            if pick_best and a > 0:
                break                                            # one row per position (drop_duplicates keep='first')
            if not pick_best and not include_cys and mut == "C":
                continue
            v = float(table[pos, a])
            rows.append({"WT Seq": p.get("wt", seq), "Model": model_name, "Dataset": dataset, "ddG_pred": "" if v != v else v,   # (NaN = a missing cell)
                         "position": pos, "wildtype": wt, "mutation": mut,
                         "neighbors": int(neighbors[pos]) if neighbors is not None else "",
                         "best_AA": best[pos] if best is not None else "", "pdb": name})
            if pick_best:
                rows[-1]["dupe_detector"] = name + str(pos)
    return rows


def write_csv(rows, path: str, pick_best: Optional[bool] = None) -> None:
    """``pick_best`` (default: whether the rows carry the column) selects the header with ``dupe_detector``."""
    if pick_best is None:
        pick_best = bool(rows) and "dupe_detector" in rows[0]
    cols = COLUMNS_PICK_BEST if pick_best else COLUMNS
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh, lineterminator="\n")           # pandas' to_csv line ends (examples/ThermoMPNN_inference_2OCJ.csv)
        w.writerow([""] + cols)
        for i, r in enumerate(rows):
            w.writerow([i] + [r[c] for c in cols])


def write_scan_csv(path: str, res: dict, model_name: str, dataset: str, pick_best: bool, include_cys: bool,
                   triples=None, n_threads: int = 0) -> int:
    """The native columnar writer on a finished scan (``dist.scan_files`` result on rank 0) -> number of rows.
    Byte-identical to ``write_csv(rows_for_protein(...))`` (tests/test_host.py compares them)."""
    from . import native_csv
    names = [n.strip(".pdb") for n in res["names"]]               # SSM.py:139 (a character-set strip)
    with native_csv.CsvWriter(path, native_csv.SCHEMA_SSM) as w:
        if triples is not None:
            w.write_listed(res["table"], res["offsets"], res["seqs"], names, triples, neighbors=res["neighbors"],
                           model=model_name, dataset=dataset)
        else:
            w.write_ssm(res["table"], res["offsets"], res["seqs"], names, neighbors=res["neighbors"], model=model_name,
                        dataset=dataset, pick_best=pick_best, include_cys=include_cys, n_threads=n_threads)
    return w.rows


def write_scan_npz(path: str, res: dict) -> None:
    """Binary result: ddg float32 [T, 21] (column a = mutation to ALPHABET[a], column 20 = 'X'), offsets int64 [n+1] (rows of
    protein i = offsets[i]:offsets[i+1]), names, seqs ('-' = no residue: that row is not a prediction), neighbors (optional).
    No text formatting at all: the format for downstream code that wants the table, not a spreadsheet."""
    extra = {} if res.get("neighbors") is None else {"neighbors": res["neighbors"]}
    with open(path, "wb") as fh:                                  # (np.savez would append ".npz" to a bare name)
        np.savez(fh, ddg=res["table"], offsets=np.asarray(res["offsets"], dtype=np.int64), names=np.array(res["names"]),
                 seqs=np.array(res["seqs"]), **extra)


def scan_to_file(engine, paths: Sequence[str], chains: Sequence, out: str, model_name: str = "ThermoMPNN",
                 dataset: str = "custom", pick_best: bool = False, include_cys: bool = False, centrality: bool = False,
                 n_threads: int = 0, **pipeline_kw):
    """Single-GPU streaming form: PDB files -> ``out`` (``.npz``: binary tables; anything else: the reference's CSV layout)
    with parse(k+1) || forward(k) || write(k-1) — a chunk's rows are formatted and written while the GPU works on the next
    chunk, and nothing but the current chunks is held in memory. -> (rows or residues written, pipeline.ScanStats)."""
    from . import native_csv, pipeline
    if out.endswith(".npz"):
        acc = dict(t=[], nb=[], lens=[], seqs=[], names=[])

        def sink(ch):
            acc["t"].append(ch.table.copy())
            if ch.neighbors is not None:
                acc["nb"].append(ch.neighbors.copy())
            acc["lens"].extend(int(x) for x in np.diff(ch.offsets))
            acc["seqs"].extend(ch.seqs())
            acc["names"].extend(ch.names)

        stats = pipeline.scan_files(engine, paths, chains, sink, centrality=centrality, **pipeline_kw)
        t0 = _now()
        res = dict(table=np.concatenate(acc["t"]) if acc["t"] else np.zeros((0, 21), np.float32),
                   neighbors=np.concatenate(acc["nb"]) if acc["nb"] else None,
                   offsets=np.concatenate([[0], np.cumsum(acc["lens"])]), seqs=acc["seqs"], names=acc["names"])
        write_scan_npz(out, res)
        stats.sink_s += _now() - t0
        stats.wall_s += _now() - t0
        return int(res["table"].shape[0]), stats
    w = native_csv.CsvWriter(out, native_csv.SCHEMA_SSM)
    nt = n_threads or max(1, pipeline.usable_cpus() - 1)
    try:
        def sink(ch):
            w.write_ssm(ch.table, ch.offsets, ch.seq_ptrs, [n.strip(".pdb") for n in ch.names], neighbors=ch.neighbors,
                        model=model_name, dataset=dataset, pick_best=pick_best, include_cys=include_cys, n_threads=nt)

        stats = pipeline.scan_files(engine, paths, chains, sink, centrality=centrality, **pipeline_kw)
    finally:
        w.close()
    return w.rows, stats


def scan_datasets(models: dict, datasets: dict, pick_best: bool = False, include_cys: bool = False, centrality: bool = False,
                  out_dir: str = ".", max_batches: Optional[int] = None, n_threads: int = 0) -> List[str]:
    """The driver loop of the reference's SSM script as a function (analysis/SSM.py:96-176): every model of ``models``
    (``TransferModel``, ``ProteinMPNNBaseline`` — anything with ``ssm_table(pdb)``) over every dataset of ``datasets``
    (``ddgBenchDataset``, ``FireProtDataset``, ... — iterables of ``(pdb, mutations)`` with a ``wt_seqs`` dictionary), one file
    ``<model>_<dataset>_SSM_preds.csv`` per pair in ``out_dir`` with the reference's columns and quirks: 'WT Seq' is the
    DATASET's wild-type string (``wt_seqs[name]``, ``name + '.pdb'`` for Megascale sets, :145-149), 'pdb' is the
    character-set-stripped structure name (:133), ``--pick_best`` / ``--include_cys`` / ``--centrality`` as there. One forward
    per protein instead of one ``model(pdb, mutations)`` call plus 20 L ``.item()`` syncs; rows by the native writer.
    ``--centrality`` needs single-chain entries: the reference indexes the FIRST chain's neighbour counts with positions that run
    over all chains (:129-143), so its own loop ends in an IndexError at the first position of a second chain; here the entry is
    refused up front with a message that says so (a restriction, not a silent difference).
    -> the files written."""
    import os
    from . import native_csv
    from .thermompnn_benchmarking import compute_centrality
    written = []
    for name, model in models.items():
        model = model.eval()
        for dataset_name, dataset in datasets.items():
            tables, seqs, cells, names, neigh = [], [], [], [], []
            for i, (mut_pdb, _listed) in enumerate(dataset):
                p = mut_pdb[0]
                with torch.no_grad():
                    tables.append(model.ssm_table(mut_pdb))
                seqs.append(p["seq"])
                key = p["name"] if "Megascale" not in dataset_name else p["name"] + ".pdb"
                wt = dataset.wt_seqs[key]
                cells.append("" if wt is None else str(wt))              # pandas writes a missing value as an empty cell
                names.append(p["name"].strip(".pdb"))
                if centrality:
                    coord_chain = [c for c in p.keys() if "coords" in c][0]
                    neigh.append(compute_centrality(p[coord_chain], basis_atom="CA", backup_atom="C", chain=coord_chain[-1],
                                                    radius=10.0, device=tables[-1].device).to(torch.int32))
                    if neigh[-1].numel() != tables[-1].shape[0]:      # the reference indexes the FIRST chain's counts by position (:131-134)
                        raise ValueError(f"{p['name']}: --centrality needs single-chain dataset entries "
                                         f"({neigh[-1].numel()} residues in chain {coord_chain[-1]}, {tables[-1].shape[0]} parsed)")
                if max_batches is not None and i >= max_batches:
                    break
            path = os.path.join(out_dir, name + "_" + dataset_name + "_SSM_preds.csv")
            lens = [int(t.shape[0]) for t in tables]
            flat = torch.cat(tables).cpu().numpy() if tables else np.zeros((0, 21), np.float32)     # one D2H copy per dataset
            nb = torch.cat(neigh).cpu().numpy() if centrality and neigh else None
            with native_csv.CsvWriter(path, native_csv.SCHEMA_SSM) as w:
                w.write_ssm(flat, np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), seqs, names, neighbors=nb, model=name,
                            dataset=dataset_name, pick_best=pick_best, include_cys=include_cys, n_threads=n_threads, wt_cells=cells)
            written.append(path)
    return written


def _now() -> float:
    import time
    return time.perf_counter()


def main(argv=None):
    ap = argparse.ArgumentParser(description="ThermoMPNN SSM over many PDB files on MI355X")
    ap.add_argument("pdbs", nargs="+", help="PDB files (the first chain is used unless --chain is given)")
    ap.add_argument("--chain", default="A")
    ap.add_argument("--model_path", default="")
    ap.add_argument("--thermompnn_dir", default=".")
    ap.add_argument("--synthetic_weights", type=int, default=None)
    ap.add_argument("--dataset_name", default="custom")
    ap.add_argument("--out", default="ThermoMPNN_custom_SSM_preds.csv",
                    help="output file; a name ending in .npz gets the binary tables (ddg [T,21], offsets, names, seqs) instead of CSV")
    ap.add_argument("--pick_best", action="store_true", default=False, help="Keep only the BEST mutation at each position")
    ap.add_argument("--include_cys", action="store_true", default=False, help="Include cysteine as potential mutation option.")
    ap.add_argument("--centrality", action="store_true", default=False, help="Calculate centrality value for each residue (# neighbors).")
    ap.add_argument("--mutations", default="", help="CSV (pdb,position,mutation): write only these mutants")
    ap.add_argument("--precision", default=None, choices=["f16x2", "bf16x3", "fp32"])
    ap.add_argument("--chunk_files", type=int, default=96, help="files per pipeline chunk (parse || forward || write overlap)")
    ap.add_argument("--device_tables", action="store_true", default=False,
                    help="multi-rank binary / listed output: keep every rank's tables on its GPU until the gather (automatic over "
                         "RCCL; this forces it for other backends)")
    ap.add_argument("--allow_pickle", action="store_true", default=False,
                    help="read --model_path with the unrestricted pickle loader (it can execute code from the file); the default "
                         "restricted loader already reads Lightning checkpoints such as thermoMPNN_default.pt")
    args = ap.parse_args(argv)

    from . import dist as tdist
    from .custom_inference import load_model
    rank, world, device = tdist.init_from_env()
    model = load_model(args.model_path, args.thermompnn_dir, args.synthetic_weights, device=device, precision=args.precision,
                       allow_pickle=args.allow_pickle or None)
    engine = model.engine()
    chains = [args.chain] * len(args.pdbs)
    with torch.cuda.device(engine.device):
        if world == 1 and not args.mutations:
            # one GPU: stream chunks straight into the output while the next ones are parsed and computed
            n_rows, stats = scan_to_file(engine, args.pdbs, chains, args.out, "ThermoMPNN", args.dataset_name, args.pick_best,
                                         args.include_cys, args.centrality, chunk_files=args.chunk_files)
            n_prot = stats.files
        elif not args.mutations and not args.out.endswith(".npz"):
            # N GPUs -> CSV: every rank formats its own LPT shard with the final running indices, the ranks exchange byte counts and
            # each places its text in the one output file (dist.scan_files_to_csv): the writer scales with the ranks
            n_rows, stats = tdist.scan_files_to_csv(engine, args.pdbs, chains, args.out, "ThermoMPNN", args.dataset_name, args.pick_best,
                                                    args.include_cys, args.centrality, chunk_files=args.chunk_files)
            n_prot = len(args.pdbs)
        else:
            # N GPUs -> binary tables / an explicit mutation list: every rank runs the pipeline on its LPT shard, ONE gather to rank
            # 0 (over RCCL straight from the device buffer the forwards wrote), which writes
            res = tdist.scan_files(engine, args.pdbs, chains, centrality=args.centrality, chunk_files=args.chunk_files,
                                   device_tables=True if args.device_tables else None)
            n_prot, n_rows = len(res["names"]), 0
            err = None
            if rank == 0:
                try:
                    if args.out.endswith(".npz"):
                        write_scan_npz(args.out, res)
                        n_rows = int(res["table"].shape[0])
                    else:
                        tri = read_mutation_list(args.mutations, [{"seq": s_, "name": n_} for s_, n_ in zip(res["seqs"], res["names"])]) \
                            if args.mutations else None
                        n_rows = write_scan_csv(args.out, res, "ThermoMPNN", args.dataset_name, args.pick_best, args.include_cys, tri)
                except Exception as e:       # noqa: BLE001 - the other ranks must not wait at the barrier for a writer that died
                    err = f"rank 0: {type(e).__name__}: {e}"
            tdist.agree_or_raise(err)
    if rank == 0:
        print(f"Saved {n_rows} rows for {n_prot} proteins to {args.out} ({world} rank(s))")
    if world > 1:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()
    return args.out


if __name__ == "__main__":
    main()
