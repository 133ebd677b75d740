"""Counterpart of /root/reference/analysis/SSM.py: site-saturation scans over MANY proteins.

The reference runs one protein per forward and appends to a pandas frame cell by cell with one device sync per
mutation (SSM.py:105-147). Here: native threaded PDB parsing -> ONE ragged batch per chunk -> one fused forward ->
one device-to-host copy -> columnar CSV writer. Post-processing options keep the reference semantics:
  --centrality   'neighbors' column = #CA within 10 A (compute_centrality, SSM.py:129-132,144-145)
  --pick_best    keep one row per position carrying best_AA = argmin ddG (retrieve_best_mutants, SSM.py:32-42,153-162)
  --include_cys  otherwise mutations to C are excluded (from the best-pick, or dropped from the listing; :164-166)
Launched under ``python -m torch.distributed.run --nproc-per-node N -m thermompnn_amd.ssm_scan ...`` the proteins
are sharded over the N GPUs (dist.ssm_scan: LPT partition, one RCCL all-gather of the tables) and rank 0 writes the
CSV; a plain ``python -m thermompnn_amd.ssm_scan`` is the same code with a world of one.
  --mutations FILE   CSV with columns pdb,position,mutation (0-based position into the parsed sequence): only the listed
                     mutants are written (BASELINE config 4: an explicit list over many proteins)
"""
from __future__ import annotations

import argparse
import csv
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import native_pdb
from .datasets import ALPHABET

AA20 = ALPHABET[:-1]
COLUMNS = ["WT Seq", "Model", "Dataset", "ddG_pred", "position", "wildtype", "mutation", "neighbors", "best_AA", "pdb"]


def retrieve_best_mutants(ddg_table: np.ndarray, allow_cys: bool = True) -> List[str]:
    """Best (lowest ddG) mutant letter at each position of a [L, 20] table; first minimum wins like
    ``idxmin`` (SSM.py:32-42). Cysteine is excluded unless ``allow_cys``."""
    t = np.array(ddg_table[:, :20], dtype=np.float64, copy=True)
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
            rows.append({"WT Seq": seq, "Model": model_name, "Dataset": dataset, "ddG_pred": float(table[pos, a]),
                         "position": pos, "wildtype": wt, "mutation": mut,
                         "neighbors": int(neighbors[pos]) if neighbors is not None else "",
                         "best_AA": best[pos] if best is not None else "", "pdb": name})
    return rows


def write_csv(rows, path: str) -> None:
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow([""] + COLUMNS)
        for i, r in enumerate(rows):
            w.writerow([i] + [r[c] for c in COLUMNS])


def main(argv=None):
    ap = argparse.ArgumentParser(description="ThermoMPNN SSM over many PDB files on MI355X")
    ap.add_argument("pdbs", nargs="+", help="PDB files (the first chain is used unless --chain is given)")
    ap.add_argument("--chain", default="A")
    ap.add_argument("--model_path", default="")
    ap.add_argument("--thermompnn_dir", default=".")
    ap.add_argument("--synthetic_weights", type=int, default=None)
    ap.add_argument("--dataset_name", default="custom")
    ap.add_argument("--out", default="ThermoMPNN_custom_SSM_preds.csv")
    ap.add_argument("--pick_best", action="store_true", default=False, help="Keep only the BEST mutation at each position")
    ap.add_argument("--include_cys", action="store_true", default=False, help="Include cysteine as potential mutation option.")
    ap.add_argument("--centrality", action="store_true", default=False, help="Calculate centrality value for each residue (# neighbors).")
    ap.add_argument("--mutations", default="", help="CSV (pdb,position,mutation): write only these mutants")
    ap.add_argument("--precision", default=None, choices=["f16x2", "bf16x3", "fp32"])
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
    # every rank parses ~2/N of the files (a strided length pre-pass + the rest of its own LPT shard), not all of them
    shard, lengths, seqs, names = tdist.parse_sharded(args.pdbs, [args.chain] * len(args.pdbs), k_neighbors=engine.K)
    with torch.cuda.device(engine.device):
        tables, neigh = scan_proteins(engine, shard, centrality=args.centrality, lengths=lengths)
    proteins = [{"seq": s_, "name": n_} for s_, n_ in zip(seqs, names)]     # what the writer needs of every protein
    if rank == 0:
        rows = []
        if args.mutations:
            tri = read_mutation_list(args.mutations, proteins)
            for i, pos, a in tri:
                p = proteins[i]
                rows.append({"WT Seq": p["seq"], "Model": "ThermoMPNN", "Dataset": args.dataset_name,
                             "ddG_pred": float(tables[i][pos, a]), "position": int(pos), "wildtype": p["seq"][pos],
                             "mutation": AA20[a], "neighbors": int(neigh[i][pos]) if neigh else "", "best_AA": "",
                             "pdb": p["name"].strip(".pdb")})
        else:
            for i, p in enumerate(proteins):
                rows += rows_for_protein(p, tables[i], neigh[i] if neigh else None, "ThermoMPNN", args.dataset_name,
                                         args.pick_best, args.include_cys)
        write_csv(rows, args.out)
        print(f"Saved {len(rows)} rows for {len(proteins)} proteins to {args.out} ({world} rank(s))")
    if world > 1:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()
    return args.out


if __name__ == "__main__":
    main()
