"""Counterpart of /root/reference/analysis/SSM.py: site-saturation scans over MANY proteins.

The reference runs one protein per forward and appends to a pandas frame cell by cell with one device sync per
mutation (SSM.py:105-147). Here: native threaded PDB parsing -> ONE ragged batch per chunk -> one fused forward ->
one device-to-host copy -> columnar CSV writer. Post-processing options keep the reference semantics:
  --centrality   'neighbors' column = #CA within 10 A (compute_centrality, SSM.py:129-132,144-145)
  --pick_best    keep one row per position carrying best_AA = argmin ddG (retrieve_best_mutants, SSM.py:32-42,153-162)
  --include_cys  otherwise mutations to C are excluded (from the best-pick, or dropped from the listing; :164-166)
With torch.distributed initialised, proteins are sharded over the ranks (dist.ssm_scan) and rank 0 writes.
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


def scan_proteins(engine, proteins: Sequence[dict], centrality: bool = False, chunk_residues: int = 1 << 18):
    """proteins: dicts from native_pdb.parse_pdb. -> (list of [L,21] ddG arrays, list of neighbour-count arrays or None).
    Proteins are processed in ragged chunks of at most ``chunk_residues`` residues (workspace ~27.5 KB / residue)."""
    from .dist import pack_proteins
    tables: List[Optional[np.ndarray]] = [None] * len(proteins)
    neigh: List[Optional[np.ndarray]] = [None] * len(proteins)
    order = sorted(range(len(proteins)), key=lambda i: -len(proteins[i]["S"]))
    i = 0
    while i < len(order):
        ids, tot = [], 0
        while i < len(order) and (not ids or tot + len(proteins[order[i]]["S"]) <= chunk_residues):
            ids.append(order[i])
            tot += len(proteins[order[i]]["S"])
            i += 1
        b = pack_proteins(proteins, ids, engine.device)
        ddg = engine.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=b["max_len"])["ddg"]
        cen = engine.centrality(b["X"], b["mask"], b["offsets"], 10.0).cpu().numpy() if centrality else None
        ddg = ddg.cpu().numpy()                                  # one D2H copy per chunk
        if not np.isfinite(ddg[:, :20]).all():
            raise RuntimeError("non-finite ddG: the default f16x2 matrix-core path needs |activations| < 65504 — "
                               "rerun with TMPNN_PRECISION=bf16x3 (full fp32 range)")
        pos = 0
        for pid in ids:
            L = len(proteins[pid]["S"])
            tables[pid] = ddg[pos:pos + L]
            if cen is not None:
                neigh[pid] = cen[pos:pos + L]
            pos += L
    return tables, (neigh if centrality else None)


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
    args = ap.parse_args(argv)

    from .custom_inference import load_model
    model = load_model(args.model_path, args.thermompnn_dir, args.synthetic_weights)
    engine = model.engine()
    proteins = native_pdb.parse_pdbs(args.pdbs, [args.chain] * len(args.pdbs))
    with torch.cuda.device(engine.device):
        tables, neigh = scan_proteins(engine, proteins, centrality=args.centrality)
    rows = []
    for i, p in enumerate(proteins):
        rows += rows_for_protein(p, tables[i], neigh[i] if neigh else None, "ThermoMPNN", args.dataset_name,
                                 args.pick_best, args.include_cys)
    write_csv(rows, args.out)
    print(f"Saved {len(rows)} rows for {len(proteins)} proteins to {args.out}")
    return args.out


if __name__ == "__main__":
    main()
