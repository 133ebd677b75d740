"""The mutation record the ThermoMPNN API takes (mirrors /root/reference/datasets.py:25-31)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

ALPHABET = "ACDEFGHIKLMNPQRSTVWYX"   # transfer_model.py:11 — index 20 ('X') = gap / unknown


@dataclass
class Mutation:
    position: int            # 0-based index into the concatenated parsed sequence
    wildtype: str
    mutation: str
    ddG: Optional[float] = None
    pdb: Optional[str] = ""


# ------------------------------------------------------------------------------------------------------------------
# CSV + PDB benchmark datasets: counterparts of ddgBenchDataset (/root/reference/datasets.py:248-317)
# and FireProtDataset (:167-245) without pandas / Bio. Each item is (pdb, mutations) exactly as the reference yields it:
# pdb = [parsed dict], mutations = [Mutation(position into the parsed sequence, wt, mut, ddG Tensor[1] | None, name)].
# Host-side bookkeeping only; the device work happens in TransferModel.forward / dist.ssm_scan.
# ------------------------------------------------------------------------------------------------------------------
import csv as _csv
import math as _math
import os as _os


def _ddg_tensor(text, sign: float):
    """CSV cell -> Tensor[1] (times ``sign``) or None for empty / NaN cells."""
    import torch
    if text is None or str(text).strip() == "":
        return None
    v = float(text)
    return None if _math.isnan(v) else torch.tensor([v * sign], dtype=torch.float32)


def _read_rows(path):
    with open(path, newline="") as fh:
        return [{(k or "").strip(): v for k, v in r.items()} for r in _csv.DictReader(fh)]


class ddgBenchDataset:
    """SSYM / S669 / myoglobin style CSV (columns PDB, MUT, DDG[, SEQ]) over a directory of PDB files
    (/root/reference/datasets.py:248-317). ``PDB`` = 4-letter id + chain letter; ``MUT`` = wt + author residue number +
    mutant; the residue number is looked up in the parser's ``resn_list``; ddG = -DDG (the CSVs store stabilisation with
    the opposite sign, :311). Rows whose residue number is not in the structure are skipped; a wild-type mismatch is
    repaired with the reference's gap-count contingency (:296-308) and otherwise raises AssertionError."""

    def __init__(self, cfg, pdb_dir: str, csv_fname: str):
        self.cfg, self.pdb_dir = cfg, pdb_dir
        self.rows = _read_rows(csv_fname)
        self.wt_names, self.mut_rows, self.wt_seqs = [], {}, {}
        for r in self.rows:                                   # pandas .unique(): order of first appearance
            name = r["PDB"]
            if name not in self.mut_rows:
                self.wt_names.append(name)
                self.mut_rows[name] = []
            self.mut_rows[name].append(r)
        if "S669" not in self.pdb_dir:
            for name in self.wt_names:
                self.wt_seqs[name[:-1]] = self.mut_rows[name][0].get("SEQ")

    def __len__(self):
        return len(self.wt_names)

    def __getitem__(self, index):
        from .pdb_io import alt_parse_PDB
        full = self.wt_names[index]
        chain = [full[-1]]
        wt_name = full.split(".pdb")[0][:-1]
        pdb = alt_parse_PDB(_os.path.join(self.pdb_dir, wt_name + ".pdb"), chain)
        resn_list, seq = pdb[0]["resn_list"], pdb[0]["seq"]
        mutations = []
        for row in self.mut_rows[full]:
            info = row["MUT"]
            wt_aa, mut_aa = info[0], info[-1]
            try:
                pdb_idx = resn_list.index(info[1:-1])
            except ValueError:                                # insertion codes etc.: skipped like the reference (:291-292)
                continue
            if seq[pdb_idx] != wt_aa:                         # mis-alignment contingency (:296-308)
                if "S669" in self.pdb_dir:
                    gaps = sum(1 for g in seq if g == "-")
                else:
                    gaps = sum(1 for g in seq[:pdb_idx + 10] if g == "-")
                pdb_idx += gaps if gaps > 0 else 1
                assert seq[pdb_idx] == wt_aa, f"{full} {info}: structure has {seq[pdb_idx]} at the aligned position"
            mutations.append(Mutation(pdb_idx, seq[pdb_idx], mut_aa, _ddg_tensor(row.get("DDG"), -1.0), wt_name))
        return pdb, mutations

    def __iter__(self):
        return (self[i] for i in range(len(self)))


def global_alignment_map(a: str, b: str):
    """Index map of a global alignment of ``a`` onto ``b`` that maximises the number of identical columns with free gaps
    (the scoring of Bio.pairwise2.align.globalxx the reference uses for mis-aligned FireProt entries, datasets.py:229-231):
    -> list m with m[i] = index in b aligned to a[i], or None when a[i] faces a gap."""
    n, m = len(a), len(b)
    score = [[0] * (m + 1) for _ in range(n + 1)]
    for i in range(n - 1, -1, -1):
        ai, row, below = a[i], score[i], score[i + 1]
        for j in range(m - 1, -1, -1):
            best = below[j + 1] + (1 if ai == b[j] else 0)
            if below[j] > best:
                best = below[j]
            if row[j + 1] > best:
                best = row[j + 1]
            row[j] = best
    out, i, j = [None] * n, 0, 0
    while i < n and j < m:
        if a[i] == b[j] and score[i][j] == score[i + 1][j + 1] + 1:
            out[i] = j
            i, j = i + 1, j + 1
        elif score[i][j] == score[i + 1][j]:
            i += 1
        elif score[i][j] == score[i][j + 1]:
            j += 1
        else:                                                 # mismatch column
            out[i] = j
            i, j = i + 1, j + 1
    return out


class FireProtDataset:
    """FireProtDB CSV (columns pdb_id_corrected, pdb_sequence, pdb_position, wild_type, mutation, ddG) with a pickled
    split dictionary {train/val/test/...: [pdb names]} (/root/reference/datasets.py:167-245). ``split='all'`` joins
    every split. Positions index ``pdb_sequence``; when the parsed structure disagrees the position is re-mapped through
    a global alignment (the reference's pairwise2 contingency) and unmappable rows are dropped."""

    def __init__(self, cfg, split: str, allow_pickle: bool = False):
        """``allow_pickle``: read the split file with the unrestricted pickle loader (it can execute code from the file).
        Default: the restricted unpickler of ``weights.safe_unpickle`` (containers, numpy arrays of names) — the
        reference's own ``dataset_splits/*.pkl`` load with it."""
        self.cfg, self.split = cfg, split
        rows = [r for r in _read_rows(cfg.data_loc.fireprot_csv) if (r.get("ddG") or "").strip() not in ("", "nan", "NaN")]
        self.rows = rows
        self.seq_to_data = {}
        for r in rows:
            self.seq_to_data.setdefault(r["pdb_sequence"], []).append(r)
        if allow_pickle:
            import pickle
            with open(cfg.data_loc.fireprot_splits, "rb") as fh:  # the reference's own split file format (a pickled dict)
                splits = pickle.load(fh)
        else:
            from .weights import safe_unpickle
            splits = safe_unpickle(cfg.data_loc.fireprot_splits)
        if not isinstance(splits, dict):
            raise ValueError(f"{cfg.data_loc.fireprot_splits}: expected a pickled dict of split name -> protein names")
        self.wt_names = [n for sub in splits.values() for n in sub] if split == "all" else list(splits[split])
        self.mut_rows = {n: [r for r in rows if r["pdb_id_corrected"] == n] for n in self.wt_names}
        self.wt_seqs = {n: self.mut_rows[n][0]["pdb_sequence"] for n in self.wt_names}

    def __len__(self):
        return len(self.wt_names)

    def __getitem__(self, index):
        from .pdb_io import alt_parse_PDB
        wt_name = self.wt_names[index]
        seq = self.wt_seqs[wt_name]
        data = self.seq_to_data[seq]
        pdb = alt_parse_PDB(_os.path.join(self.cfg.data_loc.fireprot_pdbs, f"{data[0]['pdb_id_corrected']}.pdb"), None)
        pseq = pdb[0]["seq"]
        amap = None
        mutations = []
        for row in data:
            pos = int(float(row["pdb_position"]))
            idx = pos
            ok = 0 <= idx < len(pseq) and pseq[idx] == row["wild_type"] == row["pdb_sequence"][pos]
            if not ok:
                if amap is None:
                    amap = global_alignment_map(seq, pseq.replace("-", "X"))
                idx = amap[pos] if 0 <= pos < len(amap) else None
                if idx is None:
                    continue
                assert pseq[idx] == row["wild_type"] == row["pdb_sequence"][pos], f"{wt_name}: cannot align position {pos}"
            mutations.append(Mutation(idx, pseq[idx], row["mutation"], _ddg_tensor(row.get("ddG"), 1.0), wt_name))
        return pdb, mutations

    def __iter__(self):
        return (self[i] for i in range(len(self)))
