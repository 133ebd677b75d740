"""Host-side PDB reading and batch packing for the ThermoMPNN hot path.

Behavioural contract, restated from
/root/reference/protein_mpnn_utils.py:183-350 (alt_parse_PDB*) and :353-605 (tied_featurize).
Own implementation: one pass over the file for all chains, vectorised packing.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

AA1 = "ARNDCQEGHILKMFPSTWYV-"                       # parser alphabet (protein_mpnn_utils.py:191)
AA3 = ("ALA ARG ASN ASP CYS GLN GLU GLY HIS ILE LEU LYS MET PHE PRO SER THR TRP TYR VAL GAP").split()
_AA3_TO_1 = {t: o for t, o in zip(AA3, AA1)}
MPNN_ALPHABET = "ACDEFGHIKLMNPQRSTVWYX"            # featurizer alphabet (protein_mpnn_utils.py:356)
_AA_TO_IDX = {a: i for i, a in enumerate(MPNN_ALPHABET)}
BACKBONE = ("N", "CA", "C", "O")

_DEFAULT_CHAINS = [chr(c) for c in range(ord("A"), ord("Z") + 1)] + \
                  [chr(c) for c in range(ord("a"), ord("z") + 1)] + [str(i) for i in range(300)]


class _ChainAcc:
    """Per-chain accumulator: first occurrence of (residue number, insertion code, atom) wins
    (protein_mpnn_utils.py:241-250)."""
    __slots__ = ("res", "raw_resn", "lo", "hi")

    def __init__(self):
        self.res: Dict[int, Dict[str, dict]] = {}
        self.raw_resn: Dict[str, None] = {}
        self.lo = None
        self.hi = None

    def add(self, line: str) -> None:
        atom = line[12:16].strip()
        resname = line[17:20]
        raw = line[22:27].strip()
        self.raw_resn.setdefault(raw, None)
        xyz = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
        if raw[-1].isalpha():
            ins, num = raw[-1], int(raw[:-1]) - 1
        else:
            ins, num = "", int(raw) - 1
        self.lo = num if self.lo is None or num < self.lo else self.lo
        self.hi = num if self.hi is None or num > self.hi else self.hi
        slot = self.res.setdefault(num, {}).setdefault(ins, {"name": resname, "atoms": {}})
        slot["atoms"].setdefault(atom, xyz)

    def finish(self, atoms: Sequence[str]):
        seq, coords = [], []
        nan3 = (np.nan, np.nan, np.nan)
        for num in range(self.lo, self.hi + 1):
            entry = self.res.get(num)
            if entry is None:                       # numbering gap -> '-' + NaN coords (:262-278)
                seq.append("-")
                coords.extend(nan3 for _ in atoms)
                continue
            for ins in sorted(entry):
                seq.append(_AA3_TO_1.get(entry[ins]["name"], "-"))
                got = entry[ins]["atoms"]
                coords.extend(got.get(a, nan3) for a in atoms)
        xyz = np.asarray(coords, dtype=np.float64).reshape(-1, len(atoms), 3)
        return xyz, "".join(seq), list(self.raw_resn)


def _scan_pdb(path: str, wanted: Optional[set]) -> Dict[str, _ChainAcc]:
    chains: Dict[str, _ChainAcc] = {}
    with open(path, "rb") as fh:
        for raw in fh:
            line = raw.decode("utf-8", "ignore").rstrip()
            if line[:6] == "HETATM" and line[17:20] == "MSE":     # selenomethionine -> MET (:217-220)
                line = line.replace("HETATM", "ATOM  ").replace("MSE", "MET")
            if line[:4] != "ATOM":
                continue
            ch = line[21:22]
            if wanted is not None and ch not in wanted:
                continue
            chains.setdefault(ch, _ChainAcc()).add(line)
    return chains


def alt_parse_PDB(path_to_pdb: str, input_chain_list=None, ca_only: bool = False, side_chains: bool = False):
    """PDB file -> ``[dict]`` with ``seq``, ``seq_chain_X``, ``coords_chain_X``, ``name``,
    ``num_of_chains``, ``resn_list`` (reference: protein_mpnn_utils.py:283-350).

    ``input_chain_list`` may be a list of chain ids or a string of one-letter ids (the reference
    iterates it, so ``'A'`` and ``['A']`` are equivalent — custom_inference.py:74)."""
    if side_chains:
        raise NotImplementedError("side_chains=True is not on the ThermoMPNN inference path")
    letters = list(input_chain_list) if input_chain_list else list(_DEFAULT_CHAINS)
    atoms = ("CA",) if ca_only else BACKBONE
    parsed = _scan_pdb(path_to_pdb, set(letters))

    out: dict = {"resn_list": []}
    concat = []
    n_found = 0
    last_resn: list = list("no_chain")             # reference quirk: resn_list of the LAST letter tried (:346)
    for letter in letters:
        acc = parsed.get(letter)
        if acc is None:
            last_resn = list("no_chain")
            continue
        xyz, seq, resn = acc.finish(atoms)
        last_resn = resn
        concat.append(seq)
        out["seq_chain_" + letter] = seq
        if ca_only:
            coords = {"CA_chain_" + letter: xyz.tolist()}
        else:
            coords = {f"{a}_chain_{letter}": xyz[:, i, :].tolist() for i, a in enumerate(BACKBONE)}
        out["coords_chain_" + letter] = coords
        n_found += 1
    out["name"] = path_to_pdb[path_to_pdb.rfind("/") + 1:-4]
    out["num_of_chains"] = n_found
    out["seq"] = "".join(concat)
    out["resn_list"] = last_resn
    return [out]


def _chain_letters(entry: dict) -> List[str]:
    # last character of every 'seq_chain_*' key, in insertion order (protein_mpnn_utils.py:387)
    return [k[-1:] for k in entry if k[:10] == "seq_chain_"]


def tied_featurize(batch, device, chain_dict, fixed_position_dict=None, omit_AA_dict=None,
                   tied_positions_dict=None, pssm_dict=None, bias_by_res_dict=None, ca_only=False):
    """Pack parsed proteins into padded tensors; returns the reference's 20-tuple
    (protein_mpnn_utils.py:353-605; tuple order at :605).

    Only the arguments the inference path uses are implemented (``chain_dict`` and
    ``fixed_position_dict``); the sequence-design extras raise ``NotImplementedError``."""
    for name, val in (("omit_AA_dict", omit_AA_dict), ("tied_positions_dict", tied_positions_dict),
                      ("pssm_dict", pssm_dict), ("bias_by_res_dict", bias_by_res_dict)):
        if val is not None:
            raise NotImplementedError(f"{name} is a sequence-design option outside the ThermoMPNN hot path")
    if ca_only:
        raise NotImplementedError("ca_only=True is never used by ThermoMPNN (transfer_model.py:26)")

    B = len(batch)
    lengths = np.array([len(b["seq"]) for b in batch], dtype=np.int32)
    L = int(lengths.max())
    n_alpha = len(MPNN_ALPHABET)
    X = np.full((B, L, 4, 3), np.nan, dtype=np.float64)
    residue_idx = np.full((B, L), -100, dtype=np.int32)
    chain_M = np.zeros((B, L), dtype=np.int32)
    chain_M_pos = np.zeros((B, L), dtype=np.int32)
    chain_enc = np.zeros((B, L), dtype=np.int32)
    S = np.zeros((B, L), dtype=np.int32)
    omit_AA_mask = np.zeros((B, L, n_alpha), dtype=np.int32)

    # The reference resolves masked/visible chains in a first loop and reuses the LAST element's
    # lists for every element (:382-391); identical for the B=1 calls of the hot path.
    for b in batch:
        if chain_dict is not None:
            masked, visible = chain_dict[b["name"]]
        else:
            masked, visible = _chain_letters(b), []
    order = list(masked) + list(visible)

    letters_ll, visible_ll, masked_ll, masked_len_ll, tied_ll = [], [], [], [], []
    for i, b in enumerate(batch):
        pos = 0
        c = 1
        letters, vis, msk, msk_len = [], [], [], []
        for letter in order:
            roles = [r for r, members in (("visible", visible), ("masked", masked)) if letter in members]
            for role in roles:                      # a letter listed in both is packed twice, as in the reference
                seq = "".join("X" if a == "-" else a for a in b[f"seq_chain_{letter}"])
                n = len(seq)
                coords = b[f"coords_chain_{letter}"]
                xyz = np.stack([np.asarray(coords[f"{a}_chain_{letter}"], dtype=np.float64) for a in BACKBONE], 1)
                sl = slice(pos, pos + n)
                X[i, sl] = xyz
                S[i, sl] = [_AA_TO_IDX[a] for a in seq]        # ValueError-equivalent: KeyError on unknown letter
                chain_enc[i, sl] = c
                residue_idx[i, sl] = 100 * (c - 1) + np.arange(pos, pos + n)
                letters.append(letter)
                if role == "masked":
                    chain_M[i, sl] = 1
                    fixed = np.ones(n, dtype=np.int32)
                    if fixed_position_dict is not None:
                        fp = fixed_position_dict[b["name"]][letter]
                        if fp:
                            fixed[np.asarray(fp) - 1] = 0
                    chain_M_pos[i, sl] = fixed
                    msk.append(letter)
                    msk_len.append(n)
                else:
                    chain_M_pos[i, sl] = 1
                    vis.append(letter)
                pos += n
                c += 1
        letters_ll.append(letters)
        visible_ll.append(vis)
        masked_ll.append(msk)
        masked_len_ll.append(msk_len)
        tied_ll.append([])

    mask = np.isfinite(X.sum(axis=(2, 3))).astype(np.float32)      # :576
    X = np.nan_to_num(X, nan=0.0)                                   # :577 (coords are finite or NaN)

    jumps = ((residue_idx[:, 1:] - residue_idx[:, :-1]) == 1).astype(np.float32)
    phi = np.pad(jumps, [[0, 0], [1, 0]])
    psi = np.pad(jumps, [[0, 0], [0, 1]])
    dihedral_mask = np.stack([phi, psi, psi], -1)

    f32 = dict(dtype=torch.float32, device=device)
    i64 = dict(dtype=torch.long, device=device)
    t = torch.from_numpy
    return (t(X).to(**f32), t(S).to(**i64), t(mask).to(**f32), lengths, t(chain_M).to(**f32),
            t(chain_enc).to(**i64), letters_ll, visible_ll, masked_ll, masked_len_ll,
            t(chain_M_pos).to(**f32), t(omit_AA_mask).to(**f32), t(residue_idx).to(**i64),
            t(dihedral_mask).to(**f32), tied_ll,
            torch.zeros((B, L), **f32), torch.zeros((B, L, 21), **f32),
            torch.full((B, L, 21), 10000.0, **f32) * t(_valid_rows(lengths, L)).to(**f32)[:, :, None],
            torch.zeros((B, L, 21), **f32), torch.ones(L, **f32))


def _valid_rows(lengths: np.ndarray, L: int) -> np.ndarray:
    # pssm_log_odds is 10000 on real rows and 0 on padding (:554,:557)
    return (np.arange(L)[None, :] < lengths[:, None]).astype(np.float32)


def featurize(batch, device):
    """Training-flavour packer signature (model_utils.py:19-125) -> 8-tuple
    ``(X, S, mask, lengths, chain_M, residue_idx, mask_self, chain_encoding_all)``.
    Deterministic (no chain shuffling); needs ``masked_list`` / ``visible_list`` keys."""
    B = len(batch)
    lengths = np.array([len(b["seq"]) for b in batch], dtype=np.int32)
    L = int(lengths.max())
    X = np.full((B, L, 4, 3), np.nan)
    residue_idx = np.full((B, L), -100, dtype=np.int32)
    chain_M = np.zeros((B, L), dtype=np.int32)
    mask_self = np.ones((B, L, L), dtype=np.int32)
    chain_enc = np.zeros((B, L), dtype=np.int32)
    S = np.zeros((B, L), dtype=np.int32)
    for i, b in enumerate(batch):
        masked, visible = list(b["masked_list"]), list(b["visible_list"])
        # visible chains with a sequence identical to a masked chain are promoted to masked (:43-50)
        for kv in list(visible):
            if any(b[f"seq_chain_{km}"] == b[f"seq_chain_{kv}"] for km in masked if km != kv):
                masked.append(kv)
                visible.remove(kv)
        pos, c = 0, 1
        for letter in masked + visible:
            seq = b[f"seq_chain_{letter}"]
            n = len(seq)
            coords = b[f"coords_chain_{letter}"]
            sl = slice(pos, pos + n)
            X[i, sl] = np.stack([np.asarray(coords[f"{a}_chain_{letter}"], dtype=np.float64) for a in BACKBONE], 1)
            S[i, sl] = [_AA_TO_IDX[a] for a in seq]
            chain_M[i, sl] = 1 if letter in masked else 0
            chain_enc[i, sl] = c
            residue_idx[i, sl] = 100 * (c - 1) + np.arange(pos, pos + n)
            mask_self[i, sl, sl] = 0
            pos += n
            c += 1
    mask = np.isfinite(X.sum(axis=(2, 3))).astype(np.float32)
    X = np.nan_to_num(X, nan=0.0)
    t = torch.from_numpy
    return (t(X).to(dtype=torch.float32, device=device), t(S).to(dtype=torch.long, device=device),
            t(mask).to(device), lengths, t(chain_M).to(dtype=torch.float32, device=device),
            t(residue_idx).to(dtype=torch.long, device=device),
            t(mask_self).to(dtype=torch.float32, device=device),
            t(chain_enc).to(dtype=torch.long, device=device))
