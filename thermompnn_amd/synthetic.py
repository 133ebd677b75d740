"""Synthetic proteins for the benchmark configs.

Cα = confined random walk (3.8 Å steps inside a sphere of radius 3.3·L^(1/3)+4 Å, rejection with
<= 50 tries); N/C/O = Cα + fixed offsets + N(0, 0.3²) noise; uniform 20-letter sequence.
"""
from __future__ import annotations

import numpy as np

AA20 = "ACDEFGHIKLMNPQRSTVWY"


def synthetic_backbone(L: int, seed: int = 0):
    """-> (X[L,4,3] float64 in N,CA,C,O order, seq str)."""
    rng = np.random.default_rng(seed)
    R = 3.3 * L ** (1.0 / 3.0) + 4.0
    ca = np.zeros((L, 3))
    for i in range(1, L):
        for _ in range(50):
            v = rng.normal(size=3)
            step = ca[i - 1] + 3.8 * v / np.linalg.norm(v)
            if np.linalg.norm(step) <= R:
                break
        ca[i] = step
    noise = rng.normal(0.0, 0.3, size=(L, 3, 3))
    n = ca + np.array([-1.2, 0.5, 0.3]) + noise[:, 0]
    c = ca + np.array([1.2, 0.6, -0.2]) + noise[:, 1]
    o = c + np.array([0.3, 1.1, 0.2]) + noise[:, 2]
    seq = "".join(AA20[i] for i in rng.integers(0, 20, size=L))
    return np.stack([n, ca, c, o], 1), seq


def synthetic_pdb_dict(L: int, seed: int = 0, name: str | None = None, chain: str = "A") -> dict:
    """A parsed-PDB dict in the shape ``alt_parse_PDB`` produces (single chain, no gaps)."""
    X, seq = synthetic_backbone(L, seed)
    coords = {f"{a}_chain_{chain}": X[:, i].tolist() for i, a in enumerate(("N", "CA", "C", "O"))}
    return {"resn_list": [str(i + 1) for i in range(L)], f"seq_chain_{chain}": seq,
            f"coords_chain_{chain}": coords, "name": name or f"syn_L{L}_s{seed}",
            "num_of_chains": 1, "seq": seq}


AA3 = {"A": "ALA", "R": "ARG", "N": "ASN", "D": "ASP", "C": "CYS", "Q": "GLN", "E": "GLU", "G": "GLY", "H": "HIS", "I": "ILE",
       "L": "LEU", "K": "LYS", "M": "MET", "F": "PHE", "P": "PRO", "S": "SER", "T": "THR", "W": "TRP", "Y": "TYR", "V": "VAL"}


def backbone_pdb_text(X, seq: str, chain: str = "A") -> str:
    """Backbone-only PDB text (N, CA, C, O records in the fixed columns alt_parse_PDB reads, protein_mpnn_utils.py:222-231)
    of a synthetic protein: the input files of the end-to-end (PDB -> CSV) measurement."""
    names = ((" N  ", "N"), (" CA ", "C"), (" C  ", "C"), (" O  ", "O"))
    out, serial = [], 1
    Xr = np.asarray(X, dtype=np.float64)
    for i, aa in enumerate(seq):
        res = AA3[aa]
        for a, (nm, el) in enumerate(names):
            x, y, z = Xr[i, a]
            out.append(f"ATOM  {serial:5d} {nm} {res} {chain}{i + 1:4d}    {x:8.3f}{y:8.3f}{z:8.3f}  1.00  0.00           {el}")
            serial += 1
    out.append("TER\nEND\n")
    return "\n".join(out)
