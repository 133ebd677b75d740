"""Site-saturation mutation lists (reference: /root/reference/analysis/SSM.py:16-29 and the
string -> Mutation parsing at analysis/custom_inference.py:79-90)."""
from __future__ import annotations

from typing import List, Optional

from .datasets import ALPHABET, Mutation


def get_ssm_mutations(pdb: dict) -> List[Optional[str]]:
    """20 strings ``wt + pos + aa`` per residue (aa over ALPHABET[:-1]); one ``None`` per gap ('-')."""
    out: List[Optional[str]] = []
    for pos, wt in enumerate(pdb["seq"]):
        if wt == "-":
            out.append(None)
        else:
            out.extend(f"{wt}{pos}{aa}" for aa in ALPHABET[:-1])
    return out


def mutation_objects(pdb: dict, strings=None) -> List[Optional[Mutation]]:
    """Parse ``'S0A'``-style strings into Mutation records, asserting alphabet membership with the
    reference's messages (custom_inference.py:86-87)."""
    strings = get_ssm_mutations(pdb) if strings is None else strings
    out: List[Optional[Mutation]] = []
    for m in strings:
        if m is None:
            out.append(None)
            continue
        m = m.strip()
        wt, pos, mut = m[0], int(m[1:-1]), m[-1]
        assert wt in ALPHABET, f"Wild type residue {wt} invalid, please try again with one of the following options: {ALPHABET}"
        assert mut in ALPHABET, f"Wild type residue {mut} invalid, please try again with one of the following options: {ALPHABET}"
        out.append(Mutation(position=pos, wildtype=wt, mutation=mut, ddG=None, pdb=pdb["name"]))
    return out
