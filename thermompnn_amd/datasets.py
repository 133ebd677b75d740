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
