#!/usr/bin/env python3
"""The end-to-end leg of bench.py on its own: PDB files -> CSV / binary tables, wall clock (prints one JSON object).
    python tools/e2e_bench.py [n_files]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from thermompnn_amd.engine import Engine  # noqa: E402
from thermompnn_amd.weights import synthetic_state_dict  # noqa: E402

if __name__ == "__main__":
    eng = Engine(synthetic_state_dict(0), "cuda:0", 48)
    print(json.dumps(bench.end_to_end(eng, int(sys.argv[1]) if len(sys.argv) > 1 else None), indent=1))
