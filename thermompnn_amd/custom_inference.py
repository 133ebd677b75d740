"""Counterpart of /root/reference/analysis/custom_inference.py: one PDB -> full SSM -> CSV with the reference's
schema (``,Model,Dataset,ddG_pred,position,wildtype,mutation,pdb,chain``; custom_inference.py:64,96-111).

    python -m thermompnn_amd.custom_inference --pdb examples/2OCJ.pdb --chain A --model_path models/thermoMPNN_default.pt \
        --thermompnn_dir <dir holding vanilla_model_weights/v_48_020.pt>

No Bio / omegaconf / Lightning / pandas dependency. ``--synthetic_weights SEED`` runs with the repo's deterministic
synthetic weights when the real checkpoints are not available. The whole ddG table comes back in ONE device-to-host
copy (the reference does one ``.cpu().item()`` sync per mutation, :97).
"""
from __future__ import annotations

import argparse
import csv
import os
import tempfile

import torch

from . import weights as _weights
from .datasets import ALPHABET
from .pdb_io import alt_parse_PDB
from .ssm import mutation_objects
from .transfer_model import TransferModel

MODEL_CFG = dict(hidden_dims=[64, 32], subtract_mut=True, num_final_layers=2, freeze_weights=True,
                 load_pretrained=True, lightattn=True, lr_schedule=True)      # custom_inference.py:39-47


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def first_chain(pdb_path: str) -> str:
    """Chain id of the first ATOM record (the reference asks Bio.PDB for the first chain, :21-25,:70-71)."""
    with open(pdb_path) as fh:
        for line in fh:
            if line.startswith("ATOM"):
                return line[21]
    raise ValueError(f"{pdb_path}: no ATOM records")


def pdb_id_of(path: str) -> str:
    # the reference uses rstrip('.pdb') (a character-set strip, :58); identical for ids like '2OCJ'
    return os.path.basename(path).rstrip(".pdb")


def load_model(model_path: str | None, thermompnn_dir: str | None, synthetic_seed: int | None, device="cuda",
               precision: str | None = None, allow_pickle: bool | None = None):
    if synthetic_seed is not None:
        sd = _weights.synthetic_state_dict(synthetic_seed)
        tmp = tempfile.mkdtemp(prefix="tmpnn_w_")
        os.makedirs(os.path.join(tmp, "vanilla_model_weights"))
        _weights.save_vanilla_checkpoint(os.path.join(tmp, "vanilla_model_weights", "v_48_020.pt"),
                                         _weights.split_transfer_state_dict(sd)[0], 48)
        thermompnn_dir = tmp
    else:
        sd = _weights.load_thermompnn_checkpoint(model_path, allow_pickle=allow_pickle)
    cfg = AttrDict(model=AttrDict(MODEL_CFG), platform=AttrDict(thermompnn_dir=thermompnn_dir))
    model = TransferModel(cfg)
    model.load_state_dict(sd)
    model.precision = precision
    return model.eval().to(device)


def ssm_rows(model, pdb_path: str, chain: str, model_name: str = "ThermoMPNN"):
    """-> list of row dicts in the reference's column order."""
    mut_pdb = alt_parse_PDB(pdb_path, chain)
    muts = mutation_objects(mut_pdb[0])
    with torch.no_grad():
        pred, _ = model(mut_pdb, muts)
    # (range problems never get here: Engine.ssm_forward reruns an overflowing f16x2 batch in bf16x3 or raises)
    vals = torch.cat([p["ddG"] for p in pred if p is not None]).cpu().tolist()     # one D2H copy
    rows, k = [], 0
    dataset = pdb_id_of(pdb_path)
    for m in muts:
        if m is None:
            continue
        rows.append({"Model": model_name, "Dataset": dataset, "ddG_pred": vals[k], "position": m.position,
                     "wildtype": m.wildtype, "mutation": m.mutation, "pdb": m.pdb.strip(".pdb"), "chain": chain})
        k += 1
    return rows


def ssm_to_csv(model, pdb_path: str, chain: str, csv_file: str, model_name: str = "ThermoMPNN") -> int:
    """The script's fast path: native parser -> ONE fused forward -> ONE device-to-host copy of the [L, 21] table -> native
    columnar writer (csrc/tmpnn_csv.cpp), no Python object per mutation. Same bytes as ``write_csv(ssm_rows(...))``, which
    goes through the reference-shaped ``TransferModel.forward(pdb, mutations)`` API (tests compare the two files). -> rows."""
    import numpy as np
    from . import native_csv, native_pdb
    p = native_pdb.parse_pdb(pdb_path, chain)
    L = len(p["seq"])
    assert L > 0, f"{pdb_path}: chain {chain!r} has no residues"
    engine = model.engine()
    with torch.no_grad(), torch.cuda.device(engine.device):
        out = engine.ssm_forward(p["X"], p["S"], p["mask"], p["residue_idx"], p["chain_enc"], np.array([0, L], np.int32))
        table = out["ddg"].cpu().numpy()
    with native_csv.CsvWriter(csv_file, native_csv.SCHEMA_CUSTOM_INFERENCE) as w:
        w.write_ssm(table, np.array([0, L], np.int32), [p["seq"]], [p["name"].strip(".pdb")], model=model_name,
                    dataset=pdb_id_of(pdb_path), chain=chain, include_cys=True, n_threads=1)
    return w.rows


def write_csv(rows, path: str) -> None:
    cols = ["Model", "Dataset", "ddG_pred", "position", "wildtype", "mutation", "pdb", "chain"]
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh, lineterminator="\n")           # pandas' to_csv line ends (examples/ThermoMPNN_inference_2OCJ.csv)
        w.writerow([""] + cols)                       # pandas' unnamed index column
        for i, r in enumerate(rows):
            w.writerow([i] + [r[c] for c in cols])


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--pdb", type=str, default="", help="Input PDB to use for custom inference")
    ap.add_argument("--chain", type=str, default="A", help="Chain in input PDB to use.")
    ap.add_argument("--model_path", type=str, default="", help="filepath to model to use for inference")
    ap.add_argument("--out_dir", type=str, default="./", help="Output directory in which to save predictions.")
    ap.add_argument("--thermompnn_dir", type=str, default=".", help="directory holding vanilla_model_weights/ (local.yaml: platform.thermompnn_dir)")
    ap.add_argument("--synthetic_weights", type=int, default=None, help="use synthetic weights with this seed")
    ap.add_argument("--allow_pickle", action="store_true", default=False,
                    help="read --model_path with the unrestricted pickle loader (it can execute code from the file); the default "
                         "restricted loader already reads Lightning checkpoints such as thermoMPNN_default.pt")
    ap.add_argument("--reference_shaped", action="store_true", default=False,
                    help="go through TransferModel.forward(pdb, mutations) with one Mutation object and one result dict per "
                         "mutant, as the reference script does (same file, slower host side)")
    args = ap.parse_args(argv)
    chain = args.chain if len(args.chain) >= 1 else first_chain(args.pdb)
    out_dir = os.getcwd() if args.out_dir == "./" else args.out_dir
    assert os.path.isdir(out_dir), f"{out_dir} is not a valid directory."
    model = load_model(args.model_path, args.thermompnn_dir, args.synthetic_weights, allow_pickle=args.allow_pickle or None)
    csv_file = os.path.join(out_dir, "ThermoMPNN_inference_%s.csv" % pdb_id_of(args.pdb))
    if args.reference_shaped:
        write_csv(ssm_rows(model, args.pdb, chain), csv_file)
    else:
        ssm_to_csv(model, args.pdb, chain, csv_file)
    print(f"Saved ThermoMPNN output to {csv_file}")
    return csv_file


if __name__ == "__main__":
    main()
