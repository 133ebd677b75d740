"""Counterparts of the hot-path-adjacent symbols of /root/reference/analysis/thermompnn_benchmarking.py
(SURVEY.md §8f ranks 2-4): compute_centrality (:20-35), ProteinMPNNBaseline (:38-65), get_trained_model (:78-84), and the
per-dataset evaluation loops run_prediction_default / run_prediction_keep_preds (:87-187) over the CSV datasets of
thermompnn_amd.datasets, scored with thermompnn_amd.metrics (torchmetrics / pandas / tqdm are not needed)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .datasets import ALPHABET
from .engine import _ptr, _stream, check
from .pdb_io import tied_featurize
from .protein_mpnn_utils import _EngineOwner
from .transfer_model import TransferModel, get_protein_mpnn
from .weights import load_thermompnn_checkpoint


def compute_centrality(xyz, basis_atom: str = "CA", radius: float = 10.0, core_threshold: int = 20,
                       surface_threshold: int = 15, backup_atom: str = "C", chain: str = "A",
                       device="cuda") -> torch.Tensor:
    """Number of neighbours within ``radius`` of each residue's basis atom, minus self (reference :20-35).
    ``xyz`` = a parsed ``coords_chain_X`` dict. Runs the HIP kernel; returns an int tensor on ``device``."""
    coords = np.asarray(xyz[basis_atom + f"_chain_{chain}"], dtype=np.float64)
    L = coords.shape[0]
    ok = np.isfinite(coords).all(1)
    X = np.zeros((L, 4, 3), np.float32)
    X[:, 1] = np.nan_to_num(coords)                       # the kernel reads the CA slot
    dev = torch.device(device)
    Xd = torch.from_numpy(X).to(dev)
    md = torch.from_numpy(ok.astype(np.float32)).to(dev)
    off = torch.tensor([0, L], dtype=torch.int32, device=dev)
    out = torch.empty(L, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().tmpnn_centrality(_ptr(Xd), _ptr(md), _ptr(off), 1, L, float(radius), _ptr(out), _stream()),
              "tmpnn_centrality")
    return out.long()


class ProteinMPNNBaseline(_EngineOwner):
    """ProteinMPNN as a ddG proxy: ddG = -log p(mutant aa | structure) (reference :38-65)."""

    def __init__(self, cfg, version="v_48_020.pt"):
        super().__init__()
        self.prot_mpnn = get_protein_mpnn(cfg, version=version)
        self.k_neighbors = self.prot_mpnn.k_neighbors

    def ssm_table(self, pdb) -> torch.Tensor:
        """[L, 21] on the model's device: entry [pos, a] = the ddG ``forward`` returns for a mutation to ALPHABET[a] at pos
        (= -log p), the whole scan from the one forward the reference also runs (:47-53)."""
        device = next(self.parameters()).device
        feats = tied_featurize([pdb[0] if isinstance(pdb, (list, tuple)) else pdb], device, None, None, None, None, None, None, ca_only=False)
        X, S, mask, chain_M, chain_enc, residue_idx = feats[0], feats[1], feats[2], feats[4], feats[5], feats[12]
        *_, log_probs = self.prot_mpnn(X, S, mask, chain_M, residue_idx, chain_enc, None)
        return -log_probs[0]

    def forward(self, pdb, mutations, tied_feat=True):
        device = next(self.parameters()).device
        feats = tied_featurize([pdb[0]], device, None, None, None, None, None, None, ca_only=False)
        X, S, mask, chain_M, chain_enc, residue_idx = feats[0], feats[1], feats[2], feats[4], feats[5], feats[12]
        *_, log_probs = self.prot_mpnn(X, S, mask, chain_M, residue_idx, chain_enc, None)
        live = [m for m in mutations if m is not None]
        if not live:
            return [None for _ in mutations], log_probs
        # one gather for every listed mutant (the reference indexes log_probs once per mutation, :55-62), then Tensor[1] views
        sel = torch.tensor([[m.position for m in live], [ALPHABET.index(m.mutation) for m in live]], device=log_probs.device)
        pred = log_probs[0][sel[0], sel[1]]
        dtm, ddg = iter(pred.split(1)), iter((-pred).split(1))
        return [None if m is None else {"ddG": next(ddg), "dTm": next(dtm)} for m in mutations], log_probs


def get_trained_model(model_name, config, checkpt_dir="models/", override_custom=False, allow_pickle=None):
    """Load a ThermoMPNN Lightning checkpoint into the HIP-backed TransferModel (reference :78-84) without
    importing Lightning: the ``model.`` prefix of TransferModelPL is stripped by the loader."""
    import os
    path = model_name if override_custom else os.path.join(config.platform.thermompnn_dir, checkpt_dir, model_name)
    model = TransferModel(config)
    model.load_state_dict(load_thermompnn_checkpoint(path, allow_pickle=allow_pickle))
    return model


def run_prediction_default(name, model, dataset_name, dataset, results, keep_preds: bool = False):
    """Reference loop (:87-119): ``model(pdb, mutations)`` per protein, metrics over every mutation with a measured ddG.
    Appends {"Model", "Dataset", "ddG r2" ... "ddG pearson"} to ``results``; with ``keep_preds`` also returns the raw rows
    (the run_prediction_keep_preds variant, :122-187, without pandas)."""
    from .metrics import get_metrics
    pred_all, true_all, rows = [], [], []
    for pdb, mutations in dataset:
        if not mutations:
            continue
        with torch.no_grad():
            pred, _ = model(pdb, mutations)
        keep = [(m, o) for m, o in zip(mutations, pred) if o is not None and m.ddG is not None]
        if not keep:
            continue
        vals = torch.cat([o["ddG"].reshape(1) for _, o in keep]).cpu().tolist()        # one D2H copy per protein
        for (m, _), v in zip(keep, vals):
            pred_all.append(v)
            true_all.append(float(m.ddG))
            if keep_preds:
                rows.append({"Model": name, "Dataset": dataset_name, "ddG_true": float(m.ddG), "ddG_pred": v,
                             "position": m.position, "wildtype": m.wildtype, "mutation": m.mutation,
                             "pdb": (m.pdb or "").strip(".pdb")})
    met = get_metrics(pred_all, true_all)
    column = {"Model": name, "Dataset": dataset_name}
    for k in ("r2", "mse", "rmse", "spearman", "pearson"):
        column[f"ddG {k}"] = met[k]
    column["n"] = met["n"]
    results.append(column)
    return (results, rows) if keep_preds else results


def run_prediction_batched(name, engine, dataset_name, dataset, results, group=None):
    """The same evaluation on the batched path: every protein of the dataset goes through ONE ragged dist.ssm_scan (sharded
    over ``group``'s GPUs when torch.distributed is initialised) and the listed mutations are picked out of the [L, 21]
    tables with dist.select_mutations. Equivalent to run_prediction_default for mutations whose stated wild type matches
    the structure (the datasets guarantee that) — and one forward per dataset instead of one per protein."""
    from . import dist as tdist
    from .metrics import get_metrics
    from .pdb_io import tied_featurize
    prots, triples, truth = [], [], []
    for pdb, mutations in dataset:
        muts = [m for m in mutations if m is not None and m.ddG is not None and m.mutation in ALPHABET[:20]]
        if not muts:
            continue
        f = tied_featurize([pdb[0]], "cpu", None, None, None, None, None, None, ca_only=False)
        pid = len(prots)
        prots.append(dict(X=f[0][0].numpy(), S=f[1][0].numpy().astype(np.int32), mask=f[2][0].numpy(),
                          residue_idx=f[12][0].numpy().astype(np.int32), chain_enc=f[5][0].numpy().astype(np.int32)))
        for m in muts:
            assert pdb[0]["seq"][m.position] == m.wildtype, "batched evaluation needs the structure's own wild type"
            triples.append((pid, m.position, ALPHABET.index(m.mutation)))
            truth.append(float(m.ddG))
    tables = tdist.ssm_scan(engine, prots, group=group)
    pred = tdist.select_mutations(tables, np.asarray(triples, dtype=np.int64)).cpu().numpy()
    met = get_metrics(pred, truth)
    column = {"Model": name, "Dataset": dataset_name, "n": met["n"]}
    for k in ("r2", "mse", "rmse", "spearman", "pearson"):
        column[f"ddG {k}"] = met[k]
    results.append(column)
    return results


KEEP_PREDS_COLUMNS = ["WT Seq", "Model", "Dataset", "ddG_true", "ddG_pred", "position", "wildtype", "mutation", "neighbors",
                      "best_AA", "pdb"]


def run_prediction_keep_preds(name, model, dataset_name, dataset, results, centrality: bool = False, out_dir: str = "."):
    """The reference's evaluation loop that also saves the raw predictions (:122-187): metrics appended to ``results`` as
    ``run_prediction_default`` does, and ``<name>_<dataset_name>_raw_preds.csv`` in the layout its pandas frame gets — the
    unnamed running index, the KEEP_PREDS_COLUMNS, one row per LISTED mutation: a mutation without a measured ddG still gets
    its row with only 'WT Seq' / 'Model' / 'Dataset' filled (:159-167), 'WT Seq' = ``dataset.wt_seqs[key]`` and left empty for
    S669 sets (:166), 'pdb' character-set-stripped (:150), 'neighbors' only with ``centrality``, 'best_AA' always empty.
    One forward per protein and ONE device-to-host copy of its predictions (the reference syncs twice per mutation)."""
    import csv
    import os
    from .metrics import get_metrics
    pred_all, true_all, rows = [], [], []
    for pdb, mutations in dataset:
        if not mutations:
            continue
        with torch.no_grad():
            pred, _ = model(pdb, mutations)
        p = pdb[0]
        neighbors = None
        if centrality:
            coord_chain = [c for c in p.keys() if "coords" in c][0]
            neighbors = compute_centrality(p[coord_chain], basis_atom="CA", backup_atom="C", chain=coord_chain[-1], radius=10.0,
                                           device=next(model.parameters()).device).cpu().tolist()
        measured = [(m, o) for m, o in zip(mutations, pred) if m is not None and o is not None and m.ddG is not None]
        vals = iter(torch.cat([o["ddG"].reshape(1) for _, o in measured]).cpu().tolist() if measured else [])
        for m, o in zip(mutations, pred):
            if m is None:
                continue
            row = dict.fromkeys(KEEP_PREDS_COLUMNS, "")
            if o is not None and m.ddG is not None:
                v, t = next(vals), float(torch.as_tensor(m.ddG).reshape(-1)[0])
                pred_all.append(v)
                true_all.append(t)
                row.update({"ddG_true": t, "ddG_pred": v, "position": m.position, "wildtype": m.wildtype, "mutation": m.mutation,
                            "pdb": (m.pdb or "").strip(".pdb")})
                if neighbors is not None:
                    row["neighbors"] = int(neighbors[m.position])
            row["Model"], row["Dataset"] = name, dataset_name
            if "S669" not in dataset_name:
                wt = dataset.wt_seqs[m.pdb if "Megascale" not in dataset_name else m.pdb + ".pdb"]
                row["WT Seq"] = "" if wt is None else wt
            rows.append(row)
    met = get_metrics(pred_all, true_all)
    column = {"Model": name, "Dataset": dataset_name}
    for k in ("r2", "mse", "rmse", "spearman", "pearson"):
        column[f"ddG {k}"] = met[k]
    column["n"] = met["n"]
    results.append(column)
    with open(os.path.join(out_dir, name + "_" + dataset_name + "_raw_preds.csv"), "w", newline="") as fh:
        w = csv.writer(fh, lineterminator="\n")
        w.writerow([""] + KEEP_PREDS_COLUMNS)
        for i, r in enumerate(rows):
            w.writerow([i] + [r[c] for c in KEEP_PREDS_COLUMNS])
    return results


def evaluate_datasets(models: dict, datasets: dict, keep_preds: bool = False, centrality: bool = False, out_dir: str = "."):
    """The driver loop of the reference's benchmarking script (:242-253): every model over every dataset, then
    ``ThermoMPNN_metrics.csv`` (index, Model, Dataset, ddG r2 / mse / rmse / spearman / pearson) in ``out_dir``.
    -> the list of metric rows (each also carries ``n``, the number of scored mutations)."""
    import csv
    import os
    results = []
    for name, model in models.items():
        model = model.eval()
        for dataset_name, dataset in datasets.items():
            if keep_preds:
                run_prediction_keep_preds(name, model, dataset_name, dataset, results, centrality=centrality, out_dir=out_dir)
            else:
                run_prediction_default(name, model, dataset_name, dataset, results)
    cols = ["Model", "Dataset"] + [f"ddG {k}" for k in ("r2", "mse", "rmse", "spearman", "pearson")]
    with open(os.path.join(out_dir, "ThermoMPNN_metrics.csv"), "w", newline="") as fh:
        w = csv.writer(fh, lineterminator="\n")
        w.writerow([""] + cols)
        for i, r in enumerate(results):
            w.writerow([i] + [("" if isinstance(r[c], float) and r[c] != r[c] else r[c]) for c in cols])
    return results
