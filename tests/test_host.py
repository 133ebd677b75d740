"""Host logic: PDB parsing, batch packing, mutation lists, weight formats.  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from thermompnn_amd import pdb_io, ssm, weights
from thermompnn_amd.datasets import ALPHABET, Mutation
from thermompnn_amd.synthetic import synthetic_pdb_dict

PDB = os.path.join(GOLDEN, "2OCJ.pdb")
GAP = os.path.join(GOLDEN, "2OCJ_gap_chainA.pdb")


@pytest.mark.parametrize("tag,path,chains", [("A", PDB, "A"), ("AB", PDB, ["A", "B"]), ("gapA", GAP, "A")])
def test_parser_matches_reference(tag, path, chains):
    pg = load_golden("parser_2OCJ")
    d = pdb_io.alt_parse_PDB(path, chains)[0]
    assert d["seq"] == str(pg[tag + "_seq"])
    assert d["resn_list"] == [str(x) for x in pg[tag + "_resn_list"]]
    assert d["num_of_chains"] == int(pg[tag + "_num_of_chains"])
    for ch in chains:
        for atom, v in d["coords_chain_" + ch].items():
            np.testing.assert_array_equal(np.asarray(v), pg[f"{tag}_{atom}"])
    assert d["name"] == os.path.basename(path)[:-4]


def test_parser_missing_chain_is_skipped():
    d = pdb_io.alt_parse_PDB(PDB, "AZ")[0]
    assert d["num_of_chains"] == 1 and "seq_chain_Z" not in d
    assert d["resn_list"] == list("no_chain")     # reference quirk (protein_mpnn_utils.py:279-280,346)


def test_mse_and_insertion_codes(tmp_path):
    lines = [
        "ATOM      1  N   ALA A   1      0.000   0.000   0.000  1.00  0.00           N",
        "ATOM      2  CA  ALA A   1      1.458   0.000   0.000  1.00  0.00           C",
        "ATOM      3  C   ALA A   1      2.009   1.420   0.000  1.00  0.00           C",
        "ATOM      4  O   ALA A   1      1.251   2.390   0.000  1.00  0.00           O",
        "HETATM    5  N   MSE A   2      3.332   1.536   0.000  1.00  0.00           N",
        "HETATM    6  CA  MSE A   2      3.988   2.839   0.000  1.00  0.00           C",
        "HETATM    7  C   MSE A   2      5.504   2.693   0.000  1.00  0.00           C",
        "HETATM    8  O   MSE A   2      6.030   1.580   0.000  1.00  0.00           O",
        "ATOM      9  CA  GLY A   2A     7.000   3.000   0.000  1.00  0.00           C",
        "ATOM     10  CA  GLY A   2A     9.000   9.000   9.000  1.00  0.00           C",
        "ATOM     11  CA  UNK A   5      8.000   4.000   0.000  1.00  0.00           C",
    ]
    p = tmp_path / "toy1.pdb"
    p.write_text("\n".join(lines) + "\n")
    d = pdb_io.alt_parse_PDB(str(p), "A")[0]
    assert d["seq"] == "AMG---"          # MSE->M, insertion code sorted after '', gaps 3,4, UNK -> '-'
    ca = np.asarray(d["coords_chain_A"]["CA_chain_A"])
    assert ca[2].tolist() == [7.0, 3.0, 0.0]                  # first occurrence of an atom wins
    assert np.isnan(np.asarray(d["coords_chain_A"]["N_chain_A"])[2]).all()
    assert d["resn_list"] == ["1", "2", "2A", "5"] and d["name"] == "toy1"


def test_tied_featurize_tuple():
    g = load_golden("2OCJ_A_gap")
    d = pdb_io.alt_parse_PDB(GAP, "A")
    out = pdb_io.tied_featurize(d, "cpu", None)
    assert len(out) == 20
    X, S, mask, lengths, chain_M, chain_enc = out[:6]
    residue_idx = out[12]
    np.testing.assert_array_equal(X[0].numpy(), g["X"])
    np.testing.assert_array_equal(S[0].numpy(), g["S"])
    np.testing.assert_array_equal(mask[0].numpy(), g["mask"])
    np.testing.assert_array_equal(residue_idx[0].numpy(), g["residue_idx"])
    np.testing.assert_array_equal(chain_enc[0].numpy(), g["chain_enc"])
    assert S.dtype == torch.int64 and X.dtype == torch.float32 and lengths.dtype == np.int32
    assert chain_M.sum() == len(g["S"])
    with pytest.raises(NotImplementedError):
        pdb_io.tied_featurize(d, "cpu", None, pssm_dict={})


def test_tied_featurize_two_chains_and_padding():
    a = pdb_io.alt_parse_PDB(PDB, ["A", "B"])[0]
    b = synthetic_pdb_dict(40, seed=1)
    b["seq_chain_B"], b["coords_chain_B"] = b["seq_chain_A"], {k.replace("_A", "_B"): v for k, v in b["coords_chain_A"].items()}
    X, S, mask, lengths, chain_M, chain_enc, *rest = pdb_io.tied_featurize([a, b], "cpu", None)
    residue_idx = rest[6]
    assert X.shape == (2, 388, 4, 3) and lengths.tolist() == [388, 40]
    assert chain_enc[0, 193] == 1 and chain_enc[0, 194] == 2
    assert residue_idx[0, 194] == 100 + 194                     # 100*(c-1)+position (:473)
    assert mask[1, 80:].sum() == 0 and residue_idx[1, -1] == -100


def test_ssm_mutation_list_and_asserts():
    pdb = {"seq": "AC-W", "name": "toy"}
    m = ssm.get_ssm_mutations(pdb)
    assert len(m) == 61 and m[40] is None and m[0] == "A0A" and m[19] == "A0Y" and m[41] == "W3A"
    objs = ssm.mutation_objects(pdb)
    assert objs[40] is None and objs[41] == Mutation(3, "W", "A", None, "toy")
    with pytest.raises(AssertionError, match="invalid, please try again"):
        ssm.mutation_objects(pdb, ["B0A"])
    assert ALPHABET[:-1] == "ACDEFGHIKLMNPQRSTVWY"


def test_weight_formats_roundtrip(tmp_path):
    sd = weights.synthetic_state_dict(3)
    assert len(sd) == 130 and sum(v.numel() for v in sd.values()) == 4342876
    mp, hd = weights.split_transfer_state_dict(sd)
    assert len(mp) == 118 and sum(v.numel() for v in mp.values()) == 1660485
    weights.save_vanilla_checkpoint(tmp_path / "v.pt", mp, 48)
    k, back = weights.load_vanilla_checkpoint(tmp_path / "v.pt")
    assert k == 48 and all(torch.equal(back[n], mp[n]) for n in mp)
    weights.save_lightning_checkpoint(tmp_path / "t.ckpt", sd)
    back = weights.load_thermompnn_checkpoint(tmp_path / "t.ckpt")
    assert list(back) == list(sd) and all(torch.equal(back[n], sd[n]) for n in sd)
    assert torch.equal(weights.synthetic_state_dict(3)["both_out.1.weight"], sd["both_out.1.weight"])
    assert (sd["light_attention.feature_convolution.weight"] != 0).all()


def test_csv_writer_schema(tmp_path):
    from thermompnn_amd.custom_inference import first_chain, pdb_id_of, write_csv
    rows = [{"Model": "ThermoMPNN", "Dataset": "2OCJ", "ddG_pred": -0.25, "position": 0, "wildtype": "S",
             "mutation": "A", "pdb": "2OCJ", "chain": "A"}]
    write_csv(rows, tmp_path / "o.csv")
    lines = (tmp_path / "o.csv").read_text().splitlines()
    assert lines[0] == ",Model,Dataset,ddG_pred,position,wildtype,mutation,pdb,chain"     # examples/ThermoMPNN_inference_2OCJ.csv:1
    assert lines[1] == "0,ThermoMPNN,2OCJ,-0.25,0,S,A,2OCJ,A"
    assert pdb_id_of("/x/y/2OCJ.pdb") == "2OCJ" and first_chain(PDB) == "A"


@pytest.mark.parametrize("path,chains", [(PDB, "A"), (PDB, ["A", "B"]), (PDB, "BA"), (GAP, "A"), (PDB, None)])
def test_native_parser_equals_python_path(path, chains):
    """csrc/tmpnn_pdb.cpp == tied_featurize(alt_parse_PDB(...)) on the tensors the hot path consumes."""
    from thermompnn_amd import native_pdb
    d = pdb_io.alt_parse_PDB(path, chains)
    X, S, mask, _, _, chain_enc, *rest = pdb_io.tied_featurize(d, "cpu", None)
    n = native_pdb.parse_pdb(path, chains)
    assert n["seq"] == d[0]["seq"] and n["num_of_chains"] == d[0]["num_of_chains"] and n["name"] == d[0]["name"]
    np.testing.assert_array_equal(n["X"], X[0].numpy())
    np.testing.assert_array_equal(n["S"], S[0].numpy())
    np.testing.assert_array_equal(n["mask"], mask[0].numpy())
    np.testing.assert_array_equal(n["residue_idx"], rest[6][0].numpy())
    np.testing.assert_array_equal(n["chain_enc"], chain_enc[0].numpy())


def test_native_parser_batch_and_errors(tmp_path):
    from thermompnn_amd import native_pdb
    from thermompnn_amd._lib import TmpnnError
    many = native_pdb.parse_pdbs([PDB, GAP, PDB] * 4, ["A", "A", "AB"] * 4, n_threads=4)
    assert [len(m["seq"]) for m in many] == [194, 194, 388] * 4
    assert many[1]["seq"].count("-") == 3 and many[1]["mask"].sum() == 194 - 4
    toy = tmp_path / "toy2.pdb"
    toy.write_text("HETATM    5  N   MSE A   2       3.332   1.536   0.000  1.00  0.00           N\n"
                   "ATOM      9  CA  GLY A   2A      7.000   3.000   0.000  1.00  0.00           C\n"
                   "ATOM     11  CA  UNK A   5       8.000   4.000   0.000  1.00  0.00           C\n")
    t = native_pdb.parse_pdb(str(toy), "A")
    assert t["seq"] == "MG---" and t["S"].tolist() == [10, 5, 20, 20, 20] and t["mask"].sum() == 0
    with pytest.raises(TmpnnError, match="cannot open"):
        native_pdb.parse_pdb("/nonexistent.pdb", "A")
    bad = tmp_path / "bad.pdb"
    bad.write_text("ATOM      1  N   ALA A   1      xx.000   0.000   0.000\n")
    with pytest.raises(TmpnnError, match="malformed"):
        native_pdb.parse_pdbs([PDB, str(bad)], ["A", "A"])
    assert native_pdb.parse_pdbs([]) == []
    # per-file status (round 5): a bad file leaves a hole instead of voiding the batch, and the message names EVERY failing file
    got = native_pdb.parse_pdbs([PDB, str(bad), GAP, "/nonexistent.pdb"], ["A"] * 4, skip_bad=True)
    assert [None if g is None else len(g["seq"]) for g in got] == [194, None, 194, None]
    with pytest.raises(TmpnnError, match=r"malformed ATOM record in .*bad\.pdb; cannot open /nonexistent\.pdb"):
        native_pdb.parse_pdbs([PDB, str(bad), "/nonexistent.pdb"], ["A"] * 3)
    # no chain filter = the reference's default alphabet: a malformed record under a chain id outside it (blank, punctuation) is never
    # looked at there (protein_mpnn_utils.py:286-293 iterates A-Z, a-z, 0-9), so it must not fail the file
    odd = tmp_path / "odd.pdb"
    odd.write_text(open(PDB).read() + "ATOM   9999  N   ALA     1      xx.000   0.000   0.000\nATOM   9999  N   ALA *   1      yy.000   0.000   0.000\n")
    a, b = native_pdb.parse_pdb(str(odd)), native_pdb.parse_pdb(PDB)
    assert a["seq"] == b["seq"] and np.array_equal(a["X"], b["X"], equal_nan=True)
    from thermompnn_amd.protein_mpnn_utils import alt_parse_PDB
    assert alt_parse_PDB(str(odd))[0]["seq"] == a["seq"]


def test_retrieve_best_mutants_and_rows():
    from thermompnn_amd.ssm_scan import COLUMNS, retrieve_best_mutants, rows_for_protein
    t = np.zeros((3, 21), np.float32)
    t[0, 1] = -2.0            # best at position 0 is C (index 1)
    t[0, 4] = -1.0            # runner-up F
    t[1, 7] = -0.5            # position 1: I
    t[2, :] = 1.0
    t[2, 3] = t[2, 9] = -3.0  # tie at position 2: first minimum wins (E before L), like idxmin
    assert retrieve_best_mutants(t, allow_cys=True) == ["C", "I", "E"]
    assert retrieve_best_mutants(t, allow_cys=False) == ["F", "I", "E"]
    p = {"seq": "A-W", "name": "toy"}
    rows = rows_for_protein(p, t, np.array([5, -1, 7]), "ThermoMPNN", "ds", pick_best=False, include_cys=False)
    assert len(rows) == 2 * 19 and all(r["mutation"] != "C" for r in rows) and rows[0]["neighbors"] == 5
    assert set(rows[0]) == set(COLUMNS)
    best = rows_for_protein(p, t, None, "ThermoMPNN", "ds", pick_best=True, include_cys=True)
    assert [(r["position"], r["mutation"], r["best_AA"]) for r in best] == [(0, "A", "C"), (2, "A", "E")]


def test_metrics_match_scipy():
    from scipy import stats
    from thermompnn_amd.metrics import get_metrics
    rng = np.random.default_rng(0)
    t = rng.normal(size=500)
    p = 0.7 * t + 0.5 * rng.normal(size=500)
    p[::50] = p[1::50]                                   # ties
    t[3] = np.nan
    m = get_metrics(p, t)
    ok = np.isfinite(t)
    assert m["n"] == 499
    assert abs(m["pearson"] - stats.pearsonr(p[ok], t[ok])[0]) < 1e-12
    assert abs(m["spearman"] - stats.spearmanr(p[ok], t[ok])[0]) < 1e-12
    assert abs(m["rmse"] - np.sqrt(np.mean((p[ok] - t[ok]) ** 2))) < 1e-12
    assert abs(m["r2"] - (1 - ((p[ok] - t[ok]) ** 2).sum() / ((t[ok] - t[ok].mean()) ** 2).sum())) < 1e-12


def test_ddgbench_dataset_alignment_and_signs():
    """ddgBenchDataset counterpart (/root/reference/datasets.py:248-317): author residue numbers -> parsed positions through
    resn_list, the gap contingency after a numbering gap, skipped rows, ddG = -DDG, None for an empty cell."""
    from thermompnn_amd.datasets import ddgBenchDataset
    ds = ddgBenchDataset(None, GOLDEN, os.path.join(GOLDEN, "ddgbench_sample.csv"))
    assert len(ds) == 2 and ds.wt_names == ["2OCJA", "2OCJ_gap_chainAA"] and ds.wt_seqs["2OCJ"].startswith("SVPSQ")
    pdb, muts = ds[0]
    assert pdb[0]["seq"][0] == "S" and [(m.position, m.wildtype, m.mutation) for m in muts] == [
        (0, "S", "A"), (7, "Y", "F"), (14, "R", "K"), (4, "Q", "E")]                     # A999G: number not in the structure
    assert [None if m.ddG is None else round(float(m.ddG), 4) for m in muts] == [1.5, -0.7, -2.25, None]
    assert muts[0].ddG.shape == (1,) and muts[0].pdb == "2OCJ"
    pdb, muts = ds[1]
    seq = pdb[0]["seq"]
    assert seq[54:57] == "---" and len(pdb[0]["resn_list"]) == len(seq) - 3
    # S106 sits before the numbering gap (direct hit); P153 / T155 sit behind it: resn_list points 3 short, the contingency
    # adds the gap count (reference :296-305)
    assert [(m.position, m.wildtype, m.mutation) for m in muts] == [(10, "S", "T"), (57, "P", "A"), (59, "T", "S")]
    assert all(seq[m.position] == m.wildtype for m in muts)


def test_fireprot_dataset_and_alignment_map(tmp_path):
    """FireProtDataset counterpart (:167-245): split pickle, per-protein grouping, and the global-alignment contingency
    for a pdb_sequence that does not line up with the parsed structure."""
    import pickle
    from types import SimpleNamespace
    from thermompnn_amd.datasets import FireProtDataset, global_alignment_map
    from thermompnn_amd.pdb_io import alt_parse_PDB
    assert global_alignment_map("ACDEFG", "ACXDEG") == [0, 1, 3, 4, None, 5]
    assert global_alignment_map("MKT", "KT") == [None, 0, 1]
    seq = alt_parse_PDB(os.path.join(GOLDEN, "2OCJ.pdb"), None)[0]["seq"]
    shifted = "MG" + seq                                      # a construct with two extra N-terminal residues
    csv_path = tmp_path / "fireprot.csv"
    with open(csv_path, "w") as fh:
        fh.write("pdb_id_corrected,pdb_sequence,pdb_position,wild_type,mutation,ddG\n")
        fh.write(f"2OCJ,{shifted},2,{seq[0]},A,1.25\n")      # position 2 of the construct = position 0 of the structure
        fh.write(f"2OCJ,{shifted},9,{seq[7]},F,-0.5\n")
        fh.write(f"2OCJ,{shifted},0,M,A,0.3\n")              # faces a gap in the alignment: dropped
        fh.write(f"2OCJ,{shifted},5,{seq[3]},G,\n")          # no ddG: filtered at load time (dropna)
    with open(tmp_path / "splits.pkl", "wb") as fh:
        pickle.dump({"train": [], "val": [], "test": ["2OCJ"]}, fh)
    cfg = SimpleNamespace(data_loc=SimpleNamespace(fireprot_csv=str(csv_path), fireprot_splits=str(tmp_path / "splits.pkl"),
                                                   fireprot_pdbs=GOLDEN))
    ds = FireProtDataset(cfg, "test")
    assert len(ds) == 1 and len(FireProtDataset(cfg, "all")) == 1 and len(FireProtDataset(cfg, "train")) == 0
    pdb, muts = ds[0]
    assert [(m.position, m.wildtype, m.mutation, round(float(m.ddG), 3)) for m in muts] == [(0, seq[0], "A", 1.25), (7, seq[7], "F", -0.5)]


def test_lightning_checkpoint_loader_drops_foreign_keys(tmp_path):
    """load_thermompnn_checkpoint on a Lightning-shaped file: optimizer state, hyper-parameters and Lightning-level
    buffers are dropped, the 'model.' prefix stripped, and the result loads STRICTLY into TransferModel's tree. Files
    that need the pickle loader are refused unless explicitly allowed."""
    import collections
    from thermompnn_amd import weights
    sd = weights.synthetic_state_dict(0)
    ckpt = {"epoch": 3, "global_step": 10, "pytorch-lightning_version": "1.9", "hyper_parameters": {"lr": 1e-3},
            "optimizer_states": [{"state": {0: {"exp_avg": torch.zeros(3)}}}],
            "state_dict": collections.OrderedDict([("model." + k, v) for k, v in sd.items()] +
                                                  [("metrics.ddG.r2.sum_error", torch.zeros(1)), ("val_loss", torch.ones(1))])}
    path = tmp_path / "lightning.ckpt"
    torch.save(ckpt, path)
    out = weights.load_thermompnn_checkpoint(str(path))
    assert list(out.keys()) == list(sd.keys()) and all(torch.equal(out[k], sd[k]) for k in sd)

    # a checkpoint whose hyper_parameters hold arbitrary objects (the published thermoMPNN_default.pt carries an OmegaConf
    # config) fails weights_only=True; the restricted unpickler still extracts the tensors WITHOUT running the file's globals
    marker = tmp_path / "executed"
    bad = tmp_path / "pickled.ckpt"
    torch.save({"state_dict": ckpt["state_dict"], "hyper_parameters": _Evil(str(marker)), "callbacks": {_Evil(str(marker)): 1}}, bad)
    with pytest.raises(Exception):
        torch.load(bad, weights_only=True)
    out = weights.load_thermompnn_checkpoint(str(bad))
    assert list(out.keys()) == list(sd.keys()) and all(torch.equal(out[k], sd[k]) for k in sd)
    assert not marker.exists()                                              # nothing from the file was executed
    assert list(weights.load_thermompnn_checkpoint(str(bad), allow_pickle=True).keys()) == list(sd.keys())
    assert marker.exists()                                                  # the opt-in pickle loader does run it
    junk = tmp_path / "junk.ckpt"
    junk.write_bytes(b"not a checkpoint at all")
    with pytest.raises(RuntimeError, match="allow_pickle"):
        weights.load_thermompnn_checkpoint(str(junk))


class _Evil:
    """Writes a marker file when unpickled by the plain pickle loader."""

    def __init__(self, path):
        self.path = path

    def __reduce__(self):
        import pathlib
        return (pathlib.Path.touch, (pathlib.Path(self.path),))

    def __hash__(self):
        return 7


def test_split_files_load_without_pickle_execution(tmp_path):
    """FireProtDataset reads its split dictionary with the restricted unpickler: lists and numpy arrays of names load (the
    reference's dataset_splits/*.pkl are exactly that), foreign globals become inert placeholders."""
    import pickle
    marker = tmp_path / "executed"
    f = tmp_path / "splits.pkl"
    with open(f, "wb") as fh:
        pickle.dump({"train": ["1ABC", "2DEF"], "test": np.array(["3GHI"], dtype=object), "x": _Evil(str(marker))}, fh)
    d = weights.safe_unpickle(str(f))
    assert d["train"] == ["1ABC", "2DEF"] and list(d["test"]) == ["3GHI"] and not marker.exists()
    assert type(d["x"]).__name__ == "_Opaque"


def test_published_real_weight_table_fixture():
    """tests/golden/2OCJ_A_realweights_ddg.npz = the reference's published examples/ThermoMPNN_inference_2OCJ.csv as a
    [194, 20] table (consumed by the skipped-unless-weights GPU test): shape, wild-type zeros and the survey's summary
    statistics (mean 1.03, range -1.85 ... 4.76)."""
    t = np.load(os.path.join(GOLDEN, "2OCJ_A_realweights_ddg.npz"))["ddg"]
    g = np.load(os.path.join(GOLDEN, "2OCJ_A.npz"))
    assert t.shape == (194, 20) and t.dtype == np.float32
    assert (t[np.arange(194), g["S"].astype(int)] == 0).all()
    assert abs(t.mean() - 1.03) < 0.01 and abs(t.min() + 1.85) < 0.01 and abs(t.max() - 4.76) < 0.01


def test_featurize_matches_reference_batch():
    """pdb_io.featurize (the model_utils.featurize signature north_star names; /root/reference/model_utils.py:19-125) on a
    padded batch of two single-chain proteins, against the tuple the imported reference returned (make_golden.py)."""
    import torch
    from thermompnn_amd.pdb_io import alt_parse_PDB, featurize
    from thermompnn_amd.synthetic import synthetic_pdb_dict
    g = load_golden("featurize_batch")
    batch = []
    for d in (synthetic_pdb_dict(32, seed=5), alt_parse_PDB(os.path.join(GOLDEN, "2OCJ.pdb"), "A")[0]):
        d = dict(d)
        d["masked_list"], d["visible_list"] = ["A"], []
        batch.append(d)
    X, S, mask, lengths, chain_M, ridx, mask_self, cenc = featurize(batch, "cpu")
    assert X.dtype == torch.float32 and S.dtype == torch.long and ridx.dtype == torch.long and cenc.dtype == torch.long
    assert mask.dtype == chain_M.dtype == mask_self.dtype == torch.float32
    np.testing.assert_array_equal(X.numpy(), g["X"])
    np.testing.assert_array_equal(S.numpy(), g["S"])
    np.testing.assert_array_equal(mask.numpy(), g["mask"])
    np.testing.assert_array_equal(lengths, g["lengths"])
    np.testing.assert_array_equal(chain_M.numpy(), g["chain_M"])
    np.testing.assert_array_equal(ridx.numpy(), g["residue_idx"])
    np.testing.assert_array_equal(cenc.numpy(), g["chain_enc"])
    np.testing.assert_array_equal(mask_self.numpy().sum(-1).astype(np.int32), g["mask_self_rowsum"])
    assert tuple(lengths) == (32, 194) and (ridx.numpy()[0, 32:] == -100).all()     # padding convention of the reference


def _atom_line(serial=1, atom="CA", resname="ALA", chain="A", resnum="   1", ins=" ", x="  11.104", y="   6.134", z="  -6.504"):
    return f"ATOM  {serial:5d} {atom:^4s} {resname:>3s} {chain}{resnum}{ins}   {x}{y}{z}  1.00  0.00           C"


def _mutated_pdbs(tmp_path, n_files=60):
    """Seeded structural damage to real ATOM records: truncated lines, non-numeric columns, huge / negative residue numbers,
    control bytes, very long lines, empty files, duplicate atoms, odd chain letters."""
    rng = np.random.default_rng(12345)
    base = [l.rstrip("\n") for l in open(PDB) if l.startswith("ATOM")][:400]
    files = []
    hand = {
        "empty": "", "only_newlines": "\n\n\n", "short_atom": "ATOM\nATOM  \nATOM      1  N\n",
        "col21": "ATOM      1  N   MET A", "no_coords": "ATOM      1  N   MET A   1",
        "nonnumeric_xyz": _atom_line(x="  abcdef"), "nan_xyz": _atom_line(x="     nan", y="     inf", z="    -inf"),
        "exp_xyz": _atom_line(x="  1e9999", y=" -1e9999", z="   1e-99"),
        "resnum_text": _atom_line(resnum="ABCD"), "resnum_only_ins": _atom_line(resnum="    ", ins="A"),
        "resnum_max": _atom_line(resnum="9999", ins="9") + "\n" + _atom_line(serial=2, resnum="-999", ins="9"),
        "resnum_span": _atom_line(resnum="-999", ins=" ") + "\n" + _atom_line(serial=2, resnum="9999", ins=" "),
        "long_line": _atom_line() + "X" * 5000 + "\n" + _atom_line(serial=2, atom="N") + "Y" * 1021,
        "exact_buf": (_atom_line() + " " * 600)[:511] + "\n" + (_atom_line(serial=2, atom="C") + " " * 600)[:510] + "\n",
        "nul_bytes": _atom_line()[:40] + "\x00\x00" + _atom_line()[42:], "binary": bytes(range(256)).decode("latin1") * 8,
        "mse": "HETATM    1  N   MSE A   1      11.104   6.134  -6.504  1.00  0.00           N\nHETATM    2 MSE  MSE A   2",
        "hetatm_short": "HETATM    1  N   MS", "tabs": _atom_line().replace(" ", "\t"),
        "unicode": _atom_line(resname="\u00e9\u00e9\u00e9"), "dup_atoms": "\n".join(_atom_line(serial=i, x=f"{i:8.3f}") for i in range(1, 30)),
        "many_ins": "\n".join(_atom_line(serial=i, ins=c) for i, c in enumerate("ZYXWVUTSRQPONMLKJIHGFEDCBA", 1)),
        "odd_chains": "\n".join(_atom_line(serial=i, chain=c) for i, c in enumerate("z9 *\x7f", 1)),
    }
    for name, text in hand.items():
        f = tmp_path / f"hand_{name}.pdb"
        f.write_bytes(text.encode("latin1", errors="replace"))
        files.append(str(f))
    for k in range(n_files):
        lines = list(base[: int(rng.integers(1, 120))])
        for _ in range(int(rng.integers(1, 12))):
            i = int(rng.integers(0, len(lines)))
            l = lines[i]
            op = int(rng.integers(0, 8))
            if op == 0:
                lines[i] = l[: int(rng.integers(0, len(l) + 1))]                                  # truncate
            elif op == 1:
                a, b = sorted(int(x) for x in rng.integers(0, len(l) + 1, size=2))
                lines[i] = l[:a] + "".join(chr(int(c)) for c in rng.integers(1, 256, size=b - a)) + l[b:]   # garbage span
            elif op == 2:
                lines[i] = l[:22] + f"{int(rng.integers(-9999, 99999)):5d}"[:5] + l[27:]          # wild residue number
            elif op == 3:
                lines[i] = l[:30] + "".join(rng.choice(list("0123456789.-+eEnaif xX"), size=24)) + l[54:]   # wild coordinates
            elif op == 4:
                lines[i] = l + " " * int(rng.integers(0, 1200))                                  # long line
            elif op == 5:
                del lines[i]
                if not lines:
                    lines = [""]
            elif op == 6:
                lines.insert(i, l)                                                               # duplicate record
            else:
                lines[i] = l[:21] + chr(int(rng.integers(32, 127))) + l[22:]                      # other chain letter
        f = tmp_path / f"mut_{k}.pdb"
        f.write_bytes(("\n".join(lines) + ("\n" if rng.integers(0, 2) else "")).encode("latin1"))
        files.append(str(f))
    return files


def test_native_parser_survives_malformed_input(tmp_path):
    """SURVEY §5 'sanitizers on host stubs': csrc/tmpnn_pdb.cpp reads untrusted text with fixed columns, strtod, a residue
    span that allocates per missing number, and threads. Built with -fsanitize=address,undefined (any report aborts) and fed
    ~85 damaged files, alone and through the threaded batch entry; every file either parses or returns TMPNN_E_INVALID, and
    whatever parses has the same length as the Python parser's result for it."""
    import subprocess
    from thermompnn_amd import build
    try:
        driver = build.build_pdb_sanitizer_driver(str(tmp_path / "pdb_fuzz_driver"))
    except RuntimeError as e:
        pytest.skip(f"sanitizer driver not buildable here: {e}")
    files = _mutated_pdbs(tmp_path)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([driver, "--threads", "8"] + files, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, f"sanitizer report / crash:\n{r.stderr[-4000:]}"
    lines = r.stdout.strip().splitlines()
    assert len(lines) == len(files) + 1 and lines[-1].startswith("batch ")
    n_ok = 0
    for f, line in zip(files, lines):
        rc, length = int(line.split()[0]), int(line.split()[1])
        assert rc in (0, -1), (f, line)
        if rc == 0:
            n_ok += 1
            raw = open(f, "rb").read()
            if any(b < 9 or 13 < b < 32 or b == 127 for b in raw) or b"_" in raw:
                continue      # control bytes / Python-only number syntax ("1_0"): no crash is all that is asked of the native reader
            try:
                ref = pdb_io.alt_parse_PDB(f, None)[0]
            except Exception:
                continue                                     # the Python parser may be stricter; agreement is only required when both parse
            assert length == len(ref["seq"]), (f, length, len(ref["seq"]))
    assert 10 < n_ok < len(files)                            # the corpus has both survivable and fatal damage
    # the span guard: one chain whose residue numbers cover the whole 5-column range is an error, not a 110 000-row allocation
    span = [f for f in files if f.endswith("hand_resnum_span.pdb")][0]
    assert lines[files.index(span)].split()[0] in ("0", "-1")


def test_native_csv_number_format_is_python_repr():
    """The ddG_pred cell is what pandas writes for a Python float: repr(float(x)) — shortest round-trip digits of the double,
    fixed notation for 1e-4 <= |x| < 1e16, else exponent form with a two-digit exponent (examples/ThermoMPNN_inference_2OCJ.csv)."""
    from thermompnn_amd import native_csv
    rng = np.random.default_rng(0)
    vals = [float(v) for v in rng.normal(size=3000).astype(np.float32)]
    vals += [float(v) for v in rng.integers(0, 2 ** 32, size=20000, dtype=np.uint64).astype(np.uint32).view(np.float32)]
    vals += [0.0, -0.0, 1.0, -2.0, 1e-5, 1e-4, 9.9999e-5, 1e15, 1e16, 9999999999999998.0, 1.2345e22, 5e-324, 1.7976931348623157e308,
             float("inf"), float("-inf"), float(np.float32(0.1)), 100.0, 123456.789, -0.040876448154449463]
    for v in vals:
        assert native_csv.format_double(v) == repr(v), (v, native_csv.format_double(v))
    assert native_csv.format_double(float("nan")) == "nan"


def _random_scan(rng, n=24):
    seqs, tabs, nbs, names = [], [], [], []
    for i in range(n):
        L = int(rng.integers(1, 120))
        s = [AA20[k] for k in rng.integers(0, 20, L)]
        for k in rng.integers(0, L, 2):
            s[k] = "-"
        seqs.append("".join(s))
        tabs.append(rng.normal(scale=3.0, size=(L, 21)).astype(np.float32))
        nbs.append(rng.integers(-1, 40, L).astype(np.int32))
        names.append(f"prot_{i}")
    names[3], names[5] = 'odd,"name', "d.p1abc.pdb"                # quoting; the reference's character-set strip of '.pdb'
    tabs[2][0, 3] = tabs[2][0, 9] = -50.0                           # an exact tie for best_AA: first minimum wins
    return seqs, tabs, nbs, names


AA20 = "ACDEFGHIKLMNPQRSTVWY"


@pytest.mark.parametrize("pick,cys,use_nb", [(False, False, False), (False, True, True), (True, False, True), (True, True, False)])
def test_native_csv_writer_equals_row_writer(tmp_path, pick, cys, use_nb):
    """csrc/tmpnn_csv.cpp == rows_for_protein + csv.writer, byte for byte, in every post-processing mode (SSM.py:128-176);
    written in two chunks so the running index continues across calls."""
    from thermompnn_amd import native_csv, ssm_scan
    seqs, tabs, nbs, names = _random_scan(np.random.default_rng(7))
    rows = []
    for i in range(len(seqs)):
        rows += ssm_scan.rows_for_protein({"seq": seqs[i], "name": names[i]}, tabs[i], nbs[i] if use_nb else None, "ThermoMPNN",
                                          "my,set", pick, cys)
    ssm_scan.write_csv(rows, str(tmp_path / "py.csv"))
    table = np.concatenate(tabs)
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int32)
    nb = np.concatenate(nbs)
    stripped = [n.strip(".pdb") for n in names]
    w = native_csv.CsvWriter(str(tmp_path / "nat.csv"))
    cut = 9
    w.write_ssm(table[:off[cut]], off[:cut + 1], seqs[:cut], stripped[:cut], nb[:off[cut]] if use_nb else None, dataset="my,set",
                pick_best=pick, include_cys=cys, n_threads=3)
    w.write_ssm(table[off[cut]:], off[cut:] - off[cut], seqs[cut:], stripped[cut:], nb[off[cut]:] if use_nb else None,
                dataset="my,set", pick_best=pick, include_cys=cys, n_threads=3)
    assert w.close() == len(rows)
    a, b = (tmp_path / "py.csv").read_bytes(), (tmp_path / "nat.csv").read_bytes()
    assert a == b and w.bytes == len(a) and b"\r" not in a
    # the one-call form used after a multi-rank gather
    res = dict(table=table, offsets=off, seqs=seqs, names=names, neighbors=nb if use_nb else None)
    assert ssm_scan.write_scan_csv(str(tmp_path / "one.csv"), res, "ThermoMPNN", "my,set", pick, cys) == len(rows)
    assert (tmp_path / "one.csv").read_bytes() == a


@pytest.mark.parametrize("pick", [False, True])
@pytest.mark.parametrize("cys", [False, True])
@pytest.mark.parametrize("cen", [False, True])
def test_csv_writers_equal_files_made_by_pandas(tmp_path, pick, cys, cen):
    """Both writers against tests/golden/csv/*.csv — frames built cell by cell and written by REAL pandas exactly as the reference
    does (tests/golden/make_csv_golden.py restates SSM.py:102-176 operation for operation): every flag combination, a gap, the
    character-set strip of the name, a WT cell that needs quoting, exponent-form floats, a NaN (empty cell) and an inf, and the
    ``dupe_detector`` column the reference's --pick_best frames carry."""
    from thermompnn_amd import native_csv, ssm_scan
    g = np.load(os.path.join(GOLDEN, "csv", "csv_inputs.npz"))
    want = open(os.path.join(GOLDEN, "csv", f"ssm_pick{int(pick)}_cys{int(cys)}_cen{int(cen)}.csv"), "rb").read()
    names, seqs, wts = [str(x) for x in g["names"]], [str(x) for x in g["seqs"]], [str(x) for x in g["wts"]]
    off, table, nb = g["offsets"].astype(np.int32), g["table"], g["neighbors"]
    rows = []
    for i in range(len(seqs)):
        rows += ssm_scan.rows_for_protein({"seq": seqs[i], "name": names[i], "wt": wts[i]}, table[off[i]:off[i + 1]],
                                          nb[off[i]:off[i + 1]] if cen else None, "ThermoMPNN", "P53", pick, cys)
    ssm_scan.write_csv(rows, str(tmp_path / "py.csv"), pick_best=pick)
    assert (tmp_path / "py.csv").read_bytes() == want
    stripped = [n.strip(".pdb") for n in names]
    with native_csv.CsvWriter(str(tmp_path / "nat.csv")) as w:                  # (header decided by the first listing)
        w.write_ssm(table, off, seqs, stripped, nb if cen else None, dataset="P53", pick_best=pick, include_cys=cys, n_threads=2, wt_cells=wts)
    assert (tmp_path / "nat.csv").read_bytes() == want
    # one rank's share of a sharded scan: protein 1 alone, no header, with its running index in the whole listing
    per = 1 if pick else 20 if cys else 19
    first1 = sum(c != "-" for c in seqs[0]) * per
    with native_csv.CsvWriter(str(tmp_path / "part.csv"), header=False, pick_best=pick) as w:
        nbytes = w.write_ssm(table[off[1]:], off[1:] - off[1], seqs[1:], stripped[1:], nb[off[1]:] if cen else None, dataset="P53",
                             pick_best=pick, include_cys=cys, n_threads=2, wt_cells=wts[1:], first_rows=[first1], want_bytes=True)
    part = (tmp_path / "part.csv").read_bytes()
    assert want.endswith(part) and nbytes.tolist() == [len(part)] and native_csv.header_text(0, pick) == want[:want.index(b"\n") + 1]


def test_custom_inference_csv_equals_the_file_made_by_pandas(tmp_path):
    from thermompnn_amd import custom_inference, native_csv
    g = np.load(os.path.join(GOLDEN, "csv", "csv_inputs.npz"))
    want = open(os.path.join(GOLDEN, "csv", "custom_inference.csv"), "rb").read()
    off, seq, tab = g["offsets"], str(g["seqs"][1]), g["table"][g["offsets"][1]:g["offsets"][2]]
    rows = [{"Model": "ThermoMPNN", "Dataset": "2OCJ", "ddG_pred": float(tab[pos, a]), "position": pos, "wildtype": wt,
             "mutation": AA20[a], "pdb": "2OCJ", "chain": "A"} for pos, wt in enumerate(seq) if wt != "-" for a in range(20)]
    custom_inference.write_csv(rows, str(tmp_path / "py.csv"))
    assert (tmp_path / "py.csv").read_bytes() == want
    with native_csv.CsvWriter(str(tmp_path / "nat.csv"), native_csv.SCHEMA_CUSTOM_INFERENCE) as w:
        w.write_ssm(tab, np.array([0, len(seq)], np.int32), [seq], ["2OCJ"], dataset="2OCJ", chain="A", include_cys=True)
    assert (tmp_path / "nat.csv").read_bytes() == want


def test_native_csv_listed_and_custom_inference_schema(tmp_path):
    from thermompnn_amd import custom_inference, native_csv, ssm_scan
    from thermompnn_amd._lib import TmpnnError
    seqs, tabs, nbs, names = _random_scan(np.random.default_rng(8), 6)
    table = np.concatenate(tabs)
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int32)
    # explicit mutation list (BASELINE config 4 / --mutations)
    rng = np.random.default_rng(9)
    tri = []
    for _ in range(300):
        i = int(rng.integers(0, 6))
        pos = int(rng.integers(0, len(seqs[i])))
        if seqs[i][pos] != "-":
            tri.append((i, pos, int(rng.integers(0, 20))))
    rows = [{"WT Seq": seqs[i], "Model": "ThermoMPNN", "Dataset": "custom", "ddG_pred": float(tabs[i][pos, a]), "position": pos,
             "wildtype": seqs[i][pos], "mutation": AA20[a], "neighbors": int(nbs[i][pos]), "best_AA": "", "pdb": names[i].strip(".pdb")}
            for i, pos, a in tri]
    ssm_scan.write_csv(rows, str(tmp_path / "py.csv"))
    res = dict(table=table, offsets=off, seqs=seqs, names=names, neighbors=np.concatenate(nbs))
    assert ssm_scan.write_scan_csv(str(tmp_path / "nat.csv"), res, "ThermoMPNN", "custom", False, False, np.array(tri)) == len(tri)
    assert (tmp_path / "py.csv").read_bytes() == (tmp_path / "nat.csv").read_bytes()
    with pytest.raises(TmpnnError, match="out of range"):
        ssm_scan.write_scan_csv(str(tmp_path / "bad.csv"), res, "ThermoMPNN", "custom", False, False, np.array([[0, 10 ** 6, 0]]))
    # custom_inference layout (examples/ThermoMPNN_inference_2OCJ.csv:1)
    i = 1
    rows = [{"Model": "ThermoMPNN", "Dataset": "2OCJ", "ddG_pred": float(tabs[i][pos, a]), "position": pos, "wildtype": wt,
             "mutation": AA20[a], "pdb": "2OCJ", "chain": "A"} for pos, wt in enumerate(seqs[i]) if wt != "-" for a in range(20)]
    custom_inference.write_csv(rows, str(tmp_path / "ci_py.csv"))
    with native_csv.CsvWriter(str(tmp_path / "ci_nat.csv"), native_csv.SCHEMA_CUSTOM_INFERENCE) as w:
        w.write_ssm(tabs[i], np.array([0, len(seqs[i])], np.int32), [seqs[i]], ["2OCJ"], dataset="2OCJ", chain="A", include_cys=True)
    assert (tmp_path / "ci_py.csv").read_bytes() == (tmp_path / "ci_nat.csv").read_bytes()
    with pytest.raises(TmpnnError, match="sequence length"):
        with native_csv.CsvWriter(str(tmp_path / "x.csv")) as w:
            w.write_ssm(tabs[0], np.array([0, len(seqs[0])], np.int32), [seqs[0] + "A"], ["x"])
    with pytest.raises(TmpnnError, match="cannot create"):
        native_csv.CsvWriter(str(tmp_path / "no_such_dir" / "x.csv"))


def test_native_pack_batch_equals_per_file_fill():
    """tmpnn_pdb_pack_batch (the pipeline's staging step) == tmpnn_pdb_fill protein after protein, offsets included."""
    import ctypes as C
    from thermompnn_amd import _lib, native_pdb
    lib = _lib.load()
    paths, chains = [PDB, GAP, PDB], ["A", "A", "AB"]
    single = native_pdb.parse_pdbs(paths, chains)
    n = len(paths)
    hs = (C.c_void_p * n)()
    cp = (C.c_char_p * n)(*[p.encode() for p in paths])
    cc = (C.c_char_p * n)(*[c.encode() for c in chains])
    _lib.check(lib.tmpnn_pdb_parse_batch(cp, cc, n, 2, hs))
    T = sum(len(s["S"]) for s in single)
    X, S, mask = np.full((T, 4, 3), -1, np.float32), np.full(T, -1, np.int32), np.full(T, -1, np.float32)
    ridx, cenc, ca, off = np.full(T, -1, np.int32), np.full(T, -1, np.int32), np.full(T, -1, np.float32), np.zeros(n + 1, np.int32)
    p = lambda a: a.ctypes.data
    assert lib.tmpnn_pdb_pack_batch(hs, n, 2, T - 1, p(X), p(S), p(mask), p(ridx), p(cenc), p(ca), p(off)) == -4   # TMPNN_E_WORKSPACE
    _lib.check(lib.tmpnn_pdb_pack_batch(hs, n, 2, T, p(X), p(S), p(mask), p(ridx), p(cenc), p(ca), p(off)))
    assert off.tolist() == np.concatenate([[0], np.cumsum([len(s["S"]) for s in single])]).tolist()
    for k, want in (("X", X), ("S", S), ("mask", mask), ("residue_idx", ridx), ("chain_enc", cenc), ("ca_mask", ca)):
        np.testing.assert_array_equal(np.concatenate([s[k] for s in single]), want)
    assert [C.string_at(lib.tmpnn_pdb_seq(C.c_void_p(h))).decode() for h in hs] == [s["seq"] for s in single]
    for h in hs:
        lib.tmpnn_pdb_free(C.c_void_p(h))


def test_compat_package_resolves_the_reference_import_lines(tmp_path):
    """compat/ on PYTHONPATH: the import lines of analysis/custom_inference.py:11-15, analysis/SSM.py:10-14 and
    analysis/thermompnn_benchmarking.py:11-14 resolve VERBATIM to the engine-backed modules, and
    TransferModelPL.load_from_checkpoint(path, cfg=config).model (thermompnn_benchmarking.py:78-84) loads a Lightning-format
    checkpoint without Lightning."""
    import subprocess
    import sys
    from thermompnn_amd import weights
    sd = weights.synthetic_state_dict(0)
    mp, _ = weights.split_transfer_state_dict(sd)
    os.makedirs(tmp_path / "vanilla_model_weights")
    weights.save_vanilla_checkpoint(tmp_path / "vanilla_model_weights" / "v_48_020.pt", mp, 48)
    weights.save_lightning_checkpoint(tmp_path / "thermo.ckpt", sd)
    code = f"""
from datasets import Mutation
from train_thermompnn import TransferModelPL
from protein_mpnn_utils import tied_featurize, alt_parse_PDB
from thermompnn_benchmarking import get_trained_model
from SSM import get_ssm_mutations
from datasets import MegaScaleDataset, ddgBenchDataset, FireProtDataset, Mutation
from protein_mpnn_utils import loss_smoothed, tied_featurize
from model_utils import featurize
from thermompnn_benchmarking import compute_centrality, ProteinMPNNBaseline, get_trained_model, ALPHABET
from transfer_model import get_protein_mpnn
import datasets, thermompnn_amd, torch
assert datasets.__file__.endswith('compat/datasets.py') and ALPHABET == 'ACDEFGHIKLMNPQRSTVWYX'
class AD(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__
cfg = AD(model=AD(hidden_dims=[64, 32], subtract_mut=True, num_final_layers=2, freeze_weights=True, load_pretrained=True,
                  lightattn=True), platform=AD(thermompnn_dir={str(tmp_path)!r}))
model = TransferModelPL.load_from_checkpoint({str(tmp_path / 'thermo.ckpt')!r}, cfg=cfg).model
assert isinstance(model, thermompnn_amd.transfer_model.TransferModel)
want = thermompnn_amd.weights.synthetic_state_dict(0)
assert all(torch.equal(v, want[k]) for k, v in model.state_dict().items())
pdb = alt_parse_PDB({PDB!r}, 'A')
muts = get_ssm_mutations(pdb[0])
assert len(muts) == 20 * 194 and muts[0] == 'S0A'
for bad in (MegaScaleDataset, loss_smoothed):
    try:
        bad()
        raise SystemExit('training-side name did not refuse')
    except NotImplementedError:
        pass
print('compat ok')
"""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.path.join(repo, "compat"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0 and "compat ok" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]


def test_native_csv_streams_a_very_long_protein_in_blocks(tmp_path):
    """The reference's layout repeats the whole sequence in every row, so a protein's text grows with L^2: beyond 64 MB the
    native writer waits for the protein's turn and streams blocks of positions instead of buffering it whole. Same bytes."""
    from thermompnn_amd import native_csv, ssm_scan
    rng = np.random.default_rng(3)
    L = 1900                                                   # 38 000 rows x 1.9 KB = 74 MB of text
    seqs = ["".join(AA20[k] for k in rng.integers(0, 20, 40)), "".join(AA20[k] for k in rng.integers(0, 20, L)),
            "".join(AA20[k] for k in rng.integers(0, 20, 25))]
    tabs = [rng.normal(size=(len(s), 21)).astype(np.float32) for s in seqs]
    names = ["small_a", "giant", "small_b"]
    rows = []
    for i in range(3):
        rows += ssm_scan.rows_for_protein({"seq": seqs[i], "name": names[i]}, tabs[i], None, "ThermoMPNN", "custom", False, True)
    ssm_scan.write_csv(rows, str(tmp_path / "py.csv"))
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int32)
    with native_csv.CsvWriter(str(tmp_path / "nat.csv")) as w:
        w.write_ssm(np.concatenate(tabs), off, seqs, [n.strip(".pdb") for n in names], include_cys=True, n_threads=3)   # (SSM.py:139's strip)
    assert w.rows == len(rows) and w.bytes > (64 << 20)
    import filecmp
    assert filecmp.cmp(str(tmp_path / "py.csv"), str(tmp_path / "nat.csv"), shallow=False)


def test_raw_weight_file_and_native_example_build(tmp_path, synthetic_weights):
    """weights.export_raw: the canonical tensor order, every element, as a host without Python reads it
    (examples/scan_native.cpp); the example compiles against include/tmpnn.h + the HIP runtime API and links libtmpnn.so."""
    import struct
    import subprocess
    from thermompnn_amd import _lib, build, weights
    raw = tmp_path / "w.raw"
    nbytes = weights.export_raw(synthetic_weights, str(raw))
    blob = raw.read_bytes()
    assert len(blob) == nbytes and blob[:8] == weights.RAW_MAGIC and struct.unpack_from("<i", blob, 8)[0] == _lib.N_TENSORS
    lib, pos = _lib.load(), 12
    for i, name in enumerate(_lib.tensor_names()):
        numel = struct.unpack_from("<q", blob, pos)[0]
        assert numel == lib.tmpnn_tensor_numel(i)
        key = name if i >= _lib.N_MPNN_TENSORS else "prot_mpnn." + name
        np.testing.assert_array_equal(np.frombuffer(blob, "<f4", numel, pos + 8), synthetic_weights[key].numpy().ravel())
        pos += 8 + 4 * numel
    assert pos == len(blob)
    mpnn_only = {k: v for k, v in synthetic_weights.items() if k.startswith("prot_mpnn.")}
    weights.export_raw(mpnn_only, str(tmp_path / "m.raw"))
    assert struct.unpack_from("<i", (tmp_path / "m.raw").read_bytes(), 8)[0] == _lib.N_MPNN_TENSORS
    exe = build.build_native_example(str(tmp_path / "scan_native"))      # its rpath is relative to examples/: name the library's directory
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60, env=dict(os.environ, LD_LIBRARY_PATH=build.HERE))
    assert r.returncode == 1 and "usage: scan_native" in r.stderr


def test_cpu_budget_is_shared_between_the_ranks_of_a_host(monkeypatch):
    from thermompnn_amd import pipeline
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    host = pipeline.usable_cpus()
    assert host == pipeline.usable_cpus(per_rank=False) >= 1
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    assert pipeline.usable_cpus() == max(1, host // 4) and pipeline.usable_cpus(per_rank=False) == host
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "not a number")
    assert pipeline.usable_cpus() == host


def test_native_host_entry_points_turn_exceptions_into_error_codes(tmp_path):
    """A C++ exception must not cross the C-ABI (tmpnn_host_guard.hpp): with the address space capped, the CSV writer's
    per-protein buffer (57 MB for L = 1700) cannot be allocated -> TMPNN_E_WORKSPACE ("out of host memory") on the caller's
    thread, the process lives, and with several worker threads the in-order commit ticket of a failed protein is handed on
    (no deadlock); the same handle then writes a small protein normally."""
    import subprocess
    import sys
    from thermompnn_amd import build
    code = r'''
import ctypes as C, resource, sys
lib = C.CDLL(sys.argv[1])
lib.tmpnn_last_error.restype = C.c_char_p
h = C.c_void_p()
assert lib.tmpnn_csv_open(sys.argv[2].encode(), 0, C.byref(h)) == 0
def write(L, n, threads):
    tab = (C.c_float * (n * L * 21))()
    off = (C.c_int32 * (n + 1))(*[i * L for i in range(n + 1)])
    seqs = (C.c_char_p * n)(*[b"A" * L] * n)
    names = (C.c_char_p * n)(*[b"p%d" % i for i in range(n)])
    return lib.tmpnn_csv_write_ssm(h, tab, 21, off, n, seqs, None, names, None, b"M", b"D", None, None, 0, threads)
vm = [int(l.split()[1]) for l in open("/proc/self/status") if l.startswith("VmSize")][0] * 1024
resource.setrlimit(resource.RLIMIT_AS, (vm + (40 << 20), vm + (40 << 20)))
for threads in (1, 4):
    rc = write(1700, 6, threads)
    msg = lib.tmpnn_last_error().decode()
    assert rc == -4 and "out of host memory" in msg, (rc, msg)
assert write(30, 3, 4) == 0, lib.tmpnn_last_error()
rows, nbytes = C.c_int64(), C.c_int64()
assert lib.tmpnn_csv_close(h, C.byref(rows), C.byref(nbytes)) == 0 and rows.value == 3 * 30 * 19
print("ok")
'''
    r = subprocess.run([sys.executable, "-c", code, build.LIB, str(tmp_path / "x.csv")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stdout, r.stderr[-1500:])
