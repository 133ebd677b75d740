"""PDB files -> ddG tables -> CSV / binary at engine speed: the three-stage host
pipeline (thermompnn_amd/pipeline.py) and the native columnar writer against the per-protein, per-row reference-shaped path."""
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine(synthetic_weights):
    from thermompnn_amd.engine import Engine
    return Engine(synthetic_weights, "cuda:0", 48)


def _pdb_set(tmp_path, n=23, seed=11):
    from thermompnn_amd.synthetic import backbone_pdb_text, synthetic_backbone
    rng = np.random.default_rng(seed)
    paths = []
    for i, L in enumerate(rng.integers(20, 200, size=n)):
        X, seq = synthetic_backbone(int(L), 9000 + i)
        p = tmp_path / f"syn_{i:03d}.pdb"
        p.write_text(backbone_pdb_text(X, seq))
        paths.append(str(p))
    paths.insert(3, os.path.join(GOLDEN, "2OCJ.pdb"))
    paths.insert(9, os.path.join(GOLDEN, "2OCJ_gap_chainA.pdb"))     # numbering gaps + a residue without its N atom
    return paths


def _reference_shaped_rows(engine, paths, centrality, pick_best, include_cys):
    """One protein per forward, one dict per row: the shape of analysis/SSM.py:105-166 on the same engine."""
    from thermompnn_amd import native_pdb, ssm_scan
    rows, tables = [], []
    for path in paths:
        p = native_pdb.parse_pdb(path, "A")
        L = len(p["seq"])
        off = torch.tensor([0, L], dtype=torch.int32)
        t = engine.ssm_forward(p["X"], p["S"], p["mask"], p["residue_idx"], p["chain_enc"], off)["ddg"].cpu().numpy()
        nb = engine.centrality(p["X"], p["ca_mask"], off).cpu().numpy() if centrality else None
        tables.append(t)
        rows += ssm_scan.rows_for_protein(p, t, nb, "ThermoMPNN", "custom", pick_best, include_cys)
    return rows, tables


@pytest.mark.parametrize("centrality,pick_best,include_cys", [(False, False, False), (True, True, False), (True, False, True)])
def test_pipeline_csv_is_byte_identical_to_the_row_writer(tmp_path, engine, centrality, pick_best, include_cys):
    from thermompnn_amd import ssm_scan
    paths = _pdb_set(tmp_path)
    rows, _ = _reference_shaped_rows(engine, paths, centrality, pick_best, include_cys)
    ssm_scan.write_csv(rows, str(tmp_path / "rows.csv"))
    out = str(tmp_path / "pipe.csv")
    # 7 files per chunk, at most 600 residues per forward: chunks split, several slots in flight, the running index continues
    n, stats = ssm_scan.scan_to_file(engine, paths, ["A"] * len(paths), out, pick_best=pick_best, include_cys=include_cys,
                                     centrality=centrality, chunk_files=7, chunk_residues=600, parse_threads=3)
    assert n == len(rows) and stats.files == len(paths) and stats.chunks >= 4 and stats.reruns == 0
    assert open(out, "rb").read() == (tmp_path / "rows.csv").read_bytes()


def test_pipeline_binary_output_and_cli(tmp_path, engine):
    from thermompnn_amd import native_pdb, ssm_scan
    paths = _pdb_set(tmp_path, n=9, seed=5)
    _, tables = _reference_shaped_rows(engine, paths, False, False, False)
    out = str(tmp_path / "scan.npz")
    n, stats = ssm_scan.scan_to_file(engine, paths, ["A"] * len(paths), out, centrality=True, chunk_files=4)
    z = np.load(out)
    assert n == z["ddg"].shape[0] == sum(t.shape[0] for t in tables) and list(z["offsets"]) == list(np.concatenate([[0], np.cumsum([t.shape[0] for t in tables])]))
    np.testing.assert_array_equal(z["ddg"], np.concatenate(tables))                 # ragged chunks == single forwards, bit for bit
    assert list(z["names"]) == [os.path.basename(p)[:-4] for p in paths]
    assert list(z["seqs"]) == [native_pdb.parse_pdb(p, "A")["seq"] for p in paths] and z["neighbors"].shape == (n,)
    # the CLI, CSV and binary, equals the library call
    csv_cli = ssm_scan.main(paths + ["--synthetic_weights", "0", "--out", str(tmp_path / "cli.csv"), "--chunk_files", "3"])
    ssm_scan.scan_to_file(engine, paths, ["A"] * len(paths), str(tmp_path / "lib.csv"))
    assert open(csv_cli, "rb").read() == (tmp_path / "lib.csv").read_bytes()
    npz_cli = ssm_scan.main(paths + ["--synthetic_weights", "0", "--out", str(tmp_path / "cli.npz")])
    np.testing.assert_array_equal(np.load(npz_cli)["ddg"], z["ddg"])
    # an unreadable file stops the scan with the parser's message (no hang, no partial silence)
    from thermompnn_amd._lib import TmpnnError
    with pytest.raises(TmpnnError, match="cannot open"):
        ssm_scan.scan_to_file(engine, paths[:5] + [str(tmp_path / "missing.pdb")] + paths[5:], ["A"] * (len(paths) + 1),
                              str(tmp_path / "bad.csv"), chunk_files=2)
    assert ssm_scan.scan_to_file(engine, [], [], str(tmp_path / "empty.csv"))[0] == 0


def test_pipeline_reruns_an_overflowing_chunk(tmp_path, synthetic_weights):
    """A chunk whose f16x2 forward leaves the fp16 range is rerun at the retry precision from its staging slot (the writer
    thread sees the chunk's status word): same tables as a bf16x3 engine, with a warning; no retry precision -> error."""
    from thermompnn_amd import ssm_scan
    from thermompnn_amd._lib import TmpnnRangeError
    from thermompnn_amd.engine import Engine
    W = {k: v.clone() for k, v in synthetic_weights.items()}
    W["prot_mpnn.features.edge_embedding.weight"] = W["prot_mpnn.features.edge_embedding.weight"] * 1e6
    paths = _pdb_set(tmp_path, n=6, seed=3)
    ch = ["A"] * len(paths)
    ssm_scan.scan_to_file(Engine(W, "cuda:0", 48, precision="bf16x3"), paths, ch, str(tmp_path / "want.npz"), chunk_files=3)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        _, stats = ssm_scan.scan_to_file(Engine(W, "cuda:0", 48, precision="f16x2"), paths, ch, str(tmp_path / "got.npz"), chunk_files=3)
    assert stats.reruns == stats.chunks >= 2 and any("bf16x3" in str(w.message) for w in rec)
    want = np.load(tmp_path / "want.npz")["ddg"]
    np.testing.assert_array_equal(np.load(tmp_path / "got.npz")["ddg"], want)
    with pytest.raises(TmpnnRangeError):
        ssm_scan.scan_to_file(Engine(W, "cuda:0", 48, precision="f16x2", retry_precision=None), paths, ch,
                              str(tmp_path / "strict.npz"), chunk_files=3)
    # the device-table form of the pipeline (what a multi-rank scan gathers over RCCL, round 5): the tables stay on the GPU, the sink
    # gets no host copy, and a rerun writes the chunk's rows of the SAME device buffer — on a side stream, so that the rerun's use of
    # the caller's stream (not the writer thread's default one) is exercised
    from thermompnn_amd import pipeline
    dev_table = torch.full((want.shape[0] + 7, 21), float("nan"), device="cuda:0")
    seen = []
    side = torch.cuda.Stream()
    with warnings.catch_warnings(record=True) as rec2, torch.cuda.stream(side):
        warnings.simplefilter("always")
        st = pipeline.scan_files(Engine(W, "cuda:0", 48, precision="f16x2"), paths, ch, lambda c: seen.append((c.table is None, c.row0, c.T)),
                                 chunk_files=3, device_table=dev_table)
        side.synchronize()
    assert st.reruns == st.chunks >= 2 and all(t for t, _, _ in seen) and [r for _, r, _ in seen] == list(np.cumsum([0] + [T for _, _, T in seen])[:-1])
    np.testing.assert_array_equal(dev_table[:want.shape[0]].cpu().numpy(), want)
    assert torch.isnan(dev_table[want.shape[0]:]).all()
    with pytest.raises(ValueError, match="device_table holds"):
        pipeline.scan_files(Engine(synthetic_weights, "cuda:0", 48), paths, ch, None, device_table=dev_table[:10])


def test_custom_inference_fast_path_equals_the_reference_shaped_path(tmp_path):
    """custom_inference: native parser + one forward + native writer == TransferModel.forward(pdb, mutations) with one
    Mutation / result dict per mutant + csv.writer (analysis/custom_inference.py:72-111), byte for byte."""
    from thermompnn_amd import custom_inference
    pdb = os.path.join(GOLDEN, "2OCJ.pdb")
    os.makedirs(tmp_path / "a")
    os.makedirs(tmp_path / "b")
    fast = custom_inference.main(["--pdb", pdb, "--chain", "A", "--synthetic_weights", "0", "--out_dir", str(tmp_path / "a")])
    slow = custom_inference.main(["--pdb", pdb, "--chain", "A", "--synthetic_weights", "0", "--out_dir", str(tmp_path / "b"),
                                  "--reference_shaped"])
    a, b = open(fast, "rb").read(), open(slow, "rb").read()
    assert a == b and a.startswith(b",Model,Dataset,ddG_pred,position,wildtype,mutation,pdb,chain\n0,ThermoMPNN,2OCJ,")
    assert a.count(b"\n") == 3881


def test_sharded_file_scan_two_ranks_one_device(tmp_path):
    """The many-PDB CLI under torchrun (2 ranks on cuda:0, gloo): every rank runs the pipeline on its LPT shard; CSV: every rank
    formats its shard and places its text in the one file (dist.scan_files_to_csv), binary: one gather to rank 0 — same bytes as
    the single-process streaming run, CSV with post-processing and the binary tables."""
    import subprocess
    import sys
    from test_gpu_parity import _torchrun
    from thermompnn_amd import ssm_scan
    repo = os.path.dirname(os.path.dirname(GOLDEN))
    paths = _pdb_set(tmp_path, n=10, seed=21)
    flags = ["--synthetic_weights", "0", "--centrality", "--pick_best", "--chunk_files", "3"]
    one = ssm_scan.main(paths + flags + ["--out", str(tmp_path / "one.csv")])
    r = _torchrun(["-m", "thermompnn_amd.ssm_scan"] + paths + flags + ["--out", str(tmp_path / "two.csv")],
                  {"TMPNN_ONE_DEVICE": "1", "PYTHONPATH": repo})
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    assert open(one, "rb").read() == (tmp_path / "two.csv").read_bytes()
    r = _torchrun(["-m", "thermompnn_amd.ssm_scan"] + paths + ["--synthetic_weights", "0", "--out", str(tmp_path / "two.npz")],
                  {"TMPNN_ONE_DEVICE": "1", "PYTHONPATH": repo})
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    one_npz = ssm_scan.main(paths + ["--synthetic_weights", "0", "--out", str(tmp_path / "one.npz")])
    a, b = np.load(one_npz), np.load(tmp_path / "two.npz")
    np.testing.assert_array_equal(a["ddg"], b["ddg"])
    assert list(a["names"]) == list(b["names"]) and list(a["offsets"]) == list(b["offsets"])
    # the RCCL form of the binary path (tables stay on the device between the forwards and the gather), forced under gloo, with centrality
    r = _torchrun(["-m", "thermompnn_amd.ssm_scan"] + paths + ["--synthetic_weights", "0", "--centrality", "--device_tables", "--chunk_files", "3",
                                                            "--out", str(tmp_path / "dev.npz")], {"TMPNN_ONE_DEVICE": "1", "PYTHONPATH": repo})
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    c = np.load(tmp_path / "dev.npz")
    one_c = np.load(ssm_scan.main(paths + ["--synthetic_weights", "0", "--centrality", "--out", str(tmp_path / "one_c.npz")]))
    np.testing.assert_array_equal(a["ddg"], c["ddg"])
    np.testing.assert_array_equal(one_c["neighbors"], c["neighbors"])
    # a corrupt file fails BOTH ranks promptly (no rank may be left waiting in a collective)
    bad = tmp_path / "bad.pdb"
    bad.write_text("ATOM      1  N   ALA A   1      xx.000   0.000   0.000\n")
    r = _torchrun(["-m", "thermompnn_amd.ssm_scan"] + paths[:3] + [str(bad)] + ["--synthetic_weights", "0", "--out", str(tmp_path / "x.csv")],
                  {"TMPNN_ONE_DEVICE": "1", "PYTHONPATH": repo}, timeout=300)
    assert r.returncode != 0 and "malformed" in (r.stdout + r.stderr)


def test_reference_style_driver_runs_unchanged_through_compat(tmp_path):
    """A driver written the way analysis/custom_inference.py is — the reference's own import lines, get_trained_model,
    alt_parse_PDB, get_ssm_mutations, Mutation objects, `model(mut_pdb, final_mutation_list)`, `out["ddG"].cpu().item()`
    (custom_inference.py:11-15,72-97) — with ONLY compat/ on PYTHONPATH, against the golden ddG table."""
    import subprocess
    import sys
    from conftest import load_golden
    from thermompnn_amd import weights
    sd = weights.synthetic_state_dict(0)
    os.makedirs(tmp_path / "vanilla_model_weights")
    weights.save_vanilla_checkpoint(tmp_path / "vanilla_model_weights" / "v_48_020.pt", weights.split_transfer_state_dict(sd)[0], 48)
    weights.save_lightning_checkpoint(tmp_path / "thermo.ckpt", sd)
    code = f"""
import torch
from datasets import Mutation
from train_thermompnn import TransferModelPL
from protein_mpnn_utils import tied_featurize, alt_parse_PDB
from thermompnn_benchmarking import get_trained_model
from SSM import get_ssm_mutations
import numpy as np
class AD(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__
cfg = AD(model=AD(hidden_dims=[64, 32], subtract_mut=True, num_final_layers=2, freeze_weights=True, load_pretrained=True,
                  lightattn=True, lr_schedule=True), platform=AD(thermompnn_dir={str(tmp_path)!r}))
model = get_trained_model(model_name={str(tmp_path / 'thermo.ckpt')!r}, config=cfg, override_custom=True)
model = model.eval().cuda()
mut_pdb = alt_parse_PDB({os.path.join(GOLDEN, '2OCJ.pdb')!r}, 'A')
final = []
for m in get_ssm_mutations(mut_pdb[0]):
    if m is None:
        final.append(None)
        continue
    m = m.strip()
    final.append(Mutation(position=int(m[1:-1]), wildtype=m[0], mutation=m[-1], ddG=None, pdb=mut_pdb[0]['name']))
with torch.no_grad():
    pred, _ = model(mut_pdb, final)
vals = [out["ddG"].cpu().item() for mut, out in zip(final, pred) if mut is not None]
np.save({str(tmp_path / 'ddg.npy')!r}, np.array(vals, dtype=np.float32))
"""
    repo = os.path.dirname(os.path.dirname(GOLDEN))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(tmp_path),
                       env=dict(os.environ, PYTHONPATH=os.path.join(repo, "compat")))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    got = np.load(tmp_path / "ddg.npy").reshape(194, 20)
    np.testing.assert_allclose(got, load_golden("2OCJ_A")["ddg"][:, :20], atol=1e-4, rtol=0)


@pytest.mark.parametrize("suffix", ["headA", "headB", "headC"])
def test_non_default_head_configurations_match_the_reference(tmp_path, suffix):
    """TransferModel configurations other than the released one — hidden_dims of other sizes / counts, num_final_layers 0..3,
    lightattn on or off (transfer_model.py:45-73,84-108) — run through the generic head kernels (tmpnn_ddg_head_generic) behind the
    unchanged `model(pdb, mutations)` API, against vectors the imported reference produced for those configurations."""
    from conftest import load_golden
    from thermompnn_amd import pdb_io, weights
    from thermompnn_amd.ssm import mutation_objects
    from thermompnn_amd.transfer_model import TransferModel
    g = load_golden("2OCJ_A_" + suffix)
    head = dict(hidden_dims=[int(x) for x in g["hidden_dims"]], num_final_layers=int(g["num_final_layers"]), lightattn=bool(g["lightattn"]))
    sd = weights.synthetic_state_dict(int(g["weight_seed"]), head=head)

    class AD(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    os.makedirs(tmp_path / "vanilla_model_weights")
    weights.save_vanilla_checkpoint(tmp_path / "vanilla_model_weights" / "v_48_020.pt", weights.split_transfer_state_dict(sd)[0], 48)
    cfg = AD(model=AD(subtract_mut=True, freeze_weights=True, load_pretrained=True, **head), platform=AD(thermompnn_dir=str(tmp_path)))
    model = TransferModel(cfg)
    assert model.generic_head and not model.load_state_dict(sd).missing_keys
    model = model.eval().cuda()
    pdb = pdb_io.alt_parse_PDB(os.path.join(GOLDEN, "2OCJ.pdb"), "A")
    muts = mutation_objects(pdb[0])
    with torch.no_grad():
        pred, _ = model(pdb, muts)
    got = torch.cat([p["ddG"] for p in pred]).cpu().numpy().reshape(194, 20)
    np.testing.assert_allclose(got, g["ddg"], atol=1e-4, rtol=0)
    # the head table z itself, and a stated wild type that differs from the structure (uses z, not the ddG table)
    eng = model.engine()
    f = pdb_io.tied_featurize([pdb[0]], "cuda:0", None)
    res = eng.ssm_forward(f[0][0], f[1][0], f[2][0], f[12][0], f[5][0], torch.tensor([0, 194], dtype=torch.int32), want_hidden=True, want_ddg=False)
    _, z = model._generic_tables(eng, res["hidden"], f[1][0])
    np.testing.assert_allclose(z.cpu().numpy(), g["z"], atol=1e-5, rtol=0)
    from thermompnn_amd.datasets import Mutation
    wt_other = "A" if pdb[0]["seq"][7] != "A" else "G"
    with torch.no_grad():
        p2, _ = model(pdb, [Mutation(7, wt_other, "W", None, "2OCJ")])
    want = 1.7 * (g["z"][7, "ACDEFGHIKLMNPQRSTVWY".index("W")] - g["z"][7, "ACDEFGHIKLMNPQRSTVWY".index(wt_other)])
    assert abs(p2[0]["ddG"].item() - want) <= 1e-4
    # the engine refuses layouts that are not a TransferModel head
    from thermompnn_amd._lib import TmpnnError
    with pytest.raises(TmpnnError):
        eng.ddg_head_generic([res["hidden"][2]], f[1][0], model.prot_mpnn.W_s.weight, [torch.zeros(21, 100)], [torch.zeros(21)],
                             model.ddg_out.weight, model.ddg_out.bias)


def test_pipeline_propagates_a_failing_consumer_and_stays_usable(tmp_path, engine):
    """A sink that raises (a full disk, a bad callback) stops the scan with ITS exception — no hang, no leaked staging slot —
    and the same engine runs the next scan normally (the pinned slots are kept on the engine between scans)."""
    from thermompnn_amd import pipeline, ssm_scan
    paths = _pdb_set(tmp_path, n=12, seed=9)
    seen = []

    def sink(ch):
        seen.append(ch.index)
        if ch.index == 1:
            raise OSError("disk full (test)")

    with pytest.raises(OSError, match="disk full"):
        pipeline.scan_files(engine, paths, ["A"] * len(paths), sink, chunk_files=3)
    assert seen[:2] == [0, 1] and len(engine._staging_pool) == 3
    n, stats = ssm_scan.scan_to_file(engine, paths, ["A"] * len(paths), str(tmp_path / "after.npz"), chunk_files=3)
    assert stats.chunks >= 4 and n == np.load(tmp_path / "after.npz")["ddg"].shape[0]


def test_native_host_example_matches_python_pipeline(tmp_path, engine, synthetic_weights):
    """examples/scan_native.cpp — a host that binds libtmpnn.so WITHOUT Python or torch (HIP runtime API for memory and one
    stream; weights from the flat file of weights.export_raw) — writes the same bytes as the Python pipeline: the C-ABI is
    the whole product, torch is plumbing. Also with files split over several chunks and in another precision."""
    import subprocess
    from thermompnn_amd import build, ssm_scan, weights
    exe = os.path.join(os.path.dirname(build.HERE), "examples", "scan_native")
    if not os.path.exists(exe):
        build.build_native_example()
    raw = str(tmp_path / "w.raw")
    weights.export_raw(synthetic_weights, raw)
    paths = _pdb_set(tmp_path, n=11, seed=3)
    ref = str(tmp_path / "py.csv")
    n, _ = ssm_scan.scan_to_file(engine, paths, ["A"] * len(paths), ref)
    for extra in ([], ["--chunk_files", "4", "--threads", "3"]):
        out = str(tmp_path / "native.csv")
        r = subprocess.run([exe, raw, out] + extra + paths, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.startswith(f"{n} rows, ")
        assert open(out, "rb").read() == open(ref, "rb").read()
    # another precision goes through the same entry points (tmpnn_weights_create_p): equal to the engine at that precision
    from thermompnn_amd.engine import Engine
    eng32 = Engine(synthetic_weights, "cuda:0", 48, precision="fp32")
    ssm_scan.scan_to_file(eng32, paths[:4], ["A"] * 4, ref)
    r = subprocess.run([exe, raw, str(tmp_path / "n32.csv"), "--precision", "fp32"] + paths[:4], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(str(tmp_path / "n32.csv"), "rb").read() == open(ref, "rb").read()
    # errors come back through tmpnn_last_error: a missing file, a weight file that is not one
    r = subprocess.run([exe, raw, str(tmp_path / "x.csv"), str(tmp_path / "missing.pdb")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "missing.pdb" in r.stderr
    r = subprocess.run([exe, paths[0], str(tmp_path / "x.csv"), paths[0]], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "TMPNNRAW" in r.stderr


@pytest.mark.parametrize("pick_best,include_cys,centrality", [(False, False, False), (True, False, True), (False, True, True)])
def test_scan_datasets_equals_the_reference_loop(tmp_path, pick_best, include_cys, centrality):
    """ssm_scan.scan_datasets = the models x datasets driver loop of analysis/SSM.py:96-176 (one forward per protein, native
    rows) against that loop written the reference's way — ``model(mut_pdb, mutations)`` per protein, one row dict per listed
    mutant, 'WT Seq' from ``dataset.wt_seqs`` (NOT the parsed sequence: the sample CSV's SEQ column is a different string),
    compute_centrality per protein — for ThermoMPNN and the ProteinMPNN baseline: byte-identical files."""
    from thermompnn_amd import custom_inference, ssm_scan
    from thermompnn_amd.datasets import ddgBenchDataset
    from thermompnn_amd.ssm import mutation_objects
    from thermompnn_amd.thermompnn_benchmarking import ProteinMPNNBaseline, compute_centrality
    ds = ddgBenchDataset(None, GOLDEN, os.path.join(GOLDEN, "ddgbench_sample.csv"))
    model = custom_inference.load_model(None, None, 0)
    models = {"ProteinMPNN": ProteinMPNNBaseline(model.cfg, version="v_48_020.pt").eval().cuda(), "ThermoMPNN": model}
    files = ssm_scan.scan_datasets(models, {"sample": ds}, pick_best=pick_best, include_cys=include_cys, centrality=centrality,
                                   out_dir=str(tmp_path))
    assert [os.path.basename(f) for f in files] == ["ProteinMPNN_sample_SSM_preds.csv", "ThermoMPNN_sample_SSM_preds.csv"]
    for (name, m), f in zip(models.items(), files):
        rows = []
        for mut_pdb, _ in ds:
            p = mut_pdb[0]
            muts = mutation_objects(p)
            with torch.no_grad():
                pred, _ = m(mut_pdb, muts)
            table = np.zeros((len(p["seq"]), 21), np.float32)
            live = [(mu, o) for mu, o in zip(muts, pred) if mu is not None]
            vals = torch.cat([o["ddG"].reshape(1) for _, o in live]).cpu().numpy()
            for (mu, _), v in zip(live, vals):
                table[mu.position, "ACDEFGHIKLMNPQRSTVWY".index(mu.mutation)] = v
            nb = None
            if centrality:
                ck = [c for c in p.keys() if "coords" in c][0]
                nb = compute_centrality(p[ck], basis_atom="CA", backup_atom="C", chain=ck[-1], radius=10.0).cpu().numpy()
            r = ssm_scan.rows_for_protein(p, table, nb, name, "sample", pick_best, include_cys)
            for x in r:
                x["WT Seq"] = ds.wt_seqs[p["name"]]
            rows += r
        ref = str(tmp_path / "ref.csv")
        ssm_scan.write_csv(rows, ref)
        assert len(rows) > 100 and open(f, "rb").read() == open(ref, "rb").read(), name
    # the compat module exposes the same function under the reference's module name
    import importlib.util
    repo = os.path.dirname(os.path.dirname(GOLDEN))
    spec = importlib.util.spec_from_file_location("_compat_SSM", os.path.join(repo, "compat", "SSM.py"))
    assert "scan_datasets" in open(spec.origin).read()


def test_keep_preds_loop_writes_the_reference_frame(tmp_path):
    """run_prediction_keep_preds / evaluate_datasets = analysis/thermompnn_benchmarking.py:122-187, 242-253: the raw-prediction
    CSV in the layout of the reference's frame (a listed mutation without a measurement keeps a row with only WT Seq / Model /
    Dataset; a residue number the structure lacks is dropped by the dataset; 'pdb' stripped; neighbours only with --centrality),
    values = the golden ddG tables, metrics = run_prediction_default's."""
    import csv
    from thermompnn_amd import custom_inference
    from thermompnn_amd.datasets import ddgBenchDataset
    from thermompnn_amd.thermompnn_benchmarking import compute_centrality, evaluate_datasets, run_prediction_default
    from conftest import load_golden
    TOL_DDG = 1e-4            # kcal/mol; BASELINE.json north_star
    ds = ddgBenchDataset(None, GOLDEN, os.path.join(GOLDEN, "ddgbench_sample.csv"))
    model = custom_inference.load_model(None, None, 0)
    res = evaluate_datasets({"ThermoMPNN": model}, {"sample": ds}, keep_preds=True, centrality=True, out_dir=str(tmp_path))
    want = run_prediction_default("ThermoMPNN", model, "sample", ds, [])
    assert res == want and res[0]["n"] == 6
    text = (tmp_path / "ThermoMPNN_sample_raw_preds.csv").read_text()
    lines = text.split("\n")
    assert lines[0] == ",WT Seq,Model,Dataset,ddG_true,ddG_pred,position,wildtype,mutation,neighbors,best_AA,pdb" and lines[-1] == ""
    rows = list(csv.DictReader(text.splitlines()))
    assert [r[""] for r in rows] == [str(i) for i in range(7)]                 # 8 listed, A999G has no residue in the structure
    g = {"2OCJ": load_golden("2OCJ_A")["ddg"], "2OCJ_gap_chainA": load_golden("2OCJ_A_gap")["ddg"]}
    seq_cell = ds.wt_seqs["2OCJ"]
    nb = {}
    for pdb, _ in ds:
        ck = [c for c in pdb[0].keys() if "coords" in c][0]
        nb[pdb[0]["name"]] = compute_centrality(pdb[0][ck], basis_atom="CA", backup_atom="C", chain=ck[-1], radius=10.0).cpu().tolist()
    n_scored = 0
    for r in rows:
        assert r["Model"] == "ThermoMPNN" and r["Dataset"] == "sample" and r["WT Seq"] == seq_cell and r["best_AA"] == ""
        if r["ddG_pred"] == "":                                                 # Q100E: listed without a measurement
            assert all(r[c] == "" for c in ("ddG_true", "position", "wildtype", "mutation", "neighbors", "pdb"))
            continue
        n_scored += 1
        pos, a = int(r["position"]), "ACDEFGHIKLMNPQRSTVWY".index(r["mutation"])
        assert abs(float(r["ddG_pred"]) - g[r["pdb"]][pos, a]) <= TOL_DDG and r["ddG_pred"] == repr(float(r["ddG_pred"]))
        assert int(r["neighbors"]) == nb[r["pdb"]][pos] and r["pdb"] in g
    assert n_scored == 6 and [r["ddG_true"] for r in rows if r["ddG_true"]][:3] == ["1.5", "-0.699999988079071", "-2.25"]
    met = (tmp_path / "ThermoMPNN_metrics.csv").read_text().split("\n")
    assert met[0] == ",Model,Dataset,ddG r2,ddG mse,ddG rmse,ddG spearman,ddG pearson" and met[1].startswith("0,ThermoMPNN,sample,")
    assert [float(x) for x in met[1].split(",")[3:]] == [res[0][f"ddG {k}"] for k in ("r2", "mse", "rmse", "spearman", "pearson")]
