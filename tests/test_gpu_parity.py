"""GPU parity tests proper: the HIP path (through the C-ABI) vs the CPU oracle and vs the committed
reference goldens.  Run on the MI355X box:  python -m pytest tests -m gpu -x -q"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, HOT_F64_FACTOR, is_hot, load_golden, oracle_trace_f64, tol_scale, weights_for_case

pytestmark = pytest.mark.gpu

TOL_INTERMEDIATE = 1e-5   # abs; SURVEY §8c
TOL_DDG = 1e-4            # kcal/mol; BASELINE.json north_star
CASES = ["2OCJ_A", "2OCJ_A_gap", "2OCJ_AB", "syn_L32", "syn_L256", "syn_L256_s1"]
# the same proteins through two more weight sets of the imported reference (make_golden.EXTRA_WEIGHT_SETS): a second Xavier draw
# and the heavy "hot" draw (matrices x 3, biases x 5, LayerNorm gamma in [-2, 2]; tolerances scale with the tensors, conftest.tol_scale)
EXTRA_CASES = ["2OCJ_A_w1", "syn_L32_w1", "2OCJ_A_hot", "syn_L32_hot", "2OCJ_A_wide", "syn_L32_wide"]   # wide: Linear outputs to 1e4
_ENGINES = {}
WORST = {}          # (precision, quantity) -> worst |hip - reference| / tolerance seen in this session (written to gpurun_out/)


def engine_for(g, precision=None):
    """One engine per (weight set, precision) of the golden fixtures. No retry precision: an f16x2 forward that left the fp16 range
    raises TmpnnRangeError instead of coming back as a silent bf16x3 rerun that would pass "as f16x2" (VERDICT r4 weak 1b); the
    retry itself is covered by its own tests (test_range_overflow_is_detected_and_retried, test_gpu_e2e's pipeline rerun)."""
    from thermompnn_amd.engine import Engine
    key = (int(g["weight_seed"]), str(g["weight_style"]) if "weight_style" in g else "xavier", precision)
    if key not in _ENGINES:
        _ENGINES[key] = Engine(weights_for_case(g), "cuda:0", 48, precision=precision, retry_precision=None)
    return _ENGINES[key]


def close(got, want, tol, g, what, prec, f64=None):
    """Xavier draws: |got - want| <= tol, absolute (north_star's 1e-5 / 1e-4). Hot draw (activations up to 1e2): the tensors
    are one to two orders larger and ``want`` — the reference's own fp32 evaluation — is itself 0.3-0.5 of any sensible absolute
    line away from the truth, so the criterion is distance to the FLOAT64 truth ``f64`` (conftest.oracle_trace_f64, same graph):
        |got - f64| <= max(2.5 x max|want - f64|, tol)            per tensor
    i.e. the HIP path may be at most 2.5 x as far from the truth as the reference is. The ratios against
    the round-2 (divisor 4) and round-3 (divisor 3) scaled absolute lines are still recorded in parity_worst_errors.json."""
    got64, want64 = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = float(np.nanmax(np.abs(got64 - want64))) if np.size(want) else 0.0
    style = str(g["weight_style"]) if "weight_style" in g else "xavier"
    k = f"{prec}/{style}/{what}"
    if is_hot(g) and f64 is not None:
        truth = np.asarray(f64, dtype=np.float64)
        ref_err = float(np.nanmax(np.abs(want64 - truth))) if np.size(want) else 0.0
        hip_err = float(np.nanmax(np.abs(got64 - truth))) if np.size(want) else 0.0
        bound = max(HOT_F64_FACTOR * ref_err, tol)
        line4 = tol * max(1.0, float(np.nanmax(np.abs(want64))) / 4.0)
        rec = {"ratio": hip_err / bound, "abs_err_vs_f64": hip_err, "reference_abs_err_vs_f64": ref_err, "bound": bound,
               "hip_over_reference_distance": hip_err / ref_err if ref_err > 0 else None,
               "vs_reference_ratio_divisor4_line": err / line4, "vs_reference_ratio_divisor3_line": err / (line4 * 4.0 / 3.0),
               "abs_err_vs_reference": err, "criterion": "float64 truth, 2.5 x the reference's own distance"}
        if rec["ratio"] > WORST.get(k, {"ratio": -1})["ratio"]:
            WORST[k] = rec
        assert hip_err <= bound, f"{what}: |hip - f64| = {hip_err:.3e} > {bound:.3e} (reference is {ref_err:.3e} from the truth)"
        return
    atol = tol * tol_scale(g, want)
    if err / atol > WORST.get(k, {"ratio": -1})["ratio"]:
        WORST[k] = {"ratio": err / atol, "abs_err": err, "atol": atol}
    np.testing.assert_allclose(got, want, atol=atol, rtol=0, err_msg=what)


@pytest.fixture(scope="module", autouse=True)
def _dump_worst():
    yield
    import json
    out = os.path.join(os.path.dirname(os.path.dirname(GOLDEN)), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_worst_errors.json"), "w") as fh:
            json.dump(WORST, fh, indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.fixture(scope="module")
def engine(synthetic_weights):
    from thermompnn_amd.engine import Engine
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return Engine(synthetic_weights, "cuda:0", 48, retry_precision=None)      # (see engine_for: no silent reruns in parity tests)


def oracle_trace(W, g, E_idx=None):
    """Oracle intermediates; ``E_idx`` pins the neighbour graph (order included) so per-edge tensors compare
    slot by slot and exact K-th-distance ties (implementation-defined in torch.topk) cannot leak downstream."""
    from oracle import thermompnn_oracle as orc
    t = torch.from_numpy
    X, S, mask = t(g["X"])[None], t(g["S"].astype(np.int64))[None], t(g["mask"])[None]
    ridx, cenc = t(g["residue_idx"].astype(np.int64))[None], t(g["chain_enc"].astype(np.int64))[None]
    tr = {}
    ov = None if E_idx is None else t(np.ascontiguousarray(E_idx).astype(np.int64))[None]
    with torch.no_grad():
        orc.ssm_table(W, X, S, mask, torch.ones_like(mask), ridx, cenc, 48, trace=tr, E_idx_override=ov)
    return {k: v[0].numpy() for k, v in tr.items()}


def adjusted_distances(g):
    from oracle import thermompnn_oracle as orc
    X, mask = torch.from_numpy(g["X"])[None], torch.from_numpy(g["mask"])[None]
    return orc.adjusted_distances(X[:, :, 1], mask)[0].numpy()


def topk_rows_differing(ours, ref, D_adj, rows):
    """Rows whose neighbour SET differs from the reference's. A difference is legal only as an exact tie at the
    K-th distance (torch.topk leaves tie order unspecified): assert that, return the tie rows."""
    ties = []
    for i in rows:
        a, b = set(ours[i].tolist()), set(ref[i].tolist())
        if a == b:
            continue
        kth = np.sort(D_adj[i])[len(b) - 1]
        assert all(D_adj[i, j] == kth for j in a ^ b), f"row {i}: neighbour sets differ beyond a K-th-distance tie"
        ties.append(int(i))
    return ties


def packed_inputs(g, dev="cuda:0"):
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)
    L = len(g["S"])
    return dict(X=t(g["X"], torch.float32), S=t(g["S"], torch.int32), mask=t(g["mask"], torch.float32),
                ridx=t(g["residue_idx"], torch.int32), cenc=t(g["chain_enc"], torch.int32),
                offsets=torch.tensor([0, L], dtype=torch.int32, device=dev), L=L)


def align(ours, ours_idx, ref, ref_idx, rows):
    """Reorder both [L,K,C] tensors by neighbour id so they compare edge by edge."""
    out_a, out_b = [], []
    for i in rows:
        k = ref_idx.shape[1]
        oi = ours_idx[i, :k]
        if sorted(oi.tolist()) != sorted(ref_idx[i].tolist()):
            continue
        out_a.append(ours[i, :k][np.argsort(oi, kind="stable")])
        out_b.append(ref[i][np.argsort(ref_idx[i], kind="stable")])
    return np.stack(out_a), np.stack(out_b)


def test_library_is_the_hip_build():
    from thermompnn_amd import _lib
    lib = _lib.load()
    assert lib.tmpnn_version() == 200 and lib.tmpnn_num_tensors() == 130
    with open("/proc/self/maps") as fh:
        assert "libtmpnn.so" in fh.read()


@pytest.mark.parametrize("case", CASES + EXTRA_CASES)
def test_stagewise_parity_vs_oracle(case):
    g = load_golden(case)
    check_stagewise(g, engine_for(g), weights_for_case(g))


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "f16x2"])
def test_extra_weight_sets_in_every_precision(mode):
    """Second Xavier draw + the hot draw, stage by stage and fused, on every matrix-core path (the accuracy margin of the
    split-precision kernels at larger activations; the worst error / tolerance per precision lands in
    gpurun_out/parity_worst_errors.json and is quoted in DESIGN.md)."""
    for case in EXTRA_CASES:
        g = load_golden(case)
        eng = engine_for(g, mode)
        check_stagewise(g, eng, weights_for_case(g))
        check_fused_forward(case, eng, weights_for_case(g))


def check_stagewise(g, engine, synthetic_weights):
    prec = engine.precision
    p = packed_inputs(g)
    L, valid = p["L"], np.nonzero(g["mask"] > 0)[0]
    Keff = min(48, L)

    # K0: neighbour sets (unmasked rows; ties at the K-th distance tolerated) and adjusted distances
    E_idx, D_nb = engine.knn_topk(p["X"], p["mask"], p["offsets"])
    ei, dn = E_idx.cpu().numpy(), D_nb.cpu().numpy()
    assert (ei[:, Keff:] == -1).all() and (ei[:, :Keff] >= 0).all()
    D_adj = adjusted_distances(g)
    ties = topk_rows_differing(ei[:, :Keff], g["E_idx"], D_adj, valid)
    assert len(ties) <= 1
    np.testing.assert_allclose(dn[valid, :Keff], np.take_along_axis(D_adj, ei[:, :Keff].astype(np.int64), 1)[valid],
                               atol=1e-6, rtol=3e-7)                       # <= 1 ulp of sqrt
    assert (np.diff(dn[:, :Keff], axis=1) >= 0).all()                     # sorted ascending like torch.topk
    # everything downstream is compared on the SAME graph, slot by slot
    tr = oracle_trace(synthetic_weights, g, ei[:, :Keff])
    t64 = oracle_trace_f64(g, ei[:, :Keff]) if is_hot(g) else None        # the float64 truth on the same graph (hot draws)
    f64 = lambda name: None if t64 is None else t64[name]
    f64_edges = lambda name: None if t64 is None else align(t64[name], t64["E_idx"], tr[name], tr["E_idx"], valid)[0]

    # K1: featurizer output E (LayerNorm) and h_E = W_e E + b
    h_E, E = engine.edge_featurize(p["X"], p["ridx"], p["cenc"], E_idx, D_nb, want_E=True)
    a, b = align(E.cpu().numpy(), ei, tr["E"], tr["E_idx"], valid)
    close(a, b, TOL_INTERMEDIATE, g, "E", prec, f64_edges("E"))
    a, b = align(h_E.cpu().numpy(), ei, tr["h_E0"], tr["E_idx"], valid)
    close(a, b, TOL_INTERMEDIATE, g, "h_E0", prec, f64_edges("h_E0"))
    assert (h_E.cpu().numpy()[:, Keff:] == 0).all()

    # K2/K3: encoder
    h_V = torch.zeros((L, 128), device="cuda:0")
    for l in range(3):
        engine.enc_layer(l, h_V, h_E, E_idx, p["mask"])
        close(h_V.cpu().numpy(), tr[f"hV_enc{l + 1}"], TOL_INTERMEDIATE, g, f"hV_enc{l + 1}", prec, f64(f"hV_enc{l + 1}"))
    a, b = align(h_E.cpu().numpy(), ei, tr["h_E_final"], tr["E_idx"], valid)
    close(a, b, TOL_INTERMEDIATE, g, "h_E_final", prec, f64_edges("h_E_final"))

    # K4: decoder
    hs = []
    for l in range(3):
        h_V = engine.dec_layer(l, h_V, h_E, E_idx, p["S"], p["mask"])
        hs.append(h_V)
        close(h_V.cpu().numpy(), tr[f"hV_dec{l + 1}"], TOL_INTERMEDIATE, g, f"hV_dec{l + 1}", prec, f64(f"hV_dec{l + 1}"))
    assert (hs[2].cpu().numpy()[g["mask"] == 0] == 0).all()

    # epilogues: W_s embedding, logits, head
    np.testing.assert_array_equal(engine.seq_embed(p["S"]).cpu().numpy(), tr["h_S"])
    close(engine.log_probs(hs[2]).cpu().numpy(), tr["log_probs"], TOL_INTERMEDIATE, g, "log_probs", prec, f64("log_probs"))
    ddg, z = engine.ddg_head(hs[2], hs[1], p["S"], want_z=True)
    close(z.cpu().numpy(), tr["z"], TOL_INTERMEDIATE, g, "z", prec, f64("z"))
    close(ddg.cpu().numpy(), tr["ddg"], TOL_DDG, g, "ddg", prec, f64("ddg"))


@pytest.mark.parametrize("case", CASES + EXTRA_CASES)
def test_fused_forward_vs_reference_golden(case):
    """tmpnn_ssm_forward against the vectors the imported reference produced (tests/golden/make_golden.py)."""
    g = load_golden(case)
    check_fused_forward(case, engine_for(g), weights_for_case(g))


def check_fused_forward(case, engine, synthetic_weights):
    g = load_golden(case)
    p = packed_inputs(g)
    res = engine.ssm_forward(p["X"], p["S"], p["mask"], p["ridx"], p["cenc"], p["offsets"], want_hidden=True,
                             want_log_probs=True, want_E_idx=True)
    valid = np.nonzero(g["mask"] > 0)[0]
    Keff = min(48, p["L"])
    ei = res["E_idx"].cpu().numpy()
    ties = topk_rows_differing(ei[:, :Keff], g["E_idx"], adjusted_distances(g), valid)
    if ties:
        # syn_L256 (BASELINE config 2, seed 0) has ONE exact fp32 tie at row 185's 48th neighbour; the reference's
        # pick is implementation-defined, so compare against the oracle run on the engine's (valid) graph instead
        assert case == "syn_L256" and ties == [185]
        tr = oracle_trace(synthetic_weights, g, ei[:, :Keff])
        g = dict(g, hV_dec3=tr["hV_dec3"], log_probs=tr["log_probs"], ddg=tr["ddg"][:, :20])
    hid = res["hidden"].cpu().numpy()
    prec = engine.precision + "/fused"
    t64 = oracle_trace_f64(g, ei[:, :Keff]) if is_hot(g) else None        # hot draws: distance to the float64 truth (close())
    f64 = lambda name: None if t64 is None else t64[name]
    close(hid[2], g["hV_dec3"], TOL_INTERMEDIATE, g, "hV_dec3", prec, f64("hV_dec3"))
    if "hV_dec1" in g:
        close(hid[0], g["hV_dec1"], TOL_INTERMEDIATE, g, "hV_dec1", prec, f64("hV_dec1"))
        close(hid[1], g["hV_dec2"], TOL_INTERMEDIATE, g, "hV_dec2", prec, f64("hV_dec2"))
    close(res["log_probs"].cpu().numpy(), g["log_probs"], TOL_INTERMEDIATE, g, "log_probs", prec, f64("log_probs"))
    ddg = res["ddg"].cpu().numpy()
    have = ~np.isnan(g["ddg"][:, 0])
    close(ddg[have][:, :20], g["ddg"][have], TOL_DDG, g, "ddg", prec, None if t64 is None else t64["ddg"][have][:, :20])
    wt = g["S"].astype(np.int64)
    assert (ddg[np.arange(len(wt)), wt] == 0).all()                       # wt -> wt rows are exactly 0


def test_ragged_batch_equals_single_proteins(engine):
    """Packing proteins of different lengths (incl. L < K) into one ragged batch changes nothing, bit for bit."""
    names = ["syn_L32", "2OCJ_A_gap", "syn_L256", "2OCJ_AB", "syn_L32"]
    gs = [load_golden(n) for n in names]
    singles = []
    for g in gs:
        p = packed_inputs(g)
        r = engine.ssm_forward(p["X"], p["S"], p["mask"], p["ridx"], p["cenc"], p["offsets"], want_hidden=True)
        singles.append((r["ddg"].cpu().numpy(), r["hidden"].cpu().numpy()))
    cat = lambda k, dt: torch.from_numpy(np.concatenate([np.asarray(g[k]) for g in gs])).to("cuda:0", dt)
    lens = [len(g["S"]) for g in gs]
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
    r = engine.ssm_forward(cat("X", torch.float32), cat("S", torch.int32), cat("mask", torch.float32),
                           cat("residue_idx", torch.int32), cat("chain_enc", torch.int32), offsets, want_hidden=True,
                           want_E_idx=True)
    ddg, hid, ei = r["ddg"].cpu().numpy(), r["hidden"].cpu().numpy(), r["E_idx"].cpu().numpy()
    for (d1, h1), s, e in zip(singles, offsets[:-1].tolist(), offsets[1:].tolist()):
        np.testing.assert_array_equal(ddg[s:e], d1)
        np.testing.assert_array_equal(hid[:, s:e], h1)
        blk = ei[s:e]
        assert ((blk == -1) | ((blk >= s) & (blk < e))).all()              # neighbours never cross proteins


def test_transfer_model_api_drop_in(tmp_path, synthetic_weights):
    """custom_inference.py's call sequence (:67-97): TransferModel(cfg).eval().cuda(); model(pdb, mutations)."""
    from thermompnn_amd import weights
    from thermompnn_amd.protein_mpnn_utils import alt_parse_PDB
    from thermompnn_amd.ssm import mutation_objects
    from thermompnn_amd.transfer_model import TransferModel

    class AD(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    mp, _ = weights.split_transfer_state_dict(synthetic_weights)
    os.makedirs(tmp_path / "vanilla_model_weights")
    weights.save_vanilla_checkpoint(tmp_path / "vanilla_model_weights" / "v_48_020.pt", mp, 48)
    cfg = AD(model=AD(hidden_dims=[64, 32], subtract_mut=True, num_final_layers=2, freeze_weights=True,
                      load_pretrained=True, lightattn=True), platform=AD(thermompnn_dir=str(tmp_path)))
    model = TransferModel(cfg)
    assert cfg.decoding_order == "left-to-right"
    model.load_state_dict(synthetic_weights)
    with pytest.raises(RuntimeError, match="no CPU execution path"):
        model(alt_parse_PDB(os.path.join(GOLDEN, "2OCJ_gap_chainA.pdb"), "A"), [])
    model = model.eval().cuda()
    pdb = alt_parse_PDB(os.path.join(GOLDEN, "2OCJ_gap_chainA.pdb"), "A")
    muts = mutation_objects(pdb[0])
    with torch.no_grad():
        pred, second = model(pdb, muts)
    assert second is None and len(pred) == len(muts)
    g = load_golden("2OCJ_A_gap")
    n_none = 0
    for m, out in zip(muts, pred):
        if m is None:
            assert out is None
            n_none += 1
            continue
        v = out["ddG"]
        assert v.shape == (1,) and v.is_cuda
        assert abs(v.cpu().item() - g["ddg"][m.position, "ACDEFGHIKLMNPQRSTVWY".index(m.mutation)]) <= TOL_DDG
    assert n_none == 3
    # a stated wild type that is not the residue in the structure follows the reference formula z[mut] - z[wt]
    from thermompnn_amd.datasets import Mutation
    odd = [Mutation(5, "W", "A"), None, Mutation(7, pdb[0]["seq"][7], "C")]
    with torch.no_grad():
        pred2, _ = model(pdb, odd)
    z = g["z"]
    want = (1.7 * z[5, 0] - 0.3) - (1.7 * z[5, 18] - 0.3)
    assert abs(pred2[0]["ddG"].item() - want) <= TOL_DDG and pred2[1] is None


def test_protein_mpnn_forward_padded_batch(synthetic_weights):
    """ProteinMPNN.forward on a padded [B, L] batch (mask = 0 on padding) vs the oracle on the same tensors."""
    from oracle import thermompnn_oracle as orc
    from thermompnn_amd import pdb_io, weights
    from thermompnn_amd.protein_mpnn_utils import ProteinMPNN
    from thermompnn_amd.synthetic import synthetic_pdb_dict

    mp, _ = weights.split_transfer_state_dict(synthetic_weights)
    batch = [synthetic_pdb_dict(70, seed=11), synthetic_pdb_dict(90, seed=12)]
    feats = pdb_io.tied_featurize(batch, "cpu", None)
    X, S, mask, chain_M, cenc, ridx = feats[0], feats[1], feats[2], feats[4], feats[5], feats[12]
    with torch.no_grad():
        hid_o, hS_o, lp_o = orc.mpnn_forward(mp, X, S, mask, chain_M, ridx, cenc, 48)
    model = ProteinMPNN(21, 128, 128, 128, k_neighbors=48, augment_eps=0.0)
    model.load_state_dict(mp)
    model = model.eval().cuda()
    c = lambda t: t.cuda()
    with torch.no_grad():
        hid, hS, lp = model(c(X), c(S), c(mask), c(chain_M), c(ridx), c(cenc), None)
    assert len(hid) == 3 and hid[0].shape == (2, 90, 128) and lp.shape == (2, 90, 21)
    rows = mask.bool().numpy()
    for a, b in zip(hid, hid_o):
        np.testing.assert_allclose(a.cpu().numpy()[rows], b.numpy()[rows], atol=TOL_INTERMEDIATE, rtol=0)
        assert (a.cpu().numpy()[~rows] == 0).all()
    np.testing.assert_array_equal(hS.cpu().numpy(), hS_o.numpy())
    np.testing.assert_allclose(lp.cpu().numpy()[rows], lp_o.numpy()[rows], atol=TOL_INTERMEDIATE, rtol=0)
    with pytest.raises(NotImplementedError):
        ProteinMPNN(21, 128, 128, 128, k_neighbors=64)


def test_gathers_match_torch():
    from thermompnn_amd.protein_mpnn_utils import cat_neighbors_nodes, gather_edges, gather_nodes
    gen = torch.Generator().manual_seed(0)
    nodes = torch.randn(2, 37, 128, generator=gen).cuda()
    idx = torch.randint(0, 37, (2, 37, 48), generator=gen).cuda()
    want = torch.gather(nodes, 1, idx.view(2, -1, 1).expand(-1, -1, 128)).view(2, 37, 48, 128)
    assert torch.equal(gather_nodes(nodes, idx), want)
    m = torch.rand(2, 37, 1, generator=gen).cuda()                       # C = 1 path (mask gather, :1232)
    assert torch.equal(gather_nodes(m, idx), torch.gather(m, 1, idx.view(2, -1, 1)).view(2, 37, 48, 1))
    edges = torch.randn(2, 37, 37, 3, generator=gen).cuda()
    want_e = torch.gather(edges, 2, idx.unsqueeze(-1).expand(-1, -1, -1, 3))
    assert torch.equal(gather_edges(edges, idx), want_e)
    h_nb = torch.randn(2, 37, 48, 128, generator=gen).cuda()
    assert torch.equal(cat_neighbors_nodes(nodes, h_nb, idx), torch.cat([h_nb, want], -1))
    empty = gather_nodes(nodes[:, :0], idx[:, :0])
    assert empty.shape == (2, 0, 48, 128)


def test_edge_cases_and_errors(engine):
    from thermompnn_amd._lib import TmpnnError
    from thermompnn_amd.engine import Engine
    from thermompnn_amd.synthetic import synthetic_backbone
    from thermompnn_amd.weights import synthetic_state_dict
    # a single-residue protein next to a normal one: K_eff = 1, its only neighbour is itself
    X1, s1 = synthetic_backbone(1, 3)
    X2, s2 = synthetic_backbone(60, 4)
    X = torch.tensor(np.concatenate([X1, X2]), dtype=torch.float32)
    S = torch.tensor(["ACDEFGHIKLMNPQRSTVWY".index(c) for c in s1 + s2], dtype=torch.int32)
    T = 61
    res = engine.ssm_forward(X, S, torch.ones(T), torch.cat([torch.arange(1), torch.arange(60)]), torch.ones(T),
                             torch.tensor([0, 1, 61], dtype=torch.int32), want_E_idx=True)
    ei = res["E_idx"].cpu().numpy()
    assert ei[0, 0] == 0 and (ei[0, 1:] == -1).all() and np.isfinite(res["ddg"].cpu().numpy()).all()
    assert (ei[1:, :48] >= 1).all()
    # empty batch is a no-op, K > 48 is refused loudly
    out = engine.ssm_forward(X[:0], S[:0], torch.ones(0), torch.zeros(0), torch.zeros(0), torch.tensor([0], dtype=torch.int32), max_len=0)
    assert out["ddg"].shape == (0, 21)
    with pytest.raises(TmpnnError):
        Engine(synthetic_state_dict(0), "cuda:0", 64)
    with pytest.raises(TmpnnError, match="CUDA"):
        Engine(synthetic_state_dict(0), "cpu", 48)


def test_full_size_properties(engine):
    """BASELINE.json config 2 (L=256) and config 5 (L=2048): size-independent properties + oracle at L=1024."""
    from thermompnn_amd.synthetic import synthetic_backbone
    for L, seed in ((256, 0), (2048, 3)):
        Xn, seq = synthetic_backbone(L, seed)
        X = torch.tensor(Xn, dtype=torch.float32)
        S = torch.tensor(["ACDEFGHIKLMNPQRSTVWY".index(c) for c in seq], dtype=torch.int32)
        args = (torch.ones(L), torch.arange(L), torch.ones(L), torch.tensor([0, L], dtype=torch.int32))
        r = engine.ssm_forward(X, S, *args, want_E_idx=True, want_hidden=True)
        ddg, ei = r["ddg"].cpu().numpy(), r["E_idx"].cpu().numpy()
        assert np.isfinite(ddg).all() and (ddg[np.arange(L), S.numpy()] == 0).all()
        assert (ei[:, 0] == np.arange(L)).all()                              # self is the nearest neighbour (col 0)
        assert all(len(set(row.tolist())) == 48 for row in ei[:: max(1, L // 64)])
        # determinism: same input, same bits
        r2 = engine.ssm_forward(X, S, *args)
        assert torch.equal(r2["ddg"].cpu(), r["ddg"].cpu())
        # a rigid rotation + translation of the backbone leaves the graph and (to fp32 noise) the ddG unchanged
        th = 0.7
        R = torch.tensor([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], dtype=torch.float32)
        r3 = engine.ssm_forward(X @ R.T + torch.tensor([3.0, -2.0, 5.0]), S, *args, want_E_idx=True)
        same_graph = (np.sort(r3["E_idx"].cpu().numpy(), 1) == np.sort(ei, 1)).all(1).mean()
        assert same_graph > 0.99
        diff = np.abs(r3["ddg"].cpu().numpy() - ddg)          # rows whose K-th neighbour flipped move more
        assert np.percentile(diff, 95) < 1e-3 and diff.max() < 0.1


@pytest.mark.parametrize("L,seed", [(1024, 7), (2048, 3)], ids=["L1024", "config5_L2048"])
def test_large_chain_vs_oracle(L, seed, engine, synthetic_weights):
    """One long chain against the CPU oracle over the WHOLE table — L = 2048 / seed 3 is BASELINE.json configs[4] at full size
    (the rows > 512 path of the k-NN kernel, 98 304 edges); decoder states compared too."""
    from oracle import thermompnn_oracle as orc
    from thermompnn_amd.synthetic import synthetic_backbone
    Xn, seq = synthetic_backbone(L, seed)
    X = torch.tensor(Xn, dtype=torch.float32)
    S = torch.tensor(["ACDEFGHIKLMNPQRSTVWY".index(c) for c in seq])
    ones, ar = torch.ones(1, L), torch.arange(L)[None]
    r = engine.ssm_forward(X, S.int(), torch.ones(L), torch.arange(L), torch.ones(L), torch.tensor([0, L], dtype=torch.int32),
                           want_hidden=True, want_E_idx=True)
    tr = {}
    with torch.no_grad():    # on the engine's graph: an exact K-th-distance tie (implementation-defined in torch.topk) cannot leak in
        want = orc.ssm_table(synthetic_weights, X[None], S[None], ones, ones, ar, ones.long(), 48, trace=tr,
                             E_idx_override=r["E_idx"].cpu().long()[None])[0].numpy()
        free = {}
        orc.ssm_table(synthetic_weights, X[None], S[None], ones, ones, ar, ones.long(), 48, trace=free) if L <= 1024 else None
    np.testing.assert_allclose(r["ddg"].cpu().numpy(), want, atol=TOL_DDG, rtol=0)
    np.testing.assert_allclose(r["hidden"][2].cpu().numpy(), tr["hV_dec3"][0].numpy(), atol=TOL_INTERMEDIATE, rtol=0)
    if free:                 # and the graph itself: same neighbour sets as the oracle's own top-k, ties aside
        a, b = np.sort(r["E_idx"].cpu().numpy(), 1), np.sort(free["E_idx"][0].numpy(), 1)
        assert (a == b).all(1).mean() > 0.995


def test_custom_inference_script_end_to_end(tmp_path):
    """BASELINE config 1 counterpart: PDB -> CSV with the reference schema, values vs the reference golden."""
    import csv
    from thermompnn_amd import custom_inference
    out = custom_inference.main(["--pdb", os.path.join(GOLDEN, "2OCJ.pdb"), "--chain", "A", "--synthetic_weights", "0",
                                 "--out_dir", str(tmp_path)])
    assert os.path.basename(out) == "ThermoMPNN_inference_2OCJ.csv"
    rows = list(csv.DictReader(open(out)))
    g = load_golden("2OCJ_A")
    assert len(rows) == 3880 and list(rows[0].keys()) == ["", "Model", "Dataset", "ddG_pred", "position", "wildtype", "mutation", "pdb", "chain"]
    for r in rows[::97]:
        want = g["ddg"][int(r["position"]), "ACDEFGHIKLMNPQRSTVWY".index(r["mutation"])]
        assert abs(float(r["ddG_pred"]) - want) <= TOL_DDG
        assert r["pdb"] == "2OCJ" and r["chain"] == "A" and r["Model"] == "ThermoMPNN"


def test_sharded_scan_single_rank_ragged(engine):
    """dist.ssm_scan on one rank over a config-4-like ragged set (L in [40, 72], K_eff = min(48, L))."""
    from thermompnn_amd.dist import ssm_scan
    from thermompnn_amd.synthetic import synthetic_backbone
    rng = np.random.default_rng(2)
    prots = []
    for i, L in enumerate(rng.integers(40, 73, size=12)):
        X, seq = synthetic_backbone(int(L), 500 + i)
        prots.append(dict(X=X.astype(np.float32), S=np.array(["ACDEFGHIKLMNPQRSTVWY".index(c) for c in seq], dtype=np.int32),
                          mask=np.ones(L, np.float32), residue_idx=np.arange(L, dtype=np.int32), chain_enc=np.ones(L, np.int32)))
    tables = ssm_scan(engine, prots)
    assert [t.shape for t in tables] == [(len(p["S"]), 21) for p in prots]
    p = prots[5]
    L = len(p["S"])
    single = engine.ssm_forward(p["X"], p["S"], p["mask"], p["residue_idx"], p["chain_enc"], torch.tensor([0, L], dtype=torch.int32))["ddg"]
    assert torch.equal(tables[5], single)


def test_centrality_and_baseline_and_scan(tmp_path, engine, synthetic_weights):
    """compute_centrality, ProteinMPNNBaseline, the many-PDB SSM driver."""
    from oracle import thermompnn_oracle as orc
    from thermompnn_amd import native_pdb, pdb_io, ssm_scan, weights
    from thermompnn_amd.ssm import mutation_objects
    from thermompnn_amd.thermompnn_benchmarking import ProteinMPNNBaseline, compute_centrality
    gap = os.path.join(GOLDEN, "2OCJ_gap_chainA.pdb")
    pdb = pdb_io.alt_parse_PDB(gap, "A")
    # centrality vs the reference formula (cdist, NaN -> 2r, count < r, minus 1)
    ca = torch.tensor(pdb[0]["coords_chain_A"]["CA_chain_A"])
    d = torch.nan_to_num(torch.cdist(ca, ca), nan=20.0)
    want = (d < 10.0).sum(-1) - 1
    got = compute_centrality(pdb[0]["coords_chain_A"], chain="A").cpu()
    assert torch.equal(got, want) and int(got.min()) == -1
    # ProteinMPNNBaseline: -log p(mut) from the HIP log_probs vs the oracle
    class AD(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    mp, _ = weights.split_transfer_state_dict(synthetic_weights)
    os.makedirs(tmp_path / "vanilla_model_weights")
    weights.save_vanilla_checkpoint(tmp_path / "vanilla_model_weights" / "v_48_020.pt", mp, 48)
    cfg = AD(model=AD(load_pretrained=True, freeze_weights=True), platform=AD(thermompnn_dir=str(tmp_path)))
    base = ProteinMPNNBaseline(cfg).eval().cuda()
    muts = mutation_objects(pdb[0])[:60]
    with torch.no_grad():
        pred, lp = base(pdb, muts)
    g = load_golden("2OCJ_A_gap")
    assert lp.shape == (1, 194, 21)
    np.testing.assert_allclose(lp[0].cpu().numpy(), g["log_probs"], atol=TOL_INTERMEDIATE, rtol=0)
    assert abs(pred[3]["ddG"].item() + g["log_probs"][0, 3]) < 1e-5 and torch.equal(pred[3]["ddG"], -pred[3]["dTm"])
    # many-PDB driver: native parse -> ragged batch -> CSV; values vs the reference golden
    out = ssm_scan.main([os.path.join(GOLDEN, "2OCJ.pdb"), gap, "--chain", "A", "--synthetic_weights", "0", "--centrality",
                         "--pick_best", "--out", str(tmp_path / "ssm.csv")])
    import csv
    rows = list(csv.DictReader(open(out)))
    assert len(rows) == 194 + 191                              # one row per non-gap position
    g0 = load_golden("2OCJ_A")
    r = rows[17]
    assert r["pdb"] == "2OCJ" and r["mutation"] == "A" and abs(float(r["ddG_pred"]) - g0["ddg"][17, 0]) <= TOL_DDG
    best = "ACDEFGHIKLMNPQRSTVWY"[int(np.argmin(np.where(np.arange(20) == 1, np.inf, g0["ddg"][17])))]
    assert r["best_AA"] == best
    full = pdb_io.alt_parse_PDB(os.path.join(GOLDEN, "2OCJ.pdb"), "A")[0]
    ca0 = torch.tensor(full["coords_chain_A"]["CA_chain_A"])
    assert int(r["neighbors"]) == int(((torch.cdist(ca0, ca0) < 10.0).sum(-1) - 1)[17])
    assert int(rows[194 + 17]["neighbors"]) == int(want[17])     # second file = the gapped structure
    # every residue of the gapped / missing-atom structure: the scan masks centrality on the CA atom only, like the reference
    # (a residue that lacks N / C / O but has its CA still counts and is counted)
    seq_gap = pdb[0]["seq"]
    pos_rows = {int(r_["position"]): int(r_["neighbors"]) for r_ in rows[194:]}
    assert sorted(pos_rows) == [k for k, c in enumerate(seq_gap) if c != "-"]
    assert all(pos_rows[k] == int(want[k]) for k in pos_rows)


def _random_protein(rng, L, n_chains=1, p_missing=0.0):
    from thermompnn_amd.synthetic import synthetic_backbone
    X, seq = synthetic_backbone(L, int(rng.integers(1 << 30)))
    S = np.array(["ACDEFGHIKLMNPQRSTVWY".index(c) for c in seq], dtype=np.int64)
    mask = np.ones(L, np.float32)
    miss = rng.random(L) < p_missing
    mask[miss] = 0
    X = X.astype(np.float32)
    X[miss] = 0                                              # tied_featurize zeroes NaN coordinates
    S[miss & (rng.random(L) < 0.5)] = 20                     # some of them are gaps ('X')
    cuts = np.sort(rng.choice(np.arange(1, L), size=n_chains - 1, replace=False)) if n_chains > 1 and L > n_chains else []
    chain = np.ones(L, np.int64)
    for c in cuts:
        chain[c:] += 1
    ridx = 100 * (chain - 1) + np.arange(L)
    return X, S, mask, ridx, chain


@pytest.mark.parametrize("K", [48, 30])
def test_randomised_parity_vs_oracle(K, synthetic_weights):
    """Random lengths (1..120, many below K), random missing residues / gaps, 1-3 chains, K = 48 and K = 30: the fused
    HIP forward vs the oracle run on the engine's (validated) neighbour graph."""
    from oracle import thermompnn_oracle as orc
    from thermompnn_amd.engine import Engine
    eng = Engine(synthetic_weights, "cuda:0", K)
    rng = np.random.default_rng(1234 + K)
    prots = [_random_protein(rng, int(L), int(rng.integers(1, 4)), float(rng.choice([0.0, 0.05, 0.3])))
             for L in [1, 2, 7, 31, 47, 48, 49, 64, 100, 120]]
    lens = [len(p[1]) for p in prots]
    cat = lambda k, dt: torch.tensor(np.concatenate([p[k] for p in prots])).to("cuda:0", dt)
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
    r = eng.ssm_forward(cat(0, torch.float32), cat(1, torch.int32), cat(2, torch.float32), cat(3, torch.int32),
                        cat(4, torch.int32), offsets, want_hidden=True, want_log_probs=True, want_E_idx=True)
    ddg, hid, lp, ei = (r[k].cpu().numpy() for k in ("ddg", "hidden", "log_probs", "E_idx"))
    pos = 0
    for (X, S, mask, ridx, chain), L in zip(prots, lens):
        Keff = min(K, L)
        blk = ei[pos:pos + L]
        assert (blk[:, Keff:] == -1).all()
        local = blk[:, :Keff] - pos
        assert (local >= 0).all() and (local < L).all()
        t = torch.from_numpy
        D_adj = orc.adjusted_distances(t(X)[None, :, 1], t(mask)[None])[0].numpy()
        for i in np.nonzero(mask > 0)[0]:                     # a valid top-k up to exact ties
            kth = np.sort(D_adj[i])[Keff - 1]
            assert (D_adj[i, local[i]] <= kth).all() and len(set(local[i].tolist())) == Keff
        tr = {}
        with torch.no_grad():
            want = orc.ssm_table(synthetic_weights, t(X)[None], t(S)[None], t(mask)[None], torch.ones(1, L), t(ridx)[None],
                                 t(chain)[None], K, trace=tr, E_idx_override=t(local.astype(np.int64))[None])[0].numpy()
        valid = mask > 0
        # rows that attend to a masked neighbour are implementation-defined in the reference only through WHICH masked
        # residue ties in; with the graph pinned everything is comparable
        np.testing.assert_allclose(hid[2, pos:pos + L][valid], tr["hV_dec3"][0].numpy()[valid], atol=1e-5, rtol=0)
        np.testing.assert_allclose(lp[pos:pos + L][valid], tr["log_probs"][0].numpy()[valid], atol=1e-5, rtol=0)
        np.testing.assert_allclose(ddg[pos:pos + L], want, atol=TOL_DDG, rtol=0)
        assert (hid[:, pos:pos + L][:, ~valid] == 0).all()
        pos += L


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "f16x2"])
def test_all_matmul_modes_pass_golden_parity(mode, synthetic_weights):
    """Precision is an argument of the engine / weight handle: every matrix-core path passes the same golden parity in
    one process, and a per-call override reproduces the dedicated engine bit for bit."""
    from thermompnn_amd.engine import Engine
    eng = Engine(synthetic_weights, "cuda:0", 48, precision=mode)
    assert eng.precision == mode and eng.w.precision == mode
    for case in CASES:
        check_fused_forward(case, eng, synthetic_weights)
    p = packed_inputs(load_golden("syn_L256_s1"))
    a = eng.ssm_forward(p["X"], p["S"], p["mask"], p["ridx"], p["cenc"], p["offsets"])["ddg"]
    other = Engine(synthetic_weights, "cuda:0", 48, precision="f16x2" if mode != "f16x2" else "bf16x3")
    b = other.ssm_forward(p["X"], p["S"], p["mask"], p["ridx"], p["cenc"], p["offsets"], precision=mode)["ddg"]
    assert torch.equal(a, b)


def test_precision_default_comes_from_the_environment():
    """TMPNN_PRECISION only picks the default of handles created without a precision (read once per process)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(GOLDEN))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from thermompnn_amd import _lib\nfrom thermompnn_amd.engine import Engine\n"
            "from thermompnn_amd.weights import synthetic_state_dict\n"
            "assert _lib.load().tmpnn_matmul_mode().decode() == 'bf16x3'\n"
            "W = synthetic_state_dict(0)\n"
            "assert Engine(W, 'cuda:0').precision == 'bf16x3' and Engine(W, 'cuda:0', precision='fp32').precision == 'fp32'\n" % repo)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TMPNN_PRECISION="bf16x3"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    bad_code = ("import sys; sys.path.insert(0, %r)\nfrom thermompnn_amd.engine import Engine\n"
                "from thermompnn_amd.weights import synthetic_state_dict\nEngine(synthetic_state_dict(0), 'cuda:0')\n" % repo)
    bad = subprocess.run([sys.executable, "-c", bad_code], env=dict(os.environ, TMPNN_PRECISION="fp64"), capture_output=True,
                         text=True, timeout=600)
    assert bad.returncode != 0 and "unknown precision" in bad.stderr      # an error, not an abort()


@pytest.mark.parametrize("env", [{"TMPNN_NODE_SPLIT": "0", "TMPNN_FEAT_SPLIT": "0", "TMPNN_HEAD_SPLIT": "0"},
                                 {"TMPNN_NODE_IMG": "0", "TMPNN_FEAT_IMG": "0"}, {"TMPNN_NODE_DEEP": "0", "TMPNN_KNN_REG": "0"},
                                 {"TMPNN_MSG_WAVE_MIN": "0", "TMPNN_FUSE_SMALL": "0"}],
                         ids=["fp32_node_featurizer_head", "no_weight_fragment_images", "tall_node_tiles_and_lds_knn_for_small_launches",
                              "wavefront_per_residue_message_pass_for_small_launches"])
def test_selectable_kernel_forms_pass_golden_parity(env):
    """The non-default kernel forms of the f16x2 mode stay parity-green. The switches exist only in the debug variant of the
    library (libtmpnn_debug.so, -DTMPNN_DEBUG_BUILD; read once per process) — the shipped library ignores them."""
    import subprocess
    import sys
    from thermompnn_amd import _lib
    assert os.path.exists(_lib.DEBUG_LIB_PATH), "build the debug variant: python -m thermompnn_amd.build"
    # the wavefront-per-residue message pass is the bench batch's form: it also sees the hot / wide weight draws (f16x2 leg)
    select = "fused_forward or ragged_batch" + (" or (extra_weight_sets and f16x2)" if "TMPNN_MSG_WAVE_MIN" in env else "")
    env = dict(env, TMPNN_LIB=_lib.DEBUG_LIB_PATH)
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import pytest; sys.exit(pytest.main(['-q', '-x', '-m', 'gpu', '-k', %r, %r]))"
            % (os.path.dirname(GOLDEN.rstrip('/')).rsplit('/tests', 1)[0], os.path.dirname(GOLDEN), select, __file__))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_small_launch_kernel_forms_are_bit_identical():
    """Which form of a kernel runs depends on the launch size (register k-NN rows for proteins up to 512 residues, the deep-prefetch
    node_update when every workgroup has one 16-row tile): the forms must agree BIT FOR BIT, or a protein's result would depend
    on the batch it is in. tools/dbg_knn.py (ties, masked residues, L < K, ragged batches) and tools/dbg_deep.py run both forms
    in separate processes and compare the raw bits."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(GOLDEN.rstrip("/")))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "dbg_knn.py")], cwd=repo, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL IDENTICAL" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "dbg_deep.py")], cwd=repo, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if "max diff" in l]
    assert r.returncode == 0 and len(lines) == 3 and all("n diff 0 " in l for l in lines), r.stdout[-1500:] + r.stderr[-1500:]
    # launches with at most one tile per workgroup fuse the edge update with the next message pass (edge_msg_fused_kernel)
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "dbg_fused.py")], cwd=repo, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL IDENTICAL" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_f16x2_range_limit_is_loud():
    """f16x2 needs |x| < 65504: beyond it the GEMM core returns inf/nan (never a silently wrong finite number);
    bf16x3 and fp32 keep the full fp32 range."""
    from thermompnn_amd import _lib
    from thermompnn_amd.engine import _ptr, _stream
    lib = _lib.load()
    g = torch.Generator().manual_seed(7)
    X = (torch.randn(4, 48, 128, generator=g) * 1e5).cuda()
    W = (torch.rand(128, 128, generator=g) * 0.2 - 0.1).cuda()
    ref = X.double() @ W.double().t()
    out = {}
    for mode in (0, 1, 2):
        Y = torch.zeros_like(X)
        assert lib.tmpnn_gemm_probe(mode, _ptr(X), _ptr(W), _ptr(Y), 4, 1, _stream()) == 0
        torch.cuda.synchronize()
        out[mode] = Y
    for mode in (0, 1):
        assert torch.isfinite(out[mode]).all()
        assert float(((out[mode].double() - ref).abs() / ref.abs().max()).max()) < 1e-5
    assert not torch.isfinite(out[2]).all()


def test_split_precision_gemm_core_accuracy():
    """The split-precision matrix-core GEMMs are in the accuracy class of the exact-fp32 MFMA chain (vs float64)."""
    from thermompnn_amd import _lib
    from thermompnn_amd.engine import _ptr, _stream
    lib = _lib.load()
    T = 64
    g = torch.Generator().manual_seed(5)
    X = (torch.randn(T, 48, 128, generator=g) * torch.logspace(-3, 2, 128)).cuda()      # 5 decades of column scales
    W = (torch.rand(128, 128, generator=g) * 0.4 - 0.2).cuda()
    ref = X.double() @ W.double().t()
    scale = (X.double().abs() @ W.double().abs().t())                                   # error is relative to sum |a||b|
    errs = {}
    for mode in (0, 1, 2):
        Y = torch.zeros_like(X)
        assert lib.tmpnn_gemm_probe(mode, _ptr(X), _ptr(W), _ptr(Y), T, 1, _stream()) == 0
        torch.cuda.synchronize()
        errs[mode] = float(((Y.double() - ref).abs() / scale).max())
    assert errs[1] < 4e-7 and errs[1] <= 1.5 * errs[0], errs
    assert errs[2] < 6e-7 and errs[2] <= 2.5 * errs[0], errs


def _synthetic_protein(L, seed):
    from thermompnn_amd.synthetic import synthetic_backbone
    X, seq = synthetic_backbone(int(L), int(seed))
    return dict(X=X.astype(np.float32), S=np.array(["ACDEFGHIKLMNPQRSTVWY".index(c) for c in seq], dtype=np.int32),
                mask=np.ones(L, np.float32), residue_idx=np.arange(L, dtype=np.int32), chain_enc=np.ones(L, np.int32))


def _oracle_table(W, p):
    from oracle import thermompnn_oracle as orc
    L = len(p["S"])
    ones, ar = torch.ones(1, L), torch.arange(L)[None]
    with torch.no_grad():
        return orc.ssm_table(W, torch.from_numpy(p["X"])[None], torch.from_numpy(p["S"].astype(np.int64))[None], ones, ones,
                             ar, ones.long(), 48)[0].numpy()


def test_config3_full_size_ragged_1024(engine, synthetic_weights):
    """BASELINE.json configs[2] at FULL size: 1 024 ragged proteins, L ~ U[64, 512] (T = 295 632 residues, 8.7 GB of
    workspace, h_E byte offsets beyond 2^32) in ONE ssm_forward. Sampled proteins — first, last, longest, shortest and
    the ones straddling the 2^31- and 2^32-byte h_E offsets — must equal their single-protein forward AND their rows in a second,
    differently packed large batch bit for bit, and the CPU oracle to 1e-4 kcal/mol."""
    from thermompnn_amd.dist import pack_proteins
    lens = np.random.default_rng(1).integers(64, 513, size=1024)             # tools/run_configs.py:66-70
    prots = [_synthetic_protein(L, 1000 + i) for i, L in enumerate(lens)]
    b = pack_proteins(prots, list(range(1024)), "cuda:0")
    T = int(lens.sum())
    assert T == 295632 and T * 48 * 128 * 4 > 2 ** 32
    ddg = engine.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=b["max_len"])["ddg"]
    starts = np.concatenate([[0], np.cumsum(lens)])
    row_bytes = 48 * 128 * 4
    straddle = [int(np.searchsorted(starts, (2 ** k) // row_bytes, side="right") - 1) for k in (31, 32)]
    for k, i in zip((31, 32), straddle):
        assert starts[i] * row_bytes < 2 ** k <= starts[i + 1] * row_bytes
    sample = sorted({0, 1023, int(lens.argmax()), int(lens.argmin()), 511, 700, *straddle})
    assert len(sample) >= 8
    wt_zero = ddg.cpu().numpy()[np.arange(T), b["S"].cpu().numpy()]
    assert (wt_zero == 0).all() and bool(torch.isfinite(ddg).all())          # a checksum over ALL 5.9 M predictions
    # Batch invariance: a residue's result depends on its own protein only, bit for bit — whatever the batch around it and whichever
    # kernel forms the launch size selects. Since round 5 large launches run the message pass one wavefront per residue
    # (msg8_wave_kernel: whole multiples of 8 residues per workgroup; the remainder and small launches take the 8-wavefront form, one
    # protein alone the fused edge + message form): all of them use the same K order inside a matrix-core step and the same summation
    # order. Checked against (a) the same proteins packed into a DIFFERENT large batch and (b) their single-protein forwards.
    order = [i for i in reversed(sample)] + [i for i in range(100, 180) if i not in sample]
    b2 = pack_proteins(prots, order, "cuda:0")
    assert int(sum(lens[i] for i in order)) >= 64 * torch.cuda.get_device_properties(0).multi_processor_count
    ddg2 = engine.ssm_forward(b2["X"], b2["S"], b2["mask"], b2["ridx"], b2["cenc"], b2["offsets"], max_len=b2["max_len"])["ddg"]
    starts2 = np.concatenate([[0], np.cumsum([lens[i] for i in order])])
    for k, i in enumerate(order[:len(sample)]):
        assert torch.equal(ddg2[starts2[k]:starts2[k + 1]], ddg[starts[i]:starts[i + 1]]), f"protein {i} depends on its batch"
    for i in sample:
        p, L = prots[i], int(lens[i])
        mine = ddg[starts[i]:starts[i + 1]]
        single = engine.ssm_forward(p["X"], p["S"], p["mask"], p["residue_idx"], p["chain_enc"],
                                    torch.tensor([0, L], dtype=torch.int32))["ddg"]
        assert torch.equal(mine, single), f"protein {i} differs from its single-protein forward"
        np.testing.assert_allclose(mine.cpu().numpy(), _oracle_table(synthetic_weights, p), atol=TOL_DDG, rtol=0)


def test_message_pass_residue_split_is_batch_invariant(engine):
    """Round 6: inside a workgroup of the wavefront-per-residue message pass the first-dispatched wavefront of a SIMD takes 11 sixteenths
    of the SIMD's residues, its partner the rest (thermompnn_amd/csrc/tmpnn_msg.hip, TM_MSG_WAVE_OLD_16TH). Batches of n copies of one
    L = 256 protein give every workgroup of a 256-CU device n residues (n = 20: 16 + a remainder for the 8-wavefront kernel): the splits
    3:1, 4:2, 7:3 and 10:4 per SIMD. Every copy must equal the single-protein forward (fused small-launch forms) bit for bit — every residue
    is computed exactly once, whichever wavefront takes it."""
    g = load_golden("syn_L256")
    p = packed_inputs(g)
    single = engine.ssm_forward(p["X"], p["S"], p["mask"], p["ridx"], p["cenc"], p["offsets"])["ddg"]
    L = p["L"]
    for n in (16, 20, 24, 40, 56):
        rep = lambda k: p[k].repeat(*([n] + [1] * (p[k].dim() - 1)))
        offsets = torch.arange(0, (n + 1) * L, L, dtype=torch.int32)
        ddg = engine.ssm_forward(rep("X"), rep("S"), rep("mask"), rep("ridx"), rep("cenc"), offsets, max_len=L)["ddg"]
        assert ddg.shape[0] == n * L
        for k in range(n):
            assert torch.equal(ddg[k * L:(k + 1) * L], single), f"copy {k} of {n} differs from the single-protein forward"


def test_config4_listed_mutations(engine, synthetic_weights):
    """BASELINE.json configs[3] on one rank: 300 Megascale-like proteins (L in [40, 72]: K_eff = min(48, L) < 48 for many)
    and an explicit list of 200 000 (protein, position, aa) triples drawn without replacement; the product path is
    dist.ssm_scan + dist.select_mutations. Checked against the CPU oracle on sampled proteins."""
    from thermompnn_amd.dist import select_mutations, ssm_scan
    rng = np.random.default_rng(2)
    lens = rng.integers(40, 73, size=300)
    prots = [_synthetic_protein(L, 5000 + i) for i, L in enumerate(lens)]
    T = int(lens.sum())
    flat = rng.choice(20 * T, size=200000, replace=False)
    res_of = flat // 20
    starts = np.concatenate([[0], np.cumsum(lens)])
    pid = np.searchsorted(starts, res_of, side="right") - 1
    triples = np.stack([pid, res_of - starts[pid], flat % 20], 1)
    tables = ssm_scan(engine, prots)
    vals = select_mutations(tables, triples).cpu().numpy()
    assert vals.shape == (200000,) and np.isfinite(vals).all()
    assert (lens < 48).sum() > 50                                            # the K_eff < 48 path is exercised
    for i in (0, 7, 150, 299, int(lens.argmin()), int(lens.argmax())):
        want = _oracle_table(synthetic_weights, prots[i])
        sel = pid == i
        np.testing.assert_allclose(vals[sel], want[triples[sel, 1], triples[sel, 2]], atol=TOL_DDG, rtol=0)
    with pytest.raises(IndexError):
        select_mutations(tables, np.array([[0, 9999, 0]]))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _torchrun(args, env, timeout=900):
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + args
    return subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True, timeout=timeout)


def test_two_rank_scan_with_the_real_engine(tmp_path, engine):
    """Two ranks (both on cuda:0, gloo group — the box has one GPU) run the REAL engine through dist.ssm_scan: LPT shards,
    one all-gather of [rows, 22] tables. Every table equals the 1-rank run bit for bit."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(GOLDEN))
    worker = os.path.join(repo, "tests", "dist_worker.py")
    one = str(tmp_path / "one.npz")
    r1 = subprocess.run([sys.executable, worker, one, "24"], capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    two = str(tmp_path / "two.npz")
    r2 = _torchrun([worker, two, "24"], {"TMPNN_ONE_DEVICE": "1"})
    assert r2.returncode == 0, r2.stdout[-1500:] + r2.stderr[-2500:]
    a, b = np.load(one), np.load(two)
    assert int(a["world"]) == 1 and int(b["world"]) == 2 and (b["shard_sizes"] > 0).all()
    for i in range(24):
        np.testing.assert_array_equal(a[f"t{i}"], b[f"t{i}"])
        np.testing.assert_array_equal(a[f"c{i}"], b[f"c{i}"])
    if torch.cuda.device_count() >= 2:                                       # RCCL proper needs one GPU per rank
        nc = str(tmp_path / "nccl.npz")
        r3 = _torchrun([worker, nc, "24"], {})
        assert r3.returncode == 0, r3.stdout[-1500:] + r3.stderr[-2500:]
        c = np.load(nc)
        for i in range(24):
            np.testing.assert_array_equal(a[f"t{i}"], c[f"t{i}"])


def test_two_rank_bench_and_cli(tmp_path):
    """bench.py --gpus 2 (one-device smoke mode) prints the contract line for n_gpus = 2, and the many-PDB CLI under
    torchrun writes the same CSV as the single-process run."""
    import json
    repo = os.path.dirname(os.path.dirname(GOLDEN))
    r = _torchrun([os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--proteins-per-gpu", "2"],
                  {"TMPNN_BENCH_ONE_DEVICE": "1", "TMPNN_BENCH_BACKEND": "gloo", "TMPNN_BENCH_WARMUP_SKEW": "1",
                   "TMPNN_BENCH_WATCHDOG": "120", "TMPNN_E2E_FILES": "16"})     # ranks warm up for different times: no collective may sit in that loop
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["preds_per_step"] == 2 * 2 * 256 * 20
    assert d["collective"]["ranks_seen"] == 2 and d["collective"]["world_size"] == 2
    assert d["excl_collective"]["value"] > 0 and d["excl_collective"]["ms_per_step"] > 0      # weak mode separates compute from the exchange too
    from thermompnn_amd import ssm_scan
    pdbs = [os.path.join(GOLDEN, "2OCJ.pdb"), os.path.join(GOLDEN, "2OCJ_gap_chainA.pdb")]
    single = ssm_scan.main(pdbs + ["--synthetic_weights", "0", "--centrality", "--out", str(tmp_path / "one.csv")])
    r = _torchrun(["-m", "thermompnn_amd.ssm_scan"] + pdbs + ["--synthetic_weights", "0", "--centrality", "--out",
                                                            str(tmp_path / "two.csv")], {"TMPNN_ONE_DEVICE": "1", "PYTHONPATH": repo})
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    assert open(single).read() == open(tmp_path / "two.csv").read()
    # an explicit mutation list selects rows of the same tables
    mlist = tmp_path / "muts.csv"
    mlist.write_text("pdb,position,mutation\n2OCJ,5,W\n2OCJ_gap_chainA,17,A\n2OCJ,5,A\n")
    out = ssm_scan.main(pdbs + ["--synthetic_weights", "0", "--mutations", str(mlist), "--out", str(tmp_path / "sel.csv")])
    import csv
    rows = list(csv.DictReader(open(out)))
    full = {(r_["pdb"], r_["position"], r_["mutation"]): r_["ddG_pred"] for r_ in csv.DictReader(open(single))}
    assert [(r_["pdb"], r_["position"], r_["mutation"]) for r_ in rows] == [("2OCJ", "5", "W"), ("2OCJ_gap_chainA", "17", "A"), ("2OCJ", "5", "A")]
    assert all(full[(r_["pdb"], r_["position"], r_["mutation"])] == r_["ddG_pred"] for r_ in rows)


def _bench(args, env=None, timeout=900):
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(GOLDEN))
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py")] + args, env=dict(os.environ, **(env or {})),
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]      # ONE line on stdout, nothing else (library chatter -> stderr)
    return json.loads(lines[0])


def test_profile_hook_records_one_kernel_when_asked(engine):
    """tmpnn_profile_select (include/tmpnn_debug.h): with a name set, ONLY that kernel's launches are bracketed by events —
    bench.py's timed region uses it for the dominant kernel; without one, every launch is."""
    import bench
    from thermompnn_amd import _lib
    lib = _lib.load()
    eng = engine
    # a launch with at most one tile per workgroup computes the k-NN rows inside the featurizer launch, fuses the edge update
    # with the next message pass and (round 6) the last node update with the ddG head: 13 launches, other names
    small = bench.build_batch(2, 64, 0, torch.device("cuda:0"))
    lib.tmpnn_profile_enable(1)
    eng.ssm_forward(small["X"], small["S"], small["mask"], small["ridx"], small["cenc"], small["offsets"], max_len=64)
    fused = bench.fetch_profile(lib)
    lib.tmpnn_profile_enable(0)
    assert fused["edge_msg_fused"][1] == 3 and fused["enc_msg"][1] == 1 and fused["dec_msg"][1] == 2 and "enc_edge" not in fused
    assert fused["node_update"][1] == 6 and "knn" not in fused and "head" not in fused and sum(int(v[1]) for v in fused.values()) == 13
    b = bench.build_batch(8, 64, 0, torch.device("cuda:0"))           # 512 tiles > #CUs: the 18-launch form
    fwd = lambda: eng.ssm_forward(b["X"], b["S"], b["mask"], b["ridx"], b["cenc"], b["offsets"], max_len=64)
    fwd()
    lib.tmpnn_profile_enable(1)
    fwd()
    everything = bench.fetch_profile(lib)
    lib.tmpnn_profile_enable(0)
    assert {"knn", "featurize", "enc_msg", "node_update", "enc_edge", "dec_msg", "head"} <= set(everything)
    assert everything["enc_edge"][1] == 3 and everything["node_update"][1] == 6 and sum(int(v[1]) for v in everything.values()) == 18
    try:
        lib.tmpnn_profile_select(b"enc_edge")
        lib.tmpnn_profile_enable(1)
        fwd()
        fwd()
        only = bench.fetch_profile(lib)
        lib.tmpnn_profile_enable(0)
    finally:
        lib.tmpnn_profile_select(None)
    assert set(only) == {"enc_edge"} and only["enc_edge"][1] == 6 and only["enc_edge"][0] > 0
    lib.tmpnn_profile_enable(1)
    fwd()
    again = bench.fetch_profile(lib)
    lib.tmpnn_profile_enable(0)
    assert set(again) == set(everything)


def test_bench_line_times_the_dominant_kernel_inside_the_timed_region():
    d = _bench(["--steps", "6", "--warmup", "2", "--proteins-per-gpu", "8", "--no-extras", "--no-cpu-baseline"])
    r, k = d["roofline"], d["kernels"]
    dom = r["kernel"]
    assert k[dom]["timed"].startswith("inside the timed region") and k[dom]["launches"] == 6 * k[dom]["launches_per_step"]
    assert abs(k[dom]["avg_ms"] - r["avg_launch_ms"]) < 1e-12 and r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1
    assert all(v["timed"].startswith("un-timed pass") for n, v in k.items() if n != dom)
    assert k[dom]["avg_ms"] * k[dom]["launches_per_step"] >= 0.9 * max(v["avg_ms"] * v["launches_per_step"] for v in k.values())
    assert d["pipeline"]["gpu_kernel_ms_per_step"] <= d["ms_per_step"] * 1.10      # (event-timed launches include their hand-over)


def test_bench_self_launches_and_proves_its_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (the contract command) starts itself under
    torch.distributed.run and echoes what the process group saw: world size, all_reduce(ones), one identity per rank."""
    smoke = {"TMPNN_BENCH_ONE_DEVICE": "1", "TMPNN_BENCH_BACKEND": "gloo", "TMPNN_BENCH_WATCHDOG": "300", "TMPNN_E2E_FILES": "24"}
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    os_env_backup = dict(os.environ)
    try:
        os.environ.clear()
        os.environ.update(env)
        d = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--proteins-per-gpu", "2"], smoke)
    finally:
        os.environ.clear()
        os.environ.update(os_env_backup)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["preds_per_step"] == 2 * 2 * 256 * 20
    c = d["collective"]
    assert c["world_size"] == 2 and c["ranks_seen"] == 2 and len(c["devices"]) == 2 and c["backend"].startswith("gloo")
    assert sorted(x["rank"] for x in c["devices"]) == [0, 1] and c["one_device_smoke_mode"] is True
    sp = d["ms_per_step_spread"]
    assert sp["n"] == 2 and sp["min"] <= sp["median"] <= sp["max"]
    # the many-PDB scan on both ranks rides in the same line (round 5): sharded CSV writer + gather of the binary tables
    e2e = d["end_to_end"]
    assert e2e["ranks"] == 2 and e2e["files"] == 24 and e2e["to_csv"]["rows"] == e2e["preds"] and e2e["to_npz"]["rows"] == e2e["residues"], e2e
    assert e2e["to_csv"]["preds_per_s"] > 0 and len(e2e["to_csv"]["wall_s_all_runs"]) == 3


def test_rccl_runs_on_this_box_with_one_rank():
    """The pool hands out 1-GPU boxes and RCCL refuses two ranks on one device, so the N > 1 tests above go through gloo. This one
    runs RCCL ITSELF with a world of one: init_process_group("nccl"), all_reduce / all_gather_object / all_gather_into_tensor on
    device buffers, the asynchronous double-buffered exchange of bench.py (weak and strong mode), and the product's
    dist.all_gather_tables on device memory — every call of the multi-GPU path, on a communicator of one."""
    import subprocess
    import sys
    d = _bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--proteins-per-gpu", "2", "--no-extras"], {"TMPNN_BENCH_FORCE_GROUP": "1"})
    c = d["collective"]
    assert c["backend"].startswith("nccl") and c["world_size"] == 1 and c["ranks_seen"] == 1 and c["distinct_devices"] == 1
    plain = _bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--proteins-per-gpu", "2", "--no-extras"])
    assert "collective" not in plain and d["config"]["preds_per_step"] == plain["config"]["preds_per_step"]
    s1 = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--scaling", "strong"], {"TMPNN_BENCH_FORCE_GROUP": "1"})
    s0 = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--scaling", "strong"])
    assert s1["checksum_listed"] == s0["checksum_listed"] and s1["excl_collective"]["value"] > 0     # gathered by RCCL == computed locally
    repo = os.path.dirname(os.path.dirname(GOLDEN))
    code = ("import os, sys, socket, torch, torch.distributed as dist\n"
            "sys.path.insert(0, %r)\n"
            "from thermompnn_amd.dist import all_gather_tables\n"
            "sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()\n"
            "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')\n"
            "torch.cuda.set_device(0)\n"
            "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
            "x = torch.arange(5 * 22, dtype=torch.float32, device='cuda:0').reshape(5, 22)\n"
            "got = all_gather_tables(x, [5])\n"
            "assert dist.get_backend() == 'nccl' and len(got) == 1 and got[0].is_cuda and torch.equal(got[0], x)\n"
            "dist.barrier(); dist.destroy_process_group(); print('rccl-ok')\n" % repo)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl-ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_bench_strong_scaling_mode_is_rank_invariant():
    """--scaling strong = BASELINE configs[3] (300 proteins / 200 000 listed mutants, fixed total work). The listed values
    gathered from two ranks are the same numbers as from one (their float64 sum is bit-equal)."""
    one = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--scaling", "strong"])
    two = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--scaling", "strong"],
                 {"TMPNN_BENCH_ONE_DEVICE": "1", "TMPNN_BENCH_BACKEND": "gloo", "TMPNN_BENCH_WATCHDOG": "300"})
    for d, n in ((one, 1), (two, 2)):
        assert d["scaling"] == "strong" and d["n_gpus"] == n and d["config"]["preds_per_step"] == 200000
        assert abs(d["value"] - 200000 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    assert one["checksum_listed"] == two["checksum_listed"] and np.isfinite(one["checksum_listed"])
    assert two["collective"]["ranks_seen"] == 2 and two["excl_collective"]["value"] > 0


def test_captured_graph_owns_its_buffers(synthetic_weights):
    """A graph captured on a small protein keeps replaying correctly after the engine's shared workspace has been
    reallocated by a larger forward and by a larger capture (the captured launches used to point into it)."""
    small, big = packed_inputs(load_golden("syn_L32")), packed_inputs(load_golden("syn_L256"))
    args = lambda p: (p["X"], p["S"], p["mask"], p["ridx"], p["cenc"], p["offsets"])
    from thermompnn_amd.engine import Engine
    want = Engine(synthetic_weights, "cuda:0", 48).ssm_forward(*args(small))["ddg"].clone()
    eng = Engine(synthetic_weights, "cuda:0", 48)                               # an engine whose workspace does not exist yet
    graph, out = eng.capture_graph(*args(small), max_len=small["L"])
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out["ddg"], want)
    big_ddg = eng.ssm_forward(*args(big))["ddg"].clone()                        # (re)allocates the engine's own workspace
    g2, out2 = eng.capture_graph(*args(big), max_len=big["L"])                  # and a second, larger graph
    junk = [torch.full((1 << 20,), float("nan"), device="cuda:0") for _ in range(8)]   # recycle whatever was freed
    del junk
    out["ddg"].zero_()
    graph.replay()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out["ddg"], want) and torch.equal(out2["ddg"], big_ddg)
    eng.check_graph_status(graph)
    eng.check_graph_status(g2)


def test_range_overflow_is_detected_and_retried(synthetic_weights):
    """f16x2 needs every GEMM operand below 65504. With W_edge scaled by 1e6 the featurizer overflows fp16: the head
    kernel raises TMPNN_STATUS_RANGE on the device, Engine.ssm_forward reruns the batch in bf16x3 with a warning (the
    scaled network is still finite there: the LayerNorm behind W_edge is scale-invariant) — or raises when no retry
    precision is set. TransferModel.forward goes through the same path."""
    import warnings
    from thermompnn_amd._lib import TmpnnRangeError
    from thermompnn_amd.engine import Engine
    W = {k: v.clone() for k, v in synthetic_weights.items()}
    W["prot_mpnn.features.edge_embedding.weight"] = W["prot_mpnn.features.edge_embedding.weight"] * 1e6
    p = packed_inputs(load_golden("syn_L32"))
    args = (p["X"], p["S"], p["mask"], p["ridx"], p["cenc"], p["offsets"])
    want = Engine(W, "cuda:0", 48, precision="bf16x3").ssm_forward(*args)["ddg"]
    assert bool(torch.isfinite(want).all())
    eng = Engine(W, "cuda:0", 48, precision="f16x2")
    raw = eng.ssm_forward(*args, check_status=False, want_hidden=True)
    assert not bool(torch.isfinite(raw["hidden"]).all())                     # the overflow is real: poisoned decoder states
    # (the ddG values themselves may look finite — the head's ReLUs map NaN to 0, as IEEE max does — which is exactly why
    #  the head kernel checks its INPUT and raises the device flag)
    with pytest.raises(TmpnnRangeError, match="bf16x3"):
        eng.check_last_status()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = eng.ssm_forward(*args)["ddg"]
    assert torch.equal(got, want) and any("bf16x3" in str(w.message) for w in rec)
    strict = Engine(W, "cuda:0", 48, precision="f16x2", retry_precision=None)
    with pytest.raises(TmpnnRangeError):
        strict.ssm_forward(*args)


def test_range_flag_covers_the_head_and_hidden_only_forwards(synthetic_weights):
    """(1) an overflow INSIDE the head's own f16x2 GEMMs (centre tap x 1e6: relu(Wc x + bc) leaves the fp16 range while
    every fp32 value stays finite) used to come out of the ReLUs as a finite, wrong ddG with no flag; (2) a forward that only
    returns hidden states has no head / log-probability kernel to look at them. Both raise TMPNN_STATUS_RANGE now."""
    import warnings
    from thermompnn_amd._lib import TmpnnRangeError
    from thermompnn_amd.engine import Engine
    p = packed_inputs(load_golden("syn_L32"))
    args = (p["X"], p["S"], p["mask"], p["ridx"], p["cenc"], p["offsets"])
    W = {k: v.clone() for k, v in synthetic_weights.items()}
    W["light_attention.feature_convolution.weight"] = W["light_attention.feature_convolution.weight"] * 1e6
    want = Engine(W, "cuda:0", 48, precision="fp32").ssm_forward(*args)["ddg"]
    assert bool(torch.isfinite(want).all()) and float(want.abs().max()) > 1e3
    strict = Engine(W, "cuda:0", 48, precision="f16x2", retry_precision=None)
    with pytest.raises(TmpnnRangeError):
        strict.ssm_forward(*args)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = Engine(W, "cuda:0", 48, precision="f16x2").ssm_forward(*args)["ddg"]          # reruns in bf16x3
    assert any("bf16x3" in str(w.message) for w in rec)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=1e-5 * float(want.abs().max()))   # values ~1e6: fp32 noise
    # a decoder weight: the poisoned state is caught at the head's input
    W2 = {k: v.clone() for k, v in synthetic_weights.items()}
    W2["prot_mpnn.decoder_layers.2.W1.weight"] = W2["prot_mpnn.decoder_layers.2.W1.weight"] * 1e7
    with pytest.raises(TmpnnRangeError):
        Engine(W2, "cuda:0", 48, precision="f16x2", retry_precision=None).ssm_forward(*args)
    # hidden states only
    W3 = {k: v.clone() for k, v in synthetic_weights.items()}
    W3["prot_mpnn.features.edge_embedding.weight"] = W3["prot_mpnn.features.edge_embedding.weight"] * 1e6
    with pytest.raises(TmpnnRangeError):
        Engine(W3, "cuda:0", 48, precision="f16x2", retry_precision=None).ssm_forward(*args, want_ddg=False, want_hidden=True)
    ok = Engine(synthetic_weights, "cuda:0", 48).ssm_forward(*args, want_ddg=False, want_hidden=True)
    assert bool(torch.isfinite(ok["hidden"]).all())


def test_max_len_is_guarded(engine):
    """A max_len below the longest protein can no longer overrun the kNN kernel's LDS row: host offsets are checked in
    Python, device offsets by the kernel (empty rows + TMPNN_STATUS_MAXLEN)."""
    from thermompnn_amd._lib import TmpnnError
    p = packed_inputs(load_golden("syn_L256"))
    args = (p["X"], p["S"], p["mask"], p["ridx"], p["cenc"])
    with pytest.raises(TmpnnError, match="max_len"):
        engine.ssm_forward(*args, p["offsets"], max_len=100)                 # device-resident offsets: the kernel's guard
    with pytest.raises(TmpnnError, match="smaller than the longest"):
        engine.ssm_forward(*args, p["offsets"].cpu(), max_len=100)           # host offsets: checked before the launch
    ok = engine.ssm_forward(*args, p["offsets"], max_len=300)["ddg"]
    assert torch.equal(ok, engine.ssm_forward(*args, p["offsets"])["ddg"])


def test_real_weights_2OCJ():
    """The reference's published table for examples/2OCJ.pdb (real v_48_020 + ThermoMPNN weights). Skipped
    unless TMPNN_REAL_WEIGHTS_DIR holds vanilla_model_weights/v_48_020.pt and thermoMPNN_default.pt."""
    root = os.environ.get("TMPNN_REAL_WEIGHTS_DIR", "")
    vanilla = os.path.join(root, "vanilla_model_weights", "v_48_020.pt")
    ckpt = os.path.join(root, "thermoMPNN_default.pt")
    if not (root and os.path.exists(vanilla) and os.path.exists(ckpt)):
        pytest.skip("real weights not supplied (TMPNN_REAL_WEIGHTS_DIR)")
    from thermompnn_amd import custom_inference
    model = custom_inference.load_model(ckpt, root, None)
    rows = custom_inference.ssm_rows(model, os.path.join(GOLDEN, "2OCJ.pdb"), "A")
    want = np.load(os.path.join(GOLDEN, "2OCJ_A_realweights_ddg.npz"))["ddg"]
    got = np.array([r["ddG_pred"] for r in rows], dtype=np.float64).reshape(194, 20)
    np.testing.assert_allclose(got, want, atol=TOL_DDG, rtol=0)


def test_benchmark_dataset_evaluation(tmp_path, synthetic_weights):
    """ddgBenchDataset rows -> TransferModel (per protein, the reference's loop) and the batched
    ssm_scan path -> metrics. Predictions equal the reference golden tables at the listed positions; both paths agree."""
    from thermompnn_amd import custom_inference
    from thermompnn_amd.datasets import ddgBenchDataset
    from thermompnn_amd.metrics import get_metrics
    from thermompnn_amd.thermompnn_benchmarking import run_prediction_batched, run_prediction_default
    ds = ddgBenchDataset(None, GOLDEN, os.path.join(GOLDEN, "ddgbench_sample.csv"))
    model = custom_inference.load_model(None, None, 0)
    results, rows = run_prediction_default("ThermoMPNN", model, "sample", ds, [], keep_preds=True)
    g = {"2OCJ": load_golden("2OCJ_A")["ddg"], "2OCJ_gap_chainA": load_golden("2OCJ_A_gap")["ddg"]}
    assert len(rows) == 6 and results[0]["n"] == 6                       # Q100E has no measurement, A999G no residue
    for r in rows:
        want = g[r["pdb"]][r["position"], "ACDEFGHIKLMNPQRSTVWY".index(r["mutation"])]
        assert abs(r["ddG_pred"] - want) <= TOL_DDG
    met = get_metrics([r["ddG_pred"] for r in rows], [r["ddG_true"] for r in rows])
    assert all(abs(results[0][f"ddG {k}"] - met[k]) < 1e-12 for k in ("r2", "mse", "rmse", "spearman", "pearson"))
    batched = run_prediction_batched("ThermoMPNN", model.engine(), "sample", ds, [])
    assert batched[0]["n"] == 6 and all(abs(batched[0][f"ddG {k}"] - results[0][f"ddG {k}"]) < 1e-5
                                        for k in ("r2", "mse", "rmse", "spearman", "pearson"))


def test_bench_contract_line():
    """bench.py prints ONE JSON line with the contract's keys (tiny workload; no CPU baseline leg)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(GOLDEN))
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--proteins-per-gpu", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TMPNN_E2E_FILES="48"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    e2e = d["end_to_end"]                                # PDB files -> CSV / binary wall clock (48 files here, 1 024 by default)
    assert e2e["files"] == 48 and e2e["to_csv"]["rows"] == e2e["preds"] and e2e["to_npz"]["rows"] == e2e["residues"], e2e
    assert e2e["to_csv"]["preds_per_s"] > 0 and e2e["to_npz"]["preds_per_s"] > 0 and e2e["custom_inference_2OCJ"]["rows"] == 3880
    assert d["roofline"]["issue"] is None or 0 < d["roofline"]["issue"]["frac"] < 1.5
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "preds/s" and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 2 * 256 * 20 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["bound"] == ("hbm" if rf["t_hbm_roof_us"] >= rf["t_mfma_roof_us"] else "mfma")
    assert 0 < d["pipeline"]["frac_of_binding_roof"] < 1 and d["ms_per_step_spread"]["n"] == 2
    assert d["roofline_gather"]["bound"] == "hbm" and d["roofline_gather"]["unit"] == "GB/s"
