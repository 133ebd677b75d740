"""The engine's restructured schedule (hoisted W3, split W1, folded tables) is parity-neutral."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import thermompnn_oracle as orc
from schedule_model import mpnn_schedule


@pytest.mark.parametrize("case", ["2OCJ_A", "2OCJ_A_gap", "2OCJ_AB", "syn_L32"])
def test_schedule_is_parity_neutral(case, synthetic_weights):
    g = load_golden(case)
    mp, hd = orc.split_weights(synthetic_weights)
    t = torch.from_numpy
    S = t(g["S"].astype(np.int64))
    with torch.no_grad():
        hs, h_E, _ = mpnn_schedule(mp, t(g["X"]), t(g["mask"]), S, t(g["residue_idx"].astype(np.int64)),
                                   t(g["chain_enc"].astype(np.int64)))
        _, ddg = orc.head_table(hd, [hs[2][None], hs[1][None]], mp["W_s.weight"][S][None], S[None])
    np.testing.assert_allclose(hs[2].numpy(), g["hV_dec3"], atol=1e-5, rtol=0)
    have = ~np.isnan(g["ddg"][:, 0])
    np.testing.assert_allclose(ddg[0].numpy()[have][:, :20], g["ddg"][have], atol=1e-4, rtol=0)
