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


def test_message_pass_k_order_is_one_permutation_in_every_form():
    """Round 5: every f16x2 form of the message pass feeds the matrix cores in ONE K order, so their bits agree (tmpnn_split.h: perm_c4;
    tmpnn_split.hip: prep_wimg_kernel(perm); tmpnn_msg.hip: msg8_wave_kernel). Host-side restatement of the three index maps:
      * weight image: element e of lane group q in 32-deep step c holds k = 32 c + 16 (e >> 2) + 4 q + (e & 3);
      * wavefront-per-residue form: lane (n, q) holds output columns 16 cb + 4 q + i of its accumulator block cb, and blocks 2 c, 2 c + 1
        become elements 0..3, 4..7 of step c;
      * 8-wavefront form: column group c4 (columns 4 c4 .. 4 c4 + 3) is stored at perm_c4(c4) — 16-byte chunk (perm >> 1), half (perm & 1) —
        and lane group q reads chunk 4 c + q of step c.
    All three must name the same k for the same (c, q, e), and each must be a permutation of the 128 columns."""
    import os
    import re
    from conftest import REPO
    perm_c4 = lambda c4: (c4 & ~7) | ((c4 & 3) << 1) | ((c4 >> 2) & 1)
    src = open(os.path.join(REPO, "thermompnn_amd", "csrc", "tmpnn_split.h")).read()
    assert re.search(r"perm_c4\(int c4\) \{ return \(c4 & ~7\) \| \(\(c4 & 3\) << 1\) \| \(\(c4 >> 2\) & 1\); \}", src)
    image = {(c, q, e): 32 * c + 16 * (e >> 2) + 4 * q + (e & 3) for c in range(4) for q in range(4) for e in range(8)}
    assert sorted(image.values()) == list(range(128))
    wave = {}
    for cb in range(8):
        for q in range(4):
            for i in range(4):
                wave[(cb >> 1, q, 4 * (cb & 1) + i)] = 16 * cb + 4 * q + i
    assert wave == image
    planes = {}
    assert sorted(perm_c4(c4) for c4 in range(32)) == list(range(32))
    for c4 in range(32):
        p = perm_c4(c4)
        chunk, half = p >> 1, p & 1
        for i in range(4):
            planes[(chunk >> 2, chunk & 3, 4 * half + i)] = 4 * c4 + i          # lane group q = chunk & 3 of step c = chunk >> 2
    assert planes == image
