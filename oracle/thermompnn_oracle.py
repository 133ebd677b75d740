"""CPU ORACLE for the ThermoMPNN SSM hot path.  *** TEST INFRASTRUCTURE — NOT THE PRODUCT ***

A plain torch-CPU fp32 restatement of the reference algorithm, function by function, citing the
reference file:line each one follows.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product path (thermompnn_amd/) never does.

Parity status: PINNED for synthetic weights — tests/golden/*.npz were produced by importing the
reference itself (tests/golden/make_golden.py) and tests/test_oracle_golden.py checks this
restatement against them.  Real-weight parity (examples/ThermoMPNN_inference_2OCJ.csv) is UNPINNED:
the checkpoints are absent from the reference mount (.MISSING_LARGE_BLOBS).

The weights argument ``W`` is a flat dict of tensors named as in the reference state dict
(thermompnn_amd/weights.py: transfer_param_shapes()).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# Order of the 25 atom-pair RBF blocks (protein_mpnn_utils.py:1143-1167); indices into (N, Ca, C, O, Cb)
N_, CA, C_, O_, CB = 0, 1, 2, 3, 4
PAIR_ORDER = [(CA, CA), (N_, N_), (C_, C_), (O_, O_), (CB, CB), (CA, N_), (CA, C_), (CA, O_), (CA, CB),
              (N_, C_), (N_, O_), (N_, CB), (CB, C_), (CB, O_), (O_, C_), (N_, CA), (C_, CA), (O_, CA),
              (CB, CA), (C_, N_), (O_, N_), (CB, N_), (C_, CB), (O_, CB), (C_, O_)]


def gather_edges(edges: Tensor, idx: Tensor) -> Tensor:
    """[B,L,L,C] at [B,L,K] -> [B,L,K,C] (protein_mpnn_utils.py:763-767)."""
    return torch.gather(edges, 2, idx.unsqueeze(-1).expand(-1, -1, -1, edges.size(-1)))


def gather_nodes(nodes: Tensor, idx: Tensor) -> Tensor:
    """[B,L,C] at [B,L,K] -> [B,L,K,C] (protein_mpnn_utils.py:770-778)."""
    B, L, K = idx.shape
    flat = idx.reshape(B, L * K, 1).expand(-1, -1, nodes.size(2))
    return torch.gather(nodes, 1, flat).view(B, L, K, -1)


def cat_neighbors_nodes(h_nodes: Tensor, h_neighbors: Tensor, idx: Tensor) -> Tensor:
    """[h_neighbors || h_nodes[idx]] (protein_mpnn_utils.py:788-791)."""
    return torch.cat([h_neighbors, gather_nodes(h_nodes, idx)], -1)


def layer_norm(x: Tensor, W: Dict[str, Tensor], prefix: str) -> Tensor:
    return F.layer_norm(x, (x.size(-1),), W[prefix + ".weight"], W[prefix + ".bias"], 1e-5)


def linear(x: Tensor, W: Dict[str, Tensor], prefix: str) -> Tensor:
    return F.linear(x, W[prefix + ".weight"], W.get(prefix + ".bias"))


def backbone_atoms(X: Tensor) -> List[Tensor]:
    """N, Ca, C, O and the virtual Cb (protein_mpnn_utils.py:1131-1138)."""
    n, ca, c, o = X[:, :, 0], X[:, :, 1], X[:, :, 2], X[:, :, 3]
    b = ca - n
    cc = c - ca
    a = torch.cross(b, cc, dim=-1)
    cb = -0.58273431 * a + 0.56802827 * b - 0.54067466 * cc + ca
    return [n, ca, c, o, cb]


def adjusted_distances(ca: Tensor, mask: Tensor, eps: float = 1e-6) -> Tensor:
    """D_adjust of ProteinFeatures._dist (protein_mpnn_utils.py:1102-1106)."""
    m2 = mask.unsqueeze(1) * mask.unsqueeze(2)
    d = ca.unsqueeze(1) - ca.unsqueeze(2)
    D = m2 * torch.sqrt((d ** 2).sum(3) + eps)
    return D + (1.0 - m2) * D.max(-1, keepdim=True)[0]


def knn(ca: Tensor, mask: Tensor, top_k: int, E_idx_override: Optional[Tensor] = None):
    """Masked Ca distances and the top_k smallest per row (protein_mpnn_utils.py:1101-1109).
    torch.topk leaves the order of EXACT ties unspecified; tests that must compare downstream tensors
    slot by slot pass ``E_idx_override`` (a graph already checked to be a valid top-k) to pin it."""
    D_adj = adjusted_distances(ca, mask)
    if E_idx_override is not None:
        return torch.gather(D_adj, 2, E_idx_override), E_idx_override
    return torch.topk(D_adj, min(top_k, ca.shape[1]), dim=-1, largest=False)


def rbf(D: Tensor, n: int = 16) -> Tensor:
    """16 Gaussians on [2, 22] Å, sigma 1.25 (protein_mpnn_utils.py:1111-1119)."""
    mu = torch.linspace(2.0, 22.0, n).view(1, 1, 1, -1)
    return torch.exp(-(((D.unsqueeze(-1) - mu) / ((22.0 - 2.0) / n)) ** 2))


def positional_index(residue_idx: Tensor, chain_labels: Tensor, E_idx: Tensor, max_rel: int = 32) -> Tensor:
    """Clipped relative offset, or 65 across chains (protein_mpnn_utils.py:1170-1175, 903-905)."""
    off = gather_edges((residue_idx[:, :, None] - residue_idx[:, None, :])[..., None], E_idx)[..., 0]
    same = gather_edges(((chain_labels[:, :, None] - chain_labels[:, None, :]) == 0).long()[..., None], E_idx)[..., 0]
    return torch.clip(off + max_rel, 0, 2 * max_rel) * same + (1 - same) * (2 * max_rel + 1)


def protein_features(W: Dict[str, Tensor], X: Tensor, mask: Tensor, residue_idx: Tensor,
                     chain_labels: Tensor, top_k: int, E_idx_override: Optional[Tensor] = None):
    """kNN graph + [posenc16 || RBF400] -> LN(W_edge .) (protein_mpnn_utils.py:1127-1180)."""
    atoms = backbone_atoms(X)
    D_nb, E_idx = knn(atoms[CA], mask, top_k, E_idx_override)
    blocks = [rbf(D_nb)]
    for a, b in PAIR_ORDER[1:]:
        A, Bt = atoms[a], atoms[b]
        D_ab = torch.sqrt(((A[:, :, None, :] - Bt[:, None, :, :]) ** 2).sum(-1) + 1e-6)   # :1122
        blocks.append(rbf(gather_edges(D_ab[..., None], E_idx)[..., 0]))
    d = positional_index(residue_idx, chain_labels, E_idx)
    E_pos = F.linear(F.one_hot(d, 66).float(), W["features.embeddings.linear.weight"],
                     W["features.embeddings.linear.bias"])                                  # :906-907
    E = torch.cat([E_pos] + blocks, -1)
    E = F.linear(E, W["features.edge_embedding.weight"])                                    # :1178 (no bias)
    return layer_norm(E, W, "features.norm_edges"), E_idx, D_nb


def ffn(h: Tensor, W, p: str) -> Tensor:
    """PositionWiseFeedForward (protein_mpnn_utils.py:883-893); exact-erf GELU."""
    return linear(F.gelu(linear(h, W, p + ".W_in")), W, p + ".W_out")


def message(h_EV: Tensor, W, p: str, names=("W1", "W2", "W3")) -> Tensor:
    a, b, c = names
    return linear(F.gelu(linear(F.gelu(linear(h_EV, W, f"{p}.{a}")), W, f"{p}.{b}")), W, f"{p}.{c}")


def enc_layer(W, p: str, h_V: Tensor, h_E: Tensor, E_idx: Tensor, mask: Tensor, mask_attend: Tensor):
    """EncLayer.forward (protein_mpnn_utils.py:816-839), eval mode (dropout = identity)."""
    K = h_E.size(-2)
    h_EV = torch.cat([h_V.unsqueeze(-2).expand(-1, -1, K, -1), cat_neighbors_nodes(h_V, h_E, E_idx)], -1)
    msg = mask_attend.unsqueeze(-1) * message(h_EV, W, p)
    h_V = layer_norm(h_V + msg.sum(-2) / 30.0, W, p + ".norm1")
    h_V = layer_norm(h_V + ffn(h_V, W, p + ".dense"), W, p + ".norm2")
    h_V = mask.unsqueeze(-1) * h_V
    h_EV = torch.cat([h_V.unsqueeze(-2).expand(-1, -1, K, -1), cat_neighbors_nodes(h_V, h_E, E_idx)], -1)
    h_E = layer_norm(h_E + message(h_EV, W, p, ("W11", "W12", "W13")), W, p + ".norm3")
    return h_V, h_E


def dec_layer(W, p: str, h_V: Tensor, h_ESV: Tensor, mask: Tensor):
    """DecLayer.forward (protein_mpnn_utils.py:859-880) with mask_attend=None (call at :1272)."""
    K = h_ESV.size(-2)
    h_EV = torch.cat([h_V.unsqueeze(-2).expand(-1, -1, K, -1), h_ESV], -1)
    h_V = layer_norm(h_V + message(h_EV, W, p).sum(-2) / 30.0, W, p + ".norm1")
    h_V = layer_norm(h_V + ffn(h_V, W, p + ".dense"), W, p + ".norm2")
    return mask.unsqueeze(-1) * h_V


def mpnn_forward(W: Dict[str, Tensor], X, S, mask, chain_M, residue_idx, chain_encoding_all,
                 top_k: int = 48, n_enc: int = 3, n_dec: int = 3, trace: Optional[dict] = None,
                 E_idx_override: Optional[Tensor] = None):
    """ProteinMPNN.forward (protein_mpnn_utils.py:1222-1277) on the ThermoMPNN path:
    order_mask_backward is overwritten with ones (:1259), so mask_bw = mask_i, mask_fw = 0 and the
    encoder-only branch contributes exact zeros.  Returns (reversed hidden list, h_S, log_probs)."""
    E, E_idx, D_nb = protein_features(W, X, mask, residue_idx, chain_encoding_all, top_k, E_idx_override)
    h_V = torch.zeros(E.shape[0], E.shape[1], E.shape[-1])
    h_E = linear(E, W, "W_e")
    if trace is not None:
        trace.update(E=E, E_idx=E_idx, D_nb=D_nb, h_E0=h_E)
    mask_attend = mask.unsqueeze(-1) * gather_nodes(mask.unsqueeze(-1), E_idx).squeeze(-1)
    for i in range(n_enc):
        h_V, h_E = enc_layer(W, f"encoder_layers.{i}", h_V, h_E, E_idx, mask, mask_attend)
        if trace is not None:
            trace[f"hV_enc{i + 1}"] = h_V
    if trace is not None:
        trace["h_E_final"] = h_E
    h_S = F.embedding(S, W["W_s.weight"])
    h_ES = cat_neighbors_nodes(h_S, h_E, E_idx)
    mask_bw = mask.view(mask.size(0), mask.size(1), 1, 1)          # mask_1D * ones (:1261-1263)
    hidden = []
    for i in range(n_dec):
        h_ESV = mask_bw * cat_neighbors_nodes(h_V, h_ES, E_idx)    # + h_EXV_encoder_fw == 0 (:1271)
        h_V = dec_layer(W, f"decoder_layers.{i}", h_V, h_ESV, mask)
        hidden.append(h_V)
        if trace is not None:
            trace[f"hV_dec{i + 1}"] = h_V
    log_probs = F.log_softmax(linear(h_V, W, "W_out"), dim=-1)
    return list(reversed(hidden)), h_S, log_probs


def light_attention(W, x: Tensor) -> Tensor:
    """LightAttention.forward on [1, 384, 1] (transfer_model.py:148-155), literal form."""
    o = F.conv1d(x, W["light_attention.feature_convolution.weight"],
                 W["light_attention.feature_convolution.bias"], padding=4)
    att = F.conv1d(x, W["light_attention.attention_convolution.weight"],
                   W["light_attention.attention_convolution.bias"], padding=4)
    return torch.squeeze(o * torch.softmax(att, dim=-1))


def both_out(W, y: Tensor) -> Tensor:
    """[ReLU, Linear] x3 (transfer_model.py:67-71)."""
    for i in (1, 3, 5):
        y = linear(F.relu(y), W, f"both_out.{i}")
    return y


def split_weights(W):
    mp = {k[len("prot_mpnn."):]: v for k, v in W.items() if k.startswith("prot_mpnn.")}
    hd = {k: v for k, v in W.items() if not k.startswith("prot_mpnn.")}
    return mp, hd


def transfer_forward_loop(W, X, S, mask, chain_M, residue_idx, chain_enc, mutations, alphabet,
                          top_k: int = 48):
    """TransferModel.forward, reference-shaped: the head is evaluated once per mutation
    (transfer_model.py:86-120).  ``mutations``: list of objects with .position/.wildtype/.mutation or None."""
    mp, hd = split_weights(W)
    hidden, h_S, _ = mpnn_forward(mp, X, S, mask, chain_M, residue_idx, chain_enc, top_k)
    hid = torch.cat(hidden[:2], -1)
    out = []
    for m in mutations:
        if m is None:
            out.append(None)
            continue
        x = torch.cat([hid[0][m.position], h_S[0][m.position]], -1)
        y = light_attention(hd, x.unsqueeze(-1).unsqueeze(0))
        z = linear(both_out(hd, y).unsqueeze(-1), hd, "ddg_out")
        out.append(z[alphabet.index(m.mutation)][0] - z[alphabet.index(m.wildtype)][0])
    return out


def head_table(W, hidden: List[Tensor], h_S: Tensor, S: Tensor):
    """Vectorised head: one evaluation per position.  On a length-1 sequence the
    feature conv reduces to its centre tap and softmax over a size-1 axis is 1
    (transfer_model.py:107-108,148-155).  Returns (z[B,L,21], ddg[B,L,21]) with
    ddg[..., a] = (w z_a + b) - (w z_wt + b) (transfer_model.py:110-116)."""
    x = torch.cat([hidden[0], hidden[1], h_S], -1)
    y = F.linear(x, W["light_attention.feature_convolution.weight"][:, :, 4],
                 W["light_attention.feature_convolution.bias"])
    z = both_out(W, y)
    zz = z * W["ddg_out.weight"].view(()) + W["ddg_out.bias"].view(())
    return z, zz - torch.gather(zz, -1, S.unsqueeze(-1))


def ssm_table(W, X, S, mask, chain_M, residue_idx, chain_enc, top_k: int = 48, trace=None, E_idx_override=None):
    """Full SSM, vectorised: -> ddg[B, L, 21] (column a = mutation to ALPHABET[a])."""
    mp, hd = split_weights(W)
    hidden, h_S, log_probs = mpnn_forward(mp, X, S, mask, chain_M, residue_idx, chain_enc, top_k, trace=trace,
                                          E_idx_override=E_idx_override)
    z, ddg = head_table(hd, hidden, h_S, S)
    if trace is not None:
        trace.update(h_S=h_S, log_probs=log_probs, z=z, ddg=ddg)
    return ddg
