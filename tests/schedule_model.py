"""Torch-CPU model of the HIP engine's *schedule* (test helper, never imported by the product).

Checks on CPU that the algebraic restructuring the kernels use stays inside the parity tolerance:
  * positional-encoding fold:  E_pos . W_edge[:, :16]^T  ==  PosTable[d]           (66 x 128 table)
  * W1 . [h_i | e_ij | h_j]  ==  W1a.h_i + W1b.e_ij + W1c.h_j                    (node terms once per node)
  * sum_k mask_k (W3.m_k + b3)  ==  W3.(sum_k mask_k m_k) + b3 . sum_k mask_k     (W3 once per node)
  * decoder:  W1c . W_s[S_j]  ==  SeqTable_l[S_j]                                 (21 x 128 table per layer)
"""
import torch
import torch.nn.functional as F

from oracle import thermompnn_oracle as orc


def ln(x, W, p):
    return F.layer_norm(x, (128,), W[p + ".weight"], W[p + ".bias"], 1e-5)


def node_update(W, p, h, Ssum, cnt, mask):
    dh = (F.linear(Ssum, W[p + ".W3.weight"]) + cnt[..., None] * W[p + ".W3.bias"]) / 30.0
    h = ln(h + dh, W, p + ".norm1")
    h = ln(h + orc.ffn(h, W, p + ".dense"), W, p + ".norm2")
    return mask[..., None] * h


def mpnn_schedule(W, X, mask, S, residue_idx, chain_enc, K=48):
    """Single protein, unbatched tensors [L,...]. Returns (hV list dec1..3, h_E final)."""
    Xb, mb = X[None], mask[None]
    atoms = orc.backbone_atoms(Xb)
    D_nb, E_idx = orc.knn(atoms[orc.CA], mb, K)
    E_idx = E_idx[0]
    L, Ke = E_idx.shape
    blocks = [orc.rbf(D_nb)[0]]
    for a, b in orc.PAIR_ORDER[1:]:
        d = torch.sqrt(((atoms[a][0][:, None, :] - atoms[b][0][E_idx]) ** 2).sum(-1) + 1e-6)
        blocks.append(orc.rbf(d[None])[0])
    rbf = torch.cat(blocks, -1)                                           # [L,K,400]
    We = W["features.edge_embedding.weight"]
    pos_table = (W["features.embeddings.linear.weight"].t() + W["features.embeddings.linear.bias"]) @ We[:, :16].t()
    d = orc.positional_index(residue_idx[None], chain_enc[None], E_idx[None])[0]
    E = ln(pos_table[d] + rbf @ We[:, 16:].t(), W, "features.norm_edges")
    h_E = F.linear(E, W["W_e.weight"], W["W_e.bias"])
    h = torch.zeros(L, 128)
    m_att = mask[:, None] * mask[E_idx]
    for l in range(3):
        p = f"encoder_layers.{l}"
        W1 = W[p + ".W1.weight"]
        A = F.linear(h, W1[:, :128], W[p + ".W1.bias"])
        Cn = F.linear(h, W1[:, 256:])
        m1 = F.gelu(A[:, None] + Cn[E_idx] + h_E @ W1[:, 128:256].t())
        m2 = F.gelu(F.linear(m1, W[p + ".W2.weight"], W[p + ".W2.bias"]))
        h = node_update(W, p, h, (m_att[..., None] * m2).sum(1), m_att.sum(1), mask)
        W11 = W[p + ".W11.weight"]
        A = F.linear(h, W11[:, :128], W[p + ".W11.bias"])
        Cn = F.linear(h, W11[:, 256:])
        m1 = F.gelu(A[:, None] + Cn[E_idx] + h_E @ W11[:, 128:256].t())
        m2 = F.gelu(F.linear(m1, W[p + ".W12.weight"], W[p + ".W12.bias"]))
        h_E = ln(h_E + F.linear(m2, W[p + ".W13.weight"], W[p + ".W13.bias"]), W, p + ".norm3")
    hs = []
    for l in range(3):
        p = f"decoder_layers.{l}"
        W1 = W[p + ".W1.weight"]
        seq_table = W["W_s.weight"] @ W1[:, 256:384].t()                  # [21,128]
        A = F.linear(h, W1[:, :128], W[p + ".W1.bias"])
        Dn = F.linear(h, W1[:, 384:])
        inner = h_E @ W1[:, 128:256].t() + seq_table[S][E_idx] + Dn[E_idx]
        m1 = F.gelu(A[:, None] + mask[:, None, None] * inner)
        m2 = F.gelu(F.linear(m1, W[p + ".W2.weight"], W[p + ".W2.bias"]))
        h = node_update(W, p, h, m2.sum(1), torch.full((L,), float(Ke)), mask)
        hs.append(h)
    return hs, h_E, E_idx
