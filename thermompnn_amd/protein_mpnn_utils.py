"""Drop-in counterpart of the hot-path symbols of /root/reference/protein_mpnn_utils.py.

Same names, signatures and return conventions; the arithmetic runs in libtmpnn.so
(hand-written HIP for gfx950) — this module only packs arguments. Out-of-scope symbols of the
reference file (CA_ProteinFeatures, sample/tied_sample, StructureDataset*, loss_*) are not provided.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import weights as _weights
from .engine import Engine, cat_neighbors_nodes, gather_edges, gather_nodes  # noqa: F401  (API surface)
from .pdb_io import alt_parse_PDB, featurize, tied_featurize  # noqa: F401  (API surface)


def _register_tree(root: nn.Module, shapes) -> None:
    """Register parameters under the reference's dotted state-dict names (containers only; no forward)."""
    for name, shape in shapes.items():
        mod = root
        *path, leaf = name.split(".")
        for part in path:
            if part not in mod._modules:
                mod.add_module(part, nn.Module())
            mod = mod._modules[part]
        p = torch.empty(shape)
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)                 # protein_mpnn_utils.py:1217-1219
        elif leaf == "weight":
            nn.init.ones_(p)
        else:
            nn.init.zeros_(p)
        mod.register_parameter(leaf, nn.Parameter(p))


class _EngineOwner(nn.Module):
    """Builds (and rebuilds after load_state_dict / .to()) the native engine from the module's parameters."""

    _engine = None
    _engine_key = None
    k_neighbors = 48
    precision = None          # matrix-core path of the engine ("f16x2" | "bf16x3" | "fp32"; None = library default)
    retry_precision = "bf16x3"   # Engine reruns an f16x2 forward that left the fp16 range at this precision (None: raise)
    _engine_kwargs: dict = {}    # extra Engine(...) arguments of a subclass (TransferModel with a generic head: with_head=False)

    def _state_for_engine(self):
        return {k: v for k, v in self.state_dict().items()}

    def engine(self) -> Engine:
        params = list(self.parameters())
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("thermompnn_amd runs on MI355X only: move the model to a CUDA (ROCm) device "
                               "with .cuda(); there is no CPU execution path")
        key = (tuple((p.data_ptr(), p._version) for p in params), self.precision, self.retry_precision)
        if self._engine is None or key != self._engine_key:
            self._engine = Engine(self._state_for_engine(), dev, self.k_neighbors, precision=self.precision,
                                  retry_precision=self.retry_precision, **self._engine_kwargs)
            self._engine_key = key
        return self._engine


class ProteinMPNN(_EngineOwner):
    """Constructor/forward signature of the reference class (protein_mpnn_utils.py:1184-1277)."""

    def __init__(self, num_letters, node_features, edge_features, hidden_dim, num_encoder_layers=3,
                 num_decoder_layers=3, vocab=21, k_neighbors=64, augment_eps=0.05, dropout=0.1, ca_only=False):
        super().__init__()
        if ca_only:
            raise NotImplementedError("ca_only=True (CA_ProteinFeatures) is outside the ThermoMPNN hot path")
        if (num_letters, node_features, edge_features, hidden_dim, vocab) != (21, 128, 128, 128, 21) or \
                (num_encoder_layers, num_decoder_layers) != (3, 3):
            raise NotImplementedError("the HIP engine is specialised for the ThermoMPNN configuration: "
                                      "21 letters, 128-d features, 3 encoder + 3 decoder layers")
        if not 1 <= int(k_neighbors) <= 48:
            raise NotImplementedError(f"k_neighbors={k_neighbors}: this build supports 1..48 (v_48_* weights)")
        if augment_eps and augment_eps > 0:
            raise NotImplementedError("augment_eps > 0 adds training noise; ThermoMPNN uses 0.0 (transfer_model.py:27)")
        self.node_features, self.edge_features, self.hidden_dim = node_features, edge_features, hidden_dim
        self.k_neighbors = int(k_neighbors)
        _register_tree(self, _weights.mpnn_param_shapes())

    def forward(self, X, S, mask, chain_M, residue_idx, chain_encoding_all, randn, use_input_decoding_order=False,
                decoding_order=None):
        """-> (list of 3 decoder states [B,L,128] in REVERSED order, h_S [B,L,128], log_probs [B,L,21]).
        ``chain_M``, ``randn`` and the decoding-order arguments do not influence the outputs on this path
        (the reference overwrites its order mask with ones, :1259)."""
        eng = self.engine()
        B, L = X.shape[0], X.shape[1]
        offsets = torch.arange(B + 1, dtype=torch.int32) * L
        with torch.cuda.device(eng.device):
            res = eng.ssm_forward(X.reshape(B * L, 4, 3), S.reshape(-1), mask.reshape(-1), residue_idx.reshape(-1),
                                  chain_encoding_all.reshape(-1), offsets, max_len=L, want_ddg=False,
                                  want_hidden=True, want_log_probs=True)
            h_S = eng.seq_embed(S.reshape(-1)).view(B, L, -1)
        hidden = res["hidden"].view(3, B, L, -1)
        return [hidden[2], hidden[1], hidden[0]], h_S, res["log_probs"].view(B, L, -1)
