import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thermompnn_amd.engine import Engine
from thermompnn_amd.weights import synthetic_state_dict
from thermompnn_amd.synthetic import synthetic_backbone
W = synthetic_state_dict(0)
W["prot_mpnn.features.edge_embedding.weight"] = W["prot_mpnn.features.edge_embedding.weight"] * 1e6
L = 32
X, seq = synthetic_backbone(L, 1)
S = torch.tensor(["ACDEFGHIKLMNPQRSTVWY".index(c) for c in seq], dtype=torch.int32)
eng = Engine(W, "cuda:0", 48, precision="f16x2")
r = eng.ssm_forward(torch.tensor(X, dtype=torch.float32), S, torch.ones(L), torch.arange(L), torch.ones(L), torch.tensor([0, L], dtype=torch.int32), check_status=False, want_hidden=True, want_log_probs=True)
print("status", int(eng._status.item()))
for k in range(3):
    h = r["hidden"][k]
    print("hidden", k, "finite frac", float(torch.isfinite(h).float().mean()), "nan", int(torch.isnan(h).sum()), "inf", int(torch.isinf(h).sum()))
print("ddg finite", bool(torch.isfinite(r["ddg"]).all()), "logp finite", bool(torch.isfinite(r["log_probs"]).all()))
