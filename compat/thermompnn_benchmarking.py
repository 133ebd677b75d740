"""``thermompnn_benchmarking`` of the reference (/root/reference/analysis/thermompnn_benchmarking.py) -> thermompnn_amd."""
import _repo  # noqa: F401
from thermompnn_amd.datasets import ALPHABET  # noqa: F401
from thermompnn_amd.metrics import get_metrics  # noqa: F401
from thermompnn_amd.thermompnn_benchmarking import (ProteinMPNNBaseline, compute_centrality, get_trained_model,  # noqa: F401
                                                     evaluate_datasets, run_prediction_batched,
                                                     run_prediction_default, run_prediction_keep_preds)
