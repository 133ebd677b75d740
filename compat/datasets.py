"""``datasets`` of the reference (/root/reference/datasets.py) -> thermompnn_amd."""
import _repo  # noqa: F401
from thermompnn_amd.datasets import ALPHABET, FireProtDataset, Mutation, ddgBenchDataset  # noqa: F401


class MegaScaleDataset:
    """The training / validation set (datasets.py:34-164: a 2 GB CSV plus pickled splits). Only its name is needed for the
    reference drivers' import lines; evaluating on it goes through ddgBenchDataset-style CSVs."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("MegaScaleDataset (training data) is outside the MI355X inference engine's scope")
