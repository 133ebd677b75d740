"""``transfer_model`` of the reference (/root/reference/transfer_model.py) -> thermompnn_amd."""
import _repo  # noqa: F401
from thermompnn_amd.datasets import ALPHABET  # noqa: F401
from thermompnn_amd.transfer_model import TransferModel, get_protein_mpnn  # noqa: F401
