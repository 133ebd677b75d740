"""``model_utils.featurize`` of the reference (/root/reference/model_utils.py:19-125) -> thermompnn_amd."""
import _repo  # noqa: F401
from thermompnn_amd.pdb_io import featurize  # noqa: F401
