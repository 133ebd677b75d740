"""``protein_mpnn_utils`` of the reference (/root/reference/protein_mpnn_utils.py) -> thermompnn_amd."""
import _repo  # noqa: F401
from thermompnn_amd.pdb_io import alt_parse_PDB, featurize, tied_featurize  # noqa: F401
from thermompnn_amd.protein_mpnn_utils import ProteinMPNN, cat_neighbors_nodes, gather_edges, gather_nodes  # noqa: F401


def loss_smoothed(*args, **kwargs):
    """Training loss (protein_mpnn_utils.py:1557-1566): imported by analysis/SSM.py:12, never called on the inference path."""
    raise NotImplementedError("loss_smoothed belongs to training, which is outside the MI355X inference engine's scope")
