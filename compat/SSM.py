"""``SSM`` of the reference (/root/reference/analysis/SSM.py) -> thermompnn_amd."""
import _repo  # noqa: F401
from thermompnn_amd.ssm import get_ssm_mutations  # noqa: F401
from thermompnn_amd.ssm_scan import main, retrieve_best_mutants, scan_datasets  # noqa: F401

if __name__ == "__main__":
    main()
