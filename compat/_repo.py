"""Makes ``thermompnn_amd`` importable when only ``compat/`` is on PYTHONPATH."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)
