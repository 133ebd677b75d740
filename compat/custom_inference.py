"""``analysis/custom_inference.py`` of the reference as a script name -> thermompnn_amd.custom_inference."""
import _repo  # noqa: F401
from thermompnn_amd.custom_inference import *  # noqa: F401,F403
from thermompnn_amd.custom_inference import main

if __name__ == "__main__":
    main()
