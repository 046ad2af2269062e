"""Bare-name shim for `from helper_mimo_esn_generic import trainMIMOESN_generic`
(Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:11)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from esn_ofdm_mimo_amd.helper_mimo_esn_generic import trainMIMOESN_generic  # noqa: E402,F401
