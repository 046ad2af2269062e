"""Bare-name shim: put this directory first on sys.path and the reference drivers'
`from pyESN import ESN` (e.g. Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:9) binds the HIP-backed ESN."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from esn_ofdm_mimo_amd.pyESN import ESN, correct_dimensions, identity  # noqa: E402,F401
