"""Bare-name shim for `from HelpFunc import HelpFunc` (system_model_2_all_comparision.py:4)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from esn_ofdm_mimo_amd.HelpFunc import HelpFunc  # noqa: E402,F401
