"""MI355X-native ESN OFDM/MIMO symbol detector hot path.

Drop-in modules (same names and call signatures as the reference's ``libs/``):

    from esn_ofdm_mimo_amd.pyESN import ESN
    from esn_ofdm_mimo_amd.helper_mimo_esn_generic import trainMIMOESN_generic

or put ``esn_ofdm_mimo_amd/dropin`` first on ``sys.path`` and keep the reference's
bare imports (``from pyESN import ESN``) unchanged -- see INTEGRATION.md.

All arithmetic of the path runs in the HIP library ``libesn_hip.so`` (C ABI in
``include/esn_hip.h``); there is no CPU fallback: importing the kernels without
the library, or calling them without a GPU, raises.
"""
from . import _lib  # noqa: F401

__all__ = ["pyESN", "helper_mimo_esn_generic", "batched", "montecarlo"]
