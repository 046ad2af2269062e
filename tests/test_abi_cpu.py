"""CPU-only checks of the C-ABI boundary: the library builds, loads, exports every
symbol include/esn_hip.h declares, and the host-side argument logic that needs no
GPU (sizes, geometry, error strings) behaves."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from esn_ofdm_mimo_amd import build, _lib
    build.build_library(verbose=False)
    return _lib.load()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "esn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(esn_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    names = declared_functions()
    assert len(names) >= 12
    raw = C.CDLL(os.path.join(ROOT, "esn_ofdm_mimo_amd", "libesn_hip.so"))
    for n in names:
        assert hasattr(raw, n), f"{n} declared in esn_hip.h but not exported"


def test_binding_covers_header(lib):
    from esn_ofdm_mimo_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()


def test_geometry_and_sizes(lib):
    from esn_ofdm_mimo_amd._lib import Shape, F64, F32, F16, BF16
    sh = Shape(512, 16, 8, 1, 1)
    assert lib.esn_tile_frames(F64, C.byref(sh)) == 8
    assert lib.esn_tile_frames(F32, C.byref(sh)) == 64
    assert lib.esn_tile_frames(F16, C.byref(sh)) == 128
    # float64 image: K-major [n_res+n_in+n_out][n_res] for the vector-ALU kernel, then the matrix-pipe
    # kernel's fragment-ordered copy Mp x Kp, Kp = roundup(512+16+8, 8) = 536
    assert lib.esn_packed_weights_bytes(F64, C.byref(sh)) == 8 * (512 + 16 + 8) * 512 + 8 * 512 * 536
    # MFMA images: Mp x Kp elements, Kp = roundup(512+16+8, 32) = 544
    assert lib.esn_packed_weights_bytes(F32, C.byref(sh)) == 4 * 512 * 544
    # fp16 / bf16 at 257..512 units: the 32x32x16 image (harvest, A/B runs) + the 16x16x32 kernel's copy behind it
    assert lib.esn_packed_weights_bytes(F16, C.byref(sh)) == 2 * (2 * 512 * 544)
    assert lib.esn_packed_weights_bytes(BF16, C.byref(sh)) == 2 * (2 * 512 * 544)
    assert lib.esn_packed_readout_bytes(F64, C.byref(sh)) == 8 * 8 * 528 + 8 * 16 * 536
    assert lib.esn_packed_readout_bytes(F32, C.byref(sh)) == 16 * 544 * 4 + 16
    # hi rows 0-7, lo rows 8-15 (the register-state kernel's image exists only in ESN_WITH_RS=1 experiment builds)
    assert lib.esn_packed_readout_bytes(F16, C.byref(sh)) == (16 * 544 * 2 + 16) + (17 * 1024 + 16)
    assert lib.esn_packed_readout_bytes(F16, C.byref(Shape(256, 16, 8, 1, 1))) == 16 * 288 * 2 + 16
    small = Shape(100, 2, 2, 1, 1)
    assert lib.esn_tile_frames(F32, C.byref(small)) == 64
    big = Shape(2048, 16, 8, 1, 1)
    # N_res = 2048 in float64: vector-ALU kernel only (the float64 state of 16 frames does not fit LDS)
    assert lib.esn_packed_weights_bytes(F64, C.byref(big)) == 8 * (2048 + 16 + 8) * 2048
    assert lib.esn_tile_frames(F16, C.byref(big)) == 32
    assert lib.esn_tile_frames(F32, C.byref(big)) < 0          # float32 state does not fit LDS
    assert b"unsupported" in lib.esn_last_error()
    # fp16, N_res = 2048: K padded to whole 64-deep chunks (2112); the packed read-out carries the image of the
    # launch-per-step GEMM path behind the persistent kernel's; its workspace = 2 state images + 2 partial buffers
    assert lib.esn_packed_weights_bytes(F16, C.byref(big)) == 2 * 2048 * 2112
    assert lib.esn_packed_readout_bytes(F16, C.byref(big)) == (16 * 2112 * 2 + 16) + (2048 * 64 + 512 + 16)
    assert lib.esn_predict_workspace_bytes(F16, C.byref(big), 150, 75) == 2 * 256 * 2112 * 2 + 2 * 8 * 256 * 8 * 4
    assert lib.esn_predict_workspace_bytes(F16, C.byref(sh), 150, 75) == 0       # N_res = 512: persistent kernels
    assert lib.esn_readout_solve_workspace_bytes(3, 128, 528, 8) == 3 * 8 * (128 * 528 + 8 * 528 + 2 * 128)
    assert lib.esn_readout_solve_workspace_bytes(1, 512, 104, 4) == 8 * ((104 + 4) * 512 + 4 * 512 + 2 * 104)


def test_argument_errors_are_reported(lib):
    from esn_ofdm_mimo_amd._lib import Shape, F64
    bad = Shape(0, 1, 1, 1, 1)
    assert lib.esn_tile_frames(F64, C.byref(bad)) < 0
    sh = Shape(8, 2, 2, 1, 1)
    rc = lib.esn_predict_batch(F64, C.byref(sh), None, None, None, None, None, None, None,
                               1, 1, 4, 4, 0, None, None, 0.0, 0, None, 0, 0, None, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.esn_last_error()
    rc = lib.esn_detect_count(1, 1, 1, 100, 2, 4, 1, 1, 1, 1, None, None)
    assert rc == -1 and b"power of two" in lib.esn_last_error()


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from esn_ofdm_mimo_amd import pyESN, _lib
    with pytest.raises(_lib.EsnHipError, match="no CPU fallback"):
        pyESN.ESN(2, 2, n_reservoir=10, random_state=1)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "esn_ofdm_mimo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
