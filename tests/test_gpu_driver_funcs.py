"""HIP kernels of the detector tail, the LLR leg, the linear baselines and the tap generator against the
REFERENCE'S OWN helper functions -- not against the oracle.

tests/golden/driver_funcs.npz holds seeded inputs and the outputs of the drivers' top-level functions
(compiled out of their `ast` in the build container, tests/golden/make_golden.py::case_driver_funcs).  The
same inputs go through the C ABI here:
    esn_detect_count        vs hard_bits_from_syms / reconstruct_esn_outputs_generic / (1/N) FFT
    esn_qam_llr             vs est_sigma2_from_decision + qam_llrs_maxlog (the driver's per-frame use)
    esn_mmse_detect_count   vs equalize_mmse          esn_zf_detect_count   vs equalize_zf
    esn_gen_taps (kind 0)   vs build_cdlb_mimo_taps, fed the standard normals the reference consumed
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g(golden):
    return golden("driver_funcs")


@pytest.fixture(scope="module")
def env():
    import torch
    from esn_ofdm_mimo_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")

    def to_dev(a, dtype=None):
        a = np.ascontiguousarray(a)
        t = torch.as_tensor(a.copy(), device=dev)
        return t.to(dtype) if dtype is not None else t
    return torch, _lib, lib, dev, to_dev


def _bits_layout(bits_nm_nt):
    """[N*m, n_t] int -> uint8 [1][N*m][n_t] (TxBits layout of the C ABI)."""
    return np.ascontiguousarray(bits_nm_nt.astype(np.uint8))[None]


@pytest.mark.parametrize("tag", ("v2", "nbf", "siso"))
@pytest.mark.parametrize("m", (2, 4))
def test_detect_count_hard_decisions_equal_reference(g, env, tag, m):
    """Symbols X -> time domain y = N ifft(X) sqrt(Pi) -> esn_detect_count: X_hat returns X, and the kernel's
    bits equal hard_bits_from_syms(X) -- zero errors against them, all errors against their complement."""
    torch, _lib, lib, dev, to_dev = env
    x = g[f"{tag}_hard{m}_x"]                       # [N, n_t] complex
    want = g[f"{tag}_hard{m}_bits"]
    if tag == "siso":
        x, want = x[:, :1], want[:, None]
    n, n_t = x.shape
    p_i = 3.3e-4
    y = (n * np.fft.ifft(x, axis=0) * np.sqrt(p_i))[None]                        # [1, N, n_t] complex
    Y = to_dev(np.ascontiguousarray(y).view(np.float64).reshape(1, n, 2 * n_t))
    pi_d = to_dev(np.array([p_i]))
    for bits, expect in ((want, 0), (1 - want, want.size)):
        err = torch.zeros(1, dtype=torch.int64, device=dev)
        nb = torch.zeros(1, dtype=torch.int64, device=dev)
        xh = torch.empty((1, n, 2 * n_t), dtype=torch.float64, device=dev)
        _lib.check(lib.esn_detect_count(Y.data_ptr(), 1, 1, n, n_t, m, pi_d.data_ptr(),
                                        to_dev(_bits_layout(bits)).data_ptr(), err.data_ptr(), nb.data_ptr(),
                                        xh.data_ptr(), None), "esn_detect_count")
        torch.cuda.synchronize()
        assert int(err[0]) == expect and int(nb[0]) == want.size
    got = xh.cpu().numpy().reshape(n, n_t, 2)
    np.testing.assert_allclose(got[..., 0] + 1j * got[..., 1], x, rtol=0, atol=1e-12 * np.abs(x).max())


@pytest.mark.parametrize("tag", ("v2", "nbf"))
def test_detect_count_reconstruction_equals_reference(g, env, tag):
    """reconstruct_esn_outputs_generic with the common delay every driver uses: column pairs -> complex
    sequences; the kernel's X_hat is their (1/N) FFT / sqrt(Pi)."""
    torch, _lib, lib, dev, to_dev = env
    y = g[f"{tag}_recon_y"]                         # [N + 9, 2 n_t]
    seqs = g[f"{tag}_recon_common"][:, :16]         # [n_t, N] (the block-fading variant slices one row more)
    n_t, n = seqs.shape
    p_i = 2.0e-5
    Y = to_dev(np.ascontiguousarray(y[:n])[None])   # delay - delay_min = 0: rows 0..N-1
    err = torch.zeros(1, dtype=torch.int64, device=dev)
    nb = torch.zeros(1, dtype=torch.int64, device=dev)
    xh = torch.empty((1, n, 2 * n_t), dtype=torch.float64, device=dev)
    bits = torch.zeros((1, n * 4, n_t), dtype=torch.uint8, device=dev)
    _lib.check(lib.esn_detect_count(Y.data_ptr(), 1, 1, n, n_t, 4, to_dev(np.array([p_i])).data_ptr(),
                                    bits.data_ptr(), err.data_ptr(), nb.data_ptr(), xh.data_ptr(), None), "detect")
    got = xh.cpu().numpy().reshape(n, n_t, 2)
    want = np.fft.fft(seqs, axis=1).T / n / np.sqrt(p_i)
    np.testing.assert_allclose(got[..., 0] + 1j * got[..., 1], want, rtol=0, atol=1e-12 * np.abs(want).max())


@pytest.mark.parametrize("tag", ("v2", "nbf"))
@pytest.mark.parametrize("m", (2, 4))
def test_qam_llr_equals_reference_frame(g, env, tag, m):
    """One frame the way the drivers use the functions (v2 :459-463): sigma^2 = mean over tx of
    est_sigma2_from_decision, LLRs of every column scaled by it."""
    torch, _lib, lib, dev, to_dev = env
    x = g[f"{tag}_hard{m}_x"]                       # [N, n_t]
    n, n_t = x.shape
    want = g[f"{tag}_frame_llr{m}"]                 # (N, m, n_t)
    X = to_dev(np.ascontiguousarray(x).view(np.float64).reshape(1, n, 2 * n_t))
    llr = torch.empty((1, n_t, n * m), dtype=torch.float64, device=dev)
    s2 = torch.empty(1, dtype=torch.float64, device=dev)
    _lib.check(lib.esn_qam_llr(1, n, n_t, m, X.data_ptr(), llr.data_ptr(), s2.data_ptr(), None), "esn_qam_llr")
    torch.cuda.synchronize()
    assert float(s2[0]) == pytest.approx(float(g[f"{tag}_frame_sigma2_{m}"]), rel=1e-12)
    got = llr.cpu().numpy().reshape(n_t, n, m).transpose(1, 2, 0)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10)
    # single column = the function's own sigma^2
    x1 = np.ascontiguousarray(x[:, :1])
    X1 = to_dev(x1.view(np.float64).reshape(1, n, 2))
    llr1 = torch.empty((1, 1, n * m), dtype=torch.float64, device=dev)
    _lib.check(lib.esn_qam_llr(1, n, 1, m, X1.data_ptr(), llr1.data_ptr(), s2.data_ptr(), None), "esn_qam_llr")
    torch.cuda.synchronize()
    assert float(s2[0]) == pytest.approx(float(g[f"{tag}_sigma2_{m}"]), rel=1e-12)
    np.testing.assert_allclose(llr1.cpu().numpy().reshape(n, m), g[f"{tag}_llr{m}"], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("tag", ("v2", "nbf"))
@pytest.mark.parametrize("shape", ("8x4", "2x2"))
@pytest.mark.parametrize("kind", ("mmse", "zf"))
def test_linear_equalisers_equal_reference(g, env, tag, shape, kind):
    """equalize_mmse / equalize_zf per subcarrier: the fixture's (Y_k, H_k) become one frame of N = 16
    subcarriers (time domain y = N ifft(Y), no CP) and one channel H [N][n_r][n_t]."""
    torch, _lib, lib, dev, to_dev = env
    h, yk = g[f"{tag}_eq{shape}_h"], g[f"{tag}_eq{shape}_y"]        # [N, n_r, n_t], [N, n_r]
    want = g[f"{tag}_eq{shape}_{kind}"]                              # [N, n_t]
    n, n_r, n_t = h.shape
    power_scale, nop = 0.0123, 0.004
    p_i = power_scale ** 2
    y_t = (n * np.fft.ifft(yk, axis=0))[None]                        # [1, N, n_r]
    Yd = to_dev(np.ascontiguousarray(y_t).view(np.float64))
    Hd = to_dev(np.ascontiguousarray(h[None]).view(np.float64))
    err = torch.zeros(1, dtype=torch.int64, device=dev)
    nb = torch.zeros(1, dtype=torch.int64, device=dev)
    xh = torch.empty((1, n, n_t, 2), dtype=torch.float64, device=dev)
    bits = torch.zeros((1, n * 4, n_t), dtype=torch.uint8, device=dev)
    pi_d = to_dev(np.array([p_i]))
    if kind == "mmse":
        rc = lib.esn_mmse_detect_count(1, 1, n, 0, n_t, n_r, 4, pi_d.data_ptr(), C.c_double(nop * p_i), Hd.data_ptr(),
                                       Yd.data_ptr(), bits.data_ptr(), err.data_ptr(), nb.data_ptr(), xh.data_ptr(), None)
    else:
        rc = lib.esn_zf_detect_count(1, 1, n, 0, n_t, n_r, 4, pi_d.data_ptr(), Hd.data_ptr(), Yd.data_ptr(),
                                     bits.data_ptr(), err.data_ptr(), nb.data_ptr(), xh.data_ptr(), None)
    _lib.check(rc, kind)
    torch.cuda.synchronize()
    got = xh.cpu().numpy()
    got = got[0, :, :, 0] + 1j * got[0, :, :, 1]
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9 * np.abs(want).max())


@pytest.mark.parametrize("key,n_r,n_t,ds", [("v2_taps_1247", 8, 4, 300.0), ("v2_taps_1340", 8, 4, 300.0),
                                            ("v2_taps_2x2_ds1000", 2, 2, 1000.0)])
def test_tdlb_taps_equal_reference(g, env, key, n_r, n_t, ds):
    """build_cdlb_mimo_taps: the device generator fed the standard normals the reference's Generator produced
    (per link, per path: real then imaginary) returns the reference's impulse responses."""
    torch, _lib, lib, dev, to_dev = env
    want = g[key]                                   # [n_r, n_t, isi] complex
    normals = g[key + "_normals"]                   # [n_r, n_t, 23, 2]
    gains = to_dev(normals)
    out = torch.empty((1, n_r, n_t, 8), dtype=torch.complex128, device=dev)
    _lib.check(lib.esn_gen_taps(0, 1, n_r, n_t, 8, C.c_double(2 * 1.024e6), C.c_double(ds), gains.data_ptr(), 0, 0,
                                out.data_ptr(), None), "esn_gen_taps")
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy()[0], want, rtol=1e-12, atol=1e-14)


def test_perfect_csi_channel_is_fft_of_taps(g, env):
    """esn_taps_to_freq = H_true of the block-fading drivers: FFT_N of the zero-padded impulse response."""
    torch, _lib, lib, dev, to_dev = env
    taps = g["v2_taps_1247"]                        # [8, 4, 8]
    n = 128
    H = torch.empty((1, n, 8, 4), dtype=torch.complex128, device=dev)
    _lib.check(lib.esn_taps_to_freq(1, n, 4, 8, 8, to_dev(taps[None]).data_ptr(), H.data_ptr(), None), "taps_to_freq")
    torch.cuda.synchronize()
    want = np.transpose(np.fft.fft(taps, n, axis=2), (2, 0, 1))
    np.testing.assert_allclose(H.cpu().numpy()[0], want, rtol=0, atol=1e-12)
