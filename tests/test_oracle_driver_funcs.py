"""The driver-side pieces of the oracle against the reference's OWN helper functions.

tests/golden/driver_funcs.npz was produced in the build container by compiling only the top-level
`def`s (and the two TDL-B tables) of three reference drivers out of their `ast` -- the simulation bodies
never run, pyldpc is never needed -- and calling them on seeded inputs (tests/golden/make_golden.py
::case_driver_funcs).  Every restated piece the detector tail, the LLR leg, the linear baselines and the
tap generator rest on is compared with those outputs here: constellation, bit labels (natural binary,
LSB first), hard decisions, ESN output reconstruction, MMSE / ZF solves, max-log LLRs, decision-directed
sigma^2, logistic calibration, TDL-B impulse responses.  CPU only."""
import math

import numpy as np
import pytest

from oracle import baselines, esn_oracle as eo, ldpc_oracle as lo
from oracle.ofdm_frames import LinkConfig, TDLB_NORM_DELAYS, TDLB_POW_DB, tdlb_mimo_taps

TAGS = ("v2", "nbf", "siso")


@pytest.fixture(scope="module")
def g(golden):
    return golden("driver_funcs")


@pytest.mark.parametrize("tag", TAGS)
def test_constellation_and_bit_labels(g, tag):
    for m in (2, 4, 6):
        np.testing.assert_allclose(eo.unit_qam(m), g[f"{tag}_qam{m}"], rtol=0, atol=1e-15)
        np.testing.assert_array_equal(eo.bit_labels_lsb_first(m), g[f"{tag}_labels{m}"])


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("m", (2, 4))
def test_hard_bits(g, tag, m):
    x = g[f"{tag}_hard{m}_x"]
    want = g[f"{tag}_hard{m}_bits"]
    if tag == "siso":
        got = eo.hard_bits(x[:, :1], eo.unit_qam(m), m)[:, 0]
    else:
        got = eo.hard_bits(x, eo.unit_qam(m), m)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("m", (2, 4))
def test_sigma2_and_maxlog_llr(g, tag, m):
    z = g[f"{tag}_hard{m}_x"][:, 0]
    s2 = lo.sigma2_from_decisions(z, m)
    assert s2 == pytest.approx(float(g[f"{tag}_sigma2_{m}"]), rel=1e-14)
    np.testing.assert_allclose(lo.qam_llrs_maxlog(z, m, s2), g[f"{tag}_llr{m}"], rtol=1e-13, atol=1e-13)
    # sigma^2 below the floor: the max(sigma2, 1e-12) guard
    np.testing.assert_allclose(lo.qam_llrs_maxlog(z[:4], m, 0.0), g[f"{tag}_llr{m}_tiny_sigma"], rtol=1e-13)


@pytest.mark.parametrize("tag", ("v2", "nbf"))
@pytest.mark.parametrize("shape", ("8x4", "2x2"))
def test_mmse_and_zf_solves(g, tag, shape):
    h, y = g[f"{tag}_eq{shape}_h"], g[f"{tag}_eq{shape}_y"]
    for k in range(h.shape[0]):
        np.testing.assert_allclose(baselines.equalize_mmse(y[k], h[k], 0.0123, 0.004),
                                   g[f"{tag}_eq{shape}_mmse"][k], rtol=1e-12)
        np.testing.assert_allclose(baselines.equalize_zf(y[k], h[k], 0.0123),
                                   g[f"{tag}_eq{shape}_zf"][k], rtol=1e-12)


def test_siso_scalar_equalisers(g):
    """The SISO driver's scalar forms (Demo_SISO_...:76-95) are the 1x1 case of the matrix forms
    up to its extra 1e-12 in the MMSE denominator."""
    y, h = g["siso_eq_y"], g["siso_eq_h"]
    for k in range(len(y)):
        hk, yk = h[k].reshape(1, 1), y[k].reshape(1)
        np.testing.assert_allclose(baselines.equalize_zf(yk, hk, 0.37)[0], g["siso_eq_zf"][k], rtol=1e-12)
        np.testing.assert_allclose(baselines.equalize_zf(yk, hk, 0.37)[0], g["siso_eq_ls"][k], rtol=1e-12)
        np.testing.assert_allclose(baselines.equalize_mmse(yk, hk, 0.37, 0.02 + 1e-12)[0], g["siso_eq_mmse"][k],
                                   rtol=1e-12)


@pytest.mark.parametrize("tag", ("v2", "nbf"))
def test_reconstruct_esn_outputs(g, tag):
    y = g[f"{tag}_recon_y"]
    n, n_t = 16, 3
    for name in ("common", "ragged"):
        delay, dmin = g[f"{tag}_recon_{name}_delay"], int(g[f"{tag}_recon_{name}_dmin"])
        want = g[f"{tag}_recon_{name}"]
        # the block-fading drivers slice N+1 rows (OFDM_MIMO_2-2_NBF_LDPC.py:56-64); the FFT that follows
        # (np.fft.fft(x) of N+1 points) is only ever fed N rows because the prediction has exactly N
        rows = want.shape[1]
        got = np.array(eo.outputs_to_time_signals(y, delay, dmin, rows, n_t))
        np.testing.assert_array_equal(got, want)
    want = g[f"{tag}_recon_short"]
    got = np.array(eo.outputs_to_time_signals(y[:n], np.full(2 * n_t, 3), 3, n, n_t))
    assert want.shape == (n_t, n)          # N rows even where N+1 are asked for
    np.testing.assert_array_equal(got, want)


def test_logistic_calibration(g):
    x, y = g["v2_logreg_x"], g["v2_logreg_y"]
    np.testing.assert_allclose(lo.fit_logreg_1d(x, y, maxiter=400, lr=0.1, l2=1e-3), g["v2_logreg_call"], rtol=1e-12)
    np.testing.assert_allclose(lo.fit_logreg_1d(x, y, lr=0.15), g["v2_logreg_default"], rtol=1e-12)


def test_tdlb_table_and_taps(g):
    np.testing.assert_array_equal(TDLB_NORM_DELAYS, g["v2_tdlb_delays"])
    np.testing.assert_array_equal(TDLB_POW_DB, g["v2_tdlb_pow_db"])
    cfg = LinkConfig()
    for seed in (1234 + 12 + 1, 1234 + 30 + 76):
        want = g[f"v2_taps_{seed}"]
        np.testing.assert_allclose(tdlb_mimo_taps(cfg, seed), want, rtol=1e-14, atol=1e-16)
        # unit energy per link (driver:162-164)
        np.testing.assert_allclose((np.abs(want) ** 2).sum(axis=2), 1.0, rtol=1e-12)
    cfg2 = LinkConfig(n_t=2, n_r=2, ds_ns=1000.0)
    np.testing.assert_allclose(tdlb_mimo_taps(cfg2, 5), g["v2_taps_2x2_ds1000"], rtol=1e-14, atol=1e-16)


def test_encoder_call(g):
    """ldpc_encode_bits (driver:90-93) is G u mod 2, dense or sparse."""
    want = (g["v2_enc_g"] @ g["v2_enc_u"]) % 2
    np.testing.assert_array_equal(g["v2_enc_dense"], want)
    np.testing.assert_array_equal(g["v2_enc_sparse"], want)
    # the restated systematic encoder is the same product with G = [I ; P]
    p = g["v2_enc_g"][12:].astype(np.uint8)
    gen = np.concatenate([np.eye(12, dtype=np.int64), p.astype(np.int64)])
    np.testing.assert_array_equal(lo.encode(p, g["v2_enc_u"].astype(np.uint8)), (gen @ g["v2_enc_u"]) % 2)


def test_detector_tail_on_reference_predictions(golden):
    """a10-a12 end to end with reference-made pieces only: the reference's predictions (c4.npz, made by
    libs/pyESN.py) through the restated tail give the same bits as through the reference's
    reconstruct + hard-decision functions (recomputed here from the pinned pieces: with a common delay
    the reconstruction is a column pairing, pinned above; the hard bits are pinned above)."""
    g4 = golden("c4")
    cfg = LinkConfig()
    const = eo.unit_qam(cfg.m)
    labels = eo.bit_labels_lsb_first(cfg.m)
    for pred in g4["data_pred"][:4]:
        seqs = eo.outputs_to_time_signals(pred, g4["delay"], int(g4["d_min"]), cfg.n_sub, cfg.n_t)
        x_hat = eo.time_to_freq(seqs, cfg.n_sub, cfg.p_i(float(g4["ebno_db"])))
        rx = eo.hard_bits(x_hat, const, cfg.m)
        # literal per-symbol loop of driver:95-103
        want = np.zeros_like(rx)
        for ii in range(cfg.n_sub):
            for tx in range(cfg.n_t):
                idx = int(np.argmin(np.abs(const - x_hat[ii, tx])))
                want[cfg.m * ii:cfg.m * (ii + 1), tx] = labels[idx]
        np.testing.assert_array_equal(rx, want)
        # (1/N) FFT / sqrt(Pi) (driver:439-441)
        z = pred[:cfg.n_sub, 0] + 1j * pred[:cfg.n_sub, 1]
        np.testing.assert_allclose(x_hat[:, 0], np.fft.fft(z) / cfg.n_sub / math.sqrt(cfg.p_i(12.0)), rtol=1e-13)
