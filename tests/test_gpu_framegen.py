"""HIP frame generator against oracle/ofdm_frames.py.

Deterministic leg: bits, noise and tap gains are supplied, so the kernels must reproduce the
oracle's transmitter / channel arithmetic to float64 round-off (1e-12 of max|y|).
Statistical leg: the Philox streams -- bit balance, unit-variance noise, unit-energy TDL-B taps,
and stream independence from the launch shape (a block generated alone == generated in a batch).
The oracle's recipe is itself "parity unpinned at the sample level" (driver scripts are not
importable); what is pinned here is that the HIP generator equals the oracle restatement."""
import math

import numpy as np
import pytest

from oracle import esn_oracle as eo
from oracle.ofdm_frames import LinkConfig, TDLB_NORM_DELAYS, TDLB_POW_DB, modulate

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def src():
    from esn_ofdm_mimo_amd.montecarlo import FrameSource, LinkParams
    return FrameSource, LinkParams


def _oracle_taps_from_gains(cfg, g):
    """oracle.tdlb_impulse with the path gains given (g: complex [n_paths] standard normal pairs)."""
    p = 10.0 ** (TDLB_POW_DB / 10.0)
    p = p / p.sum()
    d = TDLB_NORM_DELAYS * cfg.ds_ns * 1e-9 * cfg.fs
    h = np.zeros(cfg.isi, dtype=np.complex128)
    for k in range(len(p)):
        i0 = int(np.floor(d[k])); frac = d[k] - i0
        gp = g[k] / np.sqrt(2.0) * np.sqrt(p[k])
        if 0 <= i0 < cfg.isi:
            h[i0] += gp * (1.0 - frac)
        if 0 <= i0 + 1 < cfg.isi:
            h[i0 + 1] += gp * frac
    return h / np.sqrt(np.sum(np.abs(h) ** 2))


@pytest.mark.parametrize("n_t,n_r,n_sub,m", [(4, 8, 128, 4), (2, 2, 512, 4), (1, 1, 64, 2)])
def test_generator_matches_oracle_with_supplied_randomness(src, n_t, n_r, n_sub, m):
    import torch
    from scipy import signal
    FrameSource, LinkParams = src
    cfg = LinkConfig(n_t=n_t, n_r=n_r, n_sub=n_sub, m=m)
    prm = LinkParams(n_t=n_t, n_r=n_r, n_sub=n_sub, m=m)
    fs = FrameSource(prm, seed=3)
    rs = np.random.RandomState(n_sub + n_t)
    G, F, ebno = 2, 3, 9.0
    gains = rs.randn(G, n_r, n_t, 23) + 1j * rs.randn(G, n_r, n_t, 23)
    taps = fs.taps(G, 0, 0, gains=torch.as_tensor(gains, device=fs.device)).cpu().numpy()
    for b in range(G):
        for rx in range(n_r):
            for tx in range(n_t):
                np.testing.assert_allclose(taps[b, rx, tx], _oracle_taps_from_gains(cfg, gains[b, rx, tx]),
                                           rtol=1e-12, atol=1e-14)
    T = n_sub + cfg.cp
    bits_in = (rs.rand(G * F, n_sub * m, n_t) > 0.5).astype(np.uint8)
    noise = rs.randn(G * F, T, n_r) + 1j * rs.randn(G * F, T, n_r)
    bits, x_cp, y_cp = fs.frames(torch.as_tensor(taps, device=fs.device), F, ebno, 0, 0, 1, want_x=True,
                                 bits_in=torch.as_tensor(bits_in, device=fs.device),
                                 noise_in=torch.as_tensor(noise, device=fs.device))
    np.testing.assert_array_equal(bits.cpu().numpy(), bits_in)
    x_cp, y_cp = x_cp.cpu().numpy(), y_cp.cpu().numpy()
    for f in range(G * F):
        _, xo, x_pa = modulate(bits_in[f].astype(np.int32), cfg, ebno)
        np.testing.assert_allclose(x_cp[f], xo, rtol=0, atol=1e-12 * np.abs(xo).max())
        want = np.zeros((T, n_r), dtype=np.complex128)
        for rx in range(n_r):
            for tx in range(n_t):
                want[:, rx] += signal.lfilter(taps[f // F, rx, tx], np.array([1]), x_pa[:, tx])
            want[:, rx] += math.sqrt(T * cfg.no / 2) * noise[f, :, rx]
        np.testing.assert_allclose(y_cp[f], want, rtol=0, atol=1e-12 * np.abs(want).max())


def test_generator_streams_are_shape_independent_and_well_distributed(src):
    FrameSource, LinkParams = src
    prm = LinkParams()
    fs = FrameSource(prm, seed=11)
    F = 6
    full = fs.blocks_fast(12.0, 2, 0, 8, F)
    one = fs.blocks(12.0, 2, [5], F)                 # block 5 alone == block 5 inside the batch
    np.testing.assert_array_equal(one["pilot_y"].cpu().numpy(), full["pilot_y"][5:6].cpu().numpy())
    np.testing.assert_array_equal(one["data_y"].cpu().numpy(), full["data_y"][5 * F:6 * F].cpu().numpy())
    np.testing.assert_array_equal(one["data_bits"].cpu().numpy(), full["data_bits"][5 * F:6 * F].cpu().numpy())
    pair = fs.blocks(12.0, 2, [2, 3, 6], F)          # non-contiguous subset, rank-style
    np.testing.assert_array_equal(pair["data_y"][2 * F:].cpu().numpy(), full["data_y"][6 * F:7 * F].cpu().numpy())
    other = fs.blocks_fast(12.0, 3, 0, 8, F)         # another SNR index: different streams
    assert np.abs(other["data_y"].cpu().numpy() - full["data_y"].cpu().numpy()).max() > 0

    big = fs.blocks_fast(12.0, 0, 0, 64, 16)
    bits = big["data_bits"].cpu().numpy().astype(float)
    assert abs(bits.mean() - 0.5) < 4 * 0.5 / np.sqrt(bits.size)
    taps = big["taps"].cpu().numpy()
    np.testing.assert_allclose((np.abs(taps) ** 2).sum(-1), 1.0, rtol=1e-12)        # unit energy per link
    # received power: sum over tx of |h|^2 * E|x_pa|^2 + noise; check the AWGN floor at very low SNR
    lo = FrameSource(LinkParams(), seed=5)
    z = lo.blocks_fast(-60.0, 0, 0, 16, 8)["data_y"].cpu().numpy()                  # signal ~1e-6 of noise
    T = prm.t_frame
    var = T * prm.no / 2
    assert abs(z.real.var() / var - 1) < 0.02 and abs(z.imag.var() / var - 1) < 0.02
    assert abs(z.real.mean()) < 5 * np.sqrt(var / z.real.size)
    k = ((z.real / np.sqrt(var)) ** 4).mean()
    assert abs(k - 3.0) < 0.1                                                        # Gaussian kurtosis


def test_generated_frames_drive_the_detector_to_the_published_ber_level(src):
    """End to end on generated frames: uncoded ESN BER at 12 dB sits in the band of the reference's
    results_ber.csv (0.2445 at N_res=300) -- the statistical pin of the recipe."""
    from esn_ofdm_mimo_amd.montecarlo import DetectorSweep, LinkParams
    sw = DetectorSweep(LinkParams(), n_reservoir=300, noise=0.001, seed=1, precision="f16", fit_precision="f32")
    ber, counts = sw.run([12.0], blocks_per_snr=48, chunk_blocks=48)
    assert counts[0, 1] == 48 * sw.p.coherence_symbols * 128 * 4 * 4
    assert 0.19 < ber[0] < 0.30, ber
