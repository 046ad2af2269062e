"""CPU oracle for the baseline equaliser the reference plots the ESN against --
TEST INFRASTRUCTURE ONLY (same rules as esn_oracle.py).

float64/complex128 NumPy restatement of the north-star driver's pilot-based channel estimate and
per-subcarrier MMSE detector (``system_model_2/Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py``):
  * sparse LS pilot pattern, one Tx per subcarrier round-robin ............ :330-333
  * pilot + LS-pilot frames through the channel with the SAME noise ....... :336-356
  * (1/N) FFT, LS at the pilot subcarriers, linear inter/extrapolation ..... :358-371
  * time-domain MMSE refinement, truncation to IsiDuration taps ............ :373-378
  * per-subcarrier MMSE solve (H^H H + No/Pi I)^-1 H^H Y / sqrt(Pi) ........ :40-45, :444-448

Parity pin: the driver cannot be imported (runs at import, needs pyldpc), so this file is pinned
statistically only -- against column 3 (MMSE_uncoded) of the reference's own
``results/results_4x8_cdl_coded_uncoded/CDLB_run_01/results_ber.csv`` -- which at the same time
pins the frame recipe of ``ofdm_frames.py`` (tests/test_oracle_baseline_ber.py)."""
from __future__ import annotations

import math

import numpy as np
from scipy import interpolate, signal

from .esn_oracle import unit_qam
from .ofdm_frames import LinkConfig, random_bits

# MMSE_uncoded column of results_ber.csv (Eb/No 0,3,...,30 dB; N=128, 4x8, 16-QAM, 1000 symbols/point)
PUBLISHED_MMSE_BER = {0: 0.3196171875, 3: 0.2523603515625, 6: 0.185376953125, 9: 0.12928955078125,
                      12: 0.07861474609375, 15: 0.05450439453125, 18: 0.03449072265625,
                      21: 0.0270302734375, 24: 0.02187158203125, 27: 0.01991455078125, 30: 0.0189169921875}


def isi_magnitude(cfg: LinkConfig):
    """Exponential prior of the time-domain MMSE refinement (driver:212-214)."""
    t = cfg.cp / 9
    m = np.exp(-np.arange(cfg.cp + 1) / max(t, 1e-12))
    return m / m.sum()


def pilot_frames(cfg: LinkConfig, ebno_db, taps, rng):
    """Pilot symbol and its sparse LS companion through the channel with one shared noise draw.
    Returns dict(bits, X_p, X_LS, x_cp (pre-PA teacher), y_cp, y_ls_cp)."""
    const = unit_qam(cfg.m)
    n, m = cfg.n_sub, cfg.m
    bits = random_bits(cfg, rng)
    idx = (bits.reshape(n, m, cfg.n_t) * (1 << np.arange(m))[None, :, None]).sum(axis=1)
    x_p = const[idx]
    x_ls = np.zeros_like(x_p)
    for tx in range(cfg.n_t):
        x_ls[tx::cfg.n_t, tx] = x_p[tx::cfg.n_t, tx]
    sp, a = math.sqrt(cfg.p_i(ebno_db)), cfg.a_clip(ebno_db)

    def tx_chain(xf):
        xt = n * np.fft.ifft(xf, axis=0)
        xc = np.concatenate([xt[-cfg.cp:], xt], axis=0) * sp
        return xc, xc / np.sqrt(1 + (np.abs(xc) / a) ** 2)

    x_cp, x_pa = tx_chain(x_p)
    _, x_ls_pa = tx_chain(x_ls)
    t = n + cfg.cp
    y = np.zeros((t, cfg.n_r), dtype=np.complex128)
    y_ls = np.zeros_like(y)
    for nr in range(cfg.n_r):
        for nt in range(cfg.n_t):
            y[:, nr] += signal.lfilter(taps[nr, nt], np.array([1]), x_pa[:, nt])
            y_ls[:, nr] += signal.lfilter(taps[nr, nt], np.array([1]), x_ls_pa[:, nt])
        noise = math.sqrt(t * cfg.no / 2) * (rng.randn(t) + 1j * rng.randn(t))
        y[:, nr] += noise
        y_ls[:, nr] += noise
    return dict(bits=bits, X_p=x_p, X_LS=x_ls, x_cp=x_cp, y_cp=y, y_ls_cp=y_ls)


def estimate_channel(cfg: LinkConfig, ebno_db, x_ls, y_ls_cp, ls_only=False):
    """H_MMSE [N, n_r, n_t]: LS at every n_t-th subcarrier, linear interpolation with
    extrapolation, IFFT, truncate to isi taps, diagonal MMSE shrinkage, FFT (driver:358-382).
    ls_only: the interpolated LS estimate H_LS itself (OFDM_MIMO_2-2_NBF_LDPC.py:321-333, fed to LS-ZF)."""
    n, p_i = cfg.n_sub, cfg.p_i(ebno_db)
    y_ls = (1.0 / n) * np.fft.fft(y_ls_cp[cfg.cp:], axis=0)
    r_h = np.diag(isi_magnitude(cfg)[:cfg.isi])
    scaler = (cfg.no / p_i) / (n / 2)
    h = np.zeros((n, cfg.n_r, cfg.n_t), dtype=np.complex128)
    for nr in range(cfg.n_r):
        for tx in range(cfg.n_t):
            sc = np.arange(tx, n, cfg.n_t)
            hls = y_ls[sc, nr] / (x_ls[sc, tx] * math.sqrt(p_i) + 1e-12)
            full = interpolate.interp1d(sc, hls, kind="linear", bounds_error=False,
                                        fill_value="extrapolate")(np.arange(n))
            if ls_only:
                h[:, nr, tx] = full
                continue
            c_ls = np.fft.ifft(full)[:cfg.isi]
            c_mmse = np.linalg.solve(scaler * np.linalg.inv(r_h) + np.eye(cfg.isi), c_ls)
            h[:, nr, tx] = np.fft.fft(np.r_[c_mmse, np.zeros(n - cfg.isi)])
    return h


def equalize_mmse(y_k, h_k, power_scale, noise_over_power):
    """(H^H H + nop I)^-1 H^H y / power_scale for one subcarrier (driver:40-45)."""
    hh = h_k.conj().T
    g = hh @ h_k + noise_over_power * np.eye(h_k.shape[1], dtype=h_k.dtype)
    return np.linalg.solve(g, hh @ y_k) / power_scale


def equalize_zf(y_k, h_k, power_scale):
    """(H^H H + 1e-12 I)^-1 H^H y / power_scale for one subcarrier (driver:34-39;
    OFDM_MIMO_2-2_NBF_LDPC.py:41-47 -- Perfect-ZF with the true H, LS-ZF with the estimate, :450-460)."""
    hh = h_k.conj().T
    g = hh @ h_k + 1e-12 * np.eye(h_k.shape[1], dtype=h_k.dtype)
    return np.linalg.solve(g, hh @ y_k) / power_scale


def taps_to_freq(cfg: LinkConfig, taps):
    """True channel H [N, n_r, n_t] = FFT_N of the zero-padded taps per link
    (Perfect-CSI ZF, OFDM_MIMO_2-2_NBF_LDPC.py:281-285)."""
    return np.transpose(np.fft.fft(taps, cfg.n_sub, axis=2), (2, 0, 1))


def linear_detect(cfg: LinkConfig, ebno_db, h, y_cp, reg=None):
    """X_hat [N, n_t]: per-subcarrier MMSE (reg = No/Pi, the default) or ZF (reg = 0 -> 1e-12)
    of the received frame (driver:444-448; NBF driver:450-460)."""
    n, p_i = cfg.n_sub, cfg.p_i(ebno_db)
    y = (1.0 / n) * np.fft.fft(y_cp[cfg.cp:], axis=0)              # [N, n_r]
    reg = cfg.no / p_i if reg is None else reg
    out = np.zeros((n, cfg.n_t), dtype=np.complex128)
    for k in range(n):
        out[k] = equalize_zf(y[k], h[k], math.sqrt(p_i)) if reg == 0 else \
            equalize_mmse(y[k], h[k], math.sqrt(p_i), reg)
    return out


def mmse_detect(cfg: LinkConfig, ebno_db, h, y_cp):
    """X_hat [N, n_t] = (H^H H + No/Pi I)^-1 H^H Y / sqrt(Pi) per subcarrier (driver:40-45,444-448)."""
    return linear_detect(cfg, ebno_db, h, y_cp)
