"""CPU oracle for the OFDM/MIMO frame recipe around the ESN detector -- TEST
INFRASTRUCTURE ONLY (same rules as ``esn_oracle.py``: imported by tests,
smoke() and bench.py's cpu_baseline leg, never by the product path).

float64/complex128 NumPy restatement of the north-star driver's transmitter,
channel and noise (``system_model_2/Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py``):
  * TDL-B tap table (3GPP TR 38.901 Table 7.7.2-2) ........ :127-137
  * per-link impulse response from the table .............. :139-165
  * independent links, one Generator per redraw ........... :167-177
  * system constants (W, No, Pi, var_x, A_Clip, scaling) .. :182-238, :285-288
  * bits -> QAM -> N*ifft -> CP -> sqrt(Pi) -> PA ......... :323-345 (pilot), :397-419 (data)
  * per-link FIR (lfilter, zero state) + AWGN ............. :348-356, :421-427
and the block-fading exponential-PDP taps of
``system_model_2/OFDM_MIMO_2-2_NBF_LDPC.py:162-164,272-279``.

Parity pin: the constellation is checked against the reference's importable
``HelpFunc.UnitQamConstellation``; the rest of the recipe lives in driver
scripts that cannot be imported (they run a whole simulation at import and
need pyldpc), so it is "parity unpinned" at the sample level and pinned
STATISTICALLY against the reference's own published curve: frames from this
recipe through the restated LS/MMSE baseline (oracle/baselines.py) reproduce
column MMSE_uncoded of results/.../CDLB_run_01/results_ber.csv within 0-5 % at
every Eb/No from 0 to 30 dB (tests/test_oracle_baseline_ber.py) -- a wrong power
scaling, PA, noise variance or tap recipe moves that curve by dBs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
from scipy import signal

from .esn_oracle import unit_qam

TDLB_NORM_DELAYS = np.array([
    0.0000, 0.1072, 0.2155, 0.2095, 0.2870, 0.2986, 0.3752, 0.5055, 0.3681,
    0.3697, 0.5700, 0.5283, 1.1021, 1.2756, 1.5474, 1.7842, 2.0169, 2.8294,
    3.0219, 3.6187, 4.1067, 4.2790, 4.7834])
TDLB_POW_DB = np.array([
    0.0, -2.2, -4.0, -3.2, -9.8, -1.2, -3.4, -5.2, -7.6, -3.0, -8.9, -9.0,
    -4.8, -5.7, -7.5, -1.9, -7.6, -12.2, -9.8, -11.4, -14.9, -9.2, -11.3])


@dataclass
class LinkConfig:
    """Constants of one driver configuration (defaults = north-star 4x8 driver)."""
    n_t: int = 4
    n_r: int = 8
    n_sub: int = 128            # N
    m: int = 4                  # bits / QAM symbol
    isi: int = 8                # IsiDuration
    fs: float = 2 * 1.024e6     # W
    no: float = 1e-5
    clip_db: float = 3.0
    ds_ns: float = 300.0
    input_scaler: float = 0.005
    teacher_scale: float = 5e-7
    min_delay: int = 0
    f_d: float = 100.0

    @property
    def cp(self):
        return self.isi - 1

    @property
    def max_delay(self):
        return int(math.ceil(self.isi / 2) + 2)

    @property
    def delay(self):
        return (self.min_delay + self.max_delay) // 2

    @property
    def coherence_symbols(self):
        """L (driver:200-203)."""
        t_sym = (self.n_sub + self.isi - 1) / self.fs
        return max(1, math.floor((0.5 / max(self.f_d, 1e-9)) / t_sym))

    def p_i(self, ebno_db):
        return (10 ** (ebno_db / 10)) * self.no

    def var_x(self, ebno_db):
        return float(np.float_power(10, ebno_db / 10) * self.no * self.n_sub)

    def a_clip(self, ebno_db):
        return math.sqrt(self.var_x(ebno_db)) * float(np.float_power(10, self.clip_db / 20))

    def input_scaling(self, ebno_db):
        return self.input_scaler / math.sqrt(self.var_x(ebno_db))


def tdlb_impulse(isi, fs, ds_ns, rng):
    """One link's discrete impulse response (driver:139-165)."""
    p = 10.0 ** (TDLB_POW_DB / 10.0)
    p = p / p.sum()
    d_samp = TDLB_NORM_DELAYS * ds_ns * 1e-9 * fs
    h = np.zeros(isi, dtype=np.complex128)
    for k in range(len(p)):
        i0 = int(np.floor(d_samp[k]))
        frac = d_samp[k] - i0
        g = (rng.standard_normal() + 1j * rng.standard_normal()) / np.sqrt(2.0) * np.sqrt(p[k])
        if 0 <= i0 < isi:
            h[i0] += g * (1.0 - frac)
        if 0 <= i0 + 1 < isi:
            h[i0 + 1] += g * frac
    e = np.sum(np.abs(h) ** 2)
    if e > 0:
        h = h / np.sqrt(e)
    return h


def tdlb_mimo_taps(cfg: LinkConfig, seed):
    """taps[n_r, n_t, isi] from one Generator, rx-major draw order (driver:167-177)."""
    rng = np.random.default_rng(seed)
    c = np.zeros((cfg.n_r, cfg.n_t, cfg.isi), dtype=np.complex128)
    for nr in range(cfg.n_r):
        for nt in range(cfg.n_t):
            c[nr, nt] = tdlb_impulse(cfg.isi, cfg.fs, cfg.ds_ns, rng)
    return c


def exp_pdp_taps(cfg: LinkConfig, rng):
    """Block-fading Rayleigh taps with exponential PDP exp(-k/(cp/9)) normalised
    (OFDM_MIMO_2-2_NBF_LDPC.py:162-164,272-279); rng is a RandomState."""
    pdp = np.exp(-np.arange(cfg.isi) / max(cfg.cp / 9, 1e-12))
    pdp = pdp / pdp.sum()
    c = np.zeros((cfg.n_r, cfg.n_t, cfg.isi), dtype=np.complex128)
    for nr in range(cfg.n_r):
        for nt in range(cfg.n_t):
            c[nr, nt] = (rng.randn(cfg.isi) + 1j * rng.randn(cfg.isi)) / np.sqrt(2) * np.sqrt(pdp)
    return c


def flat_taps(cfg: LinkConfig, rng):
    """One-tap channel of unit modulus and random phase per link, the SISO driver's
    `H_true = randn + 1j randn; H_true /= abs(H_true)`
    (Demo_SISO_QPSK_AWGN_LDPC_ESN_with_ZF_LS.py:205-206); rng is a RandomState."""
    c = np.zeros((cfg.n_r, cfg.n_t, cfg.isi), dtype=np.complex128)
    for nr in range(cfg.n_r):
        for nt in range(cfg.n_t):
            h = rng.randn() + 1j * rng.randn()
            c[nr, nt, 0] = h / np.abs(h)
    return c


def modulate(bits, cfg: LinkConfig, ebno_db, const=None):
    """bits [N*m x N_t] -> (X [N x N_t], x_cp pre-PA, x_cp post-PA).

    index = sum_b bit_b 2^b (driver:406-411); x = N*ifft(X); CP prepended;
    * sqrt(Pi); PA x / sqrt(1 + (|x|/A)^2) (driver:413-419)."""
    const = unit_qam(cfg.m) if const is None else const
    n, m = cfg.n_sub, cfg.m
    idx = (bits.reshape(n, m, cfg.n_t) * (1 << np.arange(m))[None, :, None]).sum(axis=1)
    x_f = const[idx]
    x_t = n * np.fft.ifft(x_f, axis=0)
    x_cp = np.concatenate([x_t[-cfg.cp:], x_t], axis=0) * math.sqrt(cfg.p_i(ebno_db)) \
        if cfg.cp > 0 else x_t * math.sqrt(cfg.p_i(ebno_db))
    a = cfg.a_clip(ebno_db)
    x_pa = x_cp / np.sqrt(1 + (np.abs(x_cp) / a) ** 2)
    return x_f, x_cp, x_pa


def channel(x_pa, taps, cfg: LinkConfig, rng):
    """sum_tx lfilter(c[rx][tx], 1, x) + sqrt(T*No/2) (randn + j randn)  (driver:421-427).
    rng is a RandomState; draw order per rx: randn(T) real then randn(T) imag."""
    t = x_pa.shape[0]
    y = np.zeros((t, cfg.n_r), dtype=np.complex128)
    for nr in range(cfg.n_r):
        for nt in range(cfg.n_t):
            y[:, nr] += signal.lfilter(taps[nr, nt], np.array([1]), x_pa[:, nt])
        y[:, nr] += math.sqrt(t * cfg.no / 2) * (rng.randn(t) + 1j * rng.randn(t))
    return y


def random_bits(cfg: LinkConfig, rng):
    return (rng.rand(cfg.n_sub * cfg.m, cfg.n_t) > 0.5).astype(np.int32)


def make_frame(cfg: LinkConfig, ebno_db, taps, rng):
    """One transmitted+received OFDM symbol: dict(bits, X, x_cp, y_cp)."""
    bits = random_bits(cfg, rng)
    x_f, x_cp, x_pa = modulate(bits, cfg, ebno_db)
    y_cp = channel(x_pa, taps, cfg, rng)
    return dict(bits=bits, X=x_f, x_cp=x_cp, y_cp=y_cp)
