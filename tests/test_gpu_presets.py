"""The other driver configurations of the reference through the batched harness (SURVEY 8f / BASELINE
configs[1] and [2]):

  SISO QPSK over a flat channel  Demo_SISO_QPSK_AWGN_LDPC_ESN_with_ZF_LS.py:203-256 -- one pilot per Eb/No
        point, `fit(Ein, Eout)` with transient 0 and no output delay, `predict(x)` = continuation=True, CP = 0
  block fading, exponential PDP  OFDM_MIMO_2-2_NBF_LDPC.py:162-164,272-279

The frame recipe restatement (oracle/ofdm_frames.py) is parity-unpinned at the sample level (driver
scripts are not importable); what is pinned here: the HIP tap generator kinds 1 / 2 equal it with supplied
gains (1e-12), and the harness driven in the SISO semantics equals the pinned ESN oracle on the same
frames (float64 kernels: outputs 1e-8, bit-error counts exact)."""
import numpy as np
import pytest

from oracle import esn_oracle as eo
from oracle.ofdm_frames import LinkConfig, exp_pdp_taps, flat_taps, make_frame

pytestmark = pytest.mark.gpu


class Replay:
    """RandomState stand-in that returns supplied standard normals in draw order."""

    def __init__(self, values):
        self.v, self.i = np.asarray(values, dtype=float).ravel(), 0

    def randn(self, *shape):
        n = int(np.prod(shape)) if shape else 1
        out = self.v[self.i:self.i + n]
        self.i += n
        return out.reshape(shape) if shape else float(out[0])


@pytest.fixture(scope="module")
def mc():
    from esn_ofdm_mimo_amd import montecarlo
    return montecarlo


def test_exponential_pdp_and_flat_taps_vs_oracle(mc):
    import torch
    rs = np.random.RandomState(4)
    # kind 1: exponential PDP, one CN(0, pdp_k) tap per path
    prm = mc.LinkParams.block_fading(2, 2, 512)
    cfg = LinkConfig(n_t=2, n_r=2, n_sub=512)
    fs = mc.FrameSource(prm, seed=3)
    G = 3
    gains = rs.randn(G, cfg.n_r, cfg.n_t, cfg.isi) + 1j * rs.randn(G, cfg.n_r, cfg.n_t, cfg.isi)
    taps = fs.taps(G, 0, 0, gains=torch.as_tensor(gains, device=fs.device)).cpu().numpy()
    for b in range(G):
        # the oracle draws randn(isi) real then randn(isi) imaginary per link, rx-major
        feed = np.stack([gains[b].real, gains[b].imag], axis=2).ravel()
        np.testing.assert_allclose(taps[b], exp_pdp_taps(cfg, Replay(feed)), rtol=1e-12, atol=1e-15)
    # Philox-drawn taps follow the profile: E|h_k|^2 = pdp_k
    many = fs.taps(4096, 0, 0).cpu().numpy()
    pdp = np.exp(-np.arange(cfg.isi) / (cfg.cp / 9)); pdp /= pdp.sum()
    np.testing.assert_allclose((np.abs(many) ** 2).mean(axis=(0, 1, 2)), pdp, rtol=0.05, atol=1e-4)
    # kind 2: flat, unit modulus, random phase
    prm1 = mc.LinkParams.siso_awgn()
    cfg1 = LinkConfig(n_t=1, n_r=1, n_sub=512, m=2, isi=1)
    fs1 = mc.FrameSource(prm1, seed=3)
    g1 = rs.randn(5, 1, 1, 1) + 1j * rs.randn(5, 1, 1, 1)
    t1 = fs1.taps(5, 0, 0, gains=torch.as_tensor(g1, device=fs1.device)).cpu().numpy()
    for b in range(5):
        np.testing.assert_allclose(t1[b], flat_taps(cfg1, Replay([g1[b, 0, 0, 0].real, g1[b, 0, 0, 0].imag])), rtol=1e-12)
    drawn = fs1.taps(1000, 0, 0).cpu().numpy()
    np.testing.assert_allclose(np.abs(drawn), 1.0, rtol=1e-12)
    assert abs(np.angle(drawn).mean()) < 0.3 and np.angle(drawn).std() > 1.5      # phases spread over the circle


def test_siso_preset_equals_oracle_in_the_siso_drivers_semantics(mc):
    """One pilot, 24 data symbols, Eb/No 6 dB: fit without delay or transient, predict with continuation --
    the GPU harness (float64 kernels, noise 0) against the oracle ESN call for call."""
    import torch
    prm = mc.LinkParams.siso_awgn()
    assert (prm.cp, prm.delay, prm.forget, prm.continuation, prm.coherence_symbols) == (0, 0, 0, True, 400)
    cfg = LinkConfig(n_t=1, n_r=1, n_sub=512, m=2, isi=1)
    ebno, n_res, F = 6.0, 100, 24
    rs = np.random.RandomState(42)
    taps = flat_taps(cfg, rs)
    pilot = make_frame(cfg, ebno, taps, rs)
    data = [make_frame(cfg, ebno, taps, rs) for _ in range(F)]
    sweep = mc.DetectorSweep(prm, n_reservoir=n_res, noise=0.0, seed=5, precision="f64", fit_precision="f64",
                             solve_method="qr")
    w, w_in, w_fb = (sweep.bank._W[0].cpu().numpy(), sweep.bank._W_in[0].cpu().numpy(), sweep.bank._W_fb[0].cpu().numpy())
    sweep.set_snr(ebno, 1)
    dev = sweep.device
    E = sweep.train(torch.as_tensor(pilot["y_cp"][None], device=dev), torch.as_tensor(pilot["x_cp"][None], device=dev))
    assert int(sweep.bank.fit_status.sum().item()) == 0
    # oracle: Demo_SISO...py:222-226 (fit) and :253-256 (predict, FFT, de-scale)
    o = eo.OracleESN(2, 2, n_res, spectral_radius=0.9, sparsity=0.1, noise=0.0,
                     input_scaling=cfg.input_scaling(ebno) * np.ones(2), input_shift=np.zeros(2),
                     teacher_scaling=cfg.teacher_scale * np.ones(2), teacher_shift=np.zeros(2), random_state=1)
    o.W, o.W_in, o.W_feedb = w, w_in, w_fb
    ein = np.column_stack([pilot["y_cp"][:, 0].real, pilot["y_cp"][:, 0].imag])
    eout = np.column_stack([pilot["x_cp"][:, 0].real, pilot["x_cp"][:, 0].imag])
    o.fit(ein, eout)
    assert np.max(np.abs(E[0].cpu().numpy() - o._ext_states)) / np.abs(o._ext_states).max() < 1e-11
    o.W_out = sweep.bank.W_out[0].cpu().numpy()            # (512 x 102 fit at noise 0: compare predictions on one W_out)
    dy = torch.as_tensor(np.stack([fr["y_cp"] for fr in data]), device=dev)
    bits = torch.as_tensor(np.stack([fr["bits"] for fr in data]).astype(np.uint8), device=dev)
    err = torch.zeros(1, dtype=torch.int64, device=dev)
    nb = torch.zeros(1, dtype=torch.int64, device=dev)
    y = sweep.detect(dy, bits, F, err, nb).cpu().numpy()
    const = eo.unit_qam(cfg.m)
    want_err = 0
    for i, fr in enumerate(data):
        x = np.column_stack([fr["y_cp"][:, 0].real, fr["y_cp"][:, 0].imag])
        want = o.predict(x)                                  # continuation=True, transient 0
        assert np.max(np.abs(y[i] - want)) / np.abs(want).max() < 1e-8
        x_hat = eo.time_to_freq([want[:, 0] + 1j * want[:, 1]], cfg.n_sub, cfg.p_i(ebno))
        want_err += eo.count_bit_errors(fr["bits"], eo.hard_bits(x_hat, const, cfg.m))
    assert int(err.item()) == want_err and int(nb.item()) == F * cfg.n_sub * cfg.m
    assert want_err / (F * cfg.n_sub * cfg.m) < 0.2          # the detector works on this channel


@pytest.mark.parametrize("preset", ["siso", "nbf"])
def test_preset_sweeps_run_end_to_end(mc, preset):
    """Generator -> train -> detect -> counters for the two other drivers: BER falls with Eb/No."""
    prm = mc.LinkParams.siso_awgn(symbols_per_pilot=40) if preset == "siso" else mc.LinkParams.block_fading(2, 2, 512)
    sweep = mc.DetectorSweep(prm, n_reservoir=100, noise=0.001, seed=2, precision="f32", fit_precision="f32",
                             solve_method="auto")
    ber, counts = sweep.run([0.0, 9.0, 18.0], blocks_per_snr=16, chunk_blocks=16)
    assert counts[:, 1].min() > 0
    assert ber[0] > ber[1] > ber[2] and ber[2] < (0.05 if preset == "siso" else 0.2), ber    # 2x2 16-QAM block fading at N_res=100: 0.137 @ 18 dB
