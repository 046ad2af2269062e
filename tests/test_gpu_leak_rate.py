"""`leak_rate` -- an EXTENSION the reference does not have (pyESN's update is a == 1; BASELINE north_star's
formula x[t] = (1-a) x[t-1] + a tanh(W x[t-1] + W_in u[t] + W_fb y[t-1]) reduces to it).  Served by the float64 kernels
(vector ALU, float64 matrix pipe, single-sequence cluster) against the oracle's restatement of the same formula; the
noise term is added after the blend, where the reference adds it after tanh."""
import ctypes as C

import numpy as np
import pytest

from oracle import esn_oracle as eo

pytestmark = pytest.mark.gpu


def rel_err(got, want):
    return float(np.max(np.abs(got - want)) / (np.max(np.abs(want)) + 1e-300))


@pytest.mark.parametrize("n_res,B", [(100, 1), (512, 1), (100, 4), (100, 24), (300, 40)])
def test_leaky_predict_and_harvest_match_oracle(n_res, B):
    """B = 1: cluster kernel (one and several workgroups); 2..8: vector ALU; more: float64 matrix pipe."""
    from esn_ofdm_mimo_amd import batched
    rs = np.random.RandomState(n_res + B)
    n_in, n_out, t, tr, a = 4, 2, 30, 3, 0.35
    w, w_in, w_fb = eo.draw_weights(rs, n_in, n_out, n_res, 0.9, 0.1)
    bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=0.0, leak_rate=a)
    in_scale, t_scale = rs.rand(1, n_in) * 0.5 + 0.5, rs.rand(1, n_out) + 0.5
    bank.set_scaling(np.tile(in_scale, (B, 1)), None, np.tile(t_scale, (B, 1)), None)
    u, d = rs.randn(B, t, n_in), np.tanh(rs.randn(B, t, n_out))
    w_out = rs.randn(B, n_out, n_res + n_in) * 0.05
    bank.set_readout(w_out)
    x0, y0 = rs.randn(B, n_res) * 0.1, rs.randn(B, n_out) * 0.1
    y = bank.predict(u, 1, T=t, transient=tr, precision="f64", x0=x0, y0=y0, noise_mode="none").cpu().numpy()
    e = bank.harvest(u, d, precision="f64", noise_mode="none").cpu().numpy()
    bank.raise_if_cluster_timed_out()
    for b in sorted({0, B // 2, B - 1}):
        o = eo.OracleESN(n_in, n_out, n_res, noise=0.0, input_scaling=in_scale[0], teacher_scaling=t_scale[0],
                         random_state=1, weights=(w, w_in, w_fb), leak_rate=a)
        o.fit(u[b], d[b], 0)
        assert rel_err(e[b], o._ext_states) < 1e-10, (b, rel_err(e[b], o._ext_states))
        o.W_out = w_out[b]
        o.laststate, o.lastoutput = x0[b], y0[b]
        want = o.predict(u[b], tr, continuation=True)
        assert rel_err(y[b], want) < 1e-10, (b, rel_err(y[b], want))
    # the leak does something: the same call with a == 1 differs
    bank1 = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=0.0)
    bank1.set_scaling(np.tile(in_scale, (B, 1)), None, np.tile(t_scale, (B, 1)), None)
    e1 = bank1.harvest(u, d, precision="f64", noise_mode="none").cpu().numpy()
    assert rel_err(e1, e) > 1e-2


def test_leaky_dropin_and_precision_rule():
    from esn_ofdm_mimo_amd import batched, _lib
    from esn_ofdm_mimo_amd.pyESN import ESN
    rs = np.random.RandomState(5)
    u, d = rs.randn(60, 3), np.tanh(rs.randn(60, 2))
    esn = ESN(3, 2, n_reservoir=80, spectral_radius=0.9, sparsity=0.1, noise=0.0, random_state=42, leak_rate=0.5)
    o = eo.OracleESN(3, 2, n_reservoir=80, spectral_radius=0.9, sparsity=0.1, noise=0.0, random_state=42, leak_rate=0.5)
    p_gpu, p_cpu = esn.fit(u, d, 5), o.fit(u, d, 5)
    assert rel_err(esn.W_out, o.W_out) < 1e-6 and rel_err(p_gpu, p_cpu) < 1e-8
    y_gpu, y_cpu = esn.predict(u[:20]), o.predict(u[:20])
    assert rel_err(y_gpu, y_cpu) < 1e-8
    # the MFMA precisions do not carry the extension: a loud -2, never a silent a == 1
    w, w_in, w_fb = eo.draw_weights(rs, 4, 2, 64, 0.9, 0.1)
    bank = batched.ReservoirBank(4, 2, 64, w, w_in, w_fb, noise=0.0, leak_rate=0.5)
    bank.set_readout(rs.randn(1, 2, 68))
    with pytest.raises(_lib.EsnHipError, match="leak_rate"):
        bank.predict(rs.randn(3, 10, 4), 3, T=10, precision="f16", noise_mode="none")
    with pytest.raises(ValueError):
        batched.ReservoirBank(4, 2, 64, w, w_in, w_fb, leak_rate=1.5)
    sh = _lib.Shape(64, 4, 2, 1, 1, -0.2)
    assert _lib.load().esn_predict_workspace_bytes(_lib.F64, C.byref(sh), 1, 1) == 0
    assert b"leak_rate" in _lib.load().esn_last_error()
