"""float64 recurrence on the float64 matrix pipe (csrc/esn_recur_f64_mfma.hip: v_mfma_f64_16x16x4_f64,
the reference's own arithmetic batched over frames) against the CPU oracle (1e-10: float64 round-off
in a different summation order) and against the vector-ALU float64 kernel on identical inputs and
identical noise draws (debug knob f64_mfma=0)."""
import numpy as np
import pytest

from oracle import esn_oracle as eo

pytestmark = pytest.mark.gpu


def rel_err(got, want):
    return float(np.max(np.abs(got - want)) / (np.max(np.abs(want)) + 1e-300))


@pytest.fixture(scope="module")
def mods():
    from esn_ofdm_mimo_amd import batched, _lib
    return batched, _lib


def _valu(lib, fn):
    lib.debug_set("f64_mfma", "0")
    try:
        return fn()
    finally:
        lib.debug_set("f64_mfma", "1")


@pytest.mark.parametrize("n_res,n_in,n_out,G,F", [(512, 16, 8, 3, 37), (300, 16, 8, 2, 75), (100, 4, 4, 3, 19),
                                                  (40, 3, 5, 3, 37), (70, 6, 12, 2, 9), (1024, 16, 8, 2, 20),
                                                  (640, 2, 2, 1, 17)])
@pytest.mark.parametrize("noise_mode,noise", [("none", 0.0), ("counter", 1e-3), ("tensor", 1e-3)])
def test_predict_f64_matrix_pipe(mods, n_res, n_in, n_out, G, F, noise_mode, noise):
    """Ragged groups (tiles straddle groups and padding slots), per-group read-outs and scalings, initial
    state / feedback, short inputs (rows past T_in are zeros), transient, all three noise modes."""
    batched, lib = mods
    rs = np.random.RandomState(n_res + n_in)
    t_in, t, tr = 21, 24, 3
    w, w_in, w_fb = eo.draw_weights(rs, n_in, n_out, n_res, 0.9, 0.1)
    bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=noise)
    in_scale, in_shift = rs.rand(G, n_in) * 0.4 + 0.1, rs.randn(G, n_in) * 0.05
    t_scale, t_shift = rs.rand(G, n_out) + 0.5, rs.randn(G, n_out) * 0.1
    bank.set_scaling(in_scale, in_shift, t_scale, t_shift)
    w_out = rs.randn(G, n_out, n_res + n_in) * 0.05
    bank.set_readout(w_out)
    B = G * F - 4                                                   # last group is short
    u = rs.randn(B, t_in, n_in)
    x0, y0 = rs.randn(G, n_res) * 0.1, rs.randn(G, n_out) * 0.1
    kw = dict(T=t, transient=tr, precision="f64", x0=x0, y0=y0, noise_mode=noise_mode, seed=7)
    if noise_mode == "tensor":
        if n_res not in (512, 40):
            pytest.skip("tensor noise: two shapes are enough")
        kw["noise_u"] = rs.rand(B, t, n_res)
    got = bank.predict(u, F, **kw).cpu().numpy()
    assert got.shape == (B, t - tr, n_out)
    ref = _valu(lib, lambda: bank.predict(u, F, **kw).cpu().numpy())
    assert rel_err(got, ref) < 1e-11, rel_err(got, ref)             # same arithmetic, same noise, other k order
    if noise_mode != "counter":
        for b in sorted({0, min(F - 1, B - 1), min(F, B - 1), B - 1}):
            grp = b // F
            o = eo.OracleESN(n_in, n_out, n_res, noise=noise, input_scaling=in_scale[grp], input_shift=in_shift[grp],
                             teacher_scaling=t_scale[grp], teacher_shift=t_shift[grp], random_state=1)
            o.W, o.W_in, o.W_feedb, o.W_out = w, w_in, w_fb, w_out[grp]
            o.laststate, o.lastoutput = x0[grp], y0[grp]
            if noise_mode == "tensor":
                class Replay:                                        # the oracle draws rand(n_res) once per step
                    def __init__(self, rows): self.rows, self.i = rows, 0
                    def rand(self, n):
                        self.i += 1
                        return self.rows[self.i - 1]
                o.rng = Replay(kw["noise_u"][b])
            want = o.predict(np.vstack([u[b], np.zeros((t - t_in, n_in))]), tr, continuation=True)
            assert rel_err(got[b], want) < 1e-10, (b, rel_err(got[b], want))


@pytest.mark.parametrize("n_res,n_in,n_out,G", [(512, 16, 8, 37), (100, 4, 4, 21), (1024, 16, 8, 17), (70, 6, 12, 9)])
@pytest.mark.parametrize("noise", [0.0, 1e-3])
def test_harvest_f64_matrix_pipe(mods, n_res, n_in, n_out, G, noise):
    """G pilots through a shared reservoir: extended states vs the oracle (noise 0) and vs the vector-ALU
    kernel (same counter noise)."""
    batched, lib = mods
    rs = np.random.RandomState(G)
    t = 30
    w, w_in, w_fb = eo.draw_weights(rs, n_in, n_out, n_res, 0.9, 0.1)
    bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=noise)
    in_scale, in_shift, t_scale = rs.rand(G, n_in) * 0.4 + 0.1, rs.randn(G, n_in) * 0.05, rs.rand(G, n_out) + 0.5
    bank.set_scaling(in_scale, in_shift, t_scale, None)
    u, d = rs.randn(G, t, n_in), rs.randn(G, t, n_out) * 0.3
    mode = "counter" if noise else "none"
    E = bank.harvest(u, d, precision="f64", noise_mode=mode, seed=3).cpu().numpy()
    ref = _valu(lib, lambda: bank.harvest(u, d, precision="f64", noise_mode=mode, seed=3).cpu().numpy())
    assert E.shape == (G, t, n_res + n_in)
    assert rel_err(E, ref) < 1e-12
    if not noise:
        for g in (0, G // 2, G - 1):
            o = eo.OracleESN(n_in, n_out, n_res, noise=0.0, input_scaling=in_scale[g], input_shift=in_shift[g],
                             teacher_scaling=t_scale[g], random_state=1)
            o.W, o.W_in, o.W_feedb = w, w_in, w_fb
            o.fit(u[g], d[g], 0)
            assert rel_err(E[g], o._ext_states) < 1e-11


def test_per_group_reservoirs_f64_matrix_pipe(mods):
    """One (W, W_in, W_fb) per group (reference-faithful mode): whole-tile padding per group, fit + predict."""
    batched, lib = mods
    rs = np.random.RandomState(5)
    n_in, n_out, n_res, t, G, F = 4, 2, 48, 40, 11, 5
    ws = [eo.draw_weights(np.random.RandomState(100 + g), n_in, n_out, n_res, 0.9, 0.2) for g in range(G)]
    bank = batched.ReservoirBank(n_in, n_out, n_res, np.stack([w[0] for w in ws]), np.stack([w[1] for w in ws]),
                                 np.stack([w[2] for w in ws]), noise=0.0)
    u = rs.randn(G, t, n_in)
    d = np.tanh(u @ rs.randn(n_in, n_out)) + 0.3 * np.roll(u[:, :, :n_out], 1, axis=1)
    bank.fit(u, d, transient=4, precision="f64", noise_mode="none")
    assert int(bank.fit_status.sum().item()) == 0
    u2 = rs.randn(G * F, t, n_in)
    got = bank.predict(u2, F, transient=2, precision="f64").cpu().numpy()
    for g in (0, 5, G - 1):
        o = eo.OracleESN(n_in, n_out, n_res, noise=0.0, random_state=1)
        o.W, o.W_in, o.W_feedb = ws[g]
        o.fit(u[g], d[g], 4)
        o.W_out = bank.W_out[g].cpu().numpy()
        for f in range(F):
            assert rel_err(got[g * F + f], o.predict(u2[g * F + f], 2, continuation=False)) < 1e-9


def test_teacher_forcing_off_and_tail_transient(mods):
    batched, lib = mods
    rs = np.random.RandomState(9)
    n_in, n_out, n_res, t, G, F = 5, 2, 70, 17, 2, 10
    w, w_in, w_fb = eo.draw_weights(rs, n_in, n_out, n_res, 0.8, 0.2)
    bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, teacher_forcing=False, noise=0.0)
    w_out = rs.randn(G, n_out, n_res + n_in) * 0.05
    bank.set_readout(w_out)
    u = rs.randn(G * F, t, n_in) * 0.5
    for tr in (0, t - 1):
        got = bank.predict(u, F, transient=tr, precision="f64").cpu().numpy()
        for b in (0, F, G * F - 1):
            o = eo.OracleESN(n_in, n_out, n_res, noise=0.0, teacher_forcing=False, random_state=1)
            o.W, o.W_in, o.W_feedb, o.W_out = w, w_in, w_fb, w_out[b // F]
            assert rel_err(got[b], o.predict(u[b], tr, continuation=False)) < 1e-10


@pytest.mark.parametrize("amp", [0.005, 0.6])
def test_predict_f64_series_and_library_tanh(mods, amp):
    """The activation takes the 21st-order series when a wave's pre-activations are all below 0.25 and
    the library routine otherwise (esn_common.h tanh_f64_series): amp=0.005 keeps every state below 0.2
    (series everywhere, as on the OFDM workload), amp=0.6 drives some waves above (both paths in one
    launch).  Either way the result is the oracle's to float64 round-off."""
    batched, lib = mods
    n_res, n_in, n_out, G, F, t = 512, 16, 8, 2, 40, 30
    rs = np.random.RandomState(5)
    w, w_in, w_fb = eo.draw_weights(rs, n_in, n_out, n_res, 0.9, 0.1)
    bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=0.0)
    w_out = rs.randn(G, n_out, n_res + n_in) * 0.002
    bank.set_readout(w_out)
    u = rs.randn(G * F, t, n_in) * amp
    u[F:] *= 0.005 / amp                                            # the second group is always small
    got = bank.predict(u, F, T=t, precision="f64").cpu().numpy()
    ref = _valu(lib, lambda: bank.predict(u, F, T=t, precision="f64").cpu().numpy())
    assert rel_err(got, ref) < 1e-12, rel_err(got, ref)
    peak = 0.0
    for b in (0, F - 1, F, 2 * F - 1):
        o = eo.OracleESN(n_in, n_out, n_res, noise=0.0, random_state=1)
        o.W, o.W_in, o.W_feedb, o.W_out = w, w_in, w_fb, w_out[b // F]
        seen, step = [], o.step
        o.step = lambda x, uu, y: seen.append(step(x, uu, y)) or seen[-1]
        want = o.predict(u[b], 0, continuation=False)
        assert rel_err(got[b], want) < 1e-12, (b, rel_err(got[b], want))
        if b < F:
            peak = max(peak, float(np.max(np.abs(seen))))
    assert (peak < 0.2) == (amp < 0.1), peak                        # the case exercises the path it names
