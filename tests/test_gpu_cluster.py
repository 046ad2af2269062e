"""One float64 sequence on the LDS-resident cluster kernel (csrc/esn_recur_cluster.hip: the reservoir matrix stays
in the LDS of C co-resident workgroups for all T steps, the state travels through L2 as tagged granules) against
the vector-ALU kernel (debug knob cluster=0) and the CPU oracle: predict with both continuation modes, harvest,
the three noise modes, reservoirs that need 1, 10, 16 and 64 workgroups.  The reference-generated goldens reach
the same kernel through the 2-D drop-in (tests/test_gpu_dropin_semantics.py, tests/test_gpu_parity.py)."""
import ctypes as C

import numpy as np
import pytest

from oracle import esn_oracle as eo

pytestmark = pytest.mark.gpu


class ReplayRng:
    """Stands in for the oracle's RandomState: rand(n) hands out the rows of a pre-drawn uniform tensor."""

    def __init__(self, rows):
        self.rows, self.i = rows, 0

    def rand(self, n):
        r = self.rows[self.i]
        self.i += 1
        assert r.shape == (n,)
        return r


def rel_err(got, want):
    return float(np.max(np.abs(got - want)) / (np.max(np.abs(want)) + 1e-300))


@pytest.fixture(scope="module")
def mods():
    from esn_ofdm_mimo_amd import batched, _lib
    return batched, _lib


@pytest.mark.parametrize("n_res,n_in,n_out", [(100, 2, 2), (300, 16, 8), (512, 16, 8), (1024, 4, 4), (37, 3, 1)])
@pytest.mark.parametrize("noise_mode,noise", [("none", 0.0), ("tensor", 1e-3), ("counter", 1e-3)])
def test_cluster_kernel_matches_vector_kernel_and_oracle(mods, n_res, n_in, n_out, noise_mode, noise):
    batched, lib = mods
    from esn_ofdm_mimo_amd._lib import PRECISIONS
    rs = np.random.RandomState(n_res + n_in)
    w, w_in, w_fb = eo.draw_weights(rs, n_in, n_out, n_res, 0.9, 0.1)
    bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=noise)
    in_scale, in_shift = rs.rand(1, n_in) * 0.2 + 0.1, rs.randn(1, n_in) * 0.05
    t_scale, t_shift = rs.rand(1, n_out) + 0.5, rs.randn(1, n_out) * 0.1
    bank.set_scaling(in_scale, in_shift, t_scale, t_shift)
    t_in, t, tr = 40, 43, 5
    u = rs.randn(1, t_in, n_in)
    d = np.tanh(rs.randn(1, t_in, n_out))
    w_out = rs.randn(1, n_out, n_res + n_in) * 0.02
    bank.set_readout(w_out)
    x0, y0 = rs.randn(1, n_res) * 0.1, rs.randn(1, n_out) * 0.1
    assert lib.load().esn_predict_workspace_bytes(PRECISIONS["f64"], C.byref(bank.shape), 1, 1) > 0
    assert lib.load().esn_harvest_workspace_bytes(PRECISIONS["f64"], C.byref(bank.shape), 1) > 0
    kw = dict(precision="f64", noise_mode=noise_mode, seed=4)
    nz_p = rs.rand(1, t, n_res) if noise_mode == "tensor" else None
    nz_h = rs.rand(1, t_in - 1, n_res) if noise_mode == "tensor" else None

    def run():
        fresh = bank.predict(u, 1, T=t, transient=tr, noise_u=nz_p, **kw).cpu().numpy()
        cont = bank.predict(u, 1, T=t, transient=0, x0=x0, y0=y0, noise_u=nz_p, **kw).cpu().numpy()
        e = bank.harvest(u, d, noise_u=nz_h, **kw).cpu().numpy()
        bank.raise_if_cluster_timed_out()
        return fresh, cont, e

    got = run()
    lib.debug_set("cluster", "0")
    try:
        ref = run()
    finally:
        lib.debug_set("cluster", "1")
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        assert rel_err(a, b) < 1e-11, rel_err(a, b)
    if noise_mode != "counter":
        o = eo.OracleESN(n_in, n_out, n_res, noise=noise, input_scaling=in_scale[0], input_shift=in_shift[0],
                         teacher_scaling=t_scale[0], teacher_shift=t_shift[0], random_state=1, weights=(w, w_in, w_fb))
        o.W_out = w_out[0]
        o.laststate, o.lastoutput = x0[0], y0[0]
        if noise_mode == "tensor":
            o.rng = ReplayRng(nz_p[0])
        upad = np.vstack([u[0], np.zeros((t - t_in, n_in))])
        want = o.predict(upad, 0, continuation=True)
        assert rel_err(got[1], want[None]) < 1e-10
        if noise_mode == "tensor":
            o.rng = ReplayRng(nz_h[0])
        o.fit(u[0], d[0], 2)
        assert rel_err(got[2][0], o._ext_states) < 1e-10
