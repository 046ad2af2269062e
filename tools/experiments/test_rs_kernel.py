"""Register-resident-state fp16 / bf16 predict kernel (csrc/esn_recur_rs.hip; N_res 257..512, n_in 13..16,
n_out <= 8: the headline shape; opt-in by the debug knob rs=1 -- it is correct but measured slower than the
default) against the LDS-state kernel on identical inputs and identical noise draws and against
the CPU oracle.

Not part of tests/: the kernel is compiled only into the experiment build
    ESN_WITH_RS=1 python esn_ofdm_mimo_amd/build.py --variant rs
    ESN_HIP_LIB=esn_ofdm_mimo_amd/libesn_hip_rs.so python -m pytest tools/experiments/test_rs_kernel.py -m gpu"""
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


@pytest.mark.parametrize("n_res,n_in,n_out,G,F,precision", [(512, 16, 8, 5, 75, "f16"), (300, 16, 8, 3, 75, "f16"),
                                                            (512, 14, 5, 2, 301, "f16"), (512, 16, 8, 7, 64, "bf16"),
                                                            (400, 16, 8, 1, 1, "f16")])
@pytest.mark.parametrize("noise_mode,noise", [("none", 0.0), ("counter", 1e-3)])
def test_register_state_kernel_matches_lds_state_kernel(mods, n_res, n_in, n_out, G, F, precision, noise_mode, noise):
    """Ragged groups (128-slot tiles straddle up to three groups, padding slots), short inputs (rows past T_in are
    zeros), per-group read-outs and scalings, initial state / feedback (continuation), both noise modes."""
    batched, lib = mods
    rs = np.random.RandomState(n_res + G)
    t_in, t, tr = 30, 34, 4
    w, w_in, w_fb = eo.draw_weights(rs, n_in, n_out, n_res, 0.9, 0.1)
    bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=noise)
    in_scale, in_shift = rs.rand(G, n_in) * 0.2 + 0.1, rs.randn(G, n_in) * 0.05
    t_scale, t_shift = rs.rand(G, n_out) + 0.5, rs.randn(G, n_out) * 0.1
    bank.set_scaling(in_scale, in_shift, t_scale, t_shift)
    w_out = rs.randn(G, n_out, n_res + n_in) * 0.004           # weak feedback: rounding differences are not amplified
    bank.set_readout(w_out)
    B = max(1, G * F - 9)                                      # last group is short
    u = rs.randn(B, t_in, n_in)
    x0, y0 = rs.randn(G, n_res) * 0.1, rs.randn(G, n_out) * 0.1
    kw = dict(T=t, transient=tr, precision=precision, x0=x0, y0=y0, noise_mode=noise_mode, seed=5)
    old = bank.predict(u, F, **kw).cpu().numpy()
    lib.debug_set("rs", "1")
    try:
        new = bank.predict(u, F, **kw).cpu().numpy()
    finally:
        lib.debug_set("rs", "0")
    assert new.shape == old.shape == (B, t - tr, n_out)
    tol = (2e-3 if noise == 0.0 else 8e-3) * (1 if precision == "f16" else 10)
    assert rel_err(new, old) < tol, rel_err(new, old)
    if F >= 64 and noise == 0.0:      # (F < 64: the shape falls back to the LDS-state kernel by design)
        for b in sorted({0, min(F - 1, B - 1), min(F, B - 1), B - 1}):
            grp = b // F
            o = eo.OracleESN(n_in, n_out, n_res, noise=0.0, input_scaling=in_scale[grp], input_shift=in_shift[grp],
                             teacher_scaling=t_scale[grp], teacher_shift=t_shift[grp], random_state=1)
            o.W, o.W_in, o.W_feedb, o.W_out = w, w_in, w_fb, w_out[grp]
            o.laststate, o.lastoutput = x0[grp], y0[grp]
            want = o.predict(np.vstack([u[b], np.zeros((t - t_in, n_in))]), tr, continuation=True)
            assert rel_err(new[b], want) < (2e-2 if precision == "f16" else 2e-1), (b, rel_err(new[b], want))
