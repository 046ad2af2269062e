"""Reservoirs beyond 1024 units (BASELINE configs[4], N_res = 2048): the launch-per-step GEMM path
(csrc/esn_recur_big.hip) against the persistent fp16 kernel on identical inputs and identical noise
draws (debug knob big_gemm=0), against the CPU oracle, and -- through tests/test_gpu_parity.py's c5 case
-- against the reference's own predictions."""
import numpy as np
import pytest

from oracle import esn_oracle as eo

pytestmark = pytest.mark.gpu


def rel_err(got, want):
    return float(np.max(np.abs(got - want)) / (np.max(np.abs(want)) + 1e-300))


_WEIGHTS = {}


def draw_weights_cached(seed, n_in, n_out, n_res):
    """eo.draw_weights (its eigvals of a 2048 x 2048 matrix takes 10-40 s of host time) once per shape and seed;
    returns the weights and a RandomState continued from where the draw left it."""
    key = (seed, n_in, n_out, n_res)
    if key not in _WEIGHTS:
        rs = np.random.RandomState(seed)
        _WEIGHTS[key] = (eo.draw_weights(rs, n_in, n_out, n_res, 0.9, 0.1), rs.get_state())
    rs = np.random.RandomState(0)
    rs.set_state(_WEIGHTS[key][1])
    return _WEIGHTS[key][0], rs


@pytest.fixture(scope="module")
def mods():
    from esn_ofdm_mimo_amd import batched, _lib
    return batched, _lib


@pytest.mark.parametrize("n_res,n_in,n_out,G,F,precision", [(2048, 16, 8, 5, 75, "f16"), (1500, 4, 4, 3, 37, "f16"),
                                                            (1100, 2, 2, 2, 301, "f16"), (2048, 16, 8, 2, 40, "bf16")])
@pytest.mark.parametrize("noise_mode,noise", [("none", 0.0), ("counter", 1e-3), ("tensor", 1e-3)])
def test_big_gemm_path_matches_persistent_kernel(mods, n_res, n_in, n_out, G, F, precision, noise_mode, noise):
    batched, lib = mods
    (w, w_in, w_fb), rs = draw_weights_cached(n_res, n_in, n_out, n_res)
    t_in, t, tr = 30, 34, 4
    bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=noise)
    in_scale, in_shift = rs.rand(G, n_in) * 0.2 + 0.1, rs.randn(G, n_in) * 0.05
    t_scale, t_shift = rs.rand(G, n_out) + 0.5, rs.randn(G, n_out) * 0.1
    bank.set_scaling(in_scale, in_shift, t_scale, t_shift)
    w_out = rs.randn(G, n_out, n_res + n_in) * 0.002          # weak feedback: rounding differences are not amplified
    bank.set_readout(w_out)
    B = G * F - 9                                             # last group is short
    u = rs.randn(B, t_in, n_in)
    x0, y0 = rs.randn(G, n_res) * 0.1, rs.randn(G, n_out) * 0.1
    kw = dict(T=t, transient=tr, precision=precision, x0=x0, y0=y0, noise_mode=noise_mode, seed=5)
    if noise_mode == "tensor":
        if (n_res, precision) != (2048, "f16"):
            pytest.skip("tensor noise: one shape is enough")
        kw["noise_u"] = rs.rand(B, t, n_res)
    from esn_ofdm_mimo_amd._lib import PRECISIONS
    import ctypes as C
    assert lib.load().esn_predict_workspace_bytes(PRECISIONS[precision], C.byref(bank.shape), B, F) > 0
    big = bank.predict(u, F, **kw).cpu().numpy()
    lib.debug_set("big_gemm", "0")
    try:
        persistent = bank.predict(u, F, **kw).cpu().numpy()
    finally:
        lib.debug_set("big_gemm", "1")
    assert big.shape == persistent.shape == (B, t - tr, n_out)
    tol = 2e-3 if precision == "f16" else 2e-2
    assert rel_err(big, persistent) < tol, rel_err(big, persistent)
    if noise == 0.0:
        for b in (0, F - 1, F, B - 1):
            grp = b // F
            o = eo.OracleESN(n_in, n_out, n_res, noise=0.0, input_scaling=in_scale[grp], input_shift=in_shift[grp],
                             teacher_scaling=t_scale[grp], teacher_shift=t_shift[grp], random_state=1,
                             weights=(w, w_in, w_fb))
            o.W_out = w_out[grp]
            o.laststate, o.lastoutput = x0[grp], y0[grp]
            want = o.predict(np.vstack([u[b], np.zeros((t - t_in, n_in))]), tr, continuation=True)
            assert rel_err(big[b], want) < (2e-2 if precision == "f16" else 1e-1), (b, rel_err(big[b], want))


def test_big_gemm_single_frame_and_no_workspace(mods):
    """One frame (255 padding slots), and the C ABI's NULL-workspace contract: persistent kernel, same result."""
    batched, lib = mods
    n_in, n_out, n_res = 16, 8, 2048
    (w, w_in, w_fb), rs = draw_weights_cached(n_res, n_in, n_out, n_res)
    bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=0.0)
    bank.set_readout(rs.randn(1, n_out, n_res + n_in) * 0.002)
    u = rs.randn(1, 20, n_in) * 0.1
    a = bank.predict(u, 1, transient=2, precision="f16", noise_mode="none").cpu().numpy()
    lib.debug_set("big_gemm", "0")
    try:
        b = bank.predict(u, 1, transient=2, precision="f16", noise_mode="none").cpu().numpy()
    finally:
        lib.debug_set("big_gemm", "1")
    assert rel_err(a, b) < 2e-3


@pytest.mark.parametrize("n_res,n_in,n_out,G,precision,e_dtype", [(2048, 16, 8, 70, "f16", "f32"), (2048, 16, 8, 5, "f16", "f64"),
                                                                 (1500, 4, 4, 129, "f16", "f32"), (2048, 16, 8, 33, "bf16", "f32"),
                                                                 # 257..1024 units, 64 pilots or more: the same GEMM with 32-deep chunks (debug knob harvest_gemm)
                                                                 (512, 16, 8, 70, "f16", "f32"), (300, 16, 8, 64, "f16", "f64"),
                                                                 (1024, 8, 4, 65, "bf16", "f32")])
@pytest.mark.parametrize("noise_mode,noise", [("none", 0.0), ("counter", 1e-3), ("tensor", 1e-3)])
def test_big_gemm_harvest_matches_persistent_kernel(mods, n_res, n_in, n_out, G, precision, e_dtype, noise_mode, noise):
    """Teacher-forced harvest as one 128 x 64-tiled GEMM launch per step (bigh_step_kernel) against the persistent
    harvest kernel on identical inputs and identical noise draws (ragged pilot counts: padding slots, partial
    tiles), and against the CPU oracle's extended states."""
    batched, lib = mods
    (w, w_in, w_fb), rs = draw_weights_cached(n_res, n_in, n_out, n_res)
    t = 24
    bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=noise)
    in_scale, in_shift = rs.rand(G, n_in) * 0.2 + 0.1, rs.randn(G, n_in) * 0.05
    t_scale, t_shift = rs.rand(G, n_out) * 0.2 + 0.1, rs.randn(G, n_out) * 0.02
    bank.set_scaling(in_scale, in_shift, t_scale, t_shift)
    u, d = rs.randn(G, t, n_in), rs.randn(G, t, n_out)
    kw = dict(precision=precision, noise_mode=noise_mode, seed=9, e_dtype=e_dtype, group_offset=3)
    if noise_mode == "tensor":
        if (n_res, precision, G) != (2048, "f16", 70):
            pytest.skip("tensor noise: one shape is enough")
        kw["noise_u"] = rs.rand(G, t - 1, n_res)
    from esn_ofdm_mimo_amd._lib import PRECISIONS
    import ctypes as C
    lib.debug_set("harvest_gemm", "1")          # (reservoirs of 257..1024 units take the GEMM path only on request: it is slower)
    try:
        assert lib.load().esn_harvest_workspace_bytes(PRECISIONS[precision], C.byref(bank.shape), G) > 0
        big = bank.harvest(u, d, **kw).double().cpu().numpy()
    finally:
        lib.debug_set("harvest_gemm", "0")
    lib.debug_set("big_gemm", "0")
    try:
        persistent = bank.harvest(u, d, **kw).double().cpu().numpy()
    finally:
        lib.debug_set("big_gemm", "1")
    assert big.shape == persistent.shape == (G, t, n_res + n_in)
    # same operand rounding, same k order, same noise stream: the two kernels agree to accumulation round-off
    tol = 2e-3 if precision == "f16" else 2e-2
    assert rel_err(big, persistent) < tol, rel_err(big, persistent)
    np.testing.assert_array_equal(big[:, :, n_res:], persistent[:, :, n_res:])          # scaled-input columns
    assert np.all(big[:, 0, :n_res] == 0.0)
    if noise == 0.0:
        for g in (0, G - 1):
            o = eo.OracleESN(n_in, n_out, n_res, noise=0.0, input_scaling=in_scale[g], input_shift=in_shift[g],
                             teacher_scaling=t_scale[g], teacher_shift=t_shift[g], random_state=1, weights=(w, w_in, w_fb))
            o.fit(u[g], d[g], 2)
            assert rel_err(big[g], o._ext_states) < (2e-2 if precision == "f16" else 1e-1)
