"""The `esn_*_mem` entry points of include/esn_hip.h called with ESN_MEM_HOST: every array is a NumPy array on the
host (the reference's own calling convention, pyESN.py:154,218), only the packed images live in memory handed out
by esn_device_alloc -- the way a C caller without a HIP toolchain would drive the library.  Checked against the
CPU oracle (fit states, W_out, predictions) and, bit for bit, against the device-pointer entry points."""
import ctypes as C

import numpy as np
import pytest

from oracle import esn_oracle as eo

pytestmark = pytest.mark.gpu


def hp(a):
    """host pointer of a C-contiguous array (None -> NULL)"""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


@pytest.fixture(scope="module")
def mods():
    from esn_ofdm_mimo_amd import batched, _lib
    return batched, _lib


@pytest.mark.parametrize("precision,n_res,tol", [("f64", 100, 1e-10), ("f16", 512, 2e-2)])
def test_host_arrays_fit_predict_detect(mods, precision, n_res, tol):
    batched, L = mods
    lib = L.load()
    H, prec = L.MEM_HOST, L.PRECISIONS[precision]
    n_in, n_out, G, F, T, tr = 4, 4, 3, 5, 40, 6
    rs = np.random.RandomState(7)
    w, w_in, w_fb = eo.draw_weights(rs, n_in, n_out, n_res, 0.9, 0.1)
    shape = L.Shape(n_res, n_in, n_out, 1, 1)
    in_scale = np.ascontiguousarray(np.tile(rs.rand(1, n_in) * 0.2 + 0.1, (G, 1)))
    t_scale = np.ascontiguousarray(np.tile(rs.rand(1, n_out) + 0.5, (G, 1)))
    t_shift = np.ascontiguousarray(np.tile(rs.randn(1, n_out) * 0.1, (G, 1)))
    u_fit, d_fit = rs.randn(G, T, n_in), np.tanh(rs.randn(G, T, n_out))
    u = rs.randn(G * F, T, n_in)

    pw = lib.esn_device_alloc(lib.esn_packed_weights_bytes(prec, C.byref(shape)))
    pwo = lib.esn_device_alloc(lib.esn_packed_readout_bytes(prec, C.byref(shape)) * G)
    assert pw and pwo
    try:
        L.check(lib.esn_pack_weights_mem(H, prec, C.byref(shape), hp(w), hp(w_in), hp(w_fb), pw, None), "pack")
        # ---- fit: harvest + solve, all arrays on the host
        E = np.full((G, T, n_res + n_in), np.nan)
        L.check(lib.esn_harvest_batch_mem(H, prec, C.byref(shape), pw, hp(in_scale), None, hp(t_scale), hp(t_shift),
                                          hp(u_fit), hp(d_fit), G, T, 0.0, L.NOISE_NONE, None, 0, 0, hp(E), None, 0,
                                          None), "harvest")
        w_out = np.full((G, n_out, n_res + n_in), np.nan)
        status = np.full(G, -1, dtype=np.int32)
        L.check(lib.esn_readout_solve_batch_mem(H, hp(E), hp(d_fit), G, T, tr, n_res + n_in, n_out, hp(t_scale),
                                                hp(t_shift), hp(w_out), hp(status), None, None), "solve")
        assert (status == 0).all()
        # ---- predict
        L.check(lib.esn_pack_readout_mem(H, prec, C.byref(shape), G, hp(w_out), pwo, None), "pack_readout")
        Y = np.full((G * F, T - tr, n_out), np.nan)
        L.check(lib.esn_predict_batch_mem(H, prec, C.byref(shape), pw, pwo, hp(in_scale), None, hp(t_scale),
                                          hp(t_shift), hp(u), G * F, F, T, T, tr, None, None, 0.0, L.NOISE_NONE, None,
                                          0, 0, hp(Y), None, 0, None), "predict")
        assert np.isfinite(E).all() and np.isfinite(w_out).all() and np.isfinite(Y).all()

        # ---- the oracle on the same numbers
        for g in range(G):
            o = eo.OracleESN(n_in, n_out, n_res, noise=0.0, input_scaling=in_scale[g], teacher_scaling=t_scale[g],
                             teacher_shift=t_shift[g], random_state=1, weights=(w, w_in, w_fb))
            o.fit(u_fit[g], d_fit[g], tr)
            assert np.abs(E[g] - o._ext_states).max() < tol * max(1.0, np.abs(o._ext_states).max())
            if precision == "f64":
                assert np.abs(w_out[g] - o.W_out).max() < 1e-7 * np.abs(o.W_out).max()
            o.W_out = w_out[g]                        # (f16: the readout of the states this precision harvested)
            for f in range(F):
                want = o.predict(u[g * F + f], tr, continuation=False)
                assert np.abs(Y[g * F + f] - want).max() < tol * max(1.0, np.abs(want).max())

        # ---- the device-pointer entry points on the same inputs: bit-identical
        bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=0.0)
        bank.set_scaling(in_scale, None, t_scale, t_shift)
        L.debug_set("hcluster", "0")       # (the host call above lent no workspace: persistent harvest kernel on both sides)
        try:
            E_dev = bank.harvest(u_fit, d_fit, precision=precision, noise_mode="none").cpu().numpy()
        finally:
            L.debug_set("hcluster", "1")
        assert np.array_equal(E_dev, E)
        bank.set_readout(w_out)
        Y_dev = bank.predict(u, F, T=T, transient=tr, precision=precision, noise_mode="none").cpu().numpy()
        assert np.array_equal(Y_dev, Y)

        # ---- detector tail on host arrays: counters are read, added to and written back
        n_sub, n_t, m = 32, n_out // 2, 2
        Yd = np.ascontiguousarray(Y[:, :n_sub, :])
        p_i = np.full(G, float(np.mean(Yd ** 2) * 2))
        tx = rs.randint(0, 2, size=(G * F, n_sub * m, n_t)).astype(np.uint8)
        err = np.array([5, 0, 1], dtype=np.int64)
        bits = np.array([10, 0, 0], dtype=np.int64)
        L.check(lib.esn_detect_count_mem(H, hp(Yd), G * F, F, n_sub, n_t, m, hp(p_i), hp(tx), hp(err), hp(bits),
                                         None, None), "detect")
        e_dev, b_dev = bank.detect_count(Yd, tx, p_i, F, n_sub, n_t, m)
        assert np.array_equal(err - np.array([5, 0, 1]), e_dev.cpu().numpy())
        assert np.array_equal(bits - np.array([10, 0, 0]), b_dev.cpu().numpy())
        assert (bits[1:] == F * n_sub * m * n_t).all() and 0 < err[1] < bits[1]
    finally:
        assert lib.esn_device_free(pw) == 0 and lib.esn_device_free(pwo) == 0


def test_mem_kind_is_checked_and_device_kind_forwards(mods):
    batched, L = mods
    lib = L.load()
    shape = L.Shape(8, 2, 2, 1, 1)
    assert lib.esn_pack_weights_mem(7, L.F64, C.byref(shape), None, None, None, None, None) == -1
    assert b"memory kind" in lib.esn_last_error()
    # ESN_MEM_DEVICE is the plain entry point: same validation, same error text
    assert lib.esn_detect_count_mem(L.MEM_DEVICE, None, 1, 1, 8, 1, 2, None, None, None, None, None, None) < 0
    assert b"esn_detect_count" in lib.esn_last_error()
    assert lib.esn_predict_batch_mem(L.MEM_HOST, L.F64, C.byref(shape), None, None, None, None, None, None, None, 0, 1,
                                     4, 4, 0, None, None, 0.0, 0, None, 0, 0, None, None, 0, None) == -1
    assert lib.esn_device_free(None) == 0
