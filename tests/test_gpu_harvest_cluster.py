"""fp16 / bf16 harvest at 257..512 reservoir units on the cluster kernel (csrc/esn_harvest_cluster.hip: pairs of
co-resident workgroups per 16 pilots keep the weight matrix in registers and exchange their state slices as tagged
granules; clusters of 4 / 8 behind the knob) against the persistent harvest kernel (debug knob hcluster=0: same weights, same noise draws, only
the summation order differs) and the CPU oracle: ragged pilot counts (partial clusters, a lone pilot, several clusters
per XCD), the three noise modes, float32 and float64 extended states, reservoirs below the padded 512 rows."""
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


@pytest.mark.parametrize("n_res,n_in,n_out,G", [(512, 16, 8, 70), (512, 16, 8, 1), (300, 16, 8, 200), (384, 8, 6, 64),
                                                (512, 4, 4, 577)])
@pytest.mark.parametrize("noise_mode,noise", [("none", 0.0), ("counter", 1e-3), ("tensor", 1e-3)])
def test_cluster_harvest_matches_persistent_kernel_and_oracle(mods, n_res, n_in, n_out, G, noise_mode, noise):
    import ctypes as C
    batched, L = mods
    if noise_mode == "tensor" and G > 100:
        pytest.skip("tensor noise: the small shapes are enough")
    rs = np.random.RandomState(n_res + G)
    t = 31
    w, w_in, w_fb = eo.draw_weights(rs, n_in, n_out, n_res, 0.9, 0.1)
    bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=noise)
    in_scale, in_shift = rs.rand(G, n_in) * 0.2 + 0.1, rs.randn(G, n_in) * 0.05
    t_scale, t_shift = rs.rand(G, n_out) + 0.5, rs.randn(G, n_out) * 0.1
    bank.set_scaling(in_scale, in_shift, t_scale, t_shift)
    u, d = rs.randn(G, t, n_in), np.tanh(rs.randn(G, t, n_out))
    kw = dict(precision="f16", noise_mode=noise_mode, seed=5, group_offset=3)
    if noise_mode == "tensor":
        kw["noise_u"] = rs.rand(G, t - 1, n_res)
    assert L.load().esn_harvest_workspace_bytes(L.F16, C.byref(bank.shape), G) == ((G + 15) // 16) * 32768 + 64
    e32 = bank.harvest(u, d, e_dtype="f32", **kw)
    bank.raise_if_harvest_timed_out()
    e64 = bank.harvest(u, d, **kw)
    bank.raise_if_harvest_timed_out()
    L.debug_set("hcluster", "0")
    try:
        assert L.load().esn_harvest_workspace_bytes(L.F16, C.byref(bank.shape), G) == 0
        ref = bank.harvest(u, d, **kw).cpu().numpy()
    finally:
        L.debug_set("hcluster", "1")
    e32, e64 = e32.cpu().numpy(), e64.cpu().numpy()
    assert e64.shape == ref.shape == (G, t, n_res + n_in)
    # float32 states are the same fp16 values; the input columns round once
    assert np.array_equal(e32[..., :n_res].astype(np.float64), e64[..., :n_res])
    assert np.array_equal(e32[..., n_res:], e64[..., n_res:].astype(np.float32))
    # the input columns and row 0 do not depend on the kernel at all
    assert np.array_equal(e64[..., n_res:], ref[..., n_res:])
    assert np.array_equal(e64[:, 0], ref[:, 0])
    # states: fp16 roundings of sums taken in another order -- equal to a few half ulps
    assert rel_err(e64[..., :n_res], ref[..., :n_res]) < 4e-3, rel_err(e64[..., :n_res], ref[..., :n_res])
    same = np.mean(e64[..., :n_res] == ref[..., :n_res])
    assert same > 0.5, same
    if noise_mode == "none":
        for g in sorted({0, G // 2, G - 1}):
            o = eo.OracleESN(n_in, n_out, n_res, noise=0.0, input_scaling=in_scale[g], input_shift=in_shift[g],
                             teacher_scaling=t_scale[g], teacher_shift=t_shift[g], random_state=1, weights=(w, w_in, w_fb))
            o.fit(u[g], d[g], 0)
            assert rel_err(e64[g], o._ext_states) < 2e-2, (g, rel_err(e64[g], o._ext_states))


def test_cluster_harvest_bf16_and_fit(mods):
    """bf16 through the same kernel, and a whole fit (harvest + Cholesky solve) on top of it against the oracle's W_out."""
    batched, L = mods
    rs = np.random.RandomState(3)
    n_in, n_out, n_res, t, tr, G = 16, 8, 512, 138, 10, 96
    w, w_in, w_fb = eo.draw_weights(rs, n_in, n_out, n_res, 0.9, 0.1)
    bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=1e-3)
    bank.set_scaling(np.full((G, n_in), 0.15), None, np.full((G, n_out), 0.8), None)
    u, d = rs.randn(G, t, n_in), np.tanh(rs.randn(G, t, n_out))
    for precision in ("bf16", "f16"):
        e = bank.harvest(u, d, precision=precision, noise_mode="counter", seed=2, e_dtype="f32")
        bank.raise_if_harvest_timed_out()
        L.debug_set("hcluster", "0")
        try:
            ref = bank.harvest(u, d, precision=precision, noise_mode="counter", seed=2, e_dtype="f32")
        finally:
            L.debug_set("hcluster", "1")
        tol = 4e-3 if precision == "f16" else 3e-2
        assert rel_err(e.cpu().numpy(), ref.cpu().numpy()) < tol
    w_out, status = bank.solve(e, d, tr, method="chol")
    w_ref, _ = bank.solve(ref, d, tr, method="chol")
    assert int(status.ne(0).sum()) == 0
    # the two fits interpolate the same teacher from states that differ by fp16 roundings
    pred = np.einsum("gok,gtk->gto", w_out.cpu().numpy(), e.cpu().numpy().astype(np.float64)[:, tr:])
    want = d[:, tr:] * 0.8
    assert rel_err(pred, want) < 1e-6
    assert rel_err(w_out.cpu().numpy(), w_ref.cpu().numpy()) < 0.5


def test_cluster_harvest_with_several_weight_sets(mods):
    """One reservoir per coherence block (a pool of weight sets, group g -> set (group_offset + g) % n_wsets): every cluster
    serves ONE set, its pilots are that set's groups in order.  Same states as the persistent kernel, whatever the rotation."""
    batched, L = mods
    rs = np.random.RandomState(17)
    n_in, n_out, n_res, t, G, nws = 16, 8, 512, 25, 45, 3
    ws = [eo.draw_weights(rs, n_in, n_out, n_res, 0.9, 0.1) for _ in range(nws)]
    w, w_in, w_fb = (np.stack([x[i] for x in ws]) for i in range(3))
    bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=1e-3)
    bank.set_scaling(rs.rand(G, n_in) * 0.2 + 0.1, None, rs.rand(G, n_out) + 0.5, None)
    u, d = rs.randn(G, t, n_in), np.tanh(rs.randn(G, t, n_out))
    for off in (0, 5):
        kw = dict(precision="f16", noise_mode="counter", seed=8, group_offset=off)
        e = bank.harvest(u, d, **kw).cpu().numpy()
        bank.raise_if_harvest_timed_out()
        L.debug_set("hcluster", "0")
        try:
            ref = bank.harvest(u, d, **kw).cpu().numpy()
        finally:
            L.debug_set("hcluster", "1")
        assert np.array_equal(e[..., n_res:], ref[..., n_res:])
        assert rel_err(e[..., :n_res], ref[..., :n_res]) < 4e-3
    # noise-free against the oracle, group by group with ITS weight set
    bank0 = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=0.0)
    e0 = bank0.harvest(u, d, precision="f16", noise_mode="none", group_offset=5).cpu().numpy()
    for g in (0, 1, 2, 44):
        o = eo.OracleESN(n_in, n_out, n_res, noise=0.0, random_state=1, weights=ws[(5 + g) % nws])
        o.fit(u[g], d[g], 0)
        assert rel_err(e0[g], o._ext_states) < 2e-2, (g, rel_err(e0[g], o._ext_states))
