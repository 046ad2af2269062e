"""Pins oracle/esn_oracle.py (the CPU restatement) to golden vectors produced by
importing the reference itself (tests/golden/make_golden.py).  CPU only.

Tolerance: 1e-12 relative where the arithmetic is a fixed sequence of float64
ops; 1e-9 on quantities that pass through LAPACK pinv of an ill-conditioned
matrix (noise=0 cases) because gesdd rounding differs run to run only in the
last bits but is amplified by cond(S)."""
import hashlib

import numpy as np
import pytest

from oracle import esn_oracle as eo
from oracle.ofdm_frames import LinkConfig


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


PLAIN = {
    "tiny": dict(n_in=3, n_out=2, n_res=8,
                 kw=dict(spectral_radius=0.9, sparsity=0.25, input_scaling=[0.3, 0.2, 0.1],
                         input_shift=[0.0, 0.1, -0.1], teacher_scaling=0.5, teacher_shift=0.05)),
    "c2": dict(n_in=2, n_out=2, n_res=100,
               kw=dict(spectral_radius=0.9, sparsity=0.1, input_scaling=0.05 * np.ones(2),
                       input_shift=np.zeros(2), teacher_scaling=5e-3 * np.ones(2),
                       teacher_shift=np.zeros(2))),
}


@pytest.mark.parametrize("name", list(PLAIN))
def test_weights_bit_exact(golden, name):
    g, c = golden(name), PLAIN[name]
    esn = eo.OracleESN(c["n_in"], c["n_out"], c["n_res"], noise=0.0,
                       random_state=int(g["seed"]), **c["kw"])
    for nm in ("W", "W_in", "W_feedb"):
        a = getattr(esn, nm)
        assert sha(a) == str(g[nm + "_sha"]), nm
        np.testing.assert_array_equal(a.ravel()[:4], g[nm + "_head"])
        np.testing.assert_array_equal(a.ravel()[-4:], g[nm + "_tail"])


@pytest.mark.parametrize("name", list(PLAIN))
@pytest.mark.parametrize("tag,noise", [("n0", 0.0), ("n1", 0.001)])
def test_fit_predict_match_reference(golden, name, tag, noise):
    g, c = golden(name), PLAIN[name]
    tr = int(g["transient"])
    esn = eo.OracleESN(c["n_in"], c["n_out"], c["n_res"], noise=noise,
                       random_state=int(g["seed"]), **c["kw"])
    pred_train = esn.fit(g["u"], g["d"], tr)
    rt = 1e-9
    np.testing.assert_allclose(esn.laststate, g[tag + "_laststate"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(esn.lastoutput, g[tag + "_lastoutput"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(pred_train, g[tag + "_pred_train"], rtol=rt, atol=1e-9)
    scale = np.abs(g[tag + "_W_out"]).max()
    np.testing.assert_allclose(esn.W_out, g[tag + "_W_out"], rtol=1e-6, atol=1e-8 * scale)
    # use the reference's W_out from here so predict is compared op-for-op
    esn.W_out = g[tag + "_W_out"]
    np.testing.assert_allclose(esn.predict(g["u2"], 0, continuation=True),
                               g[tag + "_pred_cont"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(esn.predict(g["u2"], tr, continuation=False),
                               g[tag + "_pred_fresh"], rtol=1e-10, atol=1e-12)


HELPER = {
    "c3": (LinkConfig(n_t=2, n_r=2, n_sub=512), 100),
    "c4": (LinkConfig(), 512),
    "c4s": (LinkConfig(), 300),
}


def _helper_esn(cfg, n_res, seed, ebno, noise):
    n_in, n_out = 2 * cfg.n_r, 2 * cfg.n_t
    return eo.OracleESN(n_in, n_out, n_res, spectral_radius=0.9, sparsity=0.1, noise=noise,
                        input_shift=np.zeros(n_in),
                        input_scaling=cfg.input_scaling(ebno) * np.ones(n_in),
                        teacher_scaling=cfg.teacher_scale * np.ones(n_out),
                        teacher_shift=np.zeros(n_out), feedback_scaling=np.zeros(n_out),
                        random_state=seed)


@pytest.mark.parametrize("name", list(HELPER))
def test_helper_and_detection_match_reference(golden, name):
    g = golden(name)
    cfg, n_res = HELPER[name]
    seed, ebno = int(g["seed"]), float(g["ebno_db"])
    esn = _helper_esn(cfg, n_res, seed, ebno, 0.0)
    assert sha(esn.W) == str(g["W_sha"])
    ret = eo.train_mimo_esn(esn, 0, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub,
                            cfg.n_t, cfg.n_r, cfg.isi, g["pilot_y"], g["pilot_x"])
    x_in, x_out, _, delay, d_idx, d_min, d_max, forget, nmse = ret
    np.testing.assert_array_equal(x_in, g["esn_in"])
    np.testing.assert_array_equal(x_out, g["esn_out"])
    np.testing.assert_array_equal(delay, g["delay"])
    assert (d_idx, d_min, d_max, forget) == (int(g["d_idx"]), int(g["d_min"]),
                                             int(g["d_max"]), int(g["forget"]))
    np.testing.assert_allclose(esn.laststate, g["n0_laststate"], rtol=1e-11, atol=1e-14)
    # W_out of an (under-determined or ill-conditioned) pinv: compare through its action
    ext = esn._ext_states[forget:]
    np.testing.assert_allclose(ext @ esn.W_out.T, ext @ g["n0_W_out"].T, rtol=1e-6,
                               atol=1e-9 * np.abs(x_out).max() * cfg.teacher_scale)
    assert nmse == pytest.approx(float(g["n0_nmse"]), rel=1e-5)
    # data frames through the reference's W_out: op-for-op predict parity
    esn.W_out = g["n0_W_out"]
    if "data_y" in g:
        for y_cp, want in zip(g["data_y"], g["data_pred"]):
            got = esn.predict(eo.pack_rx(y_cp, d_max), forget, continuation=False)
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9 * np.abs(want).max())


def test_helper_with_noise_replays_rng_stream(golden):
    """noise=0.001: the oracle draws (T-1)+T+(T-1) x rand(N_res) in the reference's order."""
    g = golden("c4s")
    cfg, n_res = HELPER["c4s"]
    esn = _helper_esn(cfg, n_res, int(g["seed"]), float(g["ebno_db"]), 0.001)
    eo.train_mimo_esn(esn, 0, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub,
                      cfg.n_t, cfg.n_r, cfg.isi, g["pilot_y"], g["pilot_x"])
    np.testing.assert_allclose(esn.laststate, g["n1_laststate"], rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(esn.W_out, g["n1_W_out"], rtol=1e-5,
                               atol=1e-7 * np.abs(g["n1_W_out"]).max())


def test_constellation_matches_reference(golden):
    g = golden("constellation")
    for m in (2, 4, 6):
        np.testing.assert_allclose(eo.unit_qam(m), g[f"qam{m}"], rtol=0, atol=1e-15)


def test_bits_and_counts_on_golden_batch(golden):
    """a10-a12 on the reference's own predictions: BER plausibility + self-consistency."""
    g = golden("c4")
    cfg = LinkConfig()
    bits = np.unpackbits(g["data_bits"])[:np.prod(g["data_bits_shape"])].reshape(g["data_bits_shape"])
    const = eo.unit_qam(cfg.m)
    errs = tot = 0
    for pred, b in zip(g["data_pred"], bits):
        seqs = eo.outputs_to_time_signals(pred, g["delay"], int(g["d_min"]), cfg.n_sub, cfg.n_t)
        x_hat = eo.time_to_freq(seqs, cfg.n_sub, cfg.p_i(float(g["ebno_db"])))
        rx = eo.hard_bits(x_hat, const, cfg.m)
        errs += eo.count_bit_errors(b, rx)
        tot += b.size
    ber = errs / tot
    # published curve (N_res=300, noisy): 0.2445 @ 12 dB; N_res=512 on one channel draw
    assert 0.05 < ber < 0.45, ber


def test_misc_semantics(golden):
    g = golden("misc")
    esn = eo.OracleESN(1, 1, n_reservoir=20, spectral_radius=0.8, sparsity=0.2, noise=0.0,
                       input_scaling=0.5, input_shift=0.1, teacher_scaling=0.7,
                       teacher_shift=-0.2, teacher_forcing=False, random_state=99)
    np.testing.assert_allclose(esn.fit(g["u"], g["d"], 3), g["nofb_pred_train"], rtol=1e-8, atol=1e-10)
    esn.W_out = g["nofb_W_out"]
    np.testing.assert_allclose(esn.predict(g["u"][:17], 2, continuation=True), g["nofb_pred"],
                               rtol=1e-10, atol=1e-12)
    esn = eo.OracleESN(1, 1, n_reservoir=20, spectral_radius=1.1, noise=0.0, random_state=3)
    np.testing.assert_allclose(esn.fit(g["u"], g["d"]), g["plain_pred_train"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(esn.predict(g["u"][:9]), g["plain_pred"], rtol=1e-6, atol=1e-8)
    np.testing.assert_array_equal(eo.broadcast_arg(2.5, 4), g["cd_scalar"])
    with pytest.raises(ValueError, match=str(g["err_msgs"][0])):
        eo.OracleESN(3, 1, input_scaling=[1.0, 2.0])
    with pytest.raises(ValueError, match=str(g["err_msgs"][1])):
        eo.OracleESN(3, 1, input_scaling=np.zeros((2, 2)))
    assert eo.broadcast_arg(None, 3) is None


def test_mackey_glass_free_run(golden):
    """BASELINE configs[0]: 2000-step teacher-forced fit on a constant input, 2000-step free run
    (continuation=True), the one workload where W_feedb*y is O(1).  The series generator is pinned
    too (the fixture stores the series the reference was run on)."""
    from oracle.mackey_glass import mackey_glass
    g = golden("mackey")
    n_train, n_free = int(g["trainlen"]), int(g["future"])
    series = g["series"]
    np.testing.assert_allclose(mackey_glass(n_train + n_free), series, rtol=1e-12, atol=0)
    for tag, noise in (("n1", 0.001), ("n0", 0.0)):
        esn = eo.OracleESN(1, 1, n_reservoir=int(g["n_res"]), spectral_radius=1.5, noise=noise,
                           random_state=int(g["seed"]))
        if tag == "n1":
            assert sha(esn.W) == str(g["W_sha"]) and sha(esn.W_feedb) == str(g["W_feedb_sha"])
        pred_train = esn.fit(np.ones(n_train), series[:n_train])
        np.testing.assert_allclose(esn.laststate, g[tag + "_laststate"], rtol=1e-11, atol=1e-14)
        np.testing.assert_allclose(esn.lastoutput, g[tag + "_lastoutput"], rtol=1e-13)
        np.testing.assert_allclose(pred_train, g[tag + "_pred_train"], rtol=1e-7, atol=1e-9)
        free = esn.predict(np.ones(n_free))
        if tag == "n1":
            # with the default state noise the fitted system is a stable limit cycle: a 1e-14
            # perturbation stays ~1e-14 over the 2000 steps (measured), so the whole run is compared
            np.testing.assert_allclose(free, g["n1_free_run"], rtol=0, atol=1e-7)
            rmse = np.sqrt(np.mean((free.ravel()[:500] - series[n_train:n_train + 500]) ** 2))
            assert rmse < 0.15                  # and it does predict the series (0.094 in the reference run)
        else:
            # noise = 0: cond(E) 2.6e6 and a chaotic free run (1e-14 -> 1e-10 after 200 steps)
            np.testing.assert_allclose(free[:100], g["n0_free_run"][:100], rtol=0, atol=1e-6)


def _small_2x2_esn(g, noise):
    cfg = LinkConfig(n_t=2, n_r=2, n_sub=128)
    return cfg, _helper_esn(cfg, 100, int(g["seed"]), float(g["ebno_db"]), noise)


@pytest.mark.parametrize("tag,noise", [("n0", 0.0), ("n1", 0.001)])
def test_helper_delay_scan_matches_reference(golden, tag, noise):
    """DelayFlag=1: every delay in [Min, Max] is fitted and predicted, the lowest (mis-aligned)
    NMSE wins, then the final fit (helper:66-84) -- 15 RNG-consuming calls replayed in order."""
    g = golden("scan")
    cfg, esn = _small_2x2_esn(g, noise)
    ret = eo.train_mimo_esn(esn, 1, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub, cfg.n_t, cfg.n_r, cfg.isi,
                            g["pilot_y"], g["pilot_x"])
    x_in, x_out, _, delay, d_idx, d_min, d_max, forget, nmse = ret
    np.testing.assert_array_equal(x_in, g[tag + "_esn_in"])
    np.testing.assert_array_equal(x_out, g[tag + "_esn_out"])
    np.testing.assert_array_equal(delay, g[tag + "_delay"])
    assert (d_idx, d_min, d_max, forget) == tuple(int(g[tag + k]) for k in ("_d_idx", "_d_min", "_d_max", "_forget"))
    assert nmse == pytest.approx(float(g[tag + "_nmse"]), rel=1e-6)
    np.testing.assert_allclose(esn.laststate, g[tag + "_laststate"], rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(esn.W_out, g[tag + "_W_out"], rtol=1e-5, atol=1e-7 * np.abs(g[tag + "_W_out"]).max())


@pytest.mark.parametrize("tag,noise", [("n0", 0.0), ("n1", 0.001)])
def test_legacy_helpfunc_trainer_matches_reference(golden, tag, noise):
    """a13: HelpFunc.trainMIMOESN (7 x fit+predict, forced row 3, final fit) incl. the printed vector."""
    g = golden("legacy")
    cfg, esn = _small_2x2_esn(g, noise)
    printed = []
    ret = eo.train_mimo_esn_legacy(esn, 0, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub, cfg.n_t, cfg.n_r,
                                   cfg.isi, g["pilot_y"], g["pilot_x"], echo=printed.append)
    x_in, x_out, _, delay, d_idx, d_min, d_max, forget, nmse = ret
    np.testing.assert_array_equal(x_in, g[tag + "_esn_in"])
    np.testing.assert_array_equal(x_out, g[tag + "_esn_out"])
    np.testing.assert_array_equal(delay, g[tag + "_delay"])
    assert delay.dtype == g[tag + "_delay"].dtype
    assert (d_idx, int(d_min), int(d_max), int(forget)) == (3, 3, 3, 10)
    assert nmse == pytest.approx(float(g[tag + "_nmse"]), rel=1e-5)
    np.testing.assert_allclose(printed[0], g[tag + "_printed_values"], rtol=1e-5)
    np.testing.assert_allclose(esn.laststate, g[tag + "_laststate"], rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(esn.W_out, g[tag + "_W_out"], rtol=1e-5, atol=1e-7 * np.abs(g[tag + "_W_out"]).max())
    with pytest.raises(Exception) as ei:
        eo.train_mimo_esn_legacy(esn, 1, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub, cfg.n_t, cfg.n_r,
                                 cfg.isi, g["pilot_y"], g["pilot_x"])
    assert type(ei.value).__name__ == str(g["flag1_error"])


def test_leak_rate_extension_reduces_to_the_reference_update():
    """leak_rate is an extension: a == 1 is bit-for-bit the reference's update, a < 1 blends with the previous state."""
    rs = np.random.RandomState(3)
    u, d = rs.randn(25, 2), np.tanh(rs.randn(25, 1))
    a1 = eo.OracleESN(2, 1, n_reservoir=20, spectral_radius=0.9, noise=0.001, random_state=7)
    b1 = eo.OracleESN(2, 1, n_reservoir=20, spectral_radius=0.9, noise=0.001, random_state=7, leak_rate=1.0)
    assert np.array_equal(a1.fit(u, d), b1.fit(u, d))
    c = eo.OracleESN(2, 1, n_reservoir=20, spectral_radius=0.9, noise=0.0, random_state=7, leak_rate=0.25)
    c.fit(u, d)
    x = np.zeros(20)
    for t in range(1, 4):       # the first rows of the harvested states, by hand
        pre = c.W @ x + c.W_in @ c.scale_inputs(u)[t] + c.W_feedb @ c.scale_teacher(d)[t - 1]
        x = 0.75 * x + 0.25 * np.tanh(pre)
        assert np.allclose(c._ext_states[t, :20], x, rtol=0, atol=1e-15)
