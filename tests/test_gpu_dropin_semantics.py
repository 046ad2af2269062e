"""GPU parity, round 2: the reference-generated fixtures that round 1 only checked against the CPU
oracle now run through the HIP drop-ins (``pyESN.ESN``, ``trainMIMOESN_generic``, ``HelpFunc``):

  misc.npz    1-D inputs, scalar scalings, teacher_forcing=False, ValueError texts   (pyESN.py:4-24,168-171)
  mackey.npz  BASELINE configs[0]: 2000-step fit on a constant input + 2000-step free run (:154-255)
  scan.npz    trainMIMOESN_generic with DelayFlag=1 (helper_mimo_esn_generic.py:66-84)
  legacy.npz  HelpFunc.trainMIMOESN, SURVEY row a13 (HelpFunc.py:64-187)
  c4 / c4s    the drop-in's OWN W_out and its predictions of the golden data frames, noise 0 and
              0.001 (round 1 compared W_out only through its action on the training rows)

Tolerances: float64 kernels vs the reference's float64 NumPy -- 1e-8 relative on outputs; W_out
through the QR solve vs LAPACK pinv -- cond(E) * 1e-13, written per case.
"""
import numpy as np
import pytest

from oracle import esn_oracle as eo
from oracle.ofdm_frames import LinkConfig

pytestmark = pytest.mark.gpu


def rel_err(got, want):
    return float(np.max(np.abs(np.asarray(got) - np.asarray(want))) / (np.max(np.abs(want)) + 1e-300))


@pytest.fixture(scope="module")
def mods():
    from esn_ofdm_mimo_amd import pyESN, helper_mimo_esn_generic
    from esn_ofdm_mimo_amd.HelpFunc import HelpFunc
    return pyESN, helper_mimo_esn_generic, HelpFunc


def test_misc_fixture_through_hip_dropin(mods, golden):
    pyESN = mods[0]
    g = golden("misc")
    u, d = g["u"], g["d"]                                   # both 1-D (pyESN.py:168-171, :230-231)
    esn = pyESN.ESN(1, 1, n_reservoir=20, spectral_radius=0.8, sparsity=0.2, noise=0.0,
                    input_scaling=0.5, input_shift=0.1, teacher_scaling=0.7,
                    teacher_shift=-0.2, teacher_forcing=False, random_state=99)
    np.testing.assert_array_equal(esn.input_scaling, [0.5])           # scalar -> vector (pyESN.py:15-17)
    pred_train = esn.fit(u, d, 3)
    assert pred_train.shape == g["nofb_pred_train"].shape == (40, 1)
    assert rel_err(pred_train, g["nofb_pred_train"]) < 1e-8
    assert rel_err(esn.W_out, g["nofb_W_out"]) < 1e-7
    esn.W_out = g["nofb_W_out"]
    got = esn.predict(u[:17], 2, continuation=True)
    assert got.shape == (15, 1) and rel_err(got, g["nofb_pred"]) < 1e-9
    # defaults: no scalings at all, transient 0, continuation=True, rho > 1
    esn = pyESN.ESN(1, 1, n_reservoir=20, spectral_radius=1.1, noise=0.0, random_state=3)
    assert esn.input_scaling is None and esn.teacher_scaling is None
    assert rel_err(esn.fit(u, d), g["plain_pred_train"]) < 1e-6      # 40 x 21 system, cond 1e6 at noise 0
    assert rel_err(esn.predict(u[:9]), g["plain_pred"]) < 1e-5
    # error behaviour of correct_dimensions (pyESN.py:21,23): the reference's own texts
    np.testing.assert_array_equal(pyESN.correct_dimensions(2.5, 4), g["cd_scalar"])
    assert pyESN.correct_dimensions(None, 3) is None
    with pytest.raises(ValueError) as e0:
        pyESN.ESN(3, 1, input_scaling=[1.0, 2.0])
    assert str(e0.value) == str(g["err_msgs"][0])
    with pytest.raises(ValueError) as e1:
        pyESN.ESN(3, 1, input_scaling=np.zeros((2, 2)))
    assert str(e1.value) == str(g["err_msgs"][1])
    with pytest.raises(Exception, match="Invalid seed"):
        pyESN.ESN(2, 1, n_reservoir=5, random_state="not a seed")        # pyESN.py:83-85
    with pytest.raises(AttributeError):
        pyESN.ESN(2, 1, n_reservoir=5, random_state=1).predict(np.zeros((3, 2)), continuation=False)


def test_falsy_seed_selects_the_global_rng(mods):
    """random_state=0 is falsy -> the process-global NumPy RNG, not seed 0 (pyESN.py:81-87)."""
    pyESN = mods[0]
    np.random.seed(2468)
    esn = pyESN.ESN(2, 1, n_reservoir=12, sparsity=0.3, random_state=0, noise=0.0)
    assert esn.random_state_ is np.random.mtrand._rand
    np.random.seed(2468)
    w, w_in, w_fb = eo.draw_weights(np.random.mtrand._rand, 2, 1, 12, 0.95, 0.3)
    np.testing.assert_array_equal(esn.W, w)
    np.testing.assert_array_equal(esn.W_in, w_in)
    np.testing.assert_array_equal(esn.W_feedb, w_fb)
    # ... and the state noise of fit comes from that same global stream (pyESN.py:125)
    esn.noise = 0.01
    rs = np.random.RandomState(1)
    u, d = rs.randn(9, 2), rs.randn(9, 1)
    np.random.seed(99)
    esn.fit(u, d)
    o = eo.OracleESN(2, 1, n_reservoir=12, noise=0.01, random_state=np.random.RandomState(5))
    o.W, o.W_in, o.W_feedb = w, w_in, w_fb
    np.random.seed(99)
    o.rng = np.random.mtrand._rand
    o.fit(u, d)
    assert rel_err(esn.laststate, o.laststate) < 1e-12


@pytest.mark.parametrize("tag,noise", [("n1", 0.001), ("n0", 0.0)])
def test_mackey_glass_through_f64_dropin(mods, golden, tag, noise):
    """BASELINE configs[0] on the float64 kernels: fit(ones, series) / predict(ones) call for call."""
    pyESN = mods[0]
    g = golden("mackey")
    n_train, n_free = int(g["trainlen"]), int(g["future"])
    series = g["series"]
    esn = pyESN.ESN(n_inputs=1, n_outputs=1, n_reservoir=int(g["n_res"]), spectral_radius=1.5, noise=noise,
                    random_state=int(g["seed"]))
    pred_train = esn.fit(np.ones(n_train), series[:n_train])
    assert esn.fit_status == 0
    assert rel_err(esn.laststate, g[tag + "_laststate"]) < 1e-10
    np.testing.assert_allclose(esn.lastoutput, g[tag + "_lastoutput"], rtol=1e-13)
    assert rel_err(pred_train, g[tag + "_pred_train"]) < (1e-8 if noise else 1e-6)
    # 2000 x 101 system: cond 2.6e4 with the default noise, 2.6e6 without
    assert rel_err(esn.W_out, g[tag + "_W_out"]) < (1e-8 if noise else 1e-5)
    free = esn.predict(np.ones(n_free))                     # continuation=True: O(1) output feedback
    assert free.shape == (n_free, 1)
    if noise:
        # stable limit cycle (a 1e-14 perturbation stays 1e-14 over the run): whole run compared
        assert np.max(np.abs(free - g["n1_free_run"])) < 1e-6
        assert np.sqrt(np.mean((free.ravel()[:500] - series[n_train:n_train + 500]) ** 2)) < 0.15
        # same run with the reference's W_out: op-for-op, 2000 dependent steps
        esn2 = pyESN.ESN(n_inputs=1, n_outputs=1, n_reservoir=int(g["n_res"]), spectral_radius=1.5, noise=noise,
                         random_state=int(g["seed"]))
        esn2.fit(np.ones(n_train), series[:n_train])
        esn2.W_out = g["n1_W_out"]
        assert np.max(np.abs(esn2.predict(np.ones(n_free)) - g["n1_free_run"])) < 1e-9
    else:
        esn.W_out = g["n0_W_out"]                           # chaotic at noise 0: first 100 steps only
        assert np.max(np.abs(esn.predict(np.ones(n_free))[:100] - g["n0_free_run"][:100])) < 1e-6


def _esn_2x2(pyESN, g, noise):
    cfg = LinkConfig(n_t=2, n_r=2, n_sub=128)
    n_in, n_out = 4, 4
    ebno = float(g["ebno_db"])
    return cfg, pyESN.ESN(n_in, n_out, 100, spectral_radius=0.9, sparsity=0.1, noise=noise,
                          input_shift=np.zeros(n_in), input_scaling=cfg.input_scaling(ebno) * np.ones(n_in),
                          teacher_scaling=cfg.teacher_scale * np.ones(n_out), teacher_shift=np.zeros(n_out),
                          feedback_scaling=np.zeros(n_out), random_state=int(g["seed"]))


@pytest.mark.parametrize("tag,noise", [("n0", 0.0), ("n1", 0.001)])
def test_helper_delay_scan_through_hip(mods, golden, tag, noise):
    """DelayFlag=1 (helper:66-84): 7 x (fit, predict) + final fit, the RandomState consumed in order."""
    pyESN, helper, _ = mods
    g = golden("scan")
    cfg, esn = _esn_2x2(pyESN, g, noise)
    ret = helper.trainMIMOESN_generic(esn, 1, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub, cfg.n_t, cfg.n_r,
                                      cfg.isi, g["pilot_y"], g["pilot_x"])
    x_in, x_out, esn2, delay, d_idx, d_min, d_max, forget, nmse = ret
    assert esn2 is esn
    np.testing.assert_array_equal(x_in, g[tag + "_esn_in"])
    np.testing.assert_array_equal(x_out, g[tag + "_esn_out"])
    np.testing.assert_array_equal(delay, g[tag + "_delay"])
    assert (d_idx, d_min, d_max, forget) == tuple(int(g[tag + k]) for k in ("_d_idx", "_d_min", "_d_max", "_forget"))
    assert nmse == pytest.approx(float(g[tag + "_nmse"]), rel=1e-4)
    assert rel_err(esn.laststate, g[tag + "_laststate"]) < 1e-10
    assert rel_err(esn.W_out, g[tag + "_W_out"]) < (1e-6 if noise else 1e-4)


@pytest.mark.parametrize("tag,noise", [("n0", 0.0), ("n1", 0.001)])
def test_legacy_helpfunc_trainer_through_hip(mods, golden, capsys, tag, noise):
    """HelpFunc.trainMIMOESN -- 7 x (fit, predict), forced row 3, printed NMSE vector."""
    pyESN, _, HelpFunc = mods
    g = golden("legacy")
    cfg, esn = _esn_2x2(pyESN, g, noise)
    ret = HelpFunc.trainMIMOESN(esn, 0, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub, cfg.n_t, cfg.n_r,
                                cfg.isi, g["pilot_y"], g["pilot_x"])
    x_in, x_out, esn2, delay, d_idx, d_min, d_max, forget, nmse = ret
    assert esn2 is esn and len(ret) == 9
    np.testing.assert_array_equal(x_in, g[tag + "_esn_in"])
    np.testing.assert_array_equal(x_out, g[tag + "_esn_out"])
    np.testing.assert_array_equal(delay, g[tag + "_delay"])
    assert delay.dtype == g[tag + "_delay"].dtype
    assert (d_idx, int(d_min), int(d_max), int(forget)) == (3, 3, 3, 10)
    assert float(nmse) == pytest.approx(float(g[tag + "_nmse"]), rel=1e-4)
    printed = np.array(capsys.readouterr().out.replace("[", " ").replace("]", " ").split(), dtype=float)
    np.testing.assert_allclose(printed, g[tag + "_printed_values"], rtol=1e-4)
    assert rel_err(esn.laststate, g[tag + "_laststate"]) < 1e-10
    assert rel_err(esn.W_out, g[tag + "_W_out"]) < (1e-6 if noise else 1e-4)
    with pytest.raises(TypeError):
        HelpFunc.trainMIMOESN(esn, 1, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub, cfg.n_t, cfg.n_r,
                              cfg.isi, g["pilot_y"], g["pilot_x"])
    for m in (2, 4, 6):
        np.testing.assert_allclose(HelpFunc.UnitQamConstellation(m), golden("constellation")[f"qam{m}"], atol=1e-15)


HELPER = {"c3": (LinkConfig(n_t=2, n_r=2, n_sub=512), 100, 1e-4), "c4": (LinkConfig(), 512, 1e-7),
          "c4s": (LinkConfig(), 300, 1e-7)}


@pytest.mark.parametrize("name", list(HELPER))
def test_product_trained_readout_equals_reference(mods, golden, name):
    """The drop-in's OWN W_out (QR on the GPU) against the reference's pinv W_out, and the golden
    data frames predicted with it.  c4 / c4s are 128 x 528 / 128 x 316 (under-determined: the
    minimum-norm solution is unique, cond(E) ~ 1e3); c3 is 512 x 104 fitted without state noise
    (cond ~ 1e5), hence its looser bound."""
    pyESN, helper, _ = mods
    g = golden(name)
    cfg, n_res, tol = HELPER[name]
    n_in, n_out = 2 * cfg.n_r, 2 * cfg.n_t
    ebno = float(g["ebno_db"])
    for tag, noise in (("n0", 0.0), ("n1", 0.001)):
        esn = pyESN.ESN(n_in, n_out, n_res, spectral_radius=0.9, sparsity=0.1, noise=noise,
                        input_shift=np.zeros(n_in), input_scaling=cfg.input_scaling(ebno) * np.ones(n_in),
                        teacher_scaling=cfg.teacher_scale * np.ones(n_out), teacher_shift=np.zeros(n_out),
                        feedback_scaling=np.zeros(n_out), random_state=int(g["seed"]))
        ret = helper.trainMIMOESN_generic(esn, 0, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub, cfg.n_t,
                                          cfg.n_r, cfg.isi, g["pilot_y"], g["pilot_x"])
        assert esn.fit_status == 0
        assert rel_err(esn.laststate, g[tag + "_laststate"]) < 1e-10
        assert rel_err(esn.W_out, g[tag + "_W_out"]) < (tol if tag == "n0" else 1e-7), (name, tag)
        assert ret[8] == pytest.approx(float(g[tag + "_nmse"]), rel=1e-4)
        if tag == "n0":                      # the golden frames were predicted by the noise-free reference ESN
            d_max, forget = int(g["d_max"]), int(g["forget"])
            for y_cp, want in zip(g["data_y"], g["data_pred"]):
                got = esn.predict(eo.pack_rx(y_cp, d_max), forget, continuation=False)
                assert rel_err(got, want) < 100 * tol, (name, rel_err(got, want))
