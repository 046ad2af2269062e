#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (where /root/reference exists):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's own ``libs/pyESN.py``, ``libs/helper_mimo_esn_generic.py``
and ``libs/HelpFunc.py`` (nothing is copied), drives them with seeded synthetic
inputs and stores inputs + the reference's outputs as small .npz fixtures.
Weights are not stored: they are re-drawn from the seed (RandomState draw order
is part of the contract) and only pinned by digest + corner samples.

Cases:
  tiny   N_res=8    n_in=3  n_out=2  T=12
  c2     N_res=100  SISO  n_in=2  n_out=2  T=512  transient 0  (fit/predict continuation=True)
  c3     N_res=100  2x2   N=512 CP=7 d=3  (helper)
  c4     N_res=512  4x8   N=128 CP=7 d=3  (helper + 16-frame detection batch)
  c4s    N_res=300  4x8   N=128           (driver default reservoir)
  c5     N_res=2048 4x8   N=128           (helper, W_out only + predictions)
  mackey N_res=100  1 -> 1  2000-step teacher-forced fit on a constant input + 2000-step free run
         (BASELINE configs[0]; the series comes from oracle/mackey_glass.py and is stored)
  scan   N_res=100  2x2   N=128  trainMIMOESN_generic with DelayFlag=1 (delay scan, helper:66-81)
  legacy N_res=100  2x2   N=128  HelpFunc.trainMIMOESN (7 fit+predict, forced d=3; HelpFunc.py:64-187)
  driver_funcs  the top-level functions of the reference's driver SCRIPTS (hard-bit decision, per-tone MMSE/ZF
         equalisers, LLR demapper, TDL-B taps, logistic-regression detector, encoder): the scripts run a whole sweep
         at import, so only their `def`s (and the `_TDLB_*` tables) are compiled, via `ast` (driver_functions())
"""
import hashlib
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/libs")

import numpy as np  # noqa: E402

import pyESN as ref_pyesn  # noqa: E402  (the reference)
from helper_mimo_esn_generic import trainMIMOESN_generic as ref_train  # noqa: E402
from HelpFunc import HelpFunc as RefHelp  # noqa: E402

from oracle.ofdm_frames import LinkConfig, make_frame, tdlb_mimo_taps, exp_pdp_taps  # noqa: E402
from oracle.mackey_glass import mackey_glass  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def weight_pins(esn):
    out = {}
    for name in ("W", "W_in", "W_feedb"):
        a = getattr(esn, name)
        out[name + "_sha"] = digest(a)
        out[name + "_head"] = a.ravel()[:4].copy()
        out[name + "_tail"] = a.ravel()[-4:].copy()
    out["rho"] = np.max(np.abs(np.linalg.eigvals(esn.W)))
    return out


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"{name}.npz  {os.path.getsize(path)/1024:.1f} KiB")


def make_ref(n_in, n_out, n_res, seed, noise, **kw):
    return ref_pyesn.ESN(n_inputs=n_in, n_outputs=n_out, n_reservoir=n_res,
                         noise=noise, random_state=seed, **kw)


def case_plain(name, n_in, n_out, n_res, t_len, seed, transient, kw, t_pred=None):
    """G1-G5: init digests, fit (noise 0 and 0.001), predict both continuation modes."""
    rs = np.random.RandomState(seed + 7)
    u = rs.randn(t_len, n_in)
    d = np.tanh(rs.randn(t_len, n_out) * 0.5)
    u2 = rs.randn(t_pred or t_len, n_in)
    out = dict(seed=seed, transient=transient, u=u, d=d, u2=u2)
    for tag, noise in (("n0", 0.0), ("n1", 0.001)):
        esn = make_ref(n_in, n_out, n_res, seed, noise, **kw)
        if tag == "n0":
            out.update(weight_pins(esn))
        out[tag + "_pred_train"] = esn.fit(u, d, transient)
        out[tag + "_W_out"] = esn.W_out.copy()
        out[tag + "_laststate"] = esn.laststate.copy()
        out[tag + "_lastoutput"] = esn.lastoutput.copy()
        out[tag + "_pred_cont"] = esn.predict(u2, 0, continuation=True)
        out[tag + "_pred_fresh"] = esn.predict(u2, transient, continuation=False)
    save(name, **out)


def case_helper(name, cfg, n_res, seed, ebno_db, channel_kind, n_frames, store_frames=True):
    """G6/G7: trainMIMOESN_generic on a synthetic pilot, then predict on data frames."""
    rs = np.random.RandomState(seed + 11)
    if channel_kind == "tdlb":
        taps = tdlb_mimo_taps(cfg, seed + 1234)
    else:
        taps = exp_pdp_taps(cfg, rs)
    pilot = make_frame(cfg, ebno_db, taps, rs)
    n_in, n_out = 2 * cfg.n_r, 2 * cfg.n_t
    kw = dict(spectral_radius=0.9, sparsity=0.1,
              input_shift=np.zeros(n_in),
              input_scaling=cfg.input_scaling(ebno_db) * np.ones(n_in),
              teacher_scaling=cfg.teacher_scale * np.ones(n_out),
              teacher_shift=np.zeros(n_out),
              feedback_scaling=np.zeros(n_out))
    out = dict(seed=seed, ebno_db=ebno_db, taps=taps,
               pilot_y=pilot["y_cp"], pilot_x=pilot["x_cp"])
    for tag, noise in (("n0", 0.0), ("n1", 0.001)):
        esn = make_ref(n_in, n_out, n_res, seed, noise, **kw)
        if tag == "n0":
            out.update(weight_pins(esn))
        ret = ref_train(esn, 0, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub,
                        cfg.n_t, cfg.n_r, cfg.isi, pilot["y_cp"], pilot["x_cp"])
        esn_in, esn_out, esn, delay, d_idx, d_min, d_max, forget, nmse = ret
        out[tag + "_W_out"] = esn.W_out.copy()
        out[tag + "_laststate"] = esn.laststate.copy()
        out[tag + "_nmse"] = nmse
        if tag == "n0":
            out.update(esn_in=esn_in, esn_out=esn_out, delay=delay, d_idx=d_idx,
                       d_min=d_min, d_max=d_max, forget=forget)
            # data frames through the trained (noise=0) ESN
            rs2 = np.random.RandomState(seed + 13)
            ys, bits, preds = [], [], []
            for _ in range(n_frames):
                fr = make_frame(cfg, ebno_db, taps, rs2)
                u = np.zeros((cfg.n_sub + d_max + cfg.cp, n_in))
                for rx in range(cfg.n_r):
                    u[:, 2 * rx] = np.r_[fr["y_cp"][:, rx].real, np.zeros(d_max)]
                    u[:, 2 * rx + 1] = np.r_[fr["y_cp"][:, rx].imag, np.zeros(d_max)]
                preds.append(esn.predict(u, forget, continuation=False))
                ys.append(fr["y_cp"])
                bits.append(fr["bits"])
            if store_frames:
                out["data_y"] = np.array(ys)
            out["data_bits"] = np.packbits(np.array(bits).astype(np.uint8))
            out["data_bits_shape"] = np.array(np.array(bits).shape)
            out["data_pred"] = np.array(preds)
            out["data_seed"] = seed + 13
    save(name, **out)


def case_constellation():
    out = {}
    for m in (2, 4, 6):
        out[f"qam{m}"] = RefHelp.UnitQamConstellation(m)
    save("constellation", **out)


def case_misc():
    """G5/G8: teacher_forcing=False, scalar scalings, 1-D input, error cases."""
    rs = np.random.RandomState(5)
    u = rs.randn(40)
    d = np.sin(np.arange(40) * 0.3)
    out = dict(u=u, d=d)
    esn = ref_pyesn.ESN(1, 1, n_reservoir=20, spectral_radius=0.8, sparsity=0.2, noise=0.0,
                        input_scaling=0.5, input_shift=0.1, teacher_scaling=0.7,
                        teacher_shift=-0.2, teacher_forcing=False, random_state=99)
    out["nofb_pred_train"] = esn.fit(u, d, 3)
    out["nofb_W_out"] = esn.W_out.copy()
    out["nofb_pred"] = esn.predict(u[:17], 2, continuation=True)
    esn = ref_pyesn.ESN(1, 1, n_reservoir=20, spectral_radius=1.1, noise=0.0, random_state=3)
    out["plain_pred_train"] = esn.fit(u, d)
    out["plain_pred"] = esn.predict(u[:9])
    # correct_dimensions error cases
    errs = []
    for bad in ([1.0, 2.0], np.zeros((2, 2))):
        try:
            ref_pyesn.ESN(3, 1, input_scaling=bad)
            errs.append("")
        except ValueError as e:
            errs.append(str(e))
    out["err_msgs"] = np.array(errs)
    out["cd_scalar"] = ref_pyesn.correct_dimensions(2.5, 4)
    save("misc", **out)


def case_mackey():
    """BASELINE configs[0]: the upstream pyESN demo the reference's results/mecky_glass.png.png
    shows, at N_res=100 -- fit(ones, series) then predict(ones) with continuation=True, i.e. a free
    run in which W_feedb*y is O(1).  Default noise (0.001) and noise=0."""
    train, future = 2000, 2000
    data = mackey_glass(train + future)
    out = dict(series=data, trainlen=train, future=future, n_res=100, rho=1.5, seed=42)
    for tag, noise in (("n1", 0.001), ("n0", 0.0)):
        esn = ref_pyesn.ESN(n_inputs=1, n_outputs=1, n_reservoir=100, spectral_radius=1.5,
                            noise=noise, random_state=42)
        if tag == "n1":
            out.update(weight_pins(esn))
        out[tag + "_pred_train"] = esn.fit(np.ones(train), data[:train])
        out[tag + "_W_out"] = esn.W_out.copy()
        out[tag + "_laststate"] = esn.laststate.copy()
        out[tag + "_lastoutput"] = esn.lastoutput.copy()
        out[tag + "_free_run"] = esn.predict(np.ones(future))
    save("mackey", **out)


def _small_2x2(seed, ebno_db=12):
    cfg = LinkConfig(n_t=2, n_r=2, n_sub=128)
    rs = np.random.RandomState(seed + 11)
    taps = exp_pdp_taps(cfg, rs)
    pilot = make_frame(cfg, ebno_db, taps, rs)
    n_in, n_out = 2 * cfg.n_r, 2 * cfg.n_t
    kw = dict(spectral_radius=0.9, sparsity=0.1, input_shift=np.zeros(n_in),
              input_scaling=cfg.input_scaling(ebno_db) * np.ones(n_in),
              teacher_scaling=cfg.teacher_scale * np.ones(n_out), teacher_shift=np.zeros(n_out),
              feedback_scaling=np.zeros(n_out))
    return cfg, pilot, n_in, n_out, kw


def case_scan():
    """trainMIMOESN_generic with DelayFlag=1: fit+predict for every d in [Min, Max], keep the
    lowest (mis-aligned) NMSE, final fit (helper:66-84)."""
    seed, ebno = 41, 12
    cfg, pilot, n_in, n_out, kw = _small_2x2(seed, ebno)
    out = dict(seed=seed, ebno_db=ebno, pilot_y=pilot["y_cp"], pilot_x=pilot["x_cp"])
    for tag, noise in (("n0", 0.0), ("n1", 0.001)):
        esn = make_ref(n_in, n_out, 100, seed, noise, **kw)
        ret = ref_train(esn, 1, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub, cfg.n_t, cfg.n_r,
                        cfg.isi, pilot["y_cp"], pilot["x_cp"])
        esn_in, esn_out, esn, delay, d_idx, d_min, d_max, forget, nmse = ret
        out.update({tag + "_esn_in": esn_in, tag + "_esn_out": esn_out, tag + "_delay": delay,
                    tag + "_d_idx": d_idx, tag + "_d_min": d_min, tag + "_d_max": d_max,
                    tag + "_forget": forget, tag + "_nmse": nmse, tag + "_W_out": esn.W_out.copy(),
                    tag + "_laststate": esn.laststate.copy()})
    save("scan", **out)


def case_legacy():
    """HelpFunc.trainMIMOESN (2x2 only): scans d = 0..Max with fit+predict each, prints the NMSE
    vector, forces Delay_Idx = 3 and fits again (HelpFunc.py:101-187)."""
    import contextlib
    import io
    seed, ebno = 43, 12
    cfg, pilot, n_in, n_out, kw = _small_2x2(seed, ebno)
    out = dict(seed=seed, ebno_db=ebno, pilot_y=pilot["y_cp"], pilot_x=pilot["x_cp"])
    for tag, noise in (("n0", 0.0), ("n1", 0.001)):
        esn = make_ref(n_in, n_out, 100, seed, noise, **kw)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ret = RefHelp.trainMIMOESN(esn, 0, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub, cfg.n_t,
                                       cfg.n_r, cfg.isi, pilot["y_cp"], pilot["x_cp"])
        esn_in, esn_out, esn, delay, d_idx, d_min, d_max, forget, nmse = ret
        out.update({tag + "_esn_in": esn_in, tag + "_esn_out": esn_out, tag + "_delay": delay,
                    tag + "_d_idx": d_idx, tag + "_d_min": d_min, tag + "_d_max": d_max,
                    tag + "_forget": forget, tag + "_nmse": nmse, tag + "_W_out": esn.W_out.copy(),
                    tag + "_laststate": esn.laststate.copy(),
                    tag + "_printed_values": np.array(buf.getvalue().replace("[", " ").replace("]", " ").split(),
                                                      dtype=float)})
    # the DelayFlag != 0 branch of the legacy trainer is broken (np.zeros(shape, 1): HelpFunc.py:76)
    try:
        RefHelp.trainMIMOESN(make_ref(n_in, n_out, 100, seed, 0.0, **kw), 1, cfg.min_delay, cfg.max_delay, cfg.cp,
                             cfg.n_sub, cfg.n_t, cfg.n_r, cfg.isi, pilot["y_cp"], pilot["x_cp"])
        out["flag1_error"] = np.array("")
    except Exception as e:          # noqa: BLE001
        out["flag1_error"] = np.array(type(e).__name__)
    save("legacy", **out)


# ------------------------------------------------------------------------------------------------
# Driver-side helper functions (round 3).  The driver scripts run a whole simulation at import and
# need pyldpc, so they are NOT imported: the file is parsed with `ast`, and only its top-level
# `def` nodes and the two `_TDLB_*` tables are compiled into a namespace holding numpy / math /
# scipy.sparse.  The simulation body never runs and nothing is copied into the repo: the fixture
# holds seeded inputs and the reference functions' outputs.
DRIVERS = {
    "v2": "/root/reference/system_model_2/Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py",       # :17-119, :125-177
    "nbf": "/root/reference/system_model_2/OFDM_MIMO_2-2_NBF_LDPC.py",                 # :22-111
    "siso": "/root/reference/system_model_2/Demo_SISO_QPSK_AWGN_LDPC_ESN_with_ZF_LS.py",  # :15-95
}


def driver_functions(path):
    import ast
    import math
    import scipy.sparse as sp
    tree = ast.parse(open(path).read(), filename=path)
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            keep.append(node)
        elif isinstance(node, ast.Assign) and all(isinstance(t, ast.Name) and t.id.startswith("_TDLB_")
                                                  for t in node.targets):
            keep.append(node)
    ns = {"np": np, "math": math, "sp": sp}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


def case_driver_funcs():
    """Pins for the detector tail, the LLR leg, the linear equalisers and the TDL-B tap recipe:
    constellation + bit labels, hard decisions, ESN output reconstruction, MMSE / ZF per-subcarrier
    solves, max-log LLRs, decision-directed sigma^2, logistic calibration, TDL-B impulse responses."""
    rs = np.random.RandomState(20260)
    out = {}
    for tag, path in DRIVERS.items():
        f = driver_functions(path)
        for m in (2, 4, 6):
            const = f["unit_qam_constellation"](m)
            out[f"{tag}_qam{m}"] = const
            out[f"{tag}_labels{m}"] = f["qam_bit_labels"](2 ** m, m)
        for m in (2, 4):
            const = np.array(f["unit_qam_constellation"](m)).astype(complex)
            labels = f["qam_bit_labels"](2 ** m, m)
            n, n_t = 64, (1 if tag == "siso" else 3)
            # symbols = constellation points + noise, some far outside the grid, a few exact ties avoided
            idx = rs.randint(0, 2 ** m, size=(n, n_t))
            x = const[idx] + 0.35 * (rs.randn(n, n_t) + 1j * rs.randn(n, n_t))
            x[::7] *= 3.0
            out[f"{tag}_hard{m}_x"] = x
            if tag == "v2":
                out[f"{tag}_hard{m}_bits"] = f["hard_bits_from_syms"](x, const, m)
            elif tag == "nbf":
                out[f"{tag}_hard{m}_bits"] = f["hard_bits_from_syms"](x, const, m, None)
            else:
                out[f"{tag}_hard{m}_bits"] = f["hard_bits_from_syms"](x[:, 0], const, m, None)
            z = x[:, 0].copy()
            s2 = f["est_sigma2_from_decision"](z, const)
            out[f"{tag}_sigma2_{m}"] = np.float64(s2)
            out[f"{tag}_llr{m}"] = np.asarray(f["qam_llrs_maxlog"](z, const, labels, s2)).reshape(n, m)
            out[f"{tag}_llr{m}_tiny_sigma"] = np.asarray(f["qam_llrs_maxlog"](z[:4], const, labels, 0.0)).reshape(4, m)
            if tag != "siso":
                # one frame the way the drivers do it (v2 :459-463): sigma^2 = mean over tx of the per-column
                # estimates, every column's LLRs scaled by that mean
                s2f = np.mean([f["est_sigma2_from_decision"](x[:, tx], const) for tx in range(n_t)])
                out[f"{tag}_frame_sigma2_{m}"] = np.float64(s2f)
                out[f"{tag}_frame_llr{m}"] = np.stack(
                    [np.asarray(f["qam_llrs_maxlog"](x[:, tx], const, labels, s2f)).reshape(n, m) for tx in range(n_t)],
                    axis=2)                                                     # (N, m, N_t)
        # linear equalisers
        if tag == "siso":
            yk = rs.randn(16) + 1j * rs.randn(16)
            hk = rs.randn(16) + 1j * rs.randn(16)
            out[f"{tag}_eq_y"], out[f"{tag}_eq_h"] = yk, hk
            out[f"{tag}_eq_zf"] = np.array([f["equalize_zf"](yk[i], hk[i], 0.37) for i in range(16)])
            out[f"{tag}_eq_ls"] = np.array([f["equalize_ls"](yk[i], hk[i], 0.37) for i in range(16)])
            out[f"{tag}_eq_mmse"] = np.array([f["equalize_mmse"](yk[i], hk[i], 0.37, 0.02) for i in range(16)])
        else:
            for (n_r, n_t) in ((8, 4), (2, 2)):
                k = 16
                hk = (rs.randn(k, n_r, n_t) + 1j * rs.randn(k, n_r, n_t)) / np.sqrt(2)
                yk = rs.randn(k, n_r) + 1j * rs.randn(k, n_r)
                key = f"{tag}_eq{n_r}x{n_t}"
                out[key + "_h"], out[key + "_y"] = hk, yk
                out[key + "_mmse"] = np.array([f["equalize_mmse"](yk[i], hk[i], 0.0123, 0.004) for i in range(k)])
                out[key + "_zf"] = np.array([f["equalize_zf"](yk[i], hk[i], 0.0123) for i in range(k)])
        # ESN output reconstruction: a common delay (what every driver uses) and per-column delays
        if tag != "siso":
            n, n_t = 16, 3
            y = rs.randn(n + 9, 2 * n_t)
            for name, delay, dmin in (("common", np.full(2 * n_t, 3), 3), ("ragged", np.array([2, 4, 3, 3, 5, 2]), 2)):
                xs = f["reconstruct_esn_outputs_generic"](y, delay, dmin, n, n_t)
                out[f"{tag}_recon_{name}"] = np.array(xs)
                out[f"{tag}_recon_{name}_delay"] = delay
                out[f"{tag}_recon_{name}_dmin"] = dmin
            out[f"{tag}_recon_y"] = y
            # the block-fading drivers ask for N+1 rows of an N-row array (OFDM_MIMO_2-2_NBF_LDPC.py:56-64)
            xs = f["reconstruct_esn_outputs_generic"](y[:n], np.full(2 * n_t, 3), 3, n, n_t)
            out[f"{tag}_recon_short"] = np.array(xs)
    f = driver_functions(DRIVERS["v2"])
    # logistic LLR calibration at the call site's arguments (driver :519-520) and at the defaults
    x = 4.0 * rs.randn(600)
    y = (rs.rand(600) < 1.0 / (1.0 + np.exp(0.8 * x - 0.2))).astype(float)
    out["v2_logreg_x"], out["v2_logreg_y"] = x, y
    out["v2_logreg_call"] = np.array(f["fit_logreg_1d"](x, y, maxiter=400, lr=0.1, l2=1e-3))
    out["v2_logreg_default"] = np.array(f["fit_logreg_1d"](x, y))
    out["v2_sigmoid"] = f["sigmoid"](np.linspace(-30, 30, 13))
    # TDL-B taps: the driver's call (:321) at two seeds, plus the standard normals it consumed
    # (per link, per path: real then imaginary) so a device generator can be fed the same gains
    for seed in (1234 + 12 + 1, 1234 + 30 + 76):
        taps = np.array(f["build_cdlb_mimo_taps"](8, 4, 8, 2 * 1.024e6, 300.0, seed=seed))
        out[f"v2_taps_{seed}"] = taps
        out[f"v2_taps_{seed}_normals"] = np.random.default_rng(seed).standard_normal((8, 4, 23, 2))
    out["v2_taps_2x2_ds1000"] = np.array(f["build_cdlb_mimo_taps"](2, 2, 8, 2 * 1.024e6, 1000.0, seed=5))
    out["v2_taps_2x2_ds1000_normals"] = np.random.default_rng(5).standard_normal((2, 2, 23, 2))
    out["v2_tdlb_delays"], out["v2_tdlb_pow_db"] = f["_TDLB_NORM_DELAYS"], f["_TDLB_POW_DB"]
    # the encoder call (driver :90-93) with a dense and a sparse generator
    import scipy.sparse as sp
    g = (rs.rand(24, 12) < 0.3).astype(np.int64)
    u = (rs.rand(12) < 0.5).astype(np.int64)
    out["v2_enc_g"], out["v2_enc_u"] = g, u
    out["v2_enc_dense"] = f["ldpc_encode_bits"](g, u)
    out["v2_enc_sparse"] = f["ldpc_encode_bits"](sp.csr_matrix(g), u)
    save("driver_funcs", **out)


def main():
    only = set(sys.argv[1:])          # e.g. `make_golden.py mackey scan` regenerates just those
    if only:
        for name in only:
            globals()["case_" + name]()
        return
    case_constellation()
    case_misc()
    case_mackey()
    case_scan()
    case_legacy()
    case_driver_funcs()
    case_plain("tiny", 3, 2, 8, 12, seed=42, transient=2,
               kw=dict(spectral_radius=0.9, sparsity=0.25, input_scaling=[0.3, 0.2, 0.1],
                       input_shift=[0.0, 0.1, -0.1], teacher_scaling=0.5, teacher_shift=0.05))
    case_plain("c2", 2, 2, 100, 512, seed=7, transient=0,
               kw=dict(spectral_radius=0.9, sparsity=0.1, input_scaling=0.05 * np.ones(2),
                       input_shift=np.zeros(2), teacher_scaling=5e-3 * np.ones(2),
                       teacher_shift=np.zeros(2)), t_pred=64)
    c3 = LinkConfig(n_t=2, n_r=2, n_sub=512)
    case_helper("c3", c3, 100, seed=21, ebno_db=12, channel_kind="exp", n_frames=2)
    c4 = LinkConfig()
    case_helper("c4", c4, 512, seed=31, ebno_db=12, channel_kind="tdlb", n_frames=16)
    case_helper("c4s", c4, 300, seed=33, ebno_db=18, channel_kind="tdlb", n_frames=2)
    case_helper("c5", c4, 2048, seed=35, ebno_db=12, channel_kind="tdlb", n_frames=1)


if __name__ == "__main__":
    main()
