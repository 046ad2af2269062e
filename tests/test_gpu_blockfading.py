"""The block-fading drivers' comparison (BASELINE configs[2]; OFDM_MIMO_2-2_NBF_LDPC.py and its siblings) on the
GPU: LS-ZF on the interpolated LS estimate, Perfect-ZF on the true channel, the "train at a fixed Eb/No" second
ESN and the LDPC leg decoded on every 4th symbol with the drivers' uncalibrated LLRs.

Deterministic: H_LS, H_true and the ZF detector against oracle/baselines.py (whose equalize_zf / equalize_mmse are
pinned by the reference's own functions, tests/test_oracle_driver_funcs.py) on the same frames.
Statistical: one Eb/No point of the whole comparison at 2x2 -- the orderings every published block-fading plot
shows (Perfect-ZF <= LS-ZF, coding helps the linear detectors at high SNR) and the bookkeeping of the cadence."""
import numpy as np
import pytest

from oracle import esn_oracle as eo
from oracle.baselines import estimate_channel, linear_detect, pilot_frames, taps_to_freq
from oracle.ofdm_frames import LinkConfig, exp_pdp_taps, make_frame

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_t,n_r,n_sub", [(2, 2, 64), (4, 8, 128)])
def test_ls_estimate_true_channel_and_zf_match_oracle(n_t, n_r, n_sub):
    import torch
    from esn_ofdm_mimo_amd.montecarlo import FrameSource, LinkParams
    cfg = LinkConfig(n_t=n_t, n_r=n_r, n_sub=n_sub)
    fs = FrameSource(LinkParams.block_fading(n_t, n_r, n_sub), seed=1)
    rs = np.random.RandomState(n_sub + 1)
    ebno, G, F = 15.0, 3, 3
    taps = [exp_pdp_taps(cfg, rs) for _ in range(G)]
    pil = [pilot_frames(cfg, ebno, t, rs) for t in taps]
    data = [[make_frame(cfg, ebno, t, rs) for _ in range(F)] for t in taps]
    dev = fs.device
    pbits = torch.as_tensor(np.stack([p["bits"] for p in pil]).astype(np.uint8), device=dev)
    yls = torch.as_tensor(np.stack([p["y_ls_cp"] for p in pil]), device=dev)
    H_ls = fs.estimate_channel(pbits, yls, ebno, ls_only=True)
    H_true = fs.true_channel(torch.as_tensor(np.stack(taps), device=dev))
    for b in range(G):
        want = estimate_channel(cfg, ebno, pil[b]["X_LS"], pil[b]["y_ls_cp"], ls_only=True)
        np.testing.assert_allclose(H_ls[b].cpu().numpy(), want, rtol=0, atol=1e-10 * np.abs(want).max())
        np.testing.assert_allclose(H_true[b].cpu().numpy(), taps_to_freq(cfg, taps[b]), rtol=0, atol=1e-12)
    dy = torch.as_tensor(np.stack([fr["y_cp"] for d in data for fr in d]), device=dev)
    dbits = np.stack([fr["bits"] for d in data for fr in d]).astype(np.uint8)
    const = eo.unit_qam(cfg.m)
    for H in (H_ls, H_true):
        err, nb, xh = fs.mmse_detect_count(H, dy, torch.as_tensor(dbits, device=dev), F, ebno, want_xhat=True, zf=True)
        Hn, xh = H.cpu().numpy(), xh.cpu().numpy()
        want_err = np.zeros(G, dtype=np.int64)
        for i in range(G * F):
            x = linear_detect(cfg, ebno, Hn[i // F], data[i // F][i % F]["y_cp"], reg=0)
            np.testing.assert_allclose(xh[i], x, rtol=0, atol=1e-8 * np.abs(x).max())
            want_err[i // F] += eo.count_bit_errors(dbits[i], eo.hard_bits(x, const, cfg.m))
        np.testing.assert_array_equal(err.cpu().numpy(), want_err)


def test_block_fading_comparison_point_2x2():
    """configs[2] at N = 128 (the drivers' FAST size): five detectors, uncoded and coded."""
    from esn_ofdm_mimo_amd.coded import LdpcCode
    from esn_ofdm_mimo_amd.montecarlo import DetectorSweep, LinkParams, block_fading_point
    prm = LinkParams.block_fading(2, 2, 128)
    kw = dict(n_reservoir=100, noise=0.001, seed=5, precision="f16", fit_precision="f16")
    sw = DetectorSweep(prm, **kw)
    sw_fixed = DetectorSweep(prm, train_ebno=12.0, **kw)
    code = LdpcCode(prm.n_sub * prm.m, 4, 8, seed=3)
    L = prm.coherence_symbols
    res = {}
    for si, ebno in enumerate((6.0, 24.0)):
        r = block_fading_point(sw, code, ebno, si, 96, fixed_sweep=sw_fixed, seed=1)
        res[ebno] = r
        print(ebno, {k: round(v, 5) for k, v in r.items()})
        kk = np.arange(96)[:, None] * L + 2 + np.arange(L - 1)[None, :]
        assert r["decoded_symbols"] == int((kk % 4 == 1).sum())               # the every-4th-symbol cadence
        for k in ("ESN_matched", "ESN_trainFixed", "LS_ZF", "MMSE", "PerfectZF"):
            assert 0.0 <= r["BER_" + k] < 0.5 and 0.0 <= r["BERC_" + k] <= 0.5
        assert r["BER_MMSE"] <= r["BER_LS_ZF"] * 1.02                         # the refined estimate + MMSE beats LS-ZF
    # the true channel beats its estimate where estimation noise dominates; at high Eb/No the LS estimate has absorbed
    # the PA's compression into an effective channel that the true taps do not know (measured: 0.082 vs 0.080 at 24 dB)
    assert res[6.0]["BER_PerfectZF"] < res[6.0]["BER_LS_ZF"]
    assert res[24.0]["BER_PerfectZF"] < 1.1 * res[24.0]["BER_LS_ZF"]
    for k in ("ESN_matched", "LS_ZF", "MMSE", "PerfectZF"):
        assert res[24.0]["BER_" + k] < res[6.0]["BER_" + k]                    # falling with Eb/No
    assert res[24.0]["BERC_PerfectZF"] < res[24.0]["BER_PerfectZF"]           # coding pays at high SNR
    # trained at 12 dB, evaluated at 24 dB: worse than (or equal to) the SNR-matched ESN, better than at 6 dB
    assert res[24.0]["BER_ESN_trainFixed"] < res[6.0]["BER_ESN_trainFixed"]


def test_sweep_with_fixed_training_snr_runs_end_to_end():
    """DetectorSweep(train_ebno=...) through `run`: the pilot is regenerated at the fixed Eb/No over the block's taps
    and the input scaling of that Eb/No is used at train and detect time (OFDM_MIMO_2-2_NBF_LDPC.py:347-367,440-448);
    at the training Eb/No itself it coincides with the SNR-matched sweep."""
    from esn_ofdm_mimo_amd.montecarlo import DetectorSweep, LinkParams
    prm = LinkParams.block_fading(2, 2, 128)
    kw = dict(n_reservoir=100, noise=0.001, seed=5, precision="f32", fit_precision="f32")
    _, c_match = DetectorSweep(prm, **kw).run([12.0, 24.0], 32, frames_per_block=6)
    _, c_fixed = DetectorSweep(prm, train_ebno=12.0, **kw).run([12.0, 24.0], 32, frames_per_block=6)
    np.testing.assert_array_equal(c_fixed[0], c_match[0])                      # same pilot, same scaling at 12 dB
    assert c_fixed[1, 1] == c_match[1, 1] and c_fixed[1, 0] != c_match[1, 0]
