"""Self-consistency of the LDPC restatement (oracle/ldpc_oracle.py): pyldpc is absent, so parity is
unpinned; these checks pin the algebra (H c = 0, regularity, rank) and that the sum-product decoder
corrects what a (4,8)-regular code of this length should."""
import numpy as np

from oracle import ldpc_oracle as lo


def test_gallager_code_algebra():
    rng = np.random.RandomState(1)
    H = lo.gallager_parity_check(512, 4, 8, rng)
    assert H.shape == (256, 512)
    assert np.all(H.sum(0) == 4) and np.all(H.sum(1) == 8)
    H_sys, P, k, order = lo.systematic_code(H)
    assert k == 512 - (256 - 3)                      # d_v - 1 dependent rows
    u = (rng.rand(50, k) > 0.5).astype(np.uint8)
    c = lo.encode(P, u)
    assert c.shape == (50, 512)
    assert not ((c.astype(int) @ H_sys.T.astype(int)) % 2).any()
    np.testing.assert_array_equal(c[:, :k], u)       # systematic: message = first k bits


def test_bp_decoder_corrects_awgn():
    rng = np.random.RandomState(2)
    H = lo.gallager_parity_check(512, 4, 8, rng)
    H_sys, P, k, _ = lo.systematic_code(H)
    u = (rng.rand(40, k) > 0.5).astype(np.uint8)
    c = lo.encode(P, u)
    snr_db = 3.0                                     # Es/N0 of the BPSK observations
    var = 10 ** (-snr_db / 10)
    y = (1.0 - 2.0 * c) + np.sqrt(var) * rng.randn(*c.shape)
    raw_ber = np.mean((y < 0) != (c == 1))
    dec = lo.decode_bp(H_sys, y, snr_db, maxiter=50)
    dec_ber = np.mean(dec[:, :k] != u)
    assert raw_ber > 0.03 and dec_ber < raw_ber / 20, (raw_ber, dec_ber)
    # noiseless: decodes exactly in one sweep
    np.testing.assert_array_equal(lo.decode_bp(H_sys, 1.0 - 2.0 * c, 10.0, 5), c)
