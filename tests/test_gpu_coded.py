"""Coded leg on the GPU against oracle/ldpc_oracle.py (pyldpc restated; parity
unpinned -- the package is absent): encoder bit-exact, LLRs 1e-12, calibration 1e-9, sum-product
decisions identical on the same graph and observations."""
import numpy as np
import pytest

from oracle import esn_oracle as eo
from oracle import ldpc_oracle as lo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def code():
    from esn_ofdm_mimo_amd.coded import LdpcCode
    return LdpcCode(512, 4, 8, seed=5)


def test_code_algebra_and_encoder(code):
    import torch
    assert code.H.shape == (256, 512) and code.k == 259
    assert np.all(code.H.sum(0) == 4) and np.all(code.H.sum(1) == 8)
    rs = np.random.RandomState(0)
    u = (rs.rand(6, 4, code.k) > 0.5).astype(np.uint8)
    bits = code.encode(torch.as_tensor(u), 4).cpu().numpy()              # [B, n, n_t]
    cw = np.transpose(bits, (0, 2, 1))                                     # [B, n_t, n]
    np.testing.assert_array_equal(cw, lo.encode(code.P, u))
    assert not ((cw.reshape(-1, 512).astype(int) @ code.H.T.astype(int)) % 2).any()


def test_llrs_and_sigma2_match_oracle(code):
    import torch
    rs = np.random.RandomState(1)
    m, N, n_t, B = 4, 128, 4, 3
    const = eo.unit_qam(m)
    x = const[rs.randint(0, 16, size=(B, N, n_t))] + 0.15 * (rs.randn(B, N, n_t) + 1j * rs.randn(B, N, n_t))
    llr, s2 = code.llrs(torch.as_tensor(x, device=code.device), m)
    llr, s2 = llr.cpu().numpy(), s2.cpu().numpy()
    for b in range(B):
        want_s2 = np.mean([lo.sigma2_from_decisions(x[b, :, tx], m) for tx in range(n_t)])
        assert abs(s2[b] - want_s2) < 1e-12 * want_s2
        for tx in range(n_t):
            want = lo.qam_llrs_maxlog(x[b, :, tx], m, want_s2).reshape(-1)
            np.testing.assert_allclose(llr[b, tx], want, rtol=1e-10, atol=1e-10)


def test_calibration_matches_oracle(code):
    import torch
    rs = np.random.RandomState(2)
    m, B, n_t, N = 4, 5, 4, 128
    bits = (rs.rand(B, N * m, n_t) > 0.5).astype(np.uint8)
    llr = rs.randn(B, n_t, N * m) * 3 - (2.0 * np.transpose(bits, (0, 2, 1)) - 1.0) * 1.5
    a, b = code.fit_calibration(torch.as_tensor(llr, device=code.device), torch.as_tensor(bits, device=code.device), m)
    a, b = a.cpu().numpy(), b.cpu().numpy()
    for bit in range(m):
        xs = llr.reshape(B, n_t, N, m)[..., bit].reshape(-1)
        ys = np.transpose(bits, (0, 2, 1)).reshape(B, n_t, N, m)[..., bit].reshape(-1).astype(float)
        wa, wb = lo.fit_logreg_1d(xs, ys, maxiter=400, lr=0.1, l2=1e-3)
        assert abs(a[bit] - wa) < 1e-9 and abs(b[bit] - wb) < 1e-9
        assert wa < 0                                   # positive LLR means bit 0


def test_sum_product_decoder_matches_oracle(code):
    import torch
    rs = np.random.RandomState(3)
    n_cw = 24
    u = (rs.rand(n_cw, code.k) > 0.5).astype(np.uint8)
    c = lo.encode(code.P, u)
    # calibrated LLRs of mixed quality: some codewords decodable, some not
    sig = np.repeat(np.array([0.6, 0.8, 1.0, 1.3]), n_cw // 4)[:, None]
    llr = 2.0 * ((1.0 - 2.0 * c) + sig * rs.randn(*c.shape)) / sig ** 2
    a = torch.full((4,), -1.0, dtype=torch.float64, device=code.device)       # -(a x + b) = x
    b = torch.zeros(4, dtype=torch.float64, device=code.device)
    x_llr = torch.as_tensor(llr.reshape(n_cw // 4, 4, 512), device=code.device)
    err, nb, xo = code.decode_count(x_llr, a, b, torch.as_tensor(u), cw_per_group=4, bits_per_sym=4,
                                    maxiter=30, want_bits=True)
    xo = xo.cpu().numpy()
    yobs = 0.5 * np.clip(llr, -20, 20)
    want = lo.decode_bp(code.H, yobs, 1.0, 30)
    same = (xo == want).all(axis=1)
    assert same.mean() >= 0.95, same
    conv = ~((want.astype(int) @ code.H.T.astype(int)) % 2).any(axis=1)
    np.testing.assert_array_equal(xo[conv], want[conv])                       # converged words: identical
    want_err = np.array([(want[g * 4:(g + 1) * 4, :code.k] != u[g * 4:(g + 1) * 4]).sum() for g in range(n_cw // 4)])
    got_err = err.cpu().numpy()
    assert np.abs(got_err - want_err).max() <= 0.02 * code.k * 4              # non-converged words may differ in a few bits
    np.testing.assert_array_equal(nb.cpu().numpy(), [4 * code.k] * (n_cw // 4))
    assert 6 <= conv.sum() < n_cw                                             # the test exercises both regimes
