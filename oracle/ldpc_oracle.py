"""CPU oracle for the coded leg of the north-star driver -- TEST INFRASTRUCTURE ONLY.

The reference uses the third-party package ``pyldpc`` (``requirements-sm2.txt:5``: ``pyldpc>=0.8.0``,
unpinned, NOT vendored in /root/reference and not installable here), so this file restates the
package's published algorithms, anchored on the reference's own call sites:

  * ``make_ldpc(n, d_v, d_c, systematic=True, sparse=True)``  Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:250
        regular Gallager parity-check matrix (first block: d_c consecutive ones per row; the other
        d_v - 1 blocks are column permutations of it) and a systematic generator obtained by GF(2)
        elimination.  pyldpc draws the permutations from an unseeded RNG, so no two runs of the
        reference share a code: only the ensemble (n=512, d_v=4, d_c=8) is reproducible.
  * ``decode(H, y, snr, maxiter)``                              :495-496, :505-506
        flooding log-domain sum-product: var = 10^(-snr/10), Lc = 2 y / var, check update
        Lr = log((1+X)/(1-X)) with X = prod tanh(Lq/2), bit update Lq = Lc + sum Lr, hard decision
        x = (L_post <= 0), stop when H x = 0 or after maxiter sweeps.
  * ``get_message(G, d)``                                       :496, :506
        for a systematic generator the message is the first k code bits.
  * max-log LLRs, decision-directed sigma^2, logistic calibration: driver :66-88, :90-93, :108-119.

Parity: **unpinned** (the dependency is absent and the reference has no fixture at this boundary).
Statistical anchor: columns ESN_coded / MMSE_coded of the reference's results_ber.csv."""
from __future__ import annotations

import numpy as np

from .esn_oracle import bit_labels_lsb_first, unit_qam


def gallager_parity_check(n, d_v, d_c, rng):
    """Regular (d_v, d_c) Gallager matrix [n d_v / d_c, n]."""
    if n % d_c:
        raise ValueError("d_c must divide n")
    rows_per_block = n // d_c
    block = np.zeros((rows_per_block, n), dtype=np.uint8)
    for i in range(rows_per_block):
        block[i, i * d_c:(i + 1) * d_c] = 1
    blocks = [block] + [block[:, rng.permutation(n)] for _ in range(d_v - 1)]
    return np.concatenate(blocks, axis=0)


def systematic_code(H):
    """GF(2) elimination -> (H_sys, P, k): columns of H permuted so that the code is
    {c = [u ; P u mod 2]} with u the first k bits, H_sys c = 0 (dependent rows of H are kept in
    H_sys -- the decoder uses the full, regular graph)."""
    m, n = H.shape
    A = H.copy() % 2
    perm = np.arange(n)
    r = 0
    piv_cols = []
    for c in range(n):
        if r == m:
            break
        rows = np.nonzero(A[r:, c])[0]
        if rows.size == 0:
            continue
        p = r + rows[0]
        if p != r:
            A[[r, p]] = A[[p, r]]
        others = np.nonzero(A[:, c])[0]
        others = others[others != r]
        A[others] ^= A[r]
        piv_cols.append(c)
        r += 1
    rank = r
    k = n - rank
    free = [c for c in range(n) if c not in set(piv_cols)]
    order = np.array(free + piv_cols)                 # info bits first, parity bits (pivots) last
    # reduced rows: A[i, piv_cols[i]] = 1 and pivots form an identity -> parity_i = sum_free A[i, f] u_f
    P = A[:rank][:, free].astype(np.uint8)            # [rank, k]
    H_sys = H[:, order].astype(np.uint8)
    return H_sys, P, k, order


def encode(P, u):
    """c = [u ; P u mod 2] for u [..., k]."""
    par = (u.astype(np.int64) @ P.T.astype(np.int64)) % 2
    return np.concatenate([u, par], axis=-1).astype(np.uint8)


def decode_bp(H, y, snr_db, maxiter):
    """pyldpc.decode restated (flooding log-BP, float64).  y [n_cw, n] 'observations' with the BPSK
    convention bit 0 <-> +1; returns hard bits [n_cw, n]."""
    m, n = H.shape
    var = 10.0 ** (-snr_db / 10.0)
    Lc = 2.0 * np.asarray(y, dtype=np.float64) / var
    ci, vi = np.nonzero(H)                              # edges sorted by check
    n_cw = Lc.shape[0]
    Lq = Lc[:, vi].copy()
    x = (Lc <= 0).astype(np.uint8)
    done = np.zeros(n_cw, dtype=bool)
    out = x.copy()
    order_c = np.argsort(ci, kind="stable")
    assert np.all(order_c == np.arange(len(ci)))
    dc = np.bincount(ci, minlength=m)
    starts = np.concatenate([[0], np.cumsum(dc)])[:-1]
    for _ in range(maxiter):
        t = np.tanh(0.5 * Lq)
        Lr = np.empty_like(Lq)
        for e in range(len(ci)):
            s, cnt = starts[ci[e]], dc[ci[e]]
            idx = [q for q in range(s, s + cnt) if q != e]
            X = np.prod(t[:, idx], axis=1)
            num, den = 1.0 + X, 1.0 - X
            with np.errstate(divide="ignore", invalid="ignore"):
                v = np.log(num / den)
            v = np.where(num == 0, -1.0, np.where(den == 0, 1.0, v))
            Lr[:, e] = v
        tot = np.zeros((n_cw, n))
        np.add.at(tot, (slice(None), vi), Lr)
        L_post = Lc + tot
        Lq = L_post[:, vi] - Lr
        x = (L_post <= 0).astype(np.uint8)
        synd = (x.astype(np.int64) @ H.T.astype(np.int64)) % 2
        ok = ~synd.any(axis=1)
        newly = ok & ~done
        out[newly] = x[newly]
        done |= ok
        if done.all():
            break
    out[~done] = x[~done]
    return out


def qam_llrs_maxlog(z, m, sigma2):
    """LLR[n, b] = (min_{s: bit b = 1} |z-s|^2 - min_{s: bit b = 0} |z-s|^2) / max(sigma2, 1e-12)
    (positive = bit 0 more likely; driver :66-88)."""
    const = unit_qam(m)
    labels = bit_labels_lsb_first(m)
    d = np.abs(z.reshape(-1, 1) - const.reshape(1, -1)) ** 2
    out = np.zeros((z.size, m))
    for b in range(m):
        d0 = d[:, labels[:, b] == 0].min(axis=1)
        d1 = d[:, labels[:, b] == 1].min(axis=1)
        out[:, b] = (d1 - d0) / max(sigma2, 1e-12)
    return out


def sigma2_from_decisions(x_col, m):
    """mean |z - nearest point|^2 + 1e-12 (driver :90-93)."""
    const = unit_qam(m)
    idx = np.argmin(np.abs(x_col.reshape(-1, 1) - const.reshape(1, -1)) ** 2, axis=1)
    return float(np.mean(np.abs(x_col - const[idx]) ** 2) + 1e-12)


def fit_logreg_1d(x, y, maxiter=400, lr=0.1, l2=1e-3):
    """p(y=1|x) = sigmoid(a x + b) by plain gradient descent from (1, 0) (driver :108-119)."""
    a, b = 1.0, 0.0
    n = len(x)
    for _ in range(maxiter):
        p = 1.0 / (1.0 + np.exp(-(a * x + b)))
        ga = np.dot(p - y, x) / n + l2 * a
        gb = np.sum(p - y) / n
        a -= lr * ga
        b -= lr * gb
    return float(a), float(b)
