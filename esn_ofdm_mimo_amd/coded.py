"""Coded leg of the north-star driver: regular Gallager LDPC code, encoder, max-log
LLRs, per-bit logistic LLR calibration and sum-product decoding, batched on the GPU.

The reference delegates the code to the un-vendored package ``pyldpc`` (``requirements-sm2.txt:5``;
call sites ``Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:250,495-496,505-506``).  What is restated here:

  * ``LdpcCode``            ``make_ldpc(n, d_v, d_c, systematic=True, sparse=True)``: Gallager
                            construction + GF(2) elimination to a systematic generator.  Host-side,
                            once per sweep (pyldpc draws its permutations from an unseeded RNG, so
                            the reference's code differs from run to run; only the ensemble is fixed).
  * ``encode``              ``G.dot(u) % 2`` (:90-93)            -> ``esn_ldpc_encode``
  * ``llrs``                ``qam_llrs_maxlog`` + ``est_sigma2_from_decision`` (:66-93, :459-469)
                                                                 -> ``esn_qam_llr``
  * ``fit_calibration``     ``fit_logreg_1d`` (:108-119, :513-523): 400 plain gradient steps from
                            (1, 0), lr 0.1, l2 1e-3 -- per-SNR host logic of the driver, run as
                            torch tensor ops on the device (4 scalars per bit position).
  * ``decode_count``        ``-(a llr + b)`` clipped to +-20, ``yobs = 0.5 llr``, ``decode(H, yobs,
                            snr=1.0, maxiter=100)``, ``get_message`` and the info-bit error count
                            (:483-511)                           -> ``esn_ldpc_decode_count``
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import check, ptr

LLR_CLIP = 20.0            # driver :245
LDPC_MAXITER = 100         # driver :244
SNR_FOR_LDPC = 1.0         # driver :484 (pyldpc: var = 10^(-snr/10))


class LdpcCode:
    """Regular (d_v, d_c) Gallager code of length n in systematic form c = [u ; P u]."""

    def __init__(self, n, d_v=4, d_c=8, seed=0, device=None):
        if n % d_c:
            raise ValueError(f"LDPC_dc={d_c} must divide n_code={n}.")       # driver :248-249
        torch = _lib.require_gpu()
        self.torch, self.lib = torch, _lib.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        rs = np.random.RandomState(seed)
        rows = n // d_c
        base = np.zeros((rows, n), dtype=np.uint8)
        base[np.repeat(np.arange(rows), d_c), np.arange(n)] = 1
        H = np.concatenate([base] + [base[:, rs.permutation(n)] for _ in range(d_v - 1)], axis=0)
        # GF(2) row reduction; pivot columns become the parity positions
        A = H.copy()
        m = A.shape[0]
        pivots, r = [], 0
        for c in range(n):
            if r == m:
                break
            nz = np.flatnonzero(A[r:, c])
            if nz.size == 0:
                continue
            p = r + nz[0]
            if p != r:
                A[[r, p]] = A[[p, r]]
            hit = np.flatnonzero(A[:, c])
            hit = hit[hit != r]
            A[hit] ^= A[r]
            pivots.append(c)
            r += 1
        rank = r
        pset = set(pivots)
        free = [c for c in range(n) if c not in pset]
        order = np.array(free + pivots)
        self.n, self.k, self.rank = n, n - rank, rank
        self.P = np.ascontiguousarray(A[:rank][:, free]).astype(np.uint8)         # [n-k, k]
        self.H = np.ascontiguousarray(H[:, order]).astype(np.uint8)               # decoder graph
        ci, vi = np.nonzero(self.H)                                                # check-major edges
        e = len(ci)
        chk_ptr = np.concatenate([[0], np.cumsum(np.bincount(ci, minlength=m))]).astype(np.int32)
        by_var = np.argsort(vi, kind="stable").astype(np.int32)
        var_ptr = np.concatenate([[0], np.cumsum(np.bincount(vi, minlength=n))]).astype(np.int32)
        self.m_checks, self.n_edges = m, e
        with torch.cuda.device(self.device):
            dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=self.device)
            self._P = dev(self.P)
            self._chk_ptr, self._edge_var = dev(chk_ptr), dev(vi.astype(np.int32))
            self._var_ptr, self._var_edge = dev(var_ptr), dev(by_var)

    # ------------------------------------------------------------------ transmitter side
    def encode(self, u, n_t):
        """u uint8 [B, n_t, k] -> TxBits uint8 [B, n, n_t]  (feed to FrameSource.frames(bits_in=...))."""
        torch = self.torch
        u = u.to(device=self.device, dtype=torch.uint8).contiguous()
        b = u.shape[0]
        with torch.cuda.device(self.device):
            bits = torch.empty((b, self.n, n_t), dtype=torch.uint8, device=self.device)
            check(self.lib.esn_ldpc_encode(b, n_t, self.k, self.n, ptr(self._P), ptr(u), ptr(bits),
                                           _lib.stream_handle()), "esn_ldpc_encode")
        return bits

    # ------------------------------------------------------------------ receiver side
    def llrs(self, x_hat, bits_per_sym):
        """x_hat complex128 [B, N, n_t] -> (llr float64 [B, n_t, N*m], sigma2 [B])."""
        torch = self.torch
        x_hat = x_hat.contiguous()
        b, n_sub, n_t = x_hat.shape
        with torch.cuda.device(self.device):
            llr = torch.empty((b, n_t, n_sub * bits_per_sym), dtype=torch.float64, device=self.device)
            s2 = torch.empty(b, dtype=torch.float64, device=self.device)
            check(self.lib.esn_qam_llr(b, n_sub, n_t, bits_per_sym, ptr(x_hat), ptr(llr), ptr(s2),
                                       _lib.stream_handle()), "esn_qam_llr")
        return llr, s2

    def fit_calibration(self, llr, bits, bits_per_sym, maxiter=400, lr=0.1, l2=1e-3):
        """Per bit position b: p(bit=1 | llr) = sigmoid(a_b llr + b_b), gradient descent from (1, 0).
        llr [B, n_t, N*m], bits uint8 [B, N*m, n_t] -> (a [m], b [m]) device tensors."""
        torch, m = self.torch, bits_per_sym
        x = llr.reshape(llr.shape[0], llr.shape[1], -1, m).permute(3, 0, 1, 2).reshape(m, -1)      # [m, S]
        y = bits.permute(0, 2, 1).reshape(bits.shape[0], bits.shape[2], -1, m).permute(3, 0, 1, 2) \
            .reshape(m, -1).to(torch.float64)
        a = torch.ones(m, dtype=torch.float64, device=self.device)
        b = torch.zeros(m, dtype=torch.float64, device=self.device)
        n = x.shape[1]
        for _ in range(maxiter):
            p = torch.sigmoid(a[:, None] * x + b[:, None])
            ga = ((p - y) * x).sum(1) / n + l2 * a
            gb = (p - y).sum(1) / n
            a = a - lr * ga
            b = b - lr * gb
        return a, b

    def decode_count(self, llr, a, b, u_true, cw_per_group, bits_per_sym, err=None, nbits=None,
                     maxiter=LDPC_MAXITER, want_bits=False):
        """Calibrated LLRs -> sum-product decode -> info-bit errors per group (driver :483-511)."""
        torch, m = self.torch, bits_per_sym
        bsz, n_t, n = llr.shape
        cal = -(a.view(1, 1, 1, m) * llr.view(bsz, n_t, -1, m) + b.view(1, 1, 1, m))
        yobs = (0.5 * cal.clamp(-LLR_CLIP, LLR_CLIP)).reshape(bsz * n_t, n).contiguous()
        n_cw = bsz * n_t
        g = (n_cw + cw_per_group - 1) // cw_per_group
        u_true = None if u_true is None else u_true.to(device=self.device, dtype=torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            if err is None:
                err = torch.zeros(g, dtype=torch.int64, device=self.device)
            if nbits is None:
                nbits = torch.zeros(g, dtype=torch.int64, device=self.device)
            xo = torch.empty((n_cw, n), dtype=torch.uint8, device=self.device) if want_bits else None
            check(self.lib.esn_ldpc_decode_count(
                n_cw, n, self.k, self.m_checks, self.n_edges, ptr(self._chk_ptr), ptr(self._edge_var),
                ptr(self._var_ptr), ptr(self._var_edge), ptr(yobs), SNR_FOR_LDPC, int(maxiter), ptr(u_true),
                int(cw_per_group), ptr(xo), ptr(err), ptr(nbits), _lib.stream_handle()), "esn_ldpc_decode_count")
        return (err, nbits, xo) if want_bits else (err, nbits)
