"""Drop-in for the reference's ``libs/helper_mimo_esn_generic.py``:
``trainMIMOESN_generic`` with the same arguments and the same 9-element return
list (helper_mimo_esn_generic.py:5-86), driving the HIP-backed ``ESN``.

The call sequence is the reference's -- fit, predict on the training input,
(delay scan when ``DelayFlag`` is set), final fit (helper:44-45,84) -- so the
ESN's RandomState is consumed exactly as in the reference and a seeded run
reproduces its W_out.  The reported NMSE keeps the reference's slice
(helper:47-55; mis-aligned by d for DelayFlag == 0).

``pack_frames`` / ``trainMIMOESN_batch`` are the batched siblings used by the
Monte-Carlo harness: G pilots trained in one harvest + one solve launch.
"""
from __future__ import annotations

import numpy as np


def _build_io_for_delay(y_CP, x_CP, d, N, CyclicPrefixLen, N_t, N_r):
    """Re/Im interleave + d trailing zero input rows + d leading zero teacher rows (helper:26-38)."""
    T = N + CyclicPrefixLen
    X_in = np.zeros((T + d, 2 * N_r), dtype=float)
    X_out = np.zeros((T + d, 2 * N_t), dtype=float)
    X_in[:T, 0::2] = y_CP.real
    X_in[:T, 1::2] = y_CP.imag
    X_out[d:d + T, 0::2] = x_CP.real
    X_out[d:d + T, 1::2] = x_CP.imag
    return X_in, X_out


def trainMIMOESN_generic(esn, DelayFlag, Min_Delay, Max_Delay,
                         CyclicPrefixLen, N, N_t, N_r, IsiDuration,
                         y_CP, x_CP):
    y_CP = np.asarray(y_CP)
    x_CP = np.asarray(x_CP)

    def nmse_for_delay(d):
        X_in, X_out = _build_io_for_delay(y_CP, x_CP, d, N, CyclicPrefixLen, N_t, N_r)
        nForget = d + CyclicPrefixLen
        esn.fit(X_in, X_out, nForget)
        pred = esn.predict(X_in, nForget, continuation=False)
        nmse_sum = 0.0
        for tx in range(N_t):
            x_hat = pred[d:d + N + 1, 2 * tx] + 1j * pred[d:d + N + 1, 2 * tx + 1]
            x_true = x_CP[IsiDuration - 1:, tx]
            M = min(len(x_hat), len(x_true))
            if M > 0:
                nmse_sum += np.linalg.norm(x_hat[:M] - x_true[:M]) ** 2 / \
                    (np.linalg.norm(x_true[:M]) ** 2 + 1e-12)
        return nmse_sum, X_in, X_out, nForget

    if DelayFlag == 0:
        d = int((Min_Delay + Max_Delay) // 2)
        nmse, ESN_input, ESN_output, nForgetPoints = nmse_for_delay(d)
        Delay_Idx = d - Min_Delay
        NMSE_ESN = float(nmse)
    else:
        best_nmse, best = 1e9, None
        Delay_Idx = 0
        for dd in range(Min_Delay, Max_Delay + 1):
            nmse, Xin, Xout, nF = nmse_for_delay(dd)
            if nmse < best_nmse:
                best_nmse, best, Delay_Idx = nmse, (Xin, Xout, nF, dd), dd - Min_Delay
        ESN_input, ESN_output, nForgetPoints, d = best
        NMSE_ESN = float(best_nmse)

    Delay = np.full(2 * N_t, int(d), dtype=int)
    esn.fit(ESN_input, ESN_output, nForgetPoints)
    return [ESN_input, ESN_output, esn, Delay, Delay_Idx, int(d), int(d), nForgetPoints, NMSE_ESN]


# ---------------------------------------------------------------------------
# batched siblings (extension): the packing is a VIEW, not a copy
# ---------------------------------------------------------------------------
def complex_frames_as_esn_io(z):
    """complex128 [..., T, n] -> float64 view [..., T, 2n] with Re/Im interleaved, which is
    exactly the layout helper:30-37 / driver:433-436 build column by column."""
    z = np.ascontiguousarray(z, dtype=np.complex128)
    return z.view(np.float64).reshape(*z.shape[:-1], 2 * z.shape[-1])


def trainMIMOESN_batch(bank, y_CP, x_CP, d, CyclicPrefixLen, precision="f64", noise_mode="counter", seed=0):
    """G pilots at once: y_CP [G,T,N_r], x_CP [G,T,N_t] complex -> bank.W_out (one harvest + one
    solve launch).  Returns (E, nForget).  Teacher rows are delayed by d; the d trailing input rows
    are the zeros the kernel synthesises beyond T_in -- here materialised because harvest takes
    equal-length U and D."""
    y = complex_frames_as_esn_io(y_CP)
    x = complex_frames_as_esn_io(x_CP)
    g, t = y.shape[0], y.shape[1]
    U = np.zeros((g, t + d, y.shape[2]))
    D = np.zeros((g, t + d, x.shape[2]))
    U[:, :t] = y
    D[:, d:d + t] = x
    n_forget = d + CyclicPrefixLen
    E = bank.fit(U, D, transient=n_forget, precision=precision, noise_mode=noise_mode, seed=seed)
    return E, n_forget
