"""Drop-in for the reference's ``libs/HelpFunc.py``: the legacy 2x2 trainer
``HelpFunc.trainMIMOESN`` and the constellation helper ``HelpFunc.UnitQamConstellation``, as
``system_model_2/system_model_2_all_comparision.py:4,434-446`` and
``Demo_MIMO_2x2_all_DL_model_comparion.py`` call them -- every fit and predict inside runs on the HIP
kernels through ``pyESN.ESN``.

Behaviour kept from ``HelpFunc.py:64-187`` (DelayFlag == 0, the only branch that can run):
two receive / two transmit streams are hard-wired; delay row j = [j, j, j, j] for j = 0..Max_Delay;
every row is fitted and predicted on the training input (7 fits + 7 predicts at Max_Delay = 6, so
the ESN's RandomState advances exactly as in the reference); the row index is then forced to 3
(:159), the NMSE vector is printed (:161), NMSE_ESN is the minimum over all rows, and the ESN is
fitted once more at d = 3.  ``DelayFlag != 0`` raises the reference's own TypeError
(``np.zeros(shape, 1)``, :76).
"""
from __future__ import annotations

import math

import numpy as np


class HelpFunc:
    @staticmethod
    def UnitQamConstellation(Bi):
        """Unit-mean-power square QAM, index = i*side + j -> Re pam[i], Im pam[j] (HelpFunc.py:6-39)."""
        side = math.ceil(math.sqrt(2 ** Bi) / 2) * 2
        pam = np.arange(-(side - 1), side, 2).astype(np.int32)
        c = (pam[:, None] + 1j * pam[None, :]).reshape(-1).astype(np.complex128)
        return c / math.sqrt(np.mean(abs(c) ** 2))

    @staticmethod
    def ComputeChannelCorrMatrix(IsiMagnitude):
        """R_f[n, m] = r_f[n - m] below the diagonal, conj(r_f[m - n]) on and above it, r_f = FFT of
        the tap powers (HelpFunc.py:41-62; no driver calls it)."""
        r = np.fft.fft(np.asarray(IsiMagnitude))
        n = len(r)
        idx = np.arange(n)[:, None] - np.arange(n)[None, :]
        return np.where(idx > 0, r[np.abs(idx)], np.conjugate(r)[np.abs(idx)]).astype(np.complex128)

    @staticmethod
    def trainMIMOESN(esn, DelayFlag, Min_Delay, Max_Delay, CyclicPrefixLen, N, N_t, N_r, IsiDuration, y_CP, x_CP):
        if DelayFlag:
            np.zeros(((Max_Delay + 1 - Min_Delay) ** 2,), 1)         # the reference's TypeError (:76)
        y_CP, x_CP = np.asarray(y_CP), np.asarray(x_CP)
        table = np.zeros((Max_Delay + 1 - Min_Delay, 4), dtype=np.int32)
        table[np.arange(0, Max_Delay + 1)] = np.arange(0, Max_Delay + 1, dtype=np.int32)[:, None]
        hi, lo = table.max(axis=1), table.min(axis=1)
        span = N + CyclicPrefixLen

        def esn_io(j):
            X_in = np.zeros((span + hi[j], N_t * 2))
            X_out = np.zeros((span + hi[j], N_t * 2))
            X_in[:span, 0:4:2] = y_CP[:, :2].real
            X_in[:span, 1:4:2] = y_CP[:, :2].imag
            for col in range(4):
                part = x_CP[:, col // 2].real if col % 2 == 0 else x_CP[:, col // 2].imag
                X_out[table[j, col]:table[j, col] + span, col] = part
            return X_in, X_out

        target = x_CP[IsiDuration - 1:, :]
        scores = np.zeros(table.shape[0])
        for j in range(table.shape[0]):
            X_in, X_out = esn_io(j)
            forget = lo[j] + CyclicPrefixLen
            esn.fit(X_in, X_out, forget)
            est = esn.predict(X_in, forget, continuation=False)
            off = table[j] - lo[j]
            for tx in range(2):
                z = est[off[2 * tx]:off[2 * tx] + N + 1, 2 * tx] + 1j * est[off[2 * tx + 1]:off[2 * tx + 1] + N + 1, 2 * tx + 1]
                scores[j] += np.linalg.norm(z - target[:, tx]) ** 2 / np.linalg.norm(target[:, tx]) ** 2
        Delay_Idx = 3                                                # forced (HelpFunc.py:159)
        print(scores)
        ESN_input, ESN_output = esn_io(Delay_Idx)
        nForgetPoints = lo[Delay_Idx] + CyclicPrefixLen
        esn.fit(ESN_input, ESN_output, nForgetPoints)
        return [ESN_input, ESN_output, esn, table[Delay_Idx, :], Delay_Idx, lo[Delay_Idx], hi[Delay_Idx],
                nForgetPoints, np.amin(scores)]
