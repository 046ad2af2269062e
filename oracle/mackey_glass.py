"""Mackey-Glass series for BASELINE configs[0] -- TEST INFRASTRUCTURE ONLY (same rules as
``esn_oracle.py``).

The reference holds no Mackey-Glass code or data (only the plot
``results/mecky_glass.png.png`` of the upstream pyESN demo -- a 2000-step teacher-forced fit on a
constant input followed by a 2000-step free run), so the series is generated here:

    dx/dt = beta x(t - tau) / (1 + x(t - tau)^p) - gamma x(t),   beta 0.2, gamma 0.1, p 10, tau 17

integrated with classical RK4 at h = 0.1 (the delayed value at the half step is the mean of its
two grid neighbours), sampled every 1.0 after a 1000-unit wash-out from the constant history 1.2.
The golden fixture stores the series itself next to the reference's outputs on it, so the tests do
not depend on this integrator being reproduced bit for bit.
"""
from __future__ import annotations

import numpy as np


def mackey_glass(n, tau=17.0, beta=0.2, gamma=0.1, p=10, h=0.1, x0=1.2, washout=1000):
    lag = int(round(tau / h))
    per = int(round(1.0 / h))
    total = (n + washout) * per
    x = np.empty(total + lag + 1)
    x[:lag + 1] = x0

    def f(xt, xd):
        return beta * xd / (1.0 + xd ** p) - gamma * xt

    for i in range(lag, total + lag):
        d0, d1 = x[i - lag], x[i - lag + 1]
        dm = 0.5 * (d0 + d1)
        k1 = f(x[i], d0)
        k2 = f(x[i] + 0.5 * h * k1, dm)
        k3 = f(x[i] + 0.5 * h * k2, dm)
        k4 = f(x[i] + h * k3, d1)
        x[i + 1] = x[i] + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
    return x[lag + washout * per::per][:n].copy()
