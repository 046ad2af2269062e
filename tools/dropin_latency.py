#!/usr/bin/env python3
"""Latency of the UNCHANGED reference call pattern -- one `fit` / one `predict` per OFDM frame through the 2-D
drop-in (libs/pyESN.py surface) -- at the headline shape: the LDS-resident cluster kernel (default), the
vector-ALU kernel (debug knob cluster=0) and the NumPy oracle on this host.  Also times the device work alone
(HIP events around the kernel calls, no host copies)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from esn_ofdm_mimo_amd import _lib, pyESN  # noqa: E402
from oracle import esn_oracle as eo  # noqa: E402

rs = np.random.RandomState(0)
n_in, n_out, n_res, T = 16, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 512, 138
u, d = rs.randn(T, n_in) * 0.1, rs.randn(T, n_out) * 0.1
kw = dict(spectral_radius=0.9, sparsity=0.1, noise=0.001, input_scaling=0.05 * np.ones(n_in), input_shift=np.zeros(n_in),
          teacher_scaling=5e-3 * np.ones(n_out), teacher_shift=np.zeros(n_out))
e = pyESN.ESN(n_in, n_out, n_res, random_state=1, **kw)
o = eo.OracleESN(n_in, n_out, n_res, random_state=1, **kw)


def timed(m, name):
    m.fit(u, d, 10); m.predict(u, 10, continuation=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m.fit(u, d, 10)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(20):
        m.predict(u, 10, continuation=False)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: fit {1e3*(t1-t0)/5:.2f} ms, predict {1e3*(t2-t1)/20:.2f} ms per {T}-sample sequence (host wall, "
          f"incl. RandomState draws and copies)")


def device_only(name):
    bank = e._get_bank()
    ud = torch.as_tensor(u[None], device=bank.device)
    nz = torch.rand((1, T, n_res), dtype=torch.float64, device=bank.device)
    for _ in range(3):
        bank.predict(ud, 1, transient=10, precision="f64", noise_mode="tensor", noise_u=nz)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); bank.predict(ud, 1, transient=10, precision="f64", noise_mode="tensor", noise_u=nz); b.record()
    torch.cuda.synchronize()
    ms = np.median([a.elapsed_time(b) for a, b in ev])
    print(f"{name}: predict kernel(s) alone {ms:.3f} ms = {1e3*ms/T:.2f} us per step")


timed(e, "hip drop-in (cluster kernel)")
device_only("hip drop-in (cluster kernel)")
_lib.debug_set("cluster", "0")
timed(e, "hip drop-in (vector-ALU kernel)")
device_only("hip drop-in (vector-ALU kernel)")
_lib.debug_set("cluster", "1")
timed(o, "numpy oracle")
