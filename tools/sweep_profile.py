#!/usr/bin/env python3
"""The user-facing sweep alone (bench.py's `sweep` sub-record) -- run under
`rocprofv3 --kernel-trace --stats` to see where a DetectorSweep.run spends the GPU.
usage: sweep_profile.py [precision] [n_res]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from esn_ofdm_mimo_amd.montecarlo import LinkParams  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
n_res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
params = LinkParams()
rec = bench.run_sweep(torch, params, precision=prec, fit_precision="auto", n_res=n_res,
                      F=params.coherence_symbols, solve="auto")
print(json.dumps(rec))
