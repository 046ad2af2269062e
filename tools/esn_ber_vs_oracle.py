#!/usr/bin/env python3
"""GPU sweep vs the CPU oracle run of tools/cpu_oracle_ber.py at the reference's published configuration
(4x8 TDL-B, N = 128, N_res = 300, state noise on): ESN uncoded BER at a few Eb/No points for
  * one fresh reservoir per coherence block (what the reference and the oracle run do),
  * the benchmark's pool of 8 pre-drawn reservoirs, and the shared reservoir,
in fp16 and float64 -- to tell a reservoir-pool effect from an arithmetic one.
usage: esn_ber_vs_oracle.py [blocks]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from esn_ofdm_mimo_amd.montecarlo import DetectorSweep, LinkParams  # noqa: E402

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 240
ebno = [15.0, 21.0, 27.0]
oracle = {15.0: 0.21417, 21.0: 0.17134, 27.0: 0.15600}            # tools/cpu_oracle_ber.py, 240 blocks, stderr 0.0011
published = {15.0: 0.20868, 21.0: 0.16521, 27.0: 0.16198}         # results_ber.csv, 14 blocks per point
rows = []
for name, kw in (("per_block_fresh", dict(reservoirs="per_block", pool=blocks)),
                 ("pool_of_8", dict(reservoirs="per_block", pool=8)),
                 ("pool_of_16", dict(reservoirs="per_block", pool=16)),
                 ("shared", dict(reservoirs="shared"))):
    for prec in ("f16", "f64"):
        if prec == "f64" and name not in ("per_block_fresh", "shared"):
            continue
        for seed in (7, 8):
            sw = DetectorSweep(LinkParams(), n_reservoir=300, noise=0.001, seed=seed, precision=prec,
                               fit_precision=prec, **kw)
            ber, counts = sw.run(ebno, blocks)
            rec = dict(reservoirs=name, precision=prec, seed=seed, blocks=blocks,
                       ber={str(e): float(b) for e, b in zip(ebno, ber)},
                       vs_oracle={str(e): float(b) / oracle[e] for e, b in zip(ebno, ber)})
            print(json.dumps(rec), flush=True)
print(json.dumps(dict(oracle=oracle, published=published)))
