#!/usr/bin/env python3
"""BASELINE configs[4]: 4x8 TDL-B, large reservoir N_res = 2048, a 10^7-frame uncoded BER sweep over 0..20 dB on ONE
MI355X (the config names 8; the sweep shards by block, `bench.py --gpus N` measures the scaling).  Writes the reference's
CSV header with the one column this sweep has, plus wall time and rate.

    python tools/big_sweep.py [--frames 1e7] [--n-res 2048] [--out profiles/r03_big_sweep_nres2048.csv]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=float, default=1e7)
    ap.add_argument("--n-res", type=int, default=2048)
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    from esn_ofdm_mimo_amd.montecarlo import DetectorSweep, LinkParams
    prm = LinkParams()
    F = prm.coherence_symbols
    ebno = [float(e) for e in range(0, 21, 2)]
    t0 = time.perf_counter()
    sw = DetectorSweep(prm, n_reservoir=a.n_res, noise=0.001, seed=7, precision=a.precision, fit_precision=a.precision)
    t_init = time.perf_counter() - t0                        # dominated by the host eigvals of the reservoir draw
    chunk = sw.default_chunk_blocks(F)
    blocks = max(chunk, int(round(a.frames / (len(ebno) * F) / chunk)) * chunk)
    sw.run([ebno[0]], chunk, frames_per_block=F)             # warm-up
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ber, counts = sw.run(ebno, blocks, frames_per_block=F)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    frames = len(ebno) * blocks * F
    lines = ["EbNo(dB),ESN_uncoded"] + [f"{int(e)},{b}" for e, b in zip(ebno, ber)]
    lines.append(f"# N_res={a.n_res} {a.precision} shared reservoir, {blocks} blocks x {F} symbols per point = {frames} frames "
                 f"in {dt:.2f} s = {frames / dt:.0f} symbols/s on one GPU (generation, training, detection, host syncs); "
                 f"reservoir draw {t_init:.1f} s; bit errors {int(counts[:, 0].sum())} of {int(counts[:, 1].sum())}; "
                 f"fits repaired {sw.fits_repaired}")
    print("\n".join(lines))
    if a.out:
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
