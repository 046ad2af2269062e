#!/usr/bin/env python3
"""ESN uncoded BER of the NumPy oracle (= the reference algorithm, float64, one frame per call, a FRESH reservoir
per coherence block as the reference draws it, state noise on) at the reference's published configuration
(4x8 TDL-B, N = 128, N_res = 300) -- the CPU-side number to set beside the GPU sweep and the published column
`ESN_uncoded` of results/results_4x8_cdl_coded_uncoded/CDLB_run_01/results_ber.csv.

    python tools/cpu_oracle_ber.py --ebno 21 --blocks 240 --procs 6

VERDICT r2 weak #1: the GPU sweep sits 3-7 % above the published column at >= 15 dB (0.1773 vs 0.1652 at 21 dB);
the published run has 14 channel draws per point.  This run says what the oracle itself gives on hundreds."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def one_block(job):
    ebno, n_res, block, frames = job
    import numpy as np
    from threadpoolctl import threadpool_limits
    from oracle import esn_oracle as eo
    from oracle.ofdm_frames import LinkConfig, make_frame, tdlb_mimo_taps
    cfg = LinkConfig()
    n_in, n_out = 2 * cfg.n_r, 2 * cfg.n_t
    with threadpool_limits(limits=1):
        rs = np.random.RandomState(7000 + block)
        taps = tdlb_mimo_taps(cfg, 1234 + int(ebno) + block * 75 + 1)          # the driver's seed pattern (:321)
        esn = eo.OracleESN(n_in, n_out, n_res, spectral_radius=0.9, sparsity=0.1, noise=0.001,
                           input_scaling=cfg.input_scaling(ebno) * np.ones(n_in), input_shift=np.zeros(n_in),
                           teacher_scaling=cfg.teacher_scale * np.ones(n_out), teacher_shift=np.zeros(n_out),
                           random_state=rs)                                     # weights AND noise from one stream
        pilot = make_frame(cfg, ebno, taps, rs)
        ret = eo.train_mimo_esn(esn, 0, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub, cfg.n_t, cfg.n_r, cfg.isi,
                                pilot["y_cp"], pilot["x_cp"])
        _, _, _, delay, _, d_min, d_max, forget, _ = ret
        const = eo.unit_qam(cfg.m)
        errs = tot = 0
        for _ in range(frames):
            fr = make_frame(cfg, ebno, taps, rs)
            _, rx = eo.detect_frame(esn, fr["y_cp"], delay, d_min, d_max, forget, cfg.n_sub, cfg.n_t,
                                    cfg.p_i(ebno), const, cfg.m)
            errs += eo.count_bit_errors(fr["bits"], rx)
            tot += rx.size
    return errs, tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ebno", type=float, nargs="+", default=[21.0])
    ap.add_argument("--blocks", type=int, default=240)
    ap.add_argument("--frames", type=int, default=75)
    ap.add_argument("--n-res", type=int, default=300)
    ap.add_argument("--procs", type=int, default=6)
    args = ap.parse_args()
    import multiprocessing as mp
    import numpy as np
    out = []
    for ebno in args.ebno:
        t0 = time.time()
        with mp.get_context("spawn").Pool(args.procs) as pool:
            res = pool.map(one_block, [(ebno, args.n_res, b, args.frames) for b in range(args.blocks)], chunksize=2)
        e = np.array([r[0] for r in res], dtype=float)
        n = np.array([r[1] for r in res], dtype=float)
        ber = e.sum() / n.sum()
        per_block = e / n
        sem = per_block.std(ddof=1) / np.sqrt(len(per_block))              # blocks are the independent units
        rec = dict(ebno_db=ebno, n_res=args.n_res, blocks=args.blocks, frames_per_block=args.frames, ber=ber,
                   stderr_over_blocks=sem, seconds=time.time() - t0)
        print(json.dumps(rec), flush=True)
        out.append(rec)


if __name__ == "__main__":
    main()
