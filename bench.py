#!/usr/bin/env python3
"""Headline benchmark: OFDM symbols/s through the ESN detector, 4x8, N_res=512 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the detector hot path over a resident batch of synthetic input:
G coherence blocks x L data frames per rank (4x8 TDL-B, N=128, CP=7, d=3, 16-QAM, Eb/No 12 dB,
shared reservoir N_res=512, state noise on):

    train   G pilots: harvest (float32 MFMA) + Householder-QR readout solve (float64)   [a6/a8]
    predict G*L frames through the recurrence + readout                                  [a7, a9]
    detect  reconstruct + (1/N) FFT / sqrt(Pi) + 16-QAM slicer + bit-error count         [a10-a12]

so training is amortised at the reference ratio (one pilot per L = 75 symbols).  Inputs are
resident in HBM before the timed region.  Ranks shard blocks (weak scaling: per-GPU work fixed),
no data-path collective; the int64 error counters are summed with one all_reduce at the end.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the predict recurrence):
algorithmic FLOPs per launch (2 T [N_res (N_res+n_in+n_out) + n_out (N_res+n_in)]
per frame) / mean launch duration measured with HIP events on the launch stream.
`cpu_baseline` times the NumPy oracle (the reference algorithm, one frame per call, float64)
on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "f16": 2500.0, "bf16": 2500.0, "f64": 78.6}   # MI355X_MICROARCH.md (dense)


def flop_per_frame(n_res, n_in, n_out, T):
    return 2 * T * (n_res * (n_res + n_in + n_out) + n_out * (n_res + n_in))


def cpu_baseline(params, n_res, ebno, n_blocks, frames_per_block):
    """NumPy oracle (kind 'port'): train once per block + detect every frame, single BLAS thread."""
    import numpy as np
    from threadpoolctl import threadpool_limits
    from oracle import esn_oracle as eo
    from oracle.ofdm_frames import LinkConfig, make_frame, tdlb_mimo_taps
    cfg = LinkConfig(n_t=params.n_t, n_r=params.n_r, n_sub=params.n_sub, m=params.m)
    n_in, n_out = 2 * cfg.n_r, 2 * cfg.n_t
    rs = np.random.RandomState(123)
    const = eo.unit_qam(cfg.m)
    esn = eo.OracleESN(n_in, n_out, n_res, spectral_radius=0.9, sparsity=0.1, noise=0.001,
                       input_scaling=cfg.input_scaling(ebno) * np.ones(n_in), input_shift=np.zeros(n_in),
                       teacher_scaling=cfg.teacher_scale * np.ones(n_out), teacher_shift=np.zeros(n_out),
                       random_state=rs)
    frames = []
    for b in range(n_blocks):
        taps = tdlb_mimo_taps(cfg, 1000 + b)
        frames.append((make_frame(cfg, ebno, taps, rs), [make_frame(cfg, ebno, taps, rs)
                                                         for _ in range(frames_per_block)]))
    errs = tot = 0
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        for pilot, data in frames:
            ret = eo.train_mimo_esn(esn, 0, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub, cfg.n_t,
                                    cfg.n_r, cfg.isi, pilot["y_cp"], pilot["x_cp"])
            _, _, _, delay, _, d_min, d_max, forget, _ = ret
            for fr in data:
                _, rx = eo.detect_frame(esn, fr["y_cp"], delay, d_min, d_max, forget, cfg.n_sub, cfg.n_t,
                                        cfg.p_i(ebno), const, cfg.m)
                errs += eo.count_bit_errors(fr["bits"], rx)
                tot += rx.size
        dt = time.perf_counter() - t0
    n = n_blocks * frames_per_block
    return dict(value=n / dt, unit="OFDM symbols/s", cores=1, kind="port",
                sample=f"{n_blocks} blocks x {frames_per_block} frames (train + detect), NumPy float64 oracle, "
                       f"1 BLAS thread, {dt:.1f} s, BER {errs / max(tot, 1):.3f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="f16", choices=["f32", "f16", "bf16", "f64"])
    ap.add_argument("--fit-precision", default="auto", choices=["auto", "f16", "bf16", "f32", "f64"],
                    help="harvest arithmetic; auto = the predict precision for f16/bf16 (states are then rounded\n"
                         "the same way at train and detect time), f32 otherwise")
    ap.add_argument("--n-res", type=int, default=512)
    ap.add_argument("--blocks", type=int, default=0, help="coherence blocks per rank per step (0 = auto)")
    ap.add_argument("--frames-per-block", type=int, default=0, help="0 = L of the reference (75 at N=128)")
    ap.add_argument("--ebno", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-blocks", type=int, default=32, help="cpu_baseline sample: blocks of L frames (~10-15 s)")
    ap.add_argument("--predict-only", action="store_true", help="time the predict+detect leg only")
    ap.add_argument("--solve", default="auto", choices=["auto", "qr", "chol"])
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from esn_ofdm_mimo_amd import _lib
    from esn_ofdm_mimo_amd.montecarlo import DetectorSweep, LinkParams, reduce_counters

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if args.fit_precision == "auto":
        args.fit_precision = args.precision if args.precision in ("f16", "bf16") else "f32"
    params = LinkParams()                               # 4x8 TDL-B, N=128, 16-QAM
    F = args.frames_per_block or params.coherence_symbols
    tile = {"f32": 64, "f16": 128, "bf16": 128, "f64": 8}[args.precision]
    # auto: a whole number of workgroup tiles per CU (tiles = G * ceil16(F) / tile), ~5-10 rounds
    fpad = ((F + 15) // 16) * 16
    G = args.blocks or max(1, (5 * 256 * 128) // fpad)
    sweep = DetectorSweep(params, n_reservoir=args.n_res, noise=0.001, seed=1234,   # same reservoir on every rank
                         
                          precision=args.precision, fit_precision=args.fit_precision,
                          reservoirs="shared", rank=rank, world_size=world, solve_method=args.solve)
    data = sweep.src.blocks_fast(args.ebno, 0, rank * G, G, F)
    sweep.set_snr(args.ebno, G)
    err = torch.zeros(G, dtype=torch.int64, device=sweep.device)
    nb = torch.zeros(G, dtype=torch.int64, device=sweep.device)
    T = params.t_frame + params.delay
    y_out = torch.empty((G * F, params.n_sub, sweep.n_out), dtype=torch.float64, device=sweep.device)
    sweep.train(data["pilot_y"], data["pilot_x"], seed=1)      # W_out exists for predict-only mode

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

    def step(i, timed):
        if not args.predict_only:
            sweep.train(data["pilot_y"], data["pilot_x"], seed=i)
        U = torch.view_as_real(data["data_y"]).reshape(G * F, params.t_frame, sweep.n_in)
        if timed:
            ev[i][0].record()
        y = sweep.bank.predict(U, F, T=T, transient=params.delay + params.cp, precision=args.precision,
                               noise_mode="counter", seed=i, out=y_out)
        if timed:
            ev[i][1].record()
        sweep.bank.detect_count(y, data["data_bits"], sweep.p_i, F, params.n_sub, params.n_t, params.m,
                                err=err, bits=nb)

    for i in range(args.warmup):
        step(i % max(args.steps, 1), False)
    err.zero_(); nb.zero_()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    counters = torch.stack([err.sum(), nb.sum()]).view(1, 2)
    reduce_counters(counters, dist if world > 1 else None, world)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=sweep.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else float("nan")
    frames_per_step = G * F
    flop = flop_per_frame(args.n_res, sweep.n_in, sweep.n_out, T) * frames_per_step
    achieved = flop / (kernel_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[args.precision]
    c = counters.cpu().numpy()[0]

    # HBM bytes per launch of the dominant kernel from the last committed PMC pass, if it was taken
    # on this workload (tools/pmc_traffic.py on the GPU box; FETCH_SIZE/WRITE_SIZE in their own passes)
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic_%s.json" % args.precision)
    if os.path.exists(pmc_path):
        try:
            pj = json.load(open(pmc_path))
            if pj.get("blocks") == G and pj.get("traffic_bytes_per_launch"):
                traffic = pj["traffic_bytes_per_launch"]
        except Exception:
            traffic = None

    if rank == 0:
        out = {
            "metric": "OFDM symbols/s through the ESN detector (4x8 TDL-B, N=128, N_res=%d; train+predict+detect)" % args.n_res,
            "value": world * frames_per_step * args.steps / dt,
            "unit": "OFDM symbols/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "f16": "f16 operands / f32 accumulate", "bf16": "bf16 operands / f32 accumulate",
                      "f64": "f64"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "configs[3]: OFDM 4x8 MIMO, TDL-B taps, 16-QAM, N=128, CP=7, d=3, N_res=%d, "
                                   "Eb/No %g dB, uncoded" % (args.n_res, args.ebno),
                       "blocks_per_rank": G, "frames_per_block": F, "frames_per_step": world * frames_per_step,
                       "reservoir": "shared", "state_noise": 0.001, "fit": "harvest %s + %s f64 solve" % (args.fit_precision, args.solve),
                       "timed": "predict+detect" if args.predict_only else "train+predict+detect",
                       "parallelism": "blocks sharded over %d rank(s), one all_reduce of counters" % world},
            "ber": float(c[0]) / max(float(c[1]), 1.0),
            "fit_groups_flagged": int(sweep.bank.fit_status.sum().item()),
            "predict_kernel_ms": kernel_ms,
            "predict_only_symbols_per_s": world * frames_per_step / (kernel_ms * 1e-3),
            "roofline": {"bound": "mfma", "kernel": "esn::recur_mfma_kernel (predict)", "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "traffic_unit": "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_traffic_*.json)",
                         "algorithmic_bytes_per_launch": frames_per_step * 8 * (params.t_frame * sweep.n_in + params.n_sub * sweep.n_out),
                         "flop_per_frame": flop_per_frame(args.n_res, sweep.n_in, sweep.n_out, T),
                         "frames_per_launch": frames_per_step},
            "device": _lib.device_info(),
        }
        if not args.no_cpu_baseline and args.cpu_blocks > 0 and world == 1:      # reported once, at N=1 only
            out["cpu_baseline"] = cpu_baseline(params, args.n_res, args.ebno, args.cpu_blocks, F)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
