#!/usr/bin/env python3
"""Headline benchmark: OFDM symbols/s through the ESN detector, 4x8, N_res=512 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a
torch.distributed.run child, one process per GPU, rendezvous on 127.0.0.1) BEFORE anything in this
process touches the GPU, relays the children's output and exits with their code.  A rank whose
WORLD_SIZE differs from --gpus exits non-zero.

One "step" = one pass of the detector hot path over a resident batch of synthetic input:
G coherence blocks x L data frames per rank (4x8 TDL-B, N=128, CP=7, d=3, 16-QAM, Eb/No 12 dB,
shared reservoir N_res=512, state noise on):

    train   G pilots: harvest (MFMA recurrence) + float64 readout solve                  [a6/a8]
    predict G*L frames through the recurrence + readout                                  [a7, a9]
    detect  reconstruct + (1/N) FFT / sqrt(Pi) + 16-QAM slicer + bit-error count         [a10-a12]

so training is amortised at the reference ratio (one pilot per L = 75 symbols).  Inputs are
resident in HBM before the timed region.  Ranks shard blocks (weak scaling: per-GPU work fixed),
no data-path collective; the int64 error counters are summed with one all_reduce at the end.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the predict recurrence):
algorithmic FLOPs per launch (2 T [N_res (N_res+n_in+n_out) + n_out (N_res+n_in)]
per frame) / mean launch duration measured with HIP events on the launch stream.  Beside the
headline (fp16 operands, shared reservoir) the same line carries, measured in the same run on
the same workload (N=1 only):
    `precisions`   the step in float32 (exact f32 MFMA) and float64 (the reference's arithmetic)
    `reservoirs`   the fp16 step with one reservoir per coherence block (reference-faithful mode)
    `large_reservoir`  the fp16 step at N_res = 2048 (BASELINE configs[4]; one GEMM launch per timestep)
`cpu_baseline` times the NumPy oracle (the reference algorithm, one frame per call, float64) on
this host on bounded samples of the same workload: one BLAS thread, default BLAS threads, and one
single-threaded process per physical core over disjoint block shards (`value`).
`sweep` is what a user of `DetectorSweep.run` gets: wall clock of an Eb/No sweep including the frame generator,
training, detection and every host synchronisation.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "f16": 2500.0, "bf16": 2500.0, "f64": 78.6}   # MI355X_MICROARCH.md (dense)
DTYPE_NAME = {"f32": "f32", "f16": "f16 operands / f32 accumulate", "bf16": "bf16 operands / f32 accumulate",
              "f64": "f64"}


def flop_per_frame(n_res, n_in, n_out, T):
    return 2 * T * (n_res * (n_res + n_in + n_out) + n_out * (n_res + n_in))


# ------------------------------------------------------------------------------------------------
# CPU baseline (NumPy oracle = the reference algorithm; runs BEFORE this process touches the GPU)
# ------------------------------------------------------------------------------------------------
def _cpu_shard(job):
    """Train once per block + detect every frame with the float64 oracle; returns (frames, s, errs, bits)."""
    n_t, n_r, n_sub, m, n_res, ebno, first_block, n_blocks, frames_per_block, threads = job
    import numpy as np
    from threadpoolctl import threadpool_limits
    from oracle import esn_oracle as eo
    from oracle.ofdm_frames import LinkConfig, make_frame, tdlb_mimo_taps
    cfg = LinkConfig(n_t=n_t, n_r=n_r, n_sub=n_sub, m=m)
    n_in, n_out = 2 * cfg.n_r, 2 * cfg.n_t
    rs = np.random.RandomState(123 + first_block)
    const = eo.unit_qam(cfg.m)
    import contextlib
    with (threadpool_limits(limits=threads) if threads else contextlib.nullcontext()):
        # (the limit also covers the 512 x 512 eigvals of the constructor: eight workers with eight BLAS
        #  threads each spent 25 s there)
        esn = eo.OracleESN(n_in, n_out, n_res, spectral_radius=0.9, sparsity=0.1, noise=0.001,
                           input_scaling=cfg.input_scaling(ebno) * np.ones(n_in), input_shift=np.zeros(n_in),
                           teacher_scaling=cfg.teacher_scale * np.ones(n_out), teacher_shift=np.zeros(n_out),
                           random_state=np.random.RandomState(123))
        frames = []
        for b in range(first_block, first_block + n_blocks):
            taps = tdlb_mimo_taps(cfg, 1000 + b)
            frames.append((make_frame(cfg, ebno, taps, rs), [make_frame(cfg, ebno, taps, rs)
                                                             for _ in range(frames_per_block)]))
        errs = tot = 0
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
    return n_blocks * frames_per_block, dt, errs, tot


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """(logical CPUs this process may run on, physical cores among them, CPU quota of the cgroup or None)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    phys = set()
    for c in allowed:
        try:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            phys.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            phys.add(("?", str(c)))
    quota = None
    try:                                              # cgroup v2: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                          # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return len(allowed), len(phys), quota


def profiler_attached():
    """True under rocprofv3 & co: their preloaded library initialises the GPU in every child process, so the
    CPU legs (which fork worker processes) are skipped there."""
    pre = os.environ.get("LD_PRELOAD", "") + os.environ.get("HSA_TOOLS_LIB", "") + os.environ.get("ROCP_TOOL_LIB", "")
    return any(k in pre for k in ("rocprof", "roctracer", "rocprofiler"))


def cpu_baseline(shape, n_res, ebno, n_blocks, frames_per_block, max_procs=0):
    """Three legs on bounded samples (each ~10 s).  `value` = ONE single-threaded process per PHYSICAL core
    the process may use (capped by the cgroup's CPU quota when there is one, never by a constant), each over
    its own disjoint blocks.  Order: the process-per-core leg first, from a spawn context started before
    this process has run any BLAS or GPU code; then one process with one BLAS thread; then one process
    with the default BLAS threading."""
    import multiprocessing as mp
    n_t, n_r, n_sub, m = shape
    logical, physical, quota = host_cores()
    procs = physical
    if quota is not None:
        procs = min(procs, max(1, int(quota + 0.5)))
    if max_procs:
        procs = min(procs, max_procs)
    job = lambda first, nb, thr: (n_t, n_r, n_sub, m, n_res, ebno, first, nb, frames_per_block, thr)   # noqa: E731
    per = max(1, n_blocks // 2)                     # blocks per worker: about half the single-process sample
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(_cpu_shard, [job(1000 * (w + 1), per, 1) for w in range(procs)], chunksize=1)
    wall = time.perf_counter() - t0
    n_all = sum(r[0] for r in res)
    dt_all = max(r[1] for r in res)                 # timed regions run side by side; the slowest bounds the rate
    n1, dt1, e1, b1 = _cpu_shard(job(0, n_blocks, 1))
    nd, dtd, _, _ = _cpu_shard(job(0, n_blocks, 0))
    legs = {
        "single_thread": {"value": n1 / dt1, "cores": 1, "sample": f"{n_blocks} blocks x {frames_per_block} frames, {dt1:.1f} s"},
        "default_blas_threads": {"value": nd / dtd, "cores": logical,
                                 "sample": f"{n_blocks} blocks x {frames_per_block} frames, {dtd:.1f} s, one process, BLAS default threading"},
        "process_per_core": {"value": n_all / dt_all, "cores": procs,
                             "sample": f"{procs} processes x {per} blocks x {frames_per_block} frames, 1 BLAS thread each, "
                                       f"slowest {dt_all:.1f} s (wall incl. process start, reservoir draw and frame "
                                       f"generation {wall:.1f} s)"},
    }
    return dict(value=n_all / dt_all, unit="OFDM symbols/s", cores=procs, kind="port",
                sample=f"train + detect with the NumPy float64 oracle (reference algorithm, one frame per call), "
                       f"{procs} single-threaded processes (one per physical core"
                       f"{'' if quota is None else ', capped by the cgroup CPU quota %.1f' % quota}) x {per} blocks x "
                       f"{frames_per_block} frames; BER {e1 / max(b1, 1):.3f}",
                host_cpu=_cpu_model(), host_logical_cpus=logical, host_physical_cores=physical,
                host_cpu_quota=quota, host_cores_total=os.cpu_count(), legs=legs)


# ------------------------------------------------------------------------------------------------
# multi-rank launch
# ------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """Start n ranks of this script (one per GPU).  Nothing in this process has touched the GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------
# one timed configuration of the step
# ------------------------------------------------------------------------------------------------
def run_config(torch, dist, params, *, precision, fit_precision, reservoirs, n_res, G, F, ebno, steps, warmup,
               solve, rank, world, predict_only=False, pool=8):
    from esn_ofdm_mimo_amd.montecarlo import DetectorSweep, reduce_counters
    import numpy as np
    if fit_precision == "auto":
        fit_precision = precision if precision in ("f16", "bf16") else ("f64" if precision == "f64" else "f32")
    sweep = DetectorSweep(params, n_reservoir=n_res, noise=0.001, seed=1234,   # same reservoir(s) on every rank
                          precision=precision, fit_precision=fit_precision, reservoirs=reservoirs, pool=pool,
                          rank=rank, world_size=world, solve_method=solve)
    data = sweep.src.blocks_fast(ebno, 0, rank * G, G, F)
    sweep.set_snr(ebno, G)
    err = torch.zeros(G, dtype=torch.int64, device=sweep.device)
    nb = torch.zeros(G, dtype=torch.int64, device=sweep.device)
    T = params.t_frame + params.delay
    y_out = torch.empty((G * F, params.n_sub, sweep.n_out), dtype=torch.float64, device=sweep.device)
    off = rank * G                  # global index of this rank's first block: noise and weight sets follow it
    sweep.train(data["pilot_y"], data["pilot_x"], seed=1, group_offset=off)      # W_out exists for predict-only mode
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

    def step(i, timed):
        if not predict_only:
            sweep.train(data["pilot_y"], data["pilot_x"], seed=i, group_offset=off)
        U = torch.view_as_real(data["data_y"]).reshape(G * F, params.t_frame, sweep.n_in)
        if timed:
            ev[i][0].record()
        y = sweep.bank.predict(U, F, T=T, transient=params.delay + params.cp, precision=precision,
                               noise_mode="counter", seed=i, out=y_out, group_offset=off)
        if timed:
            ev[i][1].record()
        sweep.bank.detect_count(y, data["data_bits"], sweep.p_i, F, params.n_sub, params.n_t, params.m,
                                err=err, bits=nb)

    for i in range(warmup):
        step(i % max(steps, 1), False)
    err.zero_(); nb.zero_()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
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
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if steps else float("nan")
    frames = G * F
    fpf = flop_per_frame(n_res, sweep.n_in, sweep.n_out, T)
    achieved = fpf * frames / (kernel_ms * 1e-3) / 1e12
    c = counters.cpu().numpy()[0]
    return dict(value=world * frames * steps / dt, ms_per_step=1e3 * dt / max(steps, 1),
                predict_kernel_ms=kernel_ms, predict_only_symbols_per_s=world * frames / (kernel_ms * 1e-3),
                achieved_tflops=achieved, frac=achieved / PEAK_TFLOPS[precision], peak=PEAK_TFLOPS[precision],
                ber=float(c[0]) / max(float(c[1]), 1.0), fit_groups_flagged=int(sweep.bank.fit_status.sum().item()),
                frames_per_step=frames, blocks=G, fit_precision=fit_precision, flop_per_frame=fpf,
                n_in=sweep.n_in, n_out=sweep.n_out, T=T)


def run_sweep(torch, params, *, precision, fit_precision, n_res, F, solve, reservoirs="shared", points=(6.0, 12.0, 18.0),
              chunks_per_point=3):
    """What a user of the harness gets: `DetectorSweep.run` over `points` Eb/No values, wall clock around the
    whole call -- frame generation, training, detection, every host synchronisation and the final read-back of
    the counters included (nothing is resident beforehand except the packed reservoir)."""
    from esn_ofdm_mimo_amd.montecarlo import DetectorSweep
    if fit_precision == "auto":
        fit_precision = precision if precision in ("f16", "bf16") else ("f64" if precision == "f64" else "f32")
    sweep = DetectorSweep(params, n_reservoir=n_res, noise=0.001, seed=1234, precision=precision,
                          fit_precision=fit_precision, reservoirs=reservoirs, solve_method=solve)
    chunk = sweep.default_chunk_blocks(F)
    blocks = chunk * chunks_per_point
    sweep.run([points[0]], chunk, frames_per_block=F)            # warm-up: allocator, packed images, code objects
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ber, counts = sweep.run(list(points), blocks, frames_per_block=F)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frames = len(points) * blocks * F
    return dict(value=frames / dt, unit="OFDM symbols/s", wall_s=dt, frames=frames, ebno_db=list(points),
                blocks_per_point=blocks, chunk_blocks=chunk, ber=[float(b) for b in ber],
                fits_repaired=int(sweep.fits_repaired), reservoir=reservoirs,
                includes="frame generator + train + predict + detect + host syncs + counter read-back")


def predict_kernel_name(precision, n_res):
    """Name of the kernel the roofline record is about (csrc/esn_api.hip picks it from precision and shape)."""
    if precision == "f64":
        return "esn::recur_f64_mfma_kernel (predict)"
    if precision in ("f16", "bf16") and 256 < n_res <= 512 and os.environ.get("ESN_S16") != "0":
        return "esn::recur_skew16_kernel (predict, v_mfma_f32_16x16x32)"
    return "esn::recur_mfma_kernel (predict)"


def pmc_traffic(precision, G, F):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC pass taken on THIS
    workload (tools/pmc_traffic.py on the GPU box: FETCH_SIZE and WRITE_SIZE in separate passes,
    corrected as MI355X_MICROARCH.md prescribes).  Returns (bytes or None, provenance or None)."""
    import glob
    best = (None, None)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s.json" % precision))):
        try:
            pj = json.load(open(path))
        except Exception:
            continue
        if pj.get("blocks") == G and pj.get("frames_per_block", F) == F and pj.get("traffic_bytes_per_launch"):
            best = (pj["traffic_bytes_per_launch"],
                    {"file": os.path.relpath(path, ROOT), "launches_averaged": pj.get("FETCH_SIZE_launches"),
                     "kernel_match": pj.get("kernel_match"), "read_bytes": pj.get("read_bytes_per_launch"),
                     "write_bytes": pj.get("write_bytes_per_launch")})
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="f16", choices=["f32", "f16", "bf16", "f64"])
    ap.add_argument("--fit-precision", default="auto", choices=["auto", "f16", "bf16", "f32", "f64"],
                    help="harvest arithmetic; auto = the predict precision for f16/bf16 (states are then rounded\n"
                         "the same way at train and detect time), f64 for f64, f32 otherwise")
    ap.add_argument("--n-res", type=int, default=512)
    ap.add_argument("--blocks", type=int, default=0, help="coherence blocks per rank per step (0 = auto)")
    ap.add_argument("--frames-per-block", type=int, default=0, help="0 = L of the reference (75 at N=128)")
    ap.add_argument("--ebno", type=float, default=12.0)
    ap.add_argument("--reservoirs", default="shared", choices=["shared", "per_block"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-blocks", type=int, default=24, help="cpu_baseline single-process sample: blocks of L frames (~10 s)")
    ap.add_argument("--no-extra", action="store_true", help="skip the f32 / f64 / per-block sub-records")
    ap.add_argument("--predict-only", action="store_true", help="time the predict+detect leg only")
    ap.add_argument("--solve", default="auto", choices=["auto", "qr", "chol"])
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing test: rendezvous, sharding offsets, barrier, counter all_reduce and MAX over ranks "
                         "with NO compute (gloo on CPU); prints a line with \"dry_run\": true and value null")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0 and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))           # before any GPU call in this process
    world = max(world, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")

    if args.dry_run:
        return dry_run(args, rank, world)

    from esn_ofdm_mimo_amd.montecarlo import LinkParams
    params = LinkParams()                               # 4x8 TDL-B, N=128, 16-QAM
    F = args.frames_per_block or params.coherence_symbols

    cpu = None
    if not args.no_cpu_baseline and args.cpu_blocks > 0 and world == 1 and not profiler_attached():
        # once, at N=1, before this process touches the GPU or BLAS
        cpu = cpu_baseline((params.n_t, params.n_r, params.n_sub, params.m), args.n_res, args.ebno, args.cpu_blocks, F)

    import torch
    import torch.distributed as dist
    from esn_ofdm_mimo_amd import _lib
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # auto: a whole number of workgroup tiles per CU (tiles = G * ceil16(F) / tile), ~5 rounds
    fpad = ((F + 15) // 16) * 16
    G = args.blocks or max(1, (5 * 256 * 128) // fpad)
    common = dict(n_res=args.n_res, F=F, ebno=args.ebno, solve=args.solve, rank=rank, world=world)
    head = run_config(torch, dist, params, precision=args.precision, fit_precision=args.fit_precision,
                      reservoirs=args.reservoirs, G=G, steps=args.steps, warmup=args.warmup,
                      predict_only=args.predict_only, **common)

    extra_prec, extra_res = {}, {}
    if not args.no_extra and world == 1 and not args.predict_only:
        def sub(rec, **note):
            keep = ("value", "ms_per_step", "predict_kernel_ms", "achieved_tflops", "peak", "frac", "ber", "blocks",
                    "frames_per_step", "fit_precision", "fit_groups_flagged")
            out = {k: rec[k] for k in keep}
            out.update(unit="OFDM symbols/s", **note)
            return out
        for prec, g_sub in (("f32", G), ("f64", max(1, G // 4))):     # f64: 512 blocks = 1280 tiles of 32 frames = 5 per CU
            if prec == args.precision:
                continue
            rec = run_config(torch, dist, params, precision=prec, fit_precision="auto", reservoirs="shared",
                             G=g_sub, steps=3, warmup=1, **common)
            extra_prec[prec] = sub(rec, dtype=DTYPE_NAME[prec], reservoir="shared", steps=3, warmup=1)
        if args.reservoirs == "shared":
            rec = run_config(torch, dist, params, precision=args.precision, fit_precision=args.fit_precision,
                             reservoirs="per_block", G=G, steps=3, warmup=1, **common)
            extra_res["per_block"] = sub(rec, dtype=DTYPE_NAME[args.precision], steps=3, warmup=1,
                                         reservoir="per_block: weight set = block index mod 8 (pool of 8 pre-drawn "
                                                   "reservoirs; the reference draws one per block, SURVEY F5)")

    large = None
    if not args.no_extra and world == 1 and not args.predict_only and args.n_res != 2048:
        g_big = 512                                   # 512 x 80 slots = 160 frame tiles x 8 row tiles = 5 workgroups per CU
        rec = run_config(torch, dist, params, precision=args.precision, fit_precision=args.fit_precision,
                         reservoirs="shared", n_res=2048, G=g_big, steps=2, warmup=1, F=F, ebno=args.ebno,
                         solve=args.solve, rank=rank, world=world)
        large = {k: rec[k] for k in ("value", "ms_per_step", "predict_kernel_ms", "achieved_tflops", "peak", "frac", "ber",
                                     "blocks", "frames_per_step", "fit_precision", "fit_groups_flagged", "flop_per_frame")}
        large.update(unit="OFDM symbols/s", dtype=DTYPE_NAME[args.precision], steps=2, warmup=1,
                     workload="configs[4]: 4x8 TDL-B, N=128, N_res=2048, shared reservoir",
                     kernel="esn::big_prep_kernel + esn::big_step_kernel, one pair of launches per timestep "
                            "(predict_kernel_ms spans all of them)")

    sweep_rec = None
    if not args.no_extra and world == 1 and not args.predict_only:
        sweep_rec = run_sweep(torch, params, precision=args.precision, fit_precision=args.fit_precision,
                              n_res=args.n_res, F=F, solve=args.solve)
        sweep_rec["vs_headline"] = sweep_rec["value"] / head["value"]

    traffic, traffic_src = pmc_traffic(args.precision, G, F)
    if rank == 0:
        out = {
            "metric": "OFDM symbols/s through the ESN detector (4x8 TDL-B, N=128, N_res=%d; train+predict+detect)" % args.n_res,
            "value": head["value"],
            "unit": "OFDM symbols/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_NAME[args.precision],
            "data": "synthetic",
            "config": {"workload": "configs[3]: OFDM 4x8 MIMO, TDL-B taps, 16-QAM, N=128, CP=7, d=3, N_res=%d, "
                                   "Eb/No %g dB, uncoded" % (args.n_res, args.ebno),
                       "blocks_per_rank": G, "frames_per_block": F, "frames_per_step": world * head["frames_per_step"],
                       "reservoir": args.reservoirs, "state_noise": 0.001,
                       "fit": "harvest %s + %s f64 solve" % (head["fit_precision"], args.solve),
                       "timed": "predict+detect" if args.predict_only else "train+predict+detect",
                       "parallelism": "blocks sharded over %d rank(s), one all_reduce of counters" % world},
            "ber": head["ber"],
            "fit_groups_flagged": head["fit_groups_flagged"],
            "predict_kernel_ms": head["predict_kernel_ms"],
            "predict_only_symbols_per_s": head["predict_only_symbols_per_s"],
            "roofline": {"bound": "mfma", "kernel": predict_kernel_name(args.precision, args.n_res),
                         "achieved": head["achieved_tflops"],
                         "peak": head["peak"], "unit": "TFLOP/s", "frac": head["frac"], "traffic": traffic,
                         "traffic_unit": "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": head["frames_per_step"] * 8 * (params.t_frame * head["n_in"] + params.n_sub * head["n_out"]),
                         "flop_per_frame": head["flop_per_frame"],
                         "frames_per_launch": head["frames_per_step"]},
            "device": _lib.device_info(),
        }
        if extra_prec:
            out["precisions"] = extra_prec
        if extra_res:
            out["reservoirs"] = extra_res
        if large:
            out["large_reservoir"] = large
        if sweep_rec:
            out["sweep"] = sweep_rec
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def dry_run(args, rank, world):
    """No compute: the launcher, rendezvous, block sharding, barriers, the counters' all_reduce and the
    MAX over ranks, on gloo / CPU.  Used by tests/test_bench_launch_cpu.py; never a benchmark."""
    import torch
    import torch.distributed as dist
    from esn_ofdm_mimo_amd.montecarlo import reduce_counters
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    G = args.blocks or 4
    first_block = rank * G                                   # the offset run_config hands to blocks_fast
    counters = torch.tensor([[first_block + 1, G]], dtype=torch.int64)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    reduce_counters(counters, dist if world > 1 else None, world)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        print(json.dumps({"dry_run": True, "value": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "first_block_sum": int(counters[0, 0]), "blocks_total": int(counters[0, 1]),
                          "ms_per_step": 1e3 * dt}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
