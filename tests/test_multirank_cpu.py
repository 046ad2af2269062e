"""world_size-2 gloo test of the multi-GPU path's host logic: contiguous block
ranges per rank, per-block random streams independent of rank, ONE all_reduce(SUM) of int64
counters -> summed error counts bit-identical to the single-process result.  The per-block
detector here is the CPU oracle on a tiny configuration (the GPU kernels are covered by the
-m gpu tests); what is under test is the partition + reduction the product uses."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SNR, N_BLOCKS, F = 2, 7, 2


def block_counts(snr_idx, block):
    """(errors, bits) of one coherence block; depends on (snr_idx, block) only."""
    from oracle import esn_oracle as eo
    from oracle.ofdm_frames import LinkConfig, make_frame, tdlb_mimo_taps
    cfg = LinkConfig(n_t=2, n_r=2, n_sub=16, m=2)
    ebno = 6.0 + 6.0 * snr_idx
    rs = np.random.RandomState(1000 * snr_idx + block)
    taps = tdlb_mimo_taps(cfg, 77 + 100 * snr_idx + block)
    esn = eo.OracleESN(4, 4, 24, spectral_radius=0.9, sparsity=0.1, noise=0.0,
                       input_scaling=cfg.input_scaling(ebno) * np.ones(4),
                       teacher_scaling=cfg.teacher_scale * np.ones(4), random_state=5)
    pilot = make_frame(cfg, ebno, taps, rs)
    ret = eo.train_mimo_esn(esn, 0, cfg.min_delay, cfg.max_delay, cfg.cp, cfg.n_sub, cfg.n_t, cfg.n_r,
                            cfg.isi, pilot["y_cp"], pilot["x_cp"])
    _, _, _, delay, _, d_min, d_max, forget, _ = ret
    const = eo.unit_qam(cfg.m)
    e = b = 0
    for _ in range(F):
        fr = make_frame(cfg, ebno, taps, rs)
        _, rx = eo.detect_frame(esn, fr["y_cp"], delay, d_min, d_max, forget, cfg.n_sub, cfg.n_t,
                                cfg.p_i(ebno), const, cfg.m)
        e += eo.count_bit_errors(fr["bits"], rx)
        b += rx.size
    return e, b


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    from esn_ofdm_mimo_amd.montecarlo import blocks_for_rank, reduce_counters
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    counters = torch.zeros((N_SNR, 2), dtype=torch.int64)
    for si in range(N_SNR):
        for blk in blocks_for_rank(rank, world, N_BLOCKS):
            e, b = block_counts(si, blk)
            counters[si, 0] += e
            counters[si, 1] += b
    reduce_counters(counters, dist, world)
    if rank == 0:
        np.save(out_path, counters.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_exact_cover():
    from esn_ofdm_mimo_amd.montecarlo import blocks_for_rank
    for world in (1, 2, 3, 8):
        got = sorted(b for r in range(world) for b in blocks_for_rank(r, world, 29))
        assert got == list(range(29))


@pytest.mark.timeout(300)
def test_two_rank_gloo_counts_equal_single_process(tmp_path):
    single = np.zeros((N_SNR, 2), dtype=np.int64)
    for si in range(N_SNR):
        for blk in range(N_BLOCKS):
            e, b = block_counts(si, blk)
            single[si] += (e, b)
    out = str(tmp_path / "c.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    np.testing.assert_array_equal(np.load(out), single)
    assert single[:, 1].min() > 0
