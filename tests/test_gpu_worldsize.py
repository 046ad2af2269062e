"""Sweep results do not depend on world size or chunking (SURVEY 4 tier 4 / 8e), with the state noise ON.

Every random stream of the sweep is keyed by global indices: the frame generator by (seed, snr, global block /
frame), the state noise by (seed, snr, leg, GLOBAL frame, step, row) through `group_offset` of
esn_predict_batch / esn_harvest_batch, the weight set of per-block reservoirs by the global block.  So the int64
counters of one 1-rank sweep must equal, bit for bit, the sum of the two ranks of a 2-rank sweep (run back to
back on the one GPU of the box) and a sweep cut into different chunks.  The same file pushes the counters
through a 1-rank `nccl` (= RCCL) process group: the collective branch of `reduce_counters` on a device tensor."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EBNO = [6.0, 15.0]
BLOCKS, F = 22, 9                    # 22 blocks: ragged halves of the pool of 8, chunks of 5 leave a tail


def _sweep(reservoirs, rank=0, world=1, precision="f16", n_res=100, **kw):
    from esn_ofdm_mimo_amd.montecarlo import DetectorSweep, LinkParams
    fit = precision if precision in ("f16", "f64") else "f32"
    return DetectorSweep(LinkParams(), n_reservoir=n_res, noise=0.001, seed=7, precision=precision,
                         fit_precision=fit, reservoirs=reservoirs, pool=8, rank=rank, world_size=world, **kw)


@pytest.mark.parametrize("reservoirs", ("shared", "per_block"))
@pytest.mark.parametrize("precision", ("f16", "f64"))
def test_counters_independent_of_world_size_and_chunking(reservoirs, precision):
    _, one = _sweep(reservoirs, precision=precision).run(EBNO, BLOCKS, frames_per_block=F, chunk_blocks=BLOCKS)
    assert one[:, 1].min() == BLOCKS * F * 128 * 4 * 4 and one[:, 0].min() > 0
    # two ranks, run back to back: same counters in total
    parts = [_sweep(reservoirs, rank=r, world=2, precision=precision).run(EBNO, BLOCKS, frames_per_block=F)[1]
             for r in range(2)]
    np.testing.assert_array_equal(parts[0] + parts[1], one)
    assert parts[0][:, 1].min() > 0 and parts[1][:, 1].min() > 0
    # three ranks with ragged shares
    parts = [_sweep(reservoirs, rank=r, world=3, precision=precision).run(EBNO, BLOCKS, frames_per_block=F)[1]
             for r in range(3)]
    np.testing.assert_array_equal(sum(parts), one)
    # other chunkings of the one-rank sweep
    for chunk in (5, 16):
        _, c = _sweep(reservoirs, precision=precision).run(EBNO, BLOCKS, frames_per_block=F, chunk_blocks=chunk)
        np.testing.assert_array_equal(c, one)


def test_noise_and_weight_set_follow_the_global_group():
    """The kernels themselves: groups [4, 16) of a 24-group launch == a 12-group launch with group_offset = 4 (same
    outputs bit for bit, counter noise on, four weight sets), predict and harvest, fp16 / f32 / f64.  (Both
    launches hold more than 8 sequences: float64 batches of up to 8 run on the vector-ALU kernel, whose
    summation order differs from the matrix-pipe kernel's at the 1e-16 level.)"""
    import torch
    from esn_ofdm_mimo_amd import batched
    from oracle import esn_oracle as eo
    rs = np.random.RandomState(3)
    n_in, n_out, n_res, G, Fr, T = 4, 4, 64, 24, 7, 20
    ws = [eo.draw_weights(np.random.RandomState(10 + i), n_in, n_out, n_res, 0.9, 0.1) for i in range(4)]
    bank = batched.ReservoirBank(n_in, n_out, n_res, np.stack([w[0] for w in ws]), np.stack([w[1] for w in ws]),
                                 np.stack([w[2] for w in ws]), noise=0.01)
    u = rs.randn(G * Fr, T, n_in) * 0.3
    w_out = rs.randn(G, n_out, n_res + n_in) * 0.01
    up, dp = rs.randn(G, T, n_in) * 0.3, np.tanh(rs.randn(G, T, n_out))
    lo, hi = 4, 16
    for prec in ("f16", "f32", "f64"):
        bank.set_readout(w_out)
        full = bank.predict(u, Fr, precision=prec, noise_mode="counter", seed=11).cpu().numpy()
        bank.set_readout(w_out[lo:hi])
        part = bank.predict(u[lo * Fr:hi * Fr], Fr, precision=prec, noise_mode="counter", seed=11,
                            group_offset=lo).cpu().numpy()
        np.testing.assert_array_equal(part, full[lo * Fr:hi * Fr])
        wrong = bank.predict(u[lo * Fr:hi * Fr], Fr, precision=prec, noise_mode="counter", seed=11).cpu().numpy()
        assert np.abs(wrong - part).max() > 0                      # without the offset: other noise, other weights
        e_full = bank.harvest(up, dp, precision=prec, noise_mode="counter", seed=5).cpu().numpy()
        e_part = bank.harvest(up[lo:hi], dp[lo:hi], precision=prec, noise_mode="counter", seed=5,
                              group_offset=lo).cpu().numpy()
        np.testing.assert_array_equal(e_part, e_full[lo:hi])
    torch.cuda.synchronize()


def test_large_reservoir_path_follows_the_global_group():
    """N_res = 2048 (one GEMM launch per timestep): a chunk with group_offset equals the same groups of the whole."""
    from esn_ofdm_mimo_amd import batched
    from oracle import esn_oracle as eo
    rs = np.random.RandomState(4)
    n_in, n_out, n_res, G, Fr, T = 16, 8, 2048, 4, 40, 12
    w = eo.draw_weights(np.random.RandomState(2), n_in, n_out, n_res, 0.9, 0.1)
    bank = batched.ReservoirBank(n_in, n_out, n_res, *w, noise=0.01)
    u = rs.randn(G * Fr, T, n_in) * 0.05
    w_out = rs.randn(G, n_out, n_res + n_in) * 0.002
    bank.set_readout(w_out)
    full = bank.predict(u, Fr, precision="f16", noise_mode="counter", seed=3).cpu().numpy()
    bank.set_readout(w_out[2:])
    part = bank.predict(u[2 * Fr:], Fr, precision="f16", noise_mode="counter", seed=3, group_offset=2).cpu().numpy()
    np.testing.assert_array_equal(part, full[2 * Fr:])


def test_counters_through_a_one_rank_rccl_group():
    """`reduce_counters` on a device int64 tensor through torch.distributed's nccl backend (RCCL): world size 1
    still creates the communicator and runs the all_reduce kernel path the N-GPU benchmark uses."""
    import torch
    import torch.distributed as dist
    from esn_ofdm_mimo_amd.montecarlo import reduce_counters
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        c = torch.tensor([[123456789012, 7], [5, 11]], dtype=torch.int64, device="cuda:0")
        want = c.clone()
        reduce_counters(c, dist, 1)          # a process group was handed in: the collective runs, one rank
        t = torch.tensor([3.5], dtype=torch.float64, device="cuda:0")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                    # the bench's MAX over ranks
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.equal(c, want) and float(t) == 3.5
        # and a whole sweep reduced through it
        sw = _sweep("shared")
        ber, counts = sw.run([12.0], 4, frames_per_block=5, dist=dist)
        assert counts[0, 1] == 4 * 5 * 128 * 16 and 0.05 < ber[0] < 0.5
    finally:
        dist.destroy_process_group()
