"""Monte-Carlo harness around the ESN detector: the callers either side of the hot path
(SURVEY 8f), rebuilt batched and device-resident.

    LinkParams      constants of a driver configuration
                    (Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:182-238, :285-288)
    FrameSource     bits -> QAM -> N*ifft -> CP -> sqrt(Pi) -> PA -> per-link 8-tap FIR -> AWGN
                    (:323-356 pilot, :397-427 data), TDL-B taps (:127-177) or the exponential-PDP
                    Rayleigh taps of OFDM_MIMO_2-2_NBF_LDPC.py:162-164,272-279.  Round 1 builds the
                    frames with torch tensor ops on the device (a stand-in: the generator is row f-1
                    of the scope table, next to move into HIP kernels); every random stream is keyed
                    by (seed, snr index, block index), so a block is identical on any rank.
    DetectorSweep   per SNR point: G coherence blocks at a time -> one harvest + one solve launch
                    (training, helper_mimo_esn_generic.py:58-86), one predict launch over G*L data
                    frames, one fused detect/count launch; int64 counters [n_snr, {err, bits}]
                    reduced over ranks with a single all_reduce (RCCL) at the end.

The reference redraws a reservoir per coherence block from the global RNG; the sweep
supports that ("per_block" reservoirs from a pre-drawn pool) and the shared-reservoir mode the
throughput target assumes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from . import _lib
from .batched import ReservoirBank

TDLB_NORM_DELAYS = (0.0000, 0.1072, 0.2155, 0.2095, 0.2870, 0.2986, 0.3752, 0.5055, 0.3681,
                    0.3697, 0.5700, 0.5283, 1.1021, 1.2756, 1.5474, 1.7842, 2.0169, 2.8294,
                    3.0219, 3.6187, 4.1067, 4.2790, 4.7834)
TDLB_POW_DB = (0.0, -2.2, -4.0, -3.2, -9.8, -1.2, -3.4, -5.2, -7.6, -3.0, -8.9, -9.0,
               -4.8, -5.7, -7.5, -1.9, -7.6, -12.2, -9.8, -11.4, -14.9, -9.2, -11.3)


@dataclass
class LinkParams:
    n_t: int = 4
    n_r: int = 8
    n_sub: int = 128
    m: int = 4
    isi: int = 8
    fs: float = 2 * 1.024e6
    no: float = 1e-5
    clip_db: float = 3.0
    ds_ns: float = 300.0
    input_scaler: float = 0.005
    teacher_scale: float = 5e-7
    min_delay: int = 0
    f_d: float = 100.0
    channel: str = "tdlb"         # "tdlb" | "exp" | "awgn"

    @property
    def cp(self):
        return self.isi - 1

    @property
    def max_delay(self):
        return int(math.ceil(self.isi / 2) + 2)

    @property
    def delay(self):
        return (self.min_delay + self.max_delay) // 2

    @property
    def t_frame(self):
        return self.n_sub + self.cp

    @property
    def coherence_symbols(self):
        t_sym = (self.n_sub + self.isi - 1) / self.fs
        return max(1, math.floor((0.5 / max(self.f_d, 1e-9)) / t_sym))

    def p_i(self, ebno_db):
        return (10 ** (ebno_db / 10)) * self.no

    def var_x(self, ebno_db):
        return (10 ** (ebno_db / 10)) * self.no * self.n_sub

    def a_clip(self, ebno_db):
        return math.sqrt(self.var_x(ebno_db)) * 10 ** (self.clip_db / 20)

    def input_scaling(self, ebno_db):
        return self.input_scaler / math.sqrt(self.var_x(ebno_db))


def unit_qam_table(m):
    """Unit-power square QAM, index = i*side + j <-> (pam[i], pam[j])  (driver:17-28)."""
    side = math.ceil(math.sqrt(2 ** m) / 2) * 2
    pam = np.arange(-(side - 1), side, 2).astype(float)
    re, im = np.meshgrid(pam, pam, indexing="ij")
    c = (re + 1j * im).reshape(-1)
    return c / math.sqrt(np.mean(np.abs(c) ** 2))


class FrameSource:
    def __init__(self, params: LinkParams, device=None, seed=0):
        torch = _lib.require_gpu()
        self.torch, self.p = torch, params
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.seed = int(seed)
        self.const = torch.as_tensor(unit_qam_table(params.m), device=self.device)
        self.pow2 = (2 ** torch.arange(params.m, device=self.device)).view(1, 1, params.m, 1)
        # TDL-B: h = g @ M with fixed split of each path between floor/ceil taps (driver:139-165)
        p_lin = 10.0 ** (np.array(TDLB_POW_DB) / 10.0)
        p_lin /= p_lin.sum()
        d = np.array(TDLB_NORM_DELAYS) * params.ds_ns * 1e-9 * params.fs
        M = np.zeros((len(p_lin), params.isi))
        for k in range(len(p_lin)):
            i0 = int(np.floor(d[k])); frac = d[k] - i0
            if 0 <= i0 < params.isi:
                M[k, i0] += 1.0 - frac
            if 0 <= i0 + 1 < params.isi:
                M[k, i0 + 1] += frac
        self._tdl_M = torch.as_tensor(M * np.sqrt(p_lin)[:, None], device=self.device).to(torch.complex128)
        pdp = np.exp(-np.arange(params.isi) / max(params.cp / 9, 1e-12))
        self._exp_pdp = torch.as_tensor(np.sqrt(pdp / pdp.sum()), device=self.device)

    def _gen(self, *key):
        g = self.torch.Generator(device=self.device)
        h = self.seed
        for k in key:
            h = (h * 1000003 + int(k) + 0x9E3779B9) % (2 ** 63 - 1)
        g.manual_seed(h)
        return g

    def _cn(self, shape, gen):
        torch = self.torch
        re = torch.randn(shape, generator=gen, device=self.device, dtype=torch.float64)
        im = torch.randn(shape, generator=gen, device=self.device, dtype=torch.float64)
        return torch.complex(re, im)

    def taps(self, n_blocks, gen):
        """[G, n_r, n_t, isi] complex128, i.i.d. links, unit energy per link (TDL-B)."""
        torch, p = self.torch, self.p
        if p.channel == "tdlb":
            g = self._cn((n_blocks, p.n_r, p.n_t, self._tdl_M.shape[0]), gen) / math.sqrt(2.0)
            h = g @ self._tdl_M
            e = (h.abs() ** 2).sum(-1, keepdim=True)
            return h / torch.sqrt(torch.where(e > 0, e, torch.ones_like(e)))
        if p.channel == "exp":
            return self._cn((n_blocks, p.n_r, p.n_t, p.isi), gen) / math.sqrt(2.0) * self._exp_pdp
        if p.channel == "awgn":                      # flat unit-modulus channel (SISO driver :205-206)
            h = self._cn((n_blocks, p.n_r, p.n_t, 1), gen)
            h = h / h.abs()
            return torch.cat([h, torch.zeros((n_blocks, p.n_r, p.n_t, p.isi - 1), dtype=h.dtype,
                                             device=self.device)], -1)
        raise ValueError(p.channel)

    def transmit(self, n_frames, ebno_db, gen):
        """bits [B, N*m, n_t] uint8, x_cp (pre-PA) and x_pa (post-PA) [B, T, n_t] complex128."""
        torch, p = self.torch, self.p
        bits = (torch.rand((n_frames, p.n_sub * p.m, p.n_t), generator=gen, device=self.device) > 0.5)
        idx = (bits.view(n_frames, p.n_sub, p.m, p.n_t).to(torch.int64) * self.pow2).sum(2)
        x_f = self.const[idx]                                             # [B, N, n_t]
        x_t = p.n_sub * torch.fft.ifft(x_f, dim=1)
        if p.cp > 0:
            x_t = torch.cat([x_t[:, -p.cp:], x_t], dim=1)
        x_cp = x_t * math.sqrt(p.p_i(ebno_db))
        x_pa = x_cp / torch.sqrt(1 + (x_cp.abs() / p.a_clip(ebno_db)) ** 2)
        return bits.to(torch.uint8), x_cp, x_pa

    def receive(self, x_pa, taps, frames_per_block, gen):
        """y[b,t,rx] = sum_tx sum_k c[rx,tx,k] x[b,t-k,tx] + sqrt(T No/2)(randn + j randn)."""
        torch, p = self.torch, self.p
        b, t, _ = x_pa.shape
        g = taps.shape[0]
        xp = torch.cat([torch.zeros((b, p.isi - 1, p.n_t), dtype=x_pa.dtype, device=self.device), x_pa], 1)
        win = xp.unfold(1, p.isi, 1).flip(-1)                            # [B, T, n_t, isi]: x[t-k]
        win = win.reshape(g, frames_per_block, t, p.n_t * p.isi)
        c = taps.reshape(g, p.n_r, p.n_t * p.isi).transpose(1, 2)         # [G, n_t*isi, n_r]
        y = torch.matmul(win.reshape(g, frames_per_block * t, -1), c).reshape(b, t, p.n_r)
        return y + math.sqrt(t * p.no / 2) * self._cn((b, t, p.n_r), gen)

    def blocks(self, ebno_db, snr_idx, block_ids, frames_per_block):
        """Pilot + data frames of the given coherence blocks.  Returns a dict of device tensors:
        pilot_y [G,T,n_r], pilot_x [G,T,n_t] (pre-PA teacher), data_y [G*F,T,n_r], data_bits."""
        torch = self.torch
        outs = []
        for bid in block_ids:                     # one generator per block: rank-independent
            gen = self._gen(snr_idx, bid)
            taps = self.taps(1, gen)
            _, px, ppa = self.transmit(1, ebno_db, gen)
            py = self.receive(ppa, taps, 1, gen)
            bits, _, dpa = self.transmit(frames_per_block, ebno_db, gen)
            dy = self.receive(dpa, taps, frames_per_block, gen)
            outs.append((py, px, dy, bits))
        return dict(pilot_y=torch.cat([o[0] for o in outs]), pilot_x=torch.cat([o[1] for o in outs]),
                    data_y=torch.cat([o[2] for o in outs]), data_bits=torch.cat([o[3] for o in outs]))

    def blocks_fast(self, ebno_db, snr_idx, first_block, n_blocks, frames_per_block):
        """Same recipe, all blocks from ONE generator keyed by (snr, first_block) -- for benchmarks,
        where only shapes and statistics matter (not rank-independent per block)."""
        gen = self._gen(snr_idx, first_block, n_blocks)
        taps = self.taps(n_blocks, gen)
        _, px, ppa = self.transmit(n_blocks, ebno_db, gen)
        py = self.receive(ppa, taps, 1, gen)
        bits, _, dpa = self.transmit(n_blocks * frames_per_block, ebno_db, gen)
        dy = self.receive(dpa, taps, frames_per_block, gen)
        return dict(pilot_y=py, pilot_x=px, data_y=dy, data_bits=bits)


def _view_real(z):
    """complex128 [..., T, n] -> float64 view [..., T, 2n] (Re/Im interleaved; driver:433-436)."""
    import torch
    z = z.contiguous()
    return torch.view_as_real(z).reshape(*z.shape[:-1], 2 * z.shape[-1])


complex_as_io = _view_real


def blocks_for_rank(rank, world_size, n_blocks):
    """Round-robin deal of coherence blocks to ranks: the union over ranks is
    range(n_blocks) and block b's random streams depend on b only, never on the rank."""
    return list(range(rank, n_blocks, world_size))


def reduce_counters(counters, dist=None, world_size=1):
    """The path's only collective: one all_reduce(SUM) of the int64 [n_snr, 2] (errors, bits)
    tensor (RCCL over xGMI on GPUs, gloo on CPU).  Integer sums are order independent, so the
    result is bit-identical for any world size."""
    if dist is not None and world_size > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
    return counters


def draw_reservoir(n_in, n_out, n_res, spectral_radius, sparsity, seed):
    """(W, W_in, W_feedb) in the reference's draw order (pyESN.py:93-109), host-side init."""
    rs = np.random.RandomState(seed)
    w = rs.rand(n_res, n_res) - 0.5
    w[rs.rand(n_res, n_res) < sparsity] = 0
    w *= spectral_radius / np.max(np.abs(np.linalg.eigvals(w)))
    return w, rs.rand(n_res, n_in) * 2 - 1, rs.rand(n_res, n_out) * 2 - 1


class DetectorSweep:
    """SNR sweep x Monte-Carlo blocks, sharded over ranks by block."""

    def __init__(self, params: LinkParams, n_reservoir=512, spectral_radius=0.9, sparsity=0.1, noise=0.001,
                 seed=0, precision="f32", fit_precision="f64", reservoirs="shared", pool=8, device=None,
                 rank=0, world_size=1, solve_method="auto"):
        torch = _lib.require_gpu()
        self.torch, self.p = torch, params
        self.rank, self.world = rank, world_size
        self.precision, self.fit_precision = precision, fit_precision
        self.solve_method = solve_method
        self.n_in, self.n_out, self.n_res = 2 * params.n_r, 2 * params.n_t, n_reservoir
        self.seed = seed
        self.src = FrameSource(params, device, seed)
        self.device = self.src.device
        n_sets = 1 if reservoirs == "shared" else int(pool)
        ws = [draw_reservoir(self.n_in, self.n_out, n_reservoir, spectral_radius, sparsity, seed * 7919 + 17 + i)
              for i in range(n_sets)]
        self.bank = ReservoirBank(self.n_in, self.n_out, n_reservoir, np.stack([w[0] for w in ws]),
                                  np.stack([w[1] for w in ws]), np.stack([w[2] for w in ws]),
                                  teacher_forcing=True, noise=noise, device=self.device)
        self.n_sets = n_sets

    def set_snr(self, ebno_db, n_groups):
        torch, p = self.torch, self.p
        ones_in = torch.ones((n_groups, self.n_in), dtype=torch.float64, device=self.device)
        ones_out = torch.ones((n_groups, self.n_out), dtype=torch.float64, device=self.device)
        self.bank.set_scaling(ones_in * p.input_scaling(ebno_db), None, ones_out * p.teacher_scale, None)
        self.p_i = torch.full((n_groups,), p.p_i(ebno_db), dtype=torch.float64, device=self.device)

    def train(self, pilot_y, pilot_x, seed=0):
        """helper_mimo_esn_generic.py:58-86 for G blocks: delay d, nForget = d + CP, one harvest + solve."""
        torch, p = self.torch, self.p
        d = p.delay
        g, t = pilot_y.shape[0], pilot_y.shape[1]
        U = torch.zeros((g, t + d, self.n_in), dtype=torch.float64, device=self.device)
        D = torch.zeros((g, t + d, self.n_out), dtype=torch.float64, device=self.device)
        U[:, :t] = _view_real(pilot_y)
        D[:, d:d + t] = _view_real(pilot_x)
        self._fit_io = (U, D, d + p.cp)
        return self.bank.fit(U, D, transient=d + p.cp, precision=self.fit_precision, noise_mode="counter",
                             seed=seed, method=self.solve_method)

    def repair_fit(self, E):
        """Host-synchronising check of the last fit: groups the Cholesky path flagged are re-solved
        with the QR kernel (GPU).  Returns how many were."""
        U, D, tr = self._fit_io
        n = self.bank.resolve_failed(E, D, tr, self.bank.W_out, self.bank.fit_status)
        if n:
            self.bank.set_readout(self.bank.W_out)
        return n

    def detect(self, data_y, data_bits, frames_per_block, err, bits, seed=0, out=None):
        """driver:433-456 for all data frames of G blocks: predict (d trailing zero rows synthesised
        in-kernel) -> fused FFT/slicer/count."""
        p = self.p
        U = _view_real(data_y)
        y = self.bank.predict(U, frames_per_block, T=p.t_frame + p.delay, transient=p.delay + p.cp,
                              precision=self.precision, noise_mode="counter", seed=seed, out=out)
        self.bank.detect_count(y, data_bits, self.p_i, frames_per_block, p.n_sub, p.n_t, p.m, err=err, bits=bits)
        return y

    def run(self, ebno_list, blocks_per_snr, frames_per_block=None, chunk_blocks=64, dist=None):
        """Returns BER[n_snr] (identical on every rank).  Blocks are dealt round-robin to ranks."""
        torch = self.torch
        F = frames_per_block or self.p.coherence_symbols
        n_snr = len(ebno_list)
        counters = torch.zeros((n_snr, 2), dtype=torch.int64, device=self.device)
        for si, ebno in enumerate(ebno_list):
            mine = blocks_for_rank(self.rank, self.world, blocks_per_snr)
            for c0 in range(0, len(mine), chunk_blocks):
                ids = mine[c0:c0 + chunk_blocks]
                g = len(ids)
                data = self.src.blocks(ebno, si, ids, F)
                self.set_snr(ebno, g)
                E = self.train(data["pilot_y"], data["pilot_x"], seed=self.seed + 1000 * si + ids[0])
                self.repair_fit(E)
                err = torch.zeros(g, dtype=torch.int64, device=self.device)
                nb = torch.zeros(g, dtype=torch.int64, device=self.device)
                self.detect(data["data_y"], data["data_bits"], F, err, nb, seed=self.seed + 1000 * si + ids[0])
                counters[si, 0] += err.sum()
                counters[si, 1] += nb.sum()
        reduce_counters(counters, dist, self.world)
        c = counters.cpu().numpy()
        return c[:, 0] / np.maximum(c[:, 1], 1), c
