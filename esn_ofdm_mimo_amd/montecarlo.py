"""Monte-Carlo harness around the ESN detector: the callers either side of the hot path
(SURVEY 8f), rebuilt batched and device-resident.

    LinkParams      constants of a driver configuration
                    (Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:182-238, :285-288)
    FrameSource     bits -> QAM -> N*ifft -> CP -> sqrt(Pi) -> PA -> per-link 8-tap FIR -> AWGN
                    (:323-356 pilot, :397-427 data), TDL-B taps (:127-177) or the exponential-PDP
                    Rayleigh taps of OFDM_MIMO_2-2_NBF_LDPC.py:162-164,272-279: HIP kernels
                    (esn_gen_taps / esn_gen_frames, csrc/esn_gen.hip) with Philox counter streams keyed
                    by (seed, snr index, global block / frame index), so a block is identical on any rank.
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
from ._lib import check, ptr

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
    # driver variants (defaults = the north-star 4x8 driver: d = (Min+Max)//2, nForget = d + CP, fresh state)
    delay_fixed: int = -1         # >= 0: output delay d of the trainer (the SISO driver trains without delay)
    forget_fixed: int = -1        # >= 0: rows dropped from the fit and from every prediction
    continuation: bool = False    # True: every predict starts from the training-final state / teacher output
    coherence_fixed: int = 0      # > 0: data symbols per pilot instead of the Doppler formula

    @classmethod
    def siso_awgn(cls, n_sub=512, symbols_per_pilot=400):
        """Demo_SISO_QPSK_AWGN_LDPC_ESN_with_ZF_LS.py: 1x1, QPSK, N=512, CP=0 (:107-111), flat unit-modulus
        channel drawn once per Eb/No point (:203-206), `esn.fit(Ein, Eout)` with transient 0 and no output
        delay (:224-226), `esn.predict(ESN_input)` = continuation=True (:253-254), 400 symbols per pilot."""
        return cls(n_t=1, n_r=1, n_sub=n_sub, m=2, isi=1, channel="awgn", delay_fixed=0, forget_fixed=0,
                   continuation=True, coherence_fixed=symbols_per_pilot)

    @classmethod
    def block_fading(cls, n_t=2, n_r=2, n_sub=512):
        """OFDM_{SISO,SIMO_1-2,MIMO_2-2}_NBF_LDPC.py / Demo_MIMO_4x8_ChannelRank_..._fast.py: exponential-PDP
        Rayleigh taps exp(-k/(CP/9)) redrawn every coherence block (:162-164,:272-279), 16-QAM, same trainer
        as the 4x8 driver (delay (0+6)//2 = 3, nForget = 10)."""
        return cls(n_t=n_t, n_r=n_r, n_sub=n_sub, m=4, isi=8, channel="exp")

    @property
    def cp(self):
        return self.isi - 1

    @property
    def max_delay(self):
        return int(math.ceil(self.isi / 2) + 2)

    @property
    def delay(self):
        return self.delay_fixed if self.delay_fixed >= 0 else (self.min_delay + self.max_delay) // 2

    @property
    def forget(self):
        return self.forget_fixed if self.forget_fixed >= 0 else self.delay + self.cp

    @property
    def t_frame(self):
        return self.n_sub + self.cp

    @property
    def coherence_symbols(self):
        if self.coherence_fixed > 0:
            return self.coherence_fixed
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


class FrameSource:
    """HIP frame generator (esn_gen_taps / esn_gen_frames of include/esn_hip.h).  Counter-based
    random streams: frame f of block b at SNR index s is a pure function of (seed, s, b, f)."""

    CHANNEL_KIND = {"tdlb": 0, "exp": 1, "awgn": 2}

    def __init__(self, params: LinkParams, device=None, seed=0):
        torch = _lib.require_gpu()
        self.torch, self.p = torch, params
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.seed = int(seed)

    def _key(self, *parts):
        h = self.seed & (2 ** 64 - 1)
        for k in parts:
            h = (h * 6364136223846793005 + int(k) + 0x9E3779B97F4A7C15) % (2 ** 64)
        return h

    def taps(self, n_blocks, snr_idx, first_block, gains=None):
        """[G, n_r, n_t, isi] complex128.  Link l of block b draws from counter (first_block + b)."""
        torch, p = self.torch, self.p
        with torch.cuda.device(self.device):
            out = torch.empty((n_blocks, p.n_r, p.n_t, p.isi), dtype=torch.complex128, device=self.device)
            check(self.lib.esn_gen_taps(self.CHANNEL_KIND[p.channel], n_blocks, p.n_r, p.n_t, p.isi, p.fs, p.ds_ns,
                                        ptr(gains), self._key(snr_idx, 1), int(first_block) * p.n_r * p.n_t,
                                        ptr(out), _lib.stream_handle()), "esn_gen_taps")
        return out

    def frames(self, taps, frames_per_block, ebno_db, snr_idx, first_frame, stream_id, want_x=False,
               bits_in=None, noise_in=None, ls_pattern=False):
        """frames_per_block frames per block of `taps` -> (bits uint8 [B,N*m,n_t], x_cp or None, y_cp).
        stream_id separates pilots (0) from data (1); first_frame is the global frame counter."""
        torch, p = self.torch, self.p
        g = taps.shape[0]
        b = g * frames_per_block
        with torch.cuda.device(self.device):
            p_i = torch.full((g,), p.p_i(ebno_db), dtype=torch.float64, device=self.device)
            a_clip = torch.full((g,), p.a_clip(ebno_db), dtype=torch.float64, device=self.device)
            bits = torch.empty((b, p.n_sub * p.m, p.n_t), dtype=torch.uint8, device=self.device)
            x_cp = torch.empty((b, p.t_frame, p.n_t), dtype=torch.complex128, device=self.device) if want_x else None
            y_cp = torch.empty((b, p.t_frame, p.n_r), dtype=torch.complex128, device=self.device)
            check(self.lib.esn_gen_frames(b, frames_per_block, p.n_sub, p.cp, p.n_t, p.n_r, p.isi, p.m,
                                          1 if ls_pattern else 0, ptr(p_i),
                                          ptr(a_clip), p.no, ptr(taps), ptr(bits_in), ptr(noise_in),
                                          self._key(snr_idx, 2 + stream_id), int(first_frame), ptr(bits), ptr(x_cp),
                                          ptr(y_cp), _lib.stream_handle()), "esn_gen_frames")
        return bits, x_cp, y_cp

    def blocks(self, ebno_db, snr_idx, block_ids, frames_per_block, with_ls_pilot=False):
        """Pilot + data frames of the given coherence blocks (any subset, any order: every block is
        generated from its own global index, so the result does not depend on the rank that asks).
        Returns pilot_y [G,T,n_r], pilot_x [G,T,n_t] (pre-PA teacher), data_y [G*F,T,n_r], data_bits."""
        torch = self.torch
        ids = list(block_ids)
        runs, start = [], 0                          # contiguous runs of block ids -> one launch each
        for i in range(1, len(ids) + 1):
            if i == len(ids) or ids[i] != ids[i - 1] + 1:
                runs.append((ids[start], i - start)); start = i
        outs = [self.blocks_fast(ebno_db, snr_idx, b0, n, frames_per_block, with_ls_pilot) for b0, n in runs]
        return {k: torch.cat([o[k] for o in outs]) for k in outs[0]}

    def blocks_fast(self, ebno_db, snr_idx, first_block, n_blocks, frames_per_block, with_ls_pilot=False):
        """Blocks first_block .. first_block + n_blocks - 1 in three launches (taps, pilots, data)."""
        taps = self.taps(n_blocks, snr_idx, first_block)
        pbits, px, py = self.frames(taps, 1, ebno_db, snr_idx, first_block, 0, want_x=True)
        bits, _, dy = self.frames(taps, frames_per_block, ebno_db, snr_idx, first_block * frames_per_block, 1)
        out = dict(pilot_y=py, pilot_x=px, pilot_bits=pbits, data_y=dy, data_bits=bits, taps=taps)
        if with_ls_pilot:     # same bits, same noise, sparse pattern (driver:330-356)
            _, _, out["pilot_y_ls"] = self.frames(taps, 1, ebno_db, snr_idx, first_block, 0, ls_pattern=True)
        return out

    # ---- baseline equaliser ------------------------------------------------------
    def estimate_channel(self, pilot_bits, pilot_y_ls, ebno_db, ls_only=False):
        """LS + time-domain MMSE channel estimate H [G, N, n_r, n_t] (driver:358-382); ls_only: the interpolated
        LS estimate H_LS of the block-fading drivers' LS-ZF detector (OFDM_MIMO_2-2_NBF_LDPC.py:321-333)."""
        torch, p = self.torch, self.p
        g = pilot_bits.shape[0]
        with torch.cuda.device(self.device):
            p_i = torch.full((g,), p.p_i(ebno_db), dtype=torch.float64, device=self.device)
            H = torch.empty((g, p.n_sub, p.n_r, p.n_t), dtype=torch.complex128, device=self.device)
            check(self.lib.esn_channel_estimate(g, p.n_sub, p.cp, p.n_t, p.n_r, p.isi, p.m, ptr(p_i), p.no,
                                                ptr(pilot_bits.contiguous()), ptr(pilot_y_ls.contiguous()),
                                                1 if ls_only else 0, ptr(H), _lib.stream_handle()),
                  "esn_channel_estimate")
        return H

    def true_channel(self, taps):
        """H_true [G, N, n_r, n_t] = FFT_N of the zero-padded taps (OFDM_MIMO_2-2_NBF_LDPC.py:273-279)."""
        torch, p = self.torch, self.p
        g = taps.shape[0]
        with torch.cuda.device(self.device):
            H = torch.empty((g, p.n_sub, p.n_r, p.n_t), dtype=torch.complex128, device=self.device)
            check(self.lib.esn_taps_to_freq(g, p.n_sub, p.n_t, p.n_r, p.isi, ptr(taps.contiguous()), ptr(H),
                                            _lib.stream_handle()), "esn_taps_to_freq")
        return H

    def mmse_detect_count(self, H, data_y, data_bits, frames_per_block, ebno_db, err=None, bits=None,
                          want_xhat=False, zf=False):
        """Per-subcarrier MMSE detector + error counters (driver:444-456); zf=True: equalize_zf (driver:34-39),
        LS-ZF with an estimated H, Perfect-ZF with `true_channel` (OFDM_MIMO_2-2_NBF_LDPC.py:450-460)."""
        torch, p = self.torch, self.p
        g, b = H.shape[0], data_y.shape[0]
        with torch.cuda.device(self.device):
            p_i = torch.full((g,), p.p_i(ebno_db), dtype=torch.float64, device=self.device)
            if err is None:
                err = torch.zeros(g, dtype=torch.int64, device=self.device)
            if bits is None:
                bits = torch.zeros(g, dtype=torch.int64, device=self.device)
            xh = torch.empty((b, p.n_sub, p.n_t), dtype=torch.complex128, device=self.device) if want_xhat else None
            if zf:
                check(self.lib.esn_zf_detect_count(b, int(frames_per_block), p.n_sub, p.cp, p.n_t, p.n_r, p.m,
                                                   ptr(p_i), ptr(H), ptr(data_y.contiguous()),
                                                   ptr(data_bits.contiguous()), ptr(err), ptr(bits), ptr(xh),
                                                   _lib.stream_handle()), "esn_zf_detect_count")
            else:
                check(self.lib.esn_mmse_detect_count(b, int(frames_per_block), p.n_sub, p.cp, p.n_t, p.n_r, p.m,
                                                     ptr(p_i), p.no, ptr(H), ptr(data_y.contiguous()),
                                                     ptr(data_bits.contiguous()), ptr(err), ptr(bits), ptr(xh),
                                                     _lib.stream_handle()), "esn_mmse_detect_count")
        return (err, bits, xh) if want_xhat else (err, bits)


def _view_real(z):
    """complex128 [..., T, n] -> float64 view [..., T, 2n] (Re/Im interleaved; driver:433-436)."""
    import torch
    z = z.contiguous()
    return torch.view_as_real(z).reshape(*z.shape[:-1], 2 * z.shape[-1])


complex_as_io = _view_real


def blocks_for_rank(rank, world_size, n_blocks):
    """Contiguous deal of coherence blocks to ranks: rank r owns blocks
    [r n / W, (r+1) n / W); the union over ranks is range(n_blocks).  A block's random streams, state
    noise and (per-block reservoirs) weight set depend on its GLOBAL index only -- never on the rank,
    the chunk or the launch it lands in -- so the summed counters are identical for any world size.
    (Contiguous ranges, so that a chunk of a rank's blocks is one run of global indices and one
    `group_offset` describes it to the kernels.)"""
    lo = (rank * n_blocks) // world_size
    hi = ((rank + 1) * n_blocks) // world_size
    return list(range(lo, hi))


def reduce_counters(counters, dist=None, world_size=1):
    """The path's only collective: one all_reduce(SUM) of the int64 [n_snr, 2] (errors, bits)
    tensor (RCCL over xGMI on GPUs, gloo on CPU).  Integer sums are order independent, so the
    result is bit-identical for any world size.  Runs whenever a process group is handed in (a
    one-rank group included); `dist=None` is the single-process path."""
    if dist is not None and (world_size > 1 or dist.is_initialized()):
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
                 rank=0, world_size=1, solve_method="auto", train_ebno=None):
        torch = _lib.require_gpu()
        self.torch, self.p = torch, params
        self.rank, self.world = rank, world_size
        self.precision, self.fit_precision = precision, fit_precision
        self.solve_method = solve_method
        self.train_ebno = train_ebno      # not None: every ESN is trained at this fixed Eb/No
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

    def stream_seed(self, snr_idx, leg):
        """64-bit seed of the state-noise stream of one Eb/No point; leg 0 = training (harvest), 1 = detection.
        With the kernels' global frame index this makes the noise a function of (seed, snr, leg, global frame,
        step, row) -- independent of chunking and world size."""
        h = (int(self.seed) * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) % (2 ** 64)
        for k in (snr_idx, leg):
            h = ((h ^ (h >> 31)) * 0xBF58476D1CE4E5B9 + int(k) + 1) % (2 ** 64)
        return h

    def train(self, pilot_y, pilot_x, seed=0, group_offset=0):
        """helper_mimo_esn_generic.py:58-86 for G blocks: delay d, nForget = d + CP, one harvest + solve.
        group_offset = global index of the first block (noise key and weight set follow the global block)."""
        torch, p = self.torch, self.p
        d = p.delay
        g, t = pilot_y.shape[0], pilot_y.shape[1]
        # zero-padded pilot buffers are kept between calls of the same shape (the padding rows are never written)
        io = getattr(self, "_train_bufs", None)
        if io is None or io[0].shape != (g, t + d, self.n_in):
            io = self._train_bufs = (torch.zeros((g, t + d, self.n_in), dtype=torch.float64, device=self.device),
                                     torch.zeros((g, t + d, self.n_out), dtype=torch.float64, device=self.device))
        U, D = io
        U[:, :t] = _view_real(pilot_y)
        D[:, d:d + t] = _view_real(pilot_x)
        self._fit_io = (U, D, p.forget)
        # float32 extended states on the all-GPU fast path (fp16/bf16 harvest + Cholesky): the state
        # columns are exactly representable, the fit is unchanged to ~1e-7
        rows, cols = t + d - p.forget, self.bank.n_reservoir + self.n_in
        chol = self.solve_method == "chol" or (self.solve_method == "auto" and self.bank.chol_fits(rows, cols))
        e_dtype = "f32" if (chol and self.fit_precision in ("f16", "bf16")) else "f64"
        E = self.bank.fit(U, D, transient=p.forget, precision=self.fit_precision, noise_mode="counter",
                          seed=seed, method=self.solve_method, e_dtype=e_dtype, group_offset=group_offset)
        self._cont = None
        if p.continuation:      # laststate / lastoutput of pyESN.py:195-197: training-final state, scaled teacher
            y_last = D[:, -1, :]
            if self.bank.t_scale is not None:
                y_last = y_last * self.bank.t_scale[:g]
            if self.bank.t_shift is not None:
                y_last = y_last + self.bank.t_shift[:g]
            self._cont = (E[:, -1, :self.bank.n_reservoir].double().contiguous(), y_last.contiguous())
        return E

    def repair_fit(self, E):
        """Host-synchronising check of the last fit: groups the Cholesky path flagged are re-solved
        with the QR kernel (GPU).  Returns how many were."""
        U, D, tr = self._fit_io
        n = self.bank.resolve_failed(E, D, tr, self.bank.W_out, self.bank.fit_status)
        if n:
            self.bank.set_readout(self.bank.W_out)
        return n

    def detect(self, data_y, data_bits, frames_per_block, err, bits, seed=0, out=None, group_offset=0):
        """driver:433-456 for all data frames of G blocks: predict (d trailing zero rows synthesised
        in-kernel) -> fused FFT/slicer/count."""
        p = self.p
        U = _view_real(data_y)
        x0, y0 = self._cont if (p.continuation and getattr(self, "_cont", None)) else (None, None)
        y = self.bank.predict(U, frames_per_block, T=p.t_frame + p.delay, transient=p.forget, x0=x0, y0=y0,
                              precision=self.precision, noise_mode="counter", seed=seed, out=out,
                              group_offset=group_offset)
        self.bank.detect_count(y, data_bits, self.p_i, frames_per_block, p.n_sub, p.n_t, p.m, err=err, bits=bits)
        return y

    def default_chunk_blocks(self, frames_per_block):
        """Blocks per launch that fill the chip with whole rounds of workgroup tiles: about five tiles per CU
        (the benchmark's choice), i.e. 5 * CUs * tile_frames slots at ceil16(F) slots per block."""
        info = _lib.device_info()
        tile = self.bank.tile_frames(self.precision)
        fpad = ((frames_per_block + 15) // 16) * 16
        return max(1, (5 * info["cu_count"] * tile) // fpad)

    def _chunk(self, ebno, si, ids, F, repair):
        """One launch group: generate, train, detect the contiguous global blocks `ids`; returns the device
        tensor [errors, bits, flagged fits] (int64) without synchronising the host unless `repair`."""
        torch = self.torch
        g = len(ids)
        data = self.src.blocks_fast(ebno, si, ids[0], g, F)
        self.set_snr(ebno, g)
        if self.train_ebno is not None:
            # the "train@fixed Eb/No" ESN of the block-fading drivers (OFDM_MIMO_2-2_NBF_LDPC.py:181-183,347-367):
            # pilot generated at the training Eb/No over the SAME taps, input scaling of that Eb/No at train AND
            # detect time, evaluated on the data frames of the actual Eb/No (:440-448)
            _, px, py = self.src.frames(data["taps"], 1, self.train_ebno, si, ids[0], 0, want_x=True)
            data["pilot_y"], data["pilot_x"] = py, px
            ones_in = torch.ones((g, self.n_in), dtype=torch.float64, device=self.device)
            self.bank.in_scale = ones_in * self.p.input_scaling(self.train_ebno)
        E = self.train(data["pilot_y"], data["pilot_x"], seed=self.stream_seed(si, 0), group_offset=ids[0])
        if repair:
            self.repair_fit(E)
        err = torch.zeros(g, dtype=torch.int64, device=self.device)
        nb = torch.zeros(g, dtype=torch.int64, device=self.device)
        self.detect(data["data_y"], data["data_bits"], F, err, nb, seed=self.stream_seed(si, 1), group_offset=ids[0])
        flagged = self.bank.fit_status.ne(0).sum().to(torch.int64) if not repair else torch.zeros(
            (), dtype=torch.int64, device=self.device)
        # (a timed-out harvest cluster counts as a flagged fit: the chunk is then redone with `repair`, whose host
        #  read raises)
        ht = getattr(self.bank, "harvest_timeout", None)
        if ht is not None:
            if repair:
                self.bank.raise_if_harvest_timed_out()
            else:
                flagged = flagged + ht.ne(0).sum().to(torch.int64)
        return torch.stack([err.sum(), nb.sum(), flagged])

    def run(self, ebno_list, blocks_per_snr, frames_per_block=None, chunk_blocks=None, dist=None):
        """Returns (BER[n_snr], counters [n_snr, 2]) -- identical on every rank and for every world size and
        chunking (contiguous block ranges per rank; every stream keyed by global indices).  No host
        synchronisation inside an Eb/No point: the Cholesky status flags are summed on the device and read once
        per point; a chunk with a flagged fit (none on any run so far) is redone with the QR repair."""
        torch = self.torch
        F = frames_per_block or self.p.coherence_symbols
        chunk = int(chunk_blocks or self.default_chunk_blocks(F))
        n_snr = len(ebno_list)
        counters = torch.zeros((n_snr, 2), dtype=torch.int64, device=self.device)
        mine = blocks_for_rank(self.rank, self.world, blocks_per_snr)
        self.fits_repaired = 0
        for si, ebno in enumerate(ebno_list):
            chunks = [mine[c0:c0 + chunk] for c0 in range(0, len(mine), chunk)]
            if not chunks:
                continue
            res = torch.stack([self._chunk(ebno, si, ids, F, repair=False) for ids in chunks])   # [n_chunks, 3]
            if int(res[:, 2].sum().item()):                     # the point's one host read
                for ci in torch.nonzero(res[:, 2]).flatten().tolist():
                    self.fits_repaired += int(res[ci, 2].item())
                    res[ci] = self._chunk(ebno, si, chunks[ci], F, repair=True)
            counters[si] += res[:, :2].sum(dim=0)
        reduce_counters(counters, dist, self.world)
        c = counters.cpu().numpy()
        return c[:, 0] / np.maximum(c[:, 1], 1), c


def coded_ber_point(sweep, code, ebno_db, snr_idx, n_blocks, frames_per_block=None, cal_frac=0.3, seed=0):
    """One Eb/No point of the reference's coded + uncoded comparison, batched on the device
    (Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:283-530): LDPC-coded payloads on every data symbol, ESN and
    LS/MMSE detection, max-log LLRs, logistic LLR calibration fitted on the first `cal_frac` of the
    blocks (the reference: the first 30 % of the symbols, :266,:476-482,:513-523) and sum-product
    decoding of the rest.  Returns dict(ESN_uncoded, MMSE_uncoded, ESN_coded, MMSE_coded, a_esn, ...)."""
    torch, p, src = sweep.torch, sweep.p, sweep.src
    F = frames_per_block or p.coherence_symbols
    G = n_blocks
    gen = torch.Generator(device=sweep.device)
    gen.manual_seed((seed * 1000003 + snr_idx * 7919 + 12345) % (2 ** 63 - 1))
    u = torch.randint(0, 2, (G * F, p.n_t, code.k), generator=gen, device=sweep.device, dtype=torch.uint8)
    tx_bits = code.encode(u, p.n_t)                                          # [B, N*m, n_t]
    taps = src.taps(G, snr_idx, 0)
    pbits, px, py = src.frames(taps, 1, ebno_db, snr_idx, 0, 0, want_x=True)
    _, _, py_ls = src.frames(taps, 1, ebno_db, snr_idx, 0, 0, ls_pattern=True)
    _, _, dy = src.frames(taps, F, ebno_db, snr_idx, 0, 1, bits_in=tx_bits)
    # ESN
    sweep.set_snr(ebno_db, G)
    E = sweep.train(py, px, seed=sweep.stream_seed(snr_idx, 0) + seed)
    sweep.repair_fit(E)
    U = _view_real(dy)
    y = sweep.bank.predict(U, F, T=p.t_frame + p.delay, transient=p.forget, precision=sweep.precision,
                           noise_mode="counter", seed=sweep.stream_seed(snr_idx, 1) + seed)
    e_esn, n_esn, xh = sweep.bank.detect_count(y, tx_bits, sweep.p_i, F, p.n_sub, p.n_t, p.m, want_xhat=True)
    x_esn = torch.view_as_complex(xh.view(G * F, p.n_sub, p.n_t, 2).contiguous())
    # LS/MMSE baseline
    H = src.estimate_channel(pbits, py_ls, ebno_db)
    e_mm, n_mm, x_mm = src.mmse_detect_count(H, dy, tx_bits, F, ebno_db, want_xhat=True)
    out = dict(ESN_uncoded=float(e_esn.sum()) / float(n_esn.sum()), MMSE_uncoded=float(e_mm.sum()) / float(n_mm.sum()))
    n_cal = max(1, int(round(cal_frac * G))) * F                             # frames used for calibration
    for name, xhat in (("ESN", x_esn), ("MMSE", x_mm)):
        llr, _ = code.llrs(xhat, p.m)
        a, b = code.fit_calibration(llr[:n_cal], tx_bits[:n_cal], p.m)
        err, nb = code.decode_count(llr[n_cal:].contiguous(), a, b, u[n_cal:], p.n_t * F, p.m)
        out[name + "_coded"] = float(err.sum()) / max(float(nb.sum()), 1.0)
        out["a_" + name.lower()] = a.cpu().numpy()
        out["b_" + name.lower()] = b.cpu().numpy()
    return out


def block_fading_point(sweep, code, ebno_db, snr_idx, n_blocks, fixed_sweep=None, decode_every=4, llr_scale=1.5,
                       seed=0):
    """One Eb/No point of the block-fading drivers' comparison (OFDM_{SISO,SIMO_1-2,MIMO_2-2}_NBF_LDPC.py /
    Demo_MIMO_4x8_ChannelRank_..._fast.py :266-521), batched on the device: per coherence block one pilot and
    L - 1 LDPC-coded data symbols (the pilot symbol carries no data here, :387); detectors ESN (SNR-matched),
    ESN trained at a fixed Eb/No (`fixed_sweep`: a DetectorSweep built with train_ebno=12), LS-ZF,
    MMSE and Perfect-ZF (:450-460); uncoded BER over every data symbol, coded BER on every `decode_every`-th
    symbol of the run (kk % 4 == 1, :202,389) with the drivers' uncalibrated LLRs: per-stream decision-directed
    sigma^2, x LLR_SCALE 1.5, clip +-20 (:478-485).  Returns the reference's holder names (BER_* / BERC_*)."""
    torch, p, src = sweep.torch, sweep.p, sweep.src
    L = p.coherence_symbols
    F, G = L - 1, n_blocks
    gen = torch.Generator(device=sweep.device)
    gen.manual_seed((seed * 1000003 + snr_idx * 7919 + 4242) % (2 ** 63 - 1))
    u = torch.randint(0, 2, (G * F, p.n_t, code.k), generator=gen, device=sweep.device, dtype=torch.uint8)
    tx_bits = code.encode(u, p.n_t)
    taps = src.taps(G, snr_idx, 0)
    pbits, px, py = src.frames(taps, 1, ebno_db, snr_idx, 0, 0, want_x=True)
    _, _, py_ls = src.frames(taps, 1, ebno_db, snr_idx, 0, 0, ls_pattern=True)
    _, _, dy = src.frames(taps, F, ebno_db, snr_idx, 0, 1, bits_in=tx_bits)
    xhat = {}
    err = {}

    def esn_leg(sw, name, pilot_y, pilot_x, scale_ebno):
        sw.set_snr(ebno_db, G)
        if scale_ebno != ebno_db:
            ones_in = torch.ones((G, sw.n_in), dtype=torch.float64, device=sw.device)
            sw.bank.in_scale = ones_in * p.input_scaling(scale_ebno)
        E = sw.train(pilot_y, pilot_x, seed=sw.stream_seed(snr_idx, 0) + seed)
        sw.repair_fit(E)
        y = sw.bank.predict(_view_real(dy), F, T=p.t_frame + p.delay, transient=p.forget, precision=sw.precision,
                            noise_mode="counter", seed=sw.stream_seed(snr_idx, 1) + seed)
        e, nb, xh = sw.bank.detect_count(y, tx_bits, sw.p_i, F, p.n_sub, p.n_t, p.m, want_xhat=True)
        xhat[name] = torch.view_as_complex(xh.view(G * F, p.n_sub, p.n_t, 2).contiguous())
        err[name] = (e, nb)

    esn_leg(sweep, "ESN_matched", py, px, ebno_db)
    if fixed_sweep is not None:
        t_eb = fixed_sweep.train_ebno
        _, px_f, py_f = src.frames(taps, 1, t_eb, snr_idx, 0, 0, want_x=True)     # same pilot bits at the fixed power
        esn_leg(fixed_sweep, "ESN_trainFixed", py_f, px_f, t_eb)
    H_ls = src.estimate_channel(pbits, py_ls, ebno_db, ls_only=True)
    H_mmse = src.estimate_channel(pbits, py_ls, ebno_db)
    H_true = src.true_channel(taps)
    for name, H, zf in (("LS_ZF", H_ls, True), ("MMSE", H_mmse, False), ("PerfectZF", H_true, True)):
        e, nb, xh = src.mmse_detect_count(H, dy, tx_bits, F, ebno_db, want_xhat=True, zf=zf)
        xhat[name], err[name] = xh, (e, nb)
    out = {"BER_" + k: float(v[0].sum()) / float(v[1].sum()) for k, v in err.items()}
    # coded leg: symbol kk (1-based over the run; block b holds kk = L b + 1 (pilot) .. L b + L) decodes iff kk % every == 1
    kk = (torch.arange(G, device=sweep.device)[:, None] * L + 2 + torch.arange(F, device=sweep.device)[None, :]).reshape(-1)
    sel = torch.nonzero((kk % decode_every) == 1).flatten()
    a = torch.full((p.m,), -float(llr_scale), dtype=torch.float64, device=sweep.device)   # -(a llr + b) = scale * llr
    b = torch.zeros(p.m, dtype=torch.float64, device=sweep.device)
    for name, xh in xhat.items():
        xs = xh[sel]                                                           # [S, N, n_t]
        per_stream = xs.permute(0, 2, 1).reshape(-1, p.n_sub, 1).contiguous()  # sigma^2 per (frame, tx) column (:479)
        llr, _ = code.llrs(per_stream, p.m)                                    # [S n_t, 1, N m]
        e, nb = code.decode_count(llr.view(xs.shape[0], p.n_t, -1), a, b, u[sel], max(1, xs.shape[0] * p.n_t), p.m)
        out["BERC_" + name] = float(e.sum()) / max(float(nb.sum()), 1.0)
    out["decoded_symbols"] = int(sel.numel())
    return out
