"""Batched, device-resident form of the ESN hot path.

``ReservoirBank`` holds G echo-state networks on one GPU -- either one shared
reservoir (W, W_in, W_feedb) with G readouts ("shared-reservoir" mode) or one
reservoir per group ("reference-faithful" mode, the reference draws a fresh
reservoir per coherence block) -- and runs the reference's three
operations for all of them at once through the C ABI of ``include/esn_hip.h``:

    harvest  -> state-collection loop of ESN.fit      (libs/pyESN.py:176-189)
    solve    -> pinv readout solve of ESN.fit         (libs/pyESN.py:191-192)
    predict  -> ESN.predict                           (libs/pyESN.py:218-255)
    detect   -> reconstruct/FFT/slicer/error count    (Demo_MIMO_4x8_..._v2.py:47-58,439-456)

torch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import PRECISIONS, Shape, check, ptr


def _as_dev(x, torch, device, dtype=None):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        t = x.to(device=device, dtype=dtype or torch.float64)
    else:
        a = np.ascontiguousarray(x)
        if not a.flags.writeable:                 # (arrays out of an .npz are read-only; torch wants ownership)
            a = a.copy()
        t = torch.as_tensor(a, device=device).to(dtype or torch.float64)
    return t.contiguous()


class ReservoirBank:
    def __init__(self, n_inputs, n_outputs, n_reservoir, W, W_in, W_feedb,
                 teacher_forcing=True, noise=0.001, device=None, leak_rate=1.0):
        torch = _lib.require_gpu()
        self.torch = torch
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.n_inputs, self.n_outputs, self.n_reservoir = int(n_inputs), int(n_outputs), int(n_reservoir)
        self.noise = float(noise)
        W = np.asarray(W, dtype=np.float64)
        W_in = np.asarray(W_in, dtype=np.float64)
        W_feedb = np.asarray(W_feedb, dtype=np.float64)
        if W.ndim == 2:
            W, W_in, W_feedb = W[None], W_in[None], W_feedb[None]
        self.n_wsets = W.shape[0]
        if W.shape[1:] != (n_reservoir, n_reservoir) or W_in.shape != (self.n_wsets, n_reservoir, n_inputs) \
                or W_feedb.shape != (self.n_wsets, n_reservoir, n_outputs):
            raise ValueError("weight shapes do not match (n_wsets, n_reservoir, ...)")
        if not 0.0 < float(leak_rate) <= 1.0:
            raise ValueError("leak_rate must be in (0, 1]")
        self.leak_rate = float(leak_rate)       # extension (the reference has none): float64 kernels only when != 1
        self.shape = Shape(n_reservoir, n_inputs, n_outputs, 1 if teacher_forcing else 0, self.n_wsets, self.leak_rate)
        with torch.cuda.device(self.device):
            self._W = _as_dev(W, torch, self.device)
            self._W_in = _as_dev(W_in, torch, self.device)
            self._W_fb = _as_dev(W_feedb, torch, self.device)
        self._packed = {}
        self._packed_wout = {}
        self.in_scale = self.in_shift = self.t_scale = self.t_shift = None
        self.W_out = None

    # ------------------------------------------------------------------ packing
    def tile_frames(self, precision):
        rc = self.lib.esn_tile_frames(PRECISIONS[precision], C.byref(self.shape))
        if rc <= 0:
            check(rc if rc else -2, "esn_tile_frames")
        return rc

    def packed_weights(self, precision):
        if precision not in self._packed:
            torch, p = self.torch, PRECISIONS[precision]
            nbytes = self.lib.esn_packed_weights_bytes(p, C.byref(self.shape))
            if nbytes == 0:
                raise _lib.EsnHipError(f"precision {precision} does not support n_reservoir={self.n_reservoir}")
            with torch.cuda.device(self.device):
                buf = torch.empty(nbytes * self.n_wsets, dtype=torch.uint8, device=self.device)
                check(self.lib.esn_pack_weights(p, C.byref(self.shape), ptr(self._W), ptr(self._W_in),
                                                ptr(self._W_fb), ptr(buf), _lib.stream_handle()),
                      "esn_pack_weights")
            self._packed[precision] = buf
        return self._packed[precision]

    def set_scaling(self, in_scale=None, in_shift=None, t_scale=None, t_shift=None):
        """Per-group scalings, each [G, n] (or None = identity): pyESN.py:127-152."""
        torch = self.torch
        self.in_scale = _as_dev(in_scale, torch, self.device)
        self.in_shift = _as_dev(in_shift, torch, self.device)
        self.t_scale = _as_dev(t_scale, torch, self.device)
        self.t_shift = _as_dev(t_shift, torch, self.device)

    def set_readout(self, W_out):
        """W_out [G, n_out, n_res + n_in] (float64)."""
        self.W_out = _as_dev(W_out, self.torch, self.device)
        if self.W_out.ndim == 2:
            self.W_out = self.W_out[None].contiguous()
        self._packed_wout = {}

    def packed_readout(self, precision):
        if self.W_out is None:
            raise AttributeError("W_out: fit (or set_readout) before predict")
        if precision not in self._packed_wout:
            torch, p = self.torch, PRECISIONS[precision]
            g = self.W_out.shape[0]
            nbytes = self.lib.esn_packed_readout_bytes(p, C.byref(self.shape))
            with torch.cuda.device(self.device):
                buf = torch.empty(nbytes * g, dtype=torch.uint8, device=self.device)
                check(self.lib.esn_pack_readout(p, C.byref(self.shape), g, ptr(self.W_out), ptr(buf),
                                                _lib.stream_handle()), "esn_pack_readout")
            self._packed_wout[precision] = buf
        return self._packed_wout[precision]

    # ------------------------------------------------------------------ fit
    def harvest(self, U, D, precision="f64", noise_mode="counter", noise_u=None, seed=0, e_dtype="f64",
                group_offset=0):
        """U [G,T,n_in], D [G,T,n_out] -> extended states E [G,T,n_res+n_in] (device).

        group_offset: global index of group 0 in the caller's sweep -- the counter noise and (with several
        weight sets) the weight set follow the GLOBAL group, so a sweep cut into chunks or ranks gives the
        same states (include/esn_hip.h).

        e_dtype "f32" (MFMA precisions only) stores E as float32 -- exact for the state columns, 6e-8
        relative on the scaled inputs -- which halves the store tail here and the reads of `solve`."""
        torch = self.torch
        U = _as_dev(U, torch, self.device)
        D = _as_dev(D, torch, self.device)
        g, t = U.shape[0], U.shape[1]
        if D.shape[0] != g or D.shape[1] != t:
            raise ValueError("inputs and teacher disagree on [G, T]")
        self._check_groups(g)
        nm, nz = self._noise_args(noise_mode, noise_u, (g, t - 1, self.n_reservoir))
        with torch.cuda.device(self.device):
            if e_dtype not in ("f64", "f32"):
                raise ValueError("e_dtype must be 'f64' or 'f32'")
            f32 = e_dtype == "f32"
            fn = self.lib.esn_harvest_batch_f32 if f32 else self.lib.esn_harvest_batch
            E = torch.empty((g, t, self.n_reservoir + self.n_inputs),
                            dtype=torch.float32 if f32 else torch.float64, device=self.device)
            # reservoirs beyond 1024 units harvest as one GEMM launch per step out of a caller-owned workspace
            wbytes = self.lib.esn_harvest_workspace_bytes(PRECISIONS[precision], C.byref(self.shape), g)
            ws = None
            if wbytes:
                ws = getattr(self, "_harvest_ws", None)
                if ws is None or ws.numel() < wbytes:
                    ws = self._harvest_ws = torch.empty(wbytes, dtype=torch.uint8, device=self.device)
            check(fn(PRECISIONS[precision], C.byref(self.shape), ptr(self.packed_weights(precision)),
                     ptr(self.in_scale), ptr(self.in_shift), ptr(self.t_scale), ptr(self.t_shift),
                     ptr(U), ptr(D), g, t, self.noise, nm, ptr(nz), int(seed) & (2**64 - 1), int(group_offset),
                     ptr(E), ptr(ws), wbytes, _lib.stream_handle()), "esn_harvest_batch")
            if wbytes and precision == "f64" and g == 1:
                self._cluster_err = ws[wbytes - 64:wbytes - 60]      # error word of the single-sequence cluster kernel
            # fp16/bf16 at 257..512 units: clusters of eight workgroups with the matrix resident in LDS
            # (csrc/esn_harvest_cluster.hip); their bounded waits raise the same kind of error word
            self.harvest_timeout = (ws[wbytes - 64:wbytes - 60].view(torch.int32)
                                    if wbytes and precision in ("f16", "bf16") and self.n_reservoir <= 512 else None)
        return E

    def raise_if_harvest_timed_out(self):
        """Host-synchronising check of the last fp16/bf16 harvest on the cluster kernel (see harvest)."""
        w = getattr(self, "harvest_timeout", None)
        if w is not None and int(w.item()) != 0:
            raise _lib.EsnHipError("harvest cluster kernel: a workgroup timed out waiting for the others "
                                   "(the device is oversubscribed); the extended states are invalid")

    def chol_fits(self, rows, cols):
        """Shapes the Cholesky solve covers (esn_readout_solve_chol_batch): Gram dimension up to 128 in LDS,
        up to 512 out of a workspace."""
        return min(rows, cols) <= 512 and self.n_outputs <= 8

    def solve(self, E, D, transient, method="qr"):
        """W_out[g] = (pinv(E[g][transient:]) @ scale(D[g][transient:])).T ; returns (W_out, status).

        method "qr": float64 Householder QR (accurate to cond(E) eps; the drop-in's choice).
        method "chol": float64 normal equations on the float64 matrix pipe (min(rows, cols) <= 512, n_out <= 8;
        Gram + factor in LDS up to 128, in a workspace beyond), an order of magnitude faster; groups whose pivot test fails are re-solved
        with QR on the GPU.  "auto" = "chol" when the shape fits."""
        torch = self.torch
        e32 = isinstance(E, torch.Tensor) and E.dtype == torch.float32       # as written by harvest(e_dtype="f32")
        E = _as_dev(E, torch, self.device, torch.float32 if e32 else None)
        D = _as_dev(D, torch, self.device)
        g, t, cols = E.shape
        rows = t - transient
        fits = self.chol_fits(rows, cols)
        if method == "auto":
            method = "chol" if fits else "qr"
        if e32 and method != "chol":
            E = E.double()                                                  # the QR kernel works in place on float64
        if method == "chol":
            if not fits:
                raise ValueError("method='chol' needs min(rows, cols) <= 512 and n_outputs <= 8")
            with torch.cuda.device(self.device):
                W_out = torch.empty((g, self.n_outputs, cols), dtype=torch.float64, device=self.device)
                status = torch.empty(g, dtype=torch.int32, device=self.device)
                fn = self.lib.esn_readout_solve_chol_batch_f32 if e32 else self.lib.esn_readout_solve_chol_batch
                wbytes = self.lib.esn_readout_chol_workspace_bytes(g, rows, cols)
                ws = None
                if wbytes:
                    ws = getattr(self, "_chol_ws", None)
                    if ws is None or ws.numel() < wbytes:
                        ws = self._chol_ws = torch.empty(wbytes, dtype=torch.uint8, device=self.device)
                check(fn(ptr(E), ptr(D), g, t, int(transient), cols, self.n_outputs, ptr(self.t_scale),
                         ptr(self.t_shift), ptr(W_out), ptr(status), ptr(ws), wbytes, _lib.stream_handle()),
                      "esn_readout_solve_chol_batch")
            self.last_solve_status = status        # checked lazily: no host sync on the fast path
            return W_out, status
        with torch.cuda.device(self.device):
            wbytes = self.lib.esn_readout_solve_workspace_bytes(g, rows, cols, self.n_outputs)
            work = torch.empty(wbytes, dtype=torch.uint8, device=self.device)
            W_out = torch.empty((g, self.n_outputs, cols), dtype=torch.float64, device=self.device)
            status = torch.empty(g, dtype=torch.int32, device=self.device)
            check(self.lib.esn_readout_solve_batch(ptr(E), ptr(D), g, t, int(transient), cols, self.n_outputs,
                                                   ptr(self.t_scale), ptr(self.t_shift), ptr(W_out),
                                                   ptr(status), ptr(work), _lib.stream_handle()),
                  "esn_readout_solve_batch")
        return W_out, status

    def resolve_failed(self, E, D, transient, W_out, status):
        """Re-solve with QR (on the GPU) the groups a "chol" solve flagged; returns their count."""
        torch = self.torch
        bad = torch.nonzero(status).flatten()
        nbad = int(bad.numel())
        if nbad:
            keep = (self.t_scale, self.t_shift)
            self.t_scale = None if keep[0] is None else keep[0][bad].contiguous()
            self.t_shift = None if keep[1] is None else keep[1][bad].contiguous()
            try:
                w2, st2 = self.solve(E[bad].contiguous(), _as_dev(D, torch, self.device)[bad].contiguous(),
                                     transient, method="qr")
            finally:
                self.t_scale, self.t_shift = keep
            W_out[bad] = w2
            status[bad] = st2
        return nbad

    def fit(self, U, D, transient=0, precision="f64", noise_mode="counter", noise_u=None, seed=0,
            method="qr", e_dtype="f64", group_offset=0):
        E = self.harvest(U, D, precision, noise_mode, noise_u, seed, e_dtype=e_dtype, group_offset=group_offset)
        W_out, status = self.solve(E, D, transient, method=method)
        self.set_readout(W_out)
        ht = getattr(self, "harvest_timeout", None)
        if ht is not None:
            # a harvest cluster that timed out waiting for its peer left invalid states: every group's status says so
            # (-9; device-side, no host sync here -- whoever reads fit_status sees it)
            status = self.torch.where(ht.ne(0).expand_as(status), self.torch.full_like(status, -9), status)
        self.fit_status = status
        return E

    # ------------------------------------------------------------------ predict
    def predict(self, U, frames_per_group, T=None, transient=0, precision="f32", x0=None, y0=None,
                noise_mode="counter", noise_u=None, seed=0, out=None, group_offset=0):
        """U [B,T_in,n_in] (frames ordered by group) -> Y [B,T-transient,n_out] (device, unscaled).
        group_offset: global index of group 0 (noise key and weight set follow the global group)."""
        torch = self.torch
        U = _as_dev(U, torch, self.device)
        b, t_in = U.shape[0], U.shape[1]
        T = t_in if T is None else int(T)
        if not (0 <= int(transient) < T and t_in <= T):
            raise ValueError(f"need 0 <= transient < T and T_in <= T (transient={transient}, T_in={t_in}, T={T})")
        g = (b + frames_per_group - 1) // frames_per_group
        self._check_groups(g)
        if self.W_out is None:
            raise AttributeError("W_out: fit (or set_readout) before predict")
        if self.W_out.shape[0] < g:
            raise ValueError(f"readout holds {None if self.W_out is None else self.W_out.shape[0]} groups, batch needs {g}")
        x0 = _as_dev(x0, torch, self.device)
        y0 = _as_dev(y0, torch, self.device)
        nm, nz = self._noise_args(noise_mode, noise_u, (b, T, self.n_reservoir))
        with torch.cuda.device(self.device):
            if out is None:
                out = torch.empty((b, T - transient, self.n_outputs), dtype=torch.float64, device=self.device)
            # reservoirs beyond 1024 units run one GEMM launch per step out of a caller-owned workspace
            wbytes = self.lib.esn_predict_workspace_bytes(PRECISIONS[precision], C.byref(self.shape), b,
                                                          int(frames_per_group))
            ws = None
            if wbytes:
                ws = getattr(self, "_workspace", None)
                if ws is None or ws.numel() < wbytes:
                    ws = self._workspace = torch.empty(wbytes, dtype=torch.uint8, device=self.device)
            check(self.lib.esn_predict_batch(
                PRECISIONS[precision], C.byref(self.shape), ptr(self.packed_weights(precision)),
                ptr(self.packed_readout(precision)), ptr(self.in_scale), ptr(self.in_shift),
                ptr(self.t_scale), ptr(self.t_shift), ptr(U), b, int(frames_per_group), t_in, T,
                int(transient), ptr(x0), ptr(y0), self.noise, nm, ptr(nz), int(seed) & (2**64 - 1),
                int(group_offset), ptr(out), ptr(ws), wbytes, _lib.stream_handle()), "esn_predict_batch")
            if wbytes and precision == "f64" and b == 1:
                self._cluster_err = ws[wbytes - 64:wbytes - 60]      # error word of the single-sequence cluster kernel
        return out

    def raise_if_cluster_timed_out(self):
        """Host-synchronising check after a single-sequence float64 call (the 2-D drop-in makes it when it copies the
        result to the host): the cluster kernel's workgroups wait for each other with bounded spins and raise an
        error word instead of hanging (include/esn_hip.h, esn_predict_batch: workspace)."""
        w = getattr(self, "_cluster_err", None)
        if w is not None:
            self._cluster_err = None
            if int(w.view(self.torch.int32).item()) != 0:
                raise _lib.EsnHipError("single-sequence cluster kernel: a workgroup timed out waiting for the others "
                                       "(the device is oversubscribed); outputs are invalid")

    # ------------------------------------------------------------------ detector tail
    def detect_count(self, Y, tx_bits, p_i, frames_per_group, n_sub, n_t, bits_per_sym,
                     err=None, bits=None, want_xhat=False):
        """Y [B,N,2 n_t] -> per-group int64 (errors, bits) accumulated into err/bits."""
        torch = self.torch
        Y = _as_dev(Y, torch, self.device)
        b = Y.shape[0]
        g = (b + frames_per_group - 1) // frames_per_group
        p_i = _as_dev(p_i, torch, self.device)
        tx_bits = _as_dev(tx_bits, torch, self.device, dtype=torch.uint8)
        with torch.cuda.device(self.device):
            if err is None:
                err = torch.zeros(g, dtype=torch.int64, device=self.device)
            if bits is None:
                bits = torch.zeros(g, dtype=torch.int64, device=self.device)
            xh = torch.empty((b, n_sub, 2 * n_t), dtype=torch.float64, device=self.device) if want_xhat else None
            check(self.lib.esn_detect_count(ptr(Y), b, int(frames_per_group), int(n_sub), int(n_t),
                                            int(bits_per_sym), ptr(p_i), ptr(tx_bits), ptr(err), ptr(bits),
                                            ptr(xh), _lib.stream_handle()), "esn_detect_count")
        return (err, bits, xh) if want_xhat else (err, bits)

    # ------------------------------------------------------------------ helpers
    def _check_groups(self, g):
        for name in ("in_scale", "in_shift", "t_scale", "t_shift"):
            t = getattr(self, name)
            if t is not None and t.shape[0] < g:
                raise ValueError(f"{name} holds {t.shape[0]} groups, batch has {g}")
        if self.n_wsets > 1 and g % self.n_wsets and g > self.n_wsets:
            pass  # weight set = group % n_wsets by contract

    def _noise_args(self, noise_mode, noise_u, shape):
        if self.noise == 0.0 or noise_mode in (None, "none"):
            return _lib.NOISE_NONE, None
        if noise_mode == "tensor":
            nz = _as_dev(noise_u, self.torch, self.device)
            if nz is None or tuple(nz.shape) != tuple(shape):
                raise ValueError(f"noise_u must have shape {shape}")
            return _lib.NOISE_TENSOR, nz
        if noise_mode == "counter":
            return _lib.NOISE_COUNTER, None
        raise ValueError("noise_mode must be 'none', 'tensor' or 'counter'")
