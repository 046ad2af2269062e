"""Drop-in for the reference's ``libs/pyESN.py``: same module-level names
(``correct_dimensions``, ``identity``, ``ESN``), same constructor / ``fit`` /
``predict`` signatures, argument meaning, attributes and error behaviour
(pyESN.py:4-24, :27, :33-91, :154-216, :218-255) -- with every timestep of the
reservoir and the readout solve running in HIP kernels on the MI355X.

2-D inputs ``[T, n]`` reproduce the reference call for call (float64 kernels, the
ESN's own ``RandomState`` supplies the state noise in the reference's draw
order, so a seeded run matches the NumPy reference to float64 round-off).
3-D inputs ``[B, T, n]`` are the batched extension the reference lacks: B
independent sequences through one trained ESN in a single launch
(``precision=`` selects the float32 / fp16 / bf16 MFMA kernels).

What stays on the host: drawing the weights from ``RandomState`` and the
spectral-radius rescale via LAPACK ``eigvals`` (init-time, pyESN.py:93-109 --
draw order and ``eigvals`` are what make weights bit-identical to the
reference for a given seed).  Nothing per-timestep runs on the CPU and there
is no CPU fallback: without the HIP library or a GPU, construction raises.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .batched import ReservoirBank


def correct_dimensions(s, targetlength):
    """None -> None; scalar -> length-n vector; 1-D of the right length -> itself;
    anything else -> ValueError (pyESN.py:4-24)."""
    if s is not None:
        s = np.array(s)
        if s.ndim == 0:
            s = np.array([s] * targetlength)
        elif s.ndim == 1:
            if not len(s) == targetlength:
                raise ValueError("arg must have length " + str(targetlength))
        else:
            raise ValueError("Invalid argument")
    return s


def identity(x):
    return x


class ESN:
    def __init__(self, n_inputs, n_outputs, n_reservoir=200,
                 spectral_radius=0.95, sparsity=0, noise=0.001, input_shift=None,
                 input_scaling=None, teacher_forcing=True, feedback_scaling=None,
                 teacher_scaling=None, teacher_shift=None,
                 out_activation=identity, inverse_out_activation=identity,
                 random_state=None, silent=True, *, precision="f32", device=None, leak_rate=1.0):
        self.n_inputs = n_inputs
        self.n_reservoir = n_reservoir
        self.n_outputs = n_outputs
        self.spectral_radius = spectral_radius
        self.sparsity = sparsity
        self.noise = noise
        self.input_shift = correct_dimensions(input_shift, n_inputs)
        self.input_scaling = correct_dimensions(input_scaling, n_inputs)
        self.teacher_scaling = teacher_scaling      # stored raw, broadcast later (pyESN.py:70-71)
        self.teacher_shift = teacher_shift
        # feedback_scaling is accepted and ignored, exactly like the reference (pyESN.py:35)
        self.out_activation = out_activation
        self.inverse_out_activation = inverse_out_activation
        if out_activation is not identity or inverse_out_activation is not identity:
            raise NotImplementedError(
                "the HIP kernels implement the identity output activation only "
                "(no reference driver uses another); there is no CPU fallback")
        self.random_state = random_state
        if isinstance(random_state, np.random.RandomState):
            self.random_state_ = random_state
        elif random_state:
            try:
                self.random_state_ = np.random.RandomState(random_state)
            except TypeError as e:
                raise Exception("Invalid seed: " + str(e))
        else:
            self.random_state_ = np.random.mtrand._rand
        self.teacher_forcing = teacher_forcing
        self.silent = silent
        self.precision = precision          # kernel used for 3-D (batched) predict
        # extension (the reference has no leak rate): x[t] = (1-a) x[t-1] + a tanh(...) + noise; float64 only
        self.leak_rate = float(leak_rate)
        if self.leak_rate != 1.0:
            self.precision = "f64"
        self.device = device
        _lib.load()
        _lib.require_gpu()
        self.initweights()

    # ---- init (host; pyESN.py:93-109) -----------------------------------------
    def initweights(self):
        rs = self.random_state_
        W = rs.rand(self.n_reservoir, self.n_reservoir) - 0.5
        W[rs.rand(*W.shape) < self.sparsity] = 0
        radius = np.max(np.abs(np.linalg.eigvals(W)))
        self.W = W * (self.spectral_radius / radius)
        self.W_in = rs.rand(self.n_reservoir, self.n_inputs) * 2 - 1
        self.W_feedb = rs.rand(self.n_reservoir, self.n_outputs) * 2 - 1
        self._bank = None

    def _get_bank(self):
        if self._bank is None:
            self._bank = ReservoirBank(self.n_inputs, self.n_outputs, self.n_reservoir, self.W, self.W_in,
                                       self.W_feedb, teacher_forcing=self.teacher_forcing, noise=self.noise,
                                       device=self.device, leak_rate=self.leak_rate)
            ts = None if self.teacher_scaling is None else \
                np.broadcast_to(np.asarray(self.teacher_scaling, dtype=float), (self.n_outputs,))[None]
            tsh = None if self.teacher_shift is None else \
                np.broadcast_to(np.asarray(self.teacher_shift, dtype=float), (self.n_outputs,))[None]
            isc = None if self.input_scaling is None else np.asarray(self.input_scaling, dtype=float)[None]
            ish = None if self.input_shift is None else np.asarray(self.input_shift, dtype=float)[None]
            self._bank.set_scaling(isc, ish, ts, tsh)
        self._bank.noise = float(self.noise)
        return self._bank

    # ---- teacher scaling of the few host-side values fit() returns (pyESN.py:137-152) --------
    def _scale_teacher(self, teacher):
        scale = 1.0 if self.teacher_scaling is None else self.teacher_scaling
        shift = 0.0 if self.teacher_shift is None else self.teacher_shift
        return teacher * scale + shift

    def _unscale_teacher(self, teacher_scaled):
        scale = 1.0 if self.teacher_scaling is None else self.teacher_scaling
        shift = 0.0 if self.teacher_shift is None else self.teacher_shift
        return (teacher_scaled - shift) / scale

    def _report(self, text):
        if not self.silent:
            print(text)

    # ---- fit (pyESN.py:154-216) --------------------------------------------------
    def fit(self, inputs, outputs, transient=0, inspect=False):
        inputs = np.asarray(inputs, dtype=float)
        outputs = np.asarray(outputs, dtype=float)
        if inputs.ndim < 2:
            inputs = np.reshape(inputs, (len(inputs), -1))
        if outputs.ndim < 2:
            outputs = np.reshape(outputs, (len(outputs), -1))
        if inputs.ndim != 2 or outputs.ndim != 2:
            raise ValueError("fit takes one training sequence: inputs [T, n_in], outputs [T, n_out]")
        bank = self._get_bank()
        n = inputs.shape[0]
        # the reference draws rand(n_reservoir) once per update, noise or not (pyESN.py:124-125)
        noise_u = self.random_state_.rand(max(n - 1, 0), self.n_reservoir)
        self._report("harvesting states...")
        E = bank.harvest(inputs[None], outputs[None], precision="f64",
                         noise_mode="tensor" if self.noise else "none", noise_u=noise_u[None])
        self._report("fitting...")
        W_out, status = bank.solve(E, outputs[None], transient)
        bank.set_readout(W_out)
        self.fit_status = int(status[0].item())
        self.W_out = W_out[0].cpu().numpy()
        bank.raise_if_cluster_timed_out()
        ext = E[0]
        self.laststate = ext[-1, :self.n_reservoir].cpu().numpy()
        self.lastinput = inputs[-1, :]
        self.lastoutput = self._scale_teacher(outputs)[-1, :]
        if inspect:                     # the reference's state picture (pyESN.py:200-207), from device memory
            from matplotlib import pyplot
            picture = ext.cpu().numpy().T
            pyplot.figure(figsize=(picture.shape[1] * 0.0025, picture.shape[0] * 0.01))
            pyplot.imshow(picture, aspect="auto", interpolation="nearest")
            pyplot.colorbar()
        pred_train = self._unscale_teacher(bank.torch.matmul(ext, W_out[0].T).cpu().numpy())
        self._report("training error:")
        self._report(np.sqrt(np.mean((pred_train - outputs) ** 2)))
        return pred_train

    # ---- predict (pyESN.py:218-255) ------------------------------------------------
    def predict(self, inputs, transient=0, continuation=True, *, precision=None, noise_mode=None, seed=0):
        inputs = np.asarray(inputs, dtype=float) if not hasattr(inputs, "data_ptr") else inputs
        if inputs.ndim < 2:
            inputs = np.reshape(inputs, (len(inputs), -1))
        bank = self._get_bank()
        if bank.W_out is None:
            if not hasattr(self, "W_out"):
                raise AttributeError("'ESN' object has no attribute 'W_out'")
            bank.set_readout(np.asarray(self.W_out)[None])
        elif hasattr(self, "W_out") and self.W_out is not getattr(self, "_w_out_seen", None):
            bank.set_readout(np.asarray(self.W_out)[None])   # honour a W_out assigned by the caller
        self._w_out_seen = self.W_out
        x0 = y0 = None
        if continuation:
            x0 = np.asarray(self.laststate, dtype=float)[None]
            y0 = np.asarray(self.lastoutput, dtype=float)[None]
        if inputs.ndim == 2:
            n = inputs.shape[0]
            noise_u = self.random_state_.rand(n, self.n_reservoir)
            y = bank.predict(inputs[None], frames_per_group=1, transient=transient, precision=precision or "f64",
                             x0=x0, y0=y0, noise_mode="tensor" if self.noise else "none",
                             noise_u=noise_u[None])
            y = y[0].cpu().numpy()
            bank.raise_if_cluster_timed_out()
            return y
        if inputs.ndim != 3:
            raise ValueError("predict takes [T, n_in] or a batch [B, T, n_in]")
        b = inputs.shape[0]
        y = bank.predict(inputs, frames_per_group=b, transient=transient, precision=precision or self.precision,
                         x0=x0, y0=y0, noise_mode=noise_mode or ("counter" if self.noise else "none"), seed=seed)
        return y if hasattr(inputs, "data_ptr") else y.cpu().numpy()
