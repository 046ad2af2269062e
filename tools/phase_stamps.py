#!/usr/bin/env python3
"""Diagnostic: where does a timestep of the predict kernel spend its cycles?  Loads the
-DESN_STAMPS build (libesn_hip_stamps.so), runs one predict launch of the bench workload and
prints, per wave of workgroup 0, the summed s_memtime cycles of each phase.  Shares only:
the stamped build forbids overlaps the real kernel has (guide 7, In-kernel stamps)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from esn_ofdm_mimo_amd import _lib, build  # noqa: E402

if "ESN_STAMPS_LIB" not in os.environ:
    build.build_library(stamps=True, verbose=False)
_lib.LIB_PATH = os.path.join(ROOT, "esn_ofdm_mimo_amd", os.environ.get("ESN_STAMPS_LIB", "libesn_hip_stamps.so"))
from esn_ofdm_mimo_amd.montecarlo import DetectorSweep, LinkParams  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 512
params = LinkParams()
F = params.coherence_symbols
sweep = DetectorSweep(params, n_reservoir=512, noise=0.001, seed=1, precision=prec, fit_precision=os.environ.get("ESN_FIT", "f32"))
lib = _lib.load()
lib.esn_debug_set_stamp_buffer.argtypes = [C.c_void_p]
buf = torch.zeros(16 * 8, dtype=torch.int64, device="cuda")
lib.esn_debug_set_stamp_buffer(buf.data_ptr())
data = sweep.src.blocks_fast(12.0, 0, 0, G, F)
sweep.set_snr(12.0, G)
sweep.train(data["pilot_y"], data["pilot_x"], seed=1)
if len(sys.argv) > 3 and sys.argv[3] == "harvest":      # stamps of the last harvest launch (workgroup 0)
    buf.zero_()
    sweep.train(data["pilot_y"], data["pilot_x"], seed=1)
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(16, 8)
    Th = params.t_frame + params.delay
    if raw[4:8, :7].sum() == 0:       # the 4-wave cluster kernel (esn_harvest_cluster.hip): 10 ns ticks and poll rounds
        print("harvest cluster kernel, workgroup 0: microseconds per timestep and poll rounds per gather, per wave")
        print("wave   fetch+GEMM   E+publish   stage+gather   poll rounds   E-row stores   loop edge")
        for w in range(4):
            v = raw[w, :6].astype(float) / Th
            print(f"{w:4d} {v[0] / 100:12.2f} {v[1] / 100:11.2f} {v[2] / 100:14.2f} {v[3]:13.2f} {v[4] / 100:14.2f} {v[5] / 100:11.2f}")
        sys.exit(0)
    names = ["G gemm", "-", "-", "wait", "E + staging", "wait", "E-row copy"]
    print("harvest: cycles per timestep (workgroup 0), per wave")
    print("wave " + " ".join(f"{n:>12s}" for n in names) + "        total")
    for w in range(8):
        v = raw[w, :7].astype(float) / Th
        print(f"{w:4d} " + " ".join(f"{x:12.0f}" for x in v) + f" {v.sum():12.0f}")
    sys.exit(0)
U = torch.view_as_real(data["data_y"]).reshape(G * F, params.t_frame, sweep.n_in)
T = params.t_frame + params.delay
for _ in range(2):
    sweep.bank.predict(U, F, T=T, transient=params.delay + params.cp, precision=prec, noise_mode="counter", seed=3)
torch.cuda.synchronize()
raw = buf.cpu().numpy().reshape(16, 8)
if os.environ.get("ESN_RS", "0") == "1" and prec in ("f16", "bf16"):     # register-state kernel (opt-in: ESN_RS=1)
    print("register-state kernel: cycles per timestep (workgroup 0), per wave")
    print("wave   event: vmcnt wait   event: barrier     tile chunks    read-out chunks   boundary      total")
    for w in range(4):
        v = raw[w, :5].astype(float) / T
        print(f"{w:4d} {v[0]:18.0f} {v[1]:16.0f} {v[2]:15.0f} {v[3]:18.0f} {v[4]:10.0f} {v[2] + v[3] + v[4]:10.0f}")
    sys.exit(0)
hw = raw[:8, 6]
print("HW_ID per wave: " + " ".join(f"w{w}:simd{(int(v) >> 4) & 3}/cu{(int(v) >> 8) & 15}" for w, v in enumerate(hw)))
st = raw[:8, :6].astype(float) / T
names = ["G1 gemm+ro", "wait Ba", "G2 in+fb", "wait Bb", "E act+noise", "wait Bc"]
print(f"precision {prec}: cycles per timestep (workgroup 0), per wave")
print("wave " + " ".join(f"{n:>13s}" for n in names) + "        total")
for w in range(8):
    print(f"{w:4d} " + " ".join(f"{v:13.0f}" for v in st[w]) + f" {st[w].sum():12.0f}")
print("mean " + " ".join(f"{v:13.0f}" for v in st.mean(0)) + f" {st.mean(0).sum():12.0f}")

if raw[8:, :6].any():
    sub = raw[8:16, :6].astype(float) / T
    print("inside P2 (skewed schedule): uf_groups, E first half, prefetch issue, E second half, yU, input fetch")
    for w in range(8):
        print(f"{w:4d} " + " ".join(f"{v:13.0f}" for v in sub[w]))
