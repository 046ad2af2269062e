#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path when the boundary hands over HOST buffers (esn_predict_batch_mem with
ESN_MEM_HOST, pageable NumPy arrays) beside the device-resident rate bench.py reports -- the figure DESIGN.md 1
quotes; it is never bench.py's `value`.   usage: tools/host_abi_rate.py [blocks] [precision]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from esn_ofdm_mimo_amd import _lib as L, batched
    from esn_ofdm_mimo_amd.montecarlo import draw_reservoir
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    precision = sys.argv[2] if len(sys.argv) > 2 else "f16"
    n_res, n_in, n_out, F, T, tr = 512, 16, 8, 75, 138, 10
    lib, prec = L.load(), L.PRECISIONS[precision]
    w, w_in, w_fb = draw_reservoir(n_in, n_out, n_res, 0.9, 0.1, 1)
    rs = np.random.RandomState(0)
    B = blocks * F
    U = rs.randn(B, T - 3, n_in) * 5e-3
    w_out = rs.randn(blocks, n_out, n_res + n_in) * 0.02
    Y = np.empty((B, T - tr, n_out))
    shape = L.Shape(n_res, n_in, n_out, 1, 1)
    hp = lambda a: C.c_void_p(a.ctypes.data)
    pw = lib.esn_device_alloc(lib.esn_packed_weights_bytes(prec, C.byref(shape)))
    pwo = lib.esn_device_alloc(lib.esn_packed_readout_bytes(prec, C.byref(shape)) * blocks)
    L.check(lib.esn_pack_weights_mem(L.MEM_HOST, prec, C.byref(shape), hp(w), hp(w_in), hp(w_fb), pw, None), "pack")
    L.check(lib.esn_pack_readout_mem(L.MEM_HOST, prec, C.byref(shape), blocks, hp(w_out), pwo, None), "pack_readout")

    def host_call():
        L.check(lib.esn_predict_batch_mem(L.MEM_HOST, prec, C.byref(shape), pw, pwo, None, None, None, None, hp(U), B, F,
                                          T - 3, T, tr, None, None, 1e-3, L.NOISE_COUNTER, None, 1, 0, hp(Y), None, 0,
                                          None), "predict")
    host_call()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        host_call()
    t_host = (time.perf_counter() - t0) / reps

    bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=1e-3)
    bank.set_readout(w_out)
    Ud = torch.as_tensor(U, device="cuda")
    yd = bank.predict(Ud, F, T=T, transient=tr, precision=precision, noise_mode="counter", seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        yd = bank.predict(Ud, F, T=T, transient=tr, precision=precision, noise_mode="counter", seed=1)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / reps
    same = bool(np.array_equal(yd.cpu().numpy(), Y))
    gb = (U.nbytes + Y.nbytes) / 1e9
    print(f"predict, {precision}, {blocks} blocks x {F} frames = {B} frames, N_res={n_res}: "
          f"host arrays (ESN_MEM_HOST, pageable, {gb:.2f} GB over PCIe per call) {t_host * 1e3:.1f} ms = "
          f"{B / t_host / 1e6:.3f} M symbols/s ({gb / t_host:.1f} GB/s); device-resident {t_dev * 1e3:.2f} ms = "
          f"{B / t_dev / 1e6:.2f} M symbols/s; outputs identical: {same}")
    lib.esn_device_free(pw)
    lib.esn_device_free(pwo)


if __name__ == "__main__":
    main()
