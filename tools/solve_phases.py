"""Time budget of the LDS-resident Cholesky read-out solve (csrc/esn_solve.hip readout_chol_kernel) on the
headline shape: the whole kernel, then with phases switched off through the `chol_skip` debug knob
(1: Gram over one k-chunk only, 2: one 16-column block of the factorisation, 4: no triangular solves,
8: no W_out = A^T alpha).  Differences of the totals are the phases' shares.  Results with a knob set are
wrong by construction -- this is a profiling tool.

    python tools/solve_phases.py [--groups 2048] [--e32]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from esn_ofdm_mimo_amd import _lib, batched
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=2048)
    ap.add_argument("--rows", type=int, default=139)       # S + 1 of the headline frame (135 + 3 + 1)
    ap.add_argument("--transient", type=int, default=11)   # -> 128 fitted rows
    ap.add_argument("--n-res", type=int, default=512)
    ap.add_argument("--n-in", type=int, default=16)
    ap.add_argument("--n-out", type=int, default=8)
    ap.add_argument("--e32", action="store_true", help="float32 design matrix (the fp16/f32 harvest's output)")
    a = ap.parse_args()
    rs = np.random.RandomState(0)
    w = rs.rand(a.n_res, a.n_res) - 0.5
    bank = batched.ReservoirBank(a.n_in, a.n_out, a.n_res, w, rs.rand(a.n_res, a.n_in), rs.rand(a.n_res, a.n_out))
    dev = torch.device("cuda:0")
    cols = a.n_res + a.n_in
    E = torch.randn(a.groups, a.rows, cols, device=dev, dtype=torch.float32 if a.e32 else torch.float64) * 0.1
    D = torch.randn(a.groups, a.rows, a.n_out, device=dev, dtype=torch.float64)

    def timed(skip):
        _lib.debug_set("chol_skip", str(skip))
        for _ in range(2):
            bank.solve(E, D, a.transient, method="chol")
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for s, e in ev:
            s.record()
            bank.solve(E, D, a.transient, method="chol")
            e.record()
        torch.cuda.synchronize()
        return float(np.median([s.elapsed_time(e) for s, e in ev]))

    full = timed(0)
    print(f"shape {a.rows - a.transient} x {cols}, {a.groups} systems, E {'f32' if a.e32 else 'f64'}: {full:.3f} ms")
    rows = [("gram k-chunks 2..", 1), ("factorisation blocks 2..", 2), ("triangular solves", 4), ("W_out = A^T alpha", 8)]
    for name, bit in rows:
        t = timed(bit)
        print(f"  without {name:28s} {t:.3f} ms   (share {full - t:+.3f} ms)")
    t = timed(15)
    print(f"  all four off {'':23s} {t:.3f} ms   (launch, staging, first chunk, first block)")
    _lib.debug_set("chol_skip", "0")


if __name__ == "__main__":
    main()
