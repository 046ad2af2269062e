import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from esn_ofdm_mimo_amd import batched
G, rows, cols, n_out, tr = 2048, 128, 528, 8, 10
bank = batched.ReservoirBank(cols - 2, n_out, 2, np.zeros((2, 2)), np.zeros((2, cols - 2)), np.zeros((2, n_out)))
E = torch.randn(G, rows + tr, cols, dtype=torch.float64, device="cuda")
D = torch.randn(G, rows + tr, n_out, dtype=torch.float64, device="cuda")
for _ in range(2): bank.solve(E, D, tr, method="chol")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): bank.solve(E, D, tr, method="chol")
torch.cuda.synchronize(); print("skip", os.environ.get("ESN_CHOL_SKIP", "0"), "ms/solve", (time.perf_counter() - t0) / 5 * 1e3)
