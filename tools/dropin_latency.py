import time, numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
from esn_ofdm_mimo_amd import pyESN
from oracle import esn_oracle as eo
rs = np.random.RandomState(0)
n_in, n_out, n_res, T = 16, 8, 512, 138
u, d = rs.randn(T, n_in) * 0.1, rs.randn(T, n_out) * 0.1
kw = dict(spectral_radius=0.9, sparsity=0.1, noise=0.001, input_scaling=0.05 * np.ones(n_in), input_shift=np.zeros(n_in),
          teacher_scaling=5e-3 * np.ones(n_out), teacher_shift=np.zeros(n_out))
e = pyESN.ESN(n_in, n_out, n_res, random_state=1, **kw)
o = eo.OracleESN(n_in, n_out, n_res, random_state=1, **kw)
for name, m in (("hip drop-in", e), ("numpy oracle", o)):
    m.fit(u, d, 10); m.predict(u, 10, continuation=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): m.fit(u, d, 10)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(20): m.predict(u, 10, continuation=False)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: fit {1e3*(t1-t0)/5:.1f} ms, predict {1e3*(t2-t1)/20:.1f} ms per 138-sample sequence")
