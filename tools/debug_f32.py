import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from esn_ofdm_mimo_amd import batched
from oracle import esn_oracle as eo
rs = np.random.RandomState(3)
n_in, n_out, n_res, t, G = 4, 2, 40, 4, 2
w, w_in, w_fb = eo.draw_weights(rs, n_in, n_out, n_res, 0.9, 0.1)
bank = batched.ReservoirBank(n_in, n_out, n_res, w, w_in, w_fb, noise=0.0)
u, d = rs.randn(G, t, n_in), rs.randn(G, t, n_out)
E64 = bank.harvest(u, d, precision="f64").cpu().numpy()
E32 = bank.harvest(u, d, precision="f32").cpu().numpy()
np.set_printoptions(precision=4, linewidth=200, suppress=True)
for row in range(t):
    print("row", row, "max diff", np.abs(E64[0,row]-E32[0,row]).max())
print(E64[0,1,:12]); print(E32[0,1,:12])
print(E64[0,1,-4:], E32[0,1,-4:])
pre = w_in @ u[0,1] + w_fb @ d[0,0]
print("expected x1", np.tanh(pre)[:12])
print("w_in@u only", np.tanh(w_in @ u[0,1])[:12])
print("w_fb@d only", np.tanh(w_fb @ d[0,0])[:12])
print("sorted expected", np.sort(E64[0,1,:40])[:10])
print("sorted got     ", np.sort(E32[0,1,:40])[:10])
# structured probe: W_in picks input c into rows 4c..4c+3, no feedback
w_in2 = np.zeros_like(w_in); 
for r in range(n_res): w_in2[r, r % n_in] = 0.1 * (1 + r)
bank2 = batched.ReservoirBank(n_in, n_out, n_res, w * 0, w_in2, w_fb * 0, noise=0.0)
u2 = np.zeros((1, 2, n_in)); u2[0, 1] = [1, 2, 3, 4]
d2 = np.zeros((1, 2, n_out))
e64 = bank2.harvest(u2, d2, precision="f64").cpu().numpy()[0, 1, :40]
e32 = bank2.harvest(u2, d2, precision="f32").cpu().numpy()[0, 1, :40]
print(np.arctanh(e64)); print(np.arctanh(np.clip(e32, -0.999999, 0.999999)))
