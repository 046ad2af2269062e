import os, sys, numpy as np, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tools/..
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch
    from esn_ofdm_mimo_amd.montecarlo import DetectorSweep, LinkParams
    p = LinkParams(); F = 75; G = 64
    sw = DetectorSweep(p, n_reservoir=512, noise=0.001, seed=1234, precision="f16", fit_precision="f16")
    d = sw.src.blocks_fast(12.0, 0, 0, G, F); sw.set_snr(12.0, G)
    sw.train(d["pilot_y"], d["pilot_x"], seed=1)
    U = torch.view_as_real(d["data_y"]).reshape(G * F, p.t_frame, sw.n_in)
    y = sw.bank.predict(U, F, T=p.t_frame + p.delay, transient=p.forget, precision="f16", noise_mode="counter", seed=3)
    np.save(sys.argv[1], y.cpu().numpy())
else:
    for name, lib in (("base", "libesn_hip.so"), ("pk16", "libesn_hip_pk16.so"), ("pk16f", "libesn_hip_pk16f.so")):
        subprocess.run([sys.executable, __file__, f"/tmp/y_{name}.npy"], env=dict(os.environ, ESN_HIP_LIB=os.path.join(ROOT, "esn_ofdm_mimo_amd", lib)), check=True)
    a = np.load("/tmp/y_base.npy")
    for n in ("pk16", "pk16f"):
        b = np.load(f"/tmp/y_{n}.npy"); print(n, "max rel diff", np.abs(a - b).max() / np.abs(a).max(), "identical", np.array_equal(a, b))
