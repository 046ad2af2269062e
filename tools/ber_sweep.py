#!/usr/bin/env python3
"""BER-vs-Eb/No sweep on the GPU in the reference's own output format (the CSV the north-star
driver writes, Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:636-641): EbNo(dB),ESN_uncoded,MMSE_uncoded
(the LDPC-coded columns belong to SURVEY row f-4, not built).  Same configuration as the reference's
published run (N=128, 4x8 TDL-B, 16-QAM, N_res=300 by default) but thousands of channel draws per
point instead of 14.

    python tools/ber_sweep.py [--n-res 300] [--blocks 1024] [--precision f16] [--out file.csv]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PUBLISHED = {  # results/results_4x8_cdl_coded_uncoded/CDLB_run_01/results_ber.csv (N_res=300, 14 draws/point)
    0: (0.39036, 0.31962), 3: (0.35693, 0.25236), 6: (0.32307, 0.18538), 9: (0.28086, 0.12929),
    12: (0.24451, 0.07861), 15: (0.20868, 0.05450), 18: (0.18600, 0.03449), 21: (0.16521, 0.02703),
    24: (0.15912, 0.02187), 27: (0.16198, 0.01991), 30: (0.15690, 0.01892)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-res", type=int, default=300)
    ap.add_argument("--blocks", type=int, default=1024)
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--reservoirs", default="shared", choices=["shared", "per_block"])
    ap.add_argument("--out", default=None)
    ap.add_argument("--coded", action="store_true", help="add the LDPC-coded columns (SURVEY row f-4)")
    a = ap.parse_args()
    import torch
    from esn_ofdm_mimo_amd.montecarlo import DetectorSweep, LinkParams
    prm = LinkParams()
    F = prm.coherence_symbols
    fitp = a.precision if a.precision in ("f16", "bf16") else "f32"
    sw = DetectorSweep(prm, n_reservoir=a.n_res, noise=0.001, seed=7, precision=a.precision, fit_precision=fitp,
                       reservoirs=a.reservoirs, pool=16)
    t0 = time.perf_counter()
    if a.coded:
        from esn_ofdm_mimo_amd.coded import LdpcCode
        from esn_ofdm_mimo_amd.montecarlo import coded_ber_point
        pub_coded = {0: (0.39209, 0.31793), 3: (0.35876, 0.25365), 6: (0.31658, 0.17010), 9: (0.27584, 0.07085),
                     12: (0.24670, 0.006145), 15: (0.20165, 0.000163), 18: (0.18246, 0.0), 21: (0.15071, 0.0),
                     24: (0.14270, 0.0), 27: (0.13496, 0.0), 30: (0.12669, 0.0)}
        code = LdpcCode(prm.n_sub * prm.m, 4, 8, seed=11)
        print(f"[LDPC] regular code: n={code.n}, k={code.k}, rate={code.k / code.n:.3f}")
        lines = ["EbNo(dB),ESN_uncoded,MMSE_uncoded,ESN_coded,MMSE_coded"]      # the reference's header (:639)
        for si, ebno in enumerate(sorted(PUBLISHED)):
            r = coded_ber_point(sw, code, float(ebno), si, a.blocks, F, seed=7)
            lines.append(f"{ebno},{r['ESN_uncoded']},{r['MMSE_uncoded']},{r['ESN_coded']},{r['MMSE_coded']}")
            print(f"Eb/No {ebno:2d} dB  ESN {r['ESN_uncoded']:.5f}/{r['ESN_coded']:.5f} (pub {PUBLISHED[ebno][0]:.5f}/"
                  f"{pub_coded[ebno][0]:.5f})   MMSE {r['MMSE_uncoded']:.5f}/{r['MMSE_coded']:.6f} "
                  f"(pub {PUBLISHED[ebno][1]:.5f}/{pub_coded[ebno][1]:.6f})   a_mmse[0] {r['a_mmse'][0]:+.3f}", flush=True)
        print(f"{len(PUBLISHED)} coded points in {time.perf_counter() - t0:.1f} s")
        if a.out:
            open(a.out, "w").write("\n".join(lines) + "\n")
        return
    lines = ["EbNo(dB),ESN_uncoded,MMSE_uncoded"]
    for si, ebno in enumerate(sorted(PUBLISHED)):
        d = sw.src.blocks_fast(float(ebno), si, 0, a.blocks, F, with_ls_pilot=True)
        sw.set_snr(float(ebno), a.blocks)
        E = sw.train(d["pilot_y"], d["pilot_x"], seed=si)
        sw.repair_fit(E)
        err = torch.zeros(a.blocks, dtype=torch.int64, device=sw.device)
        nb = torch.zeros(a.blocks, dtype=torch.int64, device=sw.device)
        sw.detect(d["data_y"], d["data_bits"], F, err, nb, seed=si)
        H = sw.src.estimate_channel(d["pilot_bits"], d["pilot_y_ls"], float(ebno))
        e2, n2 = sw.src.mmse_detect_count(H, d["data_y"], d["data_bits"], F, float(ebno))
        torch.cuda.synchronize()
        esn = float(err.sum()) / float(nb.sum())
        mmse = float(e2.sum()) / float(n2.sum())
        lines.append(f"{ebno},{esn},{mmse}")
        print(f"Eb/No {ebno:2d} dB  ESN {esn:.5f} (published {PUBLISHED[ebno][0]:.5f})   "
              f"MMSE {mmse:.5f} (published {PUBLISHED[ebno][1]:.5f})", flush=True)
    dt = time.perf_counter() - t0
    print(f"{len(PUBLISHED)} points x {a.blocks} blocks x {F} symbols in {dt:.1f} s "
          f"({len(PUBLISHED) * a.blocks * F / dt:.0f} symbols/s incl. generation, baseline and host sync)")
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
