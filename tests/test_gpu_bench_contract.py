"""bench.py prints ONE JSON line with the fields the driver and the judge read (task contract,
section 4 of the tier framing): metric / value / unit / n_gpus / steps / warmup / ms_per_step /
higher_is_better / scaling / vs_baseline / dtype / data / config.workload + roofline + cpu_baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_small_run():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--blocks", "64", "--cpu-blocks", "1"], capture_output=True, text=True, cwd=ROOT,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["unit"] == "OFDM symbols/s" and d["value"] > 1e5
    # value = frames per step * steps / wall time
    assert abs(d["value"] - d["config"]["frames_per_step"] * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["unit"] == "TFLOP/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"]
    assert set(c["legs"]) == {"single_thread", "default_blas_threads", "process_per_core"}
    assert c["legs"]["single_thread"]["cores"] == 1 and c["host_cpu"] and c["host_physical_cores"] >= 1
    assert c["cores"] <= c["host_logical_cpus"] and (c["host_cpu_quota"] is None or c["cores"] <= c["host_cpu_quota"] + 0.5)
    # what a user of DetectorSweep.run gets (generation + train + predict + detect + host syncs)
    sw = d["sweep"]
    assert sw["unit"] == d["unit"] and sw["frames"] >= 1e6 and len(sw["ber"]) == len(sw["ebno_db"]) >= 3
    assert abs(sw["value"] - sw["frames"] / sw["wall_s"]) < 1e-6 * sw["value"] and sw["fits_repaired"] == 0
    assert all(a > b for a, b in zip(sw["ber"], sw["ber"][1:]))                      # falling with Eb/No
    # reference-precision and reference-faithful sub-records ride in the same line (N=1)
    for prec, peak in (("f32", 157.3), ("f64", 78.6)):
        p_ = d["precisions"][prec]
        assert p_["value"] > 0 and p_["peak"] == peak and abs(p_["frac"] - p_["achieved_tflops"] / peak) < 1e-9
        assert 0.15 < p_["ber"] < 0.30
    assert d["large_reservoir"]["value"] > 0 and 0.1 < d["large_reservoir"]["frac"] < 1 and 0.1 < d["large_reservoir"]["ber"] < 0.3
    assert d["reservoirs"]["per_block"]["value"] > 0 and 0.15 < d["reservoirs"]["per_block"]["ber"] < 0.30
    # the detector works: BER of the 4x8 ESN at 12 dB is ~0.23 for both the GPU and the oracle sample
    assert 0.15 < d["ber"] < 0.30
