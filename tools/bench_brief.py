#!/usr/bin/env python3
"""Runs bench.py (no CPU leg) and prints the few numbers used when tuning the predict kernel."""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "bench.py", "--cpu-blocks", "0", *sys.argv[1:]], capture_output=True, text=True)
if out.returncode != 0:
    print(out.stderr[-2000:])
    sys.exit(1)
d = json.loads(out.stdout.strip().splitlines()[-1])
print(f"symbols/s {d['value']:.4g}  ms/step {d['ms_per_step']:.2f}  predict_ms {d['predict_kernel_ms']:.2f}  "
      f"frac {d['roofline']['frac']:.3f}  ber {d['ber']:.6f}")
