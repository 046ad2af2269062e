"""Shared helpers of the PMC tools: run one rocprofv3 counter pass of bench.py (counters alone,
no trace domains; the program directly after `--`) and classify kernel names."""
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pass(counters, outdir, bench_args):
    cmd = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", outdir, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), *bench_args]
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, check=True, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = []
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


_TPL = re.compile(r"recur_mfma_kernel<([^>]*)>")


def recur_kind(name):
    """'predict' / 'harvest' for an instantiation of recur_mfma_kernel<TR, NW, MT, NT, HARVEST, NOISE, SKEW>
    (the HARVEST template argument is the FIFTH one), else None."""
    if "recur_skew16_kernel" in name:          # fp16/bf16 predict at 257..512 units on 16x16x32 MFMAs: predict only
        return "predict"
    m = _TPL.search(name)
    if not m:
        return None
    args = [a.strip() for a in m.group(1).split(",")]
    if len(args) < 5:
        return None
    return "harvest" if args[4] in ("true", "1") else "predict"


def kernel_class(name):
    k = recur_kind(name)
    if k:
        return ("recur_skew16_" if "recur_skew16_kernel" in name else "recur_mfma_") + k
    for tag in ("harvest_cluster", "recur_rs", "bigh_step", "big_step", "big_prep", "recur_cluster", "readout_chol_big", "recur_f64_mfma", "recur_f64", "readout_chol", "readout_qr", "detect_count", "pack_readout", "pack_weights",
                "gen_frames", "gen_taps"):
        if tag in name:
            return tag
    return None
