#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from rocprofv3 PMC counters (run ON the GPU box).

Two separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; counters are
collected alone, without trace domains -- MI355X_MICROARCH.md 'rocprofv3 PMC slots' / 'HBM'):

    python tools/pmc_traffic.py [--precision f16] -> profiles/<tag>_pmc_traffic.json

Corrections applied exactly as the guide prescribes: both counters are in KiB (x1024);
on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads, so the read
side is DOUBLED; WRITE_SIZE is exact for 16-B-per-lane streaming stores (our stores are
narrower, so the write side is a lower-confidence figure and is reported separately).
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pass(counter, outdir, bench_args):
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", outdir, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), *bench_args]
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, check=True, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = []
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def per_kernel(rows, counter, match):
    vals = [float(r["Counter_Value"]) for r in rows
            if r.get("Counter_Name") == counter and match(r.get("Kernel_Name", ""))]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--blocks", default="2048")
    a = ap.parse_args()
    bench_args = ["--precision", a.precision, "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                  "--blocks", a.blocks]
    is_predict = lambda n: "recur_mfma_kernel" in n and "false" in n.split("recur_mfma_kernel")[1]
    out = {"precision": a.precision, "blocks": int(a.blocks), "kernel": "esn::recur_mfma_kernel (predict)"}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(ROOT, "gpurun_out", f"pmc_{counter.lower()}_{a.precision}")
        rows = run_pass(counter, d, bench_args)
        v, n = per_kernel(rows, counter, is_predict)
        out[counter + "_KiB_per_launch"] = v
        out[counter + "_launches"] = n
    if out["FETCH_SIZE_KiB_per_launch"] is not None and out["WRITE_SIZE_KiB_per_launch"] is not None:
        rd = out["FETCH_SIZE_KiB_per_launch"] * 1024 * 2      # gfx950 correction (guide, HBM section)
        wr = out["WRITE_SIZE_KiB_per_launch"] * 1024
        out.update(read_bytes_per_launch=rd, write_bytes_per_launch=wr, traffic_bytes_per_launch=rd + wr)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    path = os.path.join(ROOT, "profiles", f"{a.tag}_pmc_traffic_{a.precision}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
