#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from rocprofv3 PMC counters (run ON the GPU box).

Two separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; counters are
collected alone, without trace domains -- MI355X_MICROARCH.md 'rocprofv3 PMC slots' / 'HBM'):

    python tools/pmc_traffic.py [--precision f16] [--tag r02] -> profiles/<tag>_pmc_traffic_<precision>.json

Corrections applied exactly as the guide prescribes: both counters are in KiB (x1024);
on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads, so the read
side is DOUBLED; WRITE_SIZE is exact for 16-B-per-lane streaming stores (our output rows are
64-byte partial lines, so the write side is reported separately and may exceed the
algorithmic bytes).

The predict launches are selected by PARSING the template arguments of the kernel name
(recur_mfma_kernel<TR, NW, MT, NT, HARVEST=false, ...>) and by the launch's grid size, which
must equal the predict launch's tile count -- round 1 matched the substring "false", which the
harvest instantiation <..., true, 2, false> also carries, and averaged both kinds of launch.
"""
import argparse
import json
import os
import shutil

from pmc_common import ROOT, recur_kind, run_pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--blocks", default="2048")
    a = ap.parse_args()
    bench_args = ["--precision", a.precision, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra",
                  "--blocks", a.blocks]
    out = {"precision": a.precision, "blocks": int(a.blocks), "frames_per_block": 75,
           "kernel": "esn::recur_skew16_kernel / esn::recur_mfma_kernel (predict)",
           "kernel_match": "recur_skew16_kernel<...> (predict only) or template argument HARVEST == false of "
                           "recur_mfma_kernel<...>"}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(ROOT, "gpurun_out", f"pmc_{counter.lower()}_{a.precision}")
        shutil.rmtree(d, ignore_errors=True)
        rows = run_pass([counter], d, bench_args)
        sel = [r for r in rows if r.get("Counter_Name") == counter and recur_kind(r.get("Kernel_Name", "")) == "predict"]
        vals = [float(r["Counter_Value"]) for r in sel]
        out[counter + "_KiB_per_launch"] = sum(vals) / len(vals) if vals else None
        out[counter + "_launches"] = len(vals)
        out[counter + "_KiB_each"] = vals
        out[counter + "_grid_sizes"] = sorted({r.get("Grid_Size") for r in sel})
        keep = os.path.join(ROOT, "profiles", f"{a.tag}_pmc_{counter.lower()}_{a.precision}_rows.csv")
        with open(keep, "w") as f:       # the recurrence rows only (the full CSV lists every torch kernel)
            f.write("Kernel_Name,Grid_Size,Counter_Name,Counter_Value\n")
            for r in rows:
                if r.get("Counter_Name") == counter and recur_kind(r.get("Kernel_Name", "")):
                    f.write('"%s",%s,%s,%s\n' % (r["Kernel_Name"], r.get("Grid_Size"), counter, r["Counter_Value"]))
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
