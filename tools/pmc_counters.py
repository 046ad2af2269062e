#!/usr/bin/env python3
"""MFMA utilisation, issue / wait shares, LDS conflicts and L2 hit rate per kernel of the bench step
(run ON the GPU box): separate rocprofv3 --pmc passes (8 SQ slots / 4 TCC slots per pass on gfx950),
counters alone, the program directly after `--`.

    python tools/pmc_counters.py [--tag r02] [--precision f16]
        -> profiles/<tag>_pmc_counters_<precision>.json   (per kernel class: mean counter values per launch
                                                           + the derived ratios below)

Derived per kernel (units: MI355X_MICROARCH.md 'Per-instruction cycle constants', row s_memtime):
    mfma_busy_frac     SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 [per XCD] * 1024 SIMDs) -- share of the
                       SIMD-cycles of the launch in which the matrix pipe is busy (the rocprof MfmaUtil formula)
    mfma_coexec_frac   SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES
    wait / issue       SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (quad-cycles)
    lds_conflict_frac  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
    l2_hit             TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
    clock_ghz          GRBM_GUI_ACTIVE / 8 / dispatch duration of the SAME (profiled) pass -- 'DVFS give-back':
                       profiled passes clock lower than un-profiled ones; a >= 10 ms dispatch reads within 3 %
"""
import argparse
import collections
import json
import os
import shutil

from pmc_common import ROOT, kernel_class, run_pass

PASSES = [
    ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
     "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE"],
    ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
     "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCP_TCC_READ_REQ_sum"],
    ["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQ_IFETCH", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
     "SQ_INSTS_LDS", "SQ_WAVE_CYCLES"],
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--blocks", default="2048")
    ap.add_argument("--extra", default="", help="extra bench.py arguments (quoted)")
    a = ap.parse_args()
    bench_args = ["--precision", a.precision, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra",
                  "--blocks", a.blocks, *a.extra.split()]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for i, counters in enumerate(PASSES):
        d = os.path.join(ROOT, "gpurun_out", f"pmc_pass{i}_{a.precision}")
        shutil.rmtree(d, ignore_errors=True)
        try:
            rows = run_pass(counters, d, bench_args)
        except Exception as e:          # noqa: BLE001  (a counter the driver refuses: keep the other passes)
            print(f"pass {i} failed: {e}")
            continue
        for r in rows:
            kc = kernel_class(r.get("Kernel_Name", ""))
            if kc:
                acc[kc][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Counter_Name"] == counters[0] and r.get("End_Timestamp"):
                    acc[kc]["_duration_ns_pass%d" % i].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out = {"precision": a.precision, "blocks": int(a.blocks), "bench_args": bench_args, "kernels": {}}
    for kc, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        m["launches"] = max(len(v) for v in cs.values())
        g = m.get("GRBM_GUI_ACTIVE")
        if g and m.get("_duration_ns_pass0"):
            m["clock_ghz"] = g / 8.0 / m["_duration_ns_pass0"]
        if g and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            m["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (g / 8.0 * 1024.0)
        if m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            m["mfma_coexec_frac"] = m.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / m["SQ_VALU_MFMA_BUSY_CYCLES"]
        w = m.get("SQ_WAVE_CYCLES")
        if w:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in m:
                    m[c.lower() + "_frac"] = m[c] / w
        if m.get("SQ_LDS_IDX_ACTIVE"):
            m["lds_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
        if m.get("SQC_ICACHE_REQ"):
            m["icache_miss_frac"] = m.get("SQC_ICACHE_MISSES", 0.0) / m["SQC_ICACHE_REQ"]
        if m.get("TCC_HIT_sum") is not None and (m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)) > 0:
            m["l2_hit"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        out["kernels"][kc] = m
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    path = os.path.join(ROOT, "profiles", f"{a.tag}_pmc_counters_{a.precision}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
