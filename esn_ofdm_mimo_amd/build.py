"""Builds libesn_hip.so (the C-ABI library of include/esn_hip.h) in-tree with hipcc
for gfx950.  Called by __graft_entry__.build(); also runnable as a script."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libesn_hip.so")
SOURCES = ["esn_api.hip", "esn_host.hip", "esn_pack.hip", "esn_recur_f64.hip", "esn_recur_f64_mfma.hip", "esn_recur_mfma.hip",
           "esn_recur_big.hip", "esn_recur_cluster.hip", "esn_recur_mfma_f32.hip", "esn_recur_mfma_f16.hip", "esn_recur_mfma_bf16.hip", "esn_recur_skew16.hip", "esn_harvest_cluster.hip",
           "esn_solve.hip", "esn_detect.hip", "esn_gen.hip", "esn_baseline.hip", "esn_coded.hip"]
# the register-resident-state predict kernel is a kept negative result: it is compiled only into
# experiment builds (`ESN_WITH_RS=1 python esn_ofdm_mimo_amd/build.py --variant rs`), never into the product library
WITH_RS = os.environ.get("ESN_WITH_RS") == "1"
# -fno-slp-vectorize: SLP turns adjacent float32 adds/fmas into v_pk_*_f32, which issue far slower
# than two scalar ops beside MFMAs (measured: predict kernel 13.97 -> 13.42 ms)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
           [os.path.join(PKG, "..", "include", "esn_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, stamps=False):
    """stamps=True builds the diagnostic variant libesn_hip_stamps.so (-DESN_STAMPS: in-kernel
    s_memtime phase stamps; never used for timing or by the product)."""
    global LIB
    if stamps:
        extra = ["-DESN_STAMPS"] + os.environ.get("ESN_EXTRA_FLAGS", "").split()
        return _build(os.path.join(PKG, "libesn_hip_stamps.so"), "build_stamps", extra, verbose)
    if not force and not _stale():
        return LIB
    return _build(LIB, "build", [], verbose)


def _build(LIB, bdir, extra, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(PKG, bdir), exist_ok=True)
    sources = SOURCES + (["esn_recur_rs.hip"] if WITH_RS and LIB.endswith("_rs.so") else [])
    if len(sources) > len(SOURCES):
        extra = list(extra) + ["-DESN_WITH_RS"]
    for src in sources:
        obj = os.path.join(PKG, bdir, src.replace(".hip", ".o"))
        objs.append(obj)
        per_file = os.environ.get("ESN_FLAGS_" + src.replace(".hip", "").upper(), "").split()
        cmd = [hipcc, *FLAGS, *extra, *per_file, "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
        if verbose and out.strip():
            sys.stderr.write(out.decode())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:     # tuning builds: libesn_hip_<name>.so with ESN_EXTRA_FLAGS (load via ESN_HIP_LIB)
        name = sys.argv[sys.argv.index("--variant") + 1]
        print(_build(os.path.join(PKG, f"libesn_hip_{name}.so"), "build_" + name,
                     os.environ.get("ESN_EXTRA_FLAGS", "").split(), True))
    else:
        print(build_library(force="--force" in sys.argv, stamps="--stamps" in sys.argv))
