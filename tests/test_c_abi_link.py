"""The boundary is a plain-C ABI: include/esn_hip.h must compile as C (gcc -std=c99 -pedantic) and a
C program must link against libesn_hip.so and call the host-only entry points (no GPU needed)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SRC = r'''
#include <stdio.h>
#include <string.h>
#include "esn_hip.h"
int main(void) {
    esn_shape_t sh = {512, 16, 8, 1, 1};
    if (esn_abi_version() < 1) return 1;
    if (esn_tile_frames(ESN_F16, &sh) != 128) return 2;
    if (esn_tile_frames(ESN_F32, &sh) != 64) return 3;
    if (esn_packed_weights_bytes(ESN_F16, &sh) != (size_t)2 * 2 * 512 * 544) return 4;   /* two image cuts */
    sh.n_res = 0;
    if (esn_tile_frames(ESN_F64, &sh) >= 0) return 5;
    if (!strstr(esn_last_error(), "invalid shape")) return 6;
    /* argument errors are reported without touching a device */
    if (esn_detect_count(0, 1, 1, 128, 4, 4, 0, 0, 0, 0, 0, 0) != -1) return 7;
    if (esn_gen_frames(1, 1, 100, 7, 4, 8, 8, 4, 0, (const double*)1, (const double*)1, 1e-5, (const double*)1,
                       0, 0, 0, 0, (uint8_t*)1, 0, (double*)1, 0) != -1) return 8;   /* N not a power of two */
    /* the host-memory front ends validate before they stage anything */
    if (esn_predict_batch_mem(2, ESN_F64, &sh, 0, 0, 0, 0, 0, 0, 0, 1, 1, 4, 4, 0, 0, 0, 0.0, 0, 0, 0, 0, 0, 0, 0, 0) != -1)
        return 9;
    if (!strstr(esn_last_error(), "memory kind")) return 10;
    if (esn_detect_count_mem(ESN_MEM_HOST, 0, 1, 1, 128, 4, 4, 0, 0, 0, 0, 0, 0) != -1) return 11;
    if (esn_device_free(0) != 0) return 12;
    printf("abi %d ok\n", esn_abi_version());
    return 0;
}
'''


def test_header_is_c_and_library_links_from_c(tmp_path):
    from esn_ofdm_mimo_amd import build
    lib = build.build_library(verbose=False)
    src = tmp_path / "abi.c"
    src.write_text(C_SRC)
    exe = tmp_path / "abi"
    libdir = os.path.dirname(lib)
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
           "-o", str(exe), "-L", libdir, "-lesn_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "ok" in r.stdout


def test_bench_flop_accounting_matches_survey():
    sys.path.insert(0, ROOT)
    import bench
    # C4 76 909 056 (N=512: 290 916 864), C5 1 175 751 168, C3 11 709 504, C2 10 858 496
    assert bench.flop_per_frame(512, 16, 8, 138) == 76909056
    assert bench.flop_per_frame(512, 16, 8, 522) == 290916864
    assert bench.flop_per_frame(2048, 16, 8, 138) == 1175751168
    assert bench.flop_per_frame(100, 4, 4, 522) == 11709504
    assert bench.flop_per_frame(100, 2, 2, 512) == 10858496
