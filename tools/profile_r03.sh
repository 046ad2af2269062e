#!/bin/bash
# Run ON the GPU box: the evidence of round 3.  Everything lands in gpurun_out/profiles_r03/ (merged back by gpurun);
# copy what is to be judged into profiles/ afterwards.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export ESN_STAMPS_LIB=libesn_hip_stamps.so          # prebuilt in-tree (python esn_ofdm_mimo_amd/build.py --stamps)
PY=$(python3 -c 'import sys; print(sys.executable)')
OUT=gpurun_out/profiles_r03
mkdir -p $OUT profiles
step() { echo "[r03] $1"; }

step "default bench line"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > profiles/r03_bench_default.json 2> $OUT/bench_default.err || echo "bench failed"

step "headline kernel stats + PMC + stamps"
timeout -k 10 500 bash tools/profile_round.sh r03 f16 > $OUT/profile_round.log 2>&1 || tail -5 $OUT/profile_round.log
timeout -k 10 300 bash tools/profile_precisions.sh r03 > $OUT/profile_precisions.log 2>&1 || tail -5 $OUT/profile_precisions.log

step "configs[4] kernel stats"
rm -rf gpurun_out/prof_c5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -- "$PY" bench.py --n-res 2048 --blocks 512 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > profiles/r03_bench_c5_f16.json 2> $OUT/c5.err
python3 tools/trim_kernel_stats.py "$(find gpurun_out/prof_c5 -name '*kernel_stats.csv' | head -1)" profiles/r03_bench_c5_f16_kernel_stats.csv 12

step "sweep kernel stats"
rm -rf gpurun_out/prof_sweep
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sweep -- "$PY" tools/sweep_profile.py f16 512 > profiles/r03_sweep_profiled.json 2> $OUT/sweep.err
python3 tools/trim_kernel_stats.py "$(find gpurun_out/prof_sweep -name '*kernel_stats.csv' | head -1)" profiles/r03_sweep_kernel_stats.csv 14

step "drop-in latency, generator phases, 1e7-frame N_res=2048 sweep"
timeout -k 10 120 python tools/dropin_latency.py 512 > profiles/r03_dropin_latency.txt 2>&1
timeout -k 10 120 python tools/time_gen.py > profiles/r03_time_gen.txt 2>&1
timeout -k 10 300 python tools/big_sweep.py --frames 1e7 --out profiles/r03_big_sweep_nres2048.csv > $OUT/big_sweep.log 2>&1 || tail -3 $OUT/big_sweep.log

cp profiles/r03_* $OUT/ 2>/dev/null
ls -la $OUT | tail -40
