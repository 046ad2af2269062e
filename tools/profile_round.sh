#!/bin/bash
# Run ON the GPU box: the rocprof evidence of a round, written under profiles/ (tracked) --
#   kernel-trace stats of the default bench command, PMC traffic, PMC counters, phase stamps.
# usage: tools/profile_round.sh r02 [precision]
set -e -o pipefail
TAG=${1:-r02}; PREC=${2:-f16}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
PY=$(python3 -c 'import sys; print(sys.executable)')   # the real interpreter: no exec hop under the profiler
mkdir -p gpurun_out profiles
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- "$PY" bench.py --precision $PREC --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/prof_${TAG}_bench.json 2> gpurun_out/prof_${TAG}.err
STATS=$(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1)
python3 tools/trim_kernel_stats.py "$STATS" profiles/${TAG}_bench_${PREC}_kernel_stats.csv 14
cp gpurun_out/prof_${TAG}_bench.json profiles/${TAG}_bench_${PREC}_profiled.json
echo "[profile] kernel stats done"
python3 tools/pmc_traffic.py --precision $PREC --tag $TAG > gpurun_out/pmc_traffic_$TAG.log 2>&1 || { tail -5 gpurun_out/pmc_traffic_$TAG.log; exit 1; }
echo "[profile] traffic done"
python3 tools/pmc_counters.py --precision $PREC --tag $TAG > gpurun_out/pmc_counters_$TAG.log 2>&1 || { tail -5 gpurun_out/pmc_counters_$TAG.log; exit 1; }
echo "[profile] counters done"
python3 tools/phase_stamps.py $PREC 512 > profiles/${TAG}_phase_stamps_${PREC}.txt 2> gpurun_out/stamps_$TAG.err || { tail -5 gpurun_out/stamps_$TAG.err; exit 1; }
echo "[profile] stamps done"
