#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel-trace stats of the float32 and float64 bench configurations (the sub-records
# `precisions.f32` / `precisions.f64` of the default bench line), trimmed into profiles/.
# usage: tools/profile_precisions.sh r02
set -e -o pipefail
TAG=${1:-r02}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
PY=$(python3 -c 'import sys; print(sys.executable)')   # the real interpreter: no exec hop under the profiler
mkdir -p gpurun_out profiles
for spec in "f32 2048" "f64 512"; do
  set -- $spec; PREC=$1; BLOCKS=$2
  rm -rf gpurun_out/prof_${TAG}_$PREC
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$PREC -- "$PY" bench.py --precision $PREC --fit-precision auto --blocks $BLOCKS --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/prof_${TAG}_${PREC}_bench.json 2> gpurun_out/prof_${TAG}_$PREC.err
  STATS=$(find gpurun_out/prof_${TAG}_$PREC -name '*kernel_stats.csv' | head -1)
  python3 tools/trim_kernel_stats.py "$STATS" profiles/${TAG}_bench_${PREC}_kernel_stats.csv 10
  cp gpurun_out/prof_${TAG}_${PREC}_bench.json profiles/${TAG}_bench_${PREC}_profiled.json
  echo "[profile] $PREC done"
done
