#!/bin/bash
# A/B of one library build under two settings of an environment knob, interleaved on ONE device:
#   tools/ab_knob.sh ESN_S16 0 1      (3 rounds of bench.py per value; AB_ARGS overrides the bench arguments)
cd "$(dirname "$0")/.."
KNOB=$1; shift
ARGS=${AB_ARGS:-"--no-cpu-baseline --no-extra --steps 10 --warmup 3"}
for round in 1 2 3; do
  for v in "$@"; do
    env $KNOB=$v python3 bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $round $KNOB=%-3s value %.4g  ms/step %.2f  predict_ms %.3f  frac %.4f  ber %.6f' % ('$v', d['value'], d['ms_per_step'], d['predict_kernel_ms'], d['roofline']['frac'], d['ber']))"
  done
done
