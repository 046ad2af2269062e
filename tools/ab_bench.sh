#!/bin/bash
# A/B of library variants on ONE device in one call: tools/ab_bench.sh base variant1 variant2 ...
# ("base" = libesn_hip.so, other names = libesn_hip_<name>.so built by build.py --variant); 3 interleaved rounds.
cd "$(dirname "$0")/.."
ARGS=${AB_ARGS:-"--no-cpu-baseline --no-extra --steps 10 --warmup 3"}
for round in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = base ]; then lib=esn_ofdm_mimo_amd/libesn_hip.so; else lib=esn_ofdm_mimo_amd/libesn_hip_$v.so; fi
    ESN_HIP_LIB=$PWD/$lib python3 bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $round %-10s value %.4g  ms/step %.2f  predict_ms %.3f  frac %.4f  ber %.6f' % ('$v', d['value'], d['ms_per_step'], d['predict_kernel_ms'], d['roofline']['frac'], d['ber']))"
  done
done
