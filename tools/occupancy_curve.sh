#!/bin/bash
# microseconds per batched step of the default step path against the number of envs (one launch per env step; 65 536 envs = one wave per SIMD):
# what a second, third ... wave per SIMD costs -- the argument about VERDICT r5 #5's option A.
#   bash tools/occupancy_curve.sh
set -eo pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
BARGS="--steps 2000 --warmup 200 --preroll-steps 0 --cpu-seconds 0 --classic-steps 0 --sustained-seconds 0 --fused-seconds 0 --fixed-policy 0 --rk45-steps 0 --large-batch 0 --single-env-steps 0 --side-legs 0"
for n in 16384 32768 65536 98304 131072 196608 262144 524288; do
  python3 bench.py --envs $n $BARGS 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('envs %7d: %.2f us per batched step, %.3e env-steps/s, kernel %s, frac %.3f' % ($n, d['ms_per_step'] * 1e3, d['value'], d['roofline']['kernel'], d['roofline']['frac']))"
done | tee gpurun_out/occupancy_curve.txt
