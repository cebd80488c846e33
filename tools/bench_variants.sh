#!/bin/bash
# Bench variants quoted in DESIGN.md section 6 (GPU box, one session so that the figures are comparable).
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/bench_variants.log
: > $out
run() { echo "## $*" >> $out; "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'kernel_avg_ms':d['roofline']['kernel_avg_ms'],'hbm_frac':d['roofline']['frac'],'valu_frac':(d.get('valu_roofline') or {}).get('frac'),'workload':d['config']['workload'][:60]}))" >> $out || echo FAILED >> $out; }
B="python3 bench.py --cpu-seconds 0 --classic-steps 0"
run $B
run $B --micro 16
run $B --graded 0 --micro 8 --micro-coarse 1
run $B --graded 0 --micro 8 --micro-coarse 2
run $B --graded 0 --micro 8 --micro-coarse 0 --steps 1000 --warmup 100
run env PVDER_ETD_VARIANT=full $B --steps 400 --warmup 40
run $B --envs 4096 --n-sub 4
run $B --envs 4096 --n-sub 4 --integrator rk4 --steps 20 --warmup 2 --time-every 1
run $B --integrator rk4 --steps 3 --warmup 1 --time-every 1
run $B --integrator rk45 --steps 2 --warmup 1 --time-every 1
run $B --envs 262144 --steps 1000 --warmup 100
run $B --envs 1048576 --steps 400 --warmup 40
cat $out
