#!/bin/bash
set -o pipefail
out=gpurun_out/r03_bench_variants.log
: > $out
run() { echo "## $*" >> $out; "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=d.get('fused_rollout') or {}
print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'kernel_avg_ms':d['roofline']['kernel_avg_ms'],'hbm_frac':d['roofline']['frac'],'valu_frac':(d.get('valu_roofline') or {}).get('frac'),'fused':f.get('value'),'workload':d['config']['workload'][:60]}))" >> $out || echo FAILED >> $out; }
B="python3 bench.py --cpu-seconds 0 --classic-steps 0 --sustained-seconds 0 --fixed-policy 0 --rk45-steps 0 --large-batch 0 --single-env-steps 0"
run $B --steps 2000 --warmup 200
run $B --integrator etdrk4 --steps 1000 --warmup 100
run env PVDER_ETD_VARIANT=full $B --steps 400 --warmup 40 --fused-seconds 0 --preroll-steps 0
run $B --envs 4096 --n-sub 4 --steps 2000 --warmup 200
run $B --envs 4096 --n-sub 4 --integrator rk4 --steps 20 --warmup 2 --time-every 1
run $B --envs 262144 --steps 1000 --warmup 100 --preroll-steps 2000
run $B --envs 1048576 --steps 400 --warmup 40 --preroll-steps 500 --fused-steps 40
cat $out
