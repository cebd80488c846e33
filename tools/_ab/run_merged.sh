set -e
mkdir -p gpurun_out
{
for n in 1 64 4096; do timeout -k 10 200 python3 tools/hybrid_batch_rate.py $n 8 2>&1 | grep "^N"; done
timeout -k 10 900 python3 -m pytest tests/test_gpu_etdlin.py tests/test_gpu_parity.py -m gpu -x -q -k "duty_cycle or hybrid or fixed_action_ramps or off_range or graph or saturat or integrate" 2>&1 | tail -3
BARGS="--steps 20 --warmup 5 --cpu-seconds 0 --classic-steps 0 --sustained-seconds 0 --fused-seconds 0 --fixed-policy 0 --rk45-steps 0 --single-env-steps 0 --side-legs 0"
for m in 0 3; do echo -n "PVDER_SAT_HYBRID=$m (3: the kernel behind every launch): "; PVDER_SAT_HYBRID=$m python3 bench.py $BARGS --preroll-steps 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.2f us per step; large batch %.2f us' % (d['ms_per_step']*1e3, d['large_batch']['ms_per_step']*1e3))"; done
} | tee gpurun_out/merged.txt
