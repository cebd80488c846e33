BARGS="--steps 20 --warmup 5 --cpu-seconds 0 --classic-steps 0 --sustained-seconds 0 --fused-seconds 0 --fixed-policy 0 --rk45-steps 0 --single-env-steps 0 --side-legs 0"
for rep in 1 2; do for m in 0 3; do echo -n "PVDER_SAT_HYBRID=$m: "; PVDER_SAT_HYBRID=$m python3 bench.py $BARGS --preroll-steps 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.2f us per step; large batch %.2f us' % (d['ms_per_step']*1e3, d['large_batch']['ms_per_step']*1e3))"; done; done
