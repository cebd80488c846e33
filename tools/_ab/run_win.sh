BARGS="--steps 20 --warmup 5 --preroll-steps 0 --cpu-seconds 0 --classic-steps 0 --sustained-seconds 0 --fused-seconds 0 --fixed-policy 0 --rk45-steps 0 --large-batch 0 --single-env-steps 0 --side-legs 0"
mkdir -p gpurun_out
for rep in 1 2 3; do for g in 1 0; do
python3 bench.py $BARGS --graph-window $g 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('graph-window $g: %.2f us per step, %.3e, host enqueue %.2f us, following %s' % (d['ms_per_step']*1e3, d['value'], d['host_enqueue_ms_per_step']*1e3, d.get('duty_cycle_limit_kernel_following')))"
done; done | tee gpurun_out/win_ab.txt
