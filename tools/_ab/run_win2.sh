mkdir -p gpurun_out
P='import json,sys
d=json.loads(sys.stdin.read()); print("%.2f us per step, %.3e, following %s, preroll %s" % (d["ms_per_step"]*1e3, d["value"], d.get("duty_cycle_limit_kernel_following"), d["config"].get("preroll_steps")))'
BARGS="--cpu-seconds 0 --classic-steps 0 --sustained-seconds 0 --fused-seconds 0 --fixed-policy 0 --rk45-steps 0 --large-batch 0 --single-env-steps 0 --side-legs 0"
{
for rep in 1 2 3; do echo -n "driver flags + legs off: "; python3 bench.py --steps 20 --warmup 5 $BARGS 2>/dev/null | tail -1 | python3 -c "$P"; done
for pr in 0 5000 10000 15000 19000 21000 25000; do echo -n "preroll $pr: "; python3 bench.py --steps 20 --warmup 5 --preroll-steps $pr $BARGS 2>/dev/null | tail -1 | python3 -c "$P"; done
echo -n "driver command as is: "; python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "$P"
} | tee gpurun_out/win_ab2.txt
