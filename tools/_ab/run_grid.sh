mkdir -p gpurun_out
BARGS="--steps 20 --warmup 5 --preroll-steps 0 --cpu-seconds 0 --classic-steps 0 --sustained-seconds 0 --fused-seconds 0 --fixed-policy 0 --rk45-steps 0 --large-batch 0 --single-env-steps 0 --side-legs 0"
{
for g in 512 1024 2048 512 1024; do
  echo "== grid $g"
  export PVDER_LIB_PATH=$PWD/tools/_ab/libpvder_grid$g.so
  for n in 16384 65536; do timeout -k 10 200 python3 tools/hybrid_batch_rate.py $n 8 2>&1 | grep "^N"; done
  PVDER_SAT_HYBRID=3 python3 bench.py $BARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('followed launch that finds nothing: %.2f us per step' % (d['ms_per_step']*1e3))"
done
} | tee gpurun_out/grid_ab.txt
