mkdir -p gpurun_out
BARGS="--steps 20 --warmup 5 --preroll-steps 0 --cpu-seconds 0 --classic-steps 0 --sustained-seconds 0 --fused-seconds 0 --fixed-policy 0 --rk45-steps 0 --single-env-steps 0 --side-legs 0"
{
for v in block quarter block quarter; do
  echo "== $v"
  if [ $v = block ]; then export PVDER_LIB_PATH=$PWD/tools/_ab/libpvder_block.so; else unset PVDER_LIB_PATH; fi
  for n in 1 64 4096 65536; do timeout -k 10 200 python3 tools/hybrid_batch_rate.py $n 8 2>&1 | grep "^N"; done
  PVDER_SAT_HYBRID=3 python3 bench.py $BARGS 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('followed launch that finds nothing: %.2f us per step; large batch %.2f us' % (d['ms_per_step']*1e3, d['large_batch']['ms_per_step']*1e3))"
done
unset PVDER_LIB_PATH
timeout -k 10 900 python3 -m pytest tests/test_gpu_etdlin.py tests/test_gpu_parity.py -m gpu -x -q -k "duty_cycle or hybrid or fixed_action_ramps or off_range or graph or saturat or integrate" 2>&1 | tail -3
} | tee gpurun_out/quarter_ab.txt
