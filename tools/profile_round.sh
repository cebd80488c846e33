#!/bin/bash
# The round's profile set (GPU box): bench line, rocprofv3 kernel trace + stats, the two PMC passes.
#   bash tools/profile_round.sh r01
set -eo pipefail
tag=${1:-r01}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
BARGS="--steps 400 --warmup 100 --preroll-steps 0 --cpu-seconds 0 --classic-steps 0 --sustained-seconds 0 --fused-seconds 0.1 --fixed-policy 0 --rk45-steps 0 --large-batch 0 --single-env-steps 0 --side-legs 0"
export PVDER_PMC_KERNEL="${PVDER_PMC_KERNEL:-step_etdlin_kernel_w1<3>}"
echo "[profile] kernel trace + stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py $BARGS > $out/stats.log 2>&1
echo "[profile] pmc FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 bench.py $BARGS > $out/fetch.log 2>&1
echo "[profile] pmc WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 bench.py $BARGS > $out/write.log 2>&1
python3 tools/reduce_pmc.py $tag $out/fetch $out/write $out/stats "$BARGS" > $out/reduce.log 2>&1
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_* gpurun_out/profiles_$tag/
if [ -n "$PROFILE_SKIP_BENCH" ]; then cat $out/reduce.log | head -30; exit 0; fi
echo "[profile] full bench"; python3 bench.py > $out/bench_full.log 2>&1
tail -1 $out/bench_full.log > gpurun_out/profiles_$tag/${tag}_bench.json.log
python3 bench.py --steps 20 --warmup 5 > $out/bench_20.log 2>&1
tail -1 $out/bench_20.log > gpurun_out/profiles_$tag/${tag}_bench_steps20.json.log
cat $out/reduce.log | head -30
tail -1 $out/bench_full.log
