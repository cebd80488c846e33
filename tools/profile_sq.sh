#!/bin/bash
# SQ counters of the step kernel (GPU box): two --pmc passes of their own (8 SQ slots per pass; no --stats, no
# other trace domain), reduced by tools/reduce_sq.py -> profiles/<tag>_pmc_sq_summary.json.
#   bash tools/profile_sq.sh r04a
set -eo pipefail
tag=${1:-r04}
export TMPDIR=/tmp
out=gpurun_out/sq_$tag
rm -rf $out; mkdir -p $out
BARGS="--steps 200 --warmup 50 --preroll-steps 0 --cpu-seconds 0 --classic-steps 0 --sustained-seconds 0 --fused-seconds 0 --fixed-policy 0 --rk45-steps 0 --large-batch 0 --single-env-steps 0 --side-legs 0"
export PVDER_PMC_KERNEL="${PVDER_PMC_KERNEL:-step_etdlin_kernel_w1<3>}"
echo "[sq] pass 1: wave cycles"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $out/p1 -- python3 bench.py $BARGS > $out/p1.log 2>&1
echo "[sq] pass 2: instruction mix + matrix pipe"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $out/p2 -- python3 bench.py $BARGS > $out/p2.log 2>&1
python3 tools/reduce_sq.py $tag $out/p1 $out/p2 "$BARGS" | tee $out/reduce.log
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_pmc_sq* gpurun_out/profiles_$tag/
