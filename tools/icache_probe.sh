#!/bin/bash
# GPU box: instruction-cache / fetch counters of the step kernel per launch while every STRIDE-th env sits at the current
# limit (tools/sparse_careful_envs.py) -- is the careful pass's "many workgroups" slowdown an instruction-fetch effect?
#   bash tools/icache_probe.sh
set -eo pipefail
export TMPDIR=/tmp
out=gpurun_out/icache
rm -rf $out; mkdir -p $out
rocprofv3 --list-avail > $out/avail.txt 2>&1 || true
grep -i -E "ICACHE|IFETCH|SQC_INST|INST_LEVEL|SQ_WAIT_IFETCH" $out/avail.txt | head -60 > $out/avail_ifetch.txt || true
echo "[icache] pass A"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pA -- python3 tools/sparse_careful_envs.py > $out/pA.log 2>&1
echo "[icache] pass B"
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/pB -- python3 tools/sparse_careful_envs.py > $out/pB.log 2>&1
python3 tools/icache_reduce.py $out/pA $out/pB | tee $out/reduce.txt
