#!/bin/bash
# A/B builds of the library with extra hipcc flag sets (debug tool):  bash tools/ab_flags.sh "name1:flags..." "name2:flags..."
set -e
mkdir -p gpurun_out/ab
CS="gym-solarpvder-environment_amd/csrc"
LIB=gym-solarpvder-environment_amd/gym_PVDER_amd/lib/libpvder_hip.so
cp $LIB gpurun_out/ab/base.so
names="base"
for spec in "$@"; do
  n="${spec%%:*}"; f="${spec#*:}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $f -Iinclude -I$CS -o gpurun_out/ab/$n.so $CS/pvder_kernels.hip
  names="$names $n"
done
for rep in 1 2; do for v in $names; do
  cp gpurun_out/ab/$v.so $LIB
  python3 bench.py --cpu-seconds 0 --classic-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done; done
cp gpurun_out/ab/base.so $LIB
