#!/bin/bash
# A/B: plain vs non-temporal state loads / result stores (two builds of the library, one session)
set -e
mkdir -p gpurun_out/nt
CS="gym-solarpvder-environment_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DPVDER_PLAIN_MEM -Iinclude -I$CS -o gpurun_out/nt/libplain.so $CS/pvder_kernels.hip
cp gym-solarpvder-environment_amd/gym_PVDER_amd/lib/libpvder_hip.so gpurun_out/nt/libnt.so
for v in plain nt plain nt; do
  cp gpurun_out/nt/lib$v.so gym-solarpvder-environment_amd/gym_PVDER_amd/lib/libpvder_hip.so
  python3 bench.py --cpu-seconds 0 --classic-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
done
cp gpurun_out/nt/libnt.so gym-solarpvder-environment_amd/gym_PVDER_amd/lib/libpvder_hip.so
