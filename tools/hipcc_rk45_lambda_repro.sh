#!/bin/bash
# Reproducer attempt of the hipcc miscompile of step_rk45_kernel<double, 3> reported in round 2 (DESIGN.md "RK45": the sim
# time step factored into a helper taking the right-hand side as a lambda returned NaN states for the 23-state instance
# while the 11-state instance of the same template was correct).  -DPVDER_RK45_LAMBDA builds that variant of the shipped
# source (rk45_sim_step<T, NS, F> in csrc/pvder_kernels.hip: the written-out loop, line for line); the RK45 parity tests
# then say whether this compiler still gets it wrong.  GPU box:
#     bash tools/hipcc_rk45_lambda_repro.sh [extra hipcc flags]
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p tools/_ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=12 -fPIC -shared \
  -Iinclude -Igym-solarpvder-environment_amd/csrc -DPVDER_RK45_LAMBDA "$@" -o tools/_ab/librk45l.so \
  gym-solarpvder-environment_amd/csrc/pvder_agent.hip gym-solarpvder-environment_amd/csrc/pvder_kernels.hip || exit 1
PVDER_LIB_PATH=$PWD/tools/_ab/librk45l.so python -m pytest tests -m gpu -q -x -k "rk45 and not etdlin64" 2>&1 | tail -15
