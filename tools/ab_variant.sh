#!/bin/bash
# A/B of a variant build against the shipped library inside ONE gpurun session (boxes differ by +-15 %):
#   bash tools/ab_variant.sh "-DPVDER_PLAIN_MEM" [more flag sets ...]
# Variant libraries are built into gpurun_out/ and selected with PVDER_LIB_PATH -- the in-tree library
# is never touched (the old ab_*.sh copied variants over it and could leave one behind).
set -eo pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
PKG=gym-solarpvder-environment_amd
echo "== shipped library"; python3 tools/dbg_lin.py
i=0
for flags in "$@"; do
  i=$((i+1)); lib=$PWD/gpurun_out/libpvder_variant_$i.so
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=12 $flags -Iinclude -I$PKG/csrc -o $lib $PKG/csrc/pvder_kernels.hip $PKG/csrc/pvder_agent.hip 2>/dev/null
  echo "== variant $i: $flags"; PVDER_LIB_PATH=$lib python3 tools/dbg_lin.py
done
echo "== shipped library again"; python3 tools/dbg_lin.py
