#!/bin/bash
# Register / scratch / LDS use of every kernel of the library (CPU-side: hipcc cross-compiles gfx950).
#   tools/kres.sh [extra hipcc flags]
cd "$(dirname "$0")/.."
mkdir -p .scratch
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=12 -Iinclude \
  -Igym-solarpvder-environment_amd/csrc -Rpass-analysis=kernel-resource-usage "$@" -o .scratch/kres.so \
  gym-solarpvder-environment_amd/csrc/pvder_kernels.hip 2>&1 |
  python3 -c '
import re, sys
cur = None
rows = []
for line in sys.stdin:
    if "error" in line or "warning" in line:
        sys.stderr.write(line)
    m = re.search(r"remark: [^ ]* *Function Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+([\w][\w \[\]/]*?): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
import subprocess
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    print("%-44s VGPR %3d AGPR %3d SGPR %3d scratch %5d spill v%d s%d LDS %6d occ %d" % (n[:44], r.get("VGPRs", -1), r.get("AGPRs", -1),
          r.get("TotalSGPRs", -1), r.get("ScratchSize [bytes/lane]", -1), r.get("VGPRs Spill", -1), r.get("SGPRs Spill", -1),
          r.get("LDS Size [bytes/block]", -1), r.get("Occupancy [waves/SIMD]", -1)))
'
