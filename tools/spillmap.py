#!/usr/bin/env python3
"""Where a kernel spills: per basic block of the gfx950 ISA, instruction / scratch store / scratch load /
MFMA counts (CPU-side; `hipcc --cuda-device-only -S` of the library source).
    python tools/spillmap.py <kernel-name-substring> [asm-file]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, ".scratch", "k.s")
if len(sys.argv) <= 2:
    os.makedirs(os.path.dirname(asm), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-mllvm", "-amdgpu-kernarg-preload-count=12",
                           "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "gym-solarpvder-environment_amd/csrc"),
                           "--cuda-device-only", "-S", "-o", asm,
                           os.path.join(root, "gym-solarpvder-environment_amd/csrc/pvder_kernels.hip")], stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(sys.argv[1]) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".end_amdhsa_kernel") or lines[i].strip() == "s_endpgm" and ".Lfunc_end" in lines[i + 1])
blk = ["entry", 0, 0, 0, 0]
stats = [blk]
for ln in lines[start + 1:end]:
    t = ln.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        blk = [t.split(":")[0], 0, 0, 0, 0]; stats.append(blk); continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    blk[1] += 1
    blk[2] += t.startswith("scratch_store")
    blk[3] += t.startswith("scratch_load")
    blk[4] += "mfma" in t
print("block  instr  sst  sld  mfma")
for b in stats:
    if b[1] > 20 or b[2] or b[3]:
        print("%-10s %5d %4d %4d %4d" % tuple(b))
print("total instr %d, scratch stores %d, loads %d" % (sum(b[1] for b in stats), sum(b[2] for b in stats), sum(b[3] for b in stats)))
