#!/usr/bin/env python3
"""CPU: how hipcc interleaves matrix-core and vector instructions in a kernel of the library -- run-length pattern of the
ISA (M = v_mfma, V = other vector ALU, O = memory / LDS, S = scalar, W = s_waitcnt / s_nop) from a fresh device-only compile.
    python tools/isa_pattern.py [kernel-name-substring] [extra hipcc flags ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "step_etdlin_rollout_kernelILi3"
wd = os.path.join(ROOT, ".scratch", "co"); os.makedirs(wd, exist_ok=True)
csrc = os.path.join(ROOT, "gym-solarpvder-environment_amd", "csrc")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                       "-mllvm", "-amdgpu-kernarg-preload-count=12", "-I" + os.path.join(ROOT, "include"), "-I" + csrc] + sys.argv[2:] +
                      ["-o", os.path.join(wd, "dev.o"), os.path.join(csrc, "pvder_kernels.hip")], stderr=subprocess.DEVNULL)
llvm = "/opt/rocm/lib/llvm/bin/"
subprocess.check_call([llvm + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + os.path.join(wd, "dev.o"),
                       "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + os.path.join(wd, "dev.co")])
asm = subprocess.run([llvm + "llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(wd, "dev.co")], capture_output=True, text=True).stdout.split("\n")
start = end = None
for i, l in enumerate(asm):
    if re.match(r"^[0-9a-f]+ <.*%s.*>:" % re.escape(name), l) and start is None:
        start = i
    elif start is not None and re.match(r"^[0-9a-f]+ <", l):
        end = i; break
seq = []
for l in asm[start + 1:end]:
    m = re.match(r"\s+(\S+)", l)
    if not m: continue
    op = m.group(1)
    seq.append("M" if op.startswith("v_mfma") else "W" if op.startswith(("s_waitcnt", "s_nop")) else "V" if op.startswith("v_") else "S" if op.startswith("s_") else "O")
runs, cur, n = [], seq[0], 0
for k in seq:
    if k == cur: n += 1
    else: runs.append((cur, n)); cur, n = k, 1
runs.append((cur, n))
print("%s: %d instructions, %d MFMA, %d VALU; longest pure-VALU runs: %s; longest pure-MFMA runs: %s" % (
    name, len(seq), seq.count("M"), seq.count("V"), sorted([n for k, n in runs if k == "V"])[-8:], sorted([n for k, n in runs if k == "M"])[-8:]))
if os.environ.get("FULL"):
    print(" ".join("%s%d" % r for r in runs))
