set -e
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=12 -DPVDER_DEBUG_COUNTERS -Iinclude -Igym-solarpvder-environment_amd/csrc -o gpurun_out/libdbg.so gym-solarpvder-environment_amd/csrc/pvder_kernels.hip gym-solarpvder-environment_amd/csrc/pvder_agent.hip 2>/dev/null
PVDER_LIB_PATH=$PWD/gpurun_out/libdbg.so python3 - <<'PY'
import ctypes, os, sys
sys.path[:0]=[os.getcwd(), os.getcwd()+"/gym-solarpvder-environment_amd", os.getcwd()+"/tests"]
import torch
from gym_PVDER_amd import _native
from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec
dev=torch.device("cuda:0"); L=_native.lib(); buf=(ctypes.c_ulonglong*16)()
env=PVDERVec(65536,device=dev,seed=2024,auto_reset=False,events_spec=both_events_spec()); env.reset()
a=torch.full((65536,),4,dtype=torch.uint8,device=dev)
for t in range(135):
    L.pvder_debug_counters(buf,1); env.step(a,return_tensors=False); torch.cuda.synchronize(); L.pvder_debug_counters(buf,1)
    if t in (60,61,84,99,100,101,102,103,130): print("step",t,"tail lanes",buf[10],"| lane x sim steps (s>0)",buf[14],"of which lane_fine",buf[13],"| fine passes executed at s>0 (waves)",buf[15])
PY
