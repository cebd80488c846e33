"""Debug probe (GPU box): step rates of the etdlin path with / without the hand-over guard.
  rocprofv3 --kernel-trace --stats -d gpurun_out/p -- python3 tools/dbg_lin.py 1e9"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gym-solarpvder-environment_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
from gym_PVDER_amd.vec_env import PVDERVec, default_events_spec
dev = torch.device("cuda:0")
spec = default_events_spec(); spec["insolation"]["ENABLE"] = True
guard = float(sys.argv[1]) if len(sys.argv) > 1 else None
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
env = PVDERVec(65536, device=dev, seed=2024, events_spec=spec, auto_reset=True)
if guard is not None:
    env.est_guard = guard
env.reset()
gen = torch.Generator(device=dev); gen.manual_seed(1)
acts = torch.randint(0, 5, (steps, 65536), generator=gen, device=dev, dtype=torch.uint8)
for fused in (False, True, False, True):
    env.rollout(acts[:100], fused=fused); torch.cuda.synchronize()
    t0 = time.perf_counter(); env.rollout(acts, fused=fused); th = time.perf_counter() - t0; torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print("guard %s fused %d: host %.2f us/step, total %.2f us/step" % (guard, fused, th / steps * 1e6, t1 / steps * 1e6))
