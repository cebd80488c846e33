import ctypes, os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
import torch
from gym_PVDER_amd import _native
from gym_PVDER_amd.vec_env import PVDERVec
from helpers import both_events_spec
dev = torch.device("cuda:0")
L = _native.lib(); buf = (ctypes.c_ulonglong * 16)()
names = ["lanes run", "std deferred", "careful lanes", "careful->deferred", "careful tick trip", "careful step trip", "hybrid steps", "2nd", "3rd", "defer&careful", "tail lanes", "tail WGs", "seq"]
N = 65536
gen = torch.Generator(device=dev); gen.manual_seed(1)
acts = torch.randint(0, 5, (20, N), generator=gen, device=dev, dtype=torch.uint8)
env = PVDERVec(N, device=dev, model_type="model_1", seed=2, events_spec=both_events_spec(), auto_reset=True)
env.reset()
for rep in range(4):
    L.pvder_debug_counters(buf, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.rollout(acts, fused=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    L.pvder_debug_counters(buf, 1)
    f = env.fstate
    print("steps %d..: %.1f us/step" % (20 * rep, dt * 1e6), {n: buf[k] / 20.0 for k, n in enumerate(names)}, "Q_ref range %.3f..%.3f" % (float(f[11].min()), float(f[11].max())), flush=True)
