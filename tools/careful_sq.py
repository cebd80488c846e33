#!/usr/bin/env python3
"""GPU box: SQ counters of the step kernel along the reference's 'dec' episode (action 1 held for 160 env steps, 65 536 envs):
what a wave executes per env step on the standard path (steps 40..79) and once the whole batch sits at the current limit
(steps 140..155: the careful pass) -- instruction counts, busy / wait cycles.

    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES \\
        --kernel-trace --output-format csv -d gpurun_out/csq/p1 -- python3 tools/careful_sq.py run
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_IFETCH SQ_INSTS_BRANCH \\
        --kernel-trace --output-format csv -d gpurun_out/csq/p2 -- python3 tools/careful_sq.py run
    python3 tools/careful_sq.py reduce gpurun_out/csq/p1 gpurun_out/csq/p2"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd"), os.path.join(ROOT, "tests")]
if sys.argv[1] == "run":
    import torch
    from gym_PVDER_amd.vec_env import PVDERVec
    from helpers import both_events_spec
    dev = torch.device("cuda:0")
    act = int(os.environ.get("ACT", "1"))
    env = PVDERVec(65536, device=dev, seed=2024, auto_reset=False, events_spec=both_events_spec())
    env.reset()
    a = torch.full((65536,), act, dtype=torch.uint8, device=dev)
    for t in range(160):
        env.step(a, return_tensors=False)
    torch.cuda.synchronize()
    sys.exit(0)
tot = {}
for d in sys.argv[2:]:
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = {}
    for r in csv.DictReader(open(f)):
        if "step_etdlin_kernel" in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(rows)
    assert len(ids) == 160, len(ids)
    for name, lo, hi in (("standard path (steps 40..79)", 40, 80), ("at the current limit (steps 140..155)", 140, 156)):
        for k in rows[ids[lo]]:
            tot.setdefault(name, {})[k] = sum(rows[ids[i]][k] for i in range(lo, hi)) / (hi - lo)
for name, c in tot.items():
    w = c.get("SQ_WAVES", 1024.0)
    print(name + ": per launch, per wave (%d waves)" % w)
    for k in sorted(c):
        print("   %-28s %14.0f %12.1f" % (k, c[k], c[k] / w))
