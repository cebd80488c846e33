#!/usr/bin/env python3
"""Reduce tools/icache_probe.sh: counters per launch of step_etdlin_kernel, averaged over each block of 120 launches
(= one stride setting of tools/sparse_careful_envs.py: 3 rollouts x 40 steps)."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print(d, "no counter file"); continue
    per = collections.OrderedDict()
    for row in csv.DictReader(open(files[0])):
        if "step_etdlin_kernel" not in row["Kernel_Name"]:
            continue
        per.setdefault(int(row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
    ids = sorted(per)
    names = sorted({k for v in per.values() for k in v})
    print(d, "launches", len(ids))
    print("%-8s" % "block", " ".join("%16s" % n[-16:] for n in names))
    for b in range(0, len(ids), 120):
        blk = ids[b:b + 120][40:]            # (skip the first rollout of the setting: it moves the envs to the limit)
        if not blk: continue
        print("%-8d" % (b // 120), " ".join("%16.0f" % (sum(per[i].get(n, 0.0) for i in blk) / len(blk)) for n in names))
