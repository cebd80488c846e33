#!/usr/bin/env python
"""Reduce rocprofv3 --pmc passes over tools/a4_hybrid_numbers.py to the SQ counters of step_sat_hybrid_kernel per dispatch
(the dispatches that had an env to advance: the upper half by wave cycles).   python tools/hybrid_sq.py <dir> [<dir> ...]"""
import csv, glob, os, sys
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for row in csv.DictReader(open(f)):
            if "step_sat_hybrid_kernel" in row["Kernel_Name"]:
                per.setdefault(row["Counter_Name"], {})[row["Dispatch_Id"]] = float(row["Counter_Value"])
        if not per:
            continue
        key = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in per else sorted(per)[0]
        ids = sorted(per[key], key=lambda i: per[key][i])
        busy = ids[len(ids) // 2:]
        print(d, "dispatches", len(ids), "busy half", len(busy))
        for name in sorted(per):
            v = [per[name].get(i, 0.0) for i in busy]
            print("  %-28s mean %.4g   max %.4g" % (name, sum(v) / len(v), max(v)))
