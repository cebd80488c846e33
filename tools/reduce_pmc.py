#!/usr/bin/env python
"""Reduce the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --pmc + --kernel-trace
only) and the --kernel-trace --stats pass of one bench command to profiles/<tag>_* summaries.

  python tools/reduce_pmc.py <tag> <fetch_dir> <write_dir> <stats_dir> "<bench args>"
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNEL = os.environ.get("PVDER_PMC_KERNEL", "step_etdlin_kernel_w1<3>")      # PVDERVec.kernel_name() of the bench workload
ENVS = 65536
ALGO_BYTES = 269 * ENVS


def counter_mean(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under " + d
    vals = []
    for row in csv.DictReader(open(files[0])):
        if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == counter:
            vals.append(float(row["Counter_Value"]))
    return files[0], vals


def main():
    tag, fdir, wdir, sdir, bargs = sys.argv[1:6]
    prof = os.path.join(ROOT, "profiles")
    ff, fv = counter_mean(fdir, "FETCH_SIZE")
    wf, wv = counter_mean(wdir, "WRITE_SIZE")
    shutil.copy(ff, os.path.join(prof, tag + "_pmc_fetch_size_counter_collection.csv"))
    shutil.copy(wf, os.path.join(prof, tag + "_pmc_write_size_counter_collection.csv"))
    stats = glob.glob(os.path.join(sdir, "**", "*kernel_stats.csv"), recursive=True)
    assert stats, "no kernel_stats.csv under " + sdir
    shutil.copy(stats[0], os.path.join(prof, tag + "_bench_kernel_stats.csv"))
    avg_ns = calls = None
    for row in csv.DictReader(open(stats[0])):
        if KERNEL in row["Name"]:
            avg_ns, calls = float(row["AverageNs"]), int(row["Calls"])
    # warm launches only (the first ones fault pages in)
    fv, wv = fv[len(fv) // 4:], wv[len(wv) // 4:]
    fkb, wkb = sum(fv) / len(fv), sum(wv) / len(wv)
    rd, wr = fkb * 1024 * 2, wkb * 1024
    out = {
        "kernel": KERNEL, "envs": ENVS,
        "kernel_src_sha": __import__("bench").kernel_source_hash(),   # bench.py reports this traffic only for the same sources
        "command_fetch": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py " + bargs,
        "command_write": "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py " + bargs,
        "command_stats": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py " + bargs,
        "FETCH_SIZE_KB_mean_per_launch": fkb, "launches_fetch": len(fv),
        "WRITE_SIZE_KB_per_launch": wkb, "launches_write": len(wv),
        "correction": "MI355X_MICROARCH.md HBM: counters are in KB (x1024); on gfx950 FETCH_SIZE reports 1/2 of the "
                      "streamed read bytes -> x2; WRITE_SIZE exact",
        "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": ALGO_BYTES,
        "kernel_trace_avg_ns": avg_ns, "kernel_trace_calls": calls,
        "note": "expected from the kernel's own accesses: reads 25 fp32 rows + 3 int32 rows + action + event rows "
                "= ~8.2 MB, writes 25+3+13 rows = 10.75 MB",
    }
    with open(os.path.join(prof, tag + "_pmc_step_kernel_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
