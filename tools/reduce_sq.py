#!/usr/bin/env python
"""Reduce the SQ counter passes of tools/profile_sq.sh to profiles/<tag>_pmc_sq_summary.json (per launch and per wave of
the step kernel; warm launches only) and keep the raw counter rows of that kernel as <tag>_pmc_sq_pass{1,2}.csv.

  python tools/reduce_sq.py <tag> <pass1_dir> <pass2_dir> "<bench args>"
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNEL = os.environ.get("PVDER_PMC_KERNEL", "step_etdlin_kernel_w1<3>")
ENVS = int(os.environ.get("PVDER_PMC_ENVS", "65536"))


def collect(d, keep_path):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under " + d
    per = {}
    meta = {}
    with open(files[0]) as f, open(keep_path, "w", newline="") as g:
        rd = csv.DictReader(f)
        wr = csv.DictWriter(g, fieldnames=rd.fieldnames)
        wr.writeheader()
        for row in rd:
            if KERNEL in row["Kernel_Name"]:
                per.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                meta = {k: row[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                           "Workgroup_Size", "Grid_Size")}
                row["Kernel_Name"] = KERNEL            # (the mangled template name is 2 KB per row)
                wr.writerow(row)
    return {k: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for k, v in per.items()}, meta


def main():
    tag, d1, d2, bargs = sys.argv[1:5]
    prof = os.path.join(ROOT, "profiles")
    c1, meta = collect(d1, os.path.join(prof, tag + "_pmc_sq_pass1.csv"))
    c2, _ = collect(d2, os.path.join(prof, tag + "_pmc_sq_pass2.csv"))
    per_launch = dict(c1)
    per_launch.update(c2)
    waves = ENVS // 64
    per_wave = {k: v / waves for k, v in per_launch.items()}
    wc = per_launch.get("SQ_WAVE_CYCLES", 0.0)
    share = {}
    if wc:
        share = {"active_inst": per_launch["SQ_ACTIVE_INST_ANY"] / wc,
                 "wait_any (parked: s_waitcnt, barrier)": per_launch["SQ_WAIT_ANY"] / wc,
                 "wait_inst_any (issue stall: MFMA read-after-write, pipe)": per_launch["SQ_WAIT_INST_ANY"] / wc}
    out = {"kernel": KERNEL, "envs": ENVS, "waves_per_launch": waves, "kernel_resources": meta,
           "kernel_src_sha": __import__("bench").kernel_source_hash(),
           "per_launch": per_launch, "per_wave": per_wave, "share_of_wave_cycles": share,
           "mfma_busy_cycles_per_wave": per_wave.get("SQ_VALU_MFMA_BUSY_CYCLES"),
           "mfma_count_per_wave_at_8_cycles": per_wave.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 8.0,
           "command": "rocprofv3 --pmc <counters of one pass> --kernel-trace --output-format csv -- python3 bench.py " + bargs,
           "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); "
                   "SQ_VALU_MFMA_BUSY_CYCLES counts cycles; two passes (tools/profile_sq.sh), warm launches only"}
    with open(os.path.join(prof, tag + "_pmc_sq_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
