#!/usr/bin/env python3
"""Condense the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_profile_*.sh into profiles/<tag>_hbm_traffic.json.

  python tools/summarize_traffic.py <tag> <profile_dir> [--workload c2]

<profile_dir> holds pmc_fetch/ and pmc_write/ (rocprofv3 --pmc, csv).  Per kernel, averaged over its launches: bytes per
launch = (2*FETCH_SIZE + WRITE_SIZE) KiB -- FETCH_SIZE counts half the bytes of 16-B/lane streams on gfx950
(MI355X_MICROARCH.md, HBM section).  The same figures tools/summarize_prof.py writes when it is given a --stats run."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:80]
        acc[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return {k: acc[k] / len(n[k]) for k in acc}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    workload = "c2"
    if "--workload" in sys.argv:
        workload = sys.argv[sys.argv.index("--workload") + 1]
        args.remove(workload)
    tag, d = args[0], args[1]
    f, w = per_launch(os.path.join(d, "pmc_fetch"), "FETCH_SIZE"), per_launch(os.path.join(d, "pmc_write"), "WRITE_SIZE")
    out = {k: {"FETCH_SIZE_KiB_avg": f[k], "WRITE_SIZE_KiB_avg": w.get(k, 0.0),
               "hbm_bytes_per_launch": (2 * f[k] + w.get(k, 0.0)) * 1024} for k in f}
    json.dump({"workload": workload, "note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB", "kernels": out},
              open(os.path.join(ROOT, "profiles", tag + "_hbm_traffic.json"), "w"), indent=1)
    print("wrote profiles/%s_hbm_traffic.json" % tag)


if __name__ == "__main__":
    main()
