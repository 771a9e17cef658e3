#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into small tracked files under profiles/.

  python tools/summarize_prof.py <round-tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir>] [--workload c2]

Writes profiles/<tag>_kernel_stats.csv (kernel names shortened) and, when PMC passes are given,
profiles/<tag>_hbm_traffic.json with per-kernel HBM bytes per launch =
(2*FETCH_SIZE + WRITE_SIZE) KiB -- FETCH_SIZE reads half the bytes of 16-B/lane streams on gfx950
(MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:80]


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit("no file matches " + pattern)
    return files[0]


def stats_rows(stats_dir):
    """Per-kernel statistics from rocprofv3 --kernel-trace --stats output: the csv when it was asked for
    (--output-format csv), else computed from the kernels view of the default sqlite file (*_results.db)."""
    files = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)
    if files:
        return list(csv.DictReader(open(files[0])))
    import sqlite3
    db = sqlite3.connect(one(os.path.join(stats_dir, "**", "*_results.db")))
    q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
         "group by name order by sum(duration) desc")
    got = list(db.execute(q))
    total = float(sum(r[2] for r in got)) or 1.0
    return [{"Name": r[0], "Calls": r[1], "TotalDurationNs": r[2], "AverageNs": "%.1f" % r[3],
             "Percentage": "%.2f" % (100.0 * r[2] / total), "MinNs": r[4], "MaxNs": r[5]} for r in got]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    workload = "c2"
    if "--workload" in sys.argv:
        workload = sys.argv[sys.argv.index("--workload") + 1]
        args.remove(workload)
    tag, stats_dir = args[0], args[1]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    rows = stats_rows(stats_dir)
    with open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"), "w") as f:
        wr = csv.writer(f)
        wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            wr.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                         r["MinNs"], r["MaxNs"]])
    if len(args) >= 4:
        tot = {}
        for key, d in (("FETCH_SIZE", args[2]), ("WRITE_SIZE", args[3])):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(one(os.path.join(d, "**", "*counter_collection.csv")))):
                if r["Counter_Name"] == key:
                    acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                tot.setdefault(k, {})[key + "_KiB_avg"] = sum(v) / len(v)
        for k, v in tot.items():
            v["hbm_bytes_per_launch"] = (2 * v.get("FETCH_SIZE_KiB_avg", 0.0) + v.get("WRITE_SIZE_KiB_avg", 0.0)) * 1024
        json.dump({"workload": workload, "note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB",
                   "kernels": tot}, open(os.path.join(ROOT, "profiles", tag + "_hbm_traffic.json"), "w"), indent=1)
    print("wrote profiles/%s_*" % tag)


if __name__ == "__main__":
    main()
