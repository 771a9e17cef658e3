#!/usr/bin/env python3
"""Condense a rocprofv3 --pmc pass (csv output) into profiles/<tag>_clock_mfma.json.

  python tools/summarize_pmc.py <tag> <pmc_dir> [--workload c3s]

Per kernel (averaged over its launches): the raw counters, the launch duration under the counter pass,
effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration, and the share of SIMD cycles the matrix pipes were busy =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")[:60]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    workload = "c2"
    if "--workload" in sys.argv:
        workload = sys.argv[sys.argv.index("--workload") + 1]
        args.remove(workload)
    tag, d = args[0], args[1]
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if (r["Dispatch_Id"]) not in seen:
            seen.add(r["Dispatch_Id"])
            acc[k]["dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out = {}
    for k, v in acc.items():
        e = {c: sum(x) / len(x) for c, x in v.items()}
        e["launches"] = len(v["dur_ns"])
        if "GRBM_GUI_ACTIVE" in e and e["dur_ns"] > 0:
            e["effective_clock_GHz"] = e["GRBM_GUI_ACTIVE"] / 8.0 / e["dur_ns"]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e["GRBM_GUI_ACTIVE"] > 0:
                e["mfma_busy_frac_of_simd_cycles"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0)
        out[k] = e
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["dur_ns"] * kv[1]["launches"]))
    json.dump({"workload": workload,
               "note": "effective clock = GRBM_GUI_ACTIVE/8/duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES/(1024 SIMDs x "
                       "cycles); per-launch averages, durations under the counter pass",
               "kernels": out}, open(os.path.join(ROOT, "profiles", tag + "_clock_mfma.json"), "w"), indent=1)
    print("wrote profiles/%s_clock_mfma.json" % tag)


if __name__ == "__main__":
    main()
