"""Timeline of the last tg_leak_rollout call in a rocprofv3 --kernel-trace CSV (tools/gpu_leak_check.sh): per kernel
start (us after the call's first kernel), duration, queue; then per queue the busy time and the span."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "k_leak_prep" in r["Kernel_Name"]]
i0 = first[-1]
t0 = int(rows[i0]["Start_Timestamp"])
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 9


def name(s):
    m = re.search(r"(k_[a-z_0-9]+)", s)
    return m.group(1) if m else s[:30]


busy, end = {}, 0
for r in rows[i0:]:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = r["Queue_Id"]
    busy[q] = busy.get(q, 0) + b - a
    end = max(end, b)
    if a / 1e3 < lim:
        print("%8.1f %7.1f q%s %s" % (a / 1e3, (b - a) / 1e3, q, name(r["Kernel_Name"])))
print("span %.1f us; busy per queue:" % (end / 1e3), {q: round(v / 1e3, 1) for q, v in busy.items()})
