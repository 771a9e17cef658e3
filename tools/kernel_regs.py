#!/usr/bin/env python3
"""VGPRs / spills / LDS per kernel of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_regs.py textgan-pytorch_amd/csrc/tg_dis.hip [-DTG_DIAG=1 ...]
"""
import re
import subprocess
import sys

src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|"
                  r"Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
demangle = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                          capture_output=True, text=True).stdout.splitlines()
print(f"{'VGPR':>5} {'AGPR':>5} {'spill':>6} {'scratch':>8} {'occ':>4} {'LDS':>7}  kernel")
for r, d in zip(rows, demangle):
    d = re.sub(r"^void ", "", d.replace("(anonymous namespace)::", ""))
    d = re.sub(r"\(.*$", "", d)
    print(f"{r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('VGPRs Spill', '?'):>6} "
          f"{r.get('ScratchSize', '?'):>8} {r.get('Occupancy', '?'):>4} {r.get('LDS Size', '?'):>7}  {d}")
