#!/usr/bin/env python3
"""Kernel resource table (VGPRs / SGPRs / scratch / occupancy / LDS) of one csrc/*.hip translation unit, from hipcc's
-Rpass-analysis=kernel-resource-usage.  Usage: scripts/kres.py s360_forward.hip [extra hipcc flags]"""
import re
import subprocess
import sys
from pathlib import Path

csrc = Path(__file__).resolve().parent.parent / "splatter360_amd" / "csrc"
src, extra = sys.argv[1], sys.argv[2:]
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
if src == "s360_backward_em.hip":
    flags.append("-fno-slp-vectorize")
r = subprocess.run(["/opt/rocm/bin/hipcc", *flags, *extra, "-Rpass-analysis=kernel-resource-usage", "-c", str(csrc / src), "-o", "/tmp/kres.o"],
                   capture_output=True, text=True)
rows, cur = [], None
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r"remark: [^:]*:\d+:\d+: +Name: (\S+)", line) or re.search(r":\d+:\d+: +Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r" (TotalSGPRs|VGPRs|AGPRs|ScratchSize|Occupancy|LDS Size|VGPRs Spill)[^:]*: (\d+)", line)
    if m and cur is not None:
        cur.setdefault(m.group(1), int(m.group(2)))
if r.returncode:
    print(r.stderr[-3000:])
for q in rows:
    name = subprocess.run(["c++filt", q["name"]], capture_output=True, text=True).stdout.split("(")[0].strip()
    print("%-58s vgpr %4d sgpr %4d scratch %4d occ %2d lds %6d" % (name[:58], q.get("VGPRs", -1), q.get("TotalSGPRs", -1), q.get("ScratchSize", -1),
                                                                  q.get("Occupancy", -1), q.get("LDS Size", -1)))
