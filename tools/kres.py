"""Kernel resource table of one csrc/*.hip (registers, spills, occupancy): python3 tools/kres.py tip_unet.hip [filter]"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tissue_image_processing_amd import build as b  # noqa: E402

src = os.path.join(b.CSRC, sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run([b._hipcc()] + b.CFLAGS + ["-c", src, "-o", "/tmp/_kres.o", "-Rpass-analysis=kernel-resource-usage"],
                     capture_output=True, text=True).stderr
NAME = re.compile(r"remark:\s+Function Name: (\S+)")
FIELD = re.compile(r"remark:\s+([A-Za-z][A-Za-z /\[\]]*?): (\d+) \[-Rpass")
cur, rows = None, {}
for line in out.splitlines():
    m = NAME.search(line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = FIELD.search(line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    print("%-62s VGPR %3d AGPR %3d SGPR %3d spillV %d spillS %d occ %d scratch %d"
          % (name[:62], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("TotalSGPRs", -1), v.get("VGPRs Spill", -1),
             v.get("SGPRs Spill", -1), v.get("Occupancy [waves/SIMD]", -1), v.get("ScratchSize [bytes/lane]", -1)))
