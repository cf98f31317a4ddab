#!/usr/bin/env python3
"""Print VGPR/AGPR/SGPR/occupancy/spill per kernel for a .hip file (hipcc -Rpass-analysis)."""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    extra = sys.argv[2:]
    cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"] + extra
    out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key in ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]",
                    "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and cur is not None and key not in cur:
                cur[key] = int(m.group(1))
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void tnf::", "")
        print("%-58s vgpr %3d agpr %3d sgpr %3d occ %d spill %d scratch %d lds %d" % (
            name[:58], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1),
            r.get("Occupancy [waves/SIMD]", -1), r.get("VGPRs Spill", -1),
            r.get("ScratchSize [bytes/lane]", -1), r.get("LDS Size [bytes/block]", -1)))


if __name__ == "__main__":
    main()
