#!/usr/bin/env python3
"""Registers / scratch / LDS / occupancy of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py gcfm.hip [name-filter]
"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "optimal_crowds_amd", "csrc")


def main():
    src = os.path.join(CSRC, sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    r = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-c", src,
                        "-o", "/tmp/_kres.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = None
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    for c in rows:
        name = subprocess.run(["c++filt", c["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name)
        if flt and flt not in name:
            continue
        print("%-52s VGPR %3d AGPR %3d SGPR %3d scratch %4d LDS %6d occ %d" % (
            name[:52], c.get("VGPRs", -1), c.get("AGPRs", -1), c.get("TotalSGPRs", c.get("SGPRs", -1)),
            c.get("ScratchSize", -1), c.get("LDS Size", -1), c.get("Occupancy", -1)) +
              (" SPILL s%d v%d" % (c.get("SGPRs Spill", 0), c.get("VGPRs Spill", 0)) if c.get("SGPRs Spill", 0) or c.get("VGPRs Spill", 0) else ""))


if __name__ == "__main__":
    main()
