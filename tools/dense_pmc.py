#!/usr/bin/env python3
"""SQ counters of the DENSE k_stageV launch (every tile computed: hjb_active = 0), GPU box:
    bash tools/dense_pmc.sh [workload]        # two rocprofv3 --pmc passes over this script, then the per-wavefront means
"""
import os, sys, glob, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "--read":
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_stageV" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {c: sum(v) / len(v) for c, v in acc.items()}
    for c in sorted(m):
        print("   %-26s n=%d mean=%.5g" % (c, len(acc[c]), m[c]))
    w = m.get("SQ_WAVES", 0)
    if w:
        print("per wavefront: VALU %.0f  SALU %.0f  SMEM %.0f  LDS %.0f  VMEM %.0f" % tuple(
            m.get(k, 0) / w for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM")))
    sys.exit(0)

import bench, torch
wl = sys.argv[1] if len(sys.argv) > 1 else "metro_station"
sim, cfg = bench.build_sim(wl)
eng = sim._engine
eng.set_tuning("hjb_active", 0)
for k, v in (a.split("=") for a in sys.argv[2:]):
    eng.set_tuning(k, int(v))
for rep in range(2):
    st, _ = eng.solve(0, sim.T, sim._nt0, m=None)
torch.cuda.synchronize()
print("dense solve %.3f ms, attempts %d" % (1e3 * st["seconds"], st["n_attempt"]))
