#!/usr/bin/env python3
"""The kernels of ONE HJB solve in time order (GPU box):
    rocprofv3 --kernel-trace -d gpurun_out/trace -o solve -- python3 tools/solve_trace.py [workload]     # writes the trace
    python3 tools/solve_trace.py --read gpurun_out/trace                                                  # prints the last solve
"""
import os, sys, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "--read":
    files = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the last solve = the kernels after the last gap of more than 2 ms
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > 2_000_000:
            cut = i
    sol = rows[cut:]
    t0 = sol[0][0]
    print("%d kernels, %.3f ms from the first start to the last end" % (len(sol), (sol[-1][1] - t0) * 1e-6))
    prev_end = t0
    agg = {}
    for k, (s, e, n) in enumerate(sol):
        short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
        a = agg.setdefault(short, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += (e - s) * 1e-3
        a[2] += max(0, s - prev_end) * 1e-3
        if k < 44 or k >= len(sol) - 8:
            print("%4d  start %9.2f us  gap %6.2f  dur %7.2f  %s" % (k, (s - t0) * 1e-3, (s - prev_end) * 1e-3, (e - s) * 1e-3, short))
        prev_end = max(prev_end, e)
    print("by kernel: count, total us, gaps before them us")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-62s %4d %9.1f %9.1f" % (n, a[0], a[1], a[2]))
    sys.exit(0)

import time
import bench, torch
wl = sys.argv[1] if len(sys.argv) > 1 else "metro_station"
sim, cfg = bench.build_sim(wl)
eng = sim._engine
for rep in range(3):
    torch.cuda.synchronize()
    time.sleep(0.02)
    st, _ = eng.solve(0, sim.T, sim._nt0, m=None)
    torch.cuda.synchronize()
print("solve %.3f ms, attempts %d" % (1e3 * st["seconds"], st["n_attempt"]))
