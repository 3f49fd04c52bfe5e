#!/usr/bin/env python3
"""Where the wall time of Engine.solve_keys goes (GPU box): python tools/ab_keys.py [workload]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch

wl = sys.argv[1] if len(sys.argv) > 1 else "metro_station"
sim, cfg = bench.build_sim(wl)
eng = sim._engine
K = len(sim._key_names)
jobs = [(k, sim.T, sim._nt0, None) for k in range(K)]
for rep in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sts = eng.solve_keys(jobs)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("solve_keys %.3f ms (+%.3f sync) | inside the library per key: %s ms" % (
        1e3 * (t1 - t0), 1e3 * (t2 - t1), " ".join("%.3f" % (1e3 * s["seconds"]) for s in sts)), flush=True)
eng.overlap_solves = False
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sts = eng.solve_keys(jobs)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("one after the other %.3f ms | inside the library per key: %s ms" % (
        1e3 * (t1 - t0), " ".join("%.3f" % (1e3 * s["seconds"]) for s in sts)), flush=True)
