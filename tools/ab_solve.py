#!/usr/bin/env python3
"""A/B of the HJB solve's launch sequences on one workload (GPU box): python tools/ab_solve.py [workload] [knob=v,v ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch

wl = sys.argv[1] if len(sys.argv) > 1 else "metro_station"
knobs = [a.split("=") for a in sys.argv[2:]] or [["hjb_tail", "1,0"]]
sim, cfg = bench.build_sim(wl)
eng = sim._engine
name, vals = knobs[0]
for v in [int(x) for x in vals.split(",")] * 2:
    eng.set_tuning(name, v)
    ts = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st, _ = eng.solve(0, sim.T, sim._nt0, m=None)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print("%s=%d: solve %s ms (lib %.3f ms)  attempts %d accepted %d" % (name, v, " ".join("%.3f" % (1e3 * t) for t in ts), 1e3 * st["seconds"], st["n_attempt"], st["n_accept"]))
