#!/usr/bin/env python3
"""A/B of knob COMBINATIONS on the HJB solve of one workload (GPU box):
    python tools/ab_combo.py [workload] hjb_persist=0,hjb_tail_fast=0 hjb_persist=1,hjb_tail_fast=1 ...
Every combination is timed twice (interleaved), 4 solves each; also both keys concurrently (Engine.solve_keys) when K > 1."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch

args = sys.argv[1:]
wl = args.pop(0) if args and "=" not in args[0] else "metro_station"
combos = [[kv.split("=") for kv in a.split(",")] for a in args] or [[["hjb_persist", "0"]], [["hjb_persist", "2"]]]
sim, cfg = bench.build_sim(wl)
eng = sim._engine
K = len(sim._key_names)
for combo in combos * 2:
    for name, v in combo:
        eng.set_tuning(name, int(v))
    ts, tk = [], []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st, _ = eng.solve(0, sim.T, sim._nt0, m=None)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    for rep in range(4 if K > 1 else 0):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sts = bench.solve_round(sim)
        torch.cuda.synchronize()
        tk.append(time.perf_counter() - t0)
    print("%-40s solve %s ms | %d keys at once %s ms | attempts %d accepted %d persistent %d (exit %d, %d blocks)" % (
        ",".join("%s=%s" % (n, v) for n, v in combo), " ".join("%.3f" % (1e3 * t) for t in ts), K,
        " ".join("%.3f" % (1e3 * t) for t in tk), st["n_attempt"], st["n_accept"], st["persist_attempts"], st["persist_exit"],
        st["persist_blocks"]), flush=True)
