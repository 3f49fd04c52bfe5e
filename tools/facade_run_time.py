"""Wall time of the facade's own run() on a bench workload (untimed: construction): python tools/facade_run_time.py [workload]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch   # noqa: E402
import bench   # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "metro_station"
for rep in range(2):
    sim, cfg = bench.build_sim(wl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        sim.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s run(): %d steps in %.1f ms = %.0f steps/s, inside %d of %d" % (wl, sim.simu_step, dt * 1e3, sim.simu_step / dt,
                                                                           sim.inside, sim.N))
    del sim
