import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
sim, cfg = bench.build_sim("metro_station")
eng = sim._engine
bench.run_steps(sim, 20, sim._nt0)
torch.cuda.synchronize()
for prof in (False, True):
    eng.set_profiling(prof); eng.reset_kernel_stats()
    t0 = time.perf_counter()
    for s in range(300):
        eng.step(30 + s, noise=None, seed=1, sync=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("profiling", prof, "host enqueue us/step %.1f   total us/step %.1f" % ((t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6))
