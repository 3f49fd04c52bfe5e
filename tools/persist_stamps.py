#!/usr/bin/env python3
"""Where an attempt of the persistent HJB launch spends its time (diagnostic build, GPU box):

    OC_HIP_LIB=$PWD/optimal_crowds_amd/liboc_hip_diag.so OC_PERSIST_STAMPS=/tmp/st.bin python tools/persist_stamps.py [workload]

k_stageV<.., PERSIST> stamps every attempt with s_memrealtime (100 MHz) in its LAST block (the one that runs the controller):
attempt entered | controller block read | items done | ticket drawn | tail done | generation stored."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, torch

wl = sys.argv[1] if len(sys.argv) > 1 else "metro_station"
path = os.environ.get("OC_PERSIST_STAMPS", "/tmp/persist_stamps.bin")
os.environ["OC_PERSIST_STAMPS"] = path
sim, cfg = bench.build_sim(wl)
eng = sim._engine
eng.set_tuning("hjb_persist", 2)
for rep in range(2):
    st, _ = eng.solve(0, sim.T, sim._nt0, m=None)
torch.cuda.synchronize()
print("solve %.3f ms, attempts %d, persistent %d on %d blocks, exit %d" % (1e3 * st["seconds"], st["n_attempt"], st["persist_attempts"],
                                                                        st["persist_blocks"], st["persist_exit"]))
a = np.fromfile(path, dtype=np.uint64).reshape(-1, 16).astype(np.int64)
rows = a[a[:, 5] > 0]
last = rows[:, :8]
d = np.diff(last[:, :6], axis=1) * 0.01          # us
names = ["read ctl", "items", "ticket", "tail", "publish"]
print("LAST block of %d attempts (us): " % len(rows) + "  ".join("%s %.2f" % (n, v) for n, v in zip(names, d.mean(axis=0))))
gap = (last[1:, 0] - last[:-1, 5]) * 0.01        # generation stored -> the next attempt's last block entered
per = (last[1:, 5] - last[:-1, 5]) * 0.01
print("attempt period %.2f us (min %.2f max %.2f); publish -> next last block entered %.2f us; items per attempt mean %.1f max %d"
      % (per.mean(), per.min(), per.max(), gap.mean(), last[:, 7].mean(), last[:, 7].max()))
cb = rows[rows[:, 15] == 1]
if len(cb):
    # inside the tail: ticket drawn [3] -> control_body: loads + reduction issued [8], first barrier passed [9], serial decision
    # done [10], `beside` + barrier [11], tables + last barrier [12] -> tail done [4]
    seq = np.stack([cb[:, 3], cb[:, 8], cb[:, 9], cb[:, 10], cb[:, 11], cb[:, 12], cb[:, 4]], axis=1)
    dd = np.diff(seq, axis=1) * 0.01
    print("inside the tail (us): to reduction issued %.2f | barrier %.2f | SERIAL decision %.2f | beside + barrier %.2f | tables %.2f | "
          "list + controller block out %.2f" % tuple(dd.mean(axis=0)))
for q in (5, 30, 60, 90):
    if q < len(rows):
        print("attempt %3d: items %3d  " % (q + 1, last[q, 7]) + "  ".join("%s %.2f" % (n, v) for n, v in zip(names, d[q])))
