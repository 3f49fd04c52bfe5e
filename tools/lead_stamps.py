#!/usr/bin/env python3
"""Where an attempt of the LEAD launch chain spends its time (diagnostic build, GPU box):

    OC_HIP_LIB=$PWD/optimal_crowds_amd/liboc_hip_diag.so OC_LEAD_STAMPS=/tmp/st.bin python tools/lead_stamps.py [workload]

k_stageV<.., LEAD> stamps every attempt with s_memrealtime (100 MHz) in its LEADER block (list position 0):
entered | loads landed | decision formed | prologue done | tile done | stores out."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, torch

wl = sys.argv[1] if len(sys.argv) > 1 else "metro_station"
path = os.environ.get("OC_LEAD_STAMPS", "/tmp/lead_stamps.bin")
os.environ["OC_LEAD_STAMPS"] = path
sim, cfg = bench.build_sim(wl)
eng = sim._engine
eng.set_tuning("hjb_persist", 0)
eng.set_tuning("hjb_lead", 1)
for rep in range(2):
    st, _ = eng.solve(0, sim.T, sim._nt0, m=None)
torch.cuda.synchronize()
print("solve %.3f ms, attempts %d" % (1e3 * st["seconds"], st["n_attempt"]))
a = np.fromfile(path, dtype=np.uint64).reshape(-1, 16).astype(np.int64)
idx = np.nonzero(a[:, 15] == 2)[0]
rows = a[idx]
d = np.diff(rows[:, :6], axis=1) * 0.01
names = ["loads", "decision", "write out", "tile", "stores"]
print("LEADER block of %d no-tail attempts (us): " % len(rows) + "  ".join("%s %.2f" % (n, v) for n, v in zip(names, np.median(d, axis=0))))
consec = np.nonzero(np.diff(idx) == 1)[0]
per = (rows[consec + 1, 0] - rows[consec, 0]) * 0.01
gap = (rows[consec + 1, 0] - rows[consec, 5]) * 0.01
print("attempt period %.2f us (min %.2f max %.2f); leader's stores out -> next leader entered %.2f us (min %.2f); items mean %.1f"
      % (per.mean(), per.min(), per.max(), gap.mean(), gap.min(), rows[:, 7].mean()))
seq = np.stack([rows[:, 1], rows[:, 13], rows[:, 14], rows[:, 6], rows[:, 8], rows[:, 9], rows[:, 10], rows[:, 11], rows[:, 12], rows[:, 2]], axis=1)
dd = np.diff(seq, axis=1) * 0.01
print("inside the decision, medians (us): loads landed -> controller's first barrier %.2f | records in LDS %.2f | shares written %.2f | max and count %.2f | barrier %.2f | "
      "SERIAL %.2f | beside + barrier %.2f | tables %.2f | rest %.2f" % tuple(np.median(dd, axis=0)))
for q in (5, 30, 60, 90):
    if q < len(rows):
        print("attempt %3d: items %3d  " % (idx[q], rows[q, 7]) + "  ".join("%s %.2f" % (n, v) for n, v in zip(names, d[q])))

# a block in the MIDDLE of the list (a computed tile): entered | prologue done | loads landed, tile tested | six stages | stores issued |
# reduction + record | stores out
mid = a[2048:][a[2048:, 15] == 4]
if len(mid):
    seq = np.stack([mid[:, 0], mid[:, 3], mid[:, 8], mid[:, 9], mid[:, 10], mid[:, 11], mid[:, 5]], axis=1)
    dd = np.diff(seq, axis=1) * 0.01
    print("MIDDLE block of %d attempts, medians (us): prologue %.2f | tile loads + test %.2f | six stages %.2f | outputs formed, stores issued %.2f | "
          "reduction + record %.2f | stores out %.2f | total %.2f" % ((len(mid),) + tuple(np.median(dd, axis=0)) + (np.median(seq[:, -1] - seq[:, 0]) * 0.01,)))
