"""Diagnostic (GPU): h_abs0 / attempts of three keys solved one after the other and concurrently, several trials."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oc_oracle as oc
from test_hjb_rows import _density_like, _setup
from optimal_crowds_amd.engine import Engine
cfg, room, Nx, Ny, X0, Y0, _ = _setup("gold_b", 0.05)
X, Y = np.meshgrid(X0, Y0)
names = list(room["targets"].keys())
keysets = [[names[0]], [names[-1]], names[:2]]
Vs = [oc.finalize_potential(oc.create_potential(room, ks, X, Y), cfg) for ks in keysets]
T, nt = 3.0, 150
dtype = sys.argv[1] if len(sys.argv) > 1 else "mixed"
m = torch.as_tensor(_density_like(Ny, Nx))
for trial in range(5):
    for overlap in (False, True):
        eng = Engine(cfg, room["room_length"], room["room_height"], Nx, Ny, X0, Y0, len(Vs), 4096, dtype=dtype)
        eng.overlap_solves = overlap
        eng.set_tuning("hjb_active", 2)
        for k, V in enumerate(Vs):
            eng.set_potential(k, V, np.zeros((0, 4)))
            eng.alloc_field(k, nt - 1)
        s1 = eng.solve_keys([(k, T, nt, m) for k in range(len(Vs))])
        s2 = eng.solve_keys([(k, T, nt - (25 if k == 1 else 0), m) for k in range(len(Vs))])
        print(trial, "overlap" if overlap else "serial ", " ".join("%.17g/%d" % (s["h_abs0"], s["n_attempt"]) for s in s1 + s2), flush=True)
        eng.close()
