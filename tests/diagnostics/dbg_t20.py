"""Diagnostic (GPU): float32 against float64 HJB solves of room_test at T = 20 -- which cells differ and by how much.
Lives under tests/ because it uses the oracle as a checker (only tests/, smoke() and bench.py's cpu_baseline may)."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oc_oracle as oc
from optimal_crowds_amd import room_library
from optimal_crowds_amd.engine import Engine
cfg = copy.deepcopy(oc.DEFAULT_CONFIG)
room = room_library.room_test()
Nx, Ny = oc.grid_dims(room["room_length"], room["room_height"], cfg["grid_step"])
X0, Y0 = oc.axes(room["room_length"], room["room_height"], Nx, Ny)
X, Y = np.meshgrid(X0, Y0)
V = oc.finalize_potential(oc.create_potential(room, ["door"], X, Y), cfg)
T = 20.0; nt = round(T / cfg["dt"])
res = {}
keepk = [nt - 1 - s for s in (0, 3, 250, 600, 950, nt - 2)]
for dt in ("float64", "float32"):
    eng = Engine(cfg, room["room_length"], room["room_height"], Nx, Ny, X0, Y0, 1, 16, dtype=dt)
    eng.set_potential(0, V, np.zeros((0, 4)))
    eng.alloc_field(0, nt - 1)
    st, phi = eng.solve(0, T, nt, keep=keepk)
    res[dt] = (st, phi.cpu().numpy(), [t.double().cpu().numpy() for t in eng.field_interior(0)])
    print(dt, {k: st[k] for k in ("nfev", "n_attempt", "n_accept", "n_rescale", "final_exponent")})
    eng.close()
p64, p32 = res["float64"][1], res["float32"][1]
for q, k in enumerate(keepk):
    rel = np.abs(p32[q] - p64[q]) / np.abs(p64[q])
    s = nt - 1 - k
    gx, gy = res["float64"][2][0][s], res["float64"][2][1][s]
    vx, vy = res["float32"][2][0][s], res["float32"][2][1][s]
    both = (np.hypot(gx, gy) > .5) & (np.hypot(vx, vy) > .5)
    ang = np.abs(np.arctan2(gx * vy - gy * vx, gx * vx + gy * vy)) * both
    i, j = np.unravel_index(ang.argmax(), ang.shape)
    print("slice", s, "phi rel err max %.2e median %.2e | max angle %.3e at interior (%d,%d) phi64 there %.4e rel err there %.2e; neighbours rel err" % (
        rel.max(), np.median(rel), ang.max(), i, j, p64[q][i+1, j+1], rel[i+1, j+1]), rel[i:i+3, j:j+3].round(6).tolist())
