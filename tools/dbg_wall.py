import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_hip_parity import _engine, _rects
g = np.load('tests/golden/G6_forces.npz')
eng, room, cfg, X0, Y0 = _engine("gold_b", "float32", n_keys=2)
names = [str(s) for s in g["key_names"]]
for k, name in enumerate(names):
    eng.set_potential(k, g["V_%d" % k], _rects(room, name))
n = len(g["wp"])
for k in range(2):
    eng.set_agents(g["wp"][:, 0], g["wp"][:, 1], g["wv"][:, 0], g["wv"][:, 1], g["wvdes"], np.full(n, k, dtype=np.int32))
    idx, f = eng.wall_force()
    f = f.cpu().numpy(); w = g["wforce"][k]
    err = np.abs(f - w); rel = err / np.maximum(np.abs(w), 1e-300)
    bad = np.argsort(-np.nan_to_num(rel.max(axis=1)))[:5]
    for b in bad: print(k, b, f[b], w[b], rel[b], g["wp"][b], g["wv"][b])
