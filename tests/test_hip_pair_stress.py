"""The pair kernel (gcfm.hip: tile boxes, neighbour lists, conservative float32 masks, two-candidate list entries,
chunked packed phase 2) against the reference's definition evaluated for EVERY ordered pair in NumPy float64:
sum over active j != i with sqrt(dx^2+dy^2) < cutoff of agents_repulsion (pedestrians.py:216-280,
simulations.py:285-295).  Crowds made to stress the culling: clusters next to voids (dense and empty tiles), a line of
agents spaced exactly the cutoff apart (the `<` decision at the boundary), stationary agents (Q19), inactive agents,
a few agents outside the room."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oc_oracle as oc                     # noqa: E402  (checker only)
from tests.test_hip_parity import _engine              # noqa: E402


def _all_pairs(x, y, vx, vy, vdes, status, cfg):
    n = len(x)
    F = np.zeros((n, 2))
    P = np.stack([x, y], 1)
    Vv = np.stack([vx, vy], 1)
    act = status.astype(bool)
    for i0 in range(0, n, 256):
        i1 = min(n, i0 + 256)
        dx = x[None, :] - x[i0:i1, None]
        dy = y[None, :] - y[i0:i1, None]
        near = (np.sqrt(dx * dx + dy * dy) < cfg["repulsion_cutoff"]) & act[None, :]
        near[np.arange(i1 - i0), np.arange(i0, i1)] = False
        f = oc.pair_force(P[i0:i1, None, :], Vv[i0:i1, None, :], vdes[i0:i1, None], P[None, :, :], Vv[None, :, :], cfg)
        f = np.where(near[..., None], f, 0.0)
        F[i0:i1] = f.sum(axis=1)
    F[~act] = 0.0
    return F


def test_pair_sums_match_all_pairs_numpy():
    cfg = copy.deepcopy(oc.DEFAULT_CONFIG)
    cfg["grid_step"] = 0.1
    eng, room, cfg, X0, Y0 = _engine("exit_opposite", "float32", n_keys=2, max_agents=8192, cfg=cfg)
    rng = np.random.RandomState(21)
    L = room["room_length"]
    xs, ys = [], []
    for _ in range(14):                                   # clusters of very different density, voids between them
        cx, cy, r, m = rng.uniform(5, L - 5), rng.uniform(5, L - 5), rng.uniform(0.8, 6.0), rng.randint(30, 400)
        xs.append(cx + r * rng.normal(size=m))
        ys.append(cy + r * rng.normal(size=m))
    xs.append(np.arange(12) * 4.0 + 2.0)                  # neighbours exactly at the cutoff: sqrt(16) < 4 is false
    ys.append(np.full(12, 1.5))
    xs.append(np.arange(12) * np.nextafter(4.0, 0.0) + 2.0)       # ... and one ulp inside it
    ys.append(np.full(12, 48.5))
    xs.append(np.array([-1.0, L + 0.7, 20.0]))            # outside the room
    ys.append(np.array([10.0, 30.0, -0.9]))
    x, y = np.concatenate(xs), np.concatenate(ys)
    n = len(x)
    vx, vy = rng.normal(0, 0.7, n), rng.normal(0, 0.7, n)
    still = rng.rand(n) < 0.05
    vx[still] = vy[still] = 0.0                           # |v| = 0: heading (1, 0), no vision (Q19)
    vdes = rng.normal(1.34, 0.26, n).clip(0.5, 2.2)
    status = rng.rand(n) > 0.06
    eng.set_agents(x, y, vx, vy, vdes, np.zeros(n, dtype=np.int32), status=status)
    got = eng.pair_force().cpu().numpy()
    want = _all_pairs(x, y, vx, vy, vdes, status, cfg)
    ok = status
    tol = 5e-5 * np.maximum(np.abs(want).max(axis=1), 1e-3)
    err = np.abs(got - want).max(axis=1)
    assert np.all(err[ok] <= tol[ok]), (int(np.argmax(err - tol)), err.max())
    assert np.all(got[~status] == 0.0)
    # the cutoff decision itself, on the two lines alone: at exactly the cutoff nobody interacts (sqrt(16) < 4 is
    # false); one ulp inside it, neighbours do
    lx = np.concatenate([np.arange(12) * 4.0 + 2.0, np.arange(12) * np.nextafter(4.0, 0.0) + 2.0])
    ly = np.concatenate([np.full(12, 1.5), np.full(12, 48.5)])
    lv = rng.normal(0, 0.7, (24, 2))
    eng.set_agents(lx, ly, lv[:, 0], lv[:, 1], np.full(24, 1.3), np.zeros(24, dtype=np.int32))
    got = eng.pair_force().cpu().numpy()
    want = _all_pairs(lx, ly, lv[:, 0], lv[:, 1], np.full(24, 1.3), np.ones(24), cfg)
    assert np.all(want[:12] == 0.0) and np.all(got[:12] == 0.0)
    assert np.any(want[12:] != 0.0) and np.array_equal(got[12:] != 0.0, want[12:] != 0.0)
    assert np.all(np.abs(got - want).max(axis=1) <= 5e-5 * np.maximum(np.abs(want).max(axis=1), 1e-3))
    eng.close()


def test_pair_sums_when_every_tile_neighbours_every_tile():
    """6000 agents in an 8 m x 6 m room (unphysical: 125 per m^2): 94 tiles that all lie within the cutoff of each
    other, so the neighbour lists overflow (> 63 tiles -> every tile scans all tiles), nearly every pair is a
    candidate and a virtual j-tile exceeds the candidate list (two rounds of 32 records)."""
    eng, room, cfg, X0, Y0 = _engine("gold_b", "float32", n_keys=2, max_agents=8192)
    rng = np.random.RandomState(4)
    n = 6000
    x, y = rng.uniform(0.3, room["room_length"] - 0.3, n), rng.uniform(0.3, room["room_height"] - 0.3, n)
    vx, vy = rng.normal(0, 0.7, n), rng.normal(0, 0.7, n)
    vdes = rng.normal(1.34, 0.26, n).clip(0.5, 2.2)
    status = rng.rand(n) > 0.02
    eng.set_agents(x, y, vx, vy, vdes, np.zeros(n, dtype=np.int32), status=status)
    got = eng.pair_force().cpu().numpy()
    want = _all_pairs(x, y, vx, vy, vdes, status, cfg)
    # sums of ~5000 terms of size <= 4 in float32 chunks: relative to the sum of magnitudes, not to the (cancelling) sum
    scale = np.maximum(np.abs(want).max(axis=1), 1.0) * 50
    assert np.all(np.abs(got - want).max(axis=1)[status] <= 5e-5 * scale[status])
    assert np.all(got[~status] == 0.0)
    # the same crowd through the step kernels with 1, 2 and 4 wavefronts per tile (virtual j-tiles, shared LDS sums):
    # one step from identical states; the speed clamp (v_max) maps the huge sums to directions
    X, Y = np.meshgrid(X0, Y0)
    names = []
    for box in room["initial_boxes"].values():
        k = " or ".join(box[5:])
        if k not in names:
            names.append(k)
    for k, name in enumerate(names):
        eng.set_potential(k, oc.finalize_potential(oc.create_potential(room, name.split(" or "), X, Y), cfg),
                          np.zeros((0, 4)))
    ref = None
    for split in (1, 2, 4):
        eng.set_tuning("pair_split", split)
        eng.set_agents(x, y, vx, vy, vdes, np.zeros(n, dtype=np.int32), status=status)
        eng.step(0, noise=np.zeros((n, 2)))
        st = eng.get_state()
        v = np.stack([st["vx"], st["vy"]], 1)
        if ref is None:
            ref = v
        assert np.abs(v - ref).max() <= 2e-4
    eng.close()
