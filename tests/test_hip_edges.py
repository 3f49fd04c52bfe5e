"""Edge cases of the HIP path (through the C-ABI): empty / single / inactive agents, agents outside the room,
grids the vector kernels cannot take, stale neighbour lists, in-place edits, exits."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oc_oracle as oc                     # noqa: E402  (checker only)
from optimal_crowds_amd import room_library            # noqa: E402
from tests.test_hip_parity import _engine, _rects      # noqa: E402


def _ready(dtype="float32", T=1.0, room="gold_b"):
    eng, room_d, cfg, X0, Y0 = _engine(room, dtype, n_keys=2)
    X, Y = np.meshgrid(X0, Y0)
    names = []
    for box in room_d["initial_boxes"].values():
        k = " or ".join(box[5:])
        if k not in names:
            names.append(k)
    nt = round(T / cfg["dt"])
    Vs = []
    for k, name in enumerate(names):
        V = oc.finalize_potential(oc.create_potential(room_d, name.split(" or "), X, Y), cfg)
        Vs.append(V)
        eng.set_potential(k, V, _rects(room_d, name))
        eng.alloc_field(k, nt - 1)
        eng.solve(k, T, nt)
    return eng, room_d, cfg, X0, Y0, names, Vs


def test_zero_one_and_inactive_agents():
    eng, room, cfg, X0, Y0, names, Vs = _ready()
    z = np.zeros(0)
    eng.set_agents(z, z, z, z, z, np.zeros(0, dtype=np.int32))
    assert eng.step(0, noise=np.zeros((0, 2))) == (0, 0, 0)
    eng.set_agents([3.0], [3.0], [0.3], [0.1], [1.3], np.zeros(1, dtype=np.int32))
    assert np.all(eng.pair_force().cpu().numpy() == 0.0)         # nobody to repel
    ex, bad, act = eng.step(0, noise=np.zeros((1, 2)))
    assert (ex, bad, act) == (0, 0, 1)
    n = 100
    rng = np.random.RandomState(0)
    x, y = rng.uniform(1, 7, n), rng.uniform(1, 5, n)
    eng.set_agents(x, y, np.zeros(n), np.zeros(n), np.ones(n), np.zeros(n, dtype=np.int32), status=np.zeros(n))
    assert eng.step(0, noise=np.zeros((n, 2))) == (0, 0, 0)
    st = eng.get_state()
    assert np.array_equal(st["x"], x) and np.array_equal(st["y"], y)       # inactive agents never move
    assert np.all(st["agent_time"] == 0)
    eng.close()


def test_inactive_agents_are_invisible_to_others():
    eng, room, cfg, X0, Y0, names, Vs = _ready()
    rng = np.random.RandomState(1)
    n = 200
    x, y = rng.uniform(1, 7, n), rng.uniform(1, 5, n)
    vx, vy = rng.normal(0, 0.5, n), rng.normal(0, 0.5, n)
    status = rng.rand(n) < 0.5
    eng.set_agents(x, y, vx, vy, np.full(n, 1.3), np.zeros(n, dtype=np.int32), status=status)
    f = eng.pair_force().cpu().numpy()
    act = np.nonzero(status)[0]
    eng.set_agents(x[act], y[act], vx[act], vy[act], np.full(len(act), 1.3), np.zeros(len(act), dtype=np.int32))
    f2 = eng.pair_force().cpu().numpy()
    assert np.allclose(f[act], f2, rtol=1e-6, atol=1e-9)       # only the summation order may differ
    assert np.all(f[~status] == 0.0)
    eng.close()


def test_agents_outside_the_room_lookup_wraps_like_numpy():
    eng, room, cfg, X0, Y0, names, Vs = _ready("float64")
    opt = oc.Optimals(Vs[0].copy(), 1.0, room["room_length"], room["room_height"], cfg)
    vx, vy = eng.field_interior(0)
    opt.vx_opt[:] = vx.cpu().numpy()
    opt.vy_opt[:] = vy.cpu().numpy()
    pos = np.array([[-0.03, 2.0], [-0.2, -0.1], [3.0, -0.07], [-1.0, 3.0], [7.99, 5.99], [8.3, 6.2]])
    n = len(pos)
    eng.set_agents(pos[:, 0], pos[:, 1], np.zeros(n), np.zeros(n), np.full(n, 1.2), np.zeros(n, dtype=np.int32))
    des, err = eng.lookup(5)
    des = des.cpu().numpy()
    assert err == 0
    for q in range(n):
        want = 1.2 * opt.choose_optimal_velocity(pos[q], 5)       # negative indices wrap (NumPy semantics)
        assert np.allclose(des[q], want, atol=1e-12)
    eng.close()


def test_grid_not_multiple_of_four_uses_simple_kernels():
    """Nx = 101: the 16-byte vector kernels do not apply; the one-array-per-pass kernels must give the same answer
    as the oracle's solve."""
    from optimal_crowds_amd.engine import Engine
    cfg = copy.deepcopy(oc.DEFAULT_CONFIG)
    room = room_library.gold_a()
    room["room_length"] = 5.03
    Nx, Ny = oc.grid_dims(room["room_length"], room["room_height"], cfg["grid_step"])
    assert Nx % 4 != 0
    X0, Y0 = oc.axes(room["room_length"], room["room_height"], Nx, Ny)
    X, Y = np.meshgrid(X0, Y0)
    V = oc.finalize_potential(oc.create_potential(room, ["east"], X, Y), cfg)
    T = 1.0
    nt = round(T / cfg["dt"])
    eng = Engine(cfg, room["room_length"], room["room_height"], Nx, Ny, X0, Y0, 1, 16, dtype="float64")
    eng.set_potential(0, V, np.zeros((0, 4)))
    eng.alloc_field(0, nt - 1)
    st, phi = eng.solve(0, T, nt, keep=[nt - 1])
    opt = oc.Optimals(V.copy(), T, room["room_length"], room["room_height"], cfg)
    keep = {}
    opt.compute_optimal_velocity(0.0, np.zeros_like(V), keep_phi=keep)
    assert st["nfev"] == opt.stats["nfev"]
    assert np.allclose(phi.cpu().numpy()[0].ravel(), keep[nt - 1], rtol=1e-9)
    assert np.allclose(eng.field_interior(0)[0].cpu().numpy(), opt.vx_opt, atol=1e-7)
    eng.close()


def _run_vs_oracle(eng, golden, nsteps, resort_every=None):
    g = golden("G7_step.npz")
    cfg = copy.deepcopy(oc.DEFAULT_CONFIG)
    np.random.seed(int(g["seed"]))
    sim = oc.Simulation(room_library.ROOMS["gold_b"](), float(g["T"]), cfg=cfg, mode="synchronous")
    for k in sim.key_names:
        sim.targets[k].compute_optimal_velocity(0.0, np.zeros_like(sim.V))
    sim.x, sim.y, sim.vx, sim.vy = (g[n].copy() for n in ("x", "y", "vx", "vy"))
    sim.status = g["status"].copy()
    sim.inside = int(sim.status.sum())
    sim.simu_step = int(g["simu_step"])
    if resort_every:
        eng.set_resort_every(resort_every)
    rng = np.random.RandomState(5)
    exits = 0
    for s in range(nsteps):
        noise = rng.normal(size=(sim.N, 2))
        step = sim.simu_step
        sim.step(perm=np.arange(sim.N), noise=noise)
        exits += eng.step(step, noise=noise)[0]
    st = eng.get_state()
    assert np.array_equal(st["status"].astype(bool), sim.status)
    assert exits == int(g["status"].sum() - sim.status.sum())
    assert np.abs(st["x"] - sim.x).max() <= 1e-3 and np.abs(st["y"] - sim.y).max() <= 1e-3
    return sim


def test_stale_neighbour_lists_fall_back_to_full_scan(golden):
    """No re-sort for 70 steps: agents move metres, far beyond the drift margin the neighbour lists were built
    with; the per-step validity check must route k_forces to the full tile scan and results stay right."""
    from tests.test_hip_parity import _g7_setup
    g, eng = _g7_setup(golden, "float32")
    _run_vs_oracle(eng, golden, 70, resort_every=100000)
    eng.close()


def test_in_place_edit_needs_agents_changed(golden):
    from tests.test_hip_parity import _g7_setup
    g, eng = _g7_setup(golden, "float32")
    eng.step(int(g["simu_step"]), noise=np.zeros((len(g["x"]), 2)))
    # teleport everybody (in place, same tensors), tell the library, and compare forces with a fresh engine state
    a = eng.agents
    a["x"].add_(0.37)
    a["y"].mul_(0.9).add_(0.2)
    eng.agents_changed()
    f1 = eng.pair_force().cpu().numpy()
    st = eng.get_state()
    eng.set_agents(st["x"], st["y"], st["vx"], st["vy"], g["v_des"], g["key_of_agent"], status=st["status"])
    f2 = eng.pair_force().cpu().numpy()
    assert np.array_equal(f1, f2)
    eng.close()


def test_dense_cluster_many_candidates():
    """200 agents inside a 1.5 m disc: every pair of every tile pair is a candidate (list capacity path)."""
    eng, room, cfg, X0, Y0, names, Vs = _ready()
    rng = np.random.RandomState(2)
    n = 200
    r, a = 0.75 * np.sqrt(rng.rand(n)), rng.uniform(0, 2 * np.pi, n)
    x, y = 4.0 + r * np.cos(a), 3.0 + r * np.sin(a)
    vx, vy = rng.normal(0, 0.6, n), rng.normal(0, 0.6, n)
    vd = rng.normal(1.34, 0.26, n)
    eng.set_agents(x, y, vx, vy, vd, np.zeros(n, dtype=np.int32))
    f = eng.pair_force().cpu().numpy()
    want = np.zeros((n, 2))
    P = np.column_stack([x, y]); Vv = np.column_stack([vx, vy])
    for i in range(n):
        js = np.array([j for j in range(n) if j != i])
        want[i] = oc.pair_force(P[i], Vv[i], vd[i], P[js], Vv[js], cfg).sum(axis=0)
    assert np.abs(f - want).max() <= 5e-5 * np.abs(want).max()
    eng.close()
