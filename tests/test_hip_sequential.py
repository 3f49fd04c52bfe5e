"""The reference's SEQUENTIAL update order on the GPU (oc_gcfm_step_sequential) and the gap between it and the
synchronous step, against what the reference itself produced:

* G7_step.npz holds ONE step of the reference's real `simulation.step` (seq_*: agents advanced one by one in the
  recorded permutation, simulations.py:271 + pedestrians.py:140-162) next to the frozen-snapshot composition (syn_*);
* G8_run_room_test.npz holds every 50th trajectory point of the reference's complete T = 20 run (1001 sequential steps).

Stated tolerances: one sequential step |dx| <= 1e-6 m (float32 pair arithmetic), |dv| <= 5e-5 m/s; the complete
room_test run reproduced to <= 2e-3 m at every recorded point with the float64 HJB state (the residual is float32 pair
forces through 1000 steps) and the same evacuees at the same steps.  The synchronous step differs from the reference's
sequential one by up to 2.8e-3 m / 0.12 m/s after ONE step of a dense crowd (the fixture's own two columns) -- that gap,
not arithmetic, is what update='sequential' removes.
"""
import copy
import io
import contextlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oc_oracle as oc                     # noqa: E402  (checker only)
from optimal_crowds_amd import room_library            # noqa: E402
from test_hip_parity import _g7_setup                  # noqa: E402


@pytest.mark.parametrize("dtype,tol", [("float64", 2e-7), ("float32", 1e-6)])
def test_sequential_step_reproduces_the_reference_step_g7(golden, dtype, tol):
    g, eng = _g7_setup(golden, dtype)
    exited, bad, active, levels = eng.step_sequential(int(g["simu_step"]), g["perm"], noise=g["noise"])
    st = eng.get_state()
    assert np.abs(st["x"] - g["seq_x"]).max() <= tol and np.abs(st["y"] - g["seq_y"]).max() <= tol
    assert np.abs(st["vx"] - g["seq_vx"]).max() <= 50 * tol and np.abs(st["vy"] - g["seq_vy"]).max() <= 50 * tol
    assert np.array_equal(st["status"].astype(bool), g["seq_status"])
    assert bad == 0 and active == int(g["seq_status"].sum()) and exited == int(g["status"].sum()) - active
    assert 2 <= levels <= len(g["x"]) + 4           # a dense 34-agent crowd: a chain of dependencies, not one level
    # the gap the mode exists for: the same state advanced synchronously is far outside that tolerance
    g2, eng2 = _g7_setup(golden, dtype)
    eng2.step(int(g["simu_step"]), noise=g["noise"])
    s2 = eng2.get_state()
    gap = max(np.abs(s2["x"] - g["seq_x"]).max(), np.abs(s2["y"] - g["seq_y"]).max())
    assert 1e-3 < gap < 5e-3                        # 2.8e-3 m after one step (fixture: |syn - seq|)
    eng.close()
    eng2.close()


def test_sequential_equals_oracle_sequential_over_many_steps(golden):
    """60 sequential steps from the G7 state, fresh permutations and noise: GPU against the oracle's sequential mode
    (which test_oracle_golden pins to the reference's own step) to 1e-4 m; every permutation needs several levels."""
    g, eng = _g7_setup(golden, "float64")
    cfg = copy.deepcopy(oc.DEFAULT_CONFIG)
    np.random.seed(int(g["seed"]))
    sim = oc.Simulation(room_library.ROOMS["gold_b"](), float(g["T"]), cfg=cfg, mode="sequential")
    for k in sim.key_names:
        sim.targets[k].compute_optimal_velocity(0.0, np.zeros_like(sim.V))
    sim.x, sim.y, sim.vx, sim.vy = (g[n].copy() for n in ("x", "y", "vx", "vy"))
    sim.status = g["status"].copy()
    sim.inside = int(sim.status.sum())
    sim.simu_step = int(g["simu_step"])
    rng = np.random.RandomState(5)
    nsteps = min(60, sim.targets[sim.key_names[0]].nt_opt - 2 - sim.simu_step)
    lv = []
    for s in range(nsteps):
        perm = rng.permutation(sim.N)
        noise = rng.normal(size=(sim.N, 2))
        step = sim.simu_step
        sim.step(perm=perm, noise=noise)
        exited, bad, active, levels = eng.step_sequential(step, perm, noise=noise)
        lv.append(levels)
        assert active == sim.inside
    st = eng.get_state()
    assert np.array_equal(st["status"].astype(bool), sim.status)
    assert np.abs(st["x"] - sim.x).max() <= 1e-4 and np.abs(st["y"] - sim.y).max() <= 1e-4
    assert min(lv) >= 2
    eng.close()


def test_full_run_trajectories_vs_reference_g8(golden):
    """simulation('room_test', 20, update='sequential').run() with the reference's seed: the same np.random stream, the
    same update order -> the reference's trajectories themselves (fixture G8, every 50th point of all 21 agents)."""
    from optimal_crowds_amd import simulations
    g = golden("G8_run_room_test.npz")
    np.random.seed(int(g["seed"]))
    with contextlib.redirect_stdout(io.StringIO()):
        sim = simulations.simulation("room_test", float(g["T"]), config=copy.deepcopy(oc.DEFAULT_CONFIG), noise="host",
                                     update="sequential", dtype="float64")
        sim.run()
    assert sim.simu_step == int(g["nsteps"]) and sim.inside == int(g["inside"])
    st = sim._engine.get_state()
    assert np.array_equal(st["status"].astype(bool), g["status"])
    assert np.allclose(st["agent_time"], g["agent_time"], atol=1e-9)        # every agent left in the reference's step
    rec = sim._rec.cpu().numpy()                        # (steps + 1, 4, N)
    worst = 0.0
    for n, s in enumerate(g["idx"]):
        ok = ~np.isnan(g["traj"][n, :, 0])              # agents still inside after s steps
        worst = max(worst, np.abs(rec[s, 0][ok] - g["traj"][n, ok, 0]).max(initial=0.0),
                    np.abs(rec[s, 1][ok] - g["traj"][n, ok, 1]).max(initial=0.0))
    assert worst <= 2e-3, worst
    assert max(sim.levels) >= 2


def test_sequential_margin_violation_is_reported(golden):
    """the level schedule rests on a displacement bound; a step that breaks it must fail loudly, not silently"""
    from optimal_crowds_amd._lib import OcError
    g, eng = _g7_setup(golden, "float64")
    eng.set_tuning("seq_margin_mm", 1)                  # 1 mm: agents at ~1 m/s move 20 mm per step
    with pytest.raises(OcError, match="seq_margin"):
        eng.step_sequential(int(g["simu_step"]), g["perm"], noise=g["noise"])
    eng.close()


def test_sequential_steps_at_config2_size_vs_oracle():
    """BASELINE configs[1] (exit_opposite: 512 x 512 grid, 5 000 agents at 2.5 / m^2, two keys): eight steps in the
    reference's update order -- permutation and noise drawn from np.random exactly as simulations.py:271,303 does -- on the
    GPU and in the oracle's sequential mode.  A dense crowd: every agent has ~120 neighbours within the cutoff, the
    permutation's dependency graph is tens of levels deep, and any agent advanced out of turn would show up as a
    difference of millimetres (the synchronous step differs by 2.8e-3 m after one step, G7).  After ONE step every agent
    agrees to 2e-6 m (float32 pair arithmetic and direction field against float64); after eight the median is 1e-7 m
    and the worst agent 5e-4 m (8e-5 measured: in a crowd this dense with random velocities a perturbation grows
    several-fold per step, which is the dynamics); the same agents are inside."""
    import os
    from optimal_crowds_amd import host, simulations
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = copy.deepcopy(host.load_config(os.path.join(REPO, "optimal_crowds_amd", "config.json")))
    cfg["grid_step"] = 0.1
    T = 1.0
    room = room_library.ROOMS["exit_opposite"]()
    np.random.seed(77)
    with contextlib.redirect_stdout(io.StringIO()):
        sim = simulations.simulation("exit_opposite", T, config=cfg, noise="host", update="sequential", record=False,
                                     history=False, wall_search_radius=0.0)
        sim._solve_all()
    np.random.seed(77)
    ref = oc.Simulation(room, T, cfg=cfg, mode="sequential")
    assert np.array_equal(sim._x0, ref.x) and np.array_equal(sim._v_des, ref.v_des)         # same crowd
    ref.solve_all()
    # a crowd that already moves (k = 0 for a stationary agent, pedestrians.py:259-262): the same velocities on both sides
    rng = np.random.RandomState(5)
    v0 = rng.normal(0.0, 0.6, size=(sim.N, 2))
    ref.vx[:], ref.vy[:] = v0[:, 0], v0[:, 1]
    a = sim._engine.agents
    import torch
    a["vx"].copy_(torch.as_tensor(v0[:, 0])), a["vy"].copy_(torch.as_tensor(v0[:, 1]))
    sim._engine.agents_changed()
    for s in range(8):
        state = np.random.get_state()
        with contextlib.redirect_stdout(io.StringIO()):
            sim.step(sim.dt)
        np.random.set_state(state)
        ref.step()
        if s == 0:
            st = sim._engine.get_state()
            d1 = np.hypot(st["x"] - ref.x, st["y"] - ref.y)
            assert d1.max() <= 2e-6, d1.max()
            assert np.abs(st["vx"] - ref.vx).max() <= 1e-4 and np.abs(st["vy"] - ref.vy).max() <= 1e-4
    st = sim._engine.get_state()
    assert np.array_equal(st["status"].astype(bool), ref.status) and sim.inside == ref.inside
    d = np.hypot(st["x"] - ref.x, st["y"] - ref.y)
    assert np.median(d) <= 1e-7 and d.max() <= 5e-4, (np.median(d), d.max())
    assert min(sim.levels) >= 10            # a deep dependency graph: the schedule is doing real work
    print("levels per step:", sim.levels)
