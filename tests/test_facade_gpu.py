"""The reference-shaped Python surface (simulations.simulation / optimals / pedestrians) on the GPU."""
import copy
import io
import contextlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oc_oracle as oc                     # noqa: E402  (checker only)
from optimal_crowds_amd import room_library            # noqa: E402


def _cfg():
    return copy.deepcopy(oc.DEFAULT_CONFIG)


def test_run_matches_oracle_synchronous_and_reference_statistics(golden):
    """simulation('room_test', T).run() with host noise consumes np.random exactly like the reference, so the
    oracle in synchronous mode (same seed) is the trajectory-level checker; the reference's own sequential run
    (fixture G8) is the statistical one."""
    from optimal_crowds_amd import simulations
    T = 6.0
    np.random.seed(1)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sim = simulations.simulation("room_test", T, config=_cfg(), noise="host")
        sim.run()
    printed = buf.getvalue()
    assert printed.startswith("ABM simulation room created!\nComputing trajectories at time 0.0\n"
                              "Optimal trajectories have been learnt for door\n")
    np.random.seed(1)
    ref = oc.Simulation(room_library.ROOMS["room_test"](), T, cfg=_cfg(), mode="synchronous")
    g1 = golden("G1_room_test.npz")
    assert np.array_equal(sim._x0, g1["x"]) and np.array_equal(sim._v_des, g1["v_des"])   # same start as reference
    ref.run()
    assert sim.simu_step == ref.simu_step and sim.inside == ref.inside
    assert sim.time == pytest.approx(ref.time, abs=1e-12)
    st = sim._engine.get_state()
    assert np.array_equal(st["status"].astype(bool), ref.status)
    assert np.abs(st["x"] - ref.x).max() < 2e-3 and np.abs(st["y"] - ref.y).max() < 2e-3
    assert np.allclose(st["agent_time"], ref.agent_time, atol=1e-9)
    # facade views
    a = sim.agents[3]
    assert a.target == "door" and len(a.traj) == int(round(a.time / sim.dt)) + 1
    assert np.allclose(a.traj[0], a.initial_position) and np.allclose(a.traj[-1], a.position())
    xs, ys = sim.initial_positions()
    assert np.array_equal(xs, g1["x"])
    with pytest.raises(ValueError):
        sim.evac_times()
    # history frame: active agents + density; density equals the oracle's on the same snapshot
    t1 = sorted(sim.history)[10]
    frame = sim.history[t1]
    assert frame[-1].shape == (sim.Ny, sim.Nx) and len(frame) - 1 <= sim.N
    d = sim.gaussian_density(0.5)
    want = oc.gaussian_density(st["x"], st["y"], st["status"].astype(bool), ref.X, ref.Y, ref.V, 0.5)
    assert np.abs(d - want).max() <= 2e-7 * max(want.max(), 1e-30)
    # lazily exposed direction field
    vx = sim.targets["door"].vx_opt
    assert vx.shape == (round(T / 0.02) - 1, sim.Ny - 2, sim.Nx - 2)
    o = ref.targets["door"].vx_opt[5]
    assert (np.abs(vx[5] - o) < 1e-2).mean() > 0.995


def test_full_run_statistics_vs_reference_g8(golden):
    """T = 20 run of room_test: the synchronous GPU run evacuates the same number of agents (+-2) as the
    reference's sequential run with the same seed, at similar times."""
    from optimal_crowds_amd import simulations
    g = golden("G8_run_room_test.npz")
    np.random.seed(int(g["seed"]))
    with contextlib.redirect_stdout(io.StringIO()):
        sim = simulations.simulation("room_test", float(g["T"]), config=_cfg(), noise="host")
        sim.run()
    assert sim.simu_step == int(g["nsteps"])
    assert abs(sim.inside - int(g["inside"])) <= 2
    t_gpu = np.sort(sim._engine.get_state()["agent_time"])
    t_ref = np.sort(g["agent_time"])
    k = min((t_gpu < 20).sum(), (t_ref < 20).sum())
    assert np.abs(t_gpu[:k] - t_ref[:k]).max() < 1.5          # exit times of the evacuated agents, seconds


def test_recompute_horizon_semantics(golden):
    """recompute=True: nt_opt shrinks at every recompute exactly like the reference (quirk Q9)."""
    from optimal_crowds_amd import simulations
    g = golden("G8_run_gold_b_recompute.npz")
    np.random.seed(int(g["seed"]))
    nts = []
    with contextlib.redirect_stdout(io.StringIO()):
        sim = simulations.simulation("gold_b", float(g["T"]), recompute=True, config=_cfg(), noise="host")
        for key, opt in sim.targets.items():
            f = opt.compute_optimal_velocity

            def wrapped(t, m, _f=f, _o=opt):
                _f(t, m)
                nts.append((sim.simu_step, _o.nt_opt))
            opt.compute_optimal_velocity = wrapped
        sim.run()
    assert np.array_equal(np.array(nts), g["nt_after"])
    assert sim.simu_step == int(g["nsteps"])
    assert abs(sim.inside - int(g["inside"])) <= 3


def test_single_point_api():
    from optimal_crowds_amd import simulations
    np.random.seed(12)
    with contextlib.redirect_stdout(io.StringIO()):
        sim = simulations.simulation("gold_b", 1.0, config=_cfg(), noise="host")
        sim.targets["east"].compute_optimal_velocity(0.0, None)
    v = sim.targets["east"].choose_optimal_velocity(np.array((7.2, 3.1)), 3)
    assert v.shape == (2,) and abs(np.hypot(*v) - 1.0) < 1e-5
    assert np.all(sim.targets["east"].choose_optimal_velocity(np.array((3.0, 3.0)), 10 ** 6) == 0)
    with pytest.raises(IndexError):
        sim.targets["east"].choose_optimal_velocity(np.array(((sim.Nx - 3) * sim.dx + 0.01, 3.0)), 3)
    a = sim.agents[0]
    f = a.wall_repulsion(sim.X_opt, sim.Y_opt, sim.Vs[a.target])
    ref = oc.wall_force(a.position(), a.velocity(), a.v_des, sim.X_opt, sim.Y_opt, sim.Vs[a.target], _cfg())
    assert np.allclose(f, ref, rtol=1e-10, atol=1e-14)
    pj, vj = a.position() + np.array((0.4, 0.1)), np.array((-0.5, 0.2))
    fr = a.agents_repulsion(pj, vj)
    want = oc.pair_force(a.position(), a.velocity(), a.v_des, pj, vj, _cfg())
    assert np.allclose(fr, want, rtol=5e-5, atol=1e-7)


def test_run_without_a_round_trip_per_step_ends_where_the_stepwise_loop_ends():
    """run() with device noise enqueues the steps and reads the device-side run status every 16 steps; it must stop
    -- time, simu_step, inside, agent state, history marks -- exactly where the step-by-step loop (verbose=True)
    stops, also when the evacuation completes between two looks."""
    import contextlib
    import io
    from optimal_crowds_amd import host, simulations
    cfg = copy.deepcopy(host.load_config())
    cfg["grid_step"] = 0.1
    between = 0
    for seed in (5, 6, 7):              # (which step the last agent leaves in depends on the noise stream: one of these
        out = []                        #  runs has to end between two looks, all of them have to end alike)
        for verbose in (False, True):
            np.random.seed(3)
            with contextlib.redirect_stdout(io.StringIO()) as buf:
                sim = simulations.simulation("room_test", 25.0, config=cfg, noise="device", seed=seed, record=True,
                                             history=True, wall_search_radius=0.0)
                sim.run(verbose=verbose)
            st = sim._engine.get_state()
            out.append((sim.simu_step, sim.time, sim.inside, st, sorted(sim.history.keys()), buf.getvalue().splitlines()[-1],
                        sim.evac_times() if sim.inside == 0 else None))
            sim._engine.close()
        a, b = out
        assert a[2] == 0 and b[2] == 0, "the crowd should be out well before T = 25 s"
        assert a[0] == b[0] and a[1] == b[1]
        between += 1 if a[0] % 16 != 0 else 0
        for f in ("x", "y", "vx", "vy", "status", "agent_time"):
            assert np.array_equal(a[3][f], b[3][f])
        assert a[4] == b[4]
        assert a[5] == b[5] and a[5].startswith("Evacuation complete")
        assert np.array_equal(a[6], b[6])
    assert between >= 1                 # (ends between two looks)


def test_long_horizon_switches_the_hjb_state_to_float64():
    """T = 30 s: phi reaches e^150, beyond what float32's range management covers (the solve used to end with a NaN
    error norm); the facade then runs the same kernels on a float64 state ('mixed': the direction field stays float32)
    -- checked against the oracle's solve -- while T = 20 s stays float32."""
    from optimal_crowds_amd import simulations
    from optimal_crowds_amd.engine import hjb_state_dtype
    cfg = _cfg()
    assert hjb_state_dtype(cfg, 20.0) == "float32" and hjb_state_dtype(cfg, 25.0) == "float32"
    assert hjb_state_dtype(cfg, 30.0) == "mixed" and hjb_state_dtype(cfg, 30.0, "float64") == "float64"
    assert hjb_state_dtype(cfg, 20.0, "mixed") == "mixed"
    T = 30.0
    np.random.seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        sim = simulations.simulation("gold_a", T, config=cfg, noise="host", record=False)
        assert sim.hjb_dtype == "mixed"         # float64 state, float32 direction field
        key = sim._key_names[0]
        sim.targets[key].compute_optimal_velocity(0.0, None)
    st = sim.targets[key].stats
    assert st["n_rescale"] == 0 and st["n_slices_written"] == round(T / cfg["dt"]) - 1
    opt = oc.Optimals(sim.Vs[key].copy(), T, sim.room_length, sim.room_height, cfg)
    opt.compute_optimal_velocity(0.0, np.zeros((sim.Ny, sim.Nx)))
    # (1500 slices, ~570 steps at the stability limit: a rejection whose error norm sits at 1 +- 1e-12 may fall the other
    # way in the two float64 implementations)
    assert abs(st["n_accept"] - opt.stats["n_accept"]) <= 3 and abs(st["nfev"] - opt.stats["nfev"]) <= 0.02 * st["nfev"]
    vx, vy = sim.targets[key].vx_opt, sim.targets[key].vy_opt
    for s in (0, 700, 1498):
        gx, gy, hx, hy = opt.vx_opt[s], opt.vy_opt[s], vx[s], vy[s]
        assert not np.isnan(hx).any()
        same = (np.hypot(gx, gy) > 0.5) == (np.hypot(hx, hy) > 0.5)
        assert same.mean() >= 0.999
        # two float64 implementations whose roundings differ (the kernels fuse their multiply-adds) drift apart in the
        # integrator's marginally stable mode over ~570 steps (tests/hjb_compare.py): 5e-4 measured at T = 30
        assert np.abs(hx - gx)[same].max() <= 2e-3 and np.abs(hy - gy)[same].max() <= 2e-3


def test_reference_import_line_works_unmodified():
    """`from optimal_crowds import simulations` (the reference's README recipe) resolves to this implementation"""
    import optimal_crowds
    from optimal_crowds import simulations as s1
    from optimal_crowds_amd import simulations as s2
    import optimal_crowds.optimals as o1
    assert s1 is s2 and o1 is optimal_crowds.optimals and hasattr(s1, "simulation")


def test_index_error_in_the_enqueued_run_leaves_the_state_of_the_offending_step():
    """An agent that walks into the reference's IndexError band (quirk Q8: x in [(Nx-3) dx, L - dx)) makes step() raise
    inside that step.  run() with device noise enqueues 16 steps between two looks at the device-side status; when the
    flag shows up, simu_step / time AND the agent state must be those the step-by-step loop stops with."""
    from optimal_crowds_amd import simulations
    cfg = _cfg()
    cfg["grid_step"] = 0.1
    room = {"room_length": 10.0, "room_height": 10.0,
            "initial_boxes": {"b": [8.2, 5.0, 1.0, 3.0, 1.0, "door"]},
            "targets": {"door": [9.95, 5.0, 0.3, 6.0]},         # reached only through the band 9.7 <= x < 9.8
            "walls": {}, "holes": {}, "cylinders": {}}
    out = []
    for verbose in (False, True):
        np.random.seed(4)
        with contextlib.redirect_stdout(io.StringIO()):
            sim = simulations.simulation(room, 4.0, config=cfg, noise="device", seed=9, record=True)
            with pytest.raises(IndexError):
                sim.run(verbose=verbose)
        st = sim._engine.get_state()
        out.append((sim.simu_step, sim.time, sim.inside, st))
        assert 5 < sim.simu_step < 150
        sim._engine.close()
    (s0, t0, in0, a), (s1, t1, in1, b) = out
    assert (s0, t0, in0) == (s1, t1, in1)
    for f in ("x", "y", "vx", "vy", "status"):
        assert np.array_equal(a[f], b[f]), f
    assert np.allclose(a["agent_time"], b["agent_time"], atol=1e-12)
