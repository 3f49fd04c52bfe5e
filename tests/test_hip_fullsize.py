"""BASELINE-size checks (2048 x 2048 grid, 100 000 agents, 2 keys): properties that do not need a full-size oracle
run, plus oracle spot checks on samples.  Takes ~30 s on an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oc_oracle as oc                     # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def c4():
    import bench
    sim, cfg = bench.build_sim("metro_station")
    stats = bench.solve_round(sim)
    yield sim, cfg, stats
    import torch
    sim._engine.close()             # 67 GB of direction field: give it back before the next module's fixtures
    sim._engine.fields = []
    del sim
    torch.cuda.empty_cache()


def test_c4_rhs_full_grid_vs_oracle(c4):
    import torch
    sim, cfg, stats = c4
    eng = sim._engine
    hp = cfg["hjb_params"]
    rng = np.random.RandomState(0)
    phi = np.exp(rng.normal(size=sim.Ny * sim.Nx))
    V = sim.Vs[sim._key_names[0]]
    eng.make_coef(0, None)
    got = eng.rhs(torch.as_tensor(phi.astype(np.float32)).to(eng.device)).double().cpu().numpy()
    want = oc.hjb_rhs(phi.astype(np.float32).astype(np.float64), V, np.zeros_like(V), hp["sigma"], hp["mu"], hp["g"],
                      cfg["grid_step"], cfg["grid_step"])
    assert np.abs(got - want).max() <= 3e-6 * np.abs(want).max()
    # linearity of the operator (walls frozen): rhs(2 phi) = 2 rhs(phi) exactly in floating point
    got2 = eng.rhs(torch.as_tensor((2 * phi).astype(np.float32)).to(eng.device)).double().cpu().numpy()
    assert np.array_equal(got2, 2 * got)


def test_c4_direction_field_is_unit_or_zero(c4):
    sim, cfg, stats = c4
    for k in range(2):
        assert stats[k]["n_slices_written"] == sim._nt0 - 1 and stats[k]["n_accept"] > 50
        vx, vy = sim._engine.field_interior(k)
        for s in (0, 1, 400, sim._nt0 - 2):
            n = torch_norm(vx[s], vy[s])
            unit = (n > 0.5)
            assert float((n[unit] - 1).abs().max()) < 1e-5
            assert float(n[~unit].abs().max()) == 0.0
            assert 0.0 < float(unit.float().mean()) < 1.0
        full = sim._engine.fields[k][0]
        assert float(full[:, 0, :].abs().max()) == 0 and float(full[:, :, -1].abs().max()) == 0     # padded ring
        # later slices (closer to T) reach less far: the moving fraction shrinks with the slice time
        f0 = float((torch_norm(vx[0], vy[0]) > 0.5).float().mean())
        f1 = float((torch_norm(vx[sim._nt0 - 2], vy[sim._nt0 - 2]) > 0.5).float().mean())
        assert f0 > f1


def torch_norm(a, b):
    return (a.double() ** 2 + b.double() ** 2).sqrt()


def test_c4_forces_lookup_and_bookkeeping_on_samples(c4):
    sim, cfg, stats = c4
    eng = sim._engine
    nt = sim._nt0
    for s in range(25):                                   # a moving crowd (noise-driven), crosses one re-sort
        eng.step(s, noise=None, seed=7, sync=False)
    st = eng.get_state()
    x, y, vx, vy = st["x"], st["y"], st["vx"], st["vy"]
    status = st["status"].astype(bool)
    c = eng.counters.cpu().numpy()
    assert c[2] == status.sum() and c[1] == 0
    rng = np.random.RandomState(3)
    sample = rng.choice(np.nonzero(status)[0], 150, replace=False)
    # pair forces of sampled agents against ALL agents
    f = eng.pair_force().cpu().numpy()
    for i in sample:
        d = np.hypot(x - x[i], y - y[i])
        sel = status & (d < cfg["repulsion_cutoff"])
        sel[i] = False
        js = np.nonzero(sel)[0]
        want = oc.pair_force(np.array((x[i], y[i])), np.array((vx[i], vy[i])), sim._v_des[i],
                             np.column_stack([x[js], y[js]]), np.column_stack([vx[js], vy[js]]), cfg).sum(axis=0)
        assert np.abs(f[i] - want).max() <= 5e-5 * max(np.abs(want).max(), 1e-3)
    # nearest wall cell: exact index wherever the search ran; skipped agents really are beyond their radius
    idx, fw = eng.wall_force()
    idx = idx.cpu().numpy()
    wms = [oc.WallMap(sim._X0, sim._Y0, sim.Vs[k]) for k in sim._key_names]
    for i in sample[:60]:
        want = wms[sim._key_of[i]].query(x[i], y[i])
        if idx[i] >= 0:
            assert idx[i] == want
        else:
            r, cc = divmod(want, sim.Nx)
            dist = np.hypot(sim._X0[cc] - x[i], sim._Y0[r] - y[i])
            s_i = np.hypot(vx[i], vy[i])
            a = cfg["b_min"] + cfg["tau_a"] * s_i
            assert dist > a * a + cfg["eta_walls"] * (1 + s_i) * 33.0
    # lookup against a host copy of the slice
    step = 25
    des, err = eng.lookup(step)
    des = des.cpu().numpy()
    assert err == 0
    for k in range(2):
        fx, fy = (t[step].double().cpu().numpy() for t in eng.field_interior(k))
        ii = [i for i in sample if sim._key_of[i] == k]
        i0, i1, j0, j1, bad = oc.lookup_indices(x[ii], y[ii], sim.room_length, sim.room_height, sim.dx, sim.dy, sim.Nx,
                                                sim.Ny)
        assert not bad.any()
        assert np.allclose(des[ii, 0], sim._v_des[ii] * 0.5 * (fx[i0, j0] + fx[i1, j1]), atol=1e-12)
        assert np.allclose(des[ii, 1], sim._v_des[ii] * 0.5 * (fy[i0, j0] + fy[i1, j1]), atol=1e-12)


def test_t20_fp32_range_management_vs_oracle():
    """T = 20 s: phi spans e^100 ~ 2.7e43 (beyond float32); the solve carries phi * 2^-e.  room_test (200 x 200),
    fp32 direction field against the oracle's fp64 solve: angle <= 1e-2 rad for 99 % of the cells whose |v| is
    not within 10 % of `lim` (max 6e-2, at the front), unit/zero flips on <= 0.5 % of the cells, no NaN."""
    import copy
    from optimal_crowds_amd import room_library
    from optimal_crowds_amd.engine import Engine
    cfg = copy.deepcopy(oc.DEFAULT_CONFIG)
    room = room_library.room_test()
    Nx, Ny = oc.grid_dims(room["room_length"], room["room_height"], cfg["grid_step"])
    X0, Y0 = oc.axes(room["room_length"], room["room_height"], Nx, Ny)
    X, Y = np.meshgrid(X0, Y0)
    V = oc.finalize_potential(oc.create_potential(room, ["door"], X, Y), cfg)
    T = 20.0
    nt = round(T / cfg["dt"])
    eng = Engine(cfg, room["room_length"], room["room_height"], Nx, Ny, X0, Y0, 1, 16, dtype="float32")
    eng.set_potential(0, V, np.zeros((0, 4)))
    eng.alloc_field(0, nt - 1)
    st, _ = eng.solve(0, T, nt)
    assert st["n_rescale"] >= 1 and st["final_exponent"] > 60
    opt = oc.Optimals(V.copy(), T, room["room_length"], room["room_height"], cfg)
    keep = {}
    opt.compute_optimal_velocity(0.0, np.zeros_like(V), keep_phi=keep)
    assert abs(st["n_accept"] - opt.stats["n_accept"]) <= 5
    hp = cfg["hjb_params"]
    vx, vy = (t.double().cpu().numpy() for t in eng.field_interior(0))
    assert not np.isnan(vx).any() and not np.isnan(vy).any()
    for s in (0, 3, 250, 600, 950, nt - 2):
        gx, gy = opt.vx_opt[s], opt.vy_opt[s]
        # |v| of the oracle: the SURVEY 8d tolerance applies to cells at least 10 % away from the threshold `lim`
        ph = keep[nt - 1 - s].reshape(Ny, Nx)
        p = ph * (ph > oc.LIM) + oc.LIM * (ph < oc.LIM)
        nrm = np.hypot((p[1:-1, 2:] - p[1:-1, :-2]) / (2 * cfg["grid_step"]),
                       (p[2:, 1:-1] - p[:-2, 1:-1]) / (2 * cfg["grid_step"])) / (hp["mu"] * p[1:-1, 1:-1])
        clear = np.abs(nrm - oc.LIM) > 0.1 * oc.LIM
        both = (np.hypot(gx, gy) > 0.5) & (np.hypot(vx[s], vy[s]) > 0.5)
        ang = np.abs(np.arctan2(gx * vy[s] - gy * vx[s], gx * vx[s] + gy * vy[s]))
        # The fp32 solve takes slightly different adaptive steps (386 vs 383 accepted), so phi differs from the oracle
        # by up to ~1e-3 relative (the solver's own rtol; median 2e-6).  At the advancing front, where phi is only
        # 0.3 % above its far-field value, the direction is a difference of such values: errors of a few 1e-2 rad in
        # a handful of cells are inherent to rtol = 1e-3, not to float32 (tests/diagnostics/dbg_t20.py prints the evidence).
        a_clear = ang[both & clear]
        assert a_clear.size == 0 or (np.percentile(a_clear, 99) <= 1e-2 and a_clear.max() <= 6e-2)
        flip = (np.hypot(gx, gy) > 0.5) != (np.hypot(vx[s], vy[s]) > 0.5)
        assert flip.mean() <= 5e-3          # the front shifts by less than a cell
    eng.close()


from hjb_compare import assert_steps_acceptable, clear_cells, describe, direction_errors, phi_errors, record_margin      # noqa: E402


def test_c4_value_grid_and_direction_field_vs_c_oracle(c4):
    """The headline grid itself (2048 x 2048, T = 20, key 0) against the C oracle's complete float64 solve, at three
    t_eval points (the last slice, t = 8 s, t = 0):
      free run    both solvers adapt on their own: accepted-step counts within 5;
      forced run  the oracle takes the GPU's accepted step sequence (every one of those steps passes the oracle's own
                  error test), so what remains is float32 arithmetic alone.
    Stated float64 -> float32 tolerances at this size (measured values in brackets; tests/hjb_compare.py explains why phi
    carries a checkerboard ripple that the direction field does not see):
      phi        2x2-block averages within 1e-2 [6.8e-3], pointwise within 1.5e-1 [9.8e-2, next to a door at t = 8 s];
      direction  angle <= 1e-2 rad on at least 95 % of the cells at least 10 % away from the unit-or-zero threshold
                 [97.1 % at t = 8 s, all of them at t = 0 and at the last slice], never beyond 0.085 rad [5.6e-2: where the
                 ripple's amplitude changes from cell to cell, next to the doors]; unit/zero flips on <= 0.1 % of the cells.
    The float64 mode of the same kernels, on the same grid, reproduces the oracle to 1e-5 (next test): the float32
    deviation is rounding noise amplified by the integrator, not a difference of algorithm."""
    from oracle import oc_oracle_c as occ
    sim, cfg, stats = c4
    eng = sim._engine
    hp = cfg["hjb_params"]
    nt = sim._nt0
    slices = [nt - 2, 400, 0]                                   # field slices s  <->  t_eval index k = nt - 1 - s
    keep = [nt - 1 - s for s in slices]
    st, phi_keep = eng.solve(0, sim.T, nt, m=None, keep=keep, want_steps=True)
    got_phi = phi_keep.cpu().numpy().reshape(len(keep), -1)
    P = occ.params(cfg, sim.room_length, sim.room_height, sim.Nx, sim.Ny)
    V = np.ascontiguousarray(sim.Vs[sim._key_names[0]])
    zero = np.zeros_like(V)
    free_phi, fst = occ.rk45_solve(P, V, zero, sim.T, nt, keep=keep)
    assert abs(fst["n_accept"] - st["n_accept"]) <= 5
    forced_phi, zst = occ.rk45_solve(P, V, zero, sim.T, nt, keep=keep, forced=st["t_steps"])
    assert zst["n_accept"] == st["n_accept"]
    assert_steps_acceptable(zst["err"])
    vx, vy = eng.field_interior(0)
    for q, s in enumerate(slices):
        for ref, name in ((free_phi[q], "free"), (forced_phi[q], "forced")):
            point, avg = phi_errors(got_phi[q], ref, V, sim.Ny, sim.Nx)
            print("C4 slice %d vs %s run: phi pointwise %.3e averaged %.3e" % (s, name, point, avg))
            # (bounds = 1.5 x the largest value on record, profiles/r04_parity_margins.json: 6.8e-3 / 9.8e-2 in round 3's step
            #  sequence, 1.4e-3 / 9.7e-3 in round 4's -- the ripple's amplitude is chaotic in the step sequence)
            assert avg <= 1.0e-2 and point <= 1.5e-1, (s, point, avg)
        hx, hy = vx[s].double().cpu().numpy(), vy[s].double().cpu().numpy()
        gx, gy = occ.vels(P, forced_phi[q])
        a, flips = direction_errors(gx, gy, hx, hy, clear_cells(forced_phi[q], sim.Ny, sim.Nx, cfg["grid_step"], hp["mu"]))
        print("C4 slice %d: angle %s; flips %.2e" % (s, describe(a), flips))
        point_f, avg_f = phi_errors(got_phi[q], forced_phi[q], V, sim.Ny, sim.Nx)
        record_margin("C4 metro_station 2048x2048", s, point_f, avg_f, a, flips, err=zst["err"])
        assert a.size > (1000 if s == 0 else 100)                 # (the last slice moves next to the doors only)
        assert (a > 1e-2).mean() <= 4.5e-2 and a.max() <= 0.085        # (on record: 2.9 % / 5.6e-2 rad in round 3, 0 / 3.3e-3 in round 4)
        assert flips <= 1e-3


@pytest.mark.parametrize("dtype", ["float64", "mixed"])
def test_c4_float64_mode_reproduces_the_oracle_at_full_size(c4, dtype):
    """The same 2048 x 2048 solve on a float64 state (same kernels, templated) against the C oracle on the GPU's step
    sequence: phi within 1e-5 pointwise at t = 0 [8.4e-7: two float64 roundings, amplified by the marginally stable
    mode], direction within 1e-5 rad.  'float64' = state and field; 'mixed' = float64 state, float32 direction field (what
    a user who needs the reference's field at this size would run: no float32 ripple, no doubled field memory).  Ten
    output slices only (nt = 11 over the full T = 20)."""
    import torch
    from oracle import oc_oracle_c as occ
    from optimal_crowds_amd.engine import Engine
    sim, cfg, stats = c4
    hp = cfg["hjb_params"]
    V = np.ascontiguousarray(sim.Vs[sim._key_names[0]])
    eng = Engine(cfg, sim.room_length, sim.room_height, sim.Nx, sim.Ny, sim._X0, sim._Y0, 1, 16, dtype=dtype)
    eng.set_potential(0, V, np.zeros((0, 4)))
    nt = 11
    eng.alloc_field(0, nt - 1)
    keep = [nt - 1, 5]
    st, phi_keep = eng.solve(0, sim.T, nt, keep=keep, want_steps=True)
    got = phi_keep.cpu().numpy().reshape(2, -1)
    P = occ.params(cfg, sim.room_length, sim.room_height, sim.Nx, sim.Ny)
    want, zst = occ.rk45_solve(P, V, np.zeros_like(V), sim.T, nt, keep=keep, forced=st["t_steps"])
    assert zst["n_accept"] == st["n_accept"]
    assert_steps_acceptable(zst["err"])
    vx, vy = eng.field_interior(0)
    assert vx.dtype == (torch.float32 if dtype == "mixed" else torch.float64)
    for q, k in enumerate(keep):
        point, avg = phi_errors(got[q], want[q], V, sim.Ny, sim.Nx)
        assert point <= 1e-5, (k, point, avg)
        s = nt - 1 - k
        gx, gy = occ.vels(P, want[q])
        a, flips = direction_errors(gx, gy, vx[s].double().cpu().numpy(), vy[s].double().cpu().numpy(),
                                    clear_cells(want[q], sim.Ny, sim.Nx, cfg["grid_step"], hp["mu"]))
        assert a.size > 1000 and a.max() <= 1e-5 and flips <= 1e-5
    eng.close()


def test_t20_fp32_error_alone_on_the_gpu_step_sequence():
    """The T = 20 solve of room_test with solver noise taken out: the C oracle (float64) is driven along the step
    sequence the float32 GPU solve accepted (each of those steps passes the oracle's own error test), so the two differ
    by float32 arithmetic only.  Then SURVEY 8d's tolerances hold without loosening: angle <= 1e-2 rad on EVERY cell at
    least 10 % away from the unit-or-zero threshold, phi within 5e-3 on block averages (1e-2 pointwise: the checkerboard
    mode, tests/hjb_compare.py), at six slices from t = 0 to t = T."""
    import copy
    from oracle import oc_oracle_c as occ
    from optimal_crowds_amd import room_library
    from optimal_crowds_amd.engine import Engine
    cfg = copy.deepcopy(oc.DEFAULT_CONFIG)
    room = room_library.room_test()
    Nx, Ny = oc.grid_dims(room["room_length"], room["room_height"], cfg["grid_step"])
    X0, Y0 = oc.axes(room["room_length"], room["room_height"], Nx, Ny)
    X, Y = np.meshgrid(X0, Y0)
    V = oc.finalize_potential(oc.create_potential(room, ["door"], X, Y), cfg)
    T = 20.0
    nt = round(T / cfg["dt"])
    slices = [0, 3, 250, 600, 950, nt - 2]
    keep = [nt - 1 - s for s in slices]
    eng = Engine(cfg, room["room_length"], room["room_height"], Nx, Ny, X0, Y0, 1, 16, dtype="float32")
    eng.set_potential(0, V, np.zeros((0, 4)))
    eng.alloc_field(0, nt - 1)
    st, phi_keep = eng.solve(0, T, nt, keep=keep, want_steps=True)
    P = occ.params(cfg, room["room_length"], room["room_height"], Nx, Ny)
    want, zst = occ.rk45_solve(P, V, np.zeros_like(V), T, nt, keep=keep, forced=st["t_steps"])
    assert zst["n_accept"] == st["n_accept"]
    assert_steps_acceptable(zst["err"])
    got = phi_keep.cpu().numpy().reshape(len(keep), -1)
    vx, vy = (t.double().cpu().numpy() for t in eng.field_interior(0))
    for q, s in enumerate(slices):
        point, avg = phi_errors(got[q], want[q], V, Ny, Nx)
        assert avg <= 5e-3 and point <= 1e-2, (s, point, avg)
        gx, gy = occ.vels(P, want[q])
        a, flips = direction_errors(gx, gy, vx[s], vy[s], clear_cells(want[q], Ny, Nx, cfg["grid_step"],
                                                                       cfg["hjb_params"]["mu"]))
        assert a.size == 0 or a.max() <= 1e-2
        assert flips <= 1e-3
    eng.close()


def test_c4_lookups_through_the_ring_equal_lookups_in_the_slices(c4):
    """At the headline size: the desired velocities of all 100 000 agents (both keys) looked up through the interpolant ring
    -- the product path: vels evaluated at the one or two cells a lookup reads -- are the bit patterns found in the slices a
    slice-writing solve (`field='slices'`: oc_hjb_solve, the reference's fill loop optimals.py:200-204) leaves, at early,
    middle and late steps, and so are slices materialised from the ring (oc_hjb_ring_slices)."""
    import bench
    import torch
    sim, cfg, stats = c4
    eng = sim._engine
    assert eng.fields[0].is_ring and eng.field_mode == "ring"
    ref, _ = bench.build_sim("metro_station", field="slices")          # same seed: the same crowd
    try:
        rstats = bench.solve_round(ref)
        for a, b in zip(stats, rstats):
            assert (a["n_attempt"], a["n_accept"], a["nfev"], a["final_exponent"]) == (b["n_attempt"], b["n_accept"], b["nfev"],
                                                                                       b["final_exponent"])
        st0 = eng.get_state()           # (earlier tests of this module have stepped the fixture's crowd: the same state here)
        ref._engine.set_agents(st0["x"], st0["y"], st0["vx"], st0["vy"], sim._v_des, sim._key_of, status=st0["status"])
        moving = 0
        for step in (0, 1, 37, 250, 600, 997, 998):
            d0, e0 = eng.lookup(step)
            d1, e1 = ref._engine.lookup(step)
            assert e0 == e1
            assert torch.equal(d0, d1), step
            moving += int((d0.abs().sum(dim=1) > 0).sum())
        assert moving > 1000            # (most of this crowd stands in the far field, whose direction is zero: probes below)
        # probe positions where the field lives: around each key's exits
        rng = np.random.RandomState(4)
        for k, name in enumerate(sim._key_names):
            rect = np.array([sim._room["targets"][t] for t in sim._possible[name]])[0]
            n = 20000
            xs = np.clip(rect[0] + rng.uniform(-25, 25, n), 0.5, sim.room_length - 0.5)
            ys = np.clip(rect[1] + rng.uniform(-25, 25, n), 0.5, sim.room_height - 0.5)
            nz = 0
            for step in (0, 300, 900):
                p0, pe0 = eng.lookup_points(k, xs, ys, step)
                p1, pe1 = ref._engine.lookup_points(k, xs, ys, step)
                assert pe0 == pe1 and np.array_equal(p0, p1)
                nz += int((np.abs(p0).sum(axis=1) > 0).sum())
            assert nz > 3000                                            # (probe-steps that see a direction)
        ref._engine.set_agents(st0["x"], st0["y"], st0["vx"], st0["vy"], sim._v_des, sim._key_of, status=st0["status"])
        eng.set_agents(st0["x"], st0["y"], st0["vx"], st0["vy"], sim._v_des, sim._key_of, status=st0["status"])
        for k in range(2):
            for s in (0, 400, 998):
                vx, vy = eng.ring_slices(k, s, s + 1)
                assert torch.equal(vx[0], ref._engine.fields[k][0][s]) and torch.equal(vy[0], ref._engine.fields[k][1][s])
    finally:
        ref._engine.close()
        ref._engine.fields = []
        del ref
        torch.cuda.empty_cache()
