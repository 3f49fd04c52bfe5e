"""The interpolant ring (oc_hjb_solve_ring / oc_set_field_ring / oc_hjb_ring_slices): a solve that keeps what SciPy's dense
output of each accepted step is made of (rk.py:178-180, 560-575) instead of every slice of vx_opt / vy_opt
(reference optimals.py:200-204), and a lookup (optimals.py:212-250) that evaluates `vels` (optimals.py:168-186) at the one or
two cells it reads.  The bar is equality, not closeness: every test compares with the slice-writing solve (oc_hjb_solve)
on the same engine inputs and asks for identical bit patterns --
  * slices materialised from the ring == the slices oc_hjb_solve writes (all of them, every dtype, with the exact active
    set forced on, with a density term, at T = 20 with power-of-two rescales, after a re-solve with a shorter horizon);
  * lookups through the ring == lookups in the slices, on crowds of probe positions at many steps, default and
    OC_FIX_LOOKUP (bilinear) forms, IndexError-band flags included;
  * a ring that is too small is reported (OC_ERR_CAPACITY) and grown, with the same result;
  * whole runs of the facade: field='ring' and field='slices' leave the same agent states (to the rounding of the pair
    sums' float64 atomics, the only thing that differs between two launches of the same step).
"""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oc_oracle as oc                     # noqa: E402  (room rasterisation for the inputs only)
from optimal_crowds_amd import room_library            # noqa: E402
from test_hjb_rows import _density_like, _setup        # noqa: E402


def _engine(setup, dtype, mode, n_agents=4096):
    from optimal_crowds_amd.engine import Engine
    cfg, room, Nx, Ny, X0, Y0, V = setup
    eng = Engine(cfg, room["room_length"], room["room_height"], Nx, Ny, X0, Y0, 1, n_agents, dtype=dtype, field_mode=mode)
    eng.set_potential(0, V, np.zeros((0, 4)))
    return eng


def _probe_agents(eng, room, n, seed):
    rng = np.random.RandomState(seed)
    L, H = room["room_length"], room["room_height"]
    gs = L / (eng.Nx - 1)
    x = rng.uniform(2.5 * gs, L - 3.5 * gs, n)
    y = rng.uniform(2.5 * gs, H - 3.5 * gs, n)
    # a few on the special branches of optimals.py:234-245: x <= dx, x >= L - dx, and exactly on grid lines
    x[:8] = [0.3 * gs, gs, 1.0000001 * gs, L - 0.5 * gs, L - gs, 7 * gs, 0.0, L - 2.2 * gs]
    y[8:16] = [0.3 * gs, gs, 1.0000001 * gs, H - 0.5 * gs, H - gs, 7 * gs, 0.0, H - 2.2 * gs]
    eng.set_agents(x, y, np.zeros(n), np.zeros(n), rng.normal(1.34, 0.26, n), np.zeros(n, dtype=np.int32))


def _lookups(eng, steps):
    out = []
    for s in steps:
        des, err = eng.lookup(int(s))
        out.append((des.cpu().numpy().copy(), err))
    return out


CASES = [
    # room, grid_step, dtype, T, nt, density term, active set forced at this size
    ("gold_a", 0.05, "float32", 3.0, 150, False, 0),
    ("gold_a", 0.05, "float32", 3.0, 150, True, 2),
    ("gold_b", 0.05, "float64", 2.0, 100, False, 2),
    ("gold_b", 0.05, "mixed", 2.0, 100, True, 2),
    ("room_test", 0.05, "float32", 20.0, 1000, False, 2),       # 200 x 200, 383 accepted steps, rescales
    ("exit_opposite", 0.1, "float32", 20.0, 1000, False, 2),    # 512 x 512: the front crosses the room
    ("slalom", 0.1, "float32", 20.0, 1000, False, 1),           # 1024 x 1024: the default rule switches the active set on
]


@pytest.mark.parametrize("room_name,gs,dtype,T,nt,with_m,active", CASES)
def test_ring_equals_slices(room_name, gs, dtype, T, nt, with_m, active):
    import torch
    setup = _setup(room_name, gs)
    cfg, room, Nx, Ny, X0, Y0, V = setup
    m = torch.as_tensor(_density_like(Ny, Nx)) if with_m else None
    res = {}
    for mode in ("slices", "ring"):
        eng = _engine(setup, dtype, mode)
        eng.set_tuning("hjb_active", active)
        eng.alloc_field(0, nt - 1)
        keep = [0, 1, nt // 3, nt - 1]
        st, phi = eng.solve(0, T, nt, m=m, keep=keep, want_steps=True)
        if mode == "ring":
            assert eng.fields[0].is_ring and eng.fields[0].vx is None          # nothing of the field is in memory
        _probe_agents(eng, room, 3000, 5)
        steps = [0, 1, 2, nt // 7, nt // 2, nt - 3, nt - 2, nt - 1, nt + 5]
        look = _lookups(eng, steps)
        vx, vy = eng.fields[0][0], eng.fields[0][1]             # ring: materialised here (oc_hjb_ring_slices)
        res[mode] = (st, phi.cpu(), look, vx.cpu(), vy.cpu())
        if mode == "ring":
            # a sub-range lands where it was asked for
            a, b = nt // 5, nt // 5 + 7
            sx, sy = eng.ring_slices(0, a, b)
            assert torch.equal(sx.cpu(), res["ring"][3][a:b]) and torch.equal(sy.cpu(), res["ring"][4][a:b])
        eng.close()
        del eng
    (st_s, phi_s, look_s, vx_s, vy_s), (st_r, phi_r, look_r, vx_r, vy_r) = res["slices"], res["ring"]
    for k in ("nfev", "n_attempt", "n_accept", "n_rescale", "final_exponent"):
        assert st_r[k] == st_s[k], (k, st_r[k], st_s[k])
    assert np.array_equal(st_r["t_steps"], st_s["t_steps"])
    assert torch.equal(phi_r, phi_s)                            # the audit output comes from the ring too
    assert torch.equal(vx_r, vx_s) and torch.equal(vy_r, vy_s)  # every slice, every cell, every bit
    assert float(vx_s.abs().max()) > 0.5                        # (not a comparison of zeros)
    for (d_s, e_s), (d_r, e_r) in zip(look_s, look_r):
        assert e_s == e_r
        assert np.array_equal(d_s, d_r)
    assert any(np.abs(d).max() > 0 for d, _ in look_s)
    if dtype == "float32" and T >= 20.0:
        assert st_s["n_rescale"] >= 1


def test_ring_bilinear_lookup_and_shorter_horizon():
    """OC_FIX_LOOKUP through the ring; then a re-solve with a shorter horizon (reference optimals.py:139-142: nt shrinks,
    the span does not -- quirk Q9): slices at and beyond nt - 1 are never read (lookup returns 0)."""
    setup = _setup("gold_a", 0.05)
    cfg, room, Nx, Ny, X0, Y0, V = setup
    T, nt0, nt1 = 3.0, 150, 97
    res = {}
    for mode in ("slices", "ring"):
        eng = _engine(setup, "float32", mode)
        eng.alloc_field(0, nt0 - 1)
        eng.solve(0, T, nt0)
        _probe_agents(eng, room, 2000, 9)
        steps = [0, 3, 50, 95, 96, 97, 120, 148, 149]
        out = [_lookups(eng, steps)]
        eng.set_fix_flags(["lookup"])
        out.append(_lookups(eng, steps))
        eng.set_fix_flags([])
        eng.solve(0, T, nt1)                                     # the recompute
        out.append(_lookups(eng, steps))
        res[mode] = out
        if mode == "ring":
            for d, _ in out[2][4:]:                              # steps >= nt1 - 1: zero desired velocity
                assert not d.any()
            vx = eng.fields[0][0]
            assert vx.shape[0] == nt0 - 1 and not bool(vx[nt1 - 1:].any())
        eng.close()
    for part_s, part_r in zip(res["slices"], res["ring"]):
        for (d_s, e_s), (d_r, e_r) in zip(part_s, part_r):
            assert e_s == e_r and np.array_equal(d_s, d_r)


def test_ring_too_small_is_reported_and_grown():
    import ctypes as C
    import torch
    from optimal_crowds_amd import _lib
    setup = _setup("gold_a", 0.05)
    T, nt = 3.0, 150
    eng = _engine(setup, "float32", "ring")
    eng.alloc_field(0, nt - 1)
    st, _ = eng.solve(0, T, nt)
    want = eng.fields[0][0].clone()
    assert eng.fields[0].ring["slots"] >= st["n_accept"]
    # the C-ABI reports it ...
    ring = eng.make_ring(0, 5)
    eng.make_coef(0, None)
    stats = _lib.OcHjbStats()
    rc = eng.lib.oc_hjb_solve_ring(eng.h, eng.solve_dtype, C.c_void_p(eng.coef.data_ptr()), T, nt, C.byref(ring["struct"]),
                                   None, 0, None, None, 0, C.byref(stats), eng.stream())
    assert rc == _lib.OC_ERR_CAPACITY and b"ring is too small" in eng.lib.oc_last_error()
    # ... and the engine grows the ring until the solve fits: 5 -> 10 -> 20 -> ... slots
    st2, _ = eng.solve(0, T, nt)
    assert st2["n_accept"] == st["n_accept"] and eng.fields[0].ring["slots"] >= st["n_accept"]
    assert eng.fields[0].ring["slots"] == 5 * 2 ** int(np.ceil(np.log2(st["n_accept"] / 5)))
    assert torch.equal(eng.fields[0][0], want)
    eng.close()


def test_facade_ring_and_slices_runs_are_identical():
    """simulations.simulation(...).run() with field='ring' (default) and field='slices': same np.random stream, same
    agent states at the end (1e-10 m), same exits -- on a room with two keys and a recompute=True run (density term, shrinking
    horizon every 50 steps)."""
    from optimal_crowds_amd import simulations
    out = {}
    for mode in ("ring", "slices"):
        for room, T, rec in (("gold_a", 4.0, False), ("gold_b", 4.0, True)):
            np.random.seed(12)
            sim = simulations.simulation(room, T, recompute=rec, field=mode)
            sim.run()
            st = sim._engine.get_state()
            out[(mode, room)] = (st["x"].copy(), st["y"].copy(), st["vx"].copy(), st["status"].copy(), sim.simu_step, sim.inside)
            if mode == "ring":
                key = next(iter(sim.targets))
                vx = sim.targets[key].vx_opt
                assert vx.shape == (round(T / sim.dt) - 1, sim.Ny - 2, sim.Nx - 2)
                a = vx[3]
                assert a.shape == (sim.Ny - 2, sim.Nx - 2) and a.dtype == np.float64
                assert np.array_equal(vx[2:5][1], a) and np.array_equal(vx[3, 5:9, 2], a[5:9, 2])
            sim._engine.close()
    # (the lookups are bit-identical -- the tests above; the pair sums of a step are float64 LDS atomics whose order is not
    # fixed from launch to launch, so two RUNS agree to rounding, 1e-16 per step, whatever the field mode)
    for room in ("gold_a", "gold_b"):
        r, s = out[("ring", room)], out[("slices", room)]
        assert r[4] == s[4] and r[5] == s[5]
        assert np.array_equal(r[3], s[3])
        for a, b in zip(r[:3], s[:3]):
            assert np.allclose(a, b, rtol=0, atol=1e-10)


def test_ring_equals_slices_at_config2_with_agents():
    """C2 (exit_opposite 512 x 512, 5 000 agents, 2 keys): 120 synchronous steps from the same start, device noise --
    the same positions (1e-10 m) with the ring and with slices."""
    from optimal_crowds_amd import simulations
    fin = {}
    for mode in ("ring", "slices"):
        np.random.seed(20240807)
        sim = simulations.simulation("exit_opposite", 20.0, field=mode, noise="device", record=False,
                                     config=_cfg(0.1))
        sim._solve_all()
        for s in range(120):
            sim._engine.step(s, noise=None, seed=7, sync=False)
        st = sim._engine.get_state()
        fin[mode] = (st["x"].copy(), st["y"].copy(), st["vx"].copy(), st["vy"].copy())
        sim._engine.close()
    for a, b in zip(fin["ring"], fin["slices"]):
        assert np.allclose(a, b, rtol=0, atol=1e-10)           # (equal up to the order of float64 atomic additions)
    assert np.abs(fin["ring"][2]).max() > 0.1


def _cfg(gs):
    cfg = copy.deepcopy(oc.DEFAULT_CONFIG)
    cfg["grid_step"] = gs
    return cfg


@pytest.mark.parametrize("dtype,with_m", [("float32", False), ("mixed", True)])
def test_concurrent_key_solves_equal_one_after_the_other(dtype, with_m):
    """Engine.solve_keys: three keys of one room (three different target sets -> three potentials) solved concurrently --
    each on its own handle, stream and host thread -- leave the rings, the statistics and the lookups that solving them
    one after the other leaves (the reference's order: simulations.py:419-420), bit for bit."""
    import torch
    from optimal_crowds_amd.engine import Engine
    cfg, room, Nx, Ny, X0, Y0, _ = _setup("gold_b", 0.05)
    X, Y = np.meshgrid(X0, Y0)
    names = list(room["targets"].keys())
    keysets = [[names[0]], [names[-1]], names[:2]][:3]
    Vs = [oc.finalize_potential(oc.create_potential(room, ks, X, Y), cfg) for ks in keysets]
    T, nt = 3.0, 150
    m = torch.as_tensor(_density_like(Ny, Nx)) if with_m else None
    res = []
    for overlap in (False, True):
        eng = Engine(cfg, room["room_length"], room["room_height"], Nx, Ny, X0, Y0, len(Vs), 4096, dtype=dtype)
        eng.overlap_solves = overlap
        eng.set_tuning("hjb_active", 2)
        for k, V in enumerate(Vs):
            eng.set_potential(k, V, np.zeros((0, 4)))
            eng.alloc_field(k, nt - 1)
        stats = eng.solve_keys([(k, T, nt, m) for k in range(len(Vs))])
        assert len(eng._lanes) == (len(Vs) - 1 if overlap else 0)
        # a second round on the same engine (re-solve into used rings, shorter horizon for one key)
        stats2 = eng.solve_keys([(k, T, nt - (25 if k == 1 else 0), m) for k in range(len(Vs))])
        fields = [tuple(t.cpu().numpy() for t in eng.ring_slices(k, 0, nt - 1)) for k in range(len(Vs))]
        rng = np.random.RandomState(5)
        n = 2000
        gs = room["room_length"] / (Nx - 1)
        eng.set_agents(rng.uniform(2.5 * gs, room["room_length"] - 3.5 * gs, n), rng.uniform(2.5 * gs, room["room_height"] - 3.5 * gs, n),
                       np.zeros(n), np.zeros(n), rng.normal(1.34, 0.26, n), (np.arange(n) % len(Vs)).astype(np.int32))
        looks = _lookups(eng, [0, 7, 60, 120, 148])
        res.append((stats, stats2, fields, looks))
        eng.close()
    (sa, sa2, fa, la), (sb, sb2, fb, lb) = res
    for x, y in zip(sa + sa2, sb + sb2):
        for name in ("nfev", "n_attempt", "n_accept", "n_rescale", "h_abs0", "h_min", "h_max", "final_exponent"):
            assert x[name] == y[name], name
    assert len({s["n_attempt"] for s in sa}) > 1            # (the keys' solves really differ)
    for k in range(len(Vs)):
        assert np.array_equal(fa[k][0], fb[k][0]) and np.array_equal(fa[k][1], fb[k][1])
        assert np.abs(fa[k][0]).max() > 0.5
    for (da, ea), (db, eb) in zip(la, lb):
        assert np.array_equal(da, db) and ea == eb


def test_concurrent_key_solves_grow_a_ring_that_is_too_small_in_the_callers_thread():
    """A lane thread of Engine.solve_keys allocates and frees nothing (its tensors would belong to the lane stream's pool while
    the step kernels read them on the caller's stream): a ring that runs out of slots in a lane (OC_ERR_CAPACITY) is grown and
    its key solved again by the CALLER's thread -- same statistics and slices as with rings that were large enough."""
    import torch
    from optimal_crowds_amd.engine import Engine
    cfg, room, Nx, Ny, X0, Y0, _ = _setup("gold_b", 0.05)
    X, Y = np.meshgrid(X0, Y0)
    names = list(room["targets"].keys())
    Vs = [oc.finalize_potential(oc.create_potential(room, ks, X, Y), cfg) for ks in ([names[0]], [names[-1]])]
    T, nt = 3.0, 150
    res = []
    for tiny in (False, True):
        eng = Engine(cfg, room["room_length"], room["room_height"], Nx, Ny, X0, Y0, len(Vs), 4096, dtype="float32")
        for k, V in enumerate(Vs):
            eng.set_potential(k, V, np.zeros((0, 4)))
            eng.alloc_field(k, nt - 1)
            if tiny:
                eng.make_ring(k, 2)                     # (both lanes -- the caller's own and the auxiliary one -- run out)
        stats = eng.solve_keys([(k, T, nt, None) for k in range(len(Vs))])
        if tiny:
            assert all(eng.fields[k].ring["slots"] >= stats[k]["n_accept"] > 2 for k in range(len(Vs)))
        res.append((stats, [tuple(t.cpu().numpy() for t in eng.ring_slices(k, 0, nt - 1)) for k in range(len(Vs))]))
        eng.close()
    (sa, fa), (sb, fb) = res
    for x, y in zip(sa, sb):
        for name in ("nfev", "n_attempt", "n_accept", "h_abs0", "h_min", "h_max"):
            assert x[name] == y[name], name
    for k in range(len(Vs)):
        assert np.array_equal(fa[k][0], fb[k][0]) and np.array_equal(fa[k][1], fb[k][1])


@pytest.mark.parametrize("active", [0, 2])
def test_tail_variants_make_the_same_decisions(active):
    """oc_set_tuning "hjb_tail": 0 = stage launch + k_control (+ k_list), 1 = the attempt's last block is its controller
    while at most 640 blocks take part, 2 = always (two-level tickets beyond 640).  slalom at 1024 x 1024 has 520 stage
    tiles -- under the flat ticket's range -- so the two-level path is taken on metro_station's 2048 x 2048
    grid (BASELINE configs[3]: 1 957 tiles) with every tile computed (active = 0) and, with the active set, in its first, list-less attempt."""
    import torch
    setup = _setup("metro_station", 0.1)
    cfg, room, Nx, Ny, X0, Y0, V = setup
    assert Nx % 4 == 0 and -(-Nx // 112) * -(-Ny // 20) > 640
    T, nt = 3.0, 150
    res = {}
    for tail in (0, 1, 2):
        eng = _engine(setup, "float32", "ring")
        eng.set_tuning("hjb_active", active)
        eng.set_tuning("hjb_tail", tail)
        eng.alloc_field(0, nt - 1)
        st, phi = eng.solve(0, T, nt, keep=[0, nt - 1], want_steps=True)
        _probe_agents(eng, room, 2000, 11)
        look = _lookups(eng, [0, 1, nt // 2, nt - 2])
        res[tail] = (st, phi.cpu(), look)
        eng.close()
        del eng
    st0, phi0, look0 = res[0]
    assert st0["n_accept"] > 10
    for tail in (1, 2):
        st, phi, look = res[tail]
        for k in ("nfev", "n_attempt", "n_accept", "n_rescale", "final_exponent"):
            assert st[k] == st0[k], (tail, k, st[k], st0[k])
        assert np.array_equal(st["t_steps"], st0["t_steps"])
        assert torch.equal(phi, phi0)
        for (d0, e0), (d1, e1) in zip(look0, look):
            assert e0 == e1 and np.array_equal(d0, d1)


@pytest.mark.parametrize("room_name,gs,active,T,nt", [("room_test", 0.05, 0, 20.0, 1000), ("metro_station", 0.1, 2, 6.0, 300),
                                                      ("exit_opposite", 0.1, 2, 20.0, 1000)])
def test_persistent_launch_and_fast_tail_make_the_same_decisions(room_name, gs, active, T, nt):
    """oc_set_tuning "hjb_persist" = 2: the attempts of a ring solve inside ONE launch of a co-resident grid (k_stageV<..,
    PERSIST>: items dealt to the blocks, ticket barrier, the last arrival runs controller and list builder, the others wait for
    the generation word) -- and "hjb_tail_fast" = 0 / 1: the fused tail reading every tile's words or the listed tiles' records.
    All of them leave the ring, the statistics and the lookups of the launch chain, bit for bit; a barrier deadline that
    passes at once (hjb_persist_wait_us = 10) makes the launch give up and the chain finish the solve -- with the same bits.
    room_test: no active set, one tile per block, rescales and rejected attempts over T = 20; metro_station: active set, short
    lists (the fast tail's range); exit_opposite with the active set forced: lists beyond the fast tail's range, blocks
    that carry two tiles."""
    import torch
    setup = _setup(room_name, gs)
    cfg, room, Nx, Ny, X0, Y0, V = setup
    res = {}
    variants = {"chain": dict(hjb_persist=0, hjb_tail_fast=0), "chain+fast": dict(hjb_persist=0, hjb_tail_fast=1),
                "persist": dict(hjb_persist=2, hjb_tail_fast=0), "persist+fast": dict(hjb_persist=2, hjb_tail_fast=1),
                "deadline": dict(hjb_persist=2, hjb_tail_fast=1, hjb_persist_wait_us=10),
                # a sparse attempt's launch sized BELOW its work list does nothing; the host drains and re-enqueues (grid_resyncs)
                "grid too small": dict(hjb_persist=0, hjb_grid_margin=-12), "grid not fitted": dict(hjb_persist=0, hjb_grid_fit=0),
                # "hjb_lead" = 1: the controller at the start of the NEXT launch, in every block (k_stageV<.., LEAD>)
                "lead": dict(hjb_persist=0, hjb_lead=1), "lead, grid too small": dict(hjb_persist=0, hjb_lead=1, hjb_grid_margin=-12),
                "lead, grid not fitted": dict(hjb_persist=0, hjb_lead=1, hjb_grid_fit=0)}
    for name, knobs in variants.items():
        eng = _engine(setup, "float32", "ring")
        eng.set_tuning("hjb_active", active)
        for k, v in knobs.items():
            eng.set_tuning(k, v)
        eng.alloc_field(0, nt - 1)
        st, phi = eng.solve(0, T, nt, keep=[0, nt // 3, nt - 1], want_steps=True)
        _probe_agents(eng, room, 2000, 11)
        look = _lookups(eng, [0, 1, nt // 2, nt - 2])
        res[name] = (st, phi.cpu(), look)
        eng.close()
        del eng
    st0, phi0, look0 = res["chain"]
    assert st0["n_accept"] > 10 and st0["persist_exit"] == 0
    assert res["persist"][0]["persist_exit"] in (1, 2) and res["persist"][0]["persist_attempts"] > 10
    assert res["deadline"][0]["persist_exit"] in (1, 2, 3)
    if room_name == "metro_station":
        assert res["grid too small"][0]["grid_resyncs"] > 0 and res["chain"][0]["grid_resyncs"] == 0
        assert res["lead, grid too small"][0]["grid_resyncs"] > 0
    if active:
        assert res["lead"][0]["lead_attempts"] > 10 and res["chain"][0]["lead_attempts"] == 0
    for name in variants:
        st, phi, look = res[name]
        for k in ("nfev", "n_attempt", "n_accept", "n_rescale", "final_exponent", "h_abs0", "h_min", "h_max"):
            assert st[k] == st0[k], (name, k, st[k], st0[k])
        assert np.array_equal(st["t_steps"], st0["t_steps"]), name
        assert torch.equal(phi, phi0), name
        for (d0, e0), (d1, e1) in zip(look0, look):
            assert e0 == e1 and np.array_equal(d0, d1), name


@pytest.mark.parametrize("room_name,gs,dtype,with_m", [("room_test", 0.05, "float32", False), ("slalom", 0.1, "float32", False),
                                                        ("slalom", 0.1, "mixed", True), ("exit_opposite", 0.1, "float32", False)])
def test_far_field_substitution_changes_nothing(room_name, gs, dtype, with_m):
    """oc_set_tuning "hjb_subst" (ring solves with the exact active set): 1 (default) = a trivial tile's cells are not kept up
    to date in the state buffers -- whoever loads them substitutes the controller's far-field value and K0 = 0 -- so a
    change of scale no longer puts every tile on the list to rewrite the far field; 0 = the buffers hold every tile.  Same
    attempts, same steps, same phi, same lookups: T = 20 s with rescales (float32) and rejected attempts."""
    import torch
    setup = _setup(room_name, gs)
    cfg, room, Nx, Ny, X0, Y0, V = setup
    T, nt = 20.0, 1000
    m = torch.as_tensor(_density_like(Ny, Nx)) if with_m else None
    res = {}
    for subst in (0, 1):
        eng = _engine(setup, dtype, "ring")
        eng.set_tuning("hjb_active", 2)
        eng.set_tuning("hjb_subst", subst)
        eng.alloc_field(0, nt - 1)
        st, phi = eng.solve(0, T, nt, m=m, keep=[0, 1, nt // 2, nt - 1], want_steps=True)
        _probe_agents(eng, room, 3000, 17)
        look = _lookups(eng, [0, 1, 2, nt // 7, nt // 2, nt - 3, nt - 2])
        vx, vy = eng.ring_slices(0, nt // 3, nt // 3 + 5)
        res[subst] = (st, phi.cpu(), look, vx.cpu(), vy.cpu())
        eng.close()
        del eng
    (st0, phi0, look0, vx0, vy0), (st1, phi1, look1, vx1, vy1) = res[0], res[1]
    for k in ("nfev", "n_attempt", "n_accept", "n_rescale", "final_exponent"):
        assert st1[k] == st0[k], (k, st1[k], st0[k])
    assert np.array_equal(st1["t_steps"], st0["t_steps"])
    assert st0["n_attempt"] > st0["n_accept"]                   # rejected attempts on the way
    if dtype == "float32":
        assert st0["n_rescale"] >= 1
    assert torch.equal(phi1, phi0) and torch.equal(vx1, vx0) and torch.equal(vy1, vy0)
    for (d0, e0), (d1, e1) in zip(look0, look1):
        assert e0 == e1 and np.array_equal(d0, d1)
    assert any(np.abs(d).max() > 0 for d, _ in look0)
