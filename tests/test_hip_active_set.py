"""The exact active set of the HJB solve (csrc/hjb.hip, k_stageV / k_out): tiles on which phi is one constant power of two,
K0 = 0 and coef = 0 (the far field of an m = 0 solve: optimals.py:144-164 evaluates to exactly 0 there) are neither
loaded nor computed.  Everything here compares the default (active set on) with `hjb_active = 0` (every tile of every
attempt computed) on the SAME engine and buffers: the step sequence, every statistic and every value of the direction
field must be identical -- not close: equal.  Cases: small rooms where the front crosses the whole grid, a density term
(coef != 0 in blobs: active from the first attempt there), T = 20 in float32 (power-of-two rescales between the buffers'
scales: the store-skipping rule), float64 / mixed states, a second solve into buffers a previous solve has dirtied, and
the headline grid (C4 key 0, where 96 % of the tile-attempts are trivial)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_hjb_rows import _density_like, _new_engine, _setup      # noqa: E402


def _solve(eng, T, nt, m, active, keep=None):
    import torch
    eng.set_tuning("hjb_active", 2 if active else 0)           # 2: at any grid size (1 = only from 96 stage tiles on)
    mt = None if m is None else torch.as_tensor(m)
    st, phi = eng.solve(0, T, nt, m=mt, keep=keep, want_steps=True)
    vx, vy = eng.fields[0][0], eng.fields[0][1]
    return st, vx.clone(), vy.clone(), (None if phi is None else phi.clone())


def _same(a, b):
    import torch
    return bool(torch.equal(a, b))


@pytest.mark.parametrize("room_name,gs,dtype,T,nt,with_m", [
    ("gold_a", 0.05, "float32", 3.0, 150, False),
    ("gold_a", 0.05, "float32", 3.0, 150, True),
    ("gold_b", 0.05, "float64", 2.0, 100, False),
    ("gold_b", 0.05, "mixed", 2.0, 100, True),
    ("exit_opposite", 0.1, "float32", 20.0, 1000, False),       # 512 x 512: the front reaches most of the room; rescales
    ("slalom", 0.1, "float32", 20.0, 1000, False),              # 1024 x 1024: walls shape the front
])
@pytest.mark.parametrize("field_mode", ["slices", "ring"])
def test_active_set_equals_dense_sweep(room_name, gs, dtype, T, nt, with_m, field_mode):
    """field_mode 'slices': the slice writer's own use of the flags (zero-filled tiles; the poison below catches a tile it
    forgot); 'ring': the stage launch's flags in the ring's slots (the field is materialised from the ring)."""
    setup = _setup(room_name, gs)
    cfg, room, Nx, Ny, X0, Y0, V = setup
    m = _density_like(Ny, Nx) if with_m else None
    eng = _new_engine(*setup, dtype, field_mode)
    eng.alloc_field(0, nt - 1)
    st_d, vx_d, vy_d, _ = _solve(eng, T, nt, m, active=False)
    # poison the field so that a tile the active-set path forgot to write cannot pass for a zero
    eng.fields[0][0].fill_(7.0)
    eng.fields[0][1].fill_(7.0)
    st_a, vx_a, vy_a, _ = _solve(eng, T, nt, m, active=True)
    for k in ("nfev", "n_attempt", "n_accept", "n_rescale", "n_slices_written", "final_exponent"):
        assert st_a[k] == st_d[k], (k, st_a[k], st_d[k])
    assert np.array_equal(st_a["t_steps"], st_d["t_steps"])
    assert st_a["h_min"] == st_d["h_min"] and st_a["h_max"] == st_d["h_max"]
    assert _same(vx_a, vx_d) and _same(vy_a, vy_d)
    if dtype == "float32" and T >= 20.0:
        assert st_a["n_rescale"] >= 1                               # the rescale path was exercised
    # and again into the buffers the first active solve left behind, with a different problem first (dirty state arrays)
    other = _density_like(Ny, Nx, seed=11)
    _solve(eng, T, nt, other, active=True)
    st_b, vx_b, vy_b, _ = _solve(eng, T, nt, m, active=True)
    assert st_b["n_attempt"] == st_d["n_attempt"] and _same(vx_b, vx_d) and _same(vy_b, vy_d)
    eng.close()


def test_active_set_keeps_the_audit_output():
    """`keep` (phi at chosen t_eval points) takes k_out's full path; the stage launches still skip trivial tiles."""
    setup = _setup("gold_a", 0.05)
    eng = _new_engine(*setup, "float32")
    T, nt, keep = 3.0, 150, [1, 40, 149]
    eng.alloc_field(0, nt - 1)
    st_d, vx_d, vy_d, phi_d = _solve(eng, T, nt, None, active=False, keep=keep)
    st_a, vx_a, vy_a, phi_a = _solve(eng, T, nt, None, active=True, keep=keep)
    assert st_a["n_attempt"] == st_d["n_attempt"]
    assert _same(phi_a, phi_d) and _same(vx_a, vx_d) and _same(vy_a, vy_d)
    far = phi_a[:, 2:6, 2:6]                        # a corner the front has not reached at t = T - dt: exactly 1
    assert float(far[0].min()) == 1.0 and float(far[0].max()) == 1.0
    eng.close()


def test_active_set_at_the_headline_size():
    """C4 key 0 (2048 x 2048, T = 20, nt = 1000, m = 0): equal fields, equal steps, and the share of computed tiles."""
    import torch
    import bench
    sim, cfg = bench.build_sim("metro_station")
    eng = sim._engine
    eng.set_profiling(1)
    out = {}
    for active in (False, True):
        eng.set_tuning("hjb_active", 1 if active else 0)
        eng.reset_kernel_stats()
        st, _ = eng.solve(0, sim.T, sim._nt0, m=None, want_steps=True)
        nl, ms, by = eng.kernel_stats("hjb_stage")
        out[active] = (st, eng.fields[0][0].clone(), by / max(nl, 1), ms / max(nl, 1))
        if not active:
            eng.fields[0][0].fill_(7.0)
    (st_d, vx_d, by_d, ms_d), (st_a, vx_a, by_a, ms_a) = out[False], out[True]
    assert st_a["n_attempt"] == st_d["n_attempt"] and np.array_equal(st_a["t_steps"], st_d["t_steps"])
    assert bool(torch.equal(vx_a, vx_d))
    share = by_a / by_d
    print("C4 key 0: computed-tile share %.3f, stage launch %.1f us (dense %.1f us)" % (share, 1e3 * ms_a, 1e3 * ms_d))
    assert 0.0005 < share < 0.25
    del vx_a, vx_d, out
    eng.close()
    eng.fields = []
    torch.cuda.empty_cache()
