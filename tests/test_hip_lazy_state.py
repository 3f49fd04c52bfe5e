"""lazy_state (oc_set_tuning / oc_agents_sync, include/oc_hip.h): a step may leave x / y / vx / vy / agent_time in the library's
slot-ordered snapshot instead of scattering them into the caller's arrays; whoever reads the arrays -- the caller through
oc_agents_sync (Engine.agents[...] does it), the periodic re-sort, oc_density, the diagnostic exports, the sequential step --
finds them current.  The bar: the same states as with the scatter in every step (reference simulations.py:303-326 writes the
agent's own fields in place), to the order of the pair sums' float64 atomics."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.test_hip_parity import _engine                               # noqa: E402
from tests.test_hip_step_shapes import _big, _crowd                     # noqa: E402


def _fresh(lazy):
    eng, room, cfg, X0, Y0 = _engine("exit_opposite", "float32", n_keys=2, max_agents=8192, cfg=_big(), wall_search_radius=0.0)
    eng.set_lazy_state(lazy)
    state = _crowd(eng, room, cfg, X0, Y0, n=6000, T=2.0)
    x, y, vx, vy, vdes, key = state
    # a cluster right in front of the east door: agents LEAVE during the run (their last state is written when they do)
    x[:300] = np.linspace(48.5, 50.0, 300)
    y[:300] = 25.575 + np.linspace(-1.5, 1.5, 300)
    vx[:300], vy[:300], key[:300] = 1.2, 0.0, 0
    eng.set_agents(x, y, vx, vy, vdes, key)
    return eng


def test_lazy_state_equals_the_scatter_in_every_step():
    import torch
    res = {}
    for lazy in (False, True):
        eng = _fresh(lazy)
        assert eng.lazy_state == lazy
        snaps, dens, exits = [], [], 0
        for s in range(75):                                             # (re-sorts at steps 32 and 64)
            out = eng.step(s, noise=None, seed=3)
            exits += out[0]
            if s in (0, 9, 31, 32, 33, 70):
                snaps.append(eng.get_state())
            if s == 40:                                                 # another entry point that reads the arrays
                dens.append(eng.density(0.5, None).double().cpu().numpy())
                dens.append(eng.wall_force()[1].cpu().numpy())
        if lazy:
            # the laziness is real: behind the hook the tensors are older than the snapshot until they are synchronised
            eng.step(75, noise=None, seed=3, sync=False)
            raw = dict.__getitem__(eng.agents, "x").clone()
            cur = eng.agents["x"].clone()
            assert not torch.equal(raw, cur)
            eng.step(76, noise=None, seed=3, sync=False)
            # the sequential step works on the arrays: it must find them current
            perm = torch.randperm(eng.agents["N"], device=eng.device, dtype=torch.int32)
            eng.step_sequential(77, perm, noise=None, seed=3)
        res[lazy] = (snaps, dens, exits, eng.get_state())
        eng.close()
    (sa, da, ea, _), (sb, db, eb, fin) = res[False], res[True]
    assert ea == eb and ea > 50                                         # agents did leave, the same ones
    for a, b in zip(sa, sb):
        assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["agent_time"], b["agent_time"])
        for f in ("x", "y", "vx", "vy"):
            assert np.allclose(a[f], b[f], rtol=0, atol=1e-9), f
    for a, b in zip(da, db):
        assert np.allclose(a, b, rtol=0, atol=1e-9)
    assert np.isfinite(fin["x"]).all() and fin["agent_time"].max() == pytest.approx(78 * 0.02)
    gone = ~sb[-1]["status"].astype(bool)
    assert 50 < gone.sum() <= eb and (sb[-1]["agent_time"][gone] < 71 * 0.02 + 1e-12).all()     # (their clocks stopped)


def test_replacing_the_agents_after_a_lazy_step_writes_nowhere_else():
    """A lazy step leaves the library owing the caller's arrays a flush (raw pointers to them).  Replacing the agent set
    (Engine.set_agents, a different N) must pay or drop that debt BEFORE the old tensors are released: a later flush into
    freed blocks lands in whatever the caching allocator handed them to next (advisor finding, round 3)."""
    import torch
    eng = _fresh(True)
    for s in range(3):
        eng.step(s, noise=None, seed=5, sync=False)
    old = eng.get_state()                                               # (synchronises: what the old arrays must end up as)
    for s in range(3, 5):
        eng.step(s, noise=None, seed=5, sync=False)                     # ... and the debt is back
    n_old = eng.agents["N"]
    # a smaller crowd: the old tensors are released by set_agents; grab the allocator's blocks right away with canaries
    x = np.linspace(20.0, 30.0, 1000); y = np.full(1000, 25.0)
    eng.set_agents(x, y, np.zeros(1000), np.zeros(1000), np.full(1000, 1.3), np.zeros(1000, dtype=np.int32))
    canaries = [torch.full((n_old,), 7.25, dtype=torch.float64, device=eng.device) for _ in range(12)]
    eng.step(5, noise=None, seed=5)
    eng.step(6, noise=None, seed=5)
    new = eng.get_state()
    torch.cuda.synchronize()
    for c in canaries:
        assert bool((c == 7.25).all()), "a flush of the replaced agent set wrote into reallocated memory"
    assert new["x"].shape == (1000,) and np.isfinite(new["x"]).all() and np.allclose(new["y"], 25.0, atol=0.2)
    assert old["x"].shape == (n_old,)
    eng.close()
