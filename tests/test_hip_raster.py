"""Device-side room rasteriser (oc_rasterise_potential) against the host restatement of
simulation.create_potential + the finalisation of optimals.__init__ (reference simulations.py:516-576,
optimals.py:89-91): bit-identical arrays, and a facade that behaves the same whichever side rasterises."""
import contextlib
import copy
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oc_oracle as oc                     # noqa: E402  (checker only)
from optimal_crowds_amd import host, room_library      # noqa: E402


def _engine(room, cfg):
    from optimal_crowds_amd.engine import Engine
    Nx, Ny = host.grid_dims(room["room_length"], room["room_height"], cfg["grid_step"])
    X0, Y0 = host.axes(room["room_length"], room["room_height"], Nx, Ny)
    return Engine(cfg, room["room_length"], room["room_height"], Nx, Ny, X0, Y0, 1, 16), X0, Y0


@pytest.mark.parametrize("name,grid_step", [("room_test", 0.05), ("gold_a", 0.05), ("gold_b", 0.05), ("gold_b", 0.013),
                                            ("exit_opposite", 0.05), ("slalom", 0.05), ("metro_station", 0.05)])
def test_rasteriser_bit_identical_to_host_and_oracle(name, grid_step):
    cfg = copy.deepcopy(oc.DEFAULT_CONFIG)
    cfg["grid_step"] = grid_step
    room = room_library.ROOMS[name]()
    eng, X0, Y0 = _engine(room, cfg)
    hp = cfg["hjb_params"]
    keys = []
    for box in room["initial_boxes"].values():
        if box[5:] not in keys:
            keys.append(box[5:])
    keys.append([])                                     # no target at all: walls, holes, cylinders and the border only
    for targets in keys:
        raw = host.create_potential(room, targets, X0, Y0)
        got_raw = eng.rasterise_potential(room, targets, hp["wall_potential"], hp["target_potential"],
                                          finalise=False).cpu().numpy()
        assert np.array_equal(got_raw.astype(np.float64), raw)
        fin = host.finalize_potential(raw.copy(), cfg)
        got = eng.rasterise_potential(room, targets, hp["wall_potential"], hp["target_potential"]).cpu().numpy()
        assert np.array_equal(got.astype(np.float64), fin)
    if eng.Nx * eng.Ny <= 200 * 200:                    # the oracle's full-meshgrid restatement of the reference
        X, Y = np.meshgrid(X0, Y0)
        assert np.array_equal(oc.create_potential(room, keys[0], X, Y), host.create_potential(room, keys[0], X0, Y0))
    eng.close()


def test_rasteriser_overlaps_holes_and_cylinders():
    """overlapping walls count twice, a hole resets whatever the walls drew, cylinders subtract after the holes, the
    border overrides all of them and targets are drawn last (the reference's order of drawing)"""
    cfg = copy.deepcopy(oc.DEFAULT_CONFIG)
    room = {"room_length": 6, "room_height": 4,
            "initial_boxes": {"b": [1.0, 2.0, 1.0, 1.0, 1.0, "t"]},
            "targets": {"t": [5.8, 2.0, 0.6, 1.0], "u": [3.0, 0.1, 1.0, 0.4]},
            "walls": {"w1": [3.0, 2.0, 2.0, 0.5], "w2": [3.5, 2.0, 2.0, 1.5], "w3": [0.0, 0.0, 1.0, 1.0]},
            "holes": {"h1": [3.2, 2.0, 0.6, 3.0]},
            "cylinders": {"c1": [3.2, 2.6, 0.2], "c2": [1.5, 3.0, 0.45], "c3": [6.0, 4.0, 0.7]}}
    eng, X0, Y0 = _engine(room, cfg)
    for targets in (["t"], ["t", "u"], ["u"]):
        raw = host.create_potential(room, targets, X0, Y0)
        assert set(np.unique(raw)) >= {-2.0, -1.0, 0.0, 1.0}
        got = eng.rasterise_potential(room, targets, -100.0, 1.0, finalise=False).cpu().numpy()
        assert np.array_equal(got.astype(np.float64), raw)
        X, Y = np.meshgrid(X0, Y0)
        assert np.array_equal(oc.create_potential(room, targets, X, Y), raw)
    eng.close()


def test_facade_is_the_same_whichever_side_rasterises():
    from optimal_crowds_amd import simulations
    cfg = copy.deepcopy(oc.DEFAULT_CONFIG)
    sims = {}
    for mode in ("host", "device"):
        np.random.seed(5)
        with contextlib.redirect_stdout(io.StringIO()):
            sim = simulations.simulation("gold_b", 2.0, config=copy.deepcopy(cfg), noise="device", seed=3,
                                         rasterise=mode)
            sim.run()
        sims[mode] = sim
    a, b = sims["host"], sims["device"]
    assert list(a.Vs) == list(b.Vs)
    for k in a.Vs:
        assert np.array_equal(a.Vs[k], b.Vs[k]) and np.array_equal(a.targets[k].V, b.targets[k].V)
    assert np.array_equal(a.V, b.V)
    sa, sb = a._engine.get_state(), b._engine.get_state()
    assert np.array_equal(sa["status"], sb["status"])
    for f in ("x", "y", "vx", "vy"):                    # (float64 LDS atomics of the pair sums are applied in no fixed
        assert np.allclose(sa[f], sb[f], rtol=0, atol=1e-9)     # order: two runs of ONE configuration differ likewise)
    assert a.simu_step == b.simu_step and a.inside == b.inside
    assert np.allclose(a.gaussian_density(0.4), b.gaussian_density(0.4), rtol=1e-9, atol=1e-12)
    with pytest.raises(ValueError):
        simulations.simulation("gold_b", 2.0, config=copy.deepcopy(cfg), rasterise="gpu")
