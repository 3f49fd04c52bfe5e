"""Tiny cached engines for standalone `ped` objects (single-agent queries outside a simulation)."""
import numpy as np

from . import host

_cache = {}


def engine_for(p):
    from .engine import Engine
    V = np.asarray(p.V, dtype=np.float64)
    key = (V.shape, float(p.room_length), float(p.room_height), hash(V.tobytes()))
    if key not in _cache:
        cfg = dict(host.load_config(None))
        cfg["grid_step"] = p.step
        cfg["tau_a"], cfg["b_min"], cfg["b_max"] = p.tau_a, p.b_min, p.b_max
        cfg["eta"], cfg["eta_walls"] = p.eta, p.eta_walls
        Ny, Nx = V.shape
        X0, Y0 = host.axes(p.room_length, p.room_height, Nx, Ny)
        eng = Engine(cfg, p.room_length, p.room_height, Nx, Ny, X0, Y0, 1, 16, dtype="float32")
        eng.set_potential(0, V, np.zeros((0, 4)))
        _cache.clear()
        _cache[key] = eng
    return _cache[key]
