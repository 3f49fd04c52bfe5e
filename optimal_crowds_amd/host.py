"""Host-side (NumPy) pieces of the path that run once per simulation: config/room loading, grid,
room -> potential rasteriser and the initial-placement sampler.  One-off initialisation, not kernels
(SURVEY.md section 2 rows 10, 11, 16, 17); results are identical to the reference's, only evaluated
on 1-D axes / local windows instead of full meshgrids so that 4096 x 4096 rooms initialise quickly.
"""
import json
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
R_IN = 0.2        # exclusion radius of the placement sampler, reference simulations.py:111


def load_config(config=None):
    """config.json: dict given > ./optimal_crowds/config.json (reference simulations.py:42) > packaged copy."""
    if isinstance(config, dict):
        return config
    for path in ([config] if config else []) + ["optimal_crowds/config.json", os.path.join(PKG, "config.json")]:
        if path and os.path.exists(path):
            with open(path) as f:
                return json.loads(f.read())
    raise FileNotFoundError("config.json")


def load_room(room):
    """rooms/<room>.json relative to the cwd (reference simulations.py:47), else the packaged room library."""
    if isinstance(room, dict):
        return room
    for path in (os.path.join("rooms", room + ".json"), os.path.join(PKG, "rooms", room + ".json")):
        if os.path.exists(path):
            with open(path) as f:
                return json.loads(f.read())
    raise FileNotFoundError("rooms/%s.json" % room)


def grid_dims(room_length, room_height, grid_step):
    """reference simulations.py:63-64 (Python float floor division)."""
    return int(room_length // grid_step + 1), int(room_height // grid_step + 1)


def axes(room_length, room_height, Nx, Ny):
    """1-D axes of the reference's meshgrid, simulations.py:69-70."""
    return np.linspace(0, room_length, Nx), np.linspace(0, room_height, Ny)


def _rect_mask(X0, Y0, r):
    return np.outer(abs(Y0 - r[1]) < r[3] / 2, abs(X0 - r[0]) < r[2] / 2)


def create_potential(room, targets, X0, Y0):
    """Room description -> V in {-k, 0, 1} (reference simulations.py:516-576): every wall adds -1, holes reset
    to 0, every cylinder adds -1, the border is -1, the key's targets are +1 (drawn last)."""
    Ny, Nx = len(Y0), len(X0)
    V = np.zeros((Ny, Nx))
    for w in room["walls"].values():
        V -= _rect_mask(X0, Y0, w)
    for h in room["holes"].values():
        V[_rect_mask(X0, Y0, h)] = 0
    hx, hy = X0[1] - X0[0], Y0[1] - Y0[0]
    for c in room["cylinders"].values():
        j0, j1 = max(int((c[0] - c[2]) / hx) - 1, 0), min(int((c[0] + c[2]) / hx) + 3, Nx)
        i0, i1 = max(int((c[1] - c[2]) / hy) - 1, 0), min(int((c[1] + c[2]) / hy) + 3, Ny)
        if j1 > j0 and i1 > i0:
            d = np.sqrt((X0[None, j0:j1] - c[0]) ** 2 + (Y0[i0:i1, None] - c[1]) ** 2)
            V[i0:i1, j0:j1] -= (d < c[2])
    V[:, 0] = -1
    V[:, -1] = -1
    V[0, :] = -1
    V[-1, :] = -1
    for t in targets:
        V[_rect_mask(X0, Y0, room["targets"][t])] = 1
    return V


def finalize_potential(V, cfg):
    """In place: walls -> wall_potential, targets -> target_potential (reference optimals.py:89-91)."""
    hp = cfg["hjb_params"]
    V[V < 0] = hp["wall_potential"]
    V[V > 0] = hp["target_potential"]
    return V


def sample_box(box, X0, Y0, place_ped, r_in=R_IN):
    """Rejection sampler of one initial box (reference simulations.py:122-140).  Same draws from np.random, same
    accept/reject decisions and the same marks in place_ped; the mask is evaluated on the window of grid nodes
    that can lie within r_in instead of the whole grid."""
    loc_N = int(box[4] * box[2] * box[3])
    xs, ys = np.empty(loc_N), np.empty(loc_N)
    hx, hy = X0[1] - X0[0], Y0[1] - Y0[0]
    Nx, Ny = len(X0), len(Y0)
    lo_x, hi_x = box[0] - box[2] / 2, box[0] + box[2] / 2
    lo_y, hi_y = box[1] - box[3] / 2, box[1] + box[3] / 2
    placed = 0
    uniform = np.random.uniform
    while placed < loc_N:
        x_in = uniform(lo_x, hi_x, 1)
        y_in = uniform(lo_y, hi_y, 1)
        j0 = max(int((x_in[0] - r_in) / hx) - 1, 0)
        j1 = min(int((x_in[0] + r_in) / hx) + 3, Nx)
        i0 = max(int((y_in[0] - r_in) / hy) - 1, 0)
        i1 = min(int((y_in[0] + r_in) / hy) + 3, Ny)
        if j1 > j0 and i1 > i0:
            mask = np.sqrt((X0[None, j0:j1] - x_in) ** 2 + (Y0[i0:i1, None] - y_in) ** 2) < r_in
            sub = place_ped[i0:i1, j0:j1]
            if sub[mask].any():
                continue
            sub[mask] = 1
        xs[placed] = x_in[0]
        ys[placed] = y_in[0]
        placed += 1
    v_des = np.random.normal(1.34, 0.26, size=loc_N)
    return xs, ys, v_des
