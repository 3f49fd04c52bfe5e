"""CPU oracle (float64, NumPy) for the optimal_crowds hot path.  TEST INFRASTRUCTURE ONLY.

This module is a plain restatement of the reference's algorithm for the path named in
BASELINE.json (HJB solve + GCFM agent update).  It exists to CHECK the HIP path:
only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it.
The product package (optimal_crowds_amd/) never imports it and has no CPU fallback.

Pinning: every function below is checked in tests/test_oracle_golden.py against fixtures
captured by importing the reference itself (tests/golden/make_goldens.py).
The time integrator is third-party: scipy.integrate.solve_ivp(method='RK45'), SciPy 1.15.3
(scipy/integrate/_ivp/rk.py, common.py, ivp.py, base.py) -- not pinned by the reference
(no requirements file); `rk45_solve` restates its published Dormand-Prince algorithm and is
pinned by fixtures G3 (step sequence, nfev, dense output) generated with SciPy 1.15.3.

Citations are `file:line` into the reference (optimals.py / pedestrians.py /
simulations.py) or `scipy:` into the SciPy files above.
"""
import json

import numpy as np

# ------------------------------------------------------------------------------------------
# configuration (config.json keys, simulations.py:42-97, optimals.py:36-74)
# ------------------------------------------------------------------------------------------
DEFAULT_CONFIG = {
    "dt": 0.02, "grid_step": 0.05, "v_max": 2, "relaxation": 0.1, "sigma_convolution": 0.5,
    "a_min": 0.18, "b_min": 0.2, "b_max": 0.25, "tau_a": 0.53, "eta": 0.05, "eta_walls": 0.15,
    "repulsion_cutoff": 4, "recompute_frequency": 50,
    "hjb_params": {"sigma": 0.2, "mu": 5, "wall_potential": -100, "target_potential": 1, "g": -0.005},
}
LIM = 10e-3                    # optimals.py:95, simulations.py:74
R_IN = 0.2                     # simulations.py:111
COS_VISION = np.cos(0.7 * np.pi)   # pedestrians.py:262


def grid_dims(room_length, room_height, grid_step):
    """simulations.py:63-64 / optimals.py:55-56 (float floor-division quirk Q1)."""
    return int(room_length // grid_step + 1), int(room_height // grid_step + 1)


def axes(room_length, room_height, Nx, Ny):
    """simulations.py:69-70: geometry grid is linspace(0, L, N) (spacing != grid_step, Q2)."""
    return np.linspace(0, room_length, Nx), np.linspace(0, room_height, Ny)


# ------------------------------------------------------------------------------------------
# room -> potential  (simulations.py:516-576, optimals.py:89-91, simulations.py:121,157)
# ------------------------------------------------------------------------------------------
def create_potential(room, targets, X, Y):
    """V in {-k, 0, 1}: walls add -1 each, holes reset to 0, cylinders add -1, border -1, targets 1."""
    Ny, Nx = X.shape
    V = np.zeros((Ny, Nx))
    for w in room["walls"].values():
        m = (abs(X - w[0]) < w[2] / 2) * (abs(Y - w[1]) < w[3] / 2)
        Vt = np.zeros((Ny, Nx))
        Vt[m] = -1
        V += Vt
    for h in room["holes"].values():
        m = (abs(X - h[0]) < h[2] / 2) * (abs(Y - h[1]) < h[3] / 2)
        V[m] = 0
    for c in room["cylinders"].values():
        Vt = np.zeros((Ny, Nx))
        Vt[np.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2) < c[2]] = -1
        V += Vt
    V[:, 0] = -1
    V[:, -1] = -1
    V[0, :] = -1
    V[-1, :] = -1
    for t in targets:
        t = room["targets"][t]
        m = (abs(X - t[0]) < t[2] / 2) * (abs(Y - t[1]) < t[3] / 2)
        V[m] = 1
    return V


def finalize_potential(V, cfg):
    """optimals.py:89-91: walls -> wall_potential, targets -> target_potential (in place, Q3)."""
    hp = cfg["hjb_params"]
    V[V < 0] = hp["wall_potential"]
    V[V > 0] = hp["target_potential"]
    return V


# ------------------------------------------------------------------------------------------
# HJB right-hand side  (optimals.py:144-164)
# ------------------------------------------------------------------------------------------
def hjb_rhs(phi, V, m, sigma, mu, g, dx, dy):
    ny, nx = V.shape
    p = np.empty((ny + 2, nx + 2))
    p[1:-1, 1:-1] = phi.reshape(ny, nx)
    p[0, :] = p[2, :]
    p[-1, :] = p[-3, :]
    p[:, -1] = p[:, -3]
    p[:, 0] = p[:, 2]
    lap = (p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] - 4 * p[1:-1, 1:-1]) / (dx * dy)
    out = -0.5 * sigma ** 2 * lap - ((V + g * m) * p[1:-1, 1:-1]) / (mu * sigma ** 2)
    out[V < 0] = 0
    return out.reshape(nx * ny)


# ------------------------------------------------------------------------------------------
# Dormand-Prince RK45 with scipy's controller and dense output
# scipy:rk.py:14-72 (rk_step), :111-176 (_step_impl), :377-404 (tables), :552-575 (dense out),
# scipy:common.py:63-66 (RMS norm), :68-134 (initial step), scipy:ivp.py:598-616,706-723
# ------------------------------------------------------------------------------------------
RK_C = np.array([0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1])
RK_A = np.array([
    [0, 0, 0, 0, 0],
    [1 / 5, 0, 0, 0, 0],
    [3 / 40, 9 / 40, 0, 0, 0],
    [44 / 45, -56 / 15, 32 / 9, 0, 0],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729, 0],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656]])
RK_B = np.array([35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84])
RK_E = np.array([-71 / 57600, 0, 71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525, 1 / 40])
RK_P = np.array([
    [1, -8048581381 / 2820520608, 8663915743 / 2820520608, -12715105075 / 11282082432],
    [0, 0, 0, 0],
    [0, 131558114200 / 32700410799, -68118460800 / 10900136933, 87487479700 / 32700410799],
    [0, -1754552775 / 470086768, 14199869525 / 1410260304, -10690763975 / 1880347072],
    [0, 127303824393 / 49829197408, -318862633887 / 49829197408, 701980252875 / 199316789632],
    [0, -282668133 / 205662961, 2019193451 / 616988883, -1453857185 / 822651844],
    [0, 40617522 / 29380423, -110615467 / 29380423, 69997945 / 29380423]])
SAFETY, MIN_FACTOR, MAX_FACTOR = 0.9, 0.2, 10.0


def _rms(x):
    return np.linalg.norm(x) / x.size ** 0.5


def select_initial_step(fun, t0, y0, t_bound, f0, direction, order, rtol, atol):
    interval_length = abs(t_bound - t0)
    if interval_length == 0.0:
        return 0.0
    scale = atol + np.abs(y0) * rtol
    d0 = _rms(y0 / scale)
    d1 = _rms(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    h0 = min(h0, interval_length)
    y1 = y0 + h0 * direction * f0
    f1 = fun(t0 + h0 * direction, y1)
    d2 = _rms((f1 - f0) / scale) / h0
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = max(1e-6, h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1, d2)) ** (1 / (order + 1))
    return min(100 * h0, h1, interval_length)


def rk45_solve(fun, t0, tf, y0, t_eval, rtol=1e-3, atol=1e-6, on_slice=None):
    """Integrate y' = fun(t, y) from t0 to tf (either direction) exactly as solve_ivp(RK45, t_eval).

    Returns (ys, stats): ys has shape (n, len(t_eval)) unless `on_slice(k, y)` is given, in which
    case each dense-output slice is handed to the callback (k = index into t_eval) and ys is None.
    stats: nfev, n_attempt, n_accept, t_steps (accepted step end times), h_abs0.
    """
    t_eval = np.asarray(t_eval, dtype=float)
    direction = np.sign(tf - t0) if tf != t0 else 1.0
    y = np.array(y0, dtype=float)
    n = y.size
    nfev = 0

    def f(t, yy):
        nonlocal nfev
        nfev += 1
        return fun(t, yy)

    t = float(t0)
    fcur = f(t, y)
    h_abs = select_initial_step(f, t, y, tf, fcur, direction, 4, rtol, atol)
    h_abs0 = h_abs
    K = np.empty((7, n))
    err_exp = -1 / 5
    if direction > 0:
        te = t_eval
        ti = 0
    else:
        te = t_eval[::-1]          # increasing, scipy:ivp.py:609-613
        ti = te.shape[0]
    ys = None if on_slice is not None else np.empty((n, t_eval.size))
    n_attempt = n_accept = 0
    t_steps = [t]
    while True:
        if n == 0 or t == tf:      # scipy:base.py:191-196
            break
        # ---- _step_impl
        min_step = 10 * np.abs(np.nextafter(t, direction * np.inf) - t)
        h_abs = max(h_abs, min_step)
        accepted = rejected = False
        while not accepted:
            if h_abs < min_step:
                raise RuntimeError("step size too small")
            h = h_abs * direction
            t_new = t + h
            if direction * (t_new - tf) > 0:
                t_new = tf
            h = t_new - t
            h_abs = np.abs(h)
            n_attempt += 1
            K[0] = fcur
            for s in range(1, 6):
                dy = np.dot(K[:s].T, RK_A[s, :s]) * h
                K[s] = f(t + RK_C[s] * h, y + dy)
            y_new = y + h * np.dot(K[:-1].T, RK_B)
            f_new = f(t + h, y_new)
            K[6] = f_new
            scale = atol + np.maximum(np.abs(y), np.abs(y_new)) * rtol
            err = _rms(np.dot(K.T, RK_E) * h / scale)
            if err < 1:
                factor = MAX_FACTOR if err == 0 else min(MAX_FACTOR, SAFETY * err ** err_exp)
                if rejected:
                    factor = min(1, factor)
                h_abs *= factor
                accepted = True
            else:
                h_abs *= max(MIN_FACTOR, SAFETY * err ** err_exp)
                rejected = True
        n_accept += 1
        t_old, y_old = t, y
        t, y, fcur = t_new, y_new, f_new
        t_steps.append(t)
        # ---- t_eval handling, scipy:ivp.py:706-723
        if direction > 0:
            ti_new = np.searchsorted(te, t, side="right")
            idx = np.arange(ti, ti_new)
        else:
            ti_new = np.searchsorted(te, t, side="left")
            idx = np.arange(ti_new, ti)[::-1]
        if idx.size > 0:
            Q = K.T.dot(RK_P)
            hh = t - t_old
            for k in idx:
                x = (te[k] - t_old) / hh
                p = np.cumprod(np.tile(x, 4))
                yk = hh * np.dot(Q, p) + y_old
                kk = k if direction > 0 else t_eval.size - 1 - k
                if on_slice is not None:
                    on_slice(kk, yk)
                else:
                    ys[:, kk] = yk
            ti = ti_new
        if direction * (t - tf) >= 0:
            break
    stats = dict(nfev=nfev, n_attempt=n_attempt, n_accept=n_accept,
                 t_steps=np.array(t_steps), h_abs0=h_abs0)
    return ys, stats


# ------------------------------------------------------------------------------------------
# direction field (optimals.py:168-186) and the fill loop (optimals.py:139-142,193-204)
# ------------------------------------------------------------------------------------------
def vels(phi, ny, nx, mu, dx, dy, lim=LIM):
    p = phi.reshape(ny, nx).copy()
    p = p * (p > lim) + lim * (p < lim)
    gx = (p[1:-1, 2:] - p[1:-1, :-2]) / (2 * dx)
    gy = (p[2:, 1:-1] - p[:-2, 1:-1]) / (2 * dy)
    pc = p[1:-1, 1:-1] * (p[1:-1, 1:-1] > lim) + lim * (p[1:-1, 1:-1] < lim)
    with np.errstate(all="ignore"):
        vx = gx / (mu * pc)
        vy = gy / (mu * pc)
        norm = np.sqrt(vx ** 2 + vy ** 2)
        den = norm * (norm > lim) + (norm < lim)
        return (vx * (norm > lim)) / den, (vy * (norm > lim)) / den


class Optimals:
    """State of one target key: V (finalised), vx_opt / vy_opt (nt0-1, Ny-2, Nx-2), nt_opt."""

    def __init__(self, V, T, room_length, room_height, cfg):
        hp = cfg["hjb_params"]
        self.cfg = cfg
        self.room_length, self.room_height = room_length, room_height
        gs = cfg["grid_step"]
        self.Nx, self.Ny = grid_dims(room_length, room_height, gs)
        self.dx = self.dy = gs
        self.sigma, self.mu, self.g = hp["sigma"], hp["mu"], hp["g"]
        self.dt, self.T = cfg["dt"], T
        self.nt_opt = round(T / self.dt)
        self.vx_opt = np.empty((self.nt_opt - 1, self.Ny - 2, self.Nx - 2))
        self.vy_opt = np.empty((self.nt_opt - 1, self.Ny - 2, self.Nx - 2))
        self.V = finalize_potential(V, cfg)
        self.stats = None

    def rhs(self, phi, m):
        return hjb_rhs(phi, self.V, m, self.sigma, self.mu, self.g, self.dx, self.dy)

    def compute_optimal_velocity(self, t, m, keep_phi=None):
        """optimals.py:124-206 incl. the Q9 horizon semantics.  keep_phi: dict to receive phi slices."""
        self.t = t
        self.nt_opt = round((self.T - t) / self.dt)
        nt = self.nt_opt
        t_eval = np.linspace(self.T, 0, nt)
        y0 = np.ones(self.Ny * self.Nx)

        def on_slice(k, yk):
            if keep_phi is not None:
                keep_phi[k] = yk.copy()
            if 1 <= k <= nt - 1:
                i = nt - k          # optimals.py:200-204: vx_opt[i-1] = vels(sol.y[:, nt-i])
                vx, vy = vels(yk, self.Ny, self.Nx, self.mu, self.dx, self.dy)
                self.vx_opt[i - 1] = vx
                self.vy_opt[i - 1] = vy

        _, self.stats = rk45_solve(lambda tt, yy: self.rhs(yy, m), self.T, 0.0, y0, t_eval, on_slice=on_slice)

    def choose_optimal_velocity(self, pos, t, fixed=False):
        """optimals.py:212-250 (Q7 diagonal pair, Q8 IndexError band, Q9 time guard).  fixed=True: the corrected
        lookup (fix flag 'lookup'), lookup_bilinear below."""
        x, y = pos
        if t >= self.nt_opt - 1:
            return np.array((0., 0.))
        if fixed:
            return lookup_bilinear(self.vx_opt[t], self.vy_opt[t], x, y, self.room_length, self.room_height)
        if x < self.room_length - self.dx:
            j = int(x // self.dx)
            if x > self.dx:
                j = [j, j + 1]
        else:
            j = self.Nx - 3
        if y < self.room_height - self.dy:
            i = int(y // self.dy)
            if y > self.dy:
                i = [i, i + 1]
        else:
            i = self.Ny - 3
        return np.array((np.mean(self.vx_opt[t][i, j]), np.mean(self.vy_opt[t][i, j])))


def lookup_bilinear(vx_int, vy_int, x, y, room_length, room_height):
    """Fix flag 'lookup' (Q7/Q8 corrected; no reference lines -- this is what optimals.py:234-248 intends): bilinear
    interpolation of the direction field over the four grid NODES around (x, y).  vx_int, vy_int: one slice, the
    reference's interior arrays (Ny-2, Nx-2), whose element [i, j] belongs to node (i+1, j+1) at
    (X0[j+1], Y0[i+1]) = ((j+1) L/(Nx-1), (i+1) H/(Ny-1)).  Positions outside the interior nodes are clamped."""
    Ny, Nx = vx_int.shape[0] + 2, vx_int.shape[1] + 2

    def split(g, n):
        fl = np.floor(g)
        if not fl >= 1.0:
            fl, g = 1.0, 1.0
        if fl >= n - 2:
            fl = float(n - 2)
            g = fl
        i0 = int(fl)
        return i0, min(i0 + 1, n - 2), g - fl

    c0, c1, fx = split(x / (room_length / (Nx - 1)), Nx)
    r0, r1, fy = split(y / (room_height / (Ny - 1)), Ny)
    w00, w01, w10, w11 = (1 - fy) * (1 - fx), (1 - fy) * fx, fy * (1 - fx), fy * fx
    out = []
    for f in (vx_int, vy_int):
        out.append(w00 * float(f[r0 - 1, c0 - 1]) + w01 * float(f[r0 - 1, c1 - 1]) + w10 * float(f[r1 - 1, c0 - 1])
                   + w11 * float(f[r1 - 1, c1 - 1]))
    return np.array(out)


def lookup_indices(x, y, room_length, room_height, dx, dy, Nx, Ny):
    """Vectorised index logic of choose_optimal_velocity: returns (i0, i1, j0, j1, bad).

    The reference averages cells (i0,j0) and (i1,j1) when both indices are 2-lists (diagonal pair),
    (i0,j0),(i0,j1) when only j is a list, (i0,j0),(i1,j0) when only i is -> in every case the mean of
    the two cells (i0,j0) and (i1,j1) with i1=i0 or j1=j0 where the index is scalar; and a single cell
    if both are scalar.  bad marks the IndexError band (index Nx-2 / Ny-2 on an axis of that size).
    """
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    jx = np.floor_divide(x, dx).astype(np.int64)
    iy = np.floor_divide(y, dy).astype(np.int64)
    in_x = x < room_length - dx
    in_y = y < room_height - dy
    two_x = in_x & (x > dx)
    two_y = in_y & (y > dy)
    j0 = np.where(in_x, jx, Nx - 3)
    i0 = np.where(in_y, iy, Ny - 3)
    j1 = np.where(two_x, j0 + 1, j0)
    i1 = np.where(two_y, i0 + 1, i0)
    # numpy fancy-index semantics: negative indices wrap; indices >= size raise IndexError
    bad = (j1 > Nx - 3) | (i1 > Ny - 3) | (j0 < -(Nx - 2)) | (i0 < -(Ny - 2))
    # both lists -> diagonal pair; exactly one list -> 2 cells along that axis; none -> 1 cell
    return i0, i1, j0, j1, bad


# ------------------------------------------------------------------------------------------
# GCFM forces (pedestrians.py:216-334), vectorised over pairs / agents, same formulas
# ------------------------------------------------------------------------------------------
def pair_force(pi, vi, vdes_i, pj, vj, cfg, unit=False):
    """Repulsion of j on i (pedestrians.py:216-280).  All inputs (..., 2) / (...,) broadcastable.
    unit=True: fix flag 'pair_force' (Q12 corrected): -rep * e along the unit vector instead of -rep * R."""
    a_min = cfg["b_min"]            # Q5: a_min is b_min (simulations.py:88, pedestrians.py:74)
    tau, b_min, b_max, eta = cfg["tau_a"], cfg["b_min"], cfg["b_max"], cfg["eta"]
    pi, vi, pj, vj = (np.asarray(a, dtype=float) for a in (pi, vi, pj, vj))
    with np.errstate(all="ignore"):
        si = np.sqrt(vi[..., 0] ** 2 + vi[..., 1] ** 2)
        sj = np.sqrt(vj[..., 0] ** 2 + vj[..., 1] ** 2)
        a_i = a_min + tau * si
        b_i = b_max - (b_max - b_min) * np.minimum(si / vdes_i, 1)
        a_j = a_min + tau * sj
        b_j = b_max - (b_max - b_min) * np.minimum(sj / vdes_i, 1)      # Q6: i's v_des
        R = pj - pi
        r = np.sqrt(R[..., 0] ** 2 + R[..., 1] ** 2)
        e = R / r[..., None]
        vr = vj - vi
        d = vr[..., 0] * (-e[..., 0]) + vr[..., 1] * (-e[..., 1])
        v_rel = 0.5 * (d + np.abs(d))
        k = np.maximum((vi[..., 0] * e[..., 0] + vi[..., 1] * e[..., 1]) / si - COS_VISION, 0) / (1 - COS_VISION)
        k = np.where(si > 0, k, 0.0)
        al_i = np.arctan2(R[..., 1], R[..., 0])
        be_i = np.arctan2(vi[..., 1], vi[..., 0])
        al_j = np.arctan2(-R[..., 1], -R[..., 0])
        be_j = np.arctan2(vj[..., 1], vj[..., 0])
        q_i = np.sqrt(1 / ((np.cos(al_i - be_i) / a_i) ** 2 + (np.sin(al_i - be_i) / b_i) ** 2))
        q_j = np.sqrt(1 / ((np.cos(al_j - be_j) / a_j) ** 2 + (np.sin(al_j - be_j) / b_j) ** 2))
        dist = r - q_i - q_j
        rep = np.minimum(k * np.exp(-dist / (eta * (1 + v_rel))), 1)
        return -rep[..., None] * (e if unit else R)


def nearest_wall_index_bruteforce(px, py, X, Y, V):
    """pedestrians.py:311-313: argmin over the whole grid of sqrt(.)+V*10e3, first minimum."""
    d = np.sqrt((X - px) ** 2 + (Y - py) ** 2)
    return int(np.argmin(d + V * 10e3))


class WallMap:
    """Exact fast nearest-'wall' query, same answer as nearest_wall_index_bruteforce.

    The argmin of fl(fl(dist) + fl(V*10e3)) is attained in the lowest-valued V class present
    (V in {-100, 0, 1}: classes are >= 1e4 apart, distances < 1e4 m).  Inside that class ("sites")
    and inside one grid row, the objective is a non-decreasing function of |X[j]-px|, so the row
    minimum is at the nearest site on the left or on the right of px; ties after rounding between
    same-side sites of one row cannot occur for grid spacings >= 1e-3 m and rooms < 1e4 m
    (distance gaps >= spacing^2/(2*dist) >> ulp(1e6) = 1.2e-10).  So scanning every row with those
    two candidates, in row order, with a strict '<' update reproduces numpy's first-minimum rule.
    """

    def __init__(self, X0, Y0, V):
        self.X0, self.Y0 = X0, Y0
        vmin = V.min()
        self.off = vmin * 10e3
        site = (V == vmin)
        Ny, Nx = V.shape
        cols = np.arange(Nx)
        # left[i, j] = largest site column <= j (or -1); right[i, j] = smallest site column >= j (or Nx)
        self.left = np.maximum.accumulate(np.where(site, cols, -1), axis=1)
        self.right = np.minimum.accumulate(np.where(site, cols, Nx)[:, ::-1], axis=1)[:, ::-1]
        self.Nx, self.Ny = Nx, Ny

    def query(self, px, py):
        X0, Y0, Nx = self.X0, self.Y0, self.Nx
        jx = int(np.clip(np.searchsorted(X0, px, side="right") - 1, 0, Nx - 1))   # X0[jx] <= px
        jl = self.left[:, jx]
        jr = self.right[:, min(jx + 1, Nx - 1)] if jx + 1 < Nx else np.full(self.Ny, Nx)
        rows = np.arange(self.Ny)
        best = np.inf
        bi = -1
        dy2 = (Y0 - py) ** 2
        for cand in (jl, jr):
            ok = (cand >= 0) & (cand < Nx)
            c = np.where(ok, cand, 0)
            f = np.sqrt((X0[c] - px) ** 2 + dy2) + self.off
            f = np.where(ok, f, np.inf)
            # flat index for tie-breaking: row-major first minimum
            flat = rows * Nx + c
            m = f.min()
            if m < best or (m == best and flat[f == m].min() < bi):
                best = m
                bi = int(flat[f == m].min())
            elif m == best:
                pass
        return bi


def wall_force(p, v, vdes, X, Y, V, cfg, index=None, fixed=False):
    """pedestrians.py:282-334 for one agent; `index` may carry a precomputed nearest-cell index.
    fixed=True: fix flag 'wall_force' (Q12 corrected): q_i with its square root, -rep * e, no factor 3."""
    a_min = cfg["b_min"]
    tau, b_min, b_max, eta_w = cfg["tau_a"], cfg["b_min"], cfg["b_max"], cfg["eta_walls"]
    if index is None:
        index = nearest_wall_index_bruteforce(p[0], p[1], X, Y, V)
    ind = np.unravel_index(index, X.shape)
    R = np.array((X[ind], Y[ind])) - p
    with np.errstate(all="ignore"):
        r = np.linalg.norm(R)
        e = R / r
        d = v[0] * e[0] + v[1] * e[1]
        v_rel = 0.5 * (d + abs(d))
        al = np.arctan2(R[1], R[0])
        be = np.arctan2(v[1], v[0])
        s = np.linalg.norm(v)
        a_i = a_min + tau * s
        b_i = b_max - (b_max - b_min) * np.minimum(s / vdes, 1)
        q_i = 1 / ((np.cos(al - be) / a_i) ** 2 + (np.sin(al - be) / b_i) ** 2)    # Q12: no sqrt
        if fixed:
            q_i = np.sqrt(q_i)
        dist = r - q_i
        rep = np.minimum(np.exp(-dist / (eta_w * (1 + v_rel))), 1)
        if fixed:
            return -rep * e
        return -3 * rep * R


def gaussian_density(x, y, status, X, Y, V_all, sigma, fixed_norm=False):
    """simulations.py:453-487 (Q11 normaliser 2*pi*sigma; mask where every key has a wall).
    fixed_norm=True: fix flag 'density_norm', C = 2*pi*sigma^2."""
    d = np.zeros(X.shape)
    C = 2 * np.pi * sigma ** 2 if fixed_norm else np.sqrt(4 * np.pi ** 2 * sigma ** 2)
    for k in range(len(x)):
        if status[k]:
            d += np.exp(-((X - x[k]) ** 2 + (Y - y[k]) ** 2) / (2 * sigma ** 2)) / C
    d[V_all < 0] = 0
    return d


# ------------------------------------------------------------------------------------------
# placement sampler (simulations.py:122-140), windowed but decision- and RNG-identical
# ------------------------------------------------------------------------------------------
def sample_box(box, X0, Y0, place_ped, r_in=R_IN):
    loc_N = int(box[4] * box[2] * box[3])
    xs = np.empty(loc_N)
    ys = np.empty(loc_N)
    placed = 0
    hx = X0[1] - X0[0]
    hy = Y0[1] - Y0[0]
    Nx, Ny = len(X0), len(Y0)
    while placed < loc_N:
        x_in = np.random.uniform(box[0] - box[2] / 2, box[0] + box[2] / 2, 1)
        y_in = np.random.uniform(box[1] - box[3] / 2, box[1] + box[3] / 2, 1)
        j0 = max(int((x_in[0] - r_in) / hx) - 1, 0)
        j1 = min(int((x_in[0] + r_in) / hx) + 3, Nx)
        i0 = max(int((y_in[0] - r_in) / hy) - 1, 0)
        i1 = min(int((y_in[0] + r_in) / hy) + 3, Ny)
        if j1 <= j0 or i1 <= i0:
            mask = np.zeros((0, 0), dtype=bool)
            sub = place_ped[0:0, 0:0]
        else:
            mask = np.sqrt((X0[None, j0:j1] - x_in) ** 2 + (Y0[i0:i1, None] - y_in) ** 2) < r_in
            sub = place_ped[i0:i1, j0:j1]
        if 1 in sub[mask]:
            continue
        sub[mask] = 1
        xs[placed] = x_in[0]
        ys[placed] = y_in[0]
        placed += 1
    v_des = np.random.normal(1.34, 0.26, size=loc_N)
    return xs, ys, v_des


# ------------------------------------------------------------------------------------------
# the simulation (simulations.py:20-162 init, :252-342 step, :401-451 run)
# ------------------------------------------------------------------------------------------
class Simulation:
    """SoA restatement.  mode='sequential' reproduces the reference (Gauss-Seidel in a fresh random
    permutation, consuming np.random exactly like it); mode='synchronous' is the frozen-snapshot
    (Jacobi) update the GPU implements, with the SAME permutation-ordered noise assignment."""

    def __init__(self, room, T, recompute=False, cfg=None, mode="sequential", fast_wall=True, fixes=()):
        """fixes: subset of {'lookup', 'wall_force', 'pair_force', 'density_norm', 'horizon'} -- the opt-in
        corrections of the product (oc_hip.h OC_FIX_*), restated here so that they too have a checker."""
        self.fixes = frozenset(fixes)
        self.cfg = cfg = dict(DEFAULT_CONFIG if cfg is None else cfg)
        self.room = room if isinstance(room, dict) else json.load(open(room))
        room = self.room
        self.recompute, self.mode, self.fast_wall = recompute, mode, fast_wall
        self.room_length, self.room_height = room["room_length"], room["room_height"]
        gs = cfg["grid_step"]
        self.Nx, self.Ny = grid_dims(self.room_length, self.room_height, gs)
        self.dx = self.dy = gs
        self.X0, self.Y0 = axes(self.room_length, self.room_height, self.Nx, self.Ny)
        self.X, self.Y = np.meshgrid(self.X0, self.Y0)
        self.T, self.time, self.simu_step, self.dt = T, 0., 0, cfg["dt"]
        self.key_names, self.targets, self.Vs = [], {}, {}
        self.V = np.zeros((self.Ny, self.Nx)) + 1
        self.place_ped = np.zeros((self.Ny, self.Nx))
        xs, ys, vd, kk, poss = [], [], [], [], []
        for box in room["initial_boxes"].values():
            tg = box[5:]
            key = " or ".join(tg)
            V = create_potential(room, tg, self.X, self.Y)
            self.Vs[key] = V
            self.targets[key] = Optimals(V, T, self.room_length, self.room_height, cfg)
            if key not in self.key_names:
                self.key_names.append(key)
            self.V += V
            bx, by, bv = sample_box(box, self.X0, self.Y0, self.place_ped)
            xs.append(bx); ys.append(by); vd.append(bv)
            kk.append(np.full(len(bx), self.key_names.index(key)))
            poss += [tg] * len(bx)
        self.V[self.V > np.min(self.V)] = 0
        self.x = np.concatenate(xs); self.y = np.concatenate(ys)
        self.vx = np.zeros_like(self.x); self.vy = np.zeros_like(self.x)
        self.v_des = np.concatenate(vd)
        self.key = np.concatenate(kk).astype(np.int64)
        self.possible = poss
        self.N = len(self.x)
        self.status = np.ones(self.N, dtype=bool)
        self.agent_time = np.zeros(self.N)
        self.inside = self.N
        self.wallmaps = {k: WallMap(self.X0, self.Y0, self.Vs[k]) for k in self.key_names}
        self.history_inside = []

    # -- forces on agent i given the state arrays (current values)
    def _forces(self, i, x, y, vx, vy, status):
        key = self.key_names[self.key[i]]
        pos = np.array((x[i], y[i]))
        des = self.v_des[i] * self.targets[key].choose_optimal_velocity(pos, self.simu_step,
                                                                        fixed="lookup" in self.fixes)
        dist = np.sqrt((x - x[i]) ** 2 + (y - y[i]) ** 2)
        sel = status & (dist < self.cfg["repulsion_cutoff"])
        sel[i] = False
        js = np.nonzero(sel)[0]
        rep = np.array((0., 0.))
        if len(js):
            f = pair_force(pos, np.array((vx[i], vy[i])), self.v_des[i],
                           np.column_stack([x[js], y[js]]), np.column_stack([vx[js], vy[js]]), self.cfg,
                           unit="pair_force" in self.fixes)
            for row in f:                      # ascending j, sequential fp64 sum (Q20)
                rep = rep + row
        idx = self.wallmaps[key].query(x[i], y[i]) if self.fast_wall else None
        wall = wall_force(pos, np.array((vx[i], vy[i])), self.v_des[i], self.X, self.Y, self.Vs[key], self.cfg, idx,
                          fixed="wall_force" in self.fixes)
        return des, rep, wall

    def _advance(self, i, des, rep, wall, noise, x, y, vx, vy):
        c = self.cfg
        dt = self.dt
        cvx = vx[i] + c["hjb_params"]["sigma"] / 2 * noise[0] + rep[0] + wall[0]
        cvy = vy[i] + c["hjb_params"]["sigma"] / 2 * noise[1] + rep[1] + wall[1]
        ax = (des[0] - cvx) / c["relaxation"]
        ay = (des[1] - cvy) / c["relaxation"]
        nx = x[i] + cvx * dt + 0.5 * ax * dt ** 2
        ny = y[i] + cvy * dt + 0.5 * ay * dt ** 2
        nvx = cvx + ax * dt
        nvy = cvy + ay * dt
        norm = np.sqrt(nvx ** 2 + nvy ** 2)
        if not norm < c["v_max"]:
            nvx, nvy = nvx * (c["v_max"] / norm), nvy * (c["v_max"] / norm)
        return nx, ny, nvx, nvy

    def _exited(self, i, px, py):
        for t in self.possible[i]:
            d = self.room["targets"][t]
            if abs(px - d[0]) < d[2] * 0.5 and abs(py - d[1]) < d[3] * 0.5:
                return True
        return False

    def step(self, dt=None, perm=None, noise=None):
        """simulations.py:252-342.  perm/noise given -> used instead of np.random (noise[i] is agent i's)."""
        N = self.N
        if perm is None:
            perm = np.random.choice(np.arange(N), N, replace=False)
        if self.mode == "sequential":
            for i in perm:
                if self.status[i]:
                    des, rep, wall = self._forces(i, self.x, self.y, self.vx, self.vy, self.status)
                    nz = np.random.normal(size=2) if noise is None else noise[i]
                    r = self._advance(i, des, rep, wall, nz, self.x, self.y, self.vx, self.vy)
                    self.x[i], self.y[i], self.vx[i], self.vy[i] = r
                    self.agent_time[i] += self.dt
                    if self._exited(i, self.x[i], self.y[i]):
                        self.status[i] = False
                        self.inside -= 1
        else:
            x, y, vx, vy, st = (a.copy() for a in (self.x, self.y, self.vx, self.vy, self.status))
            for i in perm:
                if st[i]:
                    des, rep, wall = self._forces(i, x, y, vx, vy, st)
                    nz = np.random.normal(size=2) if noise is None else noise[i]
                    r = self._advance(i, des, rep, wall, nz, x, y, vx, vy)
                    self.x[i], self.y[i], self.vx[i], self.vy[i] = r
                    self.agent_time[i] += self.dt
                    if self._exited(i, self.x[i], self.y[i]):
                        self.status[i] = False
                        self.inside -= 1
        self.time += self.dt
        self.simu_step += 1

    def density(self, sigma=None):
        sigma = self.cfg["sigma_convolution"] if sigma is None else sigma
        return gaussian_density(self.x, self.y, self.status, self.X, self.Y, self.V, sigma,
                                fixed_norm="density_norm" in self.fixes)

    def solve_all(self):
        m = self.density() * (self.simu_step > 0)
        for k in self.key_names:
            # 'horizon' (Q9 corrected): every solve covers the full horizon, slice s = time s*dt
            self.targets[k].compute_optimal_velocity(0.0 if "horizon" in self.fixes else self.time, m)

    def run(self, on_step=None):
        """simulations.py:401-451 (without history/plots; on_step(sim) is called where write_history is)."""
        self.solve_all()
        rs = self.cfg["recompute_frequency"]
        while (self.inside > 0) and (self.time < self.T):
            if self.recompute and (self.simu_step % rs == 0) and self.simu_step > 0:
                self.solve_all()
            if on_step is not None:
                on_step(self)
            self.step()
