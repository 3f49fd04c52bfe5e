"""ctypes wrapper of oracle/liboc_oracle.so (C restatement of the hot path).  TEST INFRASTRUCTURE ONLY:
used by tests/ and by bench.py's cpu_baseline leg, never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboc_oracle.so")


class Params(C.Structure):
    _fields_ = [("Ny", C.c_int32), ("Nx", C.c_int32)] + [(n, C.c_double) for n in (
        "dx", "dy", "L", "H", "dt", "sigma", "mu", "g", "lim", "a_min", "tau_a", "b_min", "b_max", "eta", "eta_walls",
        "cutoff", "v_max", "relaxation", "noise_amp")]


def params(cfg, room_length, room_height, Nx, Ny):
    hp = cfg["hjb_params"]
    p = Params()
    p.Ny, p.Nx = Ny, Nx
    p.dx = p.dy = cfg["grid_step"]
    p.L, p.H, p.dt = room_length, room_height, cfg["dt"]
    p.sigma, p.mu, p.g, p.lim = hp["sigma"], hp["mu"], hp["g"], 10e-3
    p.a_min = cfg["b_min"]
    p.tau_a, p.b_min, p.b_max = cfg["tau_a"], cfg["b_min"], cfg["b_max"]
    p.eta, p.eta_walls, p.cutoff = cfg["eta"], cfg["eta_walls"], cfg["repulsion_cutoff"]
    p.v_max, p.relaxation, p.noise_amp = cfg["v_max"], cfg["relaxation"], hp["sigma"] / 2
    return p


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", HERE])
        _lib = C.CDLL(LIB)
        _lib.occ_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def hjb_rhs(P, phi, V, m):
    out = np.empty(P.Ny * P.Nx)
    load().occ_hjb_rhs(C.byref(P), _p(f64(phi)), _p(f64(V)), _p(f64(m)), _p(out))
    return out


def vels(P, phi):
    vx = np.empty((P.Ny - 2, P.Nx - 2))
    vy = np.empty_like(vx)
    load().occ_vels(C.byref(P), _p(f64(phi)), _p(vx), _p(vy))
    return vx, vy


def lincomb(y, Ks, cs):
    out = np.empty_like(y)
    arr = (C.c_void_p * len(Ks))(*[k.ctypes.data for k in Ks])
    cs = f64(cs)
    load().occ_lincomb(_p(y), arr, _p(cs), len(Ks), C.c_size_t(y.size), _p(out))
    return out


def pair_forces(P, pi, vi, vdes, pj, vj):
    pi, vi, vdes, pj, vj = (f64(a) for a in (pi, vi, vdes, pj, vj))
    f = np.empty_like(pi)
    load().occ_pair_forces(C.byref(P), len(pi), _p(pi), _p(vi), _p(vdes), _p(pj), _p(vj), _p(f))
    return f


class WallMap:
    def __init__(self, X0, Y0, V):
        self.X0, self.Y0 = f64(X0), f64(Y0)
        V = f64(V)
        self.Ny, self.Nx = V.shape
        self.vmin = float(V.min())
        self.off = self.vmin * 10e3
        self.left = np.empty(V.shape, dtype=np.int32)
        self.right = np.empty(V.shape, dtype=np.int32)
        load().occ_wallmap_build(self.Ny, self.Nx, _p(V), C.c_double(self.vmin), _p(self.left), _p(self.right))

    def query(self, px, py):
        px, py = f64(np.atleast_1d(px)), f64(np.atleast_1d(py))
        out = np.empty(len(px), dtype=np.int64)
        load().occ_wall_indices(self.Ny, self.Nx, _p(self.X0), _p(self.Y0), _p(self.left), _p(self.right),
                                C.c_double(self.off), len(px), _p(px), _p(py), _p(out))
        return out


def step_sync(P, x, y, vx, vy, vdes, status, key, fields, X0, Y0, wallmaps, rects, noise, i0=0, i1=None):
    """fields: list per key of (vx_slice, vy_slice) float64 (Ny-2, Nx-2) or None; rects: list per key of (n,4)."""
    N = len(x)
    i1 = N if i1 is None else i1
    x, y, vx, vy, vdes, noise = (f64(a) for a in (x, y, vx, vy, vdes, noise))
    status = np.ascontiguousarray(status, dtype=np.uint8)
    key = np.ascontiguousarray(key, dtype=np.int32)
    nk = len(wallmaps)
    keep = []

    def parr(items):
        keep.extend(items)
        return (C.c_void_p * nk)(*[(a.ctypes.data if a is not None else None) for a in items])
    fx = parr([f64(f[0]) if f is not None else None for f in fields])
    fy = parr([f64(f[1]) if f is not None else None for f in fields])
    lefts, rights = parr([w.left for w in wallmaps]), parr([w.right for w in wallmaps])
    offs = f64([w.off for w in wallmaps])
    max_rect = max([len(r) for r in rects] + [1])
    rr = np.zeros((nk, max_rect, 4))
    nrect = np.zeros(nk, dtype=np.int32)
    for k, r in enumerate(rects):
        r = np.asarray(r, dtype=np.float64).reshape(-1, 4)
        rr[k, :len(r)] = r
        nrect[k] = len(r)
    nx, ny, nvx, nvy = (np.empty(N) for _ in range(4))
    nst = np.empty(N, dtype=np.uint8)
    load().occ_step_sync(C.byref(P), N, i0, i1, _p(x), _p(y), _p(vx), _p(vy), _p(vdes), _p(status), _p(key), fx, fy,
                         _p(f64(X0)), _p(f64(Y0)), lefts, rights, _p(offs), _p(rr), _p(nrect), max_rect, _p(noise),
                         _p(nx), _p(ny), _p(nvx), _p(nvy), _p(nst))
    return nx, ny, nvx, nvy, nst


def dense_output(y, Ks, hh, x):
    """SciPy's dense output at one t_eval point (occ_dense_output: rk.py:178-180, 560-575) from a step's y_old and its
    seven stage derivatives Ks; x = (t_eval - t_old) / h."""
    y = f64(y)
    Ks = [f64(k) for k in Ks]
    assert len(Ks) == 7
    out = np.empty_like(y)
    arr = (C.c_void_p * 7)(*[k.ctypes.data for k in Ks])
    lib = load()
    lib.occ_dense_output.restype = None
    lib.occ_dense_output(_p(y), arr, C.c_double(hh), C.c_double(x), C.c_size_t(y.size), _p(out))
    return out


def rk45_solve(P, V, m, T, nt, keep=(), forced=None, rtol=1e-3, atol=1e-6):
    """The whole HJB solve of optimals.py:193-196 in C (occ_rk45_solve).  Returns (phi_keep (len(keep), Ny*Nx), stats);
    forced: step end times to take instead of the controller's (stats['err'] = the error norm of each forced step)."""
    n = P.Ny * P.Nx
    keep = np.ascontiguousarray(keep, dtype=np.int32)
    phi = np.empty((max(len(keep), 1), n))
    t_steps = np.zeros(100000)
    st = np.zeros(4, dtype=np.int64)
    fz = err = None
    if forced is not None:
        fz = f64(forced)
        err = np.zeros(len(fz))
    lib = load()
    lib.occ_rk45_solve.restype = C.c_int
    rc = lib.occ_rk45_solve(C.byref(P), _p(f64(V)), _p(f64(m)), C.c_double(T), C.c_int(nt), C.c_double(rtol),
                            C.c_double(atol), _p(keep), C.c_int(len(keep)), _p(phi), _p(fz) if fz is not None else None,
                            C.c_int(len(fz) if fz is not None else 0), _p(err) if err is not None else None,
                            _p(t_steps), C.c_int(len(t_steps)), _p(st))
    if rc != 0:
        raise RuntimeError("occ_rk45_solve failed (%d)" % rc)
    stats = dict(nfev=int(st[0]), n_attempt=int(st[1]), n_accept=int(st[2]), t_steps=t_steps[:int(st[3])].copy())
    if err is not None:
        stats["err"] = err[:int(st[2])].copy()
    return phi[:len(keep)], stats
