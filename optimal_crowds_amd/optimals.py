"""`optimals.optimals(room, V, T, target)` -- the reference's HJB / floor-field object
(reference optimals.py:11-250) backed by the HIP solver.

compute_optimal_velocity(t, m) runs the Cole-Hopf HJB solve (RK45 with SciPy's controller and dense output)
on the GPU and leaves in HBM what the direction field is evaluated from -- by default the accepted steps' interpolants
(the "ring": 7.8 GB per key at 2048^2, T = 20), from which a lookup computes the one or two cells it reads; vx_opt / vy_opt
are array-like views that materialise slices on demand (all of them would be 33.5 GB per key at 2048^2, nt = 1000).
"""
import numpy as np
import torch

from . import host


class DeviceField:
    """Read-only array-like over a (nslice, Ny-2, Nx-2) device tensor; indexing returns float64 NumPy."""

    def __init__(self, tensor):
        self._t = tensor

    @property
    def shape(self):
        return tuple(self._t.shape)

    @property
    def dtype(self):
        return np.dtype(np.float64)

    def __len__(self):
        return self._t.shape[0]

    def __getitem__(self, idx):
        return self._t[idx].double().cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self._t.double().cpu().numpy()
        return a if dtype is None else a.astype(dtype)


class RingField:
    """vx_opt / vy_opt of a key whose solve kept the interpolant ring (the default): the (nslice, Ny-2, Nx-2) array of
    the reference (optimals.py:80-81), of which a slice exists only while somebody looks at it -- indexing materialises
    the slices asked for with the slice writer's own kernel (Engine.ring_slices) and returns float64 NumPy.  Slices at
    and beyond the current horizon nt_opt - 1 read as zeros (the reference keeps whatever an earlier solve left there;
    its lookup never reads them, optimals.py:231-232)."""

    def __init__(self, engine, key, comp):
        self._e, self._k, self._c = engine, key, comp

    @property
    def shape(self):
        f = self._e.fields[self._k]
        return (f.nslice, f.rows - 2, self._e.Nx - 2)

    @property
    def dtype(self):
        return np.dtype(np.float64)

    def __len__(self):
        return self.shape[0]

    def _slices(self, lo, hi):
        f = self._e.fields[self._k]
        n = hi - lo
        out = np.zeros((n, f.rows - 2, self._e.Nx - 2))
        top = min(hi, max(f.nt - 1, 0))
        if top > lo:
            t = self._e.ring_slices(self._k, lo, top)[self._c]
            out[:top - lo] = t[:, 1:-1, 1:-1].double().cpu().numpy()
        return out

    def __getitem__(self, idx):
        first, rest = (idx[0], idx[1:]) if isinstance(idx, tuple) else (idx, ())
        n = self.shape[0]
        if isinstance(first, (int, np.integer)):
            i = int(first) + (n if first < 0 else 0)
            if not 0 <= i < n:
                raise IndexError("index %d is out of bounds for axis 0 with size %d" % (first, n))
            a = self._slices(i, i + 1)[0]
            return a[rest] if rest else a
        if isinstance(first, slice):
            lo, hi, st = first.indices(n)
            if st == 1:
                a = self._slices(lo, max(hi, lo))
                return a[(slice(None),) + rest] if rest else a
        return np.asarray(self)[idx]

    def __array__(self, dtype=None, copy=None):
        a = self._slices(0, self.shape[0])
        return a if dtype is None else a.astype(dtype)


class optimals:
    def __init__(self, room, V, T, target, *, config=None, device="cuda:0", dtype="float32", _sim=None,
                 _key_index=0):
        var_config = _sim._cfg if _sim is not None else host.load_config(config)
        var_room = _sim._room if _sim is not None else host.load_room(room)
        self.room_length = var_room['room_length']
        self.room_height = var_room['room_height']
        self.grid_step = var_config['grid_step']
        self.Nx, self.Ny = host.grid_dims(self.room_length, self.room_height, self.grid_step)
        self.dx = self.dy = self.grid_step
        self._X0, self._Y0 = host.axes(self.room_length, self.room_height, self.Nx, self.Ny)
        hp = var_config['hjb_params']
        self.sigma, self.mu = hp['sigma'], hp['mu']
        self.pot, self.pot_target, self.g = hp['wall_potential'], hp['target_potential'], hp['g']
        self.dt = var_config['dt']
        self.T = T
        self.nt_opt = round(self.T / self.dt)
        self.target = target
        # in place, like the reference (optimals.py:89-91); V is None when the simulation rasterised the room on the
        # device (already finalised there): the attribute is then the simulation's lazily downloaded array
        self._V = host.finalize_potential(V, var_config) if V is not None else None
        self.lim = 10e-3
        self.t = 0.
        self.stats = None
        self._k = _key_index
        self._sim = _sim
        self._fix_horizon = _sim is not None and "horizon" in getattr(_sim, "fixes", ())
        if _sim is not None:
            self._engine = _sim._engine
        else:
            from .engine import Engine, hjb_state_dtype
            self._engine = Engine(var_config, self.room_length, self.room_height, self.Nx, self.Ny, self._X0,
                                  self._Y0, 1, 16, device=device, dtype=hjb_state_dtype(var_config, T, dtype))
            rects = np.array([var_room['targets'][t] for t in target.split(' or ') if t in var_room['targets']],
                             dtype=np.float64).reshape(-1, 4)
            self._engine.set_potential(0, self.V, rects)
            self._engine.alloc_field(0, self.nt_opt - 1)

    @property
    def V(self):
        return self._V if self._V is not None else self._sim.Vs[self.target]

    @property
    def X_opt(self):
        return np.meshgrid(self._X0, self._Y0)[0]

    @property
    def Y_opt(self):
        return np.meshgrid(self._X0, self._Y0)[1]

    @property
    def vx_opt(self):
        if self._engine.fields[self._k].is_ring:
            return RingField(self._engine, self._k, 0)
        return DeviceField(self._engine.field_interior(self._k)[0])

    @property
    def vy_opt(self):
        if self._engine.fields[self._k].is_ring:
            return RingField(self._engine, self._k, 1)
        return DeviceField(self._engine.field_interior(self._k)[1])

    def _begin_solve(self, t):
        """the horizon bookkeeping of compute_optimal_velocity (reference optimals.py:139-142) -> (key index, T, nt)"""
        self.t = t
        self.nt_opt = round((self.T - self.t) / self.dt)
        if self._fix_horizon:       # Q9 corrected: always the full horizon, slice s = time s*dt (simulation(fixes=...))
            self.nt_opt = round(self.T / self.dt)
        return self._k, self.T, self.nt_opt

    def _end_solve(self, stats):
        self.stats = stats
        print('Optimal trajectories have been learnt for ' + self.target)

    def compute_optimal_velocity(self, t, m):
        """reference optimals.py:124-206.  m: (Ny,Nx) density (NumPy array or device tensor) or None/0."""
        self._begin_solve(t)
        mt = None
        if m is not None and not np.isscalar(m):
            mt = m if torch.is_tensor(m) else torch.as_tensor(np.ascontiguousarray(m, dtype=np.float64))
        self._end_solve(self._engine.solve(self._k, self.T, self.nt_opt, m=mt)[0])

    def choose_optimal_velocity(self, pos, t):
        """reference optimals.py:212-250 for one position (device lookup; raises IndexError like the reference)."""
        des, err = self._engine.lookup_points(self._k, np.array([pos[0]]), np.array([pos[1]]), int(t))
        if err:
            raise IndexError("index out of bounds for the floor field")
        return des[0]
