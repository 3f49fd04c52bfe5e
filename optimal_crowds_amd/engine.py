"""Device engine: owns the liboc_hip handle and the HBM buffers (torch tensors used purely as device
memory + stream plumbing) for one simulation.  Everything numerical happens in the HIP library.

Buffers in HBM (row-major, N agents, K keys, grid Ny x Nx, nt0 = round(T/dt)):
  V[k]      float32 (Ny,Nx)      finalised potential of key k ({-100,0,1})
  coef      dtype   (Ny,Nx)      (V + g m)/(mu sigma^2), NaN on walls
  vx[k],vy[k] dtype (nt0-1, Ny, Nx)   direction field of key k, padded rows (33.5 GB/key at 2048^2 fp32);
                                      the reference's (nt0-1, Ny-2, Nx-2) arrays are the interior [:, 1:-1, 1:-1]
  x,y,vx,vy,v_des,agent_time float64 (N);  status uint8 (N);  key int32 (N)
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import (FIX_BITS, OC_ERR_CAPACITY, OC_F32, OC_F64, OC_F64_F32, OcAgents, OcConfig, OcError, OcHjbRing, OcHjbRingJob,
                   OcHjbStats, check)

LIM = 10e-3          # reference optimals.py:95


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def hjb_state_dtype(cfg, T, requested="float32"):
    """The arithmetic type of the HJB state for horizon T.  float32 carries phi * 2^-e (csrc/hjb.hip, range management):
    phi grows to e^(T V_target / (mu sigma^2)) while the far field stays at 1, so the scaled far field 2^-e -- and the
    clip level lim * 2^-e -- leave float32's normal range when log2(e) T V_target / (mu sigma^2) - 64 exceeds ~118
    (T > 25 s with the reference's config).  Beyond that the same kernels run on a float64 state ('mixed': the direction
    field stays float32, so its memory does not double)."""
    if requested not in ("float32", "f32", torch.float32):
        return requested if requested == "mixed" else "float64"
    hp = cfg["hjb_params"]
    bits = 1.4426950408889634 * float(T) * abs(hp["target_potential"]) / (hp["mu"] * hp["sigma"] ** 2)
    return "float32" if bits - 64.0 <= 118.0 else "mixed"


class _RingTooSmall(Exception):
    """a lane's solve ran out of ring slots (OC_ERR_CAPACITY): solve_keys grows the ring in the caller's thread"""


class _Field:
    """Direction field of one key, indexed like the list it replaces: [0], [1] = vx, vy as (nslice, rows, Nx) device
    tensors, [2] = allocated slices, [3] = current horizon nt_opt.  In 'ring' mode the key holds an interpolant ring
    (oc_hjb_solve_ring) and the tensors are MATERIALISED from it on first access (oc_hjb_ring_slices: the slice writer's
    own arithmetic) and dropped at the next solve; the agent step never needs them."""

    def __init__(self, eng, key, nslice, rows, held=True):
        self.eng, self.key, self.nslice, self.rows, self.held = eng, key, nslice, rows, held
        self.nt = 0
        self.vx = self.vy = None        # slices (mode 'slices'), or the materialised cache of a ring
        self.ring = None                # dict(data, meta, slots, struct) once a ring solve has run
        self.is_ring = False

    def __getitem__(self, i):
        if i == 2:
            return self.nslice
        if i == 3:
            return self.nt
        if i not in (0, 1):
            raise IndexError(i)
        if not self.held:
            return None
        if self.is_ring and self.vx is None:
            self.vx, self.vy = self.eng.ring_slices(self.key, 0, max(self.nt - 1, 0), full=True)
        return self.vx if i == 0 else self.vy

    def __setitem__(self, i, v):
        if i == 3:
            self.nt = v
        elif i == 0:
            self.vx = v
        elif i == 1:
            self.vy = v
        else:
            raise IndexError(i)

    def __iter__(self):
        return iter((self[0], self[1], self.nslice, self.nt))


class _AgentState(dict):
    """The agents' SoA tensors.  With lazy_state (Engine.lazy_state, the default) a step leaves x / y / vx / vy / agent_time
    in the library's slot-ordered snapshot instead of scattering them into these tensors (oc_hip.h: oc_agents_sync); reading
    one of them here brings them up to date first, so that callers keep seeing "the step updated the state in place"."""
    _LAZY = ("x", "y", "vx", "vy", "agent_time")

    def __init__(self, eng, *a, **kw):
        super().__init__(*a, **kw)
        self._eng = eng

    def __getitem__(self, k):
        if k in self._LAZY:
            self._eng.flush_state()
        return super().__getitem__(k)

    def get(self, k, default=None):
        return self[k] if k in self else default

    def values(self):
        self._eng.flush_state()
        return super().values()

    def items(self):
        self._eng.flush_state()
        return super().items()


class Engine:
    def __init__(self, cfg, room_length, room_height, Nx, Ny, X0, Y0, n_keys, max_agents,
                 device="cuda:0", dtype="float32", wall_search_radius=0.0, field_mode="ring"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.OcError("no HIP device visible: optimal_crowds_amd needs an MI355X (there is no CPU path)")
        self.device = torch.device(device)
        self.dev_index = self.device.index or 0
        # dtype: 'float32' (product), 'float64' (audit: state AND field), 'mixed' (float64 HJB state, float32 field: the
        # reference's direction field to 1e-6 -- no float32 ripple, DESIGN.md 3.1 -- without doubling the field's memory)
        self.mixed = dtype == "mixed"
        self.dtype = OC_F32 if dtype in ("float32", "f32", torch.float32) else OC_F64             # HJB state, coef, m
        self.tdtype = torch.float32 if self.dtype == OC_F32 else torch.float64
        self.field_dtype = OC_F32 if self.mixed else self.dtype                                    # direction field
        self.field_tdtype = torch.float32 if self.field_dtype == OC_F32 else torch.float64
        self.solve_dtype = OC_F64_F32 if self.mixed else self.dtype
        self.Nx, self.Ny, self.n_keys = Nx, Ny, n_keys
        # field_mode: 'ring' (default) -- a solve keeps the accepted steps' interpolants (oc_hjb_solve_ring) and lookups
        # evaluate the direction from them: the slices nobody reads are never written (reference optimals.py:200-204
        # fills all of them, step() reads one, :231-248); 'slices' -- every slice is written (oc_hjb_solve), as the
        # reference does.  The ring needs the vector kernels (Nx % 4 == 0); other grids use slices.
        if field_mode not in ("ring", "slices"):
            raise ValueError("field_mode must be 'ring' or 'slices'")
        self.field_mode = field_mode if Nx % 4 == 0 else "slices"
        hp = cfg["hjb_params"]
        c = OcConfig()
        c.Ny, c.Nx = Ny, Nx
        c.dx = c.dy = cfg["grid_step"]
        c.room_length, c.room_height = room_length, room_height
        c.dt = cfg["dt"]
        c.sigma, c.mu, c.g = hp["sigma"], hp["mu"], hp["g"]
        c.lim, c.rtol, c.atol = LIM, cfg.get("rtol", 1e-3), cfg.get("atol", 1e-6)
        c.a_min = cfg["b_min"]        # reference quirk: a_min := b_min (simulations.py:88)
        c.tau_a, c.b_min, c.b_max = cfg["tau_a"], cfg["b_min"], cfg["b_max"]
        c.eta, c.eta_walls = cfg["eta"], cfg["eta_walls"]
        c.repulsion_cutoff, c.v_max, c.relaxation = cfg["repulsion_cutoff"], cfg["v_max"], cfg["relaxation"]
        c.noise_amp = hp["sigma"] / 2
        c.wall_search_radius = wall_search_radius
        c.max_agents, c.max_keys = max(int(max_agents), 1), max(int(n_keys), 1)
        self.cfg_struct = c
        X0 = np.ascontiguousarray(X0, dtype=np.float64)
        Y0 = np.ascontiguousarray(Y0, dtype=np.float64)
        h = C.c_void_p()
        check(self.lib.oc_create(self.dev_index, C.byref(c), X0.ctypes.data_as(C.c_void_p),
                                 Y0.ctypes.data_as(C.c_void_p), C.byref(h)))
        self.h = h
        self._axes = (X0, Y0)
        # independent keys' solves run concurrently (solve_keys): lane 0 = this handle on the caller's stream, lanes 1.. =
        # auxiliary handles (their own HJB scratch) on their own streams, created by the first call that needs them
        self._lanes = []
        self._tuning = {}
        self._hjb_variant = 0
        self._prof_mode = 0
        self.overlap_solves = os.environ.get("OC_OVERLAP_SOLVES", "1") != "0"
        # lazy_state: steps keep the agents' state in the library's snapshot; the tensors in self.agents are brought up to
        # date when somebody reads them (_AgentState) -- OC_LAZY_STATE=0 / set_lazy_state(False): scattered in every step
        self._state_behind = False
        self.lazy_state = False
        self.set_lazy_state(os.environ.get("OC_LAZY_STATE", "1") != "0")
        self.V = [None] * n_keys
        self.fields = [None] * n_keys
        self.coef = torch.empty((Ny, Nx), dtype=self.tdtype, device=self.device)
        self.counters = torch.zeros(4, dtype=torch.int32, device=self.device)
        self.agents = None
        self._astruct = None

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            torch.cuda.synchronize(self.device)
            for ln in getattr(self, "_lanes", []):
                self.lib.oc_destroy(ln["h"])
            self._lanes = []
            self.lib.oc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_profiling(self, mode=3):
        """mode bits: 1 HJB kernels, 2 GCFM kernels every step, 4 GCFM kernels every 16th step; 0/False off."""
        self._prof_mode = int(mode)
        check(self.lib.oc_set_profiling(self.h, int(mode)))
        for ln in self._lanes:
            check(self.lib.oc_set_profiling(ln["h"], int(mode)))

    def kernel_stats(self, name):
        """(launches, total_ms, total_algorithmic_bytes) of a kernel class since the last reset (summed over the handles
        of concurrent solves, solve_keys)."""
        tot = [0, 0.0, 0.0]
        for h in [self.h] + [ln["h"] for ln in self._lanes]:
            n, ms, by = C.c_int64(), C.c_double(), C.c_double()
            check(self.lib.oc_kernel_stats(h, name.encode(), C.byref(n), C.byref(ms), C.byref(by)))
            tot = [tot[0] + n.value, tot[1] + ms.value, tot[2] + by.value]
        return tuple(tot)

    def reset_kernel_stats(self):
        check(self.lib.oc_reset_kernel_stats(self.h))
        for ln in self._lanes:
            check(self.lib.oc_reset_kernel_stats(ln["h"]))

    def set_resort_every(self, steps):
        check(self.lib.oc_set_resort_every(self.h, int(steps)))

    def agents_changed(self):
        """Call after modifying the agent tensors in place between steps (tensors taken from self.agents[...] are current:
        reading them there synchronises first)."""
        self._state_behind = False
        check(self.lib.oc_agents_changed(self.h))

    def set_lazy_state(self, on):
        """oc_set_tuning("lazy_state"): leave x / y / vx / vy / agent_time in the library's snapshot between steps"""
        self.flush_state()
        check(self.lib.oc_set_tuning(self.h, b"lazy_state", 1 if on else 0))
        self.lazy_state = bool(on)

    def flush_state(self):
        """bring the tensors of self.agents up to date with the steps enqueued so far (oc_agents_sync; on the current stream)"""
        if self._state_behind:
            self._state_behind = False
            check(self.lib.oc_agents_sync(self.h, self.stream()))

    def set_hjb_variant(self, variant):
        self._hjb_variant = int(variant)
        for ln in self._lanes:
            check(self.lib.oc_set_hjb_variant(ln["h"], int(variant)))
        check(self.lib.oc_set_hjb_variant(self.h, int(variant)))
        if int(variant) != 0:               # the cross-check variants write slices
            self.field_mode = "slices"

    def set_tuning(self, name, value):
        """Launch-shape knobs of the step kernels (oc_hip.h: oc_set_tuning)."""
        check(self.lib.oc_set_tuning(self.h, name.encode(), int(value)))
        self._tuning[name] = int(value)
        if name.startswith("hjb_") or name == "out_tile":       # (the solves' knobs also go to the concurrent solves' handles)
            for ln in self._lanes:
                check(self.lib.oc_set_tuning(ln["h"], name.encode(), int(value)))

    def set_fix_flags(self, names):
        """Opt-in corrections of reference quirks: an iterable of 'lookup', 'wall_force', 'pair_force',
        'density_norm' (oc_hip.h: OC_FIX_*).  Empty = the reference's behaviour (default)."""
        mask = 0
        for n in names:
            if n not in FIX_BITS:
                raise ValueError("unknown fix %r (known: %s)" % (n, ", ".join(sorted(FIX_BITS))))
            mask |= FIX_BITS[n]
        check(self.lib.oc_set_fix_flags(self.h, mask))
        self.fix_flags = mask

    # ------------------------------------------------------------------ per-key data
    def rasterise_potential(self, room, targets, wall_potential, target_potential, finalise=True):
        """Room dict -> potential of the key whose targets are `targets`, rasterised on the device
        (oc_rasterise_potential: reference simulations.py:516-576 + optimals.py:89-91).  float32 (Ny,Nx) tensor."""
        def arr(items, w):
            a = np.ascontiguousarray(np.array(list(items), dtype=np.float64).reshape(-1, w))
            return a, (a.ctypes.data_as(C.c_void_p) if len(a) else None), len(a)
        walls = arr(room["walls"].values(), 4)
        holes = arr(room["holes"].values(), 4)
        cyls = arr(room["cylinders"].values(), 3)
        tg = arr([room["targets"][t] for t in targets], 4)
        out = torch.empty((self.Ny, self.Nx), dtype=torch.float32, device=self.device)
        check(self.lib.oc_rasterise_potential(self.h, walls[1], walls[2], holes[1], holes[2], cyls[1], cyls[2], tg[1], tg[2],
                                              float(wall_potential), float(target_potential), int(bool(finalise)),
                                              _ptr(out), self.stream()))
        return out

    def set_potential(self, key, V_host, rects):
        """V_host: finalised potential (Ny,Nx) of key -- a NumPy array, or a float32 device tensor (rasterise_potential);
        rects: (n,4) exit rectangles of its targets."""
        if torch.is_tensor(V_host):
            Vt = V_host.to(device=self.device, dtype=torch.float32).contiguous()
            vmin = float(Vt.min().item())
        else:
            Vt = torch.as_tensor(np.ascontiguousarray(V_host, dtype=np.float32)).to(self.device)
            vmin = float(V_host.min())
        self.V[key] = Vt
        check(self.lib.oc_wallmap_build(self.h, key, _ptr(Vt), C.c_float(vmin), self.stream()))
        r = np.ascontiguousarray(rects, dtype=np.float64).reshape(-1, 4)
        check(self.lib.oc_set_key_targets(self.h, key, r.ctypes.data_as(C.c_void_p), len(r)))

    # ------------------------------------------------------------------ several GPUs
    def enable_sharding(self, halo_comm, step_comm):
        """From now on this engine is one rank of a multi-GPU simulation with REPLICATED agents: `solve` solves this
        rank's row block of the HJB grid (oc_hjb_solve_rows through halo_comm, a parallel.DistHaloComm /
        ThreadHaloComm on a RowShard) and keeps only that block of the direction field, `step` does this rank's share
        of the forces and lookups and sums them over the ranks (oc_gcfm_step_sharded through step_comm).  Every rank
        must make the same calls with the same arguments; the agent states stay bit-identical across the ranks."""
        comms = halo_comm if isinstance(halo_comm, dict) else {k: halo_comm for k in range(self.n_keys)}
        for c in comms.values():
            if c is not None and c.shard.world > 1 and c.shard.field_margin < 1:
                raise ValueError("sharded lookups need halo >= 8")
        # per key: the communicator of the rank group that solves it (keys first, then row blocks: parallel
        # .key_row_plan), or None when this rank holds no part of that key's field
        self._halo_comms, self._step_comm = comms, step_comm

    @property
    def sharded(self):
        return getattr(self, "_halo_comms", None) is not None

    def holds_key(self, key):
        return not self.sharded or self._halo_comms.get(key) is not None

    def _key_rows(self, key):
        """the communicator of a key solved in row blocks by several ranks, else None (whole grid here, or not held)"""
        c = self._halo_comms.get(key) if self.sharded else None
        return c if (c is not None and c.shard.world > 1) else None

    def alloc_field(self, key, nslice):
        # slices are stored padded as (Ny, Nx) so that every row is 16-byte aligned for the vector kernels;
        # the reference's (Ny-2, Nx-2) array is the interior block [1:-1, 1:-1] (see field_interior)
        if not self.holds_key(key):             # another rank group solves and answers this key
            self.fields[key] = _Field(self, key, nslice, 0, held=False)
            return None, None
        rc = self._key_rows(key)
        rows = rc.shard.Ny_local if rc is not None else self.Ny                  # row-sharded: this rank's block
        f = self.fields[key] = _Field(self, key, nslice, rows)
        if self.field_mode == "ring":           # (the ring itself is allocated by the first solve, which knows T)
            f.is_ring = True
            return None, None
        shape = (max(nslice, 0), rows, self.Nx)
        f.vx = torch.zeros(shape, dtype=self.field_tdtype, device=self.device)
        f.vy = torch.zeros(shape, dtype=self.field_tdtype, device=self.device)
        return f.vx, f.vy

    # ------------------------------------------------------------------ the interpolant ring
    def _ring(self, key, T, grow=False):
        """the key's ring, allocated for horizon T (oc_hjb_ring_slots_hint) or doubled after OC_ERR_CAPACITY"""
        f = self.fields[key]
        if f.ring is not None and not grow:
            return f.ring
        slots = int(self.lib.oc_hjb_ring_slots_hint(self.h, float(T))) if f.ring is None else 2 * f.ring["slots"]
        return self.make_ring(key, slots)

    def make_ring(self, key, slots):
        """(re)allocate the key's ring with room for `slots` accepted steps"""
        f = self.fields[key]
        f.ring = None                                   # (frees the old buffers before the larger ones are made)
        f.vx = f.vy = None
        slots = max(int(slots), 1)
        per = int(self.lib.oc_hjb_ring_slot_elems(self.h, int(f.rows)))
        nmeta = int(self.lib.oc_hjb_ring_meta_bytes(self.h, self.solve_dtype, slots, int(f.nslice), int(f.rows)))
        data = torch.empty((slots, per), dtype=self.tdtype, device=self.device)
        meta = torch.zeros(nmeta, dtype=torch.uint8, device=self.device)
        st = OcHjbRing()
        st.data, st.meta = data.data_ptr(), meta.data_ptr()
        st.slots, st.nslice, st.rows, st.reserved = slots, int(f.nslice), int(f.rows), 0
        f.ring = dict(data=data, meta=meta, slots=slots, struct=st)
        return f.ring

    def ring_slices(self, key, s_lo, s_hi, full=False):
        """slices [s_lo, s_hi) of a ring-backed key's direction field, written by the slice writer itself
        (oc_hjb_ring_slices): (s_hi - s_lo, rows, Nx) tensors -- or, full=True, (nslice, rows, Nx) with the slices in place
        and zeros elsewhere (what alloc_field's zero-initialised arrays hold beyond a solve's horizon)."""
        f = self.fields[key]
        if not f.is_ring or f.ring is None:
            raise ValueError("key %d has no ring (field_mode='slices', or not solved yet)" % key)
        s_hi = min(int(s_hi), max(f.nt - 1, 0))
        s_lo = min(int(s_lo), s_hi)
        n_out = f.nslice if full else s_hi - s_lo
        vx = torch.zeros((max(n_out, 0), f.rows, self.Nx), dtype=self.field_tdtype, device=self.device)
        vy = torch.zeros_like(vx)
        if s_hi > s_lo:
            ox, oy = (vx[s_lo:], vy[s_lo:]) if full else (vx, vy)
            check(self.lib.oc_hjb_ring_slices(self.h, self.solve_dtype, C.byref(f.ring["struct"]), s_lo, s_hi, _ptr(ox),
                                              _ptr(oy), self.stream()))
        return vx, vy

    def _register_field(self, key):
        """(re-)register the key's field with the agent step: ring or slices, whole grid or this rank's row block"""
        f = self.fields[key]
        if not f.held:
            return
        rc = self._key_rows(key)
        if f.is_ring:
            if f.ring is None:
                return
            sh = rc.shard if rc is not None else None
            check(self.lib.oc_set_field_ring(self.h, key, self.solve_dtype, C.byref(f.ring["struct"]), int(f.nt),
                                             int(sh.row0) if sh else -1, int(sh.g0) if sh else 0, int(sh.g1) if sh else 0))
        elif rc is not None:
            sh = rc.shard
            check(self.lib.oc_set_field_rows(self.h, key, self.field_dtype, _ptr(f.vx), _ptr(f.vy), f.nslice, int(f.nt),
                                             int(sh.row0), int(sh.Ny_local), int(sh.g0), int(sh.g1)))
        else:
            check(self.lib.oc_set_field(self.h, key, self.field_dtype, _ptr(f.vx), _ptr(f.vy), f.nslice, int(f.nt)))

    def field_interior(self, key):
        """(vx_opt, vy_opt) views of shape (nslice, Ny-2, Nx-2), the reference's arrays (optimals.py:80-81)."""
        vx, vy = self.fields[key][0], self.fields[key][1]
        if vx is None:
            raise ValueError("key %d is held by another rank group" % key)
        return vx[:, 1:-1, 1:-1], vy[:, 1:-1, 1:-1]

    def field_slice(self, key, s):
        """(vx_opt[s], vy_opt[s]) of the reference (optimals.py:80-81) as (Ny-2, Nx-2) device tensors, whatever the field
        mode: a view of the stored slice, or -- ring -- that one slice written by oc_hjb_ring_slices (zeros at and beyond
        the horizon nt_opt - 1).  For the 4096 x 4096 room, where all slices at once would be 125 GB."""
        f = self.fields[key]
        if not f.held:
            raise ValueError("key %d is held by another rank group" % key)
        if not 0 <= s < f.nslice:
            raise IndexError("slice %d of %d" % (s, f.nslice))
        if f.is_ring and f.vx is None:
            vx, vy = self.ring_slices(key, s, s + 1)
            if vx.shape[0] == 0:
                vx = torch.zeros((1, f.rows, self.Nx), dtype=self.field_tdtype, device=self.device)
                vy = torch.zeros_like(vx)
            return vx[0, 1:-1, 1:-1], vy[0, 1:-1, 1:-1]
        return f.vx[s, 1:-1, 1:-1], f.vy[s, 1:-1, 1:-1]

    def make_coef(self, key, m=None, out=None):
        out = self.coef if out is None else out
        if m is not None:
            m = m.to(device=self.device, dtype=self.tdtype).contiguous()
        check(self.lib.oc_hjb_make_coef(self.h, self.dtype, _ptr(self.V[key]), _ptr(m), _ptr(out), self.stream()))
        return out

    def rhs(self, phi, coef=None):
        coef = self.coef if coef is None else coef
        out = torch.empty_like(phi)
        check(self.lib.oc_hjb_rhs(self.h, self.dtype, _ptr(phi), _ptr(coef), _ptr(out), self.stream()))
        return out

    def ctl_power(self, x):
        """err ** -0.2 as the device step controller forms it (test hook: oc_hjb_ctl_power); x: float64 device tensor"""
        out = torch.empty_like(x)
        check(self.lib.oc_hjb_ctl_power(self.h, _ptr(x), x.numel(), _ptr(out), self.stream()))
        return out

    def vels(self, phi64, scale=1.0):
        vx = torch.zeros((self.Ny, self.Nx), dtype=self.field_tdtype, device=self.device)
        vy = torch.zeros_like(vx)
        check(self.lib.oc_hjb_vels(self.h, self.field_dtype, _ptr(phi64), float(scale), _ptr(vx), _ptr(vy), self.stream()))
        return vx, vy

    def solve(self, key, T, nt, m=None, keep=None, want_steps=False):
        """HJB solve for `key` with horizon nt = round((T - t)/dt): slices 0..nt-2 of the key's field -- written
        (field_mode 'slices'), or kept as the accepted steps' interpolants and evaluated at lookup ('ring').
        keep: list of t_eval indices whose phi is returned (float64, (len, Ny, Nx))."""
        f = self.fields[key]
        nslice = f.nslice
        if nt - 1 > nslice:
            raise ValueError("horizon exceeds allocated field slices")
        if not self.holds_key(key):
            f.nt = nt
            return None, None
        rc = self._key_rows(key)
        if rc is not None:
            if want_steps:
                raise ValueError("want_steps is not available on a row-sharded key")
            st, _, phi_keep = self.solve_rows(key, T, nt, rc, m=m, keep=keep, field=(f.vx, f.vy), _own=True)
            f.nt = nt
            self._register_field(key)
            return st, phi_keep
        st, phi_keep = self._solve_whole(key, T, nt, m, keep, want_steps, None)
        f.nt = nt
        self._register_field(key)
        return st, phi_keep

    def _solve_whole(self, key, T, nt, m, keep, want_steps, lane):
        """the library call of a whole-grid solve of `key` on a lane (None: this handle, its coef buffer, the caller's
        stream; else a dict of solve_keys: handle, coef buffer -- the caller has made the lane's stream current).  Does not
        touch the handle's key tables: the caller registers the field afterwards."""
        f = self.fields[key]
        h = self.h if lane is None else lane["h"]
        coef = self.coef if lane is None else lane["coef"]
        if m is not None:
            m = m.to(device=self.device, dtype=self.tdtype).contiguous()
        check(self.lib.oc_hjb_make_coef(h, self.dtype, _ptr(self.V[key]), _ptr(m), _ptr(coef), self.stream()))
        stats = OcHjbStats()
        keep_arr = phi_keep = None
        n_keep = 0
        if keep is not None and len(keep):
            keep_arr = np.ascontiguousarray(keep, dtype=np.int32)
            n_keep = len(keep_arr)
            phi_keep = torch.empty((n_keep, self.Ny, self.Nx), dtype=torch.float64, device=self.device)
        cap = 100000 if want_steps else 0
        tsteps = np.zeros(max(cap, 1))
        kp = keep_arr.ctypes.data_as(C.c_void_p) if keep_arr is not None else None
        tp = tsteps.ctypes.data_as(C.c_void_p) if cap else None
        if f.is_ring:
            if lane is None:
                f.vx = f.vy = None                  # (a materialised copy of the previous solve; a lane's caller did this)
            ring = self._ring(key, T)
            while True:
                rcode = self.lib.oc_hjb_solve_ring(h, self.solve_dtype, _ptr(coef), float(T), int(nt),
                                                   C.byref(ring["struct"]), kp, n_keep, _ptr(phi_keep), tp, cap,
                                                   C.byref(stats), self.stream())
                if rcode != OC_ERR_CAPACITY:
                    break
                if lane is not None:                # a lane thread allocates and frees nothing: the caller's thread grows
                    raise _RingTooSmall(key)        # the ring under ITS stream and solves the key again (solve_keys)
                ring = self._ring(key, T, grow=True)        # more accepted steps than slots: a larger ring, again
            check(rcode)
        else:
            check(self.lib.oc_hjb_solve(h, self.solve_dtype, _ptr(coef), float(T), int(nt), _ptr(f.vx), _ptr(f.vy),
                                        kp, n_keep, _ptr(phi_keep), tp, cap, C.byref(stats), self.stream()))
        st = stats.as_dict()
        if want_steps:
            st["t_steps"] = tsteps[:st["n_accept"] + 1].copy()
        return st, phi_keep

    # ------------------------------------------------------------------ independent keys solved concurrently
    def _lane(self, i):
        """lane i >= 1 of solve_keys: an auxiliary handle (same configuration; it only ever runs HJB solves, so what it
        allocates is the solver's scratch -- 12 state arrays), a stream and a coef buffer of its own"""
        while len(self._lanes) < i:
            h = C.c_void_p()
            check(self.lib.oc_create(self.dev_index, C.byref(self.cfg_struct), self._axes[0].ctypes.data_as(C.c_void_p),
                                     self._axes[1].ctypes.data_as(C.c_void_p), C.byref(h)))
            check(self.lib.oc_set_hjb_variant(h, self._hjb_variant))
            for name, value in self._tuning.items():
                if name.startswith("hjb_") or name == "out_tile":
                    check(self.lib.oc_set_tuning(h, name.encode(), value))
            check(self.lib.oc_set_profiling(h, self._prof_mode))
            self._lanes.append(dict(h=h, stream=torch.cuda.Stream(device=self.device),
                                    coef=torch.empty((self.Ny, self.Nx), dtype=self.tdtype, device=self.device)))
        return self._lanes[i - 1]

    def solve_keys(self, jobs, max_lanes=4):
        """Solve several keys: jobs = [(key, T, nt, m), ...] -> list of stats (None for a key another rank group holds).
        The reference solves its keys one after the other (simulations.py:419-420, 432-435); the solves are independent,
        and with the interpolant ring and the exact active set one of them is a chain of ~100 short, latency-bound
        launches that leaves most of the chip idle -- so keys whose whole grid lives on this GPU (ring fields) are solved
        CONCURRENTLY, each on its own handle and stream, by ONE library call per round of `max_lanes` keys
        (oc_hjb_solve_ring_many: the host threads are the library's).  Same kernels, same arithmetic, same results as
        solve() key by key (tests/test_hip_ring.py).  OC_OVERLAP_SOLVES=0 / overlap_solves = False: one after the other."""
        jobs = list(jobs)
        par = [q for q, (key, T, nt, m) in enumerate(jobs)
               if self.holds_key(key) and self._key_rows(key) is None and nt - 1 <= self.fields[key].nslice
               and self.fields[key].is_ring]
        if not self.overlap_solves or len(par) < 2:
            return [self.solve(key, T, nt, m=m)[0] for key, T, nt, m in jobs]
        out = [None] * len(jobs)
        for q, (key, T, nt, m) in enumerate(jobs):      # (row blocks / keys held elsewhere: the ordinary path, first)
            if q not in par:
                out[q] = self.solve(key, T, nt, m=m)[0]
        n_lanes = min(len(par), max(int(max_lanes), 1))
        lanes = [None] + [self._lane(i) for i in range(1, n_lanes)]
        main = torch.cuda.current_stream(self.device)
        # everything the lanes read is put in place HERE, by the caller's thread on the caller's stream: the densities go to
        # the device once (a lane thread does no host-to-device copy and frees no tensor), the rings exist
        on_dev = {}
        for q in par:
            key, T, nt, m = jobs[q]
            if m is not None and id(m) not in on_dev:
                on_dev[id(m)] = m.to(device=self.device, dtype=self.tdtype).contiguous()
            if self.fields[key].is_ring:
                self.fields[key].vx = self.fields[key].vy = None    # (a materialised copy of the previous solve)
                self._ring(key, T)
        jobs = [(key, T, nt, on_dev.get(id(m), m) if m is not None else None) for key, T, nt, m in jobs]
        ready = torch.cuda.Event()
        ready.record(main)                  # the potentials, the density and the last steps' lookups of the old rings
        regrow = []                         # keys whose ring was too small (OC_ERR_CAPACITY in a lane): redone below
        # ONE library call per round of n_lanes keys (oc_hjb_solve_ring_many): the library runs the jobs on host threads of its
        # own -- nothing of the interpreter lies between the solves' launches (threads started from here cost 0.3 - 0.7 ms per
        # round of two 2 ms solves).  Lane 0 = this handle on the caller's stream.
        for r0 in range(0, len(par), n_lanes):
            batch = par[r0:r0 + n_lanes]
            arr = (OcHjbRingJob * len(batch))()
            keep_alive = []
            for li, q in enumerate(batch):
                key, T, nt, m = jobs[q]
                ln = lanes[li]
                st = main if ln is None else ln["stream"]
                if ln is not None:
                    st.wait_event(ready)
                jb = arr[li]
                jb.h = (self.h if ln is None else ln["h"]).value
                jb.coef_dtype, jb.dtype = self.dtype, self.solve_dtype
                jb.V = self.V[key].data_ptr()
                jb.m = m.data_ptr() if m is not None else None
                jb.coef = (self.coef if ln is None else ln["coef"]).data_ptr()
                jb.T, jb.nt = float(T), int(nt)
                jb.ring = C.pointer(self.fields[key].ring["struct"])
                jb.stream = st.cuda_stream
                keep_alive.append(m)
            rcode = self.lib.oc_hjb_solve_ring_many(arr, len(batch))
            for li, q in enumerate(batch):
                if arr[li].rc == OC_ERR_CAPACITY:
                    regrow.append(q)
                elif arr[li].rc != 0:
                    raise OcError("liboc_hip: error %d: %s" % (arr[li].rc, arr[li].error.decode(errors="replace")))
                else:
                    out[q] = arr[li].stats.as_dict()
            del rcode
        for q in sorted(regrow):            # by the caller's thread on the caller's stream: the larger ring belongs to its pool
            key, T, nt, m = jobs[q]
            self._ring(key, T, grow=True)
            out[q] = self._solve_whole(key, T, nt, m, None, False, None)[0]
        # (every solve call returned after its stream had drained: whatever follows on the caller's stream is ordered
        # behind all of them by the host)
        for q in par:
            key, T, nt, m = jobs[q]
            self.fields[key].nt = nt
            self._register_field(key)
        return out

    def solve_rows(self, key, T, nt, comm, m=None, keep=None, field=None, _own=False):
        """This rank's part of a row-block HJB solve (oc_hjb_solve_rows; `comm`: parallel.DistHaloComm or
        ThreadHaloComm built on a parallel.RowShard).  Returns (stats, (vx, vy), phi_keep): the direction field
        (nt-1, Ny_local, Nx) and kept phi (len(keep), Ny_local, Nx) cover the LOCAL rows comm.shard.rows().
        (_own: called by solve() for a key whose field this engine manages -- with field_mode 'ring' the row block's
        interpolants go to the key's ring, oc_hjb_solve_rows_ring, and (vx, vy) is (None, None).)"""
        sh = comm.shard
        self.make_coef(key, m)
        coef_rows = self.coef[sh.rows()]
        use_ring = _own and self.fields[key].is_ring
        if use_ring:
            pass
        elif field is None:
            shape = (max(nt - 1, 0), sh.Ny_local, self.Nx)
            field = (torch.zeros(shape, dtype=self.field_tdtype, device=self.device),
                     torch.zeros(shape, dtype=self.field_tdtype, device=self.device))
        vx, vy = (None, None) if use_ring else field
        stats = OcHjbStats()
        keep_arr = phi_keep = None
        n_keep = 0
        if keep is not None and len(keep):
            keep_arr = np.ascontiguousarray(keep, dtype=np.int32)
            n_keep = len(keep_arr)
            phi_keep = torch.empty((n_keep, sh.Ny_local, self.Nx), dtype=torch.float64, device=self.device)
        st = comm.struct()
        kp = keep_arr.ctypes.data_as(C.c_void_p) if n_keep else None
        if use_ring:
            f = self.fields[key]
            f.vx = f.vy = None
            ring = self._ring(key, T)
            while True:
                rc = self.lib.oc_hjb_solve_rows_ring(self.h, self.solve_dtype, _ptr(coef_rows), float(T), int(nt),
                                                     C.byref(ring["struct"]), C.byref(st), kp, n_keep, _ptr(phi_keep),
                                                     C.byref(stats), self.stream())
                if rc != OC_ERR_CAPACITY or comm.error is not None:
                    break
                ring = self._ring(key, T, grow=True)    # (every rank sees the same step count: all grow together)
        else:
            rc = self.lib.oc_hjb_solve_rows(self.h, self.solve_dtype, _ptr(coef_rows), float(T), int(nt), _ptr(vx), _ptr(vy),
                                            C.byref(st), kp, n_keep, _ptr(phi_keep), C.byref(stats), self.stream())
        if comm.error is not None:
            err, comm.error = comm.error, None
            raise err
        check(rc)
        return stats.as_dict(), (vx, vy), phi_keep

    def set_field_rows(self, key, vx, vy, nt, shard):
        """Register a ROW BLOCK of key's direction field (as returned by solve_rows on `shard`, a parallel.RowShard):
        this rank answers the lookups whose first cell lies in the rows it owns (oc_set_field_rows)."""
        if shard.world > 1 and shard.field_margin < 1:
            raise ValueError("a lookup's second cell may lie one row beyond the owned rows: the row block needs "
                             "halo >= 8 (the field is exact halo - 7 rows beyond the block)")
        nslice = int(vx.shape[0])
        f = self.fields[key] = _Field(self, key, nslice, int(shard.Ny_local))
        f.vx, f.vy, f.nt = vx, vy, int(nt)
        check(self.lib.oc_set_field_rows(self.h, key, self.field_dtype, _ptr(vx), _ptr(vy), nslice, int(nt), int(shard.row0),
                                         int(shard.Ny_local), int(shard.g0), int(shard.g1)))

    def set_field_horizon(self, key, nt):
        self.fields[key].nt = nt
        self._register_field(key)

    # ------------------------------------------------------------------ agents
    def set_agents(self, x, y, vx, vy, v_des, key, status=None, agent_time=None):
        dev, f64 = self.device, torch.float64
        N = len(x)
        mk = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        st = np.ones(N, dtype=np.uint8) if status is None else np.ascontiguousarray(status, dtype=np.uint8)
        # lazy_state: the library may still owe the OLD tensors a flush (state_dirty + raw pointers to them).  Pay it while
        # they are alive, then drop the pending state -- a later flush would write into blocks the allocator has handed on.
        if getattr(self, "agents", None) is not None:
            self.flush_state()
            check(self.lib.oc_agents_sync(self.h, self.stream()))
        check(self.lib.oc_agents_changed(self.h))
        self._state_behind = False
        self.agents = _AgentState(
            self, N=N, x=mk(x), y=mk(y), vx=mk(vx), vy=mk(vy), v_des=mk(v_des),
            status=torch.as_tensor(st).to(dev),
            key=torch.as_tensor(np.ascontiguousarray(key, dtype=np.int32)).to(dev),
            agent_time=mk(np.zeros(N) if agent_time is None else agent_time))
        a = OcAgents()
        a.N = N
        for f in ("x", "y", "vx", "vy", "v_des", "status", "key", "agent_time"):
            setattr(a, f, self.agents[f].data_ptr())
        self._astruct = a

    def get_state(self):
        a = self.agents
        return {k: a[k].cpu().numpy() for k in ("x", "y", "vx", "vy", "status", "agent_time")}

    def lookup(self, simu_step):
        N = self.agents["N"]
        des = torch.empty((N, 2), dtype=torch.float64, device=self.device)
        err = torch.zeros(2, dtype=torch.int32, device=self.device)
        check(self.lib.oc_field_lookup(self.h, C.byref(self._astruct), int(simu_step), _ptr(des), _ptr(err),
                                       self.stream()))
        return des, int(err[0].item())

    def _with_probe(self, x, y, vx, vy, v_des, key, fn):
        """Run fn() with a temporary agent set (single-point queries of the facade), then restore."""
        self.flush_state()
        saved = (self.agents, self._astruct)
        try:
            self.set_agents(x, y, vx, vy, v_des, key)
            return fn()
        finally:
            self.agents, self._astruct = saved

    def lookup_points(self, key, xs, ys, simu_step):
        n = len(xs)

        def fn():
            des, err = self.lookup(simu_step)
            return des.cpu().numpy(), err
        return self._with_probe(xs, ys, np.zeros(n), np.zeros(n), np.ones(n), np.full(n, key, dtype=np.int32), fn)

    def wall_force(self):
        N = self.agents["N"]
        idx = torch.empty(N, dtype=torch.int64, device=self.device)
        f = torch.empty((N, 2), dtype=torch.float64, device=self.device)
        check(self.lib.oc_wall_force(self.h, C.byref(self._astruct), _ptr(idx), _ptr(f), self.stream()))
        return idx, f

    def wall_last_index(self):
        """the nearest-wall cell each agent's LAST step used (oc_wall_last_index; -1: none within its search radius)"""
        N = self.agents["N"]
        idx = torch.empty(N, dtype=torch.int32, device=self.device)
        check(self.lib.oc_wall_last_index(self.h, C.byref(self._astruct), _ptr(idx), self.stream()))
        return idx

    def pair_force(self):
        N = self.agents["N"]
        f = torch.empty((N, 2), dtype=torch.float64, device=self.device)
        check(self.lib.oc_pair_force(self.h, C.byref(self._astruct), _ptr(f), self.stream()))
        return f

    def step(self, simu_step, noise=None, seed=0, sync=True):
        """One synchronous GCFM step in place.  noise: (N,2) float64 tensor/array or None (device Philox)."""
        if self.sharded:
            return self.step_sharded(simu_step, self._step_comm, noise=noise, seed=seed, sync=sync)
        if noise is not None and not torch.is_tensor(noise):
            noise = torch.as_tensor(np.ascontiguousarray(noise, dtype=np.float64)).to(self.device)
        check(self.lib.oc_gcfm_step(self.h, C.byref(self._astruct), int(simu_step), _ptr(noise),
                                    C.c_uint64(int(seed)), _ptr(self.counters), self.stream()))
        self._state_behind = self.lazy_state
        if sync:
            c = self.counters.cpu().numpy()
            return int(c[0]), int(c[1]), int(c[2])
        return None

    def step_sequential(self, simu_step, perm, noise=None, seed=0):
        """One SEQUENTIAL step in the reference's update order (oc_gcfm_step_sequential): agents move one by one in
        the permutation `perm` and see the already advanced state of those before them (simulations.py:271,
        pedestrians.py:140-162).  Returns (exited, bad, active, levels)."""
        if self.sharded:
            raise ValueError("the sequential step is a single-GPU mode")
        if noise is not None and not torch.is_tensor(noise):
            noise = torch.as_tensor(np.ascontiguousarray(noise, dtype=np.float64)).to(self.device)
        if torch.is_tensor(perm):
            perm = perm.to(device=self.device, dtype=torch.int32).contiguous()
        else:
            perm = torch.as_tensor(np.ascontiguousarray(perm, dtype=np.int32)).to(self.device)
        if perm.numel() != self.agents["N"]:
            raise ValueError("the permutation must list every agent once")
        rounds = C.c_int32(0)
        check(self.lib.oc_gcfm_step_sequential(self.h, C.byref(self._astruct), int(simu_step), _ptr(perm), _ptr(noise),
                                               C.c_uint64(int(seed)), _ptr(self.counters), C.byref(rounds),
                                               self.stream()))
        c = self.counters.cpu().numpy()
        return int(c[0]), int(c[1]), int(c[2]), int(rounds.value)

    def step_sharded(self, simu_step, comm, noise=None, seed=0, sync=True):
        """One synchronous GCFM step on comm.world ranks with replicated agents (oc_gcfm_step_sharded): this rank does
        the pair / wall forces of its share of the tiles and the lookups its field rows cover, `comm` (parallel
        .DistStepComm or ThreadStepComm) sums the exchange buffer over the ranks, every rank finishes all agents."""
        if noise is not None and not torch.is_tensor(noise):
            noise = torch.as_tensor(np.ascontiguousarray(noise, dtype=np.float64)).to(self.device)
        n = int(self.lib.oc_xchg_doubles(int(self.agents["N"]), int(comm.world)))
        if getattr(self, "_xchg", None) is None or self._xchg.numel() != n:
            self._xchg = torch.zeros(n, dtype=torch.float64, device=self.device)
        comm.bind(self._xchg)
        st = comm.struct()
        rc = self.lib.oc_gcfm_step_sharded(self.h, C.byref(self._astruct), int(simu_step), _ptr(noise),
                                           C.c_uint64(int(seed)), _ptr(self.counters), C.byref(st), self.stream())
        if comm.error is not None:
            err, comm.error = comm.error, None
            raise err
        check(rc)
        if sync:
            c = self.counters.cpu().numpy()
            return int(c[0]), int(c[1]), int(c[2])
        return None

    def run_status(self, reset=False):
        """(active agents, 1 + step in which the last agent left or 0, 1 + agent in the IndexError band or 0, its step):
        what run() needs when it does not read the counters after every step (oc_run_status; synchronises)."""
        out = (C.c_int32 * 4)()
        check(self.lib.oc_run_status(self.h, out, int(bool(reset)), self.stream()))
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def density(self, sigma, V_all=None, trunc_sigmas=6.0, out_dtype=None, rows=None):
        """rows = (lo, hi): only those grid rows are computed (oc_density_rows), the others are zero -- what a rank of a
        row-block solve needs (density_rows_needed)"""
        dt = self.dtype if out_dtype is None else out_dtype
        tdt = torch.float32 if dt == OC_F32 else torch.float64
        if rows is None or (rows[0] <= 0 and rows[1] >= self.Ny):
            out = torch.empty((self.Ny, self.Nx), dtype=tdt, device=self.device)
            check(self.lib.oc_density(self.h, dt, C.byref(self._astruct), float(sigma), _ptr(V_all),
                                      float(trunc_sigmas), _ptr(out), self.stream()))
            return out
        out = torch.zeros((self.Ny, self.Nx), dtype=tdt, device=self.device)
        check(self.lib.oc_density_rows(self.h, dt, C.byref(self._astruct), float(sigma), _ptr(V_all), float(trunc_sigmas),
                                       int(max(rows[0], 0)), int(min(rows[1], self.Ny)), _ptr(out), self.stream()))
        return out

    def density_rows_needed(self):
        """the grid rows whose density the HJB solves of THIS rank read (simulations.py:432-435 feeds the density of the whole
        grid to every key's solve): the union of the local row ranges of the keys held here -- every row for a key held whole,
        nothing for a rank that holds no key (its share of the step needs no density)"""
        lo, hi = self.Ny, 0
        for key in range(self.n_keys):
            if not self.holds_key(key):
                continue
            rc = self._key_rows(key)
            if rc is None:
                return (0, self.Ny)
            sh = rc.shard
            lo, hi = min(lo, int(sh.row0)), max(hi, int(sh.row0) + int(sh.Ny_local))
        return (lo, hi) if hi > lo else (0, 0)
