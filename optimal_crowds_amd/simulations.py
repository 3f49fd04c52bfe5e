"""`simulations.simulation(room, T, recompute=False).run()` -- the reference's driver surface
(reference simulations.py:18-451) on top of the MI355X engine.

Same constructor arguments, attributes, printed lines and room / config formats; the per-step work
(floor-field lookup, all-pairs GCFM repulsion, wall repulsion, noisy relaxation step, exit test), the HJB
solves and the density run in HIP kernels through liboc_hip.so.  Agent state lives in HBM as SoA buffers;
`agents[i]` are light views (pedestrians.ped) onto it.

Differences a user can observe (DESIGN.md "Deviations"):
  * the agent update is synchronous by default (all agents see the positions of the step's start); the reference
    updates agents one by one in a random order.  The random permutation is still drawn (noise='host') so that the
    global np.random stream is consumed exactly like the reference's.  update='sequential' reproduces the reference's
    order itself (oc_gcfm_step_sequential: the permutation is level-scheduled on the device), at one pair of launches
    per dependency level instead of one per step.
  * noise='device' (default above `host_noise_max_agents`) draws the step noise with Philox on the GPU.
"""
import warnings

import numpy as np
import torch

from . import host, optimals as optimals_mod, pedestrians
from .engine import Engine, hjb_state_dtype

warnings.filterwarnings("ignore")

HOST_NOISE_MAX_AGENTS = 20000


class _History(dict):
    """history[time] -> frame, built lazily from the device recorder (reference simulations.py:579-589)."""

    def __init__(self, sim):
        super().__init__()
        self._sim = sim
        self._steps = {}

    def _mark(self, time, step):
        self._steps[time] = step
        dict.__setitem__(self, time, None)

    def __getitem__(self, time):
        v = dict.__getitem__(self, time)
        if v is None:
            v = self._sim._frame(self._steps[time])
            dict.__setitem__(self, time, v)
        return v

    def _truncate(self, last_step):
        """drop the frames marked for steps beyond last_step (run() enqueues a few steps past the last exit)"""
        for t in [t for t, st in self._steps.items() if st >= last_step]:
            del self._steps[t]
            dict.__delitem__(self, t)

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]


class _LazyPotentials(dict):
    """Vs[key] -> (Ny,Nx) float64 array like the reference's; with device rasterisation the array is downloaded from
    the GPU the first time somebody asks for it."""

    def __init__(self):
        super().__init__()
        self._dev = {}

    def _set_dev(self, key, tensor):
        self._dev[key] = tensor
        dict.__setitem__(self, key, None)

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if v is None:
            v = self._dev[key].double().cpu().numpy()
            dict.__setitem__(self, key, v)
        return v

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]


RASTER_DEVICE_MIN_CELLS = 1 << 22      # 2048 x 2048 and larger: rasterise the room on the device


class simulation:
    def __init__(self, room, T, recompute=False, *, config=None, device="cuda:0", dtype="float32",
                 noise=None, seed=20240807, history=None, record=None, wall_search_radius=None, sharding=None,
                 fixes=(), rasterise=None, update="synchronous", field="ring"):
        """rasterise: 'host' (NumPy, like the reference), 'device' (oc_rasterise_potential, bit-identical; Vs[key] and
        V become lazily downloaded arrays) or None = 'device' from 2048 x 2048 grid nodes on.
        field: 'ring' (default) -- an HJB solve keeps the accepted steps' interpolants and the step's lookups evaluate the
        direction from them (same values, bit for bit; nothing of the ~950 slices per re-solve that are never read is
        written) -- or 'slices': every slice of vx_opt / vy_opt is written, as the reference does (optimals.py:200-204).
        fixes: opt-in corrections of reference quirks, default none = bug-compatible (SURVEY quirk register):
        'lookup' (Q7/Q8: bilinear interpolation over the four nodes, no IndexError band), 'wall_force' and 'pair_force'
        (Q12: unit vectors, q_i with its square root, no factor 3), 'density_norm' (Q11: 2 pi sigma^2), 'horizon' (Q9: a
        recompute solves the full horizon again, so slice s keeps meaning time s*dt and the field does not go
        stale from t = T/2 on); 'all' = every one of them.
        sharding: None (one GPU) or a callable engine -> (halo_comm, step_comm), e.g. parallel.dist_sharding():
        this process is then one rank of a multi-GPU simulation with replicated agents (Engine.enable_sharding);
        every rank constructs the same simulation (same room, config, seed) and makes the same calls."""
        self.recompute = recompute
        if update not in ("synchronous", "sequential"):
            raise ValueError("update must be 'synchronous' or 'sequential'")
        self.update = update
        self.levels = []            # update='sequential': dependency levels of each step's permutation
        var_config = host.load_config(config)
        var_room = host.load_room(room)
        self._cfg, self._room = var_config, var_room
        self.room_length = var_room["room_length"]
        self.room_height = var_room["room_height"]
        self.grid_step = var_config["grid_step"]
        self.Nx, self.Ny = host.grid_dims(self.room_length, self.room_height, self.grid_step)
        self.dx = self.dy = self.grid_step
        self._X0, self._Y0 = host.axes(self.room_length, self.room_height, self.Nx, self.Ny)
        self.sigma_convolution = var_config["sigma_convolution"]
        self.pot = var_config["hjb_params"]["wall_potential"]
        self.lim = 10e-3
        self.T = T
        self.time = 0.
        self.simu_step = 0
        self.dt = var_config["dt"]
        self.relaxation = var_config["relaxation"]
        self.noise_intensity = var_config["hjb_params"]["sigma"]
        self.a_min = var_config["b_min"]
        self.tau_a = var_config["tau_a"]
        self.b_min = var_config["b_min"]
        self.b_max = var_config["b_max"]
        self.eta = var_config["eta"]
        self.eta_walls = var_config["eta_walls"]
        self.repulsion_cutoff = var_config["repulsion_cutoff"]
        self.v_max = var_config["v_max"]
        self.recompute_step = var_config["recompute_frequency"]

        # ---- keys and the crowd (reference simulations.py:104-160; the sampler never looks at the potentials, Q16)
        self.place_ped = np.zeros((self.Ny, self.Nx))
        key_names, xs, ys, vd, kk, box_keys = [], [], [], [], [], []
        self._possible = {}
        for box in var_room["initial_boxes"].values():
            targets = box[5:]
            key = " or ".join(targets)
            if key not in key_names:
                key_names.append(key)
            self._possible[key] = targets
            box_keys.append(key)
            bx, by, bv = host.sample_box(box, self._X0, self._Y0, self.place_ped)
            xs.append(bx), ys.append(by), vd.append(bv)
            kk.append(np.full(len(bx), key_names.index(key), dtype=np.int32))
        cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dtype=dt)
        x0, y0 = cat(xs, float), cat(ys, float)
        self._v_des = cat(vd, float)
        self._key_of = cat(kk, np.int32)
        self._key_names = key_names
        self.N = len(x0)
        self.inside = self.N
        self._x0, self._y0 = x0.copy(), y0.copy()

        # ---- device side
        nt0 = round(self.T / self.dt)
        self._nt0 = nt0
        if wall_search_radius is None:
            wall_search_radius = -1.0          # per-agent radius beyond which the wall force is < 1e-12 m/s
        # float32 HJB state unless the horizon takes phi beyond what its range management covers (T > ~25 s): 'mixed' then
        # (float64 state, float32 field); dtype='mixed' / 'float64' can also be asked for
        self.hjb_dtype = hjb_state_dtype(var_config, T, dtype)
        self._engine = Engine(var_config, self.room_length, self.room_height, self.Nx, self.Ny, self._X0, self._Y0,
                              len(key_names), self.N, device=device, dtype=self.hjb_dtype,
                              wall_search_radius=wall_search_radius, field_mode=field)
        eng = self._engine
        # ---- potentials (reference simulations.py:104-121, 516-576; finalised like optimals.__init__ does, Q3)
        if rasterise is None:
            rasterise = "device" if self.Nx * self.Ny >= RASTER_DEVICE_MIN_CELLS else "host"
        if rasterise not in ("host", "device"):
            raise ValueError("rasterise must be 'host', 'device' or None")
        self._raster_dev = rasterise == "device"
        self.Vs = _LazyPotentials()
        hp = var_config["hjb_params"]
        if self._raster_dev:
            for key in key_names:
                self.Vs._set_dev(key, eng.rasterise_potential(var_room, self._possible[key], hp["wall_potential"],
                                                              hp["target_potential"]))
            V_all = torch.ones((self.Ny, self.Nx), dtype=torch.float32, device=eng.device)
            for key in box_keys:                        # one term per BOX, like the reference's loop
                V_all += self.Vs._dev[key]
            V_all[V_all > V_all.min()] = 0
            self._V_all_dev, self._V_host = V_all, None
        else:
            self._V_host = np.zeros((self.Ny, self.Nx)) + 1
            for key in box_keys:
                if key not in self.Vs:
                    V = host.create_potential(var_room, self._possible[key], self._X0, self._Y0)
                    self.Vs[key] = host.finalize_potential(V, var_config)
                self._V_host += self.Vs[key]
            self._V_host[self._V_host > np.min(self._V_host)] = 0
            self._V_all_dev = torch.as_tensor(self._V_host.astype(np.float32)).to(eng.device)
        fixes = {fixes} if isinstance(fixes, str) else set(fixes)
        if "all" in fixes:
            fixes = {"lookup", "wall_force", "pair_force", "density_norm", "horizon"}
        self.fixes = frozenset(fixes)
        eng.set_fix_flags(self.fixes - {"horizon"})
        if sharding is not None:
            eng.enable_sharding(*sharding(eng))
        self.targets = {}
        for k, key in enumerate(key_names):
            rects = np.array([var_room["targets"][t] for t in self._possible[key]], dtype=np.float64)
            eng.set_potential(k, self.Vs._dev[key] if self._raster_dev else self.Vs[key], rects)
            eng.alloc_field(k, nt0 - 1)
            self.targets[key] = optimals_mod.optimals(room, None if self._raster_dev else self.Vs[key], T, key,
                                                      _sim=self, _key_index=k)
        eng.set_agents(x0, y0, np.zeros(self.N), np.zeros(self.N), self._v_des, self._key_of)
        self._noise_mode = noise or ("host" if self.N <= HOST_NOISE_MAX_AGENTS or update == "sequential" else "device")
        if update == "sequential" and sharding is not None:
            raise ValueError("update='sequential' is a single-GPU mode")
        self._seed = seed
        self._status_host = np.ones(self.N, dtype=bool)
        # trajectory recorder (ped.traj / ped.vels / history): (steps+1, 4, N) float64 in HBM when it fits
        if record is None:
            record = (nt0 + 3) * max(self.N, 1) * 32 <= 8 << 30
        self._rec = None
        if record and self.N > 0:
            self._rec = torch.empty((nt0 + 3, 4, self.N), dtype=torch.float64, device=eng.device)
            self._rec_status = torch.empty((nt0 + 3, self.N), dtype=torch.uint8, device=eng.device)
            eng.set_lazy_state(False)       # (the recorder reads the state after every step: keep it current in the arrays)
            self._record(0)
        self._history_on = (self._rec is not None) if history is None else bool(history)
        self.history = _History(self)
        self._agents = None
        print('ABM simulation room created!')

    # ------------------------------------------------------------------ reference attributes
    @property
    def V(self):
        """1 + sum of the boxes' potentials, everything above the minimum set to 0 (reference simulations.py:119-121)"""
        if self._V_host is None:
            self._V_host = self._V_all_dev.double().cpu().numpy()
        return self._V_host

    @V.setter
    def V(self, value):
        self._V_host = np.asarray(value, dtype=float)
        self._V_all_dev = torch.as_tensor(self._V_host.astype(np.float32)).to(self._engine.device)

    @property
    def X_opt(self):
        return np.meshgrid(self._X0, self._Y0)[0]

    @property
    def Y_opt(self):
        return np.meshgrid(self._X0, self._Y0)[1]

    @property
    def agents(self):
        if self._agents is None:
            self._agents = np.array([pedestrians.ped._view(self, i) for i in range(self.N)], dtype=object)
        return self._agents

    # ------------------------------------------------------------------ recorder
    def _record(self, slot):
        if self._rec is None or slot >= self._rec.shape[0]:
            return
        a = self._engine.agents
        r = self._rec[slot]
        r[0].copy_(a["x"]), r[1].copy_(a["y"]), r[2].copy_(a["vx"]), r[3].copy_(a["vy"])
        self._rec_status[slot].copy_(a["status"])

    def _frame(self, step):
        if self._rec is None:
            raise RuntimeError("history needs the trajectory recorder (record=True)")
        st = self._rec[step].cpu().numpy()
        act = self._rec_status[step].cpu().numpy().astype(bool)
        frame = [[np.array((st[0, i], st[1, i])), np.array((st[2, i], st[3, i])),
                  self._key_names[self._key_of[i]], self._v_des[i]] for i in np.nonzero(act)[0]]
        frame.append(self._density_of(self._rec[step][0], self._rec[step][1], self._rec_status[step],
                                      self.sigma_convolution))
        return frame

    def write_history(self, time):
        """reference simulations.py:579-589; frames (agents + density) are materialised on access."""
        if self._history_on:
            self.history._mark(time, self.simu_step)

    # ------------------------------------------------------------------ step (reference simulations.py:252-342)
    def step(self, dt, verbose=False):
        eng = self._engine
        if dt != self.dt:
            raise ValueError("the device step uses config dt (%g)" % self.dt)
        noise = perm = None
        if self._noise_mode == "host" or self.update == "sequential":
            # same draws, in the same order, as the reference: one permutation, then 2 normals per active agent
            perm = np.random.choice(np.arange(self.N), self.N, replace=False)
        if self._noise_mode == "host":
            noise = np.zeros((self.N, 2))
            act = perm[self._status_host[perm]]
            noise[act] = np.random.normal(size=(len(act), 2))
        if self.update == "sequential":     # the reference's own order: reference simulations.py:271, pedestrians.py:140-162
            exited, bad, active, levels = eng.step_sequential(self.simu_step, perm, noise=noise, seed=self._seed)
            self.levels.append(levels)
        else:
            exited, bad, active = eng.step(self.simu_step, noise=noise, seed=self._seed)
        if exited:
            self.inside -= exited
            self._status_host = eng.agents["status"].cpu().numpy().astype(bool)
        if bad:
            # (the device has applied the whole synchronous step; the reference raises in the middle of its loop over
            # the agents, before time / simu_step advance)
            raise IndexError("agent %d left the region covered by the floor field (index %d is out of bounds)"
                             % (bad - 1, bad - 1))
        self.time += dt
        self.simu_step += 1
        self._record(self.simu_step)
        if verbose:
            print('t = {:.2f}s exit = {}/{}'.format(self.time, self.N - self.inside, self.N) + 10 * ' ', end='\n')

    def evac_times(self, draw=False):
        if self.inside > 0:
            raise ValueError('There are still people inside!')
        return self._engine.agents["agent_time"].cpu().numpy().copy()

    def initial_positions(self):
        return self._x0.copy(), self._y0.copy()

    # ------------------------------------------------------------------ run (reference simulations.py:401-451)
    def _solve_all(self):
        """every key's compute_optimal_velocity(time, m) (reference simulations.py:419-420, 432-435).  The solves are
        independent: the engine runs those whose whole grid lives on this GPU concurrently (Engine.solve_keys); the
        bookkeeping and the printed lines stay in the reference's key order."""
        m = self._density_dev(self.sigma_convolution) if self.simu_step > 0 else None
        tg = list(self.targets.values())
        if len(tg) < 2 or any(type(o).compute_optimal_velocity is not optimals_mod.optimals.compute_optimal_velocity
                              or "compute_optimal_velocity" in vars(o) for o in tg):
            for key in self.targets:        # (one key, or a caller that wrapped the method: the plain loop)
                self.targets[key].compute_optimal_velocity(self.time, m)
            return
        jobs = [o._begin_solve(self.time) for o in tg]
        stats = self._engine.solve_keys([(k, T, nt, m) for k, T, nt in jobs])
        for o, st in zip(tg, stats):
            o._end_solve(st)

    def _run_enqueued(self):
        """The loop of run() without a host round trip per step (device noise, no per-step printing): steps are
        enqueued and the device-side run status is read every LOOK steps and before every recompute.  The status says
        in which step the last agent left (or an agent entered the IndexError band), so the loop ends -- and `time`,
        `simu_step`, the recorder -- stand exactly where the step-by-step loop would have ended: steps enqueued
        after the last exit change nothing (no active agent) and are not counted."""
        LOOK = 16
        eng = self._engine
        eng.run_status(reset=True)
        first = self.simu_step
        while (self.inside > 0) & (self.time < self.T):
            if self.recompute and (self.simu_step % self.recompute_step == 0) and self.simu_step > 0:
                print(f"Computing trajectories at time {self.time}")
                self._solve_all()
            self.write_history(self.time)
            eng.step(self.simu_step, noise=None, seed=self._seed, sync=False)
            self.time += self.dt
            self.simu_step += 1
            self._record(self.simu_step)
            if (self.simu_step - first) % LOOK == 0 or not (self.time < self.T) or \
                    (self.recompute and self.simu_step % self.recompute_step == 0):
                active, done, bad, bad_step = eng.run_status()
                if bad:
                    # the reference raises inside the offending step (before time / simu_step advance); up to LOOK - 1
                    # further steps were already enqueued: the state goes back to where the step-by-step loop stops,
                    # the offending step applied
                    self.simu_step, self.time = bad_step, self._time_at(first, bad_step)
                    self._restore(bad_step + 1)
                    raise IndexError("agent %d left the region covered by the floor field (index %d is out of "
                                     "bounds)" % (bad - 1, bad - 1))
                self.inside = active
                if done:
                    self.simu_step, self.time = done, self._time_at(first, done)
                    self.history._truncate(done)            # write_history marks the frame BEFORE step s as s
        self._status_host = eng.agents["status"].cpu().numpy().astype(bool)

    def _restore(self, step):
        """agent state := the recorder's slot `step` (the state after `step` steps).  Without a recorder
        (record=False) the state after an IndexError raised by the enqueued loop is that of the last enqueued step."""
        if self._rec is None or step >= self._rec.shape[0]:
            return
        a, r = self._engine.agents, self._rec[step]
        a["x"].copy_(r[0]), a["y"].copy_(r[1]), a["vx"].copy_(r[2]), a["vy"].copy_(r[3])
        a["status"].copy_(self._rec_status[step])
        alive = (self._rec_status[:step + 1] != 0).sum(dim=0).clamp(max=step)      # steps each agent has taken
        times = np.concatenate([[0.0], np.cumsum(np.full(step, self.dt))]) if step > 0 else np.zeros(1)
        a["agent_time"].copy_(torch.as_tensor(times, device=alive.device)[alive])
        self._status_host = a["status"].cpu().numpy().astype(bool)
        self.inside = int(self._status_host.sum())
        self.history._truncate(step)
        self._engine.agents_changed()

    def _time_at(self, first_step, step):
        """`time` after the loop has advanced from first_step to step, accumulated like the reference does (time += dt)"""
        t = self._time_first
        for _ in range(step - first_step):
            t += self.dt
        return t

    def run(self, verbose=False, draw=False, mode='scatter'):
        print(f"Computing trajectories at time {self.time}")
        self._solve_all()
        if self._noise_mode == "device" and self.update == "synchronous" and not verbose and not draw:
            self._time_first = self.time
            self._run_enqueued()
        while (self.inside > 0) & (self.time < self.T):
            if self.recompute and (self.simu_step % self.recompute_step == 0) and self.simu_step > 0:
                print(f"Computing trajectories at time {self.time}")
                self._solve_all()
            self.write_history(self.time)
            self.step(self.dt, verbose=verbose)
            if draw and (self.simu_step % 10) == 0:
                self.draw(mode)
        if self.inside == 0:
            print('Evacuation complete in {:.2f}s!'.format(self.time))
        else:
            print('Evacuation failed!' + 10 * ' ')

    # ------------------------------------------------------------------ density (reference simulations.py:453-487)
    def _density_dev(self, sigma):
        # (sharded: only the rows this rank's solves read -- the agents are replicated, each rank splats into its own row
        # block without communication)
        eng = self._engine
        rows = eng.density_rows_needed() if getattr(eng, "key_plan", None) else None
        return eng.density(sigma, self._V_all_dev, rows=rows)

    def _density_of(self, x, y, status, sigma):
        eng = self._engine
        eng.flush_state()
        saved, sstruct = eng.agents, eng._astruct
        try:
            eng.set_agents(x.cpu().numpy(), y.cpu().numpy(), np.zeros(self.N), np.zeros(self.N), self._v_des,
                           self._key_of, status=status.cpu().numpy())
            return eng.density(sigma, self._V_all_dev, out_dtype=1).cpu().numpy()
        finally:
            eng.agents, eng._astruct = saved, sstruct

    def gaussian_density(self, sigma):
        return self._engine.density(sigma, self._V_all_dev, out_dtype=1).cpu().numpy()

    def create_potential(self, var_room, targets):
        return host.create_potential(var_room, targets, self._X0, self._Y0)

    # ------------------------------------------------------------------ plotting: out of scope
    def draw(self, mode='scatter'):
        raise NotImplementedError("plotting is outside the accelerated path; read sim.agents / sim.history")

    draw_final_trajectories = draw_history = draw
