"""`pedestrians.ped` -- the reference's agent object (reference pedestrians.py:14-356).

Inside a simulation the agents are SoA buffers in HBM; `sim.agents[i]` is a view that reads agent i's fields
from the device (and its trajectory from the recorder).  A ped constructed directly with the reference's
signature is a standalone agent whose repulsion methods run on the GPU on a 2-agent snapshot.
"""
import numpy as np


class ped:
    def __init__(self, X, Y, step, V, target, all_targets, possible_targets, x, y, vx, vy, room_length, room_height,
                 v_des, a_min, tau_a, b_min, b_max, eta, eta_walls):
        self._sim, self._i = None, -1
        self.initial_position = np.array((x, y), dtype=float)
        self.initial_velocity = np.array((vx, vy), dtype=float)
        self.v_des = v_des
        self.a_min = b_min                   # reference quirk (pedestrians.py:74)
        self.tau_a, self.b_min, self.b_max, self.eta, self.eta_walls = tau_a, b_min, b_max, eta, eta_walls
        self._status = True
        self.target = target
        self.possible_targets = possible_targets
        self.all_targets = all_targets
        self._time = 0
        self.room_length, self.room_height = room_length, room_height
        self._traj = [self.initial_position]
        self._vels = [self.initial_velocity]
        self.X, self.Y, self.V, self.step = X, Y, V, step

    @classmethod
    def _view(cls, sim, i):
        p = cls.__new__(cls)
        p._sim, p._i = sim, i
        p.initial_position = np.array((sim._x0[i], sim._y0[i]), dtype=float)
        p.initial_velocity = np.zeros(2)
        p.v_des = sim._v_des[i]
        p.a_min, p.tau_a, p.b_min, p.b_max = sim.a_min, sim.tau_a, sim.b_min, sim.b_max
        p.eta, p.eta_walls = sim.eta, sim.eta_walls
        p.target = sim._key_names[sim._key_of[i]]
        p.possible_targets = sim._possible[p.target]
        p.all_targets = sim._room["targets"]
        p.room_length, p.room_height = sim.room_length, sim.room_height
        p.V = sim.Vs[p.target]
        p.step = sim.grid_step
        return p

    # ---- state
    def _dev(self, name):
        return float(self._sim._engine.agents[name][self._i].item())

    @property
    def status(self):
        return bool(self._dev("status")) if self._sim is not None else self._status

    @status.setter
    def status(self, v):
        if self._sim is not None:
            raise AttributeError("status of a simulation agent is owned by the device state")
        self._status = v

    @property
    def time(self):
        return self._dev("agent_time") if self._sim is not None else self._time

    @property
    def X_grid(self):
        return self._sim.X_opt if self._sim is not None else self.X

    def _n_records(self):
        return int(round(self.time / self._sim.dt)) + 1

    @property
    def traj(self):
        if self._sim is None:
            return self._traj
        rec = self._sim._rec
        if rec is None:
            raise RuntimeError("trajectories were not recorded (record=False)")
        n = self._n_records()
        a = rec[:n, 0:2, self._i].cpu().numpy()
        return [a[k].copy() for k in range(n)]

    @property
    def vels(self):
        if self._sim is None:
            return self._vels
        rec = self._sim._rec
        if rec is None:
            raise RuntimeError("trajectories were not recorded (record=False)")
        n = self._n_records()
        a = rec[:n, 2:4, self._i].cpu().numpy()
        return [a[k].copy() for k in range(n)]

    def position(self):
        if self._sim is None:
            return np.array(self._traj[-1], dtype=float)
        return np.array((self._dev("x"), self._dev("y")), dtype=float)

    def velocity(self):
        if self._sim is None:
            return np.array(self._vels[-1], dtype=float)
        return np.array((self._dev("vx"), self._dev("vy")), dtype=float)

    def evolve(self, x, y, vx, vy, dt):
        if self._sim is not None:
            raise RuntimeError("simulation agents are advanced by simulation.step() on the device")
        self._traj.append(np.array((x, y), dtype=float))
        self._vels.append(np.array((vx, vy), dtype=float))
        self._time += dt

    def check_status(self):
        x, y = self.position()
        for target in self.possible_targets:
            door = self.all_targets[target]
            if abs(x - door[0]) < door[2] * 0.5 and abs(y - door[1]) < door[3] * 0.5:
                if self._sim is None:
                    self._status = False

    def evac_time(self):
        if self.status:
            raise ValueError('This pedestrian has not exited the room yet!')
        return self.time

    def distance(self, pos):
        x, y = self.position()
        return np.sqrt((pos[0] - x) ** 2 + (pos[1] - y) ** 2)

    # ---- repulsions (reference pedestrians.py:216-334) evaluated by the HIP kernels on a tiny snapshot
    def _probe_engine(self):
        if self._sim is not None:
            return self._sim._engine
        from . import _probe
        return _probe.engine_for(self)

    def agents_repulsion(self, pos_j, vel_j):
        """Repulsion of an agent at pos_j moving with vel_j on this agent (no cutoff applied, like the reference)."""
        eng = self._probe_engine()
        pi, vi = self.position(), self.velocity()
        x = np.array([pi[0], pos_j[0]]); y = np.array([pi[1], pos_j[1]])
        vx = np.array([vi[0], vel_j[0]]); vy = np.array([vi[1], vel_j[1]])

        def fn():
            return eng.pair_force().cpu().numpy()[0]
        f = eng._with_probe(x, y, vx, vy, np.array([self.v_des, 1.0]), np.zeros(2, dtype=np.int32), fn)
        if np.hypot(x[1] - x[0], y[1] - y[0]) >= eng.cfg_struct.repulsion_cutoff:
            raise ValueError("agents_repulsion on the device is defined inside the repulsion cutoff only")
        return f

    def wall_repulsion(self, X, Y, V):
        eng = self._probe_engine()
        pi, vi = self.position(), self.velocity()
        key = 0
        if self._sim is not None:
            key = self._sim._key_names.index(self.target)

        def fn():
            return eng.wall_force()[1].cpu().numpy()[0]
        return eng._with_probe(np.array([pi[0]]), np.array([pi[1]]), np.array([vi[0]]), np.array([vi[1]]),
                               np.array([self.v_des]), np.array([key], dtype=np.int32), fn)
