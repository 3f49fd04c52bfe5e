#!/usr/bin/env python3
"""Generate the parity fixtures under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference).  Nothing of the reference is
copied: it is imported from a staging directory outside the repo, driven on rooms authored
by optimal_crowds_amd/room_library.py, and only inputs + outputs (numbers) are stored.

    python tests/golden/make_goldens.py [G1 G2 ...]      # default: all

Staging recipe: the reference has no package __init__ and opens
'optimal_crowds/config.json' and 'rooms/<room>.json' relative to the cwd, so we create
/tmp/oc_stage/{optimal_crowds -> /root/reference, rooms/} and chdir there; `seaborn`
(import-only dependency of the plotting code) is replaced by an empty module.

Fixtures (each .npz also stores numpy/scipy versions):
  G1_<room>.npz   potential grids, X/Y axes, sampler output (positions, v_des, keys)
  G2_rhs.npz      HJB right-hand side on seeded random + real phi
  G3_solve_<room>.npz  RK45: accepted step times, nfev, phi at selected t_eval slices
  G4 (inside G3)  vx_opt / vy_opt at selected slices
  G5_lookup.npz   choose_optimal_velocity on probing positions (incl. IndexError band)
  G6_forces.npz   pair force and wall force (+ argmin index) on seeded random inputs
  G7_step.npz     one sequential reference step and one frozen-snapshot synchronous step
  G8_run_*.npz    full run() of room_test (T=20) and a recompute=True run of gold_b
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
STAGE = "/tmp/oc_stage"
REFERENCE = "/root/reference"



def _load_by_path(name, path):
    """Load one source file as a stand-alone module WITHOUT putting its directory on sys.path.  The repository root
    holds a regular package `optimal_crowds/` (the drop-in alias of the MI355X implementation); with the root on the
    path that package would win over the staged reference (a namespace portion) and this script would record the
    implementation's own outputs as "reference" fixtures."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


room_library = _load_by_path("_oc_room_library", os.path.join(REPO, "optimal_crowds_amd", "room_library.py"))


def _under(path, root):
    path, root = os.path.realpath(path), os.path.realpath(root)
    return path == root or path.startswith(root + os.sep)


def stage(stage_dir=STAGE, reference=REFERENCE):
    """Stage the reference for import and return its three modules.  Refuses (RuntimeError) unless every one of them
    was loaded from a file under `reference` (reference simulations.py:10-11 imports `from optimal_crowds import
    pedestrians / optimals`, hence the directory name)."""
    if not os.path.isdir(reference):
        raise RuntimeError("%s is absent: fixtures can only be generated in the build container" % reference)
    os.makedirs(os.path.join(stage_dir, "rooms"), exist_ok=True)
    link = os.path.join(stage_dir, "optimal_crowds")
    if os.path.islink(link) and os.path.realpath(link) != os.path.realpath(reference):
        os.unlink(link)
    if not os.path.islink(link):
        os.symlink(reference, link)
    room_library.write_rooms(os.path.join(stage_dir, "rooms"))
    os.chdir(stage_dir)
    # nothing of the repository may answer to the name `optimal_crowds`: drop the repo root (and '' / '.' = a cwd that
    # could be the repo root) from the path, forget modules a caller imported before
    repo_real = os.path.realpath(REPO)
    sys.path[:] = [p for p in sys.path
                   if p not in ("", ".") and os.path.realpath(p) != repo_real]
    for name in [m for m in sys.modules if m == "optimal_crowds" or m.startswith("optimal_crowds.")]:
        del sys.modules[name]
    import importlib
    importlib.invalidate_caches()
    sys.path.insert(0, stage_dir)
    sys.modules["seaborn"] = types.ModuleType("seaborn")
    import matplotlib
    matplotlib.use("Agg")
    from optimal_crowds import simulations, optimals, pedestrians
    check_is_reference((simulations, optimals, pedestrians), reference)
    return simulations, optimals, pedestrians


def check_is_reference(mods, reference=REFERENCE):
    for m in mods:
        f = getattr(m, "__file__", None)
        if not f or not _under(f, reference):
            raise RuntimeError("refusing to write fixtures: %s was loaded from %s, not from %s"
                               % (m.__name__, f, reference))


def versions():
    import scipy
    return dict(numpy_version=np.__version__, scipy_version=scipy.__version__)


OUT = HERE          # --out DIR writes elsewhere (e.g. to compare a regeneration with the committed fixtures)
_MODS = ()


def save(name, **arrs):
    check_is_reference(_MODS)          # again at every write: a fixture is only ever the reference's output
    if not _MODS:
        raise RuntimeError("refusing to write fixtures: stage() has not run")
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs, **versions())
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024))


class SolveRecorder:
    """Wraps the `solve_ivp` name inside the reference's optimals module."""

    def __init__(self, optimals_mod):
        self.mod = optimals_mod
        self.real = optimals_mod.solve_ivp
        self.calls = []

    def __enter__(self):
        def wrapped(fun, t_span, y0, **kw):
            sol = self.real(fun, t_span, y0, **kw)
            self.calls.append(dict(fun=fun, t_span=t_span, y0=y0.copy(), kw=kw, sol=sol))
            return sol
        self.mod.solve_ivp = wrapped
        return self

    def __exit__(self, *a):
        self.mod.solve_ivp = self.real


def agent_state(sim):
    x = np.array([a.position()[0] for a in sim.agents])
    y = np.array([a.position()[1] for a in sim.agents])
    vx = np.array([a.velocity()[0] for a in sim.agents])
    vy = np.array([a.velocity()[1] for a in sim.agents])
    st = np.array([a.status for a in sim.agents])
    return x, y, vx, vy, st


def keys_of(sim):
    names = list(sim.targets)
    return names, np.array([names.index(a.target) for a in sim.agents])


# ----------------------------------------------------------------------------- G1
def g1(simulations, optimals, pedestrians):
    for room, seed in (("gold_a", 11), ("gold_b", 12), ("room_test", 1)):
        np.random.seed(seed)
        sim = simulations.simulation(room, 4.0)
        names, kidx = keys_of(sim)
        x, y, vx, vy, st = agent_state(sim)
        arrs = dict(seed=seed, Nx=sim.Nx, Ny=sim.Ny, X0=sim.X_opt[0], Y0=sim.Y_opt[:, 0],
                    V_all=sim.V, key_names=np.array(names), key_of_agent=kidx,
                    x=x, y=y, v_des=np.array([a.v_des for a in sim.agents]),
                    place_ped=sim.place_ped.astype(np.int8),
                    rng_next=np.random.random_sample(4))
        for k, name in enumerate(names):
            arrs["V_%d" % k] = sim.Vs[name]
        save("G1_%s.npz" % room, **arrs)


# ----------------------------------------------------------------------- G2 G3 G4
def g234(simulations, optimals, pedestrians):
    from scipy.integrate import RK45
    rhs_saved = False
    for room, T, seed in (("gold_a", 4.0, 11), ("gold_a", 6.0, 11), ("gold_b", 3.0, 12)):
        np.random.seed(seed)
        sim = simulations.simulation(room, T)
        names, _ = keys_of(sim)
        key = names[-1]
        opt = sim.targets[key]
        # non-trivial density input m (the reference passes zero at step 0; a recompute passes
        # gaussian_density): use the real density of the initial crowd
        m = sim.gaussian_density(sim.sigma_convolution)
        with SolveRecorder(optimals) as rec:
            opt.compute_optimal_velocity(0.0, m)
        call = rec.calls[0]
        sol = call["sol"]
        fun = call["fun"]
        nt = opt.nt_opt
        # accepted step times via scipy's own solver class on the captured RHS
        solver = RK45(lambda t, y: fun(t, y, m), T, call["y0"], 0.0)
        h0 = solver.h_abs
        ts = [solver.t]
        while solver.status == "running":
            solver.step()
            ts.append(solver.t)
        assert solver.nfev == sol.nfev, (solver.nfev, sol.nfev)
        sl = np.unique(np.concatenate([[0, 1, 2, 3, nt // 4, nt // 2, nt - 3, nt - 2, nt - 1],
                                       np.linspace(5, nt - 5, 6).astype(int)]))
        vs = np.array([nt - 1 - k for k in sl if 1 <= k <= nt - 1])   # vx_opt index of t_eval[k]
        save("G3_solve_%s_T%d.npz" % (room, int(T)),
             room=room, T=T, seed=seed, key=key, V=sim.Vs[key], m=m, nt=nt,
             t_eval=call["kw"]["t_eval"], nfev=sol.nfev, h_abs0=h0, t_steps=np.array(ts),
             slices=sl, phi=sol.y[:, sl].T.reshape(len(sl), sim.Ny, sim.Nx),
             v_slices=vs, vx=opt.vx_opt[vs], vy=opt.vy_opt[vs])
        if not rhs_saved:
            rng = np.random.RandomState(5)
            phis = [np.exp(rng.normal(size=sim.Ny * sim.Nx) * 2.0),
                    rng.uniform(0.0, 3.0, size=sim.Ny * sim.Nx),
                    sol.y[:, nt // 2].copy(), sol.y[:, nt - 1].copy()]
            outs = [fun(0.0, p.copy(), m) for p in phis]
            outs0 = [fun(0.0, p.copy(), 0 * m) for p in phis]
            save("G2_rhs.npz", room=room, V=sim.Vs[key], m=m, phi=np.array(phis),
                 rhs_m=np.array(outs), rhs_0=np.array(outs0))
            rhs_saved = True


# ----------------------------------------------------------------------------- G5
def g5(simulations, optimals, pedestrians):
    np.random.seed(11)
    T = 2.0
    sim = simulations.simulation("gold_a", T)
    names, _ = keys_of(sim)
    opt = sim.targets[names[0]]
    opt.compute_optimal_velocity(0.0, 0 * sim.V)
    rng = np.random.RandomState(7)
    L, H, dx = sim.room_length, sim.room_height, sim.dx
    pos = np.column_stack([rng.uniform(0, L, 160), rng.uniform(0, H, 160)])
    special = [(dx * 0.5, 1.0), (dx, 1.0), (dx * 1.0000001, 1.0), (1.0, dx * 0.5), (1.0, dx),
               (L - dx, 2.0), (L - 0.5 * dx, 2.0), (L - 1.5 * dx, 2.0), (L - 2.5 * dx, 2.0),
               ((sim.Nx - 3) * dx, 2.0), ((sim.Nx - 3) * dx - 1e-9, 2.0),
               (2.0, H - dx), (2.0, H - 0.5 * dx), (2.0, H - 1.5 * dx), (2.0, (sim.Ny - 3) * dx),
               (2.0, (sim.Ny - 3) * dx - 1e-9), (0.0, 0.0), (L, H), (0.3, 0.3), (0.15, 0.15),
               (2.0 + 1e-12, 2.0), (1.0, 1.0), (1.05, 1.05), (3.0, 3.0 - 1e-13)]
    pos = np.vstack([pos, np.array(special)])
    steps = np.concatenate([rng.randint(0, opt.nt_opt - 1, len(pos) - 6),
                            [opt.nt_opt - 2, opt.nt_opt - 1, opt.nt_opt, opt.nt_opt + 5, 0, 1]])
    rng.shuffle(steps)
    out = np.zeros((len(pos), 2))
    err = np.zeros(len(pos), dtype=np.int8)      # 1 = IndexError
    for n, (p, s) in enumerate(zip(pos, steps)):
        try:
            out[n] = opt.choose_optimal_velocity(p, int(s))
        except IndexError:
            err[n] = 1
            out[n] = np.nan
    # after a recompute the horizon shrinks (nt_opt changes, stale slices remain)
    opt.compute_optimal_velocity(0.5, 0 * sim.V)
    nt2 = opt.nt_opt
    out2 = np.zeros((len(pos), 2))
    err2 = np.zeros(len(pos), dtype=np.int8)
    for n, (p, s) in enumerate(zip(pos, steps)):
        try:
            out2[n] = opt.choose_optimal_velocity(p, int(s))
        except IndexError:
            err2[n] = 1
            out2[n] = np.nan
    save("G5_lookup.npz", room="gold_a", T=T, V=sim.Vs[names[0]], key=names[0], pos=pos, steps=steps,
         out=out, err=err, nt_opt=round(T / sim.dt), nt_opt2=nt2, t2=0.5, out2=out2, err2=err2)


# ----------------------------------------------------------------------------- G6
def g6(simulations, optimals, pedestrians):
    np.random.seed(12)
    sim = simulations.simulation("gold_b", 2.0)
    names, _ = keys_of(sim)
    cfg = json.load(open("optimal_crowds/config.json"))
    rng = np.random.RandomState(21)
    n = 2000
    pi = rng.uniform(1, 7, (n, 2))
    # mostly near pairs, some far
    rad = np.where(rng.rand(n) < 0.8, rng.uniform(0.05, 1.2, n), rng.uniform(1.2, 6.0, n))
    ang = rng.uniform(0, 2 * np.pi, n)
    pj = pi + np.column_stack([rad * np.cos(ang), rad * np.sin(ang)])
    vi = rng.normal(0, 0.9, (n, 2))
    vj = rng.normal(0, 0.9, (n, 2))
    vdes = rng.normal(1.34, 0.26, n)
    # special cases: stationary i, stationary j, both, head-on, j exactly behind, axis aligned
    vi[:40] = 0.0
    vj[20:60] = 0.0
    vi[60:80] = (pj[60:80] - pi[60:80])           # heading straight at j
    vi[80:100] = -(pj[80:100] - pi[80:100])       # j exactly behind
    vi[100:110, 1] = 0.0
    vj[110:120, 0] = 0.0
    vi[120:130] *= 3.0                            # faster than v_des
    force = np.zeros((n, 2))
    for k in range(n):
        a = pedestrians.ped(sim.Y_opt, sim.X_opt, sim.grid_step, sim.Vs[names[0]], names[0], {}, [],
                            pi[k, 0], pi[k, 1], vi[k, 0], vi[k, 1], sim.room_length, sim.room_height,
                            vdes[k], sim.a_min, sim.tau_a, sim.b_min, sim.b_max, sim.eta, sim.eta_walls)
        force[k] = a.agents_repulsion(pj[k], vj[k])
    # wall force on both potentials
    nw = 300
    wp = np.column_stack([rng.uniform(0.1, 7.9, nw), rng.uniform(0.1, 5.9, nw)])
    wv = rng.normal(0, 0.9, (nw, 2))
    wv[:30] = 0.0
    wvd = rng.normal(1.34, 0.26, nw)
    # exact-tie probes: points equidistant from wall cells (on grid nodes / cell centres)
    X0, Y0 = sim.X_opt[0], sim.Y_opt[:, 0]
    wp[30:40, 0] = X0[rng.randint(5, 150, 10)]
    wp[30:40, 1] = Y0[rng.randint(5, 110, 10)]
    wp[40:50, 0] = 0.5 * (X0[60:70] + X0[61:71])
    wp[40:50, 1] = 3.0
    wforce = np.zeros((len(names), nw, 2))
    windex = np.zeros((len(names), nw), dtype=np.int64)
    for kk, name in enumerate(names):
        V = sim.Vs[name]
        for k in range(nw):
            a = pedestrians.ped(sim.Y_opt, sim.X_opt, sim.grid_step, V, name, {}, [],
                                wp[k, 0], wp[k, 1], wv[k, 0], wv[k, 1], sim.room_length, sim.room_height,
                                wvd[k], sim.a_min, sim.tau_a, sim.b_min, sim.b_max, sim.eta, sim.eta_walls)
            wforce[kk, k] = a.wall_repulsion(sim.X_opt, sim.Y_opt, V)
            # index the reference picked: recover it from the returned vector is ill-posed, so
            # evaluate the same selection rule on the same arrays (numpy argmin, first minimum)
            d = np.sqrt((sim.X_opt - wp[k, 0]) ** 2 + (sim.Y_opt - wp[k, 1]) ** 2)
            windex[kk, k] = np.argmin(d + V * 10e3)
    save("G6_forces.npz", room="gold_b", pi=pi, pj=pj, vi=vi, vj=vj, vdes=vdes, force=force,
         a_min=sim.a_min, tau_a=sim.tau_a, b_min=sim.b_min, b_max=sim.b_max, eta=sim.eta,
         eta_walls=sim.eta_walls, wp=wp, wv=wv, wvdes=wvd, wforce=wforce, windex=windex,
         key_names=np.array(names), V_0=sim.Vs[names[0]], V_1=sim.Vs[names[1]],
         X0=X0, Y0=Y0)


# ----------------------------------------------------------------------------- G7
def g7(simulations, optimals, pedestrians):
    np.random.seed(12)
    T = 3.0
    sim = simulations.simulation("gold_b", T)
    names, kidx = keys_of(sim)
    for name in names:
        sim.targets[name].compute_optimal_velocity(0.0, 0 * sim.V)
    for _ in range(70):                           # warm-up: a moving, interacting crowd
        sim.step(sim.dt)
    x, y, vx, vy, st = agent_state(sim)
    step0 = sim.simu_step
    vdes = np.array([a.v_des for a in sim.agents])
    # --- frozen-snapshot synchronous step composed from the reference's own methods.
    seed_step = 777
    np.random.seed(seed_step)
    perm = np.random.choice(np.arange(sim.N), sim.N, replace=False)
    noise = np.zeros((sim.N, 2))
    for i in perm:
        if sim.agents[i].status:
            noise[i] = np.random.normal(size=2)
    des = np.zeros((sim.N, 2)); rep = np.zeros((sim.N, 2)); wall = np.zeros((sim.N, 2))
    for i in range(sim.N):
        a = sim.agents[i]
        if not a.status:
            continue
        des[i] = a.v_des * sim.targets[a.target].choose_optimal_velocity(a.position(), sim.simu_step)
        for j in range(sim.N):
            b = sim.agents[j]
            if b.status and j != i and a.distance(b.position()) < sim.repulsion_cutoff:
                rep[i] = rep[i] + np.array(a.agents_repulsion(b.position(), b.velocity()), dtype=float)
        wall[i] = a.wall_repulsion(sim.X_opt, sim.Y_opt, sim.Vs[a.target])
    # integration formulas of the step driver (reference simulations.py:303-326), applied to the
    # frozen forces; exit test by the reference's own check_status on a scratch copy afterwards
    dt = sim.dt
    cv = np.column_stack([vx, vy]) + sim.noise_intensity / 2 * noise + rep + wall
    acc = (des - cv) / sim.relaxation
    nx_ = x + cv[:, 0] * dt + 0.5 * acc[:, 0] * dt ** 2
    ny_ = y + cv[:, 1] * dt + 0.5 * acc[:, 1] * dt ** 2
    nv = cv + acc * dt
    nrm = np.sqrt(nv[:, 0] ** 2 + nv[:, 1] ** 2)
    sc = np.where(nrm < sim.v_max, 1.0, sim.v_max / nrm)
    nv = nv * sc[:, None]
    act = st.copy()
    nx_ = np.where(act, nx_, x); ny_ = np.where(act, ny_, y)
    nv = np.where(act[:, None], nv, np.column_stack([vx, vy]))
    # --- the reference's true sequential step from the same snapshot and RNG seed
    np.random.seed(seed_step)
    sim.step(sim.dt)
    sx, sy, svx, svy, sst = agent_state(sim)
    # synchronous exit status: same rectangles, evaluated on the synchronous positions
    targets = json.load(open("rooms/gold_b.json"))["targets"]
    syn_st = st.copy()
    for i in range(sim.N):
        if st[i]:
            for tname in sim.agents[i].possible_targets:
                d = targets[tname]
                if abs(nx_[i] - d[0]) < d[2] * 0.5 and abs(ny_[i] - d[1]) < d[3] * 0.5:
                    syn_st[i] = False
    save("G7_step.npz", room="gold_b", T=T, seed=12, warm_steps=70, simu_step=step0, seed_step=seed_step,
         x=x, y=y, vx=vx, vy=vy, status=st, v_des=vdes, key_of_agent=kidx, key_names=np.array(names),
         perm=perm, noise=noise, des=des, rep=rep, wall=wall,
         syn_x=nx_, syn_y=ny_, syn_vx=nv[:, 0], syn_vy=nv[:, 1], syn_status=syn_st,
         seq_x=sx, seq_y=sy, seq_vx=svx, seq_vy=svy, seq_status=sst)


# ----------------------------------------------------------------------------- G8
def g8(simulations, optimals, pedestrians):
    import io
    import contextlib
    for tag, room, T, seed, rec in (("room_test", "room_test", 20.0, 1, False),
                                    ("gold_b_recompute", "gold_b", 4.0, 12, True)):
        np.random.seed(seed)
        sim = simulations.simulation(room, T, recompute=rec)
        nts = []
        orig = {}
        for name, opt in sim.targets.items():
            orig[name] = opt.compute_optimal_velocity

            def wrapped(t, m, _o=opt, _f=orig[name]):
                _f(t, m)
                nts.append((sim.simu_step, _o.nt_opt))
            opt.compute_optimal_velocity = wrapped
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            sim.run()
        printed = buf.getvalue()
        nsteps = sim.simu_step
        every = 50
        idx = np.arange(0, nsteps + 1, every)
        N = sim.N
        traj = np.full((len(idx), N, 2), np.nan)
        vel = np.full((len(idx), N, 2), np.nan)
        for i, a in enumerate(sim.agents):
            tr = np.array(a.traj); ve = np.array(a.vels)
            for n, s in enumerate(idx):
                if s < len(tr):
                    traj[n, i] = tr[s]; vel[n, i] = ve[s]
        times = np.array([a.time for a in sim.agents])
        hist_t = np.array(sorted(sim.history))
        inside_t = np.array([len(sim.history[t]) - 1 for t in hist_t])
        dens_idx = np.array([0, len(hist_t) // 3, len(hist_t) - 1])
        dens = np.array([sim.history[hist_t[k]][-1] for k in dens_idx])
        names, kidx = keys_of(sim)
        save("G8_run_%s.npz" % tag, room=room, T=T, seed=seed, recompute=rec, nsteps=nsteps,
             final_time=sim.time, inside=sim.inside, idx=idx, traj=traj, vel=vel,
             agent_time=times, status=np.array([a.status for a in sim.agents]),
             n_traj=np.array([len(a.traj) for a in sim.agents]),
             hist_t=hist_t, inside_t=inside_t, dens_idx=dens_idx, dens=dens,
             nt_after=np.array(nts), printed=np.array(printed), key_of_agent=kidx,
             key_names=np.array(names), v_des=np.array([a.v_des for a in sim.agents]))


ALL = dict(G1=g1, G234=g234, G5=g5, G6=g6, G7=g7, G8=g8)


def compare(dir_a, dir_b):
    """Array-by-array comparison of two fixture directories; returns the list of differences (empty = bit-identical)."""
    bad = []
    for f in sorted(x for x in os.listdir(dir_a) if x.endswith(".npz")):
        if not os.path.exists(os.path.join(dir_b, f)):
            bad.append("%s: missing" % f)
            continue
        a, b = np.load(os.path.join(dir_a, f), allow_pickle=False), np.load(os.path.join(dir_b, f), allow_pickle=False)
        if sorted(a.files) != sorted(b.files):
            bad.append("%s: different arrays" % f)
            continue
        for k in a.files:
            if a[k].dtype.kind in "fc":
                same = a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=True)
            else:
                same = a[k].shape == b[k].shape and np.array_equal(a[k], b[k])
            if not same:
                bad.append("%s[%s]" % (f, k))
    return bad


if __name__ == "__main__":
    args = sys.argv[1:]
    if "--out" in args:
        i = args.index("--out")
        OUT = os.path.abspath(args[i + 1])
        os.makedirs(OUT, exist_ok=True)
        del args[i:i + 2]
    check = "--check" in args          # regenerate into --out (default: a scratch directory) and compare with tests/golden
    if check:
        args.remove("--check")
        if OUT == HERE:
            OUT = "/tmp/oc_goldens_regen"
            os.makedirs(OUT, exist_ok=True)
    _MODS = stage()
    print("reference modules:", ", ".join(m.__file__ for m in _MODS))
    which = args or list(ALL)
    for w in which:
        print("==", w)
        ALL[w](*_MODS)
    if check:
        bad = compare(OUT, HERE)
        print("regenerated fixtures vs committed:", "bit-identical" if not bad else "DIFFERENT: " + ", ".join(bad))
        sys.exit(1 if bad else 0)
