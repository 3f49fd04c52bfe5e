#!/usr/bin/env python3
"""bench.py -- simulation steps/sec of the optimal_crowds hot path (HJB solve + GCFM update) on MI355X.

Workload (BASELINE.json configs[3], the config the metric is quoted on): room `metro_station`,
204.75 m -> 2048 x 2048 grid (grid_step 0.1), 100 000 agents, 2 target keys, T = 20 s -> nt = 1000.

A "step" is one simulation step of `simulation.run()`: one GCFM update of every agent; the HJB solve of every
key happens once per nt steps, exactly where run() does it (before step 0), INSIDE the timed region.  With the
default --steps 1000 the timed region is one complete run: K solves + 1000 steps.  For other K the region holds
ceil(K/nt) solve rounds.  Inputs (potentials, agent state) are resident in HBM when the clock starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload metro_station|open_room|slalom|exit_opposite|room_test]
                    [--mode auto|sharded|ensemble]

N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL):
  sharded   ONE simulation on N GPUs as SURVEY 8e lays it out -- keys first (independent solves on disjoint rank groups),
            then row blocks of the HJB grid with halo exchange, agents replicated with one all-reduce per step.
            value = K / max-over-ranks time, scaling = strong.  Default (`auto`) for metro_station and open_room; the
            line also carries an `ensemble` object measured right after it, for comparison.
  ensemble  every rank runs its own member of an ensemble (own placement and noise seed, no data-path collective):
            value = N * K / max-over-ranks time = aggregate replica throughput, scaling = weak.  Default for the small
            rooms ("replicas only" where halos cost more than they save).
"""
import argparse
import copy
import hashlib
import json
import os

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")     # (kernel arguments in device memory: optimal_crowds_amd/__init__.py)
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WORKLOADS = {   # name -> (room, grid_step, T, recompute)
    "metro_station": ("metro_station", 0.1, 20.0, False),
    "slalom": ("slalom", 0.1, 20.0, False),
    "exit_opposite": ("exit_opposite", 0.1, 20.0, False),
    "room_test": ("room_test", 0.05, 20.0, False),
    "open_room": ("open_room", 0.1, 20.0, True),      # configs[4]: 4096^2, 500k agents, recompute=True
}
SHARDED_BY_DEFAULT = ("metro_station", "open_room")
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s


def csrc_hash():
    """sha256 over the kernel sources' CODE (comments and white space dropped: rewording a comment does not make a counter
    file stale): profiles/*pmc*.json carry it, so a counter file of other kernels is never quoted"""
    import re
    hsh = hashlib.sha256()
    d = os.path.join(REPO, "optimal_crowds_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            src = open(os.path.join(d, name), "r", errors="replace").read()
            src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
            src = re.sub(r"//[^\n]*", " ", src)
            hsh.update(name.encode())
            hsh.update(" ".join(src.split()).encode())
    return hsh.hexdigest()[:16]


def build_sim(workload, seed=20240807, wall_search_radius=-1.0, sharding=None, dtype="float32", field="ring"):
    import io
    import contextlib
    from optimal_crowds_amd import host, simulations
    room, gs, T, recompute = WORKLOADS[workload]
    cfg = copy.deepcopy(host.load_config(os.path.join(REPO, "optimal_crowds_amd", "config.json")))
    cfg["grid_step"] = gs
    np.random.seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        sim = simulations.simulation(room, T, recompute, config=cfg, noise="device", seed=seed, record=False, history=False,
                                     wall_search_radius=wall_search_radius, sharding=sharding, dtype=dtype, field=field)
    return sim, cfg


def solve_round(sim):
    """what run() does before the first step: one HJB solve per key (m = 0 at step 0) -- Engine.solve_keys, as the facade's
    run() calls it: keys whose whole grid lives on this GPU are solved concurrently.  Sharded: a rank solves only the keys
    (or row blocks of keys) its group holds; the others' stats are None."""
    return sim._engine.solve_keys([(k, sim.T, sim._nt0, None) for k in range(len(sim._key_names))])


def run_steps(sim, K, nt, timed_stats=None, update="synchronous", levels=None):
    """update='sequential': the reference's own one-by-one order (simulations.py:271; oc_gcfm_step_sequential) -- a fresh
    random permutation per step (drawn on the device), level-scheduled by the library; `levels` collects each step's depth"""
    import torch
    eng = sim._engine
    done = 0
    while done < K:
        s = done % nt
        if s == 0:
            st = solve_round(sim)
            if timed_stats is not None:
                timed_stats.append(st)
        elif sim.recompute and s % sim.recompute_step == 0:
            # run() with recompute=True (simulations.py:432-435): density of the current crowd, then every key again
            # on the shrunken horizon nt = round((T - t) / dt)  (quirk Q9)
            m = sim._density_dev(sim.sigma_convolution)
            st = eng.solve_keys([(k, sim.T, round((sim.T - s * sim.dt) / sim.dt), m) for k in range(len(sim._key_names))])
            if timed_stats is not None:
                timed_stats.append(st)
        if update == "sequential":
            perm = torch.randperm(sim.N, device=eng.device, dtype=torch.int32)
            out = eng.step_sequential(s, perm, noise=None, seed=sim._seed + done // nt)
            if levels is not None:
                levels.append(out[3])
        else:
            eng.step(s, noise=None, seed=sim._seed + done // nt, sync=False)
        done += 1
    # lazy_state (the engine's default, include/oc_hip.h: oc_agents_sync): the steps keep the agents' state in the library's
    # slot-ordered snapshot; the caller's arrays are brought up to date at the re-sorts and HERE, inside the timed region
    eng.flush_state()


def host_cores():
    """(threads to use, what limits them): the CPUs this process may run on, capped by the container's CPU quota"""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per) + 0.5))
    except Exception:
        pass
    return (min(aff, quota) if quota else aff), {"host_cpus": os.cpu_count(), "affinity": aff, "cgroup_quota": quota}


def cpu_baseline(sim, cfg, workload, stt, budget_s=25.0):
    """The oracle's C restatement (oracle/liboc_oracle.so: float64, OpenMP, kind = "port": the reference itself is
    Python and cannot travel to this box) timed on this host's cores, all of those this process may use:
      HJB   one COMPLETE solve of key 0 up to the dense output (every RK attempt: occ_rk45_solve, the same algorithm
            as the GPU's) + the dense output and direction field of a sample of slices, scaled to nt - 1 slices;
      GCFM  one COMPLETE synchronous step of all agents (lookup, all pairs with the cutoff test, nearest wall cell,
            integration) on the state the timed region STARTED from (after the warm-up steps: at the end of a run of a
            small room most agents have left) -- or, if a probe says it would take longer than the budget, a contiguous
            share of the agents against all, scaled.
    value = nt / (K t_solve + nt t_step): the same "one complete run" the GPU line times.  For the two small rooms the
    NumPy/SciPy-stack restatement (oracle/oc_oracle.py: the reference's own software stack, effectively one core) is
    timed beside it."""
    cores, lim = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    from oracle import oc_oracle_c as occ
    P = occ.params(cfg, sim.room_length, sim.room_height, sim.Nx, sim.Ny)
    key = sim._key_names[0]
    V = np.ascontiguousarray(sim.Vs[key])
    m = np.zeros_like(V)
    nt = sim._nt0
    # ONE complete solve (every RK attempt) that keeps three slices -- early, middle, late -- for the direction-field
    # timing; the dense output (SciPy's quartic interpolant, rk.py:560-575, one pass over y and the seven K per t_eval
    # point) and `vels` are timed DIRECTLY on arrays of the solve's size, `ns` evaluations each, and scaled to the nt - 1
    # slices the reference fills (optimals.py:200-204).  (Round 2 took the dense output as the difference of two ~3 s
    # solves: +-40 % from run to run.)
    keep = sorted(set(int(k) for k in np.linspace(1, nt - 1, 3)))
    t0 = time.perf_counter()
    phi, st = occ.rk45_solve(P, V, m, sim.T, nt, keep=keep)
    t_solve_kept = time.perf_counter() - t0
    ns = 24
    rng0 = np.random.RandomState(0)
    ysyn = rng0.uniform(0.5, 1.5, V.size)
    Ksyn = [rng0.uniform(-1.0, 1.0, V.size) for _ in range(7)]
    occ.dense_output(ysyn, Ksyn, -0.2, 0.5)                 # (first touch of the output pages, thread team start)
    t0 = time.perf_counter()
    for q in range(ns):
        occ.dense_output(ysyn, Ksyn, -0.2, (q + 0.5) / ns)
    t_dense = (time.perf_counter() - t0) / ns
    del ysyn, Ksyn
    occ.vels(P, phi[0])
    t0 = time.perf_counter()
    for q in range(ns):
        occ.vels(P, phi[q % len(keep)])
    t_vels = (time.perf_counter() - t0) / ns
    t_attempts = max(t_solve_kept - len(keep) * t_dense, 0.0)
    t_attempts_and_sample = t_solve_kept
    t_hjb = t_attempts + (nt - 1) * (t_dense + t_vels)
    # ---- GCFM: complete step(s) on the state the timed region started from
    N = sim.N
    wms = [occ.WallMap(sim._X0, sim._Y0, sim.Vs[k]) for k in sim._key_names]
    rects = [[sim._room["targets"][t] for t in sim._possible[k]] for k in sim._key_names]
    rng = np.random.RandomState(1)
    noise = rng.normal(size=(N, 2))

    def part(i0, i1):
        t = time.perf_counter()
        occ.step_sync(P, stt["x"], stt["y"], stt["vx"], stt["vy"], sim._v_des, stt["status"], sim._key_of,
                      [None] * len(wms), sim._X0, sim._Y0, wms, rects, noise, i0=i0, i1=i1)
        return time.perf_counter() - t
    n_probe = min(N, 512)
    t_probe = part(0, n_probe)
    est = t_probe / n_probe * N
    gcfm_budget = max(budget_s - t_attempts_and_sample - ns * (t_dense + t_vels), 5.0)
    if est <= gcfm_budget:
        t_step, share = part(0, N), N
    else:
        share = max(n_probe, int(N * gcfm_budget / est))
        t_step = part(0, share) / share * N
    K = len(sim._key_names)
    value = nt / (K * t_hjb + nt * t_step)
    out = {"value": value, "unit": "steps/s", "cores": cores, "kind": "port",
           "sample": "C/OpenMP float64 oracle on %d threads (%s): HJB = one complete RK45 solve of one key, %d attempts "
                     "(%.2f s) + dense output and direction field timed directly, %d evaluations each, scaled to %d slices (%.4f + %.4f s each); "
                     "GCFM = one synchronous step of %d of %d agents against all (%.2f s%s); value = nt / (K t_solve + "
                     "nt t_step), K = %d, nt = %d" % (cores, json.dumps(lim), st["n_attempt"], t_attempts, ns,
                                                      nt - 1, t_dense, t_vels, share, N, t_step,
                                                      "" if share == N else ", scaled", K, nt),
           "t_hjb_solve_s": t_hjb, "t_step_s": t_step, "rk_attempts": st["n_attempt"], **lim}
    if workload in ("room_test", "exit_opposite") and os.environ.get("OC_BENCH_NUMPY_STACK", "1") != "0":
        out["numpy_stack"] = numpy_stack_baseline(sim, cfg, stt)
    return out


def numpy_stack_baseline(sim, cfg, stt, steps=2):
    """the NumPy/SciPy-stack restatement of the reference (oracle/oc_oracle.py, float64) on this host:
    one complete HJB solve of one key (its own RK45 in NumPy, every slice through vels) and `steps` sequential steps
    in the reference's update order."""
    from oracle import oc_oracle as oc
    room = sim._room
    np.random.seed(sim._seed)
    o = oc.Simulation(room, sim.T, cfg=cfg, mode="sequential")
    key = o.key_names[0]
    t0 = time.perf_counter()
    o.targets[key].compute_optimal_velocity(0.0, np.zeros_like(o.V))
    t_solve = time.perf_counter() - t0
    o.x, o.y, o.vx, o.vy = (stt[n].copy() for n in ("x", "y", "vx", "vy"))
    o.status = stt["status"].astype(bool).copy()
    o.inside = int(o.status.sum())
    for k in o.key_names[1:]:                       # (the lookup needs a field for every key: reuse the solved one)
        o.targets[k].vx_opt, o.targets[k].vy_opt = o.targets[key].vx_opt, o.targets[key].vy_opt
    t0 = time.perf_counter()
    for _ in range(steps):
        o.step()
    t_step = (time.perf_counter() - t0) / steps
    K, nt = len(o.key_names), sim._nt0
    return {"value": nt / (K * t_solve + nt * t_step), "unit": "steps/s", "cores": 1, "kind": "port (NumPy/SciPy stack)",
            "t_hjb_solve_s": t_solve, "t_step_s": t_step,
            "sample": "oracle/oc_oracle.py: one complete solve of one key + %d sequential steps of %d agents" % (steps, sim.N)}


def gcfm_work(sim, cfg, t_forces_s, t_step_s):
    """SURVEY 8d's algorithmic work of one GCFM step on the state the run ended with (untimed): ordered pairs
    P = N_a (N_a - 1), of which P_near lie within the cutoff (counted here with torch on the GPU, float64), nominal
    8 flop per far pair and 60 per near pair; 88 B of state traffic per active agent.  The far pairs are never
    visited (tile boxes and neighbour lists cull them): only the near-pair figure is an executed rate."""
    a = sim._engine.agents
    act = a["status"].bool()
    x, y = a["x"][act], a["y"][act]
    na = int(x.numel())
    near = 0
    for i0 in range(0, na, 2048):
        dx = x[i0:i0 + 2048, None] - x[None, :]
        dy = y[i0:i0 + 2048, None] - y[None, :]
        near += int(((dx * dx + dy * dy).sqrt() < float(cfg["repulsion_cutoff"])).sum().item())
    near -= na                                              # the agent itself
    P = na * (na - 1)
    return {"active_agents": na, "ordered_pairs": P, "near_pairs": near,
            "flop_near_pairs": 60.0 * near,
            "near_pair_TFLOPs": 60.0 * near / t_forces_s / 1e12,
            "near_pair_frac_of_fp32_vector_peak": 60.0 * near / t_forces_s / 157.3e12,
            "all_pairs_equivalent_flop": 8.0 * (P - near) + 60.0 * near,
            "state_bytes_frac_of_hbm_peak": 88.0 * na / t_step_s / (HBM_PEAK_GBS * 1e9)}


def pmc_file(pattern, workload="metro_station"):
    """newest committed PMC summary of this workload whose kernels are the ones in this tree (stamped with csrc_hash),
    else None"""
    import glob
    for f in reversed(sorted(glob.glob(os.path.join(REPO, "profiles", pattern)))):
        try:
            d = json.load(open(f))
            if d.get("csrc_sha") == csrc_hash() and d.get("workload", "metro_station") == workload:
                return d
        except Exception:
            pass
    return None


def pmc_traffic(workload, kernel_class="hjb_out", per_attempt=False):
    """HBM bytes per launch of a kernel class from the committed PMC summary (profiles/rNN_pmc_traffic*.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very command, gfx950 FETCH_SIZE x2 correction applied by
    tools/collect_profiles.py).  Counters cannot be collected from inside the timed run, hence the file; null if the
    file is absent, belongs to another workload or was taken with other kernel sources."""
    d = pmc_file("r*_pmc_traffic*.json", workload)
    if not d:
        return None
    key = {"hjb_out": "k_out_hbm_bytes_per_launch", "hjb_stage": "k_stageV_hbm_bytes_per_launch",
           "gcfm_forces": "k_forces_hbm_bytes_per_launch", "density": "k_density_hbm_bytes_per_launch"}.get(kernel_class)
    if per_attempt and kernel_class == "hjb_stage":
        # active set: all HBM bytes of the solves' stage launches over the launches that did any work -- the same set of
        # launches `achieved` averages over (the per-launch key is the mean of the few launches that look at every tile)
        return d.get("k_stageV_hbm_bytes_per_attempt")
    return d.get(key)


def pmc_step(workload):
    d = pmc_file("r*_pmc_step*.json", workload)
    return d.get("k_forces") if d else None


def measure(sim, args, nt, world, dist, red_dev, keep_state=False, levels=None):
    """W warmup steps, then exactly K timed steps bracketed by barrier + synchronize; max over ranks."""
    import torch
    from optimal_crowds_amd import parallel
    eng = sim._engine
    K, W = args.steps, args.warmup

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    upd = getattr(args, "update", "synchronous")
    if W > 0:
        run_steps(sim, W, nt, update=upd)
    barrier()
    # HJB kernels on every 8th attempt; GCFM kernels on every 16th step -- every 4th in a short region (the driver's --steps 20:
    # 5 samples instead of 1): event records cost gaps, so a long run samples sparsely
    eng.set_profiling(1 | (8 if K <= 64 else 4))
    eng.reset_kernel_stats()
    stats = []
    state0 = eng.get_state() if keep_state else None        # the crowd the timed region starts from (CPU baseline's input)
    barrier()
    t0 = time.perf_counter()
    run_steps(sim, K, nt, stats, update=upd, levels=levels)
    barrier()
    elapsed = time.perf_counter() - t0
    return parallel.max_over_ranks(elapsed, device=red_dev), stats, state0


def launch_ranks(n):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port <free> bench.py
    <the same arguments>` as a child process; its stdout (rank 0's JSON line) is relayed.  Returns its exit code."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # (dmabuf IPC: RCCL across processes needs it on this host driver)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in p.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return p.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="metro_station", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-work-count", action="store_true",
                    help="skip breakdown.gcfm_work (its torch kernels clutter a rocprofv3 kernel-stats run)")
    ap.add_argument("--wall-radius", type=float, default=-1.0,
                    help="diagnostic: wall search radius in m (default -1: per-agent radius from the 1e-12 m/s force bound)")
    ap.add_argument("--mode", default="auto", choices=["auto", "ensemble", "sharded"],
                    help="N > 1: sharded = ONE simulation on N GPUs (keys first, then HJB row blocks; agents replicated, one "
                         "all-reduce per step; strong scaling) -- auto picks it for metro_station / open_room; ensemble = "
                         "one independent simulation per GPU (aggregate replica throughput, weak scaling)")
    ap.add_argument("--no-ensemble-line", action="store_true", help="sharded mode: do not also measure the ensemble")
    ap.add_argument("--no-rows-variant", action="store_true",
                    help="sharded mode with at least two ranks per key: do not also measure the keys x row-blocks variant")
    ap.add_argument("--no-dense-line", action="store_true",
                    help="do not also run the HJB solves with every tile computed (profiling passes: keeps the kernel "
                         "averages to the product path)")
    ap.add_argument("--tune", default="", help="launch-shape knobs, e.g. pair_split=4 (oc_set_tuning)")
    ap.add_argument("--field", default="ring", choices=["ring", "slices"],
                    help="what an HJB solve leaves in HBM: ring = the accepted steps' interpolants, lookups evaluate the "
                         "direction from them (default, the product path); slices = every slice of vx_opt / vy_opt is written "
                         "like the reference does (optimals.py:200-204)")
    ap.add_argument("--update", default="synchronous", choices=["synchronous", "sequential"],
                    help="agent update order: synchronous (frozen snapshot, the fast path and the default) or sequential = the "
                         "reference's one-by-one order in a fresh random permutation per step (simulations.py:271), "
                         "level-scheduled on the device: what trajectory-level parity with the reference costs (single GPU)")
    ap.add_argument("--hjb-dtype", default="float32", choices=["float32", "mixed", "float64"],
                    help="HJB state: float32 (BASELINE's configs, default), mixed (float64 state, float32 field), float64")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
            raise SystemExit("--gpus %d under a launcher of %d ranks" % (args.gpus, world))
        # `python bench.py --gpus N` without a launcher: start the ranks ourselves -- a CHILD process (this one has not
        # touched the GPU: nothing is re-executed from a process that has) running torch.distributed.run with one rank per
        # GPU, whose single JSON line and exit code are passed on
        sys.exit(launch_ranks(args.gpus))
    import torch
    from optimal_crowds_amd import parallel
    # OC_BENCH_ONE_DEVICE=1 (rehearsal on a 1-GPU box): every rank uses cuda:0 and the collectives run over gloo
    one_dev = os.environ.get("OC_BENCH_ONE_DEVICE") == "1"
    torch.cuda.set_device(0 if one_dev else local_rank)
    backend = "gloo" if one_dev else "nccl"
    # OC_BENCH_FORCE_SHARDED=1 (rehearsal of the RCCL code path on a 1-GPU box): a process group of ONE rank over nccl and
    # the sharded mode on it -- every collective of the design (the solve's all-gather, the step's all-reduce) is then
    # issued through RCCL from the library's callbacks, on the launch stream, exactly as with N > 1
    force_sharded = os.environ.get("OC_BENCH_FORCE_SHARDED") == "1"
    if force_sharded and world == 1:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        tdist.init_process_group(backend, rank=0, world_size=1,
                                 **({"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}))
    rank, world = parallel.init_distributed(backend)
    dist = None
    if world > 1 or force_sharded:
        import torch.distributed as dist
    red_dev = "cpu" if one_dev else "cuda"

    import optimal_crowds_amd.engine  # noqa: F401  (fails loudly without the HIP library)
    mode = args.mode
    if mode == "auto":
        mode = "sharded" if args.workload in SHARDED_BY_DEFAULT else "ensemble"
    sharded = mode == "sharded" and (world > 1 or force_sharded)
    t_init = time.perf_counter()
    if sharded:     # one simulation on all ranks: same seed everywhere (replicated agents)
        sim, cfg = build_sim(args.workload, wall_search_radius=args.wall_radius, sharding=parallel.dist_sharding(),
                             dtype=args.hjb_dtype, field=args.field)
    else:           # N = 1, or an ensemble: every rank simulates its own member (own placement and noise seed)
        sim, cfg = build_sim(args.workload, seed=20240807 + rank, wall_search_radius=args.wall_radius, dtype=args.hjb_dtype,
                             field=args.field)
    t_init = time.perf_counter() - t_init
    eng = sim._engine
    for kv in filter(None, args.tune.split(",")):
        name, val = kv.split("=")
        eng.set_tuning(name, int(val))
    nt = sim._nt0
    K, W = args.steps, args.warmup
    # the exact active set of the HJB solve (csrc/hjb.hip): on by default from 96 stage tiles (112 x 20 cells) on, single
    # GPU / whole-key solves only; `--tune hjb_active=0` computes every tile of every attempt
    tune_active = [kv.split("=")[1] for kv in args.tune.split(",") if kv.startswith("hjb_active=")]
    active_knob = int(tune_active[-1]) if tune_active else int(os.environ.get("OC_HJB_ACTIVE", "1"))
    n_stage_tiles = -(-sim.Nx // 112) * -(-sim.Ny // 20)
    rows_split = sharded and any(len(p) > 1 for p in getattr(eng, "key_plan", []))
    active_set = (active_knob == 2 or (active_knob == 1 and n_stage_tiles >= 96)) and n_stage_tiles <= 12000 and not rows_split
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    if args.update == "sequential" and (world > 1 or sharded):
        raise SystemExit("--update sequential is a single-GPU mode")
    seq_levels = [] if args.update == "sequential" else None
    elapsed, stats, state0 = measure(sim, args, nt, world, dist, red_dev, keep_state=want_cpu, levels=seq_levels)

    # ---- breakdown (untimed extras for the report)
    mine = [s for rnd in stats for s in rnd if s is not None]
    t_hjb = sum(s["seconds"] for s in mine)
    first = mine[0] if mine else None
    ring = any(f is not None and f.is_ring for f in eng.fields)
    kstats = {}
    for name in ("hjb_out", "hjb_stage", "hjb_list", "hjb_persist", "gcfm_forces", "gcfm_finish", "density"):
        nl, ms, by = eng.kernel_stats(name)
        kstats[name] = {"launches_timed": nl, "avg_ms": (ms / nl if nl else None),
                        "designed_GB_per_launch": (by / nl / 1e9 if nl else None),
                        "achieved_GBps": (by / (ms * 1e-3) / 1e9 if ms > 0 and by > 0 else None)}
    # ---- the dominant kernel of the timed region: launches x average event time per class (the HJB classes are sampled
    # every 8th attempt, the step's every 16th step; the launch counts are exact)
    n_att = sum(s_["n_attempt"] for s_ in mine)
    n_acc = sum(s_["n_accept"] for s_ in mine)
    # (the list builder is a launch of its own only where a ring solve's attempt is NOT fused -- lists beyond the flat ticket's
    # range before the solve goes dense: none at the BASELINE rooms; with the fused tail it runs inside the stage launch)
    launches = {"gcfm_forces": K, "gcfm_finish": K, "hjb_stage": n_att,
                "hjb_out": 0 if ring else n_acc, "density": max(len(stats) - max(1, -(-K // nt)), 0) if sim.recompute else 0}
    kstats.pop("hjb_list", None)
    kp = kstats.pop("hjb_persist")
    n_persist = sum(s_.get("persist_attempts", 0) for s_ in mine)
    for name, n_l in launches.items():
        kstats[name]["launches_in_timed_region"] = n_l
        kstats[name]["est_total_ms"] = (kstats[name]["avg_ms"] or 0.0) * n_l
    # (the controller / list-builder launches and k_finish are latency-sized helpers: reported, never "the" kernel)
    dominant = max((n_ for n_ in launches if n_ != "gcfm_finish"), key=lambda n_: kstats[n_]["est_total_ms"])
    kf = kstats["gcfm_forces"]
    gw = None
    if rank == 0 and kf.get("avg_ms") and not args.no_work_count and not sharded:
        kn_ms = kstats["gcfm_finish"]["avg_ms"] or 0.0
        gw = gcfm_work(sim, cfg, kf["avg_ms"] * 1e-3, (kf["avg_ms"] + kn_ms) * 1e-3)

    def hbm_roof(name, kernel, note=None):
        k_ = kstats[name]
        if not k_["launches_timed"] or not k_["achieved_GBps"]:
            return None
        # (the PMC passes are taken with the default float32 state: no figure for another HJB dtype; and on a --steps 20 run:
        # where that run's stage launches are of another kind than the timed ones -- open_room's first, sparse solve against
        # the dense re-solves of a complete run -- the sample says nothing about them)
        traffic = pmc_traffic(args.workload, name, per_attempt=bool(active_set)) if sim.hjb_dtype == "float32" else None
        if traffic is not None and k_["designed_GB_per_launch"] and traffic < 0.25e9 * k_["designed_GB_per_launch"]:
            traffic = None
        r = {"bound": "hbm", "kernel": kernel, "achieved": k_["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": k_["achieved_GBps"] / HBM_PEAK_GBS,
             "traffic": traffic,
             "traffic_unit": "HBM bytes per launch (rocprofv3 PMC passes, "
             "profiles/; null unless taken with these kernel sources)",
             "launches_timed": k_["launches_timed"], "avg_launch_ms": k_["avg_ms"],
             "algorithmic_GB_per_launch": k_["designed_GB_per_launch"],
             "share_of_timed_region_gpu_ms": k_["est_total_ms"]}
        if note:
            r["note"] = note
        return r
    stage_note = None
    if active_set:
        stage_note = ("exact active set: only the tiles where phi is not constant are loaded and computed (bytes = the designed "
                      "passes x the solve's mean share of such tiles); with a few dozen listed tiles the launch is one tile's "
                      "trip through its block -- latency, not bandwidth; the same launch with every tile computed is in "
                      "`dense_sweep.k_stageV`.  `traffic` = all HBM bytes of the solves' stage launches / the launches that did "
                      "any work: ~80 % of them belong to the ~7 launches per solve that look at EVERY tile (the first "
                      "attempt, which has no list yet, and the attempts around a change of scale, which rewrite the far "
                      "field) and to the listed tiles that are read only to be found trivial -- bytes the algorithmic figure "
                      "(computed tiles x designed passes) does not count")
    roofs = {
        "hjb_out": lambda: hbm_roof("hjb_out", "k_out (HJB dense output + direction-field writer)",
                                    ("exact active set: the far field's slices are written as zeros without reads -- (almost) "
                                     "a pure write stream of the whole direction field") if active_set else None),
        "hjb_stage": lambda: hbm_roof("hjb_stage", "k_stageV (all six RK stages of an attempt in one launch%s)"
                                      % ("; stores the step's interpolant into the ring" if ring else ""), stage_note),
        "density": lambda: hbm_roof("density", "k_density (separable Gaussian splat)"),
    }
    roof = None
    if dominant in roofs:
        roof = roofs[dominant]()
    elif dominant in ("gcfm_forces", "gcfm_finish") and kf.get("avg_ms"):
        # The step's force kernel is bound by vector-instruction issue and latency -- neither by HBM nor by MFMA (the pair
        # "contraction" has inner dimension 2; 88 B of state per agent and step).  SURVEY 8d prescribes the fp32-vector
        # fraction as its binding figure, with the HBM fraction beside it: achieved = 60 flop x near pairs / launch time.
        roof = {"bound": "valu", "bound_note": "fp32 vector ALU (SURVEY 8d: the GCFM step is VALU/transcendental-bound, not "
                "HBM, not MFMA); the contract's hbm / mfma choices do not apply to this kernel",
                "kernel": "k_forces (pair forces + nearest-wall search + floor-field lookup)",
                "achieved": gw["near_pair_TFLOPs"] if gw else None, "peak": 157.3, "unit": "TFLOP/s",
                "frac": gw["near_pair_frac_of_fp32_vector_peak"] if gw else None,
                "flop_model": "60 flop per ordered pair within the cutoff (SURVEY 8d), pairs counted on the final state",
                "traffic": pmc_traffic(args.workload, "gcfm_forces"),
                "traffic_unit": "HBM bytes per launch (rocprofv3 PMC passes, profiles/; null unless taken with these sources)",
                "algorithmic_bytes_per_launch": 88.0 * (gw["active_agents"] if gw else sim.N),
                "hbm_frac": (88.0 * gw["active_agents"] / (kf["avg_ms"] * 1e-3) / (HBM_PEAK_GBS * 1e9)) if gw else None,
                "launches_timed": kf["launches_timed"], "avg_launch_ms": kf["avg_ms"],
                "share_of_timed_region_gpu_ms": kf["est_total_ms"]}
    # the HBM-bound kernel that matters most, when the dominant kernel is not HBM-bound
    roof_hbm = None
    if roof is None or roof.get("bound") != "hbm":
        cand = [n_ for n_ in ("hjb_stage", "hjb_out", "density") if kstats[n_]["est_total_ms"] > 0]
        if cand:
            roof_hbm = roofs[max(cand, key=lambda n_: kstats[n_]["est_total_ms"])]()
    keys_n = len(sim._key_names)
    parallelism = "single GPU"
    if world > 1:
        parallelism = ("one simulation on %d GPUs: keys first, then HJB row blocks; replicated agents" % world if sharded
                       else "ensemble of %d replicas, one per GPU" % world)
    kn = kstats["gcfm_finish"]
    kernel_step_ms = (kf["avg_ms"] + kn["avg_ms"]) if kf["avg_ms"] and kn["avg_ms"] else None
    # the complete run (one solve round of all keys + nt steps) from this region's own parts -- the wall time of
    # its solve rounds and the step kernels' event times (the steps of a short region drain behind the host: their wall time
    # is not separable) -- so that a --steps 20 record carries the figure a --steps 1000 run measures directly
    projected = None
    if kernel_step_ms and stats and not sim.recompute:
        rounds = [max(s_["seconds"] for s_ in rnd if s_ is not None) for rnd in stats if any(s_ is not None for s_ in rnd)]
        if rounds:
            t_round = float(np.mean(rounds))
            projected = {"steps_per_s": nt / (t_round + nt * kernel_step_ms * 1e-3), "t_solve_round_s": t_round,
                         "t_step_kernels_s": kernel_step_ms * 1e-3, "nt": nt,
                         "formula": "nt / (t_solve_round + nt * t_step_kernels); t_solve_round = the longest key's library time per "
                                    "round (keys are solved concurrently), t_step_kernels = k_forces + k_finish event times; the "
                                    "re-sort every 32 steps (~0.1 ms) is not included"}
    result = {
        "metric": "sim steps/sec (HJB sweep + GCFM update), %s %d^2 grid x %d agents" % (args.workload, sim.Nx, sim.N),
        "value": (1 if sharded or world == 1 else world) * K / elapsed, "unit": "steps/s", "n_gpus": world, "steps": K,
        "warmup": W, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True,
        "scaling": "strong" if sharded else "weak", "vs_baseline": None,
        "dtype": "f32" if sim.hjb_dtype == "float32" else "f64", "data": "synthetic",
        "config": {"workload": "%s %dx%d grid, %d agents, %d keys, T=%g (nt=%d)"
                               % (args.workload, sim.Nx, sim.Ny, sim.N, keys_n, sim.T, nt)
                   + (", recompute=True" if sim.recompute else ""),
                   "parallelism": parallelism, "mode": "sharded" if sharded else ("ensemble" if world > 1 else "single"),
                   "noise": "device Philox", "agent_state": "f64", "pair_math": "f32",
                   "agent_arrays": ("lazy_state: the caller's x / y / vx / vy / agent_time are scattered at the re-sorts and at "
                                    "the end of the timed region (oc_agents_sync), not in every step") if eng.lazy_state
                   else "scattered in every step",
                   "update": ("sequential: the reference's one-by-one order in a fresh permutation per step, level-scheduled "
                              "(mean %.1f levels per step, max %d)" % (float(np.mean(seq_levels)), max(seq_levels))
                              if seq_levels else "synchronous (frozen snapshot)"),
                   "hjb_state": {"float32": "f32", "mixed": "f64", "float64": "f64"}[sim.hjb_dtype],
                   "hjb_field_type": "f64" if sim.hjb_dtype == "float64" else "f32",
                   "hjb_field": ("ring: the accepted steps' interpolants stay in HBM, lookups evaluate the direction from them "
                                 "(bit-identical to the slices)") if ring else "slices: every slice of vx_opt / vy_opt is written",
                   "hjb_active_set": ("exact: tiles where phi is constant and K = 0 are not computed (bit-identical to the "
                                      "dense sweep)") if active_set
                   else "off: every tile of every attempt is computed"},
        "roofline": roof,
        "roofline_hbm": roof_hbm,
        "dominant_kernel_class": dominant,
        "breakdown": {"hjb_solve_s_total_this_rank": t_hjb, "hjb_solves_this_rank": len(mine),
                      "gcfm_kernel_ms_per_step": kernel_step_ms,
                      "gcfm_kernel_ms_note": "k_forces + k_finish HIP-event times, sampled every 16th step (the steps are "
                                             "enqueued and drain behind later host work, so wall-clock differences do not "
                                             "isolate them)",
                      "hjb_persistent_launch": {"attempts_inside": n_persist, "blocks": max((s_.get("persist_blocks", 0) for s_ in mine), default=0),
                                                "exits": sorted(set(s_.get("persist_exit", 0) for s_ in mine)),
                                                "avg_ms_per_attempt": kp["avg_ms"],
                                                "note": "attempts that ran inside ONE co-resident launch (oc_set_tuning hjb_persist; "
                                                        "default: grids of a few dozen tiles only -- csrc/hjb.hip)"},
                      "hjb_launch_resyncs": sum(s_.get("grid_resyncs", 0) for s_ in mine),
                      "hjb_launch_resyncs_note": "attempts whose launch did nothing and were drained and re-enqueued: a grid smaller than "
                                                 "its work list, or a launch that found a controller block not yet written (seen only "
                                                 "where two processes share one GPU)",
                      "projected_complete_run": projected,
                      "nfev": first["nfev"] if first else None, "rk_attempts": first["n_attempt"] if first else None,
                      "rk_accepted": first["n_accept"] if first else None, "kernels": kstats, "init_s": t_init,
                      "active_agents_end": int(eng.counters.cpu()[2]), "csrc_sha": csrc_hash()},
    }
    if world > 1 or sharded:
        sc = getattr(eng, "_step_comm", None)
        comms = [c for c in getattr(eng, "_halo_comms", {}).values() if c is not None] if sharded else []
        rowc = [c for c in comms if c.shard.world > 1]
        result["rccl"] = {
            "backend": backend, "ranks": world, "mode": "sharded" if sharded else "ensemble",
            "issued_by": ("the library itself over its own RCCL communicators (oc_comm_create: no host callback; the "
                          "callback counters below stay 0)") if getattr(eng, "rccl_native", False)
            else "callbacks into torch.distributed, enqueued on the launch stream",
            "hjb_key_plan": [list(p) for p in getattr(eng, "key_plan", [])] if sharded else None,
            "step_allgathers": sc.n_allgather if (sharded and sc) else 0,
            "step_allgather_bytes": sc.bytes_allgather if (sharded and sc) else 0,
            "step_allreduces": sc.n_allreduce if (sharded and sc) else 0,
            "step_allreduce_bytes": sc.bytes_allreduce if (sharded and sc) else 0,
            "step_exchange": ("one all-gather of the ranks' tile shares (pair sums + wall forces, 32 B per agent slot); the "
                              "desired velocities by a sum of their own (16 B per slot) where the floor fields are sharded, "
                              "inside the all-gather (48 B per slot) where every rank holds them") if sharded else None,
            "collectives_per_step": ((1 + (1 if sc and sc.n_allreduce else 0)) if sharded else 0),
            "hjb_allgathers_this_rank": sum(c.n_allgather for c in rowc),
            "hjb_halo_exchanges_this_rank": sum(c.n_exchange for c in rowc),
            "hjb_halo_message_bytes": (rowc[0].halo_bytes if rowc and hasattr(rowc[0], "halo_bytes") else 0),
            "hjb_collectives_per_attempt": ({"allgather_16B_per_rank": 1, "neighbour_halo_exchange": 1} if rowc else {}),
            "counts_cover": "warm-up + timed region",
        }
    nv = pmc_step(args.workload)
    if kf.get("avg_ms") and nv and nv.get("SQ_INSTS_VALU"):
        # the step's dominant kernel is bound by vector-instruction issue, not by HBM or MFMA: a plain (unpacked) VALU
        # instruction of a wavefront occupies its SIMD for 4 cycles -> peak = 1024 SIMDs x 2.4 GHz / 4
        peak = 256 * 4 * 2.4e9 / 4
        ach = nv["SQ_INSTS_VALU"] / (kf["avg_ms"] * 1e-3)
        result["roofline_valu"] = {"bound": "valu-issue", "kernel": "k_forces (pair forces + wall search + lookup)",
                                   "achieved": ach / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s",
                                   "frac": ach / peak, "valu_instructions_per_launch": nv["SQ_INSTS_VALU"],
                                   "avg_launch_ms": kf["avg_ms"], "source": "rocprofv3 --pmc SQ_INSTS_VALU (profiles/)"}
        wc, bc, nw = (nv.get(c) for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES"))
        if wc and bc and nw:
            # SQ_WAVE_CYCLES counts quad-cycles summed over the waves; SQ_BUSY_CYCLES is summed over the chip's 32
            # shader engines (8 XCDs x 4), so busy/32 is the launch's length in shader cycles
            result["roofline_valu"]["occupancy"] = {
                "waves_per_simd_hardware": 8, "mean_resident_waves_per_simd": 4.0 * wc / (bc / 32.0) / 1024.0,
                "waves_per_launch": nw, "source": "SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAVES (profiles/)"}
    if gw is not None:
        result["breakdown"]["gcfm_work"] = gw
    if rank == 0 and world == 1 and active_set and not args.no_dense_line:
        # the same solve with every tile computed (one untimed round, right after): the dense sweep's kernels beside it
        eng.set_tuning("hjb_active", 0)
        eng.reset_kernel_stats()
        torch.cuda.synchronize()
        ov, eng.overlap_solves = eng.overlap_solves, False      # (one key after the other: clean per-launch times)
        t0 = time.perf_counter()
        dstat = [s_ for s_ in solve_round(sim) if s_ is not None]
        torch.cuda.synchronize()
        t_dense = time.perf_counter() - t0
        eng.overlap_solves = ov
        dk = {}
        for name in ("hjb_out", "hjb_stage"):
            nl, ms, by = eng.kernel_stats(name)
            dk[name] = {"launches_timed": nl, "avg_ms": (ms / nl if nl else None),
                        "designed_GB_per_launch": (by / nl / 1e9 if nl else None),
                        "achieved_GBps": (by / (ms * 1e-3) / 1e9 if ms > 0 and by > 0 else None)}
        eng.set_tuning("hjb_active", active_knob)
        result["dense_sweep"] = {
            "what": "the same HJB solves with hjb_active=0 (every tile of every attempt computed), one untimed round, key after key",
            "hjb_solve_s_per_key": t_dense / max(len(dstat), 1), "rk_attempts": dstat[0]["n_attempt"] if dstat else None,
            "k_out": dict(dk["hjb_out"], frac_of_hbm_peak=(dk["hjb_out"]["achieved_GBps"] or 0) / HBM_PEAK_GBS),
            "k_stageV": dict(dk["hjb_stage"], frac_of_hbm_peak=(dk["hjb_stage"]["achieved_GBps"] or 0) / HBM_PEAK_GBS)}
    if want_cpu:
        result["cpu_baseline"] = cpu_baseline(sim, cfg, args.workload, state0,
                                              budget_s=float(os.environ.get("OC_BENCH_CPU_BUDGET", "25")))
    if sharded and world >= 2 * keys_n and keys_n >= 2 and not args.no_rows_variant:
        # BASELINE configs[3] as SURVEY 8e writes it -- "2 targets on 4 GPUs = 2 keys x 2-way row split": the keys-first plan
        # above keeps a 2048 x 2048 key whole (a sparse solve is a chain of short launches: splitting it adds two collectives per
        # attempt and wins nothing), so the row-split variant is measured beside it, on the same ranks, for the record
        eng.close()
        del sim, eng
        torch.cuda.empty_cache()
        simr, _ = build_sim(args.workload, wall_search_radius=args.wall_radius, sharding=parallel.dist_sharding(min_cells=1),
                            dtype=args.hjb_dtype, field=args.field)
        elr, str_, _ = measure(simr, args, nt, world, dist, red_dev)
        engr = simr._engine
        rowc = [c for c in getattr(engr, "_halo_comms", {}).values() if c is not None and c.shard.world > 1]
        miner = [s_ for rnd in str_ for s_ in rnd if s_ is not None]
        result["rows_variant"] = {
            "what": "the same run with every key's grid split into row blocks over its rank group (OC_ROW_MIN_CELLS = 1): one "
                    "16-byte all-gather and one neighbour halo exchange per RK attempt, the exact active set per rank",
            "value": K / elr, "unit": "steps/s", "ms_per_step": 1e3 * elr / K,
            "hjb_key_plan": [list(p) for p in getattr(engr, "key_plan", [])],
            "hjb_solve_s_total_this_rank": sum(s_["seconds"] for s_ in miner),
            "hjb_halo_message_bytes": (rowc[0].halo_bytes if rowc and hasattr(rowc[0], "halo_bytes") else 0),
            "hjb_halo_exchanges_this_rank": sum(c.n_exchange for c in rowc),
            "hjb_allgathers_this_rank": sum(c.n_allgather for c in rowc),
            "hjb_collectives_per_attempt": {"allgather_16B_per_rank": 1, "neighbour_halo_exchange": 1} if rowc else {}}
        sim, eng = simr, engr
    if sharded and not args.no_ensemble_line:
        # the other way to use N GPUs, measured right after: one independent replica per GPU (no data-path collective)
        eng.close()
        del sim, eng
        torch.cuda.empty_cache()
        sim2, _ = build_sim(args.workload, seed=20240807 + rank, wall_search_radius=args.wall_radius, field=args.field)
        el2, _, _ = measure(sim2, args, nt, world, dist, red_dev)
        result["ensemble"] = {"value": world * K / el2, "unit": "steps/s (aggregate over %d independent replicas)" % world,
                              "scaling": "weak", "ms_per_step_per_replica": 1e3 * el2 / K}
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
