#!/usr/bin/env python3
"""bench.py -- simulation steps/sec of the optimal_crowds hot path (HJB solve + GCFM update) on MI355X.

Workload (BASELINE.json configs[3], the config the metric is quoted on): room `metro_station`,
204.75 m -> 2048 x 2048 grid (grid_step 0.1), 100 000 agents, 2 target keys, T = 20 s -> nt = 1000.

A "step" is one simulation step of `simulation.run()`: one GCFM update of every agent; the HJB solve of every
key happens once per nt steps, exactly where run() does it (before step 0), INSIDE the timed region.  With the
default --steps 1000 the timed region is one complete run: K solves + 1000 steps.  For other K the region holds
ceil(K/nt) solve rounds.  Inputs (potentials, agent state) are resident in HBM when the clock starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload metro_station|slalom|exit_opposite|room_test]

For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank runs its own member of an ensemble of the
workload (own initial placement and noise seed; "replicas only"): value = N * K /
max-over-ranks time, scaling = weak.
"""
import argparse
import copy
import json
import os
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
    "open_room": ("open_room", 0.1, 20.0, True),      # configs[4] on ONE GPU: 4096^2, 500k agents, recompute=True
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s


def build_sim(workload, seed=20240807, wall_search_radius=-1.0, sharding=None):
    import io
    import contextlib
    from optimal_crowds_amd import host, simulations
    room, gs, T, recompute = WORKLOADS[workload]
    cfg = copy.deepcopy(host.load_config(os.path.join(REPO, "optimal_crowds_amd", "config.json")))
    cfg["grid_step"] = gs
    np.random.seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        sim = simulations.simulation(room, T, recompute, config=cfg, noise="device", seed=seed, record=False, history=False,
                                     wall_search_radius=wall_search_radius, sharding=sharding)
    return sim, cfg


def solve_round(sim):
    """what run() does before the first step: one HJB solve per key (m = 0 at step 0)."""
    stats = []
    for k, key in enumerate(sim._key_names):
        st, _ = sim._engine.solve(k, sim.T, sim._nt0, m=None)
        stats.append(st)
    return stats


def run_steps(sim, K, nt, timed_stats=None):
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
            st = [eng.solve(k, sim.T, round((sim.T - s * sim.dt) / sim.dt), m=m)[0] for k in range(len(sim._key_names))]
            if timed_stats is not None:
                timed_stats.append(st)
        eng.step(s, noise=None, seed=sim._seed + done // nt, sync=False)
        done += 1


def cpu_baseline(sim, cfg, n_attempt, budget_s=20.0):
    """The oracle's C restatement (oracle/liboc_oracle.so: float64, OpenMP) timed on this host on a bounded sample
    of the same workload, then scaled by the GPU run's own work counts:
      HJB   : full-grid RHS / stage-combination / vels passes -> one RK attempt = 6 RHS + 7 combinations,
              one output slice = 1 combination + 1 vels; x (attempts, nt-1 slices) of the GPU solve, per key
      GCFM  : complete synchronous step (lookup, all-pairs with cutoff, wall search, integration) for a sample of
              agents against all N, scaled to the active agents.
    kind = "port" (the reference itself cannot travel to this box)."""
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    os.environ["OMP_NUM_THREADS"] = str(cores)
    from oracle import oc_oracle_c as occ
    P = occ.params(cfg, sim.room_length, sim.room_height, sim.Nx, sim.Ny)
    n = sim.Nx * sim.Ny
    key = sim._key_names[0]
    V = np.ascontiguousarray(sim.Vs[key])
    m = np.zeros_like(V)
    rng = np.random.RandomState(1)
    phi = np.exp(rng.normal(size=n))
    Ks = [rng.normal(size=n) for _ in range(7)]

    def timeit(fn, share):
        fn()
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < budget_s * share or k < 2:
            fn()
            k += 1
        return (time.perf_counter() - t0) / k, k
    t_rhs, n_rhs = timeit(lambda: occ.hjb_rhs(P, phi, V, m), 0.12)
    t_lin3, _ = timeit(lambda: occ.lincomb(phi, Ks[:3], np.ones(3)), 0.06)
    t_lin7, _ = timeit(lambda: occ.lincomb(phi, Ks, np.ones(7)), 0.06)
    t_vels, n_vels = timeit(lambda: occ.vels(P, phi), 0.1)
    nt = sim._nt0
    t_attempt = 6 * t_rhs + 5 * t_lin3 + 2 * t_lin7            # stages 1-5 average 3 arrays; y_new and error use 6-7
    t_hjb = n_attempt * t_attempt + (nt - 1) * (t_lin7 + t_vels)
    # GCFM sample on the current state
    st = sim._engine.get_state()
    N = sim.N
    wms = [occ.WallMap(sim._X0, sim._Y0, sim.Vs[k]) for k in sim._key_names]
    rects = [[sim._room["targets"][t] for t in sim._possible[k]] for k in sim._key_names]
    noise = rng.normal(size=(N, 2))
    ns = min(N, 256)
    t0 = time.perf_counter()
    done = 0
    while done < N and (time.perf_counter() - t0 < budget_s * 0.5 or done == 0):
        occ.step_sync(P, st["x"], st["y"], st["vx"], st["vy"], sim._v_des, st["status"], sim._key_of,
                      [None] * len(wms), sim._X0, sim._Y0, wms, rects, noise, i0=done, i1=min(N, done + ns))
        done = min(N, done + ns)
    t_agent = (time.perf_counter() - t0) / done
    t_step = t_agent * int(st["status"].sum())
    K = len(sim._key_names)
    value = nt / (K * t_hjb + nt * t_step)
    return {"value": value, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": "C/OpenMP float64 oracle on %d threads: %d full-grid RHS (%.4f s), stage combinations, %d vels "
                      "(%.4f s) -> one RK attempt %.3f s x %d attempts + %d output slices per key; GCFM: complete "
                      "synchronous step for %d of %d agents vs all agents (%.2e s/agent); scaled to K=%d keys, "
                      "nt=%d steps" % (cores, n_rhs, t_rhs, n_vels, t_vels, t_attempt, n_attempt, nt - 1, done, N,
                                       t_agent, K, nt),
            "t_hjb_solve_s": t_hjb, "t_step_s": t_step}


def gcfm_work(sim, cfg, t_forces_s, t_step_s):
    """SURVEY 8d's algorithmic work of one GCFM step on the state the run ended with (untimed): ordered pairs
    P = N_a (N_a - 1), of which P_near lie within the cutoff (counted here with torch on the GPU, float64), nominal
    8 flop per far pair and 60 per near pair; 88 B of state traffic per active agent.  The far pairs are never
    visited (tile boxes and neighbour lists cull them), so the all-pairs figure is an equivalent rate, not executed
    flops; the near-pair figure is what the pair kernel really evaluates."""
    import torch
    a = sim._engine.agents
    act = a["status"].bool()
    x, y = a["x"][act], a["y"][act]
    na = int(x.numel())
    c2 = float(cfg["repulsion_cutoff"]) ** 2
    near = 0
    for i0 in range(0, na, 2048):
        dx = x[i0:i0 + 2048, None] - x[None, :]
        dy = y[i0:i0 + 2048, None] - y[None, :]
        near += int(((dx * dx + dy * dy).sqrt() < float(cfg["repulsion_cutoff"])).sum().item())
    near -= na                                              # the agent itself
    P = na * (na - 1)
    f_all = 8.0 * (P - near) + 60.0 * near
    return {"active_agents": na, "ordered_pairs": P, "near_pairs": near,
            "flop_all_pairs_equivalent": f_all, "flop_near_pairs": 60.0 * near,
            "all_pairs_equivalent_TFLOPs": f_all / t_forces_s / 1e12,
            "near_pair_TFLOPs": 60.0 * near / t_forces_s / 1e12,
            "near_pair_frac_of_fp32_vector_peak": 60.0 * near / t_forces_s / 157.3e12,
            "state_bytes_frac_of_hbm_peak": 88.0 * na / t_step_s / (HBM_PEAK_GBS * 1e9)}


def pmc_traffic(workload):
    """HBM bytes per k_out launch from the committed PMC summary (profiles/rNN_pmc_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very command, gfx950 FETCH_SIZE x2 correction applied by
    tools/collect_profiles.py).  Counters cannot be collected from inside the timed run, hence the file; None if
    absent or for another workload."""
    import glob
    if workload != "metro_station":
        return None
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")))
    for f in reversed(files):
        try:
            v = json.load(open(f)).get("k_out_hbm_bytes_per_launch")
            if v:
                return v
        except Exception:
            pass
    return None


def pmc_valu(workload, counter="SQ_INSTS_VALU"):
    """A per-launch SQ counter of k_forces (default: wave-level VALU instructions) from the committed summary
    (profiles/rNN_pmc_step_kernels.txt, tools/pmc_step.sh: rocprofv3 --pmc passes of this very command); None if
    absent or for another workload."""
    import glob
    import re
    if workload != "metro_station":
        return None
    for f in reversed(sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_step_kernels.txt")))):
        try:
            txt = open(f).read()
            blk = txt[txt.index("k_forces"):]
            blk = blk[:blk.index("k_finish")] if "k_finish" in blk else blk
            m = re.search(counter + r"\s+n=\d+\s+mean=([0-9.e+]+)", blk)
            if m:
                return float(m.group(1))
        except Exception:
            pass
    return None


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
    ap.add_argument("--mode", default="ensemble", choices=["ensemble", "sharded"],
                    help="N > 1: ensemble = one independent simulation per GPU (default, weak scaling); sharded = ONE "
                         "simulation on N GPUs: HJB grid in row blocks, agents replicated, one all-reduce per step "
                         "(strong scaling; DESIGN.md section 7)")
    ap.add_argument("--tune", default="", help="launch-shape knobs, e.g. pair_split=4 (oc_set_tuning)")
    args = ap.parse_args()

    import torch
    from optimal_crowds_amd import parallel
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    # OC_BENCH_ONE_DEVICE=1 (rehearsal on a 1-GPU box): every rank uses cuda:0 and the timing reductions use gloo
    one_dev = os.environ.get("OC_BENCH_ONE_DEVICE") == "1"
    torch.cuda.set_device(0 if one_dev else local_rank)
    rank, world = parallel.init_distributed("gloo" if one_dev else "nccl")
    dist = None
    if world > 1:
        import torch.distributed as dist
    red_dev = "cpu" if one_dev else "cuda"

    import optimal_crowds_amd.engine  # noqa: F401  (fails loudly without the HIP library)
    t_init = time.perf_counter()
    sharded = args.mode == "sharded"
    if sharded:     # one simulation on all ranks: same seed everywhere (replicated agents), row blocks of the HJB grid
        # (N = 1: the same code path with a one-rank "world", to see what the sharded path itself costs)
        shd = parallel.dist_sharding() if world > 1 else parallel.thread_sharding(0, parallel.ThreadWorld(1),
                                                                                  parallel.ThreadWorld(1))
        sim, cfg = build_sim(args.workload, wall_search_radius=args.wall_radius, sharding=shd)
    else:           # N > 1: an ensemble -- every rank simulates its own member (own placement and noise seed)
        sim, cfg = build_sim(args.workload, seed=20240807 + rank, wall_search_radius=args.wall_radius)
    t_init = time.perf_counter() - t_init
    eng = sim._engine
    for kv in filter(None, args.tune.split(",")):
        name, val = kv.split("=")
        eng.set_tuning(name, int(val))
    nt = sim._nt0
    K, W = args.steps, args.warmup

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup (untimed): one solve round + W steps
    if W > 0:
        run_steps(sim, W, nt)
    barrier()
    # ---- timed region: exactly K steps (+ the solve rounds run() would do)
    eng.set_profiling(1 | 4)        # HJB kernels always, GCFM kernels on every 16th step (event records cost gaps)
    eng.reset_kernel_stats()
    stats = []
    barrier()
    t0 = time.perf_counter()
    run_steps(sim, K, nt, stats)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(elapsed, device=red_dev)

    # ---- breakdown (untimed extras for the report)
    t_hjb = sum(s["seconds"] for rnd in stats for s in rnd)
    nfev = stats[0][0]["nfev"] if stats else 0
    n_att = stats[0][0]["n_attempt"] if stats else 0
    n_acc = stats[0][0]["n_accept"] if stats else 0
    ncell = sim.Nx * sim.Ny
    kstats = {}
    for name in ("hjb_out", "hjb_stage", "gcfm_forces", "gcfm_finish"):
        nl, ms, by = eng.kernel_stats(name)
        kstats[name] = {"launches": nl, "avg_ms": (ms / nl if nl else None), "total_ms": ms,
                        "algorithmic_GB_per_launch": (by / nl / 1e9 if nl else None),
                        "achieved_GBps": (by / (ms * 1e-3) / 1e9 if ms > 0 else None)}
    # dominant HBM-bound kernel: the dense-output + direction-field writer (one launch per accepted RK step)
    roof = None
    ko = kstats["hjb_out"]
    if ko["launches"]:
        ach = ko["achieved_GBps"]
        roof = {"bound": "hbm", "kernel": "k_out (HJB dense output + direction-field writer)",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args.workload), "traffic_unit": "HBM bytes per launch (rocprofv3 PMC passes)",
                "launches": ko["launches"], "avg_launch_ms": ko["avg_ms"],
                "algorithmic_GB_per_launch": ko["algorithmic_GB_per_launch"]}
    result = {
        "metric": "sim steps/sec (HJB sweep + GCFM update), 2048^2 grid x 100k agents",
        "value": (1 if sharded else world) * K / elapsed, "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "strong" if sharded else "weak",
        "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s %dx%d grid, %d agents, %d keys, T=%g (nt=%d)"
                               % (args.workload, sim.Nx, sim.Ny, sim.N, len(sim._key_names), sim.T, nt)
                   + (", recompute=True" if sim.recompute else ""),
                   "parallelism": ("one simulation on %d GPUs: HJB row blocks + replicated agents" % world if sharded else
                                   "ensemble of %d replicas, one per GPU" % world) if world > 1 else "single GPU",
                   "noise": "device Philox", "agent_state": "f64", "pair_math": "f32", "hjb_state": "f32"},
        "roofline": roof,
        "breakdown": {"hjb_solve_s_total": t_hjb, "hjb_solves": sum(len(r) for r in stats),
                      "gcfm_ms_per_step": 1e3 * (elapsed - t_hjb) / K, "nfev": nfev, "rk_attempts": n_att,
                      "rk_accepted": n_acc, "kernels": kstats, "init_s": t_init, "active_agents_end": int(eng.counters.cpu()[2])},
    }
    kf = kstats.get("gcfm_forces", {})
    nv = pmc_valu(args.workload)
    if kf.get("avg_ms") and nv:
        # the step's dominant kernel is bound by vector-instruction issue, not by HBM or MFMA: a plain (unpacked) VALU
        # instruction of a wavefront occupies its SIMD for 4 cycles -> peak = 1024 SIMDs x 2.4 GHz / 4
        peak = 256 * 4 * 2.4e9 / 4
        ach = nv / (kf["avg_ms"] * 1e-3)
        result["roofline_valu"] = {"bound": "valu-issue", "kernel": "k_forces (pair forces + wall search + lookup)",
                                   "achieved": ach / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s",
                                   "frac": ach / peak, "valu_instructions_per_launch": nv,
                                   "avg_launch_ms": kf["avg_ms"], "source": "rocprofv3 --pmc SQ_INSTS_VALU (profiles/)"}
        wc, bc, nw = (pmc_valu(args.workload, c) for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES"))
        if wc and bc and nw:
            # SQ_WAVE_CYCLES counts quad-cycles summed over the waves; SQ_BUSY_CYCLES is summed over the chip's 32
            # shader engines (8 XCDs x 4), so busy/32 is the launch's length in shader cycles
            resident = 4.0 * wc / (bc / 32.0)
            result["roofline_valu"]["occupancy"] = {
                "waves_per_simd_allowed_by_registers": 4, "waves_per_simd_hardware": 8,
                "mean_resident_waves_per_simd": resident / 1024.0, "waves_per_launch": nw,
                "mean_wave_lifetime_us": 4.0 * wc / nw / (bc / 32.0) * kf["avg_ms"] * 1e3,
                "source": "SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAVES (profiles/)"}
    if rank == 0 and kf.get("avg_ms") and not args.no_work_count:
        result["breakdown"]["gcfm_work"] = gcfm_work(sim, cfg, kf["avg_ms"] * 1e-3,
                                                     result["breakdown"]["gcfm_ms_per_step"] * 1e-3)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(sim, cfg, n_att, budget_s=float(os.environ.get("OC_BENCH_CPU_BUDGET", "20")))
        result["cpu_baseline"]["host_cpus"] = os.cpu_count()
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
