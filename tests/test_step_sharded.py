"""The GCFM step on several ranks with replicated agents (oc_gcfm_step_sharded): every rank does the pair /
wall forces of its share of the tiles and the floor-field lookups its rows cover, ONE sum over the ranks, every rank
finishes all agents.  Virtual ranks = threads of one process on cuda:0, one engine (library handle) each; the collective
is parallel.ThreadStepComm (parallel.DistStepComm is the same callback over torch.distributed / RCCL).

Against the single-GPU step: every force is computed by exactly one rank with the single-GPU arithmetic and the sum
adds zeros, so the states agree to the order in which a tile's wavefronts add their float64 partial sums (1e-12)."""
import copy
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oc_oracle as oc                     # noqa: E402  (checker only)
from optimal_crowds_amd import room_library            # noqa: E402
from tests.test_hip_parity import _engine, _rects      # noqa: E402
from tests.test_hip_step_shapes import _big            # noqa: E402

T, NT = 1.0, 50


def _potentials(room, cfg, X0, Y0):
    X, Y = np.meshgrid(X0, Y0)
    names = []
    for box in room["initial_boxes"].values():
        k = " or ".join(box[5:])
        if k not in names:
            names.append(k)
    return names, [oc.finalize_potential(oc.create_potential(room, n.split(" or "), X, Y), cfg) for n in names]


def _state(room, nkeys, n=6000, seed=5):
    rng = np.random.RandomState(seed)
    L, H = room["room_length"], room["room_height"]
    x, y = rng.uniform(0.8, L - 0.8, n), rng.uniform(0.8, H - 0.8, n)
    vx, vy = rng.normal(0, 0.6, n), rng.normal(0, 0.6, n)
    vdes = rng.normal(1.34, 0.2, n).clip(0.6, 2.0)
    key = rng.randint(0, nkeys, n).astype(np.int32)
    status = rng.rand(n) > 0.03
    return x, y, vx, vy, vdes, key, status


FIXES = ("lookup", "wall_force", "pair_force")          # the opt-in corrections (oc_set_fix_flags) shard the same way


def _new(room_name, fixes=()):
    eng, room, cfg, X0, Y0 = _engine(room_name, "float32", n_keys=2, max_agents=8192, cfg=_big(), wall_search_radius=0.0)
    eng.set_fix_flags(fixes)
    names, Vs = _potentials(room, cfg, X0, Y0)
    for k, name in enumerate(names):
        eng.set_potential(k, Vs[k], _rects(room, name))
    return eng, room, cfg, names


def _run_ranks(world, body):
    from optimal_crowds_amd import parallel
    tw = parallel.ThreadWorld(world)
    out, errs = [None] * world, []

    def work(rank):
        try:
            out[rank] = body(rank, tw)
        except BaseException as e:      # noqa: BLE001
            errs.append((rank, e))
            tw.barrier.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    if errs:
        raise errs[0][1]
    return out


def _reference(room_name, state, noise, steps, fixes=()):
    eng, room, cfg, names = _new(room_name, fixes)
    for k in range(len(names)):
        eng.alloc_field(k, NT - 1)
        eng.solve(k, T, NT)
    x, y, vx, vy, vdes, key, status = state
    eng.set_agents(x, y, vx, vy, vdes, key, status=status)
    eng.set_resort_every(2)
    outs = [eng.step(s, noise=noise[s]) for s in range(steps)]
    st = eng.get_state()
    eng.close()
    return st, outs


@pytest.mark.parametrize("world,fixes", [(1, ()), (2, ()), (3, ()), (2, FIXES)])
def test_sharded_step_with_replicated_fields(world, fixes):
    from optimal_crowds_amd import parallel
    room = room_library.ROOMS["exit_opposite"]()
    state = _state(room, 2)
    n = len(state[0])
    steps = 4
    noise = np.random.RandomState(1).normal(size=(steps, n, 2))
    ref, ref_out = _reference("exit_opposite", state, noise, steps, fixes)

    def body(rank, tw):
        eng, room_, cfg, names = _new("exit_opposite", fixes)
        for k in range(len(names)):
            eng.alloc_field(k, NT - 1)
            eng.solve(k, T, NT)                      # every rank holds the whole field: the tile's owner looks up
        x, y, vx, vy, vdes, key, status = state
        eng.set_agents(x, y, vx, vy, vdes, key, status=status)
        eng.set_resort_every(2)
        comm = parallel.ThreadStepComm(rank, tw)
        tw.barrier.wait(timeout=120)
        outs = [eng.step_sharded(s, comm, noise=noise[s]) for s in range(steps)]
        st = eng.get_state()
        eng.close()
        return st, outs, (comm.n_allgather, comm.n_allreduce, comm.bytes_allgather)

    res = _run_ranks(world, body)
    for st, outs, nred in res:
        # every field whole on every rank: ONE all-gather per step and nothing else; 48 bytes per (padded) agent slot +
        # the ranks' block tails
        assert nred[:2] == (steps, 0)
        tiles = -(-n // 64)
        assert nred[2] == 8 * world * (-(-tiles // world) * 384 + 8)
        assert [(o[0], o[2]) for o in outs] == [(o[0], o[2]) for o in ref_out]        # exits, active agents
        assert np.array_equal(st["status"], ref["status"])
        for f in ("x", "y", "vx", "vy"):
            assert np.array_equal(st[f], res[0][0][f])                               # replicas stay bit-identical
            assert np.abs(st[f] - ref[f]).max() <= 1e-11, f


@pytest.mark.parametrize("fixes", [(), FIXES])
def test_sharded_step_with_row_block_fields(fixes):
    """The direction field domain-decomposed by rows (oc_hjb_solve_rows -> oc_set_field_rows): a lookup is answered by
    the rank that owns the first cell's row.  Against the single-GPU run: the two solves differ by float32 round-off in
    the step controller's error sums (test_hjb_rows), so the desired velocities agree to 1e-4 on all but a few cells at
    the unit-or-zero threshold."""
    import torch
    from optimal_crowds_amd import parallel
    world, halo = 2, 8
    room = room_library.ROOMS["exit_opposite"]()
    state = _state(room, 2)
    n = len(state[0])
    steps = 3
    noise = np.random.RandomState(2).normal(size=(steps, n, 2))
    ref, ref_out = _reference("exit_opposite", state, noise, steps, fixes)

    def body(rank, tw):
        eng, room_, cfg, names = _new("exit_opposite", fixes)
        sh = parallel.RowShard(eng.Ny, rank, world, halo)
        hcomm = parallel.ThreadHaloComm(sh, eng.Nx, eng.tdtype, eng.device, tw)
        tw.barrier.wait(timeout=120)
        for k in range(len(names)):
            _, (vx_, vy_), _ = eng.solve_rows(k, T, NT, hcomm)
            eng.set_field_rows(k, vx_, vy_, NT, sh)
        x, y, vx, vy, vdes, key, status = state
        eng.set_agents(x, y, vx, vy, vdes, key, status=status)
        eng.set_resort_every(2)
        tw.comms = [None] * world                       # the step's collective replaces the solve's in the registry
        tw.barrier.wait(timeout=120)
        comm = parallel.ThreadStepComm(rank, tw)
        tw.barrier.wait(timeout=120)
        outs = [eng.step_sharded(s, comm, noise=noise[s]) for s in range(steps)]
        st = eng.get_state()
        eng.close()
        return st, outs

    res = _run_ranks(world, body)
    for st, outs in res:
        assert np.array_equal(st["status"], ref["status"])
        for f in ("x", "y", "vx", "vy"):
            assert np.array_equal(st[f], res[0][0][f])
        d = np.hypot(st["x"] - ref["x"], st["y"] - ref["y"])
        assert np.median(d) <= 1e-7 and (d > 1e-4).mean() <= 2e-3


# ------------------------------------------------------------------------------------------------ two processes, gloo
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dist_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from optimal_crowds_amd import parallel
    try:
        parallel.init_distributed("gloo")
        room = room_library.ROOMS["exit_opposite"]()
        state = _state(room, 2, n=3000, seed=8)
        n = len(state[0])
        steps = 3
        noise = np.random.RandomState(3).normal(size=(steps, n, 2))
        eng, room_, cfg, names = _new("exit_opposite")
        sh = parallel.RowShard(eng.Ny, rank, world, halo=8)
        hcomm = parallel.DistHaloComm(sh, eng.Nx, eng.tdtype, eng.device)
        for k in range(len(names)):
            _, (vx_, vy_), _ = eng.solve_rows(k, T, NT, hcomm)
            eng.set_field_rows(k, vx_, vy_, NT, sh)
        x, y, vx, vy, vdes, key, status = state
        eng.set_agents(x, y, vx, vy, vdes, key, status=status)
        comm = parallel.DistStepComm()
        outs = [eng.step_sharded(s, comm, noise=noise[s]) for s in range(steps)]
        res = dict(rank=rank, st=eng.get_state(), outs=outs, nred=comm.n_allreduce, ngat=comm.n_allgather,
                   bytes=(comm.bytes_allgather, comm.bytes_allreduce))
        if rank == 0:
            ref, ref_out = _reference("exit_opposite", state, noise, steps)
            res.update(ref=ref, ref_out=ref_out)
        q.put(res)
        dist.barrier()
        eng.close()
        dist.destroy_process_group()
    except BaseException as e:      # noqa: BLE001
        q.put(dict(rank=rank, error=repr(e)))
        raise


def test_sharded_step_two_processes_gloo():
    """World-size-2 torch.distributed run (gloo, both ranks on cuda:0): row-block solve, row-block fields, sharded
    steps -- the code path of a multi-GPU node with the backend swapped (nccl = RCCL moves device tensors directly)."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    for r in res:
        assert "error" not in r, r
    res.sort(key=lambda r: r["rank"])
    ref = res[0]["ref"]
    for r in res:
        # row blocks: the desired velocities go by their own sum (16 B per slot), the forces by the all-gather (32 B)
        assert r["nred"] == 3 and r["ngat"] == 3
        n_slots = 64 * -(-len(ref["x"]) // 64)
        assert r["bytes"][1] == 16 * n_slots and r["bytes"][0] == 8 * world * (-(-(n_slots // 64) // world) * 256 + 8)
        assert np.array_equal(r["st"]["status"], ref["status"])
        for f in ("x", "y", "vx", "vy"):
            assert np.array_equal(r["st"][f], res[0]["st"][f])                       # replicas bit-identical
        d = np.hypot(r["st"]["x"] - ref["x"], r["st"]["y"] - ref["y"])
        assert np.median(d) <= 1e-7 and (d > 1e-4).mean() <= 2e-3


# ------------------------------------------------------------------------------------------------ the façade, sharded
def test_simulation_run_sharded_matches_single_gpu():
    """simulations.simulation(..., sharding=...).run() on 2 virtual ranks with recompute=True (density, re-solves on
    the shrinking horizon, sharded steps) against the same run on one engine: same exits, positions to 1e-6 m on all
    but a few agents (the row-block and the single solve differ by float32 round-off in the step controller)."""
    import contextlib
    import io
    from optimal_crowds_amd import host, parallel, simulations
    cfg = copy.deepcopy(host.load_config())
    cfg["grid_step"] = 0.1
    Tsim = 3.0

    def make(sharding):
        np.random.seed(7)
        with contextlib.redirect_stdout(io.StringIO()):
            sim = simulations.simulation("room_test", Tsim, True, config=cfg, noise="device", seed=11, record=False,
                                         history=False, wall_search_radius=0.0, sharding=sharding)
            sim.run()
        st = sim._engine.get_state()
        out = (st, sim.inside, sim.simu_step, [sim.targets[k].nt_opt for k in sim.targets])
        sim._engine.close()
        return out

    ref = make(None)
    world = 2
    tw_solve, tw_step = parallel.ThreadWorld(world), parallel.ThreadWorld(world)
    # np.random is process-global: the ranks construct one after the other (same seed -> same placement), then run
    lock = threading.Lock()
    sims = [None] * world

    def body(rank, tw):
        with lock:
            np.random.seed(7)
            with contextlib.redirect_stdout(io.StringIO()):
                sims[rank] = simulations.simulation("room_test", Tsim, True, config=cfg, noise="device", seed=11,
                                                    record=False, history=False, wall_search_radius=0.0,
                                                    sharding=parallel.thread_sharding(rank, tw_solve, tw_step))
        tw.barrier.wait(timeout=120)
        sim = sims[rank]
        buf = io.StringIO()
        # (redirect_stdout is process-wide; one rank's redirection covers both, harmless)
        with contextlib.redirect_stdout(buf):
            sim.run()
        st = sim._engine.get_state()
        out = (st, sim.inside, sim.simu_step, [sim.targets[k].nt_opt for k in sim.targets])
        sim._engine.close()
        return out

    res = _run_ranks(world, body)
    for st, inside, nstep, nts in res:
        assert (inside, nstep, nts) == (ref[1], ref[2], ref[3])
        assert np.array_equal(st["status"], ref[0]["status"])
        for f in ("x", "y", "vx", "vy"):
            assert np.array_equal(st[f], res[0][0][f])
        d = np.hypot(st["x"] - ref[0]["x"], st["y"] - ref[0]["y"])
        assert np.median(d) <= 1e-6 and (d > 1e-3).mean() <= 0.1


@pytest.mark.parametrize("world", [2, 4, 3])
def test_simulation_run_keys_first_then_rows(world):
    """SURVEY 8e "partition by key first": gold_b has two target keys.  2 virtual ranks = one whole key per rank (no
    solve-time collective at all); 4 ranks = 2 keys x 2-way row blocks; 3 ranks with one of them holding no key.  A rank holds no field of the other group's key,
    the step's lookups for those agents are answered by the ranks that do and travel with the all-reduce.  Against the
    same run on one engine: same exits, same horizons, positions as in the row-block test."""
    import contextlib
    import io
    from optimal_crowds_amd import host, parallel, simulations
    cfg = copy.deepcopy(host.load_config())
    Tsim = 2.0

    def make(sharding):
        np.random.seed(12)
        with contextlib.redirect_stdout(io.StringIO()):
            sim = simulations.simulation("gold_b", Tsim, True, config=cfg, noise="device", seed=11, record=False,
                                         history=False, wall_search_radius=0.0, sharding=sharding)
        return sim

    def finish(sim):
        with contextlib.redirect_stdout(io.StringIO()):
            sim.run()
        st = sim._engine.get_state()
        held = [k for k in range(len(sim._key_names)) if sim._engine.holds_key(k)]
        out = (st, sim.inside, sim.simu_step, [sim.targets[k].nt_opt for k in sim.targets], held)
        sim._engine.close()
        return out

    ref = finish(make(None))
    # world 3: the plan of a grid that is not worth splitting on 3 ranks -- keys whole on ranks 0 and 2, rank 1 holds NO
    # field at all and only takes its share of the step (parallel.key_row_plan with the grid's size)
    plan = parallel.key_row_plan(2, world) if world != 3 else parallel.key_row_plan(2, 3, ncell=1 << 22)
    assert plan == {2: [(0,), (1,)], 4: [(0, 1), (2, 3)], 3: [(0,), (2,)]}[world]
    tw_keys = [parallel.ThreadWorld(len(r)) for r in plan]
    tw_step = parallel.ThreadWorld(world)
    lock = threading.Lock()
    sims = [None] * world

    def body(rank, tw):
        with lock:          # np.random is process-global: the ranks construct one after the other
            sims[rank] = make(parallel.thread_sharding(rank, tw_keys, tw_step, plan=plan))
        tw.barrier.wait(timeout=120)
        return finish(sims[rank])

    res = _run_ranks(world, body)
    for rank, (st, inside, nstep, nts, held) in enumerate(res):
        assert held == [k for k in range(2) if rank in plan[k]]
        assert (inside, nstep, nts) == (ref[1], ref[2], ref[3])
        assert np.array_equal(st["status"], ref[0]["status"])
        for f in ("x", "y", "vx", "vy"):
            assert np.array_equal(st[f], res[0][0][f])                  # replicas bit-identical
        d = np.hypot(st["x"] - ref[0]["x"], st["y"] - ref[0]["y"])
        assert np.median(d) <= 1e-6 and (d > 1e-3).mean() <= 0.1


def test_index_error_band_is_seen_by_every_rank():
    """An agent in the reference's IndexError band (quirk Q8) is noticed by the rank that holds its key's field; the flag
    travels with the exchange buffer, so EVERY rank reports it (and none blocks in the next collective)."""
    from optimal_crowds_amd import parallel
    world = 2
    tw_keys = [parallel.ThreadWorld(1), parallel.ThreadWorld(1)]

    def body(rank, tw):
        eng, room, cfg, names = _new("gold_b")
        x, y, vx, vy, vdes, key, status = _state(room, 2, n=500)
        # one agent of key 1 inside the band x in [(Nx-3) dx, L - dx)
        Nx = eng.Nx
        x[7], key[7], status[7] = (Nx - 3) * cfg["grid_step"] + 0.01, 1, True
        # keys dealt to ranks: rank r holds key r whole
        eng.enable_sharding(*parallel.thread_sharding(rank, tw_keys, tw, plan=[(0,), (1,)])(eng))
        for k in range(2):
            eng.alloc_field(k, NT - 1)
            eng.solve(k, T, NT)
        eng.set_agents(x, y, vx, vy, vdes, key, status=status)
        exited, bad, active = eng.step(3, seed=5)
        eng.close()
        return bad

    res = _run_ranks(world, body)
    assert res[0] == res[1] == 8                                        # 1 + agent index, on both ranks
