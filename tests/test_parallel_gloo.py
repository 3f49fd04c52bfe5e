"""N > 1 host logic on CPU: world_size-2 gloo runs of the key-first / row-block HJB partition (the communicators the
product uses: halo exchange + all-gather of the error norm), of the step's all-reduce and of the bench's timing
reductions.  The oracle's RHS stands in for the HIP kernel in the row-block RK45 model."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StubEngine:
    """what parallel.dist_sharding needs of an engine (no GPU here)"""

    def __init__(self, n_keys, Ny, Nx):
        self.n_keys, self.Ny, self.Nx, self.tdtype, self.device = n_keys, Ny, Nx, torch.float64, torch.device("cpu")


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from optimal_crowds_amd import parallel
    r, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    tmax = parallel.max_over_ranks(1.0 + rank)
    tsum = parallel.sum_over_ranks(10.0)
    # keys first, then row blocks: 2 keys on 2 ranks -> one whole key per rank, no solve-time collective;
    # 1 key on 2 ranks -> two row blocks with a neighbour each
    eng2 = _StubEngine(2, 64, 16)
    comms2, step2 = parallel.dist_sharding(halo=8)(eng2)
    held = [k for k in range(2) if comms2[k] is not None]
    shape2 = [(comms2[k].shard.world, comms2[k].shard.g0, comms2[k].shard.g1) for k in held]
    eng1 = _StubEngine(1, 64, 16)
    comms1, step1 = parallel.dist_sharding(halo=8, min_cells=1)(eng1)      # (by default a grid this small stays whole)
    sh = comms1[0].shard
    # the two collectives of a row-block solve through the key's communicator
    comms1[0].red_local.copy_(torch.tensor([1.0 + rank, 10.0 * (rank + 1)], dtype=torch.float64))
    comms1[0].allgather_red()
    red = comms1[0].red_all.clone().numpy()
    comms1[0].send_up.fill_(100.0 + rank)
    comms1[0].send_dn.fill_(200.0 + rank)
    comms1[0].exchange_halo()
    got = (float(comms1[0].recv_up[0, 0, 0]) if sh.has_up else None, float(comms1[0].recv_dn[0, 0, 0]) if sh.has_dn else None)
    q.put((rank, tmax, tsum, eng2.key_plan, held, shape2, eng1.key_plan, (sh.world, sh.g0, sh.g1), red.tolist(), got,
           (step1.rank, step1.world)))
    dist.barrier()
    dist.destroy_process_group()


def test_key_first_plan_and_collectives_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, tmax, tsum, plan2, held, shape2, plan1, shape1, red, got, stepc in res:
        assert tmax == 2.0 and tsum == 20.0
        assert plan2 == [(0,), (1,)] and held == [rank] and shape2 == [(1, 0, 64)]
        assert plan1 == [(0, 1)] and shape1 == (2, 32 * rank, 32 * rank + 32)
        assert red == [[1.0, 10.0], [2.0, 20.0]]
        assert stepc == (rank, 2)
    assert res[0][9] == (None, 101.0) and res[1][9] == (200.0, None)      # my lower neighbour's send_up / upper's send_dn


def test_key_row_plan():
    from optimal_crowds_amd import parallel
    assert parallel.key_row_plan(2, 4) == [(0, 1), (2, 3)]                 # C4 on 4 GPUs: 2 keys x 2-way rows
    assert parallel.key_row_plan(2, 8) == [(0, 1, 2, 3), (4, 5, 6, 7)]
    assert parallel.key_row_plan(1, 8) == [tuple(range(8))]                # C5: one key over all ranks
    assert parallel.key_row_plan(2, 1) == [(0,), (0,)]
    assert parallel.key_row_plan(3, 2) == [(0,), (1,), (0,)]
    assert parallel.key_row_plan(2, 3) == [(0, 1), (2,)]
    # with the grid's size: row blocks only where a block keeps ROW_MIN_CELLS cells (a smaller one is latency-bound anyway)
    assert parallel.key_row_plan(2, 4, ncell=2048 * 2048) == [(0,), (2,)]          # C4: keys whole at any N
    assert parallel.key_row_plan(2, 8, ncell=2048 * 2048) == [(0,), (4,)]
    assert parallel.key_row_plan(1, 8, ncell=4096 * 4096) == [(0, 1, 2, 3)]        # C5: 4-way rows
    assert parallel.key_row_plan(1, 2, ncell=4096 * 4096) == [(0, 1)]
    assert parallel.key_row_plan(1, 2, ncell=1024 * 1024) == [(0,)]
    assert parallel.key_row_plan(1, 2, ncell=1024 * 1024, min_cells=1) == [(0, 1)]
    for K in range(1, 6):
        for W in range(1, 9):
            plan = parallel.key_row_plan(K, W)
            assert len(plan) == K and all(len(p) >= 1 for p in plan)
            if W >= K:                                                      # every rank solves exactly one key
                assert sorted(r for p in plan for r in p) == list(range(W))


def test_row_blocks_cover_grid():
    from optimal_crowds_amd import parallel
    for Ny in (16, 2048, 4097):
        for world in (1, 2, 3, 8):
            b = parallel.row_blocks(Ny, world)
            assert b[0][0] == 0 and b[-1][1] == Ny
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [r1 - r0 for r0, r1 in b]
            assert max(sizes) - min(sizes) <= 1


# ------------------------------------------------------------------------------ wide-halo row-block RK45 over gloo
def _rows_model_solve(oc, comm, V_loc, m_loc, hp, dx, T, nt, n_cells_global):
    """Python model of oc_hjb_solve_rows' control flow (csrc/hjb.hip hjb_solve_t with a shard): local arrays of
    Ny_local rows, the RHS mirrors at the LOCAL edges (as the kernels do), error norms over owned rows completed by
    comm.allgather_red, (y_new, K6) halo rows refreshed by comm.exchange_halo after every accepted step.
    Returns (phi slices on the local rows, n_attempt, n_accept)."""
    sh = comm.shard
    Nyl, Nx = V_loc.shape
    lo, hi, H = sh.own_lo, sh.own_hi, sh.halo
    f = lambda yy: oc.hjb_rhs(yy.ravel(), V_loc, m_loc, hp["sigma"], hp["mu"], hp["g"], dx, dx).reshape(Nyl, Nx)
    rtol, atol = 1e-3, 1e-6

    def gsum(local_sq, local_max=0.0):
        comm.red_local.copy_(torch.tensor([local_sq, local_max], dtype=torch.float64))
        assert comm._red(None) == 0, comm.error
        ra = comm.red_all.numpy()
        return float(sum(ra[r, 0] for r in range(sh.world))), float(max(ra[:, 1]))

    def refresh(y_new, k6):
        for buf, rows in ((comm.send_up, slice(lo, lo + H)), (comm.send_dn, slice(hi - H, hi))):
            buf[0].copy_(torch.as_tensor(y_new[rows]))
            buf[1].copy_(torch.as_tensor(k6[rows]))
        assert comm._halo(None) == 0, comm.error
        if sh.has_up:
            y_new[lo - H:lo], k6[lo - H:lo] = comm.recv_up[0].numpy(), comm.recv_up[1].numpy()
        if sh.has_dn:
            y_new[hi:hi + H], k6[hi:hi + H] = comm.recv_dn[0].numpy(), comm.recv_dn[1].numpy()

    own = slice(lo, hi)
    rms = lambda v: np.sqrt(gsum(float((v[own] ** 2).sum()))[0]) / np.sqrt(n_cells_global)
    y = np.ones((Nyl, Nx))
    t, tf, direction = T, 0.0, -1.0
    fcur = f(y)
    scale = atol + np.abs(y) * rtol
    d0, d1 = rms(y / scale), rms(fcur / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    h0 = min(h0, T)
    f1 = f(y + h0 * direction * fcur)
    d2 = rms((f1 - fcur) / scale) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** 0.2
    h_abs = min(100 * h0, h1, T)
    te = np.linspace(T, 0, nt)
    k_next, out = 0, {}
    K = np.empty((7, Nyl, Nx))
    n_attempt = n_accept = 0
    while t != tf:
        min_step = 10 * abs(np.nextafter(t, -np.inf) - t)
        h_abs = max(h_abs, min_step)
        accepted = rejected = False
        while not accepted:
            h = h_abs * direction
            t_new = t + h
            if direction * (t_new - tf) > 0:
                t_new = tf
            h = t_new - t
            h_abs = abs(h)
            n_attempt += 1
            K[0] = fcur
            for s in range(1, 6):
                K[s] = f(y + np.tensordot(oc.RK_A[s, :s], K[:s], axes=1) * h)
            y_new = y + h * np.tensordot(oc.RK_B, K[:6], axes=1)
            K[6] = f(y_new)
            sc = atol + np.maximum(np.abs(y), np.abs(y_new)) * rtol
            e = np.tensordot(oc.RK_E, K, axes=1) * h / sc
            err = np.sqrt(gsum(float((e[own] ** 2).sum()))[0]) / np.sqrt(n_cells_global)
            if err < 1:
                factor = 10.0 if err == 0 else min(10.0, 0.9 * err ** -0.2)
                h_abs *= min(1, factor) if rejected else factor
                accepted = True
            else:
                h_abs *= max(0.2, 0.9 * err ** -0.2)
                rejected = True
        n_accept += 1
        refresh(y_new, K[6])
        while k_next < nt and te[k_next] >= t_new:
            x = (te[k_next] - t) / h
            Q = np.tensordot(K, oc.RK_P, axes=([0], [0]))              # (Nyl, Nx, 4)
            out[k_next] = y + h * (Q @ np.cumprod(np.tile(x, 4)))
            k_next += 1
        t, y, fcur = t_new, y_new, K[6].copy()
    return out, n_attempt, n_accept


def _rows_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from oracle import oc_oracle as oc
    from optimal_crowds_amd import parallel
    parallel.init_distributed("gloo")
    rng = np.random.RandomState(5)
    Ny, Nx, dx = 30, 16, 0.1
    V = np.where(rng.rand(Ny, Nx) < 0.12, -100.0, 0.0)
    V[0], V[-1], V[:, 0], V[:, -1] = -100, -100, -100, -100
    V[12:16, -1] = 1.0
    m = rng.rand(Ny, Nx)
    hp = oc.DEFAULT_CONFIG["hjb_params"]
    T, nt = 1.0, 11
    sh = parallel.RowShard(Ny, rank, world, halo=7)
    comm = parallel.DistHaloComm(sh, Nx, torch.float64, "cpu")
    out, n_att, n_acc = _rows_model_solve(oc, comm, V[sh.rows()], m[sh.rows()], hp, dx, T, nt, Ny * Nx)
    ys, st = oc.rk45_solve(lambda t, y: oc.hjb_rhs(y, V, m, hp["sigma"], hp["mu"], hp["g"], dx, dx), T, 0.0,
                           np.ones(Ny * Nx), np.linspace(T, 0, nt))
    ys = ys.reshape(Ny, Nx, nt)
    worst = max(np.abs(out[k][sh.own_lo:sh.own_hi] / ys[sh.g0:sh.g1, :, k] - 1).max() for k in range(nt))
    q.put((rank, n_att == st["n_attempt"], n_acc == st["n_accept"], float(worst), comm.n_exchange, comm.n_allgather))
    dist.barrier()
    dist.destroy_process_group()


def test_wide_halo_row_block_rk45_world2():
    """The halo-depth argument of oc_hjb_solve_rows (include/oc_hip.h) and DistHaloComm over gloo: a Python model of
    the decomposed controller reproduces the oracle's single-domain RK45 (same steps, phi to 1e-11 relative)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_att, same_acc, worst, n_ex, n_ag in res:
        assert same_att and same_acc
        assert worst <= 1e-11
        assert n_ex > 0 and n_ag >= n_ex + 3


def test_row_shard_geometry():
    from optimal_crowds_amd import parallel
    import pytest
    for Ny, world, halo in ((2048, 8, 8), (4096, 8, 16), (100, 3, 7)):
        prev = None
        for r in range(world):
            s = parallel.RowShard(Ny, r, world, halo)
            assert s.own_hi - s.own_lo == s.g1 - s.g0 >= halo
            assert s.row0 + s.own_lo == s.g0
            assert s.Ny_local == (s.g1 - s.g0) + halo * (int(s.has_up) + int(s.has_dn))
            assert s.rows().start >= 0 and s.rows().stop <= Ny
            if prev is not None:
                assert prev.g1 == s.g0
            prev = s
        assert prev.g1 == Ny
    with pytest.raises(ValueError):
        parallel.RowShard(40, 0, 8, 7)              # blocks of 5 rows cannot carry the halo
    with pytest.raises(ValueError):
        parallel.RowShard(2048, 0, 8, 6)            # the output reads the local K6 one row beyond the block: halo >= 7


# ------------------------------------------------------------------------------------------------ sharded-step plumbing
def _step_comm_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from optimal_crowds_amd import parallel
    parallel.init_distributed("gloo")
    # the exchange buffer of oc_gcfm_step_sharded (oc_hip.h: oc_step_comm): `world` blocks -- a rank writes its own, the
    # in-place all-gather brings the others' -- and, behind them, the values that are summed (one writer per element)
    per, nsum = 100, 6 * 16
    rs = np.random.RandomState(5)
    blocks = rs.normal(size=(world, per))
    vals = rs.normal(size=nsum)
    owner = np.arange(nsum) % world
    buf = torch.zeros(world * per + 7 + nsum, dtype=torch.float64)         # (7 values between the two areas stay untouched)
    buf[rank * per:(rank + 1) * per] = torch.as_tensor(blocks[rank])
    buf[world * per + 7:] = torch.as_tensor(np.where(owner == rank, vals, 0.0))
    comm = parallel.DistStepComm()
    comm.bind(buf)
    st = comm.struct()
    assert st.rank == rank and st.world == world and st.xchg == buf.data_ptr()
    rc = comm._ag(None, buf.data_ptr(), per)                    # what the library's callbacks do
    rc_b = comm._ar(None, buf.data_ptr() + 8 * (world * per + 7), nsum)
    ok = (rc == 0 and rc_b == 0 and comm.error is None and comm.n_allgather == 1 and comm.n_allreduce == 1
          and np.array_equal(buf[:world * per].numpy(), blocks.reshape(-1)) and not buf[world * per:world * per + 7].any()
          and np.array_equal(buf[world * per + 7:].numpy(), vals)
          and comm.bytes_allgather == 8 * per * world and comm.bytes_allreduce == 8 * nsum)
    # a buffer the library did not announce is refused (and must not raise through the C frames)
    other = torch.zeros(4, dtype=torch.float64)
    rc2 = comm._ar(None, other.data_ptr(), 4)
    refused = rc2 == -1 and isinstance(comm.error, AssertionError)
    # row-block geometry the sharded lookup relies on: the owned rows tile the grid, halo >= 8 leaves a margin row
    sh = parallel.RowShard(64, rank, world, halo=8)
    q.put((rank, ok, refused, (sh.g0, sh.g1, sh.row0, sh.Ny_local, sh.field_margin)))
    dist.barrier()
    dist.destroy_process_group()


def test_step_comm_and_row_shards_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_step_comm_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(r[1] and r[2] for r in res)
    (g0a, g1a, row0a, nla, mga), (g0b, g1b, row0b, nlb, mgb) = res[0][3], res[1][3]
    assert (g0a, g1a, g0b, g1b) == (0, 32, 32, 64)
    assert (row0a, nla) == (0, 40) and (row0b, nlb) == (24, 40) and mga == mgb == 1
