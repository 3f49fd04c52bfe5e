"""N > 1 host logic on CPU: world_size-2 gloo runs of the row-block HJB partition (halo exchange + scalar
all-reduce) and of the bench's timing reductions.  The oracle's RHS stands in for the HIP kernel."""
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


def _rhs_block(padded, V_blk, m_blk, sigma, mu, g, dx):
    """5-point RHS of optimals.py:144-164 on a row block given its two halo rows (columns mirror locally)."""
    p = np.empty((padded.shape[0], padded.shape[1] + 2))
    p[:, 1:-1] = padded
    p[:, 0] = p[:, 2]
    p[:, -1] = p[:, -3]
    lap = (p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] - 4 * p[1:-1, 1:-1]) / (dx * dx)
    out = -0.5 * sigma ** 2 * lap - ((V_blk + g * m_blk) * p[1:-1, 1:-1]) / (mu * sigma ** 2)
    out[V_blk < 0] = 0
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from oracle import oc_oracle as oc
    from optimal_crowds_amd import parallel
    r, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    rng = np.random.RandomState(0)                       # same data on every rank
    Ny, Nx = 37, 24
    phi = np.exp(rng.normal(size=(Ny, Nx)))
    V = np.where(rng.rand(Ny, Nx) < 0.15, -100.0, 0.0)
    V[0], V[-1], V[:, 0], V[:, -1] = -100, -100, -100, -100
    V[0, 5:9] = 1.0                                       # a target on the border: the mirror rows matter
    V[-1, 10:14] = 1.0
    m = rng.rand(Ny, Nx)
    full = oc.hjb_rhs(phi.ravel(), V, m, 0.2, 5, -0.005, 0.05, 0.05).reshape(Ny, Nx)
    r0, r1 = parallel.row_blocks(Ny, world)[rank]
    blk = torch.as_tensor(phi[r0:r1].copy())
    above, below = parallel.halo_exchange(blk, rank, world)
    pad = parallel.padded_block(blk, above, below).numpy()
    mine = _rhs_block(pad, V[r0:r1], m[r0:r1], 0.2, 5, -0.005, 0.05)
    ok_rhs = np.array_equal(mine, full[r0:r1])
    s, mx = parallel.error_norm_allreduce(float((mine ** 2).sum()), float(np.abs(mine).max()))
    ok_red = abs(s - (full ** 2).sum()) <= 1e-9 * (full ** 2).sum() and mx == np.abs(full).max()
    tmax = parallel.max_over_ranks(1.0 + rank)
    tsum = parallel.sum_over_ranks(10.0)
    keys = parallel.key_partition(5, world, rank)
    q.put((rank, ok_rhs, ok_red, tmax, tsum, keys, (r0, r1)))
    dist.barrier()
    dist.destroy_process_group()


def test_row_block_rhs_and_reductions_world2():
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
    for rank, ok_rhs, ok_red, tmax, tsum, keys, rows in res:
        assert ok_rhs and ok_red
        assert tmax == 2.0 and tsum == 20.0
    assert res[0][5] == [0, 2, 4] and res[1][5] == [1, 3]
    assert res[0][6] == (0, 19) and res[1][6] == (19, 37)


def test_row_blocks_cover_grid():
    from optimal_crowds_amd import parallel
    for Ny in (16, 2048, 4097):
        for world in (1, 2, 3, 8):
            b = parallel.row_blocks(Ny, world)
            assert b[0][0] == 0 and b[-1][1] == Ny
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [r1 - r0 for r0, r1 in b]
            assert max(sizes) - min(sizes) <= 1
