"""Multi-GPU plumbing for the path (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm).

This round the bench runs independent replicas: no data-path collective.  The helpers below
are the host side of the partition that does shard -- the HJB grid by row blocks, or whole keys over ranks -- and
are covered by world-size-2 gloo tests on CPU (tests/test_parallel_gloo.py).  They move tensors only; the numerics
stay in the HIP library (on CPU tests the oracle's RHS stands in for it).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend, **kw)
    return rank, world


def max_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def key_partition(n_keys, world, rank):
    """Independent HJB solves: key k goes to rank k % world (zero communication during the solve)."""
    return [k for k in range(n_keys) if k % world == rank]


def row_blocks(Ny, world):
    """Row-block decomposition of the (Ny, Nx) grid: contiguous blocks, sizes differ by at most one row."""
    base, rem = divmod(Ny, world)
    out, r0 = [], 0
    for r in range(world):
        r1 = r0 + base + (1 if r < rem else 0)
        out.append((r0, r1))
        r0 = r1
    return out


def halo_exchange(block, rank, world, group=None):
    """block: (rows, Nx) tensor of this rank's rows.  Returns (above, below): the neighbour rows adjacent to the
    block (None at the room border, where the RHS mirrors: optimals.py:149-152).  One row per neighbour and RHS
    evaluation: 8 KB at Nx = 2048 -- latency-bound, so plain point-to-point, no ring."""
    above = below = None
    reqs = []
    if rank > 0:
        above = torch.empty_like(block[0])
        reqs.append(dist.isend(block[0].contiguous(), rank - 1, group=group))
        reqs.append(dist.irecv(above, rank - 1, group=group))
    if rank < world - 1:
        below = torch.empty_like(block[-1])
        reqs.append(dist.isend(block[-1].contiguous(), rank + 1, group=group))
        reqs.append(dist.irecv(below, rank + 1, group=group))
    for r in reqs:
        r.wait()
    return above, below


def padded_block(block, above, below):
    """Stack the halo rows (or the mirrored second line at the room border) around the block: (rows+2, Nx)."""
    top = above if above is not None else block[1]
    bot = below if below is not None else block[-2]
    return torch.cat([top[None], block, bot[None]], dim=0)


def error_norm_allreduce(local_sumsq, local_max, device="cpu"):
    """Scalar reduction of one RK attempt: sum of squared scaled errors and max |y| (scipy rk.py:146-147)."""
    t = torch.tensor([float(local_sumsq)], dtype=torch.float64, device=device)
    m = torch.tensor([float(local_max)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return float(t.item()), float(m.item())
