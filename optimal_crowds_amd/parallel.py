"""Multi-GPU plumbing for the path (one process per GPU, torch.distributed; backend "nccl" = RCCL on ROCm).

ONE simulation on N GPUs as SURVEY 8e lays it out (what `bench.py --gpus N` measures for the multi-key / large-grid
workloads):
* keys first, then the HJB grid in row blocks: `key_row_plan` deals the keys to groups of ranks, `RowShard` is the
  geometry of one rank's block, `DistHaloComm` carries the two collectives of `oc_hjb_solve_rows` (include/oc_hip.h) --
  halo rows to the neighbours, all-gather of the error norm -- enqueued on the launch stream, no host synchronisation;
* replicated agents: `DistStepComm` carries the step's exchange of `oc_gcfm_step_sharded` -- one all-gather of the ranks'
  tile shares, plus a sum of the desired velocities when the floor fields are sharded;
* `dist_sharding()` is the `sharding=` argument that switches a whole `simulations.simulation` to this mode;
* `ThreadHaloComm` / `ThreadStepComm` / `thread_sharding` run several virtual ranks inside one process on one GPU (tests).
Over nccl the collectives move device tensors (RCCL over xGMI) on the launch stream; over gloo they are staged through
host memory.  The numerics stay in the HIP library; this module only moves tensors.
"""
import ctypes as C
import os
import threading

import torch
import torch.distributed as dist

from ._lib import ALLGATHER_CB, ALLREDUCE_CB, HALO_CB, OcHjbShard, OcStepComm


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


ROW_MIN_CELLS = 1 << 22        # a row block below ~4 M cells is a chain of latency-bound launches: nothing to win by splitting


def key_row_plan(n_keys, world, ncell=None, min_cells=None):
    """partition by key first (the solves of different keys are independent: zero communication), then split
    each key's grid into row blocks over the ranks of its group.  Returns, per key, the tuple of world ranks that solve
    it, in row-block order.  world >= n_keys: contiguous groups of world // n_keys ranks (the first world % n_keys
    groups get one more): 2 keys on 4 GPUs = 2 groups of 2.  world < n_keys: key k goes whole to rank k % world.
    ncell (cells of one key's grid): a group only uses as many of its ranks as leave every row block at least
    `min_cells` cells (default ROW_MIN_CELLS, or the environment's OC_ROW_MIN_CELLS) -- "only where grid / agent counts make
    it natural": on one MI355X a 2048^2 solve is ~100 launches of 18 us on a few dozen CUs, splitting it only adds two
    collectives per attempt, so C4's keys stay whole at any N (ranks without a key still take their share of the step) and
    C5's 4096^2 grid goes 4-way.  (A row-block solve keeps the exact active set per rank since round 4; `bench.py --gpus N`
    measures the row-split variant of a multi-key workload beside the keys-first line.)  ncell=None: every rank of a group gets
    a block."""
    if world >= n_keys:
        base, rem = divmod(world, n_keys)
        plan, r0 = [], 0
        for k in range(n_keys):
            n = base + (1 if k < rem else 0)
            use = n
            if ncell is not None:
                mc = int(min_cells if min_cells is not None else os.environ.get("OC_ROW_MIN_CELLS", ROW_MIN_CELLS))
                use = max(1, min(n, int(ncell) // max(mc, 1)))
            plan.append(tuple(range(r0, r0 + use)))
            r0 += n
        return plan
    return [(k % world,) for k in range(n_keys)]


def row_blocks(Ny, world):
    """Row-block decomposition of the (Ny, Nx) grid: contiguous blocks, sizes differ by at most one row."""
    base, rem = divmod(Ny, world)
    out, r0 = [], 0
    for r in range(world):
        r1 = r0 + base + (1 if r < rem else 0)
        out.append((r0, r1))
        r0 = r1
    return out


# ---------------------------------------------------------------------------------------------- row-block HJB solve
class RowShard:
    """One rank's row block of the (Ny, Nx) grid plus `halo` rows of each neighbour (oc_hip.h: oc_hjb_shard).
    halo >= 7 for the solve alone (one RK step applies the stencil 6 times, and the interpolant's coefficients are formed
    from the LOCAL K6 one row beyond the block); the direction field is exact on owned rows +- (halo - 7) = field_margin,
    and a sharded LOOKUP needs field_margin >= 1, i.e. halo >= 8 (Engine.enable_sharding, dist_sharding enforce it)."""

    def __init__(self, Ny, rank, world, halo=8):
        r0, r1 = row_blocks(Ny, world)[rank]
        if world > 1 and (halo < 7 or Ny // world < halo):
            raise ValueError("row blocks of %d rows cannot carry a halo of %d rows" % (Ny // world, halo))
        self.Ny, self.rank, self.world, self.halo = Ny, rank, world, halo
        self.has_up, self.has_dn = rank > 0, rank < world - 1
        self.g0, self.g1 = r0, r1                                   # owned global rows [g0, g1)
        self.row0 = r0 - (halo if self.has_up else 0)               # global row of local row 0
        self.own_lo = halo if self.has_up else 0
        self.own_hi = self.own_lo + (r1 - r0)
        self.Ny_local = self.own_hi + (halo if self.has_dn else 0)
        self.field_margin = halo - 7                                # rows beyond the block where the field is exact

    def rows(self):
        """slice of the global grid rows held locally (owned + halo)"""
        return slice(self.row0, self.row0 + self.Ny_local)


class _HaloComm:
    """Staging buffers + the two host callbacks of oc_hjb_solve_rows.  Subclasses implement the collectives."""

    def __init__(self, shard, Nx, tdtype, device):
        self.shard, self.device = shard, torch.device(device)
        mk = lambda: torch.zeros((2, shard.halo, Nx), dtype=tdtype, device=self.device)
        self.send_up, self.recv_up, self.send_dn, self.recv_dn = mk(), mk(), mk(), mk()
        self.red_local = torch.zeros(2, dtype=torch.float64, device=self.device)
        self.red_all = torch.zeros((shard.world, 2), dtype=torch.float64, device=self.device)
        self.error = None
        self.n_allgather = self.n_exchange = 0
        self.comm_slot = -1             # >= 0: the library issues both collectives itself over its RCCL communicator
        self._cb_red = HALO_CB(self._red)
        self._cb_halo = HALO_CB(self._halo)

    def _red(self, user):
        try:
            self.allgather_red()
            self.n_allgather += 1
            return 0
        except BaseException as e:        # an exception must not unwind through the C frames
            self.error = e
            return -1

    def _halo(self, user):
        try:
            self.exchange_halo()
            self.n_exchange += 1
            return 0
        except BaseException as e:
            self.error = e
            return -1

    def struct(self):
        sh, st = self.shard, OcHjbShard()
        st.Ny_local, st.Ny_global = sh.Ny_local, sh.Ny
        st.own_lo, st.own_hi, st.halo = sh.own_lo, sh.own_hi, sh.halo
        st.has_up, st.has_dn, st.world = int(sh.has_up), int(sh.has_dn), sh.world
        st.send_up, st.recv_up = self.send_up.data_ptr(), self.recv_up.data_ptr()
        st.send_dn, st.recv_dn = self.send_dn.data_ptr(), self.recv_dn.data_ptr()
        st.red_local, st.red_all = self.red_local.data_ptr(), self.red_all.data_ptr()
        st.allgather_red, st.exchange_halo = self._cb_red, self._cb_halo
        st.user = None
        st.comm_slot, st.reserved = int(self.comm_slot), 0
        return st


class DistHaloComm(_HaloComm):
    """The collectives over torch.distributed.  `ranks`: global ranks of the group's members in block order (default:
    the whole world).  Backend nccl (= RCCL): device tensors, enqueued on the current stream, neighbour messages as one
    batched send/recv pair per side over xGMI; backend gloo: staged through host memory (tests)."""

    def __init__(self, shard, Nx, tdtype, device, group=None, ranks=None):
        super().__init__(shard, Nx, tdtype, device)
        self.group = group
        self.ranks = list(range(dist.get_world_size())) if ranks is None else list(ranks)
        assert len(self.ranks) == shard.world
        self.host_staged = dist.get_backend(group) != "nccl"
        self.halo_bytes = self.send_up.numel() * self.send_up.element_size()      # one message to one neighbour
        if self.host_staged:      # gloo: persistent host mirrors of the staging buffers (no allocation per exchange)
            mk = lambda t: torch.empty(t.shape, dtype=t.dtype, device="cpu")
            self._h = {n: mk(getattr(self, n)) for n in ("send_up", "recv_up", "send_dn", "recv_dn")}
            self._h_red = torch.empty(2, dtype=torch.float64)
            self._h_all = [torch.empty(2, dtype=torch.float64) for _ in range(shard.world)]

    def allgather_red(self):
        if self.host_staged:
            self._h_red.copy_(self.red_local)
            dist.all_gather(self._h_all, self._h_red, group=self.group)
            self.red_all.copy_(torch.stack(self._h_all))
        else:       # RCCL: 16 bytes per rank, enqueued on the launch stream -- no host synchronisation
            dist.all_gather_into_tensor(self.red_all, self.red_local, group=self.group)

    def exchange_halo(self):
        """send_up -> the upper neighbour's recv_dn, send_dn -> the lower neighbour's recv_up: one batched pair of
        point-to-point messages per side (xGMI is point-to-point: no ring), straight between the staging buffers."""
        sh = self.shard
        ops = []
        if self.host_staged:
            for n in ("send_up", "send_dn"):
                self._h[n].copy_(getattr(self, n))
        buf = (lambda n: self._h[n]) if self.host_staged else (lambda n: getattr(self, n))
        if sh.has_up:
            peer = self.ranks[sh.rank - 1]
            ops += [dist.P2POp(dist.isend, buf("send_up"), peer, group=self.group),
                    dist.P2POp(dist.irecv, buf("recv_up"), peer, group=self.group)]
        if sh.has_dn:
            peer = self.ranks[sh.rank + 1]
            ops += [dist.P2POp(dist.isend, buf("send_dn"), peer, group=self.group),
                    dist.P2POp(dist.irecv, buf("recv_dn"), peer, group=self.group)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()          # (nccl: waits on the stream, not on the host)
        if self.host_staged:
            for n in ("recv_up", "recv_dn"):
                getattr(self, n).copy_(self._h[n])


class ThreadWorld:
    """Shared state of `world` virtual ranks living in threads of one process (one GPU): a barrier and the comms."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.comms = [None] * world

    def sync(self):
        torch.cuda.synchronize()
        self.barrier.wait()


class ThreadHaloComm(_HaloComm):
    def __init__(self, shard, Nx, tdtype, device, tworld):
        super().__init__(shard, Nx, tdtype, device)
        self.tw = tworld
        tworld.comms[shard.rank] = self

    def allgather_red(self):
        self.tw.sync()                                  # every rank's red_local is final
        for r, c in enumerate(self.tw.comms):
            self.red_all[r].copy_(c.red_local)
        self.tw.sync()                                  # nobody overwrites red_local before it was read

    def exchange_halo(self):
        sh = self.shard
        self.tw.sync()
        if sh.has_up:
            self.recv_up.copy_(self.tw.comms[sh.rank - 1].send_dn)
        if sh.has_dn:
            self.recv_dn.copy_(self.tw.comms[sh.rank + 1].send_up)
        self.tw.sync()


# ---------------------------------------------------------------------------------------------- sharded agent step
class _StepComm:
    """The collectives of oc_gcfm_step_sharded (oc_hip.h: oc_step_comm), both in place on the exchange buffer and enqueued
    on the current stream: ONE all-gather per step of the ranks' blocks (each rank's share of the tiles: pair sums, wall
    forces and -- when every rank holds every floor field -- desired velocities: 32 or 48 bytes per agent in total), and,
    only when the floor fields are sharded, a SUM of the desired velocities (16 bytes per agent, one non-zero contribution
    per element).  Subclasses implement allgather(per) and allreduce(view)."""

    def __init__(self, rank, world):
        self.rank, self.world = int(rank), int(world)
        self.buf = None
        self.error = None
        self.n_allgather = self.n_allreduce = 0
        self.bytes_allgather = self.bytes_allreduce = 0         # payload of the last step (all ranks' blocks / the sum)
        self.comm_slot = -1             # >= 0: the library issues both collectives itself over its RCCL communicator
        self._cb_ag = ALLGATHER_CB(self._ag)
        self._cb_ar = ALLREDUCE_CB(self._ar)

    def bind(self, buf):
        self.buf = buf

    def struct(self):
        st = OcStepComm()
        st.rank, st.world = self.rank, self.world
        st.xchg = self.buf.data_ptr()
        st.allgather, st.allreduce_sum, st.user = self._cb_ag, self._cb_ar, None
        st.comm_slot, st.reserved = int(self.comm_slot), 0
        return st

    def _view(self, ptr, count):
        off = (ptr - self.buf.data_ptr()) // 8
        assert self.buf is not None and 0 <= off and off + count <= self.buf.numel()
        return self.buf[off:off + count]

    def _ag(self, user, ptr, per):
        try:
            assert ptr == self.buf.data_ptr() and per * self.world <= self.buf.numel()
            self.allgather(int(per))
            self.n_allgather += 1
            self.bytes_allgather = 8 * int(per) * self.world
            return 0
        except BaseException as e:        # an exception must not unwind through the C frames
            self.error = e
            return -1

    def _ar(self, user, ptr, count):
        try:
            self.allreduce(self._view(ptr, int(count)))
            self.n_allreduce += 1
            self.bytes_allreduce = 8 * int(count)
            return 0
        except BaseException as e:
            self.error = e
            return -1


class DistStepComm(_StepComm):
    """torch.distributed: backend nccl (= RCCL) gathers / sums the device tensor over xGMI on the current stream, in place
    (C4: 100 000 agents -> 3.2 MB all-gather + 1.6 MB sum with sharded fields, 4.8 MB all-gather without); gloo stages
    through host memory (tests)."""

    def __init__(self, group=None):
        super().__init__(dist.get_rank(group), dist.get_world_size(group))
        self.group = group
        self.host_staged = dist.get_backend(group) != "nccl"

    def allgather(self, per):
        mine = self.buf[self.rank * per:(self.rank + 1) * per]
        if self.host_staged:
            parts = [torch.empty(per, dtype=self.buf.dtype) for _ in range(self.world)]
            dist.all_gather(parts, mine.cpu(), group=self.group)
            self.buf[:per * self.world].copy_(torch.cat(parts))
        else:       # (in place: rank r's input IS block r of the output)
            dist.all_gather_into_tensor(self.buf[:per * self.world], mine, group=self.group)

    def allreduce(self, view):
        if self.host_staged:
            hbuf = view.cpu()
            dist.all_reduce(hbuf, op=dist.ReduceOp.SUM, group=self.group)
            view.copy_(hbuf)
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)


class ThreadStepComm(_StepComm):
    """Virtual ranks in threads of one process on one GPU (tests): device copies between the ranks' buffers."""

    def __init__(self, rank, tworld):
        super().__init__(rank, tworld.world)
        self.tw = tworld
        tworld.comms[rank] = self

    def allgather(self, per):
        self.tw.sync()                                  # every rank's block is written
        for r, c in enumerate(self.tw.comms):
            if r != self.rank:
                self.buf[r * per:(r + 1) * per].copy_(c.buf[r * per:(r + 1) * per])
        self.tw.sync()                                  # nobody rewrites its block before it was read

    def allreduce(self, view):
        self._view_now = view
        self.tw.sync()                                  # every rank's part is written
        if self.rank == 0:
            total = torch.zeros_like(view)
            for c in self.tw.comms:
                total += c._view_now
            self._total = total
        self.tw.sync()
        view.copy_(self.tw.comms[0]._total)
        self.tw.sync()                                  # rank 0 keeps _total until everybody has copied it


# ---------------------------------------------------------------------------------------------- one call for a façade
def native_comm(eng, slot, group=None, ranks=None):
    """An RCCL communicator owned by the LIBRARY (oc_comm_create, csrc/comm.hip) for the ranks of `group` (None: the whole
    world): the group's first rank draws the unique id, torch.distributed only carries its 128 bytes to the others, and
    from then on the collectives of that group leave from C on the launch stream -- no Python on the launch path.
    Collective: every member calls it (and every rank of the world takes part in the broadcast of a sub-group's id only
    if it is a member: torch.distributed sub-group semantics).  Returns the slot."""
    import ctypes as C_
    ranks = list(range(dist.get_world_size())) if ranks is None else list(ranks)
    me = dist.get_rank()
    assert me in ranks
    idb = (C_.c_ubyte * 128)()
    if me == ranks[0]:
        from ._lib import check
        check(eng.lib.oc_comm_unique_id(idb))
    box = [bytes(idb)]
    if len(ranks) > 1:
        dist.broadcast_object_list(box, src=ranks[0], group=group)
    from ._lib import check
    buf = (C_.c_ubyte * 128).from_buffer_copy(box[0])
    rc = eng.lib.oc_comm_create(eng.h, int(slot), buf, ranks.index(me), len(ranks))
    if rc != 0 and len(ranks) > 1:
        # ncclCommInitRank is a collective: the other members are blocked inside it and no agreement can be reached any more.
        # End the job loudly (the launcher tears the other ranks down) instead of raising here and leaving them waiting.
        import os
        import sys
        sys.stderr.write("optimal_crowds_amd.parallel: rank %d could not create the library's RCCL communicator for ranks %s "
                         "(error %d: %s) -- aborting the job\n"
                         % (me, ranks, rc, eng.lib.oc_last_error().decode(errors="replace")))
        sys.stderr.flush()
        os._exit(70)
    check(rc)
    return int(slot)


def _rccl_usable(eng):
    """can liboc_hip bind librccl in this process? (oc_comm_available: dlopen + symbols -- no unique id is drawn: each
    ncclGetUniqueId starts a bootstrap listener that is never joined)"""
    try:
        return eng.lib.oc_comm_available() == 0
    except Exception:       # noqa: BLE001
        return False


_SUBGROUPS = {}         # rank tuple -> process group: a simulation built twice in one process reuses its sub-groups


def _subgroup(ranks):
    """the process group of `ranks` (None: the whole world); created once per process and rank set, since communicators
    are never freed behind a simulation's back (every rank must call this with the same rank sets in the same order)"""
    ranks = tuple(ranks)
    if len(ranks) <= 1 or len(ranks) >= dist.get_world_size():
        return None
    if ranks not in _SUBGROUPS:
        _SUBGROUPS[ranks] = dist.new_group(list(ranks))
    return _SUBGROUPS[ranks]


def dist_sharding(halo=8, by_key=True, native=None, min_cells=None):
    """`sharding=` argument of simulations.simulation for a torch.distributed job (one process per GPU): ONE simulation
    on all ranks.  HJB: keys first, then row blocks where a block keeps at least min_cells cells (key_row_plan;
    by_key=False: every key over all ranks) with `halo`
    rows -- at least 8 here: 7 for the solve itself (RowShard) and one more so that the floor field is exact on the row
    beyond the block, which a lookup's second cell may read -- each key's collectives on its own sub-group (nccl = RCCL over
    xGMI, or gloo); agents replicated, the step's exchange over the whole world (DistStepComm).
    native: the collectives are issued by the library itself over RCCL communicators it owns (native_comm: slot 0 = the
    world, slot 1 + k = key k's rank group) instead of through callbacks into torch.distributed.  None = on for an nccl
    job unless the environment says OC_RCCL_NATIVE=0: with callbacks a step pays two trips through the interpreter (one-rank
    RCCL group at C4: 0.42 ms per step against 0.28, profiles/r03_rccl_*one_rank.json); if a communicator cannot be created
    the callbacks take over (a line on stderr says so).
    Returns engine -> ({key: halo comm or None}, step comm)."""
    if native is None:
        native = os.environ.get("OC_RCCL_NATIVE", "1") != "0"
    if halo < 8:
        raise ValueError("sharded lookups need halo >= 8 (the field is exact halo - 7 rows beyond a block)")

    def make(eng):
        world, rank = dist.get_world_size(), dist.get_rank()
        plan = (key_row_plan(eng.n_keys, world, ncell=eng.Nx * eng.Ny, min_cells=min_cells) if by_key
                else [tuple(range(world))] * eng.n_keys)
        use_native = bool(native and dist.get_backend() == "nccl")
        if use_native:
            # every rank must take the same path: try the library's RCCL binding HERE, on all ranks, and agree on the outcome
            # before any communicator is created
            ok = torch.tensor([1 if _rccl_usable(eng) else 0], dtype=torch.int32, device=eng.device)
            if world > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if rank == 0:
                    import sys
                    print("optimal_crowds_amd.parallel: librccl could not be bound by liboc_hip on every rank -- collectives "
                          "through torch.distributed callbacks", file=sys.stderr)
                use_native = False
        comms = {}
        # a single key that is not worth splitting: EVERY rank solves it whole (the same kernels on the same inputs: the same
        # bits) instead of rank 0 alone with the others waiting -- every rank then holds the whole floor field, lookups stay
        # local and the step needs its all-gather only.  (It also keeps the ranks' view of "are the fields sharded" the same:
        # oc_gcfm_step_sharded decides it from the fields registered on the rank.)
        replicate = by_key and eng.n_keys == 1 and len(plan[0]) == 1 and world > 1
        eng.key_replicated = replicate
        for k, ranks in enumerate(plan):
            # (every rank asks for every group, member or not, in the same order: torch.distributed requires it)
            grp = _subgroup(ranks) if len(ranks) > 1 else None
            if replicate:
                comms[k] = DistHaloComm(RowShard(eng.Ny, 0, 1, halo), eng.Nx, eng.tdtype, eng.device, group=None, ranks=(rank,))
            elif rank in ranks:
                sh = RowShard(eng.Ny, ranks.index(rank), len(ranks), halo)
                comms[k] = DistHaloComm(sh, eng.Nx, eng.tdtype, eng.device, group=grp, ranks=ranks)
                if use_native and k + 1 < 16:
                    comms[k].comm_slot = native_comm(eng, 1 + k, group=grp, ranks=ranks)
            else:
                comms[k] = None
        eng.key_plan = plan
        step = DistStepComm(None)
        if use_native:
            step.comm_slot = native_comm(eng, 0)
        eng.rccl_native = use_native
        return comms, step
    return make


def thread_sharding(rank, tw_solve, tw_step, halo=8, plan=None):
    """The same for virtual ranks in threads of one process (tests): tw_step = ThreadWorld of all ranks (the step's
    exchange); tw_solve = ThreadWorld of all ranks (every key in row blocks over all of them), or -- with `plan`
    (key_row_plan) -- a list with one ThreadWorld per key, sized to the key's rank group."""
    if halo < 8:
        raise ValueError("sharded lookups need halo >= 8 (the field is exact halo - 7 rows beyond a block)")

    def make(eng):
        if plan is None:
            sh = RowShard(eng.Ny, rank, tw_solve.world, halo)
            return (ThreadHaloComm(sh, eng.Nx, eng.tdtype, eng.device, tw_solve), ThreadStepComm(rank, tw_step))
        comms = {}
        for k, ranks in enumerate(plan):
            if rank in ranks:
                sh = RowShard(eng.Ny, ranks.index(rank), len(ranks), halo)
                comms[k] = ThreadHaloComm(sh, eng.Nx, eng.tdtype, eng.device, tw_solve[k])
            else:
                comms[k] = None
        eng.key_plan = plan
        return comms, ThreadStepComm(rank, tw_step)
    return make
