"""RCCL inside the library (include/oc_hip.h: oc_comm_unique_id / oc_comm_create / oc_comm_destroy, csrc/comm.hip): with
`comm_slot >= 0` in oc_step_comm / oc_hjb_shard the collectives of the sharded step and of the row-block solve are issued
by the library itself on the launch stream -- no host callback.  A one-GPU box can exercise ONE rank of it: a
communicator of world size 1, through which the very same calls (ncclAllGather in place, ncclAllReduce, the grouped
send/recv of the halo exchange degenerate to nothing) run as they do with N ranks; results must equal the single-GPU
entry points bit for bit (a one-rank exchange moves nothing)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from optimal_crowds_amd import parallel                # noqa: E402
from test_hjb_rows import _setup                        # noqa: E402


def _new_engine(cfg, room, Nx, Ny, X0, Y0, V, dtype):
    from optimal_crowds_amd.engine import Engine
    eng = Engine(cfg, room["room_length"], room["room_height"], Nx, Ny, X0, Y0, 1, 4096, dtype=dtype)
    eng.set_potential(0, V, np.zeros((0, 4)))
    return eng


def _make_comm(eng, slot):
    from optimal_crowds_amd._lib import check
    idb = (C.c_ubyte * 128)()
    check(eng.lib.oc_comm_unique_id(idb))
    assert any(idb)
    check(eng.lib.oc_comm_create(eng.h, slot, idb, 0, 1))
    return slot


class _NoCallbackStepComm(parallel._StepComm):
    """every collective must come from the library: a callback that is reached fails the test"""

    def allgather(self, per):
        raise AssertionError("the allgather callback was called although comm_slot >= 0")

    def allreduce(self, view):
        raise AssertionError("the allreduce callback was called although comm_slot >= 0")


class _NoCallbackHaloComm(parallel._HaloComm):
    def allgather_red(self):
        raise AssertionError("the allgather_red callback was called although comm_slot >= 0")

    def exchange_halo(self):
        raise AssertionError("the exchange_halo callback was called although comm_slot >= 0")


def test_one_rank_communicator_step_and_row_solve():
    import torch
    from optimal_crowds_amd._lib import OcError, check
    setup = _setup("gold_a", 0.05)
    cfg, room, Nx, Ny, X0, Y0, V = setup
    T, nt = 3.0, 150
    rng = np.random.RandomState(2)
    n = 700
    x, y = rng.uniform(0.3, room["room_length"] - 0.4, n), rng.uniform(0.3, room["room_height"] - 0.4, n)
    vx, vy = rng.normal(0, 0.4, n), rng.normal(0, 0.4, n)
    vdes = rng.normal(1.34, 0.26, n)
    noise = rng.normal(size=(4, n, 2))
    out = {}
    for native in (False, True):
        eng = _new_engine(*setup, "float32")
        eng.alloc_field(0, nt - 1)
        if native:
            slot = _make_comm(eng, 3)
            with pytest.raises(OcError):                # the slot is taken
                check(eng.lib.oc_comm_create(eng.h, 3, (C.c_ubyte * 128)(), 0, 1))
            sh = parallel.RowShard(Ny, 0, 1, 8)
            hc = _NoCallbackHaloComm(sh, Nx, eng.tdtype, eng.device)
            hc.comm_slot = slot
            eng.enable_sharding({0: hc}, None)
            # (a world of one row block: solve() takes the whole-grid path; the row-block entry point itself:)
            st, _, phi = eng.solve_rows(0, T, nt, hc, keep=[1, 75, 149], _own=True)
            eng.fields[0].nt = nt
            eng._register_field(0)
            assert hc.error is None and hc.n_allgather == 0 and hc.n_exchange == 0
        else:
            st, phi = eng.solve(0, T, nt, keep=[1, 75, 149])
        eng.set_agents(x, y, vx, vy, vdes, np.zeros(n, dtype=np.int32))
        steps = []
        if native:
            sc = _NoCallbackStepComm(0, 1)
            sc.comm_slot = slot
            for s in range(4):
                steps.append(eng.step_sharded(s, sc, noise=noise[s]))
            assert sc.error is None and sc.n_allgather == 0 and sc.n_allreduce == 0
            check(eng.lib.oc_comm_destroy(eng.h, slot))
            check(eng.lib.oc_comm_destroy(eng.h, slot))         # (idempotent)
        else:
            for s in range(4):
                steps.append(eng.step(s, noise=noise[s]))
        stt = eng.get_state()
        out[native] = (st, phi.cpu(), eng.fields[0][0].cpu(), stt, steps)
        eng.close()
    (st_a, phi_a, vx_a, s_a, steps_a), (st_b, phi_b, vx_b, s_b, steps_b) = out[False], out[True]
    assert st_a["n_accept"] == st_b["n_accept"] and st_a["n_attempt"] == st_b["n_attempt"]
    # the row-block solve computes every tile and sums its error norm over ONE rank's partials: the same arithmetic per
    # cell; decisions may differ in the last bits of the norm only (tolerances of tests/test_hjb_rows.py)
    assert torch.allclose(phi_a, phi_b, rtol=1e-5, atol=0) and float((vx_a - vx_b).abs().max()) <= 1e-4
    assert steps_a == steps_b
    for f in ("x", "y", "vx", "vy"):
        assert np.abs(s_a[f] - s_b[f]).max() <= 1e-9, f
    assert np.array_equal(s_a["status"], s_b["status"])
