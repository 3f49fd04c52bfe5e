// RCCL inside the library (oc_set_comm): communicators the library owns, so that the collectives of
// oc_hjb_solve_rows[_ring] and oc_gcfm_step_sharded are issued from C on the launch stream -- no host callback, no Python,
// no GIL on the launch path.  RCCL is bound at run time (dlopen of librccl.so.1: the instance PyTorch-ROCm has loaded, if
// any, else ROCm's), so the library neither links against it nor needs it on a single GPU.  xGMI is point-to-point: the halo
// rows go as one grouped send/recv pair per neighbour, never through a ring.
#include "oc_internal.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;

int rccl_load() {
    if (g_rccl.lib) return OC_OK;
    void* lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) {
        oc_set_error("oc_comm: librccl.so.1 not found (%s)", dlerror());
        return OC_ERR_STATE;
    }
    RcclApi a;
    a.lib = lib;
#define OC_SYM(field, name)                                           \
    a.field = (decltype(a.field))dlsym(lib, name);                    \
    if (!a.field) {                                                   \
        oc_set_error("oc_comm: %s missing from librccl", name);       \
        return OC_ERR_STATE;                                          \
    }
    OC_SYM(GetUniqueId, "ncclGetUniqueId")
    OC_SYM(CommInitRank, "ncclCommInitRank")
    OC_SYM(CommDestroy, "ncclCommDestroy")
    OC_SYM(AllGather, "ncclAllGather")
    OC_SYM(AllReduce, "ncclAllReduce")
    OC_SYM(Send, "ncclSend")
    OC_SYM(Recv, "ncclRecv")
    OC_SYM(GroupStart, "ncclGroupStart")
    OC_SYM(GroupEnd, "ncclGroupEnd")
    OC_SYM(GetErrorString, "ncclGetErrorString")
#undef OC_SYM
    g_rccl = a;
    return OC_OK;
}

#define OC_RCCL_CHECK(expr)                                                                          \
    do {                                                                                             \
        ncclResult_t _r = (expr);                                                                    \
        if (_r != ncclSuccess) {                                                                     \
            oc_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));   \
            return OC_ERR_HIP;                                                                       \
        }                                                                                            \
    } while (0)

oc_comm* slot_of(oc_handle* h, int slot) {
    if (!h || slot < 0 || slot >= OC_MAX_COMMS) return nullptr;
    return &h->comms[slot];
}

}   // namespace

extern "C" int oc_comm_available(void) { return rccl_load(); }

extern "C" int oc_comm_unique_id(void* id_out) {
    OC_REQUIRE(id_out, "null argument");
    static_assert(OC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "oc_hip.h: OC_COMM_ID_BYTES");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    OC_RCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return OC_OK;
}

extern "C" int oc_comm_create(oc_handle* h, int slot, const void* id, int rank, int world) {
    oc_comm* c = slot_of(h, slot);
    OC_REQUIRE(c != nullptr && id != nullptr, "bad communicator slot / null id");
    OC_REQUIRE(world >= 1 && world <= OC_MAX_WORLD && rank >= 0 && rank < world, "bad rank / world");
    OC_REQUIRE(c->comm == nullptr, "communicator slot in use (oc_comm_destroy first)");
    int rc = rccl_load();
    if (rc) return rc;
    OC_HIP_CHECK(hipSetDevice(h->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    OC_RCCL_CHECK(g_rccl.CommInitRank(&comm, world, uid, rank));
    c->comm = comm;
    c->rank = rank;
    c->world = world;
    return OC_OK;
}

extern "C" int oc_comm_destroy(oc_handle* h, int slot) {
    oc_comm* c = slot_of(h, slot);
    OC_REQUIRE(c != nullptr, "bad communicator slot");
    if (c->comm) {
        OC_HIP_CHECK(hipSetDevice(h->device));
        (void)hipDeviceSynchronize();
        OC_RCCL_CHECK(g_rccl.CommDestroy((ncclComm_t)c->comm));
        c->comm = nullptr;
        c->rank = c->world = 0;
    }
    return OC_OK;
}

// ---- the collectives the solvers issue (internal; hjb.hip / gcfm.hip) --------------------------------------------------
// in place: block r (count doubles at buf + r * count) is rank r's
int oc_comm_allgather_f64(oc_handle* h, int slot, double* buf, int64_t count, hipStream_t s) {
    oc_comm* c = slot_of(h, slot);
    OC_REQUIRE(c != nullptr && c->comm != nullptr, "no communicator in this slot (oc_comm_create)");
    OC_RCCL_CHECK(g_rccl.AllGather(buf + (int64_t)c->rank * count, buf, (size_t)count, ncclDouble, (ncclComm_t)c->comm, s));
    return OC_OK;
}

// out[r * count ..] = rank r's `in` (count doubles): the error norm's (sum, max) pairs of a row-block solve
int oc_comm_allgather_f64_oop(oc_handle* h, int slot, const double* in, double* out, int64_t count, hipStream_t s) {
    oc_comm* c = slot_of(h, slot);
    OC_REQUIRE(c != nullptr && c->comm != nullptr, "no communicator in this slot (oc_comm_create)");
    OC_RCCL_CHECK(g_rccl.AllGather(in, out, (size_t)count, ncclDouble, (ncclComm_t)c->comm, s));
    return OC_OK;
}

int oc_comm_allreduce_sum_f64(oc_handle* h, int slot, double* buf, int64_t count, hipStream_t s) {
    oc_comm* c = slot_of(h, slot);
    OC_REQUIRE(c != nullptr && c->comm != nullptr, "no communicator in this slot (oc_comm_create)");
    OC_RCCL_CHECK(g_rccl.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)c->comm, s));
    return OC_OK;
}

// halo rows of a row-block solve: send_up -> rank - 1's recv_dn, send_dn -> rank + 1's recv_up, `bytes` each way: one
// grouped send/recv pair per neighbour, straight between the staging buffers
int oc_comm_exchange_halo(oc_handle* h, int slot, const void* send_up, void* recv_up, const void* send_dn, void* recv_dn,
                          size_t bytes, int has_up, int has_dn, hipStream_t s) {
    oc_comm* c = slot_of(h, slot);
    OC_REQUIRE(c != nullptr && c->comm != nullptr, "no communicator in this slot (oc_comm_create)");
    OC_REQUIRE((!has_up || c->rank > 0) && (!has_dn || c->rank < c->world - 1), "neighbour outside the communicator");
    ncclComm_t comm = (ncclComm_t)c->comm;
    OC_RCCL_CHECK(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    if (has_up) {
        if (r == ncclSuccess) r = g_rccl.Send(send_up, bytes, ncclChar, c->rank - 1, comm, s);
        if (r == ncclSuccess) r = g_rccl.Recv(recv_up, bytes, ncclChar, c->rank - 1, comm, s);
    }
    if (has_dn) {
        if (r == ncclSuccess) r = g_rccl.Send(send_dn, bytes, ncclChar, c->rank + 1, comm, s);
        if (r == ncclSuccess) r = g_rccl.Recv(recv_dn, bytes, ncclChar, c->rank + 1, comm, s);
    }
    const ncclResult_t e = g_rccl.GroupEnd();
    OC_RCCL_CHECK(r);
    OC_RCCL_CHECK(e);
    return OC_OK;
}

void oc_comm_destroy_all(oc_handle* h) {
    for (int q = 0; q < OC_MAX_COMMS; ++q)
        if (h->comms[q].comm && g_rccl.CommDestroy) {
            (void)g_rccl.CommDestroy((ncclComm_t)h->comms[q].comm);
            h->comms[q].comm = nullptr;
        }
}
