"""ctypes binding of liboc_hip.so (C-ABI: include/oc_hip.h).

There is no CPU fallback: if the shared library is missing, cannot be loaded, or no MI355X is
present, the first use raises.  `build()` compiles the library in-tree with hipcc for gfx950.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OC_HIP_LIB") or os.path.join(HERE, "liboc_hip.so")     # (OC_HIP_LIB: A/B builds, tools/)
CSRC = os.path.join(HERE, "csrc")

OC_ABI_VERSION = 3       # include/oc_hip.h: bumped whenever a signature or a struct of the C-ABI changes
OC_F32, OC_F64, OC_F64_F32 = 0, 1, 2
# opt-in corrections of reference quirks (oc_hip.h: oc_set_fix_flags); 'horizon' (Q9) is host logic in the facade
FIX_BITS = {"lookup": 1, "wall_force": 2, "pair_force": 4, "density_norm": 8}


class OcConfig(C.Structure):
    _fields_ = [
        ("Ny", C.c_int32), ("Nx", C.c_int32),
        ("dx", C.c_double), ("dy", C.c_double),
        ("room_length", C.c_double), ("room_height", C.c_double),
        ("dt", C.c_double),
        ("sigma", C.c_double), ("mu", C.c_double), ("g", C.c_double),
        ("lim", C.c_double), ("rtol", C.c_double), ("atol", C.c_double),
        ("a_min", C.c_double), ("tau_a", C.c_double), ("b_min", C.c_double), ("b_max", C.c_double),
        ("eta", C.c_double), ("eta_walls", C.c_double),
        ("repulsion_cutoff", C.c_double), ("v_max", C.c_double), ("relaxation", C.c_double),
        ("noise_amp", C.c_double), ("wall_search_radius", C.c_double),
        ("max_agents", C.c_int32), ("max_keys", C.c_int32),
    ]


class OcHjbStats(C.Structure):
    _fields_ = [
        ("nfev", C.c_int32), ("n_attempt", C.c_int32), ("n_accept", C.c_int32), ("n_rescale", C.c_int32),
        ("h_abs0", C.c_double), ("h_min", C.c_double), ("h_max", C.c_double),
        ("final_exponent", C.c_int32), ("n_slices_written", C.c_int32),
        ("seconds", C.c_double),
        ("persist_attempts", C.c_int32), ("persist_blocks", C.c_int32), ("persist_exit", C.c_int32), ("grid_resyncs", C.c_int32),
        ("lead_attempts", C.c_int32), ("reserved_", C.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


HALO_CB = C.CFUNCTYPE(C.c_int, C.c_void_p)     # int (*)(void* user)
ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)     # int (*)(void* user, double* buf, int64 n)
ALLGATHER_CB = ALLREDUCE_CB                                                 # (same shape: n = doubles per rank block)


class OcStepComm(C.Structure):
    """oc_hip.h: oc_step_comm -- the sharded step's exchange buffer and its two collectives"""
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("xchg", C.c_void_p),
                ("allgather", ALLGATHER_CB), ("allreduce_sum", ALLREDUCE_CB), ("user", C.c_void_p),
                ("comm_slot", C.c_int32), ("reserved", C.c_int32)]


class OcHjbShard(C.Structure):
    _fields_ = [
        ("Ny_local", C.c_int32), ("Ny_global", C.c_int32),
        ("own_lo", C.c_int32), ("own_hi", C.c_int32),
        ("halo", C.c_int32), ("has_up", C.c_int32), ("has_dn", C.c_int32), ("world", C.c_int32),
        ("send_up", C.c_void_p), ("recv_up", C.c_void_p), ("send_dn", C.c_void_p), ("recv_dn", C.c_void_p),
        ("red_local", C.c_void_p), ("red_all", C.c_void_p),
        ("allgather_red", HALO_CB), ("exchange_halo", HALO_CB),
        ("user", C.c_void_p),
        ("comm_slot", C.c_int32), ("reserved", C.c_int32),
    ]


class OcHjbRing(C.Structure):
    """oc_hip.h: oc_hjb_ring -- the interpolant ring of one key (device buffers owned by the caller)"""
    _fields_ = [("data", C.c_void_p), ("meta", C.c_void_p),
                ("slots", C.c_int32), ("nslice", C.c_int32), ("rows", C.c_int32), ("reserved", C.c_int32)]


class OcHjbRingJob(C.Structure):
    """oc_hip.h: oc_hjb_ring_job -- one of several ring solves oc_hjb_solve_ring_many runs at the same time"""
    _fields_ = [("h", C.c_void_p), ("coef_dtype", C.c_int32), ("dtype", C.c_int32), ("V", C.c_void_p), ("m", C.c_void_p),
                ("coef", C.c_void_p), ("T", C.c_double), ("nt", C.c_int32), ("rc", C.c_int32),
                ("ring", C.POINTER(OcHjbRing)), ("stream", C.c_void_p), ("stats", OcHjbStats), ("error", C.c_char * 256)]


OC_ERR_CAPACITY = -5


class OcAgents(C.Structure):
    _fields_ = [
        ("N", C.c_int32),
        ("x", C.c_void_p), ("y", C.c_void_p), ("vx", C.c_void_p), ("vy", C.c_void_p),
        ("v_des", C.c_void_p), ("status", C.c_void_p), ("key", C.c_void_p), ("agent_time", C.c_void_p),
    ]


# name -> (restype, argtypes): every symbol include/oc_hip.h declares
SIGNATURES = {
    "oc_abi_version": (C.c_int, []),
    "oc_last_error": (C.c_char_p, []),
    "oc_create": (C.c_int, [C.c_int, C.POINTER(OcConfig), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "oc_destroy": (C.c_int, [C.c_void_p]),
    "oc_hjb_make_coef": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "oc_hjb_rhs": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "oc_hjb_ctl_power": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "oc_hjb_solve": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                               C.POINTER(OcHjbStats), C.c_void_p]),
    "oc_hjb_solve_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.POINTER(OcHjbShard), C.c_void_p, C.c_int32, C.c_void_p,
                                    C.POINTER(OcHjbStats), C.c_void_p]),
    "oc_hjb_ring_slot_elems": (C.c_int64, [C.c_void_p, C.c_int32]),
    "oc_hjb_ring_meta_bytes": (C.c_int64, [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_int32]),
    "oc_hjb_ring_slots_hint": (C.c_int32, [C.c_void_p, C.c_double]),
    "oc_hjb_solve_ring": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_int32, C.POINTER(OcHjbRing),
                                    C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                    C.POINTER(OcHjbStats), C.c_void_p]),
    "oc_hjb_solve_ring_many": (C.c_int, [C.POINTER(OcHjbRingJob), C.c_int32]),
    "oc_hjb_solve_rows_ring": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_int32, C.POINTER(OcHjbRing),
                                         C.POINTER(OcHjbShard), C.c_void_p, C.c_int32, C.c_void_p,
                                         C.POINTER(OcHjbStats), C.c_void_p]),
    "oc_hjb_ring_slices": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(OcHjbRing), C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "oc_set_field_ring": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(OcHjbRing), C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32]),
    "oc_hjb_vels": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "oc_set_field": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "oc_set_field_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "oc_set_key_targets": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int32]),
    "oc_wallmap_build": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p]),
    "oc_field_lookup": (C.c_int, [C.c_void_p, C.POINTER(OcAgents), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "oc_wall_force": (C.c_int, [C.c_void_p, C.POINTER(OcAgents), C.c_void_p, C.c_void_p, C.c_void_p]),
    "oc_wall_last_index": (C.c_int, [C.c_void_p, C.POINTER(OcAgents), C.c_void_p, C.c_void_p]),
    "oc_pair_force": (C.c_int, [C.c_void_p, C.POINTER(OcAgents), C.c_void_p, C.c_void_p]),
    "oc_gcfm_step": (C.c_int, [C.c_void_p, C.POINTER(OcAgents), C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p,
                               C.c_void_p]),
    "oc_gcfm_step_sequential": (C.c_int, [C.c_void_p, C.POINTER(OcAgents), C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64,
                                          C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "oc_padded_agents": (C.c_int64, [C.c_int32]),
    "oc_xchg_doubles": (C.c_int64, [C.c_int32, C.c_int32]),
    "oc_gcfm_step_sharded": (C.c_int, [C.c_void_p, C.POINTER(OcAgents), C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p,
                                       C.POINTER(OcStepComm), C.c_void_p]),
    "oc_comm_available": (C.c_int, []),
    "oc_comm_unique_id": (C.c_int, [C.c_void_p]),
    "oc_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "oc_comm_destroy": (C.c_int, [C.c_void_p, C.c_int]),
    "oc_run_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_void_p]),
    "oc_set_resort_every": (C.c_int, [C.c_void_p, C.c_int32]),
    "oc_agents_changed": (C.c_int, [C.c_void_p]),
    "oc_agents_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "oc_density": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(OcAgents), C.c_double, C.c_void_p, C.c_double,
                             C.c_void_p, C.c_void_p]),
    "oc_density_rows": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(OcAgents), C.c_double, C.c_void_p, C.c_double,
                                 C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "oc_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "oc_kernel_stats": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double)]),
    "oc_reset_kernel_stats": (C.c_int, [C.c_void_p]),
    "oc_set_hjb_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "oc_set_tuning": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "oc_set_fix_flags": (C.c_int, [C.c_void_p, C.c_uint32]),
    "oc_rasterise_potential": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_void_p,
                                         C.c_void_p]),
}

_lib = None


class OcError(RuntimeError):
    pass


def build(verbose=False):
    """Compile liboc_hip.so in-tree (hipcc --offload-arch=gfx950).  Works without a GPU."""
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
        print(r.stderr)
    if r.returncode != 0:
        raise OcError("building liboc_hip.so failed:\n" + r.stderr[-4000:])
    return LIB_PATH


def load():
    """Load the shared library and declare every entry point; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OcError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(there is no CPU fallback)" % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime (libamdhip64.so in torch/lib).  Load it first so that liboc_hip.so binds
    # to the SAME runtime instance as the tensors it is handed: with /opt/rocm's copy loaded first the process ends up
    # with two runtimes and the second one reports "no ROCm-capable device".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.oc_abi_version() != OC_ABI_VERSION:
        raise OcError("liboc_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise OcError("liboc_hip: error %d: %s" % (rc, load().oc_last_error().decode(errors="replace")))
