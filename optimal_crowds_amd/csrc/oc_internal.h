// Internal declarations shared by the translation units of liboc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>
#include <map>

#include "../../include/oc_hip.h"

void oc_set_error(const char* fmt, ...);

#define OC_HIP_CHECK(expr)                                                                      \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            oc_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));  \
            return OC_ERR_HIP;                                                                  \
        }                                                                                       \
    } while (0)

#define OC_REQUIRE(cond, msg)                                              \
    do {                                                                   \
        if (!(cond)) {                                                     \
            oc_set_error("%s:%d: %s (%s)", __FILE__, __LINE__, msg, #cond); \
            return OC_ERR_ARG;                                             \
        }                                                                  \
    } while (0)

// ---- the interpolant ring (hjb.hip writes it, gcfm.hip's lookup and the slice writer read it) -------------------------
// Instead of every slice of the direction field, a solve can keep what the dense output of each ACCEPTED step is made of
// (scipy rk.py:178-180, 560-575): y_old, K0 = Q0 and Q1 .. Q3 -- five values per cell and step instead of two per cell and
// t_eval point (C4: 93 steps x 5 arrays against 999 slices x 2) -- and the direction (optimals.py:168-186) is evaluated
// where it is looked up, with the same FMA chain as the slice writer k_out.
//   data   per slot (= accepted step) five planes of rows x Nx values: y | K0 | Q1 | Q2 | Q3 (OC_RING_PLANAR below: every store
//          of the stage kernel is a contiguous 16 B per lane -- whole lines per wavefront row, streamed past the caches; a
//          lookup cell's five values come from five lines, and there are few lookups).  Only the stage tiles that were
//          COMPUTED in the accepted attempt are written (exact active set); the others read as (u, 0, 0, 0, 0).  The active
//          set only grows during a solve (a tile that was computed once is computed in every later attempt), so ONE word per
//          tile says it all: first[t] = the first slot in which tile t was computed (INT_MAX-like: never) -- trivial in
//          slot s <=> s < first[t].
//   meta   OcRingHead | OcRingSlot<T> x slots | OcRingSlice<T> x nslice | int32 first[stage tiles]
#define OC_RING_VALS 5
// Layout of a slot's values.  OC_RING_PLANAR 1: five planes of rows x Nx values (y | K0 | Q1 | Q2 | Q3) -- every store of the
// stage kernel is a contiguous 16 B per lane, whole lines per wavefront row; a lookup cell's five values come from five lines.
// 0: blocks of 20 values per four cells (y[4], K0[4], Q1[4], Q2[4], Q3[4]) -- one or two lines per lookup cell, but each
// store instruction of the stage kernel writes 16-byte pieces at an 80-byte stride.
#ifndef OC_RING_PLANAR
#define OC_RING_PLANAR 1
#endif
// offset of value a (0 .. 4) of cell idx (= local row * Nx + column) in a slot of n = rows * Nx cells
__host__ __device__ inline size_t oc_ring_off(size_t idx, int a, size_t n) {
#if OC_RING_PLANAR
    return (size_t)a * n + idx;
#else
    (void)n;
    return (idx >> 2) * (4 * OC_RING_VALS) + 4 * (size_t)a + (idx & 3);
#endif
}
// stage tile of the product path: k_stageV<T, 6, 1, true, OC_ST_RWV, OC_ST_RH> works on a region of 4 OC_ST_RWV x OC_ST_RH
// cells (1024 threads x 4 cells) of which the inner (4 OC_ST_RWV - 16) x (OC_ST_RH - 12) are the tile: 32 x 32 -> 112 x 20
// (a 64 x 64 region -- 61 % of it tile instead of 55 % -- measured slower: 172 against 162 us per dense launch at 4096^2)
#ifndef OC_ST_RWV
#define OC_ST_RWV 32
#define OC_ST_RH 32
#endif
#define OC_RING_STX (4 * OC_ST_RWV - 16)
#define OC_RING_STY (OC_ST_RH - 12)
struct OcRingHead {
    int32_t n_slots;             // accepted steps of the last solve
    int32_t nt;                  // its horizon: slices 0 .. nt - 2 are covered by the slice table
    int32_t cap, nslice;
    double te_step, Tfinal;      // t_eval[k] = k * te_step + Tfinal  (np.linspace(T, 0, nt))
    int32_t pad[8];
};
template <typename T>
struct OcRingSlot {
    double out_t, out_hh, out_unscale;      // the step: start time, signed length, 2^expo of its scale
    int32_t k_lo, k_hi;                     // t_eval indices [k_lo, k_hi) lie in the step
    int32_t dense, pad;                     // 1: every stage tile of the slot holds data (first[] is not consulted)
    T lim_s, u;                             // the clip level in the slot's scale; the far field's constant value of y
};
template <typename T>
struct OcRingSlice {
    int32_t slot, pad;
    T h[4];                                 // h x, h x^2, h x^3, h x^4 of the slice's t_eval point in its step
};
#define OC_RING_NEVER 0x7F7F7F7F            // first[t] of a tile that was never computed (memset 0x7F)
constexpr size_t oc_ring_align(size_t v) { return (v + 255) & ~(size_t)255; }
#define OC_RING_SLOTS_OFF oc_ring_align(sizeof(OcRingHead))     // (the slot records follow the head: hjb.hip's controller finds the head from them)
inline int oc_ring_ntile(int rows, int Nx) {
    return ((Nx + OC_RING_STX - 1) / OC_RING_STX) * ((rows + OC_RING_STY - 1) / OC_RING_STY);
}
inline size_t oc_ring_slots_off() { return OC_RING_SLOTS_OFF; }
inline size_t oc_ring_slices_off(size_t tsize, int slots) {
    return oc_ring_slots_off() + oc_ring_align((tsize == 4 ? sizeof(OcRingSlot<float>) : sizeof(OcRingSlot<double>)) * (size_t)slots);
}
inline size_t oc_ring_flags_off(size_t tsize, int slots, int nslice) {
    return oc_ring_slices_off(tsize, slots) +
           oc_ring_align((tsize == 4 ? sizeof(OcRingSlice<float>) : sizeof(OcRingSlice<double>)) * (size_t)std::max(nslice, 1));
}
inline size_t oc_ring_meta_size(size_t tsize, int slots, int nslice, int rows, int Nx) {
    return oc_ring_flags_off(tsize, slots, nslice) + oc_ring_align(sizeof(int32_t) * (size_t)oc_ring_ntile(rows, Nx));
}

struct oc_field_desc {
    int dtype = -1;
    const void* vx = nullptr;
    const void* vy = nullptr;
    // interpolant ring instead of slices (oc_set_field_ring): dtype above = the FIELD's type (what a slice would hold)
    const void* ring_data = nullptr;
    const void* ring_meta = nullptr;
    int ring_state_dtype = -1;              // OC_F32 / OC_F64: type of the ring's values
    int32_t ring_slots = 0;
    int32_t nslice_alloc = 0;
    int32_t nt_opt = 0;
    int32_t row0 = 0, rows = -1;            // row block held locally (rows < 0: the whole grid)
    int32_t own_lo = 0, own_hi = 1 << 30;   // grid rows whose lookups this rank answers
};

struct oc_wallmap {
    void* left = nullptr;       // int2 [Nx][Ny]: (nearest site column <= j or -1, nearest site column >= j+1 or Nx)
    void* right = nullptr;      // float (Ny,Nx): capped distance from each grid node to the nearest site
    void* gap8 = nullptr;       // int32 [Nx][ceil(Ny/8)]: smallest column gap to a site within 8 grid rows
    double edt_cap = 0.0;
    double off = 0.0;           // vmin * 10e3 (the reference adds V*10e3 before the argmin)
    bool built = false;
};

#define OC_MAX_TARGET_RECTS 8

struct oc_key_rects {
    int32_t n = 0;
    double r[OC_MAX_TARGET_RECTS][4];
};

struct oc_timer {
    std::vector<hipEvent_t> ev;   // pairs (begin, end), reused after a reset
    size_t used = 0;
    double bytes = 0.0;
    double add_ms = 0.0;          // launches whose event times the caller resolved itself (hjb_solve_dev)
    int64_t add_n = 0;
};

#define OC_MAX_COMMS 16
struct oc_comm {                // an RCCL communicator the library owns (comm.hip)
    void* comm = nullptr;       // ncclComm_t
    int rank = 0, world = 0;
};

struct oc_handle {
    int device = 0;
    oc_comm comms[OC_MAX_COMMS];
    oc_config cfg;
    int64_t ncell = 0;
    double* X0 = nullptr;       // device (Nx)
    double* Y0 = nullptr;       // device (Ny)
    bool axes_linspace = false; // X0, Y0 are exactly np.linspace(0, L, Nx), np.linspace(0, H, Ny)
    // ---- HJB scratch (allocated lazily per dtype size)
    size_t hjb_bytes = 0;       // size of the scratch below
    void* hjb_buf = nullptr;    // y, ynew, ytmp, K0..K6, phi_tmp  (11 arrays)
    double* red_partial = nullptr;   // per-block partial sums
    double* red_out = nullptr;       // device: few doubles
    double* red_host = nullptr;      // pinned host mirror; [2*OC_MAX_WORLD ..+2]: mailbox (sum, max, sequence number)
    void* red_mailbox_dev = nullptr; // device address of the mailbox
    long long red_seq = 0;
    int32_t red_blocks = 0;
#define OC_CTL_HOST_LOG 4096        // attempt-log words and ...
#define OC_CTL_HOST_TS 4096         // ... step-log doubles the pinned block takes at the end of a solve
#define OC_CTL_HOST_BYTES (8192 + 4 * OC_CTL_HOST_LOG + 8 * OC_CTL_HOST_TS)
    // pinned staging block: [0, 2048) the controller block going up, [2048, 2056) the persistent launch's sync words,
    // [3072, 3072 + sizeof(OcRingHead)) the ring's head going up, [4096, 8192) the two controller blocks coming down, then the logs
    void* hjb_ctl_host = nullptr;    // pinned staging copy of the controller block (host-controlled cross-check variants)
    void* hjb_ctl = nullptr;         // device-resident step controller (hjb.hip: HjbCtl), accepted step times, attempt log
    double* hjb_tsteps = nullptr;
    int32_t* hjb_log = nullptr;
    std::vector<hipEvent_t> hjb_events;
    int32_t* hjb_tail = nullptr;     // ticket counter of the fused tail (hjb.hip: k_stageV<.., FUSED>)
    int hjb_subst = 1;               // ring solves with the active set: far-field substitution (Stage3Args::undef) on / off
    int hjb_tail_fuse = 1;           // 1: ring solves run controller + list builder inside the stage launch when few blocks take part
    int hjb_grid_margin = 64;        // ... plus this many (and half as many again): room for the list to grow until the host sees it
    int hjb_active_rows = 1;         // 1: row-block solves use the active set too (per rank; hjb.hip: HjbCtl::band_lo)
    int hjb_grid_fit = 1;            // 1: a sparse attempt's (fused) launch has about as many blocks as the work list has entries
    int hjb_lead = 0;                // 1: sparse ring solves run an attempt's controller at the START of the next launch, in every block (hjb.hip: LEAD)
    int hjb_tail_fast = 1;           // 1: the fused tail reads short work lists by list position (hjb.hip: TailRec) instead of every tile's words
    int hjb_persist = 1;             // the attempts of a ring solve inside ONE persistent launch (hjb.hip: k_stageV<.., PERSIST>): 0 never,
                                     // 1 grids of a few dozen tiles (where it measures faster), 2 wherever it can run
    int hjb_persist_blocks = 0;      // its grid (0: 128 blocks, half the CUs -- two keys solved concurrently both fit)
    int hjb_persist_wait_us = 50000; // deadline of a block's wait at the in-kernel barrier (a grid that is not co-resident gives up)
    int persist_cap_f32 = -1, persist_cap_f64 = -1;     // block slots of the chip for that kernel (occupancy x CUs), -1: not asked yet
    int32_t* hjb_flags = nullptr;    // active-set flags of the stage tiles, two attempts (hjb.hip: k_stageV)
    size_t hjb_flags_n = 0;
    // ---- per-key data
    std::vector<oc_field_desc> fields;
    std::vector<oc_wallmap> wallmaps;
    std::vector<oc_key_rects> rects;
    void* key_tables_dev = nullptr;  // device copy of per-key descriptors for the step kernels
    void* rect_tab_dev = nullptr;    // compact exit-rectangle table (gcfm.hip: RECT_TAB doubles per key)
    void* key_steps_dev = nullptr;   // per key: what a ring lookup needs of the current step (gcfm.hip: KeyStep)
    int64_t ring_step_ready = -1;    // the simulation step key_steps_dev is resolved for (-1: none)
    bool any_ring = false;           // some key's field is an interpolant ring
    bool key_tables_dirty = true;
    // ---- GCFM scratch
    int32_t resort_every = 32;  // steps between re-sorts (16: 7 500 steps/s at C4, 32: 7 680, 48: 7 710; wider lists beyond)
    int32_t steps_since_sort = 1 << 30;
    int32_t sorted_N = -1;
    bool packed_valid = false;  // the packed snapshot matches the caller's state arrays
    const void* packed_x = nullptr;
    int32_t packed_N = -1;
    int lazy_state = 0;         // 1: oc_gcfm_step leaves the caller's state arrays to k_flush_state (oc_set_tuning "lazy_state")
    bool state_dirty = false;   // the caller's x / y / vx / vy / agent_time are behind the packed snapshot (flush_state)
    oc_agents dirty_agents;     // ... namely these arrays
    hipStream_t dirty_stream = nullptr;     // ... left behind by a step on this stream (a flush that is not ordered by the
                                            // caller's stream -- reallocation, oc_set_tuning -- runs there)
    bool wprev_valid = false;   // the wall searches' incumbents (gcfm.hip: Scratch::wprev) belong to the packed agents
    int wall_warm = 1;          // 1: a wall search starts from the cell the previous step found (oc_set_tuning "wall_warm")
    bool packed_seq = false;    // the snapshot has only been advanced by sequential steps since its re-sort
    int seq_rounds_hint = 8;    // sequential mode: dependency levels of the last step (rounds to enqueue before looking)
    double seq_margin = 0.25;   // sequential mode: bound on one step's displacement (m), checked every round
    int pair_split = 4;         // pair wavefronts per tile (1, 2 or 4): occupancy / balance of k_forces (gcfm.hip)
    int diag = 0;               // diagnostic builds of k_forces (gcfm.hip: GcfmParams::diag) -- results are WRONG with it
    int prep_blocks = 256;      // blocks of k_forces that do the wall search / field lookup of the tiles
    bool work_valid = false;    // tile_work holds a step's measured pair work (for the dispatch order)
    int work_tiles = -1;
    void* gcfm_buf = nullptr;   // carved in gcfm.hip
    size_t gcfm_bytes = 0;
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    void* shape_buf = nullptr;  // room shapes of oc_rasterise_potential (raster.hip)
    size_t shape_bytes = 0;
    void* dens_buf = nullptr;   // scratch of oc_density (density.hip)
    size_t dens_bytes = 0;
    uint32_t fix_flags = 0;     // OC_FIX_* (oc_set_fix_flags): 0 = bug-compatible with the reference
    // ---- profiling
    bool profiling = false;
    int prof_mode = 0;
    int64_t prof_tick = 0;          // GCFM step counter for sampled timers
    int hjb_variant = 0;
    int out_tile = -1;              // tile of k_out: -1 = by grid size, 0 = 64 x 16, 1 = 128 x 8, 2 = 256 x 4 cells
    int hjb_fuse = 0;               // stage launches of an RK attempt: 0 = all six stages in one, 1 = 2 x three stages (hjb.hip)
    // exact active set (hjb.hip): 0 off, 1 on for grids of >= OC_ACTIVE_MIN_TILES stage tiles, 2 on at any size.  (The threshold
    // was 1024 while a sparse attempt cost 22 us; at 15.5 us the list pays from ~100 tiles on: exit_opposite, 130 tiles, 1.76 ->
    // 1.61 ms per solve, slalom, 520 tiles, 2.82 -> 1.66 ms; room_test's 20 tiles run faster whole, inside the persistent launch.)
#define OC_ACTIVE_MIN_TILES 96
    int hjb_active = 1;
    std::map<std::string, oc_timer> timers;
};

// ---- device arithmetic shared by the slice writer (hjb.hip) and the lookup that evaluates the ring (gcfm.hip)
// Fused multiply-add, written out: the library is compiled with -ffp-contract=off (the float64 index decisions of the
// agent step must round like NumPy), so the HJB kernels ask for their FMAs explicitly -- ALL variants in the same
// places, so that they stay bit-identical to each other: a_0 k_0 + a_1 k_1 + ... is one rounding per term instead of two.
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double fast_rcp(double x) { return 1.0 / x; }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double fast_rsq(double x) { return 1.0 / sqrt(x); }

// Unit-or-zero direction of v = grad/(mu*phi_c) (optimals.py:179-186) without forming v: in float32 the ratio
// overflows next to walls late in the solve (|grad| ~ 2^64 against phi_c = lim*2^-e), although its direction and
// the decision |v| > lim are perfectly representable.  |v| > lim  <=>  m*nn > lim*mu*phi_c with m = max(|gx|,|gy|),
// nn = |(gx,gy)/m| in [1, sqrt 2].  (|v| == lim, 0/0 in the reference, is treated as "not greater": zero.)
template <typename T>
__device__ __forceinline__ void direction_of(T gx, T gy, T mupc, T lim, T& ox, T& oy) {
    const T m = fmax(fabs(gx), fabs(gy));
    ox = (T)0;
    oy = (T)0;
    if (m > (T)0) {
        const T im = fast_rcp(m);
        const T ax = gx * im, ay = gy * im;
        const T n2 = ax * ax + ay * ay;          // in [1, 2]
        const T inn = fast_rsq(n2);
        if (m * (n2 * inn) > lim * mupc) {
            ox = ax * inn;
            oy = ay * inn;
        }
    } else if (m != m) {                 // NaN input stays visible
        ox = m;
        oy = m;
    }
}

// RCCL collectives issued by the library itself (comm.hip)
int oc_comm_allgather_f64(oc_handle* h, int slot, double* buf, int64_t count, hipStream_t s);
int oc_comm_allgather_f64_oop(oc_handle* h, int slot, const double* in, double* out, int64_t count, hipStream_t s);
int oc_comm_allreduce_sum_f64(oc_handle* h, int slot, double* buf, int64_t count, hipStream_t s);
int oc_comm_exchange_halo(oc_handle* h, int slot, const void* send_up, void* recv_up, const void* send_dn, void* recv_dn,
                          size_t bytes, int has_up, int has_dn, hipStream_t s);
void oc_comm_destroy_all(oc_handle* h);

// lazy_state (gcfm.hip): bring the caller's state arrays up to date with the packed snapshot, if they are behind
int oc_flush_state(oc_handle* h, hipStream_t s);

// timers (oc_api.hip)
void oc_timer_begin(oc_handle* h, const char* name, hipStream_t s);
void oc_timer_end(oc_handle* h, const char* name, hipStream_t s, double bytes = 0.0);

// sort.hip: order[] = agent indices sorted by (cellkey, index); keys/vals scratch live in the handle
int oc_sort_pairs(oc_handle* h, uint32_t* keys_in, uint32_t* keys_out, int32_t* vals_in, int32_t* vals_out,
                  int n, int end_bit, hipStream_t s);
