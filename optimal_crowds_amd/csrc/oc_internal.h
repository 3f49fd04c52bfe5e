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

struct oc_field_desc {
    int dtype = -1;
    const void* vx = nullptr;
    const void* vy = nullptr;
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

struct oc_handle {
    int device = 0;
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
    void* hjb_ctl_host = nullptr;    // pinned staging copy of the controller block (host-controlled cross-check variants)
    void* hjb_ctl = nullptr;         // device-resident step controller (hjb.hip: HjbCtl), accepted step times, attempt log
    double* hjb_tsteps = nullptr;
    int32_t* hjb_log = nullptr;
    std::vector<hipEvent_t> hjb_events;
    int32_t* hjb_flags = nullptr;    // active-set flags of the stage tiles, two attempts (hjb.hip: k_stageV)
    size_t hjb_flags_n = 0;
    // ---- per-key data
    std::vector<oc_field_desc> fields;
    std::vector<oc_wallmap> wallmaps;
    std::vector<oc_key_rects> rects;
    void* key_tables_dev = nullptr;  // device copy of per-key descriptors for the step kernels
    void* rect_tab_dev = nullptr;    // compact exit-rectangle table (gcfm.hip: RECT_TAB doubles per key)
    bool key_tables_dirty = true;
    // ---- GCFM scratch
    int32_t resort_every = 32;  // steps between re-sorts (16: 7 500 steps/s at C4, 32: 7 680, 48: 7 710; wider lists beyond)
    int32_t steps_since_sort = 1 << 30;
    int32_t sorted_N = -1;
    bool packed_valid = false;  // the packed snapshot matches the caller's state arrays
    const void* packed_x = nullptr;
    int32_t packed_N = -1;
    bool packed_seq = false;    // the snapshot has only been advanced by sequential steps since its re-sort
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
    int hjb_active = 1;             // exact active set (hjb.hip): 0 off, 1 on for grids of >= 1024 stage tiles, 2 on at any size
    std::map<std::string, oc_timer> timers;
};

// timers (oc_api.hip)
void oc_timer_begin(oc_handle* h, const char* name, hipStream_t s);
void oc_timer_end(oc_handle* h, const char* name, hipStream_t s, double bytes = 0.0);

// sort.hip: order[] = agent indices sorted by (cellkey, index); keys/vals scratch live in the handle
int oc_sort_pairs(oc_handle* h, uint32_t* keys_in, uint32_t* keys_out, int32_t* vals_in, int32_t* vals_out,
                  int n, int end_bit, hipStream_t s);
