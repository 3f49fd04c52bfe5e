// Handle lifetime, error reporting and kernel timers of liboc_hip.so.
#include "oc_internal.h"
#include <cstdarg>
#include <cstdlib>
#include <algorithm>

static thread_local char g_err[1024] = "";

void oc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int oc_abi_version(void) { return OC_ABI_VERSION; }
extern "C" const char* oc_last_error(void) { return g_err; }

extern "C" int oc_create(int device, const oc_config* cfg, const double* X0_host, const double* Y0_host,
                         oc_handle** out) {
    OC_REQUIRE(cfg && out && X0_host && Y0_host, "null argument");
    OC_REQUIRE(cfg->Nx >= 3 && cfg->Ny >= 3, "grid must be at least 3x3");
    OC_REQUIRE(cfg->max_keys >= 1 && cfg->max_agents >= 0, "bad capacities");
    int ndev = 0;
    OC_HIP_CHECK(hipGetDeviceCount(&ndev));
    OC_REQUIRE(device >= 0 && device < ndev, "no such HIP device");
    OC_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    OC_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        oc_set_error("liboc_hip is built for gfx950 (MI355X); device %d is %s", device, prop.gcnArchName);
        return OC_ERR_STATE;
    }
    oc_handle* h = new oc_handle();
    h->device = device;
    h->cfg = *cfg;
    h->ncell = (int64_t)cfg->Nx * cfg->Ny;
    h->fields.resize(cfg->max_keys);
    h->wallmaps.resize(cfg->max_keys);
    h->rects.resize(cfg->max_keys);
    OC_HIP_CHECK(hipMalloc(&h->X0, sizeof(double) * cfg->Nx));
    OC_HIP_CHECK(hipMalloc(&h->Y0, sizeof(double) * cfg->Ny));
    OC_HIP_CHECK(hipMemcpy(h->X0, X0_host, sizeof(double) * cfg->Nx, hipMemcpyHostToDevice));
    OC_HIP_CHECK(hipMemcpy(h->Y0, Y0_host, sizeof(double) * cfg->Ny, hipMemcpyHostToDevice));
    {   // np.linspace(0, len, n): fl(j * step), last element = len (gcfm.hip: struct Axis)
        auto is_linspace = [](const double* a, int n, double len) {
            if (n < 2) return false;
            const double step = len / (n - 1);
            for (int j = 0; j < n - 1; ++j)
                if (a[j] != (double)j * step) return false;
            return a[n - 1] == len;
        };
        h->axes_linspace = is_linspace(X0_host, cfg->Nx, cfg->room_length) && is_linspace(Y0_host, cfg->Ny, cfg->room_height);
    }
    h->red_blocks = std::max(1024, ((cfg->Nx + 31) / 32) * ((cfg->Ny + 15) / 16));
    OC_HIP_CHECK(hipMalloc(&h->red_partial, sizeof(double) * (2 * (size_t)h->red_blocks + 2 * 4 * 128 + 2 * 1024)));     // (+ the fused tail's records, of two attempts: hjb.hip TAIL_REC_CAP; + a second set of partial pairs: k_init_a)
    OC_HIP_CHECK(hipMalloc(&h->red_out, sizeof(double) * 16));
    OC_HIP_CHECK(hipHostMalloc(&h->red_host, sizeof(double) * (2 * OC_MAX_WORLD + 8), hipHostMallocMapped));
    memset(h->red_host, 0, sizeof(double) * (2 * OC_MAX_WORLD + 8));
    OC_HIP_CHECK(hipHostGetDevicePointer(&h->red_mailbox_dev, h->red_host + 2 * OC_MAX_WORLD, 0));
    if (const char* e = getenv("OC_HJB_FUSE")) {        // A/B of the stage-launch shapes over a whole test run
        const int v = atoi(e);
        if (v >= 0 && v <= 1) h->hjb_fuse = v;
    }
    if (const char* e = getenv("OC_HJB_ACTIVE")) {      // 0: compute every tile of every attempt, 2: active set at any grid size
        const int v = atoi(e);
        if (v >= 0 && v <= 2) h->hjb_active = v;
    }
    *out = h;
    return OC_OK;
}

extern "C" int oc_destroy(oc_handle* h) {
    if (!h) return OC_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    oc_comm_destroy_all(h);
    (void)hipFree(h->X0);
    (void)hipFree(h->Y0);
    (void)hipFree(h->hjb_buf);
    (void)hipFree(h->red_partial);
    (void)hipFree(h->red_out);
    (void)hipHostFree(h->red_host);
    (void)hipFree(h->hjb_ctl);
    if (h->hjb_ctl_host) (void)hipHostFree(h->hjb_ctl_host);
    (void)hipFree(h->hjb_tsteps);
    (void)hipFree(h->hjb_log);
    (void)hipFree(h->hjb_flags);
    (void)hipFree(h->hjb_tail);
    for (hipEvent_t e : h->hjb_events) (void)hipEventDestroy(e);
    for (auto& w : h->wallmaps) {
        (void)hipFree(w.left);
        (void)hipFree(w.right);
        (void)hipFree(w.gap8);
    }
    (void)hipFree(h->key_tables_dev);
    (void)hipFree(h->rect_tab_dev);
    (void)hipFree(h->key_steps_dev);
    (void)hipFree(h->gcfm_buf);
    (void)hipFree(h->sort_tmp);
    (void)hipFree(h->dens_buf);
    (void)hipFree(h->shape_buf);
    for (auto& kv : h->timers)
        for (hipEvent_t e : kv.second.ev) (void)hipEventDestroy(e);
    delete h;
    return OC_OK;
}

extern "C" int oc_set_profiling(oc_handle* h, int mode) {
    OC_REQUIRE(h, "null handle");
    h->profiling = mode != 0;
    h->prof_mode = mode;
    return OC_OK;
}

// mode bits: 1 = HJB kernels, 2 = GCFM kernels every step, 4 = GCFM kernels every 16th step, 8 = every 4th step (event
// records between the 2 launches of a step cost ~14 us of a 144 us step, so the bench samples them), 16 = HJB: no persistent
// launch (every attempt is a launch of its own and can be bracketed)
static bool timer_on(oc_handle* h, const char* name) {
    if (!h->profiling) return false;
    if (name[0] == 'h' || name[0] == 'd') return (h->prof_mode & 1) != 0;     // HJB kernels, density
    if (h->prof_mode & 2) return true;
    if ((h->prof_mode & 8) && (h->prof_tick % 4 == 0)) return true;
    return (h->prof_mode & 4) && (h->prof_tick % 16 == 0);
}

void oc_timer_begin(oc_handle* h, const char* name, hipStream_t s) {
    if (!timer_on(h, name)) return;
    oc_timer& t = h->timers[name];
    if (t.used + 2 > t.ev.size()) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        t.ev.push_back(a);
        t.ev.push_back(b);
    }
    (void)hipEventRecord(t.ev[t.used], s);
}

void oc_timer_end(oc_handle* h, const char* name, hipStream_t s, double bytes) {
    if (!timer_on(h, name)) return;
    oc_timer& t = h->timers[name];
    if (t.used + 2 > t.ev.size()) return;
    (void)hipEventRecord(t.ev[t.used + 1], s);
    t.used += 2;
    t.bytes += bytes;
}

extern "C" int oc_kernel_stats(oc_handle* h, const char* name, int64_t* launches, double* total_ms,
                               double* total_bytes) {
    OC_REQUIRE(h && name, "null argument");
    auto it = h->timers.find(name);
    int64_t n = 0;
    double ms = 0.0, bytes = 0.0;
    if (it != h->timers.end()) {
        oc_timer& t = it->second;
        for (size_t q = 0; q + 1 < t.used; q += 2) {
            OC_HIP_CHECK(hipEventSynchronize(t.ev[q + 1]));
            float e = 0.f;
            OC_HIP_CHECK(hipEventElapsedTime(&e, t.ev[q], t.ev[q + 1]));
            ms += e;
            ++n;
        }
        bytes = t.bytes;
        ms += t.add_ms;
        n += t.add_n;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (total_bytes) *total_bytes = bytes;
    return OC_OK;
}

extern "C" int oc_reset_kernel_stats(oc_handle* h) {
    OC_REQUIRE(h, "null handle");
    for (auto& kv : h->timers) {
        kv.second.used = 0;
        kv.second.bytes = 0.0;
        kv.second.add_ms = 0.0;
        kv.second.add_n = 0;
    }
    return OC_OK;
}

extern "C" int oc_set_hjb_variant(oc_handle* h, int variant) {
    OC_REQUIRE(h && variant >= 0 && variant <= 2, "bad argument");
    h->hjb_variant = variant;
    return OC_OK;
}

extern "C" int oc_set_field(oc_handle* h, int key, int dtype, const void* vx, const void* vy, int32_t nslice_alloc,
                            int32_t nt_opt) {
    OC_REQUIRE(h && key >= 0 && key < (int)h->fields.size(), "bad key");
    OC_REQUIRE(dtype == OC_F32 || dtype == OC_F64, "bad dtype");
    oc_field_desc& f = h->fields[key];
    f.dtype = dtype;
    f.vx = vx;
    f.vy = vy;
    f.nslice_alloc = nslice_alloc;
    f.nt_opt = nt_opt;
    f.row0 = 0;
    f.rows = -1;
    f.own_lo = 0;
    f.own_hi = 1 << 30;
    f.ring_data = f.ring_meta = nullptr;
    f.ring_state_dtype = -1;
    f.ring_slots = 0;
    h->key_tables_dirty = true;
    return OC_OK;
}

// The interpolant ring of a key in the place of its slices: lookups evaluate the direction from it (gcfm.hip: ring_dir).
extern "C" int oc_set_field_ring(oc_handle* h, int key, int dtype, const oc_hjb_ring* ring, int32_t nt_opt, int32_t row0,
                                 int32_t own_lo, int32_t own_hi) {
    OC_REQUIRE(h && key >= 0 && key < (int)h->fields.size(), "bad key");
    OC_REQUIRE(dtype == OC_F32 || dtype == OC_F64 || dtype == OC_F64_F32, "bad dtype");
    OC_REQUIRE(ring && ring->data && ring->meta && ring->slots >= 1 && ring->rows >= 3, "bad ring");
    OC_REQUIRE(h->cfg.Nx % 4 == 0, "the interpolant ring needs Nx % 4 == 0");
    oc_field_desc& f = h->fields[key];
    f.dtype = dtype == OC_F64 ? OC_F64 : OC_F32;        // what a slice would hold: float32 unless the audit mode
    f.vx = f.vy = nullptr;
    f.nslice_alloc = ring->nslice;
    f.nt_opt = nt_opt;
    f.ring_data = ring->data;
    f.ring_meta = ring->meta;
    f.ring_state_dtype = dtype == OC_F32 ? OC_F32 : OC_F64;
    f.ring_slots = ring->slots;
    if (row0 < 0) {                 // the whole grid
        OC_REQUIRE(ring->rows == h->cfg.Ny, "a whole-grid ring has Ny rows");
        f.row0 = 0;
        f.rows = -1;
        f.own_lo = 0;
        f.own_hi = 1 << 30;
    } else {
        OC_REQUIRE(row0 + ring->rows <= h->cfg.Ny, "row block outside the grid");
        OC_REQUIRE(own_lo >= row0 && own_hi <= row0 + ring->rows && own_lo < own_hi, "owned rows outside the row block");
        f.row0 = row0;
        f.rows = ring->rows;
        f.own_lo = own_lo;
        f.own_hi = own_hi;
    }
    h->key_tables_dirty = true;
    return OC_OK;
}

extern "C" int oc_set_field_rows(oc_handle* h, int key, int dtype, const void* vx, const void* vy, int32_t nslice_alloc,
                                 int32_t nt_opt, int32_t row0, int32_t rows, int32_t own_lo, int32_t own_hi) {
    int rc = oc_set_field(h, key, dtype, vx, vy, nslice_alloc, nt_opt);
    if (rc) return rc;
    OC_REQUIRE(row0 >= 0 && rows > 0 && row0 + rows <= h->cfg.Ny, "row block outside the grid");
    OC_REQUIRE(own_lo >= row0 && own_hi <= row0 + rows && own_lo < own_hi, "owned rows outside the row block");
    oc_field_desc& f = h->fields[key];
    f.row0 = row0;
    f.rows = rows;
    f.own_lo = own_lo;
    f.own_hi = own_hi;
    return OC_OK;
}

extern "C" int oc_set_key_targets(oc_handle* h, int key, const double* rects, int32_t n) {
    OC_REQUIRE(h && key >= 0 && key < (int)h->rects.size(), "bad key");
    OC_REQUIRE(n >= 0 && n <= OC_MAX_TARGET_RECTS, "too many target rectangles for one key");
    OC_REQUIRE(n == 0 || rects, "null rects");
    h->rects[key].n = n;
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 4; ++c) h->rects[key].r[i][c] = rects[4 * i + c];
    h->key_tables_dirty = true;
    return OC_OK;
}

extern "C" int oc_set_tuning(oc_handle* h, const char* name, int value) {
    OC_REQUIRE(h && name, "null argument");
    if (!strcmp(name, "lazy_state")) {
        OC_REQUIRE(value == 0 || value == 1, "lazy_state must be 0 or 1");
        if (value == 0 && h->state_dirty) {     // (back to "every step leaves the arrays current": make them so now)
            const hipStream_t fs = h->dirty_stream;      // (the stream of the step that left the snapshot)
            int rcf = oc_flush_state(h, fs);
            if (rcf) return rcf;
            OC_HIP_CHECK(hipStreamSynchronize(fs));
        }
        h->lazy_state = value;
    } else if (!strcmp(name, "pair_split")) {
        OC_REQUIRE(value == 1 || value == 2 || value == 4, "pair_split must be 1, 2 or 4");
        h->pair_split = value;
    } else if (!strcmp(name, "prep_blocks")) {
        OC_REQUIRE(value >= 1 && value <= 65536, "prep_blocks out of range");
        h->prep_blocks = value;
    } else if (!strcmp(name, "resort_every")) {
        OC_REQUIRE(value >= 1 && value <= 4096, "resort_every out of range");
        h->resort_every = value;
    } else if (!strcmp(name, "diag")) {
#ifdef OC_DIAG_BUILD
        OC_REQUIRE(value >= 0 && value <= 15, "diag out of range");
        h->diag = value;        // liboc_hip_diag.so only: parts of k_forces cut out to time the rest -- results are WRONG
#else
        oc_set_error("oc_set_tuning: 'diag' exists only in the diagnostic build (make -C csrc diag; OC_HIP_LIB=.../liboc_hip_diag.so)");
        return OC_ERR_ARG;
#endif
    } else if (!strcmp(name, "out_tile")) {
        OC_REQUIRE(value >= -1 && value <= 2, "out_tile out of range");
        h->out_tile = value;
    } else if (!strcmp(name, "hjb_active")) {
        OC_REQUIRE(value >= 0 && value <= 2, "hjb_active out of range");
        h->hjb_active = value;
    } else if (!strcmp(name, "wall_warm")) {
        OC_REQUIRE(value >= 0 && value <= 1, "wall_warm out of range");
        h->wall_warm = value;
        h->wprev_valid = false;
    } else if (!strcmp(name, "hjb_subst")) {
        OC_REQUIRE(value >= 0 && value <= 1, "hjb_subst out of range");
        h->hjb_subst = (int)value;
    } else if (!strcmp(name, "hjb_active_rows")) {
        OC_REQUIRE(value >= 0 && value <= 1, "hjb_active_rows must be 0 or 1");
        h->hjb_active_rows = value;
    } else if (!strcmp(name, "hjb_grid_fit")) {
        OC_REQUIRE(value >= 0 && value <= 1, "hjb_grid_fit must be 0 or 1");
        h->hjb_grid_fit = value;
    } else if (!strcmp(name, "hjb_grid_margin")) {
        OC_REQUIRE(value >= -100000 && value <= 100000, "hjb_grid_margin out of range");
        h->hjb_grid_margin = value;
    } else if (!strcmp(name, "hjb_tail_fast")) {
        OC_REQUIRE(value >= 0 && value <= 1, "hjb_tail_fast must be 0 or 1");
        h->hjb_tail_fast = value;
    } else if (!strcmp(name, "hjb_lead")) {
        OC_REQUIRE(value >= 0 && value <= 1, "hjb_lead must be 0 or 1");
        h->hjb_lead = value;
    } else if (!strcmp(name, "hjb_persist")) {
        OC_REQUIRE(value >= 0 && value <= 2, "hjb_persist must be 0, 1 or 2");
        h->hjb_persist = value;
    } else if (!strcmp(name, "hjb_persist_blocks")) {
        OC_REQUIRE(value >= 0 && value <= 4096, "hjb_persist_blocks out of range");
        h->hjb_persist_blocks = value;
    } else if (!strcmp(name, "hjb_persist_wait_us")) {
        OC_REQUIRE(value >= 10 && value <= 10000000, "hjb_persist_wait_us out of range");
        h->hjb_persist_wait_us = value;
    } else if (!strcmp(name, "hjb_tail")) {
        OC_REQUIRE(value >= 0 && value <= 2, "hjb_tail out of range");
        h->hjb_tail_fuse = value;
    } else if (!strcmp(name, "hjb_fuse")) {
        OC_REQUIRE(value >= 0 && value <= 1, "hjb_fuse out of range");
        h->hjb_fuse = value;
    } else if (!strcmp(name, "seq_margin_mm")) {
        OC_REQUIRE(value >= 1 && value <= 1500, "seq_margin_mm out of range");
        h->seq_margin = 1e-3 * value;
    } else {
        oc_set_error("oc_set_tuning: unknown knob '%s'", name);
        return OC_ERR_ARG;
    }
    return OC_OK;
}

extern "C" int oc_set_fix_flags(oc_handle* h, uint32_t flags) {
    OC_REQUIRE(h, "null handle");
    OC_REQUIRE((flags & ~(uint32_t)OC_FIX_ALL) == 0, "unknown OC_FIX_* bit");
    h->fix_flags = flags;
    h->packed_valid = false;        // (the per-agent wall search radius depends on OC_FIX_WALL_FORCE)
    h->wprev_valid = false;
    return OC_OK;
}

extern "C" int oc_set_resort_every(oc_handle* h, int32_t steps) {
    OC_REQUIRE(h, "null handle");
    h->resort_every = steps > 0 ? steps : 32;
    return OC_OK;
}
