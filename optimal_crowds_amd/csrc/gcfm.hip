// GCFM agent update (Generalized Centrifugal Force Model), synchronous (frozen-snapshot) form.
// Replaces reference simulations.py:252-342 (step driver), pedestrians.py:216-280 (agents_repulsion),
// pedestrians.py:282-334 (wall_repulsion), optimals.py:212-250 (choose_optimal_velocity).
//
// Layout: caller-owned SoA state in agent order (float64 x,y,vx,vy,v_des; uint8 status; int32 key).
// The library keeps a packed snapshot in SLOT order (agents sorted along a Hilbert curve over 2 m coarse cells; a
// tile = 64 slots = one wavefront):
//   ppos  double2 (x,y)      pvel  float4 (vx,vy,|v|,1/a^2)     phead float2 (unit heading, (1,0) if v=0)
//   tile_box double4 (minx,miny,maxx,maxy) of a tile's active agents, qbox: the same for its four quarters (16
//   consecutive records);  nb_list: tiles that can come within the cutoff before the next re-sort (boxes at sort
//   time, cutoff + 2D margin; validity is re-checked every step)
//   pvd double2 (float64 velocity), pdk (v_des, key): slot-ordered copies, so a step gathers nothing by agent index
// One step = 2 launches:
//   k_forces  blocks of 4 wavefronts.  PAIR blocks carry 4/NS tiles, NS wavefronts per tile sharing the tile's i
//             records and float64 force sums in LDS; a wavefront takes its j-range of the candidate j-tiles (equal work
//             per share; candidates culled against the box of that j-range), NS tiles at a time as one 64-record
//             virtual j-tile: phase 1 builds per-lane candidate masks
//             with a conservative float32 test, the candidates are compacted into an LDS list (two per entry), phase 2
//             walks the list: exact float64 cutoff decision (the reference's sqrt(dx^2+dy^2) < cutoff), packed float32
//             force.  PREP blocks do what does not need the pair sums, one lane per agent: exact nearest-wall-cell
//             search + wall force, floor-field lookup -- chains of dependent loads that hide beside the pair work.
//   k_finish  per agent, float64, one level of slot-ordered loads: noisy relaxation step, speed clamp, exit test; then
//             the next step's packed record and tile box.
#include "oc_internal.h"
#include <cmath>
#include <algorithm>
#include <cstdlib>

namespace {

struct KeyDev {
    const void* vx;
    const void* vy;
    const int2* lr;        // [Nx][Ny]: (nearest site column <= j or -1, nearest site column >= j+1 or Nx) of row r
    const float* edt;      // (Ny,Nx) distance from each grid node to the nearest site, capped at edt_cap (lower bound)
    const int32_t* gap8;   // [Nx][ceil(Ny/8)]: min over 8 grid rows of the column gaps to the row's nearest sites (left | right << 16)
    double off;
    double edt_cap;
    int32_t dtype, nslice_alloc, nt_opt, nrect;
    int32_t row0, rows, own_lo, own_hi;     // row block of the field held here / grid rows whose lookups are answered here
    // interpolant ring in the place of the slices (oc_set_field_ring; layout: oc_internal.h)
    const void* ring_data;                  // NULL: slices (vx, vy)
    const void* ring_slots;                 // OcRingSlot<T> [slots]
    const void* ring_slices;                // OcRingSlice<T> [nslice]
    const int32_t* ring_first;              // [ring_ntile]: first slot in which each stage tile holds data
    int64_t ring_slot_elems;
    int32_t ring_state_dtype, ring_ntile, ring_tnbx, ring_pad;
    double rect[OC_MAX_TARGET_RECTS][4];
};

// What a lookup through the interpolant ring needs of (key, simulation step), resolved ONCE per step and key instead of
// by every lane through two dependent loads (slice table -> slot record): written by k_ring_step, or by k_finish for the
// step that follows it.  Values of the ring's type T are held as doubles (exact for float).
struct KeyStep {
    const void* slot;           // the slot's data, or NULL: no ring / the step lies beyond the horizon
    const int32_t* first;       // per stage tile the first slot that holds data for it, or NULL: every tile does
    double h[4], lim_s, u;
    int32_t slot_index, pad;
};


struct GcfmParams {
    int32_t Ny, Nx;
    double dx, dy, L, H, dt;
    double a_min, tau_a, b_min, b_max, eta, eta_walls, cutoff, v_max, relaxation, noise_amp;
    double wall_radius;
    double cut_d2max;   // largest double d2 with sqrt(d2) < cutoff: `d2 <= cut_d2max` IS the reference's decision
    double cell;        // coarse cell size for ordering
    double drift;       // D: how far a tile box may grow beyond its sort-time box before the lists are invalid
    int32_t ncx, ncy, horder;
    double mu, lim;     // hjb mu and the clip level lim (optimals.py:95): `vels` evaluated from the interpolant ring
    uint32_t fix;       // OC_FIX_* bits (oc_set_fix_flags); 0 = the reference's behaviour
    uint32_t diag;      // OC_DIAG_BUILD only (liboc_hip_diag.so, `make diag`): cuts parts out of k_forces to time the rest --
                        // 1 = no phase 2, 2 = no prep work, 4 = no compaction either, 8 = no lookups; the results are WRONG
};
// The release library has no diagnostic cuts: the tests compile to constants and the hot kernel carries no branch for them.
#ifdef OC_DIAG_BUILD
#define OC_DIAG(P, bit) (((P).diag & (bit)) != 0u)
#else
#define OC_DIAG(P, bit) false
#endif

#define NB_MAX 63
// ctrl words
#define C_EXIT 0
#define C_ERR 1
#define C_ACTIVE 2
#define C_INVALID 6     // neighbour lists no longer cover the tile boxes -> full tile scan (sticky until the re-sort)
#define C_NACTIVE 7     // active agents (set at the re-sort, decremented by exits): no hot same-address atomics
// words 12..15 survive the re-sort (oc_run_status): what a driver that does not read the counters every step needs
#define C_RS_DONE 12    // 1 + the step in which the last active agent left (0: somebody is still inside)
#define C_RS_ERRA 13    // 1 + an agent that was in the IndexError band, and the (first) step in which it was
#define C_RS_ERRS 14
#define CTRL_VOLATILE_BYTES 48      // words 0..11 are reset at every re-sort

// What the nearest-wall search of one agent needs (wall_prepare).
struct WEntry {
    double px, py;          // the agent's position
    double lim2;            // min(search radius^2, upper bound^2 on the distance to the nearest site)
    double off;             // the key's objective offset V_min * 10e3
    const int2* col;        // lr + jx * Ny: the agent's column of the transposed nearest-site table
    const int32_t* g8;      // gap8 + jx * nb8
    int32_t b0;             // 8-row block of the nearest grid row
    int32_t prune_ok;       // column-gap bounds valid (agent within one cell of column jx)
    int32_t jx, pad;        // the agent's column: X0[jx] <= px < X0[jx + 1] inside the room
};

struct Snap {               // the packed snapshot
    double2* ppos;
    float4* pvel;
    float2* phead;
    int32_t* pidx;
    double4* tile_box;
    double4* qbox;          // [4 * ntiles]: boxes of the tile's four quarters (records 16q .. 16q + 15)
};

struct PDK {                // per slot, fixed between re-sorts
    double v_des;
    int32_t key, pad;
};

struct Scratch {
    uint32_t *keys_a, *keys_b;
    int32_t *vals_a, *vals_b;
    Snap snap;
    double2* pvd;           // float64 velocity per slot (only the tile's own finisher touches it: single buffer)
    double* ptime;          // agent_time per slot (ped.time): advanced here every step, scattered to the caller's array with the
                            // rest of the state -- every step, or (lazy_state) when somebody needs it
    PDK* pdk;
    double4* tile_box0;
    int32_t* nb_list;       // [ntiles][NB_MAX + 1]: count (or -1 = overflow) then tile ids
    int32_t* tile_work;     // candidate-list entries each tile processed in the last step (its pair work)
    int32_t* tile_order;    // tiles by decreasing work: k_forces dispatches the heavy ones first
    // per-slot results of k_forces for k_finish: element of slot k at base[(k >> 6) * ts + (k & 63)] (xs()): plain arrays
    // (ts = 64), or -- sharded step -- the caller's exchange buffer, blocked by tile so that a rank's share of the tiles is
    // ONE contiguous block for the all-gather: [pair sums x 64][wall forces x 64]([desired velocities x 64]) per tile
    double2 *f_pair, *f_wall, *des;
    int32_t ts_pair, ts_wall, ts_des;   // double2 per tile (64: plain arrays)
    int32_t x_tpr, x_blk;               // exchange buffer: tiles per rank block (0: no blocks), double2 per block
    long long* widx;
    int32_t* wprev;         // per AGENT: the nearest-wall cell the last step's search found (-1: none) -- the next search's
                            // incumbent (an agent moves a few centimetres per step; the sites do not move at all)
    int32_t* ctrl;
    int32_t* seq_rank;      // sequential mode: position of the slot's agent in the step's permutation
    int32_t* seq_state;     // 0 = not yet updated this step, 1 = being updated in this round, 2 = updated (or no agent)
    int32_t* seq_dep;       // earlier agents within reach that have not moved yet (k_seq_count; released by k_finish<true>)
    double2* seq_pos0;      // the slot's position at the start of the step (the reach test's positions; 1e300: no agent)
    int32_t* seq_ctl;       // [0..3]: agents still pending after round r (slot r & 3); [4]: displacement-bound violations;
                            // [5]: rounds that advanced somebody (= the permutation's dependency levels)
    int32_t Np, ntiles;
};

// element of slot k of one of the per-slot result arrays (Scratch::f_pair ...); ts = that array's tile stride.  In the
// sharded step's exchange buffer the tiles of rank r form block r (x_tpr tiles + the block's 8-double tail).
__device__ __forceinline__ double2& xs(const Scratch& S, double2* base, int ts, int k) {
    const int t = k >> 6;
    if (S.x_tpr != 0 && ts != 64) {
        const int r = t / S.x_tpr;
        return base[(size_t)r * S.x_blk + (size_t)(t - r * S.x_tpr) * ts + (k & 63)];
    }
    return base[(size_t)t * ts + (k & 63)];
}


// A grid axis.  The reference builds both with np.linspace(0, len, n) (simulations.py:69-70), whose elements are
// fl(j * step), step = len / (n - 1), the last one set to len; when the caller's arrays are exactly that (checked in
// oc_create) coordinates are computed instead of loaded -- the nearest-wall search is a chain of dependent loads.
struct Axis {
    const double* a;
    double h, len;
    int32_t n, lin;
    __device__ __forceinline__ double operator[](int j) const {
        return lin ? (j == n - 1 ? len : (double)j * h) : a[j];
    }
};

__device__ __forceinline__ double wmin(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
// box of a tile's agents and of its four quarters (16 consecutive records: compact pieces of the Hilbert order).
// px/py = +-inf pairs of the lane (an absent agent passes +inf, +inf, -inf, -inf).  Lanes 0, 16, 32, 48 store the
// quarter boxes, lane 0 the tile box.
__device__ __forceinline__ double4 tile_boxes(double lox, double loy, double hix, double hiy, int lane, int it,
                                              double4* __restrict__ qbox) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        lox = fmin(lox, __shfl_xor(lox, o, 64));
        loy = fmin(loy, __shfl_xor(loy, o, 64));
        hix = fmax(hix, __shfl_xor(hix, o, 64));
        hiy = fmax(hiy, __shfl_xor(hiy, o, 64));
    }
    if ((lane & 15) == 0) qbox[it * 4 + (lane >> 4)] = make_double4(lox, loy, hix, hiy);
#pragma unroll
    for (int o = 32; o > 8; o >>= 1) {
        lox = fmin(lox, __shfl_xor(lox, o, 64));
        loy = fmin(loy, __shfl_xor(loy, o, 64));
        hix = fmax(hix, __shfl_xor(hix, o, 64));
        hiy = fmax(hiy, __shfl_xor(hiy, o, 64));
    }
    return make_double4(lox, loy, hix, hiy);
}

// ---------------------------------------------------------------------------------- ordering
// Hilbert-curve index of coarse cell (x, y) on a 2^order x 2^order grid: consecutive indices are adjacent cells, so
// any 64 consecutive agents form a compact blob (a row-major order makes tiles that wrap a cell row span the room).
__device__ __forceinline__ uint32_t hilbert_index(uint32_t x, uint32_t y, int order) {
    uint32_t d = 0;
    for (uint32_t s = 1u << (order - 1); s > 0; s >>= 1) {
        const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0) {                       // rotate the quadrant
            if (rx == 1) {
                x = s - 1 - x;
                y = s - 1 - y;
            }
            const uint32_t t = x;
            x = y;
            y = t;
        }
    }
    return d;
}

__global__ void k_cell_keys(const double* __restrict__ x, const double* __restrict__ y,
                            const uint8_t* __restrict__ status, int N, GcfmParams P, uint32_t* __restrict__ keys,
                            int32_t* __restrict__ vals) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    uint32_t key = 1u << (2 * P.horder);            // inactive agents sort to the end (curve indices are < 4^horder)
    if (status[i]) {
        double fx = x[i] / P.cell, fy = y[i] / P.cell;
        int cx = (fx > 0) ? (fx < P.ncx - 1 ? (int)fx : P.ncx - 1) : 0;
        int cy = (fy > 0) ? (fy < P.ncy - 1 ? (int)fy : P.ncy - 1) : 0;
        key = hilbert_index((uint32_t)cx, (uint32_t)cy, P.horder);
    }
    keys[i] = key;
    vals[i] = i;
}

// Exact nearest-site query (same cell as numpy's argmin over the full grid, see oracle WallMap for the proof sketch):
// per grid row only the nearest site left and right of the agent's column can be the row minimum (wall_search below).
// search radius when P.wall_radius < 0: beyond it the force 3|R| exp(-(|R| - q)/(eta_w (1 + v_rel))) is below
// 1e-12 m/s for this agent's own speed (q <= a^2, v_rel <= |v|; 33 = ln(3 * 60 m / 1e-12 m/s) rounded up)
__device__ __forceinline__ double wall_radius_of(const GcfmParams& P, double s) {
    if (P.wall_radius >= 0) return P.wall_radius;
    double a = fmax(P.a_min + P.tau_a * s, P.b_max);        // q <= max(a, b)^2  (<= max(a, b) with OC_FIX_WALL_FORCE)
    return ((P.fix & OC_FIX_WALL_FORCE) ? a : a * a) + P.eta_walls * (1.0 + s) * 33.0;
}

// Can this agent feel a wall?  Uses the capped distance transform at the nearest grid node:
// d(node) - |p - node| <= d*(p) <= d(node) + |p - node|.  Returns false when d*(p) provably exceeds the agent's
// search radius (force < 1e-12 m/s); ub2 = squared upper bound on d*(p) (or inf).
__device__ __forceinline__ bool wall_needed(const GcfmParams& P, const KeyDev& kd, const Axis X0,
                                            const Axis Y0, double px, double py, double speed,
                                            double& rad2, double& ub2) {
    const int Nx = P.Nx, Ny = P.Ny;
    const double hx = P.L / (Nx - 1), hy = P.H / (Ny - 1);
    const double rad = wall_radius_of(P, speed);
    rad2 = rad > 0 ? rad * rad : INFINITY;
    int jn = (int)floor(px / hx + 0.5), in = (int)floor(py / hy + 0.5);
    jn = jn < 0 ? 0 : (jn > Nx - 1 ? Nx - 1 : jn);
    in = in < 0 ? 0 : (in > Ny - 1 ? Ny - 1 : in);
    const double ex = X0[jn] - px, ey = Y0[in] - py;
    const double delta = sqrt(ex * ex + ey * ey) * (1.0 + 1e-12) + 1e-12;
    const double dn = (double)kd.edt[(size_t)in * Nx + jn];
    ub2 = INFINITY;
    // a capped node (+inf) only proves "farther than the cap" (a radius beyond it -- an agent faster than v_max --
    // still gets its search)
    const double dlow = (dn < kd.edt_cap) ? dn * (1.0 - 1e-6) : kd.edt_cap;
    if ((rad > 0) && (dlow - delta > rad)) return false;
    if (dn < kd.edt_cap) {
        double ub = dn * (1.0 + 1e-6) + delta + 1e-9;
        ub2 = ub * ub;
    }
    return true;
}

// wall_needed + the start of the search (column, nearest row): what wall_search needs of one agent
__device__ __forceinline__ bool wall_prepare(const GcfmParams& P, const KeyDev& kd, const Axis X0, const Axis Y0, double px,
                                             double py, double speed, WEntry& e) {
    double rad2, ub2;
    if (!wall_needed(P, kd, X0, Y0, px, py, speed, rad2, ub2)) return false;
    const int Nx = P.Nx, Ny = P.Ny;
    const double hx = P.L / (Nx - 1), hy = P.H / (Ny - 1);
    int jx = (int)floor(px / hx);
    jx = jx < 0 ? 0 : (jx > Nx - 1 ? Nx - 1 : jx);
    while (jx > 0 && X0[jx] > px) --jx;
    while (jx < Nx - 1 && X0[jx + 1] <= px) ++jx;
    int r0 = (int)floor(py / hy + 0.5);
    r0 = r0 < 0 ? 0 : (r0 > Ny - 1 ? Ny - 1 : r0);
    e.px = px;
    e.py = py;
    e.lim2 = fmin(rad2, ub2);
    e.off = kd.off;
    e.col = kd.lr + (size_t)jx * Ny;
    e.g8 = kd.gap8 + (size_t)jx * ((Ny + 7) >> 3);
    e.b0 = r0 >> 3;
    e.prune_ok = fabs(px - X0[jx]) <= hx;                          // false only for agents outside the room
    e.jx = jx;
    e.pad = 0;
    return true;
}

// wall force from the selected cell, float64 (pedestrians.py:317-334); zero if no cell was selected.
// R = 0 (agent exactly on a wall node) gives NaN in the reference; here the force is 0 (deviation D1).
__device__ __forceinline__ double2 wall_force_of(const GcfmParams& P, long long flat, double px, double py, double vxd,
                                                 double vyd, double vdes, const Axis X0,
                                                 const Axis Y0) {
    if (flat < 0) return make_double2(0.0, 0.0);
    int r = (int)(flat / P.Nx), cc = (int)(flat % P.Nx);
    double Rx = X0[cc] - px, Ry = Y0[r] - py;
    double rr = sqrt(Rx * Rx + Ry * Ry);
    if (!(rr > 0.0)) return make_double2(0.0, 0.0);     // D1
    double ex = Rx / rr, ey = Ry / rr;
    double d = vxd * ex + vyd * ey;
    double vrel = 0.5 * (d + fabs(d));
    double s = sqrt(vxd * vxd + vyd * vyd);
    double hx = 1.0, hy = 0.0;
    if (s > 0) {
        hx = vxd / s;
        hy = vyd / s;
    }
    double a = P.a_min + P.tau_a * s;
    double b = P.b_max - (P.b_max - P.b_min) * fmin(s / vdes, 1.0);
    double ca = (ex * hx + ey * hy) / a, sb = (hx * ey - hy * ex) / b;
    double q = 1.0 / (ca * ca + sb * sb);                 // Q12: no square root
    if (P.fix & OC_FIX_WALL_FORCE) q = sqrt(q);
    double dist = rr - q;
    double rep = fmin(exp(-dist / (P.eta_walls * (1.0 + vrel))), 1.0);
    if (P.fix & OC_FIX_WALL_FORCE) return make_double2(-rep * ex, -rep * ey);
    return make_double2(-3.0 * rep * Rx, -3.0 * rep * Ry);
}

// packed record of one agent
__device__ __forceinline__ void make_record(const GcfmParams& P, double vxd, double vyd, float4& pv, float2& ph) {
    double s = sqrt(vxd * vxd + vyd * vyd);
    double a = P.a_min + P.tau_a * s;
    pv = make_float4((float)vxd, (float)vyd, (float)s, (float)(1.0 / (a * a)));
    ph = (s > 0) ? make_float2((float)(vxd / s), (float)(vyd / s)) : make_float2(1.f, 0.f);
}

// full pack from the state arrays (after a re-sort): one wavefront per tile
__global__ void __launch_bounds__(64) k_pack(const double* __restrict__ x, const double* __restrict__ y,
                                             const double* __restrict__ vx, const double* __restrict__ vy,
                                             const double* __restrict__ v_des, const uint8_t* __restrict__ status,
                                             const int32_t* __restrict__ key_of, const int32_t* __restrict__ order,
                                             int N, GcfmParams P, Scratch S, const double* __restrict__ agent_time) {
    const Snap& C = S.snap;
    int k = blockIdx.x * 64 + threadIdx.x;
    int i = (k < N) ? order[k] : -1;
    bool act = (i >= 0) && status[i];
    double px = 1e300, py = 1e300, vxd = 0.0, vyd = 0.0;
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 ph = make_float2(1.f, 0.f);
    PDK dk = {1.0, 0, 0};
    if (act) {
        px = x[i];
        py = y[i];
        vxd = vx[i];
        vyd = vy[i];
        make_record(P, vxd, vyd, pv, ph);
        dk.v_des = v_des[i];
        dk.key = key_of[i];
    }
    C.ppos[k] = make_double2(px, py);
    C.pvel[k] = pv;
    C.phead[k] = ph;
    C.pidx[k] = act ? i : -1;
    // inactive agents sort to the end, so the active ones occupy slots [0, n): the last of them publishes n
    {
        const int inext = (k + 1 < N) ? order[k + 1] : -1;
        const bool act_next = (inext >= 0) && status[inext];
        if (act && !act_next) S.ctrl[C_NACTIVE] = k + 1;
    }
    S.pvd[k] = make_double2(vxd, vyd);
    S.pdk[k] = dk;
    S.ptime[k] = (act && agent_time != nullptr) ? agent_time[i] : 0.0;
    const double4 tb = tile_boxes(act ? px : INFINITY, act ? py : INFINITY, act ? px : -INFINITY, act ? py : -INFINITY,
                                  threadIdx.x, blockIdx.x, C.qbox);
    if (threadIdx.x == 0) {
        C.tile_box[blockIdx.x] = tb;
        S.tile_box0[blockIdx.x] = tb;
    }
}

__device__ __forceinline__ bool box_hit(const double4& a, const double4& b, double c) {
    return !(b.x > a.z + c || b.z < a.x - c || b.y > a.w + c || b.w < a.y - c);
}

// neighbour-tile lists from the sort-time boxes: one wavefront per tile
__global__ void __launch_bounds__(64) k_tile_lists(GcfmParams P, Scratch S, int have_work) {
    const int it = blockIdx.x, lane = threadIdx.x;
    const double4 bi = S.tile_box0[it];
    const double c = P.cutoff + 2.0 * P.drift;
    int32_t* out = S.nb_list + (size_t)it * (NB_MAX + 1);
    int cnt = 0;
    for (int tb = 0; tb < S.ntiles; tb += 64) {
        int t = tb + lane;
        bool hit = (t < S.ntiles) && box_hit(bi, S.tile_box0[t], c);
        unsigned long long m = __ballot(hit);
        int pos = cnt + __popcll(m & ((1ull << lane) - 1));
        if (hit && pos < NB_MAX) out[1 + pos] = t;
        cnt += __popcll(m);
    }
    if (lane == 0) {
        out[0] = (cnt <= NB_MAX) ? cnt : -1;
        // sort key for the dispatch order (ascending sort -> decreasing work): the tile's measured work of the last
        // step (same place along the curve, nearly the same agents), or its neighbour count before any step ran
        const int est = have_work ? S.tile_work[it] : cnt;
        S.keys_a[it] = 0x00FFFFFFu - (uint32_t)min(max(est, 0), 0x00FFFFFF);
        S.vals_a[it] = it;
    }
}

// ---------------------------------------------------------------------------------- pair forces
#define COS_VISION (-0.58778525229247312917f)   // cos(0.7 pi), pedestrians.py:262

// inclusive prefix sum over the 64 lanes by DPP row shifts and row broadcasts: six VALU-latency steps where six
// ds_bpermute round trips (__shfl_up) sat in the middle of every virtual tile's chain
__device__ __forceinline__ int wave_incl_scan(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);      // row_shr:1  (a row = 16 lanes; out of row: 0)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);      // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return x;
}

typedef float f2v __attribute__((ext_vector_type(2)));

// force of j on i (pedestrians.py:242-280) in trig-free form: cos(alpha-beta) = e.h, sin = h x e.
// float32 with hardware rcp/rsq/exp2 (1 ulp); R is the float64-differenced relative vector rounded to float32.
// TWO pairs (i, jA), (i, jB) of the same agent i per call, one in each half of packed float2 registers: a plain
// float32 VALU instruction occupies the SIMD for 4 cycles per wavefront, v_pk_fma/mul/add_f32 do two lanes' worth in
// the same 4 -- the kernel is VALU-issue-bound (PMC: VALU active 2/3 of the time), so packing is what pays.
__device__ __forceinline__ void pair_term2(f2v Rx, f2v Ry, float vxi, float vyi, float si, float hxi, float hyi,
                                           float ia2_i, float ib2_i, float inv_vdes_i, f2v vjx, f2v vjy, f2v sj,
                                           f2v ia2_j, f2v hjx, f2v hjy, float b_max, float db,
                                           float neg_inv_eta_log2e, bool unit_dir, f2v& fx, f2v& fy, bool okA = true,
                                           bool okB = true) {
    const f2v one = {1.f, 1.f}, zero = {0.f, 0.f};
    const f2v r2 = __builtin_elementwise_fma(Rx, Rx, Ry * Ry);
    const f2v inv_r = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
    const f2v r = r2 * inv_r, ex = Rx * inv_r, ey = Ry * inv_r;
    // closing speed (pedestrians.py:254-255)
    const f2v vxi2 = {vxi, vxi}, vyi2 = {vyi, vyi};
    const f2v vrel = __builtin_elementwise_max(__builtin_elementwise_fma(vxi2 - vjx, ex, (vyi2 - vjy) * ey), zero);
    // ellipse radii along the centre line (pedestrians.py:266-271)
    const f2v hxi2 = {hxi, hxi}, hyi2 = {hyi, hyi};
    const f2v ci = __builtin_elementwise_fma(ex, hxi2, ey * hyi2);
    const f2v sn_i = __builtin_elementwise_fma(hxi2, ey, -(hyi2 * ex));
    // vision factor (pedestrians.py:259-262): v_i.e/|v_i| = h_i.e when |v_i| > 0
    const f2v cv = {COS_VISION, COS_VISION};
    const float kscale = (si > 0.f) ? (1.f / (1.f - COS_VISION)) : 0.f;
    const f2v kk = __builtin_elementwise_max(ci - cv, zero) * (f2v){kscale, kscale};
    const f2v ai = __builtin_elementwise_fma(ci * ci, (f2v){ia2_i, ia2_i}, sn_i * sn_i * (f2v){ib2_i, ib2_i});
    const f2v q_i = {__builtin_amdgcn_rsqf(ai.x), __builtin_amdgcn_rsqf(ai.y)};
    const f2v b_j = __builtin_elementwise_fma((f2v){-db, -db},
                                              __builtin_elementwise_min(sj * (f2v){inv_vdes_i, inv_vdes_i}, one),
                                              (f2v){b_max, b_max});                      // Q6: i's v_des
    const f2v cj = __builtin_elementwise_fma(ex, hjx, ey * hjy);
    const f2v sn_j = __builtin_elementwise_fma(hjx, ey, -(hjy * ex));
    // q_j = 1 / sqrt(c^2/a_j^2 + s^2/b_j^2) = b_j / sqrt(c^2 (b_j/a_j)^2 + s^2): no reciprocal of b_j^2 (a quarter-rate
    // instruction per pair; b_j depends on the PAIR through i's v_des, quirk Q6, so it cannot be tabulated per agent)
    const f2v aj = __builtin_elementwise_fma(cj * cj, (b_j * b_j) * ia2_j, sn_j * sn_j);
    const f2v q_j = b_j * (f2v){__builtin_amdgcn_rsqf(aj.x), __builtin_amdgcn_rsqf(aj.y)};
    const f2v dist = r - q_i - q_j;
    const f2v den = one + vrel;
    const f2v arg = dist * (f2v){neg_inv_eta_log2e, neg_inv_eta_log2e} *
                    (f2v){__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const f2v ex2 = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
    f2v rep = __builtin_elementwise_min(kk * ex2, one);
    // a pair that does not count (beyond the cutoff, coincident, the padding half of an odd entry) is switched off HERE,
    // one select per pair instead of one per force component and word of its float64 image: rep = 0 makes both components
    // (-0 * R, R finite) zero
    rep.x = okA ? rep.x : 0.f;
    rep.y = okB ? rep.y : 0.f;
    if (unit_dir) {                     // OC_FIX_PAIR_FORCE (wave-uniform): along the unit vector
        fx = -(rep * ex);
        fy = -(rep * ey);
        // (coincident agents: e = 0 * inf = NaN, and 0 * NaN is NaN -- this variant selects on the components too)
        fx.x = okA ? fx.x : 0.f; fx.y = okB ? fx.y : 0.f;
        fy.x = okA ? fy.x : 0.f; fy.y = okB ? fy.y : 0.f;
    } else {                            // the reference scales the distance vector (pedestrians.py:280)
        fx = -(rep * Rx);
        fy = -(rep * Ry);
    }
}

#ifndef OC_PAIR_WALK
#define OC_PAIR_WALK 1          // phase 2 walks the candidate masks (pair_share); 0: round 3's compacted candidate list (A/B builds)
#endif
#define PAIR_LIST_CAP 1024      // list entries (two candidates of one agent each): 64 agents x 32 j always fit
#ifndef OC_FORCES_OCC
#define OC_FORCES_OCC 4         // k_forces: wavefronts per SIMD the register budget is set for (A/B builds)
#endif
#define PAIR_UNIT_CAP 132       // units of a virtual tile (64 lanes x 2 mask words) + the end marker
#define PAIR_SPLIT_MAX 4

struct JRec {                   // one lane's share of a staged j-tile (prefetched into registers)
    double2 pos;
    float4 vel;
    float2 head;
};

// LDS of a block of 4 wavefronts: 4/NS tiles (i records and force sums, shared by the tile's NS wavefronts) and,
// per wavefront, its share of the staged j-tile and its candidate list.
template <int NS>
struct PairLds {
    static constexpr int NJ = 64 / NS, TPB = 4 / NS;
    double2 li_pos[TPB][64];
    float4 li_vel[TPB][64];
    float4 li_aux[TPB][64];     // (hx, hy, 1/b_i^2, 1/v_des_i)
    double accx[TPB][64], accy[TPB][64];
#if !OC_PAIR_WALK
    double2 l_pos[4][64];       // per wavefront: the staged virtual j-tile (64 records)
    float4 l_vel[4][64];
    float2 l_head[4][64];
#endif
    alignas(16) float l_rx[4][64];
    alignas(16) float l_ry[4][64];
#if OC_PAIR_WALK
    uint2 l_unit[4][PAIR_UNIT_CAP];     // per wavefront: the virtual tile's units (mask word, running entry count << 8 | lane << 1 | word)
    uint8_t l_start[4][64];             // ... and per lane the unit its share of the entries starts in
    // the staged records once more, field by field: phase 2 reads the fields of TWO records (the entry's candidates) into the
    // two halves of packed operands -- from an array of structures every field cost a register move on top of the wide load
    // (30 moves per trip of 156 vector instructions), and two records' wide loads collide in the banks; field arrays give
    // conflict-free 4-byte loads straight into the register pairs
    double l_px[4][64], l_py[4][64];
    float l_f[4][6][64];                // vx, vy, |v|, 1/a^2, hx, hy
#else
    uint32_t list[4][PAIR_LIST_CAP];
#endif
};

// the i-tile in LDS (in phase 2 any lane works on any agent) and zeroed force sums: one wavefront per tile
template <int NS>
__device__ __forceinline__ void pair_load_tile(const GcfmParams& P, const Snap& C, const PDK* __restrict__ pdk,
                                               PairLds<NS>& L, int it, int tl, int lane) {
    const int k = it * 64 + lane;
    const int i = C.pidx[k];
    const float4 vi = C.pvel[k];
    const float2 hi = C.phead[k];
    float inv_vdes = 0.f, ib2_i = 0.f;
    if (i >= 0) {
        const double vd = pdk[k].v_des;
        inv_vdes = (float)(1.0 / vd);
        const double b = P.b_max - (P.b_max - P.b_min) * fmin((double)vi.z / vd, 1.0);
        ib2_i = (float)(1.0 / (b * b));
    }
    L.li_pos[tl][lane] = C.ppos[k];
    L.li_vel[tl][lane] = vi;
    L.li_aux[tl][lane] = make_float4(hi.x, hi.y, ib2_i, inv_vdes);
    L.accx[tl][lane] = 0.0;
    L.accy[tl][lane] = 0.0;
}


// One WAVEFRONT per (i-tile, share).  The NS wavefronts of a tile sit in one block and share the tile's i records
// and force sums in LDS; share `sub` owns the records j = sub*64/NS .. (sub+1)*64/NS - 1 of EVERY candidate j-tile, so
// the shares of a tile carry equal work whatever the geometry.  Splitting exists for occupancy and balance: a tile's
// work is one long chain, the chip holds only ~1.5 tiles per SIMD, and the launch ends with its slowest wavefront.
// A share gathers its 64/NS records of NS candidate tiles at a time into one VIRTUAL j-tile of 64 records, so the
// per-tile overheads (staging wait, prefix scan, barriers, short chunks) are paid once per 64 records whatever NS.
// Per virtual j-tile:
//   stage    its 64 packed records go to wave-private LDS; the next one's records are already being loaded
//            (register prefetch) while this one is processed;
//   phase 1  per lane (= agent i) a 64-bit candidate mask from a conservative float32 distance test on box-relative
//            coordinates: packed float2 math, sign bits collected with v_alignbit (3 VALU per j);
//   compact  candidates are appended, two per entry, to a wave-private LDS list, lane-major (sorted by agent); dense
//            tile pairs (more than PAIR_LIST_CAP entries) are taken in two rounds of 32 records;
//   phase 2  every lane walks a contiguous chunk of the list: exact float64 cutoff decision, packed float32 force,
//            float64 LDS atomic accumulation when the agent changes.
// The caller brackets this with block barriers (i records loaded before, sums complete after).
template <int NS>
__device__ __forceinline__ void pair_share(const GcfmParams& P, const Scratch& S, const Snap& C, PairLds<NS>& L, int it,
                                           int tl, int sub, int w, int lane) {
    constexpr int NJ = 64 / NS;                   // records of a real tile owned by this share
    constexpr int NW = 2;                         // 32-bit mask words per lane (64 records)
    constexpr int WB = 32;
    [[maybe_unused]] constexpr int NROUND = 2;                     // rounds of 32 records for a dense virtual tile
    const float b_max = (float)P.b_max, db = (float)(P.b_max - P.b_min);
    const float neg_inv_eta_log2e = (float)(-1.4426950408889634 / P.eta);
    const bool unit_dir = (P.fix & OC_FIX_PAIR_FORCE) != 0;
    const double c = P.cutoff, c2 = c * c;
    const float c2lo = (float)(c2 * (1.0 - 2e-6)), c2hi = (float)(c2 * (1.0 + 2e-6));
    const int k = it * 64 + lane;
    const int i = C.pidx[k];
    const double2 pi = L.li_pos[tl][lane];
    const double4 bi = C.tile_box[it];
    const double ox = bi.x, oy = bi.y;             // origin of the box-relative float32 coordinates
    const float ext_i = (float)((bi.z - bi.x) + (bi.w - bi.y) + 2.0 * P.cutoff);
    const float xi_f = (float)(pi.x - ox), yi_f = (float)(pi.y - oy);
    const f2v xi2 = {xi_f, xi_f}, yi2 = {yi_f, yi_f};
    const float my_speed = L.li_vel[tl][lane].z;
#if !OC_PAIR_WALK
    uint32_t* lst = L.list[w];
#endif
    int work = 0;                                   // list entries processed: the tile's measured work
    const int j0 = sub * NJ;                        // this share's records of a j-tile: j0 .. j0 + NJ - 1
    const int jl = lane & (NJ - 1);                 // lanes >= NJ duplicate the loads of lanes < NJ (same cache lines)

    // candidate tiles: the neighbour list (if valid) or every tile, filtered by the live boxes
    const int32_t* nb = S.nb_list + (size_t)it * (NB_MAX + 1);
    const int nb_cnt = nb[0];
    const bool use_list = (nb_cnt >= 0) && (S.ctrl[C_INVALID] == 0);
    const int nchunk = use_list ? 1 : (S.ntiles + 63) / 64;
    for (int ch = 0; ch < nchunk; ++ch) {
        int t;
        if (use_list) t = (lane < nb_cnt) ? nb[1 + lane] : -1;
        else t = (ch * 64 + lane < S.ntiles) ? ch * 64 + lane : -1;
        // this share only ever reads quarter(s) `sub` of a candidate tile: test that piece's own box (on a crowd at
        // 2.4 agents/m^2 the quarters of the ~12.5 tiles within reach add up to ~9 tiles' worth of records)
        bool hit = false;
        if (t >= 0) {
            if (NS == 4) hit = box_hit(bi, C.qbox[t * 4 + sub], c);
            else if (NS == 2) hit = box_hit(bi, C.qbox[t * 4 + 2 * sub], c) || box_hit(bi, C.qbox[t * 4 + 2 * sub + 1], c);
            else hit = box_hit(bi, C.tile_box[t], c);
        }
        unsigned long long tmask = __ballot(hit);
        if (!tmask) continue;
        // next virtual tile: the NS lowest remaining hits; lane l fetches record j0 + l % NJ of hit l / NJ
        int jts[NS];
        JRec cur;
        float ext = 0.f;
        auto fetch = [&]() {
            ext = 0.f;
            int jt_l = -1;
#pragma unroll
            for (int b = 0; b < NS; ++b) {
                jts[b] = -1;
                if (tmask) {
                    jts[b] = __builtin_amdgcn_readlane(t, __ffsll((long long)tmask) - 1);
                    tmask &= tmask - 1;
                    // conservative float32 bound: rounding of the relative coordinates scales with their size, and a staged
                    // record lies within the cutoff of the i box, inside its tile's box: |x_j - ox| + |y_j - oy| is at most
                    // the i box's extent + 2 cutoff + its own box's extent (3 float64 operations per tile; the exact extent
                    // of the two boxes' union took 14)
                    const double4 bj = C.tile_box[jts[b]];
                    ext = fmaxf(ext, (float)((bj.z - bj.x) + (bj.w - bj.y)));
                }
                if (lane / NJ == b) jt_l = jts[b];
            }
            cur.pos = make_double2(1e300, 1e300);
            cur.vel = make_float4(0.f, 0.f, 0.f, 0.f);
            cur.head = make_float2(1.f, 0.f);
            if (jt_l >= 0) {
                cur.pos = C.ppos[jt_l * 64 + j0 + jl];
                cur.vel = C.pvel[jt_l * 64 + j0 + jl];
                cur.head = C.phead[jt_l * 64 + j0 + jl];
            }
        };
        fetch();
        for (;;) {
            // ---- stage the virtual tile (registers -> wave-private LDS; LDS ops of one wavefront complete in order,
            // so the single buffer is safe: the previous tile's reads were issued before these writes)
#if OC_PAIR_WALK
            L.l_px[w][lane] = cur.pos.x;
            L.l_py[w][lane] = cur.pos.y;
            L.l_f[w][0][lane] = cur.vel.x;
            L.l_f[w][1][lane] = cur.vel.y;
            L.l_f[w][2][lane] = cur.vel.z;
            L.l_f[w][3][lane] = cur.vel.w;
            L.l_f[w][4][lane] = cur.head.x;
            L.l_f[w][5][lane] = cur.head.y;
#else
            L.l_pos[w][lane] = cur.pos;
            L.l_vel[w][lane] = cur.vel;
            L.l_head[w][lane] = cur.head;
#endif
            L.l_rx[w][lane] = (float)(cur.pos.x - ox);
            L.l_ry[w][lane] = (float)(cur.pos.y - oy);
            const float extt = (ext + ext_i) * 1.0001f;
            const float c2f = (float)c2 * 1.00001f + 1e-6f * extt * (extt + (float)c);
            const f2v nc2 = {-c2f, -c2f};
            int bself = -1;                                 // position of the i-tile itself in this virtual tile
            int nfill = 0;                                  // real tiles in it (the last one of a chunk is partly padding)
#pragma unroll
            for (int b = 0; b < NS; ++b) {
                bself = (jts[b] == it) ? b : bself;
                nfill += (jts[b] >= 0) ? 1 : 0;
            }
            // ---- start loading the next virtual tile while this one is processed
            const bool more = tmask != 0ull;
            if (more) fetch();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- phase 1: sign(d^2 - c2f) of records 0..63; record 32*h + q sits at bit 31 - q of word h
            uint32_t mw[NW];
#pragma unroll
            for (int h = 0; h < NW; ++h) {
                uint32_t acc = 0u;
                // the records of real tile b are 16*(4/NS)*b ..: pieces of the word that hold only padding are skipped
                constexpr int PW = (NJ < WB) ? WB / NJ : 1;         // tile pieces per mask word
                constexpr int GP = WB / 4 / PW;                     // groups of 4 records per piece
#pragma unroll
                for (int pc = 0; pc < PW; ++pc) {
                    if ((32 * h + 4 * GP * pc) / NJ < nfill) {
#pragma unroll
                        for (int gg = 0; gg < GP; ++gg) {
                            const int g = pc * GP + gg;
                            const float4 xs = *reinterpret_cast<const float4*>(&L.l_rx[w][32 * h + 4 * g]);
                            const float4 ys = *reinterpret_cast<const float4*>(&L.l_ry[w][32 * h + 4 * g]);
                            const f2v xa = {xs.x, xs.y}, xb = {xs.z, xs.w}, ya = {ys.x, ys.y}, yb = {ys.z, ys.w};
                            const f2v dxa = xa - xi2, dxb = xb - xi2, dya = ya - yi2, dyb = yb - yi2;
                            const f2v ta = __builtin_elementwise_fma(dxa, dxa, __builtin_elementwise_fma(dya, dya, nc2));
                            const f2v tb = __builtin_elementwise_fma(dxb, dxb, __builtin_elementwise_fma(dyb, dyb, nc2));
                            acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(ta.x), 31);
                            acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(ta.y), 31);
                            acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(tb.x), 31);
                            acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(tb.y), 31);
                        }
                    } else {
                        acc <<= (4 * GP) & 31;                      // (PW == 1: the whole word, left as zero)
                        if (PW == 1) acc = 0u;
                    }
                }
                mw[h] = acc << (32 - WB);
            }
            if (bself >= 0 && (lane / NJ) == sub) {                                           // j != i
                const int rs = bself * NJ + jl;
                mw[rs >> 5] &= ~(0x80000000u >> (rs & 31));
            }
            if (i < 0 || !(my_speed > 0.f)) {           // no agent, or a stationary one: k = 0 for every j (Q19)
#pragma unroll
                for (int h = 0; h < NW; ++h) mw[h] = 0u;
            }
#if OC_PAIR_WALK
            // ---- phase 2 without a candidate list: the lanes WALK the masks.  (Round 3: the masks were turned into an LDS list of
            // entries, every lane appending its own candidates -- a loop as long as the BUSIEST lane's count, 424 vector
            // instructions per virtual tile of which 87 wrote entries: a fifth of the launch's instructions.)  A UNIT is one
            // lane's mask word that has candidates; as before two candidates of a word make an entry (an odd last one is paired
            // with itself, masked).  The units go to a wave-private table in lane order -- sorted by agent -- with the running
            // entry count; every lane takes the same number R of consecutive ENTRIES: a binary search in the running counts
            // finds the unit its share starts in, the entries of that unit that belong to the lane before it are dropped from
            // the mask, and from there the lane takes two candidates off the current unit's mask per trip and moves on to the
            // next unit when the mask is empty (fetched one unit ahead).  Same pairs, same agent order, same float64 sums.
            {
                uint32_t m0 = mw[0], m1 = mw[1];
                if (OC_DIAG(P, 4u)) m0 = m1 = 0u;
                const int c0 = (__popc(m0) + 1) >> 1, c1 = (__popc(m1) + 1) >> 1, cntl = c0 + c1;
                const int incl = wave_incl_scan(cntl);
                const int total = __builtin_amdgcn_readlane(incl, 63);
                if (total != 0) {
                    work += total;
                    uint2* const ut = L.l_unit[w];
                    const unsigned long long b0 = __ballot(m0 != 0u), b1 = __ballot(m1 != 0u);
                    const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
                                      (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
                    const int nunit = __popcll(b0) + __popcll(b1);
                    // unit: (mask, running entry count incl. this unit << 8 | lane << 1 | word)
                    if (m0 != 0u) ut[below] = make_uint2(m0, ((uint32_t)(incl - c1) << 8) | ((uint32_t)lane << 1));
                    if (m1 != 0u) ut[below + (m0 != 0u ? 1 : 0)] = make_uint2(m1, ((uint32_t)incl << 8) | ((uint32_t)lane << 1) | 1u);
                    if (lane == 0) ut[nunit] = make_uint2(0u, (uint32_t)total << 8);      // (read ahead by the last unit's walker)
                    const int R = (total + 63) >> 6;
                    const int e_beg = lane * R, e_end = min(e_beg + R, total);
                    // Where does lane L's share (entries L R ...) start?  The unit that holds entry L R tells L: a unit with
                    // entries [a, b) is the start of the lanes ceil(a / R) .. ceil(b / R) - 1 (at most 17 of them, 2 - 4 as a
                    // rule) -- a short scatter by the units' owners where a binary search in the running counts cost every
                    // lane seven dependent LDS reads.  (x / R for x <= 2111, R <= 32 as (x m) >> 16, m = ceil(2^16 / R): exact.)
                    {
                        const uint32_t mg = (65536u + (uint32_t)R - 1u) / (uint32_t)R;         // (wave-uniform)
                        uint8_t* const stt = L.l_start[w];
                        const int a0 = incl - cntl, a1 = incl - c1, a2 = incl;      // word 0: [a0, a1), word 1: [a1, a2)
                        const int f0 = (int)(((uint32_t)(a0 + R - 1) * mg) >> 16), f1 = (int)(((uint32_t)(a1 + R - 1) * mg) >> 16),
                                  f2 = (int)(((uint32_t)(a2 + R - 1) * mg) >> 16);
                        for (int q = f0; q < min(f1, 64); ++q) stt[q] = (uint8_t)below;
                        const int below1 = below + (m0 != 0u ? 1 : 0);
                        for (int q = f1; q < min(f2, 64); ++q) stt[q] = (uint8_t)below1;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    int u = e_beg < total ? (int)L.l_start[w][lane] : nunit;
                    uint2 cur = ut[min(u, nunit)];
                    uint2 nxt = ut[min(u + 1, nunit)];
                    uint32_t m = e_beg < total ? cur.x : 0u;
                    uint32_t id = cur.y & 255u;
                    {   // the unit's entries that lie before the share: two candidates each, off the mask's top
                        int k = e_beg < total ? e_beg - ((int)(cur.y >> 8) - ((__popc(m) + 1) >> 1)) : 0;
                        while (k > 0) {
                            m &= ~(0x80000000u >> __clz((int)m));
                            m &= ~(0x80000000u >> __clz((int)m));
                            --k;
                        }
                    }
                    int il_cur = -1;
                    double2 pa = make_double2(0.0, 0.0);
                    float4 va = make_float4(0.f, 0.f, 0.f, 0.f), aux = va;
                    // the share's sums are float64: every pair's float32 force is added exactly, so the result does not
                    // depend on how the launch shape (pair_split, share boundaries) groups the pairs
                    double sfx = 0.0, sfy = 0.0;
                    for (int tt = e_beg; tt < (OC_DIAG(P, 1u) ? e_beg : e_end); ++tt) {
                        if (m == 0u) {                          // the unit is used up: the next one (non-empty by construction)
                            m = nxt.x;
                            id = nxt.y & 255u;
                            ++u;
                            nxt = ut[min(u + 1, nunit)];
                        }
                        const int il = (int)(id >> 1), jb = (int)(id & 1u) << 5;
                        const int q1 = __clz((int)m);
                        m &= ~(0x80000000u >> q1);
                        const bool two = m != 0u;
                        const int q2 = two ? __clz((int)m) : q1;
                        m &= ~(0x80000000u >> q2);
                        const int jA = jb + q1, jB = jb + q2;
                        const double2 pbA = make_double2(L.l_px[w][jA], L.l_py[w][jA]), pbB = make_double2(L.l_px[w][jB], L.l_py[w][jB]);
                        const float4 vbA = make_float4(L.l_f[w][0][jA], L.l_f[w][1][jA], L.l_f[w][2][jA], L.l_f[w][3][jA]),
                                     vbB = make_float4(L.l_f[w][0][jB], L.l_f[w][1][jB], L.l_f[w][2][jB], L.l_f[w][3][jB]);
                        const float2 hbA = make_float2(L.l_f[w][4][jA], L.l_f[w][5][jA]), hbB = make_float2(L.l_f[w][4][jB], L.l_f[w][5][jB]);
                        if (il != il_cur) {
                            if (il_cur >= 0) {
                                atomicAdd(&L.accx[tl][il_cur], sfx);
                                atomicAdd(&L.accy[tl][il_cur], sfy);
                            }
                            il_cur = il;
                            sfx = sfy = 0.0;
                            pa = L.li_pos[tl][il];
                            va = L.li_vel[tl][il];
                            aux = L.li_aux[tl][il];
                        }
                        const double dxA = pbA.x - pa.x, dyA = pbA.y - pa.y, dxB = pbB.x - pa.x, dyB = pbB.y - pa.y;
                        // the reference's decision sqrt(d2) < cutoff (pedestrians.py:352-356, simulations.py:291) from the
                        // float32 square of the float64-differenced vector wherever that cannot be wrong (it is good to
                        // 3e-7 relative); the exact float64 comparison decides inside the band.  Coincident agents (d2 = 0:
                        // NaN in the reference) are skipped (deviation D1).
                        const f2v Rx = {(float)dxA, (float)dxB}, Ry = {(float)dyA, (float)dyB};
                        const f2v r2f = __builtin_elementwise_fma(Rx, Rx, Ry * Ry);
                        bool okA = r2f.x < c2lo, okB = r2f.y < c2lo;
                        if ((!okA && r2f.x < c2hi) || (!okB && r2f.y < c2hi)) {         // (rare: |d - cutoff| < 4e-6 m)
                            okA = okA || dxA * dxA + dyA * dyA <= P.cut_d2max;
                            okB = okB || dxB * dxB + dyB * dyB <= P.cut_d2max;
                        }
                        okA = okA && r2f.x > 0.f;
                        okB = okB && two && r2f.y > 0.f;
                        if (okA || okB) {
                            f2v fx, fy;
                            pair_term2(Rx, Ry, va.x, va.y, va.z, aux.x,
                                       aux.y, va.w, aux.z, aux.w, (f2v){vbA.x, vbB.x}, (f2v){vbA.y, vbB.y},
                                       (f2v){vbA.z, vbB.z}, (f2v){vbA.w, vbB.w}, (f2v){hbA.x, hbB.x}, (f2v){hbA.y, hbB.y},
                                       b_max, db, neg_inv_eta_log2e, unit_dir, fx, fy, okA, okB);
                            sfx += (double)fx.x + (double)fx.y;
                            sfy += (double)fy.x + (double)fy.y;
                        }
                    }
                    if (il_cur >= 0) {
                        atomicAdd(&L.accx[tl][il_cur], sfx);
                        atomicAdd(&L.accy[tl][il_cur], sfy);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
#else
            // list entries hold TWO candidates of one agent (an odd last one is paired with itself, masked).  Dense
            // tile pairs do not fit the list: rounds of 32 j each (64 agents x 16 entries = PAIR_LIST_CAP always fit)
            // (whether the rounds are needed is known from the first pass's own prefix sum: no separate reduction)
            int nround = 1;
            if (OC_DIAG(P, 4u)) nround = 0;
            for (int rd = 0; rd < nround; ++rd) {
                uint32_t m[NW];
#pragma unroll
                for (int h = 0; h < NW; ++h) m[h] = (nround == 1 || rd == h) ? mw[h] : 0u;
                // ---- compact, lane-major: exclusive prefix sum of the lanes' entry counts, every lane appends
                int cntl = 0;           // candidates are paired inside a mask word: ceil(popc / 2) entries per word
#pragma unroll
                for (int h = 0; h < NW; ++h) cntl += (__popc(m[h]) + 1) >> 1;
                const int incl = wave_incl_scan(cntl);
                const int total = __builtin_amdgcn_readlane(incl, 63);
                if (NROUND > 1 && nround == 1 && total > PAIR_LIST_CAP) {       // a dense tile pair: again, in rounds
                    nround = NROUND;
                    rd = -1;
                    continue;
                }
                if (total == 0) continue;
                work += total;
                {   // entry: agent lane << 16 | B valid << 14 | jB << 7 | jA.  Two candidates of the same mask word per loop
                    // trip (an odd last one is paired with itself, B invalid): no carry between trips, 13 VALU per entry
                    // instead of 22 for two single-bit trips; at most one entry more per lane than pairing across the words
                    int off = incl - cntl;
                    const uint32_t ebase = (uint32_t)lane << 16;
#pragma unroll
                    for (int h = 0; h < NW; ++h) {
                        uint32_t mm = m[h];
                        while (mm) {
                            const int q1 = __clz((int)mm);
                            mm &= ~(0x80000000u >> q1);
                            const bool two = mm != 0u;
                            const int q2 = two ? __clz((int)mm) : q1;
                            mm &= ~(0x80000000u >> q2);
                            lst[off++] = ebase | (two ? (1u << 14) : 0u) | ((uint32_t)(32 * h + q2) << 7) | (uint32_t)(32 * h + q1);
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // ---- phase 2: every lane takes a contiguous chunk of R = ceil(total/64) list entries.  The list is
                // sorted by agent, so a chunk touches 1-3 agents: their data stays in registers and the forces are
                // summed in registers; LDS is updated (float64 atomic) only when the agent changes.
                const int R = (total + 63) >> 6;
                const int e_beg = lane * R, e_end = min(e_beg + R, total);
                int il_cur = -1;
                double2 pa = make_double2(0.0, 0.0);
                float4 va = make_float4(0.f, 0.f, 0.f, 0.f), aux = va;
                // the chunk's sums are float64: every pair's float32 force is added exactly, so the result does not
                // depend on how the launch shape (pair_split, chunk boundaries) groups the pairs
                double sfx = 0.0, sfy = 0.0;
                // (the NEXT entry's word is fetched one trip ahead: list word -> record addresses -> records are two dependent
                // LDS round trips per entry otherwise; a word past the chunk's end is read and dropped -- the list has room)
                uint32_t e_nxt = lst[min(e_beg, PAIR_LIST_CAP - 1)];
                for (int tt = e_beg; tt < (OC_DIAG(P, 1u) ? e_beg : e_end); ++tt) {
                    const uint32_t e = e_nxt;
                    e_nxt = lst[min(tt + 1, PAIR_LIST_CAP - 1)];
                    const int il = e >> 16, jA = e & 127, jB = (e >> 7) & 127;
                    const double2 pbA = L.l_pos[w][jA], pbB = L.l_pos[w][jB];
                    const float4 vbA = L.l_vel[w][jA], vbB = L.l_vel[w][jB];
                    const float2 hbA = L.l_head[w][jA], hbB = L.l_head[w][jB];
                    if (il != il_cur) {
                        if (il_cur >= 0) {
                            atomicAdd(&L.accx[tl][il_cur], sfx);
                            atomicAdd(&L.accy[tl][il_cur], sfy);
                        }
                        il_cur = il;
                        sfx = sfy = 0.0;
                        pa = L.li_pos[tl][il];
                        va = L.li_vel[tl][il];
                        aux = L.li_aux[tl][il];
                    }
                    const double dxA = pbA.x - pa.x, dyA = pbA.y - pa.y, dxB = pbB.x - pa.x, dyB = pbB.y - pa.y;
                    // the reference's decision sqrt(d2) < cutoff (pedestrians.py:352-356, simulations.py:291) from the
                    // float32 square of the float64-differenced vector wherever that cannot be wrong (it is good to
                    // 3e-7 relative); the exact float64 comparison decides inside the band.  Coincident agents (d2 = 0:
                    // NaN in the reference) are skipped (deviation D1).
                    const f2v Rx = {(float)dxA, (float)dxB}, Ry = {(float)dyA, (float)dyB};
                    const f2v r2f = __builtin_elementwise_fma(Rx, Rx, Ry * Ry);
                    bool okA = r2f.x < c2lo, okB = r2f.y < c2lo;
                    if ((!okA && r2f.x < c2hi) || (!okB && r2f.y < c2hi)) {         // (rare: |d - cutoff| < 4e-6 m)
                        okA = okA || dxA * dxA + dyA * dyA <= P.cut_d2max;
                        okB = okB || dxB * dxB + dyB * dyB <= P.cut_d2max;
                    }
                    okA = okA && r2f.x > 0.f;
                    okB = okB && (e & (1u << 14)) && r2f.y > 0.f;
                    if (okA || okB) {
                        f2v fx, fy;
                        pair_term2(Rx, Ry, va.x, va.y, va.z, aux.x,
                                   aux.y, va.w, aux.z, aux.w, (f2v){vbA.x, vbB.x}, (f2v){vbA.y, vbB.y},
                                   (f2v){vbA.z, vbB.z}, (f2v){vbA.w, vbB.w}, (f2v){hbA.x, hbB.x}, (f2v){hbA.y, hbB.y},
                                   b_max, db, neg_inv_eta_log2e, unit_dir, fx, fy, okA, okB);
                        sfx += (double)fx.x + (double)fx.y;
                        sfy += (double)fy.x + (double)fy.y;
                    }
                }
                if (il_cur >= 0) {
                    atomicAdd(&L.accx[tl][il_cur], sfx);
                    atomicAdd(&L.accy[tl][il_cur], sfy);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
#endif
            if (!more) break;
        }
    }
    if (lane == 0) atomicAdd(&S.tile_work[it], work);
}

// ---------------------------------------------------------------------------------- nearest-wall-cell search
// One LANE per agent (the search is a chain of dependent loads, so a wavefront should carry as many agents as it
// can).  Rows are taken in aligned blocks of 8, outward from the agent's row, 8 blocks each way per round: the round
// loads the 16 column-gap bounds (gap8, built with the wall map) in one batch, marks the blocks in which some
// candidate can still win or tie -- lower bound from the vertical distance to the block's nearest row and the smallest
// column gap of its rows -- and then visits the marked blocks nearest first: 8 table entries in one batch of loads,
// pass 1 = smallest squared distance (no square roots), pass 2 = the reference's objective for the candidates inside
// the tie band.  An agent 12 m from a pillar marks 2-3 blocks in 2 rounds instead of walking ~240 rows.
// gap8 word of a block of 8 rows: bits 0..15 = the smallest j - (nearest site column <= j), bits 16..31 = the smallest
// (nearest site column >= j + 1) - j over the block's rows; 0xFFFF = no such site in any of them (gaps saturate at 0xFFFE:
// a smaller gap is still a lower bound)
#define GAP_SIDE_NONE 0xFFFF
#define GAP_NONE ((int)0xFFFFFFFFu)
// lower bound on the squared distance to any candidate of 8-row block `blk`
// (a BOUND, evaluated 16 times per round and agent: the rows' coordinates are taken as 8 blk h and 8 blk h + 7 h --
// within an ulp of the axis values, and past the last row for the room's last block -- and 1 nm is given away instead
// of selecting among exact axis values; the candidates themselves are measured on the exact axis).  Horizontally the bound
// is exact: the nearest candidate column on either side of the agent's column over the block's rows, measured like the
// candidates themselves (X0[c] - px on the axis values) -- with the gap taken as (gap - 1) cells, as before, a straight
// wall D away kept every block within sqrt(2 D hx) of the agent's row pending (3-4 visits at D = 7 m); now the blocks
// that really hold something within the incumbent's distance.
__device__ __forceinline__ double wall_block_bound(double px, double py, int blk, int gap, bool prune_ok, int jx, int Ny,
                                                   const Axis X0, const Axis Y0, double& dyb2) {
    double ylo, yhi;
    if (Y0.lin) {
        ylo = (double)(8 * blk) * Y0.h;
        yhi = ylo + 7.0 * Y0.h;
    } else {                                    // a caller's own axis: its values
        ylo = Y0.a[8 * blk];
        yhi = Y0.a[min(8 * blk + 7, Ny - 1)];
    }
    const double dyb = fmax(fmax(ylo - py, py - yhi) - 1e-9, 0.0);
    // every candidate of the block: |dy| >= dyb, |dx| >= the distance to the nearest candidate column of either side
    double lbh = 0.0;
    if (prune_ok) {
        const int gl = gap & 0xFFFF, gr = (int)((unsigned)gap >> 16);
        const double dl = (gl != GAP_SIDE_NONE) ? fabs(px - X0[jx - gl]) : INFINITY;
        const double dr = (gr != GAP_SIDE_NONE) ? fabs(X0[jx + gr] - px) : INFINITY;
        lbh = fmin(dl, dr);
    } else if (gap == GAP_NONE) {
        lbh = INFINITY;                         // no site in the block's rows at all
    }
    dyb2 = dyb * dyb;
    return dyb2 + lbh * lbh;
}

// pass 1: smallest squared distance among the block's candidates (no square roots)
__device__ __forceinline__ double wall_block_min(double px, double py, int blk, const int2 (&c)[8], int Nx, int Ny,
                                                 const Axis X0, const Axis Y0, double thr, double lim2) {
    const int rlo = 8 * blk, rhi = min(rlo + 7, Ny - 1);
    double m = INFINITY;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = rlo + q;
        if (r <= rhi) {
            const double ddy = Y0[r] - py, ady2 = ddy * ddy;
            if (!(ady2 > thr || ady2 > lim2)) {
                if (c[q].x >= 0) {
                    const double ddx = X0[c[q].x] - px;
                    m = fmin(m, ddx * ddx + ady2);
                }
                if (c[q].y < Nx) {
                    const double ddx = X0[c[q].y] - px;
                    m = fmin(m, ddx * ddx + ady2);
                }
            }
        }
    }
    return m;
}

// pass 2: the reference's objective fl(fl(sqrt(d2)) + V*10e3) for the candidates that can still win or tie;
// ties -> lowest flat index (numpy's first minimum)
__device__ __forceinline__ void wall_block_eval(double px, double py, int blk, const int2 (&c)[8], int Nx, int Ny,
                                                const Axis X0, const Axis Y0, double off, double thr, double lim2,
                                                double& best_f, long long& best_flat) {
    const int rlo = 8 * blk, rhi = min(rlo + 7, Ny - 1);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = rlo + q;
        if (r <= rhi) {
            const double ddy = Y0[r] - py, ady2 = ddy * ddy;
            if (!(ady2 > thr || ady2 > lim2)) {
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const int cnd = side ? c[q].y : c[q].x;
                    if (cnd < 0 || cnd >= Nx) continue;
                    const double ddx = X0[cnd] - px;
                    const double d2 = ddx * ddx + ady2;
                    if (d2 > thr) continue;
                    const double f = sqrt(d2) + off;
                    const long long flat = (long long)r * Nx + cnd;
                    if (f < best_f || (f == best_f && flat < best_flat)) {
                        best_f = f;
                        best_flat = flat;
                    }
                }
            }
        }
    }
}

// Called by all 64 lanes (`go` false for lanes without a search).  Returns the flat index of the cell the reference's
// np.argmin(distances + V*10e3) selects (pedestrians.py:311-315), or -1 when nothing lies within we.lim2.
// prev_flat (>= 0): a cell of the agent's site set -- where the previous step's search ended -- taken as the incumbent: the
// search then only has to look where something can still beat or tie it (exactly the reference's cell either way: an
// incumbent is one more candidate, met again when its block is visited).
__device__ __forceinline__ long long wall_search(const GcfmParams& P, const WEntry& we, bool go, const Axis X0,
                                                 const Axis Y0, long long prev_flat = -1, int32_t* diag_cnt = nullptr) {
    const double px = we.px, py = we.py;
    const int Nx = P.Nx, Ny = P.Ny, nb8 = (Ny + 7) >> 3;
    const double lim2 = we.lim2, off = we.off;
    const int jx = we.jx;
    const int2* __restrict__ col = we.col;
    const int32_t* __restrict__ g8 = we.g8;
    const int b0 = we.b0;
    const bool prune_ok = we.prune_ok != 0;
    double best_f = INFINITY, best_d2 = INFINITY, thr = INFINITY;
    long long best_flat = -1;
    if (go && prev_flat >= 0) {
        const int pr = (int)(prev_flat / Nx), pc = (int)(prev_flat % Nx);
        const double ddx = X0[pc] - px, ddy = Y0[pr] - py, d2 = ddx * ddx + ddy * ddy;
        if (!(d2 > lim2)) {
            best_d2 = d2;
            thr = d2 + 4.8e-10 * sqrt(d2) + 1e-19;
            best_f = sqrt(d2) + off;
            best_flat = prev_flat;
        }
    }
    bool up = go, dn = go;
#ifdef OC_DIAG_BUILD
    int d_rounds = 0, d_visits = 0, d_myvis = 0;
#endif
    for (int o = 0; __any(up || dn); o += 8) {
#ifdef OC_DIAG_BUILD
        ++d_rounds;
#endif
        // ---- this round's 16 blocks: slot 2q = up-block b0 + o + q, slot 2q + 1 = down-block b0 - 1 - o - q
        int gap[16];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int bu = b0 + o + q, bd = b0 - 1 - o - q;
            gap[2 * q] = (up && bu < nb8) ? g8[bu] : GAP_NONE;
            gap[2 * q + 1] = (dn && bd >= 0) ? g8[bd] : GAP_NONE;
        }
        uint32_t pending = 0u;
        const double cut0 = fmin(thr, lim2);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int bu = b0 + o + q, bd = b0 - 1 - o - q;
            if (up) {
                if (bu >= nb8) up = false;
                else {
                    double dyb2;
                    const double lb = wall_block_bound(px, py, bu, gap[2 * q], prune_ok, jx, Ny, X0, Y0, dyb2);
                    if (dyb2 > cut0) up = false;                    // every farther block is farther still
                    else if (!(lb > cut0)) pending |= 1u << (2 * q);
                }
            }
            if (dn) {
                if (bd < 0) dn = false;
                else {
                    double dyb2;
                    const double lb = wall_block_bound(px, py, bd, gap[2 * q + 1], prune_ok, jx, Ny, X0, Y0, dyb2);
                    if (dyb2 > cut0) dn = false;
                    else if (!(lb > cut0)) pending |= 1u << (2 * q + 1);
                }
            }
        }
        // ---- visit the marked blocks, nearest first (slots are ordered by distance within a direction)
        while (__any(pending != 0u)) {
#ifdef OC_DIAG_BUILD
            ++d_visits;
            d_myvis += pending ? 1 : 0;
#endif
            if (pending) {
                const int slot = __ffs((int)pending) - 1;
                pending &= pending - 1u;
                const int q = slot >> 1;
                const int blk = (slot & 1) ? b0 - 1 - o - q : b0 + o + q;
                int gsel = GAP_NONE;
#pragma unroll
                for (int z = 0; z < 16; ++z) gsel = (z == slot) ? gap[z] : gsel;
                double dyb2;
                const double lb = wall_block_bound(px, py, blk, gsel, prune_ok, jx, Ny, X0, Y0, dyb2);
                const double cut = fmin(thr, lim2);
                if (dyb2 > cut) {                                   // this direction is finished
                    if (slot & 1) { dn = false; pending &= 0x55555555u; }
                    else { up = false; pending &= 0xAAAAAAAAu; }
                } else if (!(lb > cut)) {
                    const int rlo = 8 * blk, rhi = min(rlo + 7, Ny - 1);
                    int2 c[8];                                      // the block's table entries: one batch of loads
#pragma unroll
                    for (int z = 0; z < 8; ++z) c[z] = col[min(rlo + z, rhi)];
                    best_d2 = fmin(best_d2, wall_block_min(px, py, blk, c, Nx, Ny, X0, Y0, thr, lim2));
                    // anything farther than thr cannot tie after the rounding of sqrt(d2) + off at ulp(1e6)
                    if (best_d2 < INFINITY) thr = best_d2 + 4.8e-10 * sqrt(best_d2) + 1e-19;
                    wall_block_eval(px, py, blk, c, Nx, Ny, X0, Y0, off, thr, lim2, best_f, best_flat);
                }
            }
        }
        // ---- next round's nearest rows: stop a direction once its rows cannot matter any more
        const double cut = fmin(thr, lim2);
        const int nu = b0 + o + 8, nd = b0 - 1 - o - 8;
        if (up) {
            if (nu >= nb8) up = false;
            else {
                const double yy = Y0[8 * nu], a2 = (yy - py) * (yy - py);
                if (yy > py && a2 > cut) up = false;
            }
        }
        if (dn) {
            if (nd < 0) dn = false;
            else {
                const double yy = Y0[min(8 * nd + 7, Ny - 1)], a2 = (yy - py) * (yy - py);
                if (yy < py && a2 > cut) dn = false;
            }
        }
    }
#ifdef OC_DIAG_BUILD
    if (diag_cnt != nullptr) {          // [8] rounds of the wavefront, [9] its visit trips, [10] wavefronts, [11] lanes searching,
        const unsigned long long mg = __ballot(go);                                  // [12] block visits summed over the lanes
        int tot = d_myvis;
        for (int o2 = 32; o2 > 0; o2 >>= 1) tot += __shfl_xor(tot, o2, 64);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&diag_cnt[8], d_rounds);
            atomicAdd(&diag_cnt[9], d_visits);
            atomicAdd(&diag_cnt[10], 1);
            atomicAdd(&diag_cnt[11], __popcll(mg));
            atomicAdd(&diag_cnt[12], tot);
        }
    }
#endif
    return best_flat;
}

// capped exact distance transform of the site set on the grid nodes (when a wall map is built): every node is a query of
// the block-pruned search, radius = cap.
__global__ void __launch_bounds__(64) k_edt_nodes(GcfmParams P, const int2* __restrict__ lr,
                                                  const int32_t* __restrict__ gap8, const Axis X0, const Axis Y0,
                                                  double cap, float* __restrict__ edt) {
    const int Nx = P.Nx, Ny = P.Ny;
    const long long t = (long long)blockIdx.x * 64 + threadIdx.x;
    const bool go = t < (long long)Nx * Ny;
    const int i = go ? (int)(t / Nx) : 0, j = go ? (int)(t % Nx) : 0;
    WEntry we;
    we.px = X0[j];
    we.py = Y0[i];
    we.lim2 = cap * cap;
    we.off = 0.0;
    we.col = lr + (size_t)j * Ny;
    we.g8 = gap8 + (size_t)j * ((Ny + 7) >> 3);
    we.b0 = i >> 3;
    we.prune_ok = 1;
    we.jx = j;
    we.pad = 0;
    const long long flat = wall_search(P, we, go, X0, Y0);
    if (!go) return;
    double d = INFINITY;
    if (flat >= 0) {
        const double ddx = X0[(int)(flat % Nx)] - we.px, ddy = Y0[(int)(flat / Nx)] - we.py;
        d = sqrt(ddx * ddx + ddy * ddy);
    }
    // nodes farther than the cap from every site store +inf: a float-rounded cap must never pass for a distance (it
    // would become an upper bound on the nearest-site distance in wall_needed and prune the real site)
    edt[t] = (d < cap) ? (float)d : INFINITY;
}

// standalone wall search (diagnostic exports): one lane per slot
__global__ void __launch_bounds__(64) k_wall(GcfmParams P, Scratch S, const KeyDev* __restrict__ keys, const Axis X0,
                                             const Axis Y0) {
    const Snap& C = S.snap;
    const int k = blockIdx.x * 64 + threadIdx.x;
    WEntry we = {};
    bool go = false;
    if (C.pidx[k] >= 0) {
        const double2 p = C.ppos[k];
        go = wall_prepare(P, keys[S.pdk[k].key], X0, Y0, p.x, p.y, (double)C.pvel[k].z, we);
    }
    S.widx[k] = wall_search(P, we, go, X0, Y0);
}

// standalone pair forces (diagnostic exports): 4 tiles per block, one wavefront each
__global__ void __launch_bounds__(256) k_pair(GcfmParams P, Scratch S) {
    __shared__ PairLds<1> L;
    const Snap& C = S.snap;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, it = blockIdx.x * 4 + w;
    const bool live = it < S.ntiles;
    if (live) pair_load_tile<1>(P, C, S.pdk, L, it, w, lane);
    __syncthreads();
    if (live) pair_share<1>(P, S, C, L, it, w, 0, w, lane);
    __syncthreads();
    if (live) xs(S, S.f_pair, S.ts_pair, it * 64 + lane) = make_double2(L.accx[w][lane], L.accy[w][lane]);
}

// ---------------------------------------------------------------------------------- lookup
// Python float floor division (CPython floatobject.c float_floor_div / numpy npy_divmod)
__device__ __forceinline__ double py_floordiv(double vx, double wx) {
    double mod = fmod(vx, wx);
    double div = (vx - mod) / wx;
    if (mod != 0.0) {
        if ((wx < 0) != (mod < 0)) div -= 1.0;
    }
    double fl;
    if (div != 0.0) {
        fl = floor(div);
        if (div - fl > 0.5) fl += 1.0;
    } else {
        fl = copysign(0.0, vx / wx);
    }
    return fl;
}

template <typename T>
__device__ __forceinline__ double2 field_fetch(const KeyDev& kd, int step, int i0, int i1, int j0, int j1, int Nx,
                                               int Ny) {
    // slices are stored padded as (rows, Nx), rows = Ny or a row block starting at grid row kd.row0; the reference's
    // (Ny-2, Nx-2) array is the interior block: interior index i is grid row i + 1
    const size_t W = (size_t)Nx;
    const int rows = kd.rows < 0 ? Ny : kd.rows;
    const T* fx = (const T*)kd.vx + (size_t)step * W * rows + 1;
    const T* fy = (const T*)kd.vy + (size_t)step * W * rows + 1;
    const size_t r0 = (size_t)(i0 + 1 - kd.row0), r1 = (size_t)(i1 + 1 - kd.row0);
    double ax = (double)fx[r0 * W + j0], bx = (double)fx[r1 * W + j1];
    double ay = (double)fy[r0 * W + j0], by = (double)fy[r1 * W + j1];
    return make_double2((ax + bx) / 2, (ay + by) / 2);
}

// ---- the same values from the interpolant ring (oc_set_field_ring) -----------------------------------------------------
// Slice `step` of the direction field is not in memory; what is: for the accepted RK step that contains the slice's time,
// y_old, K0 and Q1 .. Q3 of every cell (OcRingSlot / five planes per slot) and the slice's four factors
// h x^1..4 (OcRingSlice).  ring_dir evaluates phi at the cell and its four neighbours (scipy rk.py:560-575), clips it and
// forms `vels` (optimals.py:168-186) exactly as the slice writer k_out does -- same operations in the same order on the
// same inputs, so the result is the bit pattern a slice would hold.  A stage tile that was trivial in the accepted attempt
// (exact active set: phi one constant u on the tile, every derivative zero) stored nothing: its cells read (u, 0, 0, 0, 0).
template <typename T>
struct RingCtx {
    const T* slot;              // the slot's data
    const int32_t* first;       // tile t is trivial in this slot <=> slot_index < first[t]; NULL: every tile holds data
    int slot_index;
    T h1, h2, h3, h4, lim_s, u;
    T lim, mu, inv2dx, inv2dy;
    int rows, row0, Nx, tnbx;
};

template <typename T>
__device__ __forceinline__ RingCtx<T> ring_open(const GcfmParams& P, const KeyDev& kd, const KeyStep& ks) {
    RingCtx<T> c;
    c.slot = (const T*)ks.slot;
    c.first = ks.first;
    c.slot_index = ks.slot_index;
    c.h1 = (T)ks.h[0]; c.h2 = (T)ks.h[1]; c.h3 = (T)ks.h[2]; c.h4 = (T)ks.h[3];
    c.lim_s = (T)ks.lim_s;
    c.u = (T)ks.u;
    c.mu = (T)P.mu;
    c.lim = (T)P.lim;
    c.inv2dx = (T)(1.0 / (2 * P.dx));
    c.inv2dy = (T)(1.0 / (2 * P.dy));
    c.rows = kd.rows < 0 ? P.Ny : kd.rows;
    c.row0 = kd.row0;
    c.Nx = P.Nx;
    c.tnbx = kd.ring_tnbx;
    return c;
}

// the step's entry of the table, for one key (k_ring_step / k_finish)
template <typename T>
__device__ __forceinline__ void ring_step_of(const KeyDev& kd, int step, KeyStep& out) {
    const OcRingSlice<T> sl = ((const OcRingSlice<T>*)kd.ring_slices)[step];
    const OcRingSlot<T> rs = ((const OcRingSlot<T>*)kd.ring_slots)[sl.slot];
    out.slot = (const T*)kd.ring_data + (size_t)sl.slot * (size_t)kd.ring_slot_elems;
    out.first = rs.dense ? nullptr : kd.ring_first;
    out.slot_index = sl.slot;
    for (int q = 0; q < 4; ++q) out.h[q] = (double)sl.h[q];
    out.lim_s = (double)rs.lim_s;
    out.u = (double)rs.u;
}
__device__ __forceinline__ void ring_step_prepare(const KeyDev& kd, int step, KeyStep& out) {
    out.slot = nullptr;
    out.first = nullptr;
    out.slot_index = 0;
    if (kd.ring_data == nullptr || step < 0 || step >= kd.nt_opt - 1 || step >= kd.nslice_alloc) return;
    if (kd.ring_state_dtype == OC_F32) ring_step_of<float>(kd, step, out);
    else ring_step_of<double>(kd, step, out);
}
__global__ void k_ring_step(const KeyDev* __restrict__ keys, int nkeys, int step, KeyStep* __restrict__ ksteps) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nkeys) ring_step_prepare(keys[k], step, ksteps[k]);
}

// clipped phi(t_eval) at LOCAL row r, column c (optimals.py:172)
template <typename T>
__device__ __forceinline__ T ring_phi(const RingCtx<T>& c, int r, int cc) {
    // (the loads are issued whatever the flag says -- a trivial tile's slot memory exists, its content is simply not used --
    // so that the five cells' flags and values of a lookup are all in flight together: one level of latency)
    const size_t ci = (size_t)r * c.Nx + cc, rn = (size_t)c.rows * c.Nx;
    const T* b = c.slot;
    const T y = b[oc_ring_off(ci, 0, rn)], q0 = b[oc_ring_off(ci, 1, rn)], q1 = b[oc_ring_off(ci, 2, rn)],
            q2 = b[oc_ring_off(ci, 3, rn)], q3 = b[oc_ring_off(ci, 4, rn)];
    const bool triv = c.first != nullptr && c.slot_index < c.first[(r / OC_RING_STY) * c.tnbx + cc / OC_RING_STX];
    T ph = y + fma_t(q3, c.h4, fma_t(q2, c.h3, fma_t(q1, c.h2, q0 * c.h1)));
    ph = triv ? c.u : ph;               // (u + h (0 x + 0 x^2 + ...) = u)
    return (ph > c.lim_s) ? ph : ((ph < c.lim_s) ? c.lim_s : (T)0);
}

// clip (optimals.py:172) and the direction from five clipped values (optimals.py:174-186), as in k_out
template <typename T>
__device__ __forceinline__ T ring_clip(const RingCtx<T>& c, T ph) {
    return (ph > c.lim_s) ? ph : ((ph < c.lim_s) ? c.lim_s : (T)0);
}
template <typename T, typename TO>
__device__ __forceinline__ double2 ring_vels(const RingCtx<T>& c, T pw, T pe, T ps, T pn, T p0) {
    const T pc = ring_clip(c, p0);                  // the reference clips the centre twice (:177)
    const T gx = (pe - pw) * c.inv2dx, gy = (pn - ps) * c.inv2dy;
    T ox, oy;
    direction_of<T>(gx, gy, c.mu * pc, c.lim, ox, oy);
    return make_double2((double)(TO)ox, (double)(TO)oy);
}

// the direction field's value at padded grid cell (global row gr, column gc), as float64 of what a slice of type TO holds
template <typename T, typename TO>
__device__ __forceinline__ double2 ring_dir(const RingCtx<T>& c, int Ny, int gr, int gc) {
    const int r = gr - c.row0;
    if (r < 1 || r >= c.rows - 1 || gc < 1 || gc >= c.Nx - 1) return make_double2(0.0, 0.0);   // the slices' border ring
    const T pw = ring_phi(c, r, gc - 1), pe = ring_phi(c, r, gc + 1);
    const T ps = ring_phi(c, r - 1, gc), pn = ring_phi(c, r + 1, gc);
    const T p0 = ring_phi(c, r, gc);
    return ring_vels<T, TO>(c, pw, pe, ps, pn, p0);
}

template <typename T, typename TO>
__device__ __forceinline__ double2 ring_fetch(const GcfmParams& P, const KeyDev& kd, const KeyStep& ks, int i0, int i1, int j0,
                                              int j1) {
    // interior index i of the reference's (Ny-2, Nx-2) array is grid row i + 1 (field_fetch)
    const RingCtx<T> c = ring_open<T>(P, kd, ks);
    const int r0 = i0 + 1 - c.row0, r1 = i1 + 1 - c.row0, c0 = j0 + 1, c1 = j1 + 1;      // local rows, columns
    const bool inner = r0 >= 1 && r1 + 1 < c.rows && c0 >= 1 && c1 + 1 < c.Nx && r0 <= r1 && c0 <= c1;
    // Far field first: the cells a lookup reads lie in the box rows r0-1 .. r1+1, columns c0-1 .. c1+1 -- smaller than a stage
    // tile, so at most the four tiles of its corners.  If all of them were trivial in the slice's step, phi is one constant
    // on the whole box: every difference is zero and so is the direction (what k_out writes there) -- four flag bytes
    // instead of fifty values.  At 2048 x 2048 that is ~90 % of the cells at any time of an m = 0 solve.
    if (inner && c.first != nullptr) {
        const int ta = ((r0 - 1) / OC_RING_STY) * c.tnbx, tb = ((r1 + 1) / OC_RING_STY) * c.tnbx;
        const int ua = (c0 - 1) / OC_RING_STX, ub = (c1 + 1) / OC_RING_STX;
        const int32_t f0 = c.first[ta + ua], f1 = c.first[ta + ub], f2 = c.first[tb + ua], f3 = c.first[tb + ub];
        const int si = c.slot_index;
        if (si < f0 && si < f1 && si < f2 && si < f3) return make_double2(0.0, 0.0);
    }
    if (inner && r1 == r0 + 1 && c1 == c0 + 1) {
        // the usual case, the diagonal pair (r0, c0), (r0 + 1, c0 + 1) (quirk Q7): the two crosses share two cells
        const T a_w = ring_phi(c, r0, c0 - 1), a_c = ring_phi(c, r0, c0), a_s = ring_phi(c, r0 - 1, c0);
        const T sh_e = ring_phi(c, r0, c1);         // east of A = south of B
        const T sh_n = ring_phi(c, r1, c0);         // north of A = west of B
        const T b_c = ring_phi(c, r1, c1), b_e = ring_phi(c, r1, c1 + 1), b_n = ring_phi(c, r1 + 1, c1);
        const double2 a = ring_vels<T, TO>(c, a_w, sh_e, a_s, sh_n, a_c);
        const double2 b = ring_vels<T, TO>(c, sh_n, b_e, sh_e, b_n, b_c);
        return make_double2((a.x + b.x) / 2, (a.y + b.y) / 2);
    }
    const double2 a = ring_dir<T, TO>(c, P.Ny, i0 + 1, j0 + 1), b = ring_dir<T, TO>(c, P.Ny, i1 + 1, j1 + 1);
    return make_double2((a.x + b.x) / 2, (a.y + b.y) / 2);
}

template <typename T, typename TO>
__device__ __forceinline__ double2 ring_bilinear(const GcfmParams& P, const KeyDev& kd, const KeyStep& ks, int r0, int r1, int c0,
                                                 int c1, double fy, double fx) {
    const RingCtx<T> c = ring_open<T>(P, kd, ks);
    const double2 g00 = ring_dir<T, TO>(c, P.Ny, r0, c0), g01 = ring_dir<T, TO>(c, P.Ny, r0, c1);
    const double2 g10 = ring_dir<T, TO>(c, P.Ny, r1, c0), g11 = ring_dir<T, TO>(c, P.Ny, r1, c1);
    const double w00 = (1.0 - fy) * (1.0 - fx), w01 = (1.0 - fy) * fx, w10 = fy * (1.0 - fx), w11 = fy * fx;
    return make_double2(w00 * g00.x + w01 * g01.x + w10 * g10.x + w11 * g11.x,
                        w00 * g00.y + w01 * g01.y + w10 * g10.y + w11 * g11.y);
}

// OC_FIX_LOOKUP: the four grid nodes (r0|r1, c0|c1) around the position, weights (1-fy|fy) x (1-fx|fx); the padded
// slices are indexed by grid row and column directly
template <typename T>
__device__ __forceinline__ double2 field_bilinear(const KeyDev& kd, int step, int r0, int r1, int c0, int c1, double fy,
                                                  double fx, int Nx, int Ny) {
    const size_t W = (size_t)Nx;
    const int rows = kd.rows < 0 ? Ny : kd.rows;
    const T* gx = (const T*)kd.vx + (size_t)step * W * rows;
    const T* gy = (const T*)kd.vy + (size_t)step * W * rows;
    const size_t a = (size_t)(r0 - kd.row0) * W, b = (size_t)(r1 - kd.row0) * W;
    const double w00 = (1.0 - fy) * (1.0 - fx), w01 = (1.0 - fy) * fx, w10 = fy * (1.0 - fx), w11 = fy * fx;
    return make_double2(
        w00 * (double)gx[a + c0] + w01 * (double)gx[a + c1] + w10 * (double)gx[b + c0] + w11 * (double)gx[b + c1],
        w00 * (double)gy[a + c0] + w01 * (double)gy[a + c1] + w10 * (double)gy[b + c0] + w11 * (double)gy[b + c1]);
}

__device__ __forceinline__ double2 lookup_bilinear(const GcfmParams& P, const KeyDev& kd, const KeyStep& ks, double x,
                                                   double y, double vdes, int step) {
    const int Nx = P.Nx, Ny = P.Ny;
    if (step >= kd.nslice_alloc) return make_double2(0.0, 0.0);
    // node j sits at X0[j] = j * L / (Nx - 1); the field lives on the interior nodes 1 .. Nx-2 (clamped outside)
    auto split = [](double g, int n, int& i0, int& i1, double& f) {
        double fl = floor(g);
        if (!(fl >= 1.0)) fl = 1.0, g = 1.0;                 // (also NaN)
        if (fl >= (double)(n - 2)) fl = (double)(n - 2), g = fl;
        i0 = (int)fl;
        i1 = i0 + 1 > n - 2 ? n - 2 : i0 + 1;
        f = g - fl;
    };
    int c0, c1, r0, r1;
    double fx, fy;
    split(x / (P.L / (Nx - 1)), Nx, c0, c1, fx);
    split(y / (P.H / (Ny - 1)), Ny, r0, r1, fy);
    if (r0 < kd.own_lo || r0 >= kd.own_hi) return make_double2(0.0, 0.0);       // another rank's rows
    if (kd.rows >= 0 && (r1 < kd.row0 || r1 >= kd.row0 + kd.rows)) r1 = r0;
    double2 d;
    if (kd.ring_data != nullptr) {
        if (ks.slot == nullptr) return make_double2(0.0, 0.0);
        if (kd.ring_state_dtype == OC_F32) d = ring_bilinear<float, float>(P, kd, ks, r0, r1, c0, c1, fy, fx);
        else if (kd.dtype == OC_F32) d = ring_bilinear<double, float>(P, kd, ks, r0, r1, c0, c1, fy, fx);
        else d = ring_bilinear<double, double>(P, kd, ks, r0, r1, c0, c1, fy, fx);
    } else {
        d = (kd.dtype == OC_F32) ? field_bilinear<float>(kd, step, r0, r1, c0, c1, fy, fx, Nx, Ny)
                                 : field_bilinear<double>(kd, step, r0, r1, c0, c1, fy, fx, Nx, Ny);
    }
    return make_double2(vdes * d.x, vdes * d.y);
}

// des = v_des * choose_optimal_velocity(pos, step)  (optimals.py:212-250, simulations.py:281)
__device__ __forceinline__ double2 lookup(const GcfmParams& P, const KeyDev& kd, const KeyStep& ks, double x, double y,
                                          double vdes, int step, int agent, int32_t* err) {
    if (step >= kd.nt_opt - 1) return make_double2(0.0, 0.0);
    if (P.fix & OC_FIX_LOOKUP) return lookup_bilinear(P, kd, ks, x, y, vdes, step);
    const int Nx = P.Nx, Ny = P.Ny;
    int j0, j1, i0, i1;
    if (x < P.L - P.dx) {
        j0 = (int)py_floordiv(x, P.dx);
        j1 = (x > P.dx) ? j0 + 1 : j0;
    } else {
        j0 = j1 = Nx - 3;
    }
    if (y < P.H - P.dy) {
        i0 = (int)py_floordiv(y, P.dy);
        i1 = (y > P.dy) ? i0 + 1 : i0;
    } else {
        i0 = i1 = Ny - 3;
    }
    // numpy index semantics on axes of size Nx-2 / Ny-2: negative wraps once, else IndexError (quirk Q8)
    bool bad = false;
    auto fix = [&](int& v, int n) {
        if (v < 0) v += n;
        if (v < 0 || v >= n) {
            bad = true;
            v = v < 0 ? 0 : n - 1;
        }
    };
    fix(j0, Nx - 2);
    fix(j1, Nx - 2);
    fix(i0, Ny - 2);
    fix(i1, Ny - 2);
    if (bad && err) atomicMax(err, agent + 1);
    if (step >= kd.nslice_alloc) return make_double2(0.0, 0.0);
    // domain-decomposed field: the rank that owns the first cell's grid row answers (its block carries the next row)
    if (i0 + 1 < kd.own_lo || i0 + 1 >= kd.own_hi) return make_double2(0.0, 0.0);
    if (kd.rows >= 0 && (i1 + 1 < kd.row0 || i1 + 1 >= kd.row0 + kd.rows)) i1 = i0;   // (a wrapped negative index of an
                                                                                      // agent outside the room)
    double2 d;
    if (kd.ring_data != nullptr) {
        if (ks.slot == nullptr) return make_double2(0.0, 0.0);
        if (kd.ring_state_dtype == OC_F32) d = ring_fetch<float, float>(P, kd, ks, i0, i1, j0, j1);
        else if (kd.dtype == OC_F32) d = ring_fetch<double, float>(P, kd, ks, i0, i1, j0, j1);
        else d = ring_fetch<double, double>(P, kd, ks, i0, i1, j0, j1);
    } else {
        d = (kd.dtype == OC_F32) ? field_fetch<float>(kd, step, i0, i1, j0, j1, Nx, Ny)
                                 : field_fetch<double>(kd, step, i0, i1, j0, j1, Nx, Ny);
    }
    return make_double2(vdes * d.x, vdes * d.y);
}

// ---------------------------------------------------------------------------------- noise (Philox4x32-10)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---------------------------------------------------------------------------------- the step: k_forces, k_finish
struct StepArgs {               // caller-owned state (agent order) and the step's inputs
    double *x, *y, *vx, *vy;
    uint8_t* status;
    double* agent_time;
    const double* noise;
    uint64_t seed;
    int32_t step;
    int32_t* cnt_out;           // device int32[4]
};

// Everything of simulations.py:277-299 that does not need the pair forces, for the 64 agents of tile `it` (one
// wavefront, one lane per agent): the exact nearest wall cell (pedestrians.py:311-315) and its force, and the desired
// velocity from the floor field.  A chain of dependent loads (key table -> distance bound -> gap table -> site table
// -> field) with little arithmetic: it runs in spare wavefronts beside the pair forces and leaves k_finish one level
// of loads.
__device__ __forceinline__ void prep_tile(const GcfmParams& P, const Scratch& S, const Snap& C,
                                          const KeyDev* __restrict__ keys, const KeyStep* __restrict__ ksteps, int step,
                                          const Axis X0, const Axis Y0, int it, int lane, int32_t* err, bool do_wall,
                                          bool do_lookup) {
    const int k = it * 64 + lane;
    const int i = C.pidx[k];
    const bool act = i >= 0;
    const double2 p = C.ppos[k];
    const double2 v = S.pvd[k];
    const PDK dk = S.pdk[k];
    const KeyDev& kd = keys[act ? dk.key : 0];
    // (sharded step: the tile's owner does the wall part; the lookup is answered by the owner of the field rows, or by
    // the tile's owner when every rank holds the whole field; what a rank does not compute stays zero for the sum)
    if (do_lookup) {
        double2 ds = make_double2(0.0, 0.0);
        if (act) ds = lookup(P, kd, ksteps[dk.key], p.x, p.y, dk.v_des, step, i, err);     // its loads fly during the search
        xs(S, S.des, S.ts_des, k) = ds;
    }
    if (do_wall) {
        WEntry we = {};
        const bool need = act && wall_prepare(P, kd, X0, Y0, p.x, p.y, (double)C.pvel[k].z, we);
        const long long widx = wall_search(P, we, need, X0, Y0, (need && S.wprev != nullptr) ? (long long)S.wprev[i] : -1,
                                           S.seq_ctl);
        if (act && S.wprev != nullptr) S.wprev[i] = need ? (int32_t)widx : -1;
        double2 fw = make_double2(0.0, 0.0);
        if (act) fw = wall_force_of(P, widx, p.x, p.y, v.x, v.y, dk.v_des, X0, Y0);
        xs(S, S.f_wall, S.ts_wall, k) = fw;
    }
}

struct TileShare {              // which tiles this launch works on (the whole step: 0, ntiles, 1, 0)
    int32_t lo, hi;             // pair forces and wall forces: tiles lo .. hi - 1
    int32_t by_work;            // dispatch the pair blocks by decreasing work (tile_order) instead of by index
    int32_t lookup_all;         // do the floor-field lookup of every tile (the field's row blocks decide who answers)
    int32_t prep_lookup;        // 1: the prep blocks do the lookups (every tile when lookup_all, else lo .. hi - 1);
                                // 0: each pair block looks up its own tile's agents (single GPU, the default)
};

// The step's force kernel.  Blocks of 4 wavefronts.  The first `nprep` blocks sweep the tiles with prep_tile (they
// are dispatched first; their wavefronts mostly wait on memory).  Every other block carries 4/NS tiles: barrier 1,
// the tiles' i records are in LDS; every wavefront does its share of its tile's pair forces (pair_share); barrier 2,
// the tiles' force sums are complete and go to f_pair.  128 registers and < 40 KB of LDS per block keep 4 blocks per
// CU resident.
// FIX = false: the OC_FIX_* corrections are compiled out (the default, bug-compatible kernel pays nothing for them)
template <int NS, bool FIX>
__global__ void __launch_bounds__(256, OC_FORCES_OCC) k_forces(GcfmParams P, Scratch S, const KeyDev* __restrict__ keys,
                                                const KeyStep* __restrict__ ksteps, int step, const Axis X0, const Axis Y0,
                                                int nprep, TileShare sh, int32_t* __restrict__ cnt_out) {
    __shared__ PairLds<NS> L;
    if (!FIX) __builtin_assume(P.fix == 0u);        // (launch_step picks FIX = true whenever a bit is set)
    const Snap& C = S.snap;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x < nprep) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {     // this step's counters (k_finish updates them later in the stream)
            cnt_out[C_EXIT] = cnt_out[3] = 0;
            cnt_out[C_ACTIVE] = S.ctrl[C_NACTIVE];
        }
        // prep wavefronts mostly wait on memory; when they can issue they go first, so that their chains end (and their
        // blocks' slots are free for pair blocks) as early as possible
        __builtin_amdgcn_s_setprio(3);
        for (int t = blockIdx.x * 4 + w; t < S.ntiles; t += nprep * 4) {
            const bool mine = t >= sh.lo && t < sh.hi;
            if ((mine || sh.lookup_all) && !OC_DIAG(P, 2u))
                prep_tile(P, S, C, keys, ksteps, step, X0, Y0, t, lane, &S.ctrl[C_ERR + 8], mine,
                          sh.prep_lookup != 0 && (mine || sh.lookup_all) && !OC_DIAG(P, 8u));
        }
        return;
    }
    // (Round 4, measured and not kept: PERSISTENT pair blocks -- as many as the chip holds at once, each drawing tile after tile
    // from a counter, heaviest first -- to close the half-empty second "generation" of blocks at the launch's end: 110 us
    // against 80.  A tile takes ~42 us on a CU that holds four pair blocks and ~25 us on one that holds fewer: the launch is
    // bound by what a CU gets done, ~10 us of CU time per tile, not by how the blocks are packed; the queue keeps every CU
    // full to the end, where the one-block-per-tile launch lets the last blocks run on emptier CUs.)
    const int tl = w / NS, sub = w % NS, ord = ((int)blockIdx.x - nprep) * (4 / NS) + tl;
    const bool live = ord < sh.hi - sh.lo;
    // heavy tiles first: the launch ends with its slowest wavefront
    const int it = live ? (sh.by_work ? S.tile_order[ord] : sh.lo + ord) : 0;
    if (live && sub == 0) {
        if (lane == 0) S.tile_work[it] = 0;
        pair_load_tile<NS>(P, C, S.pdk, L, it, tl, lane);
    }
    __syncthreads();
    if (live) pair_share<NS>(P, S, C, L, it, tl, sub, w, lane);
    __syncthreads();
    if (live && sub == 0) xs(S, S.f_pair, S.ts_pair, it * 64 + lane) = make_double2(L.accx[tl][lane], L.accy[tl][lane]);
    // The tile's floor-field lookups (optimals.py:212-250; through the interpolant ring: five cells' values and `vels` per
    // looked-up cell), one lane per agent, by the tile's last wavefront once its pair share is done.  In the prep blocks the
    // lookup sat on the launch's critical chain -- the prep wavefronts' dependent loads ARE the launch's length -- and cost
    // 11 us per launch with slices, 27 us through the ring; here its latency hides behind the other blocks' arithmetic.
    if (live && sub == NS - 1 && sh.prep_lookup == 0 && !OC_DIAG(P, 8u)) {
        const int k = it * 64 + lane;
        const int i = C.pidx[k];
        double2 ds = make_double2(0.0, 0.0);
        if (i >= 0) {
            const double2 p = C.ppos[k];
            const PDK dk = S.pdk[k];
            ds = lookup(P, keys[dk.key], ksteps[dk.key], p.x, p.y, dk.v_des, step, i, &S.ctrl[C_ERR + 8]);
        }
        xs(S, S.des, S.ts_des, k) = ds;
    }
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float readlane_f32(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// The candidate tiles of tile `it` in the sequential mode: its neighbour list as it stands (cutoff + 2 drift around the
// sort-time boxes: covers cutoff + 2 M around every position of the step), or every tile when the lists are not valid.  The
// dependency count and its release walk the SAME tiles with the SAME positions (seq_pos0), so an agent counted is an agent
// released.
struct SeqCand {
    const int32_t* nb;
    int cnt;
    __device__ __forceinline__ SeqCand(const Scratch& S, int it) {
        nb = S.nb_list + (size_t)it * (NB_MAX + 1);
        const int c = nb[0];
        const bool use_list = (c >= 0) && (S.ctrl[C_INVALID] == 0);
        cnt = use_list ? c : -S.ntiles;             // (< 0: all tiles 0 .. -cnt - 1)
    }
    __device__ __forceinline__ int size() const { return cnt >= 0 ? cnt : -cnt; }
    __device__ __forceinline__ int tile(int e) const { return cnt >= 0 ? nb[1 + e] : e; }
};

#define RECT_TAB (1 + 4 * OC_MAX_TARGET_RECTS)      // doubles per key in the exit-rectangle table: count, then (cx,cy,w,h)
#define RECT_KEYS_LDS 8

// simulations.py:303-332 for the 64 agents of a tile, float64, then the next step's packed snapshot in place.
// Everything is read by slot and was prepared by k_forces: one level of loads.
// SEQ (sequential mode, k_seq_go below): only the agents marked for this round (seq_state == 1) are advanced; the
// others keep their records; every agent advanced is then taken off the dependency counts of the LATER agents within reach.  `seq_round` selects the slot of the pending counter, `seq_margin` is the displacement
// bound the round's independence rests on (a larger move raises seq_ctl[4]).
// WPB: tiles (wavefronts) per block.  The synchronous step packs four tiles into a block: a quarter of the workgroups to
// dispatch.  By the kernel trace the launch gains little (C4 9.6 -> 9.1 us, C5 17.5 -> 17.6 us: profiles/r03_*_kernel_stats.csv);
// the bench's HIP-event samples of single launches at C5, taken while a solve runs on another stream, fell from 64 to 20 us.
template <bool SEQ, int WPB = 1>
__global__ void __launch_bounds__(64 * WPB) k_finish(GcfmParams P, Scratch S, const double* __restrict__ rect_tab, int nkeys,
                                               StepArgs A, int seq_round, double seq_margin,
                                               const double* __restrict__ err_slots, int n_err_slots, int64_t err_stride,
                                               const KeyDev* __restrict__ keys, KeyStep* __restrict__ ksteps, int lazy = 0) {
    __shared__ double l_rect[RECT_KEYS_LDS * RECT_TAB];
    const Snap& C = S.snap;
    static_assert(!SEQ || WPB == 1, "the sequential rounds keep one tile per block");
    const int it = (int)blockIdx.x * WPB + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63, k = it * 64 + lane;
    if (SEQ && seq_round > 0 && S.seq_ctl[(seq_round - 1) & 3] == 0) return;      // (a round enqueued past the last level)
    if (SEQ && it == 0 && lane == 0) S.seq_ctl[5] = seq_round + 1;
    // the NEXT step's ring lookups (interpolant ring: slice -> slot, factors): resolved here, once per key, off every path
    // (by block 0, the first to be dispatched: its two dependent loads per key -- slice table, slot record -- end long before
    // the launch does; in the last block they were the launch's tail)
    if (ksteps != nullptr && it == 0)
        for (int q = lane; q < nkeys; q += 64) ring_step_prepare(keys[q], A.step + 1, ksteps[q]);
    const bool tab_lds = nkeys <= RECT_KEYS_LDS;
    if (tab_lds)
        for (int q = threadIdx.x; q < nkeys * RECT_TAB; q += 64 * WPB) l_rect[q] = rect_tab[q];
    if (WPB > 1 && it >= S.ntiles) {                // (the last block's spare wavefronts: the barrier, nothing else)
        __syncthreads();
        return;
    }
    const int i = C.pidx[k];
    const int sstate = SEQ ? S.seq_state[k] : 1;
    const bool live = i >= 0;                       // an agent sits in this slot
    const bool act = live && sstate == 1;           // ... and is advanced by this launch
    bool out = false;
    const double2 p = C.ppos[k], v = S.pvd[k], ds = xs(S, S.des, S.ts_des, k), fw = xs(S, S.f_wall, S.ts_wall, k),
                  fp = xs(S, S.f_pair, S.ts_pair, k);
    double nx = (SEQ && live) ? p.x : 1e300, ny = (SEQ && live) ? p.y : 1e300;
    const PDK dk = S.pdk[k];
    // (needed only at the very end, by lane 0: fetched with the rest so that the tail holds no memory round trip)
    const double4 b0 = S.tile_box0[it];
    // IndexError-band flag of this step's lookups: the local word, or -- sharded step -- the ranks' words, which travelled
    // with the exchange buffer so that EVERY rank raises (a lookup is answered by the rank that holds the field rows)
    int err0 = 0;
    if (it == 0) {
        err0 = S.ctrl[C_ERR + 8];
        for (int r = 0; r < n_err_slots; ++r) err0 = max(err0, (int)err_slots[(size_t)r * err_stride]);
    }
    // (lazy: the time comes from its slot-ordered copy -- no gather by agent index; SEQ keeps the caller's array as the source)
    const double at0 = !act ? 0.0 : (lazy ? S.ptime[k] : (A.agent_time ? A.agent_time[i] : 0.0));
    double n0 = 0.0, n1 = 0.0;
    if (act) {      // noise: caller's standard normals, or Philox4x32-10 + Box-Muller (float32)
        if (A.noise) {
            n0 = A.noise[2 * (size_t)i];
            n1 = A.noise[2 * (size_t)i + 1];
        } else {
            uint32_t r[4];
            philox4x32_10((uint32_t)i, (uint32_t)A.step, 0u, 0u, (uint32_t)A.seed, (uint32_t)(A.seed >> 32), r);
            float u1 = ((float)(r[0] >> 8) + 0.5f) * 0x1p-24f, u2 = ((float)(r[1] >> 8) + 0.5f) * 0x1p-24f;
            float rad = sqrtf(-2.0f * __logf(u1));
            n0 = (double)(rad * __cosf(6.2831853f * u2));
            n1 = (double)(rad * __sinf(6.2831853f * u2));
        }
    }
    __syncthreads();
    if (act) {
        // simulations.py:303-326
        const double cvx = v.x + P.noise_amp * n0 + fp.x + fw.x;
        const double cvy = v.y + P.noise_amp * n1 + fp.y + fw.y;
        const double ax = (ds.x - cvx) / P.relaxation, ay = (ds.y - cvy) / P.relaxation;
        const double dt = P.dt, dt2 = dt * dt;
        nx = p.x + cvx * dt + 0.5 * ax * dt2;
        ny = p.y + cvy * dt + 0.5 * ay * dt2;
        double nvx = cvx + ax * dt, nvy = cvy + ay * dt;
        const double nrm = sqrt(nvx * nvx + nvy * nvy);
        if (!(nrm < P.v_max)) {
            const double sc = P.v_max / nrm;
            nvx *= sc;
            nvy *= sc;
        }
        S.ptime[k] = at0 + dt;
        // the caller's state arrays, in agent order: five scattered 8-byte stores per agent (at 500 000 agents more than half
        // of this launch).  lazy_state: only for an agent that LEAVES in this step -- its slot is given up below; everybody
        // else's state is scattered by k_flush_state when somebody needs the arrays (the re-sort, a read, another entry point)
        if (SEQ) {
            S.seq_state[k] = 2;
            if (!(fabs(nx - p.x) <= seq_margin && fabs(ny - p.y) <= seq_margin)) S.seq_ctl[4] = 1;
        }
        // exit test against the agent's own targets, strict (pedestrians.py:132-136)
        const double* rt = (tab_lds ? l_rect : rect_tab) + (size_t)dk.key * RECT_TAB;
        const int nrect = (int)rt[0];
        for (int q = 0; q < nrect; ++q)
            if (fabs(nx - rt[1 + 4 * q]) < rt[3 + 4 * q] * 0.5 && fabs(ny - rt[2 + 4 * q]) < rt[4 + 4 * q] * 0.5) out = true;
        if (!lazy || out) {
            A.x[i] = nx;
            A.y[i] = ny;
            A.vx[i] = nvx;
            A.vy[i] = nvy;
            if (A.agent_time) A.agent_time[i] = at0 + dt;
        }
        if (out) {
            A.status[i] = 0;
            C.ppos[k] = make_double2(1e300, 1e300);
            C.pvel[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            C.phead[k] = make_float2(1.f, 0.f);
            C.pidx[k] = -1;
        } else {
            float4 pv;
            float2 ph;
            make_record(P, nvx, nvy, pv, ph);
            C.ppos[k] = make_double2(nx, ny);
            C.pvel[k] = pv;
            C.phead[k] = ph;
            S.pvd[k] = make_double2(nvx, nvy);
        }
    }
    const bool stay = live && !out;
    if (SEQ) {      // agents that still wait for an earlier neighbour: the host launches rounds until none is left
        const unsigned long long m_pend = __ballot(live && sstate == 0);
        if (lane == 0 && m_pend) atomicAdd(&S.seq_ctl[seq_round & 3], __popcll(m_pend));
        // release: the agents advanced here no longer hold back the later agents within reach (the positions and the reach
        // of k_seq_count: earlier minus later, the same operands in the same order)
        unsigned long long am = __ballot(act);
        if (am) {
            const int rank_me = S.seq_rank[k];
            const double2 p0 = S.seq_pos0[k];
            const double reach = P.cutoff + 2.0 * seq_margin, reach2 = reach * reach;
            const SeqCand cand(S, it);
            const int nc = cand.size();
            while (am) {
                const int f = __ffsll((long long)am) - 1;
                am &= am - 1;
                const int rf = __builtin_amdgcn_readlane(rank_me, f);
                const double fx0 = readlane_f64(p0.x, f), fy0 = readlane_f64(p0.y, f);
                for (int e = 0; e < nc; e += 4) {               // (four tiles' loads in flight per trip: the chain is latency)
                    int jj[4], rj[4];
                    double2 pj[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        jj[u] = (e + u < nc) ? cand.tile(e + u) * 64 + lane : -1;
                        rj[u] = (jj[u] >= 0) ? S.seq_rank[jj[u]] : 0x7FFFFFFF;
                        pj[u] = (jj[u] >= 0) ? S.seq_pos0[jj[u]] : make_double2(1e300, 1e300);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double dx = fx0 - pj[u].x, dy = fy0 - pj[u].y;
                        if (rj[u] > rf && rj[u] != 0x7FFFFFFF && dx * dx + dy * dy < reach2) atomicSub(&S.seq_dep[jj[u]], 1);
                    }
                }
            }
        }
    }
    // counters: only exits touch them (one word takes ~88 atomics per us: 1563 tiles adding to one counter every step
    // would cost more than the rest of this kernel)
    const unsigned long long m_out = __ballot(out);
    // tile box of the new positions + validity of the neighbour lists
    const double4 tb = tile_boxes(stay ? nx : INFINITY, stay ? ny : INFINITY, stay ? nx : -INFINITY,
                                  stay ? ny : -INFINITY, lane, it, C.qbox);
    const double mnx = tb.x, mny = tb.y, mxx = tb.z, mxy = tb.w;
    if (lane == 0) {
        if (it == 0 && (!SEQ || seq_round == 0)) {  // IndexError-band flag raised by the lookups of k_forces
            const int e = err0;
            A.cnt_out[C_ERR] = e;
            S.ctrl[C_ERR + 8] = 0;
            if (e && S.ctrl[C_RS_ERRA] == 0) {
                S.ctrl[C_RS_ERRA] = e;
                S.ctrl[C_RS_ERRS] = A.step;
            }
        }
        if (m_out) {
            const int n = __popcll(m_out);
            atomicAdd(&A.cnt_out[C_EXIT], n);
            atomicSub(&A.cnt_out[C_ACTIVE], n);
            if (atomicSub(&S.ctrl[C_NACTIVE], n) == n) S.ctrl[C_RS_DONE] = A.step + 1;     // the last ones just left
        }
        C.tile_box[it] = make_double4(mnx, mny, mxx, mxy);
        const double D = P.drift;
        if (mnx <= mxx && (mnx < b0.x - D || mny < b0.y - D || mxx > b0.z + D || mxy > b0.w + D)) S.ctrl[C_INVALID] = 1;
    }
}


// lazy_state: the packed snapshot -> the caller's state arrays (agent order), for every agent that is still inside (an agent
// that left was written when it did)
__global__ void __launch_bounds__(256) k_flush_state(Scratch S, double* __restrict__ x, double* __restrict__ y,
                                                     double* __restrict__ vx, double* __restrict__ vy,
                                                     double* __restrict__ agent_time) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= S.Np) return;
    const int i = S.snap.pidx[k];
    if (i < 0) return;
    const double2 p = S.snap.ppos[k], v = S.pvd[k];
    x[i] = p.x;
    y[i] = p.y;
    vx[i] = v.x;
    vy[i] = v.y;
    if (agent_time) agent_time[i] = S.ptime[k];
}

// ---------------------------------------------------------------------------------- sequential (Gauss-Seidel) mode
// The reference updates the agents ONE BY ONE in a fresh random permutation, and an agent sees the already advanced
// positions / velocities / status of those visited before it (simulations.py:271, pedestrians.py:140-162).  That order
// is reproduced exactly by level scheduling: agent i may be advanced as soon as every agent that comes EARLIER in the
// permutation and lies within cutoff + 2 M of it has been advanced (M = bound on one step's displacement, checked by
// k_finish<true>): two agents advanced in the same round are then farther apart than the cutoff before and after their
// moves, so neither contributes to the other, and each reads exactly the state the one-by-one loop would have shown
// it -- new records of the earlier neighbours, old records of the later ones.  Once per step k_seq_count gives every agent
// the number of earlier agents within reach; a round = k_seq_go (the agents whose count has reached zero: their pair sums
// from the LIVE records) + k_finish<true> (advance them in place and take them off the counts of the later agents within
// reach); the rounds' number is the depth of the permutation's dependency graph (tens to hundreds at crowd densities), so
// this mode costs that many launch pairs per step: it exists for trajectory-level parity with the reference, the
// synchronous step is the fast path.
__global__ void k_seq_invperm(const int32_t* __restrict__ perm, int N, int32_t* __restrict__ inv) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < N) {
        const int i = perm[r];
        if (i >= 0 && i < N) inv[i] = r;
    }
}

__global__ void k_seq_init(Scratch S, const int32_t* __restrict__ inv) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) S.seq_ctl[0] = S.seq_ctl[1] = S.seq_ctl[2] = S.seq_ctl[3] = S.seq_ctl[4] = S.seq_ctl[5] = 0;
    if (k >= S.Np) return;
    const int i = S.snap.pidx[k];
    S.seq_rank[k] = (i >= 0) ? inv[i] : 0x7FFFFFFF;
    S.seq_state[k] = (i >= 0) ? 0 : 2;
    S.seq_dep[k] = 0;
    S.seq_pos0[k] = S.snap.ppos[k];
}

struct SeqLds {
    double2 pos[64];
    int32_t rank[64];
};

// Once per step: dep[i] = number of agents that come EARLIER in the permutation and lie within reach = cutoff + 2 M of agent
// i at the start of the step -- those it has to wait for.  One wavefront per tile, one lane per agent, the candidate tiles'
// positions and ranks staged through LDS.
__global__ void __launch_bounds__(64) k_seq_count(GcfmParams P, Scratch S, double margin) {
    __shared__ SeqLds L;
    const int it = blockIdx.x, lane = threadIdx.x, k = it * 64 + lane;
    const int rank_i = S.seq_rank[k];
    const double2 pi = S.seq_pos0[k];
    const bool live = S.seq_state[k] == 0;
    const double reach = P.cutoff + 2.0 * margin, reach2 = reach * reach;
    const SeqCand cand(S, it);
    int dep = 0;
    for (int e = 0; e < cand.size(); ++e) {
        const int jt = cand.tile(e);
        __syncthreads();
        L.pos[lane] = S.seq_pos0[jt * 64 + lane];
        L.rank[lane] = S.seq_rank[jt * 64 + lane];
        __syncthreads();
        if (!live) continue;
        for (int q = 0; q < 64; ++q) {
            const double2 pa = L.pos[q];
            const double dx = pa.x - pi.x, dy = pa.y - pi.y;
            dep += (L.rank[q] < rank_i && dx * dx + dy * dy < reach2) ? 1 : 0;
        }
    }
    S.seq_dep[k] = dep;
}

// One round, first launch: the agents that wait for nobody any more (dep == 0) get their pair sums from the LIVE records
// -- new records of the earlier neighbours, old ones of the later -- and are marked for k_finish<true>.  A tile without such
// an agent leaves after two loads; for a free agent the wavefront's lanes take the 64 records of a candidate tile each (two
// tiles per trip, one in each half of the packed registers), and the lanes' float64 partial sums are added up across the
// wavefront.  (Round 2 re-derived "who is free" from all pairs in every round and walked the candidate tiles one after the
// other behind two barriers each: 230 us per level at 5 000 agents.)
__global__ void __launch_bounds__(64) k_seq_go(GcfmParams P, Scratch S, int round) {
    const Snap& C = S.snap;
    const int it = blockIdx.x, lane = threadIdx.x, k = it * 64 + lane;
    if (it == 0 && lane == 0) S.seq_ctl[(round + 1) & 3] = 0;          // next round's pending counter
    // the levels are looped on the DEVICE: the host enqueues rounds a whole step ahead, and a round that finds nobody
    // pending after the one before it returns at once (one load per wavefront) -- no host round trip per level
    if (round > 0 && S.seq_ctl[(round - 1) & 3] == 0) return;
    const int i = C.pidx[k];
    const bool free_i = i >= 0 && S.seq_state[k] == 0 && S.seq_dep[k] == 0;
    unsigned long long fm = __ballot(free_i);
    if (!fm) return;
    const double2 pi = C.ppos[k];
    const float4 vi = C.pvel[k];
    const float2 hi = C.phead[k];
    float inv_vdes = 0.f, ib2_i = 0.f;
    if (i >= 0) {
        const double vd = S.pdk[k].v_des;
        inv_vdes = (float)(1.0 / vd);
        const double b = P.b_max - (P.b_max - P.b_min) * fmin((double)vi.z / vd, 1.0);
        ib2_i = (float)(1.0 / (b * b));
    }
    const float b_max = (float)P.b_max, db = (float)(P.b_max - P.b_min);
    const float neg_inv_eta_log2e = (float)(-1.4426950408889634 / P.eta);
    const bool unit_dir = (P.fix & OC_FIX_PAIR_FORCE) != 0;
    const SeqCand cand(S, it);
    const int nc = cand.size();
    double my_fx = 0.0, my_fy = 0.0;
    while (fm) {
        const int f = __ffsll((long long)fm) - 1;
        fm &= fm - 1;
        // agent f's record, wave-uniform
        const double px = readlane_f64(pi.x, f), py = readlane_f64(pi.y, f);
        const float vxf = readlane_f32(vi.x, f), vyf = readlane_f32(vi.y, f), sf = readlane_f32(vi.z, f), ia2f = readlane_f32(vi.w, f);
        const float hxf = readlane_f32(hi.x, f), hyf = readlane_f32(hi.y, f);
        const float ib2f = readlane_f32(ib2_i, f), ivdf = readlane_f32(inv_vdes, f);
        double fx = 0.0, fy = 0.0;
        for (int e0 = 0; e0 < nc; e0 += 4) {                // (four tiles' positions in flight per trip: the chain is latency)
            int jt4[4];
            double2 p4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                jt4[u] = (e0 + u < nc) ? cand.tile(e0 + u) : -1;
                p4[u] = (jt4[u] >= 0) ? C.ppos[jt4[u] * 64 + lane] : make_double2(1e300, 1e300);
            }
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int jtA = jt4[2 * h2], jtB = jt4[2 * h2 + 1];
            const double2 pa = p4[2 * h2], pb = p4[2 * h2 + 1];
            const double dxA = pa.x - px, dyA = pa.y - py, dxB = pb.x - px, dyB = pb.y - py;
            const double d2A = dxA * dxA + dyA * dyA, d2B = dxB * dxB + dyB * dyB;
            const bool okA = d2A <= P.cut_d2max && d2A > 0.0, okB = d2B <= P.cut_d2max && d2B > 0.0;
            if (!__any(okA || okB)) continue;
            float4 va = make_float4(0.f, 0.f, 0.f, 1.f), vb = va;
            float2 ha = make_float2(1.f, 0.f), hb = ha;
            if (okA) {
                va = C.pvel[jtA * 64 + lane];
                ha = C.phead[jtA * 64 + lane];
            }
            if (okB) {
                vb = C.pvel[jtB * 64 + lane];
                hb = C.phead[jtB * 64 + lane];
            }
            f2v gx, gy;
            pair_term2((f2v){okA ? (float)dxA : 1.f, okB ? (float)dxB : 1.f}, (f2v){okA ? (float)dyA : 0.f, okB ? (float)dyB : 0.f},
                       vxf, vyf, sf, hxf, hyf, ia2f, ib2f, ivdf, (f2v){va.x, vb.x}, (f2v){va.y, vb.y}, (f2v){va.z, vb.z},
                       (f2v){va.w, vb.w}, (f2v){ha.x, hb.x}, (f2v){ha.y, hb.y}, b_max, db, neg_inv_eta_log2e, unit_dir, gx, gy,
                       okA, okB);
            fx += (double)gx.x + (double)gx.y;
            fy += (double)gy.x + (double)gy.y;
          }
        }
        fx = wsum(fx);
        fy = wsum(fy);
        if (lane == f) {
            my_fx = fx;
            my_fy = fy;
        }
    }
    if (free_i) {
        xs(S, S.f_pair, S.ts_pair, k) = make_double2(my_fx, my_fy);
        S.seq_state[k] = 1;
    }
}

// ---------------------------------------------------------------------------------- diagnostics (single-shot exports)
__global__ void k_lookup_diag(GcfmParams P, Scratch S, const KeyDev* __restrict__ keys,
                              const KeyStep* __restrict__ ksteps, int step, int32_t* err) {
    const Snap& C = S.snap;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= S.Np) return;
    int i = C.pidx[k];
    double2 d = make_double2(0.0, 0.0);
    if (i >= 0) {
        double2 p = C.ppos[k];
        const PDK dk = S.pdk[k];
        d = lookup(P, keys[dk.key], ksteps[dk.key], p.x, p.y, dk.v_des, step, i, err);
    }
    S.des[k] = d;
}

__global__ void k_wall_force_diag(GcfmParams P, Scratch S, const Axis X0, const Axis Y0) {
    const Snap& C = S.snap;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= S.Np) return;
    int i = C.pidx[k];
    double2 out = make_double2(0.0, 0.0);
    if (i >= 0) {
        double2 p = C.ppos[k];
        const double2 v = S.pvd[k];
        out = wall_force_of(P, S.widx[k], p.x, p.y, v.x, v.y, S.pdk[k].v_des, X0, Y0);
    }
    S.f_wall[k] = out;
}

template <typename TV>
__global__ void k_scatter(const TV* __restrict__ src, const int32_t* __restrict__ pidx, int Np, TV* __restrict__ dst) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Np) return;
    int i = pidx[k];
    if (i >= 0) dst[i] = src[k];
}

// ---------------------------------------------------------------------------------- wall map build
// lr[j][r] = (largest site column <= j or -1, smallest site column >= j+1 or Nx) of grid row r.  One thread per row.
__global__ void k_wallmap(const float* __restrict__ V, float vmin, int Ny, int Nx, int2* __restrict__ lr) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= Ny) return;
    const float* row = V + (size_t)r * Nx;
    int last = -1;
    for (int j = 0; j < Nx; ++j) {
        if (row[j] == vmin) last = j;
        lr[(size_t)j * Ny + r].x = last;
    }
    last = Nx;
    for (int j = Nx - 1; j >= 0; --j) {
        lr[(size_t)j * Ny + r].y = last;          // nearest site at a column >= j+1
        if (row[j] == vmin) last = j;
    }
}

// gap8[j][b] = (min over rows 8b..8b+7 of j - left site column) | (min of right site column - j) << 16  (see GAP_NONE)
__global__ void __launch_bounds__(256) k_gap8(const int2* __restrict__ lr, int Nx, int Ny, int32_t* __restrict__ gap8) {
    const int nb8 = (Ny + 7) >> 3;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)Nx * nb8) return;
    const int j = (int)(t / nb8), b = (int)(t % nb8);
    int gl = GAP_SIDE_NONE, gr = GAP_SIDE_NONE;
    for (int r = 8 * b; r < min(8 * b + 8, Ny); ++r) {
        const int2 c = lr[(size_t)j * Ny + r];
        if (c.x >= 0) gl = min(gl, min(j - c.x, GAP_SIDE_NONE - 1));
        if (c.y < Nx) gr = min(gr, min(c.y - j, GAP_SIDE_NONE - 1));
    }
    gap8[t] = (int32_t)((unsigned)gl | ((unsigned)gr << 16));
}

// ---------------------------------------------------------------------------------- host helpers
Axis axis_x(const oc_handle* h) {
    return Axis{h->X0, h->cfg.room_length / (h->cfg.Nx - 1), h->cfg.room_length, h->cfg.Nx, h->axes_linspace ? 1 : 0};
}
Axis axis_y(const oc_handle* h) {
    return Axis{h->Y0, h->cfg.room_height / (h->cfg.Ny - 1), h->cfg.room_height, h->cfg.Ny, h->axes_linspace ? 1 : 0};
}

GcfmParams make_params(const oc_handle* h) {
    const oc_config& c = h->cfg;
    GcfmParams P;
    P.Ny = c.Ny; P.Nx = c.Nx;
    P.dx = c.dx; P.dy = c.dy; P.L = c.room_length; P.H = c.room_height; P.dt = c.dt;
    P.a_min = c.a_min; P.tau_a = c.tau_a; P.b_min = c.b_min; P.b_max = c.b_max; P.eta = c.eta;
    P.eta_walls = c.eta_walls; P.cutoff = c.repulsion_cutoff; P.v_max = c.v_max; P.relaxation = c.relaxation;
    P.noise_amp = c.noise_amp; P.wall_radius = c.wall_search_radius;
    P.cell = c.repulsion_cutoff > 0 ? 0.5 * c.repulsion_cutoff : 2.0;
    P.ncx = (int)(c.room_length / P.cell) + 1;
    P.ncy = (int)(c.room_height / P.cell) + 1;
    while (P.ncx > 16384 || P.ncy > 16384) {        // keep the curve index inside the 30 sorted key bits
        P.cell *= 2.0;
        P.ncx = (int)(c.room_length / P.cell) + 1;
        P.ncy = (int)(c.room_height / P.cell) + 1;
    }
    P.mu = c.mu;
    P.lim = c.lim;
    P.fix = h->fix_flags;
    P.diag = (uint32_t)h->diag;
    P.horder = 1;
    while ((1 << P.horder) < std::max(P.ncx, P.ncy)) ++P.horder;
    P.drift = 1.5 * c.v_max * c.dt * h->resort_every + 0.1;
    {   // sqrt is correctly rounded and monotone: walk down from c^2 to the first d2 whose root is below the cutoff
        double t = P.cutoff * P.cutoff;
        for (int q = 0; q < 8; ++q) t = std::nextafter(t, INFINITY);
        while (!(std::sqrt(t) < P.cutoff)) t = std::nextafter(t, 0.0);
        P.cut_d2max = t;
    }
    return P;
}

int flush_state_impl(oc_handle* h, hipStream_t s);

int carve(oc_handle* h, int N, Scratch* S, bool diag = false) {
    int Np = ((N + 63) / 64) * 64;
    if (Np == 0) Np = 64;
    const int nt = Np / 64;
    const size_t per_snap = sizeof(double2) + sizeof(float4) + sizeof(float2) + sizeof(int32_t);
    const size_t per = 2 * sizeof(uint32_t) + 2 * sizeof(int32_t) + per_snap + sizeof(double2) + sizeof(PDK) +
                       3 * sizeof(double2) + sizeof(long long) + sizeof(int32_t) + sizeof(double) + sizeof(int32_t) + sizeof(double2);
    size_t need = (per + 2 * sizeof(int32_t)) * (size_t)Np + (6 * sizeof(double4) + (NB_MAX + 3) * sizeof(int32_t)) * (size_t)nt +
                  1024 + 16384;
    if (need > h->gcfm_bytes) {
        if (h->state_dirty) {           // lazy_state: the snapshot about to be dropped is ahead of the caller's arrays
            // (on the stream of the step that left the snapshot: torch's side streams do not wait for the NULL stream)
            const hipStream_t fs = h->dirty_stream;
            int rcf = flush_state_impl(h, fs);
            if (rcf) return rcf;
            OC_HIP_CHECK(hipStreamSynchronize(fs));
        }
        if (h->gcfm_buf) OC_HIP_CHECK(hipFree(h->gcfm_buf));
        h->gcfm_buf = nullptr;
        OC_HIP_CHECK(hipMalloc(&h->gcfm_buf, need));
        OC_HIP_CHECK(hipMemset(h->gcfm_buf, 0, need));
        h->gcfm_bytes = need;
        h->packed_valid = false;
        h->work_valid = false;
        h->wprev_valid = false;
    }
    char* p = (char*)h->gcfm_buf;
    auto take = [&](size_t bytes) {
        char* q = p;
        p += (bytes + 255) & ~(size_t)255;
        return (void*)q;
    };
    // the control words come FIRST: their address must not depend on the agent count (oc_run_status reads them whatever
    // set of agents the last call packed); diagnostic exports use their own copy (prepare) and never touch the run's
    S->ctrl = (int32_t*)take(64);
    int32_t* ctrl_diag = (int32_t*)take(64);
    S->seq_ctl = (int32_t*)take(64);
    if (diag) S->ctrl = ctrl_diag;
    S->snap.ppos = (double2*)take(sizeof(double2) * Np);
    S->snap.pvel = (float4*)take(sizeof(float4) * Np);
    S->snap.phead = (float2*)take(sizeof(float2) * Np);
    S->snap.pidx = (int32_t*)take(sizeof(int32_t) * Np);
    S->snap.tile_box = (double4*)take(sizeof(double4) * nt);
    S->snap.qbox = (double4*)take(sizeof(double4) * 4 * nt);
    S->pvd = (double2*)take(sizeof(double2) * Np);
    S->pdk = (PDK*)take(sizeof(PDK) * Np);
    S->ptime = (double*)take(sizeof(double) * Np);
    S->f_pair = (double2*)take(sizeof(double2) * Np);
    S->f_wall = (double2*)take(sizeof(double2) * Np);
    S->des = (double2*)take(sizeof(double2) * Np);
    S->ts_pair = S->ts_wall = S->ts_des = 64;
    S->x_tpr = S->x_blk = 0;
    S->tile_box0 = (double4*)take(sizeof(double4) * nt);
    S->widx = (long long*)take(sizeof(long long) * Np);
    S->wprev = (int32_t*)take(sizeof(int32_t) * Np);
    if (diag || !h->wall_warm) S->wprev = nullptr;      // (single-shot exports search cold; "wall_warm" = 0 switches it off)
    S->keys_a = (uint32_t*)take(sizeof(uint32_t) * Np);
    S->keys_b = (uint32_t*)take(sizeof(uint32_t) * Np);
    S->vals_a = (int32_t*)take(sizeof(int32_t) * Np);
    S->vals_b = (int32_t*)take(sizeof(int32_t) * Np);
    S->nb_list = (int32_t*)take(sizeof(int32_t) * (NB_MAX + 1) * (size_t)nt);
    S->tile_work = (int32_t*)take(sizeof(int32_t) * nt);
    S->tile_order = (int32_t*)take(sizeof(int32_t) * nt);
    S->seq_rank = (int32_t*)take(sizeof(int32_t) * Np);
    S->seq_state = (int32_t*)take(sizeof(int32_t) * Np);
    S->seq_dep = (int32_t*)take(sizeof(int32_t) * Np);
    S->seq_pos0 = (double2*)take(sizeof(double2) * Np);
    S->Np = Np;
    S->ntiles = nt;
    return OC_OK;
}

int upload_keys(oc_handle* h, hipStream_t s) {
    if (!h->key_tables_dirty && h->key_tables_dev) return OC_OK;
    size_t nk = h->fields.size();
    std::vector<KeyDev> host(nk);
    for (size_t k = 0; k < nk; ++k) {
        KeyDev& d = host[k];
        memset(&d, 0, sizeof(d));
        d.vx = h->fields[k].vx;
        d.vy = h->fields[k].vy;
        d.dtype = h->fields[k].dtype;
        d.nslice_alloc = h->fields[k].nslice_alloc;
        d.nt_opt = (h->fields[k].vx || h->fields[k].ring_data) ? h->fields[k].nt_opt : 0;
        if (h->fields[k].ring_data) {
            const oc_field_desc& f = h->fields[k];
            const size_t ts = f.ring_state_dtype == OC_F32 ? 4 : 8;
            const int rows = f.rows < 0 ? h->cfg.Ny : f.rows;
            const char* m = (const char*)f.ring_meta;
            d.ring_data = f.ring_data;
            d.ring_slots = m + oc_ring_slots_off();
            d.ring_slices = m + oc_ring_slices_off(ts, f.ring_slots);
            d.ring_first = (const int32_t*)(m + oc_ring_flags_off(ts, f.ring_slots, f.nslice_alloc));
            d.ring_slot_elems = (int64_t)OC_RING_VALS * rows * h->cfg.Nx;
            d.ring_state_dtype = f.ring_state_dtype;
            d.ring_ntile = oc_ring_ntile(rows, h->cfg.Nx);
            d.ring_tnbx = (h->cfg.Nx + OC_RING_STX - 1) / OC_RING_STX;
        }
        d.row0 = h->fields[k].row0;
        d.rows = h->fields[k].rows;
        d.own_lo = h->fields[k].own_lo;
        d.own_hi = h->fields[k].own_hi;
        d.lr = (const int2*)h->wallmaps[k].left;
        d.edt = (const float*)h->wallmaps[k].right;
        d.gap8 = (const int32_t*)h->wallmaps[k].gap8;
        d.off = h->wallmaps[k].off;
        d.edt_cap = h->wallmaps[k].edt_cap;
        d.nrect = h->rects[k].n;
        memcpy(d.rect, h->rects[k].r, sizeof(d.rect));
    }
    std::vector<double> rt(nk * RECT_TAB, 0.0);       // compact exit-rectangle table for k_finish
    for (size_t k = 0; k < nk; ++k) {
        rt[k * RECT_TAB] = (double)h->rects[k].n;
        memcpy(&rt[k * RECT_TAB + 1], h->rects[k].r, sizeof(double) * 4 * OC_MAX_TARGET_RECTS);
    }
    if (!h->key_tables_dev) {
        OC_HIP_CHECK(hipMalloc(&h->key_tables_dev, sizeof(KeyDev) * nk));
        OC_HIP_CHECK(hipMalloc(&h->rect_tab_dev, sizeof(double) * nk * RECT_TAB));
        OC_HIP_CHECK(hipMalloc(&h->key_steps_dev, sizeof(KeyStep) * nk));
        OC_HIP_CHECK(hipMemset(h->key_steps_dev, 0, sizeof(KeyStep) * nk));
    }
    h->ring_step_ready = -1;            // (the per-step ring table was resolved from the old descriptors)
    h->any_ring = false;
    for (size_t k = 0; k < nk; ++k) h->any_ring = h->any_ring || h->fields[k].ring_data != nullptr;
    // synchronous copy of a small table: happens only when a field / target / wall map changed
    OC_HIP_CHECK(hipStreamSynchronize(s));
    OC_HIP_CHECK(hipMemcpy(h->key_tables_dev, host.data(), sizeof(KeyDev) * nk, hipMemcpyHostToDevice));
    OC_HIP_CHECK(hipMemcpy(h->rect_tab_dev, rt.data(), sizeof(double) * nk * RECT_TAB, hipMemcpyHostToDevice));
    h->key_tables_dirty = false;
    return OC_OK;
}

int check_agents(const oc_handle* h, const oc_agents* a) {
    OC_REQUIRE(h && a, "null argument");
    OC_REQUIRE(a->N >= 0 && a->N <= h->cfg.max_agents, "agent count exceeds max_agents");
    OC_REQUIRE(a->N == 0 || (a->x && a->y && a->vx && a->vy && a->v_des && a->status && a->key), "null agent array");
    return OC_OK;
}

bool wallmaps_ready(const oc_handle* h) {
    for (const auto& w : h->wallmaps)
        if (!w.built) return false;
    return true;
}

// re-sort + full pack of the current state + neighbour lists + dispatch order
int repack(oc_handle* h, const oc_agents* a, Scratch* S, const GcfmParams& P, hipStream_t s) {
    const int N = a->N;
    {   // lazy_state: the re-sort reads the caller's arrays -- bring them up to date first (same stream: ordered)
        int rcf = flush_state_impl(h, s);
        if (rcf) return rcf;
    }
    // the searches' incumbents belong to THESE agents, arrays and wall maps: anything else starts cold (a plain re-sort of
    // the same agents keeps them: they are indexed by agent)
    if (S->wprev != nullptr && (!h->wprev_valid || h->packed_x != a->x || h->packed_N != N)) {
        OC_HIP_CHECK(hipMemsetAsync(S->wprev, 0xFF, sizeof(int32_t) * (size_t)S->Np, s));
        h->wprev_valid = true;
    }
    OC_HIP_CHECK(hipMemsetAsync(S->ctrl, 0, CTRL_VOLATILE_BYTES, s));
    if (N > 0) {
        hipLaunchKernelGGL(k_cell_keys, dim3((N + 255) / 256), dim3(256), 0, s, a->x, a->y, a->status, N, P, S->keys_a,
                           S->vals_a);
        // (only the bits a key can have are sorted: 15 - 17 instead of 30 at the BASELINE rooms -- half the radix passes)
        int rc = oc_sort_pairs(h, S->keys_a, S->keys_b, S->vals_a, S->vals_b, N, std::min(30, 2 * P.horder + 1), s);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_pack, dim3(S->ntiles), dim3(64), 0, s, a->x, a->y, a->vx, a->vy, a->v_des, a->status, a->key,
                       S->vals_b, N, P, *S, (const double*)a->agent_time);
    const bool have_work = h->work_valid && h->work_tiles == S->ntiles;
    if (!have_work) OC_HIP_CHECK(hipMemsetAsync(S->tile_work, 0, sizeof(int32_t) * S->ntiles, s));
    hipLaunchKernelGGL(k_tile_lists, dim3(S->ntiles), dim3(64), 0, s, P, *S, have_work ? 1 : 0);
    {   // dispatch order of the tiles: decreasing work
        int rc = oc_sort_pairs(h, S->keys_a, S->keys_b, S->vals_a, S->tile_order, S->ntiles, 24, s);
        if (rc) return rc;
    }
    h->steps_since_sort = 0;
    h->packed_x = a->x;
    h->packed_N = N;
    h->packed_seq = false;
    return OC_OK;
}

int flush_state_impl(oc_handle* h, hipStream_t s) {
    if (!h->state_dirty) return OC_OK;
    h->state_dirty = false;
    const oc_agents& a = h->dirty_agents;
    if (a.N <= 0 || h->gcfm_buf == nullptr) return OC_OK;
    // (the packed snapshot these arrays are behind is still in the scratch: it is only ever overwritten by a repack, which
    // comes through here first, or dropped by a reallocation, which carve() precedes by a flush)
    Scratch S;
    int rc = carve(h, a.N, &S);
    if (rc) return rc;
    hipLaunchKernelGGL(k_flush_state, dim3((S.Np + 255) / 256), dim3(256), 0, s, S, a.x, a.y, a.vx, a.vy, a.agent_time);
    OC_HIP_CHECK(hipGetLastError());
    return OC_OK;
}

int prepare(oc_handle* h, const oc_agents* a, Scratch* S, GcfmParams* P, hipStream_t s) {
    int rc = carve(h, a->N, S, true);      // single-shot diagnostic exports: own control words
    if (rc) return rc;
    *P = make_params(h);
    if ((rc = upload_keys(h, s))) return rc;
    return repack(h, a, S, *P, s);
}

}   // namespace

int oc_flush_state(oc_handle* h, hipStream_t s) { return flush_state_impl(h, s); }

// ---------------------------------------------------------------------------------- C ABI
extern "C" int oc_agents_sync(oc_handle* h, void* stream) {
    OC_REQUIRE(h, "null handle");
    OC_HIP_CHECK(hipSetDevice(h->device));
    return flush_state_impl(h, (hipStream_t)stream);
}

extern "C" int oc_wallmap_build(oc_handle* h, int key, const float* V, float vmin, void* stream) {
    OC_REQUIRE(h && V && key >= 0 && key < (int)h->wallmaps.size(), "bad argument");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    oc_wallmap& w = h->wallmaps[key];
    if (!w.left) OC_HIP_CHECK(hipMalloc((void**)&w.left, sizeof(int2) * (size_t)h->ncell));
    if (!w.right) OC_HIP_CHECK(hipMalloc((void**)&w.right, sizeof(float) * (size_t)h->ncell));
    const oc_config& c = h->cfg;
    hipLaunchKernelGGL(k_wallmap, dim3((c.Ny + 63) / 64), dim3(64), 0, s, V, vmin, c.Ny, c.Nx, (int2*)w.left);
    const size_t n8 = (size_t)c.Nx * ((c.Ny + 7) / 8);
    if (!w.gap8) OC_HIP_CHECK(hipMalloc((void**)&w.gap8, sizeof(int32_t) * n8));
    hipLaunchKernelGGL(k_gap8, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, (const int2*)w.left, c.Nx, c.Ny,
                       (int32_t*)w.gap8);
    // cap: no per-agent search radius exceeds a(v_max)^2 + eta_w (1 + v_max) * 33 (wall_radius_of)
    double amax = c.a_min + c.tau_a * c.v_max;
    w.edt_cap = std::max(amax * amax, amax) + c.eta_walls * (1.0 + c.v_max) * 33.0 + 1.0;
    if (c.wall_search_radius > 0) w.edt_cap = std::max(w.edt_cap, c.wall_search_radius + 1.0);
    hipLaunchKernelGGL(k_edt_nodes, dim3((unsigned)(((size_t)c.Nx * c.Ny + 63) / 64)), dim3(64), 0, s, make_params(h),
                       (const int2*)w.left, (const int32_t*)w.gap8, axis_x(h), axis_y(h), w.edt_cap, (float*)w.right);
    OC_HIP_CHECK(hipGetLastError());
    w.off = (double)vmin * 10e3;
    w.built = true;
    h->key_tables_dirty = true;
    h->packed_valid = false;
    h->wprev_valid = false;
    return OC_OK;
}

static int require_wallmaps(oc_handle* h) {
    for (size_t k = 0; k < h->wallmaps.size(); ++k)
        if (!h->wallmaps[k].built) {
            oc_set_error("wall map of key %d not built (oc_wallmap_build)", (int)k);
            return OC_ERR_STATE;
        }
    return OC_OK;
}

extern "C" int oc_agents_changed(oc_handle* h) {
    OC_REQUIRE(h, "null handle");
    h->state_dirty = false;             // (lazy_state: the caller has written the arrays -- call oc_agents_sync BEFORE doing so)
    h->packed_valid = false;
    h->wprev_valid = false;
    return OC_OK;
}

extern "C" int oc_field_lookup(oc_handle* h, const oc_agents* a, int32_t simu_step, double* des, int32_t* err_flag,
                               void* stream) {
    int rc = check_agents(h, a);
    if (rc) return rc;
    OC_REQUIRE(des, "null output");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Scratch S;
    GcfmParams P;
    if ((rc = prepare(h, a, &S, &P, s))) return rc;
    h->packed_valid = false;
    OC_HIP_CHECK(hipMemsetAsync(des, 0, sizeof(double) * 2 * (size_t)a->N, s));
    if (err_flag) OC_HIP_CHECK(hipMemsetAsync(err_flag, 0, sizeof(int32_t) * 2, s));
    if (h->any_ring) {
        hipLaunchKernelGGL(k_ring_step, dim3(1), dim3(64 * (((int)h->fields.size() + 63) / 64)), 0, s,
                           (const KeyDev*)h->key_tables_dev, (int)h->fields.size(), simu_step, (KeyStep*)h->key_steps_dev);
        h->ring_step_ready = simu_step;
    }
    hipLaunchKernelGGL(k_lookup_diag, dim3((S.Np + 255) / 256), dim3(256), 0, s, P, S, (const KeyDev*)h->key_tables_dev,
                       (const KeyStep*)h->key_steps_dev, simu_step, err_flag);
    hipLaunchKernelGGL(k_scatter<double2>, dim3((S.Np + 255) / 256), dim3(256), 0, s, S.des, S.snap.pidx, S.Np,
                       (double2*)des);
    OC_HIP_CHECK(hipGetLastError());
    return OC_OK;
}

extern "C" int oc_wall_force(oc_handle* h, const oc_agents* a, int64_t* index_out, double* force_out, void* stream) {
    int rc = check_agents(h, a);
    if (rc) return rc;
    if ((rc = require_wallmaps(h))) return rc;
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Scratch S;
    GcfmParams P;
    if ((rc = prepare(h, a, &S, &P, s))) return rc;
    h->packed_valid = false;
    hipLaunchKernelGGL(k_wall, dim3(S.ntiles), dim3(64), 0, s, P, S, (const KeyDev*)h->key_tables_dev, axis_x(h),
                       axis_y(h));
    hipLaunchKernelGGL(k_wall_force_diag, dim3((S.Np + 255) / 256), dim3(256), 0, s, P, S, axis_x(h), axis_y(h));
    if (index_out) {
        OC_HIP_CHECK(hipMemsetAsync(index_out, 0xFF, sizeof(int64_t) * (size_t)a->N, s));
        hipLaunchKernelGGL(k_scatter<long long>, dim3((S.Np + 255) / 256), dim3(256), 0, s, S.widx, S.snap.pidx,
                           S.Np, (long long*)index_out);
    }
    if (force_out) {
        OC_HIP_CHECK(hipMemsetAsync(force_out, 0, sizeof(double) * 2 * (size_t)a->N, s));
        hipLaunchKernelGGL(k_scatter<double2>, dim3((S.Np + 255) / 256), dim3(256), 0, s, S.f_wall, S.snap.pidx,
                           S.Np, (double2*)force_out);
    }
    OC_HIP_CHECK(hipGetLastError());
    return OC_OK;
}

extern "C" int oc_pair_force(oc_handle* h, const oc_agents* a, double* force_out, void* stream) {
    int rc = check_agents(h, a);
    if (rc) return rc;
    OC_REQUIRE(force_out, "null output");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Scratch S;
    GcfmParams P;
    if ((rc = prepare(h, a, &S, &P, s))) return rc;
    h->packed_valid = false;
    hipLaunchKernelGGL(k_pair, dim3((S.ntiles + 3) / 4), dim3(256), 0, s, P, S);
    OC_HIP_CHECK(hipMemsetAsync(force_out, 0, sizeof(double) * 2 * (size_t)a->N, s));
    hipLaunchKernelGGL(k_scatter<double2>, dim3((S.Np + 255) / 256), dim3(256), 0, s, S.f_pair, S.snap.pidx, S.Np,
                       (double2*)force_out);
    OC_HIP_CHECK(hipGetLastError());
    return OC_OK;
}

__global__ void k_err_publish(const int32_t* __restrict__ ctrl, double* __restrict__ slot) {
    slot[0] = (double)ctrl[C_ERR + 8];              // (the tail of this rank's block of the exchange buffer)
}

// layout of the sharded step's exchange buffer (oc_hip.h: oc_step_comm): `world` blocks of XB doubles -- a rank's share of
// the tiles, [pair x 64][wall x 64]([des x 64]) double2 per tile, then 8 doubles of which the first is the rank's
// IndexError-band word -- followed by one double2 per slot: the desired velocities when the floor fields are sharded
struct XchgLayout {
    int tpr;            // tiles per rank
    int td;             // doubles per tile in a block: 384 with the desired velocities, 256 without
    int64_t block;      // doubles per rank as exchanged (tpr * td + 8)
    int64_t block_max;  // doubles between two ranks' blocks in memory (tpr * 384 + 8)
};
static XchgLayout xchg_layout(int ntiles, int world, bool des_separate) {
    XchgLayout x;
    x.tpr = (ntiles + world - 1) / world;
    x.td = des_separate ? 256 : 384;
    x.block = (int64_t)x.tpr * x.td + 8;
    x.block_max = (int64_t)x.tpr * 384 + 8;
    return x;
}

// one step, or one rank's part of it: k_forces on the tiles of `share`, [the caller's sum over the ranks], k_finish
static int launch_step(oc_handle* h, const oc_agents* a, Scratch& S, const GcfmParams& P, int32_t simu_step,
                       const double* noise, uint64_t seed, int32_t* counters, TileShare share,
                       const oc_step_comm* comm, bool des_separate, hipStream_t s) {
    StepArgs A;
    A.x = a->x; A.y = a->y; A.vx = a->vx; A.vy = a->vy;
    A.status = a->status;
    A.agent_time = a->agent_time;
    A.noise = noise;
    A.seed = seed;
    A.step = simu_step;
    A.cnt_out = counters ? counters : S.ctrl;
    const KeyDev* kd = (const KeyDev*)h->key_tables_dev;
    const int lazy = (h->lazy_state != 0 && comm == nullptr) ? 1 : 0;       // (the sharded step keeps the arrays current)
    const int ns = h->pair_split, tpb = 4 / ns, ntl = share.hi - share.lo;
    const int npair = (ntl + tpb - 1) / tpb;
    // prep blocks: one tile per wavefront while that takes at most `prep_blocks` blocks; beyond, k tiles per wavefront
    // and the fewest blocks that do it with k -- every prep wavefront then carries the same number of tiles and the
    // blocks free their slots together (C4: 1563 tiles, k = 2, 196 blocks: 7 750 steps/s against 7 690 with 256 blocks,
    // where half of the wavefronts carry one tile and half two; 176 blocks -- some wavefronts with three -- 7 440)
    int nprep = std::max((S.ntiles + 3) / 4, 1);
    if (nprep > h->prep_blocks) {
        const int k = (S.ntiles + 4 * h->prep_blocks - 1) / (4 * h->prep_blocks);
        nprep = (S.ntiles + 4 * k - 1) / (4 * k);
    }
    auto kf = P.fix ? (ns == 4 ? k_forces<4, true> : (ns == 2 ? k_forces<2, true> : k_forces<1, true>))
                    : (ns == 4 ? k_forces<4, false> : (ns == 2 ? k_forces<2, false> : k_forces<1, false>));
    KeyStep* const ksteps = (KeyStep*)h->key_steps_dev;
    const int nkeys = (int)h->fields.size();
    if (h->any_ring && h->ring_step_ready != (int64_t)simu_step) {     // (the previous step's k_finish resolves it otherwise)
        hipLaunchKernelGGL(k_ring_step, dim3(1), dim3(64 * ((nkeys + 63) / 64)), 0, s, kd, nkeys, simu_step, ksteps);
        h->ring_step_ready = simu_step;
    }
    oc_timer_begin(h, "gcfm_forces", s);
    hipLaunchKernelGGL(kf, dim3(nprep + npair), dim3(256), 0, s, P, S, kd, (const KeyStep*)ksteps, simu_step, axis_x(h),
                       axis_y(h), nprep, share, A.cnt_out);
    oc_timer_end(h, "gcfm_forces", s);
    h->work_valid = share.by_work != 0;     // (a rank's share measures only its own tiles)
    h->work_tiles = S.ntiles;
    const double* err_slots = nullptr;
    int n_err_slots = 0;
    int64_t err_stride = 1;
    if (comm) {
        // the ranks' shares: ONE all-gather of equal blocks (a rank sends its block once and receives the others': the bytes
        // of the whole buffer cross the wire once, against twice for the sum-all-reduce of a zero-padded buffer it replaces)
        const XchgLayout X = xchg_layout(S.ntiles, comm->world, des_separate);
        err_slots = comm->xchg + (int64_t)X.tpr * X.td;
        err_stride = X.block;
        n_err_slots = comm->world;
        hipLaunchKernelGGL(k_err_publish, dim3(1), dim3(1), 0, s, S.ctrl,
                           comm->xchg + (int64_t)comm->rank * X.block + (int64_t)X.tpr * X.td);
        if (comm->comm_slot >= 0) {     // RCCL from here, on the launch stream (comm.hip): no host callback
            int rcc = oc_comm_allgather_f64(h, comm->comm_slot, comm->xchg, X.block, s);
            if (rcc) return rcc;
        } else if (comm->allgather(comm->user, comm->xchg, X.block) != 0) {
            oc_set_error("oc_gcfm_step_sharded: the allgather callback failed");
            return OC_ERR_STATE;
        }
        if (des_separate) {     // desired velocities answered by the holders of the field rows: zero elsewhere, summed
            if (comm->comm_slot >= 0) {
                int rcc = oc_comm_allreduce_sum_f64(h, comm->comm_slot, (double*)S.des, 2 * (int64_t)S.Np, s);
                if (rcc) return rcc;
            } else if (comm->allreduce_sum(comm->user, (double*)S.des, 2 * (int64_t)S.Np) != 0) {
                oc_set_error("oc_gcfm_step_sharded: the allreduce_sum callback failed");
                return OC_ERR_STATE;
            }
        }
    }
    oc_timer_begin(h, "gcfm_finish", s);
    constexpr int FIN_WPB = 4;
    hipLaunchKernelGGL((k_finish<false, FIN_WPB>), dim3((S.ntiles + FIN_WPB - 1) / FIN_WPB), dim3(64 * FIN_WPB), 0, s, P, S,
                       (const double*)h->rect_tab_dev,
                       (int)h->fields.size(), A, 0, 0.0, err_slots, n_err_slots, err_stride, kd,
                       h->any_ring ? ksteps : (KeyStep*)nullptr, lazy);
    if (lazy) {
        h->state_dirty = true;
        h->dirty_agents = *a;
        h->dirty_stream = s;
    }
    if (h->any_ring) h->ring_step_ready = (int64_t)simu_step + 1;
    oc_timer_end(h, "gcfm_finish", s);
    OC_HIP_CHECK(hipGetLastError());
#ifdef OC_DIAG_BUILD
    if (getenv("OC_WALL_COUNTS") && (simu_step % 100) == 50) {
        int32_t c[16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(c, S.seq_ctl, sizeof(c), hipMemcpyDeviceToHost);
        fprintf(stderr, "[wall counts up to step %d] wavefronts %d: rounds %.2f, visit trips %.2f per wavefront; lanes searching %.1f per "
                "wavefront, block visits %.2f per searching lane\n", simu_step, c[10], (double)c[8] / c[10], (double)c[9] / c[10],
                (double)c[11] / c[10], (double)c[12] / std::max(c[11], 1));
        (void)hipMemset(S.seq_ctl + 8, 0, 5 * sizeof(int32_t));
    }
#endif
    h->packed_valid = true;
    h->packed_seq = false;          // (held to the synchronous step's drift bound only)
    ++h->steps_since_sort;
    ++h->prof_tick;
    return OC_OK;
}

extern "C" int oc_gcfm_step(oc_handle* h, const oc_agents* a, int32_t simu_step, const double* noise, uint64_t seed,
                            int32_t* counters, void* stream) {
    int rc = check_agents(h, a);
    if (rc) return rc;
    if ((rc = require_wallmaps(h))) return rc;
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Scratch S;
    if ((rc = carve(h, a->N, &S))) return rc;
    GcfmParams P = make_params(h);
    if ((rc = upload_keys(h, s))) return rc;
    // the packed snapshot written by the previous step is reused unless the caller's arrays changed
    if (!h->packed_valid || h->packed_x != a->x || h->packed_N != a->N || h->steps_since_sort >= h->resort_every) {
        if ((rc = repack(h, a, &S, P, s))) return rc;
    }
    const TileShare all = {0, S.ntiles, 1, 0, 0};
    return launch_step(h, a, S, P, simu_step, noise, seed, counters, all, nullptr, false, s);
}


// One SEQUENTIAL step (see k_seq_go): the caller's permutation and noise, rounds until every active agent has moved.
extern "C" int oc_gcfm_step_sequential(oc_handle* h, const oc_agents* a, int32_t simu_step, const int32_t* perm,
                                       const double* noise, uint64_t seed, int32_t* counters, int32_t* rounds_out,
                                       void* stream) {
    int rc = check_agents(h, a);
    if (rc) return rc;
    if ((rc = require_wallmaps(h))) return rc;
    OC_REQUIRE(perm || a->N == 0, "the sequential step needs the permutation");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if ((rc = flush_state_impl(h, s))) return rc;      // (this mode reads and writes the caller's arrays every round)
    Scratch S;
    if ((rc = carve(h, a->N, &S))) return rc;
    GcfmParams P = make_params(h);
    if ((rc = upload_keys(h, s))) return rc;
    const double margin = h->seq_margin;
    OC_REQUIRE(margin > 0 && margin < P.drift, "seq_margin must stay below the neighbour lists' drift allowance");
    // the neighbour lists must cover cutoff + 2 M around the LIVE boxes: they do while no box has drifted more than
    // drift - M from its sort-time box (k_finish<true> checks exactly that); a snapshot left by synchronous steps was
    // held to the weaker bound, so it is rebuilt
    if (!h->packed_valid || h->packed_x != a->x || h->packed_N != a->N || h->steps_since_sort >= h->resort_every ||
        !h->packed_seq) {
        if ((rc = repack(h, a, &S, P, s))) return rc;
    }
    h->packed_seq = true;
    StepArgs A;
    A.x = a->x; A.y = a->y; A.vx = a->vx; A.vy = a->vy;
    A.status = a->status;
    A.agent_time = a->agent_time;
    A.noise = noise;
    A.seed = seed;
    A.step = simu_step;
    A.cnt_out = counters ? counters : S.ctrl;
    const KeyDev* kd = (const KeyDev*)h->key_tables_dev;
    const int N = a->N;
    if (N > 0) {
        hipLaunchKernelGGL(k_seq_invperm, dim3((N + 255) / 256), dim3(256), 0, s, perm, N, S.vals_a);
    }
    hipLaunchKernelGGL(k_seq_init, dim3((S.Np + 255) / 256), dim3(256), 0, s, S, (const int32_t*)S.vals_a);
    // wall forces and desired velocities of ALL agents from the state at the start of the step: an agent's own record
    // does not change before its turn (the prep blocks of k_forces alone; they also reset the step's counters)
    {
        const int nprep = std::max((S.ntiles + 3) / 4, 1);
        const TileShare all = {0, S.ntiles, 0, 0, 1};      // (prep blocks only: they do the lookups)
        auto kf = P.fix ? k_forces<4, true> : k_forces<4, false>;
        if (h->any_ring && h->ring_step_ready != (int64_t)simu_step) {
            hipLaunchKernelGGL(k_ring_step, dim3(1), dim3(64 * (((int)h->fields.size() + 63) / 64)), 0, s, kd,
                               (int)h->fields.size(), simu_step, (KeyStep*)h->key_steps_dev);
            h->ring_step_ready = simu_step;
        }
        hipLaunchKernelGGL(kf, dim3(nprep), dim3(256), 0, s, P, S, kd, (const KeyStep*)h->key_steps_dev, simu_step, axis_x(h),
                           axis_y(h), nprep, all, A.cnt_out);
    }
    // every agent's count of earlier agents within reach, once per step (start-of-step positions: k_seq_init kept them)
    hipLaunchKernelGGL(k_seq_count, dim3(S.ntiles), dim3(64), 0, s, P, S, margin);
    GcfmParams Pf = P;
    Pf.drift = P.drift - margin;            // validity bound of the neighbour lists in this mode (see above)
    // ONE host synchronisation per step: as many rounds as the last step needed (+ a margin) are enqueued at once; rounds
    // past the last level return at once on the device.  Only if agents are still pending after them -- the permutation
    // of this step is deeper than the last one's -- more rounds follow.
    int rounds = 0, pending = N > 0 ? 1 : 0, levels = 0;
    int32_t ctl[8];
    int batch = std::max(8, h->seq_rounds_hint + std::max(4, h->seq_rounds_hint / 8));
    while (pending > 0) {
        for (int b = 0; b < batch; ++b, ++rounds) {
            hipLaunchKernelGGL(k_seq_go, dim3(S.ntiles), dim3(64), 0, s, P, S, rounds);
            hipLaunchKernelGGL((k_finish<true>), dim3(S.ntiles), dim3(64), 0, s, Pf, S, (const double*)h->rect_tab_dev,
                               (int)h->fields.size(), A, rounds, margin, (const double*)nullptr, 0, (int64_t)1, kd,
                               (KeyStep*)nullptr);
        }
        OC_HIP_CHECK(hipMemcpyAsync(ctl, S.seq_ctl, sizeof(ctl), hipMemcpyDeviceToHost, s));
        OC_HIP_CHECK(hipStreamSynchronize(s));
        pending = ctl[(rounds - 1) & 3];
        levels = ctl[5];
        if (ctl[4]) {
            oc_set_error("sequential step %d: an agent moved farther than seq_margin = %g m in one step; the level "
                         "schedule is not valid (raise it with oc_set_tuning(\"seq_margin_mm\"))", simu_step, margin);
            h->packed_valid = false;
            return OC_ERR_STATE;
        }
        if (rounds > N + batch) {
            oc_set_error("sequential step %d did not drain (%d agents pending)", simu_step, pending);
            h->packed_valid = false;
            return OC_ERR_STATE;
        }
        batch = std::max(8, rounds / 2);
    }
    h->seq_rounds_hint = levels;
    rounds = levels;                        // (what the caller is told: the levels, not the rounds enqueued)
    OC_HIP_CHECK(hipGetLastError());
    if (rounds_out) *rounds_out = rounds;
    h->packed_valid = true;
    h->work_valid = false;
    ++h->steps_since_sort;
    ++h->prof_tick;
    return OC_OK;
}

extern "C" int oc_run_status(oc_handle* h, int32_t* out4, int reset, void* stream) {
    OC_REQUIRE(h && out4, "null argument");
    if (!h->gcfm_buf) {                 // no step has run yet: nothing to report (the scratch is born zeroed)
        out4[0] = out4[1] = out4[2] = out4[3] = 0;
        return OC_OK;
    }
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Scratch S;
    int rc = carve(h, h->packed_N > 0 ? h->packed_N : 0, &S);
    if (rc) return rc;
    int32_t w[16];
    OC_HIP_CHECK(hipMemcpyAsync(w, S.ctrl, sizeof(w), hipMemcpyDeviceToHost, s));
    OC_HIP_CHECK(hipStreamSynchronize(s));
    out4[0] = w[C_NACTIVE];
    out4[1] = w[C_RS_DONE];
    out4[2] = w[C_RS_ERRA];
    out4[3] = w[C_RS_ERRS];
    if (reset) OC_HIP_CHECK(hipMemsetAsync(S.ctrl + C_RS_DONE, 0, 4 * sizeof(int32_t), s));
    return OC_OK;
}

extern "C" int64_t oc_padded_agents(int32_t N) {
    int64_t Np = ((int64_t)N + 63) / 64 * 64;
    return Np == 0 ? 64 : Np;
}

static int64_t xchg_total(int32_t N, int32_t world) {
    const int64_t Np = oc_padded_agents(N);
    const XchgLayout X = xchg_layout((int)(Np / 64), world > 0 ? world : 1, false);
    return (int64_t)(world > 0 ? world : 1) * X.block_max + 2 * Np;
}

extern "C" int64_t oc_xchg_doubles(int32_t N, int32_t world) { return xchg_total(N, world); }

extern "C" int oc_gcfm_step_sharded(oc_handle* h, const oc_agents* a, int32_t simu_step, const double* noise,
                                    uint64_t seed, int32_t* counters, const oc_step_comm* comm, void* stream) {
    int rc = check_agents(h, a);
    if (rc) return rc;
    if ((rc = require_wallmaps(h))) return rc;
    OC_REQUIRE(comm && comm->world >= 1 && comm->world <= OC_MAX_WORLD && comm->rank >= 0 && comm->rank < comm->world &&
               comm->xchg && (comm->allgather || comm->comm_slot >= 0), "bad oc_step_comm");
    OC_REQUIRE(comm->comm_slot < 0 || (comm->comm_slot < OC_MAX_COMMS && h->comms[comm->comm_slot].comm != nullptr &&
                                      h->comms[comm->comm_slot].rank == comm->rank &&
                                      h->comms[comm->comm_slot].world == comm->world),
               "oc_step_comm.comm_slot names no communicator of this rank / world (oc_comm_create)");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if ((rc = flush_state_impl(h, s))) return rc;
    Scratch S;
    if ((rc = carve(h, a->N, &S))) return rc;
    GcfmParams P = make_params(h);
    if ((rc = upload_keys(h, s))) return rc;
    // every rank re-sorts the same replicated state at the same steps: identical slot orders everywhere
    if (!h->packed_valid || h->packed_x != a->x || h->packed_N != a->N || h->steps_since_sort >= h->resort_every) {
        if ((rc = repack(h, a, &S, P, s))) return rc;
    }
    // who answers a lookup: with row blocks -- or with keys dealt to different ranks, where this rank holds no field of
    // some key -- every rank walks all tiles and the holder of the field rows answers (the desired velocities then travel
    // by their own zero-padded sum); with every field whole on every rank the tile's owner does, and they ride in its block
    bool rows = false;
    for (const auto& f : h->fields) rows = rows || f.rows >= 0 || (f.vx == nullptr && f.ring_data == nullptr);
    OC_REQUIRE(!rows || comm->allreduce_sum || comm->comm_slot >= 0, "sharded floor fields need the allreduce_sum callback");
    const XchgLayout X = xchg_layout(S.ntiles, comm->world, rows);
    // pair sums, wall forces (and desired velocities) of tile t live in the block of the rank that owns t
    double* base = comm->xchg;
    S.f_pair = (double2*)base;
    S.f_wall = S.f_pair + 64;
    S.ts_pair = S.ts_wall = X.td / 2;                   // (double2 units)
    S.x_tpr = X.tpr;
    S.x_blk = (int32_t)(X.block / 2);
    if (rows) {
        S.des = (double2*)(base + (int64_t)comm->world * X.block_max);
        S.ts_des = 64;
        OC_HIP_CHECK(hipMemsetAsync(S.des, 0, sizeof(double2) * (size_t)S.Np, s));
    } else {
        S.des = S.f_pair + 128;
        S.ts_des = X.td / 2;
    }
    // (tile t belongs to rank t / tpr and sits at r * X.block + (t % tpr) * td doubles: xs())
    const TileShare mine = {std::min(comm->rank * X.tpr, S.ntiles), std::min((comm->rank + 1) * X.tpr, S.ntiles), 0,
                            rows ? 1 : 0, rows ? 1 : 0};
    return launch_step(h, a, S, P, simu_step, noise, seed, counters, mine, comm, rows, s);
}

// The nearest-wall cell each agent's last step used (the warm-started search's result, kept as the next step's incumbent):
// int32 (N), -1 where nothing lay within the agent's search radius or no step has run.  For tests: it must equal what the
// cold search of oc_wall_force finds at the positions that step started from.
extern "C" int oc_wall_last_index(oc_handle* h, const oc_agents* a, int32_t* index_out, void* stream) {
    int rc = check_agents(h, a);
    if (rc) return rc;
    OC_REQUIRE(index_out, "null output");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Scratch S;
    if ((rc = carve(h, a->N, &S))) return rc;
    if (S.wprev == nullptr || !h->wprev_valid || h->packed_x != a->x || h->packed_N != a->N) {
        OC_HIP_CHECK(hipMemsetAsync(index_out, 0xFF, sizeof(int32_t) * (size_t)a->N, s));
        return OC_OK;
    }
    OC_HIP_CHECK(hipMemcpyAsync(index_out, S.wprev, sizeof(int32_t) * (size_t)a->N, hipMemcpyDeviceToDevice, s));
    return OC_OK;
}

