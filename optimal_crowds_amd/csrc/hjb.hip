// HJB solve on the room grid: Cole-Hopf-linearised equation integrated backwards in time with
// Dormand-Prince RK45 (SciPy's controller and dense output), reduced to a unit-or-zero direction field.
// Replaces reference optimals.py:124-206 (closure hjb :144-164, vels :168-186, fill loop :200-204) and the
// SciPy arithmetic it calls (scipy/integrate/_ivp/rk.py:14-180,552-575; common.py:63-134; ivp.py:598-723).
//
// Data layout (HBM): every grid array is row-major (Ny, Nx) of T (float = product, double = audit).
//   coef   : (V + g m)/(mu sigma^2) on free cells, NaN on walls  (make_coef)
//   y,K0..6: RK state / stage derivatives, carried as phi * 2^-e (e tracked on the host; the PDE is linear and
//            wall cells are frozen, so a power-of-two rescale is exact) -- keeps fp32 in range for e^{T/(mu s^2)}.
//   vx,vy  : (nslice, Ny-2, Nx-2) direction field, slice s <-> time s*T/(nt-1).
#include "oc_internal.h"
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace {

// Block slots of the chip held by persistent launches of THIS process (every handle, every host thread: Engine.solve_keys
// solves its keys concurrently).  A persistent grid waits at a barrier inside the kernel, so its blocks must all be resident:
// the launches of one process never ask for more slots than the chip has.  (Another process's kernels are not counted --
// the in-kernel deadline covers that case.)
std::atomic<int> g_persist_in_use{0};
int oc_persist_reserve(int want, int cap) {
    want = (want / 8) * 8;                      // (a multiple of 8: a block's items stay on its XCD)
    for (;;) {
        int cur = g_persist_in_use.load();
        const int got = (std::min(want, cap - cur) / 8) * 8;
        if (got < 8) return 0;
        if (g_persist_in_use.compare_exchange_weak(cur, cur + got)) return got;
    }
}
void oc_persist_release(int n) { g_persist_in_use.fetch_sub(n); }

// ---- Dormand-Prince tables (scipy rk.py:377-404), double on the host
const double RK_A[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
const double RK_B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
const double RK_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
const double RK_P[7][4] = {
    {1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
    {0, 0, 0, 0},
    {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
    {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
    {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408, 701980252875.0 / 199316789632},
    {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
    {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};


// ---- device copies of the tables: the step controller runs on the device (k_control)
__constant__ double dRK_A[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
__constant__ double dRK_B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
__constant__ double dRK_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
// ---- device-resident step controller (hjb variant 0) -------------------------------------------------------------
// SciPy's _step_impl (rk.py:111-176) and the t_eval bookkeeping of solve_ivp (ivp.py:706-723) restated as ONE thread of
// k_control: after the stage launches of an attempt it turns the error sums into accept / reject, the next step size,
// the list of t_eval points the accepted step covers and the coefficient tables of the NEXT attempt, all in this
// struct in device memory.  The stage, output and rescale kernels take their pointers' roles, coefficients and slice
// ranges from here, so an attempt is a fixed sequence of launches that the host enqueues without waiting for any of
// them: no host round trip per attempt (it cost 2-3 ms of a 16 ms solve), and in the row-block solve no host
// synchronisation around the collectives.  The host only watches `ticks` / `done` in a pinned mailbox to know when to
// stop enqueuing; attempts enqueued past the end find `done` set and return at once.
#define HJB_MAXS 24
template <typename T>
struct HjbCtl {
    double t, tf, Tfinal, h_abs, rtol, atol0, n_rms, te_step, lim;
    double hh, t_new;                   // the attempt about to run: step and its end time
    double out_t, out_hh, out_unscale;  // the step just accepted (k_out): its start, length and 2^expo
    double h_min, h_max;
    double h_abs0;                      // the initial step select_initial_step chose (k_init_step)
    double act_sum;                     // stage tiles with a non-zero error share, summed over the attempts (active-set statistics)
    int32_t nt, k_next;                 // t_eval = linspace(T, 0, nt); next index to emit
    int32_t parity;                     // y = Y[parity], y_new = Y[1 - parity], K0 = KX[parity], K6 = KX[1 - parity]
    int32_t expo;                       // state = phi * 2^-expo
    int32_t rejected, done, status;
    int32_t accepted, out_parity, k_lo, k_hi, do_scale;     // decision of the last attempt
    int32_t n_attempt, n_accept, n_rescale, n_tsteps, ticks, is_f32;
    T cA[3][6], cB[3][6], he[7];        // h * a_sj of launch A (stages 1-3) and B (stages 4, 5, y_new); h * E_j
    T atol_s, out_lim_s;
    // Range management without a pass over the arrays: after a step whose max |y_new| passed 2^64, y_new and K6 stay
    // in memory as they are and the NEXT step multiplies them by `in_scale` (an exact power of two) as it loads them
    // -- as y and K0, in its stage launches and in its k_out -- so everything it writes is in the new scale.
    T in_scale, out_in_scale;
    T u_bg;                             // active set: the far field's value of y in the current y buffer (a power of two)
    int32_t work_count;                 // active set: tiles the next stage launch has to look at (build_work_list)
    int32_t dense_mode;                 // active set: every tile has become active -- the rest of the solve runs without flags and list
    int32_t list_todo;                  // active set: k_control's order for the list builder (block 0 of k_out): 1 = build, 2 = go dense
    int32_t n_hx;                       // k_out's per-slice factors h x, h x^2, h x^3, h x^4 (0: more than HJB_MAXS slices,
    T hx[HJB_MAXS][4];                  // computed in the kernel)
    // row-block solves with the active set: the tile rows [0, band_lo) and [band_hi, tnby) touch rows RECEIVED from a neighbour
    // (their region reaches into the halo the exchange refreshes): activity arrives there from outside, so they are always on
    // the work list and test themselves every attempt.  Whole-grid solves: 0 and tnby (no such rows).
    int32_t band_lo, band_hi;
    int32_t sh_all_active, pad_;        // row-block solves: this rank's tiles ALL had an error share (k_reduce_local)
};

// first part of an attempt (rk.py:121-137): clamp the step, its end time ...
template <typename T>
__host__ __device__ inline void ctl_prepare_step(HjbCtl<T>& c, bool new_step) {
    const double direction = -1.0;
    const double min_step = 10 * fabs(nextafter(c.t, direction * INFINITY) - c.t);
    if (new_step) {
        if (c.h_abs < min_step) c.h_abs = min_step;
    } else if (c.h_abs < min_step) {
        c.status = OC_ERR_SOLVER;       // "step size too small"
        c.done = 1;
        return;
    }
    double hh = c.h_abs * direction;
    double t_new = c.t + hh;
    if (direction * (t_new - c.tf) > 0) t_new = c.tf;
    hh = t_new - c.t;
    c.h_abs = fabs(hh);
    c.hh = hh;
    c.t_new = t_new;
    c.atol_s = (T)(c.atol0 * ldexp(1.0, -c.expo));
}

// ... and entry e (0 .. 42) of its coefficient tables: h a_sj of stages 1-3 (cA), of stages 4, 5 and the y_new row (cB),
// h E_j (he).  On the device the entries are dealt to the lanes of k_control.
template <typename T>
__host__ __device__ inline void ctl_prepare_entry(HjbCtl<T>& c, int e) {
#if defined(__HIP_DEVICE_COMPILE__)
#define TAB_A dRK_A
#define TAB_B dRK_B
#define TAB_E dRK_E
#else
#define TAB_A RK_A
#define TAB_B RK_B
#define TAB_E RK_E
#endif
    const double hh = c.hh;
    if (e < 18) {
        const int q = e / 6, j = e % 6;
        c.cA[q][j] = j < 1 + q ? (T)(TAB_A[1 + q][j] * hh) : (T)0;
    } else if (e < 36) {
        const int q = (e - 18) / 6, j = (e - 18) % 6;
        c.cB[q][j] = q < 2 ? (j < 4 + q ? (T)(TAB_A[4 + q][j] * hh) : (T)0) : (T)(TAB_B[j] * hh);
    } else if (e < 43) {
        c.he[e - 36] = (T)(TAB_E[e - 36] * hh);
    }
#undef TAB_A
#undef TAB_B
#undef TAB_E
}

template <typename T>
__host__ __device__ inline void ctl_prepare_attempt(HjbCtl<T>& c, bool new_step) {     // (the host's first attempt)
    ctl_prepare_step(c, new_step);
    for (int e = 0; e < 43; ++e) ctl_prepare_entry(c, e);
}

template <typename T>
__host__ __device__ inline double ctl_te(const HjbCtl<T>& c, int k) {      // np.linspace(T, 0, nt)[k]
    return (k == c.nt - 1) ? 0.0 : (double)k * c.te_step + c.Tfinal;
}

template <typename T>
struct Lin8 {   // out = base + sum_j c[j] * K[j]
    const T* k[8];
    T c[8];
    int n;
};

// ------------------------------------------------------------------------------------------ kernels
template <typename T>
__global__ void k_make_coef(const float* __restrict__ V, const T* __restrict__ m, double g, double inv_musig2,
                            T* __restrict__ coef, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = V[i];
    if (v < 0.f) {
        coef[i] = (T)NAN;
    } else {
        double mm = m ? (double)m[i] : 0.0;
        coef[i] = (T)(((double)v + g * mm) * inv_musig2);
    }
}

// hjb(t, phi, m) at one cell (optimals.py:154-160): a_lap (N + S + W + E - 4 C) - coef C, the one expression every
// variant evaluates (4 C is exact, so the fused form of the sum rounds like the reference's)
template <typename T>
__device__ __forceinline__ T rhs_value(T up, T dn, T lft, T rgt, T C, T a_lap, T cf) {
    const T sum = fma_t((T)-4, C, up + dn + lft + rgt);
    return a_lap * sum - cf * C;
}

template <typename T>
__device__ __forceinline__ T rhs_point(const T* __restrict__ p, const T* __restrict__ coef, int i, int j, int Ny,
                                       int Nx, T a_lap) {
    int64_t c = (int64_t)i * Nx + j;
    T cf = coef[c];
    if (cf != cf) return (T)0;                        // wall: optimals.py:162
    int iu = (i == 0) ? 1 : i - 1;                    // mirror ghosts: optimals.py:149-152
    int id = (i == Ny - 1) ? Ny - 2 : i + 1;
    int jl = (j == 0) ? 1 : j - 1;
    int jr = (j == Nx - 1) ? Nx - 2 : j + 1;
    T C = p[c];
    return rhs_value(p[(int64_t)iu * Nx + j], p[(int64_t)id * Nx + j], p[(int64_t)i * Nx + jl], p[(int64_t)i * Nx + jr], C,
                     a_lap, cf);
}

template <typename T>
__global__ void k_rhs(const T* __restrict__ p, const T* __restrict__ coef, T* __restrict__ out, int Ny, int Nx,
                      T a_lap) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (i >= Ny || j >= Nx) return;
    out[(int64_t)i * Nx + j] = rhs_point(p, coef, i, j, Ny, Nx, a_lap);
}

template <typename T>
__global__ void k_lincomb(const T* __restrict__ base, Lin8<T> L, T* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T acc = (T)0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (j < L.n) acc = fma_t(L.c[j], L.k[j][i], acc);
    out[i] = base[i] + acc;
}

template <typename T>
__global__ void k_fill(T* __restrict__ p, T v, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

template <typename T>
__global__ void k_scale(T* __restrict__ a, T* __restrict__ b, T s, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        a[i] *= s;
        b[i] *= s;
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}

// block (256 threads) reduction of (sum, max) -> partial[2*block]
__device__ __forceinline__ void block_reduce_store(double s, double m, double* __restrict__ partial) {
    __shared__ double ls[4], lm[4];
    s = wave_sum(s);
    m = wave_max(m);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        ls[w] = s;
        lm[w] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = ls[0] + ls[1] + ls[2] + ls[3];
        partial[2 * blockIdx.x + 1] = fmax(fmax(lm[0], lm[1]), fmax(lm[2], lm[3]));
    }
}

// mode 0: sum (a/scale)^2, scale = atol + |y| rtol                      (select_initial_step d0/d1)
// mode 1: sum ((a-b)/scale)^2                                            (d2)
template <typename T, int MODE>
__global__ void k_norm_scaled(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ y, T atol,
                              T rtol, int64_t n, double* __restrict__ partial) {
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        T sc = atol + fabs(y[i]) * rtol;
        T v = (MODE == 0) ? a[i] / sc : (a[i] - b[i]) / sc;
        s += (double)v * (double)v;
    }
    block_reduce_store(s, 0.0, partial);
}

// One cell's term of the error norm, error / scale (scipy rk.py:146-147, common.py:63-66): the IEEE quotient in every
// variant.  (The hardware reciprocal -- two instructions where the division expands to ten -- was tried for float: the
// attempt sequence of a stability-limited solve is sensitive to the last bit of this norm, the full-size comparisons with
// the forced-step oracle moved by their whole tolerance, and a denormal scale at the end of fp32's range gave 0 * inf.)
template <typename T>
__device__ __forceinline__ T err_quot(T e, T sc) { return e / sc; }

// error norm of a step (scipy rk.py:146-147) + max |y_new| for range management
template <typename T>
__global__ void k_err_norm(const T* __restrict__ y, const T* __restrict__ ynew, Lin8<T> E /* h*E_j */, T atol,
                           T rtol, int64_t n, double* __restrict__ partial) {
    double s = 0.0, mx = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        T e = (T)0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < E.n) e = fma_t(E.c[j], E.k[j][i], e);
        T yn = ynew[i];
        T sc = atol + fmax(fabs(y[i]), fabs(yn)) * rtol;
        T v = err_quot(e, sc);
        s += (double)v * (double)v;
        mx = fmax(mx, (double)fabs(yn));
    }
    block_reduce_store(s, mx, partial);
}

// `mailbox` (optional): pinned host memory mapped into the device.  The result is also stored there, followed by the
// sequence number `seq`, so that the host can read it by polling three words instead of a copy + stream synchronise.
__global__ void __launch_bounds__(1024) k_reduce_final(const double* __restrict__ partial, int nblocks,
                                                        double* __restrict__ out, volatile double* mailbox = nullptr,
                                                        double seq = 0.0) {
    // the host waits for this kernel: all loads of a thread are issued before the first is used (4 per round; the
    // partials were written by other XCDs a moment ago, so each round trip is a full memory latency)
    const double2* __restrict__ p2 = reinterpret_cast<const double2*>(partial);
    double s = 0.0, m = 0.0;
    const int nt = blockDim.x;
    for (int i0 = threadIdx.x; i0 < nblocks; i0 += 4 * nt) {
        double2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i0 + u * nt < nblocks) ? p2[i0 + u * nt] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s += v[u].x;
            m = fmax(m, v[u].y);
        }
    }
    __shared__ double ls[16], lm[16];
    s = wave_sum(s);
    m = wave_max(m);
    const int w = threadIdx.x >> 6, nw = (nt + 63) >> 6;
    if ((threadIdx.x & 63) == 0) {
        ls[w] = s;
        lm[w] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int q = 0; q < nw; ++q) {
            a += ls[q];
            b = fmax(b, lm[q]);
        }
        out[0] = a;
        out[1] = b;
        if (mailbox) {
            mailbox[0] = a;
            mailbox[1] = b;
            __threadfence_system();         // the two values reach host memory before the sequence number does
            mailbox[2] = seq;
            __threadfence_system();
        }
    }
}

// The step controller: error norm of the attempt that just ran -> accept / reject, next step size, the t_eval points
// of an accepted step, range management, the next attempt's coefficients (scipy rk.py:139-176, ivp.py:706-723).
// parts: nparts pairs (sum of squared scaled errors, max |y_new|) -- the stage kernel's per-block partials, or the
// ranks' pairs after the all-gather of a row-block solve (summed in rank order on every rank: identical decisions).
// att_log[attempt]: bit 0 = a real attempt, bit 1 = accepted, bits 8.. = direction-field slices its k_out writes.
// mailbox (pinned host memory): ONE word = ticks (control launches so far) + 2^32 * done -- a single store, so the host
// never sees a torn pair and the kernel needs no system-scope fence.
// what the attempt's blocks wrote for the controller, read by the LAST block of the same launch (fused tail, SC1): every such
// word was stored `sc1` by one lane of its block before that block's ticket (MI355X_MICROARCH.md, inter-workgroup
// visibility: sc1 stores + drained + agent-scope add; the last adder reads with sc1 loads).  Plain loads behind a kernel boundary.
template <bool SC1>
__device__ __forceinline__ double ld_part(const double* p) {
    return SC1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool SC1>
__device__ __forceinline__ int32_t ld_flag(const int32_t* p) {
    return SC1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}

// Short work lists (the rule of a sparse solve: ~25 tiles of 1957 at 2048 x 2048): the fused tail reads what the attempt's
// blocks left BY LIST POSITION -- one 32-byte record per listed tile (error share, max |y_new|, tile | flag << 32) -- instead
// of every tile's partial pair and flag: one batch of TAIL_FAST_MAX loads where 2 x 1957 partials and 1957 flags were fetched,
// reduced and scanned (the tail was 9.8 of an attempt's 20 us).  Same sums, bit for bit: a thread adds the shares of the
// tiles = its number mod 256 in ascending order, exactly the terms of the full reduction that are not zero.
#define TAIL_REC_CAP 128                // records an attempt's blocks write (list positions 0 .. TAIL_REC_CAP - 1)
#define TAIL_FAST_MAX 96                // the fast tail's range of list lengths
template <typename T>
struct CtlLds {
    HjbCtl<T> c;                        // the controller block is worked on in LDS (one coalesced load, one coalesced store):
                                        // thread 0's serial code would otherwise pay a memory round trip per field
    double red_s[256];                  // the workers' shares of the error sum (reduced by wavefront 0: wave_tree_sum)
    unsigned long long m_bits;          // max |y_new| as the bits of a non-negative double (LDS atomic max)
    int na_cnt;                         // tiles with a non-zero error share
    int logw;                           // the attempt's log word
    int spec_k_hi, spec_nv, spec_n_hx;  // the accepted step's bookkeeping, formed by wavefront 1 beside the decision
    double spec_unscale;
    T spec_lim_s;
    int all_active;                     // every tile had an error share: all were computed, the next work list is every tile
    int fast_total;                     // fast tail: length of the list fast_list built
    double rec_s[TAIL_FAST_MAX], rec_m[TAIL_FAST_MAX];      // fast tail: the attempt's records
    int32_t rec_tile[TAIL_FAST_MAX], rec_flag[TAIL_FAST_MAX];
    int32_t slot_cnt[256];              // fast tail: records per worker slot (tile mod 256)
};

// The next attempt's work list from the attempt's records, by ONE wavefront (called beside the controller's serial part): the
// list is the 3 x 3 dilation of the tiles that were computed (flag 0) -- exactly build_work_list's rule "off the list iff
// trivial together with all eight neighbours", given that tiles off the list are trivial -- formed as a bitmap in LDS and
// written out in ascending tile order (the order build_work_list produces).  bits: (ntile + 31) / 32 LDS words.
__device__ __forceinline__ void fast_list(const int32_t* rec_tile, const int32_t* rec_flag, int nrec, int tnbx, int tnby,
                                          uint32_t* bits, int32_t* __restrict__ work_out, int* total_out, int lane) {
    const int ntile = tnbx * tnby, nw = (ntile + 31) >> 5, per = (nw + 63) >> 6;
    for (int w = lane; w < nw; w += 64) bits[w] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = lane; e < nrec; e += 64) {
        if (rec_flag[e] != 0) continue;
        const int t = rec_tile[e], bi = t / tnbx, bj = t - bi * tnbx;
#pragma unroll
        for (int di = -1; di <= 1; ++di)
#pragma unroll
            for (int dj = -1; dj <= 1; ++dj) {
                const int ni = bi + di, nj = bj + dj;
                if (ni >= 0 && ni < tnby && nj >= 0 && nj < tnbx) {
                    const int n = ni * tnbx + nj;
                    atomicOr(&bits[n >> 5], 1u << (n & 31));
                }
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int w_lo = min(lane * per, nw), w_hi = min(w_lo + per, nw);
    int cnt = 0;
    for (int w = w_lo; w < w_hi; ++w) cnt += __popc(bits[w]);
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    int off = incl - cnt;
    for (int w = w_lo; w < w_hi; ++w) {
        uint32_t b = bits[w];
        while (b) {
            const int q = __ffs((int)b) - 1;
            b &= b - 1u;
            work_out[off++] = 32 * w + q;
        }
    }
    if (lane == 63) *total_out = incl;
}

// Called by EVERY thread of the block (barriers inside); threads 0 .. 255 do the work -- the same reduction tree whatever the
// block's size, so the launch shape never changes a decision.  ctl_in: the attempt's controller block; ctl_out: where the
// next attempt's goes (the two alternate, so that a late block of the attempt's own launch never reads a block that the
// fused tail is already rewriting).
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// `beside`: called by EVERY thread between the controller's serial section (thread 0) and the barrier behind it -- the
// other wavefronts run it while thread 0 decides (the fused tail's list builder does its decision-independent part there)
// UNR: error shares in flight per worker thread and trip (the sums are formed in the same order whatever it is).  The
// stand-alone launch behind a sweep of every tile reads thousands of shares: 16 per trip there (4096 x 4096: 30 per thread,
// two round trips to the L2 instead of eight); the fused tail has the stage kernel's 64 registers and keeps 4.
#ifdef OC_DIAG_BUILD
__shared__ unsigned long long g_cb_stamp[8];    // diagnostic build: s_memrealtime inside control_body (thread 0)
#define CB_STAMP(k) do { if (threadIdx.x == 0) g_cb_stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CB_STAMP(k) do { } while (0)
#endif

// entry e (0 .. 42) of the attempt's coefficient tables as (table value, is it a non-zero entry): ctl_prepare_entry's
// factors, fetched BEFORE the decision is known (constant memory: a round trip) and scaled by the new step behind it
__device__ __forceinline__ double ctl_table_value(int e, bool& valid) {
    valid = false;
    double v = 0.0;
    if (e < 18) {
        const int q = e / 6, j = e % 6;
        valid = j < 1 + q;
        if (valid) v = dRK_A[1 + q][j];
    } else if (e < 36) {
        const int q = (e - 18) / 6, j = (e - 18) % 6;
        valid = q < 2 ? j < 4 + q : true;
        if (valid) v = q < 2 ? dRK_A[4 + q][j] : dRK_B[j];
    } else if (e < 43) {
        valid = true;
        v = dRK_E[e - 36];
    }
    return v;
}
template <typename T>
__device__ __forceinline__ void ctl_set_entry(HjbCtl<T>& c, int e, T v) {
    if (e < 18) c.cA[e / 6][e % 6] = v;
    else if (e < 36) c.cB[(e - 18) / 6][(e - 18) % 6] = v;
    else if (e < 43) c.he[e - 36] = v;
}

// the other lane of a quad / half row / row by DPP (no LDS round trip): v (op) partner, for both words of a double
__device__ __forceinline__ double dpp_partner(double v, const int ctrl_case) {
    const long long b = __double_as_longlong(v);
    int lo = (int)b, hi = (int)(b >> 32);
    switch (ctrl_case) {
        case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false); break;    // quad_perm [1,0,3,2]
        case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, false); break;    // quad_perm [2,3,0,1]
        case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xf, 0xf, false); break;  // row_half_mirror
        default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xf, 0xf, false); break; // row_mirror
    }
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double lane_value(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)b, lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// sum over the 64 lanes of a wavefront in a FIXED association (every caller of the controller gets the same bits): pairs inside
// quads, quads inside half rows, half rows inside rows -- four DPP steps, after which every lane of a row holds the row's sum --
// then (row 0 + row 1) + (row 2 + row 3).  (Six ds_bpermute round trips per word as __shfl_down steps.)
__device__ __forceinline__ double wave_tree_sum(double v) {
    v += dpp_partner(v, 0);
    v += dpp_partner(v, 1);
    v += dpp_partner(v, 2);
    v += dpp_partner(v, 3);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// err^(-1/5), the step controller's power (scipy rk.py: ERROR_EXPONENT = -1 / (error_estimator_order + 1)), for err in
// [1e-6, 1e6] -- beyond that range the controller's clamps (factor 10 / 0.2) decide without it.  The hardware's log2 / exp2 in
// float give a start value (relative error ~2e-7), two Newton steps on y^-5 = x in float64 (e -> 3 e^2) leave ~2 ulp: 30
// dependent instructions where pow() is 200, on the ONE thread the next attempt's every block waits for.
__device__ __forceinline__ double inv_fifth_root(double x) {
    double y = (double)__builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf((float)x));
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double y2 = y * y, y4 = y2 * y2, y5 = y4 * y;
        const double r = fma(-x, y5, 1.0);
        y = fma(y * 0.2, r, y);
    }
    return y;
}

// max over the 64 lanes of a wavefront (non-negative values; exact in any association): the same four DPP steps
__device__ __forceinline__ double wave_tree_max(double v) {
    v = fmax(v, dpp_partner(v, 0));
    v = fmax(v, dpp_partner(v, 1));
    v = fmax(v, dpp_partner(v, 2));
    v = fmax(v, dpp_partner(v, 3));
    return fmax(fmax(lane_value(v, 0), lane_value(v, 16)), fmax(lane_value(v, 32), lane_value(v, 48)));
}

template <typename T, bool SC1, typename HOOK = NoHook, int UNR = 4>
__device__ __forceinline__ void control_body(const HjbCtl<T>* __restrict__ ctl_in, HjbCtl<T>* __restrict__ ctl_out,
                                             const double* parts, int nparts, double* __restrict__ t_steps, int t_cap,
                                             int32_t* __restrict__ att_log, int log_cap, volatile double* mailbox,
                                             int active_set, OcRingSlot<T>* __restrict__ rslots,
                                             OcRingSlice<T>* __restrict__ rslices, int ring_cap, CtlLds<T>& L,
                                             bool store_out = true, bool ctl_loaded = false, HOOK beside = HOOK(),
                                             const double* recs = nullptr, int nrec = 0, bool fx = true, bool recs_loaded = false,
                                             double rs_in = 0.0, double rm_in = 0.0, long long rtb_in = 0) {
    // fx = false (LEAD kernels, every block but the leader): the decision is formed for this block's own use -- nothing is
    // written to memory (step log, ring records, attempt log, mailbox); recs_loaded: the caller fetched this thread's record
    // What lies between the attempt's last block and the next attempt's first load is a chain of dependent steps, so the
    // controller is laid out for LATENCY: everything that can be known before the decision is fetched or formed before it or
    // beside it, and thread 0's serial float64 part is the decision alone --
    //   * the reduction: per-thread shares (tile = thread mod 256, ascending), max and count by LDS atomics (exact in any
    //     order), the sum by ONE wavefront in a fixed DPP tree (wave_tree_sum);
    //   * the 43 table entries of the next attempt are fetched from constant memory up front and scaled behind the decision;
    //   * the ACCEPTED step's bookkeeping -- the t_eval points it covers (ivp.py:706-723), the dense-output factors, the ring's
    //     slot record and slice table, the step log -- depends on t and t + h only, which are known before the attempt runs:
    //     wavefront 1 forms it while thread 0 decides ("speculatively": what it writes to memory belongs to slot n_accept and
    //     to slices of this step's range, all rewritten by the attempt that is finally accepted there), thread 0 commits the
    //     scalars if the step is accepted.
    // Same values as the strictly serial form (round 3), field by field.
    HjbCtl<T>& c = L.c;
    int& all_active = L.all_active;
    const bool worker = threadIdx.x < 256;
    const int nth = 256;
    if (threadIdx.x == 0) {
        all_active = 0;
        L.m_bits = 0ull;
        L.na_cnt = 0;
    }
    if (nrec > 0 && worker) {
        L.slot_cnt[threadIdx.x] = 0;
        L.red_s[threadIdx.x] = 0.0;
    }
    constexpr int NW32 = (int)(sizeof(HjbCtl<T>) / 4);
    static_assert(sizeof(HjbCtl<T>) % 4 == 0, "controller block must be a whole number of words");
    if (worker && !ctl_loaded) {        // (the fused tail finds it in LDS already: copied when the block started)
        const uint32_t* src = reinterpret_cast<const uint32_t*>(ctl_in);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&c);
        for (int i = threadIdx.x; i < NW32; i += nth) dst[i] = src[i];
    }
    bool tab_valid = false;
    const double tab = (int)threadIdx.x < 43 ? ctl_table_value((int)threadIdx.x, tab_valid) : 0.0;
    // the attempt's error sums (issued together with the block's load; unused if the solve is over)
    double s = 0.0, m = 0.0, na = 0.0;
    double rs = rs_in, rm = rm_in;
    long long rtb = rtb_in;
    if (!recs_loaded && nrec > 0 && (int)threadIdx.x < nrec) {
        const double* r = recs + 4 * threadIdx.x;
        rs = ld_part<true>(r);
        rm = ld_part<true>(r + 1);
        rtb = __hip_atomic_load(reinterpret_cast<const long long*>(r + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                    // (the controller block and the words above are in LDS)
    CB_STAMP(5);
    // what wavefront 1 needs of the controller block for the accepted step's bookkeeping, read while nobody writes the block
    const bool spec_wave = ((int)threadIdx.x >> 6) == 1;
    double z_t = 0.0, z_tnew = 0.0, z_hh = 0.0, z_unscale = 0.0;
    T z_lim_s = (T)0, z_u = (T)0;
    int z_knext = 0, z_nt = 0, z_nacc = 0, z_ntsteps = 0, z_done = 1, z_dense = 0;
    if (spec_wave) {
        z_done = c.done;
        z_t = c.t; z_tnew = c.t_new; z_hh = c.hh;
        z_knext = c.k_next; z_nt = c.nt; z_nacc = c.n_accept; z_ntsteps = c.n_tsteps; z_dense = c.dense_mode;
        z_unscale = ldexp(1.0, c.expo);
        z_lim_s = (T)(c.lim * ldexp(1.0, -c.expo));
        z_u = c.u_bg * c.in_scale;      // what the far field reads in the buffer that is y once the step is accepted
    }
    if (nrec > 0) {
        // fast tail: the listed tiles' records (CtlLds) -- every other tile's share is (0, 0), which the full reduction adds
        // without changing a bit.  Worker slot t of the full reduction adds the shares of the tiles = t mod 256 in ascending
        // order from 0.0: with ONE listed tile in the slot (the rule: the list is a few dozen neighbouring tiles) that is the
        // tile's share itself, written by the record's thread; a slot with several is summed in list (= ascending tile) order
        // by the thread of the first of them.  Max and count by the LDS atomics below, exact in any order.  (A loop of every
        // worker over every record -- dependent LDS reads -- was 2.4 of the tail's 7.7 us.)
        const int my_slot = (int)(rtb & (long long)(nth - 1));
        if ((int)threadIdx.x < nrec) {
            L.rec_s[threadIdx.x] = rs;
            L.rec_m[threadIdx.x] = rm;
            L.rec_tile[threadIdx.x] = (int32_t)(rtb & 0xFFFFFFFFll);
            L.rec_flag[threadIdx.x] = (int32_t)(rtb >> 32);
            atomicAdd(&L.slot_cnt[my_slot], 1);
        }
        __syncthreads();
        CB_STAMP(6);
        if ((int)threadIdx.x < nrec) {
            if (L.slot_cnt[my_slot] == 1) {
                L.red_s[my_slot] = rs;
            } else {
                bool first = true;
                double acc = 0.0;
                for (int e = 0; e < nrec; ++e) {
                    if ((L.rec_tile[e] & (nth - 1)) == my_slot) {
                        if (e < (int)threadIdx.x) first = false;
                        acc += L.rec_s[e];
                    }
                }
                if (first) L.red_s[my_slot] = acc;
            }
        }
        // (max |y_new| and the count of tiles with an error share: formed by wavefront 0 from the records below -- as LDS
        // atomics of every record's thread on one word each they were 1.5 us)
    } else if (worker) {
        for (int i0 = threadIdx.x; i0 < nparts; i0 += UNR * nth) {
            double2 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int i = i0 + u * nth;
                v[u] = (i < nparts) ? make_double2(ld_part<SC1>(parts + 2 * i), ld_part<SC1>(parts + 2 * i + 1))
                                    : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                s += v[u].x;
                m = fmax(m, v[u].y);
                na += v[u].x > 0.0 ? 1.0 : 0.0;
            }
        }
    }
    CB_STAMP(7);
    if (worker && nrec == 0) {
        // (maximum and count: a DPP tree per worker wavefront, then FOUR LDS atomics -- one per worker and word they were
        // 1.5 us of every tail that sweeps the tiles' partials: small grids, dense attempts, k_control)
        L.red_s[threadIdx.x] = s;
        const double wm = wave_tree_max(m), wn = wave_tree_sum(na);         // (m >= 0, never NaN: fmax from 0; na a small integer)
        if ((threadIdx.x & 63) == 0) {
            if (wm > 0.0) atomicMax(&L.m_bits, (unsigned long long)__double_as_longlong(wm));   // (non-negative doubles order like their bits)
            if (wn > 0.0) atomicAdd(&L.na_cnt, (int)wn);
        }
    }
    CB_STAMP(0);
    __syncthreads();
    CB_STAMP(1);
    if (threadIdx.x < 64) {
        const int l = (int)threadIdx.x;
        const double esum = wave_tree_sum(((L.red_s[l] + L.red_s[l + 64]) + L.red_s[l + 128]) + L.red_s[l + 192]);
        double rec_max = 0.0;
        int rec_cnt = 0;
        if (nrec > 0) {
            static_assert(TAIL_FAST_MAX <= 128, "two records per lane");
            const bool h0 = l < nrec, h1 = l + 64 < nrec;
            const double m0 = h0 ? L.rec_m[l] : 0.0, m1 = h1 ? L.rec_m[l + 64] : 0.0;
            double mm = fmax(m0, m1);
            mm = mm > 0.0 ? mm : 0.0;                   // (a NaN never enters the maximum: as `if (m > 0) atomicMax`)
            rec_max = wave_tree_max(mm);
            rec_cnt = __popcll(__ballot(h0 && L.rec_s[l] > 0.0)) + __popcll(__ballot(h1 && L.rec_s[l + 64] > 0.0));
        }
        if (threadIdx.x == 0) {
            if (c.done) {                               // an attempt enqueued past the end of the solve
                c.accepted = 0;
                c.do_scale = 0;
                L.logw = 0;
            } else {
                double ymax = nrec > 0 ? rec_max : __longlong_as_double((long long)L.m_bits);
                // (active set: tiles off the work list took no part in the launch -- their share is 0 and the far field's u)
                if (active_set == 1 && c.work_count < nparts) ymax = fmax(ymax, fabs((double)c.u_bg * (double)c.in_scale));
                const double nact = nrec > 0 ? (double)rec_cnt : (double)L.na_cnt;
                c.act_sum += nact;
                // (active_set = 2: a row-block solve -- the pairs are the RANKS'; whether this rank's tiles are all active was
                // left by k_reduce_local)
                all_active = active_set == 2 ? c.sh_all_active : ((active_set != 0 && nact == (double)nparts) ? 1 : 0);
                const double err = sqrt(esum) / sqrt(c.n_rms);
                ++c.n_attempt;
                c.accepted = 0;
                c.do_scale = 0;
                L.logw = 1;
                if (!(err == err)) {
                    c.status = OC_ERR_SOLVER - 100;     // NaN error norm (state overflow)
                    c.done = 1;
                } else if (err < 1) {
                    double factor = (err < 1e-6) ? 10.0 : fmin(10.0, 0.9 * inv_fifth_root(err));    // (0.9 err^-0.2 > 10 below 5.9e-6)
                    if (c.rejected) factor = fmin(1.0, factor);
                    const double hstep = fabs(c.hh);
                    c.h_abs *= factor;
                    c.h_min = fmin(c.h_min, hstep);
                    c.h_max = fmax(c.h_max, hstep);
                    c.accepted = 1;
                    c.out_parity = c.parity;
                    c.parity ^= 1;
                    c.out_t = c.t;
                    c.out_hh = c.hh;
                    c.out_in_scale = c.in_scale;
                    c.u_bg *= c.in_scale;               // what the far field reads in the buffer that is y from now on
                    c.in_scale = (T)1;
                    c.t = c.t_new;
                    ++c.n_accept;
                    // (k_lo, k_next, k_hi, n_hx, out_unscale, out_lim_s, n_tsteps: committed below from wavefront 1's results)
                    if (c.is_f32 && ymax > 0x1p64) {    // range management: keep max |y| near 2^32
                        const int e = (int)floor(log2(ymax)) - 32;
                        c.in_scale = (T)ldexp(1.0, -e);
                        c.do_scale = 1;
                        c.expo += e;
                        ++c.n_rescale;
                    }
                    c.rejected = 0;
                    L.logw |= 2;
                    // (interpolant ring: the slot's record and the slice table are wavefront 1's; here only the capacity)
                    const bool ring_full = rslots != nullptr && c.n_accept - 1 >= ring_cap;
                    if (ring_full) {
                        c.status = OC_ERR_SOLVER - 200; // more accepted steps than ring slots: the caller retries with a larger ring
                        c.accepted = 0;
                        c.done = 1;
                    } else if (-(c.t - c.tf) >= 0) c.done = 1; // direction = -1
                    else ctl_prepare_step(c, true);
                } else {
                    c.h_abs *= (err > 1e6) ? 0.2 : fmax(0.2, 0.9 * inv_fifth_root(err));           // (0.9 err^-0.2 < 0.2 beyond 1845)
                    c.rejected = 1;
                    ctl_prepare_step(c, false);
                }
            }
            // ---- active set: the order for the list builder (every tile active for the rest of the solve -- decided on an
            // ACCEPTED attempt only: with the far-field substitution the buffer a rejected attempt leaves as y may still hold
            // tiles that were never written)
            c.list_todo = (active_set != 0 && !c.done) ? (((all_active && c.accepted) || c.dense_mode) ? 2 : 1) : 0;
        }
    } else if (spec_wave && !z_done) {
        // ---- the ACCEPTED step's bookkeeping, beside the decision.  Lane 0 of this wavefront walks the t_eval points ...
        const int l = (int)threadIdx.x & 63;
        int k_hi = z_knext, nv = 0;
        if (l == 0) {
            while (k_hi < z_nt && ctl_te(c, k_hi) >= z_tnew) {      // (ctl_te reads nt, te_step, Tfinal: never written here)
                nv += k_hi >= 1 ? 1 : 0;
                ++k_hi;
            }
        }
        k_hi = __builtin_amdgcn_readfirstlane(k_hi);
        nv = __builtin_amdgcn_readfirstlane(nv);
        const int k_lo = z_knext, n_hx = (k_hi - k_lo <= HJB_MAXS) ? k_hi - k_lo : 0;
        const int slot = z_nacc;                        // the slot of the step this attempt tries
        if (l == 0) {
            L.spec_k_hi = k_hi;
            L.spec_nv = nv;
            L.spec_n_hx = n_hx;
            L.spec_unscale = z_unscale;
            L.spec_lim_s = z_lim_s;
            if (fx && z_ntsteps < t_cap) t_steps[z_ntsteps] = z_tnew;
            if (fx && rslots != nullptr && slot < ring_cap) {     // what the step's dense output is made of stays in this slot
                OcRingSlot<T> rs_;
                rs_.out_t = z_t;
                rs_.out_hh = z_hh;
                rs_.out_unscale = z_unscale;
                rs_.k_lo = k_lo;
                rs_.k_hi = k_hi;
                rs_.dense = (active_set == 0 || z_dense != 0) ? 1 : 0;
                rs_.pad = 0;
                rs_.lim_s = z_lim_s;
                rs_.u = z_u;                            // the far field's y in the step's own scale
                rslots[slot] = rs_;
            }
        }
        // ... the lanes the dense-output factors of those points (rk.py:560-575: x = (t_eval - t_old) / h, h x^1..4) and the
        // ring's slice table (slice nt - 1 - k <-> t_eval[k], optimals.py:200-204; k = 0 is never looked up)
        for (int q = l; q < k_hi - k_lo; q += 64) {
            const int k = k_lo + q;
            const double x = (ctl_te(c, k) - z_t) / z_hh;
            double xp = x;
            T hq[4];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                hq[pp] = (T)(z_hh * xp);
                xp *= x;
            }
            if (q < n_hx) {
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) c.hx[q][pp] = hq[pp];
            }
            if (fx && rslices != nullptr && k >= 1 && slot < ring_cap) {
                OcRingSlice<T> e;
                e.slot = slot;
                e.pad = 0;
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) e.h[pp] = hq[pp];
                rslices[z_nt - 1 - k] = e;
            }
        }
    }
    CB_STAMP(2);
    beside();
    __syncthreads();
    CB_STAMP(3);
    // behind the decision: the accepted step's scalars are committed, the next attempt's 43 coefficients scaled
    if (threadIdx.x == 0) {
        if (c.accepted) {
            c.k_lo = c.k_next;
            c.k_next = c.k_hi = L.spec_k_hi;
            c.n_hx = L.spec_n_hx;
            c.out_unscale = L.spec_unscale;
            c.out_lim_s = L.spec_lim_s;
            if (c.n_tsteps < t_cap) ++c.n_tsteps;
            L.logw |= L.spec_nv << 8;
        }
        const int tk = ++c.ticks;
        // the solve is over: its accepted steps into the ring's head (the host wrote the rest of the head before the solve)
        if (fx && c.done && rslots != nullptr)
            reinterpret_cast<OcRingHead*>(reinterpret_cast<char*>(rslots) - OC_RING_SLOTS_OFF)->n_slots = c.n_accept;
        if (fx && tk - 1 < log_cap) att_log[tk - 1] = L.logw;
        if (fx && mailbox) mailbox[0] = (double)tk + (c.done ? 4294967296.0 : 0.0);
    }
    if (!c.done && (int)threadIdx.x < 43) ctl_set_entry(c, (int)threadIdx.x, tab_valid ? (T)(tab * c.hh) : (T)0);
    __syncthreads();
    CB_STAMP(4);
    if (worker && store_out) {          // (the fused tail lets the list builder work on the LDS copy first)
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c);
        uint32_t* dst = reinterpret_cast<uint32_t*>(ctl_out);
        for (int i = threadIdx.x; i < NW32; i += nth) dst[i] = src[i];
        // the solve is over: attempts enqueued past its end alternate between the two blocks -- both must say so
        if (c.done && ctl_in != ctl_out) {
            uint32_t* dst2 = reinterpret_cast<uint32_t*>(const_cast<HjbCtl<T>*>(ctl_in));
            for (int i = threadIdx.x; i < NW32; i += nth) dst2[i] = src[i];
        }
    }
}

// Row-block solve with the active set: this rank's pair for the all-gather.  The tiles off the work list took no part in the
// attempt -- their error share is 0 and their max |y_new| the far field's u (what k_control adds on one GPU) -- and whether
// ALL of this rank's tiles had an error share (then the rank goes dense: a local matter, the collectives do not depend on
// it) is left in the controller block for k_control.
template <typename T>
__global__ void __launch_bounds__(1024) k_reduce_local(const double* __restrict__ partial, int nblocks, HjbCtl<T>* __restrict__ ctl,
                                                        double* __restrict__ out) {
    const double2* __restrict__ p2 = reinterpret_cast<const double2*>(partial);
    double s = 0.0, m = 0.0, na = 0.0;
    const int nt = blockDim.x;
    for (int i0 = threadIdx.x; i0 < nblocks; i0 += 4 * nt) {
        double2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i0 + u * nt < nblocks) ? p2[i0 + u * nt] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s += v[u].x;
            m = fmax(m, v[u].y);
            na += v[u].x > 0.0 ? 1.0 : 0.0;
        }
    }
    __shared__ double ls[16], lm[16], la[16];
    s = wave_sum(s);
    m = wave_max(m);
    na = wave_sum(na);
    const int w = threadIdx.x >> 6, nw = (nt + 63) >> 6;
    if ((threadIdx.x & 63) == 0) {
        ls[w] = s;
        lm[w] = m;
        la[w] = na;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int q = 0; q < nw; ++q) {
            a += ls[q];
            b = fmax(b, lm[q]);
            c += la[q];
        }
        if (!ctl->done && ctl->dense_mode == 0 && ctl->work_count < nblocks) b = fmax(b, fabs((double)ctl->u_bg * (double)ctl->in_scale));
        ctl->sh_all_active = (c == (double)nblocks) ? 1 : 0;
        out[0] = a;
        out[1] = b;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_control(const HjbCtl<T>* __restrict__ ctl_in, HjbCtl<T>* __restrict__ ctl_out,
                                                 const double* parts, int nparts, double* __restrict__ t_steps, int t_cap,
                                                 int32_t* __restrict__ att_log, int log_cap, volatile double* mailbox,
                                                 int active_set = 0, OcRingSlot<T>* __restrict__ rslots = nullptr,
                                                 OcRingSlice<T>* __restrict__ rslices = nullptr, int ring_cap = 0) {
    __shared__ CtlLds<T> L;
    control_body<T, false, NoHook, 16>(ctl_in, ctl_out, parts, nparts, t_steps, t_cap, att_log, log_cap, mailbox, active_set, rslots,
                                       rslices, ring_cap, L);
}

// SciPy's select_initial_step (common.py:68-134, order 4) without the host: the three RMS norms are reduced on the device
// (k_norm_scaled + k_reduce_final into red[2], red[4], red[6]) and this kernel turns them into h0 (phase 0: red[8], d1 in
// red[9]) and into the first step, finishing the controller block the host uploaded (phase 1: h_abs and the first attempt's
// coefficient tables) -- three host round trips (0.3 ms of a 3 ms solve) gone.
template <typename T>
__global__ void __launch_bounds__(64) k_init_step(HjbCtl<T>* __restrict__ ctl, double* __restrict__ red, int phase,
                                                  double n_rms, double interval) {
    __shared__ HjbCtl<T> c;
    if (phase == 0) {
        if (threadIdx.x == 0) {
            const double d0 = sqrt(red[2]) / sqrt(n_rms), d1 = sqrt(red[4]) / sqrt(n_rms);
            double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
            h0 = fmin(h0, interval);
            red[8] = h0;
            red[9] = d1;
        }
        return;
    }
    constexpr int NW32 = (int)(sizeof(HjbCtl<T>) / 4);
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(ctl);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&c);
        for (int i = threadIdx.x; i < NW32; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double h0 = red[8], d1 = red[9];
        const double d2 = sqrt(red[6]) / sqrt(n_rms) / h0;
        double h1;
        if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
        else h1 = pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
        c.h_abs = fmin(fmin(100 * h0, h1), interval);
        c.h_abs0 = c.h_abs;
        ctl_prepare_step(c, true);
    }
    __syncthreads();
    if ((int)threadIdx.x < 43) ctl_prepare_entry(c, (int)threadIdx.x);
    __syncthreads();
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c);
        uint32_t* dst = reinterpret_cast<uint32_t*>(ctl);
        for (int i = threadIdx.x; i < NW32; i += blockDim.x) dst[i] = src[i];
    }
}

// A solve's small resets in ONE launch (as hipMemsetAsync calls they were four, which the runtime turned into seven fill
// kernels, ~2.5 us of stream time each): up to four word ranges, each set to its own value.
struct WordFills {
    uint32_t* p[4];
    int32_t n[4];
    uint32_t v[4];
};
__global__ void __launch_bounds__(256) k_fill_words(WordFills F) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < F.n[r]; i += gridDim.x * blockDim.x) F.p[r][i] = F.v[r];
}

// ---- select_initial_step (scipy common.py:68-134) of a whole-grid solve in FOUR launches (round 3: twelve -- fill, rhs, two
// norms with a reduction each, h0, axpy, rhs, norm, reduction, first step).  Same arithmetic per cell, same element -> thread
// mapping and the same summation orders as the kernels they replace (k_fill, k_rhs, k_norm_scaled, k_reduce_final, k_axpy_h0,
// k_init_step): h_abs0 and everything that follows are bit-identical.
// (1) y0 = 1 (optimals.py:190: phi(T) = 1), f0 = hjb(T, y0) = K0 of the first step, the partials of d0 = |y0 / scale| and
//     d1 = |f0 / scale| (k_norm_scaled<0>'s grid-stride mapping: the caller launches its grid)
template <typename T>
__global__ void __launch_bounds__(256) k_init_a(const T* __restrict__ coef, T* __restrict__ y, T* __restrict__ k0, T a_lap, T atol,
                                                T rtol, int64_t n, double* __restrict__ partial_y, double* __restrict__ partial_f) {
    double s0 = 0.0, s1 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T one = (T)1, cf = coef[i];
        // (rhs_point on a constant field: the mirror ghosts are the constant too)
        const T f = (cf != cf) ? (T)0 : rhs_value(one, one, one, one, one, a_lap, cf);
        y[i] = one;
        k0[i] = f;
        const T sc = atol + fabs(one) * rtol;
        const T v0 = one / sc, v1 = f / sc;
        s0 += (double)v0 * (double)v0;
        s1 += (double)v1 * (double)v1;
    }
    block_reduce_store(s0, 0.0, partial_y);
    __syncthreads();                    // (block_reduce_store's LDS words are used again)
    block_reduce_store(s1, 0.0, partial_f);
}
// k_reduce_final's sum of `nblocks` (sum, max) pairs, by a block of 1024 threads; the value is thread 0's
__device__ __forceinline__ double reduce_final_sum(const double* __restrict__ partial, int nblocks, double* ls /* [16] LDS */) {
    const double2* __restrict__ p2 = reinterpret_cast<const double2*>(partial);
    double s = 0.0;
    const int nt = blockDim.x;
    for (int i0 = threadIdx.x; i0 < nblocks; i0 += 4 * nt) {
        double2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i0 + u * nt < nblocks) ? p2[i0 + u * nt] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u].x;
    }
    s = wave_sum(s);
    const int w = threadIdx.x >> 6, nw = (nt + 63) >> 6;
    __syncthreads();                    // (ls may still be read from a previous call)
    if ((threadIdx.x & 63) == 0) ls[w] = s;
    __syncthreads();
    double a = 0.0;
    if (threadIdx.x == 0)
        for (int q = 0; q < nw; ++q) a += ls[q];
    return a;
}
// (2) the two reductions and h0 (k_init_step's phase 0)
__global__ void __launch_bounds__(1024) k_init_r0(const double* __restrict__ partial_y, const double* __restrict__ partial_f,
                                                   int nblocks, double* __restrict__ red, double n_rms, double interval) {
    __shared__ double ls[16];
    const double sy = reduce_final_sum(partial_y, nblocks, ls);
    const double sf = reduce_final_sum(partial_f, nblocks, ls);
    if (threadIdx.x == 0) {
        red[2] = sy;
        red[4] = sf;
        const double d0 = sqrt(sy) / sqrt(n_rms), d1 = sqrt(sf) / sqrt(n_rms);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        h0 = fmin(h0, interval);
        red[8] = h0;
        red[9] = d1;
    }
}
// (3) y1 = y0 + h0 * direction * f0 (k_axpy_h0), f1 = hjb(y1) (k_rhs), the partials of d2 = |(f1 - f0) / scale|
//     (k_norm_scaled<1>) -- y1 and f1 are formed in registers and never stored
//     (y0 is the constant 1 that k_init_a wrote: not read back)
template <typename T>
__global__ void __launch_bounds__(256) k_init_b(const T* __restrict__ coef, const T* __restrict__ k0,
                                                const double* __restrict__ red, double direction, int Ny, int Nx, T a_lap, T atol,
                                                T rtol, int64_t n, double* __restrict__ partial) {
    const T cc = (T)(red[8] * direction), one = (T)1;
    auto y1 = [&](int64_t c) -> T { return one + fma_t(cc, k0[c], (T)0); };
    double s = 0.0;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(c / Nx), j = (int)(c - (int64_t)i * Nx);
        const T cf = coef[c];
        T f1 = (T)0;
        if (!(cf != cf)) {              // (rhs_point)
            const int iu = (i == 0) ? 1 : i - 1, id = (i == Ny - 1) ? Ny - 2 : i + 1;
            const int jl = (j == 0) ? 1 : j - 1, jr = (j == Nx - 1) ? Nx - 2 : j + 1;
            f1 = rhs_value(y1((int64_t)iu * Nx + j), y1((int64_t)id * Nx + j), y1((int64_t)i * Nx + jl), y1((int64_t)i * Nx + jr),
                           y1(c), a_lap, cf);
        }
        const T sc = atol + fabs(one) * rtol;
        const T v = (f1 - k0[c]) / sc;
        s += (double)v * (double)v;
    }
    block_reduce_store(s, 0.0, partial);
}

// (4) the third reduction, the first step (k_init_step's phase 1: h_abs and the first attempt's coefficient tables into the
//     controller block the host uploaded) and the solve's small resets (k_fill_words)
//     The controller block and the ring's head arrive as KERNEL ARGUMENTS (as copies from pinned memory they were two blit
//     launches in the solve's chain).
template <typename T>
struct InitBlocks {
    HjbCtl<T> ctl;
    OcRingHead head;
    OcRingHead* head_dst;       // or NULL
};
template <typename T>
__global__ void __launch_bounds__(1024) k_init_r1(HjbCtl<T>* __restrict__ ctl, const InitBlocks<T> B, const double* __restrict__ partial,
                                                   int nblocks, double* __restrict__ red, double n_rms, double interval, WordFills F) {
    __shared__ double ls[16];
    __shared__ HjbCtl<T> c;
    if (threadIdx.x == 0) {
        c = B.ctl;
        if (B.head_dst != nullptr) *B.head_dst = B.head;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        for (int i = threadIdx.x; i < F.n[r]; i += blockDim.x) F.p[r][i] = F.v[r];
    const double s2 = reduce_final_sum(partial, nblocks, ls);
    constexpr int NW32 = (int)(sizeof(HjbCtl<T>) / 4);
    __syncthreads();
    if (threadIdx.x == 0) {
        red[6] = s2;
        const double h0 = red[8], d1 = red[9];
        const double d2 = sqrt(s2) / sqrt(n_rms) / h0;
        double h1;
        if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
        else h1 = pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
        c.h_abs = fmin(fmin(100 * h0, h1), interval);
        c.h_abs0 = c.h_abs;
        ctl_prepare_step(c, true);
    }
    __syncthreads();
    if ((int)threadIdx.x < 43) ctl_prepare_entry(c, (int)threadIdx.x);
    __syncthreads();
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c);
        uint32_t* dst = reinterpret_cast<uint32_t*>(ctl);
        for (int i = threadIdx.x; i < NW32; i += blockDim.x) dst[i] = src[i];
    }
}

// y1 = y0 + h0 * direction * f0 with h0 on the device (red[8]): the k_lincomb of select_initial_step
template <typename T>
__global__ void k_axpy_h0(const T* __restrict__ base, const T* __restrict__ k, const double* __restrict__ red,
                          double direction, T* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T cc = (T)(red[8] * direction);
    out[i] = base[i] + fma_t(cc, k[i], (T)0);
}

// dense output (scipy rk.py:560-575): phi = y_old + h * sum_p (sum_j K_j P_jp) x^(p+1)
template <typename T>
__global__ void k_dense(const T* __restrict__ yold, Lin8<T> W /* c_j = h * sum_p P_jp x^(p+1) */,
                        T* __restrict__ out, double* __restrict__ keep, double unscale, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T acc = (T)0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (j < W.n) acc = fma_t(W.c[j], W.k[j][i], acc);
    T v = yold[i] + acc;
    if (out) out[i] = v;
    if (keep) keep[i] = (double)v * unscale;
}

// vels (optimals.py:168-186) on a phi carried as phi*scale: lim -> lim*scale; everything else is a ratio.
template <typename TI, typename TO>
__global__ void k_vels(const TI* __restrict__ phi, TO* __restrict__ vx, TO* __restrict__ vy, int Ny, int Nx,
                       TI lim_s, TI lim, TI mu, TI inv2dx, TI inv2dy) {
    int j = blockIdx.x * blockDim.x + threadIdx.x + 1;
    int i = blockIdx.y * blockDim.y + threadIdx.y + 1;
    if (i >= Ny - 1 || j >= Nx - 1) return;
    auto clip = [&](TI p) { return (p > lim_s) ? p : ((p < lim_s) ? lim_s : (TI)0); };   // p==lim -> 0 (Q15)
    TI pe = clip(phi[(int64_t)i * Nx + j + 1]);
    TI pw = clip(phi[(int64_t)i * Nx + j - 1]);
    TI pn = clip(phi[(int64_t)(i + 1) * Nx + j]);
    TI ps = clip(phi[(int64_t)(i - 1) * Nx + j]);
    TI pc = clip(phi[(int64_t)i * Nx + j]);
    TI gx = (pe - pw) * inv2dx;
    TI gy = (pn - ps) * inv2dy;
    TI ox, oy;
    direction_of<TI>(gx, gy, mu * pc, lim, ox, oy);
    int64_t o = (int64_t)i * Nx + j;                 // padded (Ny, Nx) slice
    vx[o] = (TO)ox;
    vy[o] = (TO)oy;
}

// ------------------------------------------------------------------------------------------ fused kernels
// Tile = 64 columns x 16 rows per 256-thread block; thread (cx, ry) owns 4 consecutive cells of one row, so every
// global access is one aligned 16-byte (float) / 32-byte (double) vector per lane.  The stage's working value on the
// tile (+ a 1-cell halo: top/bottom rows, left/right columns; corners are not needed by the 5-point stencil) goes
// through LDS with a 72-element row stride (interior starts at element 4 -> b128-aligned, conflict-free b128 rows).
#define HT_X 64
#define HT_Y 16
#define HT_LS 72                 // LDS row stride (elements)
#define HT_LH (HT_Y + 2)

template <typename T>
struct alignas(16) V4 {
    T v[4];
};

template <typename T>
__device__ __forceinline__ V4<T> ld4(const T* p) { return *reinterpret_cast<const V4<T>*>(p); }
template <typename T>
__device__ __forceinline__ void st4(T* p, const V4<T>& x) { *reinterpret_cast<V4<T>*>(p) = x; }

// the same at a 32-bit BYTE offset from a wave-uniform base: `global_load/store ... v_offset, s[base]` -- no 64-bit vector
// address arithmetic per array (k_stageV: ten arrays per thread in an issue-bound launch)
template <typename T>
__device__ __forceinline__ V4<T> ld4b(const T* p, uint32_t bo) {
    return *reinterpret_cast<const V4<T>*>(reinterpret_cast<const char*>(p) + bo);
}
template <typename T>
__device__ __forceinline__ void st4b(T* p, uint32_t bo, const V4<T>& x) {
    *reinterpret_cast<V4<T>*>(reinterpret_cast<char*>(p) + bo) = x;
}
template <typename T>
__device__ __forceinline__ void st4b_nt(T* p, uint32_t bo, const V4<T>& x) {
    typedef T vt __attribute__((ext_vector_type(4)));
    vt v = {x.v[0], x.v[1], x.v[2], x.v[3]};
    __builtin_nontemporal_store(v, reinterpret_cast<vt*>(reinterpret_cast<char*>(p) + bo));
}

template <typename T>
__device__ __forceinline__ void st4_nt(T* p, const V4<T>& x) {      // streaming store: the field is written once
    typedef T vt __attribute__((ext_vector_type(4)));
    vt v = {x.v[0], x.v[1], x.v[2], x.v[3]};
    __builtin_nontemporal_store(v, reinterpret_cast<vt*>(p));
}

// a vector of the interpolant ring (k_stageV).  Planar layout: whole lines per wavefront row, written once and read back by
// a few lookups only -> streaming stores (OC_RING_NT 0: plain).  Interleaved layout: plain stores, which the L2 merges (as
// streaming stores its 16-byte pieces at an 80-byte stride went out as partial writes: 147 us per launch at 2048 x 2048).
#ifndef OC_RING_NT
#define OC_RING_NT OC_RING_PLANAR
#endif
// value a of the thread's vector (32-bit byte offset bo into a plane) of a ring slot of n cells
template <typename T>
__device__ __forceinline__ void st4_ring(T* slot, uint32_t bo, int a, size_t n, const V4<T>& x) {
#if OC_RING_PLANAR
    T* const plane = slot + (size_t)a * n;              // (wave-uniform)
#if OC_RING_NT
    st4b_nt(plane, bo, x);
#else
    st4b(plane, bo, x);
#endif
#else
    st4(slot + oc_ring_off((size_t)(bo / sizeof(T)), a, n), x);
#endif
}

template <typename T>
struct StageArgs {
    const T* y;
    const T* k[6];
    T c[6];
    const T* coef;
    T* out;         // K_s (or K6)
    T* ynew;        // FINAL only
    T he[6];        // FINAL: h*E_j for the arrays in k[] (same order) ...
    T he6;          // ... and h*E_6 for the new K6
    T atol, rtol, a_lap;
    double* partial;
    int Ny, Nx;
    int row_lo, row_hi;   // rows whose error / max|y| this launch accounts for (row-block solves: the owned rows)
};

// halo role of a thread: wave 0 lanes 0-15 top row, wave 1 lanes 0-15 bottom row (4 cells each),
// wave 2 lanes 0-15 left column, wave 3 lanes 0-15 right column (1 cell each)
struct HaloRole {
    int kind;      // 0 none, 1 top, 2 bottom, 3 left, 4 right
    int gi, gj;    // global cell (first of 4 for top/bottom), clamped into the grid
    int li, lj;    // LDS row / element
};

__device__ __forceinline__ HaloRole halo_role(int i0, int j0, int Ny, int Nx) {
    HaloRole r;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    r.kind = 0;
    r.gi = r.gj = r.li = r.lj = 0;
    if (l >= 16) return r;
    if (w == 0) { r.kind = 1; r.gi = i0 - 1; r.gj = j0 + 4 * l; r.li = 0; r.lj = 4 + 4 * l; }
    else if (w == 1) { r.kind = 2; r.gi = i0 + HT_Y; r.gj = j0 + 4 * l; r.li = HT_LH - 1; r.lj = 4 + 4 * l; }
    else if (w == 2) { r.kind = 3; r.gi = i0 + l; r.gj = j0 - 1; r.li = l + 1; r.lj = 3; }
    else { r.kind = 4; r.gi = i0 + l; r.gj = j0 + HT_X; r.li = l + 1; r.lj = 4 + HT_X; }
    // cells outside the grid are never used (the stencil mirrors inside the tile, vels skips the border)
    bool ok = r.gi >= 0 && r.gi < Ny && r.gj >= 0 && r.gj < Nx;
    if (!ok) r.kind = 0;
    return r;
}

// XCD-aware tile order (speed only): workgroups are dealt round-robin over the 8 XCDs, so workgroup b works on
// logical tile (b % 8) * per + b / 8: every XCD sweeps a contiguous band of tiles in row-major order and the halo
// lines of neighbouring tiles are served by that XCD's own L2 instead of the fabric.
__device__ __forceinline__ bool tile_of_block(int Ny, int Nx, int& i0, int& j0, int& logical) {
    const int nbx = (Nx + HT_X - 1) / HT_X, nby = (Ny + HT_Y - 1) / HT_Y, nb = nbx * nby;
    const int per = (nb + 7) / 8;
    logical = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
    if ((int)(blockIdx.x / 8) >= per || logical >= nb) return false;
    i0 = (logical / nbx) * HT_Y;
    j0 = (logical % nbx) * HT_X;
    return true;
}

template <typename T, int S, bool FINAL>
__global__ void __launch_bounds__(256) k_stage(StageArgs<T> A) {
    __shared__ alignas(16) T tile[HT_LH][HT_LS];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int Ny = A.Ny, Nx = A.Nx;
    int i0, j0, logical;
    if (!tile_of_block(Ny, Nx, i0, j0, logical)) return;
    const int gi = i0 + ry, gj = j0 + 4 * cx;
    const bool valid = gi < Ny && gj < Nx;             // Nx % 4 == 0: a vector is inside or outside as a whole
    const int64_t idx = (int64_t)gi * Nx + gj;
    V4<T> kreg[S];
    V4<T> yv, tv;
    if (valid) {
        yv = ld4(A.y + idx);
#pragma unroll
        for (int q = 0; q < 4; ++q) tv.v[q] = (T)0;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            kreg[j] = ld4(A.k[j] + idx);
#pragma unroll
            for (int q = 0; q < 4; ++q) tv.v[q] = fma_t(A.c[j], kreg[j].v[q], tv.v[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) tv.v[q] = yv.v[q] + tv.v[q];
        st4(&tile[ry + 1][4 + 4 * cx], tv);
        if (FINAL) st4(A.ynew + idx, tv);
    }
    const HaloRole hr = halo_role(i0, j0, Ny, Nx);
    if (hr.kind == 1 || hr.kind == 2) {
        const int64_t hidx = (int64_t)hr.gi * Nx + hr.gj;
        V4<T> hv = ld4(A.y + hidx), acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc.v[q] = (T)0;
#pragma unroll
        for (int j = 0; j < S; ++j) {
            V4<T> kv = ld4(A.k[j] + hidx);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc.v[q] = fma_t(A.c[j], kv.v[q], acc.v[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) hv.v[q] = hv.v[q] + acc.v[q];
        st4(&tile[hr.li][hr.lj], hv);
    } else if (hr.kind >= 3) {
        const int64_t hidx = (int64_t)hr.gi * Nx + hr.gj;
        T acc = (T)0;
#pragma unroll
        for (int j = 0; j < S; ++j) acc = fma_t(A.c[j], A.k[j][hidx], acc);
        tile[hr.li][hr.lj] = A.y[hidx] + acc;
    }
    __syncthreads();
    double esum = 0.0, ymax = 0.0;
    if (valid) {
        // mirror ghosts of the room border (optimals.py:149-152) are taken from inside the tile
        const int lu = (gi == 0) ? ry + 2 : ry, ld = (gi == Ny - 1) ? ry : ry + 2;
        const V4<T> up = ld4(&tile[lu][4 + 4 * cx]), dn = ld4(&tile[ld][4 + 4 * cx]);
        const T lft = (gj == 0) ? tv.v[1] : tile[ry + 1][3 + 4 * cx];
        const T rgt = (gj + 3 == Nx - 1) ? tv.v[2] : tile[ry + 1][8 + 4 * cx];
        const V4<T> cf = ld4(A.coef + idx);
        V4<T> f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const T C = tv.v[q];
            const T l = (q == 0) ? lft : tv.v[q - 1], r = (q == 3) ? rgt : tv.v[q + 1];
            f.v[q] = (cf.v[q] != cf.v[q]) ? (T)0 : rhs_value(up.v[q], dn.v[q], l, r, C, A.a_lap, cf.v[q]);
        }
        st4(A.out + idx, f);
        if (FINAL && gi >= A.row_lo && gi < A.row_hi) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                T e = A.he6 * f.v[q];
#pragma unroll
                for (int j = 0; j < S; ++j) e = fma_t(A.he[j], kreg[j].v[q], e);
                const T sc = A.atol + fmax(fabs(yv.v[q]), fabs(tv.v[q])) * A.rtol;
                const T qq = err_quot(e, sc);
                esum += (double)qq * (double)qq;
                ymax = fmax(ymax, (double)fabs(tv.v[q]));
            }
        }
    }
    if (FINAL) {
        __shared__ double ls[4], lm[4];
        esum = wave_sum(esum);
        ymax = wave_max(ymax);
        if ((threadIdx.x & 63) == 0) {
            ls[threadIdx.x >> 6] = esum;
            lm[threadIdx.x >> 6] = ymax;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            A.partial[2 * logical] = ls[0] + ls[1] + ls[2] + ls[3];
            A.partial[2 * logical + 1] = fmax(fmax(lm[0], lm[1]), fmax(lm[2], lm[3]));
        }
    }
}

template <typename T>
struct Stage3Args {         // arguments of the multi-stage kernel (k_stageV)
    const HjbCtl<T>* ctl;   // roles of the buffers (parity), coefficients, atol: written by k_control
    T* Y[2];                // y / y_new
    T* KX[2];               // K0 / K6 (FSAL: they swap on every accepted step)
    T* KM[5];               // K1 .. K5
    const T* coef;
    T rtol, a_lap;
    T p[6][3];              // rows 0, 2, 3, 4, 5, 6 / columns 1 .. 3 of P: the FINAL launch leaves Q1 .. Q3 for k_out
    double* partial;
    int Ny, Nx;
    int row_lo, row_hi;     // rows whose error / max|y| this launch accounts for
    // exact active set (k_stageV below): per tile 1 = trivial at that attempt, 0 = computed; NULL = every tile is computed
    const int32_t* flags_prev;      // the previous attempt's flags (NULL at the first attempt)
    int32_t* flags_cur;
    const int32_t* work;            // the attempt's work list (ctl->work_count tiles, build_work_list), or NULL: every tile
    // interpolant ring (RING kernels): y, K0, Q1 .. Q3 of the step go to slot ctl->n_accept (five planes, oc_internal.h)
    // instead of Q1 .. Q3 to KM -- an attempt that is rejected is overwritten by the next one
    T* ring;
    size_t ring_slot_elems;         // 5 * rows * Nx
    int32_t* ring_first;            // [ntile]: the first slot in which the tile was computed (OC_RING_NEVER: not yet)
    int32_t ring_cap, ring_ntile;
    // fused tail (FUSED kernels): the LAST block of the launch to finish is the attempt's step controller and list builder
    // -- one launch per attempt instead of three.  Every participating block stores its words for the controller `sc1`,
    // drains them and takes a ticket (agent-scope add); the block whose ticket is the last runs control_body and
    // build_work_list on what the others wrote (sc1 loads) and leaves the NEXT attempt's controller block in ctl_out.
    HjbCtl<T>* ctl_out;
    int32_t* tail_counter;          // zero between launches (the tail resets it)
    double* t_steps;
    int32_t* att_log;
    int32_t t_cap, log_cap;
    volatile double* mailbox;
    int32_t active_set;
    OcRingSlot<T>* rslots;
    OcRingSlice<T>* rslices;
    int32_t* flags_next;            // the next attempt's flags / work list (the builder's outputs)
    int32_t* work_next;
    // ring solves with the active set: undef[b * ntile + t] = 1 -- tile t's cells in state buffer b (Y[b], KX[b]) are NOT
    // maintained: the tile was trivial when b was last written, its y is the controller's far-field value and its K0 is 0,
    // and whoever loads such cells substitutes exactly that (k_stageV).  NULL: every buffer holds every tile (the other paths).
    int32_t* undef;
    // persistent launch (PERSIST kernels): ONE launch runs attempts p_a0 .. p_amax - 1 of a sparse ring solve on a co-resident
    // grid -- the blocks deal the attempt's work list among themselves, meet at a ticket barrier whose last arrival is the
    // attempt's controller and list builder (the fused tail), and go on to the next attempt without leaving the chip.  The
    // per-attempt pointers (controller block pair, flag / list / mark arrays) rotate inside the kernel exactly as the host
    // rotates them between launches (hjb_solve_dev: enqueue_attempt).
    HjbCtl<T>* p_ctl[2];
    int32_t* p_flags;               // 6 * nb words: flags of two attempts, two work lists, the two buffers' undefined marks
    int32_t* p_sync;                // [0]: attempts whose tail has run (absolute index + 1), [1]: abort (a wait timed out)
    int32_t p_a0, p_amax;
    int32_t p_max_items;            // a list longer than this is the launch chain's business (every CU busy anyway)
    int32_t p_wait_ticks;           // s_memrealtime ticks (100 MHz) a block waits at the barrier before it gives the launch up
    // OC_DIAG_BUILD (liboc_hip_diag.so, OC_PERSIST_STAMPS=<file>): 16 words per attempt -- the LAST block's s_memrealtime
    // stamps [0..5] = attempt entered, controller block read, items done, ticket drawn, tail done, generation stored,
    // [6] = its blockIdx, [7] = the attempt's items; [8..15] the same of block 0 (tools/persist_stamps.py)
    unsigned long long* p_stamps;
    int32_t att_expected;           // one-attempt launches: the attempt number (controller ticks) this launch was enqueued for, or -1
    // fused tail, short lists: 4 doubles per list position (error share, max |y_new|, tile | flag << 32, spare) written by the
    // block that takes the position (TAIL_REC_CAP of them), or NULL
    double* tail_rec;
    int32_t p_prev_lead;            // LEAD: the launch of the attempt before this one was a LEAD launch too (else that attempt ended with a tail or k_control)
};
#ifdef OC_DIAG_BUILD
#define P_STAMP(k) do { if ((PERSIST || LEAD) && A.p_stamps != nullptr && threadIdx.x == 0) p_st[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
__shared__ unsigned long long g_tl_stamp[4];        // inside do_tile: loads landed and tested | stages done | stores issued | tile's record out
#define T_STAMP(k) do { if (threadIdx.x == 0) g_tl_stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define P_STAMP(k) do { } while (0)
#define T_STAMP(k) do { } while (0)
#endif

// value of the previous / next lane of the wavefront (DPP wave shifts: no LDS traffic; lane 0 / 63 get 0)
__device__ __forceinline__ float lane_prev(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));      // wave_shr:1
}
__device__ __forceinline__ float lane_next(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));      // wave_shl:1
}
__device__ __forceinline__ double lane_prev(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double lane_next(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// a finite non-zero power of two (sums 2u, 3u, 4u of such a value are exact: the active-set argument of k_stageV)
__device__ __forceinline__ bool is_pow2(float v) {
    const uint32_t b = __float_as_uint(v), e = (b >> 23) & 0xFFu;
    return (b & 0x007FFFFFu) == 0u && e != 0u && e != 0xFFu;
}
__device__ __forceinline__ bool is_pow2(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v), e = (b >> 52) & 0x7FFull;
    return (b & 0x000FFFFFFFFFFFFFull) == 0ull && e != 0ull && e != 0x7FFull;
}

// what the list builder can do before the controller has decided (list_prepare): the thread's tiles, which of them go on
// the list if the buffers hold the trivial values already, and the wavefront's prefix sum of the counts
struct ListPre {
    uint32_t listed, listed_hi;
    int mine, incl, t_lo, t_hi;
    bool valid;
};
template <typename T, bool SC1>
__device__ __forceinline__ void build_work_list(HjbCtl<T>* __restrict__ ctl, const int32_t* __restrict__ flags_cur,
                                                int32_t* __restrict__ flags_next, int32_t* __restrict__ work,
                                                double* __restrict__ parts_rw, int tnbx, int tnby, uint8_t* sflag, int nbytes,
                                                volatile double* mailbox, bool flags_ready, const ListPre* pre = nullptr,
                                                int32_t* wsum_ext = nullptr, int32_t* undef = nullptr);
__device__ __forceinline__ ListPre list_prepare(int tnbx, int tnby, const uint8_t* sflag, int32_t* wsum);

// KEEP_SGPR: ONE empty asm statement whose operands must sit in scalar registers -- the loads that produce them are all
// issued, and waited for once, before it (k_stageV's prologue)
__device__ __forceinline__ int32_t sgpr_bits(int32_t v) { return v; }
__device__ __forceinline__ int32_t sgpr_bits(float v) { return __builtin_bit_cast(int32_t, v); }
__device__ __forceinline__ int64_t sgpr_bits(double v) { return __builtin_bit_cast(int64_t, v); }
template <typename P>
__device__ __forceinline__ int64_t sgpr_bits(P* v) { return (int64_t)reinterpret_cast<uintptr_t>(v); }
#define SB(x) "s"(sgpr_bits(x))
// a wave-uniform word of memory that THIS launch does not write (the scalar cache is invalidated between launches): a
// scalar load -- through a plain pointer the compiler takes a vector load and a readfirstlane, with a wait in between
template <typename T>
__device__ __forceinline__ T uniform_load(const T* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(p));
}

// fused tail: tickets on one counter up to TAIL_FLAT_MAX participating blocks, in two levels beyond (k_stageV: fused_tail)
#define TAIL_FLAT_MAX 640
#define TAIL_GROUPS 64
#define TAIL_COUNTER_BYTES (128 * (1 + TAIL_GROUPS))

// ---- lean multi-stage kernel: four cells of a row per thread, no per-cell index arithmetic ------------------------
// (Round 1's multi-stage kernels dealt the region's cells to the threads round-robin; that cost ~100 instructions of
// index and predicate arithmetic per cell and stage -- 450 SIMD cycles per wavefront-cell-stage, the launch time scaled
// with the cell count, not with the bytes.)  Here a thread owns FOUR consecutive cells of one region row for the whole
// launch: its inputs are 16-byte loads, the working value of a stage goes to LDS as one ds_write_b128, the stencil
// needs two ds_read_b128 (the rows above and below) at fixed offsets from the thread's own slot, and the cells left
// and right of the vector come from the neighbouring LANES (DPP wave shifts; as LDS scalars their 16-byte stride made
// them 4-way bank-conflicted: 55 % of the LDS cycles of the first version).  Nothing is predicated inside the stages: every thread
// computes every stage, values within NQ cells of the region's edge are garbage by then and are simply not stored
// (halo NQ rows, and 4 ceil(NQ/4) columns so that vectors stay aligned).  The mirror ghosts of the room border
// (optimals.py:149-152): rows -1 / Ny are extra LDS stores by the threads of rows 1 / Ny-2, columns -1 / Nx are the
// vector's own cells 1 / 2.
//   NQ = 3: two launches per attempt (A: K1..K3 from y, K0; B: Q1..Q3, y_new, K6 + error), region 128 x 32 -> tile 120 x 26
//   NQ = 6: one launch (writes Q1..Q3, K6, y_new: 3 arrays read, 5 written), region 128 x 32 -> tile 112 x 20
// Same arithmetic per cell as k_stage (one stage per launch, the cross-check variant): bit-identical results.
// a wave-uniform value that reached a vector register (an LDS word, an sc1 load of one address) back in scalar registers
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double uni(double v) {
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)b), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// one word that another CU wrote a moment ago, inside this very launch (persistent launch: never through the scalar cache)
__device__ __forceinline__ int32_t ld_fresh(const int32_t* p) {
    return uni(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ float ld_fresh(const float* p) {
    return uni(__int_as_float(__hip_atomic_load(reinterpret_cast<const int32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
}
__device__ __forceinline__ double ld_fresh(const double* p) {
    return uni(__longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
}

// PERSIST (FUSED ring kernels): the launch is a co-resident grid that runs attempt after attempt of a sparse solve -- see
// Stage3Args::p_ctl.  What changes against the one-attempt launch: the per-attempt pointers rotate in here; everything the
// previous attempt's blocks wrote is read behind an agent-scope acquire and never through the scalar cache or a load the
// compiler may take for invariant (the controller block comes through LDS, single words through ld_fresh); a block takes the
// items blockIdx.x, blockIdx.x + gridDim.x, ... of the attempt's list; the ticket barrier is taken by EVERY block, once per
// attempt, behind an agent-scope release of the block's stores; the blocks that are not last wait for the generation word --
// with a deadline: a grid that is not co-resident after all (another process's kernels on the chip) gives the launch up
// instead of hanging, and the host goes on with the launch chain from the controller's last state.
// (The persistent variant gets twice the registers -- ONE block per CU: with the attempt and item loops around it the tile
// code spilled 155 vector registers at 64; a sparse attempt lists fewer tiles than the chip has CUs anyway.)
// LEAD (FUSED ring kernels, oc_set_tuning "hjb_tail" = 3): the attempt's controller runs at the START of the NEXT attempt's
// launch, by every block for itself, instead of in the last block of its own launch.  While the work list is short (what the
// fast tail covers: lead_ok) an attempt's blocks only leave their records; the next launch's blocks each read the previous
// controller block and those records (one batch of loads), form the decision, the next list and their own coefficients
// redundantly -- same inputs, same arithmetic, same bits in every block -- and go on to their tile; the block at list
// position 0 (the leader) alone writes the new controller block, the ring's records, the logs and the mailbox.  Gone from an
// attempt's chain: the ticket round trip, the last block's fetch of the records behind it, the controller block's trip to
// memory and back.  Attempts that do not qualify (the solve's first builds, lists beyond TAIL_FAST_MAX, dense mode) keep the
// fused tail; which of the two an attempt used is a function of ITS controller block, so the next launch knows where to look.
template <typename T>
__device__ __forceinline__ bool lead_ok(const HjbCtl<T>& c, const Stage3Args<T>& A, int ntiles) {
    // (work_count < ntiles: a list that is not every tile cannot come out "all active" -- the decision's order for the list
    // builder is then "build" or, at the end of the solve, nothing: what the records alone can serve)
    return A.active_set != 0 && A.undef != nullptr && A.tail_rec != nullptr && c.done == 0 && c.dense_mode == 0 &&
           c.n_attempt >= 2 && c.work_count <= TAIL_FAST_MAX && c.work_count < ntiles;
}

template <typename T, int NQ, int S0, bool FINAL, int RWV, int RH, bool RING = false, bool FUSED = false, bool PERSIST = false,
          bool LEAD = false>
__global__ void __launch_bounds__(RWV * RH, (sizeof(T) == 4 ? 8 : 4) / (PERSIST ? 2 : 1)) k_stageV(Stage3Args<T> A) {      // float: 64 VGPRs, two blocks per CU
    static_assert(!RING || (NQ == 6 && FINAL), "the ring is written by the one-launch attempt");
    static_assert(!FUSED || RING, "the fused tail belongs to the ring path");
    static_assert(!PERSIST || FUSED, "the persistent launch is the fused-tail launch in a loop");
    static_assert(!LEAD || (FUSED && !PERSIST), "the leading controller belongs to the one-attempt fused launch");
    constexpr bool ROT = PERSIST || LEAD;           // the per-attempt pointers rotate in the kernel; the controller block is used out of LDS
    __shared__ HjbCtl<T> s_prev;                    // LEAD: the previous attempt's controller block
    __shared__ int s_last;
    __shared__ CtlLds<T> s_tail;        // FUSED: the controller's LDS block (unreferenced, hence absent, otherwise)
    __shared__ int32_t s_wsum[16];      // FUSED: the list builder's wavefront sums
    constexpr int HX = 4 * ((NQ + 3) / 4), HY = NQ, RW = 4 * RWV, NT = RWV * RH;
    constexpr int TX = RW - 2 * HX, TY = RH - 2 * HY, LS = RW + 8;
    constexpr int NK = S0 + NQ;                     // stage derivatives alive at the end of the launch
    static_assert(NQ == 3 || (NQ == 6 && S0 == 1 && FINAL), "three stages per launch, or the whole attempt");
    __shared__ alignas(16) T buf[2][RH + 2][LS];    // slot of region cell (r, c): [r + 1][c + 4]; one spare ring
    constexpr int NW32 = (int)(sizeof(HjbCtl<T>) / 4), CW = (NW32 + 255) / 256;
    __shared__ T s_p[6][3];             // PERSIST: the dense output's P columns (kernel arguments: 18 scalar registers held
    if (ROT && threadIdx.x < 18)        // across the attempt and item loops otherwise; the first barrier below publishes them)
        s_p[threadIdx.x / 3][threadIdx.x % 3] = A.p[threadIdx.x / 3][threadIdx.x % 3];
#ifdef OC_DIAG_BUILD
    unsigned long long p_st[6] = {0, 0, 0, 0, 0, 0};
#endif
  for (int att = ROT ? A.p_a0 : 0;; ++att) {        // (PERSIST: the attempts of this launch; otherwise ONE pass -- LEAD: attempt p_a0)
    if (PERSIST && att >= A.p_amax) return;
    P_STAMP(0);
    // the attempt's controller block, flags, work list (PERSIST: rotated here as hjb_solve_dev rotates them between launches)
    const HjbCtl<T>* __restrict__ ctl = ROT ? A.p_ctl[att & 1] : A.ctl;
    HjbCtl<T>* const ctl_out = ROT ? A.p_ctl[(att + 1) & 1] : A.ctl_out;
    const int nb_all = ((A.Nx + TX - 1) / TX) * ((A.Ny + TY - 1) / TY);
    const bool p_act = ROT && A.active_set != 0;
    const int32_t* const flags_prev = ROT ? (p_act && att > 0 ? A.p_flags + (size_t)((att - 1) & 1) * nb_all : nullptr) : A.flags_prev;
    int32_t* const flags_cur = ROT ? (p_act ? A.p_flags + (size_t)(att & 1) * nb_all : nullptr) : A.flags_cur;
    const int32_t* const work = ROT ? (p_act && att > 0 ? A.p_flags + (size_t)(2 + (att & 1)) * nb_all : nullptr) : A.work;
    int32_t* const flags_next = ROT ? (p_act ? A.p_flags + (size_t)((att + 1) & 1) * nb_all : nullptr) : A.flags_next;
    int32_t* const work_next = ROT ? (p_act ? A.p_flags + (size_t)(2 + ((att + 1) & 1)) * nb_all : nullptr) : A.work_next;
    // the records of an attempt's listed tiles (fused tail's short-list path; LEAD: read by the NEXT launch, hence by parity)
    double* const rec_mine = A.tail_rec != nullptr ? A.tail_rec + (ROT ? (size_t)(att & 1) * 4 * TAIL_REC_CAP : 0) : nullptr;
    // Everything the launch reads of the controller block (264 contiguous bytes, fresh in every attempt: the first wavefront
    // of a CU misses the scalar cache on each of its lines) is loaded in ONE batch up front, together with this block's entry
    // of the work list: left to itself the compiler issues these loads where the values are first used, behind the early
    // exits -- a chain of five dependent round trips to the L2 (3 - 5 us) before a block of a sparse attempt, alone on its CU,
    // issues its first vector load.  (keep_sgpr: an empty asm statement that needs the values in scalar registers.)
    const int Ny = A.Ny, Nx = A.Nx;
    const int nbx = (Nx + TX - 1) / TX, nby = (Ny + TY - 1) / TY, nb = nbx * nby;
    const int per = (nb + 7) / 8;
    // this block's position among the launch's blocks in the XCD-aware order (tile_of_block; gridDim.x is a multiple of 8): the
    // tile of that number, or the entry of that number of the work list -- and, where the launch has fewer blocks than the
    // attempt has items (the host sizes a sparse attempt's grid by the list length it last saw), every gridDim.x-th one after it
    const int per_g = (int)gridDim.x / 8;
    const int logical = PERSIST ? 0 : (int)(blockIdx.x % 8) * per_g + (int)(blockIdx.x / 8);
    const bool listed = NQ == 6 && flags_cur != nullptr && work != nullptr;
    int c_done, par, c_dense, c_wcount, c_nacc, c_rej, w_entry, c_ticks = 0;
    T c_oiscale, c_ubg, atol_s, in_scale;
    T cc[NQ][6], he[7];
    bool lead_has_tail = true;                      // LEAD: does THIS attempt end with the fused tail (else its records wait for the next launch)
    if constexpr (LEAD) {
        // ---- both controller blocks, the previous attempt's records and (for the case that it had a tail) this block's
        // entry of the work list: ONE batch of loads
        const int L = att;
        const HjbCtl<T>* const prevp = A.p_ctl[(L + 1) & 1];        // S(L-1): what attempt L-1 ran with
        uint32_t wa[CW], wb[CW];
        if (threadIdx.x < 256) {
            const uint32_t* sa = reinterpret_cast<const uint32_t*>(prevp);
            const uint32_t* sb = reinterpret_cast<const uint32_t*>(ctl);
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                const bool inr = (int)threadIdx.x + 256 * q < NW32;
                wa[q] = inr ? sa[threadIdx.x + 256 * q] : 0u;
                wb[q] = inr ? sb[threadIdx.x + 256 * q] : 0u;
            }
        }
        double rs = 0.0, rm = 0.0;
        long long rtb = 0, rtag = -1;
        if (A.tail_rec != nullptr && (int)threadIdx.x < TAIL_FAST_MAX) {
            const double* r = A.tail_rec + (size_t)((L + 1) & 1) * 4 * TAIL_REC_CAP + 4 * threadIdx.x;
            // (tile | flag and the tag first, the sums behind them: the writer stores in the opposite order, all four words lie in
            // one 32-byte sector -- a record whose tag is the attempt's is whole)
            rtb = __hip_atomic_load(reinterpret_cast<const long long*>(r + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            rtag = __hip_atomic_load(reinterpret_cast<const long long*>(r + 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            rs = ld_part<true>(r);
            rm = ld_part<true>(r + 1);
        }
        const int w_spec = listed ? work[min(logical, nb - 1)] : 0;
        if (threadIdx.x < 256) {
            uint32_t* da = reinterpret_cast<uint32_t*>(&s_prev);
            uint32_t* db = reinterpret_cast<uint32_t*>(&s_tail.c);
#pragma unroll
            for (int q = 0; q < CW; ++q)
                if ((int)threadIdx.x + 256 * q < NW32) {        // (S(L-1) in both: the rule is that the decision is formed here)
                    da[threadIdx.x + 256 * q] = wa[q];
                    db[threadIdx.x + 256 * q] = wa[q];
                }
        }
        __syncthreads();
        P_STAMP(1);
        const bool had_tail = L == 0 || A.p_prev_lead == 0 || !lead_ok(s_prev, A, nb);       // (a function of S(L-1) alone: every block agrees)
        auto not_mine = [&]() {                     // a controller block / records that are not this launch's: see att_expected
            if (threadIdx.x == 0) {
                atomicAdd(A.tail_counter + TAIL_COUNTER_BYTES / 4 + 5, 1);
                A.mailbox[2] = (double)(L + 1);
            }
        };
        if (had_tail) {
            if (L > 0 && uni(s_prev.done) != 0) {
                // the solve is over, found by the previous launch's blocks (its leader wrote the final block where this launch
                // finds its "previous" one): the leader copies it into the other block -- the launches enqueued past the end
                // alternate between the two, and the host reads the one of the attempt it enqueued last
                if (logical == 0 && threadIdx.x < 256) {
                    uint32_t* dst = reinterpret_cast<uint32_t*>(const_cast<HjbCtl<T>*>(ctl));
#pragma unroll
                    for (int q = 0; q < CW; ++q)
                        if ((int)threadIdx.x + 256 * q < NW32) dst[threadIdx.x + 256 * q] = wa[q];
                }
                return;
            }
            __syncthreads();
            if (threadIdx.x < 256) {                // the attempt's own block, left by the previous attempt's tail
                uint32_t* db = reinterpret_cast<uint32_t*>(&s_tail.c);
#pragma unroll
                for (int q = 0; q < CW; ++q)
                    if ((int)threadIdx.x + 256 * q < NW32) db[threadIdx.x + 256 * q] = wb[q];
            }
            __syncthreads();
            if (uni(s_tail.c.ticks) != L) {
                not_mine();
                return;
            }
            w_entry = w_spec;
        } else {
            if (uni(s_prev.ticks) != L - 1) {
                not_mine();
                return;
            }
            const int nrec = min(uni(s_prev.work_count), nb);
            // every record must be attempt L-1's (a launch that overtook its predecessor would find older ones)
            if (__syncthreads_or(((int)threadIdx.x < nrec && rtag != (long long)(L - 1)) ? 1 : 0)) {
                not_mine();
                return;
            }
            const bool leader = logical == 0;
            uint32_t* const lbits = reinterpret_cast<uint32_t*>(&buf[1][0][0]);
            int32_t* const llist = reinterpret_cast<int32_t*>(&buf[0][0][0]);       // (the tile buffers are free until the tile starts)
            static_assert(sizeof(buf) / 2 >= 4 * 9 * TAIL_FAST_MAX, "the next list (at most nine tiles per computed tile) fits a tile buffer");
            auto beside = [&]() {
                if (((int)threadIdx.x >> 6) == 2)
                    fast_list(s_tail.rec_tile, s_tail.rec_flag, nrec, nbx, nby, lbits, llist, &s_tail.fast_total, (int)threadIdx.x & 63);
            };
            control_body<T, true>(prevp, const_cast<HjbCtl<T>*>(ctl), A.partial, nb, A.t_steps, A.t_cap, A.att_log, A.log_cap, nullptr, A.active_set,
                                  A.rslots, A.rslices, A.ring_cap, s_tail, false, true, beside, nullptr, nrec, leader, true, rs, rm, rtb);
            // (S(L) is in s_tail.c now; the list of attempt L in llist, fast_total long)
            P_STAMP(2);
            if (threadIdx.x == 0 && s_tail.c.list_todo == 1) s_tail.c.work_count = s_tail.fast_total;
            __syncthreads();
            if (leader) {
                // the leader alone leaves S(L) in memory (the next launch's "previous" block), the list (for a launch that
                // falls back to the tail's slow builder) and the host's mailbox
                if (threadIdx.x < 256) {
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(&s_tail.c);
                    uint32_t* dst = reinterpret_cast<uint32_t*>(const_cast<HjbCtl<T>*>(ctl));
                    for (int i = threadIdx.x; i < NW32; i += 256) dst[i] = src[i];
                }
                if (s_tail.c.list_todo == 1 && work != nullptr) {
                    int32_t* wg = const_cast<int32_t*>(work);
                    for (int i = threadIdx.x; i < s_tail.fast_total; i += NT) wg[i] = llist[i];
                }
                if (threadIdx.x == 0) {
                    const HjbCtl<T>& c = s_tail.c;
                    if ((c.ticks & 3) == 0 || c.done || c.dense_mode || c.work_count > TAIL_FLAT_MAX) {
                        A.mailbox[1] = c.dense_mode ? (double)nb + 0.5 : (double)c.work_count;
                        A.mailbox[0] = (double)c.ticks + (c.done ? 4294967296.0 : 0.0);
                    }
                }
            }
            if (s_tail.c.list_todo != 1) {
                // the solve is over (or the rule says "every tile from now on", which a short list cannot bring about): nothing
                // to compute here.  (done: the next launch's leader copies the block into the other one as well.)
                if (s_tail.c.done == 0 && threadIdx.x == 0 && leader) A.mailbox[2] = (double)(L + 1);
                return;
            }
            w_entry = logical < s_tail.fast_total ? llist[min(logical, (int)(sizeof(buf) / 8) - 1)] : 0;
            w_entry = uni(w_entry);
            __syncthreads();                        // (llist / lbits are the tile buffers)
        }
        const HjbCtl<T>& lc = s_tail.c;
        c_done = uni(lc.done); par = uni(lc.parity); c_dense = uni(lc.dense_mode); c_wcount = uni(lc.work_count);
        c_nacc = uni(lc.n_accept); c_rej = uni(lc.rejected); c_ticks = L;
        c_oiscale = uni(lc.out_in_scale); c_ubg = uni(lc.u_bg); atol_s = uni(lc.atol_s); in_scale = uni(lc.in_scale);
        lead_has_tail = !lead_ok(lc, A, nb);
        P_STAMP(3);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 6; ++j) cc[q][j] = (T)0;
#pragma unroll
        for (int j = 0; j < 7; ++j) he[j] = (T)0;
    } else if (PERSIST) {
        // every block copies the attempt's controller block into LDS -- sc1 loads: the previous attempt's tail, on another CU,
        // wrote it a moment ago -- together with its first entry of the work list; the fields then come out of LDS into scalar
        // registers (a plain load here could be taken for invariant across the attempts, a scalar load could hit a stale line)
        uint32_t cwp[CW];
        if (threadIdx.x < 256) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(ctl);
#pragma unroll
            for (int q = 0; q < CW; ++q)
                cwp[q] = ((int)threadIdx.x + 256 * q < NW32)
                             ? __hip_atomic_load(src + threadIdx.x + 256 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
        w_entry = listed ? ld_fresh(work + min((int)blockIdx.x, nb - 1)) : 0;
        if (threadIdx.x < 256) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(&s_tail.c);
#pragma unroll
            for (int q = 0; q < CW; ++q)
                if ((int)threadIdx.x + 256 * q < NW32) dst[threadIdx.x + 256 * q] = cwp[q];
        }
        __syncthreads();
        const HjbCtl<T>& lc = s_tail.c;
        c_done = uni(lc.done); par = uni(lc.parity); c_dense = uni(lc.dense_mode); c_wcount = uni(lc.work_count);
        c_nacc = uni(lc.n_accept); c_rej = uni(lc.rejected);
        c_oiscale = uni(lc.out_in_scale); c_ubg = uni(lc.u_bg); atol_s = uni(lc.atol_s); in_scale = uni(lc.in_scale);
        P_STAMP(1);
        // (the 43 coefficients of the attempt are NOT taken into registers here: with the attempt and item loops around the
        // tile code they -- and the splat pairs the packed instructions want -- were alive all the way and the scalar registers
        // overflowed: 330 spilled, 300 v_readlane in the tile code.  A stage fetches its own from the LDS copy: coef_of)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 6; ++j) cc[q][j] = (T)0;
#pragma unroll
        for (int j = 0; j < 7; ++j) he[j] = (T)0;
    } else {
        c_done = ctl->done; par = ctl->parity; c_dense = ctl->dense_mode; c_wcount = ctl->work_count;
        c_nacc = ctl->n_accept; c_rej = ctl->rejected; c_ticks = ctl->ticks;
        c_oiscale = ctl->out_in_scale; c_ubg = ctl->u_bg;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (NQ == 6) cc[q][j] = q < 3 ? ctl->cA[q][j] : ctl->cB[q - 3][j];
                else cc[q][j] = FINAL ? ctl->cB[q][j] : ctl->cA[q][j];
            }
#pragma unroll
        for (int j = 0; j < 7; ++j) he[j] = FINAL ? ctl->he[j] : (T)0;
        atol_s = ctl->atol_s; in_scale = ctl->in_scale;
        w_entry = listed ? work[min(logical, nb - 1)] : 0;      // (read before it is known to be needed: in bounds)
        asm volatile("" ::SB(c_done), SB(par), SB(c_dense), SB(c_wcount), SB(c_nacc), SB(c_rej), SB(c_ticks), SB(w_entry), SB(c_oiscale), SB(c_ubg),
                     SB(atol_s), SB(in_scale), SB(cc[0][0]), SB(cc[NQ / 2 - 1][5]), SB(cc[NQ / 2][0]), SB(cc[NQ - 1][5]), SB(he[0]),
                     SB(he[6]), SB(A.coef), SB(flags_prev));
    }
    if (c_done) {                                   // an attempt enqueued past the end of the solve (PERSIST: every block sees it)
        if (!PERSIST && A.att_expected == 0 && threadIdx.x == 0) atomicAdd(A.tail_counter + TAIL_COUNTER_BYTES / 4 + 4, 1);     // (diagnostics: never at attempt 0)
        return;
    }
    const T* __restrict__ y = A.Y[par];
    const T* kin[4] = {A.KX[par], A.KM[0], A.KM[1], A.KM[2]};
    if (!PERSIST) asm volatile("" ::SB(y), SB(kin[0]));   // (the two pointers picked by the parity: in registers before the loads
                                                    // below are issued, or a late scalar load would wait for all of those)
    T* ynew = A.Y[par ^ 1];
    const bool sparse = NQ == 6 && flags_cur != nullptr && c_dense == 0;
    // PERSIST: the launch is for sparse attempts -- once every tile is active, or the list outgrows what this grid should
    // carry, the launch chain takes over (the controller block is between two attempts: every block reads the same words)
    const int n_items = (sparse && work != nullptr) ? min(c_wcount, nb) : nb;
    if (PERSIST && ((p_act && c_dense != 0) || n_items > A.p_max_items)) return;
    // blocks that take part in the attempt (the others have returned above or do so just below): what the tail waits for
    // (PERSIST: the blocks that have an item -- the others neither draw a ticket nor poll eagerly: they look in on the
    // generation word now and then and take part again when the list has grown to reach them)
    const int n_part = min(PERSIST && !(sparse && work != nullptr) ? 8 * per : n_items, (int)gridDim.x);
    if constexpr (!PERSIST) {
        // The host sizes a sparse attempt's grid by the list length it last saw and enqueues attempts ahead of the device.  A
        // launch that finds itself too small for the attempt's items does NOTHING -- no tile, no ticket, no controller -- but
        // say so in the mailbox; the launches enqueued behind it find a controller block that is not theirs (ticks !=
        // att_expected) and do nothing either; the host drains the stream and goes on from the controller's state with a
        // grid that fits (hjb_solve_dev).  (Never seen in practice: the list grows by a tile every few attempts.)
        if (!LEAD && A.att_expected >= 0 && c_ticks != A.att_expected) {
            // Not this launch's controller block: either the launches before it did nothing (a grid too small, below), or --
            // seen where TWO PROCESSES share one GPU (the N = 2 rehearsal on a one-GPU box), never otherwise -- this launch's
            // blocks started before the previous launch's last block had written the block (an sc1 load finds the old words
            // too: it is the launch order, not a cache).  Do nothing and ask the host to drain and re-enqueue from the
            // controller's state; an attempt some of whose blocks ran is simply run again (it changes none of its inputs).
            if (threadIdx.x == 0) {
                atomicAdd(A.tail_counter + TAIL_COUNTER_BYTES / 4 + 5, 1);      // (diagnostics)
                A.mailbox[2] = (double)(A.att_expected + 1);
            }
            return;
        }
        if (n_items > (int)gridDim.x) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                A.mailbox[1] = (double)n_items;
                A.mailbox[2] = (double)(c_ticks + 1);
            }
            return;
        }
        if (logical >= n_items) return;             // no item for this block
    }
    // PERSIST: wait for the attempt's tail (generation > att) -- with a deadline; false: the launch is given up
    auto wait_gen = [&](const int nap) __attribute__((always_inline)) -> bool {
        if (threadIdx.x == 0) {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            int ok = 1;
            for (;;) {
                const long long w = __hip_atomic_load(reinterpret_cast<const long long*>(A.p_sync), __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(w >> 32) != 0) {          // somebody's deadline passed
                    ok = 0;
                    break;
                }
                if ((int)(w & 0xFFFFFFFFll) > att) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > (uint64_t)(uint32_t)A.p_wait_ticks) {
                    __hip_atomic_store(A.p_sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
                if (nap) __builtin_amdgcn_s_sleep(32);
                else __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_last = ok ? 0 : -1;
        }
        __syncthreads();
        const bool ok = s_last == 0;
        __syncthreads();                    // (s_last is written again by the next barrier's ticket)
        return ok;
    };
    if constexpr (PERSIST) {
        if ((int)blockIdx.x >= n_part) {    // no item for this block in this attempt
            if (!wait_gen(1)) return;
            continue;
        }
    }
    auto st_part = [&](double* p, double v) {           // a word for the controller (thread 0 only)
        if (FUSED) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *p = v;
    };
    auto st_flag = [&](int32_t* p, int32_t v) {
        if (FUSED) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *p = v;
    };
    // FUSED: every participating block copies the controller block into LDS as it starts (1 KB out of the L2): the block that
    // turns out to be the last one has it at hand instead of behind another round trip.  The words are LOADED here and stored
    // below, behind the tile's own loads (store_ctl_words) -- stored on the spot, the first four wavefronts of every block sat
    // out a round trip to the L2 before they issued anything else.
    uint32_t cw[CW];
    if (FUSED && !ROT && threadIdx.x < 256) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(ctl);
#pragma unroll
        for (int q = 0; q < CW; ++q) cw[q] = ((int)threadIdx.x + 256 * q < NW32) ? src[threadIdx.x + 256 * q] : 0u;
    }
    auto store_ctl_words = [&]() {
        if (FUSED && !ROT && threadIdx.x < 256) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(&s_tail.c);
#pragma unroll
            for (int q = 0; q < CW; ++q)
                if ((int)threadIdx.x + 256 * q < NW32) dst[threadIdx.x + 256 * q] = cw[q];
        }
    };
    auto tail_work = [&]() __attribute__((always_inline)) {                            // the attempt's controller and list builder: every thread of ONE block
        // Everything the tail reads of the other blocks goes out in ONE batch: the tiles' flags for the list builder here,
        // the error partials inside control_body right behind them -- two dependent round trips to the L2 otherwise.
        // (the flags by wavefronts 4 .. 7, idle otherwise, straight into the builder's LDS bytes, while wavefronts 0 .. 3 run
        // the controller: its error partials and the flags are in flight together)
        uint8_t* const sflag = reinterpret_cast<uint8_t*>(&buf[0][0][0]);
        static_assert(NT >= 512, "the tail's helper wavefronts");
        // short lists (the rule of a sparse solve): reduction and list from the listed tiles' records -- CtlLds, fast_list.
        // From the solve's third build on (the first two also write the words of the tiles they leave off the list), with the
        // far-field substitution (tiles may stay off the list whatever the decision).
        const int nrec = (FUSED && sparse && work != nullptr && A.tail_rec != nullptr && n_items <= TAIL_FAST_MAX) ? n_items : 0;
        const bool fast_ok = nrec > 0 && A.active_set != 0 && A.undef != nullptr && s_tail.c.n_attempt + 1 > 2;
        uint32_t* const lbits = reinterpret_cast<uint32_t*>(&buf[1][0][0]);
        static_assert(sizeof(buf) / 2 >= 4 * ((12000 + 31) / 32), "the fast list's bitmap fits one tile buffer");
        const bool pre_ok = !fast_ok && sparse && A.active_set != 0 && nb <= (int)sizeof(buf);     // (dense mode: no flags, no list)
        if (pre_ok && threadIdx.x >= 256 && threadIdx.x < 512) {
            for (int t0 = (int)threadIdx.x - 256; t0 < nb; t0 += 8 * 256) {
                int32_t f[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) f[u] = (t0 + u * 256 < nb) ? ld_flag<true>(flags_cur + t0 + u * 256) : 0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (t0 + u * 256 < nb) sflag[t0 + u * 256] = f[u] != 0 ? 1 : 0;
            }
        }
        // While thread 0 runs the controller's serial float64 part (~2.7 us), the other wavefronts form the list builder's
        // decision-independent half from the flags just fetched: which tiles go on the next list if tiles may stay off it,
        // the counts, the wavefronts' prefix sums (list_prepare; the barrier inside control_body that precedes the serial
        // part has made the flag bytes visible, the one behind it publishes wsum).
        ListPre lpre;
        lpre.valid = false;
        auto beside = [&]() {
            if (fast_ok) {
                if (((int)threadIdx.x >> 6) == 2)       // (wavefront 1 is busy with the accepted step's bookkeeping)
                    fast_list(s_tail.rec_tile, s_tail.rec_flag, nrec, nbx, nby, lbits, work_next, &s_tail.fast_total, (int)threadIdx.x & 63);
            } else if (pre_ok && s_tail.c.dense_mode == 0) lpre = list_prepare(nbx, nby, sflag, s_wsum);
        };
        control_body<T, true>(ctl, ctl_out, A.partial, nb, A.t_steps, A.t_cap, A.att_log, A.log_cap, nullptr, A.active_set,
                              A.rslots, A.rslices, A.ring_cap, s_tail, false, true, beside, rec_mine, nrec);
        // (the tile buffers are free: the builder's flag bytes)
        static_assert(sizeof(buf) >= 12000, "the builder's flag bytes fit the tile buffers");
        if (A.active_set) {
            if (fast_ok && s_tail.c.list_todo == 1) {       // the list is written (fast_list, beside the decision): its length
                if (threadIdx.x == 0) s_tail.c.work_count = s_tail.fast_total;
            } else {
                build_work_list<T, true>(&s_tail.c, flags_cur, flags_next, work_next, A.partial, nbx, nby, sflag,
                                         (int)sizeof(buf), nullptr, pre_ok, &lpre, s_wsum, A.undef);
            }
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            constexpr int NW32 = (int)(sizeof(HjbCtl<T>) / 4);
            const uint32_t* src = reinterpret_cast<const uint32_t*>(&s_tail.c);
            uint32_t* dst = reinterpret_cast<uint32_t*>(ctl_out);
            for (int i = threadIdx.x; i < NW32; i += 256) dst[i] = src[i];
            if (s_tail.c.done) {        // (both blocks: see control_body)
                uint32_t* dst2 = reinterpret_cast<uint32_t*>(const_cast<HjbCtl<T>*>(ctl));
                for (int i = threadIdx.x; i < NW32; i += 256) dst2[i] = src[i];
            }
        }
        // the host's mailbox last (pinned host memory: writes over PCIe -- the launch cannot end, and the persistent launch's
        // next barrier cannot be passed, before they are acknowledged: ~1 us).  The host only paces its enqueuing by it and
        // needs to see the end: every 4th tick, a change of mode, the end.  The persistent launch posts nothing per attempt
        // (the host waits for the launch and reads the controller block).
        if (threadIdx.x == 0 && !PERSIST) {
            const HjbCtl<T>& c = s_tail.c;
            if ((c.ticks & 3) == 0 || c.done || c.dense_mode || c.work_count > TAIL_FLAT_MAX) {
                A.mailbox[1] = A.active_set ? (c.dense_mode ? (double)nb + 0.5 : (double)c.work_count) : (double)nb;
                A.mailbox[0] = (double)c.ticks + (c.done ? 4294967296.0 : 0.0);
            }
        }
    };
    const int part_pos = logical;                       // this block's number among the participants (0 .. n_part - 1)
    auto fused_tail = [&]() __attribute__((always_inline)) {                           // called by every thread of a participating block, at its end
        if (threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this block's sc1 words have left
            // One counter takes ~90 tickets per us: fine for the active set's short lists, not for a launch of thousands of
            // blocks.  Those draw in two levels: TAIL_GROUPS group counters (each in its own 128-byte line), and the block that
            // completes a group draws on the launch's counter.  (Every block's stores precede its ticket, a group's tickets
            // precede its draw on the top counter: whoever completes that one has everybody's words behind it.)
            int last = 0;
            if (n_part <= TAIL_FLAT_MAX) {
                const int ticket = __hip_atomic_fetch_add(A.tail_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = (ticket == n_part - 1) ? 1 : 0;
            } else {
                const int g = part_pos % TAIL_GROUPS, gsize = (n_part - g + TAIL_GROUPS - 1) / TAIL_GROUPS;
                int32_t* const gc = A.tail_counter + 32 * (1 + g);
                const int t1 = __hip_atomic_fetch_add(gc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t1 == gsize - 1) {
                    __hip_atomic_store(gc, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int t2 = __hip_atomic_fetch_add(A.tail_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last = (t2 == TAIL_GROUPS - 1) ? 1 : 0;
                }
            }
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        if (threadIdx.x == 0) __hip_atomic_store(A.tail_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tail_work();
    };
    // PERSIST: the attempt's barrier.  Every block that has an item arrives once per attempt: its waves' stores are out (the barrier's own
    // workgroup release), one lane writes them back for the other XCDs and draws the ticket; the last arrival runs the tail
    // and publishes the generation, the others wait for it -- behind a deadline.  false: the launch is given up.
    auto persist_barrier = [&]() __attribute__((always_inline)) -> bool {
        __syncthreads();
        P_STAMP(2);
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const int ticket = __hip_atomic_fetch_add(A.tail_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (ticket == n_part - 1) ? 1 : 0;
        }
        P_STAMP(3);
        __syncthreads();
#ifdef OC_DIAG_BUILD
        (void)0;
#endif
        if (s_last) {
            // (no acquire here: everything the tail reads of the other blocks it reads with sc1 loads; the invalidation this
            // block needs before ITS next tile loads comes behind the publication, off everybody else's path)
            if (threadIdx.x == 0) __hip_atomic_store(A.tail_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tail_work();
            __syncthreads();
            P_STAMP(4);
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(A.p_sync, att + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            P_STAMP(5);
            if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
#ifdef OC_DIAG_BUILD
            if (A.p_stamps != nullptr && threadIdx.x == 0) {
                unsigned long long* o = A.p_stamps + (size_t)att * 16;
                for (int k = 0; k < 6; ++k) o[k] = p_st[k];
                o[6] = blockIdx.x;
                o[7] = (unsigned long long)n_items;
                for (int k = 0; k < 5; ++k) o[8 + k] = g_cb_stamp[k];
                o[15] = 1;
            }
#endif
            return true;
        }
        return wait_gen(0);
    };
    // the attempt's coefficients where a stage needs them (PERSIST: out of the block's LDS copy of the controller block)
    auto coef_of = [&](int q, int j) __attribute__((always_inline)) -> T {
        if constexpr (ROT) return uni(q < 3 ? s_tail.c.cA[q][j] : s_tail.c.cB[q - 3][j]);
        else return cc[q][j];
    };
    auto he_of = [&](int j) __attribute__((always_inline)) -> T {
        if constexpr (ROT) return uni(s_tail.c.he[j]);
        else return he[j];
    };
    // the tile's record for the fused tail's short-list path (thread 0; e_pos: the tile's position on the work list, or -1)
    auto put_rec = [&](int e_pos, int tile, double es, double ym, int flag) __attribute__((always_inline)) {
        if (FUSED && rec_mine != nullptr && e_pos >= 0 && e_pos < TAIL_REC_CAP) {
            double* r = rec_mine + 4 * e_pos;
            __hip_atomic_store(r, es, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(r + 1, ym, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<long long*>(r + 2), (long long)(uint32_t)tile | ((long long)flag << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (LEAD) __hip_atomic_store(reinterpret_cast<long long*>(r + 3), (long long)att, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // ---- one tile of the attempt (every thread of the block; block-uniform control flow)
    auto do_tile = [&](const int logical, const int e_pos) {
    const int vx = (int)threadIdx.x % RWV, ry = (int)threadIdx.x / RWV;
    const int gi = (logical / nbx) * TY - HY + ry, gj = (logical % nbx) * TX - HX + 4 * vx;
    const bool in = gi >= 0 && gi < Ny && gj >= 0 && gj < Nx;      // (Nx % 4 == 0: a vector is inside or outside as a whole)
    const int64_t idx = (int64_t)gi * Nx + gj;
    // the same as an unsigned 32-bit element offset (the grid has fewer than 2^30 cells: hjb_solve_dev checks): with a
    // wave-uniform array pointer the access is `base in scalar registers + 32-bit vector offset` -- one shift for all of the
    // launch's ten arrays, where a 64-bit index cost two 64-bit vector operations per array (the launch is issue-bound)
    const uint32_t ui = in ? (uint32_t)idx * (uint32_t)sizeof(T) : 0u;     // (BYTE offset)
    const int rslot = RING ? c_nacc : 0;                     // the ring slot of the step this attempt tries
    const bool ring_ok = RING && rslot < A.ring_cap;                // (beyond the capacity k_control ends the solve)
    // The thread's four cells as two PAIRS: float pairs are packed-math operands (v_pk_fma_f32 / v_pk_add_f32 /
    // v_pk_mul_f32: one instruction for two cells -- the launch is bound by VALU issue, 71 % busy before this), double
    // pairs compile to two scalar operations each.  Same IEEE operations per cell either way.
    typedef T T2 __attribute__((ext_vector_type(2)));
    T2 yv[2], cf[2], al[2], kv[NK][2], t[2];
    const T2 zero2 = {(T)0, (T)0};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        yv[p] = cf[p] = al[p] = t[p] = zero2;       // al: the wall mask (coef = NaN: optimals.py:162) folded into the two
#pragma unroll                                      // coefficients once: a wall cell evaluates 0 * sum - 0 * C = 0 in every
        for (int j = 0; j < NK; ++j) kv[j][p] = zero2;      // stage, without a compare and a select per stage
    }
    // ---- the exact active set.  A tile is TRIVIAL when on its whole region (tile + halo) y is one and the same power of
    // two u, K0 = 0 and coef = 0 or wall: then every stage evaluates a (4u - 4u) - 0 u = 0 exactly, so K1..K6 = 0, Q = 0,
    // y_new = u, error share 0, max|y_new| = u -- the block writes that (or finds it already written) without computing
    // anything.  With m = 0 that is the whole far field: phi stays exactly 1 until the stencil has carried the targets'
    // influence there, 6 cells per step (C4: 4 % of the tiles are non-trivial on average over a solve, 10 % at its end).
    // Results are bit-identical to computing every tile (tests/test_hip_active_set.py).
    //   flags (per tile and attempt: 1 trivial, 0 computed).  A tile that was trivial together with its 8 neighbours is
    //   trivial in the next attempt too (their outputs cover its region; a rejected attempt leaves the state as it was):
    //   build_work_list leaves such tiles off the next attempt's WORK LIST, and the launch's blocks beyond the list exit at once;
    //   a listed tile loads its region and tests it.  Its stores are skipped when the target buffers already hold the
    //   trivial values: it was trivial at the previous attempt and no rescale separates the buffers' scales.
    bool trivial = false, own_prev = false;
    T u_raw = (T)0;
    const bool track = sparse;
    const int bi = logical / nbx, bj = logical % nbx;
    // Every load of the prologue is issued before anything waits for one of them: y, coef and K0, then -- as scalar loads,
    // which wait on their own counter -- the tile's far-field probe and its previous flag (as first written the compiler had
    // put K0 behind the triviality test of y and coef and the flag behind the probe: three dependent round trips where one
    // does).
    V4<T> y4, c4, k4in[S0];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        y4.v[q] = c4.v[q] = (T)0;
#pragma unroll
        for (int j = 0; j < S0; ++j) k4in[j].v[q] = (T)0;
    }
    if (in) {
        y4 = ld4b(y, ui);
        c4 = ld4b(A.coef, ui);
#pragma unroll
        for (int j = 0; j < S0; ++j) k4in[j] = ld4b(kin[j], ui);
    }
    // Far-field substitution (ring solves): a trivial tile's cells are not kept up to date in the state buffers -- not when
    // it is found trivial, not across a change of scale (which used to put EVERY tile on the list twice, to rewrite the far
    // field of both buffers: 7 of a C4 solve's 103 launches carried 80 % of its HBM bytes).  Cells of a tile marked
    // undefined in the buffer being read are replaced by what they stand for: y = the controller's far-field value (raw, as
    // the buffer would hold it), K0 = 0.  A vector's four cells lie in one tile (TX = 4 * 28).
    const bool subst = RING && track && A.undef != nullptr;
    int32_t flag_prev = 0, my_undef = 0, own_undef = 0;
    if (subst && in) my_undef = A.undef[(size_t)par * nb + (gi / TY) * nbx + gj / TX];
    if (track) {
        // (PERSIST: other blocks of this launch wrote these words in the previous attempt -- ld_fresh, not the scalar cache)
        const T* const probe = y + ((int64_t)(bi * TY) * Nx + bj * TX);     // (the tile's first cell; y is read-only here)
        u_raw = PERSIST ? ld_fresh(probe) : uniform_load(probe);
        // (unconditional: behind a branch the flag's load waited for the probe's)
        const int32_t* const fp = (flags_prev != nullptr ? flags_prev : flags_cur) + logical;
        flag_prev = PERSIST ? ld_fresh(fp) : uniform_load(fp);
        if (subst) {
            const int32_t* const up = A.undef + ((size_t)par * nb + logical);
            own_undef = PERSIST ? ld_fresh(up) : uniform_load(up);
        }
    }
    store_ctl_words();
    own_prev = flags_prev != nullptr && flag_prev != 0;
    if (subst) {
        if (own_undef) u_raw = c_ubg;
        if (my_undef) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                y4.v[q] = c_ubg;
#pragma unroll
                for (int j = 0; j < S0; ++j) k4in[j].v[q] = (T)0;
            }
        }
    }
    // a tile that was computed in the previous attempt is computed again without the test (the front only advances; were
    // it ever trivial again, computing it is still right): the test's barrier is paid by the ring around the active region
    const bool need_test = track && (flags_prev == nullptr || own_prev);
    bool cell_trivial = true;
    if (in) {
        if (need_test) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                cell_trivial = cell_trivial && y4.v[q] == u_raw && (c4.v[q] == (T)0 || c4.v[q] != c4.v[q]);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            yv[p] = (T2){y4.v[2 * p], y4.v[2 * p + 1]} * (T2){in_scale, in_scale};      // (1 unless the previous step
            const bool w0 = c4.v[2 * p] != c4.v[2 * p], w1 = c4.v[2 * p + 1] != c4.v[2 * p + 1];     // ordered a rescale)
            al[p] = (T2){w0 ? (T)0 : A.a_lap, w1 ? (T)0 : A.a_lap};
            cf[p] = (T2){w0 ? (T)0 : c4.v[2 * p], w1 ? (T)0 : c4.v[2 * p + 1]};
        }
#pragma unroll
        for (int j = 0; j < S0; ++j) {
            const V4<T> k4 = k4in[j];
            kv[j][0] = (T2){k4.v[0], k4.v[1]};
            kv[j][1] = (T2){k4.v[2], k4.v[3]};
            if (need_test && j == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) cell_trivial = cell_trivial && k4.v[q] == (T)0;
            }
        }
        kv[0][0] *= (T2){in_scale, in_scale};
        kv[0][1] *= (T2){in_scale, in_scale};
    }
    auto quad = [](const T2& a, const T2& b) {
        V4<T> r;
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y;
        return r;
    };
    if (track) {
        // (u_raw == the controller's far-field value: every trivial tile then provably holds ONE and the same power of two --
        // what k_control's ymax for the tiles off the list and the ring's per-slot `u` assume; a tile that is constant at
        // any other value is simply computed)
        if (need_test) trivial = __syncthreads_and((cell_trivial && is_pow2(u_raw) && u_raw == c_ubg) ? 1 : 0) != 0;
        T_STAMP(0);
        if (trivial) {
            const T un = u_raw * in_scale;
            const bool can_skip = subst || (own_prev && (c_rej != 0 || (in_scale == (T)1 && c_oiscale == (T)1)));
            if (!can_skip && in && ry >= HY && ry < RH - HY && vx >= HX / 4 && vx < RWV - HX / 4) {
                V4<T> z, uu;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    z.v[q] = (T)0;
                    uu.v[q] = un;
                }
                if (!RING) {                    // (Q1 .. Q3 of a trivial tile: with the ring its flag says "all zero")
                    st4b(A.KM[1], ui, z);
                    st4b(A.KM[2], ui, z);
                    st4b(A.KM[3], ui, z);
                }
                st4b(A.KX[par ^ 1], ui, z);
                st4b(ynew, ui, uu);
            }
            if (threadIdx.x == 0) {
                const int r_lo = bi * TY, r_hi = min(r_lo + TY, Ny);
                const double um = (r_hi > A.row_lo && r_lo < A.row_hi) ? (double)fabs(un) : 0.0;
                st_part(A.partial + 2 * logical, 0.0);
                st_part(A.partial + 2 * logical + 1, um);
                st_flag(flags_cur + logical, 1);
                put_rec(e_pos, logical, 0.0, um, 1);
                if (subst) A.undef[(size_t)(par ^ 1) * nb + logical] = 1;      // (read by later attempts only)
            }
            return;
        }
        if (threadIdx.x == 0) {
            st_flag(flags_cur + logical, 0);
            if (subst) A.undef[(size_t)(par ^ 1) * nb + logical] = 0;          // (the tile's cells of y_new / K6 are written below)
            // (ring: the tile holds data from this slot on -- it was tested just now, so it was trivial until here; a tile
            // that skips the test has been computed before and is on record)
            if (RING && need_test && rslot < A.ring_first[logical]) A.ring_first[logical] = rslot;
        }
    }
    T* const slot0 = &buf[0][ry + 1][4 + 4 * vx];
    constexpr int BUF = (RH + 2) * LS;              // elements between the two buffers
    const T2 m4 = {(T)-4, (T)-4};
    // A wavefront holds 64 / RWV whole region rows.  Stage q's working value is only needed on rows at least q rows inside the
    // region (the rest is edge garbage that nothing valid reads), its derivative only on rows q + 1 inside: a wavefront all of
    // whose rows are nearer to the edge skips that part of the stage -- a wave-uniform branch; the barrier is outside it.
    // With 2 rows per wavefront and six stages that is 18 of the 96 wavefront-stages of a block.
    static_assert(64 % RWV == 0 && RH % (64 / RWV) == 0, "a wavefront holds whole region rows");
    constexpr int RPW = 64 / RWV;
    const int wr0 = (ry / RPW) * RPW;               // the wavefront's first row; its deepest row is wr0 + RPW - 1 or wr0
    const int depth = __builtin_amdgcn_readfirstlane(max(min(wr0 + RPW - 1, RH - 1 - (wr0 + RPW - 1)), min(wr0, RH - 1 - wr0)));
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        T* const slot = slot0 + (q & 1) * BUF;
        T lft_n = (T)0, rgt_n = (T)0;
        if (depth >= q) {
            T2 acc0 = zero2, acc1 = zero2;
#pragma unroll
            for (int j = 0; j < S0 + q; ++j) {
                const T cqj = coef_of(q, j);
                const T2 c2 = {cqj, cqj};
                acc0 = __builtin_elementwise_fma(c2, kv[j][0], acc0);
                acc1 = __builtin_elementwise_fma(c2, kv[j][1], acc1);
            }
            t[0] = yv[0] + acc0;
            t[1] = yv[1] + acc1;
            if (in) {
                const V4<T> t4 = quad(t[0], t[1]);
                st4(slot, t4);
                if (gi == 1 && ry >= 1) st4(slot - 2 * LS, t4);                  // ghost row -1 = row 1
                if (gi == Ny - 2 && ry + 2 <= RH - 1) st4(slot + 2 * LS, t4);   // ghost row Ny = row Ny-2
            }
            // left / right neighbours of the vector: the adjacent lanes' cells (a region row is RWV consecutive lanes; what
            // crosses a row's end is region-edge garbage by construction), or the mirror cell at the room's border
            lft_n = lane_prev(t[1].y);
            rgt_n = lane_next(t[0].x);
        }
        __syncthreads();
        if (depth >= q + 1) {
            const T lft = (gj == 0) ? t[0].y : lft_n, rgt = (gj + 3 == Nx - 1) ? t[1].x : rgt_n;
            const V4<T> up = ld4(slot - LS), dn = ld4(slot + LS);
            // hjb(t, phi, m) (rhs_value): a (N + S + W + E - 4 C) - coef C with the sum in the reference's order
            const T2 mid = {t[0].y, t[1].x};        // right neighbours of cells 0, 1 = left neighbours of cells 2, 3
            const T2 l0 = {lft, t[0].x}, r1 = {t[1].y, rgt};
            T2 s0 = (((T2){up.v[0], up.v[1]} + (T2){dn.v[0], dn.v[1]}) + l0) + mid;
            T2 s1 = (((T2){up.v[2], up.v[3]} + (T2){dn.v[2], dn.v[3]}) + mid) + r1;
            s0 = __builtin_elementwise_fma(m4, t[0], s0);
            s1 = __builtin_elementwise_fma(m4, t[1], s1);
            kv[S0 + q][0] = al[0] * s0 - cf[0] * t[0];
            kv[S0 + q][1] = al[1] * s1 - cf[1] * t[1];
        }
    }
    T_STAMP(1);
    const bool store = in && ry >= HY && ry < RH - HY && vx >= HX / 4 && vx < RWV - HX / 4;
    double esum = 0.0, ymax = 0.0;
    T ymax_t = (T)0;                    // (the maximum is exact in the state's own type: one conversion per thread)
    if (store) {
        if (FINAL) {
            // What k_out needs of K0 .. K6 is the quartic interpolant's Q = K^T P (scipy rk.py:178-180).  Column 0 of P is
            // e_0 (Q0 = K0, already in memory), row 1 is zero: the three columns Q1 .. Q3 are formed here, where every
            // K_j of the cell is in a register, and take the place of K2 .. K5 in memory -- one array pass less for this
            // launch, and k_out reads y + 4 arrays instead of y + 6.  (Same FMA chain as interp_coeffs: bit-identical.)
            // RING: y (its pending rescale applied, as k_out applies it), K0 and Q1 .. Q3 go to the step's ring slot instead
            // -- one 16-byte vector per plane and thread -- and nothing reads them before a lookup or the slice writer does.
            constexpr int rows[6] = {0, 2, 3, 4, 5, 6};
            // (idx is a multiple of 4: the thread's vector; value a of its four cells at rb + oc_ring_off(idx, a, n))
            T* const rb = RING ? A.ring + (size_t)rslot * A.ring_slot_elems : nullptr;
            const size_t rn = A.ring_slot_elems / OC_RING_VALS;
            // (streaming stores of whole lines: st4_ring; with the interleaved layout -- five 16-byte pieces per thread at an
            // 80-byte stride -- a dense launch took 49.7 us at 2048 x 2048 against 37.8 us with planes)
            if (RING && ring_ok) {
                st4_ring(rb, ui, 0, rn, quad(yv[0], yv[1]));
                st4_ring(rb, ui, 1, rn, quad(kv[0][0], kv[0][1]));
            }
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) {
                T2 q0 = zero2, q1 = zero2;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const T pj = ROT ? uni(s_p[j][pp]) : A.p[j][pp];
                    const T2 p2 = {pj, pj};
                    q0 = __builtin_elementwise_fma(kv[rows[j]][0], p2, q0);
                    q1 = __builtin_elementwise_fma(kv[rows[j]][1], p2, q1);
                }
                if (!RING) st4b(A.KM[1 + pp], ui, quad(q0, q1));
                else if (ring_ok) st4_ring(rb, ui, 2 + pp, rn, quad(q0, q1));
            }
            st4b(A.KX[par ^ 1], ui, quad(kv[NK - 1][0], kv[NK - 1][1]));      // K6 = the next step's K0 (FSAL)
        } else {
            st4b(A.KM[0], ui, quad(kv[S0][0], kv[S0][1]));
            st4b(A.KM[1], ui, quad(kv[S0 + 1][0], kv[S0 + 1][1]));
            st4b(A.KM[2], ui, quad(kv[S0 + 2][0], kv[S0 + 2][1]));
        }
        if (FINAL) {
            st4b(ynew, ui, quad(t[0], t[1]));       // the last stage's working value IS y_new
            if (gi >= A.row_lo && gi < A.row_hi) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const T he6 = he_of(6);
                    T2 e = (T2){he6, he6} * kv[NK - 1][p];
#pragma unroll
                    for (int j = 0; j < NK - 1; ++j) {
                        const T hej = he_of(j);
                        e = __builtin_elementwise_fma((T2){hej, hej}, kv[j][p], e);
                    }
                    const T e2[2] = {e.x, e.y}, y2[2] = {yv[p].x, yv[p].y}, t2[2] = {t[p].x, t[p].y};
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const T sc = atol_s + fmax(fabs(y2[c]), fabs(t2[c])) * A.rtol;
                        const T qq = err_quot(e2[c], sc);
                        esum += (double)qq * (double)qq;
                        ymax_t = fmax(ymax_t, fabs(t2[c]));
                    }
                }
            }
        }
    }
    T_STAMP(2);
    if (FINAL) {
        // The block's error share and max |y_new|: every thread leaves its pair in the tile buffer the last stage did not
        // use (all reads of it ended before that stage's barrier), ONE wavefront adds them up -- 16 cross-lane float64
        // reductions (12 ds_bpermute round trips each) per block became one; the launch is bound by instruction issue.
        static_assert(sizeof(buf) / 2 >= sizeof(double2) * NT, "the reduction's pairs fit one tile buffer");
        double2* const red = reinterpret_cast<double2*>(&buf[NQ & 1][0][0]);
        ymax = (double)ymax_t;
        red[threadIdx.x] = make_double2(esum, ymax);
        __syncthreads();
        if (threadIdx.x < 64) {
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int q = 0; q < NT / 64; ++q) {
                const double2 v = red[threadIdx.x + 64 * q];
                a += v.x;
                b = fmax(b, v.y);
            }
            // (DPP trees in a fixed association: as six __shfl_down steps each -- 24 ds_bpermute round trips -- the pair was
            // 1 us at the end of every tile, on the attempt's chain)
            a = wave_tree_sum(a);
            b = wave_tree_max(b);
            if (threadIdx.x == 0) {
                st_part(A.partial + 2 * logical, a);
                st_part(A.partial + 2 * logical + 1, b);
                put_rec(e_pos, logical, a, b, 0);
            }
        }
    }
    T_STAMP(3);
    };      // do_tile
    if constexpr (!PERSIST) {
        // this block's item: position `logical` of the work list, or tile `logical` (NO loop over further items here: with a
        // loop around it the tile code's invariants are hoisted and spilled -- 20 % on the whole solve; a launch too small
        // for its list does nothing and says so: att_expected above)
        const bool from_list = sparse && work != nullptr;
        do_tile(from_list ? w_entry : logical, from_list ? logical : -1);
        P_STAMP(4);
        if constexpr (FUSED) {
            if (!LEAD || lead_has_tail) {
                __syncthreads();        // (the tail reuses the tile buffers and the reduction's LDS words)
                fused_tail();
            }
        }
#ifdef OC_DIAG_BUILD
        if (LEAD && A.p_stamps != nullptr && threadIdx.x == 0 && logical == 0 && att < 4096) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            p_st[5] = __builtin_amdgcn_s_memrealtime();
            unsigned long long* o = A.p_stamps + (size_t)att * 16;
            for (int k = 0; k < 6; ++k) o[k] = p_st[k];
            o[6] = blockIdx.x;
            o[7] = (unsigned long long)n_items;
            for (int k = 0; k < 5; ++k) o[8 + k] = g_cb_stamp[k];
            o[6] = g_cb_stamp[7];
            o[13] = g_cb_stamp[5];
            o[14] = g_cb_stamp[6];
            o[15] = lead_has_tail ? 3 : 2;
        }
        if (LEAD && A.p_stamps != nullptr && threadIdx.x == 0 && logical == n_items / 2 && att < 2048) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            p_st[5] = __builtin_amdgcn_s_memrealtime();
            unsigned long long* o = A.p_stamps + (size_t)(2048 + att) * 16;
            for (int k = 0; k < 6; ++k) o[k] = p_st[k];
            for (int k = 0; k < 4; ++k) o[8 + k] = g_tl_stamp[k];
            o[7] = (unsigned long long)n_items;
            o[15] = lead_has_tail ? 5 : 4;
        }
#endif
        return;
    } else {
    // PERSIST: this block's items of the attempt -- entries blockIdx.x, blockIdx.x + gridDim.x, ... of the work list, or of
    // all tiles in the XCD-aware order of the one-attempt launch (a block stays on its XCD: gridDim.x is a multiple of 8)
    {
        const bool from_list = sparse && work != nullptr;
        const int n_loop = from_list ? n_items : 8 * per;
        for (int e = (int)blockIdx.x; e < n_loop; e += (int)gridDim.x) {
            const int lg = from_list ? (e == (int)blockIdx.x ? w_entry : ld_fresh(work + e)) : (e % 8) * per + e / 8;
            if (lg < nb) do_tile(lg, from_list ? e : -1);
            __syncthreads();            // (the next tile, or the tail, reuses the tile buffers and the reduction's words)
        }
    }
    if (!persist_barrier()) return;
    }
  }
}

// Dense output + direction field for every t_eval point of one accepted step, in one pass: the quartic
// interpolant's coefficients Q = K^T P (scipy rk.py:178-180) stay in registers, each slice's phi goes through a
// double-buffered LDS tile, and vels (optimals.py:168-186) writes the slice of the (padded) direction field.
template <typename T>
struct OutArgs {
    const HjbCtl<T>* ctl;   // the accepted step: buffer roles, t, h, its t_eval range [k_lo, k_hi), 2^expo
    T* Y[2];
    T* KX[2];
    T* KM[5];
    T p[6][4];              // rows 0, 2, 3, 4, 5, 6 of P (row 1 is zero)
    void* vx_base;          // direction field (nslice, Ny, Nx) of the kernel's field type TO (= T, or float with a
    void* vy_base;          // double state: OC_F64_F32): slice nt - 1 - k  <->  t_eval[k]
    size_t slice_elems;
    double* keep_base;      // audit: phi(t_eval[k]) unscaled, float64, slot keep_slot[k] (or NULL)
    const int32_t* keep_slot;
    T lim, mu, inv2dx, inv2dy;
    int Ny, Nx;
    const int32_t* flags;   // the accepted attempt's stage-tile flags (stx x sty cells per tile), or NULL
    int stx, sty;
    // block 0 builds the next attempt's work list (build_work_list) when `work` is set
    HjbCtl<T>* ctl_rw;
    int32_t* flags_next;
    int32_t* work;
    double* parts_rw;
    const T* ring_slot;     // RING: the slot's data (k_ring_ctl filled ctl from its record)
};

template <typename T, bool QIN, bool RING = false>
__device__ __forceinline__ void interp_coeffs(const T* __restrict__ y, const T* const (&k)[6], const T (&P)[6][4],
                                              T in_scale, int64_t idx, int n, V4<T>& yq, V4<T> (&Q)[4], size_t rn = 0) {
    if (RING) {
        // y = a ring slot (planes y | K0 | Q1 | Q2 | Q3, oc_internal.h): the values k_stageV stored, the rescale already
        // applied -- what the other branch holds after its loads and multiplications
        // (rn: cells per slot -- the plane stride of the planar layout, oc_internal.h)
        if (n == 4) {
            yq = ld4(y + oc_ring_off((size_t)idx, 0, rn));
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) Q[pp] = ld4(y + oc_ring_off((size_t)idx, 1 + pp, rn));
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) yq.v[q] = (T)0;
            yq.v[0] = y[oc_ring_off((size_t)idx, 0, rn)];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
                for (int q = 0; q < 4; ++q) Q[pp].v[q] = (T)0;
                Q[pp].v[0] = y[oc_ring_off((size_t)idx, 1 + pp, rn)];
            }
        }
        return;
    }
    // n = 4: one vector of cells, n = 1: a single cell (element 0).  y and K0 (k[0]) carry the step's pending rescale
    // (HjbCtl).  Q[pp].v[q]: coefficient pp of cell q.
    // QIN: k = {K0, Q1, Q2, Q3} -- the stage launch (k_stageV) already formed the columns of Q = K^T P (Q0 = K0);
    // otherwise k = {K0, K2 .. K6} (the host-controlled cross-check variants) and Q is formed here, by the same FMAs.
    constexpr int NA = QIN ? 4 : 6;
    V4<T> kk[NA];
    if (n == 4) {
        yq = ld4(y + idx);
#pragma unroll
        for (int j = 0; j < NA; ++j) kk[j] = ld4(k[j] + idx);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) yq.v[q] = (T)0;
        yq.v[0] = y[idx];
#pragma unroll
        for (int j = 0; j < NA; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) kk[j].v[q] = (T)0;
            kk[j].v[0] = k[j][idx];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        yq.v[q] *= in_scale;
        kk[0].v[q] *= in_scale;
    }
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
        if (QIN) {
            Q[pp] = kk[pp];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                T acc = (T)0;
#pragma unroll
                for (int j = 0; j < 6; ++j) acc = fma_t(kk[j].v[q], P[j][pp], acc);
                Q[pp].v[q] = acc;
            }
        }
    }
}

// The next attempt's work list (exact active set, k_stageV), built by block 0 of k_out beside the output blocks.  A tile that
// was trivial in the attempt just run, with all 8 neighbours, is trivial in the next one whatever the decision was; if the
// target buffers hold the trivial values already (the attempt was rejected, or no rescale lies between the buffers) it needs
// no block at all: its flag and error share are set here.  sflag: the block's LDS (nbytes of it).
template <typename T, bool SC1>
__device__ __forceinline__ void build_work_list(HjbCtl<T>* __restrict__ ctl, const int32_t* __restrict__ flags_cur,
                                                int32_t* __restrict__ flags_next, int32_t* __restrict__ work,
                                                double* __restrict__ parts_rw, int tnbx, int tnby, uint8_t* sflag, int nbytes,
                                                volatile double* mailbox, bool flags_ready, const ListPre* pre,
                                                int32_t* wsum_ext, int32_t* undef) {
    // Called by every thread of the block (barriers inside), and every thread works: the builder is a chain of dependent
    // integer and LDS instructions per tile, so its length is the tiles per THREAD (8 per thread with 256 threads at
    // 2048 x 2048 took 8 us of the fused tail; 2 per thread with the stage block's 1024).  SC1: flags_cur was written by
    // other blocks of this very launch (fused tail) -- sc1 loads, see ld_part.
    // flags_ready: sflag[] already holds the tiles' flags (the fused tail's idle wavefronts fetched them meanwhile)
    __shared__ int32_t wsum_own[16];
    int32_t* const wsum = wsum_ext != nullptr ? wsum_ext : wsum_own;
    const int todo = ctl->list_todo, nth = (int)blockDim.x;
    const int ntile = tnbx * tnby;
    const int band_lo = ctl->band_lo, band_hi = ctl->band_hi;
    if (todo == 0) return;
    if (todo == 2) {                                // every tile is active and stays so: no flags, no list from now on
        if (threadIdx.x == 0) {
            ctl->dense_mode = 1;
            if (mailbox) mailbox[1] = (double)ntile + 0.5;       // (+ 0.5: "every tile, for the rest of the solve" -- the host
                                                                 // stops enqueuing the list builder's launch)
        }
        return;
    }
    // undef (ring solves, far-field substitution: Stage3Args): nobody reads a trivial tile's cells, so a change of scale does
    // not have to rewrite them -- tiles may stay off the list whatever the decision was
    const bool skip_ok =
        (undef != nullptr || ctl->rejected != 0 || (ctl->in_scale == (T)1 && ctl->out_in_scale == (T)1)) && ntile <= nbytes;
    // every thread owns a contiguous run of tiles (an ordered list keeps neighbouring tiles in neighbouring blocks)
    const int per_thr = (ntile + nth - 1) / nth;
    int t_lo = min((int)threadIdx.x * per_thr, ntile), t_hi = min(t_lo + per_thr, ntile);
    uint32_t listed = 0xFFFFFFFFu, listed_hi = 0xFFFFFFFFu;     // bit q: tile t_lo + q goes on the list (per_thr <= 64)
    // (fused tail: the tiles per thread, the masks, the counts and the wavefronts' prefix sums were formed by list_prepare
    // while thread 0 ran the controller -- valid whenever the decision allows tiles to stay off the list, which is the rule)
    const bool use_pre = pre != nullptr && pre->valid && skip_ok && wsum_ext != nullptr;
    if (use_pre) {
        listed = pre->listed;
        listed_hi = pre->listed_hi;
        t_lo = pre->t_lo;
        t_hi = pre->t_hi;
    } else if (skip_ok) {
        if (!flags_ready) {
            // (eight independent loads in flight per thread: as sc1 loads in a one-load-per-trip loop each trip waited for
            // its own round trip to the L2)
            for (int t0 = threadIdx.x; t0 < ntile; t0 += 8 * nth) {
                int32_t f[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) f[u] = (t0 + u * nth < ntile) ? ld_flag<SC1>(flags_cur + t0 + u * nth) : 0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (t0 + u * nth < ntile) sflag[t0 + u * nth] = f[u] != 0 ? 1 : 0;
            }
        }
        __syncthreads();
        int bi = t_lo / tnbx, bj = t_lo % tnbx;
        for (int t = t_lo; t < t_hi; ++t) {
            uint32_t q9 = 1u;
#pragma unroll
            for (int di = -1; di <= 1; ++di)
#pragma unroll
                for (int dj = -1; dj <= 1; ++dj) {
                    const int ni = bi + di, nj = bj + dj;
                    const bool inside = ni >= 0 && ni < tnby && nj >= 0 && nj < tnbx;
                    q9 &= inside ? (uint32_t)sflag[inside ? ni * tnbx + nj : t] : 1u;
                }
            const int q = t - t_lo;
            if (q9 && bi >= band_lo && bi < band_hi) {      // (tile rows that touch a neighbour's rows stay on the list)
                if (q < 32) listed &= ~(1u << q);
                else listed_hi &= ~(1u << (q - 32));
            }
            if (++bj == tnbx) {
                bj = 0;
                ++bi;
            }
        }
    }
    const int nmine = t_hi - t_lo;
    const uint32_t m_lo = nmine >= 32 ? 0xFFFFFFFFu : ((1u << max(nmine, 0)) - 1u);
    const uint32_t m_hi = nmine >= 64 ? 0xFFFFFFFFu : (nmine > 32 ? ((1u << (nmine - 32)) - 1u) : 0u);
    listed &= m_lo;
    listed_hi &= m_hi;
    int mine = __popc(listed) + __popc(listed_hi);
    int incl = mine;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = (nth + 63) >> 6;
    if (use_pre) {
        mine = pre->mine;
        incl = pre->incl;           // (wsum[] was posted before the controller's barriers)
    } else {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
    }
    int off = incl - mine, total = 0;
    for (int q = 0; q < nwv; ++q) {
        off += q < wv ? wsum[q] : 0;
        total += wsum[q];
    }
    // A tile off the list needs its flag (and a zero error share) in the NEXT attempt's buffers.  The active set only grows
    // -- the tiles off the list after the first attempt are all that ever will be (a change of scale without the far-field
    // substitution lists every tile twice, and every tile then writes its own words) -- so from the third build of a solve on
    // both flag buffers and the partials already hold exactly that for every tile that can be off the list: only the
    // solve's first builds write them.
    const bool write_unlisted = ctl->n_attempt <= 2;
    for (int t = t_lo; t < t_hi; ++t) {
        const int q = t - t_lo;
        const bool on = q < 32 ? ((listed >> q) & 1u) != 0 : ((listed_hi >> (q - 32)) & 1u) != 0;
        if (on) {
            work[off++] = t;
        } else if (write_unlisted) {
            flags_next[t] = 1;
            parts_rw[2 * t] = 0.0;
            parts_rw[2 * t + 1] = 0.0;
            if (undef != nullptr) undef[t] = undef[ntile + t] = 1;     // (in either state buffer: it stands for the far field)
        }
    }
    if (threadIdx.x == 0) {
        ctl->work_count = total;
        if (mailbox) mailbox[1] = (double)total;     // (the host picks the launch sequence by it: fused tail or not)
    }
}

// The decision-independent half of build_work_list, for the fused tail: called by every thread beside the controller's serial
// section (sflag[] holds the attempt's flags and a barrier lies behind their stores).  Same masks and scan as there; the
// tiles are dealt to wavefronts 1 .. only: wavefront 0 arrives when its thread 0 has decided, the others meanwhile (the list
// comes out in the same -- ascending -- order of tiles either way).
__device__ __forceinline__ ListPre list_prepare(int tnbx, int tnby, const uint8_t* sflag, int32_t* wsum) {
    ListPre r;
    const int nth = (int)blockDim.x - 64, ntile = tnbx * tnby, tid = (int)threadIdx.x - 64;
    const int per_thr = (ntile + nth - 1) / nth;
    const int t_lo = tid < 0 ? 0 : min(tid * per_thr, ntile), t_hi = tid < 0 ? 0 : min(t_lo + per_thr, ntile);
    uint32_t listed = 0xFFFFFFFFu, listed_hi = 0xFFFFFFFFu;
    int bi = t_lo / tnbx, bj = t_lo % tnbx;
    for (int t = t_lo; t < t_hi; ++t) {
        uint32_t q9 = 1u;
#pragma unroll
        for (int di = -1; di <= 1; ++di)
#pragma unroll
            for (int dj = -1; dj <= 1; ++dj) {
                const int ni = bi + di, nj = bj + dj;
                const bool inside = ni >= 0 && ni < tnby && nj >= 0 && nj < tnbx;
                q9 &= inside ? (uint32_t)sflag[inside ? ni * tnbx + nj : t] : 1u;
            }
        const int q = t - t_lo;
        if (q9) {
            if (q < 32) listed &= ~(1u << q);
            else listed_hi &= ~(1u << (q - 32));
        }
        if (++bj == tnbx) {
            bj = 0;
            ++bi;
        }
    }
    const int nmine = t_hi - t_lo;
    const uint32_t m_lo = nmine >= 32 ? 0xFFFFFFFFu : ((1u << max(nmine, 0)) - 1u);
    const uint32_t m_hi = nmine >= 64 ? 0xFFFFFFFFu : (nmine > 32 ? ((1u << (nmine - 32)) - 1u) : 0u);
    r.listed = listed & m_lo;
    r.listed_hi = listed_hi & m_hi;
    r.mine = __popc(r.listed) + __popc(r.listed_hi);
    int incl = r.mine;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wv] = incl;
    r.incl = incl;
    r.t_lo = t_lo;
    r.t_hi = t_hi;
    r.valid = per_thr <= 64;
    return r;
}

template <typename T>
__device__ __forceinline__ void build_work_list(HjbCtl<T>* __restrict__ ctl, const int32_t* __restrict__ flags_cur,
                                                int32_t* __restrict__ flags_next, int32_t* __restrict__ work,
                                                double* __restrict__ parts_rw, int tnbx, int tnby, uint8_t* sflag, int nbytes,
                                                volatile double* mailbox = nullptr, int32_t* undef = nullptr) {
    build_work_list<T, false>(ctl, flags_cur, flags_next, work, parts_rw, tnbx, tnby, sflag, nbytes, mailbox, false, nullptr,
                              nullptr, undef);
}

// With the interpolant ring no k_out launch follows an attempt: the list builder is a launch of its own (one block).
template <typename T>
__global__ void __launch_bounds__(256) k_list(HjbCtl<T>* ctl, const int32_t* __restrict__ flags_cur,
                                              int32_t* __restrict__ flags_next, int32_t* __restrict__ work,
                                              double* __restrict__ parts_rw, int tnbx, int tnby,
                                              volatile double* mailbox, int32_t* undef) {
    __shared__ uint8_t sflag[16384];
    build_work_list<T>(ctl, flags_cur, flags_next, work, parts_rw, tnbx, tnby, sflag, (int)sizeof(sflag), mailbox, undef);
}

// ---- slices from the ring (oc_hjb_ring_slices; the audit output phi_keep of a ring solve) ---------------------------
// One slot at a time: k_ring_ctl turns the slot's record into the controller block k_out reads (the t_eval range clamped
// to the slices [s_lo, s_hi) that were asked for), k_ring_fill writes (u, 0, 0, 0, 0) into the slot's stage tiles that
// were trivial in the accepted attempt -- what k_stageV leaves in the state arrays of such a tile, and what a lookup
// substitutes for it -- and k_out<.., RING> then runs exactly the arithmetic of the slice-writing solve.
template <typename T>
__global__ void k_ring_ctl(const OcRingHead* __restrict__ rh, const OcRingSlot<T>* __restrict__ rslots, int slot, int s_lo,
                           int s_hi, HjbCtl<T>* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const OcRingSlot<T> rs = rslots[slot];
    const int nt = rh->nt;
    out->accepted = 0;
    out->done = 0;
    out->dense_mode = 1;
    out->out_parity = 0;
    out->nt = nt;
    out->te_step = rh->te_step;
    out->Tfinal = rh->Tfinal;
    out->out_t = rs.out_t;
    out->out_hh = rs.out_hh;
    out->out_unscale = rs.out_unscale;
    out->out_lim_s = rs.lim_s;
    out->out_in_scale = (T)1;
    out->u_bg = rs.u;
    out->n_hx = 0;                                  // (k_out computes the slice factors itself: the same expression)
    out->list_todo = 0;
    // slice s <-> t_eval index k = nt - 1 - s: s in [s_lo, s_hi)  <=>  k in [nt - s_hi, nt - s_lo)
    const int k_lo = max(rs.k_lo, nt - s_hi), k_hi = min(rs.k_hi, nt - s_lo);
    out->k_lo = k_lo;
    out->k_hi = k_hi;
    out->work_count = rs.dense;                     // (borrowed: 1 = the slot has no trivial tiles to fill)
    if (slot < rh->n_slots && k_lo < k_hi) out->accepted = 1;
}

template <typename T>
__global__ void __launch_bounds__(256) k_ring_fill(const HjbCtl<T>* __restrict__ ctl, T* __restrict__ slot_data,
                                                   const int32_t* __restrict__ first, int slot, int Ny, int Nx) {
    if (!ctl->accepted || ctl->work_count != 0) return;
    const int t = blockIdx.x;
    if (slot >= first[t]) return;                   // the tile holds data in this slot
    const int tnbx = (Nx + OC_RING_STX - 1) / OC_RING_STX;
    const int r0 = (t / tnbx) * OC_RING_STY, c0 = (t % tnbx) * OC_RING_STX;
    const T u = ctl->u_bg;
    constexpr int VPR = OC_RING_STX / 4;            // vectors of four cells per tile row
    for (int q = threadIdx.x; q < VPR * OC_RING_STY; q += blockDim.x) {
        const int r = r0 + q / VPR, c = c0 + 4 * (q % VPR);
        if (r >= Ny || c >= Nx) continue;
        const size_t ci = (size_t)r * Nx + c, rn = (size_t)Ny * Nx;
        V4<T> uu, z;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uu.v[e] = u;
            z.v[e] = (T)0;
        }
        st4(slot_data + oc_ring_off(ci, 0, rn), uu);
#pragma unroll
        for (int a = 1; a < OC_RING_VALS; ++a) st4(slot_data + oc_ring_off(ci, a, rn), z);
    }
}

// Tile = OX x OY cells with OX * OY = 1024 (256 threads x 4 consecutive cells of a row): 64 x 16, 128 x 8 or 256 x 4.  A
// wider tile writes longer contiguous runs per row and slice (1 KB at 256 instead of 256 B), which is what counts where
// the input arrays no longer come out of the Infinity Cache (4096 x 4096).
template <typename T, int OX, typename TO, bool QIN, bool RING = false>
__global__ void __launch_bounds__(256, sizeof(T) == 4 ? 4 : 3) k_out(OutArgs<T> A) {      // float: 4 waves per SIMD (at most 128 VGPRs)
    constexpr int OY = 1024 / OX, VW = OX / 4, OLS = OX + 8;
    __shared__ alignas(16) T tile[2][OY + 2][OLS];
    const HjbCtl<T>* __restrict__ ctl = A.ctl;
    const int bshift = A.work != nullptr ? 1 : 0;
    if (bshift && blockIdx.x == 0) {                // the list builder (also after a rejected attempt)
        build_work_list<T>(A.ctl_rw, A.flags, A.flags_next, A.work, A.parts_rw, (A.Nx + A.stx - 1) / A.stx,
                           (A.Ny + A.sty - 1) / A.sty, reinterpret_cast<uint8_t*>(&tile[0][0][0]), (int)sizeof(tile));
        return;
    }
    if (!ctl->accepted) return;                     // a rejected attempt (or one past the end) has nothing to emit
    const int par = ctl->out_parity;
    const int k_lo = ctl->k_lo, k_hi = ctl->k_hi, nt = ctl->nt;
    const double t0 = ctl->out_t, hh = ctl->out_hh, unscale = ctl->out_unscale, te_step = ctl->te_step, Tf = ctl->Tfinal;
    const T lim_s = ctl->out_lim_s, in_scale = ctl->out_in_scale;
    const int n_hx = ctl->n_hx;
    const T* __restrict__ y = RING ? A.ring_slot : A.Y[par];       // (RING: everything comes from the slot's blocks)
    // K0, Q1 .. Q3 (left by k_stageV in the place of K2 .. K4)  /  K0, K2 .. K5, K6
    const T* const kk[6] = {A.KX[par], A.KM[1], A.KM[2], A.KM[3], QIN ? nullptr : A.KM[4], QIN ? nullptr : A.KX[par ^ 1]};
    const int cx = (int)threadIdx.x % VW, ry = (int)threadIdx.x / VW;
    const int Ny = A.Ny, Nx = A.Nx;
    const int nbx = (Nx + OX - 1) / OX, nby = (Ny + OY - 1) / OY, nb = nbx * nby;
    const int per = (nb + 7) / 8;
    const int bid = (int)blockIdx.x - bshift;
    const int logical = (bid % 8) * per + bid / 8;              // XCD-aware order (tile_of_block)
    if (bid / 8 >= per || logical >= nb) return;
    const int i0 = (logical / nbx) * OY, j0 = (logical % nbx) * OX;
    const int gi = i0 + ry, gj = j0 + 4 * cx;
    const bool valid = gi < Ny && gj < Nx;
    const int64_t idx = (int64_t)gi * Nx + gj;
    const int64_t ncell = (int64_t)Ny * Nx;
    // the exact active set (k_stageV): where every stage tile under this tile and its one-cell ring was trivial in the
    // accepted attempt, phi(t) is one constant there for every t of the step -- central differences 0, direction 0 -- and
    // the slices are written as zeros without reading or computing anything (the audit output `keep` takes the full path)
    if (A.flags != nullptr && A.keep_slot == nullptr && ctl->dense_mode == 0) {
        const int snbx = (Nx + A.stx - 1) / A.stx;
        const int r0 = max(i0 - 1, 0) / A.sty, r1 = min(i0 + OY, Ny - 1) / A.sty;
        const int c0 = max(j0 - 1, 0) / A.stx, c1 = min(j0 + OX, Nx - 1) / A.stx;
        bool triv = true;
        for (int r = r0; r <= r1; ++r)
            for (int c = c0; c <= c1; ++c) triv = triv && A.flags[r * snbx + c] != 0;
        if (triv) {
            if (valid) {
                V4<TO> z;
#pragma unroll
                for (int q = 0; q < 4; ++q) z.v[q] = (TO)0;
                for (int k = max(k_lo, 1); k < k_hi; ++k) {
                    st4_nt((TO*)A.vx_base + (size_t)(nt - 1 - k) * A.slice_elems + idx, z);
                    st4_nt((TO*)A.vy_base + (size_t)(nt - 1 - k) * A.slice_elems + idx, z);
                }
            }
            return;
        }
    }
    V4<T> yq, Q[4], hy, hQ[4];
    if (valid) interp_coeffs<T, QIN, RING>(y, kk, A.p, in_scale, idx, 4, yq, Q, A.slice_elems);
    // halo of the tile: threads 0 .. VW-1 the row above (a vector each), the next VW the row below, then OY threads the
    // column to the left and OY the column to the right (a cell each); cells outside the grid are never used
    HaloRole hr;
    {
        const int hidx = (int)threadIdx.x;
        hr.kind = 0;
        hr.gi = hr.gj = hr.li = hr.lj = 0;
        if (hidx < VW) { hr.kind = 1; hr.gi = i0 - 1; hr.gj = j0 + 4 * hidx; hr.li = 0; hr.lj = 4 + 4 * hidx; }
        else if (hidx < 2 * VW) { const int l = hidx - VW; hr.kind = 2; hr.gi = i0 + OY; hr.gj = j0 + 4 * l; hr.li = OY + 1; hr.lj = 4 + 4 * l; }
        else if (hidx < 2 * VW + OY) { const int l = hidx - 2 * VW; hr.kind = 3; hr.gi = i0 + l; hr.gj = j0 - 1; hr.li = l + 1; hr.lj = 3; }
        else if (hidx < 2 * VW + 2 * OY) { const int l = hidx - 2 * VW - OY; hr.kind = 4; hr.gi = i0 + l; hr.gj = j0 + OX; hr.li = l + 1; hr.lj = 4 + OX; }
        if (!(hr.gi >= 0 && hr.gi < Ny && hr.gj >= 0 && hr.gj < Nx)) hr.kind = 0;
    }
    const int hn = (hr.kind == 1 || hr.kind == 2) ? 4 : (hr.kind >= 3 ? 1 : 0);
    if (hn) interp_coeffs<T, QIN, RING>(y, kk, A.p, in_scale, (int64_t)hr.gi * Nx + hr.gj, hn, hy, hQ, A.slice_elems);
    auto clip = [&](T v) { return (v > lim_s) ? v : ((v < lim_s) ? lim_s : (T)0); };
    // the controller's per-slice factors: fetched once (a load from the controller block inside the slice loop would
    // put a memory round trip into every slice)
    int bmask = 0;                                  // cells of this thread on the grid's border ring (written as 0)
#pragma unroll
    for (int q = 0; q < 4; ++q) bmask |= (gi < 1 || gi >= Ny - 1 || gj + q < 1 || gj + q >= Nx - 1) ? (1 << q) : 0;
    __shared__ T s_hx[HJB_MAXS][4];
    if ((int)threadIdx.x < 4 * n_hx) s_hx[threadIdx.x >> 2][threadIdx.x & 3] = ctl->hx[threadIdx.x >> 2][threadIdx.x & 3];
    __syncthreads();
    int it = 0;
    for (int k = k_lo; k < k_hi; ++k) {
        // t_eval[k] <-> slice nt - 1 - k of the field; k = 0 (phi = 1 at t = T) is never used (optimals.py:200-204)
        const bool want_v = k >= 1 && A.vx_base != nullptr;
        double* keep = (A.keep_slot && A.keep_slot[k] >= 0) ? A.keep_base + (size_t)A.keep_slot[k] * ncell : nullptr;
        if (!want_v && !keep) continue;
        T h1, h2, h3, h4;
        if (n_hx) {                                 // the controller's table (uniform loads)
            h1 = s_hx[k - k_lo][0];
            h2 = s_hx[k - k_lo][1];
            h3 = s_hx[k - k_lo][2];
            h4 = s_hx[k - k_lo][3];
        } else {
            const double te = (k == nt - 1) ? 0.0 : (double)k * te_step + Tf;
            const double x = (te - t0) / hh;
            double xp = x;
            h1 = (T)(hh * xp);
            xp *= x;
            h2 = (T)(hh * xp);
            xp *= x;
            h3 = (T)(hh * xp);
            xp *= x;
            h4 = (T)(hh * xp);
        }
        TO* vxs = want_v ? (TO*)A.vx_base + (size_t)(nt - 1 - k) * A.slice_elems : nullptr;
        TO* vys = want_v ? (TO*)A.vy_base + (size_t)(nt - 1 - k) * A.slice_elems : nullptr;
        const int buf = it & 1;
        ++it;
        // The tile holds the CLIPPED phi (optimals.py:172: every use below is of the clipped value), so each cell is
        // clipped once instead of five times -- once as a centre and once by each of its four neighbours.
        V4<T> pcl;
#pragma unroll
        for (int q = 0; q < 4; ++q) pcl.v[q] = (T)0;
        if (valid) {
            V4<T> ph;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ph.v[q] = yq.v[q] + fma_t(Q[3].v[q], h4, fma_t(Q[2].v[q], h3, fma_t(Q[1].v[q], h2, Q[0].v[q] * h1)));
                pcl.v[q] = clip(ph.v[q]);
            }
            st4(&tile[buf][ry + 1][4 + 4 * cx], pcl);
            if (keep) {
#pragma unroll
                for (int q = 0; q < 4; ++q) keep[idx + q] = (double)ph.v[q] * unscale;
            }
        }
        if (hn == 4) {
            V4<T> hp;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                hp.v[q] = clip(hy.v[q] + fma_t(hQ[3].v[q], h4, fma_t(hQ[2].v[q], h3, fma_t(hQ[1].v[q], h2, hQ[0].v[q] * h1))));
            st4(&tile[buf][hr.li][hr.lj], hp);
        } else if (hn == 1) {
            tile[buf][hr.li][hr.lj] = clip(hy.v[0] + fma_t(hQ[3].v[0], h4, fma_t(hQ[2].v[0], h3, fma_t(hQ[1].v[0], h2, hQ[0].v[0] * h1))));
        }
        // left / right neighbours of the vector: the adjacent lanes (DPP wave shifts; as LDS scalars at a 16-byte stride
        // they were 4-way bank-conflicted), the tile's halo columns for the first and the last vector of a row
        const T lft_n = lane_prev(pcl.v[3]), rgt_n = lane_next(pcl.v[0]);
        __syncthreads();
        if (vxs && valid) {
            const V4<T> up = ld4(&tile[buf][ry][4 + 4 * cx]), dn = ld4(&tile[buf][ry + 2][4 + 4 * cx]);
            const T lft = (cx == 0) ? tile[buf][ry + 1][3] : lft_n, rgt = (cx == VW - 1) ? tile[buf][ry + 1][4 + OX] : rgt_n;
            V4<TO> ox, oy;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // central differences of the clipped phi (optimals.py:172-180); "north" = row i+1
                const T pw = (q == 0) ? lft : pcl.v[q - 1], pe = (q == 3) ? rgt : pcl.v[q + 1];
                const T ps = up.v[q], pn = dn.v[q];
                const T pc = clip(pcl.v[q]);                // the reference clips the centre a second time (:177)
                const T gx = (pe - pw) * A.inv2dx, gy = (pn - ps) * A.inv2dy;
                T vxo, vyo;
                direction_of<T>(gx, gy, A.mu * pc, A.lim, vxo, vyo);
                const bool border = (bmask >> q) & 1;
                ox.v[q] = border ? (TO)0 : (TO)vxo;
                oy.v[q] = border ? (TO)0 : (TO)vyo;
            }
            st4_nt(vxs + idx, ox);
            st4_nt(vys + idx, oy);
        }
    }
}

// After an attempt of a row-block solve: the `halo` owned rows next to each neighbour of (y, K0-to-be) go to the staging
// buffers (dtype (2, halo, Nx) each) and, after the caller's exchange, the neighbours' rows come back into the halo rows.
// After a rejected attempt the exchange moves rows that are already in place (every rank must still take part).
template <typename T, bool PACK>
__global__ void k_halo_stage(const HjbCtl<T>* __restrict__ ctl, T* Y0, T* Y1, T* K0, T* K1, T* up, T* dn, size_t lo,
                             size_t hi, size_t hn, int has_up, int has_dn) {
    const int par = ctl->parity;
    T* y = par ? Y1 : Y0;
    T* k = par ? K1 : K0;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 4 * hn) return;
    const int which = (int)(g / hn);
    const size_t e = g % hn;
    T* arr = (which & 1) ? k : y;
    if (which < 2) {
        if (!has_up) return;
        if (PACK) up[(which & 1) * hn + e] = arr[lo + e];
        else arr[lo - hn + e] = up[(which & 1) * hn + e];
    } else {
        if (!has_dn) return;
        if (PACK) dn[(which & 1) * hn + e] = arr[hi - hn + e];
        else arr[hi + e] = dn[(which & 1) * hn + e];
    }
}

// ------------------------------------------------------------------------------------------ host side
// k_out's tile: 128 x 8 cells while the step's input arrays fit the 256 MB Infinity Cache (C4: 6.18 TB/s against 5.69
// with 64 x 16 and 6.04 with 256 x 4), 256 x 4 beyond (C5, 4096 x 4096: 5.44 TB/s against 4.59 / 5.11) -- or the caller's
// choice (oc_set_tuning "out_tile").  QIN: the stage launch left Q1 .. Q3 (the product path); false: K2 .. K6.
template <typename T, typename TO, bool QIN, bool RING = false>
void launch_out(int shape, int Nx, int Ny, hipStream_t s, const OutArgs<T>& OA) {
    if (shape < 0) shape = (5.0 * sizeof(T) * (double)Nx * Ny > 256e6) ? 2 : 1;
    const int ox = shape == 2 ? 256 : (shape == 1 ? 128 : 64), oy = 1024 / ox;
    const int nb = ((Nx + ox - 1) / ox) * ((Ny + oy - 1) / oy);
    const dim3 grid(8 * ((nb + 7) / 8) + (OA.work != nullptr ? 1 : 0));     // (+ block 0: the work-list builder)
    if (shape == 2) hipLaunchKernelGGL((k_out<T, 256, TO, QIN, RING>), grid, dim3(256), 0, s, OA);
    else if (shape == 1) hipLaunchKernelGGL((k_out<T, 128, TO, QIN, RING>), grid, dim3(256), 0, s, OA);
    else hipLaunchKernelGGL((k_out<T, 64, TO, QIN, RING>), grid, dim3(256), 0, s, OA);
}

// typed views of a caller's ring (oc_hip.h: oc_hjb_ring; layout: oc_internal.h)
template <typename T>
struct RingView {
    T* data;
    OcRingHead* head;
    OcRingSlot<T>* slots;
    OcRingSlice<T>* slices;
    int32_t* first;             // [ntile]
    size_t slot_elems;
    int cap, nslice, rows, ntile;
};
template <typename T>
RingView<T> ring_view(const oc_hjb_ring* r, int Nx) {
    RingView<T> v;
    char* m = (char*)r->meta;
    v.data = (T*)r->data;
    v.head = (OcRingHead*)m;
    v.slots = (OcRingSlot<T>*)(m + oc_ring_slots_off());
    v.slices = (OcRingSlice<T>*)(m + oc_ring_slices_off(sizeof(T), r->slots));
    v.first = (int32_t*)(m + oc_ring_flags_off(sizeof(T), r->slots, r->nslice));
    v.slot_elems = (size_t)OC_RING_VALS * r->rows * Nx;
    v.cap = r->slots;
    v.nslice = r->nslice;
    v.rows = r->rows;
    v.ntile = oc_ring_ntile(r->rows, Nx);
    return v;
}

// slices [s_lo, s_hi) of the direction field (and / or the audit output phi at kept t_eval points) from a ring: slice s
// goes to vx_out[s - s_lo].  n_slots: the solve's accepted steps (the caller read the ring's head).
template <typename T, typename TO>
int ring_emit(oc_handle* h, const RingView<T>& R, int n_slots, int s_lo, int s_hi, void* vx_out, void* vy_out,
              double* keep_base, const int32_t* keep_slot_dev, hipStream_t s) {
    const oc_config& c = h->cfg;
    const int Ny = R.rows, Nx = c.Nx;
    if (!h->hjb_ctl) OC_HIP_CHECK(hipMalloc(&h->hjb_ctl, 8192));
    HjbCtl<T>* ctl = (HjbCtl<T>*)((char*)h->hjb_ctl + 4096);       // (behind a solve's two alternating controller blocks)
    static_assert(sizeof(HjbCtl<T>) <= 2048, "controller block too large");
    OutArgs<T> OA;
    memset(&OA, 0, sizeof(OA));
    OA.ctl = ctl;
    OA.slice_elems = (size_t)Ny * Nx;
    // slice s lands at index s - s_lo of the caller's arrays (only addresses of asked-for slices are ever formed)
    OA.vx_base = vx_out ? (void*)((TO*)vx_out - (ptrdiff_t)s_lo * (ptrdiff_t)OA.slice_elems) : nullptr;
    OA.vy_base = vy_out ? (void*)((TO*)vy_out - (ptrdiff_t)s_lo * (ptrdiff_t)OA.slice_elems) : nullptr;
    OA.keep_base = keep_base;
    OA.keep_slot = keep_slot_dev;
    OA.lim = (T)c.lim;
    OA.mu = (T)c.mu;
    OA.inv2dx = (T)(1.0 / (2 * c.dx));
    OA.inv2dy = (T)(1.0 / (2 * c.dy));
    OA.Ny = Ny;
    OA.Nx = Nx;
    OA.stx = OC_RING_STX;
    OA.sty = OC_RING_STY;
    for (int slot = 0; slot < n_slots; ++slot) {
        T* sd = R.data + (size_t)slot * R.slot_elems;
        hipLaunchKernelGGL(k_ring_ctl<T>, dim3(1), dim3(1), 0, s, (const OcRingHead*)R.head, (const OcRingSlot<T>*)R.slots, slot,
                           s_lo, s_hi, ctl);
        hipLaunchKernelGGL(k_ring_fill<T>, dim3(R.ntile), dim3(256), 0, s, (const HjbCtl<T>*)ctl, sd, (const int32_t*)R.first,
                           slot, Ny, Nx);
        OA.ring_slot = sd;
        launch_out<T, TO, true, true>(h->out_tile, Nx, Ny, s, OA);
    }
    OC_HIP_CHECK(hipGetLastError());
    return OC_OK;
}

template <typename T>
struct Solver {
    oc_handle* h;
    hipStream_t s;
    int Ny, Nx;
    int64_t n;
    T *y, *ynew, *ytmp, *K[7], *phi_tmp;
    const T* coef;
    T a_lap;
    int nfev = 0;
    const oc_hjb_shard* sh = nullptr;   // row-block solve: reductions and halos go through the caller's callbacks

    int alloc() {
        size_t need = sizeof(T) * (size_t)n * 12;
        if (h->hjb_buf == nullptr || h->hjb_bytes < need) {
            if (h->hjb_buf) OC_HIP_CHECK(hipFree(h->hjb_buf));
            h->hjb_buf = nullptr;
            h->hjb_bytes = 0;
            OC_HIP_CHECK(hipMalloc(&h->hjb_buf, need));
            h->hjb_bytes = need;
        }
        T* b = (T*)h->hjb_buf;
        y = b;
        ynew = b + n;
        ytmp = b + 2 * n;
        for (int k = 0; k < 7; ++k) K[k] = b + (3 + k) * n;
        phi_tmp = b + 10 * n;
        return OC_OK;
    }
    dim3 g1() const { return dim3((unsigned)((n + 255) / 256)); }
    dim3 b2() const { return dim3(64, 4); }
    dim3 g2(int nx, int ny) const { return dim3((nx + 63) / 64, (ny + 3) / 4); }

    void rhs(const T* p, T* out) {
        hipLaunchKernelGGL(k_rhs<T>, g2(Nx, Ny), b2(), 0, s, p, coef, out, Ny, Nx, a_lap);
        ++nfev;
    }
    int fused_blocks() const { return ((Nx + HT_X - 1) / HT_X) * ((Ny + HT_Y - 1) / HT_Y); }
    int fused_grid() const { return 8 * ((fused_blocks() + 7) / 8); }
    int reduce(double* sum, double* mx, int nblocks = 1024) {   // partials written by the preceding kernel
        if (sh) {
            // every rank contributes (sum, max) of its owned rows; the caller's collective gathers them (on `s`) and
            // every rank adds them up in rank order, so all ranks take bit-identical controller decisions
            hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s, h->red_partial, nblocks, sh->red_local);
            if (sh->comm_slot >= 0) {
                int rcc = oc_comm_allgather_f64_oop(h, sh->comm_slot, sh->red_local, sh->red_all, 2, s);
                if (rcc) return rcc;
            } else if (sh->allgather_red(sh->user) != 0) {
                oc_set_error("hjb_solve_rows: the allgather_red callback failed");
                return OC_ERR_STATE;
            }
            OC_HIP_CHECK(hipMemcpyAsync(h->red_host, sh->red_all, 2 * sizeof(double) * sh->world,
                                        hipMemcpyDeviceToHost, s));
            OC_HIP_CHECK(hipStreamSynchronize(s));
            double a = 0.0, m = 0.0;
            for (int r = 0; r < sh->world; ++r) {
                a += h->red_host[2 * r];
                m = std::max(m, h->red_host[2 * r + 1]);
            }
            *sum = a;
            if (mx) *mx = m;
            return OC_OK;
        }
        // the reduction kernel posts (sum, max, sequence number) into pinned host memory; the host polls the sequence
        // number (a copy + stream synchronise costs ~10 us more per attempt: an extra command and the wake-up)
        volatile double* mb = h->red_host + 2 * OC_MAX_WORLD;
        const double seq = (double)(++h->red_seq);
        hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s, h->red_partial, nblocks, h->red_out,
                           (volatile double*)h->red_mailbox_dev, seq);
        bool got = false;
        for (long spin = 0; spin < 20000000L; ++spin) {         // ~ a second at worst, then fall back to the stream
            if (mb[2] == seq) {
                got = true;
                break;
            }
            if ((spin & 1023) == 1023 && hipStreamQuery(s) == hipSuccess) break;    // (the stream drained: read below)
        }
        if (!got) {
            OC_HIP_CHECK(hipStreamSynchronize(s));
            if (mb[2] != seq) {             // mapped memory not visible: read the device copy
                OC_HIP_CHECK(hipMemcpy(h->red_host, h->red_out, 2 * sizeof(double), hipMemcpyDeviceToHost));
                *sum = h->red_host[0];
                if (mx) *mx = h->red_host[1];
                return OC_OK;
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        *sum = mb[0];
        if (mx) *mx = mb[1];
        return OC_OK;
    }
};

template <typename T, int SG>
void launch_stage(Solver<T>& S, StageArgs<T>& A) {
    hipLaunchKernelGGL((k_stage<T, SG, false>), dim3(S.fused_grid()), dim3(256), 0, S.s, A);
}

// the six RHS evaluations of one RK attempt as six fused launches (stages 1..5 + y_new/K6/error)
template <typename T>
void launch_stages_fused(Solver<T>& S, double hh, T atol_s, T rtol) {
    StageArgs<T> A;
    memset(&A, 0, sizeof(A));
    A.y = S.y;
    A.coef = S.coef;
    A.Ny = S.Ny;
    A.Nx = S.Nx;
    A.a_lap = S.a_lap;
    A.atol = atol_s;
    A.rtol = rtol;
    A.partial = S.h->red_partial;
    A.row_lo = S.sh ? S.sh->own_lo : 0;
    A.row_hi = S.sh ? S.sh->own_hi : S.Ny;
    for (int sg = 1; sg < 6; ++sg) {
        for (int j = 0; j < 6; ++j) {
            A.k[j] = S.K[j < sg ? j : 0];
            A.c[j] = j < sg ? (T)(RK_A[sg][j] * hh) : (T)0;
        }
        A.out = S.K[sg];
        switch (sg) {
            case 1: launch_stage<T, 1>(S, A); break;
            case 2: launch_stage<T, 2>(S, A); break;
            case 3: launch_stage<T, 3>(S, A); break;
            case 4: launch_stage<T, 4>(S, A); break;
            default: launch_stage<T, 5>(S, A); break;
        }
    }
    const int nz[5] = {0, 2, 3, 4, 5};                 // B_1 = E_1 = 0
    for (int q = 0; q < 5; ++q) {
        A.k[q] = S.K[nz[q]];
        A.c[q] = (T)(RK_B[nz[q]] * hh);
        A.he[q] = (T)(RK_E[nz[q]] * hh);
    }
    A.he6 = (T)(RK_E[6] * hh);
    A.out = S.K[6];
    A.ynew = S.ynew;
    hipLaunchKernelGGL((k_stage<T, 5, true>), dim3(S.fused_grid()), dim3(256), 0, S.s, A);
}

// ---- the solve with the device-resident controller (hjb variant 0; single GPU and row blocks) ---------------------
// Host part: SciPy's select_initial_step (three reductions, read back once each), then attempts are ENQUEUED -- stage
// launch A, stage launch B, [row blocks: local reduction, the caller's all-gather], k_control, [row blocks: pack, the
// caller's halo exchange, unpack], k_out -- a few ahead of the device, never waiting for one: the host
// watches the controller's tick count in a pinned mailbox only to stop when `done` is posted.  In a row-block solve the
// ranks must enqueue the same number of attempts (every attempt holds collectives), so there the host synchronises
// the stream once per burst of attempts and all ranks read the same `done` (their controllers are bit-identical).
template <typename T, typename TO>
int hjb_solve_dev(oc_handle* h, const void* coef_v, double Tfinal, int32_t nt, void* vx_out, void* vy_out,
                  const int32_t* keep_idx, int32_t n_keep, double* phi_keep, double* t_steps, int32_t t_steps_cap,
                  oc_hjb_stats* st, hipStream_t s, const oc_hjb_shard* sh, const oc_hjb_ring* ring = nullptr) {
    const oc_config& c = h->cfg;
    Solver<T> S;
    S.h = h;
    S.s = s;
    S.sh = sh;
    S.Ny = sh ? sh->Ny_local : c.Ny;
    S.Nx = c.Nx;
    S.n = (int64_t)S.Ny * S.Nx;
    OC_REQUIRE(S.n * (int64_t)sizeof(T) < ((int64_t)1 << 32), "grid too large: the stage kernel addresses an array by 32-bit byte offsets");
    const int own_lo = sh ? sh->own_lo : 0, own_hi = sh ? sh->own_hi : S.Ny;
    const int64_t own_off = (int64_t)own_lo * S.Nx, n_own = (int64_t)(own_hi - own_lo) * S.Nx;
    const double n_rms = sh ? (double)sh->Ny_global * S.Nx : (double)S.n;
    S.coef = (const T*)coef_v;
    S.a_lap = (T)(-0.5 * c.sigma * c.sigma / (c.dx * c.dy));
    int rc = S.alloc();
    if (rc) return rc;
    const int64_t n = S.n;
    const double rtol = c.rtol, atol0 = c.atol;
    const int RB = 1024;
    oc_hjb_stats stats;
    memset(&stats, 0, sizeof(stats));
    stats.h_min = 1e300;
    if (nt <= 1) {   // linspace(T,0,nt) has no slice that the fill loop would use (optimals.py:200)
        if (st) *st = stats;
        return OC_OK;
    }
    // ---- device-side bookkeeping (allocated once per handle)
    const int T_CAP = 65536, LOG_CAP = 65536;
    if (!h->hjb_ctl) OC_HIP_CHECK(hipMalloc(&h->hjb_ctl, 8192));
    if (!h->hjb_tsteps) OC_HIP_CHECK(hipMalloc((void**)&h->hjb_tsteps, sizeof(double) * T_CAP));
    if (!h->hjb_log) OC_HIP_CHECK(hipMalloc((void**)&h->hjb_log, sizeof(int32_t) * LOG_CAP));
    if (!h->hjb_tail) {
        OC_HIP_CHECK(hipMalloc((void**)&h->hjb_tail, TAIL_COUNTER_BYTES + 64));     // (+ the persistent launch's two words)
        OC_HIP_CHECK(hipMemset(h->hjb_tail, 0, TAIL_COUNTER_BYTES + 64));
    }
    static_assert(sizeof(HjbCtl<T>) <= 2048, "controller block too large");
    // two controller blocks: attempt a reads ctlb[a & 1], its controller leaves the next attempt's in ctlb[(a + 1) & 1]
    HjbCtl<T>* const ctlb[2] = {(HjbCtl<T>*)h->hjb_ctl, (HjbCtl<T>*)((char*)h->hjb_ctl + 2048)};
    HjbCtl<T>* ctl = ctlb[0];
    int32_t* keep_slot_dev = nullptr;
    if (n_keep > 0) {
        std::vector<int32_t> slot(nt, -1);
        for (int q = 0; q < n_keep; ++q) {
            OC_REQUIRE(keep_idx[q] >= 0 && keep_idx[q] < nt, "keep index out of range");
            slot[keep_idx[q]] = q;
        }
        OC_HIP_CHECK(hipMalloc((void**)&keep_slot_dev, sizeof(int32_t) * nt));
        OC_HIP_CHECK(hipMemcpy(keep_slot_dev, slot.data(), sizeof(int32_t) * nt, hipMemcpyHostToDevice));
    }
    struct Guard {                      // the keep table lives for this call only
        int32_t* p;
        ~Guard() { if (p) (void)hipFree(p); }
    } guard{keep_slot_dev};

    // buffers: Y[0] = y, Y[1] = y_new; KX[0] = K[0], KX[1] = K[6] (FSAL pair); KM = K1 .. K5
    T* Y[2] = {S.y, S.ynew};
    T* KX[2] = {S.K[0], S.K[6]};
    const double t0 = Tfinal, tf = 0.0, direction = -1.0;
    double* const part_b = h->red_partial + 2 * (size_t)h->red_blocks + 2 * 4 * 128;    // (a second set of 1024 pairs: oc_api.hip)
    if (sh == nullptr) {                // (whole grid: y0, f0 and the first two norms' partials in one launch -- k_init_a)
        hipLaunchKernelGGL(k_init_a<T>, dim3(RB), dim3(256), 0, s, S.coef, S.y, KX[0], S.a_lap, (T)atol0, (T)rtol, n, h->red_partial, part_b);
        ++S.nfev;
    } else {
        hipLaunchKernelGGL(k_fill<T>, S.g1(), dim3(256), 0, s, S.y, (T)1, n);
        S.rhs(S.y, KX[0]);              // f0 = K0 of the first step
    }
    // ---- select_initial_step (scipy common.py:68-134), order = 4: on the device (k_init_step) for a whole-grid solve; a
    // row-block solve completes each norm with the caller's all-gather, read back by the host
    const bool dev_init = sh == nullptr;
    double h_abs = 0.0;
    if (dev_init) {
        const double interval = std::fabs(tf - t0);
        double* red = h->red_out;
        hipLaunchKernelGGL(k_init_r0, dim3(1), dim3(1024), 0, s, (const double*)h->red_partial, (const double*)part_b, RB, red, n_rms, interval);
        hipLaunchKernelGGL(k_init_b<T>, dim3(RB), dim3(256), 0, s, S.coef, (const T*)KX[0], (const double*)red, direction, S.Ny, S.Nx,
                           S.a_lap, (T)atol0, (T)rtol, n, h->red_partial);
        ++S.nfev;
        // (k_init_r1 -- the third reduction and the first step -- runs once the controller block has been uploaded, below)
    } else {
        double interval = std::fabs(tf - t0);
        double s0, s1, s2;
        hipLaunchKernelGGL((k_norm_scaled<T, 0>), dim3(RB), dim3(256), 0, s, S.y + own_off, (const T*)nullptr,
                           S.y + own_off, (T)atol0, (T)rtol, n_own, h->red_partial);
        if ((rc = S.reduce(&s0, nullptr))) return rc;
        hipLaunchKernelGGL((k_norm_scaled<T, 0>), dim3(RB), dim3(256), 0, s, KX[0] + own_off, (const T*)nullptr,
                           S.y + own_off, (T)atol0, (T)rtol, n_own, h->red_partial);
        if ((rc = S.reduce(&s1, nullptr))) return rc;
        double d0 = std::sqrt(s0) / std::sqrt(n_rms), d1 = std::sqrt(s1) / std::sqrt(n_rms);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        h0 = std::min(h0, interval);
        Lin8<T> L;
        L.n = 1;
        L.k[0] = KX[0];
        L.c[0] = (T)(h0 * direction);
        hipLaunchKernelGGL(k_lincomb<T>, S.g1(), dim3(256), 0, s, S.y, L, S.ytmp, n);
        S.rhs(S.ytmp, S.K[1]);
        hipLaunchKernelGGL((k_norm_scaled<T, 1>), dim3(RB), dim3(256), 0, s, S.K[1] + own_off, KX[0] + own_off,
                           S.y + own_off, (T)atol0, (T)rtol, n_own, h->red_partial);
        if ((rc = S.reduce(&s2, nullptr))) return rc;
        double d2 = std::sqrt(s2) / std::sqrt(n_rms) / h0;
        double h1;
        if (d1 <= 1e-15 && d2 <= 1e-15)
            h1 = std::max(1e-6, h0 * 1e-3);
        else
            h1 = std::pow(0.01 / std::max(d1, d2), 1.0 / 5.0);
        h_abs = std::min(std::min(100 * h0, h1), interval);
    }
    stats.h_abs0 = h_abs;
    // ---- the controller's initial state
    HjbCtl<T> hc;
    memset(&hc, 0, sizeof(hc));
    hc.t = t0; hc.tf = tf; hc.Tfinal = Tfinal; hc.h_abs = h_abs; hc.rtol = rtol; hc.atol0 = atol0; hc.n_rms = n_rms;
    hc.te_step = (0.0 - Tfinal) / (double)(nt - 1);
    hc.lim = c.lim;
    hc.h_min = 1e300;
    hc.nt = nt;
    hc.is_f32 = sizeof(T) == 4 ? 1 : 0;
    hc.in_scale = hc.out_in_scale = (T)1;
    hc.u_bg = (T)1;
    hc.work_count = 1 << 30;            // (the first attempt looks at every tile)
    hc.h_abs0 = h_abs;
    {   // tile rows whose region (tile +- 6 rows) reaches into rows a neighbour sends: always on the work list
        const int TYr = OC_RING_STY, HYr = 6;       // (the one-launch attempt's tile: the only one the active set runs with)
        const int tnby0 = (S.Ny + TYr - 1) / TYr;
        hc.band_lo = 0;
        hc.band_hi = tnby0;
        if (sh && sh->has_up) hc.band_lo = std::min(tnby0, (sh->own_lo + HYr - 1) / TYr + 1);
        if (sh && sh->has_dn) {
            int b = tnby0;
            while (b > 0 && (b - 1) * TYr + TYr - 1 + HYr >= sh->own_hi) --b;      // tile row b - 1 reaches row own_hi or beyond
            hc.band_hi = b;
        }
    }
    if (!dev_init) ctl_prepare_attempt(hc, true);
    // (through a pinned staging block: no synchronisation -- the block is rewritten by the next solve at the earliest)
    if (!h->hjb_ctl_host) OC_HIP_CHECK(hipHostMalloc(&h->hjb_ctl_host, OC_CTL_HOST_BYTES, hipHostMallocDefault));
    if (!dev_init) {                    // (whole-grid solves: the block goes up as a kernel argument of k_init_r1)
        memcpy(h->hjb_ctl_host, &hc, sizeof(hc));
        OC_HIP_CHECK(hipMemcpyAsync(ctl, h->hjb_ctl_host, sizeof(hc), hipMemcpyHostToDevice, s));
    }
    InitBlocks<T> init_blocks;
    memset(&init_blocks, 0, sizeof(init_blocks));
    init_blocks.ctl = hc;
    WordFills fills;
    memset(&fills, 0, sizeof(fills));
    int n_fills = 0;
    auto add_fill = [&](void* p, size_t bytes, uint32_t v) {
        fills.p[n_fills] = (uint32_t*)p;
        fills.n[n_fills] = (int32_t)(bytes / 4);
        fills.v[n_fills] = v;
        ++n_fills;
    };
    static_assert(sizeof(HjbCtl<T>) % 4 == 0 && TAIL_COUNTER_BYTES % 4 == 0, "word fills");
    add_fill(ctlb[1], sizeof(hc), 0xFFFFFFFFu);         // (ticks = -1 until the first attempt's controller has written it)
    add_fill(h->hjb_tail, TAIL_COUNTER_BYTES, 0u);      // (a solve that failed half-way may have left tickets)
    volatile double* mb = h->red_host + 2 * OC_MAX_WORLD + 4;      // [0] ticks + 2^32 * done (k_control), [1] listed tiles,
    mb[0] = 0.0;                                                   // [2] 1 + ticks of an attempt whose launch was too small
    mb[1] = 1e9;
    mb[2] = 0.0;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    double* mb_dev = (double*)h->red_mailbox_dev + 4;

    // ---- kernel arguments (fixed for the whole solve)
    Stage3Args<T> SA;
    memset(&SA, 0, sizeof(SA));
    SA.ctl = ctl;
    SA.Y[0] = Y[0]; SA.Y[1] = Y[1];
    SA.KX[0] = KX[0]; SA.KX[1] = KX[1];
    for (int j = 0; j < 5; ++j) SA.KM[j] = S.K[1 + j];
    SA.coef = S.coef;
    SA.rtol = (T)rtol;
    SA.a_lap = S.a_lap;
    {
        const int rows[6] = {0, 2, 3, 4, 5, 6};
        for (int q = 0; q < 6; ++q)
            for (int pp = 0; pp < 3; ++pp) SA.p[q][pp] = (T)RK_P[rows[q]][pp + 1];
    }
    SA.partial = h->red_partial;
    SA.Ny = S.Ny;
    SA.Nx = S.Nx;
    SA.row_lo = own_lo;
    SA.row_hi = own_hi;
    SA.flags_prev = nullptr;
    SA.flags_cur = nullptr;
    SA.work = nullptr;
    // stage launches of an attempt (k_stageV): hjb_fuse 0 = all six stages in one launch (tile 112 x 20), 1 = two launches of
    // three stages (tile 120 x 26; measured slower, kept as a cross-check of the other)
    const int fuse = ring ? 0 : h->hjb_fuse;        // (the ring is written by the one-launch attempt)
    const int tx = fuse == 0 ? OC_RING_STX : 120, ty = fuse == 0 ? OC_RING_STY : 26;
    static_assert(OC_ST_RWV * OC_ST_RH == 1024 && 64 % OC_ST_RWV == 0, "the stage kernel's region: 1024 threads x 4 cells");
    RingView<T> R;
    memset(&R, 0, sizeof(R));
    if (ring) {
        OC_REQUIRE(ring->data && ring->meta && ring->slots >= 1, "empty ring");
        OC_REQUIRE(ring->rows == S.Ny, "the ring's row count differs from the solve's");
        OC_REQUIRE(ring->nslice >= nt - 1, "horizon exceeds the ring's slice table");
        R = ring_view<T>(ring, S.Nx);
        SA.ring = R.data;
        SA.ring_slot_elems = R.slot_elems;
        SA.ring_first = R.first;
        // (nothing has been computed yet: the stage launches record each tile's first slot)
        add_fill(R.first, sizeof(int32_t) * (size_t)R.ntile, 0x7F7F7F7Fu);
        // the ring's head but for the number of accepted steps, which the controller that ends the solve writes (control_body)
        OcRingHead* const rh0 = (OcRingHead*)((char*)h->hjb_ctl_host + 3072);
        memset(rh0, 0, sizeof(*rh0));
        rh0->n_slots = 0;
        rh0->nt = nt;
        rh0->cap = R.cap;
        rh0->nslice = R.nslice;
        rh0->te_step = hc.te_step;
        rh0->Tfinal = hc.Tfinal;
        if (dev_init) {
            init_blocks.head = *rh0;
            init_blocks.head_dst = R.head;
        } else {
            OC_HIP_CHECK(hipMemcpyAsync(R.head, rh0, sizeof(*rh0), hipMemcpyHostToDevice, s));
        }
        SA.ring_cap = R.cap;
        SA.ring_ntile = R.ntile;
    }
    const int nb3 = ((S.Nx + tx - 1) / tx) * ((S.Ny + ty - 1) / ty);
    const dim3 grid3(8 * ((nb3 + 7) / 8));
    if (nb3 > h->red_blocks) {
        oc_set_error("hjb_solve: %d tiles exceed the reduction buffer (%d)", nb3, h->red_blocks);
        return OC_ERR_STATE;
    }
    OutArgs<T> OA;
    memset(&OA, 0, sizeof(OA));
    OA.ctl = ctl;
    OA.Y[0] = Y[0]; OA.Y[1] = Y[1];
    OA.KX[0] = KX[0]; OA.KX[1] = KX[1];
    for (int j = 0; j < 5; ++j) OA.KM[j] = S.K[1 + j];
    {
        const int rows[6] = {0, 2, 3, 4, 5, 6};
        for (int q = 0; q < 6; ++q)
            for (int pp = 0; pp < 4; ++pp) OA.p[q][pp] = (T)RK_P[rows[q]][pp];
    }
    OA.vx_base = vx_out;
    OA.vy_base = vy_out;
    OA.slice_elems = (size_t)S.Ny * S.Nx;
    OA.keep_base = phi_keep;
    OA.keep_slot = keep_slot_dev;
    OA.lim = (T)c.lim;
    OA.mu = (T)c.mu;
    OA.inv2dx = (T)(1.0 / (2 * c.dx));
    OA.inv2dy = (T)(1.0 / (2 * c.dy));
    OA.Ny = S.Ny;
    OA.Nx = S.Nx;
    OA.flags = nullptr;
    OA.stx = tx;
    OA.sty = ty;
    // the exact active set (k_stageV): two flag arrays, one per attempt parity.  Single-GPU, one-launch attempts only: in a
    // row-block solve activity also arrives through the halo exchange, which the tile flags do not see.
    // On tiny grids the bookkeeping costs more than the skipped tiles save (room_test, 20 tiles: 6.4 against 5.8 ms per solve; the
    // threshold was ~1000 tiles while a sparse attempt cost 22 us): hjb_active = 1 switches by size (OC_ACTIVE_MIN_TILES), 2 forces it.
    // Above 12 000 tiles (grids beyond ~5000 x 5000) the list builder's per-thread masks and its LDS copy of the flags no
    // longer cover the grid: every tile is computed.
    // Row blocks (round 4): the same flags and lists per rank, with the tile rows that reach into a neighbour's rows always
    // listed (HjbCtl::band_lo / band_hi), the far field added in the rank's own reduction (k_reduce_local) and without the
    // far-field substitution (what a rank SENDS must be in its buffers).  The size rule looks at the whole grid's tiles.
    const int nb_rule = sh ? ((S.Nx + tx - 1) / tx) * ((sh->Ny_global + ty - 1) / ty) : nb3;
    const bool active_set = fuse == 0 && nb3 <= 12000 && (!sh || h->hjb_active_rows != 0) &&
                            (h->hjb_active == 2 || (h->hjb_active == 1 && nb_rule >= OC_ACTIVE_MIN_TILES));
    if (active_set && h->hjb_flags_n < (size_t)6 * nb3) {     // flags of two attempts + two work lists (alternating) +
        if (h->hjb_flags) OC_HIP_CHECK(hipFree(h->hjb_flags));     // the undefined-tile marks of the two state buffers
        h->hjb_flags = nullptr;
        h->hjb_flags_n = 0;
        OC_HIP_CHECK(hipMalloc((void**)&h->hjb_flags, sizeof(int32_t) * 6 * (size_t)nb3));
        h->hjb_flags_n = (size_t)6 * nb3;
    }
    // far-field substitution (Stage3Args::undef): ring solves; every tile of both buffers starts out defined
    SA.undef = (active_set && ring && !sh && h->hjb_subst != 0) ? h->hjb_flags + (size_t)4 * nb3 : nullptr;
    if (SA.undef) add_fill(SA.undef, sizeof(int32_t) * 2 * (size_t)nb3, 0u);
    if (dev_init)       // (the resets ride along with the first step)
        hipLaunchKernelGGL(k_init_r1<T>, dim3(1), dim3(1024), 0, s, ctl, init_blocks, (const double*)h->red_partial, RB, h->red_out, n_rms, std::fabs(tf - t0), fills);
    else
        hipLaunchKernelGGL(k_fill_words, dim3(8), dim3(256), 0, s, fills);
    // attempt a reads work list a & 1; the builder after it writes list (a + 1) & 1 (a late block of the attempt's own launch
    // must never see the next attempt's list: the fused tail builds it while that launch is still draining)
    auto work_list = [&](int a) -> int32_t* { return active_set ? h->hjb_flags + (2 + (size_t)(a & 1)) * nb3 : nullptr; };
    // fused tail (ring solves on one GPU): the attempt's last block is its controller and list builder -- one launch per
    // attempt instead of three.  Every participating block takes a ticket on ONE counter (~90 atomics per us on this chip),
    // so it pays while few blocks take part: small grids, or the active set's short lists.  The host picks the sequence per
    // attempt from what the device last posted (mb[1]: listed tiles); any choice is correct, only the speed differs.
    // hjb_tail = 1 (default): the fused launch while few blocks take part (the flat ticket's range), stage launch + k_control
    // otherwise; 2: every attempt fused, two-level tickets beyond that range -- measured at 4096 x 4096 with every tile
    // computed: 193 us per launch against 163 + k_control's 10 (each block's end waits for its ticket with 15 wavefronts at
    // the barrier: block slots, not atomics, are what it costs); kept for that comparison.
    const double FUSED_MAX = h->hjb_tail_fuse == 2 ? 1e18 : (double)TAIL_FLAT_MAX;
    const bool can_fuse = ring != nullptr && !sh && h->hjb_tail_fuse != 0;
    SA.tail_counter = h->hjb_tail;
    SA.tail_rec = (can_fuse && h->hjb_tail_fast != 0) ? h->red_partial + 2 * (size_t)h->red_blocks : nullptr;
    SA.t_steps = h->hjb_tsteps;
    SA.att_log = h->hjb_log;
    SA.t_cap = T_CAP;
    SA.log_cap = LOG_CAP;
    SA.mailbox = (volatile double*)mb_dev;
    SA.active_set = active_set ? 1 : 0;
    SA.rslots = R.slots;
    SA.rslices = R.slices;

    const size_t lo = (size_t)own_lo * S.Nx, hi = (size_t)own_hi * S.Nx, hn = sh ? (size_t)sh->halo * S.Nx : 0;
    const bool halos = sh && (sh->has_up || sh->has_dn);

    // profiling: events around the stage pair and around k_out of every attempt; which of them were real / accepted is
    // known only afterwards (att_log), so the sums are formed after the solve
    // (an event record between two kernels costs 3-6 us of stream time: every 8th attempt is timed, the averages are the same)
    const bool prof_on = h->profiling && (h->prof_mode & 1);
    const int PROF_EVERY = 16;
    std::vector<hipEvent_t>& ev = h->hjb_events;
    size_t ev_used = 0;
    auto event = [&](bool record = true) -> hipEvent_t {       // (record = false: the group's slot is taken, nothing is enqueued)
        if (ev_used == ev.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            ev.push_back(e);
        }
        hipEvent_t e = ev[ev_used++];
        if (record) (void)hipEventRecord(e, s);
        return e;
    };
    // LEAD launches (hjb_lead: k_stageV<.., LEAD>): sparse one-GPU ring solves with the far-field substitution and the tail's
    // records.  A LEAD launch may follow any other launch (p_prev_lead tells it which kind it follows); an un-fused launch
    // (a list beyond FUSED_MAX, as far as the host knows) follows a LEAD launch only behind a drain that makes sure the
    // attempt's controller block is in memory.
    const bool lead_live = h->hjb_lead != 0 && can_fuse && ring && active_set && SA.undef != nullptr && SA.tail_rec != nullptr;
    SA.p_ctl[0] = ctlb[0];
    SA.p_ctl[1] = ctlb[1];
    SA.p_flags = active_set ? h->hjb_flags : nullptr;
    int n_lead = 0;
    SA.p_stamps = nullptr;
#ifdef OC_DIAG_BUILD
    const char* lead_stamp_file = lead_live ? getenv("OC_LEAD_STAMPS") : nullptr;
    if (lead_stamp_file) {
        OC_HIP_CHECK(hipMalloc((void**)&SA.p_stamps, sizeof(unsigned long long) * 16 * 4096));
        OC_HIP_CHECK(hipMemsetAsync(SA.p_stamps, 0, sizeof(unsigned long long) * 16 * 4096, s));
    }
#endif
    std::vector<char> att_lead;                     // per attempt enqueued: was it a LEAD launch
    int enq = 0;
    int grid_floor = 8;                             // blocks a sparse attempt's launch has at least (raised when one was too small)
    std::vector<int> ev_attempt;                   // attempt index of every group of 4 events
    std::vector<char> ev_list;                     // ... and whether the group's second interval holds a k_list launch
    auto enqueue_attempt = [&]() -> int {
        const bool prof = prof_on && (enq % PROF_EVERY == 0);
        if (prof) ev_attempt.push_back(enq);
        if (prof) ev_list.push_back(0);
        if (prof) event();
        HjbCtl<T>* const ctl_cur = ctlb[enq & 1];
        HjbCtl<T>* const ctl_nxt = ctlb[(enq + 1) & 1];
        SA.ctl = ctl_cur;
        SA.ctl_out = ctl_nxt;
        // (row-block solves: unchecked -- every rank must enqueue the same attempts, their collectives are matched by count; a
        // drain-and-re-enqueue on one rank alone would leave the others waiting)
        SA.att_expected = sh ? -1 : enq;
        OA.ctl = ctl_nxt;                               // (k_out / the list builder act on the controller's decision)
        if (active_set) {
            SA.flags_cur = h->hjb_flags + (size_t)(enq & 1) * nb3;
            SA.flags_prev = enq > 0 ? h->hjb_flags + (size_t)((enq - 1) & 1) * nb3 : nullptr;
            SA.work = enq > 0 ? work_list(enq) : nullptr;   // (the first attempt looks at every tile: no list yet)
            SA.flags_next = h->hjb_flags + (size_t)((enq + 1) & 1) * nb3;
            SA.work_next = work_list(enq + 1);
            OA.flags = SA.flags_cur;
            OA.ctl_rw = ctl_nxt;
            OA.flags_next = SA.flags_next;
            OA.work = SA.work_next;
            OA.parts_rw = h->red_partial;
        }
        // blocks that will take part, as far as the host knows (the device is a few attempts ahead of what it posted)
        const double seen = active_set ? (enq > 0 ? (double)mb[1] : (double)nb3) : (double)nb3;
        // (nothing posted yet -- the attempts enqueued behind the solve's first sweep: the list such a sweep leaves is short as a
        // rule; fused, on a grid that holds any list -- the tail draws its tickets in two levels if it must)
        const bool unknown = active_set && enq > 0 && seen >= 1e8;
        // (the FIRST attempt of an active-set solve sweeps every tile and needs the list built behind it: the two-level fused
        // tail costs less than the two launches -- k_control and the list builder's own -- it replaces)
        bool fused = can_fuse && (seen <= FUSED_MAX || unknown || (active_set && enq == 0));
        const bool prev_lead = enq > 0 && (size_t)(enq - 1) < att_lead.size() && att_lead[enq - 1] != 0;
        if ((size_t)enq >= att_lead.size()) att_lead.resize((size_t)enq + 1, 0);
        att_lead[enq] = 0;
        if (lead_live && !fused && prev_lead) {
            // the un-fused launches need the attempt's controller block in memory, which a LEAD launch leaves there only where
            // the attempt before it ended with a tail: drain and look (once per solve, if ever)
            OC_HIP_CHECK(hipStreamSynchronize(s));
            HjbCtl<T> cur;
            OC_HIP_CHECK(hipMemcpy(&cur, ctl_cur, sizeof(cur), hipMemcpyDeviceToHost));
            if (cur.ticks != enq && cur.done == 0) fused = true;       // (not there: this attempt as a LEAD launch too)
        }
        if (ring && fused) {
            // a sparse attempt's grid follows the list: the blocks beyond it would be dispatched -- 16 wavefronts and 40 KB of
            // LDS each -- only to return at once (1957 of them per attempt at 2048 x 2048 for ~25 listed tiles).  Any size is
            // correct (a block takes every gridDim.x-th item); this one leaves room for the list to grow until the host sees it.
            dim3 gridf = grid3;
            if (active_set && enq > 0 && h->hjb_grid_fit != 0 && !unknown) {
                const int want = std::max((int)(seen * 1.5) + h->hjb_grid_margin, grid_floor);
                gridf = dim3(std::min(8u * (unsigned)((std::max(want, 8) + 7) / 8), grid3.x));
            }
            if (lead_live) {
                ++n_lead;
                att_lead[enq] = 1;
                SA.p_prev_lead = prev_lead ? 1 : 0;
                SA.p_a0 = enq;
                hipLaunchKernelGGL((k_stageV<T, 6, 1, true, OC_ST_RWV, OC_ST_RH, true, true, false, true>), gridf, dim3(OC_ST_RWV * OC_ST_RH), 0, s, SA);
            } else {
                hipLaunchKernelGGL((k_stageV<T, 6, 1, true, OC_ST_RWV, OC_ST_RH, true, true>), gridf, dim3(OC_ST_RWV * OC_ST_RH), 0, s, SA);
            }
        } else if (ring) {
            hipLaunchKernelGGL((k_stageV<T, 6, 1, true, OC_ST_RWV, OC_ST_RH, true>), grid3, dim3(OC_ST_RWV * OC_ST_RH), 0, s, SA);
        } else if (fuse == 0) {
            hipLaunchKernelGGL((k_stageV<T, 6, 1, true, OC_ST_RWV, OC_ST_RH>), grid3, dim3(OC_ST_RWV * OC_ST_RH), 0, s, SA);
        } else {
            hipLaunchKernelGGL((k_stageV<T, 3, 1, false, 32, 32>), grid3, dim3(1024), 0, s, SA);
            hipLaunchKernelGGL((k_stageV<T, 3, 4, true, 32, 32>), grid3, dim3(1024), 0, s, SA);
        }
        if (prof) event();
        if (fused) {
            // (controller and list builder ran inside the stage launch)
        } else if (sh) {
            if (active_set) hipLaunchKernelGGL(k_reduce_local<T>, dim3(1), dim3(1024), 0, s, (const double*)h->red_partial, nb3, ctl_cur, sh->red_local);
            else hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(1024), 0, s, h->red_partial, nb3, sh->red_local);
            if (sh->comm_slot >= 0) {       // RCCL from here, on the launch stream (comm.hip)
                int rcc = oc_comm_allgather_f64_oop(h, sh->comm_slot, sh->red_local, sh->red_all, 2, s);
                if (rcc) return rcc;
            } else if (sh->allgather_red(sh->user) != 0) {
                oc_set_error("hjb_solve_rows: the allgather_red callback failed");
                return OC_ERR_STATE;
            }
            hipLaunchKernelGGL(k_control<T>, dim3(1), dim3(256), 0, s, (const HjbCtl<T>*)ctl_cur, ctl_nxt,
                               (const double*)sh->red_all, sh->world, h->hjb_tsteps, T_CAP, h->hjb_log, LOG_CAP,
                               (volatile double*)mb_dev, active_set ? 2 : 0, R.slots, R.slices, R.cap);
        } else {
            hipLaunchKernelGGL(k_control<T>, dim3(1), dim3(256), 0, s, (const HjbCtl<T>*)ctl_cur, ctl_nxt,
                               (const double*)h->red_partial, nb3, h->hjb_tsteps, T_CAP, h->hjb_log, LOG_CAP,
                               (volatile double*)mb_dev, active_set ? 1 : 0, R.slots, R.slices, R.cap);
        }
        if (halos) {
            const dim3 gh((unsigned)((4 * hn + 255) / 256));
            hipLaunchKernelGGL((k_halo_stage<T, true>), gh, dim3(256), 0, s, (const HjbCtl<T>*)ctl_nxt, Y[0], Y[1], KX[0], KX[1],
                               (T*)sh->send_up, (T*)sh->send_dn, lo, hi, hn, sh->has_up, sh->has_dn);
            if (sh->comm_slot >= 0) {
                int rcc = oc_comm_exchange_halo(h, sh->comm_slot, sh->send_up, sh->recv_up, sh->send_dn, sh->recv_dn,
                                                2 * hn * sizeof(T), sh->has_up, sh->has_dn, s);
                if (rcc) return rcc;
            } else if (sh->exchange_halo(sh->user) != 0) {
                oc_set_error("hjb_solve_rows: the exchange_halo callback failed");
                return OC_ERR_STATE;
            }
            hipLaunchKernelGGL((k_halo_stage<T, false>), gh, dim3(256), 0, s, (const HjbCtl<T>*)ctl_nxt, Y[0], Y[1], KX[0], KX[1],
                               (T*)sh->recv_up, (T*)sh->recv_dn, lo, hi, hn, sh->has_up, sh->has_dn);
        }
        // (the group's second interval -- k_out, or the list builder's own launch -- is bracketed only where such a launch follows)
        const bool second = !ring || (active_set && !fused && seen != (double)nb3 + 0.5);
        if (prof) event(second);
        if (!ring) {
            launch_out<T, TO, true>(h->out_tile, S.Nx, S.Ny, s, OA);
        } else if (active_set && !fused && seen != (double)nb3 + 0.5) {
            if (prof) ev_list.back() = 1;
            // no slice writer beside which the list could be built: its own (one-block) launch -- until the device has
            // posted dense_mode (every tile active for the rest of the solve: C5's re-solves under the crowd's density,
            // 2 200 launches of ~5 us per run that did nothing)
            hipLaunchKernelGGL(k_list<T>, dim3(1), dim3(256), 0, s, ctl_nxt, (const int32_t*)OA.flags, OA.flags_next, OA.work,
                               OA.parts_rw, (S.Nx + tx - 1) / tx, (S.Ny + ty - 1) / ty, (volatile double*)mb_dev, SA.undef);
        }
        if (prof) event(second);
        ++enq;
        return OC_OK;
    };
    const int BURST = sh ? 8 : 4, LAG = 8;     // (the fused tail posts its tick count every 4th attempt)
    bool done = false;
    // ---- the persistent launch (hjb_persist): the attempts of a ring solve inside ONE launch of a co-resident grid
    // (k_stageV<.., PERSIST>), attempt after attempt, instead of one launch each.  The first attempt of an active-set solve
    // (every tile is looked at once, no list yet) goes through the launch chain, the persistent launch takes over from
    // attempt 1 and hands back (dense mode, a list that outgrew it, the deadline of a grid that turned out not to be
    // co-resident) with the controller between two attempts: the chain below continues there.
    // MEASURED (round 4, C4 2048 x 2048, ~25 listed tiles per attempt, profiles/r04_hjb_persist_stamps.txt): an attempt of the
    // persistent launch takes 18.4 us -- controller block re-read 1.1, one tile's trip 7.5, release + ticket 0.9, tail 7.7,
    // publication 0.5, the waiting blocks' wake-up 0.8 -- against 17.5 us in the launch chain: on this chip back-to-back
    // launches of one stream overlap their dispatch with the previous launch's end (~0.5 us between two attempts), and the
    // in-kernel barrier (write-back of the L2 for the other XCDs, ticket round trip, publication, poll, invalidation) costs
    // more than that.  It pays only where every tile has its own block and no list is kept: grids of a few dozen tiles
    // (room_test 200 x 200, 433 attempts: 6.85 against 7.10 ms per solve).  hjb_persist = 1 (default): those grids only;
    // 2: wherever the launch can run; 0: never.
    stats.persist_exit = 0;
    double persist_ms = 0.0;            // profiling: the persistent launch's time and the attempts it ran
    int persist_n = 0;
    if (can_fuse && h->hjb_persist != 0 && !(prof_on && (h->prof_mode & 16))) {
        const void* kfn = (const void*)k_stageV<T, 6, 1, true, OC_ST_RWV, OC_ST_RH, true, true, true>;
        int& cap = sizeof(T) == 4 ? h->persist_cap_f32 : h->persist_cap_f64;
        if (cap < 0) {
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, OC_ST_RWV * OC_ST_RH, 0) != hipSuccess) per_cu = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) cus = 0;
            cap = std::max(per_cu, 0) * std::max(cus, 0);
        }
        const int want = h->hjb_persist_blocks > 0 ? h->hjb_persist_blocks : 128;  // (half the CUs: two keys solved concurrently both fit)
        const int G = oc_persist_reserve(std::min(want, active_set ? nb3 : 8 * ((nb3 + 7) / 8)), cap);
        // (without the active set every attempt computes every tile: worth it while a block carries at most two)
        const bool p_ok = h->hjb_persist == 2 ? (active_set || nb3 <= 2 * G) : (!active_set && nb3 <= G);
        if (G >= 8 && p_ok) {
            if (active_set && (rc = enqueue_attempt())) {
                oc_persist_release(G);
                return rc;
            }
            int32_t* const psync = h->hjb_tail + TAIL_COUNTER_BYTES / 4;     // (two words behind the ticket counters)
            int32_t* const psync_host = (int32_t*)((char*)h->hjb_ctl_host + 2048);
            psync_host[0] = enq;
            psync_host[1] = 0;
            OC_HIP_CHECK(hipMemcpyAsync(psync, psync_host, 8, hipMemcpyHostToDevice, s));
            SA.p_ctl[0] = ctlb[0];
            SA.p_ctl[1] = ctlb[1];
            SA.p_flags = active_set ? h->hjb_flags : nullptr;
            SA.p_sync = psync;
            SA.p_a0 = enq;
            SA.p_amax = LOG_CAP - 2 * BURST;
            SA.p_max_items = 2 * G;
            SA.p_wait_ticks = (int32_t)std::min<int64_t>((int64_t)h->hjb_persist_wait_us * 100, 2000000000);
            SA.p_stamps = nullptr;
#ifdef OC_DIAG_BUILD
            const char* stamp_file = getenv("OC_PERSIST_STAMPS");
            const size_t stamp_bytes = sizeof(unsigned long long) * 16 * 4096;
            if (stamp_file) {
                OC_HIP_CHECK(hipMalloc((void**)&SA.p_stamps, stamp_bytes));
                OC_HIP_CHECK(hipMemsetAsync(SA.p_stamps, 0, stamp_bytes, s));
                SA.p_amax = std::min(SA.p_amax, 4096);
            }
#endif
            hipEvent_t pe0 = nullptr, pe1 = nullptr;
            if (prof_on) pe0 = event();
            hipLaunchKernelGGL((k_stageV<T, 6, 1, true, OC_ST_RWV, OC_ST_RH, true, true, true>), dim3(G), dim3(OC_ST_RWV * OC_ST_RH), 0, s, SA);
            if (prof_on) pe1 = event();
            HjbCtl<T> hc2[2];
            OC_HIP_CHECK(hipMemcpyAsync(&hc2[0], ctlb[0], sizeof(hc), hipMemcpyDeviceToHost, s));
            OC_HIP_CHECK(hipMemcpyAsync(&hc2[1], ctlb[1], sizeof(hc), hipMemcpyDeviceToHost, s));
            OC_HIP_CHECK(hipMemcpyAsync(psync_host, psync, 8, hipMemcpyDeviceToHost, s));
            const hipError_t pse = hipStreamSynchronize(s);
            oc_persist_release(G);
            OC_HIP_CHECK(pse);
#ifdef OC_DIAG_BUILD
            if (SA.p_stamps) {
                std::vector<unsigned long long> st(16 * 4096);
                OC_HIP_CHECK(hipMemcpy(st.data(), SA.p_stamps, stamp_bytes, hipMemcpyDeviceToHost));
                if (FILE* f = fopen(stamp_file, "wb")) {
                    fwrite(st.data(), 1, stamp_bytes, f);
                    fclose(f);
                }
                (void)hipFree(SA.p_stamps);
                SA.p_stamps = nullptr;
            }
#endif
            const HjbCtl<T>& last = hc2[0].ticks >= hc2[1].ticks ? hc2[0] : hc2[1];
            const int a0 = enq;
            enq = std::max((int)last.ticks, a0);        // attempts whose controller has run: the chain continues with this one
            done = last.done != 0;
            stats.persist_attempts = enq - a0;
            stats.persist_blocks = G;
            stats.persist_exit = done ? 1 : (psync_host[1] != 0 ? 3 : 2);
            // (a launch that was given up leaves tickets behind)
            OC_HIP_CHECK(hipMemsetAsync(h->hjb_tail, 0, TAIL_COUNTER_BYTES, s));
            if (prof_on && enq > a0) {
                float e = 0.f;
                (void)hipEventElapsedTime(&e, pe0, pe1);
                persist_ms = e;
                persist_n = enq - a0;
            }
            ev_used = 0;        // (the chain's bookkeeping expects groups of four events per profiled attempt)
            ev_attempt.clear();
            ev_list.clear();
        } else if (G > 0) {
            oc_persist_release(G);
        }
    }
    while (!done) {
        for (int b = 0; b < BURST; ++b)
            if ((rc = enqueue_attempt())) return rc;
        if (enq > LOG_CAP - 2 * BURST) {
            (void)hipStreamSynchronize(s);
            HjbCtl<T> two[2];
            int32_t tc[4] = {0, 0, 0, 0}, dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            (void)hipMemcpy(&two[0], ctlb[0], sizeof(hc), hipMemcpyDeviceToHost);
            (void)hipMemcpy(&two[1], ctlb[1], sizeof(hc), hipMemcpyDeviceToHost);
            (void)hipMemcpy(tc, h->hjb_tail, sizeof(tc), hipMemcpyDeviceToHost);
            (void)hipMemcpy(dbg, h->hjb_tail + TAIL_COUNTER_BYTES / 4, sizeof(dbg), hipMemcpyDeviceToHost);
            oc_set_error("RK45: more than %d attempts (early exits: done@0 %d, not-mine %d [fresh ok %d, fresh bad %d]; ticks %d / %d, attempts %d / %d, done %d / %d, "
                         "t %g / %g, listed %d / %d; tickets %d; mailbox %g %g %g)", LOG_CAP, dbg[4], dbg[5], dbg[6], dbg[7], two[0].ticks, two[1].ticks,
                         two[0].n_attempt, two[1].n_attempt, two[0].done, two[1].done, two[0].t, two[1].t, two[0].work_count,
                         two[1].work_count, tc[0], (double)mb[0], (double)mb[1], (double)mb[2]);
            return OC_ERR_SOLVER;
        }
        if (sh) {
            OC_HIP_CHECK(hipStreamSynchronize(s));
        } else {
            // let the device catch up to within LAG attempts of what is enqueued (or finish)
            double w_seen = -1.0;
            auto t_seen = std::chrono::steady_clock::now();
            for (long spin = 0;; ++spin) {
                const double w = mb[0];
                if (w >= 4294967296.0 || w >= (double)(enq - LAG) || mb[2] != 0.0) break;
                if ((spin & 4095) == 4095) {            // drained (mailbox not visible?) or failed: never spin on a dead stream
                    const hipError_t q = hipStreamQuery(s);
                    if (q == hipSuccess) break;
                    if (q != hipErrorNotReady) {
                        oc_set_error("hjb_solve: %s", hipGetErrorString(q));
                        return OC_ERR_HIP;
                    }
                    // no tick for 10 s while the stream is busy: a kernel hangs.  One blocking wait (it returns when the
                    // runtime gives the stream up), an error, no retry.
                    const auto now = std::chrono::steady_clock::now();
                    if (w != w_seen) {
                        w_seen = w;
                        t_seen = now;
                    } else if (std::chrono::duration<double>(now - t_seen).count() > 10.0) {
                        const hipError_t e = hipStreamSynchronize(s);
                        oc_set_error("hjb_solve: the controller did not advance for 10 s at attempt %d of %d enqueued (%s)",
                                     (int)w, enq, hipGetErrorString(e));
                        return OC_ERR_HIP;
                    }
                }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (mb[2] != 0.0) {
            // a sparse attempt's launch was too small for its list, or a launch found a controller block that was not its own
            // (k_stageV: att_expected): it and everything enqueued behind it did nothing.  Drain, take the attempt number
            // from the controller blocks, go on (with grids that fit).
            OC_HIP_CHECK(hipStreamSynchronize(s));
            HjbCtl<T> two[2];
            OC_HIP_CHECK(hipMemcpy(&two[0], ctlb[0], sizeof(hc), hipMemcpyDeviceToHost));
            OC_HIP_CHECK(hipMemcpy(&two[1], ctlb[1], sizeof(hc), hipMemcpyDeviceToHost));
            enq = std::max(two[0].ticks, two[1].ticks);
            if (mb[1] < 1e8) grid_floor = std::max(grid_floor, (int)mb[1]);
            ++stats.grid_resyncs;
            mb[2] = 0.0;
            __atomic_thread_fence(__ATOMIC_SEQ_CST);
            OC_HIP_CHECK(hipMemsetAsync(h->hjb_tail, 0, TAIL_COUNTER_BYTES, s));
            continue;
        }
        done = mb[0] >= 4294967296.0;
        if (!done && hipStreamQuery(s) == hipSuccess && mb[0] < (double)enq) {
            // the stream drained but the mailbox did not move: mapped memory is not visible -> read the device copy
            OC_HIP_CHECK(hipMemcpy(&hc, ctlb[enq & 1], sizeof(hc), hipMemcpyDeviceToHost));
            done = hc.done != 0;
            if (!done && lead_live) {       // (the final block is where the launch that found the end left it)
                OC_HIP_CHECK(hipMemcpy(&hc, ctlb[(enq + 1) & 1], sizeof(hc), hipMemcpyDeviceToHost));
                done = hc.done != 0;
            }
        }
    }
    // (the solve's last copies go through ITS stream: hipMemcpy works on the null stream, which serialises with whatever a
    // concurrent solve of another key -- Engine.solve_keys -- has enqueued there)
    // What the last controller left, the attempt log and the step log in ONE batch into the pinned block, one wait (as three
    // copies into pageable memory they were three host round trips, ~40 us of a 1.8 ms solve).
    static_assert(2 * sizeof(HjbCtl<T>) <= 4096, "two controller blocks in the staging block's second page");
    char* const pin = (char*)h->hjb_ctl_host;
    const int log_pre = std::min(std::min(enq, LOG_CAP), OC_CTL_HOST_LOG), ts_pre = std::min(T_CAP, OC_CTL_HOST_TS);
    OC_HIP_CHECK(hipMemcpyAsync(pin + 4096, ctlb[0], sizeof(hc), hipMemcpyDeviceToHost, s));
    OC_HIP_CHECK(hipMemcpyAsync(pin + 4096 + sizeof(hc), ctlb[1], sizeof(hc), hipMemcpyDeviceToHost, s));
    if (log_pre > 0) OC_HIP_CHECK(hipMemcpyAsync(pin + 8192, h->hjb_log, sizeof(int32_t) * log_pre, hipMemcpyDeviceToHost, s));
    if (t_steps && t_steps_cap > 0)
        OC_HIP_CHECK(hipMemcpyAsync(pin + 8192 + 4 * OC_CTL_HOST_LOG, h->hjb_tsteps, sizeof(double) * ts_pre, hipMemcpyDeviceToHost, s));
    OC_HIP_CHECK(hipStreamSynchronize(s));
    memcpy(&hc, pin + 4096 + (size_t)(enq & 1) * sizeof(hc), sizeof(hc));
    if (lead_live && hc.done == 0) {
        // LEAD: the launch that found the end wrote the final block in ITS place; the launch behind it copies it into the
        // other one -- if there was a launch behind it
        memcpy(&hc, pin + 4096 + (size_t)((enq + 1) & 1) * sizeof(hc), sizeof(hc));
        if (hc.done == 0) {
            oc_set_error("hjb_solve: the solve's final controller block is missing (ticks %d, %d attempts enqueued)", hc.ticks, enq);
            return OC_ERR_STATE;
        }
    }
    if (hc.status != 0) {
        if (hc.status == OC_ERR_SOLVER - 200) {
            oc_set_error("hjb_solve_ring: more than %d accepted steps (t = %g reached): the ring is too small", R.cap, hc.t);
            return OC_ERR_CAPACITY;
        }
        if (hc.status == OC_ERR_SOLVER) oc_set_error("RK45: step size too small at t=%g", hc.t);
        else oc_set_error("RK45: NaN error norm at t=%g (state overflow?)", hc.t);
        return OC_ERR_SOLVER;
    }
    if (ring) {
        if (n_keep > 0) {               // the audit output: phi at the kept t_eval points, from the ring
            int rc2 = ring_emit<T, TO>(h, R, hc.n_accept, 0, nt, nullptr, nullptr, phi_keep, keep_slot_dev, s);
            if (rc2) return rc2;
            OC_HIP_CHECK(hipStreamSynchronize(s));
        }
    }
    const int n_real = hc.n_attempt;
#ifdef OC_DIAG_BUILD
    if (lead_stamp_file && SA.p_stamps) {
        std::vector<unsigned long long> st(16 * 4096);
        OC_HIP_CHECK(hipMemcpy(st.data(), SA.p_stamps, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost));
        if (FILE* f = fopen(lead_stamp_file, "wb")) {
            fwrite(st.data(), 8, st.size(), f);
            fclose(f);
        }
        (void)hipFree(SA.p_stamps);
        SA.p_stamps = nullptr;
    }
#endif
    stats.lead_attempts = n_lead;
    // (the controller's ticks, not the attempts enqueued: with the fused tail an attempt past the end of the solve returns
    // before anything is logged, and its slot holds an older solve's word)
    std::vector<int32_t> log((size_t)std::max(std::min(std::min(enq, (int)hc.ticks), LOG_CAP), 0));
    if ((int)log.size() <= log_pre) {
        if (!log.empty()) memcpy(log.data(), pin + 8192, sizeof(int32_t) * log.size());
    } else {
        OC_HIP_CHECK(hipMemcpyAsync(log.data(), h->hjb_log, sizeof(int32_t) * log.size(), hipMemcpyDeviceToHost, s));
        OC_HIP_CHECK(hipStreamSynchronize(s));
    }
    for (size_t a = 0; a < log.size(); ++a)
        if (log[a] & 2) stats.n_slices_written += log[a] >> 8;
    if (prof_on) {
        oc_timer& ts = h->timers["hjb_stage"];
        oc_timer& to = h->timers["hjb_out"];
        if (persist_n > 0) {            // the persistent launch's attempts: one "stage launch" each, priced like the others
            const double share = active_set && n_real > 0 ? std::min(1.0, hc.act_sum / ((double)n_real * nb3)) : 1.0;
            ts.add_ms += persist_ms;
            ts.add_n += persist_n;
            ts.bytes += 10.0 * sizeof(T) * (double)n * share * persist_n;
            oc_timer& tp = h->timers["hjb_persist"];
            tp.add_ms += persist_ms;
            tp.add_n += persist_n;
        }
        for (size_t g = 0; g < ev_attempt.size() && 4 * g + 3 < ev_used; ++g) {
            const int a = ev_attempt[g];
            if ((size_t)a >= log.size()) break;
            float e1 = 0.f, e2 = 0.f;
            (void)hipEventElapsedTime(&e1, ev[4 * g], ev[4 * g + 1]);
            // (the second interval was bracketed only where a launch followed: k_out, or the list builder's own)
            if (!ring || (g < ev_list.size() && ev_list[g])) (void)hipEventElapsedTime(&e2, ev[4 * g + 2], ev[4 * g + 3]);
            // (active set: the state arrays are only moved on the non-trivial tiles -- priced with the solve's mean share
            // of tiles that had an error contribution; the direction field is written in full either way)
            const double share = active_set && n_real > 0 ? std::min(1.0, hc.act_sum / ((double)n_real * nb3)) : 1.0;
            if (log[a] & 1) {           // a real attempt: 8 array passes in one launch (two launches: 6 + 11)
                ts.add_ms += e1;
                ts.add_n += 1;
                ts.bytes += (ring ? 10.0 : (fuse == 0 ? 8.0 : 17.0)) * sizeof(T) * (double)n * share;    // ring: 3 read, 7 written
            }
            if (ring) {                 // (the second interval is the list builder's launch, where there was one)
                if ((log[a] & 1) && g < ev_list.size() && ev_list[g]) {
                    oc_timer& tl = h->timers["hjb_list"];
                    tl.add_ms += e2;
                    tl.add_n += 1;
                }
            } else if (log[a] & 2) {           // accepted: k_out read y, K0, Q1 .. Q3 once, wrote 2 components per slice
                const int nv = log[a] >> 8;
                to.add_ms += e2;
                to.add_n += 1;
                to.bytes += sizeof(T) * 5.0 * (double)n * share + sizeof(TO) * 2.0 * nv * (double)(S.Ny - 2) * (S.Nx - 2);
            }
        }
    }
    if (t_steps && t_steps_cap > 0) {
        t_steps[0] = t0;
        const int m = std::min(hc.n_tsteps, t_steps_cap - 1);
        if (m > 0 && m <= ts_pre) memcpy(t_steps + 1, pin + 8192 + 4 * OC_CTL_HOST_LOG, sizeof(double) * m);
        else if (m > 0) OC_HIP_CHECK(hipMemcpy(t_steps + 1, h->hjb_tsteps, sizeof(double) * m, hipMemcpyDeviceToHost));
    }
    stats.h_abs0 = hc.h_abs0;
    stats.nfev = S.nfev + 6 * n_real;
    stats.n_attempt = n_real;
    stats.n_accept = hc.n_accept;
    stats.n_rescale = hc.n_rescale;
    stats.h_min = hc.h_min;
    stats.h_max = hc.h_max;
    stats.final_exponent = hc.expo;
    if (st) *st = stats;
    return OC_OK;
}

template <typename T, typename TO = T>
int hjb_solve_t(oc_handle* h, const void* coef_v, double Tfinal, int32_t nt, void* vx_out, void* vy_out,
                const int32_t* keep_idx, int32_t n_keep, double* phi_keep, double* t_steps, int32_t t_steps_cap,
                oc_hjb_stats* st, hipStream_t s, const oc_hjb_shard* sh = nullptr, const oc_hjb_ring* ring = nullptr) {
    const oc_config& c = h->cfg;
    Solver<T> S;
    S.h = h;
    S.s = s;
    S.sh = sh;
    // row-block solve: all arrays hold Ny_local rows (owned rows + halo rows); the kernels treat them as a grid of
    // that height -- the mirror condition they apply at a halo edge only spoils rows that are never used (see
    // oc_hip.h, oc_hjb_solve_rows); error norms run over the owned rows and are completed by the collective
    S.Ny = sh ? sh->Ny_local : c.Ny;
    S.Nx = c.Nx;
    S.n = (int64_t)S.Ny * S.Nx;
    OC_REQUIRE(S.n * (int64_t)sizeof(T) < ((int64_t)1 << 32), "grid too large: the stage kernel addresses an array by 32-bit byte offsets");
    const int own_lo = sh ? sh->own_lo : 0, own_hi = sh ? sh->own_hi : S.Ny;
    const int64_t own_off = (int64_t)own_lo * S.Nx, n_own = (int64_t)(own_hi - own_lo) * S.Nx;
    const double n_rms = sh ? (double)sh->Ny_global * S.Nx : (double)S.n;     // cells of the whole grid
    S.coef = (const T*)coef_v;
    S.a_lap = (T)(-0.5 * c.sigma * c.sigma / (c.dx * c.dy));
    int rc = S.alloc();
    if (rc) return rc;
    const int64_t n = S.n;
    const double rtol = c.rtol, atol0 = c.atol;
    const int RB = 1024;
    const bool use_fused = (h->hjb_variant != 1) && (c.Nx % 4 == 0);   // vector kernels need 16-byte rows
    if (h->hjb_variant == 0 && use_fused)        // the default: three stages per launch, controller on the device
        return hjb_solve_dev<T, TO>(h, coef_v, Tfinal, nt, vx_out, vy_out, keep_idx, n_keep, phi_keep, t_steps, t_steps_cap,
                                    st, s, sh, ring);
    if (ring) {
        oc_set_error("the interpolant ring needs the product path (Nx %% 4 == 0, hjb variant 0)");
        return OC_ERR_ARG;
    }
    if (sizeof(TO) != sizeof(T)) {
        oc_set_error("the float64-state / float32-field mode needs the product path (Nx %% 4 == 0, hjb variant 0)");
        return OC_ERR_ARG;
    }
    if (sh) {
        oc_set_error("hjb_solve_rows needs the fused kernels (Nx %% 4 == 0, hjb variant 0)");
        return OC_ERR_ARG;
    }

    oc_hjb_stats stats;
    memset(&stats, 0, sizeof(stats));
    stats.h_min = 1e300;
    if (nt <= 1) {   // linspace(T,0,nt) has no slice that the fill loop would use (optimals.py:200)
        if (st) *st = stats;
        return OC_OK;
    }
    // t_eval = np.linspace(T, 0, nt)
    std::vector<double> te(nt);
    {
        double step = (0.0 - Tfinal) / (double)(nt - 1);
        for (int k = 0; k < nt; ++k) te[k] = (double)k * step + Tfinal;
        te[nt - 1] = 0.0;
    }
    std::vector<int> keep_slot(nt, -1);
    for (int q = 0; q < n_keep; ++q) {
        OC_REQUIRE(keep_idx[q] >= 0 && keep_idx[q] < nt, "keep index out of range");
        keep_slot[keep_idx[q]] = q;
    }

    int32_t* keep_slot_dev = nullptr;   // device copy of keep_slot for the fused writer (this call only)
    struct Guard {
        int32_t*& p;
        ~Guard() { if (p) (void)hipFree(p); }
    } guard{keep_slot_dev};
    int expo = 0;                       // state = phi * 2^-expo
    auto scale_of = [&]() { return std::ldexp(1.0, -expo); };

    hipLaunchKernelGGL(k_fill<T>, S.g1(), dim3(256), 0, s, S.y, (T)1, n);
    const double t0 = Tfinal, tf = 0.0, direction = -1.0;
    double t = t0;
    S.rhs(S.y, S.K[6]);                 // f0 lives in K[6] (FSAL slot)
    // ---- select_initial_step (scipy common.py:68-134), order = 4
    double h_abs;
    {
        double interval = std::fabs(tf - t0);
        double s0, s1, s2;
        hipLaunchKernelGGL((k_norm_scaled<T, 0>), dim3(RB), dim3(256), 0, s, S.y + own_off, (const T*)nullptr,
                           S.y + own_off, (T)atol0, (T)rtol, n_own, h->red_partial);
        if ((rc = S.reduce(&s0, nullptr))) return rc;
        hipLaunchKernelGGL((k_norm_scaled<T, 0>), dim3(RB), dim3(256), 0, s, S.K[6] + own_off, (const T*)nullptr,
                           S.y + own_off, (T)atol0, (T)rtol, n_own, h->red_partial);
        if ((rc = S.reduce(&s1, nullptr))) return rc;
        double d0 = std::sqrt(s0) / std::sqrt(n_rms), d1 = std::sqrt(s1) / std::sqrt(n_rms);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        h0 = std::min(h0, interval);
        Lin8<T> L;
        L.n = 1;
        L.k[0] = S.K[6];
        L.c[0] = (T)(h0 * direction);
        hipLaunchKernelGGL(k_lincomb<T>, S.g1(), dim3(256), 0, s, S.y, L, S.ytmp, n);
        S.rhs(S.ytmp, S.K[0]);
        hipLaunchKernelGGL((k_norm_scaled<T, 1>), dim3(RB), dim3(256), 0, s, S.K[0] + own_off, S.K[6] + own_off,
                           S.y + own_off, (T)atol0, (T)rtol, n_own, h->red_partial);
        if ((rc = S.reduce(&s2, nullptr))) return rc;
        double d2 = std::sqrt(s2) / std::sqrt(n_rms) / h0;
        double h1;
        if (d1 <= 1e-15 && d2 <= 1e-15)
            h1 = std::max(1e-6, h0 * 1e-3);
        else
            h1 = std::pow(0.01 / std::max(d1, d2), 1.0 / 5.0);
        h_abs = std::min(std::min(100 * h0, h1), interval);
    }
    stats.h_abs0 = h_abs;
    int n_steps_rec = 0;
    if (t_steps && t_steps_cap > 0) t_steps[n_steps_rec++] = t;

    int k_next = 0;                     // next t_eval index to emit
    const size_t slice_elems = (size_t)S.Ny * S.Nx;     // padded slices: the reference's array is [1:-1, 1:-1]
    const double err_exp = -1.0 / 5.0;

    while (t != tf) {
        double min_step = 10 * std::fabs(std::nextafter(t, direction * INFINITY) - t);
        if (h_abs < min_step) h_abs = min_step;
        bool accepted = false, rejected = false;
        double t_new = t, hh = 0;
        // FSAL: K0 <- f (swap pointers K0 <-> K6)
        std::swap(S.K[0], S.K[6]);
        while (!accepted) {
            if (h_abs < min_step) {
                oc_set_error("RK45: step size too small at t=%g", t);
                return OC_ERR_SOLVER;
            }
            hh = h_abs * direction;
            t_new = t + hh;
            if (direction * (t_new - tf) > 0) t_new = tf;
            hh = t_new - t;
            h_abs = std::fabs(hh);
            ++stats.n_attempt;
            int nred = RB;
            double atol_s = atol0 * scale_of();
            oc_timer_begin(h, "hjb_stage", s);
            if (use_fused) {
                launch_stages_fused(S, hh, (T)atol_s, (T)rtol);
                S.nfev += 6;
                nred = S.fused_blocks();
            } else {
                for (int sg = 1; sg < 6; ++sg) {
                    Lin8<T> L;
                    L.n = sg;
                    for (int j = 0; j < sg; ++j) {
                        L.k[j] = S.K[j];
                        L.c[j] = (T)(RK_A[sg][j] * hh);
                    }
                    hipLaunchKernelGGL(k_lincomb<T>, S.g1(), dim3(256), 0, s, S.y, L, S.ytmp, n);
                    S.rhs(S.ytmp, S.K[sg]);
                }
                {
                    Lin8<T> L;
                    L.n = 6;
                    for (int j = 0; j < 6; ++j) {
                        L.k[j] = S.K[j];
                        L.c[j] = (T)(RK_B[j] * hh);
                    }
                    hipLaunchKernelGGL(k_lincomb<T>, S.g1(), dim3(256), 0, s, S.y, L, S.ynew, n);
                    S.rhs(S.ynew, S.K[6]);
                }
                Lin8<T> E;
                E.n = 7;
                for (int j = 0; j < 7; ++j) {
                    E.k[j] = S.K[j];
                    E.c[j] = (T)(RK_E[j] * hh);
                }
                hipLaunchKernelGGL(k_err_norm<T>, dim3(RB), dim3(256), 0, s, S.y, S.ynew, E, (T)atol_s, (T)rtol, n,
                                   h->red_partial);
                nred = RB;
            }
            // bytes this variant is designed to move per attempt: one launch per stage = 39 array passes (the
            // one-array-per-pass kernels move more; both are cross-check paths, the default is hjb_solve_dev)
            oc_timer_end(h, "hjb_stage", s, 39.0 * sizeof(T) * (double)n);
            double esum, ymax;
            if ((rc = S.reduce(&esum, &ymax, nred))) return rc;
            double err = std::sqrt(esum) / std::sqrt(n_rms);
            if (!(err == err)) {
                oc_set_error("RK45: NaN error norm at t=%g (state overflow?)", t);
                return OC_ERR_SOLVER;
            }
            if (err < 1) {
                double factor = (err == 0) ? 10.0 : std::min(10.0, 0.9 * std::pow(err, err_exp));
                if (rejected) factor = std::min(1.0, factor);
                h_abs *= factor;
                accepted = true;
                stats.h_min = std::min(stats.h_min, std::fabs(hh));
                stats.h_max = std::max(stats.h_max, std::fabs(hh));
                // ---- dense output for t_eval points in (t_new, t]  (scipy ivp.py:706-723)
                if (use_fused) {
                    // the fused writer of the default path (k_out), its controller block filled in by the host: the
                    // one-launch-per-stage variant stays bit-identical to the default one
                    if (!h->hjb_ctl) OC_HIP_CHECK(hipMalloc(&h->hjb_ctl, 8192));
                    if (!h->hjb_ctl_host) OC_HIP_CHECK(hipHostMalloc(&h->hjb_ctl_host, OC_CTL_HOST_BYTES, hipHostMallocDefault));
                    HjbCtl<T>& oc = *(HjbCtl<T>*)h->hjb_ctl_host;
                    memset(&oc, 0, sizeof(oc));
                    oc.accepted = 1;
                    oc.out_parity = 0;
                    oc.nt = nt;
                    oc.Tfinal = Tfinal;
                    oc.te_step = (0.0 - Tfinal) / (double)(nt - 1);
                    oc.out_t = t;
                    oc.out_hh = hh;
                    oc.out_unscale = std::ldexp(1.0, expo);
                    oc.out_lim_s = (T)(c.lim * scale_of());
                    oc.out_in_scale = (T)1;
                    oc.k_lo = k_next;
                    int nv = 0;
                    while (k_next < nt && te[k_next] >= t_new) {
                        nv += k_next >= 1 ? 1 : 0;
                        ++k_next;
                    }
                    oc.k_hi = k_next;
                    oc.n_hx = 0;                    // (the kernel computes the slice factors itself)
                    if (n_keep > 0 && !keep_slot_dev) {
                        OC_HIP_CHECK(hipMalloc((void**)&keep_slot_dev, sizeof(int32_t) * nt));
                        OC_HIP_CHECK(hipMemcpy(keep_slot_dev, keep_slot.data(), sizeof(int32_t) * nt, hipMemcpyHostToDevice));
                    }
                    OC_HIP_CHECK(hipMemcpyAsync(h->hjb_ctl, &oc, sizeof(oc), hipMemcpyHostToDevice, s));
                    OutArgs<T> OA;
                    memset(&OA, 0, sizeof(OA));
                    OA.ctl = (const HjbCtl<T>*)h->hjb_ctl;
                    OA.Y[0] = OA.Y[1] = S.y;
                    OA.KX[0] = S.K[0];
                    OA.KX[1] = S.K[6];
                    for (int j = 0; j < 5; ++j) OA.KM[j] = S.K[1 + j];
                    const int prow[6] = {0, 2, 3, 4, 5, 6};
                    for (int q = 0; q < 6; ++q)
                        for (int pp = 0; pp < 4; ++pp) OA.p[q][pp] = (T)RK_P[prow[q]][pp];
                    OA.vx_base = vx_out;
                    OA.vy_base = vy_out;
                    OA.slice_elems = slice_elems;
                    OA.keep_base = phi_keep;
                    OA.keep_slot = keep_slot_dev;
                    OA.lim = (T)c.lim;
                    OA.mu = (T)c.mu;
                    OA.inv2dx = (T)(1.0 / (2 * c.dx));
                    OA.inv2dy = (T)(1.0 / (2 * c.dy));
                    OA.Ny = S.Ny;
                    OA.Nx = S.Nx;
                    oc_timer_begin(h, "hjb_out", s);
                    launch_out<T, T, false>(h->out_tile, S.Nx, S.Ny, s, OA);
                    oc_timer_end(h, "hjb_out", s, sizeof(T) * (7.0 * (double)n + 2.0 * nv * (double)(S.Ny - 2) * (S.Nx - 2)));
                    stats.n_slices_written += nv;
                    // (the pinned block is rewritten only after the next attempt's reduction has been read: copy done)
                } else {
                    while (k_next < nt && te[k_next] >= t_new) {
                        int k = k_next++;
                        bool want_v = (k >= 1);
                        bool want_keep = keep_slot[k] >= 0;
                        if (!want_v && !want_keep) continue;
                        double x = (te[k] - t) / hh;
                        double xp[4] = {x, x * x, x * x * x, x * x * x * x};
                        Lin8<T> W;
                        W.n = 7;
                        for (int j = 0; j < 7; ++j) {
                            double w = 0;
                            for (int p = 0; p < 4; ++p) w += RK_P[j][p] * xp[p];
                            W.k[j] = S.K[j];
                            W.c[j] = (T)(hh * w);
                        }
                        double* kp = want_keep ? phi_keep + (size_t)keep_slot[k] * n : nullptr;
                        hipLaunchKernelGGL(k_dense<T>, S.g1(), dim3(256), 0, s, S.y, W, S.phi_tmp, kp,
                                           std::ldexp(1.0, expo), n);
                        if (want_v) {
                            int sidx = nt - 1 - k;
                            T* vx = (T*)vx_out + (size_t)sidx * slice_elems;
                            T* vy = (T*)vy_out + (size_t)sidx * slice_elems;
                            hipLaunchKernelGGL((k_vels<T, T>), S.g2(c.Nx - 2, c.Ny - 2), S.b2(), 0, s, S.phi_tmp, vx,
                                               vy, c.Ny, c.Nx, (T)(c.lim * scale_of()), (T)c.lim, (T)c.mu,
                                               (T)(1.0 / (2 * c.dx)), (T)(1.0 / (2 * c.dy)));
                            ++stats.n_slices_written;
                        }
                    }
                }
                // ---- advance
                std::swap(S.y, S.ynew);
                t = t_new;
                ++stats.n_accept;
                if (t_steps && n_steps_rec < t_steps_cap) t_steps[n_steps_rec++] = t;
                // ---- range management (fp32): keep max|y| below 2^64
                if (sizeof(T) == 4 && ymax > 0x1p64) {
                    int e = (int)std::floor(std::log2(ymax)) - 32;
                    T sc = (T)std::ldexp(1.0, -e);
                    hipLaunchKernelGGL(k_scale<T>, S.g1(), dim3(256), 0, s, S.y, S.K[6], sc, n);
                    expo += e;
                    ++stats.n_rescale;
                }
            } else {
                h_abs *= std::max(0.2, 0.9 * std::pow(err, err_exp));
                rejected = true;
            }
        }
        if (direction * (t - tf) >= 0) break;
    }
    OC_HIP_CHECK(hipStreamSynchronize(s));
    stats.nfev = S.nfev;
    stats.final_exponent = expo;
    if (st) *st = stats;
    return OC_OK;
}

}   // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int oc_hjb_make_coef(oc_handle* h, int dtype, const float* V, const void* m, void* coef, void* stream) {
    OC_REQUIRE(h && V && coef, "null argument");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const oc_config& c = h->cfg;
    double inv = 1.0 / (c.mu * c.sigma * c.sigma);
    int64_t n = h->ncell;
    dim3 g((unsigned)((n + 255) / 256));
    if (dtype == OC_F32)
        hipLaunchKernelGGL(k_make_coef<float>, g, dim3(256), 0, s, V, (const float*)m, c.g, inv, (float*)coef, n);
    else if (dtype == OC_F64)
        hipLaunchKernelGGL(k_make_coef<double>, g, dim3(256), 0, s, V, (const double*)m, c.g, inv, (double*)coef, n);
    else
        OC_REQUIRE(false, "bad dtype");
    OC_HIP_CHECK(hipGetLastError());
    return OC_OK;
}

__global__ void k_ctl_power(const double* __restrict__ x, int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = inv_fifth_root(x[i]);
}
extern "C" int oc_hjb_ctl_power(oc_handle* h, const double* x, int64_t n, double* out, void* stream) {
    OC_REQUIRE(h && x && out && n >= 0, "oc_hjb_ctl_power: bad argument");
    if (n > 0) hipLaunchKernelGGL(k_ctl_power, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, out);
    OC_HIP_CHECK(hipGetLastError());
    return OC_OK;
}

extern "C" int oc_hjb_rhs(oc_handle* h, int dtype, const void* phi, const void* coef, void* out, void* stream) {
    OC_REQUIRE(h && phi && coef && out, "null argument");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const oc_config& c = h->cfg;
    double a = -0.5 * c.sigma * c.sigma / (c.dx * c.dy);
    dim3 b(64, 4), g((c.Nx + 63) / 64, (c.Ny + 3) / 4);
    if (dtype == OC_F32)
        hipLaunchKernelGGL(k_rhs<float>, g, b, 0, s, (const float*)phi, (const float*)coef, (float*)out, c.Ny, c.Nx,
                           (float)a);
    else if (dtype == OC_F64)
        hipLaunchKernelGGL(k_rhs<double>, g, b, 0, s, (const double*)phi, (const double*)coef, (double*)out, c.Ny,
                           c.Nx, a);
    else
        OC_REQUIRE(false, "bad dtype");
    OC_HIP_CHECK(hipGetLastError());
    return OC_OK;
}

extern "C" int oc_hjb_vels(oc_handle* h, int dtype, const void* phi, double scale, void* vx, void* vy, void* stream) {
    OC_REQUIRE(h && phi && vx && vy, "null argument");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const oc_config& c = h->cfg;
    dim3 b(64, 4), g((c.Nx - 2 + 63) / 64, (c.Ny - 2 + 3) / 4);
    double i2x = 1.0 / (2 * c.dx), i2y = 1.0 / (2 * c.dy);
    if (dtype == OC_F32)
        hipLaunchKernelGGL((k_vels<double, float>), g, b, 0, s, (const double*)phi, (float*)vx, (float*)vy, c.Ny,
                           c.Nx, c.lim * scale, c.lim, c.mu, i2x, i2y);
    else if (dtype == OC_F64)
        hipLaunchKernelGGL((k_vels<double, double>), g, b, 0, s, (const double*)phi, (double*)vx, (double*)vy, c.Ny,
                           c.Nx, c.lim * scale, c.lim, c.mu, i2x, i2y);
    else
        OC_REQUIRE(false, "bad dtype");
    OC_HIP_CHECK(hipGetLastError());
    return OC_OK;
}

extern "C" int oc_hjb_solve(oc_handle* h, int dtype, const void* coef, double T, int32_t nt, void* vx_out,
                            void* vy_out, const int32_t* keep_idx, int32_t n_keep, double* phi_keep, double* t_steps,
                            int32_t t_steps_cap, oc_hjb_stats* stats, void* stream) {
    OC_REQUIRE(h && coef, "null argument");
    OC_REQUIRE(nt <= 1 || (vx_out && vy_out), "null output field");
    OC_REQUIRE(n_keep == 0 || (keep_idx && phi_keep), "keep list without buffer");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    auto t0 = std::chrono::steady_clock::now();
    int rc;
    if (dtype == OC_F32)
        rc = hjb_solve_t<float>(h, coef, T, nt, vx_out, vy_out, keep_idx, n_keep, phi_keep, t_steps, t_steps_cap,
                                stats, s);
    else if (dtype == OC_F64)
        rc = hjb_solve_t<double>(h, coef, T, nt, vx_out, vy_out, keep_idx, n_keep, phi_keep, t_steps, t_steps_cap,
                                 stats, s);
    else if (dtype == OC_F64_F32)
        rc = hjb_solve_t<double, float>(h, coef, T, nt, vx_out, vy_out, keep_idx, n_keep, phi_keep, t_steps, t_steps_cap,
                                        stats, s);
    else {
        oc_set_error("bad dtype %d", dtype);
        return OC_ERR_ARG;
    }
    if (rc == OC_OK) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            oc_set_error("hjb_solve: %s", hipGetErrorString(e));
            return OC_ERR_HIP;
        }
        if (stats) stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return rc;
}

// Row-block variant of oc_hjb_solve (one rank of a domain-decomposed solve), see oc_hip.h.
extern "C" int oc_hjb_solve_rows(oc_handle* h, int dtype, const void* coef_rows, double T, int32_t nt, void* vx_out,
                                 void* vy_out, const oc_hjb_shard* sh, const int32_t* keep_idx, int32_t n_keep,
                                 double* phi_keep, oc_hjb_stats* stats, void* stream) {
    OC_REQUIRE(h && coef_rows && sh, "null argument");
    OC_REQUIRE(nt <= 1 || (vx_out && vy_out), "null output field");
    OC_REQUIRE(n_keep == 0 || (keep_idx && phi_keep), "keep list without buffer");
    OC_REQUIRE(sh->world >= 1 && sh->world <= OC_MAX_WORLD, "bad world size");
    OC_REQUIRE(sh->halo >= 7, "halo must cover the 6 stencil applications of one RK step + the row of K6 the output reads");
    OC_REQUIRE(sh->own_lo == (sh->has_up ? sh->halo : 0), "own_lo must equal the upper halo depth");
    OC_REQUIRE(sh->own_hi > sh->own_lo && sh->Ny_local == sh->own_hi + (sh->has_dn ? sh->halo : 0), "bad row window");
    OC_REQUIRE(!(sh->has_up || sh->has_dn) || sh->own_hi - sh->own_lo >= sh->halo, "owned block thinner than the halo");
    OC_REQUIRE(sh->Ny_local >= 3 && sh->Ny_local <= sh->Ny_global && sh->Ny_global <= h->cfg.Ny, "bad row counts");
    OC_REQUIRE(sh->red_local && sh->red_all && (sh->allgather_red || sh->comm_slot >= 0), "missing reduction buffers / callback");
    OC_REQUIRE(!(sh->has_up || sh->has_dn) || sh->exchange_halo || sh->comm_slot >= 0, "missing halo callback");
    OC_REQUIRE(!sh->has_up || (sh->send_up && sh->recv_up), "missing upper staging buffers");
    OC_REQUIRE(!sh->has_dn || (sh->send_dn && sh->recv_dn), "missing lower staging buffers");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    auto t0 = std::chrono::steady_clock::now();
    int rc;
    if (dtype == OC_F32)
        rc = hjb_solve_t<float>(h, coef_rows, T, nt, vx_out, vy_out, keep_idx, n_keep, phi_keep, nullptr, 0, stats, s, sh);
    else if (dtype == OC_F64)
        rc = hjb_solve_t<double>(h, coef_rows, T, nt, vx_out, vy_out, keep_idx, n_keep, phi_keep, nullptr, 0, stats, s,
                                 sh);
    else if (dtype == OC_F64_F32)
        rc = hjb_solve_t<double, float>(h, coef_rows, T, nt, vx_out, vy_out, keep_idx, n_keep, phi_keep, nullptr, 0, stats,
                                        s, sh);
    else {
        oc_set_error("bad dtype %d", dtype);
        return OC_ERR_ARG;
    }
    if (rc == OC_OK) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            oc_set_error("hjb_solve_rows: %s", hipGetErrorString(e));
            return OC_ERR_HIP;
        }
        if (stats) stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return rc;
}

// ------------------------------------------------------------------------------------------ the interpolant ring
static int ring_check(const oc_handle* h, int dtype, const oc_hjb_ring* ring) {
    OC_REQUIRE(h && ring, "null argument");
    OC_REQUIRE(dtype == OC_F32 || dtype == OC_F64 || dtype == OC_F64_F32, "bad dtype");
    OC_REQUIRE(ring->data && ring->meta, "null ring buffers");
    OC_REQUIRE(ring->slots >= 1 && ring->nslice >= 0 && ring->rows >= 3 && ring->rows <= h->cfg.Ny, "bad ring dimensions");
    OC_REQUIRE(h->cfg.Nx % 4 == 0, "the interpolant ring needs Nx % 4 == 0 (16-byte rows)");
    return OC_OK;
}

extern "C" int64_t oc_hjb_ring_slot_elems(const oc_handle* h, int32_t rows) {
    return h ? (int64_t)OC_RING_VALS * rows * h->cfg.Nx : 0;
}

extern "C" int64_t oc_hjb_ring_meta_bytes(const oc_handle* h, int dtype, int32_t slots, int32_t nslice, int32_t rows) {
    if (!h || slots < 1 || rows < 1) return 0;
    return (int64_t)oc_ring_meta_size(dtype == OC_F32 ? 4 : 8, slots, nslice, rows, h->cfg.Nx);
}

// Accepted steps a solve over [T, 0] will take: the controller drives the explicit pair to its stability limit on this
// equation (h ~ 3.3 / (8 (sigma^2/2) / dx^2), measured 383 steps at dx = 0.05 and 93 at dx = 0.1 for T = 20),
// so T / h_stab plus a quarter and a constant covers it; a ring that turns out too small is reported (OC_ERR_CAPACITY).
extern "C" int32_t oc_hjb_ring_slots_hint(const oc_handle* h, double T) {
    if (!h) return 0;
    const oc_config& c = h->cfg;
    const double h_stab = 3.3 / (4.0 * c.sigma * c.sigma / (c.dx * c.dy));
    const double n = std::fabs(T) / h_stab;
    return (int32_t)std::min(1.25 * n + 16.0, 1e6);
}

extern "C" int oc_hjb_solve_ring(oc_handle* h, int dtype, const void* coef, double T, int32_t nt, const oc_hjb_ring* ring,
                                 const int32_t* keep_idx, int32_t n_keep, double* phi_keep, double* t_steps,
                                 int32_t t_steps_cap, oc_hjb_stats* stats, void* stream) {
    OC_REQUIRE(h && coef, "null argument");
    int rc = ring_check(h, dtype, ring);
    if (rc) return rc;
    OC_REQUIRE(ring->rows == h->cfg.Ny, "a whole-grid solve needs a ring of Ny rows");
    OC_REQUIRE(n_keep == 0 || (keep_idx && phi_keep), "keep list without buffer");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    auto t0 = std::chrono::steady_clock::now();
    if (dtype == OC_F32)
        rc = hjb_solve_t<float>(h, coef, T, nt, nullptr, nullptr, keep_idx, n_keep, phi_keep, t_steps, t_steps_cap, stats, s,
                                nullptr, ring);
    else if (dtype == OC_F64)
        rc = hjb_solve_t<double>(h, coef, T, nt, nullptr, nullptr, keep_idx, n_keep, phi_keep, t_steps, t_steps_cap, stats, s,
                                 nullptr, ring);
    else
        rc = hjb_solve_t<double, float>(h, coef, T, nt, nullptr, nullptr, keep_idx, n_keep, phi_keep, t_steps, t_steps_cap,
                                        stats, s, nullptr, ring);
    if (rc == OC_OK) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            oc_set_error("hjb_solve_ring: %s", hipGetErrorString(e));
            return OC_ERR_HIP;
        }
        if (stats) stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return rc;
}

// Several independent whole-grid ring solves at the same time (the reference solves its keys one after the other,
// simulations.py:419-420; they share nothing): one host thread per job INSIDE the library -- job 0 on the caller's -- so that
// nothing of the interpreter lies between the solves' launches (threads started from Python cost 0.3 - 0.7 ms per round of
// two 2 ms solves: thread start and join, the calls around the solve).  Every job needs its own handle (solver scratch) and
// stream.  Returns the first job's error code that is not OC_OK (each job's own is in jobs[i].rc, its message in .error).
// The host threads of oc_hjb_solve_ring_many: started once, kept (a thread started per call cost its creation and its first
// HIP call's per-thread set-up, ~0.1 ms of a 1.7 ms solve, in every round of solves).  Never destroyed: the threads sleep on
// their condition variables until the process ends.
namespace {
struct SolveWorker {
    std::mutex m;
    std::condition_variable cv;
    oc_hjb_ring_job* job = nullptr;     // set by the caller, cleared by the worker when it is done
    void (*fn)(oc_hjb_ring_job*) = nullptr;
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return job != nullptr; });
            oc_hjb_ring_job* j = job;
            lk.unlock();
            fn(j);
            lk.lock();
            job = nullptr;
            cv.notify_all();
        }
    }
};
std::mutex g_workers_m;                  // one oc_hjb_solve_ring_many at a time uses the workers
std::vector<SolveWorker*>* g_workers = nullptr;
}

extern "C" int oc_hjb_solve_ring_many(oc_hjb_ring_job* jobs, int32_t n) {
    OC_REQUIRE(jobs != nullptr && n >= 1 && n <= 64, "bad job list");
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) OC_REQUIRE(jobs[i].h != jobs[j].h, "two jobs on one handle (each needs its own solver scratch)");
    void (*run)(oc_hjb_ring_job*) = [](oc_hjb_ring_job* jb) {
        jb->error[0] = 0;
        jb->rc = oc_hjb_make_coef(jb->h, jb->coef_dtype, jb->V, jb->m, jb->coef, jb->stream);
        if (jb->rc == OC_OK)
            jb->rc = oc_hjb_solve_ring(jb->h, jb->dtype, jb->coef, jb->T, jb->nt, jb->ring, nullptr, 0, nullptr, nullptr, 0, &jb->stats,
                                       jb->stream);
        if (jb->rc != OC_OK) {
            strncpy(jb->error, oc_last_error(), sizeof(jb->error) - 1);
            jb->error[sizeof(jb->error) - 1] = 0;
        }
    };
    {
        std::lock_guard<std::mutex> pool(g_workers_m);
        if (!g_workers) g_workers = new std::vector<SolveWorker*>();
        while ((int)g_workers->size() < n - 1) {
            SolveWorker* w = new SolveWorker();
            g_workers->push_back(w);
            std::thread([w] { w->loop(); }).detach();
        }
        for (int i = 1; i < n; ++i) {
            SolveWorker* w = (*g_workers)[i - 1];
            std::lock_guard<std::mutex> lk(w->m);
            w->fn = run;
            w->job = jobs + i;
            w->cv.notify_all();
        }
        run(jobs);
        for (int i = 1; i < n; ++i) {
            SolveWorker* w = (*g_workers)[i - 1];
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [&] { return w->job == nullptr; });
        }
    }
    for (int i = 0; i < n; ++i)
        if (jobs[i].rc != OC_OK) {
            oc_set_error("job %d: %s", i, jobs[i].error);
            return jobs[i].rc;
        }
    return OC_OK;
}

extern "C" int oc_hjb_solve_rows_ring(oc_handle* h, int dtype, const void* coef_rows, double T, int32_t nt,
                                      const oc_hjb_ring* ring, const oc_hjb_shard* sh, const int32_t* keep_idx,
                                      int32_t n_keep, double* phi_keep, oc_hjb_stats* stats, void* stream) {
    OC_REQUIRE(h && coef_rows && sh, "null argument");
    int rc = ring_check(h, dtype, ring);
    if (rc) return rc;
    OC_REQUIRE(ring->rows == sh->Ny_local, "the ring must hold the row block's Ny_local rows");
    OC_REQUIRE(n_keep == 0 || (keep_idx && phi_keep), "keep list without buffer");
    OC_REQUIRE(sh->world >= 1 && sh->world <= OC_MAX_WORLD, "bad world size");
    OC_REQUIRE(sh->halo >= 7, "halo must cover the 6 stencil applications of one RK step + the row of K6 the output reads");
    OC_REQUIRE(sh->own_lo == (sh->has_up ? sh->halo : 0), "own_lo must equal the upper halo depth");
    OC_REQUIRE(sh->own_hi > sh->own_lo && sh->Ny_local == sh->own_hi + (sh->has_dn ? sh->halo : 0), "bad row window");
    OC_REQUIRE(!(sh->has_up || sh->has_dn) || sh->own_hi - sh->own_lo >= sh->halo, "owned block thinner than the halo");
    OC_REQUIRE(sh->Ny_local >= 3 && sh->Ny_local <= sh->Ny_global && sh->Ny_global <= h->cfg.Ny, "bad row counts");
    OC_REQUIRE(sh->red_local && sh->red_all && (sh->allgather_red || sh->comm_slot >= 0), "missing reduction buffers / callback");
    OC_REQUIRE(!(sh->has_up || sh->has_dn) || sh->exchange_halo || sh->comm_slot >= 0, "missing halo callback");
    OC_REQUIRE(!sh->has_up || (sh->send_up && sh->recv_up), "missing upper staging buffers");
    OC_REQUIRE(!sh->has_dn || (sh->send_dn && sh->recv_dn), "missing lower staging buffers");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    auto t0 = std::chrono::steady_clock::now();
    if (dtype == OC_F32)
        rc = hjb_solve_t<float>(h, coef_rows, T, nt, nullptr, nullptr, keep_idx, n_keep, phi_keep, nullptr, 0, stats, s, sh, ring);
    else if (dtype == OC_F64)
        rc = hjb_solve_t<double>(h, coef_rows, T, nt, nullptr, nullptr, keep_idx, n_keep, phi_keep, nullptr, 0, stats, s, sh,
                                 ring);
    else
        rc = hjb_solve_t<double, float>(h, coef_rows, T, nt, nullptr, nullptr, keep_idx, n_keep, phi_keep, nullptr, 0, stats, s,
                                        sh, ring);
    if (rc == OC_OK) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            oc_set_error("hjb_solve_rows_ring: %s", hipGetErrorString(e));
            return OC_ERR_HIP;
        }
        if (stats) stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return rc;
}

// Slices [s_lo, s_hi) of the direction field from a ring, by the slice writer of oc_hjb_solve (k_out) itself: the arrays a
// slice-writing solve would have left, bit for bit.  vx_out / vy_out: (s_hi - s_lo, rows, Nx) of the field's type.
extern "C" int oc_hjb_ring_slices(oc_handle* h, int dtype, const oc_hjb_ring* ring, int32_t s_lo, int32_t s_hi, void* vx_out,
                                  void* vy_out, void* stream) {
    int rc = ring_check(h, dtype, ring);
    if (rc) return rc;
    OC_REQUIRE(vx_out && vy_out, "null output");
    OC_REQUIRE(s_lo >= 0 && s_lo <= s_hi, "bad slice range");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    OcRingHead rh;
    OC_HIP_CHECK(hipStreamSynchronize(s));
    OC_HIP_CHECK(hipMemcpy(&rh, ring->meta, sizeof(rh), hipMemcpyDeviceToHost));
    OC_REQUIRE(rh.cap == ring->slots && rh.nslice == ring->nslice && rh.n_slots >= 0 && rh.n_slots <= rh.cap,
               "the ring holds no finished solve (or was made with other dimensions)");
    OC_REQUIRE(s_hi <= rh.nt - 1 || s_hi == s_lo, "slices beyond the solve's horizon (nt - 1) do not exist in the ring");
    if (s_hi == s_lo) return OC_OK;
    if (dtype == OC_F32) return ring_emit<float, float>(h, ring_view<float>(ring, h->cfg.Nx), rh.n_slots, s_lo, s_hi, vx_out, vy_out, nullptr, nullptr, s);
    if (dtype == OC_F64) return ring_emit<double, double>(h, ring_view<double>(ring, h->cfg.Nx), rh.n_slots, s_lo, s_hi, vx_out, vy_out, nullptr, nullptr, s);
    return ring_emit<double, float>(h, ring_view<double>(ring, h->cfg.Nx), rh.n_slots, s_lo, s_hi, vx_out, vy_out, nullptr, nullptr, s);
}

