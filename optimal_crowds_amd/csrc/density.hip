// Gaussian crowd density on the room grid: replaces simulation.gaussian_density, reference simulations.py:453-487.
//   d[i,j] = sum_{active a} exp(-((X_j-x_a)^2 + (Y_i-y_a)^2) / (2 sigma^2)) / (2 pi sigma);   d[V_all < 0] = 0
// (the reference's normaliser, quirk Q11; 2 pi sigma^2 with OC_FIX_DENSITY_NORM)
// Agents are ordered by coarse cells of side R = trunc_sigmas*sigma; an output tile only visits the agents of the
// coarse cells that intersect tile +- R (row-major keys make each coarse row one contiguous slot range).  Everything
// inside those cells is summed (no radial test), what is dropped is < exp(-trunc^2/2) relative.  The exponential is
// evaluated in its separable form exp(-dx^2/2s^2) exp(-dy^2/2s^2): equal to the reference's single exp of the sum up
// to two float64 roundings per term.
#include "oc_internal.h"
#include <cmath>

namespace {

struct DensParams {
    int32_t Ny, Nx;
    double cell;                // side of the coarse cells the agents are ordered by (a third of the truncation radius)
    double rcut;                // truncation radius trunc_sigmas * sigma: a tile sums the agents of the cells that meet tile +- rcut
    int32_t ncx, ncy;
    double inv2s2, invC;
    double hx, hy, kx, ky;      // grid steps and exp(-h^2 / sigma^2): the Gaussian's factors along a grid line by recurrence
    int32_t recur, pad;         // 1: the axes are equally spaced (np.linspace): recurrence; 0: a true exponential per factor
    int32_t row_lo, row_hi;     // grid rows [row_lo, row_hi) are computed (oc_density_rows: a rank's row block; oc_density: all)
};

__global__ void k_dens_keys(const double* __restrict__ x, const double* __restrict__ y,
                            const uint8_t* __restrict__ status, int N, DensParams P, uint32_t* __restrict__ keys,
                            int32_t* __restrict__ vals) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    uint32_t key = 0x3FFFFFFFu;
    if (status[i]) {
        double fx = floor(x[i] / P.cell), fy = floor(y[i] / P.cell);
        int cx = fx < 0 ? 0 : (fx > P.ncx - 1 ? P.ncx - 1 : (int)fx);
        int cy = fy < 0 ? 0 : (fy > P.ncy - 1 ? P.ncy - 1 : (int)fy);
        key = (uint32_t)(cy * P.ncx + cx);
    }
    keys[i] = key;
    vals[i] = i;
}

__global__ void k_dens_gather(const double* __restrict__ x, const double* __restrict__ y,
                              const int32_t* __restrict__ order, int N, double2* __restrict__ spos) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    int i = order[k];
    spos[k] = make_double2(x[i], y[i]);
}

// cell_start[c] = first slot with key >= c   (c = 0..ncell)
__global__ void k_dens_cellstart(const uint32_t* __restrict__ keys_sorted, int N, int ncell, int32_t* __restrict__ cs) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncell) return;
    int lo = 0, hi = N;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (keys_sorted[mid] < (uint32_t)c) lo = mid + 1;
        else hi = mid;
    }
    cs[c] = lo;
}

// The Gaussian is separable: exp(-(dx^2 + dy^2)/(2 s^2)) = ex(column, agent) * ey(row, agent).  A block
// owns DT x DT = 64 x 64 output cells, a thread a 4 x 4 block of them (rows ty + 16 a, columns tx + 16 b); per batch of DB
// agents the block evaluates the 128 one-dimensional factors of every agent once into LDS and then adds the rank-one
// products: 8 LDS reads for 16 float64 FMAs per agent and thread.
// (Round 3's kernel -- 32 x 16 tiles, 2 outputs per thread -- was bound by LDS: 1.5 reads per FMA, 83 % of the index unit's
// cycles; and a tile that small sums, with its margin of 6 sigma on every side, the agents of 21 x its own area, where a 64 x 64
// tile sums 4.4 x: the work per output fell by more than the reads per FMA did.)
#define DT 64
#define DB 64
template <typename T>
__global__ void __launch_bounds__(256) k_density(DensParams P, const double2* __restrict__ spos,
                                                 const int32_t* __restrict__ cs, const double* __restrict__ X0,
                                                 const double* __restrict__ Y0, const float* __restrict__ Vall,
                                                 T* __restrict__ out) {
    // (rows padded by one double: the factor table is WRITTEN one row per lane -- agent fq = lane -- and an unpadded row
    // put all 64 lanes of a store on the same banks)
    __shared__ double lex[DB][DT + 1], ley[DB][DT + 1];    // 66.6 KB
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int j0 = blockIdx.x * DT, i0 = P.row_lo + blockIdx.y * DT;
    const int j1 = min(j0 + DT - 1, P.Nx - 1), i1 = min(i0 + DT - 1, P.row_hi - 1);
    // coarse cells intersecting tile +- cell
    auto cc = [&](double v, int n) {
        double f = floor(v / P.cell);
        return f < 0 ? 0 : (f > n - 1 ? n - 1 : (int)f);
    };
    const int cx0 = cc(X0[j0] - P.rcut, P.ncx), cx1 = cc(X0[j1] + P.rcut, P.ncx);
    const int cy0 = cc(Y0[i0] - P.rcut, P.ncy), cy1 = cc(Y0[i1] + P.rcut, P.ncy);
    // this thread's share of the factor table: agent fq of the batch, 32 of its 128 factors (part 0, 1: columns; 2, 3: rows)
    const int fq = threadIdx.x & (DB - 1), fpart = threadIdx.x >> 6;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int cy = cy0; cy <= cy1; ++cy) {
        const int s0 = cs[cy * P.ncx + cx0], s1 = cs[cy * P.ncx + cx1 + 1];
        for (int base = s0; base < s1; base += DB) {
            const int cnt = min(DB, s1 - base);
            __syncthreads();
            if (fq < cnt) {
                // The factors of one agent along a grid line are a Gaussian sampled at equal steps h: g(c + 1) = g(c) r(c),
                // r(c + 1) = r(c) k with r(c) = exp(-(2 d_c h + h^2) / 2 sigma^2), k = exp(-h^2 / sigma^2) -- two multiplications
                // per factor where a float64 exp costs ~40 instructions (the exponentials were four fifths of round 3's
                // kernel).  Restarted from true exponentials on the grid's own axis value every 16 factors: the drift stays
                // below 1e-14 relative, and what the axis deviates from equal steps (an ulp of its values) below 2e-12 of a
                // factor that is e^-18 itself there.
                const double2 p = spos[base + fq];
                const int c0 = 32 * (fpart & 1);
                const bool xs_ = fpart < 2;
                const double pc = xs_ ? p.x : p.y, h = xs_ ? P.hx : P.hy, k = xs_ ? P.kx : P.ky, sc = xs_ ? P.invC : 1.0;
                const double* __restrict__ ax = xs_ ? X0 : Y0;
                const int a0 = (xs_ ? j0 : i0), amax = (xs_ ? P.Nx : P.Ny) - 1;
                double* const row = xs_ ? lex[fq] : ley[fq];
#pragma unroll
                for (int seg = 0; seg < 2; ++seg) {
                    const int cb = c0 + 16 * seg;
                    const double d = ax[min(a0 + cb, amax)] - pc;
                    double g = exp(-(d * d) * P.inv2s2) * sc;
                    if (P.recur) {
                        double r = exp(-(2.0 * d * h + h * h) * P.inv2s2);
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            row[cb + c] = g;
                            g *= r;
                            r *= k;
                        }
                    } else {
                        for (int c = 0; c < 16; ++c) {
                            const double dc = ax[min(a0 + cb + c, amax)] - pc;
                            row[cb + c] = exp(-(dc * dc) * P.inv2s2) * sc;
                        }
                    }
                }
            }
            __syncthreads();
            for (int q = 0; q < cnt; ++q) {
                double ex[4], ey[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) ex[b] = lex[q][tx + 16 * b];
#pragma unroll
                for (int a = 0; a < 4; ++a) ey[a] = ley[q][ty + 16 * a];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] += ex[b] * ey[a];
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = i0 + ty + 16 * a;
        if (i >= P.row_hi) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + tx + 16 * b;
            if (j < P.Nx) {
                const size_t o = (size_t)i * P.Nx + j;
                double v = acc[a][b];
                if (Vall && Vall[o] < 0.f) v = 0.0;
                out[o] = (T)v;
            }
        }
    }
}

}   // namespace

extern "C" int oc_density(oc_handle* h, int dtype, const oc_agents* a, double sigma, const float* V_all,
                          double trunc_sigmas, void* out, void* stream) {
    OC_REQUIRE(h, "null handle");
    return oc_density_rows(h, dtype, a, sigma, V_all, trunc_sigmas, 0, h->cfg.Ny, out, stream);
}

extern "C" int oc_density_rows(oc_handle* h, int dtype, const oc_agents* a, double sigma, const float* V_all,
                               double trunc_sigmas, int32_t row_lo, int32_t row_hi, void* out, void* stream) {
    OC_REQUIRE(h && a && out, "null argument");
    OC_REQUIRE(row_lo >= 0 && row_lo <= row_hi && row_hi <= h->cfg.Ny, "bad row range");
    OC_REQUIRE(a->N >= 0 && a->N <= h->cfg.max_agents, "agent count exceeds max_agents");
    OC_REQUIRE(sigma > 0, "sigma must be positive");
    OC_REQUIRE(dtype == OC_F32 || dtype == OC_F64, "bad dtype");
    OC_HIP_CHECK(hipSetDevice(h->device));
    {   // lazy_state: the positions and the status are read from the caller's arrays
        int rcf = oc_flush_state(h, (hipStream_t)stream);
        if (rcf) return rcf;
    }
    hipStream_t s = (hipStream_t)stream;
    const oc_config& c = h->cfg;
    if (trunc_sigmas <= 0) trunc_sigmas = 6.0;
    DensParams P;
    P.Ny = c.Ny;
    P.Nx = c.Nx;
    P.row_lo = row_lo;
    P.row_hi = row_hi;
    P.rcut = trunc_sigmas * sigma;
    // coarse cells of a third of the radius: what a tile sums beyond its own reach shrinks with the cells (a 6.4 m tile with
    // 3 m of reach on every side: the cells it meets cover 15.4 m with 3 m cells, 13.4 m with 1 m cells), as long as the
    // coarse grid stays small
    P.cell = P.rcut / 3.0;
    while ((c.room_length / P.cell + 1.0) * (c.room_height / P.cell + 1.0) > 4.0e6) P.cell *= 2.0;
    P.ncx = (int)(c.room_length / P.cell) + 1;
    P.ncy = (int)(c.room_height / P.cell) + 1;
    P.inv2s2 = 1.0 / (2 * sigma * sigma);
    // (the recurrence needs equal steps: the reference's np.linspace axes, checked in oc_create; any other axis: true exponentials)
    P.recur = h->axes_linspace ? 1 : 0;
    P.hx = c.room_length / (c.Nx - 1);
    P.hy = c.room_height / (c.Ny - 1);
    P.kx = std::exp(-P.hx * P.hx / (sigma * sigma));
    P.ky = std::exp(-P.hy * P.hy / (sigma * sigma));
    P.invC = (h->fix_flags & OC_FIX_DENSITY_NORM) ? 1.0 / (2 * M_PI * sigma * sigma)            // Q11 corrected
                                                    : 1.0 / std::sqrt(4 * M_PI * M_PI * sigma * sigma);
    const int N = a->N, ncell = P.ncx * P.ncy;
    OC_REQUIRE((int64_t)P.ncx * P.ncy < 0x3FFFFFFF, "too many coarse cells");
    // scratch: keys/vals x2, sorted positions, cell starts
    size_t need = (size_t)N * (2 * sizeof(uint32_t) + 2 * sizeof(int32_t) + sizeof(double2)) +
                  sizeof(int32_t) * (size_t)(ncell + 2) + 4096;
    if (need > h->dens_bytes) {
        if (h->dens_buf) OC_HIP_CHECK(hipFree(h->dens_buf));
        h->dens_buf = nullptr;
        h->dens_bytes = 0;
        OC_HIP_CHECK(hipMalloc(&h->dens_buf, need));
        h->dens_bytes = need;
    }
    char* p = (char*)h->dens_buf;
    auto take = [&](size_t bytes) {
        char* q = p;
        p += (bytes + 255) & ~(size_t)255;
        return (void*)q;
    };
    double2* spos = (double2*)take(sizeof(double2) * (size_t)N);
    uint32_t* ka = (uint32_t*)take(sizeof(uint32_t) * (size_t)N);
    uint32_t* kb = (uint32_t*)take(sizeof(uint32_t) * (size_t)N);
    int32_t* va = (int32_t*)take(sizeof(int32_t) * (size_t)N);
    int32_t* vb = (int32_t*)take(sizeof(int32_t) * (size_t)N);
    int32_t* cs = (int32_t*)take(sizeof(int32_t) * (size_t)(ncell + 2));
    int nact_slots = 0;
    if (N > 0) {
        hipLaunchKernelGGL(k_dens_keys, dim3((N + 255) / 256), dim3(256), 0, s, a->x, a->y, a->status, N, P, ka, va);
        int rc = oc_sort_pairs(h, ka, kb, va, vb, N, 30, s);
        if (rc) return rc;
        hipLaunchKernelGGL(k_dens_gather, dim3((N + 255) / 256), dim3(256), 0, s, a->x, a->y, vb, N, spos);
        nact_slots = N;
    }
    hipLaunchKernelGGL(k_dens_cellstart, dim3((ncell + 1 + 255) / 256), dim3(256), 0, s, kb, nact_slots, ncell, cs);
    if (row_hi == row_lo) return OC_OK;
    dim3 g((c.Nx + DT - 1) / DT, (row_hi - row_lo + DT - 1) / DT);
    oc_timer_begin(h, "density", s);
    if (dtype == OC_F32)
        hipLaunchKernelGGL(k_density<float>, g, dim3(256), 0, s, P, spos, cs, h->X0, h->Y0, V_all, (float*)out);
    else
        hipLaunchKernelGGL(k_density<double>, g, dim3(256), 0, s, P, spos, cs, h->X0, h->Y0, V_all, (double*)out);
    // algorithmic bytes: the agents' positions once + the grid written once (the kernel is bound by float64 exp / FMA)
    oc_timer_end(h, "density", s, 16.0 * N + (dtype == OC_F32 ? 4.0 : 8.0) * (double)c.Nx * (row_hi - row_lo));
    OC_HIP_CHECK(hipGetLastError());
    return OC_OK;
}
