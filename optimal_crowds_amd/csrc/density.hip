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
    double cell;
    int32_t ncx, ncy;
    double inv2s2, invC;
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
// owns DT_X x DT_Y = 32 x 16 output cells; per batch of DB agents it evaluates the 48 one-dimensional factors of every
// agent once (24 exp per thread instead of one per cell and agent: 512 -> 48 per agent and tile) into LDS and then adds
// the rank-one products: 2 outputs per thread, one float64 FMA and 1.5 LDS reads per output and agent.
#define DT_X 32
#define DT_Y 16
#define DB 128
template <typename T>
__global__ void __launch_bounds__(256) k_density(DensParams P, const double2* __restrict__ spos,
                                                 const int32_t* __restrict__ cs, const double* __restrict__ X0,
                                                 const double* __restrict__ Y0, const float* __restrict__ Vall,
                                                 T* __restrict__ out) {
    // (rows padded by one double: the factor table is WRITTEN one row per lane -- agent fq = lane -- and an unpadded row of
    // 32 doubles put all 64 lanes of a store on the same banks: a third of the LDS cycles of the launch were conflicts)
    __shared__ double lex[DB][DT_X + 1], ley[DB][DT_Y + 1];    // 51 KB
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int j0 = blockIdx.x * DT_X, i0 = blockIdx.y * DT_Y;
    const int j1 = min(j0 + DT_X - 1, P.Nx - 1), i1 = min(i0 + DT_Y - 1, P.Ny - 1);
    // coarse cells intersecting tile +- cell
    auto cc = [&](double v, int n) {
        double f = floor(v / P.cell);
        return f < 0 ? 0 : (f > n - 1 ? n - 1 : (int)f);
    };
    const int cx0 = cc(X0[j0] - P.cell, P.ncx), cx1 = cc(X0[j1] + P.cell, P.ncx);
    const int cy0 = cc(Y0[i0] - P.cell, P.ncy), cy1 = cc(Y0[i1] + P.cell, P.ncy);
    // this thread's share of the factor table: agent fq of the batch, 24 of its 48 factors
    const int fq = threadIdx.x & (DB - 1), fpart = threadIdx.x >> 7;
    double acc0 = 0.0, acc1 = 0.0;
    for (int cy = cy0; cy <= cy1; ++cy) {
        const int s0 = cs[cy * P.ncx + cx0], s1 = cs[cy * P.ncx + cx1 + 1];
        for (int base = s0; base < s1; base += DB) {
            const int cnt = min(DB, s1 - base);
            __syncthreads();
            if (fq < cnt) {
                const double2 p = spos[base + fq];
                if (fpart == 0) {
#pragma unroll 4
                    for (int c = 0; c < 24; ++c) {
                        const double d = X0[min(j0 + c, P.Nx - 1)] - p.x;
                        lex[fq][c] = exp(-(d * d) * P.inv2s2) * P.invC;
                    }
                } else {
#pragma unroll 4
                    for (int c = 24; c < DT_X; ++c) {
                        const double d = X0[min(j0 + c, P.Nx - 1)] - p.x;
                        lex[fq][c] = exp(-(d * d) * P.inv2s2) * P.invC;
                    }
#pragma unroll 4
                    for (int c = 0; c < DT_Y; ++c) {
                        const double d = Y0[min(i0 + c, P.Ny - 1)] - p.y;
                        ley[fq][c] = exp(-(d * d) * P.inv2s2);
                    }
                }
            }
            __syncthreads();
            for (int q = 0; q < cnt; ++q) {
                const double ey = ley[q][ty];
                acc0 += lex[q][tx] * ey;
                acc1 += lex[q][tx + 16] * ey;
            }
        }
    }
    const int i = i0 + ty;
    if (i < P.Ny) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = j0 + tx + 16 * u;
            if (j < P.Nx) {
                const size_t o = (size_t)i * P.Nx + j;
                double v = u ? acc1 : acc0;
                if (Vall && Vall[o] < 0.f) v = 0.0;
                out[o] = (T)v;
            }
        }
    }
}

}   // namespace

extern "C" int oc_density(oc_handle* h, int dtype, const oc_agents* a, double sigma, const float* V_all,
                          double trunc_sigmas, void* out, void* stream) {
    OC_REQUIRE(h && a && out, "null argument");
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
    P.cell = trunc_sigmas * sigma;
    P.ncx = (int)(c.room_length / P.cell) + 1;
    P.ncy = (int)(c.room_height / P.cell) + 1;
    P.inv2s2 = 1.0 / (2 * sigma * sigma);
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
    dim3 g((c.Nx + DT_X - 1) / DT_X, (c.Ny + DT_Y - 1) / DT_Y);
    oc_timer_begin(h, "density", s);
    if (dtype == OC_F32)
        hipLaunchKernelGGL(k_density<float>, g, dim3(256), 0, s, P, spos, cs, h->X0, h->Y0, V_all, (float*)out);
    else
        hipLaunchKernelGGL(k_density<double>, g, dim3(256), 0, s, P, spos, cs, h->X0, h->Y0, V_all, (double*)out);
    // algorithmic bytes: the agents' positions once + the grid written once (the kernel is bound by float64 exp / FMA)
    oc_timer_end(h, "density", s, 16.0 * N + (dtype == OC_F32 ? 4.0 : 8.0) * (double)c.Nx * c.Ny);
    OC_HIP_CHECK(hipGetLastError());
    return OC_OK;
}
