// Room description -> potential on the grid, on the device: replaces simulation.create_potential (reference
// simulations.py:516-576) followed by the finalisation of optimals.__init__ (optimals.py:89-91) for large grids, where
// the host rasteriser's full-grid passes and the 67 MB upload per key are what initialisation waits for.
// One thread per grid node evaluates the reference's own predicates on the same float64 axes, in the reference's
// order of drawing, so the result is bit-identical to the host path:
//   every wall rectangle containing the node adds -1;  a hole resets to 0;  every cylinder containing it adds -1;
//   the border is -1;  the key's target rectangles are +1 (drawn last);  then V<0 -> wall_potential,
//   V>0 -> target_potential.
#include "oc_internal.h"

namespace {

struct RasterArgs {
    const double* shapes;       // device: walls (4 each), holes (4), cylinders (3), targets (4), in this order
    int32_t n_walls, n_holes, n_cyl, n_targets;
    int32_t Ny, Nx;
    double wall_potential, target_potential;
    int32_t finalise;
};

__device__ __forceinline__ bool in_rect(double X, double Y, const double* r) {
    // np.outer(abs(Y0 - r[1]) < r[3] / 2, abs(X0 - r[0]) < r[2] / 2)
    return (fabs(Y - r[1]) < r[3] / 2) && (fabs(X - r[0]) < r[2] / 2);
}

__global__ void __launch_bounds__(256) k_rasterise(RasterArgs A, const double* __restrict__ X0,
                                                   const double* __restrict__ Y0, float* __restrict__ V) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= A.Ny || j >= A.Nx) return;
    const double X = X0[j], Y = Y0[i];
    const double* s = A.shapes;
    double v = 0.0;
    for (int q = 0; q < A.n_walls; ++q, s += 4)
        if (in_rect(X, Y, s)) v -= 1.0;
    for (int q = 0; q < A.n_holes; ++q, s += 4)
        if (in_rect(X, Y, s)) v = 0.0;
    for (int q = 0; q < A.n_cyl; ++q, s += 3) {
        const double dx = X - s[0], dy = Y - s[1];
        if (sqrt(dx * dx + dy * dy) < s[2]) v -= 1.0;
    }
    if (i == 0 || j == 0 || i == A.Ny - 1 || j == A.Nx - 1) v = -1.0;
    for (int q = 0; q < A.n_targets; ++q, s += 4)
        if (in_rect(X, Y, s)) v = 1.0;
    if (A.finalise) {
        if (v < 0) v = A.wall_potential;
        else if (v > 0) v = A.target_potential;
    }
    V[(size_t)i * A.Nx + j] = (float)v;
}

}   // namespace

extern "C" int oc_rasterise_potential(oc_handle* h, const double* walls, int32_t n_walls, const double* holes,
                                      int32_t n_holes, const double* cylinders, int32_t n_cyl, const double* targets,
                                      int32_t n_targets, double wall_potential, double target_potential,
                                      int32_t finalise, float* V_out, void* stream) {
    OC_REQUIRE(h && V_out, "null argument");
    OC_REQUIRE(n_walls >= 0 && n_holes >= 0 && n_cyl >= 0 && n_targets >= 0, "negative shape count");
    OC_REQUIRE((n_walls == 0 || walls) && (n_holes == 0 || holes) && (n_cyl == 0 || cylinders) &&
                   (n_targets == 0 || targets), "null shape array");
    OC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    std::vector<double> host;
    host.insert(host.end(), walls, walls + 4 * (size_t)n_walls);
    host.insert(host.end(), holes, holes + 4 * (size_t)n_holes);
    host.insert(host.end(), cylinders, cylinders + 3 * (size_t)n_cyl);
    host.insert(host.end(), targets, targets + 4 * (size_t)n_targets);
    const size_t bytes = sizeof(double) * std::max<size_t>(host.size(), 1);
    if (bytes > h->shape_bytes) {
        if (h->shape_buf) OC_HIP_CHECK(hipFree(h->shape_buf));
        h->shape_buf = nullptr;
        h->shape_bytes = 0;
        OC_HIP_CHECK(hipMalloc(&h->shape_buf, bytes));
        h->shape_bytes = bytes;
    }
    // a few hundred bytes, once per key: synchronous copy (the previous call's kernel may still be reading the buffer)
    OC_HIP_CHECK(hipStreamSynchronize(s));
    if (!host.empty()) OC_HIP_CHECK(hipMemcpy(h->shape_buf, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice));
    RasterArgs A;
    A.shapes = (const double*)h->shape_buf;
    A.n_walls = n_walls; A.n_holes = n_holes; A.n_cyl = n_cyl; A.n_targets = n_targets;
    A.Ny = h->cfg.Ny; A.Nx = h->cfg.Nx;
    A.wall_potential = wall_potential; A.target_potential = target_potential;
    A.finalise = finalise;
    hipLaunchKernelGGL(k_rasterise, dim3((A.Nx + 63) / 64, (A.Ny + 3) / 4), dim3(256), 0, s, A, h->X0, h->Y0, V_out);
    OC_HIP_CHECK(hipGetLastError());
    return OC_OK;
}
