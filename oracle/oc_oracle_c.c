/* oc_oracle_c.c -- C restatement (float64, OpenMP) of the optimal_crowds hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Same role and rules as oracle/oc_oracle.py: it CHECKS the HIP path and serves as bench.py's `cpu_baseline`
 * ("port") on the GPU box's host cores; the product never links or calls it.  Every function follows the cited
 * lines of the reference (optimals.py / pedestrians.py / simulations.py) and is pinned by tests/test_oracle_c.py
 * against the fixtures captured from the reference itself (G2, G6, G7) and against the NumPy oracle.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -ffp-contract=off -shared)  ->  oracle/liboc_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t Ny, Nx;
    double dx, dy, L, H, dt;
    double sigma, mu, g, lim;
    double a_min, tau_a, b_min, b_max, eta, eta_walls, cutoff, v_max, relaxation, noise_amp;
} occ_params;

int occ_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- HJB right-hand side, optimals.py:144-164 (mirror ghosts :149-152, wall mask :162) */
void occ_hjb_rhs(const occ_params* P, const double* phi, const double* V, const double* m, double* out) {
    const int Ny = P->Ny, Nx = P->Nx;
    const double a = -0.5 * P->sigma * P->sigma, dd = P->dx * P->dy, ms = P->mu * P->sigma * P->sigma;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < Ny; ++i) {
        const int iu = (i == 0) ? 1 : i - 1, id = (i == Ny - 1) ? Ny - 2 : i + 1;
        for (int j = 0; j < Nx; ++j) {
            const int jl = (j == 0) ? 1 : j - 1, jr = (j == Nx - 1) ? Nx - 2 : j + 1;
            const size_t c = (size_t)i * Nx + j;
            if (V[c] < 0) {
                out[c] = 0.0;
                continue;
            }
            const double C = phi[c];
            const double lap = (phi[(size_t)iu * Nx + j] + phi[(size_t)id * Nx + j] + phi[(size_t)i * Nx + jl] +
                                phi[(size_t)i * Nx + jr] - 4 * C) / dd;
            out[c] = a * lap - ((V[c] + P->g * m[c]) * C) / ms;
        }
    }
}

/* ---- direction field of one phi, optimals.py:168-186; outputs (Ny-2, Nx-2) */
static inline double clipl(double p, double lim) { return p * (p > lim) + lim * (p < lim); }
void occ_vels(const occ_params* P, const double* phi, double* vx, double* vy) {
    const int Ny = P->Ny, Nx = P->Nx;
    const double lim = P->lim;
#pragma omp parallel for schedule(static)
    for (int i = 1; i < Ny - 1; ++i)
        for (int j = 1; j < Nx - 1; ++j) {
            const double pe = clipl(phi[(size_t)i * Nx + j + 1], lim), pw = clipl(phi[(size_t)i * Nx + j - 1], lim);
            const double pn = clipl(phi[(size_t)(i + 1) * Nx + j], lim), ps = clipl(phi[(size_t)(i - 1) * Nx + j], lim);
            const double pc = clipl(clipl(phi[(size_t)i * Nx + j], lim), lim);
            const double gx = (pe - pw) / (2 * P->dx), gy = (pn - ps) / (2 * P->dy);
            const double ux = gx / (P->mu * pc), uy = gy / (P->mu * pc);
            const double norm = sqrt(ux * ux + uy * uy);
            const double den = norm * (norm > lim) + (norm < lim);
            const size_t o = (size_t)(i - 1) * (Nx - 2) + (j - 1);
            vx[o] = (ux * (norm > lim)) / den;
            vy[o] = (uy * (norm > lim)) / den;
        }
}

/* ---- one stage combination y + h * sum_j a_j K_j (scipy rk.py:64-66), used to time the RK45 arithmetic */
void occ_lincomb(const double* y, const double* const* K, const double* c, int n_k, size_t n, double* out) {
#pragma omp parallel for schedule(static)
    for (size_t q = 0; q < n; ++q) {
        double acc = 0.0;
        for (int j = 0; j < n_k; ++j) acc += K[j][q] * c[j];
        out[q] = y[q] + acc;
    }
}

/* ---- pair repulsion of j on i, pedestrians.py:216-280 (trig form, as written in the reference) */
static void pair_force(const occ_params* P, double xi, double yi, double vxi, double vyi, double vdes_i, double xj,
                       double yj, double vxj, double vyj, double* fx, double* fy) {
    const double si = sqrt(vxi * vxi + vyi * vyi), sj = sqrt(vxj * vxj + vyj * vyj);
    const double a_i = P->a_min + P->tau_a * si, b_i = P->b_max - (P->b_max - P->b_min) * fmin(si / vdes_i, 1.0);
    const double a_j = P->a_min + P->tau_a * sj, b_j = P->b_max - (P->b_max - P->b_min) * fmin(sj / vdes_i, 1.0);
    const double Rx = xj - xi, Ry = yj - yi;
    const double r = sqrt(Rx * Rx + Ry * Ry);
    const double ex = Rx / r, ey = Ry / r;
    const double d = (vxj - vxi) * (-ex) + (vyj - vyi) * (-ey);
    const double v_rel = 0.5 * (d + fabs(d));
    const double cv = cos(0.7 * M_PI);
    double k = 0.0;
    if (si > 0) k = fmax((vxi * ex + vyi * ey) / si - cv, 0.0) / (1 - cv);
    const double al_i = atan2(Ry, Rx), be_i = atan2(vyi, vxi), al_j = atan2(-Ry, -Rx), be_j = atan2(vyj, vxj);
    const double ci = cos(al_i - be_i) / a_i, sn = sin(al_i - be_i) / b_i;
    const double cj = cos(al_j - be_j) / a_j, sj2 = sin(al_j - be_j) / b_j;
    const double q_i = sqrt(1 / (ci * ci + sn * sn)), q_j = sqrt(1 / (cj * cj + sj2 * sj2));
    const double dist = r - q_i - q_j;
    const double rep = fmin(k * exp(-dist / (P->eta * (1 + v_rel))), 1.0);
    *fx = -rep * Rx;
    *fy = -rep * Ry;
}

void occ_pair_forces(const occ_params* P, int n, const double* pi, const double* vi, const double* vdes,
                     const double* pj, const double* vj, double* f) {
    for (int q = 0; q < n; ++q)
        pair_force(P, pi[2 * q], pi[2 * q + 1], vi[2 * q], vi[2 * q + 1], vdes[q], pj[2 * q], pj[2 * q + 1], vj[2 * q],
                   vj[2 * q + 1], &f[2 * q], &f[2 * q + 1]);
}

/* ---- nearest "wall" cell: the reference's full-grid argmin (pedestrians.py:311-313), evaluated exactly by the row
 *      candidate scan of oracle WallMap (left/right: nearest site column <= jx / >= jx+1 per row). */
void occ_wallmap_build(int Ny, int Nx, const double* V, double vmin, int32_t* left, int32_t* right) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < Ny; ++r) {
        int last = -1;
        for (int j = 0; j < Nx; ++j) {
            if (V[(size_t)r * Nx + j] == vmin) last = j;
            left[(size_t)r * Nx + j] = last;
        }
        last = Nx;
        for (int j = Nx - 1; j >= 0; --j) {
            right[(size_t)r * Nx + j] = last;      /* nearest site at a column >= j+1 */
            if (V[(size_t)r * Nx + j] == vmin) last = j;
        }
    }
}

static long long wall_query(int Ny, int Nx, const double* X0, const double* Y0, const int32_t* left,
                            const int32_t* right, double off, double px, double py) {
    int jx = 0;
    while (jx < Nx - 1 && X0[jx + 1] <= px) ++jx;      /* X0[jx] <= px < X0[jx+1], clipped */
    double best = INFINITY;
    long long bi = -1;
    for (int r = 0; r < Ny; ++r) {
        const double dy = Y0[r] - py;
        for (int side = 0; side < 2; ++side) {
            const int c = side == 0 ? left[(size_t)r * Nx + jx] : right[(size_t)r * Nx + jx];
            if (c < 0 || c >= Nx) continue;
            const double dxx = X0[c] - px;
            const double f = sqrt(dxx * dxx + dy * dy) + off;
            const long long flat = (long long)r * Nx + c;
            if (f < best || (f == best && flat < bi)) {
                best = f;
                bi = flat;
            }
        }
    }
    return bi;
}

void occ_wall_indices(int Ny, int Nx, const double* X0, const double* Y0, const int32_t* left, const int32_t* right,
                      double off, int n, const double* px, const double* py, int64_t* out) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int q = 0; q < n; ++q) out[q] = wall_query(Ny, Nx, X0, Y0, left, right, off, px[q], py[q]);
}

/* wall force, pedestrians.py:317-334 */
static void wall_force(const occ_params* P, long long flat, const double* X0, const double* Y0, double px, double py,
                       double vxd, double vyd, double vdes, double* fx, double* fy) {
    const int r = (int)(flat / P->Nx), c = (int)(flat % P->Nx);
    const double Rx = X0[c] - px, Ry = Y0[r] - py;
    const double rr = sqrt(Rx * Rx + Ry * Ry);
    const double ex = Rx / rr, ey = Ry / rr;
    const double d = vxd * ex + vyd * ey;
    const double v_rel = 0.5 * (d + fabs(d));
    const double al = atan2(Ry, Rx), be = atan2(vyd, vxd);
    const double s = sqrt(vxd * vxd + vyd * vyd);
    const double a_i = P->a_min + P->tau_a * s, b_i = P->b_max - (P->b_max - P->b_min) * fmin(s / vdes, 1.0);
    const double ca = cos(al - be) / a_i, sb = sin(al - be) / b_i;
    const double q_i = 1 / (ca * ca + sb * sb);
    const double dist = rr - q_i;
    const double rep = fmin(exp(-dist / (P->eta_walls * (1 + v_rel))), 1.0);
    *fx = -3 * rep * Rx;
    *fy = -3 * rep * Ry;
}

/* Python float floor division (CPython float_floor_div) */
static double py_floordiv(double vx, double wx) {
    double mod = fmod(vx, wx), div = (vx - mod) / wx;
    if (mod != 0.0 && ((wx < 0) != (mod < 0))) div -= 1.0;
    if (div != 0.0) {
        double fl = floor(div);
        if (div - fl > 0.5) fl += 1.0;
        return fl;
    }
    return copysign(0.0, vx / wx);
}

/* ---- one synchronous (frozen-snapshot) step for agents i0 <= i < i1 against all N: simulations.py:277-332.
 * field: vx_opt/vy_opt slice of the current step (Ny-2, Nx-2) per key (fields_x[key], fields_y[key]) or NULL = zero.
 * rects: (n_keys, max_rect, 4) exit rectangles, nrect per key.  noise: (N,2).  Outputs (N) arrays (only [i0,i1) set). */
void occ_step_sync(const occ_params* P, int N, int i0, int i1, const double* x, const double* y, const double* vx,
                   const double* vy, const double* vdes, const uint8_t* status, const int32_t* key,
                   const double* const* fields_x, const double* const* fields_y, const double* X0, const double* Y0,
                   const int32_t* const* lefts, const int32_t* const* rights, const double* offs, const double* rects,
                   const int32_t* nrect, int max_rect, const double* noise, double* nx, double* ny, double* nvx,
                   double* nvy, uint8_t* nstatus) {
    const int Nx = P->Nx, Ny = P->Ny;
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = i0; i < i1; ++i) {
        nx[i] = x[i]; ny[i] = y[i]; nvx[i] = vx[i]; nvy[i] = vy[i]; nstatus[i] = status[i];
        if (!status[i]) continue;
        const int kk = key[i];
        /* lookup, optimals.py:212-250 */
        double desx = 0, desy = 0;
        if (fields_x && fields_x[kk]) {
            int j0, j1, r0, r1;
            if (x[i] < P->L - P->dx) { j0 = (int)py_floordiv(x[i], P->dx); j1 = (x[i] > P->dx) ? j0 + 1 : j0; }
            else j0 = j1 = Nx - 3;
            if (y[i] < P->H - P->dy) { r0 = (int)py_floordiv(y[i], P->dy); r1 = (y[i] > P->dy) ? r0 + 1 : r0; }
            else r0 = r1 = Ny - 3;
            if (j0 < 0) j0 += Nx - 2;
            if (j1 < 0) j1 += Nx - 2;
            if (r0 < 0) r0 += Ny - 2;
            if (r1 < 0) r1 += Ny - 2;
            const size_t W = (size_t)(Nx - 2);
            desx = vdes[i] * ((fields_x[kk][(size_t)r0 * W + j0] + fields_x[kk][(size_t)r1 * W + j1]) / 2);
            desy = vdes[i] * ((fields_y[kk][(size_t)r0 * W + j0] + fields_y[kk][(size_t)r1 * W + j1]) / 2);
        }
        /* pair repulsion, ascending j (simulations.py:285-295) */
        double rx = 0, ry = 0;
        for (int j = 0; j < N; ++j) {
            if (!status[j] || j == i) continue;
            const double ddx = x[j] - x[i], ddy = y[j] - y[i];
            if (!(sqrt(ddx * ddx + ddy * ddy) < P->cutoff)) continue;
            double fx, fy;
            pair_force(P, x[i], y[i], vx[i], vy[i], vdes[i], x[j], y[j], vx[j], vy[j], &fx, &fy);
            rx += fx;
            ry += fy;
        }
        /* wall repulsion */
        double wx = 0, wy = 0;
        const long long flat = wall_query(Ny, Nx, X0, Y0, lefts[kk], rights[kk], offs[kk], x[i], y[i]);
        if (flat >= 0) wall_force(P, flat, X0, Y0, x[i], y[i], vx[i], vy[i], vdes[i], &wx, &wy);
        /* simulations.py:303-326 */
        const double cvx = vx[i] + P->noise_amp * noise[2 * i] + rx + wx;
        const double cvy = vy[i] + P->noise_amp * noise[2 * i + 1] + ry + wy;
        const double ax = (desx - cvx) / P->relaxation, ay = (desy - cvy) / P->relaxation;
        const double dt = P->dt;
        nx[i] = x[i] + cvx * dt + 0.5 * ax * (dt * dt);
        ny[i] = y[i] + cvy * dt + 0.5 * ay * (dt * dt);
        double ux = cvx + ax * dt, uy = cvy + ay * dt;
        const double nrm = sqrt(ux * ux + uy * uy);
        if (!(nrm < P->v_max)) {
            ux *= P->v_max / nrm;
            uy *= P->v_max / nrm;
        }
        nvx[i] = ux;
        nvy[i] = uy;
        for (int q = 0; q < nrect[kk]; ++q) {
            const double* d = rects + ((size_t)kk * max_rect + q) * 4;
            if (fabs(nx[i] - d[0]) < d[2] * 0.5 && fabs(ny[i] - d[1]) < d[3] * 0.5) nstatus[i] = 0;
        }
    }
}
