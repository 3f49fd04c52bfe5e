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

/* ---- the whole HJB solve: solve_ivp(hjb, (T, 0), ones, 'RK45', t_eval = linspace(T, 0, nt)), optimals.py:193-196, with
 * SciPy 1.15.3's Dormand-Prince arithmetic restated: rk_step scipy rk.py:14-72, controller :111-176, tables :377-404,
 * dense output :552-575, RMS norm common.py:63-66, initial step :68-134, reversed t_eval slicing ivp.py:598-616,706-723.
 * Same algorithm as oracle/oc_oracle.py rk45_solve (which is pinned against SciPy itself and fixture G3); this copy exists
 * so that a BASELINE-size grid (2048 x 2048) can be solved on the GPU box's host cores in about a minute.
 *   keep_idx[n_keep] : t_eval indices k whose phi(t_eval[k]) goes to phi_keep (n_keep, Ny*Nx).
 *   forced[n_forced] : optional.  forced[0] = T, then the end times of the steps to take: the controller is bypassed and
 *                      every step is taken as given; err_out[s] receives the error norm the controller would have seen
 *                      for step s (< 1 means "this solver would have accepted that step too").  Used to compare two
 *                      adaptive solves on ONE step sequence, so that float32 error is separated from controller noise.
 *   stats_out[4]     : nfev, attempts, accepted, (unused); t_steps[t_cap] accepted step end times (t_steps[0] = T). */
static const double RKA[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
static const double RKB[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
static const double RKE[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
static const double RKP[7][4] = {
    {1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
    {0, 0, 0, 0},
    {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
    {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
    {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408, 701980252875.0 / 199316789632},
    {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
    {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};

static double sumsq_scaled(const double* a, const double* b, const double* y, double atol, double rtol, size_t n) {
    double s = 0.0;                                  /* sum ((a - b) / (atol + |y| rtol))^2, b may be NULL */
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (size_t q = 0; q < n; ++q) {
        const double v = (b ? a[q] - b[q] : a[q]) / (atol + fabs(y[q]) * rtol);
        s += v * v;
    }
    return s;
}

/* SciPy's quartic dense output at one t_eval point (rk.py:178-180, 560-575): out = y_old + h * (K^T P) [x, x^2, x^3, x^4],
 * x = (t_eval - t_old) / h.  K: the step's seven stage derivatives (n values each).  Exported so that the CPU baseline can
 * time it on its own (bench.py), not as the difference of two solves. */
void occ_dense_output(const double* y, const double* const* K, double hh, double x, size_t n, double* out) {
    double w[7];
    for (int j = 0; j < 7; ++j) {
        double xp = x, acc = 0.0;
        for (int p = 0; p < 4; ++p) { acc += RKP[j][p] * xp; xp *= x; }
        w[j] = acc;
    }
#pragma omp parallel for schedule(static)
    for (size_t q = 0; q < n; ++q) {
        double acc = 0.0;
        for (int j = 0; j < 7; ++j) acc += K[j][q] * w[j];
        out[q] = hh * acc + y[q];
    }
}

int occ_rk45_solve(const occ_params* P, const double* V, const double* m, double T, int nt, double rtol, double atol,
                   const int32_t* keep_idx, int n_keep, double* phi_keep, const double* forced, int n_forced,
                   double* err_out, double* t_steps, int t_cap, int64_t* stats_out) {
    const size_t n = (size_t)P->Ny * P->Nx;
    double* buf = (double*)malloc(sizeof(double) * n * 10);
    if (!buf) return -1;
    double *y = buf, *ynew = buf + n, *ytmp = buf + 2 * n, *K[7];
    for (int k = 0; k < 7; ++k) K[k] = buf + (3 + k) * n;
    int64_t nfev = 0, n_att = 0, n_acc = 0;
    for (size_t q = 0; q < n; ++q) y[q] = 1.0;
    const double t0 = T, tf = 0.0, direction = -1.0;
    double t = t0;
    occ_hjb_rhs(P, y, V, m, K[6]);                   /* f0 in the FSAL slot */
    ++nfev;
    double h_abs;
    {   /* select_initial_step, order 4 */
        const double interval = fabs(tf - t0);
        const double d0 = sqrt(sumsq_scaled(y, NULL, y, atol, rtol, n)) / sqrt((double)n);
        const double d1 = sqrt(sumsq_scaled(K[6], NULL, y, atol, rtol, n)) / sqrt((double)n);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        if (h0 > interval) h0 = interval;
#pragma omp parallel for schedule(static)
        for (size_t q = 0; q < n; ++q) ytmp[q] = y[q] + h0 * direction * K[6][q];
        occ_hjb_rhs(P, ytmp, V, m, K[0]);
        ++nfev;
        const double d2 = sqrt(sumsq_scaled(K[0], K[6], y, atol, rtol, n)) / sqrt((double)n) / h0;
        double h1;
        if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
        else h1 = pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
        h_abs = fmin(fmin(100 * h0, h1), interval);
    }
    int n_rec = 0, k_next = 0, fstep = 0;
    if (t_steps && t_cap > 0) t_steps[n_rec++] = t;
    int* slot = (int*)malloc(sizeof(int) * (size_t)(nt > 0 ? nt : 1));
    for (int k = 0; k < nt; ++k) slot[k] = -1;
    for (int q = 0; q < n_keep; ++q)
        if (keep_idx[q] >= 0 && keep_idx[q] < nt) slot[keep_idx[q]] = q;
    const double tstep = (nt > 1) ? (0.0 - T) / (double)(nt - 1) : 0.0;     /* np.linspace(T, 0, nt) */
    int rc = 0;
    while (t != tf && nt > 0) {
        double min_step = 10 * fabs(nextafter(t, direction * INFINITY) - t);
        if (h_abs < min_step) h_abs = min_step;
        int accepted = 0, rejected = 0;
        double t_new = t, hh = 0;
        double* sw = K[0]; K[0] = K[6]; K[6] = sw;  /* FSAL */
        while (!accepted) {
            if (forced) {
                if (fstep + 1 >= n_forced) { rc = -2; goto done; }
                t_new = forced[fstep + 1];
                hh = t_new - t;
                h_abs = fabs(hh);
            } else {
                if (h_abs < min_step) { rc = -3; goto done; }
                hh = h_abs * direction;
                t_new = t + hh;
                if (direction * (t_new - tf) > 0) t_new = tf;
                hh = t_new - t;
                h_abs = fabs(hh);
            }
            ++n_att;
            for (int s = 1; s < 6; ++s) {
#pragma omp parallel for schedule(static)
                for (size_t q = 0; q < n; ++q) {
                    double acc = 0.0;
                    for (int j = 0; j < s; ++j) acc += K[j][q] * RKA[s][j];
                    ytmp[q] = y[q] + acc * hh;       /* dy = np.dot(K[:s].T, a[:s]) * h  (rk.py:64-66) */
                }
                occ_hjb_rhs(P, ytmp, V, m, K[s]);
                ++nfev;
            }
#pragma omp parallel for schedule(static)
            for (size_t q = 0; q < n; ++q) {
                double acc = 0.0;
                for (int j = 0; j < 6; ++j) acc += K[j][q] * RKB[j];
                ynew[q] = y[q] + hh * acc;           /* rk.py:68 */
            }
            occ_hjb_rhs(P, ynew, V, m, K[6]);
            ++nfev;
            double es = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : es)
            for (size_t q = 0; q < n; ++q) {
                double acc = 0.0;
                for (int j = 0; j < 7; ++j) acc += K[j][q] * RKE[j];
                const double v = acc * hh / (atol + fmax(fabs(y[q]), fabs(ynew[q])) * rtol);
                es += v * v;
            }
            const double err = sqrt(es) / sqrt((double)n);
            if (forced) {
                if (err_out) err_out[fstep] = err;
                ++fstep;
                accepted = 1;
            } else if (err < 1) {
                double factor = (err == 0) ? 10.0 : fmin(10.0, 0.9 * pow(err, -0.2));
                if (rejected) factor = fmin(1.0, factor);
                h_abs *= factor;
                accepted = 1;
            } else {
                h_abs *= fmax(0.2, 0.9 * pow(err, -0.2));
                rejected = 1;
            }
        }
        /* dense output for the t_eval points in (t_new, t]  (ivp.py:706-723, rk.py:560-575) */
        while (k_next < nt) {
            double tek = (double)k_next * tstep + T;
            if (k_next == nt - 1) tek = 0.0;
            if (!(tek >= t_new)) break;
            const int k = k_next++;
            if (slot[k] < 0) continue;
            occ_dense_output(y, (const double* const*)K, hh, (tek - t) / hh, n, phi_keep + (size_t)slot[k] * n);
        }
        sw = y; y = ynew; ynew = sw;
        t = t_new;
        ++n_acc;
        if (t_steps && n_rec < t_cap) t_steps[n_rec++] = t;
        if (direction * (t - tf) >= 0) break;
    }
done:
    if (stats_out) { stats_out[0] = nfev; stats_out[1] = n_att; stats_out[2] = n_acc; stats_out[3] = n_rec; }
    free(slot);
    free(buf);
    return rc;
}
