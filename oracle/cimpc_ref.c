/*
 * cimpc_ref.c - single-thread C restatement of ContactImplicitMPC.jl's per-MPC-step
 * solver path.  TEST INFRASTRUCTURE / CPU BASELINE ONLY: nothing in the product
 * (contactimplicitmpc/jl_amd, include/, libcimpc_hip.so) links or calls this file.
 * It is validated against the numpy oracle (tests/test_oracle_c_port.py) and timed by
 * bench.py's `cpu_baseline` leg (kind = "port": a restatement of the reference
 * algorithm, not the Julia package, which cannot run here).
 *
 * Follows (paths relative to /root/reference):
 *   src/solver/qr.jl:113-158             MGS-QR, left-looking, true divisions
 *   src/solver/schur.jl:33-49,80-110     Schur complement about Dx
 *   src/controller/linearized_solver.jl  rlin! :364-373, rzlin! :378-399,
 *                                        linear_solve! :424-444 and :451-479
 *   src/controller/implicit_dynamics.jl:156-192
 *   src/controller/newton_residual.jl:113-176, newton_jacobian.jl:148-198
 *   src/controller/newton.jl:130-288     reset!, newton_solve!
 *   src/solver/lu.jl:4-12                default KKT backend: dense LU of Array(R)
 *   src/controller/newton_structure_solver/methods.jl:386-557  condensed backend
 * IP iteration control: build-defined spec (RoboDojo 0.1.3 owns it; parity unpinned),
 * identical to oracle/ip.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int nq, nu, nw, nc, nb, mode, H_ref, H;
    int nx, ny, nz, nth, nths, nd, nr, N;
    /* per knot tables */
    double *Dx, *Dy1, *Rx, *Ry1, *Ry2, *rthdyn, *rthrst, *rdyn0, *rrst0, *x0, *y10, *y20, *th0;
    double *Ai, *CAi, *CAiB;
    /* objective */
    double *Q, *R, *Qinv, *Rinv;
    /* options */
    double r_tol, kappa_tol, undercut, gamma_reg, kappa_reg, eps_min, ls_scale;
    int ip_max_iter, max_ls;
    double stall_alpha;
    double n_r_tol, beta_init, kappa;
    int n_max_iter;
    /* IP workspace */
    double *qs, *rs, *D, *xv, *tmp, *tmp2, *u, *v;
    /* arbiter mode (tests): theta of EVERY interior-point solve is perturbed in the last place (theta * (1 +- 2^-52)), so that
     * a re-run exercises the round-off sensitivity of every solve of a Newton solve, not only of the ones that read its inputs */
    unsigned long long noise;
} Ref;

static double *dalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

static int invert(const double *A, double *Ai, int n) { /* Gauss-Jordan, partial pivoting, col-major */
    double *M = dalloc((size_t)n * 2 * n);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            M[r * 2 * n + c] = A[r + c * n];
            M[r * 2 * n + n + c] = (r == c);
        }
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int r = k + 1; r < n; ++r)
            if (fabs(M[r * 2 * n + k]) > fabs(M[p * 2 * n + k])) p = r;
        if (M[p * 2 * n + k] == 0.0) { free(M); return -1; }
        if (p != k)
            for (int c = 0; c < 2 * n; ++c) { double t = M[p * 2 * n + c]; M[p * 2 * n + c] = M[k * 2 * n + c]; M[k * 2 * n + c] = t; }
        double piv = M[k * 2 * n + k];
        for (int c = 0; c < 2 * n; ++c) M[k * 2 * n + c] /= piv;
        for (int r = 0; r < n; ++r) {
            if (r == k) continue;
            double f = M[r * 2 * n + k];
            if (f == 0.0) continue;
            for (int c = 0; c < 2 * n; ++c) M[r * 2 * n + c] -= f * M[k * 2 * n + c];
        }
    }
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) Ai[r + c * n] = M[r * 2 * n + n + c];
    free(M);
    return 0;
}

Ref *ref_create(int nq, int nu, int nw, int nc, int nb, int mode, int H_ref, int H) {
    Ref *s = (Ref *)calloc(1, sizeof(Ref));
    s->nq = nq; s->nu = nu; s->nw = nw; s->nc = nc; s->nb = nb; s->mode = mode; s->H_ref = H_ref; s->H = H;
    s->nx = nq; s->ny = 2 * nc + nb; s->nz = nq + 4 * nc + 2 * nb; s->nth = 2 * nq + nu + nw + 2;
    s->nths = 2 * nq + nu;
    s->nd = mode ? nq + nc + nb : nq;
    s->nr = mode ? nq + nu + nc + nb : nq + nu;
    s->N = H * (s->nr + s->nd);
    int nx = s->nx, ny = s->ny, nth = s->nth, K = H_ref;
    s->Dx = dalloc((size_t)K * nx * nx); s->Dy1 = dalloc((size_t)K * nx * ny); s->Rx = dalloc((size_t)K * ny * nx);
    s->Ry1 = dalloc((size_t)K * ny * ny); s->Ry2 = dalloc((size_t)K * ny);
    s->rthdyn = dalloc((size_t)K * nx * nth); s->rthrst = dalloc((size_t)K * ny * nth);
    s->rdyn0 = dalloc((size_t)K * nx); s->rrst0 = dalloc((size_t)K * ny);
    s->x0 = dalloc((size_t)K * nx); s->y10 = dalloc((size_t)K * ny); s->y20 = dalloc((size_t)K * ny); s->th0 = dalloc((size_t)K * nth);
    s->Ai = dalloc((size_t)K * nx * nx); s->CAi = dalloc((size_t)K * ny * nx); s->CAiB = dalloc((size_t)K * ny * ny);
    s->Q = dalloc((size_t)H * nq * nq); s->R = dalloc((size_t)H * nu * nu);
    s->Qinv = dalloc((size_t)H * nq * nq); s->Rinv = dalloc((size_t)H * nu * nu);
    s->qs = dalloc((size_t)ny * ny); s->rs = dalloc((size_t)ny * (ny + 1) / 2); s->D = dalloc((size_t)ny * ny);
    s->xv = dalloc(ny); s->tmp = dalloc(nx + ny + nth); s->tmp2 = dalloc(nx + ny + nth); s->u = dalloc(nx); s->v = dalloc(ny);
    s->r_tol = 1e-8; s->kappa_tol = 2e-4; s->undercut = 5; s->gamma_reg = 0.1; s->kappa_reg = 1e-3;
    s->eps_min = 0.05; s->ls_scale = 0.5; s->ip_max_iter = 100; s->max_ls = 3; s->stall_alpha = 1e-13;
    s->n_r_tol = 3e-4; s->beta_init = 1e-5; s->kappa = 2e-4; s->n_max_iter = 5;
    return s;
}

void ref_set_stall(Ref *s, double stall_alpha) { s->stall_alpha = stall_alpha; }
void ref_set_noise(Ref *s, unsigned long long seed) { s->noise = seed; } /* 0 = off */
static inline unsigned long long noise_next(Ref *s) { /* xorshift64 */
    unsigned long long x = s->noise; x ^= x << 13; x ^= x >> 7; x ^= x << 17; s->noise = x ? x : 0x9E3779B97F4A7C15ull; return x; }

void ref_set_opts(Ref *s, double r_tol, double kappa_tol, double undercut, double gamma_reg, double kappa_reg,
                  double eps_min, double ls_scale, int ip_max_iter, int max_ls, double n_r_tol,
                  double beta_init, double kappa, int n_max_iter) {
    s->r_tol = r_tol; s->kappa_tol = kappa_tol; s->undercut = undercut; s->gamma_reg = gamma_reg;
    s->kappa_reg = kappa_reg; s->eps_min = eps_min; s->ls_scale = ls_scale; s->ip_max_iter = ip_max_iter;
    s->max_ls = max_ls; s->n_r_tol = n_r_tol; s->beta_init = beta_init; s->kappa = kappa; s->n_max_iter = n_max_iter;
}

void ref_destroy(Ref *s) {
    if (!s) return;
    double *p[] = {s->Dx, s->Dy1, s->Rx, s->Ry1, s->Ry2, s->rthdyn, s->rthrst, s->rdyn0, s->rrst0, s->x0, s->y10,
                   s->y20, s->th0, s->Ai, s->CAi, s->CAiB, s->Q, s->R, s->Qinv, s->Rinv, s->qs, s->rs, s->D,
                   s->xv, s->tmp, s->tmp2, s->u, s->v};
    for (size_t i = 0; i < sizeof(p) / sizeof(p[0]); ++i) free(p[i]);
    free(s);
}

/* A1: RLin / RZLin / RthLin constructors (linearized_solver.jl:67-161,224-304,325-359).
 * rz0 nz x nz, rth0 nz x nth, column-major; t is 0-based. */
int ref_set_linearization(Ref *s, int t, const double *z0, const double *th0, const double *r0,
                          const double *rz0, const double *rth0) {
    int nx = s->nx, ny = s->ny, nz = s->nz, nth = s->nth;
    double *Dx = s->Dx + (size_t)t * nx * nx, *Dy1 = s->Dy1 + (size_t)t * nx * ny, *Rx = s->Rx + (size_t)t * ny * nx;
    double *Ry1 = s->Ry1 + (size_t)t * ny * ny, *Ry2 = s->Ry2 + (size_t)t * ny;
    for (int c = 0; c < nx; ++c) {
        for (int r = 0; r < nx; ++r) Dx[r + c * nx] = rz0[r + (size_t)c * nz];
        for (int r = 0; r < ny; ++r) Rx[r + c * ny] = rz0[nx + r + (size_t)c * nz];
    }
    for (int c = 0; c < ny; ++c) {
        for (int r = 0; r < nx; ++r) Dy1[r + c * nx] = rz0[r + (size_t)(nx + c) * nz];
        for (int r = 0; r < ny; ++r) Ry1[r + c * ny] = rz0[nx + r + (size_t)(nx + c) * nz];
        Ry2[c] = rz0[nx + c + (size_t)(nx + ny + c) * nz];
    }
    for (int c = 0; c < nth; ++c) {
        for (int r = 0; r < nx; ++r) s->rthdyn[(size_t)t * nx * nth + r + c * nx] = rth0[r + (size_t)c * nz];
        for (int r = 0; r < ny; ++r) s->rthrst[(size_t)t * ny * nth + r + c * ny] = rth0[nx + r + (size_t)c * nz];
    }
    memcpy(s->rdyn0 + (size_t)t * nx, r0, nx * sizeof(double));
    memcpy(s->rrst0 + (size_t)t * ny, r0 + nx, ny * sizeof(double));
    memcpy(s->x0 + (size_t)t * nx, z0, nx * sizeof(double));
    memcpy(s->y10 + (size_t)t * ny, z0 + nx, ny * sizeof(double));
    memcpy(s->y20 + (size_t)t * ny, z0 + nx + ny, ny * sizeof(double));
    memcpy(s->th0 + (size_t)t * nth, th0, nth * sizeof(double));
    double *Ai = s->Ai + (size_t)t * nx * nx, *CAi = s->CAi + (size_t)t * ny * nx, *CAiB = s->CAiB + (size_t)t * ny * ny;
    if (invert(Dx, Ai, nx)) return -1;                                   /* schur.jl:39 */
    for (int r = 0; r < ny; ++r)
        for (int c = 0; c < nx; ++c) {                                   /* schur.jl:40 */
            double a = 0;
            for (int k = 0; k < nx; ++k) a += Rx[r + k * ny] * Ai[k + c * nx];
            CAi[r + c * ny] = a;
        }
    for (int r = 0; r < ny; ++r)
        for (int c = 0; c < ny; ++c) {                                   /* schur.jl:41 */
            double a = 0;
            for (int k = 0; k < nx; ++k) a += CAi[r + k * ny] * Dy1[k + c * nx];
            CAiB[r + c * ny] = a;
        }
    return 0;
}

int ref_set_objective(Ref *s, const double *Q, const double *R) { /* H x (col-major blocks) */
    memcpy(s->Q, Q, (size_t)s->H * s->nq * s->nq * sizeof(double));
    memcpy(s->R, R, (size_t)s->H * s->nu * s->nu * sizeof(double));
    for (int i = 0; i < s->H; ++i) {
        if (invert(Q + (size_t)i * s->nq * s->nq, s->Qinv + (size_t)i * s->nq * s->nq, s->nq)) return -1;
        if (s->nu && invert(R + (size_t)i * s->nu * s->nu, s->Rinv + (size_t)i * s->nu * s->nu, s->nu)) return -1;
    }
    return 0;
}

/* ---- qr.jl ---------------------------------------------------------------------------- */
static void mgs_factorize(Ref *s, const double *A /* ny x ny col-major */) {
    int n = s->ny, off = 0;
    double *qs = s->qs, *rs = s->rs;
    for (int j = 0; j < n; ++j) {
        double *q = qs + (size_t)j * n;
        memcpy(q, A + (size_t)j * n, n * sizeof(double));
        for (int k = 0; k < j; ++k) {
            const double *qk = qs + (size_t)k * n;
            double d = 0;
            for (int i = 0; i < n; ++i) d += q[i] * qk[i];
            rs[off] = d;
            for (int i = 0; i < n; ++i) q[i] -= qk[i] * d;
            off++;
        }
        double nn = 0;
        for (int i = 0; i < n; ++i) nn += q[i] * q[i];
        rs[off] = sqrt(nn);
        for (int i = 0; i < n; ++i) q[i] /= rs[off];
        off++;
    }
}
static inline int triu(int k, int j) { return (j * (j + 1)) / 2 + k; } /* 0-based R[k,j], k<=j */
static void qr_solve(Ref *s, const double *b, double *x) {
    int n = s->ny;
    for (int j = 0; j < n; ++j) {
        const double *q = s->qs + (size_t)j * n;
        double d = 0;
        for (int i = 0; i < n; ++i) d += q[i] * b[i];
        x[j] = d;
    }
    for (int j = n - 1; j >= 0; --j) {
        for (int k = j + 1; k < n; ++k) x[j] -= s->rs[triu(j, k)] * x[k];
        x[j] /= s->rs[triu(j, j)];
    }
}

/* ---- linearized_solver.jl --------------------------------------------------------------- */
typedef struct { double *rdyn, *rrst, *rbil; } Res;

static void rlin(Ref *s, int t, const double *z, const double *th, double kappa, const double *alt, Res r) {
    int nx = s->nx, ny = s->ny, nth = s->nth;
    const double *Dx = s->Dx + (size_t)t * nx * nx, *Dy1 = s->Dy1 + (size_t)t * nx * ny, *Rx = s->Rx + (size_t)t * ny * nx;
    const double *Ry1 = s->Ry1 + (size_t)t * ny * ny, *Ry2 = s->Ry2 + (size_t)t * ny;
    const double *Td = s->rthdyn + (size_t)t * nx * nth, *Tr = s->rthrst + (size_t)t * ny * nth;
    const double *x0 = s->x0 + (size_t)t * nx, *y10 = s->y10 + (size_t)t * ny, *y20 = s->y20 + (size_t)t * ny, *th0 = s->th0 + (size_t)t * nth;
    double *dx = s->tmp, *dy1 = s->tmp + nx, *dth = s->tmp + nx + ny;
    for (int i = 0; i < nx; ++i) dx[i] = z[i] - x0[i];
    for (int i = 0; i < ny; ++i) dy1[i] = z[nx + i] - y10[i];
    for (int i = 0; i < nth; ++i) dth[i] = th[i] - th0[i];
    for (int i = 0; i < nx; ++i) {
        double a = 0, b = 0, c = 0;
        for (int k = 0; k < nx; ++k) a += Dx[i + k * nx] * dx[k];
        for (int k = 0; k < ny; ++k) b += Dy1[i + k * nx] * dy1[k];
        for (int k = 0; k < nth; ++k) c += Td[i + k * nx] * dth[k];
        r.rdyn[i] = ((s->rdyn0[(size_t)t * nx + i] + a) + b) + c;
    }
    for (int i = 0; i < ny; ++i) {
        double a = 0, b = 0, c = 0;
        for (int k = 0; k < nx; ++k) a += Rx[i + k * ny] * dx[k];
        for (int k = 0; k < ny; ++k) b += Ry1[i + k * ny] * dy1[k];
        for (int k = 0; k < nth; ++k) c += Tr[i + k * ny] * dth[k];
        double al = (alt && i < s->nc) ? alt[i] : 0.0;
        r.rrst[i] = ((((s->rrst0[(size_t)t * ny + i] + a) + b) + Ry2[i] * (z[nx + ny + i] - y20[i])) + c) + al;
        r.rbil[i] = z[nx + i] * z[nx + ny + i] - kappa;
    }
}

static void rzlin(Ref *s, int t, const double *z, double reg) {
    int nx = s->nx, ny = s->ny;
    const double *Ry1 = s->Ry1 + (size_t)t * ny * ny, *Ry2 = s->Ry2 + (size_t)t * ny, *CAiB = s->CAiB + (size_t)t * ny * ny;
    for (int c = 0; c < ny; ++c)
        for (int r = 0; r < ny; ++r) {
            double d = Ry1[r + c * ny];
            if (r == c) {
                double y1r = fmax(z[nx + r], reg), y2r = fmax(z[nx + ny + r], reg);
                d = d - Ry2[r] * y2r / y1r;
            }
            s->D[r + c * ny] = d - CAiB[r + c * ny];
        }
    mgs_factorize(s, s->D);
}

/* schur_solve! : x = Ai*(u + B*temp), y = -temp */
static void schur_solve(Ref *s, int t, const double *u, const double *v, double *x, double *y) {
    int nx = s->nx, ny = s->ny;
    const double *Ai = s->Ai + (size_t)t * nx * nx, *CAi = s->CAi + (size_t)t * ny * nx, *B = s->Dy1 + (size_t)t * nx * ny;
    double *b = s->tmp2, *w = s->tmp2 + ny;
    for (int i = 0; i < ny; ++i) {
        double a = 0;
        for (int k = 0; k < nx; ++k) a += CAi[i + k * ny] * u[k];
        b[i] = a - v[i];
    }
    qr_solve(s, b, s->xv);
    for (int i = 0; i < nx; ++i) {
        double a = 0;
        for (int k = 0; k < ny; ++k) a += B[i + k * nx] * s->xv[k];
        w[i] = u[i] + a;
    }
    for (int i = 0; i < nx; ++i) {
        double a = 0;
        for (int k = 0; k < nx; ++k) a += Ai[i + k * nx] * w[k];
        x[i] = a;
    }
    for (int i = 0; i < ny; ++i) y[i] = -s->xv[i];
}

static void linear_solve(Ref *s, int t, const double *z, Res r, double reg, double *Delta) {
    int nx = s->nx, ny = s->ny;
    const double *Ry2 = s->Ry2 + (size_t)t * ny;
    for (int i = 0; i < ny; ++i) {
        double y1r = fmax(reg, z[nx + i]);
        s->v[i] = r.rrst[i] - Ry2[i] * r.rbil[i] / y1r;
    }
    schur_solve(s, t, r.rdyn, s->v, Delta, Delta + nx);
    for (int i = 0; i < ny; ++i) {
        double y1r = fmax(reg, z[nx + i]), y2r = fmax(reg, z[nx + ny + i]);
        Delta[nx + ny + i] = (r.rbil[i] - y2r * Delta[nx + i]) / y1r;
    }
}

static double step_length(const double *y, const double *dy, int n, double tau, double a) {
    for (int i = 0; i < n; ++i)
        if (dy[i] > 0.0) { double c = tau * y[i] / dy[i]; if (c < a) a = c; }
    return a;
}

/* one interior_point_solve!; dz: nd x nths col-major (written only on success). returns status */
static int ip_solve(Ref *s, int t, double *z, const double *th, const double *alt, double *dz, int *iters_out) {
    int nx = s->nx, ny = s->ny, nz = s->nz;
    double rbuf[3 * 64 + 8];
    double *rb = (nx + 2 * ny <= 3 * 64) ? rbuf : dalloc(nx + 2 * ny);
    Res r = {rb, rb + nx, rb + nx + ny};
    double Dl[256];
    double *Delta = (nz <= 256) ? Dl : dalloc(nz);
    double Daff[256];
    rlin(s, t, z, th, 0.0, alt, r);
    double r_vio = 0, k_vio = 0;
    for (int i = 0; i < nx; ++i) r_vio = fmax(r_vio, fabs(r.rdyn[i]));
    for (int i = 0; i < ny; ++i) { r_vio = fmax(r_vio, fabs(r.rrst[i])); k_vio = fmax(k_vio, fabs(r.rbil[i])); }
    int iters = 0;
    double reg = 0.0;
    for (int j = 0; j < s->ip_max_iter; ++j) {
        if (r_vio < s->r_tol && k_vio < s->kappa_tol) break;
        iters++;
        reg = (k_vio < s->kappa_reg) ? k_vio * s->gamma_reg : 0.0;
        rzlin(s, t, z, reg);
        linear_solve(s, t, z, r, reg, Delta);
        double a_aff = step_length(z + nx, Delta + nx, 2 * ny, 1.0, 1.0);
        double mu = 0, mu_aff = 0;
        for (int i = 0; i < ny; ++i) {
            mu += z[nx + i] * z[nx + ny + i];
            mu_aff += (z[nx + i] - a_aff * Delta[nx + i]) * (z[nx + ny + i] - a_aff * Delta[nx + ny + i]);
        }
        mu /= ny; mu_aff /= ny;
        double sg = fmin(fmax(mu_aff / mu, 0.0), 1.0);
        sg = sg * sg * sg;
        double kc = fmax(sg * mu, s->kappa_tol / s->undercut);
        memcpy(Daff, Delta, nz * sizeof(double));
        rlin(s, t, z, th, kc, alt, r);
        for (int i = 0; i < ny; ++i) r.rbil[i] += Daff[nx + i] * Daff[nx + ny + i];
        linear_solve(s, t, z, r, reg, Delta);
        double vm = fmax(r_vio, k_vio);
        double tau = fmax(1.0 - s->eps_min, 1.0 - vm * vm);
        double alpha = step_length(z + nx, Delta + nx, 2 * ny, tau, 1.0);
        if (alpha < s->stall_alpha) break;                 /* [spec] stall exit */
        for (int i = 0; i < nz; ++i) z[i] -= alpha * Delta[i];
        double k_c = 0, r_c = 0;
        for (int ls = 1; ls <= s->max_ls; ++ls) {
            rlin(s, t, z, th, 0.0, alt, r);
            k_c = 0; r_c = 0;
            for (int i = 0; i < nx; ++i) r_c = fmax(r_c, fabs(r.rdyn[i]));
            for (int i = 0; i < ny; ++i) { r_c = fmax(r_c, fabs(r.rrst[i])); k_c = fmax(k_c, fabs(r.rbil[i])); }
            if (r_c <= r_vio || k_c <= k_vio) break;
            double back = alpha * pow(s->ls_scale, ls);
            for (int i = 0; i < nz; ++i) z[i] += back * Delta[i];
        }
        k_vio = k_c; r_vio = r_c;
    }
    int ok = (r_vio < s->r_tol) && (k_vio < s->kappa_tol);
    *iters_out = iters;
    if (ok && dz) {
        double reg2 = fmax(reg, s->kappa_tol * s->gamma_reg);
        rzlin(s, t, z, reg2);
        int nth = s->nth, nd = s->nd;
        const double *Td = s->rthdyn + (size_t)t * nx * nth, *Tr = s->rthrst + (size_t)t * ny * nth;
        double xs[64], ys[64];
        for (int c = 0; c < s->nths; ++c) {                 /* only the consumed columns */
            schur_solve(s, t, Td + (size_t)c * nx, Tr + (size_t)c * ny, xs, ys);
            for (int i = 0; i < nx; ++i) dz[i + (size_t)c * nd] = -xs[i];
            if (s->mode) for (int i = 0; i < s->nc + s->nb; ++i) dz[nx + i + (size_t)c * nd] = -ys[i];
        }
    }
    if (rb != rbuf) free(rb);
    if (Delta != Dl) free(Delta);
    return ok;
}

/* implicit_dynamics! for one rollout.  window 0-based [H+2]; q [H+2][nq]; theta [H][nth];
 * gam [H][nc], bfr [H][nb] (cf);  outputs d [H][nd], dz [H][nths][nd] (col-major blocks),
 * status/iters [H], zout [H][nz] (may be NULL). */
void ref_implicit_dynamics(Ref *s, const int *window, const double *q, const double *theta, const double *gam,
                           const double *bfr, const double *alt, double *d, double *dz, int *status, int *iters,
                           double *zout) {
    int nq = s->nq, nz = s->nz, nd = s->nd;
    double zb[512];
    for (int i = 0; i < s->H; ++i) {
        int t = window[i];
        double *z = zb;
        for (int k = 0; k < nz; ++k) z[k] = 1.0;
        memcpy(z, q + (size_t)(i + 2) * nq, nq * sizeof(double));
        const double *th_i = theta + (size_t)i * s->nth;
        double thn[256];
        if (s->noise && s->nth <= 256) {
            for (int k = 0; k < s->nth; ++k) thn[k] = th_i[k] * (1.0 + ((noise_next(s) >> 33) & 1 ? 1.0 : -1.0) * 2.220446049250313e-16);
            th_i = thn;
        }
        status[i] = ip_solve(s, t, z, th_i, alt, dz + (size_t)i * s->nths * nd, &iters[i]);
        for (int k = 0; k < nq; ++k) d[(size_t)i * nd + k] = z[k] - q[(size_t)(i + 2) * nq + k];
        if (s->mode) {
            for (int k = 0; k < s->nc; ++k) d[(size_t)i * nd + nq + k] = z[nq + k] - gam[(size_t)i * s->nc + k];
            for (int k = 0; k < s->nb; ++k) d[(size_t)i * nd + nq + s->nc + k] = z[nq + s->nc + k] - bfr[(size_t)i * s->nb + k];
        }
        if (zout) memcpy(zout + (size_t)i * nz, z, nz * sizeof(double));
    }
}

/* ---- Newton layer (:configuration mode, TrackingObjective) ------------------------------ */
typedef struct { double *q, *u, *w, *th; } Tr;

static void update_theta(Ref *s, Tr T) {
    int nq = s->nq, nu = s->nu, nw = s->nw, nth = s->nth;
    for (int i = 0; i < s->H; ++i) {
        double *th = T.th + (size_t)i * nth;
        memcpy(th, T.q + (size_t)i * nq, 2 * nq * sizeof(double));
        memcpy(th + 2 * nq, T.u + (size_t)i * nu, nu * sizeof(double));
        memcpy(th + 2 * nq + nu, T.w + (size_t)i * nw, nw * sizeof(double));
    }
}

static double residual(Ref *s, Tr T, const double *nuv, const double *d, const double *dz, const double *qref,
                       const double *uref, double *r) {
    int H = s->H, nq = s->nq, nu = s->nu, nr = s->nr, nd = s->nd, nths = s->nths;
    memset(r, 0, s->N * sizeof(double));
    for (int t = 0; t < H; ++t) {                                   /* gradient! :203-219 */
        const double *Q = s->Q + (size_t)t * nq * nq, *R = s->R + (size_t)t * nu * nu;
        for (int c = 0; c < nq; ++c) {
            double a = 0;
            for (int k = 0; k < nq; ++k) a += Q[c + k * nq] * (T.q[(size_t)(t + 2) * nq + k] - qref[(size_t)(t + 2) * nq + k]);
            r[t * nr + nu + c] += a;
        }
        for (int c = 0; c < nu; ++c) {
            double a = 0;
            for (int k = 0; k < nu; ++k) a += R[c + k * nu] * (T.u[(size_t)t * nu + k] - uref[(size_t)t * nu + k]);
            r[t * nr + c] += a;
        }
    }
    for (int i = 0; i < H; ++i) {                                   /* residual! :123-136 */
        const double *dzi = dz + (size_t)i * nths * nd, *nv = nuv + (size_t)i * nd;
        if (i >= 2)
            for (int c = 0; c < nq; ++c) { double a = 0; for (int k = 0; k < nd; ++k) a += dzi[k + (size_t)c * nd] * nv[k]; r[(i - 2) * nr + nu + c] += a; }
        if (i >= 1)
            for (int c = 0; c < nq; ++c) { double a = 0; for (int k = 0; k < nd; ++k) a += dzi[k + (size_t)(nq + c) * nd] * nv[k]; r[(i - 1) * nr + nu + c] += a; }
        for (int c = 0; c < nu; ++c) { double a = 0; for (int k = 0; k < nd; ++k) a += dzi[k + (size_t)(2 * nq + c) * nd] * nv[k]; r[i * nr + c] += a; }
        for (int k = 0; k < nd; ++k) r[H * nr + i * nd + k] += d[(size_t)i * nd + k];
        for (int k = 0; k < nq; ++k) r[i * nr + nu + k] -= nv[k];
    }
    double n1 = 0;
    for (int e = 0; e < s->N; ++e) n1 += fabs(r[e]);
    return n1;
}

/* dense jacobian! (newton_jacobian.jl:148-198) + dense LU with partial pivoting (lu.jl:4-12) */
static void kkt_dense_lu(Ref *s, const double *dz, double beta, const double *r, double *Delta, double *Rm, int *piv) {
    int H = s->H, nq = s->nq, nu = s->nu, nr = s->nr, nd = s->nd, nths = s->nths, N = s->N;
    memset(Rm, 0, (size_t)N * N * sizeof(double));
#define RM(i, j) Rm[(size_t)(i) + (size_t)(j) * N]
    for (int t = 0; t < H; ++t) {
        for (int a = 0; a < nq; ++a) for (int b = 0; b < nq; ++b) RM(t * nr + nu + a, t * nr + nu + b) += s->Q[(size_t)t * nq * nq + a + b * nq];
        for (int a = 0; a < nu; ++a) for (int b = 0; b < nu; ++b) RM(t * nr + a, t * nr + b) += s->R[(size_t)t * nu * nu + a + b * nu];
        for (int k = 0; k < nd; ++k) { RM(t * nr + nu + k, H * nr + t * nd + k) -= 1.0; RM(H * nr + t * nd + k, t * nr + nu + k) -= 1.0; }
    }
    for (int i = 0; i < H; ++i) {
        const double *dzi = dz + (size_t)i * nths * nd;
        for (int k = 0; k < nd; ++k) {
            int row = H * nr + i * nd + k;
            if (i >= 2) for (int c = 0; c < nq; ++c) { double v = dzi[k + (size_t)c * nd]; RM(row, (i - 2) * nr + nu + c) += v; RM((i - 2) * nr + nu + c, row) += v; }
            if (i >= 1) for (int c = 0; c < nq; ++c) { double v = dzi[k + (size_t)(nq + c) * nd]; RM(row, (i - 1) * nr + nu + c) += v; RM((i - 1) * nr + nu + c, row) += v; }
            for (int c = 0; c < nu; ++c) { double v = dzi[k + (size_t)(2 * nq + c) * nd]; RM(row, i * nr + c) += v; RM(i * nr + c, row) += v; }
        }
        for (int k = H * nr; k < N; ++k) RM(k, k) -= beta * s->kappa;      /* reg_du quirk :185 */
    }
    /* right-looking LU, partial pivoting, column-major */
    memcpy(Delta, r, N * sizeof(double));
    for (int k = 0; k < N; ++k) {
        int p = k; double mx = fabs(RM(k, k));
        for (int i = k + 1; i < N; ++i) { double v = fabs(RM(i, k)); if (v > mx) { mx = v; p = i; } }
        piv[k] = p;
        if (p != k) {
            for (int j = 0; j < N; ++j) { double tmp = RM(k, j); RM(k, j) = RM(p, j); RM(p, j) = tmp; }
            double tb = Delta[k]; Delta[k] = Delta[p]; Delta[p] = tb;
        }
        double inv = 1.0 / RM(k, k);
        for (int i = k + 1; i < N; ++i) RM(i, k) *= inv;
        for (int j = k + 1; j < N; ++j) {
            double f = RM(k, j);
            if (f == 0.0) continue;
            double *col = &RM(0, j); const double *lk = &RM(0, k);
            for (int i = k + 1; i < N; ++i) col[i] -= lk[i] * f;
        }
    }
    for (int k = 0; k < N; ++k) { double v = Delta[k]; for (int i = k + 1; i < N; ++i) Delta[i] -= RM(i, k) * v; }
    for (int k = N - 1; k >= 0; --k) { Delta[k] /= RM(k, k); double v = Delta[k]; for (int i = 0; i < k; ++i) Delta[i] -= RM(i, k) * v; }
#undef RM
}

/* small dense helpers, column-major */
static void mm_nn(double *C, const double *A, const double *B, int m, int n, int k) {
    for (int c = 0; c < n; ++c) for (int r = 0; r < m; ++r) { double a = 0; for (int j = 0; j < k; ++j) a += A[r + j * m] * B[j + c * k]; C[r + c * m] = a; }
}
static void mm_nt_acc(double *C, const double *A, const double *B, int m, int n, int k, double alpha) { /* C += alpha A B^T, B n x k */
    for (int c = 0; c < n; ++c) for (int r = 0; r < m; ++r) { double a = 0; for (int j = 0; j < k; ++j) a += A[r + j * m] * B[c + j * n]; C[r + c * m] += alpha * a; }
}

/* condensed solve: dual Schur complement + block Cholesky (methods.jl:386-557 applied to the
 * :direct layout), same maths as oracle/newton.py kkt_solve_condensed */
static int kkt_condensed(Ref *s, const double *dz, double beta, const double *r, double *Delta, double *ws) {
    int H = s->H, nq = s->nq, nu = s->nu, nr = s->nr, nd = s->nd, nths = s->nths, n2 = nd * nd;
    double rho = H * beta * s->kappa;
    double *L0 = ws, *L1 = L0 + (size_t)H * n2, *L2 = L1 + (size_t)H * n2, *yv = L2 + (size_t)H * n2, *dn = yv + (size_t)H * nd;
    double *T0 = dn + (size_t)H * nd, *T1 = T0 + nd * nu, *T2 = T1 + nd * nq, *Y0 = T2 + nd * nq, *Y1 = Y0 + n2, *Y2 = Y1 + n2, *bet = Y2 + n2, *tv = bet + nd;
    for (int i = 0; i < H; ++i) {
        const double *A2 = dz + (size_t)i * nths * nd, *A1 = A2 + (size_t)nq * nd, *A0 = A2 + (size_t)2 * nq * nd;
        const double *Qi = s->Qinv + (size_t)i * nq * nq;
        mm_nn(T0, A0, s->Rinv + (size_t)i * nu * nu, nd, nu, nu);
        for (int k = 0; k < n2; ++k) Y0[k] = Qi[k];
        for (int k = 0; k < nd; ++k) Y0[k + k * nd] += rho;
        mm_nt_acc(Y0, T0, A0, nd, nd, nu, 1.0);
        for (int k = 0; k < nd; ++k) {
            double a = 0, b = 0;
            for (int j = 0; j < nu; ++j) a += T0[k + j * nd] * r[i * nr + j];
            for (int j = 0; j < nq; ++j) b += Qi[k + j * nq] * r[i * nr + nu + j];
            bet[k] = a - b - r[H * nr + i * nd + k];
        }
        if (i >= 1) {
            mm_nn(T1, A1, s->Qinv + (size_t)(i - 1) * nq * nq, nd, nq, nq);
            mm_nt_acc(Y0, T1, A1, nd, nd, nq, 1.0);
            for (int k = 0; k < nd; ++k) { double a = 0; for (int j = 0; j < nq; ++j) a += T1[k + j * nd] * r[(i - 1) * nr + nu + j]; bet[k] += a; }
            for (int k = 0; k < n2; ++k) Y1[k] = -T1[k];
        }
        if (i >= 2) {
            mm_nn(T2, A2, s->Qinv + (size_t)(i - 2) * nq * nq, nd, nq, nq);
            mm_nt_acc(Y0, T2, A2, nd, nd, nq, 1.0);
            for (int k = 0; k < nd; ++k) { double a = 0; for (int j = 0; j < nq; ++j) a += T2[k + j * nd] * r[(i - 2) * nr + nu + j]; bet[k] += a; }
            mm_nt_acc(Y1, T2, dz + (size_t)(i - 1) * nths * nd + (size_t)nq * nd, nd, nd, nq, 1.0);
            for (int k = 0; k < n2; ++k) Y2[k] = -T2[k];
        }
        double *l0 = L0 + (size_t)i * n2, *l1 = L1 + (size_t)i * n2, *l2 = L2 + (size_t)i * n2;
        if (i >= 2) { /* l2 = Y2 * L0_{i-2}^-T : solve X L^T = Y2 column by column */
            const double *Lp = L0 + (size_t)(i - 2) * n2;
            for (int c = 0; c < nd; ++c) for (int rr = 0; rr < nd; ++rr) {
                double a = Y2[rr + c * nd];
                for (int k = 0; k < c; ++k) a -= l2[rr + k * nd] * Lp[c + k * nd];
                l2[rr + c * nd] = a / Lp[c + c * nd];
            }
        }
        if (i >= 1) {
            if (i >= 2) mm_nt_acc(Y1, l2, L1 + (size_t)(i - 1) * n2, nd, nd, nd, -1.0);
            const double *Lp = L0 + (size_t)(i - 1) * n2;
            for (int c = 0; c < nd; ++c) for (int rr = 0; rr < nd; ++rr) {
                double a = Y1[rr + c * nd];
                for (int k = 0; k < c; ++k) a -= l1[rr + k * nd] * Lp[c + k * nd];
                l1[rr + c * nd] = a / Lp[c + c * nd];
            }
            mm_nt_acc(Y0, l1, l1, nd, nd, nd, -1.0);
        }
        if (i >= 2) mm_nt_acc(Y0, l2, l2, nd, nd, nd, -1.0);
        memset(l0, 0, n2 * sizeof(double));
        for (int c = 0; c < nd; ++c) {                              /* Cholesky */
            double dg = Y0[c + c * nd];
            for (int k = 0; k < c; ++k) dg -= l0[c + k * nd] * l0[c + k * nd];
            if (!(dg > 0.0)) return -1;
            dg = sqrt(dg);
            l0[c + c * nd] = dg;
            for (int rr = c + 1; rr < nd; ++rr) {
                double a = Y0[rr + c * nd];
                for (int k = 0; k < c; ++k) a -= l0[rr + k * nd] * l0[c + k * nd];
                l0[rr + c * nd] = a / dg;
            }
        }
        for (int k = 0; k < nd; ++k) {
            double a = bet[k];
            if (i >= 1) for (int j = 0; j < nd; ++j) a -= l1[k + j * nd] * yv[(size_t)(i - 1) * nd + j];
            if (i >= 2) for (int j = 0; j < nd; ++j) a -= l2[k + j * nd] * yv[(size_t)(i - 2) * nd + j];
            tv[k] = a;
        }
        for (int k = 0; k < nd; ++k) { double a = tv[k]; for (int j = 0; j < k; ++j) a -= l0[k + j * nd] * yv[(size_t)i * nd + j]; yv[(size_t)i * nd + k] = a / l0[k + k * nd]; }
    }
    for (int i = H - 1; i >= 0; --i) {
        const double *l0 = L0 + (size_t)i * n2;
        for (int k = 0; k < nd; ++k) {
            double a = yv[(size_t)i * nd + k];
            if (i + 1 < H) for (int j = 0; j < nd; ++j) a -= L1[(size_t)(i + 1) * n2 + j + k * nd] * dn[(size_t)(i + 1) * nd + j];
            if (i + 2 < H) for (int j = 0; j < nd; ++j) a -= L2[(size_t)(i + 2) * n2 + j + k * nd] * dn[(size_t)(i + 2) * nd + j];
            tv[k] = a;
        }
        for (int k = nd - 1; k >= 0; --k) { double a = tv[k]; for (int j = k + 1; j < nd; ++j) a -= l0[j + k * nd] * dn[(size_t)i * nd + j]; dn[(size_t)i * nd + k] = a / l0[k + k * nd]; }
    }
    for (int i = 0; i < H; ++i) {
        const double *A0 = dz + (size_t)i * nths * nd + (size_t)2 * nq * nd;
        for (int c = 0; c < nu; ++c) { double a = 0; for (int k = 0; k < nd; ++k) a += A0[k + (size_t)c * nd] * dn[(size_t)i * nd + k]; tv[c] = r[i * nr + c] - a; }
        for (int c = 0; c < nu; ++c) { double a = 0; for (int k = 0; k < nu; ++k) a += s->Rinv[(size_t)i * nu * nu + c + k * nu] * tv[k]; Delta[i * nr + c] = a; }
        for (int c = 0; c < nq; ++c) {
            double a = -dn[(size_t)i * nd + c];
            if (i + 1 < H) { const double *A1 = dz + (size_t)(i + 1) * nths * nd + (size_t)nq * nd; for (int k = 0; k < nd; ++k) a += A1[k + (size_t)c * nd] * dn[(size_t)(i + 1) * nd + k]; }
            if (i + 2 < H) { const double *A2 = dz + (size_t)(i + 2) * nths * nd; for (int k = 0; k < nd; ++k) a += A2[k + (size_t)c * nd] * dn[(size_t)(i + 2) * nd + k]; }
            tv[c] = r[i * nr + nu + c] - a;
        }
        for (int c = 0; c < nq; ++c) { double a = 0; for (int k = 0; k < nq; ++k) a += s->Qinv[(size_t)i * nq * nq + c + k * nq] * tv[k]; Delta[i * nr + nu + c] = a; }
        for (int k = 0; k < nd; ++k) Delta[H * nr + i * nd + k] = dn[(size_t)i * nd + k];
    }
    return 0;
}

int ref_kkt_solve(Ref *s, const double *dz, double beta, const double *r, double *Delta, int solver) {
    int N = s->N, H = s->H, nd = s->nd;
    if (solver == 0) {
        double *Rm = dalloc((size_t)N * N); int *piv = (int *)calloc(N, sizeof(int));
        kkt_dense_lu(s, dz, beta, r, Delta, Rm, piv);
        free(Rm); free(piv);
        return 0;
    }
    double *ws = dalloc((size_t)H * (3 * nd * nd + 2 * nd) + 8 * (size_t)nd * (nd + s->nq + s->nu) + 64);
    int rc = kkt_condensed(s, dz, beta, r, Delta, ws);
    free(ws);
    return rc;
}

static void apply_step(Ref *s, Tr dst, double *nu_dst, Tr src, const double *nu_src, const double *Delta, double alpha) {
    int H = s->H, nq = s->nq, nu = s->nu, nr = s->nr, nd = s->nd;
    for (int t = 0; t < H; ++t) {
        for (int c = 0; c < nq; ++c) dst.q[(size_t)(t + 2) * nq + c] = src.q[(size_t)(t + 2) * nq + c] - alpha * Delta[t * nr + nu + c];
        for (int c = 0; c < nu; ++c) dst.u[(size_t)t * nu + c] = src.u[(size_t)t * nu + c] - alpha * Delta[t * nr + c];
        for (int c = 0; c < nd; ++c) nu_dst[(size_t)t * nd + c] = nu_src[(size_t)t * nd + c] - alpha * Delta[H * nr + t * nd + c];
    }
    update_theta(s, dst);
}

/* newton_solve! for one rollout (cold start), :configuration mode.
 * window 0-based; ref arrays: q_ref [H+2][nq], u_ref [H][nu], w_ref [H][nw], th_ref [H][nth].
 * solver: 0 = dense LU (reference default), 1 = condensed block Cholesky.
 * out: q [H+2][nq], u [H][nu], nu_dual [H][nd]; info[0..5] = newton iters, sweeps, ip iters,
 * ip failures, r_norm/N (as double bits via rinfo). returns 0. */
int ref_newton_solve(Ref *s, const int *window, const double *q_ref, const double *u_ref, const double *w_ref,
                     const double *th_ref, const double *q0, const double *q1, int solver, double *q_out,
                     double *u_out, double *nu_out, int *info, double *rinfo) {
    int H = s->H, nq = s->nq, nu = s->nu, nw = s->nw, nth = s->nth, nd = s->nd, N = s->N, nths = s->nths;
    size_t sq = (size_t)(H + 2) * nq, su = (size_t)H * nu, sw = (size_t)H * nw, st = (size_t)H * nth;
    double *buf = dalloc(2 * (sq + su + sw + st) + 2 * (size_t)H * nd + (size_t)H * nd + (size_t)H * nths * nd + 3 * (size_t)N);
    Tr T = {buf, buf + sq, buf + sq + su, buf + sq + su + sw};
    double *p = buf + sq + su + sw + st;
    Tr C = {p, p + sq, p + sq + su, p + sq + su + sw};
    p += sq + su + sw + st;
    double *nuv = p, *nuc = p + (size_t)H * nd, *d = nuc + (size_t)H * nd, *dz = d + (size_t)H * nd;
    double *res = dz + (size_t)H * nths * nd, *resc = res + N, *Delta = resc + N;
    int *status = (int *)calloc(2 * H, sizeof(int)), *iters = status + H;
    double *Rm = NULL, *ws = NULL; int *piv = NULL;
    if (solver == 0) { Rm = dalloc((size_t)N * N); piv = (int *)calloc(N, sizeof(int)); }
    else ws = dalloc((size_t)H * (3 * nd * nd + 2 * nd) + 8 * (size_t)nd * (nd + nq + nu) + 64);
    /* reset! (cold) */
    memcpy(T.q, q_ref, sq * sizeof(double)); memcpy(T.u, u_ref, su * sizeof(double));
    memcpy(T.w, w_ref, sw * sizeof(double)); memcpy(T.th, th_ref, st * sizeof(double));
    memcpy(T.q, q0, nq * sizeof(double)); memcpy(T.q + nq, q1, nq * sizeof(double));
    {   /* update_theta!(traj, 1), (traj, 2) only */
        for (int i = 0; i < 2 && i < H; ++i) {
            double *th = T.th + (size_t)i * nth;
            memcpy(th, T.q + (size_t)i * nq, 2 * nq * sizeof(double));
            memcpy(th + 2 * nq, T.u + (size_t)i * nu, nu * sizeof(double));
            memcpy(th + 2 * nq + nu, T.w + (size_t)i * nw, nw * sizeof(double));
        }
    }
    memcpy(C.q, T.q, (sq + su + sw + st) * sizeof(double));
    double beta = s->beta_init;
    long sweeps = 0, ipit = 0, ipfail = 0;
#define SWEEP(X)                                                                              \
    do {                                                                                      \
        ref_implicit_dynamics(s, window, (X).q, (X).th, NULL, NULL, NULL, d, dz, status, iters, NULL); \
        sweeps++;                                                                             \
        for (int i_ = 0; i_ < H; ++i_) { ipit += iters[i_]; ipfail += !status[i_]; }          \
    } while (0)
    SWEEP(T);
    double r_norm = residual(s, T, nuv, d, dz, q_ref, u_ref, res);
    int l = 0;
    for (int it = 0; it < s->n_max_iter; ++it) {
        if (r_norm / N < s->n_r_tol) break;
        l++;
        if (solver == 0) kkt_dense_lu(s, dz, beta, res, Delta, Rm, piv);
        else kkt_condensed(s, dz, beta, res, Delta, ws);
        double alpha = 1.0; int ls = 0;
        apply_step(s, C, nuc, T, nuv, Delta, alpha);
        SWEEP(C);
        double r_cand = residual(s, C, nuc, d, dz, q_ref, u_ref, resc);
        while (r_cand * r_cand >= (1.0 - 0.001 * alpha) * r_norm * r_norm) {
            alpha *= 0.5; ls++;
            if (ls > 6) break;
            apply_step(s, C, nuc, T, nuv, Delta, alpha);
            SWEEP(C);
            r_cand = residual(s, C, nuc, d, dz, q_ref, u_ref, resc);
        }
        apply_step(s, T, nuv, T, nuv, Delta, alpha);
        memcpy(res, resc, N * sizeof(double));
        r_norm = r_cand;
        beta = (ls > 6) ? fmin(beta * 1.3, 1.0e2) : fmax(1.0e1, beta / 1.3);
    }
#undef SWEEP
    if (q_out) memcpy(q_out, T.q, sq * sizeof(double));
    if (u_out) memcpy(u_out, T.u, su * sizeof(double));
    if (nu_out) memcpy(nu_out, nuv, (size_t)H * nd * sizeof(double));
    if (info) { info[0] = l; info[1] = (int)sweeps; info[2] = (int)ipit; info[3] = (int)ipfail; }
    if (rinfo) rinfo[0] = r_norm / N;
    free(buf); free(status); free(Rm); free(piv); free(ws);
    return 0;
}
