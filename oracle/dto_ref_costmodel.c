/*
 * dto_ref_costmodel.c -- TEST / MEASUREMENT INFRASTRUCTURE, not product code.  Only tests/ and bench.py's `cpu_baseline`
 * leg build and run it.
 *
 * CPU restatement of HOW the reference computes the bilinear Jacobian block, for the "reference-algorithm" CPU baseline
 * of SURVEY.md section 8d (the reference itself is Julia and cannot run here):
 *
 *   src/integrators/bilinear_integrator.jl:81       f = x_{k+1} - expv(dt_k, G(u_k), x_k)
 *   src/integrators/bilinear_integrator.jl:111-131  for k = 1:N-1 (serial)  ForwardDiff.jacobian!(block, f, [z_k; z_{k+1}])
 *
 *   - ForwardDiff.jacobian! differentiates with respect to all 2z inputs in chunks of at most 12 partials
 *     (ForwardDiff's DEFAULT_CHUNK_THRESHOLD): ceil(2z/12) evaluations of f on dual numbers with 12 partials each;
 *     no sparsity of the seeds is exploited (the x_{k+1}, du, ... directions cost as much as the u directions);
 *   - every evaluation rebuilds G(u) = G_0 + sum_j u_j G_j as a matrix of duals and runs `expv`, the truncated-Taylor
 *     action of the matrix exponential (ExponentialAction.jl, compat 0.2 -- not vendored; restated here from the
 *     published algorithm it implements: Al-Mohy & Higham, "Computing the action of the matrix exponential", SIAM J.
 *     Sci. Comput. 33 (2011), Algorithm 3.2 with the theta_m table of their Table 3.1 / Higham's Table A.3 for
 *     tol = 2^-53, m_max = 55, early termination c1 + c2 <= tol ||F||_inf; the spectral shift is omitted, it does not
 *     change the operation count);
 *   - (m*, s) are chosen from alpha_p(tA) with EXACT ||(tA)^p||_1, p <= 4, once per knot (three n^3 products, 2 % of a
 *     knot's work at n = 256, included in the timing); the package estimates them per call with a few products of its own.
 *
 * Dual arithmetic is laid out as planes (value plane + 12 partial planes): y.v = A.v b.v, y.d[p] = A.v b.d[p] + A.d[p] b.v,
 * the same 25 multiply-adds per matrix entry that ForwardDiff's Dual{T,Float64,12} product performs.
 *
 * usage:  dto_ref_costmodel bench <problem.bin> <worker> <stride> <budget_s>   -> prints "<knots done> <seconds>"
 *         dto_ref_costmodel block <problem.bin> <k (0-based)> <out.bin>        -> writes the n x 2z block (column-major)
 * problem.bin: int64 n, m, N, z; double G[(m+1) n n] (column-major generators, G_0 first); double Z[N z] (knot-major,
 * x at 0, u at n, timestep at z-1).
 * build:  gcc -O3 -march=native -o oracle/_build/dto_ref_costmodel oracle/dto_ref_costmodel.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define P 12 /* ForwardDiff chunk size */

static const int THETA_M[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 35, 40, 45, 50, 55};
static const double THETA[] = {2.29e-16, 2.58e-8, 1.39e-5, 3.40e-4, 2.40e-3, 9.07e-3, 2.38e-2, 5.00e-2, 8.96e-2, 1.44e-1,
                               2.14e-1, 3.00e-1, 4.00e-1, 5.14e-1, 6.41e-1, 7.81e-1, 9.31e-1, 1.09, 1.26, 1.44,
                               1.62, 1.82, 2.01, 2.22, 2.43, 2.64, 2.86, 3.08, 3.31, 3.54, 4.7, 6.0, 7.2, 8.5, 9.9};
#define NTHETA ((int)(sizeof(THETA_M) / sizeof(THETA_M[0])))

typedef struct {
    int64_t n, m, N, z;
    double* G;
    double* Z;
} problem;

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static int load(const char* path, problem* p) {
    FILE* f = fopen(path, "rb");
    if (!f) return 1;
    int64_t h[4];
    if (fread(h, sizeof(int64_t), 4, f) != 4) return 1;
    p->n = h[0]; p->m = h[1]; p->N = h[2]; p->z = h[3];
    const size_t ng = (size_t)(p->m + 1) * p->n * p->n, nz = (size_t)p->N * p->z;
    p->G = malloc(ng * sizeof(double));
    p->Z = malloc(nz * sizeof(double));
    if (fread(p->G, sizeof(double), ng, f) != ng || fread(p->Z, sizeof(double), nz, f) != nz) return 1;
    fclose(f);
    return 0;
}

static double norm1(const double* A, int n) {
    double best = 0.0;
    for (int c = 0; c < n; ++c) {
        double s = 0.0;
        for (int r = 0; r < n; ++r) s += fabs(A[(size_t)c * n + r]);
        if (s > best) best = s;
    }
    return best;
}
static void matmul(const double* A, const double* B, double* C, int n) {
    memset(C, 0, sizeof(double) * (size_t)n * n);
    for (int c = 0; c < n; ++c)
        for (int k = 0; k < n; ++k) {
            const double b = B[(size_t)c * n + k];
            const double* a = A + (size_t)k * n;
            double* o = C + (size_t)c * n;
            for (int r = 0; r < n; ++r) o[r] += a[r] * b;
        }
}

/* (m*, s) of Al-Mohy & Higham's parameter selection (their eq. 3.11-3.13 with p_max = 4 and exact norms) */
static void choose_ms(const double* tA, int n, double* scratch, int* m_star, int* s_star) {
    double* A2 = scratch;
    double* A3 = scratch + (size_t)n * n;
    double* A4 = scratch + 2 * (size_t)n * n;
    const double n1 = norm1(tA, n);
    matmul(tA, tA, A2, n);
    matmul(A2, tA, A3, n);
    matmul(A2, A2, A4, n);
    const double d2 = sqrt(norm1(A2, n)), d3 = cbrt(norm1(A3, n)), d4 = sqrt(sqrt(norm1(A4, n)));
    const double a2 = fmax(d2, d3), a3 = fmax(d3, d4);
    long best = -1;
    for (int i = 0; i < NTHETA; ++i) {
        const int mm = THETA_M[i];
        /* alpha_p is admissible for m >= p (p - 1) - 1 */
        double alpha = n1;
        if (mm >= 1) alpha = fmin(alpha, a2);
        if (mm >= 5) alpha = fmin(alpha, a3);
        long s = (long)ceil(alpha / THETA[i]);
        if (s < 1) s = 1;
        const long cost = s * mm;
        if (best < 0 || cost < best) { best = cost; *m_star = mm; *s_star = (int)s; }
    }
}

/* y = A b on duals in plane layout: Av, Ad[P] (n x n, column-major); bv, bd[P] (n) */
static void dual_matvec(int n, const double* Av, const double* const* Ad, const double* bv, double* const* bd, double* yv,
                        double* const* yd) {
    memset(yv, 0, sizeof(double) * n);
    for (int p = 0; p < P; ++p) memset(yd[p], 0, sizeof(double) * n);
    for (int j = 0; j < n; ++j) {
        const double* av = Av + (size_t)j * n;
        const double b0 = bv[j];
        for (int i = 0; i < n; ++i) yv[i] += av[i] * b0;
        for (int p = 0; p < P; ++p) {
            const double* ad = Ad[p] + (size_t)j * n;
            const double bp = bd[p][j];
            double* y = yd[p];
            for (int i = 0; i < n; ++i) y[i] += av[i] * bp + ad[i] * b0;
        }
    }
}

typedef struct {
    int n;
    double *Av, *Ad[P];          /* t A as duals */
    double *bv, *bd[P], *fv, *fd[P], *yv, *yd[P];
    double* scratch;             /* 3 n^2 for the norms */
} work;

static void work_alloc(work* w, int n) {
    w->n = n;
    w->Av = malloc(sizeof(double) * (size_t)n * n);
    for (int p = 0; p < P; ++p) w->Ad[p] = malloc(sizeof(double) * (size_t)n * n);
    w->bv = malloc(sizeof(double) * n); w->fv = malloc(sizeof(double) * n); w->yv = malloc(sizeof(double) * n);
    for (int p = 0; p < P; ++p) {
        w->bd[p] = malloc(sizeof(double) * n); w->fd[p] = malloc(sizeof(double) * n); w->yd[p] = malloc(sizeof(double) * n);
    }
    w->scratch = malloc(sizeof(double) * 3 * (size_t)n * n);
}

static double inf_norm(const double* v, int n) {
    double m = 0.0;
    for (int i = 0; i < n; ++i) m = fmax(m, fabs(v[i]));
    return m;
}

/* One chunk of ForwardDiff.jacobian!: inputs [c0, c0 + cnt) of [z_k; z_{k+1}] carry the seeds.  Writes columns c0.. of the
 * n x 2z block.  *matvecs counts the dual products (diagnostics). */
static void chunk(const problem* pr, work* w, int64_t k, int c0, int cnt, int m_star, int s, double* block, long* matvecs) {
    const int n = (int)pr->n, m = (int)pr->m, z = (int)pr->z;
    const double* zk = pr->Z + (size_t)k * z;
    const double* zk1 = zk + z;
    const double dt = zk[z - 1];
    const size_t nn = (size_t)n * n;
    /* seeds: partial p belongs to input c0 + p */
    /* t A = dt (G_0 + sum_j u_j G_j) as duals: value plane and, for partials seeding u_j or dt, their planes */
    for (size_t e = 0; e < nn; ++e) {
        double g = pr->G[e];
        for (int j = 0; j < m; ++j) g += zk[n + j] * pr->G[(size_t)(1 + j) * nn + e];
        w->Av[e] = g;  /* G(u) for now */
    }
    for (int p = 0; p < P; ++p) {
        const int c = c0 + p;
        double* ad = w->Ad[p];
        if (p < cnt && c >= n && c < n + m) {            /* d/du_j: dt G_j */
            const double* gj = pr->G + (size_t)(1 + c - n) * nn;
            for (size_t e = 0; e < nn; ++e) ad[e] = dt * gj[e];
        } else if (p < cnt && c == z - 1) {              /* d/ddt: G(u) */
            memcpy(ad, w->Av, sizeof(double) * nn);
        } else {
            memset(ad, 0, sizeof(double) * nn);           /* ForwardDiff carries the zero partials all the same */
        }
    }
    for (size_t e = 0; e < nn; ++e) w->Av[e] *= dt;
    /* b = x_k as duals */
    for (int i = 0; i < n; ++i) { w->bv[i] = zk[i]; w->fv[i] = zk[i]; }
    for (int p = 0; p < P; ++p)
        for (int i = 0; i < n; ++i) {
            const double sd = (p < cnt && c0 + p == i) ? 1.0 : 0.0;
            w->bd[p][i] = sd; w->fd[p][i] = sd;
        }
    /* Algorithm 3.2 */
    const double tol = ldexp(1.0, -53);
    for (int st = 0; st < s; ++st) {
        double c1 = inf_norm(w->bv, n);
        for (int j = 1; j <= m_star; ++j) {
            dual_matvec(n, w->Av, (const double* const*)w->Ad, w->bv, w->bd, w->yv, w->yd);
            ++*matvecs;
            const double f = 1.0 / ((double)s * j);
            for (int i = 0; i < n; ++i) { w->bv[i] = f * w->yv[i]; w->fv[i] += w->bv[i]; }
            for (int p = 0; p < P; ++p)
                for (int i = 0; i < n; ++i) { w->bd[p][i] = f * w->yd[p][i]; w->fd[p][i] += w->bd[p][i]; }
            const double c2 = inf_norm(w->bv, n);
            if (c1 + c2 <= tol * inf_norm(w->fv, n)) break;
            c1 = c2;
        }
        memcpy(w->bv, w->fv, sizeof(double) * n);
        for (int p = 0; p < P; ++p) memcpy(w->bd[p], w->fd[p], sizeof(double) * n);
    }
    /* f = x_{k+1} - F: column of input c is -dF/dc (+ identity for the x_{k+1} inputs) */
    (void)zk1;
    for (int p = 0; p < cnt; ++p) {
        const int c = c0 + p;
        double* col = block + (size_t)c * n;
        for (int i = 0; i < n; ++i) col[i] = -w->fd[p][i];
        if (c >= z && c < z + n) col[c - z] += 1.0;
    }
}

static void jacobian_block(const problem* pr, work* w, int64_t k, double* block, long* matvecs) {
    const int n = (int)pr->n, z = (int)pr->z;
    /* parameters of expv from the value part of t A (the same for every chunk of this knot) */
    const double* zk = pr->Z + (size_t)k * z;
    const size_t nn = (size_t)n * n;
    for (size_t e = 0; e < nn; ++e) {
        double g = pr->G[e];
        for (int j = 0; j < (int)pr->m; ++j) g += zk[n + j] * pr->G[(size_t)(1 + j) * nn + e];
        w->Av[e] = zk[z - 1] * g;
    }
    int m_star = 55, s = 1;
    choose_ms(w->Av, n, w->scratch, &m_star, &s);
    for (int c0 = 0; c0 < 2 * z; c0 += P) {
        const int cnt = 2 * z - c0 < P ? 2 * z - c0 : P;
        chunk(pr, w, k, c0, cnt, m_star, s, block, matvecs);
    }
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: see the header of dto_ref_costmodel.c\n"); return 2; }
    problem pr;
    if (load(argv[2], &pr)) { fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
    work w;
    work_alloc(&w, (int)pr.n);
    double* block = calloc((size_t)pr.n * 2 * pr.z, sizeof(double));
    long matvecs = 0;
    if (!strcmp(argv[1], "block") && argc == 5) {
        jacobian_block(&pr, &w, atoll(argv[3]), block, &matvecs);
        FILE* f = fopen(argv[4], "wb");
        fwrite(block, sizeof(double), (size_t)pr.n * 2 * pr.z, f);
        fclose(f);
        printf("%ld\n", matvecs);
        return 0;
    }
    if (!strcmp(argv[1], "bench") && argc == 6) {
        const int64_t worker = atoll(argv[3]), stride = atoll(argv[4]);
        const double budget = atof(argv[5]);
        /* the reference walks k = 1..N-1 serially; W processes each walk every W-th interval */
        const double t0 = now();
        int64_t done = 0;
        for (int64_t k = worker; k < pr.N - 1; k += stride) {
            jacobian_block(&pr, &w, k, block, &matvecs);
            ++done;
            if (now() - t0 >= budget) break;
        }
        printf("%lld %.6f %ld\n", (long long)done, now() - t0, matvecs);
        return 0;
    }
    fprintf(stderr, "bad arguments\n");
    return 2;
}
