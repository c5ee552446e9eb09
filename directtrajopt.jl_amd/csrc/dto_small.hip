// dto_small.hip -- small-state path (n <= 32): ONE WORKGROUP PER KNOT INTERVAL (one wavefront for n <= 16, four for
// 17..32), everything in LDS.
//
// This is the shape the reference's own benchmarks have (4..16 states, N ~ 50..100,
// benchmark/problem_utils.jl:10-42, docs/src/benchmarks.md:83-124): the work per interval is a few
// thousand FMAs, so the batched-GEMM machinery of dto_kernels.hip (tens of launches) is all latency.
// Here a 64-lane workgroup stages (x_k, x_{k+1}, u_k, dt_k), the generators G_0..G_m and the interval's
// matrices in LDS and produces, in one launch for all intervals:
//   delta_k                                   (bilinear_integrator.jl:98-107)
//   -E_k, -dexp(A)[dt G_j] x_k, -G(u) E_k x_k (bilinear_integrator.jl:111-131)  -> Jacobian slab
//   the 5 non-zero block types of mu_k' f     (bilinear_integrator.jl:135-161)  -> Hessian slab
// with the same mathematics as the large path: Taylor degree 16 + squarings for E_k (Horner form: n is
// tiny, products are free), Taylor recurrences with Al-Mohy--Higham termination for the vector quantities.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "dto_kernels.h"

namespace dto {

namespace {

constexpr int SMALL_MAX_TERMS = 80;

__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// C = A * B (n x n, column-major, LDS), all NT threads
template <int NT>
__device__ __forceinline__ void mm(double* __restrict__ C, const double* __restrict__ A, const double* __restrict__ B, int n) {
    for (int e = threadIdx.x; e < n * n; e += NT) {
        const int r = e % n, c = e / n;
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += A[r + k * n] * B[k + c * n];
        C[e] = s;
    }
}

struct SmallLds {
    double* G;    // [(m+1)][n*n]
    double* A;    // [n*n]
    double* X;    // [n*n]
    double* Y;    // [n*n]
    double* cur;  // [T][n]
    double* nxt;  // [T][n]
    double* S;    // [T][n]
    double* S2;   // [T2][n]  (adjoint sums)
    double* W;    // [(m+1)][n] G_j' mu
    double* vec;  // [4][n] scratch: GY, GYt, Gm, mu
    unsigned long long* tn;  // [MAX_TYPES] term norms (bit patterns)
    unsigned long long* sn;  // [MAX_TYPES] sum norms
    double* pn;              // [MAX_TYPES] previous term norms
};

// Taylor recurrences for T column types starting from term0 (type 0 = v0, others 0); sums left in S.
//   transposed = 0: A v, G_g v      1: A' v, G_g' v
template <int NT>
__device__ void small_sweep(const SmallLds& L, const SweepTypes& ty, int n, int m, const double* ub, double dt,
                            const double* v0, double* S, int transposed, double beta) {
    const int T = ty.T;
    const int q = beta == beta && beta < 1e6 ? max(1, (int)ceil(beta / 9.0)) : 1;
    const double dq = dt / q;
    for (int e = threadIdx.x; e < T * n; e += NT) {
        const double v = e < n ? v0[e] : 0.0;
        L.cur[e] = v;
        S[e] = v;
    }
    __syncthreads();
    for (int round = 0; round < q; ++round) {
        if (round > 0) {
            for (int e = threadIdx.x; e < T * n; e += NT) L.cur[e] = S[e];
            __syncthreads();
        }
        for (int t = 0; t < SMALL_MAX_TERMS; ++t) {
            const double inv = 1.0 / (double)(t + 1);
            if (threadIdx.x < T) { L.tn[threadIdx.x] = 0ull; L.sn[threadIdx.x] = 0ull; }
            __syncthreads();
            for (int e = threadIdx.x; e < T * n; e += NT) {
                const int tau = e / n, i = e % n;
                const double* c = L.cur + tau * n;
                double acc = 0.0;
                for (int k = 0; k < n; ++k) acc += (transposed ? L.A[k + i * n] : L.A[i + k * n]) * c[k];
                acc /= (double)q;
                const TypeDesc td = ty.t[tau];
                for (int x = 0; x < td.n_extra; ++x) {
                    const double* g = L.G + td.gen[x] * n * n;
                    const double* cs = L.cur + td.src[x] * n;
                    double a2 = 0.0;
                    for (int k = 0; k < n; ++k) a2 += (transposed ? g[k + i * n] : g[i + k * n]) * cs[k];
                    acc += td.mult[x] * dq * a2;
                }
                acc *= inv;
                L.nxt[e] = acc;
                const double s = S[e] + acc;
                S[e] = s;
                atomicMax(&L.tn[tau], (unsigned long long)__double_as_longlong(fabs(acc)));
                atomicMax(&L.sn[tau], (unsigned long long)__double_as_longlong(fabs(s)));
            }
            __syncthreads();
            // per column type: two successive terms below tol * |sum| (Al-Mohy & Higham 2011, Alg. 3.2)
            bool ok = true;
            if (threadIdx.x < T) {
                const double tnv = __longlong_as_double((long long)L.tn[threadIdx.x]);
                const double snv = __longlong_as_double((long long)L.sn[threadIdx.x]);
                const double pv = t == 0 ? 1e300 : L.pn[threadIdx.x];
                ok = (pv + tnv <= 1.1e-16 * snv) || !(tnv == tnv) || !(snv < 1e300);
                L.pn[threadIdx.x] = tnv;
            }
            const bool done = __syncthreads_and(ok);
            for (int e = threadIdx.x; e < T * n; e += NT) L.cur[e] = L.nxt[e];
            __syncthreads();
            if (done) break;
        }
    }
}

}  // namespace

struct SmallArgs {
    KProb P;
    KBil B;
    SweepTypes ty_fw, ty_ad;
    const double* Gs;   // compact generators (m+1) x n x n
    const double* Z;
    const double* mu;
    double* cons;
    double* jac;
    double* hess;
    int mode;           // bit0 constraint values, bit1 Jacobian, bit2 Hessian
};

template <int NT>
__global__ void __launch_bounds__(NT) k_small(SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const KProb& P = a.P;
    const KBil& B = a.B;
    const int n = B.n, m = B.m, nn = n * n;
    const int64_t kl = blockIdx.x;
    const int64_t kn = P.kn_lo + kl;
    const int Tfw = a.ty_fw.T, Tad = a.ty_ad.T;
    const int Tmax = Tfw > Tad ? Tfw : Tad;
    SmallLds L;
    double* p = lds;
    L.G = p; p += (m + 1) * nn;
    L.A = p; p += nn;
    L.X = p; p += nn;
    L.Y = p; p += nn;
    L.cur = p; p += Tmax * n;
    L.nxt = p; p += Tmax * n;
    L.S = p; p += Tfw * n;
    L.S2 = p; p += Tad * n;
    L.W = p; p += (m + 1) * n;
    L.vec = p; p += 4 * n;
    L.tn = reinterpret_cast<unsigned long long*>(p); p += MAX_TYPES;
    L.sn = reinterpret_cast<unsigned long long*>(p); p += MAX_TYPES;
    L.pn = p;

    const double* zk = a.Z + kn * P.z;
    const double dt = zk[P.dt_idx];
    double ub[MAX_DRIVES + 1];
    ub[0] = 1.0;
    for (int j = 0; j < m; ++j) ub[j + 1] = zk[B.u_off + j];

    for (int e = threadIdx.x; e < (m + 1) * nn; e += NT) L.G[e] = a.Gs[e];
    __syncthreads();
    double colsum = 0.0;
    for (int e = threadIdx.x; e < nn; e += NT) {
        double s = 0.0;
        for (int j = 0; j <= m; ++j) s += ub[j] * L.G[j * nn + e];
        L.A[e] = dt * s;
    }
    __syncthreads();
    // ||A||_1 (exact): drives both the scaling of the matrix exponential and the sub-stepping of the sweeps
    double n1 = 0.0;
    for (int c = threadIdx.x; c < n; c += NT) {
        double s = 0.0;
        for (int r = 0; r < n; ++r) s += fabs(L.A[r + c * n]);
        n1 = fmax(n1, s);
    }
    n1 = wmax(n1);
    if (NT > 64) {  // several wavefronts: combine through LDS (the slot is rewritten by the sweep afterwards)
        __syncthreads();
        if ((threadIdx.x & 63) == 0) L.pn[threadIdx.x >> 6] = n1;
        __syncthreads();
        n1 = L.pn[0];
        for (int w = 1; w < NT / 64; ++w) n1 = fmax(n1, L.pn[w]);
        __syncthreads();
    }
    if (!(n1 == n1)) n1 = __longlong_as_double(0x7ff8000000000000ll);
    (void)colsum;

    const double* xk = zk + B.x_off;
    double* yv = L.S;  // forward sums: S[0] = exp(A) x, S[1+j] = dexp(A)[dt G_j] x, then h^{ij}
    small_sweep<NT>(L, a.ty_fw, n, m, ub, dt, xk, L.S, 0, n1);
    __syncthreads();
    // GY = G(u) y
    double* GY = L.vec;
    for (int i = threadIdx.x; i < n; i += NT) {
        double s = 0.0;
        for (int j = 0; j <= m; ++j) {
            double g = 0.0;
            for (int k = 0; k < n; ++k) g += L.G[j * nn + i + k * n] * yv[k];
            s += ub[j] * g;
        }
        GY[i] = s;
    }
    __syncthreads();

    if (a.mode & 1) {
        for (int r = threadIdx.x; r < n; r += NT)
            a.cons[B.lrow_off + kl * n + r] = a.Z[(kn + 1) * P.z + B.x_off + r] - yv[r];
    }

    if (a.mode & 2) {
        // E = exp(A): Taylor degree 16 (Horner) of A / 2^s, then s squarings
        int s = 0;
        if (n1 > THETA_16) s = (int)ceil(log2(n1 / THETA_16));
        if (!(n1 == n1) || s > 60) s = 60;
        const double sigma = ldexp(1.0, -s);
        for (int e = threadIdx.x; e < nn; e += NT) L.X[e] = (e % n == e / n) ? 1.0 : 0.0;
        __syncthreads();
        for (int i = TAYLOR_M; i >= 1; --i) {
            mm<NT>(L.Y, L.A, L.X, n);
            __syncthreads();
            const double f = sigma / (double)i;
            for (int e = threadIdx.x; e < nn; e += NT) L.X[e] = ((e % n == e / n) ? 1.0 : 0.0) + f * L.Y[e];
            __syncthreads();
        }
        for (int it = 0; it < s; ++it) {
            mm<NT>(L.Y, L.X, L.X, n);
            __syncthreads();
            for (int e = threadIdx.x; e < nn; e += NT) L.X[e] = L.Y[e];
            __syncthreads();
        }
        // Jacobian block of interval kn (own rows): x columns = -E, u_j = -c_j, dt = -G(u) y
        for (int e = threadIdx.x; e < nn; e += NT) {
            const int r = e % n, c = e / n;
            a.jac[jac_pos(P, P.colptr, kn, B.x_off + c, B.pre, n, 1, r)] = -L.X[e];
        }
        for (int e = threadIdx.x; e < m * n; e += NT) {
            const int j = e / n, r = e % n;
            a.jac[jac_pos(P, P.colptr, kn, B.u_off + j, B.pre, n, 1, r)] = -L.S[(1 + j) * n + r];
        }
        for (int r = threadIdx.x; r < n; r += NT) a.jac[jac_pos(P, P.colptr, kn, P.dt_idx, B.pre, n, 1, r)] = -GY[r];
    }

    if (a.mode & 4) {
        const double* muk = a.mu + B.row_off + kn * n;
        double* muv = L.vec + 3 * n;
        for (int i = threadIdx.x; i < n; i += NT) muv[i] = muk[i];
        __syncthreads();
        small_sweep<NT>(L, a.ty_ad, n, m, ub, dt, muv, L.S2, 1, n1);
        __syncthreads();
        double* GYt = L.vec + n;   // G(u)' yt
        double* Gm = L.vec + 2 * n;  // G(u)' mu
        for (int e = threadIdx.x; e < (m + 1) * n; e += NT) {
            const int j = e / n, i = e % n;
            double g = 0.0;
            for (int k = 0; k < n; ++k) g += L.G[j * nn + k + i * n] * muv[k];
            L.W[e] = g;
        }
        for (int i = threadIdx.x; i < n; i += NT) {
            double s = 0.0;
            for (int j = 0; j <= m; ++j) {
                double g = 0.0;
                for (int k = 0; k < n; ++k) g += L.G[j * nn + k + i * n] * L.S2[k];
                s += ub[j] * g;
            }
            GYt[i] = s;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += NT) {
            double s = 0.0;
            for (int j = 0; j <= m; ++j) s += ub[j] * L.W[j * n + i];
            Gm[i] = s;
        }
        __syncthreads();
        auto hadd = [&](int ca, int cb, double v) {
            const int lo = ca < cb ? ca : cb, hi = ca < cb ? cb : ca;
            atomicAdd(&a.hess[hess_pos(P, kn, lo, hi)], v);
        };
        for (int e = threadIdx.x; e < m * n; e += NT) {
            const int j = e / n, r = e % n;
            hadd(B.x_off + r, B.u_off + j, -L.S2[(1 + j) * n + r]);
        }
        for (int r = threadIdx.x; r < n; r += NT) hadd(B.x_off + r, P.dt_idx, -GYt[r]);
        // scalar blocks: one lane per entry
        const int npair = m * (m + 1) / 2;
        for (int e = threadIdx.x; e < npair + m + 1; e += NT) {
            if (e < npair) {
                int i = 0, rem = e;
                while (rem >= m - i) { rem -= m - i; ++i; }
                const int j = i + rem;
                double s = 0.0;
                for (int r = 0; r < n; ++r) s += muv[r] * L.S[(1 + m + e) * n + r];
                hadd(B.u_off + i, B.u_off + j, -s);
            } else if (e < npair + m) {
                const int j = e - npair;
                double s = 0.0;
                for (int r = 0; r < n; ++r) s += L.W[(1 + j) * n + r] * yv[r] + Gm[r] * L.S[(1 + j) * n + r];
                hadd(B.u_off + j, P.dt_idx, -s);
            } else {
                double s = 0.0;
                for (int r = 0; r < n; ++r) s += Gm[r] * GY[r];
                atomicAdd(&a.hess[hess_pos(P, kn, P.dt_idx, P.dt_idx)], -s);
            }
        }
    }
}

// identity rows of interval kn-1 in the x columns of knot kn (the constant z_{k+1} half)
__global__ void k_small_identity(KProb P, KBil B, double* __restrict__ jac) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_knots * B.n) return;
    const int64_t kn = P.kn_lo + i / B.n;
    const int r = (int)(i % B.n);
    if (kn >= 1) jac[jac_pos(P, P.colptr, kn, B.x_off + r, B.pre, B.n, 0, r)] = 1.0;
}

// Dynamic LDS beyond 64 KB has to be opted into per kernel and per DEVICE: called from dto_create (after
// hipSetDevice) with the handle's worst-case request, so that no launch depends on what another handle did.
hipError_t small_prepare(size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_small<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t launch_small(hipStream_t st, const KProb& P, const KBil& B, const double* Gs, const SweepTypes& ty_fw,
                        const SweepTypes& ty_ad, const double* dZ, const double* dmu, double* cons, double* jac, double* hess,
                        int mode) {
    if (mode & 2) {
        const int64_t nid = P.n_knots * B.n;
        if (nid > 0) hipLaunchKernelGGL(k_small_identity, dim3((unsigned)((nid + 255) / 256)), dim3(256), 0, st, P, B, jac);
    }
    if (P.n_int <= 0) return hipGetLastError();
    SmallArgs a{};
    a.P = P; a.B = B; a.ty_fw = ty_fw; a.ty_ad = ty_ad; a.Gs = Gs; a.Z = dZ; a.mu = dmu;
    a.cons = cons; a.jac = jac; a.hess = hess; a.mode = mode;
    const int n = B.n, m = B.m;
    size_t bytes = small_lds_bytes(n, m, ty_fw.T, ty_ad.T);
    if (P.debug_bad_launch) bytes += 512 * 1024;  // more LDS than a CU has: the runtime must reject the launch
    // n <= 16: one wavefront per interval; 17..32: four (the n^3 products of the matrix exponential dominate there).
    // (A variant for 33..64 -- MFMA products on zero-padded matrices, generators left in global memory -- was correct
    // but slower than the general path at N = 1000: one 130 KB workgroup per CU and scalar sweeps; dropped.)
    if (n <= 16) hipLaunchKernelGGL(k_small<64>, dim3((unsigned)P.n_int), dim3(64), bytes, st, a);
    else hipLaunchKernelGGL(k_small<256>, dim3((unsigned)P.n_int), dim3(256), bytes, st, a);
    return hipGetLastError();
}

// LDS bytes the fused kernel needs for one interval (the engine falls back to the general path beyond the CU's 160 KB)
size_t small_lds_bytes(int n, int m, int T_fw, int T_ad) {
    const int nn = n * n, Tmax = T_fw > T_ad ? T_fw : T_ad;
    return ((size_t)(m + 1) * nn + 3 * nn + 2 * Tmax * n + (size_t)T_fw * n + (size_t)T_ad * n + (size_t)(m + 1) * n + 4 * n +
            3 * MAX_TYPES) * sizeof(double);
}

}  // namespace dto
