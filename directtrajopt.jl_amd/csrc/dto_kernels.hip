// dto_kernels.hip -- hand-written gfx950 kernels of the NLP-callback engine.
//
// Hot path:
//   I1/I2  BilinearIntegrator defect and Jacobian block  (src/integrators/bilinear_integrator.jl:81,98-131)
//   I3     its Hessian of the Lagrangian                  (:135-161)
//   I4-I6  DerivativeIntegrator                            (src/integrators/derivative_integrator.jl:45-116)
//   O1-O4  regularizer / minimum-time / composite objectives (src/objectives/*.jl)
//   C1     NonlinearKnotPointConstraint, built-in g        (src/constraints/nonlinear/knot_point_constraint.jl)
//   A1/A2  value assembly in the evaluator's CSC order     (src/solvers/evaluator.jl:491-647)
//
// Two compute engines:
//   (1) the PROPAGATOR CHAIN  E_k = exp(dt_k G(u_k)) as a dense matrix, needed because -E_k IS the
//       x_k block of the Jacobian: Taylor degree 16 by Paterson-Stockmeyer (6 products) + s_k
//       squarings, every product a batched FP64 MFMA GEMM (k_bgemm), the last one storing -E_k
//       straight into the Jacobian value slab;
//   (2) the GENERATOR SWEEP   exp(A)x, dexp(A)[G_j]x, ... as Taylor recurrences on vectors where
//       every product with A = dt*sum_j ubar_j G_j is expanded over the SHARED generators, so that
//       one step for all knots is a single GEMM  [G_0 .. G_m] x (columns of all knots)  (k_sweep).
#include "dto_gemm.hip.h"
#include "dto_gemm_ring.hip.h"
#include "dto_kernels.h"
#include "dto_hostxfer.h"

#include <cstdlib>
#include <string>
#include <vector>

namespace dto {

// ============================================================================================
// small helpers
// ============================================================================================
__device__ __forceinline__ unsigned long long dbits(double v) { return (unsigned long long)__double_as_longlong(fabs(v)); }
__device__ __forceinline__ double bits_to_d(unsigned long long b) { return __longlong_as_double((long long)b); }

__device__ __forceinline__ void hess_add(const KProb& P, double* H, int64_t kn, int a, int b, double v);

// block size of the assembly kernels; option "debug_bad_launch" asks for one the hardware does not have, so that the
// tests can see a rejected launch come back through the C ABI as an error
static inline dim3 blk(const KProb& P, int threads) { return dim3(P.debug_bad_launch ? 4096 : threads); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ double block_sum_256(double v, double* sm) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ void k_fill(double* p, int64_t n, double v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}
void launch_fill(hipStream_t st, double* p, int64_t n, double v) {
    if (n <= 0) return;
    int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, st, p, n, v);
}

__global__ void k_add(double* __restrict__ dst, const double* __restrict__ src, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] += src[i];
}
__global__ void k_bits_equal(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ b, int64_t n,
                             int32_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) *flag = 0;
}
void launch_bits_equal(hipStream_t st, const double* a, const double* b, int64_t n, int32_t* flag) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_bits_equal, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const unsigned long long*>(a), reinterpret_cast<const unsigned long long*>(b), n, flag);
}
// packed[poff[r] + i] = slab[start[r] + i]: the variable runs of a value slab, back to back (dto_hostxfer.h).  One
// wavefront per run (a run is typically one 256-entry column of -E_k or a handful of entries).
__global__ void __launch_bounds__(256) k_pack_runs(const double* __restrict__ slab, const int64_t* __restrict__ start,
                                                   const int64_t* __restrict__ len, const int64_t* __restrict__ poff, int64_t n_runs,
                                                   double* __restrict__ packed) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_runs) return;
    const int lane = threadIdx.x & 63;
    const double* src = slab + start[r];
    double* dst = packed + poff[r];
    const int64_t n = len[r];
    for (int64_t i = lane; i < n; i += 64) dst[i] = __builtin_nontemporal_load(&src[i]);
}
void launch_pack_runs(hipStream_t st, const double* slab, const int64_t* start, const int64_t* len, const int64_t* poff,
                      int64_t n_runs, double* packed) {
    if (n_runs <= 0) return;
    hipLaunchKernelGGL(k_pack_runs, dim3((unsigned)((n_runs + 3) / 4)), dim3(256), 0, st, slab, start, len, poff, n_runs, packed);
}

// dst[start[r] .. start[r] + len[r]) = 0 for every run: one wavefront per run (bound outputs: the runs kernels accumulate into)
__global__ void __launch_bounds__(256) k_zero_runs(const int64_t* __restrict__ start, const int64_t* __restrict__ len, int64_t n_runs,
                                                   double* __restrict__ dst) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_runs) return;
    double* p = dst + start[r];
    const int64_t n = len[r];
    for (int64_t i = threadIdx.x & 63; i < n; i += 64) p[i] = 0.0;
}
void launch_zero_runs(hipStream_t st, const int64_t* start, const int64_t* len, int64_t n_runs, double* dst) {
    if (n_runs <= 0) return;
    hipLaunchKernelGGL(k_zero_runs, dim3((unsigned)((n_runs + 3) / 4)), dim3(256), 0, st, start, len, n_runs, dst);
}

void launch_add(hipStream_t st, double* dst, const double* src, int64_t n) {
    if (n <= 0) return;
    int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_add, dim3(grid), dim3(256), 0, st, dst, src, n);
}

// coef[b][alpha] = dt^r * prod_{i in alpha} ubar_i  (zero for padded intervals / padded multisets)
// K mode (taylor != nullptr): the set holds every multiset of degree 0..4 (idx padded with -1) and the
// coefficient of a degree-r multiset is k_r sigma^{4+r} dt^r prod ubar, i.e. the output is the right factor
// K = k0 I + k1 A + ... + k4 A^4 of the two-product Taylor evaluation (Y = A^4 K).
__global__ void k_basis_coef(KProb P, KBil B, BasisSet bs, const double* __restrict__ Z, int64_t int0, int nb,
                             const double* __restrict__ taylor) {
    const int b = blockIdx.x;
    double* c = bs.coef + (int64_t)b * bs.cntpad;
    if (b >= nb) {
        for (int a = threadIdx.x; a < bs.cntpad; a += blockDim.x) c[a] = 0.0;
        return;
    }
    const double* zk = Z + (int0 + b) * P.z;
    const double dt = zk[P.dt_idx];
    for (int a = threadIdx.x; a < bs.cntpad; a += blockDim.x) {
        double v = 0.0;
        if (a < bs.cnt) {
            v = 1.0;
            int deg = 0;
            for (int i = 0; i < bs.r; ++i) {
                const int g = bs.idx[a * bs.r + i];
                if (g < 0) continue;
                ++deg;
                v *= dt;
                if (g > 0) v *= zk[B.u_off + g - 1];
            }
            if (taylor) v *= taylor[(int64_t)b * COEF_STRIDE + COEF_K + deg];
        }
        c[a] = v;
    }
}
void launch_basis_coef(hipStream_t st, const KProb& P, const KBil& B, const BasisSet& bs, const double* dZ,
                       int64_t int0, int nb, int nbpad, const double* taylor) {
    hipLaunchKernelGGL(k_basis_coef, dim3(nbpad), dim3(128), 0, st, P, B, bs, dZ, int0, nb, taylor);
}

// Epilogue shared by the generator-subspace GEMMs: store the tile (out may be null: norms only) and leave the column
// abs-sums.  A wave's 64 rows are ONE 64-row chunk of one column of the npad x npad matrix (npad multiple of 64 and of the
// wave tile): the 16 lanes that share an interval reduce among themselves and STORE the chunk's partial sum --
// colsum[interval][matrix column][chunk], every slot written exactly once per launch; k_norm_from_colsum* adds the chunks in
// a fixed order.  (Up to round 2 the chunks met in one slot by atomicAdd: the 1-norms, and with them the squaring counts and
// the evaluation form at a boundary, depended on the order the waves arrived in.)
template <class Cfg>
__device__ __forceinline__ void basis_epilogue(const GemmAccS<Cfg>& acc, int npad, int nb, int rt, int ct, double* __restrict__ out,
                                               double* __restrict__ colsum) {
    constexpr int TM = Cfg::TM, TN = Cfg::TN;
    static_assert(Cfg::WTM == 64, "column sums assume 64-row wave tiles");
    const int64_t nn = (int64_t)npad * npad;
    const int wave_row0 = rt * TM + ((threadIdx.x >> 6) / Cfg::WC) * Cfg::WTM;
    if constexpr (Cfg::PAIRED) {
        GemmCoordP<Cfg> co;
        const int row0 = rt * TM + co.row_base, col0 = ct * TN + co.col_base;
#pragma unroll
        for (int tj = 0; tj < Cfg::NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = col0 + 16 * tj + 4 * r;
                double asum = 0.0;
#pragma unroll
                for (int p = 0; p < Cfg::MT / 2; ++p) {
                    const d2 v = {acc.v[2 * p][tj][r], acc.v[2 * p + 1][tj][r]};
                    if (out && col < nb) __builtin_nontemporal_store(v, reinterpret_cast<d2*>(&out[(int64_t)col * nn + row0 + 32 * p]));
                    asum += fabs(v.x) + fabs(v.y);
                }
                if (colsum) {
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) asum += __shfl_xor(asum, o, 64);
                    if ((threadIdx.x & 15) == 0 && col < nb) colsum[((int64_t)col * npad + wave_row0 / npad) * (npad / 64) + (wave_row0 % npad) / 64] = asum;
                }
            }
    } else {
        GemmCoordS<Cfg> co;
        const int row0 = rt * TM + co.row_base, col0 = ct * TN + co.col_base;
#pragma unroll
        for (int tj = 0; tj < Cfg::NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = col0 + 16 * tj + 4 * r;
                double asum = 0.0;
#pragma unroll
                for (int ti = 0; ti < Cfg::MT; ++ti) {
                    if (out && col < nb) __builtin_nontemporal_store(acc.v[ti][tj][r], &out[(int64_t)col * nn + row0 + 16 * ti]);
                    asum += fabs(acc.v[ti][tj][r]);
                }
                if (colsum) {
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) asum += __shfl_xor(asum, o, 64);
                    if ((threadIdx.x & 15) == 0 && col < nb) colsum[((int64_t)col * npad + wave_row0 / npad) * (npad / 64) + (wave_row0 % npad) / 64] = asum;
                }
            }
    }
}
template <class Cfg>
__device__ __forceinline__ void basis_accumulate(GemmAccS<Cfg>& acc, const BasisSet& bs, int64_t nn, int rt, int ct, double* smem) {
    const double* A = bs.S + (int64_t)rt * Cfg::TM;
    const double* B = bs.coef + (int64_t)ct * Cfg::TN * bs.cntpad;
    if constexpr (Cfg::PAIRED) gemm_accumulate_p<Cfg, true>(acc, A, (int)nn, B, bs.cntpad, bs.cntpad, smem);
    else gemm_accumulate_s<Cfg>(acc, A, (int)nn, B, bs.cntpad, bs.cntpad, nullptr, smem);
}

// out[:, b] = S * coef[:, b]  for the nb intervals of the chunk (FP64 MFMA, same GEMM core).
template <class Cfg>
__global__ void __launch_bounds__(Cfg::THREADS, 2) k_basis_gemm(int npad, int nb, BasisSet bs, double* __restrict__ out,
                                                                double* __restrict__ colsum) {
    __shared__ __attribute__((aligned(1024))) double smem[Cfg::SMEM_DOUBLES];
    const int64_t nn = (int64_t)npad * npad;
    const int row_tiles = (int)(nn / Cfg::TM);
    const int rt = blockIdx.x % row_tiles, ct = blockIdx.x / row_tiles;
    GemmAccS<Cfg> acc;
    acc.zero();
    basis_accumulate<Cfg>(acc, bs, nn, rt, ct, smem);
    basis_epilogue<Cfg>(acc, npad, nb, rt, ct, out, colsum);
}
__global__ void k_norm_from_colsum(int npad, int nb, const double* __restrict__ colsum, double* __restrict__ norms, int which,
                                   unsigned long long* d2max) {
    const int b = blockIdx.x;
    double m = 0.0;
    const int nch = npad / 64;
    auto colsum_of = [&](int c) {  // the chunks of one matrix column, added in a fixed order
        double v = 0.0;
        for (int q = 0; q < nch; ++q) v += colsum[((int64_t)b * npad + c) * nch + q];
        return v;
    };
    for (int c = threadIdx.x; c < npad; c += 64) m = fmax(m, colsum_of(c));
    m = wave_max(m);
    // NaN anywhere in the matrix must reach the scaling decision (fmax drops it)
    double bad = 0.0;
    for (int c = threadIdx.x; c < npad; c += 64) { const double v = colsum_of(c); if (!(v == v)) bad = 1.0; }
    bad = wave_max(bad);
    if (threadIdx.x == 0) {
        const double v = bad > 0.0 ? __longlong_as_double(0x7ff8000000000000ll) : m;
        norms[b * 4 + which] = v;
        if (d2max) atomicMax(d2max, dbits(sqrt(v)));  // a NaN's bit pattern exceeds every finite one
    }
}
void launch_norm_from_colsum(hipStream_t st, int npad, int nb, const double* colsum, double* norms, int which,
                             unsigned long long* d2max) {
    hipLaunchKernelGGL(k_norm_from_colsum, dim3(nb), dim3(64), 0, st, npad, nb, colsum, norms, which, d2max);
}
struct BasisMulti {
    int nsets;
    BasisSet bs[4];
    double* out[4];
    double* colsum[4];  // null: no column sums for that set
};
// Several basis sets in one grid: block -> (set, tile) with the set index fastest, so neighbouring workgroups mix the
// write-bound sets with the MFMA-bound ones.
template <class Cfg>
__global__ void __launch_bounds__(Cfg::THREADS, 2) k_basis_gemm_multi(int npad, int nb, BasisMulti M) {
    __shared__ __attribute__((aligned(1024))) double smem[Cfg::SMEM_DOUBLES];
    const int which = blockIdx.x % M.nsets, tile = blockIdx.x / M.nsets;
    BasisSet bs = M.bs[0];
    double* out = M.out[0];
    double* colsum = M.colsum[0];
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q == which) { bs = M.bs[q]; out = M.out[q]; colsum = M.colsum[q]; }
    const int64_t nn = (int64_t)npad * npad;
    const int row_tiles = (int)(nn / Cfg::TM);
    const int rt = tile % row_tiles, ct = tile / row_tiles;
    GemmAccS<Cfg> acc;
    acc.zero();
    basis_accumulate<Cfg>(acc, bs, nn, rt, ct, smem);
    basis_epilogue<Cfg>(acc, npad, nb, rt, ct, out, colsum);
}
void launch_basis_gemm_multi(hipStream_t st, int npad, int nb, int nbpad, int nsets, const BasisSet* bs, double* const* out,
                             double* const* colsum) {
    const int64_t nn = (int64_t)npad * npad;
    BasisMulti M{};
    M.nsets = nsets;
    for (int q = 0; q < nsets; ++q) { M.bs[q] = bs[q]; M.out[q] = out[q]; M.colsum[q] = colsum[q]; }
    const dim3 grid((unsigned)((nn / 128) * (nbpad / 128) * nsets));
    static const int core = tune_int("DTO_BASIS_CORE", 1);  // 1: paired-rows core (round 3), 0: the 8-byte core
    if (core == 1) {
        using C = GemmShapeP<128, 128, 2, 4>;
        hipLaunchKernelGGL((k_basis_gemm_multi<C>), grid, dim3(C::THREADS), 0, st, npad, nb, M);
    } else {
        using C = GemmShape<128, 128, 2, 4, 16>;
        hipLaunchKernelGGL((k_basis_gemm_multi<C>), grid, dim3(C::THREADS), 0, st, npad, nb, M);
    }
}
__global__ void k_basis_coef_multi(KProb P, KBil B, BasisMulti M, const double* __restrict__ Z, int64_t int0, int nb) {
    const int b = blockIdx.x;
    BasisSet bs = M.bs[0];
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q == (int)blockIdx.y) bs = M.bs[q];
    double* c = bs.coef + (int64_t)b * bs.cntpad;
    if (b >= nb) {
        for (int a = threadIdx.x; a < bs.cntpad; a += blockDim.x) c[a] = 0.0;
        return;
    }
    const double* zk = Z + (int0 + b) * P.z;
    const double dt = zk[P.dt_idx];
    for (int a = threadIdx.x; a < bs.cntpad; a += blockDim.x) {
        double v = 0.0;
        if (a < bs.cnt) {
            v = 1.0;
            for (int i = 0; i < bs.r; ++i) {
                const int g = bs.idx[a * bs.r + i];
                v *= dt;
                if (g > 0) v *= zk[B.u_off + g - 1];
            }
        }
        c[a] = v;
    }
}
void launch_basis_coef_multi(hipStream_t st, const KProb& P, const KBil& B, int nsets, const BasisSet* bs, const double* dZ,
                             int64_t int0, int nb, int nbpad) {
    BasisMulti M{};
    M.nsets = nsets;
    for (int q = 0; q < nsets; ++q) M.bs[q] = bs[q];
    hipLaunchKernelGGL(k_basis_coef_multi, dim3(nbpad, nsets), dim3(128), 0, st, P, B, M, dZ, int0, nb);
}
__global__ void k_norm_from_colsum_multi(int npad, int nb, BasisMulti M, double* __restrict__ norms) {
    const int b = blockIdx.x, which = blockIdx.y;
    const double* colsum = M.colsum[0];
#pragma unroll
    for (int q = 1; q < 3; ++q)
        if (q == which) colsum = M.colsum[q];
    double m = 0.0, bad = 0.0;
    const int nch = npad / 64;
    for (int c = threadIdx.x; c < npad; c += 64) {
        double v = 0.0;
        for (int q = 0; q < nch; ++q) v += colsum[((int64_t)b * npad + c) * nch + q];  // fixed order
        m = fmax(m, v);
        if (!(v == v)) bad = 1.0;  // NaN anywhere in the matrix must reach the scaling decision (fmax drops it)
    }
    m = wave_max(m);
    bad = wave_max(bad);
    if (threadIdx.x == 0) norms[b * 4 + 1 + which] = bad > 0.0 ? __longlong_as_double(0x7ff8000000000000ll) : m;
}
void launch_norm_from_colsum_multi(hipStream_t st, int npad, int nb, int nsets, double* const* colsum, double* norms) {
    BasisMulti M{};
    M.nsets = nsets;
    for (int q = 0; q < nsets; ++q) M.colsum[q] = colsum[q];
    hipLaunchKernelGGL(k_norm_from_colsum_multi, dim3(nb, nsets), dim3(64), 0, st, npad, nb, M, norms);
}
void launch_basis_gemm(hipStream_t st, int npad, int nb, int nbpad, const BasisSet& bs, double* out, double* colsum) {
    const int64_t nn = (int64_t)npad * npad;
    // 8 waves of 64x32 per 128x128 tile: the K loop is only 1..8 panels long, more waves hide its prologue and the
    // store-heavy epilogue better than 4 waves of 64x64 (measured -7 % per launch at 256x2000; 128x64 tiles +9 %)
    const dim3 grid((unsigned)((nn / 128) * (nbpad / 128)));
    static const int core = tune_int("DTO_BASIS_CORE", 1);
    if (core == 1) {
        using C = GemmShapeP<128, 128, 2, 4>;
        hipLaunchKernelGGL((k_basis_gemm<C>), grid, dim3(C::THREADS), 0, st, npad, nb, bs, out, colsum);
    } else {
        using C = GemmShape<128, 128, 2, 4, 16>;
        hipLaunchKernelGGL((k_basis_gemm<C>), grid, dim3(C::THREADS), 0, st, npad, nb, bs, out, colsum);
    }
}

// One wavefront per column c of the owned knots: walk the column's entries in the structure's order
// (per integrator: rows of interval k-1 then of interval k; then the constraint rows).
template <int TRANSPOSE>
__global__ void __launch_bounds__(64) k_jac_spmv(KProb P, KIntegTable T, const int64_t* __restrict__ conbase,
                                                 const int64_t* __restrict__ con_rows, const double* __restrict__ vals,
                                                 const double* __restrict__ w, double* __restrict__ y) {
    const int64_t cl = blockIdx.x;                 // local column
    const int64_t c = P.kn_lo * P.z + cl;          // global column
    const int64_t kn = c / P.z;
    const int has_prev = kn >= 1 && kn < P.N, has_own = kn < P.K;  // kn >= N: global-variable columns (constraint rows only)
    const int cnt = has_prev + has_own;
    const int64_t e0 = P.colptr[c], e1 = P.colptr[c + 1];
    const int64_t Lint = (int64_t)P.D * cnt;
    static_assert(TRANSPOSE == 1, "J w takes the row gather (launch_jac_rowgather): a column walk would meet other columns in y[row]");
    double acc = 0.0;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 64) {
        const int64_t el = e - e0;
        int64_t row;
        if (el < Lint) {
            int64_t rem = el;
            int i = 0;
            while (i < T.n - 1 && rem >= (int64_t)T.d[i] * cnt) { rem -= (int64_t)T.d[i] * cnt; ++i; }
            const int d = T.d[i];
            if (has_prev && rem < d) row = T.off[i] + (kn - 1) * d + rem;
            else row = T.off[i] + kn * d + (rem - (has_prev ? d : 0));
        } else {
            row = con_rows[conbase[c] + (el - Lint)];
        }
        const double v = vals[e - P.jac_lo];
        acc += v * w[row];
    }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) y[c] = acc;
}
// y = J w from the value slab ROW BY ROW (no atomics: the order of every row's sum is fixed).  An integrator row (integrator i,
// interval k, r) has its 2z entries at closed-form positions: part 1 of the columns of knot k, part 0 of those of knot k + 1; one
// wavefront per row takes them in ascending column order per lane and a fixed shuffle tree.
__global__ void __launch_bounds__(64) k_jac_rowgather_integ(KProb P, KIntegTable T, const double* __restrict__ vals,
                                                            const double* __restrict__ w, double* __restrict__ y) {
    const int64_t R = blockIdx.x;
    int i = 0, pre = 0;
    while (i < T.n - 1 && R >= T.off[i] + (int64_t)T.d[i] * P.K) { pre += T.d[i]; ++i; }
    const int d = T.d[i];
    const int64_t kn = (R - T.off[i]) / d;
    const int r = (int)((R - T.off[i]) % d);
    double acc = 0.0;
    for (int j = threadIdx.x; j < P.z; j += 64) {
        acc += vals[jac_pos(P, P.colptr, kn, j, pre, d, 1, r)] * w[kn * P.z + j];
        acc += vals[jac_pos(P, P.colptr, kn + 1, j, pre, d, 0, r)] * w[(kn + 1) * P.z + j];
    }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) y[R] = acc;
}
// ... a constraint row takes its entries from the row-ordered copy of the constraint pattern (ascending column)
__global__ void __launch_bounds__(64) k_jac_rowgather_con(const int64_t* __restrict__ rptr, const int64_t* __restrict__ rcol,
                                                          const int64_t* __restrict__ rpos, int64_t row0, const double* __restrict__ vals,
                                                          const double* __restrict__ w, double* __restrict__ y) {
    const int64_t r = blockIdx.x;
    double acc = 0.0;
    for (int64_t e = rptr[r] + threadIdx.x; e < rptr[r + 1]; e += 64) acc += vals[rpos[e]] * w[rcol[e]];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) y[row0 + r] = acc;
}
void launch_jac_rowgather(hipStream_t st, const KProb& P, const KIntegTable& T, int64_t n_con_rows, const int64_t* rptr, const int64_t* rcol,
                          const int64_t* rpos, int64_t n_dyn, const double* vals, const double* w, double* y) {
    if (T.n > 0 && n_dyn > 0) hipLaunchKernelGGL(k_jac_rowgather_integ, dim3((unsigned)n_dyn), dim3(64), 0, st, P, T, vals, w, y);
    if (n_con_rows > 0) hipLaunchKernelGGL(k_jac_rowgather_con, dim3((unsigned)n_con_rows), dim3(64), 0, st, rptr, rcol, rpos, n_dyn, vals, w, y);
}
void launch_jac_spmv(hipStream_t st, const KProb& P, const KIntegTable& T, const int64_t* conbase, const int64_t* con_rows,
                     const double* vals, const double* w, double* y, int transpose, int64_t global_cols) {
    const int64_t ncols = P.n_knots * P.z + global_cols;
    if (ncols <= 0) return;
    (void)transpose;  // J' w only: one wavefront per column; J w is launch_jac_rowgather
    hipLaunchKernelGGL(k_jac_spmv<1>, dim3((unsigned)ncols), dim3(64), 0, st, P, T, conbase, con_rows, vals, w, y);
}

// Zero-fill of the Jacobian slab (fill!(∂, 0), evaluator.jl:497) that skips the -E_k block of one bilinear
// integrator: the propagator chain overwrites every entry of that block anyway (1 GB of 2.2 GB at 256x2000).
__global__ void __launch_bounds__(256) k_jac_zero(KProb P, KBil B, double* __restrict__ vals) {
    const int64_t kn = P.kn_lo + blockIdx.x;
    const int has_prev = kn >= 1, has_own = kn < P.K;
    const int cnt = has_prev + has_own;
    const bool skipE = has_own && (int64_t)blockIdx.x < P.n_int;
    const int64_t e_lo = (int64_t)B.pre * cnt + (has_prev ? B.n : 0);  // offset of the E rows inside an x column
    // 16-byte stores over the 16-byte-aligned interior of [lo, hi), scalar stores at an odd head / tail; a column holds
    // a few hundred entries, so each half of the workgroup takes its own column (one wavefront per column with
    // non-temporal stores and eight workgroups per knot was measured 14 % slower)
    const int lane = threadIdx.x & 127, half = threadIdx.x >> 7;
    auto zero_range = [&](int64_t lo, int64_t hi) {
        if (hi <= lo) return;
        const int64_t a = lo + (int64_t)((reinterpret_cast<uintptr_t>(vals + lo) >> 3) & 1);
        if (lane == 0 && a > lo) vals[lo] = 0.0;
        const int64_t npair = hi > a ? (hi - a) / 2 : 0;
        d2* q = reinterpret_cast<d2*>(vals + a);
        for (int64_t i = lane; i < npair; i += 128) q[i] = d2{0.0, 0.0};
        if (lane == 0 && a + 2 * npair < hi) vals[hi - 1] = 0.0;
    };
    for (int j = 2 * blockIdx.y + half; j < P.z; j += 2 * gridDim.y) {
        const int64_t c = kn * P.z + j;
        const int64_t e0 = P.colptr[c] - P.jac_lo, len = P.colptr[c + 1] - P.colptr[c];
        const bool xcol = skipE && j >= B.x_off && j < B.x_off + B.n;
        if (xcol) {
            zero_range(e0, e0 + e_lo);
            zero_range(e0 + e_lo + B.n, e0 + len);
        } else {
            zero_range(e0, e0 + len);
        }
    }
}
void launch_jac_zero(hipStream_t st, const KProb& P, const KBil& B, double* vals) {
    if (P.n_knots <= 0) return;
    int gy = P.z < 32 ? P.z : 32;
    hipLaunchKernelGGL(k_jac_zero, dim3((unsigned)P.n_knots, gy), blk(P, 256), 0, st, P, B, vals);
}

__global__ void k_norm_bounds(KProb P, KBil B, const double* __restrict__ Z, const double* __restrict__ g1,
                              const double* __restrict__ n2, unsigned long long* out2) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double beta = 0.0, b1 = 0.0;
    if (k < P.n_int) {
        const double* zk = Z + (P.kn_lo + k) * P.z;
        const double dt = fabs(zk[P.dt_idx]);
        double ub[MAX_DRIVES + 1];
        ub[0] = 1.0;
        for (int j = 0; j < B.m; ++j) ub[j + 1] = fabs(zk[B.u_off + j]);
        double s2 = 0.0;
        for (int i = 0; i <= B.m; ++i) {
            b1 += ub[i] * g1[i];
            for (int j = 0; j <= B.m; ++j) s2 += ub[i] * ub[j] * n2[i * (B.m + 1) + j];
        }
        b1 *= dt;
        const double b2 = dt * sqrt(s2);
        beta = b1 < b2 ? b1 : b2;
        if (!(b1 == b1)) { beta = b1; }
    }
    // NaN compares false in fmax-trees: keep it by bit pattern (NaN bits exceed every finite value)
    unsigned long long vb = dbits(beta), v1 = dbits(b1);
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ob = __shfl_xor(vb, o, 64), o1 = __shfl_xor(v1, o, 64);
        vb = ob > vb ? ob : vb;
        v1 = o1 > v1 ? o1 : v1;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&out2[0], vb);
        atomicMax(&out2[1], v1);
    }
}
void launch_norm_bounds(hipStream_t st, const KProb& P, const KBil& B, const double* dZ, const double* g1,
                        const double* n2, unsigned long long* out2) {
    if (P.n_int <= 0) return;
    hipLaunchKernelGGL(k_norm_bounds, dim3((unsigned)((P.n_int + 255) / 256)), dim3(256), 0, st, P, B, dZ, g1, n2, out2);
}

// ============================================================================================
// propagator chain
// ============================================================================================

// A_b = dt * (G_0 + sum_j u_j G_j) for interval int0+b (padded npad x npad, column-major).
// G(u) is rebuilt by the reference on every call (bilinear_integrator.jl:81); here it is one
// coalesced streaming pass over the shared generators.
__global__ void __launch_bounds__(256) k_build_A(KProb P, KBil B, const double* __restrict__ Z, int64_t int0,
                                                  double* __restrict__ A) {
    const int b = blockIdx.y;
    const int64_t kn = int0 + b;
    const double* zk = Z + kn * P.z;
    const double dt = zk[P.dt_idx];
    double ub[MAX_DRIVES + 1];
    ub[0] = dt;
    for (int j = 0; j < B.m; ++j) ub[j + 1] = dt * zk[B.u_off + j];
    const int64_t nn = (int64_t)B.npad * B.npad;
    double* Ab = A + (int64_t)b * nn;
#pragma unroll 4
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nn / 2; i += (int64_t)gridDim.x * 256) {
        d2 acc = {0.0, 0.0};
        for (int j = 0; j <= B.m; ++j) {
            const d2 g = reinterpret_cast<const d2*>(B.G + (int64_t)j * nn)[i];
            acc.x += ub[j] * g.x;
            acc.y += ub[j] * g.y;
        }
        __builtin_nontemporal_store(acc, &reinterpret_cast<d2*>(Ab)[i]);
    }
}
void launch_build_A(hipStream_t st, const KProb& P, const KBil& B, const double* dZ, int64_t int0, int nb, double* A) {
    const int64_t nn2 = (int64_t)B.npad * B.npad / 2;
    int gx = (int)((nn2 + 255) / 256);
    static const int cap = tune_int("DTO_BUILDA_GX", 16);
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(k_build_A, dim3(gx, nb), dim3(256), 0, st, P, B, dZ, int0, A);
}

enum { EPI_PLAIN = 0, EPI_HORNER = 1, EPI_SQUARE = 2, EPI_DUAL = 3, EPI_DUAL5 = 4 };  // HORNER: + degree-4 polynomial; DUAL: two such outputs; DUAL5: each also + a multiple of M5
constexpr bool epi_poly(int e) { return e == EPI_HORNER || e == EPI_DUAL || e == EPI_DUAL5; }
constexpr bool epi_dual(int e) { return e == EPI_DUAL || e == EPI_DUAL5; }

struct BGemmArgs {
    const double* A;
    const double* B;
    double* C;
    int npad, nbatch;
    // poly: C = A*B + c0 I + c1 M1 + c2 M2 + c3 M3 + c4 M4, and (C2 != nullptr) C2 = A*B + the same with the
    // coefficients at coef_base2: both factors of the two-product Taylor evaluation leave one launch
    const double* M1;
    const double* M2;
    const double* M3;
    const double* M4;
    const double* M5;   // optional sixth term (the second product adds a multiple of its own left operand)
    const double* coef;
    int coef_base, coef_base2;
    double* C2;
    // square
    const int32_t* s;
    int it;
    KProb P;
    KBil Bi;
    int64_t int0;
    double* vals;
#ifdef DTO_TUNING
    int stamp_detail;             // 1: also the per-panel waits (perturbs the loop)
    unsigned long long* stamps;   // phase stamps, 8 per tile (tools/stamp_analyze.py): HW_ID | XCC_ID << 32, cycles at start / loop end / end, 100 MHz ticks at start / end, cycles waiting for panel loads / at the loop barrier
#endif
};

// Batched C_b = A_b * B_b over nbatch intervals, npad x npad x npad each (FP64 MFMA).
// PERSISTENT: the grid is sized to the chip (2 workgroups per CU) and every workgroup walks the
// (interval, tile) list with a stride of gridDim.x; this removes the workgroup re-dispatch gaps that a
// one-tile-per-workgroup grid of ~8000 short workgroups shows (measured: 25 % of wall time).
template <class Cfg, int EPI, bool DMA = false>
__global__ void __launch_bounds__(Cfg::THREADS, (Cfg::THREADS / 256) * (Cfg::SMEM_DOUBLES * 8 > 80 * 1024 ? 1 : 2))
k_bgemm(BGemmArgs a) {
    __shared__ __attribute__((aligned(1024))) double smem[Cfg::SMEM_DOUBLES];
    constexpr int TM = Cfg::TM, TN = Cfg::TN;
    const int tiles_r = a.npad / TM;
    const int tpm = tiles_r * (a.npad / TN);
    const int total = batch_tile_count(a.nbatch, tpm);
    const int64_t nn = (int64_t)a.npad * a.npad;
    GemmCoordS<Cfg> co;
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
        int b, tile;
        if (!decode_batch_tile(v, a.nbatch, tpm, b, tile)) continue;
        int s_b = 0;
        if (EPI == EPI_SQUARE) {
            s_b = a.s[b];
            if (a.it >= s_b) continue;  // this interval needs no further squaring
        }
        const int tr = tile % tiles_r, tc = tile / tiles_r;
        const double* Ab = a.A + b * nn + (int64_t)tr * TM;
        const double* Bb = a.B + b * nn + (int64_t)tc * TN * a.npad;

        GemmAccS<Cfg> acc;
        acc.zero();
        if constexpr (DMA) gemm_accumulate_dma<Cfg>(acc, Ab, a.npad, Bb, a.npad, a.npad, smem);
        else gemm_accumulate_s<Cfg>(acc, Ab, a.npad, Bb, a.npad, a.npad, nullptr, smem);

        const int row0 = tr * TM + co.row_base, col0 = tc * TN + co.col_base;

        if (EPI == EPI_SQUARE && a.it == s_b - 1) {
            // last squaring: the product is E_k; store -E_k into the Jacobian slab (x_k columns of the
            // interval's own rows), evaluator.jl:514-525 / bilinear_integrator.jl:111-131
            const int64_t kn = a.int0 + b;
            const int n = a.Bi.n;
#pragma unroll
            for (int tj = 0; tj < Cfg::NT; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = col0 + 16 * tj + 4 * r;
                    if (col >= n) continue;
                    const int64_t base = jac_pos(a.P, a.P.colptr, kn, a.Bi.x_off + col, a.Bi.pre, n, 1, 0);
#pragma unroll
                    for (int ti = 0; ti < Cfg::MT; ++ti) {
                        const int row = row0 + 16 * ti;
                        if (row < n) __builtin_nontemporal_store(-acc.v[ti][tj][r], &a.vals[base + row]);  // never re-read by the engine
                    }
                }
            continue;
        }

        double* Cb = a.C + b * nn;
        // the final polynomial product of an interval that needs no squaring IS exp(A_k): it goes into the Jacobian slab
        const bool to_slab = EPI == EPI_HORNER && a.vals != nullptr && a.s[b] == 0;
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0;
        const double *M1 = nullptr, *M2 = nullptr, *M3 = nullptr, *M4 = nullptr, *M5 = nullptr;
        double* Cb2 = nullptr;
        if (epi_poly(EPI)) {
            const double* cf = a.coef + (int64_t)b * COEF_STRIDE + a.coef_base;
            c0 = cf[0]; c1 = cf[1]; c2 = cf[2]; c3 = cf[3]; c4 = cf[4];
            M1 = a.M1 + b * nn; M2 = a.M2 + b * nn; M3 = a.M3 + b * nn; M4 = a.M4 + b * nn;

            if (epi_dual(EPI)) {
                const double* ef = a.coef + (int64_t)b * COEF_STRIDE + a.coef_base2;
                e0 = ef[0]; e1 = ef[1]; e2 = ef[2]; e3 = ef[3]; e4 = ef[4];
                Cb2 = a.C2 + b * nn;
                if (EPI == EPI_DUAL5) { M5 = a.M5 + b * nn; c5 = cf[5]; e5 = ef[5]; }
            }
        }
#pragma unroll
        for (int tj = 0; tj < Cfg::NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = col0 + 16 * tj + 4 * r;
                int64_t slab_base = 0;
                if (EPI == EPI_HORNER && to_slab && col < a.Bi.n)
                    slab_base = jac_pos(a.P, a.P.colptr, a.int0 + b, a.Bi.x_off + col, a.Bi.pre, a.Bi.n, 1, 0);
#pragma unroll
                for (int ti = 0; ti < Cfg::MT; ++ti) {
                    const int row = row0 + 16 * ti;
                    const int64_t off = (int64_t)col * a.npad + row;
                    double v2 = acc.v[ti][tj][r];
                    if (epi_poly(EPI)) {
                        // the second product of the three-product form has no A^4 term (EXPM3_*[4] = 0); in the first product of
                        // either form M4 is the launch's own A operand (L2-hot)
                        const double m1 = __builtin_nontemporal_load(&M1[off]), m2 = __builtin_nontemporal_load(&M2[off]),
                                     m3 = __builtin_nontemporal_load(&M3[off]), m4 = EPI == EPI_DUAL5 ? 0.0 : M4[off];
                        if (epi_dual(EPI)) {
                            double w2 = v2;
                            if (EPI == EPI_DUAL5) {
                                const double m5 = M5[off];  // this launch's A operand
                                w2 += e5 * m5;
                                v2 += c5 * m5;
                            }
                            w2 += e1 * m1 + e2 * m2 + e3 * m3 + e4 * m4;
                            if (row == col) w2 += e0;
                            __builtin_nontemporal_store(w2, &Cb2[off]);
                        }
                        v2 += c1 * m1 + c2 * m2 + c3 * m3 + c4 * m4;  // streamed once per stage
                        if (row == col) v2 += c0;
                    }
                    if (EPI == EPI_HORNER && to_slab) {
                        if (col < a.Bi.n && row < a.Bi.n) __builtin_nontemporal_store(-v2, &a.vals[slab_base + row]);
                        continue;
                    }
                    __builtin_nontemporal_store(v2, &Cb[off]);  // 1 GB per launch: gone from L2 before its reader starts
                }
                if (epi_dual(EPI)) __builtin_amdgcn_sched_barrier(0);  // keeps the loads of later columns from piling up (spills)
            }
    }
}

// Epilogue of the paired-rows accumulator layout, shared by k_bgemm_p and the ring kernel k_bgemm_r: tile (tr, tc) of interval b.
template <class Cfg, int EPI>
__device__ __forceinline__ void bgemm_p_epilogue(const BGemmArgs& a, const GemmAccS<Cfg>& acc, int b, int tr, int tc, int s_b, int64_t nn) {
    constexpr int TM = Cfg::TM, TN = Cfg::TN, MP = Cfg::MT / 2;
    GemmCoordP<Cfg> co;
    const int row0 = tr * TM + co.row_base, col0 = tc * TN + co.col_base;
    // -result into the Jacobian slab (x_k columns of the interval's own rows, evaluator.jl:514-525 /
    // bilinear_integrator.jl:111-131): the last squaring, or the last polynomial product of an interval that needs none
    const bool to_slab = (EPI == EPI_SQUARE && a.it == s_b - 1) || (EPI == EPI_HORNER && a.vals != nullptr && a.s[b] == 0);
    double* Cb = a.C + b * nn;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0;
    const double *M1 = nullptr, *M2 = nullptr, *M3 = nullptr, *M4 = nullptr, *M5 = nullptr;
    double* Cb2 = nullptr;
    if (epi_poly(EPI)) {
        const double* cf = a.coef + (int64_t)b * COEF_STRIDE + a.coef_base;
        c0 = cf[0]; c1 = cf[1]; c2 = cf[2]; c3 = cf[3]; c4 = cf[4];
        M1 = a.M1 + b * nn; M2 = a.M2 + b * nn; M3 = a.M3 + b * nn; M4 = a.M4 + b * nn;
        if (epi_dual(EPI)) {
            const double* ef = a.coef + (int64_t)b * COEF_STRIDE + a.coef_base2;
            e0 = ef[0]; e1 = ef[1]; e2 = ef[2]; e3 = ef[3]; e4 = ef[4];
            Cb2 = a.C2 + b * nn;
            if (EPI == EPI_DUAL5) { M5 = a.M5 + b * nn; c5 = cf[5]; e5 = ef[5]; }
        }
    }
#pragma unroll
    for (int tj = 0; tj < Cfg::NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = col0 + 16 * tj + 4 * r;
            int64_t slab_base = 0;
            if ((EPI == EPI_SQUARE || EPI == EPI_HORNER) && to_slab && col < a.Bi.n)
                slab_base = jac_pos(a.P, a.P.colptr, a.int0 + b, a.Bi.x_off + col, a.Bi.pre, a.Bi.n, 1, 0);
#pragma unroll
            for (int p = 0; p < MP; ++p) {
                const int row = row0 + 32 * p;  // even; the lane also holds row + 1
                const int64_t off = (int64_t)col * a.npad + row;
                d2 v2 = {acc.v[2 * p][tj][r], acc.v[2 * p + 1][tj][r]};
                if (epi_poly(EPI)) {
                    // the second product of the three-product form has no A^4 term (EXPM3_*[4] = 0); in the first product of
                    // either form M4 is the launch's own A operand (L2-hot)
                    const d2 m1 = __builtin_nontemporal_load(reinterpret_cast<const d2*>(&M1[off])),
                             m2 = __builtin_nontemporal_load(reinterpret_cast<const d2*>(&M2[off])),
                             m3 = __builtin_nontemporal_load(reinterpret_cast<const d2*>(&M3[off]));
                    const d2 m4 = EPI == EPI_DUAL5 ? d2{0.0, 0.0} : *reinterpret_cast<const d2*>(&M4[off]);
                    if (epi_dual(EPI)) {
                        d2 w2 = v2;
                        if (EPI == EPI_DUAL5) {
                            const d2 m5 = *reinterpret_cast<const d2*>(&M5[off]);  // this launch's A operand
                            w2 += e5 * m5;
                            v2 += c5 * m5;
                        }
                        w2 += e1 * m1 + e2 * m2 + e3 * m3 + e4 * m4;
                        if (row == col) w2.x += e0;
                        if (row + 1 == col) w2.y += e0;
                        __builtin_nontemporal_store(w2, reinterpret_cast<d2*>(&Cb2[off]));
                    }
                    v2 += c1 * m1 + c2 * m2 + c3 * m3 + c4 * m4;  // streamed once per stage
                    if (row == col) v2.x += c0;
                    if (row + 1 == col) v2.y += c0;
                }
                if ((EPI == EPI_SQUARE || EPI == EPI_HORNER) && to_slab) {
                    if (col < a.Bi.n) {  // slab columns are only 8-byte aligned; never re-read by the engine
                        if (row + 1 < a.Bi.n) __builtin_nontemporal_store(d2u{-v2.x, -v2.y}, reinterpret_cast<d2u*>(&a.vals[slab_base + row]));
                        else if (row < a.Bi.n) __builtin_nontemporal_store(-v2.x, &a.vals[slab_base + row]);
                    }
                    continue;
                }
                __builtin_nontemporal_store(v2, reinterpret_cast<d2*>(&Cb[off]));  // 1 GB per launch: gone from L2 before its reader starts
            }
            if (epi_dual(EPI)) __builtin_amdgcn_sched_barrier(0);  // keeps the loads of later columns from piling up (spills)
        }
}

// The same batched product on the paired-rows core (dto_gemm.hip.h, round 3): 16-byte fragment reads, A panels by LDS-DMA
// (DMA_A) or through registers, every epilogue access 16 bytes per lane.  Epilogue semantics are those of k_bgemm.
template <class Cfg, int EPI, bool DMA_A>
__global__ void __launch_bounds__(Cfg::THREADS, (Cfg::THREADS / 256) * 2)
k_bgemm_p(BGemmArgs a) {
    __shared__ __attribute__((aligned(1024))) double smem[Cfg::SMEM_DOUBLES];
    constexpr int TM = Cfg::TM, TN = Cfg::TN;
    const int tiles_r = a.npad / TM;
    const int tpm = tiles_r * (a.npad / TN);
    const int total = batch_tile_count(a.nbatch, tpm);
    const int64_t nn = (int64_t)a.npad * a.npad;
    for (int v = blockIdx.x; v < total; v += gridDim.x) {
        int b, tile;
        if (!decode_batch_tile(v, a.nbatch, tpm, b, tile)) continue;
        int s_b = 0;
        if (EPI == EPI_SQUARE) {
            s_b = a.s[b];
            if (a.it >= s_b) continue;  // this interval needs no further squaring
        }
        const int tr = tile % tiles_r, tc = tile / tiles_r;
        const double* Ab = a.A + b * nn + (int64_t)tr * TM;
        const double* Bb = a.B + b * nn + (int64_t)tc * TN * a.npad;
#ifdef DTO_TUNING
        unsigned long long st0 = 0, st1 = 0, rt0 = 0;
        if (a.stamps) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
#endif

        GemmAccS<Cfg> acc;
        acc.zero();
#ifdef DTO_TUNING
        unsigned long long waits[2] = {0, 0};
        gemm_accumulate_p<Cfg, DMA_A>(acc, Ab, a.npad, Bb, a.npad, a.npad, smem, a.stamps && a.stamp_detail ? waits : nullptr);
#else
        gemm_accumulate_p<Cfg, DMA_A>(acc, Ab, a.npad, Bb, a.npad, a.npad, smem);
#endif
#ifdef DTO_TUNING
        if (a.stamps) st1 = __builtin_amdgcn_s_memtime();
        struct StampAtExit {   // every path out of the tile body passes here
            unsigned long long *dst, st0, st1, rt0, w0, w1;
            __device__ ~StampAtExit() {
                if (!dst) return;
                __builtin_amdgcn_s_waitcnt(0);   // the tile's loads and stores have been issued and the loads are back
                if (threadIdx.x == 0) {
                    dst[0] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                             ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
                    dst[1] = st0; dst[2] = st1; dst[3] = __builtin_amdgcn_s_memtime(); dst[4] = rt0; dst[5] = __builtin_amdgcn_s_memrealtime();
                    dst[6] = w0; dst[7] = w1;
                }
            }
        } stamp_at_exit{a.stamps ? a.stamps + 8ll * v : nullptr, st0, st1, rt0, waits[0], waits[1]};
#endif

        bgemm_p_epilogue<Cfg, EPI>(a, acc, b, tr, tc, s_b, nn);
    }
}

// The same products on the ring core (dto_gemm_ring.hip.h): persistent, K panels of all of a workgroup's tiles in one LDS ring.
template <int EPI, int S>
__global__ void __launch_bounds__(512, 4) k_bgemm_r(BGemmArgs a) {
    using R = RingCore<8, S>;
    using Cfg = typename R::Cfg;
    __shared__ __attribute__((aligned(1024))) double smem[R::SMEM_DOUBLES];
    const int tiles_r = a.npad / 128;
    const int tpm = tiles_r * tiles_r;
    const int total = batch_tile_count(a.nbatch, tpm);
    const int64_t nn = (int64_t)a.npad * a.npad;
    R::run(smem, (int)blockIdx.x, (int)gridDim.x,
           [&](int& pos, RingTile& t) {
               int b = 0, tile = 0;
               for (; pos < total; pos += (int)gridDim.x) {
                   if (!decode_batch_tile(pos, a.nbatch, tpm, b, tile)) continue;
                   if (EPI == EPI_SQUARE && a.it >= a.s[b]) continue;  // this interval needs no further squaring
                   break;
               }
               if (pos >= total) return false;
               const int tr = tile % tiles_r, tc = tile / tiles_r;
               t.A = a.A + b * nn + (int64_t)tr * 128; t.lda = a.npad;
               t.B = a.B + b * nn + (int64_t)tc * 128 * a.npad; t.ldb = a.npad;
               t.nk = a.npad / 8;
               return true;
           },
           [&](int pos, const GemmAccS<Cfg>& acc) {
               int b = 0, tile = 0;
               decode_batch_tile(pos, a.nbatch, tpm, b, tile);
               bgemm_p_epilogue<Cfg, EPI>(a, acc, b, tile % tiles_r, tile / tiles_r, EPI == EPI_SQUARE ? a.s[b] : 0, nn);
           });
}

static int bgemm_shape_choice() {
    static int v = tune_int("DTO_BGEMM_SHAPE", -1);
    return v;
}
static int bgemm_dma_choice() {  // -1: per-epilogue default, 0/1: forced
    static int v = tune_int("DTO_BGEMM_DMA", -1);
    return v;
}
static int bgemm_wgs_choice() {  // -1: per-epilogue default, 0: one workgroup per tile, k: k persistent workgroups per CU
    static int v = tune_int("DTO_BGEMM_WGS_PER_CU", -1);
    return v;
}
// Measured in the engine at 256x2000 (ms per launch, fused-polynomial / squaring):
//   register staging, persistent 1.48 / 1.25    register staging, one WG per tile 1.37 / 1.27
//   DMA staging,      persistent 1.56 / 1.20    DMA staging,      one WG per tile 1.47 / 1.23
// so the polynomial products run one workgroup per tile with register staging, the plain products and the
// squarings run persistent with DMA staging.
template <class Cfg, int EPI>
static void launch_bgemm_shape(hipStream_t st, const BGemmArgs& a, int wgs_per_cu) {
    int grid = batch_tile_count(a.nbatch, (a.npad / Cfg::TM) * (a.npad / Cfg::TN));
    int wgs = bgemm_wgs_choice();
    if (wgs < 0) wgs = epi_poly(EPI) ? 0 : wgs_per_cu;
    else if (wgs > 0) wgs = wgs_per_cu;
    if (wgs > 0 && grid > wgs * 256) grid = wgs * 256;
    if constexpr (Cfg::TM == 128 && Cfg::TN == 128 && Cfg::KB == 16) {
        int dma = bgemm_dma_choice();
        if (dma < 0) dma = epi_poly(EPI) ? 0 : 1;
        if (dma) {
            hipLaunchKernelGGL((k_bgemm<Cfg, EPI, true>), dim3(grid), dim3(Cfg::THREADS), 0, st, a);
            return;
        }
    }
    hipLaunchKernelGGL((k_bgemm<Cfg, EPI, false>), dim3(grid), dim3(Cfg::THREADS), 0, st, a);
}
#ifdef DTO_TUNING
// DTO_STAMP_FILE=<path> DTO_STAMP_LAUNCH=<k>: launches k .. k+7 of the paired-core GEMM in this process record their phase stamps
// into <path>.<launch>.epi<EPI> (tools/stamp_analyze.py)
static unsigned long long* stamps_begin(hipStream_t st, const BGemmArgs& a, size_t& bytes, int& index) {
    static int launches = 0;
    static const int which = tune_int("DTO_STAMP_LAUNCH", -1);
    index = launches++;
    if (which < 0 || index < which || index >= which + 8 || !getenv("DTO_STAMP_FILE")) return nullptr;
    bytes = 8ull * 8 * batch_tile_count(a.nbatch, (a.npad / 128) * (a.npad / 128));
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
    (void)hipMemsetAsync(d, 0, bytes, st);
    return d;
}
static void stamps_end(hipStream_t st, unsigned long long* d, size_t bytes, int index, int epi) {
    if (!d) return;
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> h(bytes / 8);
    (void)hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost);
    const std::string path = std::string(getenv("DTO_STAMP_FILE")) + "." + std::to_string(index) + ".epi" + std::to_string(epi);
    if (FILE* f = fopen(path.c_str(), "wb")) { fwrite(h.data(), 1, bytes, f); fclose(f); }
    (void)hipFree(d);
}
#endif
// paired-rows core: persistent for the plain products and the squarings (A by DMA), one workgroup per tile for the
// polynomial products (their epilogues stream 3-4 more matrices; A through registers), as measured for the 8-byte core
template <class Cfg, int EPI>
static void launch_bgemm_p(hipStream_t st, const BGemmArgs& a_in) {
    BGemmArgs a = a_in;
#ifdef DTO_TUNING
    size_t stamp_bytes = 0;
    int stamp_index = 0;
    a.stamps = stamps_begin(st, a, stamp_bytes, stamp_index);
    a.stamp_detail = tune_int("DTO_STAMP_DETAIL", 0);
#endif
    int grid = batch_tile_count(a.nbatch, (a.npad / Cfg::TM) * (a.npad / Cfg::TN));
    int wgs = bgemm_wgs_choice();
    if (wgs < 0) wgs = epi_poly(EPI) ? 0 : 2;
    if (wgs > 0 && grid > wgs * 256) grid = wgs * 256;
    int dma = bgemm_dma_choice();
    if (dma < 0) dma = epi_poly(EPI) ? 0 : 1;
    if (dma) hipLaunchKernelGGL((k_bgemm_p<Cfg, EPI, true>), dim3(grid), dim3(Cfg::THREADS), 0, st, a);
    else hipLaunchKernelGGL((k_bgemm_p<Cfg, EPI, false>), dim3(grid), dim3(Cfg::THREADS), 0, st, a);
#ifdef DTO_TUNING
    stamps_end(st, a.stamps, stamp_bytes, stamp_index, EPI);
#endif
}
template <int EPI>
static void launch_bgemm(hipStream_t st, const BGemmArgs& a) {
    // 128x128 tiles (persistent, DMA-staged) win once the launch is large; short trajectories are better served by four
    // times as many 64x64 workgroups.  Measured crossover (tile-128 workgroups x K panels): 256x200 -17 % with 64-tiles,
    // 256x500 equal, 256x1000 and 512x100 +5..7 % with 128-tiles (DTO_BGEMM_TILE64=0/1 forces).
    static const int force64 = tune_int("DTO_BGEMM_TILE64", -1);
    const long t128 = a.npad / 128;
    // (with the ring core the 128-tiles win at every launch size from 256 states on, even 30 intervals: 256 x 250 2.12 -> 2.07 ms,
    // 256 x 100 1.35 -> 1.26 per Jacobian; at 128 states, one tile per matrix, the crossover stands -- gpurun_out/r03ag/ab_tile*.log)
    const bool small_launch = force64 >= 0 ? force64 != 0 : (t128 == 1 && a.nbatch < 3500);
    if (a.npad % 128 == 0 && small_launch && bgemm_shape_choice() < 0) {
        launch_bgemm_shape<GemmShape<64, 64, 2, 2, 16>, EPI>(st, a, 4);
        return;
    }
    // Ring core (K panels in an LDS ring, both operands by LDS-DMA, dto_gemm_ring.hip.h).  The plain products and the squarings run
    // it PERSISTENT (the next tile's panels in flight during the epilogue): 1.47 -> 1.39 ms per squaring at 256 x 2000, 2.28 -> 2.01
    // at 512 x 500, 8.16 -> 7.65 at 1024 x 500.  The polynomial products, bound by the bytes of their epilogues, keep one workgroup per
    // tile (persistent they lose 5-15 %) and still gain from the ring inside the tile: 1.60 -> 1.535 ms at 256 x 2000, 8.63 -> 8.29 at
    // 1024 x 500 (gpurun_out/r03ah/ab_ring*.log).
    static const int ring = tune_int("DTO_BGEMM_RING", -1);  // -1: as described, 0: never, 1: everything persistent
    if (a.npad % 128 == 0 && ring != 0 && bgemm_shape_choice() < 0) {
        int grid = batch_tile_count(a.nbatch, (a.npad / 128) * (a.npad / 128));
        if (grid > 2 * 256 && (ring == 1 || !epi_poly(EPI))) grid = 2 * 256;
        // (three or five slots, two or four tiles per workgroup for the polynomial products: all slower, gpurun_out/r03ah/ab_ring5.log)
        hipLaunchKernelGGL((k_bgemm_r<EPI, 4>), dim3(grid), dim3(512), 0, st, a);
        return;
    }
    static const int core = tune_int("DTO_BGEMM_CORE", 1);  // 1: paired-rows core (round 3), 0: the 8-byte core
    if (a.npad % 128 == 0 && core == 1 && bgemm_shape_choice() < 0) {
        static const int w8 = tune_int("DTO_BGEMM_P_WAVES8", -1);  // -1: 8 waves for the polynomial epilogues only
        if (w8 > 0 || (w8 < 0 && epi_poly(EPI))) launch_bgemm_p<GemmShapeP<128, 128, 2, 4>, EPI>(st, a);
        else launch_bgemm_p<GemmShapeP<128, 128, 2, 2>, EPI>(st, a);
        return;
    }
    if (a.npad % 256 == 0) {
        // shapes measured with tools/bgemm_probe2 (256x2000): see DESIGN.md
#ifdef DTO_TUNING  // the alternative shapes exist in `make TUNING=1` builds only
        switch (bgemm_shape_choice()) {
            case 0: launch_bgemm_shape<GemmShape<128, 128, 2, 2, 16>, EPI>(st, a, 2); return;
            case 1: launch_bgemm_shape<GemmShape<128, 128, 2, 2, 8>, EPI>(st, a, 2); return;
            case 2: launch_bgemm_shape<GemmShape<256, 128, 4, 2, 16>, EPI>(st, a, 1); return;
            case 3: launch_bgemm_shape<GemmShape<256, 128, 4, 2, 8>, EPI>(st, a, 1); return;
            case 4: launch_bgemm_shape<GemmShape<128, 256, 2, 4, 16>, EPI>(st, a, 1); return;
            case 5: launch_bgemm_shape<GemmShape<128, 256, 2, 4, 8>, EPI>(st, a, 1); return;
            case 6: launch_bgemm_shape<GemmShape<128, 128, 2, 4, 16>, EPI>(st, a, 2); return;
            case 7: launch_bgemm_shape<GemmShape<128, 128, 4, 2, 16>, EPI>(st, a, 2); return;
            case 8: launch_bgemm_shape<GemmShape<128, 128, 2, 4, 8>, EPI>(st, a, 3); return;
            default: break;
        }
#endif
        // measured in the engine (256x2000): the fused-polynomial epilogue hides better behind 8 waves,
        // the plain / squaring products run faster with 4 waves of 64x64
        if (epi_poly(EPI)) launch_bgemm_shape<GemmShape<128, 128, 2, 4, 16>, EPI>(st, a, 2);
        else launch_bgemm_shape<GemmShape<128, 128, 2, 2, 16>, EPI>(st, a, 2);
    } else if (a.npad % 128 == 0) {
        if (epi_poly(EPI)) launch_bgemm_shape<GemmShape<128, 128, 2, 4, 16>, EPI>(st, a, 2);
        else launch_bgemm_shape<GemmShape<128, 128, 2, 2, 16>, EPI>(st, a, 2);
    } else {
        launch_bgemm_shape<GemmShape<64, 64, 2, 2, 16>, EPI>(st, a, 4);
    }
}

void launch_bgemm_plain(hipStream_t st, int npad, int nb, const double* A, const double* Bm, double* C) {
    BGemmArgs a{};
    a.A = A; a.B = Bm; a.C = C; a.npad = npad; a.nbatch = nb;
    launch_bgemm<EPI_PLAIN>(st, a);
}
void launch_bgemm_poly(hipStream_t st, int npad, int nb, const ChainWork& w, int srcA, int srcB, int dst, int coef_base,
                       int dst2, int coef_base2, bool with_srcA, const SlabDest* slab) {
    BGemmArgs a{};
    if (slab) { a.s = w.s; a.P = slab->P; a.Bi = slab->B; a.int0 = slab->int0; a.vals = slab->vals; }
    a.A = w.W[srcA]; a.B = w.W[srcB]; a.C = w.W[dst]; a.npad = npad; a.nbatch = nb;
    a.M1 = w.W[0]; a.M2 = w.W[1]; a.M3 = w.W[2]; a.M4 = w.W[3]; a.coef = w.coef; a.coef_base = coef_base;
    if (dst2 >= 0) {
        a.C2 = w.W[dst2]; a.coef_base2 = coef_base2;
        if (with_srcA) {
            a.M5 = w.W[srcA];
            launch_bgemm<EPI_DUAL5>(st, a);
            return;
        }
        launch_bgemm<EPI_DUAL>(st, a);
        return;
    }
    launch_bgemm<EPI_HORNER>(st, a);
}

void launch_bgemm_square(hipStream_t st, int npad, int nb, const ChainWork& w, int src, int dst, int it,
                         const KProb& P, const KBil& B, int64_t int0, double* vals) {
    BGemmArgs a{};
    a.A = w.W[src]; a.B = w.W[src]; a.C = w.W[dst]; a.npad = npad; a.nbatch = nb;
    a.s = w.s; a.it = it; a.P = P; a.Bi = B; a.int0 = int0; a.vals = vals;
#ifdef DTO_TUNING
    if (tune_int("DTO_SQ_PLAIN", 0)) a.it = -1;  // timing experiment (wrong results): the last squaring stores into W like the others
#endif
    launch_bgemm<EPI_SQUARE>(st, a);
}

// 1-norms (max column abs sum) of A, A^2, A^3, A^4 per interval: exact inputs of the scaling choice.
__global__ void __launch_bounds__(256) k_norm1(int npad, ChainWork w, int only) {
    const int b = blockIdx.x, which = only >= 0 ? only : blockIdx.y;
    const double* M = w.W[which] + (int64_t)b * npad * npad;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double best = 0.0;
    for (int c = wave; c < npad; c += 4) {
        double s = 0.0;
        for (int r = lane; r < npad; r += 64) s += fabs(M[(int64_t)c * npad + r]);
        s = wave_sum(s);
        best = fmax(best, s);
    }
    __shared__ double sm[4];
    if (lane == 0) sm[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double nrm = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
        w.norms[b * 4 + which] = nrm;
        if (only == 1) atomicMax(w.d2max, dbits(sqrt(nrm)));
    }
}
// max_k min(d2_k, max(d3_k, d4_k)) from the exact norms of A^2, A^3, A^4 (non-chain callbacks at large norm)
__global__ void k_beta_from_norms(int nb, ChainWork w) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const double d2v = sqrt(w.norms[b * 4 + 1]), d3v = cbrt(w.norms[b * 4 + 2]), d4v = sqrt(sqrt(w.norms[b * 4 + 3]));
    double beta = fmin(d2v, fmax(d3v, d4v));
    if (!(d2v == d2v) || !(d3v == d3v) || !(d4v == d4v)) beta = __longlong_as_double(0x7ff8000000000000ll);
    atomicMax(w.d2max, dbits(beta));
}
void launch_beta_from_norms(hipStream_t st, int nb, const ChainWork& w) {
    hipLaunchKernelGGL(k_beta_from_norms, dim3((nb + 255) / 256), dim3(256), 0, st, nb, w);
}
// A-priori bound on the "hump" of the Taylor series of exp(A_k/q) v and on its length, per interval, from the norms the
// engine has: N_1 <= |dt| sum_g |ubar_g| ||G_g||_1 (or the exact ||A||_1 when norms[0] is finite) and the exact
// ||A^2||, ||A^3||, ||A^4|| where available (INFINITY = not computed).  Every power factorises over {1,2,3,4}:
//   ||A^k|| <= best[k] = min_p best[k-p] * N_p,      hump(q) = max_k best[k] / (q^k k!),
// a rigorous and, for non-normal A, much sharper growth measure than alpha_p^k / k! (Al-Mohy & Higham's alpha_p needs
// k >= p(p-1)).  log hump(q) bounds the digits the sums can lose to cancellation; the sweep runs in q rounds with the
// smallest q whose hump stays below e^theta_v.  out[q-1] = max_k log hump(q) (bit pattern of a non-negative double),
// out[4+q-1] = max_k (last term index with best[k]/(q^k k!) >= 1e-19), q = 1..4.
__global__ void __launch_bounds__(256) k_hump(KProb P, KBil B, const double* __restrict__ Z, const double* __restrict__ g1,
                                              int64_t int0, int nb, const double* __restrict__ norms,
                                              unsigned long long* __restrict__ out) {
    __shared__ double logk[161];  // log k: one log per thread once, instead of one per (interval, q, k)
    if (threadIdx.x <= 160) logk[threadIdx.x] = threadIdx.x ? log((double)threadIdx.x) : 0.0;
    __syncthreads();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;  // (interval, q): four threads per interval
    const int q = (gid & 3) + 1;
    const bool live = (gid >> 2) < nb;
    const int b = live ? gid >> 2 : nb - 1;  // surplus lanes repeat the last interval (they take part in the wave reduction)
    const double* zk = Z + (int0 + b) * P.z;
    double n1 = g1[0];
    for (int j = 0; j < B.m; ++j) n1 += fabs(zk[B.u_off + j]) * g1[1 + j];
    n1 *= fabs(zk[P.dt_idx]);
    double N[5];
    N[1] = fmin(n1, norms[b * 4 + 0]);  // fmin drops a NaN argument: the bound survives
    N[2] = norms[b * 4 + 1]; N[3] = norms[b * 4 + 2]; N[4] = norms[b * 4 + 3];
    bool bad = !(n1 == n1);
    for (int p = 2; p <= 4; ++p) bad = bad || !(N[p] == N[p]);
    double lN[5];
    const double lq = log((double)q);
    for (int p = 1; p <= 4; ++p) lN[p] = (N[p] > 0.0 ? log(N[p]) : -700.0) - p * lq;  // log(inf) = inf: unused
    double b0 = 0.0, b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;  // log best[k-1 .. k-4] (best[0] = 1, none before it)
    double logH = 0.0, lfact = 0.0;
    // per-step growth of log best is at most max_p lN[p]/p: once log k exceeds it the terms only shrink
    const double rate = fmax(fmax(lN[1], 0.5 * lN[2]), fmax(lN[3] * (1.0 / 3.0), 0.25 * lN[4]));
    int kend = 0;
    for (int k = 1; k <= 160; ++k) {
        const double v = fmin(fmin(b0 + lN[1], b1 + lN[2]), fmin(b2 + lN[3], b3 + lN[4]));
        b3 = b2; b2 = b1; b1 = b0; b0 = v;
        lfact += logk[k];
        const double lt = v - lfact;
        logH = fmax(logH, lt);
        if (lt >= -43.75) kend = k;  // 1e-19
        else if (logk[k] > rate) break;
    }
    if (bad) { logH = __longlong_as_double(0x7ff8000000000000ll); kend = 160; }
    // lanes with equal q (equal low two bits) reduce first: eight atomics per wavefront instead of 128 on eight addresses
    unsigned long long hb = dbits(logH), kb = (unsigned long long)kend;
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
        const unsigned long long h2 = __shfl_xor(hb, o, 64), k2 = __shfl_xor(kb, o, 64);
        hb = h2 > hb ? h2 : hb;
        kb = k2 > kb ? k2 : kb;
    }
    if ((threadIdx.x & 63) < 4) {
        atomicMax(&out[q - 1], hb);
        atomicMax(&out[4 + q - 1], kb);
    }
}
void launch_hump(hipStream_t st, const KProb& P, const KBil& B, const double* dZ, const double* g1, int64_t int0, int nb,
                 const double* norms, unsigned long long* out) {
    hipLaunchKernelGGL(k_hump, dim3((4 * nb + 255) / 256), dim3(256), 0, st, P, B, dZ, g1, int0, nb, norms, out);
}
void launch_norm1(hipStream_t st, int npad, int nb, const ChainWork& w) {
    hipLaunchKernelGGL(k_norm1, dim3(nb, 4), dim3(256), 0, st, npad, w, -1);
}
void launch_norm1_one(hipStream_t st, int npad, int nb, const ChainWork& w, int which) {
    hipLaunchKernelGGL(k_norm1, dim3(nb, 1), dim3(256), 0, st, npad, w, which);
}

// Scaling parameter s_k for both evaluation forms of the polynomial (sigma = 2^-s_k).
// alpha_p(A) = max(||A^p||^(1/p), ||A^(p+1)||^(1/(p+1))) bounds the truncation series for
// p(p-1) <= m+1 (Al-Mohy & Higham 2009, Thm 4.2); s_k = max(0, ceil(log2(alpha / theta))).  Whatever runs last for an
// interval -- its s_k-th squaring, or the final polynomial product when s_k = 0 -- stores -exp(A_k) into the Jacobian.
__global__ void k_expm_params(int nb, int s_cap, ChainWork w) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const double n1 = w.norms[b * 4 + 0];
    const double d2v = sqrt(w.norms[b * 4 + 1]);
    const double d3v = cbrt(w.norms[b * 4 + 2]);
    const double d4v = sqrt(sqrt(w.norms[b * 4 + 3]));
    double alpha = fmin(n1, fmin(fmax(d2v, d3v), fmax(d3v, d4v)));
    int s = 0;  // no squaring at all inside the radius: the last polynomial product then stores into the Jacobian itself
    if (alpha > THETA_16) {
        s = (int)ceil(log2(alpha / THETA_16));
        if (s < 1) s = 1;
    }
    if (s > s_cap) s = s_cap;          // huge norm: bounded work
    if (!(alpha == alpha)) s = 1;      // NaN input: one squaring, the NaN propagates to the output
    w.s[b] = s;
    atomicMax(&w.smax[0], s);
    atomicAdd(&w.smax[1], s);
    int s3 = 0;
    if (alpha > THETA_3P) {
        s3 = (int)ceil(log2(alpha / THETA_3P));
        if (s3 < 1) s3 = 1;
    }
    if (s3 > s_cap) s3 = s_cap;
    if (!(alpha == alpha)) s3 = 1;
    w.s3[b] = s3;
    atomicMax(&w.smax[4], s3);
    atomicAdd(&w.smax[5], s3);
    // growth rate handed to the sweep planner: ||A^t|| <= d2^t (t even) and <= max(d3,d4)^t (t >= 6)
    double beta = fmin(d2v, fmax(d3v, d4v));
    if (!(d2v == d2v) || !(d3v == d3v) || !(d4v == d4v)) beta = __longlong_as_double(0x7ff8000000000000ll);
    atomicMax(w.d2max, dbits(beta));
}
// Coefficient table of the chosen evaluation form (dto_kernels.h, EXPM2_* / EXPM3_*): every coefficient carries its power
// of sigma = 2^-s so that the kernels work on the unscaled A, A^2, A^3, A^4.
__global__ void k_expm_coef(int nb, ChainWork w, int form) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    // form 0: decided here, by every thread alike, from the chunk's squaring counts (k_expm_params): three products + s3
    // squarings against two products + s squarings, summed over the chunk; the third product (two outputs, five epilogue
    // streams: HBM-bound) costs about 1.5 squarings
    if (form == 0) form = 2 * ((int64_t)w.smax[1] - (int64_t)w.smax[5]) > 3 * (int64_t)nb ? 3 : 2;
    if (b == 0) w.smax[6] = form;
    if (b >= nb) return;
    int s = w.s[b];
    if (form == 3) { s = w.s3[b]; w.s[b] = s; }
    const double sigma = ldexp(1.0, -s);
    double* cf = w.coef + (int64_t)b * COEF_STRIDE;
    double sp = 1.0;
    const double s4 = (sigma * sigma) * (sigma * sigma);
    for (int i = 0; i <= 4; ++i) {
        if (form == 3) {
            cf[COEF_PC + i] = EXPM3_E[i] * sp;
            cf[COEF_PA + i] = EXPM3_A[i] * sp;
            cf[COEF_PB + i] = EXPM3_B[i] * sp;
            cf[COEF_K + i] = EXPM3_K[i] * sp * s4;
            cf[COEF_L + i] = (EXPM3_C[i] - EXPM3_AL * EXPM3_A[i]) * sp;
            cf[COEF_R + i] = (EXPM3_D[i] - EXPM3_BE * EXPM3_A[i]) * sp;
        } else {
            cf[COEF_PC + i] = EXPM2_C[i] * sp;
            cf[COEF_PA + i] = EXPM2_A[i] * sp;
            cf[COEF_PB + i] = EXPM2_B[i] * sp;
            cf[COEF_K + i] = EXPM2_K[i] * sp * s4;   // Y = (sigma A)^4 K(sigma A), with A^4 as the left operand
        }
        sp *= sigma;
    }
    cf[COEF_L + 5] = EXPM3_AL;
    cf[COEF_R + 5] = EXPM3_BE;
}
void launch_expm_coef(hipStream_t st, int nb, const ChainWork& w, int form) {
    hipLaunchKernelGGL(k_expm_coef, dim3((nb + 63) / 64), dim3(64), 0, st, nb, w, form);
}
void launch_expm_params(hipStream_t st, int nb, int s_cap, const ChainWork& w) {
    hipLaunchKernelGGL(k_expm_params, dim3((nb + 63) / 64), dim3(64), 0, st, nb, s_cap, w);
}

// K = k0 I + k1 A + k2 A2 + k3 A3 + k4 A4  (right factor of Y = A^4 K, scaled coefficients) -> W[5]
__global__ void __launch_bounds__(256) k_poly_h3(int npad, ChainWork w) {
    const int b = blockIdx.y;
    const int64_t nn = (int64_t)npad * npad;
    const double* cf = w.coef + (int64_t)b * COEF_STRIDE + COEF_K;
    const double c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3], c4 = cf[4];
    const double* A1 = w.W[0] + b * nn;
    const double* A2 = w.W[1] + b * nn;
    const double* A3 = w.W[2] + b * nn;
    const double* A4 = w.W[3] + b * nn;
    double* H = w.W[5] + b * nn;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nn; i += (int64_t)gridDim.x * 256) {
        double v = c1 * A1[i] + c2 * A2[i] + c3 * A3[i] + c4 * A4[i];
        if (i % npad == i / npad) v += c0;
        H[i] = v;
    }
}
void launch_poly_h3(hipStream_t st, int npad, int nb, const ChainWork& w) {
    int gx = (int)(((int64_t)npad * npad + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_poly_h3, dim3(gx, nb), dim3(256), 0, st, npad, w);
}

// ============================================================================================
// generator sweep
// ============================================================================================

// Column k of every type belongs to owned interval k (knot kn_lo + k).
// src_kind 0: p_0 = x_k (forward: exp(A) x);  1: p_0 = mu_k (adjoint: exp(A') mu).
__global__ void __launch_bounds__(256) k_sweep_init(KProb P, KBil B, SweepBuf w, int T, const double* __restrict__ Z,
                                                     const double* __restrict__ mu, int src_kind, double inv_q) {
    const int k = blockIdx.x;  // column
    const bool live = k < P.n_int;
    const int64_t kn = P.kn_lo + k;
    const int64_t colsz = w.npad;
    double mx = 0.0;
    for (int r = threadIdx.x; r < w.npad; r += blockDim.x) {
        double v = 0.0;
        if (live && r < B.n) v = src_kind == 0 ? Z[kn * P.z + B.x_off + r] : mu[B.row_off + kn * B.n + r];
        mx = fmax(mx, fabs(v));
        for (int t = 0; t < T; ++t) {
            const int64_t off = ((int64_t)t * w.Kpad + k) * colsz + r;
            const double tv = t == 0 ? v : 0.0;
            w.Z[0][off] = tv;
            w.S[off] = tv;
        }
    }
    __shared__ double sm[4];
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
        for (int t = 0; t < T; ++t) {
            const unsigned long long v = t == 0 ? dbits(mx) : 0ull;
            w.termnorm[(0 * T + t) * (int64_t)w.Kpad + k] = v;
            w.termnorm[(1 * T + t) * (int64_t)w.Kpad + k] = 0ull;
            w.termnorm[(2 * T + t) * (int64_t)w.Kpad + k] = 0ull;
            w.sumnorm[(int64_t)t * w.Kpad + k] = v;
        }
        const double dt = live ? Z[kn * P.z + P.dt_idx] : 0.0;
        w.scaleE[k] = dt * inv_q;
        w.scaleE[w.Kpad + k] = 2.0 * dt * inv_q;
        for (int j = 0; j <= B.m; ++j) {
            const double ub = live ? (j == 0 ? 1.0 : Z[kn * P.z + B.u_off + j - 1]) : 0.0;
            w.scaleU[(int64_t)j * w.Kpad + k] = ub;
            w.scaleA[(int64_t)j * w.Kpad + k] = dt * ub * inv_q;
        }
        if (k % w.TN == 0) {
            w.active[k / w.TN] = 1;
            if (w.nterms) w.nterms[k / w.TN] = 0;
        }
        if (k == 0) { w.stats[0] = w.Kpad / w.TN; w.stats[1] = 0; }
    }
}
void launch_sweep_init(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, const SweepTypes& ty,
                       const double* dZ, const double* dmu, int src_kind, int q) {
    hipLaunchKernelGGL(k_sweep_init, dim3(w.Kpad), dim3(256), 0, st, P, B, w, ty.T, dZ, dmu, src_kind, 1.0 / q);
}

__global__ void __launch_bounds__(256) k_sweep_set_type(KProb P, KBil B, SweepBuf w, int T, int type, const double* __restrict__ v) {
    const int k = blockIdx.x;
    const bool live = k < P.n_int;
    const int64_t kn = P.kn_lo + k;
    const int64_t base = ((int64_t)type * w.Kpad + k) * w.npad;
    double mx = 0.0;
    for (int r = threadIdx.x; r < w.npad; r += blockDim.x) {
        const double x = (live && r < B.n) ? v[kn * P.z + B.x_off + r] : 0.0;
        w.Z[0][base + r] = x;
        w.S[base + r] = x;
        mx = fmax(mx, fabs(x));
    }
    __shared__ double sm[4];
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
        w.termnorm[(0 * T + type) * (int64_t)w.Kpad + k] = dbits(mx);
        w.sumnorm[(int64_t)type * w.Kpad + k] = dbits(mx);
    }
}
// Start of a sweep that computes types first_type..T-1 only (frozen p terms): clear their term-0 buffers, sums and norms;
// type 0 keeps its sums (exp(A)x of the earlier callback) and counts as converged from the start.
__global__ void __launch_bounds__(256) k_sweep_init_tangents(SweepBuf w, int T) {
    const int k = blockIdx.x;
    const int64_t colsz = w.npad;
    for (int t = w.first_type; t < T; ++t)
        for (int r = threadIdx.x; r < w.npad; r += blockDim.x) {
            const int64_t off = ((int64_t)t * w.Kpad + k) * colsz + r;
            w.Z[0][off] = 0.0;
            w.S[off] = 0.0;
        }
    if (threadIdx.x == 0) {
        for (int t = 0; t < T; ++t) {
            for (int sl = 0; sl < 3; ++sl) w.termnorm[((int64_t)sl * T + t) * w.Kpad + k] = 0ull;
            if (t >= w.first_type) w.sumnorm[(int64_t)t * w.Kpad + k] = 0ull;
        }
        if (k % w.TN == 0) {
            w.active[k / w.TN] = 1;
            if (w.nterms) w.nterms[k / w.TN] = 0;
        }
        if (k == 0) { w.stats[0] = w.Kpad / w.TN; w.stats[1] = 0; }
    }
}
void launch_sweep_init_tangents(hipStream_t st, const SweepBuf& w, int T) {
    hipLaunchKernelGGL(k_sweep_init_tangents, dim3(w.Kpad), dim3(256), 0, st, w, T);
}

void launch_sweep_set_type(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, int T, int type, const double* v) {
    hipLaunchKernelGGL(k_sweep_set_type, dim3(w.Kpad), dim3(256), 0, st, P, B, w, T, type, v);
}

// Next sub-interval of a q-fold split exp(A) = exp(A/q)^q: the sums become term 0 of the new series.
__global__ void __launch_bounds__(256) k_sweep_restart(SweepBuf w, int T) {
    const int k = blockIdx.x, t = blockIdx.y;
    const int64_t base = ((int64_t)t * w.Kpad + k) * w.npad;
    double mx = 0.0;
    for (int r = threadIdx.x; r < w.npad; r += blockDim.x) {
        const double v = w.S[base + r];
        w.Z[0][base + r] = v;
        mx = fmax(mx, fabs(v));
    }
    __shared__ double sm[4];
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
        w.termnorm[(0 * T + t) * (int64_t)w.Kpad + k] = dbits(mx);
        w.termnorm[(1 * T + t) * (int64_t)w.Kpad + k] = 0ull;
        w.termnorm[(2 * T + t) * (int64_t)w.Kpad + k] = 0ull;
        w.sumnorm[(int64_t)t * w.Kpad + k] = dbits(mx);
        if (t == 0 && k % w.TN == 0) w.active[k / w.TN] = 1;
        if (t == 0 && k == 0) w.stats[0] = w.Kpad / w.TN;
    }
}
void launch_sweep_restart(hipStream_t st, const SweepBuf& w, int T) {
    hipLaunchKernelGGL(k_sweep_restart, dim3(w.Kpad, T), dim3(256), 0, st, w, T);
}

struct SweepArgs {
    KBil B;
    SweepBuf w;
    SweepTypes ty;
    const double* G;     // generators used as the left operand (G or G')
    const double* Zin;
    double* Zout;
    int t;               // producing term t+1 from term t
    int mode;            // 0: Taylor step; 1: out_j = G_j V (no sum); 2: out = sum_j ubar_j G_j V;
                         // 3: split-K Taylor step of a single column type: out_j = G_j (V .* dt ubar_j/q), active blocks only
    const double* V;
    double* out;
    int64_t seg_cols, seg_stride;  // mode 1: V's columns come in segments of seg_cols columns, seg_stride doubles apart (0: contiguous)
};

// Epilogue shared by the sweep kernels: store the new term, accumulate the sums and the column norms (Taylor
// step), or just store the product (modes 1, 2).
template <int TM, int TN>
__device__ __forceinline__ void sweep_epilogue(const SweepArgs& a, GemmAcc<TM, TN>& acc, int rt, int ct, int ty) {
    using Cfg = GemmCfg<TM, TN>;
    const int npad = a.w.npad, Kpad = a.w.Kpad;
    const int64_t typesz = (int64_t)Kpad * npad;
    GemmCoord<TM, TN> co;
    const int row0 = rt * TM + co.row_base, col0 = ct * TN + co.col_base;
    if (a.mode != 0) {
        double* O = a.out + (a.mode == 1 || a.mode == 3 ? ty * typesz : 0);
#pragma unroll
        for (int tj = 0; tj < Cfg::NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ti = 0; ti < Cfg::MT; ++ti)
                    O[(int64_t)(col0 + 16 * tj + 4 * r) * npad + row0 + 16 * ti] = acc.v[ti][tj][r];
        return;
    }
    const double inv = 1.0 / (double)(a.t + 1);
    double* Zo = a.Zout + ty * typesz;
    double* So = a.w.S + ty * typesz;
    unsigned long long* tn = a.w.termnorm + ((int64_t)((a.t + 1) % 3) * a.ty.T + ty) * Kpad;
    unsigned long long* sn = a.w.sumnorm + (int64_t)ty * Kpad;
#pragma unroll
    for (int tj = 0; tj < Cfg::NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = col0 + 16 * tj + 4 * r;
            double tmax = 0.0, smax = 0.0;
#pragma unroll
            for (int ti = 0; ti < Cfg::MT; ++ti) {
                const int64_t off = (int64_t)col * npad + row0 + 16 * ti;
                const double v = acc.v[ti][tj][r] * inv;
                const double s = So[off] + v;
                Zo[off] = v;
                So[off] = s;
                tmax = fmax(tmax, fabs(v));
                smax = fmax(smax, fabs(s));
            }
            // reduce over the 16 lanes that share this column (lane & 15 = row index)
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                tmax = fmax(tmax, __shfl_xor(tmax, o, 64));
                smax = fmax(smax, __shfl_xor(smax, o, 64));
            }
            if ((threadIdx.x & 15) == 0) {
                atomicMax(&tn[col], dbits(tmax));
                atomicMax(&sn[col], dbits(smax));
            }
        }
}

// One Taylor step of the sweep for every owned interval at once:
//   term_{t+1}[type] = 1/(t+1) * ( sum_j G_j * (term_t[type] .* dt ubar_j/q)  +  sum_extra G_g * (term_t[src] .* dt/q * mult) )
// i.e. a GEMM whose K dimension is the concatenation of the generator blocks, the per-interval
// bilinear coefficients being applied to the B panel while it is staged into LDS; an extra term shares the
// segment of its generator (both right-hand sides are combined in the staging registers).
// FROZEN: the p column is not computed -- its Taylor terms were stored by an earlier callback at the same point
// (a.w.frozen); the workgroups cover types first_type..T-1 and take the inhomogeneous term from the store.
template <int TM, int TN, bool FROZEN = false>
__global__ void __launch_bounds__(256, (TM * TN <= 64 * 64 ? 4 : 2)) k_sweep(SweepArgs a) {
    using Cfg = GemmCfg<TM, TN>;
    __shared__ __attribute__((aligned(16))) double smem[Cfg::SMEM_DOUBLES];
    const int npad = a.w.npad, Kpad = a.w.Kpad;
    const int row_tiles = npad / TM;
    const int rt = blockIdx.x % row_tiles;
    const int ct = blockIdx.x / row_tiles;
    const int ty = blockIdx.y + (FROZEN ? a.w.first_type : 0);  // column type (mode 0) or generator index (mode 1)
    if ((a.mode == 0 || a.mode == 3) && !a.w.active[(ct * TN) / a.w.TN]) return;
    const int64_t nn = (int64_t)npad * npad;
    const int64_t typesz = (int64_t)Kpad * npad;
    const int m = a.B.m;

    GemmAcc<TM, TN> acc;
    acc.zero();
    if (a.mode == 0) {
        const double* Bt = a.Zin + ty * typesz + (int64_t)ct * TN * npad;
        const TypeDesc td = a.ty.t[ty];
        // an extra term G_g (dt term_t[src]) rides in generator g's own segment: G_g (c_g term_t[type] + dt term_t[src]) --
        // the two operands are added while the panel is staged, the product is paid once (the extras of one type name
        // distinct generators: make_types)
        for (int j = 0; j <= m; ++j) {
            int e = -1;
            for (int x = 0; x < td.n_extra; ++x)
                if (td.gen[x] == j) e = x;
            if (FROZEN && e >= 0 && td.src[e] == 0) {
                // stored p term t; beyond the block's stored terms the p series had converged: nothing to add
                const int cnt = a.w.nterms_p[(ct * TN) / a.w.TN];
                if (a.t >= (cnt > 0 ? cnt : a.w.frozen_total)) e = -1;
            }
            if (e >= 0) {
                const double* Bs = (FROZEN && td.src[e] == 0 ? a.w.frozen + (int64_t)a.t * typesz : a.Zin + td.src[e] * typesz) +
                                   (int64_t)ct * TN * npad;
                // scaleE[0] carries dt/q, scaleE[1] carries 2 dt/q (the i == j second-order terms)
                gemm_accumulate2<TM, TN>(acc, a.G + j * nn + (int64_t)rt * TM, npad, Bt, npad, npad,
                                         a.w.scaleA + (int64_t)j * Kpad + ct * TN, Bs,
                                         a.w.scaleE + (td.mult[e] == 2.0 ? Kpad : 0) + ct * TN, smem);
            } else {
                gemm_accumulate<TM, TN>(acc, a.G + j * nn + (int64_t)rt * TM, npad, Bt, npad, npad,
                                        a.w.scaleA + (int64_t)j * Kpad + ct * TN, smem);
            }
        }
    } else if (a.mode == 3) {
        gemm_accumulate<TM, TN>(acc, a.G + ty * nn + (int64_t)rt * TM, npad, a.V + (int64_t)ct * TN * npad, npad, npad,
                                a.w.scaleA + (int64_t)ty * Kpad + ct * TN, smem);
    } else if (a.mode == 1) {
        const int64_t c0 = (int64_t)ct * TN;  // a tile never straddles two segments (seg_cols is a multiple of TN)
        const double* Vt = a.seg_cols ? a.V + (c0 / a.seg_cols) * a.seg_stride + (c0 % a.seg_cols) * npad : a.V + c0 * npad;
        gemm_accumulate<TM, TN>(acc, a.G + ty * nn + (int64_t)rt * TM, npad, Vt, npad, npad, nullptr, smem);
    } else {
        for (int j = 0; j <= m; ++j)
            gemm_accumulate<TM, TN>(acc, a.G + j * nn + (int64_t)rt * TM, npad, a.V + (int64_t)ct * TN * npad, npad, npad,
                                    a.w.scaleU + (int64_t)j * Kpad + ct * TN, smem);
    }

    sweep_epilogue<TM, TN>(a, acc, rt, ct, ty);
}

static int sweep_tile_choice() {
    static int v = tune_int("DTO_SWEEP_TILE", -1);
    return v;
}
static void launch_sweep_kernel(hipStream_t st, const SweepArgs& a, int ny) {
    const int npad = a.w.npad;
    // tile choice: enough workgroups to cover 256 CUs a few times over (the per-step GEMM is small)
    int choice = sweep_tile_choice();
    if (choice < 0) {
        // 64x32 measured fastest at 256x2000 (1280 workgroups = 5 per CU: balanced); smaller problems get 32x32 tiles
        // so that the launch still covers the chip (-5..10 % per step at 64 and 128 states)
        const long wgs = (long)(npad / 64) * (a.w.Kpad / 32) * ny;
        choice = wgs >= 1024 ? 5 : 6;
    }
    static const bool relax = tune_int("DTO_SWEEP_TILE_RELAX", 1) != 0;
    if (!relax && (npad % 128 != 0 || a.w.TN != 128)) choice = 0;
    if ((choice == 3 || choice == 1) && npad % 128 != 0) choice = 0;  // 128-row tiles need npad % 128 == 0
    if ((choice == 3 || choice == 2) && a.w.Kpad % 128 != 0) choice = 0;
    if (a.mode == 0 && a.w.frozen) {
        if (choice != 6) choice = 5;
        if (choice == 6)
            hipLaunchKernelGGL((k_sweep<32, 32, true>), dim3((npad / 32) * (a.w.Kpad / 32), ny), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((k_sweep<64, 32, true>), dim3((npad / 64) * (a.w.Kpad / 32), ny), dim3(256), 0, st, a);
        return;
    }
    switch (choice) {
        case 3:
            hipLaunchKernelGGL((k_sweep<128, 128>), dim3((npad / 128) * (a.w.Kpad / 128), ny), dim3(256), 0, st, a);
            break;
        case 2:
            hipLaunchKernelGGL((k_sweep<64, 128>), dim3((npad / 64) * (a.w.Kpad / 128), ny), dim3(256), 0, st, a);
            break;
        case 1:
            hipLaunchKernelGGL((k_sweep<128, 64>), dim3((npad / 128) * (a.w.Kpad / 64), ny), dim3(256), 0, st, a);
            break;
        case 4:
            hipLaunchKernelGGL((k_sweep<32, 64>), dim3((npad / 32) * (a.w.Kpad / 64), ny), dim3(256), 0, st, a);
            break;
        case 6:
            hipLaunchKernelGGL((k_sweep<32, 32>), dim3((npad / 32) * (a.w.Kpad / 32), ny), dim3(256), 0, st, a);
            break;
        case 5:
            // (a DMA-staged variant of this tile with per-segment accumulators was measured 12 % slower)
            hipLaunchKernelGGL((k_sweep<64, 32>), dim3((npad / 64) * (a.w.Kpad / 32), ny), dim3(256), 0, st, a);
            break;
        default:
            hipLaunchKernelGGL((k_sweep<64, 64>), dim3((npad / 64) * (a.w.Kpad / 64), ny), dim3(256), 0, st, a);
    }
}

// Second half of a split-K step: term_{t+1} = 1/(t+1) * sum_g partial_g in fixed order (deterministic), sums and
// column norms exactly as sweep_epilogue does them.  One workgroup per interval column.
__global__ void __launch_bounds__(256) k_sweep_finalize(SweepBuf w, double* __restrict__ Zout, const double* __restrict__ part,
                                                         int m1, int t) {
    const int col = blockIdx.x;
    if (!w.active[col / w.TN]) return;
    __shared__ double sm[8];
    const int64_t typesz = (int64_t)w.Kpad * w.npad;
    const double inv = 1.0 / (double)(t + 1);
    double tmax = 0.0, smax = 0.0;
    for (int r = threadIdx.x; r < w.npad; r += 256) {
        const int64_t off = (int64_t)col * w.npad + r;
        double v = 0.0;
        for (int g = 0; g < m1; ++g) v += part[g * typesz + off];
        v *= inv;
        const double s = w.S[off] + v;
        Zout[off] = v;
        w.S[off] = s;
        tmax = fmax(tmax, fabs(v));
        smax = fmax(smax, fabs(s));
    }
    tmax = wave_max(tmax);
    smax = wave_max(smax);
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = tmax; sm[4 + (threadIdx.x >> 6)] = smax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tmax = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
        smax = fmax(fmax(sm[4], sm[5]), fmax(sm[6], sm[7]));
        w.termnorm[(int64_t)((t + 1) % 3) * w.Kpad + col] = dbits(tmax);  // T = 1
        const double old = bits_to_d(w.sumnorm[col]);
        w.sumnorm[col] = dbits(fmax(old, smax));
    }
}

void launch_sweep_step(hipStream_t st, const KBil& B, const SweepBuf& w, const SweepTypes& ty, int transposed, int t,
                       int in_buf, int split_store) {
    SweepArgs a{};
    a.B = B; a.w = w; a.ty = ty; a.G = transposed ? B.GT : B.G;
    a.Zin = w.Z[in_buf]; a.Zout = w.Z[in_buf ^ 1]; a.t = t; a.mode = 0;
    // a single column type (eval_constraint) offers few tiles with a long K loop each: split K by generator into the
    // unused type slots of the output buffer, then sum the partials (DTO_SWEEP_SPLITK=0 disables)
    static const bool splitk = tune_int("DTO_SWEEP_SPLITK", 1) != 0;
    // (store mode with one type: the slots of the NEXT m+1 terms serve as scratch -- the caller vouches for the room)
    if (splitk && ty.T == 1 && B.m >= 1 && (split_store || (w.T_alloc >= B.m + 2 && w.Z[0] != w.Zt && a.Zin != w.Zt))) {
        const int64_t typesz = (int64_t)w.Kpad * w.npad;
        a.mode = 3; a.V = a.Zin; a.out = a.Zout + typesz;
        launch_sweep_kernel(st, a, B.m + 1);
        hipLaunchKernelGGL(k_sweep_finalize, dim3(w.Kpad), dim3(256), 0, st, w, a.Zout, a.Zout + typesz, B.m + 1, t);
        return;
    }
    // (a variant that keeps one accumulator tile per column TYPE in each wave -- generator panels reused by all types,
    // term chunks by all generators, coefficients applied in registers: ~6x less L2->LDS traffic -- was measured 7 %
    // slower at 256x2000: with 512 workgroups of 80 short barrier-separated steps it has less slack than this one)
    launch_sweep_kernel(st, a, ty.T - (w.frozen ? w.first_type : 0));
}
void launch_apply_generators(hipStream_t st, const KBil& B, const SweepBuf& w, int transposed, const double* V,
                             double* out) {
    SweepArgs a{};
    a.B = B; a.w = w; a.G = transposed ? B.GT : B.G; a.mode = 1; a.V = V; a.out = out;
    launch_sweep_kernel(st, a, B.m + 1);
}
void launch_apply_generators_cols(hipStream_t st, const KBil& B, const SweepBuf& w, int transposed, const double* V,
                                  double* out, int gen_first, int gen_count, int64_t cols, int64_t seg_cols, int64_t seg_stride) {
    SweepArgs a{};
    a.B = B; a.w = w;
    a.w.Kpad = (int32_t)cols;  // the kernel only uses Kpad as the column count / generator stride in this mode
    a.G = (transposed ? B.GT : B.G) + (int64_t)gen_first * B.npad * B.npad;
    a.mode = 1; a.V = V; a.out = out;
    a.seg_cols = seg_cols; a.seg_stride = seg_stride;
    launch_sweep_kernel(st, a, gen_count);
}

// U[a][type][k][r] = sum_{b < na} Btab[a][b] * terms[b][type][k][r]  for a < nf (valid term counts per column block)
__global__ void __launch_bounds__(256) k_pair_combine(SweepBuf ad, int T, int nf_used, int na_used,
                                                     const int32_t* __restrict__ nterms_f, int nblk_f,
                                                     const double* __restrict__ Btab, double* __restrict__ U) {
    const int64_t typesz = (int64_t)ad.Kpad * ad.npad;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (k, r)
    if (e >= typesz) return;
    const int type = blockIdx.y;
    const int col = (int)(e / ad.npad);
    int nf = nterms_f[col / nblk_f], na = ad.nterms[col / ad.nblk];
    if (nf <= 0 || nf > nf_used) nf = nf_used;
    if (na <= 0 || na > na_used) na = na_used;
    const int64_t tstride = (int64_t)T * typesz;
    const double* src = ad.Zt + type * typesz + e;
    double* dst = U + type * typesz + e;
    for (int a0 = 0; a0 < nf; a0 += 8) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < na; ++b) {
            const double v = src[b * tstride];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += Btab[(a0 + q) * PAIR_DCAP + b] * v;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (a0 + q < nf) dst[(a0 + q) * tstride] = acc[q];
    }
}
void launch_pair_combine(hipStream_t st, const SweepBuf& ad, int T, int n_types, int nf_used, int na_used,
                         const int32_t* nterms_f, int nblk_f, const double* Btab, double* U) {
    const int64_t typesz = (int64_t)ad.Kpad * ad.npad;
    hipLaunchKernelGGL(k_pair_combine, dim3((unsigned)((typesz + 255) / 256), n_types), dim3(256), 0, st, ad, T, nf_used, na_used,
                       nterms_f, nblk_f, Btab, U);
}

// (u_i, u_j) block of mu_k' f by the pairing formula.  With p_a, d^i_a the Taylor terms of ONE sweep with tangents
// (exp(A)x and its u_i-tangent) and pt_b the plain terms of the OTHER sweep (exp(A')mu),
//   d2/du_i du_j [mu' exp(A) x] = sum_{a,b} B(a,b) ( pt_b' E_i d^j_a + pt_b' E_j d^i_a ),  B(a,b) = a! b! / (a+b+1)!
// (split the word mu' ... E ... E ... x at its LEFT generator: what stands to its right is a first-order forward
// term, what stands to its left a plain adjoint term, and the Taylor terms of exp(tau A) and exp((1-tau)A')
// integrate against each other over tau in [0,1]).  With U_a = sum_b B(a,b) pt_b (k_pair_combine, type 0 only) and
// EP[j][a] = G_j' U_a (one generator product per drive and term), the block is  sum_a EP[i][a].d^j_a + EP[j][a].d^i_a.
// The scalar mu' exp(A) x = x' exp(A') mu is symmetric under (A, x, mu, G) <-> (A', mu, x, G'), and the engine uses the
// transposed reading: the tangents come from the ADJOINT sweep (which the (x,u) block needs anyway), the plain terms from a
// forward sweep of the p column alone, EP[j][a] = G_j U_a.  `fw` below is the sweep that carries the tangents.
template <int M>
__global__ void __launch_bounds__(256) k_hess_pair(KProb P, KBil B, SweepBuf fw, int nf_used,
                                                   const double* __restrict__ EP, double* __restrict__ H) {
    __shared__ double red[4][M * M];
    const int64_t kl = blockIdx.x;
    const int64_t kn = P.kn_lo + kl;
    constexpr int m = M;  // drives: compile-time so that the accumulators live in registers
    const int npad = fw.npad, T = 1 + m;
    const int64_t typesz = (int64_t)fw.Kpad * npad;
    const int64_t tstride = (int64_t)T * typesz;          // one stored Taylor term of all forward types
    const int64_t gstride = (int64_t)nf_used * typesz;    // one generator in EP (nf_used terms)
    int nf = fw.nterms[kl / fw.nblk];
    if (nf <= 0 || nf > nf_used) nf = nf_used;
    double acc[M][M];
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) acc[i][j] = 0.0;
    // 16-byte loads: a thread owns two adjacent rows (the padding rows hold zeros), the two halves of the workgroup take the
    // even and the odd terms
    const int half = threadIdx.x >> 7;
    for (int r = 2 * (threadIdx.x & 127); r < npad; r += 256) {
        const int64_t off = kl * npad + r;
        for (int a = half; a < nf; a += 2) {
            const double* Da = fw.Zt + a * tstride + off;
            const double* Ea = EP + a * typesz + off;
            d2 Et[M], D[M];
            for (int j = 0; j < m; ++j) {
                Et[j] = *reinterpret_cast<const d2*>(Ea + j * gstride);
                D[j] = *reinterpret_cast<const d2*>(Da + (1 + j) * typesz);
            }
            for (int i = 0; i < m; ++i)
                for (int j = i; j < m; ++j)
                    acc[i][j] += Et[i].x * D[j].x + Et[j].x * D[i].x + Et[i].y * D[j].y + Et[j].y * D[i].y;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = 0; i < m; ++i)
        for (int j = i; j < m; ++j) {
            const double v = wave_sum(acc[i][j]);
            if (lane == 0) red[wave][i * M + j] = v;
        }
    __syncthreads();
    if (threadIdx.x < m * m) {
        const int i = threadIdx.x / m, j = threadIdx.x % m;
        if (i <= j) {
            const int q = i * M + j;
            const double dt = fw.scaleE[kl];  // dt/q with q = 1 on this path
            hess_add(P, H, kn, B.u_off + i, B.u_off + j, -dt * (red[0][q] + red[1][q] + red[2][q] + red[3][q]));
        }
    }
}
void launch_hess_pair(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& fw, int nf_used, const double* EP,
                      double* H) {
    if (P.n_int <= 0) return;
    const dim3 grid((unsigned)P.n_int), block(256);
    switch (B.m) {
        case 1: hipLaunchKernelGGL(k_hess_pair<1>, grid, block, 0, st, P, B, fw, nf_used, EP, H); break;
        case 2: hipLaunchKernelGGL(k_hess_pair<2>, grid, block, 0, st, P, B, fw, nf_used, EP, H); break;
        case 3: hipLaunchKernelGGL(k_hess_pair<3>, grid, block, 0, st, P, B, fw, nf_used, EP, H); break;
        case 4: hipLaunchKernelGGL(k_hess_pair<4>, grid, block, 0, st, P, B, fw, nf_used, EP, H); break;
        case 5: hipLaunchKernelGGL(k_hess_pair<5>, grid, block, 0, st, P, B, fw, nf_used, EP, H); break;
        case 6: hipLaunchKernelGGL(k_hess_pair<6>, grid, block, 0, st, P, B, fw, nf_used, EP, H); break;
        default: hipLaunchKernelGGL(k_hess_pair<MAX_DRIVES>, grid, block, 0, st, P, B, fw, nf_used, EP, H); break;
    }
}

void launch_apply_Gu(hipStream_t st, const KBil& B, const SweepBuf& w, int transposed, const double* V, double* out) {
    SweepArgs a{};
    a.B = B; a.w = w; a.G = transposed ? B.GT : B.G; a.mode = 2; a.V = V; a.out = out;
    launch_sweep_kernel(st, a, 1);
}

// (Running this test inside the step kernel -- last workgroup per column block, found by an atomic counter -- was
// measured 2x SLOWER per step: the agent-scope release fence each workgroup needs before bumping the counter writes the
// XCD's L2 back (the eight L2s are not coherent with each other), which a kernel boundary does once for everybody.)
// Termination test of the Taylor recurrences (Al-Mohy & Higham 2011, Alg. 3.2 line 13 form):
// a block of TN intervals stops when, for every column of every type, two successive terms are
// below tol * |sum|.  Also recycles the term-norm slot the next step will write.
__global__ void __launch_bounds__(64) k_sweep_check(SweepBuf w, int T, int t, double tol) {
    const int ct = blockIdx.x;
    if (!w.active[ct]) return;
    const int lane = threadIdx.x;
    bool conv = true;
    const int64_t Kp = w.Kpad;
    for (int c = lane; c < w.TN; c += 64) {
        const int col = ct * w.TN + c;
        for (int ty = 0; ty < T; ++ty) {
            const double a0 = bits_to_d(w.termnorm[((int64_t)(t % 3) * T + ty) * Kp + col]);
            const double a1 = bits_to_d(w.termnorm[((int64_t)((t + 1) % 3) * T + ty) * Kp + col]);
            const double sn = bits_to_d(w.sumnorm[(int64_t)ty * Kp + col]);
            // non-finite columns cannot improve with more terms: let NaN/Inf through (the solver handles them)
            if (!(a0 + a1 <= tol * sn) && (a0 + a1 == a0 + a1) && sn < 1e300) conv = false;
            w.termnorm[((int64_t)((t + 2) % 3) * T + ty) * Kp + col] = 0ull;
        }
    }
    conv = __all(conv);
    if (lane == 0) {
        if (conv) {
            w.active[ct] = 0;
            if (w.nterms) w.nterms[ct] = t + 2;  // terms 0..t+1 are in the store
            atomicSub(&w.stats[0], 1);
        } else {
            atomicMax(&w.stats[1], t + 2);
        }
    }
}
void launch_sweep_check(hipStream_t st, const SweepBuf& w, int T, int t, double tol) {
    hipLaunchKernelGGL(k_sweep_check, dim3(w.Kpad / w.TN), dim3(64), 0, st, w, T, t, tol);
}

// ============================================================================================
// constraint values (evaluate!)
// ============================================================================================

// delta_k = x_{k+1} - exp(dt G(u_k)) x_k   (bilinear_integrator.jl:98-107); exp(A)x is sweep sum S[0].
__global__ void k_cons_bilinear(KProb P, KBil B, SweepBuf w, const double* __restrict__ Z, double* __restrict__ g) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_int * B.n) return;
    const int64_t k = i / B.n;
    const int r = (int)(i % B.n);
    const int64_t kn = P.kn_lo + k;
    g[B.lrow_off + i] = Z[(kn + 1) * P.z + B.x_off + r] - w.S[k * w.npad + r];
}
void launch_cons_bilinear(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, const double* dZ, double* g) {
    const int64_t n = P.n_int * B.n;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_cons_bilinear, dim3((unsigned)((n + 255) / 256)), blk(P, 256), 0, st, P, B, w, dZ, g);
}

// delta_k = x_{k+1} - x_k - dt_k xdot_k   (derivative_integrator.jl:55-64)
__global__ void k_cons_derivative(KProb P, KDer D, const double* __restrict__ Z, double* __restrict__ g) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_int * D.d) return;
    const int64_t k = i / D.d;
    const int r = (int)(i % D.d);
    const int64_t kn = P.kn_lo + k;
    const double* zk = Z + kn * P.z;
    g[D.lrow_off + i] = zk[P.z + D.x_off + r] - zk[D.x_off + r] - zk[P.dt_idx] * zk[D.xdot_off + r];
}
void launch_cons_derivative(hipStream_t st, const KProb& P, const KDer& Dv, const double* dZ, double* g) {
    const int64_t n = P.n_int * Dv.d;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_cons_derivative, dim3((unsigned)((n + 255) / 256)), blk(P, 256), 0, st, P, Dv, dZ, g);
}

__device__ __forceinline__ double knot_norm2(const KProb& P, const KCon& C, const double* zk) {
    double s = 0.0;
    for (int c = 0; c < C.n_comps; ++c) {
        const double v = zk[C.comps[c]];
        s += v * v;
    }
    return s;
}

// values[i] = g(z_t[comps])   (knot_point_constraint.jl:235-247)
__global__ void k_cons_knot(KProb P, KCon C, const double* __restrict__ Z, double* __restrict__ g) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C.n_times) return;
    const double s = knot_norm2(P, C, Z + C.times[i] * P.z);
    g[C.lrow[i]] = (C.kind == 1 ? sqrt(s) : s) - C.c;
}
void launch_cons_knot(hipStream_t st, const KProb& P, const KCon& C, const double* dZ, double* g) {
    if (C.n_times <= 0) return;
    hipLaunchKernelGGL(k_cons_knot, dim3((unsigned)((C.n_times + 255) / 256)), dim3(256), 0, st, P, C, dZ, g);
}

// ============================================================================================
// Jacobian assembly (A1): everything except the -E_k block, which the chain stores itself.
// The value slab is zero-filled first (evaluator.jl:497); structural zeros stay zero.
// ============================================================================================

// Per owned knot kn:  rows of interval kn-1: d(delta_{kn-1})/d(x_kn) = I;  rows of interval kn:
// u_j columns = -dexp(A)[dt G_j] x_k, dt column = -G(u) exp(A) x_k   (bilinear_integrator.jl:111-131)
__global__ void __launch_bounds__(256) k_jac_bilinear(KProb P, KBil B, SweepBuf w, double* __restrict__ vals) {
    const int64_t kl = blockIdx.x;  // local knot
    const int64_t kn = P.kn_lo + kl;
    const int n = B.n;
    if (kn >= 1) {
        for (int r = threadIdx.x; r < n; r += blockDim.x)
            vals[jac_pos(P, P.colptr, kn, B.x_off + r, B.pre, n, 0, r)] = 1.0;
    }
    if (kn < P.K && kl < P.n_int) {
        const int64_t typesz = (int64_t)w.Kpad * w.npad;
        for (int j = 0; j < B.m; ++j) {
            const int64_t base = jac_pos(P, P.colptr, kn, B.u_off + j, B.pre, n, 1, 0);
            const double* c = w.S + (1 + j) * typesz + kl * w.npad;
            for (int r = threadIdx.x; r < n; r += blockDim.x) vals[base + r] = -c[r];
        }
        const int64_t base = jac_pos(P, P.colptr, kn, P.dt_idx, B.pre, n, 1, 0);
        const double* gy = w.GY + kl * w.npad;
        for (int r = threadIdx.x; r < n; r += blockDim.x) vals[base + r] = -gy[r];
    }
}
void launch_jac_bilinear(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, double* vals) {
    if (P.n_knots <= 0) return;
    hipLaunchKernelGGL(k_jac_bilinear, dim3((unsigned)P.n_knots), blk(P, 256), 0, st, P, B, w, vals);
}

// DerivativeIntegrator block: d/dx_k = -I, d/dxdot_k = -dt I, d/ddt = -xdot_k, d/dx_{k+1} = I
// (derivative_integrator.jl:68-86).  Contributions to one entry add, as they do inside the
// reference's single ForwardDiff Jacobian of the block.
__global__ void k_jac_derivative(KProb P, KDer D, const double* __restrict__ Z, double* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_knots * D.d) return;
    const int64_t kl = i / D.d;
    const int r = (int)(i % D.d);
    const int64_t kn = P.kn_lo + kl;
    if (kn >= 1) vals[jac_pos(P, P.colptr, kn, D.x_off + r, D.pre, D.d, 0, r)] = 1.0;
    if (kn < P.K) {
        // row r of the interval's block has three entries: d/dx_r = -1, d/dxdot_r = -dt, d/ddt = -xdot_r.  Components may
        // coincide (x and xdot overlapping, the timestep inside one of them): such contributions add, as they do inside the
        // reference's single ForwardDiff Jacobian of the block -- combined HERE and then ASSIGNED, so that the entry does not
        // depend on what the buffer held before (bound outputs keep their constants; no other thread or kernel writes row r of
        // this integrator)
        const double* zk = Z + kn * P.z;
        const int64_t p1 = jac_pos(P, P.colptr, kn, D.x_off + r, D.pre, D.d, 1, r);
        const int64_t p2 = jac_pos(P, P.colptr, kn, D.xdot_off + r, D.pre, D.d, 1, r);
        const int64_t p3 = jac_pos(P, P.colptr, kn, P.dt_idx, D.pre, D.d, 1, r);
        double v1 = -1.0, v2 = -zk[P.dt_idx], v3 = -zk[D.xdot_off + r];
        bool w2 = true, w3 = true;
        if (p2 == p1) { v1 += v2; w2 = false; }
        if (p3 == p1) { v1 += v3; w3 = false; }
        else if (p3 == p2 && w2) { v2 += v3; w3 = false; }
        vals[p1] = v1;
        if (w2) vals[p2] = v2;
        if (w3) vals[p3] = v3;
    }
}
void launch_jac_derivative(hipStream_t st, const KProb& P, const KDer& Dv, const double* dZ, double* vals) {
    const int64_t n = P.n_knots * Dv.d;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_jac_derivative, dim3((unsigned)((n + 255) / 256)), blk(P, 256), 0, st, P, Dv, dZ, vals);
}

// dg/dv at the listed knots (knot_point_constraint.jl:254-268); entries outside the pattern taken
// at Z0 are dropped exactly as evaluator.jl:545-547 drops them (jpos = -1).
__global__ void k_jac_knot(KProb P, KCon C, const double* __restrict__ Z, double* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C.n_times * C.n_comps) return;
    const int64_t ti = i / C.n_comps;
    const int c = (int)(i % C.n_comps);
    const int64_t p = C.jpos[i];
    if (p < 0) return;
    const double* zk = Z + C.times[ti] * P.z;
    const double v = zk[C.comps[c]];
    vals[p] = C.kind == 1 ? v / sqrt(knot_norm2(P, C, zk)) : 2.0 * v;
}
void launch_jac_knot(hipStream_t st, const KProb& P, const KCon& C, const double* dZ, double* vals) {
    const int64_t n = C.n_times * C.n_comps;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_jac_knot, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, C, dZ, vals);
}

// ============================================================================================
// matrix-free Jacobian-vector products (A3 without the value slab)
// ============================================================================================
__global__ void k_jv_bilinear(KProb P, KBil B, SweepBuf fw, int type_ew, const double* __restrict__ w, double* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_int * B.n) return;
    const int64_t kl = i / B.n;
    const int r = (int)(i % B.n);
    const int64_t kn = P.kn_lo + kl;
    const int64_t ts = (int64_t)fw.Kpad * fw.npad, col = kl * fw.npad + r;
    const double* wk = w + kn * P.z;
    double v = -fw.S[type_ew * ts + col] - fw.GY[col] * wk[P.dt_idx] + w[(kn + 1) * P.z + B.x_off + r];
    for (int j = 0; j < B.m; ++j) v -= fw.S[(1 + j) * ts + col] * wk[B.u_off + j];
    y[B.row_off + kn * B.n + r] = v;
}
void launch_jv_bilinear(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& fw, int type_ew, const double* w, double* y) {
    const int64_t n = P.n_int * B.n;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_jv_bilinear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, B, fw, type_ew, w, y);
}

__global__ void __launch_bounds__(256) k_jtv_bilinear(KProb P, KBil B, SweepBuf fw, SweepBuf ad, const double* __restrict__ w,
                                                       double* __restrict__ y) {
    __shared__ double sm[4];
    const int64_t kl = blockIdx.x;
    const int64_t kn = P.kn_lo + kl;
    const int n = B.n;
    const int64_t ts = (int64_t)fw.Kpad * fw.npad, col = kl * fw.npad;
    double* yk = y + kn * P.z;
    const bool own = kn < P.K && kl < P.n_int;
    for (int r = threadIdx.x; r < n; r += 256) {
        double v = 0.0;
        if (own) v -= ad.S[col + r];                                   // -exp(A_k)' w_k
        if (kn >= 1) v += w[B.row_off + (kn - 1) * n + r];             // +I' w_{k-1}
        yk[B.x_off + r] += v;   // (every entry of y has ONE writer per launch, launches follow each other on the stream: fixed order)
    }
    if (!own) return;
    const double* wk = w + B.row_off + kn * n;
    for (int j = 0; j <= B.m; ++j) {   // j < m: u_j entries, j == m: the timestep entry
        double s = 0.0;
        const double* c = j < B.m ? fw.S + (1 + j) * ts + col : fw.GY + col;
        for (int r = threadIdx.x; r < n; r += 256) s += c[r] * wk[r];
        s = block_sum_256(s, sm);
        __syncthreads();
        if (threadIdx.x == 0) yk[j < B.m ? B.u_off + j : P.dt_idx] -= s;
    }
}
void launch_jtv_bilinear(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& fw, const SweepBuf& ad, const double* w, double* y) {
    if (P.n_knots <= 0) return;
    hipLaunchKernelGGL(k_jtv_bilinear, dim3((unsigned)P.n_knots), dim3(256), 0, st, P, B, fw, ad, w, y);
}

// DerivativeIntegrator rows: -w_x - dt w_xdot - xdot w_dt + w_x(k+1)
__global__ void k_jv_derivative(KProb P, KDer D, const double* __restrict__ Z, const double* __restrict__ w, double* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_int * D.d) return;
    const int64_t kl = i / D.d;
    const int r = (int)(i % D.d);
    const int64_t kn = P.kn_lo + kl;
    const double* zk = Z + kn * P.z;
    const int64_t row = D.row_off + kn * D.d + r;
    const double* wk = w + kn * P.z;
    y[row] = -wk[D.x_off + r] - zk[P.dt_idx] * wk[D.xdot_off + r] - zk[D.xdot_off + r] * wk[P.dt_idx] + wk[P.z + D.x_off + r];
}
// ... and the transpose: one workgroup per knot, every entry of y with ONE writer per phase, the phases (x entries: own interval
// and the +I of the interval before; xdot entries; the timestep entry, a block sum in a fixed tree) separated by barriers so that
// components that coincide (x and xdot overlapping, the timestep inside one of them) still add in a fixed order -- no atomics.
__global__ void __launch_bounds__(256) k_jtv_derivative(KProb P, KDer D, const double* __restrict__ Z, const double* __restrict__ w,
                                                         double* __restrict__ y) {
    __shared__ double sm[4];
    const int64_t kl = blockIdx.x;
    const int64_t kn = P.kn_lo + kl;
    const bool own = kn < P.K && kl < P.n_int;
    const double* zk = Z + kn * P.z;
    double* yk = y + kn * P.z;
    const double* wk = w + D.row_off + kn * D.d;          // rows of interval kn
    for (int r = threadIdx.x; r < D.d; r += 256) {
        double v = 0.0;
        if (own) v -= wk[r];
        if (kn >= 1) v += w[D.row_off + (kn - 1) * D.d + r];
        yk[D.x_off + r] += v;
    }
    if (!own) return;
    __syncthreads();
    for (int r = threadIdx.x; r < D.d; r += 256) yk[D.xdot_off + r] -= zk[P.dt_idx] * wk[r];
    __syncthreads();
    double s = 0.0;
    for (int r = threadIdx.x; r < D.d; r += 256) s += zk[D.xdot_off + r] * wk[r];
    s = block_sum_256(s, sm);
    if (threadIdx.x == 0) yk[P.dt_idx] -= s;
}
void launch_jv_derivative(hipStream_t st, const KProb& P, const KDer& D, const double* dZ, const double* w, double* y, int transpose) {
    if (!transpose) {
        const int64_t n = P.n_int * D.d;
        if (n <= 0) return;
        hipLaunchKernelGGL(k_jv_derivative, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, D, dZ, w, y);
    } else {
        if (P.n_knots <= 0) return;
        hipLaunchKernelGGL(k_jtv_derivative, dim3((unsigned)P.n_knots), dim3(256), 0, st, P, D, dZ, w, y);
    }
}

// knot constraints: only entries inside the pattern taken at Z0 take part (evaluator.jl:545-547)
// J w: one thread per listed time sums its row over the components in order (rows of different listings are different rows).
// J' w: entry (knot, component) is written by the thread of (listing, component); listings that repeat a knot would meet in one
// entry, so a constraint whose `times` repeat a knot runs the serial form (one workgroup walks the listings in order).
__global__ void k_jv_knot(KProb P, KCon C, const double* __restrict__ Z, const double* __restrict__ w, double* __restrict__ y) {
    const int64_t ti = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ti >= C.n_times) return;
    const int64_t kn = C.times[ti];
    const double* zk = Z + kn * P.z;
    const double nrm = C.kind == 1 ? sqrt(knot_norm2(P, C, zk)) : 1.0;
    double s = 0.0;
    bool any = false;
    for (int c = 0; c < C.n_comps; ++c) {
        if (C.jpos[ti * C.n_comps + c] < 0) continue;
        const double v = zk[C.comps[c]];
        s += (C.kind == 1 ? v / nrm : 2.0 * v) * w[kn * P.z + C.comps[c]];
        any = true;
    }
    if (any) y[C.mu_off + C.tidx[ti]] += s;
}
__device__ __forceinline__ void jtv_knot_one(const KProb& P, const KCon& C, const double* Z, const double* w, double* y, int64_t i) {
    if (C.jpos[i] < 0) return;
    const int64_t ti = i / C.n_comps;
    const int c = (int)(i % C.n_comps);
    const int64_t kn = C.times[ti];
    const double* zk = Z + kn * P.z;
    const double v = zk[C.comps[c]];
    const double jac = C.kind == 1 ? v / sqrt(knot_norm2(P, C, zk)) : 2.0 * v;
    y[kn * P.z + C.comps[c]] += jac * w[C.mu_off + C.tidx[ti]];
}
// serial = 0: one thread per (listing, component); 1: thread c walks component c through the listings in order; 2: one thread all
__global__ void k_jtv_knot(KProb P, KCon C, const double* __restrict__ Z, const double* __restrict__ w, double* __restrict__ y, int serial) {
    const int64_t n = C.n_times * C.n_comps;
    if (serial == 0) {
        const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i < n) jtv_knot_one(P, C, Z, w, y, i);
    } else if (serial == 1) {
        for (int c = threadIdx.x; c < C.n_comps; c += blockDim.x)
            for (int64_t ti = 0; ti < C.n_times; ++ti) jtv_knot_one(P, C, Z, w, y, ti * C.n_comps + c);
    } else if (threadIdx.x == 0) {
        for (int64_t i = 0; i < n; ++i) jtv_knot_one(P, C, Z, w, y, i);
    }
}
void launch_jv_knot(hipStream_t st, const KProb& P, const KCon& C, const double* dZ, const double* w, double* y, int transpose) {
    const int64_t n = C.n_times * C.n_comps;
    if (n <= 0) return;
    if (!transpose) {
        hipLaunchKernelGGL(k_jv_knot, dim3((unsigned)((C.n_times + 255) / 256)), dim3(256), 0, st, P, C, dZ, w, y);
    } else if (C.repeats || C.comp_repeats) {
        // listings that repeat a knot meet in one entry: the listings are then walked in order (components of one listing are
        // different entries unless a component is listed twice: then one thread walks everything)
        hipLaunchKernelGGL(k_jtv_knot, dim3(1), dim3(64), 0, st, P, C, dZ, w, y, C.comp_repeats ? 2 : 1);
    } else {
        hipLaunchKernelGGL(k_jtv_knot, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, C, dZ, w, y, 0);
    }
}

// ============================================================================================
// objectives (O1-O4)
// ============================================================================================


// F = ||A v||^2 of the low-rank infidelity loss, v = z_k[comps]
__device__ __forceinline__ double lowrank_F(const KObj& O, const double* zk) {
    double F = 0.0;
    for (int r = 0; r < O.comp_dim; ++r) {
        double y = 0.0;
        for (int c = 0; c < O.n_comps; ++c) y += O.R[r + (int64_t)O.comp_dim * c] * zk[O.comps[c]];
        F += y * y;
    }
    return F;
}
__device__ __forceinline__ double sign0(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0); }

// partial[b] = sum over this block's times of the term value
__global__ void __launch_bounds__(256) k_objective(KProb P, KObj O, const double* __restrict__ Z, double* __restrict__ partial) {
    __shared__ double sm[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < O.n_times; i += (int64_t)gridDim.x * 256) {
        const int64_t kn = O.times[i];
        const double* zk = Z + kn * P.z;
        const double dt = zk[P.dt_idx];
        if (O.kind == 6) {  // l = |1 - ||A v||^2|  (coherent-fidelity loss; A = O.R, k = O.comp_dim rows)
            acc += O.Qs[i] * fabs(1.0 - lowrank_F(O, zk));
        } else if (O.kind == 4) {  // KnotPointObjective, l = ||v - p||^2: sum_i Q_i l (knot_point_objectives.jl:173-182)
            double s = 0.0;
            for (int c = 0; c < O.n_comps; ++c) {
                const double dv = zk[O.comps[c]] - (O.params ? O.params[i * O.n_comps + c] : 0.0);
                s += dv * dv;
            }
            acc += O.Qs[i] * s;
        } else if (O.kind == 3) {  // MinimumTimeObjective: D * sum_{k<N} dt_k  (minimum_time_objective.jl:44-50)
            acc += dt;
        } else if (O.kind == 1) {  // QuadraticRegularizer value (regularizers.jl:79-91): 1/2 r'(R.*r), r = dt*dv
            double s = 0.0;
            for (int c = 0; c < O.comp_dim; ++c) {
                const double dv = zk[O.comp_off + c] - (O.has_baseline ? O.baseline[kn * O.comp_dim + c] : 0.0);
                const double r = dt * dv;
                s += r * (O.R[c] * r);
            }
            acc += 0.5 * s;
        } else {  // LinearRegularizer (regularizers.jl:240-249): dt * R'v
            double s = 0.0;
            for (int c = 0; c < O.comp_dim; ++c) s += O.R[c] * zk[O.comp_off + c];
            acc += dt * s;
        }
    }
    const double tot = block_sum_256(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
// f += weight * (D *) sum(partial) in fixed order (deterministic)
__global__ void k_objective_final(KObj O, int nblocks, const double* __restrict__ partial, double* f) {
    double s = 0.0;
    for (int i = 0; i < nblocks; ++i) s += partial[i];
    if (O.kind == 3) s *= O.D;
    *f += O.weight * s;
}
void launch_objective(hipStream_t st, const KProb& P, const KObj& O, const double* dZ, double* partial, double* f) {
    if (O.n_times <= 0) return;
    int nb = (int)((O.n_times + 255) / 256);
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(k_objective, dim3(nb), dim3(256), 0, st, P, O, dZ, partial);
    hipLaunchKernelGGL(k_objective_final, dim3(1), dim3(1), 0, st, O, nb, partial, f);
}

// grad += weight * term gradient  (regularizers.jl:93-115, :251-271; minimum_time_objective.jl:52-66;
// composite _objectives.jl:119-128).  grad is the shard-local slab (entry kn*z+c - grad_lo).
__global__ void k_gradient(KProb P, KObj O, const double* __restrict__ Z, double* __restrict__ grad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= O.n_times) return;
    const int64_t kn = O.times[i];
    const double* zk = Z + kn * P.z;
    double* gk = grad + kn * P.z - P.grad_lo;
    const double dt = zk[P.dt_idx];
    if (O.kind == 6) {  // grad l = -2 sign(1 - F) A'(A v); same per-listing overwrite as kind 4
        if (!O.last[i]) return;
        const double sgn = sign0(1.0 - lowrank_F(O, zk));
        for (int c = 0; c < O.n_comps; ++c) {
            double g = 0.0;
            for (int r = 0; r < O.comp_dim; ++r) {
                double y = 0.0;
                for (int c2 = 0; c2 < O.n_comps; ++c2) y += O.R[r + (int64_t)O.comp_dim * c2] * zk[O.comps[c2]];
                g += O.R[r + (int64_t)O.comp_dim * c] * y;
            }
            atomicAdd(&gk[O.comps[c]], O.weight * O.Qs[i] * (-2.0 * sgn) * g);
        }
    } else if (O.kind == 4) {  // gradient! writes 2 Q_i (v - p_i) per listed time, later entries overwrite (knot_point_objectives.jl:184-207)
        if (!O.last[i]) return;
        for (int c = 0; c < O.n_comps; ++c) {
            const double dv = zk[O.comps[c]] - (O.params ? O.params[i * O.n_comps + c] : 0.0);
            atomicAdd(&gk[O.comps[c]], O.weight * O.Qs[i] * 2.0 * dv);
        }
    } else if (O.kind == 3) {
        atomicAdd(&gk[P.dt_idx], O.weight * O.D);
    } else if (O.kind == 1) {
        double s = 0.0;
        for (int c = 0; c < O.comp_dim; ++c) {
            const double dv = zk[O.comp_off + c] - (O.has_baseline ? O.baseline[kn * O.comp_dim + c] : 0.0);
            atomicAdd(&gk[O.comp_off + c], O.weight * (dt * dt * (O.R[c] * dv)));
            s += dv * (O.R[c] * dv);
        }
        atomicAdd(&gk[P.dt_idx], O.weight * (s * dt));
    } else {
        double s = 0.0;
        for (int c = 0; c < O.comp_dim; ++c) {
            atomicAdd(&gk[O.comp_off + c], O.weight * (O.R[c] * dt));
            s += O.R[c] * zk[O.comp_off + c];
        }
        atomicAdd(&gk[P.dt_idx], O.weight * s);
    }
}
void launch_gradient(hipStream_t st, const KProb& P, const KObj& O, const double* dZ, double* grad) {
    if (O.n_times <= 0) return;
    hipLaunchKernelGGL(k_gradient, dim3((unsigned)((O.n_times + 255) / 256)), dim3(256), 0, st, P, O, dZ, grad);
}

// ============================================================================================
// Hessian of the Lagrangian (A2): H zero-filled, then accumulated (+=) in the reference's order:
// integrators, constraints, sigma * objective   (evaluator.jl:560-647).  Only row <= col entries.
// ============================================================================================

__device__ __forceinline__ void hess_add(const KProb& P, double* H, int64_t kn, int a, int b, double v) {
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    atomicAdd(&H[hess_pos(P, kn, lo, hi)], v);
}

// get_full_hessian of the regularizers (regularizers.jl:142-167, :295-313).  NOTE the reference puts
// the (v, dt) cross terms at [v_idx, dt_idx] only; they survive the `row <= col` filter
// (evaluator.jl:637) only when the timestep component follows v inside the knot.
__global__ void k_hess_objective(KProb P, KObj O, const double* __restrict__ Z, double sigma, double* __restrict__ H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= O.n_times || O.kind == 3) return;
    const int64_t kn = O.times[i];
    const double* zk = Z + kn * P.z;
    const double dt = zk[P.dt_idx];
    const double sw = sigma * O.weight;
    if (O.kind == 6) {  // A' G A with G = -2 sign(1 - F) I (knot_hvp.jl:58-66), row <= col entries
        if (!O.last[i]) return;
        const double sgn = sign0(1.0 - lowrank_F(O, zk));
        if (sgn == 0.0) return;
        for (int a = 0; a < O.n_comps; ++a)
            for (int b = 0; b < O.n_comps; ++b) {
                const int ca = O.comps[a], cb = O.comps[b];
                if (ca > cb) continue;
                double h = 0.0;
                for (int r = 0; r < O.comp_dim; ++r) h += O.R[r + (int64_t)O.comp_dim * a] * O.R[r + (int64_t)O.comp_dim * b];
                if (h != 0.0) atomicAdd(&H[hess_pos(P, kn, ca, cb)], sw * O.Qs[i] * (-2.0 * sgn) * h);
            }
    } else if (O.kind == 4) {  // triu of the per-knot Hessian 2 Q_i I (knot_point_objectives.jl:224-243)
        if (!O.last[i]) return;
        for (int c = 0; c < O.n_comps; ++c) {
            bool dup = false;  // a component listed twice: hessian of ||v-p||^2 in the concatenated vector maps both onto one entry
            for (int c2 = 0; c2 < c; ++c2) dup |= O.comps[c2] == O.comps[c];
            (void)dup;
            atomicAdd(&H[hess_pos(P, kn, O.comps[c], O.comps[c])], sw * 2.0 * O.Qs[i]);
        }
    } else if (O.kind == 1) {
        double s = 0.0;
        bool dt_inside = false;
        for (int c = 0; c < O.comp_dim; ++c) {
            const int a = O.comp_off + c;
            const double r = zk[a] - (O.has_baseline ? O.baseline[kn * O.comp_dim + c] : 0.0);
            s += r * (O.R[c] * r);
            if (a == P.dt_idx) { dt_inside = true; continue; }  // overwritten by the later setindex!
            atomicAdd(&H[hess_pos(P, kn, a, a)], sw * dt * dt * O.R[c]);
            if (a < P.dt_idx) atomicAdd(&H[hess_pos(P, kn, a, P.dt_idx)], sw * 2.0 * dt * O.R[c] * r);
        }
        (void)dt_inside;
        atomicAdd(&H[hess_pos(P, kn, P.dt_idx, P.dt_idx)], sw * s);
    } else {
        for (int c = 0; c < O.comp_dim; ++c) {
            const int a = O.comp_off + c;
            if (a <= P.dt_idx) atomicAdd(&H[hess_pos(P, kn, a, P.dt_idx)], sw * O.R[c]);
        }
    }
}
void launch_hess_objective(hipStream_t st, const KProb& P, const KObj& O, const double* dZ, double sigma, double* H) {
    if (O.n_times <= 0 || O.kind == 3) return;
    hipLaunchKernelGGL(k_hess_objective, dim3((unsigned)((O.n_times + 255) / 256)), dim3(256), 0, st, P, O, dZ, sigma, H);
}

// DerivativeIntegrator: only the (xdot_i, dt) cross term, -mu_{k,i}  (derivative_integrator.jl:90-116)
__global__ void k_hess_derivative(KProb P, KDer D, const double* __restrict__ mu, double* __restrict__ H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_int * D.d) return;
    const int64_t kl = i / D.d;
    const int r = (int)(i % D.d);
    const int64_t kn = P.kn_lo + kl;
    const double m = mu[D.row_off + kn * D.d + r];
    if (D.xdot_off + r == P.dt_idx)
        atomicAdd(&H[hess_pos(P, kn, P.dt_idx, P.dt_idx)], -2.0 * m);
    else
        hess_add(P, H, kn, D.xdot_off + r, P.dt_idx, -m);
}
void launch_hess_derivative(hipStream_t st, const KProb& P, const KDer& Dv, const double* dmu, double* H) {
    const int64_t n = P.n_int * Dv.d;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_hess_derivative, dim3((unsigned)((n + 255) / 256)), blk(P, 256), 0, st, P, Dv, dmu, H);
}

// mu_i' * Hessian of g   (knot_point_constraint.jl:275-294), upper triangle of the comps x comps block
__global__ void k_hess_knot(KProb P, KCon C, const double* __restrict__ Z, const double* __restrict__ mu, double* __restrict__ H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nc2 = C.n_comps * C.n_comps;
    if (i >= C.n_times * nc2) return;
    const int64_t ti = i / nc2;
    if (!C.hess_on[ti]) return;
    const int a = (int)((i % nc2) / C.n_comps), b = (int)(i % C.n_comps);
    const int ca = C.comps[a], cb = C.comps[b];
    if (ca > cb) return;
    const int64_t kn = C.times[ti];
    const double* zk = Z + kn * P.z;
    const double m = mu[C.mu_off + C.tidx[ti]];
    double v;
    if (C.kind == 1) {
        const double s = knot_norm2(P, C, zk);
        const double rn = sqrt(s);
        v = m * ((a == b ? 1.0 / rn : 0.0) - zk[ca] * zk[cb] / (rn * s));
    } else {
        v = a == b ? 2.0 * m : 0.0;
    }
    if (v != 0.0) atomicAdd(&H[hess_pos(P, kn, ca, cb)], v);
}
void launch_hess_knot(hipStream_t st, const KProb& P, const KCon& C, const double* dZ, const double* dmu, double* H) {
    const int64_t n = C.n_times * C.n_comps * C.n_comps;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_hess_knot, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, C, dZ, dmu, H);
}

// ============================================================================================
// host-evaluated knot terms: the caller ran the reference's own closures; the
// engine only places the blocks.  Same placement rules as the built-in kinds above.
// ============================================================================================
// external integrator: defects of the owned intervals
__global__ void k_extint_cons(KProb P, KExtInt E, const double* __restrict__ src, double* __restrict__ g) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_int * E.d) return;
    g[E.lrow_off + i] = src[P.kn_lo * E.d + i];
}
void launch_extint_cons(hipStream_t st, const KProb& P, const KExtInt& E, const double* vals, double* g) {
    const int64_t n = P.n_int * E.d;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_extint_cons, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, E, vals, g);
}
// Jacobian entries of the owned knots' column blocks: rows of interval kn-1 take the z_{k+1} half of that interval's
// block, rows of interval kn the z_k half of its own (blocks: [N-1] x (d x 2z) column-major)
__global__ void k_extint_jac(KProb P, KExtInt E, const double* __restrict__ blocks, double* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)P.z * E.d;
    if (i >= P.n_knots * per) return;
    const int64_t kn = P.kn_lo + i / per;
    const int j = (int)((i % per) / E.d), r = (int)(i % E.d);
    const int64_t bs = (int64_t)E.d * 2 * P.z;
    if (kn >= 1) vals[jac_pos(P, P.colptr, kn, j, E.pre, E.d, 0, r)] = blocks[(kn - 1) * bs + (int64_t)(P.z + j) * E.d + r];
    if (kn < P.K) vals[jac_pos(P, P.colptr, kn, j, E.pre, E.d, 1, r)] = blocks[kn * bs + (int64_t)j * E.d + r];
}
void launch_extint_jac(hipStream_t st, const KProb& P, const KExtInt& E, const double* blocks, double* vals) {
    const int64_t n = P.n_knots * P.z * E.d;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_extint_jac, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, E, blocks, vals);
}
// Hessian blocks ([N-1] x (2z x 2z) column-major) into the owned knots' columns: diagonal block (kn,kn) gets the
// z_k part of interval kn and the z_{k+1} part of interval kn-1, off-diagonal block (kn-1,kn) the cross part of kn-1
__global__ void k_extint_hess(KProb P, const double* __restrict__ blocks, double* __restrict__ H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t z = P.z, per = z * z;
    if (i >= P.n_knots * per) return;
    const int64_t kn = P.kn_lo + i / per;
    const int b = (int)((i % per) / z), a = (int)(i % z);
    const int64_t bs = 4 * z * z, ld = 2 * z;
    if (a <= b) {
        double v = 0.0;
        if (kn < P.K) v += blocks[kn * bs + a + ld * b];
        if (kn >= 1) v += blocks[(kn - 1) * bs + (z + a) + ld * (z + b)];
        if (v != 0.0) atomicAdd(&H[hess_pos(P, kn, a, b)], v);
    }
    if (kn >= 1) {
        const double v = blocks[(kn - 1) * bs + a + ld * (z + b)];
        if (v != 0.0) atomicAdd(&H[hess_pos_off(P, kn, a, b)], v);
    }
}
void launch_extint_hess(hipStream_t st, const KProb& P, const KExtInt& E, const double* blocks, double* H) {
    (void)E;
    const int64_t n = P.n_knots * P.z * P.z;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_extint_hess, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, blocks, H);
}

__global__ void k_ext_cons(KCon C, const double* __restrict__ src, double* __restrict__ g) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C.n_times * C.g_dim) return;
    const int64_t ti = i / C.g_dim;
    const int r = (int)(i % C.g_dim);
    g[C.lrow[ti] + r] = src[C.tidx[ti] * C.g_dim + r];
}
void launch_ext_cons(hipStream_t st, const KCon& C, const double* vals, double* g) {
    const int64_t n = C.n_times * C.g_dim;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_ext_cons, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, C, vals, g);
}

// blocks: [n_times_total] x (g_dim x n_comps column-major); jpos: [n_times][n_comps][g_dim]
__global__ void k_ext_jac(KCon C, const double* __restrict__ blocks, double* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = C.n_comps * C.g_dim;
    if (i >= C.n_times * per) return;
    const int64_t p = C.jpos[i];
    if (p < 0) return;
    const int64_t ti = i / per;
    vals[p] = blocks[C.tidx[ti] * per + (i % per)];
}
void launch_ext_jac(hipStream_t st, const KCon& C, const double* blocks, double* vals) {
    const int64_t n = C.n_times * C.n_comps * C.g_dim;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_ext_jac, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, C, blocks, vals);
}

// global variable index of block element a of listing ti: knot part first, then the global variables
__device__ __forceinline__ int64_t ext_var(const KProb& P, const KExtTerm& E, int64_t ti, int a) {
    return a < E.nc ? E.times[ti] * P.z + E.comps[a] : P.N * P.z + E.gcomps[a - E.nc];
}
// H[row <= col entries of the listing's block] += scale * block (column-major nb x nb, nb = nc + ng)
__global__ void k_ext_hess(KProb P, KExtTerm E, double scale, const double* __restrict__ blocks, double* __restrict__ H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nb = E.nc + E.ng, nb2 = nb * nb;
    if (i >= E.n_list * nb2) return;
    const int64_t ti = i / nb2;
    const int b = (int)((i % nb2) / nb), a = (int)(i % nb);  // element (a, b)
    const int64_t R = ext_var(P, E, ti, a), C = ext_var(P, E, ti, b);
    if (R > C) return;  // row <= col (evaluator.jl:637)
    const double v = scale * blocks[E.tidx[ti] * nb2 + (i % nb2)];
    if (v == 0.0) return;
    if (b < E.nc) {  // knot column: both indices in knot times[ti]
        if (E.knot_on[ti]) atomicAdd(&H[hess_pos(P, E.times[ti], E.comps[a], E.comps[b])], v);
    } else if (E.glob_on && a < E.nc) {  // (knot row, global column): one listing per entry; (global, global): k_ext_hess_glob
        const int64_t p = hess_pos_tail(P, R, E.gcomps[b - E.nc]);
        if (p >= 0) atomicAdd(&H[p], v);
    }
}
// Entries whose row AND column are global variables receive a contribution from EVERY listing of the term
// (global_objectives.jl:270-271, :341 accumulate): one thread per entry adds them in listing order -- no atomics, fixed order.
__global__ void k_ext_hess_glob(KProb P, KExtTerm E, double scale, const double* __restrict__ blocks, double* __restrict__ H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E.ng * E.ng || !E.glob_on) return;
    const int gb = i / E.ng, ga = i % E.ng, nb = E.nc + E.ng, nb2 = nb * nb;
    const int64_t R = P.N * P.z + E.gcomps[ga], C = P.N * P.z + E.gcomps[gb];
    if (R > C) return;
    const int64_t p = hess_pos_tail(P, R, E.gcomps[gb]);
    if (p < 0) return;
    double s = 0.0;
    for (int64_t ti = 0; ti < E.n_list; ++ti) s += scale * blocks[E.tidx[ti] * nb2 + (int64_t)(E.nc + gb) * nb + (E.nc + ga)];
    // two block elements can name the same pair of global variables (a component listed twice): those meet here
    if (s != 0.0) atomicAdd(&H[p], s);
}
void launch_ext_hess(hipStream_t st, const KProb& P, const KExtTerm& E, double scale, const double* blocks, double* H) {
    const int64_t nb = E.nc + E.ng, n = E.n_list * nb * nb;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_ext_hess, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, E, scale, blocks, H);
    if (E.ng > 0 && E.glob_on)
        hipLaunchKernelGGL(k_ext_hess_glob, dim3((unsigned)((E.ng * E.ng + 63) / 64)), dim3(64), 0, st, P, E, scale, blocks, H);
}

// f += weight * sum of the counted listings' values, fixed order within the block (deterministic)
__global__ void __launch_bounds__(256) k_ext_objective(KExtTerm E, double weight, const double* __restrict__ vals, double* f) {
    __shared__ double sm[4];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < E.n_list; i += 256)
        if (E.count[i]) acc += vals[E.tidx[i]];
    const double tot = block_sum_256(acc, sm);
    if (threadIdx.x == 0) *f += weight * tot;
}
void launch_ext_objective(hipStream_t st, const KExtTerm& E, double weight, const double* vals, double* f) {
    if (E.n_list <= 0) return;
    hipLaunchKernelGGL(k_ext_objective, dim3(1), dim3(256), 0, st, E, weight, vals, f);
}

__global__ void k_ext_gradient(KProb P, KExtTerm E, double weight, const double* __restrict__ blocks, double* __restrict__ grad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nb = E.nc + E.ng;
    if (i >= E.n_list * nb) return;
    const int64_t ti = i / nb;
    const int a = (int)(i % nb);
    if (a >= E.nc || !E.knot_on[ti]) return;  // the global part: k_ext_gradient_glob
    atomicAdd(&grad[ext_var(P, E, ti, a) - P.grad_lo], weight * blocks[E.tidx[ti] * nb + a]);
}
// gradient entries of the global variables: every listing contributes; one thread per entry, listing order, no atomics
__global__ void k_ext_gradient_glob(KProb P, KExtTerm E, double weight, const double* __restrict__ blocks, double* __restrict__ grad) {
    const int ga = blockIdx.x * blockDim.x + threadIdx.x;
    if (ga >= E.ng || !E.glob_on) return;
    const int nb = E.nc + E.ng;
    double s = 0.0;
    for (int64_t ti = 0; ti < E.n_list; ++ti) s += weight * blocks[E.tidx[ti] * nb + E.nc + ga];
    atomicAdd(&grad[P.N * P.z + E.gcomps[ga] - P.grad_lo], s);
}
void launch_ext_gradient(hipStream_t st, const KProb& P, const KExtTerm& E, double weight, const double* blocks, double* grad) {
    const int64_t n = E.n_list * (E.nc + E.ng);
    if (n <= 0) return;
    hipLaunchKernelGGL(k_ext_gradient, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P, E, weight, blocks, grad);
    if (E.ng > 0 && E.glob_on)
        hipLaunchKernelGGL(k_ext_gradient_glob, dim3((unsigned)((E.ng + 63) / 64)), dim3(64), 0, st, P, E, weight, blocks, grad);
}

// Bilinear block of mu_k' f  (bilinear_integrator.jl:135-161).  With y = exp(A)x, c_j = dexp(A)[dt G_j]x,
// h_ij = d2exp(A)[dt G_i, dt G_j]x (forward sweep fw), yt = exp(A')mu, ct_j = dexp(A')[dt G_j']mu
// (adjoint sweep ad), W_j = G_j' mu, Gm = G(u)' mu = sum_j ubar_j W_j:
//   (x, u_j)   = -ct_j               (x, dt)    = -G(u)' yt         (= ad.GY)
//   (u_i,u_j)  = -mu' h_ij           (u_j, dt)  = -(W_j' y + Gm' c_j)
//   (dt, dt)   = -Gm' (G(u) y)       (G(u) y = fw.GY)
__global__ void __launch_bounds__(256) k_hess_bilinear(KProb P, KBil B, SweepBuf fw, SweepBuf ad,
                                                        const double* __restrict__ mu, double* __restrict__ H, int with_uu) {
    __shared__ double sm[4];
    const int64_t kl = blockIdx.x;
    const int64_t kn = P.kn_lo + kl;
    const int n = B.n, m = B.m, npad = fw.npad;
    const int64_t ts = (int64_t)fw.Kpad * npad;
    const int64_t col = kl * npad;
    const double* muk = mu + B.row_off + kn * n;
    // vector blocks
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        for (int j = 0; j < m; ++j) hess_add(P, H, kn, B.x_off + r, B.u_off + j, -ad.S[(1 + j) * ts + col + r]);
        hess_add(P, H, kn, B.x_off + r, P.dt_idx, -ad.GY[col + r]);
    }
    // scalar blocks: block-wide dot products
    if (!with_uu) {
        // pairing path: only the ADJOINT sweep carries tangents.  With V_l = G_l x (fw.W), gx = G(u) x, pt = exp(A')mu and
        // dt^j = L(A, dt G_j)' mu (ad.S), the identity  G L(A,E) = L(A,E) G + (exp(A) E - E exp(A))/dt  (E = dt G_j) turns
        //   d2/du_j ddt = mu' G_j exp(A) x + mu' G L(A,E) x   into   dt^j . gx + pt . V_j,
        // and  d2/ddt2 = mu' G G exp(A) x = (G' pt) . gx  since G commutes with exp(A): no forward tangent, no forward sum.
        auto gx = [&](int r) {
            double s = 0.0;
            for (int l = 0; l <= m; ++l) s += ad.scaleU[(int64_t)l * ad.Kpad + kl] * fw.W[l * ts + col + r];
            return s;
        };
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int r = threadIdx.x; r < n; r += blockDim.x)
                s += ad.S[(1 + j) * ts + col + r] * gx(r) + ad.S[col + r] * fw.W[(1 + j) * ts + col + r];
            s = block_sum_256(s, sm);
            __syncthreads();
            if (threadIdx.x == 0) hess_add(P, H, kn, B.u_off + j, P.dt_idx, -s);
        }
        double s = 0.0;
        for (int r = threadIdx.x; r < n; r += blockDim.x) s += ad.GY[col + r] * gx(r);
        s = block_sum_256(s, sm);
        if (threadIdx.x == 0) atomicAdd(&H[hess_pos(P, kn, P.dt_idx, P.dt_idx)], -s);
        return;
    }
    auto gm = [&](int r) {
        double s = 0.0;
        for (int l = 0; l <= m; ++l) s += fw.scaleU[(int64_t)l * fw.Kpad + kl] * ad.W[l * ts + col + r];
        return s;
    };
    int hidx = 1 + m;
    for (int i = 0; i < m; ++i)
        for (int j = i; j < m; ++j, ++hidx) {
            double s = 0.0;
            for (int r = threadIdx.x; r < n; r += blockDim.x) s += muk[r] * fw.S[hidx * ts + col + r];
            s = block_sum_256(s, sm);
            __syncthreads();
            if (threadIdx.x == 0) hess_add(P, H, kn, B.u_off + i, B.u_off + j, -s);
        }
    for (int j = 0; j < m; ++j) {
        double s = 0.0;
        for (int r = threadIdx.x; r < n; r += blockDim.x)
            s += ad.W[(1 + j) * ts + col + r] * fw.S[col + r] + gm(r) * fw.S[(1 + j) * ts + col + r];
        s = block_sum_256(s, sm);
        __syncthreads();
        if (threadIdx.x == 0) hess_add(P, H, kn, B.u_off + j, P.dt_idx, -s);
    }
    {
        double s = 0.0;
        for (int r = threadIdx.x; r < n; r += blockDim.x) s += gm(r) * fw.GY[col + r];
        s = block_sum_256(s, sm);
        if (threadIdx.x == 0) atomicAdd(&H[hess_pos(P, kn, P.dt_idx, P.dt_idx)], -s);
    }
}
void launch_hess_bilinear(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& fw, const SweepBuf& ad,
                          const double* dmu, double* H, int with_uu) {
    if (P.n_int <= 0) return;
    hipLaunchKernelGGL(k_hess_bilinear, dim3((unsigned)P.n_int), blk(P, 256), 0, st, P, B, fw, ad, dmu, H, with_uu);
}

}  // namespace dto
