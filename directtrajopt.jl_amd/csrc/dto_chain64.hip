// dto_chain64.hip -- the propagator chain of a 33..64-state integrator as ONE launch (gfx950, round 4).
//
// exp(dt_k G(u_k)) for every interval, -E_k stored straight into the x_k columns of the Jacobian slab
// (bilinear_integrator.jl:111-131: the x block of the interval's Jacobian).  The general path runs this as a sequence of batched
// GEMM launches over all intervals (A, three powers, norms, scaling parameters, the factor K, two or three polynomial products,
// squarings; a host readback in the middle decides the launch sequence): at 64 states each of those launches is 17-49 us for
// 0.5 GFLOP, ~20 launches and two host round trips per call -- 0.55 ms of a 0.65 ms Jacobian at 64 x 1000, an order of magnitude
// above the arithmetic (profiles/r04a_shapes.md).  A 64 x 64 matrix is 32 KB: a workgroup keeps the whole chain of ONE interval
// in LDS and registers.
//
//   workgroup = 4 wavefronts, one interval at a time (grid-stride); wavefront w owns the columns 16 w .. 16 w + 15 of every
//   result (a 64 x 16 strip: four 16 x 16 accumulator tiles, 16 doubles per lane)
//   LDS: four 64 x 64 matrices, column-major, pitch 66 (conflict-free ds_read_b64 for the right operand's fragments)
//   A = dt (G_0 + sum u_j G_j)          streamed from the shared generators (L2)
//   A^2 = A A, A^3 = A^2 A, A^4 = A^2 A^2   (FP64 MFMA 16x16x4, both operands from LDS)
//   1-norms of A^2..A^4 from the strips (column sums inside a DPP row, maxima through LDS: fixed order), alpha, the squaring
//   counts of both evaluation forms -- the form is chosen PER INTERVAL (fewer products + squarings), no host involved
//   the polynomials K, Pa, Pb, Pc (two products) or K, Pa, Pb, PL, PR, Pe (three) are evaluated into REGISTERS strip by strip
//   from the four powers, which then free three of the LDS matrices for the operands of the products
//   Y = A^4 K,  (Y + Pa)(Y + Pb) [+ Pc | -> L, R -> L R + Pe],  s squarings,  -E into the slab
// The constants, radii and the scaling rule are those of the general path (dto_kernels.h EXPM2_* / EXPM3_*, k_expm_params,
// k_expm_coef): the same approximants of the same orders, so the two paths agree to rounding.
// Per interval ~6 products of 2 * 64^3 flops: ~3 MFLOP and ~15-20 us of one CU.  Also written: the exact ||A^2||, ||A^3||, ||A^4||
// per interval (k_hump plans the sweep from them), max_k min(d2, max(d3, d4)) and the largest squaring count.
//
// Bound: FP64 MFMA per interval, launch-free.
#include "dto_gemm.hip.h"
#include "dto_kernels.h"

namespace dto {

namespace {

typedef unsigned int c64_u4 __attribute__((ext_vector_type(4)));
constexpr int C64_P = 66;              // LDS pitch of a 64 x 64 matrix (doubles)
constexpr int C64_MAT = 64 * C64_P;    // doubles per matrix

struct Chain64Args {
    KProb P;
    KBil B;
    const double* Z;
    double* vals;          // Jacobian slab (shard-local)
    double* norms;         // [n_int][4]: INF, ||A^2||_1, ||A^3||_1, ||A^4||_1
    int32_t* smax;         // [0] max squarings (atomicMax)
    unsigned long long* d2max;   // max_k min(d2, max(d3, d4)) as a bit pattern
    int32_t* sk;           // [n_int] squarings used (diagnostics, host hand-off planning)
    int s_cap, force_form;
#ifdef C64_STAMP
    unsigned long long* stamp;   // diagnostic builds (tools/chain64_probe -DC64_STAMP): s_memtime at the phase boundaries of block 0's first interval
#endif
};
#ifdef C64_STAMP
#define C64_MARK(i) do { if (blockIdx.x == 0 && tid == 0 && kl == blockIdx.x) a.stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define C64_MARK(i) do { } while (0)
#endif

// sum over the 16 lanes of a DPP row (all lanes of the row get it), fixed order
template <int CTRL>
__device__ __forceinline__ double c64_dpp(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double c64_row16_sum(double v) {
    v += c64_dpp<0x128>(v);   // row_ror:8
    v += c64_dpp<0x124>(v);
    v += c64_dpp<0x122>(v);
    v += c64_dpp<0x121>(v);
    return v;
}

// strip of X * Y owned by this wavefront: acc[ti][r] = C[16 ti + lr][16 w + 4 r + lq]
__device__ __forceinline__ void c64_product(const double* __restrict__ X, const double* __restrict__ Y, int wave, int lr, int lq, d4 (&acc)[4]) {
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) acc[ti] = d4{0.0, 0.0, 0.0, 0.0};
    const double* yp = Y + (16 * wave + lr) * C64_P + lq;     // right operand: column 16 w + lr, row 4 ks + lq
    const double* xp = X + lq * C64_P + lr;                    // left operand: column 4 ks + lq, row 16 ti + lr
    // the fragments of k-step ks + 1 are read before the MFMAs of k-step ks are issued: one wavefront per SIMD has nobody else
    // to hide its LDS latency behind
    double yf = yp[0], xf[4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) xf[ti] = xp[16 * ti];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        double yn = 0.0, xn[4] = {0.0, 0.0, 0.0, 0.0};
        if (ks + 1 < 16) {
            yn = yp[4 * (ks + 1)];
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) xn[ti] = xp[4 * (ks + 1) * C64_P + 16 * ti];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(yf, xf[ti], acc[ti], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        yf = yn;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) xf[ti] = xn[ti];
    }
}
__device__ __forceinline__ void c64_store(double* __restrict__ M, int wave, int lr, int lq, const d4 (&acc)[4]) {
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) M[(16 * wave + 4 * r + lq) * C64_P + 16 * ti + lr] = acc[ti][r];
}
// max over the strip's 16 columns of the column abs sums (all 64 rows): the wavefront's share of ||M||_1
__device__ __forceinline__ double c64_strip_norm1(const d4 (&acc)[4]) {
    double mx = 0.0;
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double s = fabs(acc[0][r]) + fabs(acc[1][r]) + fabs(acc[2][r]) + fabs(acc[3][r]);   // this lane's rows of column 4 r + lq
        s = c64_row16_sum(s);                                                                // the column's 64 rows
        bad = bad || !(s == s);
        mx = fmax(mx, s);
    }
    // the four DPP rows (lq) hold different columns: maximum across them
    mx = fmax(mx, __shfl_xor(mx, 16, 64));
    mx = fmax(mx, __shfl_xor(mx, 32, 64));
    return __any(bad) ? __longlong_as_double(0x7ff8000000000000ll) : mx;
}

__global__ void __launch_bounds__(256, 1) k_chain64(Chain64Args a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* M0 = lds;                  // A, later K, later the results of products (ping)
    double* M1 = lds + C64_MAT;        // A^2, later Y + Pa, later L
    double* M2 = lds + 2 * C64_MAT;    // A^3, later Y + Pb, later R / squaring pong
    double* M3 = lds + 3 * C64_MAT;    // A^4
    double* red = lds + 4 * C64_MAT;   // [3][4] norm shares of the wavefronts
    const int tid0 = threadIdx.x;
    const int n = a.B.n, m = a.B.m;
    for (int64_t kl = blockIdx.x; kl < a.P.n_int; kl += gridDim.x) {
        // the lane coordinates are made opaque once per interval: otherwise the compiler computes the ~100 LDS offsets of the
        // unrolled product loops ONCE, outside this loop, and keeps them alive across it -- in scratch (46 spilled registers, reloaded
        // in the middle of the MFMA streams); recomputing an offset is one integer instruction
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
        const int64_t kn = a.P.kn_lo + kl;
        const double* zk = a.Z + kn * a.P.z;
        __syncthreads();   // the previous interval's LDS is done with
        C64_MARK(0);
        // ---- A = dt (G_0 + sum_j u_j G_j): one pass over the generators (padded 64 x 64, column-major)
        {
            const double dt = zk[a.P.dt_idx];
            double ub[MAX_DRIVES + 1];
            ub[0] = dt;
            for (int j = 0; j < m; ++j) ub[j + 1] = dt * zk[a.B.u_off + j];
            if (m <= 4) {
                // five generators at most: all 40 loads of this lane in flight at once (coefficient 0 and generator 0 beyond m)
                // (buffer loads: one 32-bit lane offset, the generator and the column group in the scalar offset -- no address
                // registers to keep per load)
                const int voff = ((tid >> 5) * 64 + (tid & 31) * 2) * 8;   // 16-byte unit tid: column tid / 32, rows 2 (tid % 32)
                double* const dst0 = M0 + (tid >> 5) * C64_P + (tid & 31) * 2;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.B.G), 0, (m + 1) * 32768, 0x00020000);
                union U { c64_u4 u; d2 d; };
                U g[5][8];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int sj = (j <= m ? j : 0) * 32768;
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[j][u].u = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sj + 4096 * u, 0);   // 8 columns on
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    d2 s = d2{0.0, 0.0};
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const double cj = j <= m ? ub[j] : 0.0;
                        s.x += cj * g[j][u].d.x; s.y += cj * g[j][u].d.y;
                    }
                    *reinterpret_cast<d2*>(dst0 + 8 * u * C64_P) = s;
                }
            } else {
                for (int i = tid; i < 64 * 32; i += 256) {       // 16-byte units
                    const int c = i >> 5, r2 = (i & 31) * 2;
                    d2 s = d2{0.0, 0.0};
                    for (int j = 0; j <= m; ++j) {
                        const d2 g = *reinterpret_cast<const d2*>(a.B.G + (int64_t)j * 4096 + c * 64 + r2);
                        s.x += ub[j] * g.x; s.y += ub[j] * g.y;
                    }
                    *reinterpret_cast<d2*>(M0 + c * C64_P + r2) = s;
                }
            }
        }
        __syncthreads();
        C64_MARK(1);
        // ---- powers and their 1-norms
        d4 acc[4];
        double nrm[3];
        c64_product(M0, M0, wave, lr, lq, acc);
        nrm[0] = c64_strip_norm1(acc);
        c64_store(M1, wave, lr, lq, acc);
        __syncthreads();
        C64_MARK(2);
        c64_product(M1, M0, wave, lr, lq, acc);
        nrm[1] = c64_strip_norm1(acc);
        c64_store(M2, wave, lr, lq, acc);
        c64_product(M1, M1, wave, lr, lq, acc);
        nrm[2] = c64_strip_norm1(acc);
        c64_store(M3, wave, lr, lq, acc);
        if (lane == 0) { red[0 * 4 + wave] = nrm[0]; red[1 * 4 + wave] = nrm[1]; red[2 * 4 + wave] = nrm[2]; }
        __syncthreads();
        C64_MARK(3);
        double N[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const double v0 = red[q * 4], v1 = red[q * 4 + 1], v2 = red[q * 4 + 2], v3 = red[q * 4 + 3];
            N[q] = fmax(fmax(v0, v1), fmax(v2, v3));
            if (!(v0 == v0) || !(v1 == v1) || !(v2 == v2) || !(v3 == v3)) N[q] = __longlong_as_double(0x7ff8000000000000ll);
        }
        // ---- scaling and form (k_expm_params / k_expm_coef of the general path, per interval)
        const double d2v = sqrt(N[0]), d3v = cbrt(N[1]), d4v = sqrt(sqrt(N[2]));
        const double alpha = fmin(fmax(d2v, d3v), fmax(d3v, d4v));
        int s2 = 0, s3 = 0;
        if (alpha > THETA_16) { s2 = (int)ceil(log2(alpha / THETA_16)); if (s2 < 1) s2 = 1; }
        if (alpha > THETA_3P) { s3 = (int)ceil(log2(alpha / THETA_3P)); if (s3 < 1) s3 = 1; }
        if (s2 > a.s_cap) s2 = a.s_cap;
        if (s3 > a.s_cap) s3 = a.s_cap;
        if (!(alpha == alpha)) { s2 = 1; s3 = 1; }      // NaN input: one squaring, the NaN propagates to the output
        int form = 3 + s3 < 2 + s2 ? 3 : 2;            // products + squarings
        if (a.force_form == 2 || a.force_form == 3) form = a.force_form;
        const int s = form == 3 ? s3 : s2;
        if (tid == 0) {
            a.norms[kl * 4 + 0] = INFINITY;
            a.norms[kl * 4 + 1] = N[0]; a.norms[kl * 4 + 2] = N[1]; a.norms[kl * 4 + 3] = N[2];
            a.sk[kl] = s;
            atomicMax(&a.smax[0], s);
            double beta = fmin(d2v, fmax(d3v, d4v));
            if (!(d2v == d2v) || !(d3v == d3v) || !(d4v == d4v)) beta = __longlong_as_double(0x7ff8000000000000ll);
            atomicMax(a.d2max, (unsigned long long)__double_as_longlong(beta));   // beta >= 0 or NaN: a NaN's bits exceed every finite value's
        }
        const double sigma = ldexp(1.0, -s);
        double cK[5], cA[5], cB[5], cC[5], cL[6], cR[6];
        {
            double sp = 1.0;
            const double s4 = (sigma * sigma) * (sigma * sigma);
#pragma unroll
            for (int i = 0; i <= 4; ++i) {
                if (form == 3) {
                    cC[i] = EXPM3_E[i] * sp; cA[i] = EXPM3_A[i] * sp; cB[i] = EXPM3_B[i] * sp; cK[i] = EXPM3_K[i] * sp * s4;
                    cL[i] = (EXPM3_C[i] - EXPM3_AL * EXPM3_A[i]) * sp; cR[i] = (EXPM3_D[i] - EXPM3_BE * EXPM3_A[i]) * sp;
                } else {
                    cC[i] = EXPM2_C[i] * sp; cA[i] = EXPM2_A[i] * sp; cB[i] = EXPM2_B[i] * sp; cK[i] = EXPM2_K[i] * sp * s4;
                    cL[i] = 0.0; cR[i] = 0.0;
                }
                sp *= sigma;
            }
            cL[5] = EXPM3_AL; cR[5] = EXPM3_BE;
        }
        C64_MARK(4);
        // ---- the polynomials at this lane's 16 elements, from the four powers: element e = (ti, r) is row 16 ti + lr,
        // column 16 w + 4 r + lq
        d4 pK[4], pA[4], pB[4], pC[4], pL[4], pR[4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + lr, col = 16 * wave + 4 * r + lq;
                const int off = col * C64_P + row;
                const double a1 = M0[off], a2 = M1[off], a3 = M2[off], a4 = M3[off];
                const double id = row == col ? 1.0 : 0.0;
                pK[ti][r] = cK[0] * id + cK[1] * a1 + cK[2] * a2 + cK[3] * a3 + cK[4] * a4;
                pA[ti][r] = cA[0] * id + cA[1] * a1 + cA[2] * a2 + cA[3] * a3 + cA[4] * a4;
                pB[ti][r] = cB[0] * id + cB[1] * a1 + cB[2] * a2 + cB[3] * a3 + cB[4] * a4;
                pC[ti][r] = cC[0] * id + cC[1] * a1 + cC[2] * a2 + cC[3] * a3 + cC[4] * a4;
                pL[ti][r] = cL[0] * id + cL[1] * a1 + cL[2] * a2 + cL[3] * a3 + cL[4] * a4;
                pR[ti][r] = cR[0] * id + cR[1] * a1 + cR[2] * a2 + cR[3] * a3 + cR[4] * a4;
            }
        __syncthreads();   // every lane has read the powers: A, A^2, A^3 may go (A^4 stays: left operand of Y)
        C64_MARK(5);
        c64_store(M0, wave, lr, lq, pK);
        __syncthreads();
        C64_MARK(6);
        // ---- Y = A^4 K -> Ya = Y + Pa (M1), Yb = Y + Pb (M2)
        c64_product(M3, M0, wave, lr, lq, acc);
        d4 ya[4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) { ya[ti][r] = acc[ti][r] + pA[ti][r]; pB[ti][r] += acc[ti][r]; }
        c64_store(M1, wave, lr, lq, ya);
        c64_store(M2, wave, lr, lq, pB);
        __syncthreads();
        C64_MARK(7);
        c64_product(M1, M2, wave, lr, lq, acc);   // Ya Yb
        double* cur = M0;      // where the current result lives
        double* oth = M3;
        if (form == 2) {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ti][r] += pC[ti][r];
        } else {
            // L = Y2 + al Ya + PL, R = Y2 + be Ya + PR, then L R + Pe
            d4 Lm[4], Rm[4];
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Lm[ti][r] = acc[ti][r] + cL[5] * ya[ti][r] + pL[ti][r];
                    Rm[ti][r] = acc[ti][r] + cR[5] * ya[ti][r] + pR[ti][r];
                }
            __syncthreads();   // every wave is done with Ya, Yb as operands
            c64_store(M1, wave, lr, lq, Lm);
            c64_store(M2, wave, lr, lq, Rm);
            __syncthreads();
            c64_product(M1, M2, wave, lr, lq, acc);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ti][r] += pC[ti][r];
        }
        C64_MARK(8);
        // ---- squarings
        for (int it = 0; it < s; ++it) {
            __syncthreads();   // the operands of the product that made `acc` are free
            c64_store(cur, wave, lr, lq, acc);
            __syncthreads();
            c64_product(cur, cur, wave, lr, lq, acc);
            double* t = cur; cur = oth; oth = t;
        }
        C64_MARK(9);
        // ---- -E_k into the x_k columns of the slab: column x_off + col of knot k, rows of interval k (part 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = 16 * wave + 4 * r + lq;
            if (col < n) {
                const int64_t base = jac_pos(a.P, a.P.colptr, kn, a.B.x_off + col, a.B.pre, n, 1, 0);
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    const int row = 16 * ti + lr;
                    if (row < n) a.vals[base + row] = -acc[ti][r];
                }
            }
        }
        C64_MARK(10);
    }
}

}  // namespace

#ifdef C64_STAMP
unsigned long long* c64_stamp_buffer = nullptr;   // set by the probe
#endif

hipError_t chain64_prepare() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain64), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_chain64(hipStream_t st, const KProb& P, const KBil& B, const double* dZ, double* vals, double* norms, int32_t* smax,
                          unsigned long long* d2max, int32_t* sk, int s_cap, int force_form, int n_cu) {
    if (B.npad != 64 || P.n_int <= 0) return hipErrorInvalidValue;
    Chain64Args a{};
    a.P = P; a.B = B; a.Z = dZ; a.vals = vals; a.norms = norms; a.smax = smax; a.d2max = d2max; a.sk = sk;
    a.s_cap = s_cap; a.force_form = force_form;
#ifdef C64_STAMP
    a.stamp = c64_stamp_buffer;
#endif
    const size_t lds = ((size_t)4 * C64_MAT + 16) * sizeof(double);
    const int64_t grid = P.n_int < (int64_t)n_cu ? P.n_int : (int64_t)n_cu;   // one 135 KB workgroup per CU, grid-stride over the intervals
    hipLaunchKernelGGL(k_chain64, dim3((unsigned)grid), dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace dto
