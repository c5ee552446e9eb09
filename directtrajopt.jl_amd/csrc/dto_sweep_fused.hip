// dto_sweep_fused.hip -- the generator sweep as ONE persistent launch (gfx950).
//
// The Taylor recurrences of the sweep (exp(A_k)x_k, its u-tangents, the adjoint quantities; dto_kernels.hip, "generator
// sweep") couple nothing across intervals: column k of term t+1 depends on column k of term t alone.  So a workgroup
// that owns a few intervals (all rows, all column types) can run the whole series by itself, with its term columns in LDS
// and only the shared generators streamed from L2 -- no kernel boundary, no host readback and no re-read of the term
// panel per row tile (the step-per-launch form k_sweep re-reads it four times and pays 28 launch gaps per Jacobian).
//
//   workgroup = 4 wavefronts; wavefront w owns rows 16 MT w .. 16 MT (w+1) of the npad = 64 MT rows (npad = 64, 128, 256)
//   columns   = T types x ipw intervals, c = type*ipw + i, padded to NT tiles of 16 (T = 5, ipw = 9 -> 45 of 48 at the
//               headline shape); the term columns Zs live in LDS ([column][row], pitch npad + 2: conflict-free ds_read_b64)
//   per step  : acc = sum_g G_g * (cA_g .* Zs + cB_g .* Zs[source columns])      (FP64 MFMA 16x16x4, issued transposed)
//               A fragments come straight from global memory (L2) in the accumulator's row permutation: a wavefront reads
//               only ITS rows of the generators, nothing is shared, nothing needs a barrier; the k-steps of all generators
//               form one circular stream read 4 k-steps ahead, entered at a point that differs between workgroups (all CUs
//               asking the L2 for the same lines at the same moment cost 8 %).  B fragments are the LDS term columns scaled
//               on the fly by the bilinear coefficient dt ubar_g / q of the column's interval, plus (dt/q) mult times the
//               source type's column where the type has an inhomogeneous term on generator g (they share the product).
//               term_{t+1} = acc/(t+1): into the sums (registers), into Zs (after a barrier), into global memory only in
//               store mode (the Hessian's pairing reads every term); column norms -> LDS, Al-Mohy--Higham test per workgroup.
//   two barriers per step; no global traffic but the generator stream in the Jacobian's sweep.
//
// Bound: FP64 MFMA.  Per step and workgroup 2 * npad * (16 NT) * (m+1) npad flops on (m+1) npad^2 * 8 bytes of generator
// reads from L2 (512/(16 NT) bytes per cycle and CU at full MFMA rate: 10.7 B/cycle at NT = 3).  Measured at 256 x 2000
// (tools/sweep_fused_probe.hip, alone on the chip): 3.3 ms for 25 terms = 50 TFLOP/s issued; without the two FP64 VALU
// operations per B fragment 8 % less (vector and matrix FP64 share one pipe), without the A loads the same.
#include "dto_gemm.hip.h"
#include "dto_kernels.h"

#include <type_traits>

namespace dto {

namespace {

__device__ __forceinline__ unsigned long long fbits(double v) { return (unsigned long long)__double_as_longlong(fabs(v)); }
__device__ __forceinline__ double fbits_to_d(unsigned long long b) { return __longlong_as_double((long long)b); }

struct FusedSweepArgs {
    KProb P;
    KBil B;
    SweepBuf w;
    SweepTypes ty;
    const double* G;     // generators used as the left operand (G or G')
    const double* Zsrc;  // the NLP vector
    const double* mu;    // multipliers (src_kind 1)
    int src_kind, q, d_ub, tc, ipw, store, nslot;
    double tol;
};

constexpr int PF = 4;  // k-steps (of 4) the A fragments are loaded ahead

// LDS carve-up shared by host and device
struct FusedLds {
    int zs, scr, cg, se, tn, sn, xn, xg, xs, xm, flag, total;  // offsets in doubles
    __host__ __device__ FusedLds(int npad, int T, int m, int ipw, int nslot, int MT) {
        const int NC = T * ipw, ZS = npad + 2;
        (void)nslot; (void)MT;
        int o = 0;
        zs = o; o += NC * ZS + 4;  // + 4: the B-fragment prefetch runs one k-step past the last column
        scr = o;
        cg = o; o += (m + 1) * ipw;
        se = o; o += ipw;
        tn = o; o += 3 * NC;
        sn = o; o += NC;
        xm = o; o += 2 * T;
        xn = o; o += (T + 1) / 2;          // ints, two per double
        xg = o; o += T;                    // 2 ints per type
        xs = o; o += T;
        flag = o; o += 2;
        total = o;
    }
};

// npad == 64 MT: one row pass of the four wavefronts covers the matrix.  The sums live in registers and the new term goes
// from the accumulators straight into the LDS columns; global memory sees the terms only in store mode and the sums once,
// at the end.
// WC column groups of wavefronts: wavefront (wr, wc) owns rows 16 MT wr .. and the column tiles NT wc .. NT (wc + 1) - 1; WC = 2
// puts two wavefronts on every SIMD, each with the k-loop of a two-tile workgroup.
template <int MT, int NT, int WC>
__global__ void __launch_bounds__(256 * WC, WC) k_sweep_fused(FusedSweepArgs a) {
    constexpr int NTHREADS = 256 * WC, NWAVES = 4 * WC;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int RL = 16 * MT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int wr = wave & 3, ct0 = (wave >> 2) * NT;  // row group, first column tile
    const int npad = a.w.npad, Kpad = a.w.Kpad, T = a.ty.T, m = a.B.m, ipw = a.ipw;
    const int NC = T * ipw, ZS = npad + 2, KS = npad / 4;
    const int64_t typesz = (int64_t)Kpad * npad, nn = (int64_t)npad * npad;
    const int k0 = blockIdx.x * ipw;
    const FusedLds L(npad, T, m, ipw, a.nslot, MT);
    double* Zs = lds + L.zs;
    double* cg = lds + L.cg;
    double* sE = lds + L.se;
    unsigned long long* tn = reinterpret_cast<unsigned long long*>(lds + L.tn);
    unsigned long long* sn = reinterpret_cast<unsigned long long*>(lds + L.sn);
    double* xm = lds + L.xm;
    int* xn = reinterpret_cast<int*>(lds + L.xn);
    int* xg = reinterpret_cast<int*>(lds + L.xg);
    int* xs = reinterpret_cast<int*>(lds + L.xs);
    int* flag = reinterpret_cast<int*>(lds + L.flag);

    // ---- per-interval coefficients, type table, term 0
    if (tid < ipw) {
        const int kl = k0 + tid;
        const bool live = kl < a.P.n_int;
        const double* zk = a.Zsrc + (a.P.kn_lo + kl) * a.P.z;
        const double dt = live ? zk[a.P.dt_idx] : 0.0;
        const double inv_q = 1.0 / a.q;
        sE[tid] = dt * inv_q;
        if (kl < Kpad) {
            a.w.scaleE[kl] = dt * inv_q;
            a.w.scaleE[Kpad + kl] = 2.0 * dt * inv_q;
        }
        for (int g = 0; g <= m; ++g) {
            const double ub = live ? (g == 0 ? 1.0 : zk[a.B.u_off + g - 1]) : 0.0;
            cg[g * ipw + tid] = dt * ub * inv_q;
            if (kl < Kpad) {
                a.w.scaleU[(int64_t)g * Kpad + kl] = ub;
                a.w.scaleA[(int64_t)g * Kpad + kl] = dt * ub * inv_q;
            }
        }
    }
    if (tid < T) {
        const TypeDesc td = a.ty.t[tid];
        xn[tid] = td.n_extra;
        xg[2 * tid] = td.gen[0]; xg[2 * tid + 1] = td.gen[1];
        xs[2 * tid] = td.src[0]; xs[2 * tid + 1] = td.src[1];
        xm[2 * tid] = td.mult[0]; xm[2 * tid + 1] = td.mult[1];
    }
    if (tid < 2) flag[tid] = 0;
    for (int c = tid; c < 3 * NC; c += NTHREADS) tn[c] = 0ull;
    for (int c = tid; c < NC; c += NTHREADS) sn[c] = 0ull;
    __syncthreads();
    {
        double* Z0 = a.store ? a.w.Zt : a.w.Z[0];
        const bool to_global = a.store != 0;  // without store the terms never leave the CU
        for (int c = wave; c < NC; c += NWAVES) {
            const int ty = c / ipw, i = c - ty * ipw, kl = k0 + i;
            const bool live = ty == 0 && kl < a.P.n_int;
            const int64_t kn = a.P.kn_lo + kl;
            double mx = 0.0;
            for (int r = lane; r < npad; r += 64) {
                double v = 0.0;
                if (live && r < a.B.n) v = a.src_kind == 0 ? a.Zsrc[kn * a.P.z + a.B.x_off + r] : a.mu[a.B.row_off + kn * a.B.n + r];
                Zs[c * ZS + r] = v;
                if (kl < Kpad) {
                    const int64_t off = ((int64_t)ty * Kpad + kl) * npad + r;
                    if (to_global) Z0[off] = v;
                }
                mx = fmax(mx, fabs(v));
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
            if (lane == 0) { tn[c] = fbits(mx); sn[c] = fbits(mx); }
        }
    }
    __syncthreads();

    // B-fragment column of this lane: lane (lr, lq) of column tile tj feeds column 16 tj + lr, rows 4 ks + lq of the term
    // panel (padding tiles re-read the last real column with coefficient 0; their accumulators are never stored)
    int bcol[NT], bty[NT], bin[NT];
    bool bok[NT];
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
        const int c = 16 * (ct0 + tj) + lr;
        bok[tj] = c < NC;
        const int cc = bok[tj] ? c : NC - 1;
        bty[tj] = cc / ipw;
        bin[tj] = cc - bty[tj] * ipw;
        bcol[tj] = cc * ZS + lq;
    }
    // The generators are streamed in a rotated order that differs between workgroups (start generator and start column
    // by block index): otherwise every CU of the chip asks the L2 for the same few lines at the same moment.  The order of
    // the K summation is a function of the block index alone, so results stay reproducible run to run.
    const int g_first = blockIdx.x % (m + 1);
    const int ks_first = ((blockIdx.x / (m + 1)) & 3) * (KS / 4);
    const int64_t astep = 4 * (int64_t)npad;
    const double* const aend = a.G + (int64_t)(m + 1) * nn;

    // sums of this lane's accumulator elements: element (ti, tj, r) is row rowbase + 32 (ti/2) + 2 lr + (ti & 1) [MT >= 2] of
    // column 16 tj + 4 r + lq
    const int rowbase = wr * RL;
    d4 sreg[MT][NT];
    auto lane_row = [&](int ti) { return MT >= 2 ? rowbase + 32 * (ti / 2) + 2 * lr + (ti & 1) : rowbase + lr; };
#pragma unroll
    for (int ti = 0; ti < MT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * (ct0 + tj) + 4 * r + lq;
                sreg[ti][tj][r] = c < NC ? Zs[c * ZS + lane_row(ti)] : 0.0;
            }

    int t_exit = 0;
    bool conv = false;
    for (int round = 0; round < a.q; ++round) {
        if (round > 0) {
            // next sub-interval of exp(A) = exp(A/q)^q: the sums become term 0 of the new series
#pragma unroll
            for (int ti = 0; ti < MT; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * (ct0 + tj) + 4 * r + lq;
                        if (c < NC) Zs[c * ZS + lane_row(ti)] = sreg[ti][tj][r];
                    }
            __syncthreads();
            for (int c = wave; c < NC; c += NWAVES) {
                double mx = 0.0;
                for (int r = lane; r < npad; r += 64) mx = fmax(mx, fabs(Zs[c * ZS + r]));
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
                if (lane == 0) { tn[c] = fbits(mx); tn[NC + c] = 0ull; tn[2 * NC + c] = 0ull; sn[c] = fbits(mx); }
            }
            if (tid < 2) flag[tid] = 0;
            __syncthreads();
        }
        conv = false;
        int t = 0;
        for (; t < a.d_ub; ++t) {
            double* Zout = a.store ? a.w.Zt + (int64_t)(t + 1) * T * typesz : a.w.Z[(t + 1) & 1];
            unsigned long long* tn_new = tn + ((t + 1) % 3) * NC;
            const double inv = 1.0 / (double)(t + 1);
            d4 acc[MT][NT];
            {
                // lane's A offset: column lq of a k-step, its rows in the accumulator's permutation (tile pair p holds the
                // even rows of a 32-row group in tile 2p and the odd rows in tile 2p+1: one 16-byte load feeds both).  The
                // k-steps of all generators form ONE circular stream (G_{g+1} follows G_g in memory): a uniform base
                // pointer that advances by 4 columns per k-step, plus this 32-bit lane offset; it is read PF k-steps ahead.
                const int aoff = lq * npad + rowbase + (MT >= 2 ? 2 * lr : lr);
#pragma unroll
                for (int ti = 0; ti < MT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) acc[ti][tj] = d4{0.0, 0.0, 0.0, 0.0};
                constexpr int AP = MT >= 2 ? MT / 2 : 1;
                d2 abuf[PF][AP];
                const double* abase = a.G + ((int64_t)g_first * KS + ks_first) * astep;
                auto issue = [&](int slot) {
                    const double* p = abase + aoff;
#pragma unroll
                    for (int q2 = 0; q2 < AP; ++q2) {
                        if constexpr (MT >= 2) abuf[slot][q2] = *reinterpret_cast<const d2*>(p + 32 * q2);
                        else abuf[slot][q2] = d2{p[0], 0.0};
                    }
                    abase += astep;
                    if (abase == aend) abase = a.G;
                };
#pragma unroll
                for (int u = 0; u < PF; ++u) issue(u);
                const int nseg = ks_first ? m + 2 : m + 1;
                for (int seg = 0; seg < nseg; ++seg) {
                    int g = g_first + seg;
                    if (g > m) g -= m + 1;
                    if (seg == m + 1) g = g_first;
                    const int ks_lo = seg == 0 ? ks_first : 0, ks_hi = seg == m + 1 ? ks_first : KS;
                    // B operand of generator g, per column: cA * term[type] + cB * term[source type] -- the bilinear
                    // coefficient dt ubar_g / q of the column's interval, and (dt/q) mult where the column's type has an
                    // inhomogeneous term on this generator: both products share the generator, so they are added here
                    // (cB = 0 and the column itself as "source" where there is none)
                    double cA[NT], cB[NT];
                    const double* zp1[NT];
                    const double* zp2[NT];
                    double z1[NT], z2[NT];
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        cA[tj] = bok[tj] ? cg[g * ipw + bin[tj]] : 0.0;
                        cB[tj] = 0.0;
                        int so = bcol[tj];
                        const int ne = bok[tj] ? xn[bty[tj]] : 0;
                        for (int x = 0; x < ne; ++x)
                            if (xg[2 * bty[tj] + x] == g) {
                                cB[tj] = sE[bin[tj]] * xm[2 * bty[tj] + x];
                                so = (xs[2 * bty[tj] + x] * ipw + bin[tj]) * ZS + lq;
                            }
                        zp1[tj] = Zs + bcol[tj] + 4 * ks_lo;
                        zp2[tj] = Zs + so + 4 * ks_lo;
                        z1[tj] = zp1[tj][0];
                        z2[tj] = zp2[tj][0];
                    }
                    for (int ks0 = ks_lo; ks0 < ks_hi; ks0 += PF) {
#pragma unroll
                        for (int u = 0; u < PF; ++u) {
                            double bf[NT];
#pragma unroll
                            for (int tj = 0; tj < NT; ++tj) {
#ifdef PROBE_NO_VALU  // (tools/sweep_fused_probe.hip: what the kernel costs without one of its parts)
                                bf[tj] = z1[tj];
#else
                                bf[tj] = cA[tj] * z1[tj] + cB[tj] * z2[tj];
#endif
                            }
                            // raw term values of the next k-step, at immediate offsets from the group's base (the last read
                            // of a segment runs 4 rows past the column: pitch padding / the next column, never used)
#ifndef PROBE_NO_BLOAD
#pragma unroll
                            for (int tj = 0; tj < NT; ++tj) {
                                z1[tj] = zp1[tj][4 * (u + 1)];
                                z2[tj] = zp2[tj][4 * (u + 1)];
                            }
#endif
                            __builtin_amdgcn_sched_barrier(0);  // the reads go out BEFORE the MFMA block, which hides them
#pragma unroll
                            for (int ti = 0; ti < MT; ++ti)
#pragma unroll
                                for (int tj = 0; tj < NT; ++tj) {
                                    double afv;
                                    if constexpr (MT >= 2) afv = (ti & 1) ? abuf[u][ti / 2].y : abuf[u][ti / 2].x;
                                    else afv = abuf[u][0].x;
                                    acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[tj], afv, acc[ti][tj], 0, 0, 0);
                                }
                            __builtin_amdgcn_sched_barrier(0);
#ifndef PROBE_NO_ALOAD
                            issue(u);  // refill this slot with the k-step PF ahead (the MFMAs above have read it)
#endif
                        }
#pragma unroll
                        for (int tj = 0; tj < NT; ++tj) { zp1[tj] += 4 * PF; zp2[tj] += 4 * PF; }
                    }
                }
                // ---- new term of this row pass: sums, column norms, stores.  Accumulator register r of column tile tj
                // holds column 16 tj + 4 r + lq
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * (ct0 + tj) + 4 * r + lq;
                        const int cc = c < NC ? c : 0;
                        const int ty = cc / ipw, kl = k0 + cc - ty * ipw;
                        const bool ok = c < NC && kl < Kpad;
                        const int64_t colbase = ((int64_t)ty * Kpad + kl) * npad + rowbase;
                        double tmax = 0.0, smax = 0.0;
                        bool bad_t = false, bad_s = false;
                        if (ok) {
#pragma unroll
                            for (int p = 0; p < (MT >= 2 ? MT / 2 : 1); ++p) {
                                const int64_t off = colbase + (MT >= 2 ? 32 * p + 2 * lr : lr);
                                d2 v, sv;
                                if constexpr (MT >= 2) v = d2{acc[2 * p][tj][r] * inv, acc[2 * p + 1][tj][r] * inv};
                                else v = d2{acc[0][tj][r] * inv, 0.0};
                                if constexpr (MT >= 2) {
                                    sv = d2{sreg[2 * p][tj][r] + v.x, sreg[2 * p + 1][tj][r] + v.y};
                                    sreg[2 * p][tj][r] = sv.x; sreg[2 * p + 1][tj][r] = sv.y;
                                } else {
                                    sv = d2{sreg[0][tj][r] + v.x, 0.0};
                                    sreg[0][tj][r] = sv.x;
                                }
                                if (a.store) {
                                    if constexpr (MT >= 2) *reinterpret_cast<d2*>(Zout + off) = v;
                                    else Zout[off] = v.x;
                                }
                                tmax = fmax(tmax, fmax(fabs(v.x), fabs(v.y)));
                                smax = fmax(smax, fmax(fabs(sv.x), fabs(sv.y)));
                                bad_t = bad_t || !(v.x == v.x) || !(v.y == v.y);      // NaN must survive the max
                                bad_s = bad_s || !(sv.x == sv.x) || !(sv.y == sv.y);
                            }
                        }
                        // compare bit patterns: a NaN's exceeds every finite one
                        unsigned long long tb = bad_t ? 0x7ff8000000000000ull : fbits(tmax), sb = bad_s ? 0x7ff8000000000000ull : fbits(smax);
#pragma unroll
                        for (int o = 8; o > 0; o >>= 1) {
                            const unsigned long long t2 = __shfl_xor(tb, o, 64), s2 = __shfl_xor(sb, o, 64);
                            tb = t2 > tb ? t2 : tb;
                            sb = s2 > sb ? s2 : sb;
                        }
                        if (ok && lr == 0) {
                            atomicMax(&tn_new[c], tb);
                            atomicMax(&sn[c], sb);
                        }
                    }
            }
            __syncthreads();  // every wave is done with the old term columns
            // Al-Mohy & Higham's test (as k_sweep_check): two successive terms below tol * |sum| in every column
            if (t >= a.tc) {
                for (int c = tid; c < NC; c += NTHREADS) {
                    const double a0 = fbits_to_d(tn[(t % 3) * NC + c]), a1 = fbits_to_d(tn_new[c]), s = fbits_to_d(sn[c]);
                    if (!(a0 + a1 <= a.tol * s) && (a0 + a1 == a0 + a1) && s < 1e300) flag[t & 1] = 1;
                }
            } else if (tid == 0) {
                flag[t & 1] = 1;
            }
            for (int c = tid; c < NC; c += NTHREADS) tn[((t + 2) % 3) * NC + c] = 0ull;
            if (tid == 0) flag[(t + 1) & 1] = 0;
            // the new term becomes the B operand of the next step
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * (ct0 + tj) + 4 * r + lq;
                    if (c < NC) {
                        if constexpr (MT >= 2) {
#pragma unroll
                            for (int p = 0; p < MT / 2; ++p)
                                *reinterpret_cast<d2*>(Zs + c * ZS + rowbase + 32 * p + 2 * lr) =
                                    d2{acc[2 * p][tj][r] * inv, acc[2 * p + 1][tj][r] * inv};
                        } else {
                            Zs[c * ZS + rowbase + lr] = acc[0][tj][r] * inv;
                        }
                    }
                }
            __syncthreads();
            if (flag[t & 1] == 0) { conv = true; break; }
        }
        t_exit = conv ? t + 2 : a.d_ub + 1;  // terms 0 .. t+1 exist
        if (round + 1 < a.q) __syncthreads();
    }
    {
        // the sums leave the registers once
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * (ct0 + tj) + 4 * r + lq;
                const int cc = c < NC ? c : 0;
                const int ty = cc / ipw, kl = k0 + cc - ty * ipw;
                if (c < NC && kl < Kpad) {
                    const int64_t colbase = ((int64_t)ty * Kpad + kl) * npad + rowbase;
                    if constexpr (MT >= 2) {
#pragma unroll
                        for (int p = 0; p < MT / 2; ++p)
                            *reinterpret_cast<d2*>(a.w.S + colbase + 32 * p + 2 * lr) = d2{sreg[2 * p][tj][r], sreg[2 * p + 1][tj][r]};
                    } else {
                        a.w.S[colbase + lr] = sreg[0][tj][r];
                    }
                }
            }
    }
    if (tid == 0) {
        if (a.w.nterms) a.w.nterms[blockIdx.x] = conv ? t_exit : 0;
        atomicMax(&a.w.stats[1], t_exit);
        atomicAdd(&a.w.stats[2], t_exit);  // diagnostics: sum over the workgroups (mean = / gridDim.x)
        if (!conv) atomicAdd(&a.w.stats[0], 1);
    }
}

template <int MT, int NT, int WC = 1>
hipError_t launch_one(hipStream_t st, const FusedSweepArgs& a, int nblocks, size_t lds) {
    if (a.w.npad != 64 * MT) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_sweep_fused<MT, NT, WC>), dim3(nblocks), dim3(256 * WC), lds, st, a);
    return hipGetLastError();
}
template <int MT, int NT, int WC = 1>
hipError_t prepare_one(int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep_fused<MT, NT, WC>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

hipError_t sweep_fused_prepare() {
    const int bytes = 160 * 1024;
    hipError_t e = hipSuccess;
#define DTO_PREP(MT, NT) if (e == hipSuccess) e = prepare_one<MT, NT>(bytes)
    DTO_PREP(4, 1); DTO_PREP(4, 2); DTO_PREP(4, 3);
    DTO_PREP(2, 1); DTO_PREP(2, 2); DTO_PREP(2, 3);
    DTO_PREP(1, 1); DTO_PREP(1, 2); DTO_PREP(1, 3);
#undef DTO_PREP
    if (e == hipSuccess) e = prepare_one<4, 2, 2>(bytes);
    return e;
}

// Shape of the launch for a sweep over T column types of an integrator padded to npad states, n_int intervals:
// intervals per workgroup (ipw) and tile counts.  Returns false when the fused form does not apply.
bool sweep_fused_plan(int npad, int m, const SweepTypes& ty, int64_t n_int, int n_cu, FusedSweepPlan& out, bool shared_chip) {
    const int T = ty.T;
    if (T < 1 || n_int <= 0) return false;
    // One row pass must cover the matrix (npad = 64, 128 or 256: 4 wavefronts x 16 MT rows).  Larger matrices offer their
    // parallelism in the ROW dimension, which a workgroup-per-interval-group form cannot use without exchanging the term
    // columns between workgroups every step: there the step-per-launch sweep is faster (measured: 512 states 19.0 against
    // 22.6 ms, 1024 states 109 against 128 ms per Jacobian), as it is when the intervals are too few to give half the CUs a
    // workgroup (256 states x 200 knots: 2.25 against 2.50 ms).
    if (npad != 64 && npad != 128 && npad != 256) return false;
    const int MT = npad / 64;
    int nslot = 0;
    for (int g = 0; g <= m; ++g) {
        int mask = 0;
        for (int t = 0; t < T; ++t)
            for (int x = 0; x < ty.t[t].n_extra; ++x)
                if (ty.t[t].gen[x] == g) mask |= 1 << ty.t[t].src[x];
        nslot = nslot > __builtin_popcount(mask) ? nslot : __builtin_popcount(mask);
    }
    // cost model, 256 states: one workgroup per CU and round; a round takes NT units of MFMA time, and a narrow column tile
    // streams the generators from L2 at 512 / (16 NT) bytes per cycle and CU, which at NT = 1 is more than a CU sustains.
    // 64 and 128 states: the steps are latency-bound (two barriers and an epilogue per handful of MFMAs), several workgroups
    // share a CU, and one column tile per workgroup is fastest -- measured per sweep at 1000 knots, NT = 1 / 2 / 3:
    // 0.24 / 0.30 / 0.44 ms (64 states), 0.61 / 0.73 / 0.86 ms (128 states); three workgroups per CU count as one round.
    static const double l2_factor[4] = {0.0, 1.6, 1.15, 1.0};
    static const double t_small[2][4] = {{0.0, 0.24, 0.30, 0.44}, {0.0, 0.61, 0.73, 0.86}};
    static const int ipw_env = tune_int("DTO_SWEEP_IPW", 0);  // A/B runs (TUNING builds)
    auto search = [&](int WC) {
        bool found = false;
        double best = 0.0;
        for (int ipw = 1; ipw <= 48; ++ipw) {
            const int NC = T * ipw;
            int NT = (NC + 15) / 16;
            if (WC == 2) {
                if (NT > 4) break;
                if (NT != 4) continue;  // two column groups of two tiles each
                NT = 2;
            } else if (NT > 3) break;
            if (ipw_env > 0 && ipw != ipw_env && T == 1 + m) continue;
            const FusedLds L(npad, T, m, ipw, nslot, MT);
            const size_t bytes = (size_t)L.total * sizeof(double);
            if (bytes > 156 * 1024) break;
            const long nblocks = (long)((n_int + ipw - 1) / ipw);
            const long slots = MT <= 2 ? 3L * n_cu : n_cu;
            const long rounds = (nblocks + slots - 1) / slots;
            // ties go to the fewer workgroups (less MFMA work issued in total)
            const double cost = (MT <= 2 ? (double)rounds * t_small[MT - 1][NT] : (double)rounds * NT * l2_factor[NT]) + 1e-6 * (double)nblocks * NT;
            if (!found || cost < best) {
                found = true; best = cost;
                out.MT = MT; out.NT = NT; out.WC = WC; out.ipw = ipw; out.nslot = nslot; out.lds_bytes = bytes; out.nblocks = (int)nblocks;
            }
        }
        return found && 2 * out.nblocks >= n_cu;
    };
    // Eight wavefronts in two column groups (256 states, 49..64 columns: 12 intervals of a Jacobian sweep per workgroup): two
    // wavefronts per SIMD hide each other's operand traffic, +10 % MFMA rate per CU -- but a third fewer workgroups, each a
    // fifth longer (256 x 2000: 167 workgroups, 4.0 ms against 223, 3.3 ms).  It pays when the CUs the sweep leaves free are
    // used by another stream (`shared_chip`: the Jacobian's sweep next to the propagator chain), not when the sweep runs alone.
    if (shared_chip && npad == 256 && search(2)) return true;
    return search(1);
}

hipError_t launch_sweep_fused(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, const SweepTypes& ty,
                              const FusedSweepPlan& pl, const double* dZ, const double* dmu, int src_kind, int transposed,
                              int q, int d_ub, int tc, bool store, double tol) {
    FusedSweepArgs a{};
    a.P = P; a.B = B; a.w = w; a.ty = ty;
    a.G = transposed ? B.GT : B.G;
    a.Zsrc = dZ; a.mu = dmu; a.src_kind = src_kind;
    a.q = q; a.d_ub = d_ub; a.tc = tc; a.ipw = pl.ipw; a.store = store ? 1 : 0; a.nslot = pl.nslot; a.tol = tol;
    if (pl.WC == 2 && pl.MT == 4 && pl.NT == 2) return launch_one<4, 2, 2>(st, a, pl.nblocks, pl.lds_bytes);
    const int key = pl.MT * 10 + pl.NT;
    switch (key) {
        case 41: return launch_one<4, 1>(st, a, pl.nblocks, pl.lds_bytes);
        case 42: return launch_one<4, 2>(st, a, pl.nblocks, pl.lds_bytes);
        case 43: return launch_one<4, 3>(st, a, pl.nblocks, pl.lds_bytes);
        case 21: return launch_one<2, 1>(st, a, pl.nblocks, pl.lds_bytes);
        case 22: return launch_one<2, 2>(st, a, pl.nblocks, pl.lds_bytes);
        case 23: return launch_one<2, 3>(st, a, pl.nblocks, pl.lds_bytes);
        case 11: return launch_one<1, 1>(st, a, pl.nblocks, pl.lds_bytes);
        case 12: return launch_one<1, 2>(st, a, pl.nblocks, pl.lds_bytes);
        case 13: return launch_one<1, 3>(st, a, pl.nblocks, pl.lds_bytes);
    }
    return hipErrorInvalidValue;
}

}  // namespace dto
