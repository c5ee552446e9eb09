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
    const int32_t* plan_dev;   // k_sweep_s64 only, nullable: {q, d_ub, tc} planned ON THE DEVICE from the norm bound (k_plan_dev) -- the
                               // host then never waits for the bound
};

constexpr int PF = 4;  // k-steps (of 4) the A fragments are loaded ahead

// LDS carve-up shared by host and device
struct FusedLds {
    int zs, scr, cg, se, tn, sn, xn, xg, xs, xm, flag, total;  // offsets in doubles
    __host__ __device__ FusedLds(int npad, int T, int m, int ipw, int nslot, int MT) {
        const int NC = T * ipw, ZS = npad + 2;
        (void)nslot; (void)MT;
        int o = 0;
        zs = o; o += NC * ZS + 40; // the B-fragment prefetch runs up to two groups of four k-steps past the last column (K split)
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
// WK = 2 (256 states): TWO wavefronts per SIMD share a wave tile and split its K loop -- groups of PF k-steps alternate between
// them, each wave with its own accumulators; at the end of a term each wave hands the OTHER wave the partial sums of the row pair
// that wave owns (through the LDS term columns, which are being replaced anyway) and finishes its own pair: sums, norms, the new
// term.  One wave per SIMD cannot keep the matrix pipe busy through its own LDS and L2 latencies (MFMA utilisation 0.64); two
// that depend on nothing of each other's can.  The per-MFMA scaling arithmetic and the generator stream are NOT duplicated (the
// k-steps are disjoint), which is what the row- and column-split two-wave forms paid for.
template <int MT, int NT, int WC, int WK = 1>
__global__ void __launch_bounds__(256 * WC * WK, WC * WK) k_sweep_fused(FusedSweepArgs a) {
    static_assert(WK == 1 || (WK == 2 && MT == 4 && WC == 1), "the K split is built for 256 states");
    constexpr int NTHREADS = 256 * WC * WK, NWAVES = 4 * WC * WK;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int RL = 16 * MT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int wr = wave & 3, ct0 = ((wave >> 2) % WC) * NT;  // row group, first column tile
    const int wk = (wave >> 2) / WC;                           // K half (WK = 2)
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
    // WK = 2: a wave keeps the sums of ONE row pair (tiles 2 wk, 2 wk + 1 -> sreg[0], sreg[1]); own_row = first row of that pair
    constexpr int SR = WK == 2 ? 2 : MT;
    d4 sreg[SR][NT];
    const int own_row = rowbase + 32 * wk + 2 * lr;
    auto lane_row = [&](int ti) { return MT >= 2 ? rowbase + 32 * (ti / 2) + 2 * lr + (ti & 1) : rowbase + lr; };
#pragma unroll
    for (int ti = 0; ti < SR; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * (ct0 + tj) + 4 * r + lq;
                sreg[ti][tj][r] = c < NC ? Zs[c * ZS + (WK == 2 ? own_row + ti : lane_row(ti))] : 0.0;
            }

    int t_exit = 0;
    bool conv = false;
    for (int round = 0; round < a.q; ++round) {
        if (round > 0) {
            // next sub-interval of exp(A) = exp(A/q)^q: the sums become term 0 of the new series
#pragma unroll
            for (int ti = 0; ti < SR; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * (ct0 + tj) + 4 * r + lq;
                        if (c < NC) Zs[c * ZS + (WK == 2 ? own_row + ti : lane_row(ti))] = sreg[ti][tj][r];
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
                // (WK = 2: this wave's groups of PF k-steps are every other one of the stream, starting with group wk)
                const double* abase = a.G + ((int64_t)g_first * KS + ks_first + (WK == 2 ? wk * PF : 0)) * astep;
                if (WK == 2 && abase >= aend) abase -= (int64_t)(m + 1) * nn;
                int in_group = 0;
                auto issue = [&](int slot) {
                    const double* p = abase + aoff;
#pragma unroll
                    for (int q2 = 0; q2 < AP; ++q2) {
                        if constexpr (MT >= 2) abuf[slot][q2] = *reinterpret_cast<const d2*>(p + 32 * q2);
                        else abuf[slot][q2] = d2{p[0], 0.0};
                    }
                    abase += astep;
                    if constexpr (WK == 2) {
                        if (++in_group == PF) { in_group = 0; abase += PF * astep; }   // the other wave's group
                        if (abase >= aend) abase -= (int64_t)(m + 1) * nn;
                    } else {
                        if (abase == aend) abase = a.G;
                    }
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
                        zp1[tj] = Zs + bcol[tj] + 4 * (ks_lo + (WK == 2 ? wk * PF : 0));
                        zp2[tj] = Zs + so + 4 * (ks_lo + (WK == 2 ? wk * PF : 0));
                        z1[tj] = zp1[tj][0];
                        z2[tj] = zp2[tj][0];
                    }
                    // (every segment holds an even number of groups: KS / PF and ks_first / PF are multiples of four)
                    for (int ks0 = ks_lo + (WK == 2 ? wk * PF : 0); ks0 < ks_hi; ks0 += PF * WK) {
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
                                z1[tj] = zp1[tj][4 * (u + 1 == PF ? PF * WK : u + 1)];   // (the wave's next group)
                                z2[tj] = zp2[tj][4 * (u + 1 == PF ? PF * WK : u + 1)];
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
                        for (int tj = 0; tj < NT; ++tj) { zp1[tj] += 4 * PF * WK; zp2[tj] += 4 * PF * WK; }
                    }
                }
                if constexpr (WK == 2) {
                    // ---- K split: hand the other wave of the SIMD the partial sums of ITS row pair, finish the own pair
                    __syncthreads();  // every wave is done with the old term columns
                    // (the lane coordinates are made opaque once per term: otherwise the per-element column indices, interval numbers and
                    // store offsets of this epilogue -- 12 elements, a division each -- are computed ONCE before the term loop and kept
                    // alive across it in scratch: ~60 scratch loads per term and wave; recomputing them is integer arithmetic)
                    int lq_e = lq, lr_e = lr;
                    if constexpr (MT == 4) asm volatile("" : "+v"(lq_e), "+v"(lr_e));   // (below 256 states nothing spills and the divisions would be the loss)
                    auto finish = [&](auto own_c) {
                        constexpr int OWN = decltype(own_c)::value, OTHER = 1 - OWN;
#pragma unroll
                        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int c = 16 * (ct0 + tj) + 4 * r + lq_e;
                                if (c < NC)
                                    *reinterpret_cast<d2*>(Zs + c * ZS + rowbase + 32 * OTHER + 2 * lr_e) =
                                        d2{acc[2 * OTHER][tj][r], acc[2 * OTHER + 1][tj][r]};
                            }
                        __syncthreads();
#pragma unroll
                        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int c = 16 * (ct0 + tj) + 4 * r + lq_e;
                                const int cc = c < NC ? c : 0;
                                const int ty = cc / ipw, kl = k0 + cc - ty * ipw;
                                const bool ok = c < NC && kl < Kpad;
                                double tmax = 0.0, smax = 0.0;
                                bool bad_t = false, bad_s = false;
                                if (c < NC) {
                                    d2* zpos = reinterpret_cast<d2*>(Zs + c * ZS + rowbase + 32 * OWN + 2 * lr_e);
                                    const d2 part = *zpos;   // the other wave's k-steps
                                    const d2 v = d2{(acc[2 * OWN][tj][r] + part.x) * inv, (acc[2 * OWN + 1][tj][r] + part.y) * inv};
                                    *zpos = v;               // the new term becomes the B operand of the next step
                                    if (ok) {
                                        const d2 sv = d2{sreg[0][tj][r] + v.x, sreg[1][tj][r] + v.y};
                                        sreg[0][tj][r] = sv.x; sreg[1][tj][r] = sv.y;
                                        if (a.store) *reinterpret_cast<d2*>(Zout + ((int64_t)ty * Kpad + kl) * npad + rowbase + 32 * OWN + 2 * lr_e) = v;
                                        tmax = fmax(fabs(v.x), fabs(v.y));
                                        smax = fmax(fabs(sv.x), fabs(sv.y));
                                        bad_t = !(v.x == v.x) || !(v.y == v.y);      // NaN must survive the max
                                        bad_s = !(sv.x == sv.x) || !(sv.y == sv.y);
                                    }
                                }
                                unsigned long long tb = bad_t ? 0x7ff8000000000000ull : fbits(tmax), sb = bad_s ? 0x7ff8000000000000ull : fbits(smax);
#pragma unroll
                                for (int o = 8; o > 0; o >>= 1) {
                                    const unsigned long long t2 = __shfl_xor(tb, o, 64), s2 = __shfl_xor(sb, o, 64);
                                    tb = t2 > tb ? t2 : tb;
                                    sb = s2 > sb ? s2 : sb;
                                }
                                if (ok && lr_e == 0) {
                                    atomicMax(&tn_new[c], tb);
                                    atomicMax(&sn[c], sb);
                                }
                            }
                    };
                    if (wk == 0) finish(std::integral_constant<int, 0>{});
                    else finish(std::integral_constant<int, 1>{});
                } else {
                    int lq_e = lq, lr_e = lr;   // (opaque once per term, as in the K-split epilogue above)
                    if constexpr (MT == 4) asm volatile("" : "+v"(lq_e), "+v"(lr_e));
                    // ---- new term of this row pass: sums, column norms, stores.  Accumulator register r of column tile tj
                    // holds column 16 tj + 4 r + lq_e
    #pragma unroll
                    for (int tj = 0; tj < NT; ++tj)
    #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = 16 * (ct0 + tj) + 4 * r + lq_e;
                            const int cc = c < NC ? c : 0;
                            const int ty = cc / ipw, kl = k0 + cc - ty * ipw;
                            const bool ok = c < NC && kl < Kpad;
                            const int64_t colbase = ((int64_t)ty * Kpad + kl) * npad + rowbase;
                            double tmax = 0.0, smax = 0.0;
                            bool bad_t = false, bad_s = false;
                            if (ok) {
    #pragma unroll
                                for (int p = 0; p < (MT >= 2 ? MT / 2 : 1); ++p) {
                                    const int64_t off = colbase + (MT >= 2 ? 32 * p + 2 * lr_e : lr_e);
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
                            if (ok && lr_e == 0) {
                                atomicMax(&tn_new[c], tb);
                                atomicMax(&sn[c], sb);
                            }
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
            if constexpr (WK == 1) {
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
                    if constexpr (WK == 2) {
                        *reinterpret_cast<d2*>(a.w.S + colbase + 32 * wk + 2 * lr) = d2{sreg[0][tj][r], sreg[1][tj][r]};
                    } else if constexpr (MT >= 2) {
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

// ------------------------------------------------------------------------------------------------------------------
// 64 states (round 4): generators STATIONARY in registers, the K dimension split over the four wavefronts.
//
// k_sweep_fused<1, 1> gives a wavefront 16 rows of ONE 16-column tile: one MFMA per k-step, each behind its own generator
// fragment from L2 and two FP64 vector operations scaling the B fragment (vector and matrix FP64 share a pipe: ~140 cycles
// per k-step for a 64-cycle MFMA), 80 such steps and two barriers per Taylor term -- 0.24 ms per 1000-knot Jacobian sweep,
// four times its MFMA time.  Here wavefront w owns k = 16 w .. 16 w + 15 of EVERY generator for ALL 64 rows:
//   A fragments  (m+1) x 4 k-steps x 4 row tiles = 80 doubles per lane, loaded once per workgroup;
//   B fragments  rows 16 w .. 16 w + 15 of the 16 term columns -- rows this very wavefront produced the step before (see the
//                reduction), so the term columns in LDS need no barrier; 4 (+4 per inhomogeneous source) LDS reads and
//                20 (+20) vector operations per term instead of 160 and 160;
//   reduction    every wavefront leaves its partial 64 x 16 product in LDS (double-buffered by term parity), ONE barrier, then
//                wavefront w adds the four partials of row tile w in wavefront order (fixed: results are a function of the data
//                alone): new term = sum / (t+1) -> sums (registers), column norms (LDS atomics), its rows of the term columns.
//   termination  the norms of term t+1 are complete only behind the NEXT barrier, so the Al-Mohy--Higham test of the pair
//                (t, t+1) runs there -- after the partial products of term t+2 were formed, before anything of them is used:
//                same sums, same term count as the other forms, one wasted product per workgroup.
// Bound: FP64 MFMA, 80 per wavefront and term (5120 cycles) + ~1000 cycles of everything else.
struct S64Lds {
    static constexpr int ZS = 66;   // column pitch: the B-fragment reads of a half-wave fall on 32 distinct 8-byte banks
    int zs, pb, cg, se, tn, sn, xm, xn, xg, xs, total;  // offsets in doubles
    __host__ __device__ S64Lds(int T, int m, int ipw) {
        int o = 0;
        zs = o; o += 16 * ZS;
        pb = o; o += 2 * 4 * 4 * 2 * 64 * 2;   // [term parity][row tile][source wavefront][half][lane] 16 bytes
        cg = o; o += (m + 1) * ipw;
        se = o; o += ipw;
        tn = o; o += 4 * 16;
        sn = o; o += 4 * 16;
        xm = o; o += 2 * T;
        xn = o; o += (T + 1) / 2;   // ints, two per double
        xg = o; o += T;
        xs = o; o += T;
        total = o;
    }
};

// Column norms travel as the HIGH WORD of |v| (non-negative doubles order like their bit patterns, a NaN's exceeds every
// finite one): the norm is rounded down by at most 2^-20 of itself, which the termination test does not notice, and the
// maximum over the 16 lanes of a DPP row takes four 32-bit DPP operations instead of twelve instructions on doubles.
__device__ __forceinline__ unsigned s64_hi(double v) { return (unsigned)((unsigned long long)__double_as_longlong(fabs(v)) >> 32); }
template <int CTRL>
__device__ __forceinline__ unsigned s64_dpp_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned s64_row16_max_u32(unsigned v) {
    v = max(v, s64_dpp_u32<0x128>(v));   // row_ror:8
    v = max(v, s64_dpp_u32<0x124>(v));   // row_ror:4
    v = max(v, s64_dpp_u32<0x122>(v));   // row_ror:2
    v = max(v, s64_dpp_u32<0x121>(v));   // row_ror:1
    return v;
}

template <int MP, int NX>
__global__ void __launch_bounds__(256, 2) k_sweep_s64(FusedSweepArgs a) {
    constexpr int ZS = S64Lds::ZS;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int Kpad = a.w.Kpad, T = a.ty.T, m = a.B.m, ipw = a.ipw, NC = T * ipw;
    const int64_t typesz = (int64_t)Kpad * 64;
    const int k0 = blockIdx.x * ipw;
    const int plan_q = a.plan_dev ? a.plan_dev[0] : a.q, plan_dub = a.plan_dev ? a.plan_dev[1] : a.d_ub, plan_tc = a.plan_dev ? a.plan_dev[2] : a.tc;
    const S64Lds L(T, m, ipw);
    double* Zs = lds + L.zs;
    d2* Pb = reinterpret_cast<d2*>(lds + L.pb);
    double* cg = lds + L.cg;
    double* sE = lds + L.se;
    unsigned long long* tn = reinterpret_cast<unsigned long long*>(lds + L.tn);
    unsigned long long* sn = reinterpret_cast<unsigned long long*>(lds + L.sn);
    double* xm = lds + L.xm;
    int* xn = reinterpret_cast<int*>(lds + L.xn);
    int* xg = reinterpret_cast<int*>(lds + L.xg);
    int* xs = reinterpret_cast<int*>(lds + L.xs);

    // ---- per-interval coefficients, type table, term 0 (as k_sweep_fused)
    if (tid < ipw) {
        const int kl = k0 + tid;
        const bool live = kl < a.P.n_int;
        const double* zk = a.Zsrc + (a.P.kn_lo + kl) * a.P.z;
        const double dt = live ? zk[a.P.dt_idx] : 0.0;
        const double inv_q = 1.0 / plan_q;
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
    if (tid < 64) { tn[tid] = 0ull; sn[tid] = 0ull; }
    __syncthreads();
    {
        double* Z0 = a.store ? a.w.Zt : a.w.Z[0];
        for (int c = w; c < 16; c += 4) {
            const int cc = c < NC ? c : 0;
            const int ty = cc / ipw, i = cc - ty * ipw, kl = k0 + i;
            const bool live = c < NC && ty == 0 && kl < a.P.n_int;
            const int64_t kn = a.P.kn_lo + kl;
            double v = 0.0;
            if (live && lane < a.B.n) v = a.src_kind == 0 ? a.Zsrc[kn * a.P.z + a.B.x_off + lane] : a.mu[a.B.row_off + kn * a.B.n + lane];
            Zs[c * ZS + lane] = v;
            if (a.store && c < NC && kl < Kpad) Z0[((int64_t)ty * Kpad + kl) * 64 + lane] = v;
            double mx = fabs(v);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
            if (lane == 0) { tn[c] = fbits(mx); sn[c] = fbits(mx); }
        }
    }
    __syncthreads();

    // ---- B role: lane (lr, lq) feeds column lr, rows 16 w + 4 ks + lq of the term panel
    const bool bok = lr < NC;
    const int bty = bok ? lr / ipw : 0, bin = bok ? lr - bty * ipw : 0;
    // (the per-generator coefficients stay in LDS and are fetched where they are used: with 160 registers of stationary operand a
    // second workgroup per CU -- which hides this one's barriers and reductions -- fits only if everything else stays under 96)
    const int zoff1 = lr * ZS + 16 * w + lq;
    int zoff2[NX > 0 ? NX : 1], gx[NX > 0 ? NX : 1];
    double cb[NX > 0 ? NX : 1];
#pragma unroll
    for (int x = 0; x < NX; ++x) {
        const bool has = bok && x < xn[bty];
        gx[x] = has ? xg[2 * bty + x] : -1;
        cb[x] = has ? sE[bin] * xm[2 * bty + x] : 0.0;
        zoff2[x] = (has ? xs[2 * bty + x] * ipw + bin : lr) * ZS + 16 * w + lq;
    }
    // ---- A role: G_g[row 16 ti + lr][k = 16 w + 4 ks + lq], column-major generators
    // (generator slots beyond m hold zeros: no branch inside the MFMA stream)
    double af[MP][4][4];
#pragma unroll
    for (int g = 0; g < MP; ++g) {
        const double* Gg = a.G + (int64_t)(g <= m ? g : 0) * 4096 + (16 * w + lq) * 64 + lr;
        const double keep = g <= m ? 1.0 : 0.0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) af[g][ks][ti] = keep * Gg[ks * 256 + 16 * ti];
    }
    // ---- C role: element r of this lane is row 16 w + lr of column 4 r + lq
    d4 sreg;
    unsigned srun[4];
    bool ok[4];
    int gcol[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = 4 * r + lq, cc = c < NC ? c : 0;
        const int ty = cc / ipw, kl = k0 + cc - ty * ipw;
        ok[r] = c < NC && kl < Kpad;
        gcol[r] = (ty * Kpad + kl) * 64 + 16 * w + lr;
        sreg[r] = Zs[c * ZS + 16 * w + lr];
        srun[r] = s64_hi(sreg[r]);
    }

    int t_exit = 0;
    bool conv = false;
    for (int round = 0; round < plan_q; ++round) {
        if (round > 0) {
            // next sub-interval of exp(A) = exp(A/q)^q: the sums become term 0 of the new series
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Zs[(4 * r + lq) * ZS + 16 * w + lr] = sreg[r];
                srun[r] = s64_hi(sreg[r]);
            }
            if (tid < 64) { tn[tid] = 0ull; sn[tid] = 0ull; }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned long long mx = (unsigned long long)s64_row16_max_u32(srun[r]) << 32;
                if (lr == 0 && 4 * r + lq < NC) { atomicMax(&tn[4 * r + lq], mx); atomicMax(&sn[4 * r + lq], mx); }
            }
            __syncthreads();
        }
        conv = false;
        int t = 0;
        for (;; ++t) {
            if (t < plan_dub) {
                // ---- partial products of term t+1 over this wavefront's k range
                double z1[4], z2[NX > 0 ? NX : 1][4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    z1[ks] = Zs[zoff1 + 4 * ks];
#pragma unroll
                    for (int x = 0; x < NX; ++x) z2[x][ks] = Zs[zoff2[x] + 4 * ks];
                }
                d4 acc[4];
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) acc[ti] = d4{0.0, 0.0, 0.0, 0.0};
                // the B fragments of two generators at a time, in front of their 32 MFMAs: vector and matrix FP64 share a pipe, an
                // FMA between two MFMAs costs a bubble (per switch, not per operation)
#pragma unroll
                for (int g0 = 0; g0 < MP; g0 += 2) {
                    double bf[2][4];
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int g = g0 + gg;
                        if (g < MP) {
                            const double cA = bok && g <= m ? cg[g * ipw + bin] : 0.0;
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks) {
                                bf[gg][ks] = cA * z1[ks];
#pragma unroll
                                for (int x = 0; x < NX; ++x) bf[gg][ks] = fma(gx[x] == g ? cb[x] : 0.0, z2[x][ks], bf[gg][ks]);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int g = g0 + gg;
                        if (g < MP) {
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                                for (int ti = 0; ti < 4; ++ti)
                                    acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[gg][ks], af[g][ks][ti], acc[ti], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                d2* pw = Pb + (size_t)((t & 1) * 16 + w) * 128 + lane;   // [parity][tile][wave][half][lane]
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    pw[ti * 512] = d2{acc[ti][0], acc[ti][1]};
                    pw[ti * 512 + 64] = d2{acc[ti][2], acc[ti][3]};
                }
            }
            __syncthreads();
            // ---- Al-Mohy & Higham's test of the pair (t-1, t), whose norms every wavefront has delivered before this barrier
            if (t >= 1 && t - 1 >= plan_tc) {
                bool more = false;
                if (lane < NC) {
                    const double a0 = fbits_to_d(tn[((t - 1) & 3) * 16 + lane]), a1 = fbits_to_d(tn[(t & 3) * 16 + lane]);
                    const double s = fbits_to_d(sn[(t & 3) * 16 + lane]);
                    more = !(a0 + a1 <= a.tol * s) && (a0 + a1 == a0 + a1) && s < 1e300;
                }
                if (__ballot(more) == 0ull) { conv = true; break; }
            }
            if (t == plan_dub) break;
            // ---- term t+1 of row tile w: the four partials in wavefront order
            const double inv = 1.0 / (double)(t + 1);
            d4 v;
            {
                const d2* pr = Pb + (size_t)((t & 1) * 16 + w * 4) * 128 + lane;
                d2 s0 = pr[0], s1 = pr[64];
#pragma unroll
                for (int sw = 1; sw < 4; ++sw) {
                    const d2 p0 = pr[sw * 128], p1 = pr[sw * 128 + 64];
                    s0 = d2{s0.x + p0.x, s0.y + p0.y};
                    s1 = d2{s1.x + p1.x, s1.y + p1.y};
                }
                v = d4{s0.x * inv, s0.y * inv, s1.x * inv, s1.y * inv};
            }
            double* Zout = a.store ? a.w.Zt + (int64_t)(t + 1) * T * typesz : nullptr;
            const int slot = ((t + 1) & 3) * 16;
            unsigned tmax[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sreg[r] += v[r];
                Zs[(4 * r + lq) * ZS + 16 * w + lr] = v[r];   // this wavefront's rows: the B operand of ITS next k range
                if (a.store && ok[r]) Zout[gcol[r]] = v[r];
                tmax[r] = ok[r] ? s64_hi(v[r]) : 0u;
                srun[r] = max(srun[r], ok[r] ? s64_hi(sreg[r]) : 0u);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned tm = s64_row16_max_u32(tmax[r]), sm = s64_row16_max_u32(srun[r]);
                if (lr == 0 && ok[r]) {
                    atomicMax(&tn[slot + 4 * r + lq], (unsigned long long)tm << 32);
                    atomicMax(&sn[slot + 4 * r + lq], (unsigned long long)sm << 32);
                }
            }
            if (w == 0 && lane < 16) { tn[((t + 2) & 3) * 16 + lane] = 0ull; sn[((t + 2) & 3) * 16 + lane] = 0ull; }
        }
        t_exit = conv ? t + 1 : plan_dub + 1;  // terms 0 .. t_exit - 1 exist
        if (round + 1 < plan_q) __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (ok[r]) a.w.S[gcol[r]] = sreg[r];
    if (tid == 0) {
        if (a.w.nterms) a.w.nterms[blockIdx.x] = conv ? t_exit : 0;
        atomicMax(&a.w.stats[1], t_exit);
        atomicAdd(&a.w.stats[2], t_exit);
        if (!conv) atomicAdd(&a.w.stats[0], 1);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same sweep with the ROWS of the matrix split over a cluster of R workgroups (round 3).
//
// k_sweep_fused gives one workgroup all npad rows of a few intervals; it needs npad <= 256 and at least half a workgroup
// per CU, i.e. >= 9 x 128 intervals for a Jacobian sweep at 256 states.  A 250-knot shard (strong scaling over 8 GPUs), a
// single-column sweep (eval_constraint, the Hessian's forward sweep: 2000 columns = 125 column tiles) or 512+ states offer
// their parallelism in the ROW dimension.  Here R workgroups (on R CUs) share an interval group: each holds ALL term columns
// in LDS (the B operand of G_g * column needs the whole column) but computes only its npad / R rows of the next term, then
// the R slices are exchanged through global memory:
//
//   publish   every lane stores its new-term elements into X[cluster][t & 1][column][row] with 8-byte AGENT-scope atomic
//             stores (`global_store_dwordx2 sc1`: write-through, nothing left dirty in the XCD's L2), every wave drains
//             (`s_waitcnt vmcnt(0)`), the workgroup's barrier, then ONE lane adds 1 to the cluster's arrival counter
//   rendezvous one lane polls the counter (sc1 load + s_sleep) until all R workgroups of step t have arrived; barrier
//   collect   every lane reads the OTHER workgroups' slices with 8-byte agent-scope atomic loads (sc1: never served from
//             this CU's L1) straight into the LDS columns
//
// -- the "8-byte agent atomics on both sides" hand-off of MI355X_MICROARCH.md (Workgroup dispatch ... visibility, Valid forms
// and its table, first row: one lane of each storing workgroup signals for all its stores behind the workgroup's barrier; an
// sc1 poll of that counter; the other waves load behind a barrier the polling wave joins; hipMalloc memory; ONE workgroup per
// CU -- the launch pads its LDS request beyond half a CU's to guarantee that).  X is double buffered by step parity: a
// workgroup can only write step t+2 after every member has published t+1, i.e. has finished reading t.  Every spin is
// bounded: a member that does not arrive within ~0.5 s makes the sweep report "not converged" instead of hanging.
// Column norms are exchanged the same way (each workgroup knows only its rows' maxima), so all R members take the SAME
// termination decision from the same numbers.  Results are a function of (plan, data) alone: fixed K order per cluster.
struct ClusterArgs {
    FusedSweepArgs f;
    double* X;            // [clusters][2][NC_pad * npad + 8 * NC_pad] exchange slabs (terms, then norms)
    unsigned* arrive;     // [clusters] monotonic arrival counters (zeroed by the launcher)
    int n_groups;         // interval groups of ipw intervals
    int n_clusters;       // clusters in the grid (each walks groups cluster, cluster + n_clusters, ...)
};

__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MT, int NT, int R>
__global__ void __launch_bounds__(256, 1) k_sweep_cluster(ClusterArgs ca) {
    const FusedSweepArgs& a = ca.f;
    constexpr int NTHREADS = 256, NWAVES = 4;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int RL = 16 * MT;          // rows per wavefront
    constexpr int RW = 4 * RL;           // rows per workgroup = npad / R
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int npad = a.w.npad, Kpad = a.w.Kpad, T = a.ty.T, m = a.B.m, ipw = a.ipw;
    const int NC = T * ipw, ZS = npad + 2, KS = npad / 4;
    const int64_t typesz = (int64_t)Kpad * npad, nn = (int64_t)npad * npad;
    // cluster members sit on one XCD under the observed round-robin placement (blocks b and b + 8 share one): speed only
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int rank = jb % R, cluster = (jb / R) * 8 + xcd;
    if (cluster >= ca.n_clusters) return;
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
    const int NCP = 16 * NT;                                   // padded column count
    const int64_t xslab = (int64_t)NCP * npad + 8 * NCP;       // doubles per (cluster, parity): term slices, then [2 R][NCP] norms
    double* Xc = ca.X + (int64_t)cluster * 2 * xslab;
    unsigned* arrive = ca.arrive + cluster;
    unsigned arrivals = 0;                                      // rendezvous passed so far (x R = counter value to wait for)
    bool dead = false;                                          // a rendezvous timed out: finish without converging

    if (tid < T) {
        const TypeDesc td = a.ty.t[tid];
        xn[tid] = td.n_extra;
        xg[2 * tid] = td.gen[0]; xg[2 * tid + 1] = td.gen[1];
        xs[2 * tid] = td.src[0]; xs[2 * tid + 1] = td.src[1];
        xm[2 * tid] = td.mult[0]; xm[2 * tid + 1] = td.mult[1];
    }
    const int rowbase = rank * RW + (wave & 3) * RL;
    auto lane_row = [&](int ti) { return MT >= 2 ? rowbase + 32 * (ti / 2) + 2 * lr + (ti & 1) : rowbase + lr; };
    const int64_t astep = 4 * (int64_t)npad;
    const double* const aend = a.G + (int64_t)(m + 1) * nn;
    int t_max = 0, n_bad = 0;

    for (int grp = cluster; grp < ca.n_groups; grp += ca.n_clusters) {
        const int k0 = grp * ipw;
        __syncthreads();  // the previous group's LDS is done with
        // ---- per-interval coefficients, term 0 (every member loads the whole columns itself)
        if (tid < ipw) {
            const int kl = k0 + tid;
            const bool live = kl < a.P.n_int;
            const double* zk = a.Zsrc + (a.P.kn_lo + kl) * a.P.z;
            const double dt = live ? zk[a.P.dt_idx] : 0.0;
            const double inv_q = 1.0 / a.q;
            sE[tid] = dt * inv_q;
            if (rank == 0 && kl < Kpad) {
                a.w.scaleE[kl] = dt * inv_q;
                a.w.scaleE[Kpad + kl] = 2.0 * dt * inv_q;
            }
            for (int g = 0; g <= m; ++g) {
                const double ub = live ? (g == 0 ? 1.0 : zk[a.B.u_off + g - 1]) : 0.0;
                cg[g * ipw + tid] = dt * ub * inv_q;
                if (rank == 0 && kl < Kpad) {
                    a.w.scaleU[(int64_t)g * Kpad + kl] = ub;
                    a.w.scaleA[(int64_t)g * Kpad + kl] = dt * ub * inv_q;
                }
            }
        }
        if (tid < 2) flag[tid] = 0;
        for (int c = tid; c < 3 * NC; c += NTHREADS) tn[c] = 0ull;
        for (int c = tid; c < NC; c += NTHREADS) sn[c] = 0ull;
        __syncthreads();
        {
            double* Z0 = a.store ? a.w.Zt : a.w.Z[0];
            const bool to_global = a.store != 0 && rank == 0;
            for (int c = wave; c < NC; c += NWAVES) {
                const int ty = c / ipw, i = c - ty * ipw, kl = k0 + i;
                const bool live = ty == 0 && kl < a.P.n_int;
                const int64_t kn = a.P.kn_lo + kl;
                double mx = 0.0;
                for (int r = lane; r < npad; r += 64) {
                    double v = 0.0;
                    if (live && r < a.B.n) v = a.src_kind == 0 ? a.Zsrc[kn * a.P.z + a.B.x_off + r] : a.mu[a.B.row_off + kn * a.B.n + r];
                    Zs[c * ZS + r] = v;
                    if (to_global && kl < Kpad) Z0[((int64_t)ty * Kpad + kl) * npad + r] = v;
                    mx = fmax(mx, fabs(v));
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
                if (lane == 0) { tn[c] = fbits(mx); sn[c] = fbits(mx); }
            }
        }
        __syncthreads();

        int bcol[NT], bty[NT], bin[NT];
        bool bok[NT];
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            const int c = 16 * tj + lr;
            bok[tj] = c < NC;
            const int cc = bok[tj] ? c : NC - 1;
            bty[tj] = cc / ipw;
            bin[tj] = cc - bty[tj] * ipw;
            bcol[tj] = cc * ZS + lq;
        }
        const int g_first = grp % (m + 1);
        const int ks_first = ((grp / (m + 1)) & 3) * (KS / 4);
        d4 sreg[MT][NT];
#pragma unroll
        for (int ti = 0; ti < MT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * tj + 4 * r + lq;
                    sreg[ti][tj][r] = c < NC ? Zs[c * ZS + lane_row(ti)] : 0.0;
                }

        int t_exit = 0;
        bool conv = false;
        for (int round = 0; round < a.q; ++round) {
            if (round > 0) {
                // next sub-interval of exp(A) = exp(A/q)^q: the sums become term 0 of the new series -- every member needs all
                // rows of them, so they go through the exchange like a term
                __syncthreads();
                double* Xs = Xc + (int64_t)(arrivals & 1) * xslab;
#pragma unroll
                for (int ti = 0; ti < MT; ++ti)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = 16 * tj + 4 * r + lq;
                            if (c < NC) {
                                Zs[c * ZS + lane_row(ti)] = sreg[ti][tj][r];
                                st_agent(&Xs[(int64_t)c * npad + lane_row(ti)], sreg[ti][tj][r]);
                            }
                        }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                ++arrivals;
                if (tid == 0) {
                    __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    long spins = 0;
                    while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < arrivals * R) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1L << 19)) { flag[0] = 2; break; }
                    }
                }
                __syncthreads();
                if (flag[0] == 2) dead = true;
                for (int e = tid; e < NC * (npad - RW); e += NTHREADS) {
                    const int c = e / (npad - RW);
                    int r = e - c * (npad - RW);
                    if (r >= rank * RW) r += RW;  // skip the own slice
                    Zs[c * ZS + r] = ld_agent(&Xs[(int64_t)c * npad + r]);
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
            for (; t < a.d_ub && !dead; ++t) {
                double* Zout = a.store ? a.w.Zt + (int64_t)(t + 1) * T * typesz : a.w.Z[(t + 1) & 1];
                unsigned long long* tn_new = tn + ((t + 1) % 3) * NC;
                const double inv = 1.0 / (double)(t + 1);
                d4 acc[MT][NT];
                double* Xs = Xc + (int64_t)(arrivals & 1) * xslab;
                double* Xn = Xs + (int64_t)NCP * npad;  // norms: [rank][2][NCP] would need R slots; use [2 R][NCP] below
                {
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
                                for (int tj = 0; tj < NT; ++tj) bf[tj] = cA[tj] * z1[tj] + cB[tj] * z2[tj];
#pragma unroll
                                for (int tj = 0; tj < NT; ++tj) {
                                    z1[tj] = zp1[tj][4 * (u + 1)];
                                    z2[tj] = zp2[tj][4 * (u + 1)];
                                }
                                __builtin_amdgcn_sched_barrier(0);
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
                                issue(u);
                            }
#pragma unroll
                            for (int tj = 0; tj < NT; ++tj) { zp1[tj] += 4 * PF; zp2[tj] += 4 * PF; }
                        }
                    }
                    // ---- new term, own rows: sums, partial column norms, the store-mode copy, and the published slice
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int c = 16 * tj + 4 * r + lq;
                            const int cc = c < NC ? c : 0;
                            const int ty = cc / ipw, kl = k0 + cc - ty * ipw;
                            const bool ok = c < NC && kl < Kpad;
                            const int64_t colbase = ((int64_t)ty * Kpad + kl) * npad + rowbase;
                            double tmax = 0.0, smax = 0.0;
                            bool bad_t = false, bad_s = false;
                            if (c < NC) {
#pragma unroll
                                for (int p = 0; p < (MT >= 2 ? MT / 2 : 1); ++p) {
                                    const int ro = MT >= 2 ? 32 * p + 2 * lr : lr;
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
                                    if (a.store && ok) {
                                        if constexpr (MT >= 2) *reinterpret_cast<d2*>(Zout + colbase + ro) = v;
                                        else Zout[colbase + ro] = v.x;
                                    }
                                    st_agent(&Xs[(int64_t)c * npad + rowbase + ro], v.x);
                                    if constexpr (MT >= 2) st_agent(&Xs[(int64_t)c * npad + rowbase + ro + 1], v.y);
                                    tmax = fmax(tmax, fmax(fabs(v.x), fabs(v.y)));
                                    smax = fmax(smax, fmax(fabs(sv.x), fabs(sv.y)));
                                    bad_t = bad_t || !(v.x == v.x) || !(v.y == v.y);
                                    bad_s = bad_s || !(sv.x == sv.x) || !(sv.y == sv.y);
                                }
                            }
                            unsigned long long tb = bad_t ? 0x7ff8000000000000ull : fbits(tmax), sb = bad_s ? 0x7ff8000000000000ull : fbits(smax);
#pragma unroll
                            for (int o = 8; o > 0; o >>= 1) {
                                const unsigned long long t2 = __shfl_xor(tb, o, 64), s2 = __shfl_xor(sb, o, 64);
                                tb = t2 > tb ? t2 : tb;
                                sb = s2 > sb ? s2 : sb;
                            }
                            if (c < NC && lr == 0) {
                                atomicMax(&tn_new[c], tb);
                                atomicMax(&sn[c], sb);
                            }
                        }
                }
                __syncthreads();  // every wave is done with the old term columns; this member's partial norms are complete
                // own slice into the LDS columns; partial norms into the exchange slab
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * tj + 4 * r + lq;
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
                if (tid < NC) {
                    st_agent(&Xn[(int64_t)(2 * rank) * NCP + tid], fbits_to_d(tn_new[tid]));
                    st_agent(&Xn[(int64_t)(2 * rank + 1) * NCP + tid], fbits_to_d(sn[tid]));
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the barrier
                __syncthreads();
                ++arrivals;
                if (tid == 0) {
                    __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    long spins = 0;
                    while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < arrivals * R) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1L << 19)) { flag[t & 1] = 2; break; }   // bounded: a member that never arrives
                    }
                }
                __syncthreads();
                if (flag[t & 1] == 2) { dead = true; break; }
                // the other members' slices and norms: ALL loads of a lane are issued before the first of them is used (an
                // agent-scope load is a round trip to the fabric, ~1-2 us: one trip per lane, not one per element)
                {
                    constexpr int NOTHER = NT * 16 * (R - 1) * RW;        // upper bound on the slice elements to collect
                    constexpr int PER = (NOTHER + NTHREADS - 1) / NTHREADS;
                    constexpr int CH = PER < 8 ? PER : 8;                  // loads in flight per lane and trip
                    const int total_e = NC * (npad - RW);
                    unsigned long long tb = 0, sb = 0;
                    double nrm[2 * (R - 1)];
                    if (tid < NC) {
                        int k = 0;
#pragma unroll
                        for (int o = 0; o < R; ++o) {
                            if (o == rank) continue;
                            nrm[k++] = ld_agent(&Xn[(int64_t)(2 * o) * NCP + tid]);
                            nrm[k++] = ld_agent(&Xn[(int64_t)(2 * o + 1) * NCP + tid]);
                        }
                        tb = tn_new[tid]; sb = sn[tid];
                    }
                    for (int u0 = 0; u0 < PER; u0 += CH) {
                        double got[CH];
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            const int e = tid + (u0 + u) * NTHREADS;
                            if (e < total_e) {
                                const int c = e / (npad - RW);
                                int r = e - c * (npad - RW);
                                if (r >= rank * RW) r += RW;
                                got[u] = ld_agent(&Xs[(int64_t)c * npad + r]);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            const int e = tid + (u0 + u) * NTHREADS;
                            if (e < total_e) {
                                const int c = e / (npad - RW);
                                int r = e - c * (npad - RW);
                                if (r >= rank * RW) r += RW;
                                Zs[c * ZS + r] = got[u];
                            }
                        }
                    }
                    if (tid < NC) {
#pragma unroll
                        for (int k = 0; k < R - 1; ++k) {
                            const unsigned long long t2 = (unsigned long long)__double_as_longlong(nrm[2 * k]);
                            const unsigned long long s2 = (unsigned long long)__double_as_longlong(nrm[2 * k + 1]);
                            tb = t2 > tb ? t2 : tb;
                            sb = s2 > sb ? s2 : sb;
                        }
                        tn_new[tid] = tb;
                        sn[tid] = sb;
                    }
                }
                __syncthreads();
                // Al-Mohy & Higham's test on the cluster-wide norms: the same decision in every member
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
                __syncthreads();
                if (flag[t & 1] == 0) { conv = true; break; }
            }
            t_exit = conv ? t + 2 : a.d_ub + 1;
        }
        // the sums of this member's rows leave the registers once
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * tj + 4 * r + lq;
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
        if (tid == 0 && rank == 0) {
            if (a.w.nterms) a.w.nterms[grp] = conv ? t_exit : 0;
            t_max = t_exit > t_max ? t_exit : t_max;
            if (!conv) ++n_bad;
        }
    }
    if (tid == 0 && rank == 0) {
        atomicMax(&a.w.stats[1], t_max);
        if (n_bad) atomicAdd(&a.w.stats[0], n_bad);
    }
}

template <int MT, int NT, int WC = 1, int WK = 1>
hipError_t launch_one(hipStream_t st, const FusedSweepArgs& a, int nblocks, size_t lds) {
    if (a.w.npad != 64 * MT) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_sweep_fused<MT, NT, WC, WK>), dim3(nblocks), dim3(256 * WC * WK), lds, st, a);
    return hipGetLastError();
}
template <int MT, int NT, int WC = 1, int WK = 1>
hipError_t prepare_one(int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep_fused<MT, NT, WC, WK>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
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
    if (e == hipSuccess) e = prepare_one<4, 1, 1, 2>(bytes);
    if (e == hipSuccess) e = prepare_one<4, 2, 1, 2>(bytes);
    if (e == hipSuccess) e = prepare_one<4, 3, 1, 2>(bytes);
#define DTO_PREP64(MP, NX) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep_s64<MP, NX>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
    DTO_PREP64(5, 0); DTO_PREP64(5, 1); DTO_PREP64(5, 2); DTO_PREP64(3, 0); DTO_PREP64(3, 1); DTO_PREP64(3, 2);
#undef DTO_PREP64
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
    // 64 states, at most 16 columns per interval: the generator-stationary form, as many intervals as a 16-column tile holds
    // (measured per 1000-knot Jacobian sweep: see DESIGN.md section 4)
    static const int s64_env = tune_int("DTO_SWEEP_S64", 1);
    if (npad == 64 && s64_env && m + 1 <= 5 && T <= 16) {
        int nx = 0;
        for (int t = 0; t < T; ++t) nx = nx > ty.t[t].n_extra ? nx : ty.t[t].n_extra;
        // a workgroup's Taylor step costs the same for 1 or 16 live columns, so few intervals are spread over the CUs first
        // (one round of workgroups) and only then packed into the tile
        // (beside another kernel -- the Hessian's forward column next to its adjoint sweep -- the CU time is what counts: full tiles)
        int ipw = shared_chip ? 16 / T : (int)((n_int + n_cu - 1) / n_cu);
        ipw = ipw < 1 ? 1 : (ipw > 16 / T ? 16 / T : ipw);
        const long nblocks = (long)((n_int + ipw - 1) / ipw);
        if (nx <= 2) {
            out.MT = 1; out.NT = 1; out.WC = 1; out.WK = 1; out.ipw = ipw; out.nslot = nslot; out.nblocks = (int)nblocks;
            out.S64 = 1; out.NX = nx;
            out.lds_bytes = (size_t)S64Lds(T, m, ipw).total * sizeof(double);
            return true;
        }
    }
    out.S64 = 0;
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
    static const int wk_env = tune_int("DTO_SWEEP_WK", 2);  // A/B runs (TUNING builds): 1 = one wave per SIMD as up to round 3
    out.WK = 1;
    if (shared_chip && npad == 256 && search(2)) return true;
    if (!search(1)) return false;
    if (npad == 256 && out.WC == 1 && wk_env == 2) out.WK = 2;   // two waves per SIMD splitting the K loop
    return true;
}

// ---- row-split cluster form: plan, workspace, launch
namespace {
template <int MT, int NT, int R>
hipError_t launch_cluster_one(hipStream_t st, const ClusterArgs& a, int nblocks, size_t lds) {
    if (a.f.w.npad != 64 * MT * R) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_sweep_cluster<MT, NT, R>), dim3(nblocks), dim3(256), lds, st, a);
    return hipGetLastError();
}
template <int MT, int NT, int R>
hipError_t prepare_cluster_one(int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep_cluster<MT, NT, R>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
}  // namespace

hipError_t sweep_cluster_prepare() {
    const int bytes = 160 * 1024;
    hipError_t e = hipSuccess;
#define DTO_PREPC(MT, NT, R) if (e == hipSuccess) e = prepare_cluster_one<MT, NT, R>(bytes)
    DTO_PREPC(1, 1, 2); DTO_PREPC(1, 2, 2); DTO_PREPC(1, 3, 2);   // 128 states
    DTO_PREPC(2, 1, 2); DTO_PREPC(2, 2, 2); DTO_PREPC(2, 3, 2);   // 256 states over 2
    DTO_PREPC(1, 1, 4); DTO_PREPC(1, 2, 4); DTO_PREPC(1, 3, 4);   // 256 states over 4
    DTO_PREPC(4, 1, 2); DTO_PREPC(4, 2, 2);                       // 512 states over 2
    DTO_PREPC(2, 1, 4); DTO_PREPC(2, 2, 4);                       // 512 states over 4
    DTO_PREPC(4, 1, 4);                                           // 1024 states over 4
#undef DTO_PREPC
    return e;
}

// Shape of the cluster launch: R members per interval group, NT column tiles (ipw = 16 NT / T intervals per group), as many
// clusters as the chip holds with ONE workgroup per CU (a multiple of 8, one per XCD and slot); each cluster walks the groups
// cluster, cluster + n_clusters, ...  Cost model per Taylor step, in MFMA units of one 16-column tile over 64 MT rows: the
// tile count times the same L2 factor as the single-workgroup form (a narrow tile streams the generators faster than a CU
// takes them in) plus the exchange (rendezvous + slice traffic), which does not shrink with the tile.
bool sweep_cluster_plan(int npad, int m, const SweepTypes& ty, int64_t n_int, int n_cu, ClusterSweepPlan& out) {
    const int T = ty.T;
    if (T < 1 || n_int <= 0 || n_cu < 16) return false;
    static const double l2_factor[4] = {0.0, 1.6, 1.15, 1.0};
    static const int force_r = tune_int("DTO_CLUSTER_R", 0), force_nt = tune_int("DTO_CLUSTER_NT", 0);
    bool found = false;
    double best = 0.0;
    for (int R = 2; R <= 4; R += 2) {
        if (npad % (64 * R) != 0) continue;
        const int MT = npad / (64 * R);
        if (MT != 1 && MT != 2 && MT != 4) continue;
        if (force_r && R != force_r) continue;
        for (int NT = 1; NT <= 3; ++NT) {
            if (force_nt && NT != force_nt) continue;
            const int ipw = (16 * NT) / T;
            if (ipw < 1) continue;
            if (NT > 1 && (16 * (NT - 1)) / T == ipw) continue;  // the narrower shape holds as many intervals
            if (NT == 3 || (MT == 4 && NT > 1)) continue;  // three-tile shapes and 256 rows x 2 tiles spill beside the exchange registers
            const FusedLds L(npad, T, m, ipw, 0, MT);
            size_t bytes = (size_t)L.total * sizeof(double);
            if (bytes > 156 * 1024) continue;
            if (bytes < 82 * 1024) bytes = 82 * 1024;   // more than half a CU's LDS: one workgroup per CU (the hand-off's condition)
            const long n_groups = (long)((n_int + ipw - 1) / ipw);
            long n_clusters = ((long)(n_cu / R) / 8) * 8;
            if (n_clusters > ((n_groups + 7) / 8) * 8) n_clusters = ((n_groups + 7) / 8) * 8;
            if (n_clusters < 8) continue;
            const long rounds = (n_groups + n_clusters - 1) / n_clusters;
            const double step_us = MT * NT * l2_factor[NT] * (m + 1) * npad * 7.3e-3;   // 64 cycles per MFMA at 2.2 GHz
            const double exch_us = 3.0 + 16.0 * NT * npad * 8.0 * 1e-3 / 60.0;           // rendezvous + slices at ~60 GB/s per CU
            const double cost = rounds * (step_us + exch_us);
            if (!found || cost < best) {
                found = true; best = cost;
                out.MT = MT; out.NT = NT; out.R = R; out.ipw = ipw; out.n_groups = (int)n_groups; out.n_clusters = (int)n_clusters;
                out.nblocks = (int)(n_clusters * R); out.lds_bytes = bytes; out.step_us = step_us + exch_us;
            }
        }
    }
    return found;
}

size_t sweep_cluster_workspace_doubles(int npad, const ClusterSweepPlan& pl) {
    const size_t NCP = 16 * (size_t)pl.NT;
    return (size_t)pl.n_clusters * 2 * (NCP * npad + 8 * NCP);
}

hipError_t launch_sweep_cluster(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, const SweepTypes& ty,
                                const ClusterSweepPlan& pl, double* X, unsigned* arrive, const double* dZ, const double* dmu,
                                int src_kind, int transposed, int q, int d_ub, int tc, bool store, double tol) {
    ClusterArgs c{};
    FusedSweepArgs& a = c.f;
    a.P = P; a.B = B; a.w = w; a.ty = ty;
    a.G = transposed ? B.GT : B.G;
    a.Zsrc = dZ; a.mu = dmu; a.src_kind = src_kind;
    a.q = q; a.d_ub = d_ub; a.tc = tc; a.ipw = pl.ipw; a.store = store ? 1 : 0; a.nslot = 0; a.tol = tol;
    c.X = X; c.arrive = arrive; c.n_groups = pl.n_groups; c.n_clusters = pl.n_clusters;
    hipError_t e = hipMemsetAsync(arrive, 0, sizeof(unsigned) * (size_t)pl.n_clusters, st);
    if (e != hipSuccess) return e;
    const int key = pl.MT * 100 + pl.NT * 10 + pl.R;
    switch (key) {
        case 112: return launch_cluster_one<1, 1, 2>(st, c, pl.nblocks, pl.lds_bytes);
        case 122: return launch_cluster_one<1, 2, 2>(st, c, pl.nblocks, pl.lds_bytes);
        case 132: return launch_cluster_one<1, 3, 2>(st, c, pl.nblocks, pl.lds_bytes);
        case 212: return launch_cluster_one<2, 1, 2>(st, c, pl.nblocks, pl.lds_bytes);
        case 222: return launch_cluster_one<2, 2, 2>(st, c, pl.nblocks, pl.lds_bytes);
        case 232: return launch_cluster_one<2, 3, 2>(st, c, pl.nblocks, pl.lds_bytes);
        case 114: return launch_cluster_one<1, 1, 4>(st, c, pl.nblocks, pl.lds_bytes);
        case 124: return launch_cluster_one<1, 2, 4>(st, c, pl.nblocks, pl.lds_bytes);
        case 134: return launch_cluster_one<1, 3, 4>(st, c, pl.nblocks, pl.lds_bytes);
        case 412: return launch_cluster_one<4, 1, 2>(st, c, pl.nblocks, pl.lds_bytes);
        case 422: return launch_cluster_one<4, 2, 2>(st, c, pl.nblocks, pl.lds_bytes);
        case 214: return launch_cluster_one<2, 1, 4>(st, c, pl.nblocks, pl.lds_bytes);
        case 224: return launch_cluster_one<2, 2, 4>(st, c, pl.nblocks, pl.lds_bytes);
        case 414: return launch_cluster_one<4, 1, 4>(st, c, pl.nblocks, pl.lds_bytes);
    }
    return hipErrorInvalidValue;
}

// The step budget of a sweep from the cheap norm bound alone, on the device (one lane): the host's cheap_plan(loose) where it gives a
// single round, else its plan_sweep -- same formulas, same constants (dto_engine.cpp), so the numbers are those a host that waited
// for the bound would have used.  out = {q, d_ub, tc}.
__global__ void k_plan_dev(const unsigned long long* __restrict__ bounds, int32_t* __restrict__ out) {
    const double beta = __longlong_as_double((long long)bounds[0]);
    auto budget = [](double br) {
        int t = 8;
        double term = 1.0;
        for (int i = 1; i <= t; ++i) term *= br / i;
        while (term > 1e-19 && t < 200) { ++t; term *= br / t; }
        return t + 6;
    };
    int q = 1, d_ub = 30, tc = -1;
    if (beta == beta && beta <= 1e6) {
        q = (int)ceil(beta / 9.0);
        if (q < 1) q = 1;
        d_ub = budget(beta / q);
        if (q > 1 && beta < 40.0) {
            // the hump criterion on the bound itself: max_k beta^k / k! <= e^9 (the maximum sits at k = floor(beta) or next to it)
            double lh = 0.0;
            const int k0 = (int)floor(beta);
            for (int k = (k0 > 1 ? k0 - 1 : 1); k <= k0 + 1; ++k) lh = fmax(lh, k * log(beta) - lgamma(k + 1.0));
            if (lh <= 9.0) {
                q = 1;
                d_ub = budget(beta);
                const int a0 = d_ub / 2 - 1, a1 = (int)ceil(beta) + 4;
                tc = a0 < (a1 > 2 ? a1 : 2) ? a0 : (a1 > 2 ? a1 : 2);
            }
        }
    }
    if (tc < 0) tc = d_ub / 2 - 1;
    if (tc < 2) tc = 0;
    out[0] = q; out[1] = d_ub; out[2] = tc;
}
void launch_plan_dev(hipStream_t st, const unsigned long long* bounds, int32_t* out) {
    hipLaunchKernelGGL(k_plan_dev, dim3(1), dim3(1), 0, st, bounds, out);
}

hipError_t launch_sweep_fused(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, const SweepTypes& ty,
                              const FusedSweepPlan& pl, const double* dZ, const double* dmu, int src_kind, int transposed,
                              int q, int d_ub, int tc, bool store, double tol, const int32_t* plan_dev) {
    if (plan_dev && !pl.S64) return hipErrorInvalidValue;   // only the 64-state form reads its plan from the device
    FusedSweepArgs a{};
    a.plan_dev = plan_dev;
    a.P = P; a.B = B; a.w = w; a.ty = ty;
    a.G = transposed ? B.GT : B.G;
    a.Zsrc = dZ; a.mu = dmu; a.src_kind = src_kind;
    a.q = q; a.d_ub = d_ub; a.tc = tc; a.ipw = pl.ipw; a.store = store ? 1 : 0; a.nslot = pl.nslot; a.tol = tol;
    if (pl.S64) {
        if (w.npad != 64 || B.m + 1 > 5 || ty.T * pl.ipw > 16) return hipErrorInvalidValue;
#define DTO_GO64(MP, NX) hipLaunchKernelGGL((k_sweep_s64<MP, NX>), dim3(pl.nblocks), dim3(256), pl.lds_bytes, st, a)
        switch ((B.m + 1 <= 3 ? 0 : 10) + pl.NX) {
            case 0: DTO_GO64(3, 0); break;
            case 1: DTO_GO64(3, 1); break;
            case 2: DTO_GO64(3, 2); break;
            case 10: DTO_GO64(5, 0); break;
            case 11: DTO_GO64(5, 1); break;
            case 12: DTO_GO64(5, 2); break;
            default: return hipErrorInvalidValue;
        }
#undef DTO_GO64
        return hipGetLastError();
    }
    if (pl.WC == 2 && pl.MT == 4 && pl.NT == 2) return launch_one<4, 2, 2>(st, a, pl.nblocks, pl.lds_bytes);
    if (pl.WK == 2 && pl.MT == 4 && pl.WC == 1) {
        if (pl.NT == 1) return launch_one<4, 1, 1, 2>(st, a, pl.nblocks, pl.lds_bytes);
        if (pl.NT == 2) return launch_one<4, 2, 1, 2>(st, a, pl.nblocks, pl.lds_bytes);
        if (pl.NT == 3) return launch_one<4, 3, 1, 2>(st, a, pl.nblocks, pl.lds_bytes);
    }
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
