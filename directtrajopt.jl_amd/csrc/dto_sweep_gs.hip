// dto_sweep_gs.hip -- the generator sweep with the GENERATORS stationary in registers (gfx950, round 4).
//
// The sweep's Taylor recurrence is, per step, one small GEMM  [G_0 .. G_m] (npad x (m+1) npad)  x  (scaled term columns of all
// intervals).  The generators are shared by every interval and every step; what changes is the term panel.  The earlier forms
// stream the generators from L2 for every step and column tile (k_sweep_fused / k_sweep_cluster: a workgroup re-reads all
// (m+1) npad^2 doubles per Taylor term) or pay a kernel boundary plus a reduction launch per term (k_sweep, split-K) -- which is
// what binds single-column sweeps (eval_constraint, the Hessian's forward column) and short shards (the 250-knot share of the
// metric on 8 GPUs): there a lone 16-column tile asks L2 for the generators faster than a CU takes them in.
//
// Here the operand that never changes never moves.  A cluster of R = npad / 32 workgroups, one per CU, shares an interval group;
// member `rank` keeps rows 32 rank .. 32 rank + 31 of ALL generators in the registers of its 4 wavefronts for the whole launch
// (wavefront w holds the k-slice npad w / 4 .. npad (w+1) / 4 of every generator: (m+1) npad / 16 A fragments of 16 bytes per lane
// = 320 of the 512 registers of a one-wave-per-SIMD kernel at 256 states x 4 drives -- 64 % of the CU's register file holds the
// operand; MFMA A/B operands may be AGPRs).  Per (group, Taylor term) -- an "item":
//
//   collect    the group's term columns [columns][npad] come from global memory into LDS (every member needs whole columns: they
//              are the B operand), 16-byte sc1 loads; per-interval bilinear coefficients are rebuilt from Z (a few loads)
//   product    each wavefront: its k-slice of every generator times the columns -- B fragments from LDS scaled on the fly
//              (cA * z + cB * z_src, as in the fused form), A fragments from registers: no global or L2 traffic at all
//   reduce     the 4 partial 32 x 16 NT tiles of the wavefronts go through LDS (the slot of the columns just consumed) and are
//              added in FIXED order by the thread that owns a (column, row pair): new term = sum / (t+1), running sum (global S,
//              read and written by that one thread only), partial column norms
//   publish    the member's 32-row slice of the new term -> global memory with 16-byte sc1 (write-through) stores, partial
//              norms likewise; every wave drains (s_waitcnt vmcnt(0)), barrier, ONE lane adds 1 to the group's arrival counter
//   rendezvous before the group's next item one lane polls that counter (sc1 load + s_sleep, bounded) until all R members have
//              published; barrier; then every load of the handed-off bytes is an sc1 load
//
// -- the hand-off of MI355X_MICROARCH.md, Valid forms, first row of its table (one lane of each storing workgroup signals for all
// that workgroup's stores behind the workgroup's barrier; sc1 poll; the other waves load behind a barrier the polling wave joins;
// hipMalloc memory; one workgroup per CU -- 256 threads x 512 registers fill a CU; stores and loads all sc1, 8 or 16 bytes).
// A cluster walks its groups round-robin, so with two or more groups per cluster the data a rendezvous waits for was published a
// whole item earlier and the poll returns at once.  The term slabs are the sweep's own buffers (ping-pong Z[0/1], or the term store
// Zt of the Hessian's pairing path): a member can only write term t+2 of a group after every member has published t+1, i.e. has
// finished reading t.  Every spin is bounded: a member that never arrives makes the sweep report "not converged" -- an error
// through the ABI -- instead of hanging the device.  All R members take the same termination decision from the same exchanged
// norms (Al-Mohy & Higham's test, as k_sweep_check).
//
// Results are a function of the data alone: the K order is (wavefront k-slice, generator, k) for every column wherever it is
// computed -- independent of the plan, the cluster count and the grid.
//
// Bound: FP64 MFMA.  Per item and member 2 * 32 * 16 NT * (m+1) npad flops; nothing but the term slices moves.
#include "dto_gemm.hip.h"
#include "dto_kernels.h"

namespace dto {

namespace {

typedef unsigned int gs_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long gs_fbits(double v) { return (unsigned long long)__double_as_longlong(fabs(v)); }
__device__ __forceinline__ double gs_bits_to_d(unsigned long long b) { return __longlong_as_double((long long)b); }
__device__ __forceinline__ void gs_st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double gs_ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct GsArgs {
    KProb P;
    KBil B;
    SweepBuf w;
    SweepTypes ty;
    const double* G;     // generators used as the left operand (G or G')
    const double* Zsrc;  // the NLP vector
    const double* mu;    // multipliers (src_kind 1)
    int src_kind, d_ub, tc, ipw, store;
    double tol;
    unsigned* arrive;    // [n_groups] monotonic arrival counters (zeroed by the launcher)
    double* Xn;          // [n_groups][3 slots, term index mod 3][R][2][16 NT] partial column norms (term, sum)
    int n_groups, n_clusters;
};

constexpr int GS_AUX_SC1 = 16;  // raw_buffer aux bit: sc1 (agent-scope: write-through stores, L1-bypassing loads)

// KU = npad / 32 = members per cluster; a wavefront holds KW = 2 KU k-steps of every generator; MP >= m + 1 generator slots (absent
// ones are zero); NT column tiles per group; HAS_SRC: some column type has an inhomogeneous term (tangent columns).
template <int KU, int MP, int NT, bool HAS_SRC>
__global__ void __launch_bounds__(256, 1) k_sweep_gs(GsArgs a) {
    constexpr int R = KU, NPAD = 32 * KU, ZS = NPAD + 2, NCP = 16 * NT, NTHREADS = 256, NWAVES = 4, KW = 2 * KU;
    constexpr int FE = (16 * NCP + NTHREADS - 1) / NTHREADS;   // (column, row pair) elements a thread finishes per item
    static_assert(NWAVES * 32 <= ZS, "the column slot also holds the partial tiles [4][NCP][32]");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* Zs = lds;
    double* cg = lds + NCP * ZS + 8;      // [MP][NCP] bilinear coefficients dt ubar_g of the group's intervals
    double* sE = cg + MP * NCP;           // [NCP] dt
    int* flag = reinterpret_cast<int*>(sE + NCP);   // [0..1] "go on" of the termination test (by item parity), [2] a rendezvous timed out
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int Kpad = a.w.Kpad, T = a.ty.T, m = a.B.m, ipw = a.ipw, NC = T * ipw;
    const int64_t typesz = (int64_t)Kpad * NPAD, nn = (int64_t)NPAD * NPAD;
    // cluster members sit on one XCD under the observed round-robin placement (blocks b and b + 8 share one): speed only
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int rank = jb % R, cluster = (jb / R) * 8 + xcd;
    if (cluster >= a.n_clusters) return;
    const int row0 = 32 * rank;

    // ---- the stationary operand: this wavefront's k-slice of this member's 32 rows of every generator.  Fragment (g, u) is
    // k-step wave KW + u of generator g: lane (lr, lq) holds column k = 4 (wave KW + u) + lq, rows row0 + 2 lr and + 1 (even
    // rows -> accumulator tile 0, odd rows -> tile 1: one 16-byte load feeds both)
    d2 af[MP][KW];
#pragma unroll
    for (int g = 0; g < MP; ++g)
#pragma unroll
        for (int u = 0; u < KW; ++u) {
            if (g <= m) af[g][u] = *reinterpret_cast<const d2*>(a.G + (int64_t)g * nn + (int64_t)(4 * (wave * KW + u) + lq) * NPAD + row0 + 2 * lr);
            else af[g][u] = d2{0.0, 0.0};
        }

    // ---- B operand of this lane: column 16 tj + lr of the group, rows 4 (wave KW + u) + lq of the term columns; padding
    // columns re-read the last real one with coefficient 0.  An inhomogeneous term of the column's type rides in the segment
    // of its generator: cB * (source type's column of the same interval)
    int bcol[NT], bin[NT];
    bool bok[NT];
    int xgen0[NT], xgen1[NT], xso0[NT], xso1[NT];
    double xm0[NT], xm1[NT];
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
        const int c = 16 * tj + lr;
        bok[tj] = c < NC;
        const int cc = bok[tj] ? c : NC - 1;
        const int bty = cc / ipw;
        bin[tj] = cc - bty * ipw;
        bcol[tj] = cc * ZS + lq + 4 * wave * KW;
        xgen0[tj] = xgen1[tj] = -1; xso0[tj] = xso1[tj] = bcol[tj]; xm0[tj] = xm1[tj] = 0.0;
        if (HAS_SRC && bok[tj]) {
            const TypeDesc td = a.ty.t[bty];
            if (td.n_extra > 0) { xgen0[tj] = td.gen[0]; xso0[tj] = (td.src[0] * ipw + bin[tj]) * ZS + lq + 4 * wave * KW; xm0[tj] = td.mult[0]; }
            if (td.n_extra > 1) { xgen1[tj] = td.gen[1]; xso1[tj] = (td.src[1] * ipw + bin[tj]) * ZS + lq + 4 * wave * KW; xm1[tj] = td.mult[1]; }
        }
    }
    if (tid < 3) flag[tid] = 0;
    __syncthreads();

    const int n_local = a.n_groups > cluster ? (a.n_groups - cluster + a.n_clusters - 1) / a.n_clusters : 0;   // <= 64 (plan)
    unsigned long long active = n_local >= 64 ? ~0ull : ((1ull << n_local) - 1ull);
    int t_max = 0, n_bad = 0, item = 0;
    bool dead = false;

    for (int t = 0; t <= a.d_ub && active != 0ull && !dead; ++t) {
        // term t of every group lives in slab `in`, term t+1 goes to slab `out` (ping-pong, or the term store)
        const double* Zin = a.store ? a.w.Zt + (int64_t)t * T * typesz : a.w.Z[t & 1];
        double* Zout = a.store ? a.w.Zt + (int64_t)(t + 1) * T * typesz : a.w.Z[(t + 1) & 1];
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Zin), 0, (int)(T * typesz * 8), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(Zout, 0, (int)(T * typesz * 8), 0x00020000);
        const double inv = 1.0 / (double)(t + 1);
        for (int lg = 0; lg < n_local; ++lg) {
            if (!((active >> lg) & 1ull)) continue;
            const int grp = cluster + lg * a.n_clusters, k0 = grp * ipw;
            const int fs = item & 1;
            ++item;
            double* Xg = a.Xn + (int64_t)grp * (3 * R * 2 * NCP);
            // ---- rendezvous: all R members have published term t of this group
            if (t > 0) {
                if (tid == 0) {
                    flag[fs] = 0;
                    long spins = 0;
                    while (__hip_atomic_load(a.arrive + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(R * t)) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1L << 19)) { flag[2] = 1; break; }   // bounded: a member that never arrives
                    }
                }
                __syncthreads();
                if (flag[2]) { dead = true; break; }
                // Al-Mohy & Higham's test on the cluster-wide norms of terms t-1, t and of the sum: the same decision in every member
                if (t - 1 >= a.tc) {
                    if (tid < NC) {
                        const int p1 = t % 3, p0 = (t + 2) % 3;   // slots of terms t and t-1 (a fast member may already be writing slot (t+1) % 3)
                        unsigned long long b0 = 0ull, b1 = 0ull, bs = 0ull;
#pragma unroll
                        for (int o = 0; o < R; ++o) {
                            const unsigned long long v1 = (unsigned long long)__double_as_longlong(gs_ld_agent(Xg + ((p1 * R + o) * 2) * NCP + tid));
                            const unsigned long long vs = (unsigned long long)__double_as_longlong(gs_ld_agent(Xg + ((p1 * R + o) * 2 + 1) * NCP + tid));
                            const unsigned long long v0 = (unsigned long long)__double_as_longlong(gs_ld_agent(Xg + ((p0 * R + o) * 2) * NCP + tid));
                            b1 = v1 > b1 ? v1 : b1; bs = vs > bs ? vs : bs; b0 = v0 > b0 ? v0 : b0;
                        }
                        const double a0 = gs_bits_to_d(b0), a1 = gs_bits_to_d(b1), s = gs_bits_to_d(bs);
                        if (!(a0 + a1 <= a.tol * s) && (a0 + a1 == a0 + a1) && s < 1e300) flag[fs] = 1;
                    }
                } else if (tid == 0) {
                    flag[fs] = 1;
                }
            }
            // ---- collect: per-interval coefficients and the group's term columns
            if (t < a.d_ub) {
                if (tid < ipw) {
                    const int kl = k0 + tid;
                    const bool live = kl < a.P.n_int;
                    const double* zk = a.Zsrc + (a.P.kn_lo + kl) * a.P.z;
                    const double dt = live ? zk[a.P.dt_idx] : 0.0;
                    sE[tid] = dt;
                    if (t == 0 && rank == 0 && kl < Kpad) {
                        a.w.scaleE[kl] = dt;
                        a.w.scaleE[Kpad + kl] = 2.0 * dt;
                    }
#pragma unroll
                    for (int g = 0; g < MP; ++g) {
                        const double ub = (live && g <= m) ? (g == 0 ? 1.0 : zk[a.B.u_off + g - 1]) : 0.0;
                        cg[g * NCP + tid] = dt * ub;
                        if (t == 0 && rank == 0 && kl < Kpad && g <= m) {
                            a.w.scaleU[(int64_t)g * Kpad + kl] = ub;
                            a.w.scaleA[(int64_t)g * Kpad + kl] = dt * ub;
                        }
                    }
                }
                if (t == 0) {
                    // term 0: the state (or the multipliers) in the type-0 columns, zero elsewhere; every member loads the whole columns
                    double* Z0 = a.store ? a.w.Zt : a.w.Z[0];
                    const bool to_global = a.store != 0 && rank == 0;   // the pairing path reads term 0 from the store
                    for (int c = wave; c < NCP; c += NWAVES) {
                        const int ty = c < NC ? c / ipw : 0, kl = k0 + (c < NC ? c - ty * ipw : 0);
                        const bool live = c < NC && ty == 0 && kl < a.P.n_int;
                        const int64_t kn = a.P.kn_lo + kl;
                        for (int r = lane; r < NPAD; r += 64) {
                            double v = 0.0;
                            if (live && r < a.B.n) v = a.src_kind == 0 ? a.Zsrc[kn * a.P.z + a.B.x_off + r] : a.mu[a.B.row_off + kn * a.B.n + r];
                            Zs[c * ZS + r] = v;
                            if (to_global && c < NC && kl < Kpad) Z0[((int64_t)ty * Kpad + kl) * NPAD + r] = v;
                        }
                    }
                } else {
                    constexpr int UPC = NPAD / 2;                        // 16-byte units per column
                    constexpr int PER = (NCP * UPC + NTHREADS - 1) / NTHREADS;
                    constexpr int CH = PER < 16 ? PER : 16;              // loads in flight per lane and trip
#pragma unroll 1
                    for (int j0 = 0; j0 < PER; j0 += CH) {
                        gs_u4 got[CH];
#pragma unroll
                        for (int j = 0; j < CH; ++j) {
                            const int e = tid + (j0 + j) * NTHREADS;
                            const int c = e / UPC, un = e - c * UPC;
                            const int ty = c < NC ? c / ipw : 0, kl = k0 + (c < NC ? c - ty * ipw : 0);
                            got[j] = gs_u4{0u, 0u, 0u, 0u};
                            if (c < NC && kl < Kpad)
                                got[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)((((int64_t)ty * Kpad + kl) * NPAD + 2 * un) * 8), 0, GS_AUX_SC1);
                        }
#pragma unroll
                        for (int j = 0; j < CH; ++j) {
                            const int e = tid + (j0 + j) * NTHREADS;
                            const int c = e / UPC, un = e - c * UPC;
                            if (c < NCP) *reinterpret_cast<gs_u4*>(Zs + c * ZS + 2 * un) = got[j];
                        }
                    }
                }
            }
            __syncthreads();
            if (t > 0 && flag[fs] == 0) {
                // converged with terms 0 .. t
                active &= ~(1ull << lg);
                if (tid == 0 && rank == 0) {
                    if (a.w.nterms) a.w.nterms[grp] = t + 1;
                    t_max = t + 1 > t_max ? t + 1 : t_max;
                }
                continue;
            }
            if (t == a.d_ub) {
                active &= ~(1ull << lg);
                if (tid == 0 && rank == 0) {
                    if (a.w.nterms) a.w.nterms[grp] = 0;
                    t_max = a.d_ub + 1 > t_max ? a.d_ub + 1 : t_max;
                    ++n_bad;
                }
                continue;
            }
            // ---- the elements this thread finishes: column fc, rows row0 + 2 frp, + 1; their running sums so far
            d2 s_prev[FE];
#pragma unroll
            for (int fe = 0; fe < FE; ++fe) {
                const int e = tid + fe * NTHREADS, fc = e >> 4, frp = e & 15;
                const int fty = fc < NC ? fc / ipw : 0, fkl = k0 + (fc < NC ? fc - fty * ipw : 0);
                s_prev[fe] = d2{0.0, 0.0};
                if (fc < NC && fkl < Kpad)
                    s_prev[fe] = t == 0 ? *reinterpret_cast<const d2*>(Zs + fc * ZS + row0 + 2 * frp)
                                        : *reinterpret_cast<const d2*>(a.w.S + ((int64_t)fty * Kpad + fkl) * NPAD + row0 + 2 * frp);
            }

            // ---- product: this wavefront's k-slice of every generator
            d4 acc[2][NT];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) acc[rt][tj] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int g = 0; g < MP; ++g) {
                double cA[NT], cB[NT], z1[NT], z2[NT];
                int so[NT];
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    cA[tj] = bok[tj] ? cg[g * NCP + bin[tj]] : 0.0;
                    cB[tj] = 0.0;
                    so[tj] = bcol[tj];
                    if (HAS_SRC) {
                        const double e = sE[bin[tj]];
                        if (xgen0[tj] == g) { cB[tj] = e * xm0[tj]; so[tj] = xso0[tj]; }
                        if (xgen1[tj] == g) { cB[tj] = e * xm1[tj]; so[tj] = xso1[tj]; }
                    }
                    z1[tj] = Zs[bcol[tj]];
                    if (HAS_SRC) z2[tj] = Zs[so[tj]];
                }
#pragma unroll
                for (int u = 0; u < KW; ++u) {
                    double bf[NT];
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) {
                        if (HAS_SRC) bf[tj] = cA[tj] * z1[tj] + cB[tj] * z2[tj];
                        else bf[tj] = cA[tj] * z1[tj];
                    }
                    if (u + 1 < KW) {
                        // raw term values of the next k-step: the reads go out BEFORE the MFMA block, which hides them
#pragma unroll
                        for (int tj = 0; tj < NT; ++tj) {
                            z1[tj] = Zs[bcol[tj] + 4 * (u + 1)];
                            if (HAS_SRC) z2[tj] = Zs[so[tj] + 4 * (u + 1)];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int tj = 0; tj < NT; ++tj)
                            acc[rt][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[tj], rt ? af[g][u].y : af[g][u].x, acc[rt][tj], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();   // every wave is done with the term columns: the slot becomes the partial-sum scratch
            // partial tile of this wavefront: accumulator register r of column tile tj holds column 16 tj + 4 r + lq, rows 2 lr, 2 lr + 1
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * tj + 4 * r + lq;
                    *reinterpret_cast<d2*>(Zs + (wave * NCP + c) * 32 + 2 * lr) = d2{acc[0][tj][r], acc[1][tj][r]};
                }
            __syncthreads();
            // ---- reduce in fixed order, new term, running sum, partial norms, publish
#pragma unroll
            for (int fe = 0; fe < FE; ++fe) {
                const int e = tid + fe * NTHREADS, fc = e >> 4, frp = e & 15;
                if (e < 16 * NCP) {
                    const int fty = fc < NC ? fc / ipw : 0, fkl = k0 + (fc < NC ? fc - fty * ipw : 0);
                    const bool fok = fc < NC && fkl < Kpad;
                    const int64_t foff = ((int64_t)fty * Kpad + fkl) * NPAD + row0 + 2 * frp;
                    d2 sum = *reinterpret_cast<const d2*>(Zs + fc * 32 + 2 * frp);
#pragma unroll
                    for (int w2 = 1; w2 < NWAVES; ++w2) {
                        const d2 p = *reinterpret_cast<const d2*>(Zs + (w2 * NCP + fc) * 32 + 2 * frp);
                        sum.x += p.x; sum.y += p.y;
                    }
                    const d2 v = d2{sum.x * inv, sum.y * inv};
                    double tmax = 0.0, smax = 0.0, t0max = 0.0;
                    bool bad_t = false, bad_s = false;
                    if (fok) {
                        const d2 sv = d2{s_prev[fe].x + v.x, s_prev[fe].y + v.y};
                        *reinterpret_cast<d2*>(a.w.S + foff) = sv;
                        gs_u4 bits;
                        __builtin_memcpy(&bits, &v, 16);
                        __builtin_amdgcn_raw_buffer_store_b128(bits, rs_out, (int)(foff * 8), 0, GS_AUX_SC1);
                        tmax = fmax(fabs(v.x), fabs(v.y));
                        smax = fmax(fabs(sv.x), fabs(sv.y));
                        t0max = fmax(fabs(s_prev[fe].x), fabs(s_prev[fe].y));
                        bad_t = !(v.x == v.x) || !(v.y == v.y);       // a NaN must survive the max
                        bad_s = !(sv.x == sv.x) || !(sv.y == sv.y);
                    }
                    unsigned long long tb = bad_t ? 0x7ff8000000000000ull : gs_fbits(tmax), sb = bad_s ? 0x7ff8000000000000ull : gs_fbits(smax);
                    unsigned long long t0b = gs_fbits(t0max);
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) {
                        const unsigned long long t2 = __shfl_xor(tb, o, 64), s2 = __shfl_xor(sb, o, 64), u2 = __shfl_xor(t0b, o, 64);
                        tb = t2 > tb ? t2 : tb;
                        sb = s2 > sb ? s2 : sb;
                        t0b = u2 > t0b ? u2 : t0b;
                    }
                    if (frp == 0 && fc < NC) {
                        const int p1 = (t + 1) % 3;
                        gs_st_agent(Xg + ((p1 * R + rank) * 2) * NCP + fc, gs_bits_to_d(tb));
                        gs_st_agent(Xg + ((p1 * R + rank) * 2 + 1) * NCP + fc, gs_bits_to_d(sb));
                        if (t == 0) gs_st_agent(Xg + ((0 * R + rank) * 2) * NCP + fc, gs_bits_to_d(t0b));   // norms of term 0 (test at the first step)
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the barrier
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(a.arrive + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0 && rank == 0) {
        // groups cut off by a timed-out rendezvous count as not converged
        if (dead) {
            for (int lg = 0; lg < n_local; ++lg)
                if ((active >> lg) & 1ull) { ++n_bad; if (a.w.nterms) a.w.nterms[cluster + lg * a.n_clusters] = 0; }
            t_max = a.d_ub + 1;
        }
        atomicMax(&a.w.stats[1], t_max);
        if (n_bad) atomicAdd(&a.w.stats[0], n_bad);
    }
}

template <int KU, int MP, int NT, bool HAS_SRC>
hipError_t gs_launch_one(hipStream_t st, const GsArgs& a, int nblocks, size_t lds) {
    hipLaunchKernelGGL((k_sweep_gs<KU, MP, NT, HAS_SRC>), dim3(nblocks), dim3(256), lds, st, a);
    return hipGetLastError();
}
template <int KU, int MP, int NT, bool HAS_SRC>
hipError_t gs_prepare_one(int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep_gs<KU, MP, NT, HAS_SRC>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

size_t gs_lds_bytes(int KU, int MP, int NT) {
    const int NPAD = 32 * KU, ZS = NPAD + 2, NCP = 16 * NT;
    size_t d = (size_t)NCP * ZS + 8 + (size_t)MP * NCP + NCP + 4;
    size_t bytes = d * sizeof(double);
    if (bytes < 82 * 1024) bytes = 82 * 1024;   // more than half a CU's LDS: one workgroup per CU (the hand-off's condition)
    return bytes;
}

}  // namespace

hipError_t sweep_gs_prepare() {
    const int bytes = 160 * 1024;
    hipError_t e = hipSuccess;
#define DTO_PREPG(KU, MP, NT) \
    if (e == hipSuccess) e = gs_prepare_one<KU, MP, NT, false>(bytes); \
    if (e == hipSuccess) e = gs_prepare_one<KU, MP, NT, true>(bytes)
    DTO_PREPG(8, 5, 1); DTO_PREPG(8, 5, 2); DTO_PREPG(8, 5, 4); DTO_PREPG(8, 3, 1); DTO_PREPG(8, 3, 2); DTO_PREPG(8, 3, 4);
#undef DTO_PREPG
    return e;
}

// Shape of the launch: KU = npad / 32 members per cluster, as many clusters as the chip holds with one workgroup per CU (a multiple
// of 8: one per XCD and slot), NT column tiles per group (ipw = 16 NT / T intervals), each cluster walking the groups cluster,
// cluster + n_clusters, ... round-robin.  Cost per Taylor term: rounds x (product + collect / reduce / publish), in us.
bool sweep_gs_plan(int npad, int m, const SweepTypes& ty, int64_t n_int, int n_cu, GsSweepPlan& out) {
    const int T = ty.T;
    if (T < 1 || n_int <= 0) return false;
    if (npad != 256) return false;   // (128 states: KU = 4 -- not instantiated yet)
    if (m + 1 > 5) return false;
    const int KU = npad / 32, R = KU;
    bool has_src = false;
    for (int t = 0; t < T; ++t) {
        if (ty.t[t].n_extra > 2) return false;
        if (ty.t[t].n_extra > 0) has_src = true;
    }
    static const int force_nt = tune_int("DTO_GS_NT", 0);
    bool found = false;
    double best = 0.0;
    for (int NT = 1; NT <= 4; NT += NT < 2 ? 1 : 2) {
        if (force_nt && NT != force_nt) continue;
        const int ipw = (16 * NT) / T;
        if (ipw < 1) continue;
        const long n_groups = (long)((n_int + ipw - 1) / ipw);
        long n_clusters = ((long)(n_cu / R) / 8) * 8;
        if (n_clusters > ((n_groups + 7) / 8) * 8) n_clusters = ((n_groups + 7) / 8) * 8;
        if (n_clusters < 8) continue;
        const long rounds = (n_groups + n_clusters - 1) / n_clusters;
        if (rounds > 64) continue;
        const double prod_us = 2.0 * NT * (m + 1) * 2 * KU * 64.0 / 2200.0;   // MFMAs per wave x 64 cycles at 2.2 GHz
        const double fix_us = rounds > 1 ? 3.5 : 6.0;                            // collect + reduce + publish (+ the exposed rendezvous of a lone group)
        const double cost = rounds * (prod_us + fix_us);
        if (!found || cost < best) {
            found = true; best = cost;
            out.KU = KU; out.MP = m + 1 <= 3 ? 3 : 5; out.NT = NT; out.ipw = ipw; out.has_src = has_src ? 1 : 0;
            out.n_groups = (int)n_groups; out.n_clusters = (int)n_clusters; out.nblocks = (int)(n_clusters * R);
            out.lds_bytes = gs_lds_bytes(KU, out.MP, NT); out.term_us = cost;
        }
    }
    return found;
}

size_t sweep_gs_norm_doubles(const GsSweepPlan& pl) { return (size_t)pl.n_groups * 3 * pl.KU * 2 * 16 * pl.NT; }

hipError_t launch_sweep_gs(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, const SweepTypes& ty, const GsSweepPlan& pl,
                           double* Xn, unsigned* arrive, const double* dZ, const double* dmu, int src_kind, int transposed, int d_ub,
                           int tc, bool store, double tol) {
    if (w.npad != 32 * pl.KU) return hipErrorInvalidValue;
    if ((size_t)ty.T * w.Kpad * w.npad * 8 >= (1ull << 31)) return hipErrorInvalidValue;   // 32-bit buffer offsets
    GsArgs a{};
    a.P = P; a.B = B; a.w = w; a.ty = ty;
    a.G = transposed ? B.GT : B.G;
    a.Zsrc = dZ; a.mu = dmu; a.src_kind = src_kind;
    a.d_ub = d_ub; a.tc = tc; a.ipw = pl.ipw; a.store = store ? 1 : 0; a.tol = tol;
    a.arrive = arrive; a.Xn = Xn; a.n_groups = pl.n_groups; a.n_clusters = pl.n_clusters;
    hipError_t e = hipMemsetAsync(arrive, 0, sizeof(unsigned) * (size_t)((pl.n_groups + 3) / 4 * 4), st);
    if (e != hipSuccess) return e;
#define DTO_GS_CASE(KU_, MP_, NT_) \
    if (pl.KU == KU_ && pl.MP == MP_ && pl.NT == NT_) \
        return pl.has_src ? gs_launch_one<KU_, MP_, NT_, true>(st, a, pl.nblocks, pl.lds_bytes) : gs_launch_one<KU_, MP_, NT_, false>(st, a, pl.nblocks, pl.lds_bytes)
    DTO_GS_CASE(8, 5, 1); DTO_GS_CASE(8, 5, 2); DTO_GS_CASE(8, 5, 4); DTO_GS_CASE(8, 3, 1); DTO_GS_CASE(8, 3, 2); DTO_GS_CASE(8, 3, 4);
#undef DTO_GS_CASE
    return hipErrorInvalidValue;
}

}  // namespace dto
