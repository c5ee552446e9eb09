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

// max of a non-negative double over the 16 lanes of a DPP row (all lanes of the row get it): rotate right by 8, 4, 2, 1
template <int CTRL>
__device__ __forceinline__ double gs_dpp_mov(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double gs_row16_max(double v) {
    v = fmax(v, gs_dpp_mov<0x128>(v));   // row_ror:8
    v = fmax(v, gs_dpp_mov<0x124>(v));   // row_ror:4
    v = fmax(v, gs_dpp_mov<0x122>(v));   // row_ror:2
    v = fmax(v, gs_dpp_mov<0x121>(v));   // row_ror:1
    return v;
}

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
    int cap;             // intervals a cluster handles at most (rows of the coefficient table in LDS)
#ifdef GS_STAMP
    unsigned long long* stamp;   // diagnostic builds (tools/sweep_gs_probe -DGS_STAMP): s_memtime at the phase boundaries of block 0's items
#endif
};

#ifdef GS_STAMP
#define GS_MARK(i) do { if (blockIdx.x == 0 && tid == 0 && n_item < 64) a.stamp[n_item * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GS_MARK(i) do { } while (0)
#endif

#ifndef GS_BK
#define GS_BK 2
#endif
constexpr int GS_AUX_SC1 = 16;  // raw_buffer aux bit: sc1 (agent-scope: write-through stores, L1-bypassing loads)

// LDS carve-up shared by host and device (doubles)
struct GsLds {
    int slot, nmax, tyt, colt, gct, flag, cgt, total;   // total: without the coefficient table [(MP + 1)][cap] that follows at cgt
    __host__ __device__ constexpr GsLds(int KU, int MP, int NT) : slot(0), nmax(0), tyt(0), colt(0), gct(0), flag(0), cgt(0), total(0) {
        const int NPAD = 32 * KU, ZS = NPAD + 2, NCP = 16 * NT;
        (void)MP;
        slot = NCP * ZS + 8;            // one column slot (also the scratch of the partial tiles [4][NCP][32]); two of them
        int o = 2 * slot;
        nmax = o; o += 2 * 3 * NCP;     // [2][3][NCP] cluster-wide column norms: term t-1, term t, sum (bit patterns)
        tyt = o; o += 5 * MAX_TYPES;    // per column type: multipliers [2] (doubles), then n_extra, generators [2], sources [2] (ints)
        colt = o; o += NCP;             // per column: {offset of the column inside a term slab for interval group 0, interval within the group} (ints)
        gct = o; o += MP * NCP + (MP * NCP + 1) / 2;   // per (generator, column): multiplier of the inhomogeneous term (0: none), its source column (ints)
        flag = o; o += 2;               // ints: [0] a rendezvous timed out
        cgt = o;                        // [(MP + 1)][cap]: dt ubar_g of every interval of the cluster (row MP: dt), built once
        total = o;
    }
};

// KU = npad / 32 = members per cluster; a wavefront holds KW = 2 KU k-steps of every generator; MP >= m + 1 generator slots (absent
// ones are zero); NT column tiles per group (two column slots in LDS: NT <= 2 at 256 states); HAS_SRC: some column type has an
// inhomogeneous term (tangent columns).
//
// Software pipeline over the items of a cluster (round-robin over its active groups, term by term): while the MFMA loop of item i
// runs on one LDS slot, the columns of item i+1 are in flight from global memory into registers (issued after the poll of ITS
// rendezvous, in two chunks rotated at a generator boundary of the loop) and land in the other slot; the termination test of
// item i+1 is evaluated from the exchanged norms before item i's reduction, so the next iteration starts on ready data.  The
// publish of item i is drained and signalled at once while at most two groups are active (the next poll but one waits for it),
// and behind the NEXT item's first barrier when three or more are (nothing waits for it that soon: the drain costs nothing then).
template <int KU, int MP, int NT, bool HAS_SRC>
__global__ void __launch_bounds__(256, 1) k_sweep_gs(GsArgs a) {
    constexpr int R = KU, NPAD = 32 * KU, ZS = NPAD + 2, NCP = 16 * NT, NTHREADS = 256, NWAVES = 4, KW = 2 * KU;
    constexpr int FE = NT;                                    // (column, row pair) elements a thread finishes per item: 16 NCP / 256
    constexpr int UPC = NPAD / 2;                             // 16-byte units per column
    constexpr int PER = NCP * UPC / NTHREADS;                 // units a thread collects per item
    constexpr int NCH = PER > 8 ? PER / 8 : 1, CH = PER / NCH;   // in chunks of 8 units (32 registers in flight) ...
    constexpr int GROT = (MP + 1) / 2;                           // ... that rotate at this generator boundary of the loop
    static_assert(CH * NCH == PER && NCH <= 2, "the chunks tile the collect");
    static_assert(NWAVES * 32 <= ZS, "the column slot also holds the partial tiles [4][NCP][32]");
    static_assert(R * NCP <= NTHREADS, "one thread per (member, column) loads the exchanged norms");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr GsLds L(KU, MP, NT);
    double* tym = lds + L.tyt;                                    // [T][2] multipliers of the inhomogeneous terms
    int* tyi = reinterpret_cast<int*>(lds + L.tyt + 2 * MAX_TYPES);  // [T][5]: n_extra, generators [2], source types [2]
    double* gcm = lds + L.gct;                                       // [MP][NCP] multiplier of the inhomogeneous term of (generator, column)
    int* gcs = reinterpret_cast<int*>(lds + L.gct + MP * NCP);       // [MP][NCP] its source column
    int* colt = reinterpret_cast<int*>(lds + L.colt);                // [NCP][2]: (type Kpad + i) NPAD / 2 in 16-byte units (-1: padding column), i
    unsigned long long* nmaxb = reinterpret_cast<unsigned long long*>(lds + L.nmax);
    int* flag = reinterpret_cast<int*>(lds + L.flag);
    double* cgt = lds + L.cgt;
    const int cap = a.cap;
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
            // the fragment must STAY in a register: an opaque pass keeps the compiler from re-loading it from memory inside the loop
            // when registers get tight (it did, in a build whose loop it had left partly rolled: 176 global loads per item)
#ifdef GS_OPAQUE_AF
            double fx = af[g][u].x, fy = af[g][u].y;
            asm volatile("" : "+v"(fx), "+v"(fy));
            af[g][u] = d2{fx, fy};
#endif
        }

    // ---- B operand of this lane: column 16 tj + lr of the group, rows 4 (wave KW + u) + lq of the term columns; padding
    // columns re-read the last real one with coefficient 0.  An inhomogeneous term of the column's type rides in the segment
    // of its generator: cB * (source type's column of the same interval)
    int bcol[NT], bin[NT], bty[NT];
    bool bok[NT];
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
        const int c = 16 * tj + lr;
        bok[tj] = c < NC;
        const int cc = bok[tj] ? c : NC - 1;
        bty[tj] = cc / ipw;
        bin[tj] = cc - bty[tj] * ipw;
        bcol[tj] = cc * ZS + lq + 4 * wave * KW;
    }
    if (tid < NCP) {
        const int ty = tid / ipw, i = tid - ty * ipw;
        colt[2 * tid] = tid < NC ? (int)((((int64_t)ty * Kpad + i) * NPAD) / 2) : -1;
        colt[2 * tid + 1] = tid < NC ? i : 0;
    }
    for (int idx = tid; idx < MP * NCP; idx += NTHREADS) {
        const int g = idx / NCP, c = idx - g * NCP;
        double mult = 0.0;
        int srcc = c < NC ? c : NC - 1;
        if (c < NC) {
            const int ty = c / ipw, i = c - ty * ipw;
            const TypeDesc td = a.ty.t[ty];
            for (int x = 0; x < td.n_extra; ++x)
                if (td.gen[x] == g) { mult = td.mult[x]; srcc = td.src[x] * ipw + i; }
        }
        gcm[idx] = mult;
        gcs[idx] = srcc;
    }
    if (tid < T) {
        const TypeDesc td = a.ty.t[tid];
        tym[2 * tid] = td.mult[0]; tym[2 * tid + 1] = td.mult[1];
        tyi[5 * tid] = td.n_extra;
        tyi[5 * tid + 1] = td.gen[0]; tyi[5 * tid + 2] = td.gen[1];
        tyi[5 * tid + 3] = td.src[0]; tyi[5 * tid + 4] = td.src[1];
    }
    if (tid < 2) flag[tid] = 0;
    for (int i = tid; i < 2 * 3 * NCP; i += NTHREADS) nmaxb[i] = 0ull;
    __syncthreads();

    const int n_local = a.n_groups > cluster ? (a.n_groups - cluster + a.n_clusters - 1) / a.n_clusters : 0;   // <= 64 (plan)
    unsigned long long active = n_local >= 64 ? ~0ull : ((1ull << n_local) - 1ull);
    int t_max = 0, n_bad = 0;
    bool dead = false;
    // bilinear coefficients dt ubar_g of every interval this cluster handles, once (the loop then reads them from LDS: no
    // vector-memory load of its own stands in front of a collect chunk in the in-order vmcnt queue); the scale factors later
    // kernels read are written on the way
    for (int idx = tid; idx < n_local * ipw; idx += NTHREADS) {
        const int lg = idx / ipw, i = idx - lg * ipw;
        const int kl = (cluster + lg * a.n_clusters) * ipw + i;
        const bool live = kl < a.P.n_int;
        const double* zk = a.Zsrc + (a.P.kn_lo + (live ? kl : 0)) * a.P.z;
        const double dt = live ? zk[a.P.dt_idx] : 0.0;
        cgt[MP * cap + idx] = dt;
        if (rank == 0 && kl < Kpad) {
            a.w.scaleE[kl] = dt;
            a.w.scaleE[Kpad + kl] = 2.0 * dt;
        }
#pragma unroll
        for (int g = 0; g < MP; ++g) {
            const double ub = (live && g <= m) ? (g == 0 ? 1.0 : zk[a.B.u_off + g - 1]) : 0.0;
            cgt[g * cap + idx] = dt * ub;
            if (rank == 0 && kl < Kpad && g <= m) {
                a.w.scaleU[(int64_t)g * Kpad + kl] = ub;
                a.w.scaleA[(int64_t)g * Kpad + kl] = dt * ub;
            }
        }
    }
    __syncthreads();

    // ---- the pieces of an item.  Registers of a collect in flight: column units, per-interval coefficients, exchanged norms
    gs_u4 got[CH];
    unsigned long long nr0 = 0ull, nr1 = 0ull, nrs = 0ull;

    auto term_in = [&](int t) -> const double* { return a.store ? a.w.Zt + (int64_t)t * T * typesz : a.w.Z[t & 1]; };
    auto term_out = [&](int t) -> double* { return a.store ? a.w.Zt + (int64_t)(t + 1) * T * typesz : a.w.Z[(t + 1) & 1]; };
    // one lane waits until all R members have published term t of the group (bounded)
    auto poll = [&](int grp, int t) {
        long spins = 0;
        while (__hip_atomic_load(a.arrive + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(R * t)) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1L << 20)) { flag[0] = 1; break; }   // a member that never arrives (~0.3 s: a member may start late when
                                                                  // another stream's kernels hold CUs at launch)
        }
    };
    // chunk `ch` of the group's term-t columns -> registers (t = 0: the state or the multipliers in the type-0 columns).  Unit e =
    // tid + 256 k of the slot is 16-byte unit tid % UPC of column (tid / UPC) + (256 / UPC) k
    auto issue_chunk = [&](int grp, int t, int ch) {
        const int k0 = grp * ipw;
        const int un = tid % UPC, cbase = tid / UPC;
        // (the two sources are kept in separate, uniformly branched loops: a destination register that either of two kinds of
        // load may write makes the compiler drain vmcnt in front of every load)
        if (t > 0) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(term_in(t)), 0, (int)(T * typesz * 8), 0x00020000);
            int voff[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int c = cbase + (NTHREADS / UPC) * (ch * CH + j);
                const int cu = colt[2 * c], kl = k0 + colt[2 * c + 1];
                // a column outside the group or the slab reads beyond the descriptor's range: the hardware returns zeros
                voff[j] = (cu >= 0 && kl < Kpad) ? (cu + k0 * (NPAD / 2) + un) * 16 : 0x7ffffff0;
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) got[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j], 0, GS_AUX_SC1);
        } else {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int c = cbase + (NTHREADS / UPC) * (ch * CH + j);
                const int kl = k0 + colt[2 * c + 1];
                const bool live = colt[2 * c] >= 0 && c < ipw && kl < a.P.n_int;
                const int64_t kn = a.P.kn_lo + (live ? kl : 0);
                const double* src = a.src_kind == 0 ? a.Zsrc + kn * a.P.z + a.B.x_off : a.mu + a.B.row_off + kn * a.B.n;
                const int r0 = 2 * un < a.B.n ? 2 * un : 0, r1 = 2 * un + 1 < a.B.n ? 2 * un + 1 : 0;
                const double vx = src[r0], vy = src[r1];
                const d2 v = d2{live && 2 * un < a.B.n ? vx : 0.0, live && 2 * un + 1 < a.B.n ? vy : 0.0};
                __builtin_memcpy(&got[j], &v, 16);
            }
        }
    };
    auto land_chunk = [&](int grp, int t, int ch, int slot) {
        double* Zs = lds + slot * L.slot;
        const int k0 = grp * ipw;
        double* Z0 = a.store ? a.w.Zt : a.w.Z[0];
        const bool to_global = t == 0 && a.store != 0 && rank == 0;   // the pairing path reads term 0 from the store
        const int un = tid % UPC, cbase = tid / UPC;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int c = cbase + (NTHREADS / UPC) * (ch * CH + j);
            *reinterpret_cast<gs_u4*>(Zs + c * ZS + 2 * un) = got[j];
            if (to_global) {
                const int cu = colt[2 * c], kl = k0 + colt[2 * c + 1];
                if (cu >= 0 && kl < Kpad) *reinterpret_cast<gs_u4*>(Z0 + 2 * ((int64_t)cu + (int64_t)k0 * (NPAD / 2) + un)) = got[j];
            }
        }
    };
    // the exchanged norms of terms t-1, t and of the sum -> registers
    auto issue_small = [&](int grp, int t) {
        nr0 = nr1 = nrs = 0ull;
        if (t > 0 && t - 1 >= a.tc && tid < R * NCP) {
            const int o = tid / NCP, c = tid - o * NCP;
            const double* Xg = a.Xn + (int64_t)grp * (3 * R * 2 * NCP);
            const int p1 = t % 3, p0 = (t + 2) % 3;   // slots of terms t and t-1 (a fast member may already be writing slot (t+1) % 3)
            if (c < NC) {
                nr1 = (unsigned long long)__double_as_longlong(gs_ld_agent(Xg + ((p1 * R + o) * 2) * NCP + c));
                nrs = (unsigned long long)__double_as_longlong(gs_ld_agent(Xg + ((p1 * R + o) * 2 + 1) * NCP + c));
                nr0 = (unsigned long long)__double_as_longlong(gs_ld_agent(Xg + ((p0 * R + o) * 2) * NCP + c));
            }
        }
    };
    auto land_small = [&](int grp, int t, int slot) {
        (void)grp;
        if (t > 0 && t - 1 >= a.tc && tid < R * NCP) {
            const int c = tid % NCP;
            if (c < NC) {
                atomicMax(&nmaxb[(slot * 3 + 0) * NCP + c], nr0);
                atomicMax(&nmaxb[(slot * 3 + 1) * NCP + c], nr1);
                atomicMax(&nmaxb[(slot * 3 + 2) * NCP + c], nrs);
            }
        }
    };
    // Al-Mohy & Higham's test on the cluster-wide norms (behind a barrier after land_small): true = the series goes on.
    // Every wavefront evaluates all columns itself: the same decision everywhere without another barrier.
    auto goes_on = [&](int t, int slot) -> bool {
        if (t == 0 || t - 1 < a.tc) return true;
        bool more = false;
        if (lane < NC) {
            const double a0 = gs_bits_to_d(nmaxb[(slot * 3 + 0) * NCP + lane]), a1 = gs_bits_to_d(nmaxb[(slot * 3 + 1) * NCP + lane]);
            const double sn = gs_bits_to_d(nmaxb[(slot * 3 + 2) * NCP + lane]);
            more = !(a0 + a1 <= a.tol * sn) && (a0 + a1 == a0 + a1) && sn < 1e300;
        }
        return __any(more);
    };
    auto clear_norms = [&](int slot) {
        for (int i = tid; i < 3 * NCP; i += NTHREADS) nmaxb[slot * 3 * NCP + i] = 0ull;
    };
    auto signal = [&](int grp) { __hip_atomic_fetch_add(a.arrive + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // successor of local group lg at term t among the active groups: the next one at t, else the first one at t + 1
    auto successor = [&](int lg, int t, int& nlg, int& nt) -> bool {
        const unsigned long long later = lg >= 63 ? 0ull : active & ~((2ull << lg) - 1ull);
        if (later) { nlg = __ffsll((long long)later) - 1; nt = t; return true; }
        if (t + 1 > a.d_ub || active == 0ull) return false;
        nlg = __ffsll((long long)active) - 1; nt = t + 1;
        return true;
    };
    auto retire = [&](int lg, int t, bool converged) {
        active &= ~(1ull << lg);
        if (tid == 0 && rank == 0) {
            const int grp = cluster + lg * a.n_clusters;
            if (a.w.nterms) a.w.nterms[grp] = converged ? t + 1 : 0;   // terms 0 .. t exist
            const int te = converged ? t + 1 : a.d_ub + 1;
            t_max = te > t_max ? te : t_max;
            if (!converged) ++n_bad;
        }
    };
    // whole collect of one item, nothing overlapped (the first item, a lone group, the item after a retirement); returns "goes on".
    // On entry every wave may still read either slot: the caller has passed a barrier since the last use of `slot`.
    auto collect_sync = [&](int lg, int t, int slot) -> bool {
        const int grp = cluster + lg * a.n_clusters;
        if (t > 0) {
            if (tid == 0) poll(grp, t);
            __syncthreads();
            if (flag[0]) { dead = true; return false; }
        }
        issue_small(grp, t);
        if (t < a.d_ub) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) { issue_chunk(grp, t, ch); land_chunk(grp, t, ch, slot); }
        }
        land_small(grp, t, slot);
        __syncthreads();
        const bool go = goes_on(t, slot);
        __syncthreads();          // every wave has read the norms
        clear_norms(slot);
        return go;
    };

    // the (column, row pair) elements this thread finishes in every item: slab offset of the column for interval group 0 (doubles),
    // the column's interval within its group; -1: a padding column
    int64_t f_off[FE];
    int f_int[FE];
#pragma unroll
    for (int fe = 0; fe < FE; ++fe) {
        const int fc = (tid + fe * NTHREADS) >> 4;
        f_off[fe] = colt[2 * fc] >= 0 ? 2 * (int64_t)colt[2 * fc] + row0 + 2 * (tid & 15) : -1;
        f_int[fe] = colt[2 * fc + 1];
    }
    int cur_lg = 0, cur_t = 0, slot = 0;
    bool have = n_local > 0;
    if (have) {
        // first item: term 0 of the first group
        (void)collect_sync(0, 0, 0);
    }
    int pending = -1;   // group whose publish is not yet drained and signalled
    int n_item = 0;
    (void)n_item;
    while (have && !dead) {
        GS_MARK(0);
        const int grp = cluster + cur_lg * a.n_clusters, k0 = grp * ipw, t = cur_t;
        double* Zs = lds + slot * L.slot;
        int nx_lg = 0, nx_t = 0;
        const bool has_next = successor(cur_lg, t, nx_lg, nx_t);
        const bool prefetch = has_next && nx_lg != cur_lg;     // (the same group again: its columns depend on this item's publish)
        const int nx_grp = cluster + nx_lg * a.n_clusters;
        // ---- step 1: the previous item's publish is drained; the next item's rendezvous
        if (pending >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (prefetch && nx_t > 0 && tid == 0) poll(nx_grp, nx_t);
        __syncthreads();
        if (flag[0]) { dead = true; break; }
        if (pending >= 0 && tid == 0) signal(pending);
        pending = -1;
        GS_MARK(1);
        // ---- step 2: the next item's collect goes out; this item's running sums
        if (prefetch) {
            issue_small(nx_grp, nx_t);
            if (nx_t < a.d_ub) issue_chunk(nx_grp, nx_t, 0);
        }
        d2 s_prev[FE];
#pragma unroll
        for (int fe = 0; fe < FE; ++fe) {
            const int e = tid + fe * NTHREADS, fc = e >> 4, frp = e & 15;
            s_prev[fe] = d2{0.0, 0.0};
            if (f_off[fe] >= 0 && k0 + f_int[fe] < Kpad)
                s_prev[fe] = t == 0 ? *reinterpret_cast<const d2*>(Zs + fc * ZS + row0 + 2 * frp)
                                    : *reinterpret_cast<const d2*>(a.w.S + f_off[fe] + (int64_t)k0 * NPAD);
        }
        // ---- step 3: product -- this wavefront's k-slice of every generator
        const int cbase = cur_lg * ipw;
        d4 acc[2][NT];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) acc[rt][tj] = d4{0.0, 0.0, 0.0, 0.0};
        GS_MARK(2);
        // The k-steps of all generators form ONE list of batches (GS_BK k-steps each).  The scaled B fragments of a batch are formed in
        // one go: FP64 vector and matrix instructions share a pipe on this part, and with one wave per SIMD every switch between them
        // is a bubble of a VALU latency (measured: ~27 cycles per VALU operation when each k-step scales its own fragment).  The raw
        // term values of the NEXT batch are read from LDS before a batch's MFMAs, and the coefficients of the NEXT generator at the
        // first batch of the current one: no dependent LDS chain stands at a generator boundary.
        constexpr int BK = GS_BK < KW ? GS_BK : KW, NB = KW / BK;
        double cA[NT], cB[NT], cAn[NT], cBn[NT], zr1[BK][NT], zr2[BK][NT];
        int so[NT], son[NT];
        auto coef = [&](int g, double (&xA)[NT], double (&xB)[NT], int (&xs)[NT]) {
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
                const int cc = bok[tj] ? 16 * tj + lr : NC - 1;
                xA[tj] = bok[tj] ? cgt[g * cap + cbase + bin[tj]] : 0.0;
                xB[tj] = 0.0;
                xs[tj] = bcol[tj];
                if (HAS_SRC) {
                    xB[tj] = bok[tj] ? cgt[MP * cap + cbase + bin[tj]] * gcm[g * NCP + cc] : 0.0;
                    xs[tj] = gcs[g * NCP + cc] * ZS + lq + 4 * wave * KW;
                }
            }
        };
        auto read_raw = [&](int u0, const int (&xs)[NT]) {
#pragma unroll
            for (int b = 0; b < BK; ++b)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    zr1[b][tj] = Zs[bcol[tj] + 4 * (u0 + b)];
                    if (HAS_SRC) zr2[b][tj] = Zs[xs[tj] + 4 * (u0 + b)];
                }
        };
        coef(0, cA, cB, so);
        read_raw(0, so);
#pragma unroll
        for (int g = 0; g < MP; ++g)
#pragma unroll
        for (int ub = 0; ub < NB; ++ub) {
            const int u0 = ub * BK;
            if (ub == 0) {
                if (NCH > 1 && g == GROT && prefetch && nx_t < a.d_ub) {   // the collect rotates: chunk 0 was issued GROT generator blocks ago
                    land_chunk(nx_grp, nx_t, 0, slot ^ 1);
                    issue_chunk(nx_grp, nx_t, 1);
                }
                if (g + 1 < MP) coef(g + 1, cAn, cBn, son);
            }
            double bfs[BK][NT];
#pragma unroll
            for (int b = 0; b < BK; ++b)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
#if defined(GS_PROBE_NOLDS) || defined(GS_PROBE_NOVALU)   // timing probes (tools/sweep_gs_probe -D...; wrong results)
                    bfs[b][tj] = zr1[b][tj];
#else
                    bfs[b][tj] = HAS_SRC ? cA[tj] * zr1[b][tj] + cB[tj] * zr2[b][tj] : cA[tj] * zr1[b][tj];
#endif
                }
#ifndef GS_PROBE_NOLDS
            if (ub + 1 < NB) read_raw(u0 + BK, so);
            else if (g + 1 < MP) read_raw(0, son);
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < BK; ++b)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj)
                        acc[rt][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfs[b][tj], rt ? af[g][u0 + b].y : af[g][u0 + b].x, acc[rt][tj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ub == NB - 1 && g + 1 < MP) {
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) { cA[tj] = cAn[tj]; cB[tj] = cBn[tj]; so[tj] = son[tj]; }
            }
        }
        GS_MARK(3);
        // ---- step 4: the next item's data lands in the other slot
        if (prefetch) {
            if (nx_t < a.d_ub) land_chunk(nx_grp, nx_t, NCH - 1, slot ^ 1);
            land_small(nx_grp, nx_t, slot ^ 1);
        }
        __syncthreads();   // every wave is done with this item's columns (the slot becomes the partial-sum scratch); the next item's are in place
        GS_MARK(4);
        const bool nx_go = prefetch ? goes_on(nx_t, slot ^ 1) : false;
        // partial tile of this wavefront: accumulator register r of column tile tj holds column 16 tj + 4 r + lq, rows 2 lr, 2 lr + 1
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * tj + 4 * r + lq;
                *reinterpret_cast<d2*>(Zs + (wave * NCP + c) * 32 + 2 * lr) = d2{acc[0][tj][r], acc[1][tj][r]};
            }
        __syncthreads();
        GS_MARK(5);
        if (prefetch) clear_norms(slot ^ 1);   // (every wave has taken its decision in front of the barrier above)
        // ---- step 5: reduce in fixed order, new term, running sum, partial norms, publish
        {
            double* Xg = a.Xn + (int64_t)grp * (3 * R * 2 * NCP);
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(term_out(t), 0, (int)(T * typesz * 8), 0x00020000);
            const double inv = 1.0 / (double)(t + 1);
#pragma unroll
            for (int fe = 0; fe < FE; ++fe) {
                const int e = tid + fe * NTHREADS, fc = e >> 4, frp = e & 15;
                const bool fok = f_off[fe] >= 0 && k0 + f_int[fe] < Kpad;
                const int64_t foff = f_off[fe] + (int64_t)k0 * NPAD;
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
                // maxima over the 16 lanes that share the column (one DPP row): rotations inside the row, vector ALU only; a NaN must
                // survive the max, so the lanes' NaN flags travel as a ballot and the row's share of it decides
                tmax = gs_row16_max(tmax); smax = gs_row16_max(smax);
                if (t == 0) t0max = gs_row16_max(t0max);
                const unsigned long long nan_t = __ballot(bad_t), nan_s = __ballot(bad_s);
                const int rsh = lane & 48;
                const unsigned long long tb = ((nan_t >> rsh) & 0xffffull) ? 0x7ff8000000000000ull : gs_fbits(tmax);
                const unsigned long long sb = ((nan_s >> rsh) & 0xffffull) ? 0x7ff8000000000000ull : gs_fbits(smax);
                const unsigned long long t0b = gs_fbits(t0max);
                if (frp == 0 && fc < NC) {
                    const int p1 = (t + 1) % 3;
                    gs_st_agent(Xg + ((p1 * R + rank) * 2) * NCP + fc, gs_bits_to_d(tb));
                    gs_st_agent(Xg + ((p1 * R + rank) * 2 + 1) * NCP + fc, gs_bits_to_d(sb));
                    if (t == 0) gs_st_agent(Xg + ((0 * R + rank) * 2) * NCP + fc, gs_bits_to_d(t0b));   // norms of term 0 (test at the first step)
                }
            }
        }
        GS_MARK(6);
        // ---- step 6: drain and signal -- at once, or behind the next item's first barrier when no poll can be waiting for it yet
        if (prefetch && __popcll(active) >= 3) {
            pending = grp;
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the barrier
            __syncthreads();
            if (tid == 0) signal(grp);
        }
        GS_MARK(7);
        ++n_item;
        // ---- advance
        if (prefetch && nx_go && nx_t < a.d_ub) {
            cur_lg = nx_lg; cur_t = nx_t; slot ^= 1;
            continue;
        }
        // the prefetched item ended its group (converged, or out of steps), or nothing could be prefetched: find the next item
        // that goes on, collecting without overlap
        if (pending >= 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) signal(pending);
            pending = -1;
        }
        int at_lg = cur_lg, at_t = t;
        if (prefetch) { retire(nx_lg, nx_t, !nx_go); at_lg = nx_lg; at_t = nx_t; }
        have = false;
        __syncthreads();   // (the partial-sum scratch of this item has been read: its slot may be refilled)
        for (;;) {
            int c_lg = 0, c_t = 0;
            if (!successor(at_lg, at_t, c_lg, c_t)) break;
            const bool go = collect_sync(c_lg, c_t, slot ^ 1);
            if (dead) break;
            if (go && c_t < a.d_ub) { cur_lg = c_lg; cur_t = c_t; slot ^= 1; have = true; break; }
            retire(c_lg, c_t, !go);
            at_lg = c_lg; at_t = c_t;
        }
    }
    if (pending >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) signal(pending);
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

size_t gs_lds_bytes(int KU, int MP, int NT, int cap) {
    size_t bytes = ((size_t)GsLds(KU, MP, NT).total + (size_t)(MP + 1) * cap) * sizeof(double);
    if (bytes < 82 * 1024) bytes = 82 * 1024;   // more than half a CU's LDS: one workgroup per CU (the hand-off's condition)
    return bytes;
}

}  // namespace

#ifdef GS_STAMP
static unsigned long long* gs_stamp_buffer = nullptr;
#endif

hipError_t sweep_gs_prepare() {
    const int bytes = 160 * 1024;
    hipError_t e = hipSuccess;
#define DTO_PREPG(KU, MP, NT) \
    if (e == hipSuccess) e = gs_prepare_one<KU, MP, NT, false>(bytes); \
    if (e == hipSuccess) e = gs_prepare_one<KU, MP, NT, true>(bytes)
    DTO_PREPG(8, 5, 1); DTO_PREPG(8, 5, 2); DTO_PREPG(8, 3, 1); DTO_PREPG(8, 3, 2);
    DTO_PREPG(4, 5, 1); DTO_PREPG(4, 5, 2); DTO_PREPG(4, 3, 1); DTO_PREPG(4, 3, 2);
#undef DTO_PREPG
    return e;
}

// Shape of the launch: KU = npad / 32 members per cluster, as many clusters as the chip holds with one workgroup per CU (a multiple
// of 8: one per XCD and slot), NT column tiles per group (ipw = 16 NT / T intervals), each cluster walking the groups cluster,
// cluster + n_clusters, ... round-robin.  Cost per Taylor term: rounds x (product + collect / reduce / publish), in us.
bool sweep_gs_plan(int npad, int m, const SweepTypes& ty, int64_t n_int, int n_cu, GsSweepPlan& out) {
    const int T = ty.T;
    if (T < 1 || n_int <= 0) return false;
    if (npad != 256 && npad != 128) return false;   // clusters of 8 / 4 members of 32 rows each
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
    for (int NT = 1; NT <= 2; ++NT) {
        if (force_nt && NT != force_nt) continue;
        const int ipw = (16 * NT) / T;
        if (ipw < 1) continue;
        const long n_groups = (long)((n_int + ipw - 1) / ipw);
        long n_clusters = ((long)(n_cu / R) / 8) * 8;
        if (n_clusters > ((n_groups + 7) / 8) * 8) n_clusters = ((n_groups + 7) / 8) * 8;
        if (n_clusters < 8) continue;
        const long rounds = (n_groups + n_clusters - 1) / n_clusters;
        if (rounds > 64) continue;
        const int MPs = m + 1 <= 3 ? 3 : 5;
        const int cap = (int)(rounds * ipw);
        if (gs_lds_bytes(KU, MPs, NT, cap) > 160 * 1024) continue;
        const double prod_us = 2.0 * NT * (m + 1) * 2 * KU * 64.0 / 2200.0;   // MFMAs per wave x 64 cycles at 2.2 GHz
        const double fix_us = rounds > 1 ? 3.5 : 9.0;   // reduce + publish; a lone group per cluster also exposes its rendezvous and collect
        // (the two-tile instance with source terms at 4 drives is the one the register file cannot quite hold: 35 spilled registers,
        // measured 7 % slower per column than the one-tile instance)
        const double cost = rounds * (prod_us + fix_us) * (has_src && NT == 2 && MPs == 5 ? 1.08 : 1.0);
        if (!found || cost < best) {
            found = true; best = cost;
            out.KU = KU; out.MP = m + 1 <= 3 ? 3 : 5; out.NT = NT; out.ipw = ipw; out.has_src = has_src ? 1 : 0;
            out.n_groups = (int)n_groups; out.n_clusters = (int)n_clusters; out.nblocks = (int)(n_clusters * R);
            out.cap = cap; out.lds_bytes = gs_lds_bytes(KU, out.MP, NT, cap); out.term_us = cost;
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
    a.arrive = arrive; a.Xn = Xn; a.n_groups = pl.n_groups; a.n_clusters = pl.n_clusters; a.cap = pl.cap;
#ifdef GS_STAMP
    a.stamp = gs_stamp_buffer;
#endif
    hipError_t e = hipMemsetAsync(arrive, 0, sizeof(unsigned) * (size_t)((pl.n_groups + 3) / 4 * 4), st);
    if (e != hipSuccess) return e;
#define DTO_GS_CASE(KU_, MP_, NT_) \
    if (pl.KU == KU_ && pl.MP == MP_ && pl.NT == NT_) \
        return pl.has_src ? gs_launch_one<KU_, MP_, NT_, true>(st, a, pl.nblocks, pl.lds_bytes) : gs_launch_one<KU_, MP_, NT_, false>(st, a, pl.nblocks, pl.lds_bytes)
    DTO_GS_CASE(8, 5, 1); DTO_GS_CASE(8, 5, 2); DTO_GS_CASE(8, 3, 1); DTO_GS_CASE(8, 3, 2);
    DTO_GS_CASE(4, 5, 1); DTO_GS_CASE(4, 5, 2); DTO_GS_CASE(4, 3, 1); DTO_GS_CASE(4, 3, 2);
#undef DTO_GS_CASE
    return hipErrorInvalidValue;
}

}  // namespace dto
