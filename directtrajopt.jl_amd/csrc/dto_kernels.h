// dto_kernels.h -- launch interface between the host engine (dto_engine.cpp) and the HIP kernels
// (dto_kernels.hip).  Plain structs passed by value to kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

namespace dto {

// Tuning switches for A/B measurements exist only in builds with -DDTO_TUNING (`make TUNING=1`); the product
// library reads no environment variable.
inline int tune_int(const char* name, int dflt) {
#ifdef DTO_TUNING
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}
inline double tune_double(const char* name, double dflt) {
#ifdef DTO_TUNING
    const char* e = getenv(name);
    return e ? atof(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

constexpr int TAYLOR_M = 16;                 // degree of the matrix Taylor polynomial
constexpr double THETA_16 = 0.78028743;      // backward-error radius of T_16 in double (Al-Mohy & Higham 2011, Table 3.1 method)
constexpr int COEF_STRIDE = 32;              // doubles per interval in the coefficient table
// T_16(B) = sum_{r<=16} B^r/r! with TWO products once B..B^4 are known (they come from the generator subspace):
//   Y = B^4 K(B),   T_16(B) = (Y + Pa(B)) (Y + Pb(B)) + Pc(B),   K, Pa, Pb, Pc of degree <= 4
// (Paterson-Stockmeyer needs three).  Constants from tools/expm_two_product_coeffs.py: the polynomial identity solved to
// 1e-50, factors without cancellation (sum of |terms| / exp = 1.03 at the radius); rounding error at ||B||_1 = 0.78:
// 6e-16 against 2e-16 for the Horner form.  Table layout per interval (each entry times its power of sigma = 2^-s):
constexpr int COEF_PC = 0, COEF_PA = 5, COEF_PB = 10, COEF_K = 15;
constexpr double EXPM2_K[5] = {0.0021247619694343247, 0.00021337327385069214, 1.9238573871783716e-05, 1.748961261071247e-06, 2.1862015763390587e-07};
constexpr double EXPM2_A[5] = {0.15657629060017847, 0.11131927651852432, 0.04459108138499527, 0.00629379053080652, -0.00037523326035530425};
constexpr double EXPM2_B[5] = {6.392946474783064, 1.8714151425525498, 0.2847453199197648, 0.035205562271892026, 0.0005680709246076226};
constexpr double EXPM2_C[5] = {-0.000983845027019532, -0.004677417588383601, -0.03797734224052902, 0.005772376398520401, 0.0016659351281941514};
// THREE products reach order 26 (degree 32): with Ya = Y + Pa, Yb = Y + Pb as above,
//   Y2 = Ya Yb,   L = Y2 + al Y + Pc,   R = Y2 + be Y + Pd,   r(B) = L R + Pe = exp(B) + O(B^27)
// Backward-error radius 2.819 against 0.780 for T_16: one product more buys 1.85 squarings, so the engine takes this form
// whenever the chunk's squaring counts drop by two for (nearly) every interval (alpha above ~3.1).  Of the free parameters
// left at order 26, a4 = c4 = d4 = 0 were imposed: the second product (two outputs, HBM-bound) then streams one matrix
// less.  Constants, radius and the rounding check (1.5e-15 at the radius, i.e. what T_16 shows after two squarings) from
// tools/expm_three_product_coeffs.py.  Table layout: Pe at 0, Pa, Pb, K as above, then the epilogue polynomials of the
// second product in terms of what that launch can read (Ya, not Y): L = Y2 + al Ya + (Pc - al Pa), R likewise.
constexpr double THETA_3P = 2.8188;
constexpr int COEF_L = 20, COEF_R = 26;      // 5 polynomial coefficients + the weight of Ya, each
constexpr double EXPM3_K[5] = {0.00010379876595047617, 6.073095185901804e-06, 3.1522969877676845e-07, -5.57963065580417e-09, 1.6549758371825144e-09};
constexpr double EXPM3_A[5] = {0.06925346208946212, 0.1399878337730723, 0.008815043145043342, -1.5454750321185654e-05, 0.0};
constexpr double EXPM3_B[5] = {7.864692916335445, 1.1432871789700823, 0.08599260924787294, 0.006001949142362251, 0.000114009585419753};
constexpr double EXPM3_C[5] = {1.4279329585415197, -0.648965244154349, -0.10937992097592901, -0.0031202517331481694, 0.0};
constexpr double EXPM3_D[5] = {0.0035917931833685884, -0.8738586720421705, -0.10354040058330033, 0.0006117561678190711, 0.0};
constexpr double EXPM3_E[5] = {-0.0814706004657684, 0.10462167443095102, 0.008296003021444887, 0.0020999533880951024, 0.00010675255813813872};
constexpr double EXPM3_AL = 0.009959087291030108, EXPM3_BE = 3.6322128429901483;
static_assert(EXPM3_A[4] == 0.0 && EXPM3_C[4] == 0.0 && EXPM3_D[4] == 0.0, "the second product's epilogue does not read A^4");
constexpr int PAIR_DCAP = 80;               // Taylor terms the Hessian's pairing path can store per sweep (rows / columns of its Beta table)
constexpr int MAX_TYPES = 36;                // column types of a generator sweep (p, d^i, h^{ij})
constexpr int MAX_DRIVES = 7;

// Problem-level constants every kernel may need.
struct KProb {
    int64_t N, K;       // knots, intervals (K = N-1)
    int32_t z, dt_idx;  // components per knot, timestep component
    int32_t D;          // sum of integrator state dims (Jacobian rows per interval)
    int32_t debug_bad_launch;  // option "debug_bad_launch" (tests of the error path): launches are given an invalid configuration
    int64_t kn_lo;      // first owned knot (0-based)
    int64_t n_knots;    // owned knots
    int64_t n_int;      // owned intervals: knots kn_lo .. kn_lo+n_int-1 (each < K)
    const int64_t* colptr;  // device, global Jacobian column pointers (0-based), n_vars+1
    int64_t jac_lo;     // first Jacobian value owned
    int64_t hess_lo;    // first Hessian value owned
    int64_t grad_lo;    // first gradient entry owned (= kn_lo*z)
    // Hessian entries in global-variable columns (the tail of the CSC order; only host-evaluated Global* terms have any)
    const int64_t* tail_colptr;  // device [gd+1]
    const int64_t* tail_rows;    // device, 0-based global row indices, ascending per column
    int64_t tail_lo;             // local position of the tail's first value in this handle's slab, -1: not owned
};

// position of entry (global row R, global-variable column j) in the local Hessian slab, -1 if absent / not owned
__host__ __device__ inline int64_t hess_pos_tail(const KProb& P, int64_t R, int j) {
    if (P.tail_lo < 0) return -1;
    int64_t lo = P.tail_colptr[j], hi = P.tail_colptr[j + 1];
    while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        if (P.tail_rows[mid] < R) lo = mid + 1; else hi = mid;
    }
    return (lo < P.tail_colptr[j + 1] && P.tail_rows[lo] == R) ? P.tail_lo + lo : -1;
}

// One BilinearIntegrator (src/integrators/bilinear_integrator.jl:61-85) on the device.
struct KBil {
    int32_t n, m, npad;
    int32_t x_off, u_off;
    int32_t pre;          // rows of earlier integrators per interval
    int64_t row_off;      // 0-based global row offset of the integrator (evaluator.jl:213-217)
    int64_t lrow_off;     // offset of its rows inside the shard-local constraint buffer
    const double* G;      // (m+1) padded npad x npad generators, column-major
    const double* GT;     // their transposes
};

struct KDer {  // DerivativeIntegrator (derivative_integrator.jl:26-49)
    int32_t d, x_off, xdot_off, pre;
    int64_t row_off, lrow_off;
};

// position (inside the shard-local value slab) of Jacobian entry (row r of integrator block,
// column `comp` of knot kn); part 0 = rows of interval kn-1, part 1 = rows of interval kn.
__host__ __device__ inline int64_t jac_pos(const KProb& P, const int64_t* colptr, int64_t kn, int comp,
                                           int pre, int d_i, int part, int r) {
    const int has_prev = kn >= 1, has_own = kn < P.K;
    const int cnt = has_prev + has_own;
    return colptr[kn * P.z + comp] + (int64_t)pre * cnt + ((part == 1 && has_prev) ? d_i : 0) + r - P.jac_lo;
}

// position of upper-triangular Hessian entry (a <= b, knot-local comps) of diagonal block kn
// entry (row a of knot kn-1, column b of knot kn) of the off-diagonal block, kn >= 1
__host__ __device__ inline int64_t hess_pos_off(const KProb& P, int64_t kn, int a, int b) {
    const int64_t z = P.z;
    const int64_t tri = z * (z + 1) / 2;
    return tri + (kn - 1) * (z * z + tri) + (int64_t)b * z + (int64_t)b * (b + 1) / 2 + a - P.hess_lo;
}
__host__ __device__ inline int64_t hess_pos(const KProb& P, int64_t kn, int a, int b) {
    const int64_t z = P.z;
    const int64_t tri = z * (z + 1) / 2;
    int64_t colstart;
    if (kn == 0)
        colstart = (int64_t)b * (b + 1) / 2;
    else
        colstart = tri + (kn - 1) * (z * z + tri) + (int64_t)b * z + (int64_t)b * (b + 1) / 2 + z;
    return colstart + a - P.hess_lo;
}

struct TypeDesc {     // one column type of a generator sweep
    int32_t n_extra;  // extra segments G_gen * Z_src * scaleE * mult
    int32_t gen[2];
    int32_t src[2];
    double mult[2];
};

struct SweepTypes {
    int32_t T;
    TypeDesc t[MAX_TYPES];
};

// ------------------------------------------------------------------ launch wrappers (dto_kernels.hip)
struct ChainWork {   // per-chunk workspace of the propagator chain: C matrices npad x npad each
    double* W[9];    // A, A2, A3, A4, then Y+Pa / squaring ping, K / polynomial / squaring pong, Y+Pb, L, R
    double* norms;   // [C][4]
    double* colsum;  // [3][C][npad][npad/64] column abs-sums of A^2..A^4 per 64-row chunk (basis path; added in fixed order)
    double* coef;    // [C][COEF_STRIDE]
    int32_t* s;      // [C] squarings per interval (of the evaluation form in use)
    int32_t* s3;     // [C] squarings per interval with the three-product form
    int32_t* smax;   // [0] max, [1] sum of s over the chunk for the two-product form, [4], [5] for the three-product form
                     // (read back by the host, which then picks the form; [2..3] hold d2max)
    unsigned long long* d2max;  // bit pattern of max_k ||A_k^2||_1^(1/2) over the chunk (exact; plans the sweep)
};

void launch_build_A(hipStream_t st, const KProb& P, const KBil& B, const double* dZ, int64_t int0, int nb, double* A);
void launch_bgemm_plain(hipStream_t st, int npad, int nb, const double* A, const double* Bm, double* C);
void launch_norm1(hipStream_t st, int npad, int nb, const ChainWork& w);
// 1-norm of matrix `which` only; also folds max_k sqrt(norm) into w.d2max (used when the chain is not run)
void launch_norm1_one(hipStream_t st, int npad, int nb, const ChainWork& w, int which);
void launch_expm_params(hipStream_t st, int nb, int s_cap, const ChainWork& w);
// Coefficient table (and w.s) for the evaluation form: 2 = two products (degree 16), 3 = three products (order 26),
// 0 = decided ON THE DEVICE from the squaring counts k_expm_params left in w.smax (three products when they save more than
// 1.5 squarings per interval); the form taken lands in w.smax[6] for the host's readback.
void launch_expm_coef(hipStream_t st, int nb, const ChainWork& w, int form);
void launch_poly_h3(hipStream_t st, int npad, int nb, const ChainWork& w);
// where exp(A_k) of the intervals that need no squaring goes: the x_k columns of the Jacobian slab
struct SlabDest {
    KProb P;
    KBil B;
    int64_t int0;
    double* vals;
};
// W[dst] = W[srcA] W[srcB] + poly(coef_base);  dst2 >= 0: also W[dst2] = W[srcA] W[srcB] + poly(coef_base2);
// with_srcA: each polynomial has a sixth coefficient, the weight of W[srcA] itself;
// slab (single-output form, the LAST polynomial product): intervals with s_k = 0 store -result into the Jacobian slab
void launch_bgemm_poly(hipStream_t st, int npad, int nb, const ChainWork& w, int srcA, int srcB, int dst, int coef_base,
                       int dst2, int coef_base2, bool with_srcA = false, const SlabDest* slab = nullptr);
void launch_bgemm_square(hipStream_t st, int npad, int nb, const ChainWork& w, int src, int dst, int it,
                         const KProb& P, const KBil& B, int64_t int0, double* vals);

// ---- the whole propagator chain of a 33..64-state integrator in ONE launch (dto_chain64.hip): a workgroup per interval keeps
// A .. A^4, the polynomial factors and the squarings in LDS and registers, chooses the evaluation form per interval and stores
// -E_k into the slab; norms [n_int][4] (INF, ||A^2||, ||A^3||, ||A^4||: k_hump's input), smax[0] = largest squaring count,
// d2max = max_k min(d2, max(d3, d4)) (bit pattern), sk [n_int] squarings used
hipError_t chain64_prepare();
hipError_t launch_chain64(hipStream_t st, const KProb& P, const KBil& B, const double* dZ, double* vals, double* norms, int32_t* smax,
                          unsigned long long* d2max, int32_t* sk, int s_cap, int force_form, int n_cu);

struct SweepBuf {      // generator sweep ("expmv") workspace for one bilinear integrator
    int32_t npad, Kpad, TN;   // Kpad multiple of TN
    double* Z[2];      // [T][Kpad][npad] ping-pong terms
    double* S;         // [T][Kpad][npad] sums
    double* GY;        // [Kpad][npad]   G(u) * S0
    double* W;         // [(m+1)][Kpad][npad]  G_j' mu   (Hessian)
    double* scaleA;    // [(m+1)][Kpad]  dt*ubar_j/q
    double* scaleU;    // [(m+1)][Kpad]  ubar_j
    double* scaleE;    // [2][Kpad]      dt/q and 2*dt/q (the i == j second-order terms)
    unsigned long long* termnorm;  // [3][T][Kpad]
    unsigned long long* sumnorm;   // [T][Kpad]
    int32_t* active;   // [Kpad/TN]
    int32_t* stats;    // [0] active blocks after the last check, [1] terms used (max)
    // term store (Hessian pairing path): every Taylor term of the sweep is kept, [dcap][T][Kpad][npad]
    double* Zt;
    int32_t dcap, T_alloc;  // T_alloc: column types the Z/S buffers were sized for
    int32_t* nterms;   // [Kpad/TN] number of valid terms of a converged column block (0: use all launched)
    // option reuse_forward_sweep: the Taylor terms of the p column kept by an earlier callback at the same point
    // ([dcap][Kpad][npad] at the head of Zt) let a later sweep run its tangent columns alone (types first_type..T-1)
    int32_t* nterms_p;       // [Kpad/TN] valid stored p terms per column block (0: all frozen_total)
    const double* frozen;    // stored p terms (nullptr: the sweep computes its own p column)
    int32_t frozen_total;    // terms 0 .. frozen_total-1 were stored
    int32_t first_type;      // first column type the sweep computes (1 with frozen p terms, else 0)
    int32_t nblk;            // intervals per entry of nterms / nterms_p (the last sweep's convergence blocks: TN for the
                             // step-per-launch form, the workgroup's interval count for the fused form)
};

// src_kind: 0 = state x_k of the integrator (forward sweep), 1 = multipliers mu_k (adjoint sweep)
void launch_sweep_init(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, const SweepTypes& ty,
                       const double* dZ, const double* dmu, int src_kind, int q);
void launch_sweep_restart(hipStream_t st, const SweepBuf& w, int T);
// term 0 (and sum) of column type `type` := v[kn*z + x_off + :]  (a second start vector, Z layout): exp(A) v rides
// along the same sweep as an extra column type
void launch_sweep_set_type(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, int T, int type, const double* v);
// matrix-free Jacobian-vector products (evaluator.jl:406-456 without materialising J)
//   Jw:  y[rows of the integrator] = -exp(A)w_x - sum_j c_j w_uj - G(u)y w_dt + w_x(k+1)   (type_ew = column type of exp(A) w_x)
//   Jtw: y[knot entries] += -exp(A')w_k (+ w_{k-1}), -c_j'w_k, -(G(u)y)'w_k                 (ad.S[0] = exp(A') w_k)
void launch_jv_bilinear(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& fw, int type_ew, const double* w, double* y);
void launch_jtv_bilinear(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& fw, const SweepBuf& ad, const double* w, double* y);
void launch_jv_derivative(hipStream_t st, const KProb& P, const KDer& D, const double* dZ, const double* w, double* y, int transpose);

// ---- the whole sweep in one persistent launch (dto_sweep_fused.hip): a workgroup owns `ipw` intervals, all rows, all types
struct FusedSweepPlan {
    int MT, NT, ipw, nslot, nblocks;
    int WC = 1;  // column groups of wavefronts (4 WC wavefronts per workgroup), NT column tiles per group
    int WK = 1;  // 2: two wavefronts per SIMD split the K loop of a wave tile (256 states)
    int S64 = 0; // 1: the generator-stationary 64-state form (k_sweep_s64), NX = most inhomogeneous sources of a column type
    int NX = 0;
    size_t lds_bytes;
};
hipError_t sweep_fused_prepare();  // per-device opt-in to the kernels' dynamic LDS (dto_create)
bool sweep_fused_plan(int npad, int m, const SweepTypes& ty, int64_t n_int, int n_cu, FusedSweepPlan& out, bool shared_chip = false);
// Runs q rounds of at most d_ub Taylor steps (termination test from step tc on, per workgroup), terms into w.Zt (store) or
// ping-pong w.Z[0/1], sums into w.S, scale factors into w.scale*, valid term counts into w.nterms[workgroup],
// w.stats[0] += workgroups that did not converge, w.stats[1] = max terms used (the caller zeroes w.stats).
hipError_t launch_sweep_fused(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, const SweepTypes& ty,
                              const FusedSweepPlan& pl, const double* dZ, const double* dmu, int src_kind, int transposed,
                              int q, int d_ub, int tc, bool store, double tol, const int32_t* plan_dev = nullptr);
// {q, d_ub, tc} of a sweep from the norm bound (bit pattern in bounds[0]), computed on the device: launch_sweep_fused(..., plan_dev)
void launch_plan_dev(hipStream_t st, const unsigned long long* bounds, int32_t* out);

// ---- the same sweep with the rows of the matrix split over a cluster of R workgroups that exchange their slices of every new
// term through global memory (dto_sweep_fused.hip): short shards, single-column sweeps, 512+ states
struct ClusterSweepPlan {
    int MT, NT, R, ipw, n_groups, n_clusters, nblocks;
    size_t lds_bytes;
    double step_us;  // the cost model's time per Taylor step and round of clusters
};
hipError_t sweep_cluster_prepare();
bool sweep_cluster_plan(int npad, int m, const SweepTypes& ty, int64_t n_int, int n_cu, ClusterSweepPlan& out);
size_t sweep_cluster_workspace_doubles(int npad, const ClusterSweepPlan& pl);
hipError_t launch_sweep_cluster(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, const SweepTypes& ty,
                                const ClusterSweepPlan& pl, double* X, unsigned* arrive, const double* dZ, const double* dmu,
                                int src_kind, int transposed, int q, int d_ub, int tc, bool store, double tol);

// ---- the sweep with the generators STATIONARY in registers (dto_sweep_gs.hip, round 4): a cluster of npad / 32 workgroups shares an
// interval group, each member holding 32 rows of every generator for the whole launch; only the term slices move (through the
// sweep's own term slabs, sc1 on both sides).  128 and 256 states, at most 4 drives, no sub-stepping (q = 1).
struct GsSweepPlan {
    int KU, MP, NT, ipw, has_src, n_groups, n_clusters, nblocks, cap;
    size_t lds_bytes;
    double term_us;  // the cost model's time per Taylor term
};
hipError_t sweep_gs_prepare();
bool sweep_gs_plan(int npad, int m, const SweepTypes& ty, int64_t n_int, int n_cu, GsSweepPlan& out);
size_t sweep_gs_norm_doubles(const GsSweepPlan& pl);   // exchange slab of the partial column norms
// arrive: n_groups (rounded up to 4) counters, zeroed by the launch; terms into w.Zt (store) or ping-pong w.Z[0/1], sums into w.S,
// valid term counts into w.nterms[group], w.stats as launch_sweep_fused
hipError_t launch_sweep_gs(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, const SweepTypes& ty, const GsSweepPlan& pl,
                           double* Xn, unsigned* arrive, const double* dZ, const double* dmu, int src_kind, int transposed, int d_ub,
                           int tc, bool store, double tol);

void launch_sweep_step(hipStream_t st, const KBil& B, const SweepBuf& w, const SweepTypes& ty, int transposed,
                       int t, int in_buf, int split_store = 0);
void launch_sweep_check(hipStream_t st, const SweepBuf& w, int T, int t, double tol);
// start of a sweep over types first_type..T-1 only: their terms, sums and norms are cleared, type 0 keeps its sums
void launch_sweep_init_tangents(hipStream_t st, const SweepBuf& w, int T);
// out[j] = G_j (or G_j') * V for every generator j (no summation): out [(m+1)][Kpad][npad]
void launch_apply_generators(hipStream_t st, const KBil& B, const SweepBuf& w, int transposed, const double* V,
                             double* out);
// out[g] = G_{gen_first+g} * V over `cols` columns (cols multiple of the sweep tile): the pairing path's E_j * terms
void launch_apply_generators_cols(hipStream_t st, const KBil& B, const SweepBuf& w, int transposed, const double* V,
                                  double* out, int gen_first, int gen_count, int64_t cols, int64_t seg_cols = 0,
                                  int64_t seg_stride = 0);
// U[a][type][k][:] = sum_b Btab[a][b] * terms[b][type][k][:]   (Beta-function weights of the pairing formula)
void launch_pair_combine(hipStream_t st, const SweepBuf& ad, int T, int n_types, int nf_used, int na_used,
                         const int32_t* nterms_f, int nblk_f, const double* Btab, double* U);
// (u_i,u_j) block of the bilinear Hessian from the pairing formula (see k_hess_pair)
void launch_hess_pair(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& fw, int nf_used, const double* EP,
                      double* H);
// GY = sum_j ubar_j G_j (or G_j') * V
void launch_apply_Gu(hipStream_t st, const KBil& B, const SweepBuf& w, int transposed, const double* V, double* out);

void launch_cons_bilinear(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, const double* dZ, double* g);
void launch_cons_derivative(hipStream_t st, const KProb& P, const KDer& Dv, const double* dZ, double* g);
void launch_jac_bilinear(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& w, double* vals);
void launch_jac_derivative(hipStream_t st, const KProb& P, const KDer& Dv, const double* dZ, double* vals);

struct KCon {  // NonlinearKnotPointConstraint with a built-in g
    int32_t kind, n_comps;
    double c;
    const int32_t* comps;    // device [n_comps]
    const int64_t* times;    // device, owned 0-based knots [n_times]
    const int64_t* lrow;     // device, local row of each owned time in the shard-local g buffer
    const int64_t* jpos;     // device [n_times][n_comps] local Jacobian positions (-1: not in pattern)
    int64_t n_times;
    int64_t mu_off;          // global 0-based row of the constraint's first row (for mu), evaluator.jl:219-223
    const int64_t* tidx;     // device, index of each owned time inside the constraint's full `times`
    const int32_t* hess_on;  // device, 0 where a LATER entry of `times` names the same knot: the reference's
                             // ForwardDiff.hessian! into the block view overwrites (knot_point_constraint.jl:285-291)
    int32_t g_dim, external; // outputs per listed time (1 for the built-in kinds); external: values come from the host
    int32_t repeats, comp_repeats;  // the owned `times` name a knot twice / `comps` name a component twice (J' w then adds in listing order)
};
void launch_cons_knot(hipStream_t st, const KProb& P, const KCon& C, const double* dZ, double* g);
void launch_jv_knot(hipStream_t st, const KProb& P, const KCon& C, const double* dZ, const double* w, double* y, int transpose);
void launch_jac_knot(hipStream_t st, const KProb& P, const KCon& C, const double* dZ, double* vals);
void launch_hess_knot(hipStream_t st, const KProb& P, const KCon& C, const double* dZ, const double* dmu, double* H);

// host-evaluated integrator (DTO_INTEGRATOR_EXTERNAL): placement of the caller's per-interval blocks
struct KExtInt {
    int32_t d, pre;        // rows per interval; rows of the integrators before it (per interval)
    int64_t row_off, lrow_off;
};
void launch_extint_cons(hipStream_t st, const KProb& P, const KExtInt& E, const double* vals, double* g);
void launch_extint_jac(hipStream_t st, const KProb& P, const KExtInt& E, const double* blocks, double* vals);
void launch_extint_hess(hipStream_t st, const KProb& P, const KExtInt& E, const double* blocks, double* H);

// TimeDependentBilinearIntegrator with a parametrised generator family, evaluated on the device (dto_tdb.hip): fills the same
// per-interval blocks as a host-evaluated integrator, which k_extint_* then place
struct KTdb {
    int32_t n, m, x_off, u_off, t_off, order, substeps, nmod;
    int64_t row_off;             // 0-based global row offset of the integrator (for mu)
    const double* G;             // (m+1) compact n x n matrices, column-major
    const double* H;             // [nmod][(m+1)] compact n x n matrices
    const int32_t* mod_kind;     // [nmod] 1 = cos(omega t), 2 = sin(omega t)
    const double* mod_omega;     // [nmod]
};
bool tdb_supported(const KTdb& T);
size_t tdb_scratch_doubles(const KTdb& T, int need);
// blocks of intervals i_lo .. i_lo + count - 1 (global, 0-based) into vals [K][n], jac [K][2z][n], hess [K][2z][2z]
hipError_t launch_tdb(hipStream_t st, const KProb& P, const KTdb& T, const double* dZ, const double* dmu, int need, int64_t i_lo,
                      int64_t count, double* vals, double* jac, double* hess, double* scratch, size_t scratch_stride);

// host-evaluated knot terms (DTO_CONSTRAINT_EXTERNAL / DTO_OBJECTIVE_EXTERNAL_KNOT): scatter of caller-supplied blocks
void launch_ext_cons(hipStream_t st, const KCon& C, const double* vals, double* g);
void launch_ext_jac(hipStream_t st, const KCon& C, const double* blocks, double* vals);
// One host-evaluated term's placement data: per listing i the variables [z_t[comps] ; global_data[gcomps]]
// (times[i] == N: no knot part).  `knot_on` / `glob_on`: this handle places the listing's knot part / its entries in
// global-variable columns; `count`: this handle adds the listing's value to the objective.
struct KExtTerm {
    int32_t nc, ng;
    const int32_t* comps;    // device [nc]
    const int32_t* gcomps;   // device [ng]
    const int64_t* times;    // device [n_list] 0-based knots
    const int64_t* tidx;     // device [n_list] index of the listing in the caller's arrays
    const int32_t* knot_on;  // device [n_list]
    const int32_t* count;    // device [n_list]
    int32_t glob_on, pad;
    int64_t n_list;
};
void launch_ext_hess(hipStream_t st, const KProb& P, const KExtTerm& E, double scale, const double* blocks, double* H);
void launch_ext_objective(hipStream_t st, const KExtTerm& E, double weight, const double* vals, double* f);
void launch_ext_gradient(hipStream_t st, const KProb& P, const KExtTerm& E, double weight, const double* blocks, double* grad);

struct KObj {  // objective term
    int32_t kind, comp_off, comp_dim, has_baseline;
    double weight, D;
    const double* R;         // device [comp_dim]
    const double* baseline;  // device [comp_dim x N] or null
    const int64_t* times;    // device, owned 0-based knots
    int64_t n_times;
    // KNOT_SQDIST (kind 4): l(v, p) = ||v - p||^2 on a component list, per-time targets and weights
    const int32_t* comps;    // device [n_comps]
    int32_t n_comps, pad1;
    const double* params;    // device [n_times][n_comps] (owned times) or null
    const double* Qs;        // device [n_times]
    const int32_t* last;     // device [n_times]: 0 where a later listed time names the same knot -- the reference's
                             // gradient!/hessian! overwrite per listed time (knot_point_objectives.jl:198, 235)
};
void launch_objective(hipStream_t st, const KProb& P, const KObj& O, const double* dZ, double* partial, double* f);
void launch_gradient(hipStream_t st, const KProb& P, const KObj& O, const double* dZ, double* grad);
void launch_hess_objective(hipStream_t st, const KProb& P, const KObj& O, const double* dZ, double sigma, double* H);
void launch_hess_derivative(hipStream_t st, const KProb& P, const KDer& Dv, const double* dmu, double* H);
void launch_hess_bilinear(hipStream_t st, const KProb& P, const KBil& B, const SweepBuf& fw, const SweepBuf& ad,
                          const double* dmu, double* H, int with_uu);

void launch_fill(hipStream_t st, double* p, int64_t n, double v);
void launch_zero_runs(hipStream_t st, const int64_t* start, const int64_t* len, int64_t n_runs, double* dst);

// Small-state path (dto_small.hip): one wavefront per interval, everything in LDS.  mode bits: 1 constraint
// values, 2 Jacobian block, 4 Hessian block.  Gs = compact (m+1) x n x n generators.
size_t small_lds_bytes(int n, int m, int T_fw, int T_ad);
hipError_t small_prepare(size_t lds_bytes);  // per-device opt-in to > 64 KB of dynamic LDS (dto_create)
hipError_t launch_small(hipStream_t st, const KProb& P, const KBil& B, const double* Gs, const SweepTypes& ty_fw,
                        const SweepTypes& ty_ad, const double* dZ, const double* dmu, double* cons, double* jac, double* hess,
                        int mode);

// y = J w / y = J' w from the value slab in CSC order (A3: evaluator.jl:406-456; the reference also
// materialises the Jacobian values first, on the host).  One wavefront per column.
struct KIntegTable {
    int32_t n;
    int32_t d[8];
    int64_t off[8];
};
void launch_jac_spmv(hipStream_t st, const KProb& P, const KIntegTable& T, const int64_t* conbase, const int64_t* con_rows,
                     const double* vals, const double* w, double* y, int transpose, int64_t global_cols = 0);
// y = J w from the value slab row by row (fixed summation order; the transpose takes launch_jac_spmv, one wavefront per column)
void launch_jac_rowgather(hipStream_t st, const KProb& P, const KIntegTable& T, int64_t n_con_rows, const int64_t* rptr, const int64_t* rcol,
                          const int64_t* rpos, int64_t n_dyn, const double* vals, const double* w, double* y);
void launch_add(hipStream_t st, double* dst, const double* src, int64_t n);  // dst += src
// *flag = 0 if a and b differ in any bit (flag preset to 1 by the caller)
void launch_bits_equal(hipStream_t st, const double* a, const double* b, int64_t n, int32_t* flag);

// Powers of A_k from the generator subspace: A_k = dt*sum_j ubar_j G_j lives in an (m+1)-dimensional
// matrix space, so A_k^r = sum over multisets alpha of size r of (dt^r prod ubar_alpha) * S_alpha with
// S_alpha = sum of the distinct orderings of the product G_alpha1 ... G_alphar, shared by all knots.
// One GEMM  [vec(S_alpha)] (npad^2 x cnt) x coef (cnt x intervals)  then yields A_k^r for every
// interval of the chunk -- 2*npad^2*cnt flops per interval instead of 2*npad^3.
struct BasisSet {
    int32_t r, cnt, cntpad;     // degree, number of multisets, padded to 16
    const double* S;            // device [cntpad][npad^2]
    const int32_t* idx;         // device [cnt][r] generator indices of each multiset (-1 = unused slot)
    double* coef;               // device [capacity][cntpad] per-interval coefficients
};
void launch_basis_coef(hipStream_t st, const KProb& P, const KBil& B, const BasisSet& bs, const double* dZ,
                       int64_t int0, int nb, int nbpad, const double* taylor);
void launch_jac_zero(hipStream_t st, const KProb& P, const KBil& B, double* vals);
// colsum (nullable): [nb][npad] accumulators of the column abs-sums of every output matrix (zeroed by
// the caller); the exact 1-norms then cost no extra pass over the matrices.
void launch_basis_gemm(hipStream_t st, int npad, int nb, int nbpad, const BasisSet& bs, double* out, double* colsum);
// the same for several basis sets in ONE launch (A^2, A^3, A^4): their tiles are interleaved, so the write-bound sets
// (few multisets, short K loop) overlap with the MFMA-bound ones instead of running one after the other
void launch_basis_gemm_multi(hipStream_t st, int npad, int nb, int nbpad, int nsets, const BasisSet* bs, double* const* out,
                             double* const* colsum);
// coefficients of several basis sets in one launch
void launch_basis_coef_multi(hipStream_t st, const KProb& P, const KBil& B, int nsets, const BasisSet* bs, const double* dZ,
                             int64_t int0, int nb, int nbpad);
// norms[b*4 + 1 + r] = max column sum of set r, r < nsets (one launch)
void launch_norm_from_colsum_multi(hipStream_t st, int npad, int nb, int nsets, double* const* colsum, double* norms);
// norms[b*4 + which] = max_c colsum[b][c]
void launch_beta_from_norms(hipStream_t st, int nb, const ChainWork& w);
void launch_hump(hipStream_t st, const KProb& P, const KBil& B, const double* dZ, const double* g1, int64_t int0, int nb,
                 const double* norms, unsigned long long* out);
void launch_norm_from_colsum(hipStream_t st, int npad, int nb, const double* colsum, double* norms, int which,
                             unsigned long long* d2max = nullptr);
// out2[0] = max_k min(b1_k, b2_k), out2[1] = max_k b1_k (bit patterns of non-negative doubles), where
// b1_k >= ||A_k||_1 and b2_k >= ||A_k^2||_1^(1/2) follow from the generator norms g1[j] = ||G_j||_1,
// n2[i][j] = ||G_i G_j||_1 and the triangle inequality.
void launch_norm_bounds(hipStream_t st, const KProb& P, const KBil& B, const double* dZ, const double* g1,
                        const double* n2, unsigned long long* out2);

}  // namespace dto
