// dto_engine.cpp -- host side of the C ABI declared in include/dto_engine.h.
//
// Owns the handle, builds the sparsity structure in the reference's exact order
// (src/solvers/evaluator.jl:119-209, closed forms of SURVEY.md §3.6 + a per-column prefix sum for
// the value-dependent constraint entries), and orchestrates the HIP kernels of dto_kernels.hip.
// There is NO CPU fallback: every evaluation runs on the GPU or fails with an error code.
#include "../../include/dto_engine.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <string>
#include <vector>

#include "dto_comm.h"
#include "dto_hostxfer.h"
#include "dto_kernels.h"

#include <memory>

using namespace dto;

namespace {

thread_local std::string g_create_error;

struct HipError {
    std::string msg;
};

#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            char buf_[512];                                                                      \
            snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                        \
            throw HipError{buf_};                                                                \
        }                                                                                        \
    } while (0)

template <class T>
T* dalloc(size_t n) {
    void* p = nullptr;
    if (n == 0) n = 1;
    HIP_CHECK(hipMalloc(&p, n * sizeof(T)));
    return static_cast<T*>(p);
}
template <class T>
T* dupload(const std::vector<T>& v) {
    T* p = dalloc<T>(v.size());
    if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}

struct BilHost {
    KBil k;
    SweepBuf fw{}, ad{};
    int T_alloc = 0;
    std::vector<double> g1;   // ||G_j||_1
    std::vector<double> n2;   // ||G_i G_j||_1, (m+1)^2
    double* d_g1 = nullptr;
    double* d_n2 = nullptr;
    ChainWork chain{};
    int chain_cap = 0;
    unsigned long long* d_hump = nullptr;  // [8] k_hump output: log hump(q), last term index, q = 1..4 (max over intervals)
    double hump_logH[4] = {0, 0, 0, 0};
    int hump_kend[4] = {0, 0, 0, 0};
    bool hump_valid = false;
    // reuse_forward_sweep: what b.fw still holds for the cached Z -- 0 nothing, 1 the p sums (S type 0), 2 p and d^j sums
    // and GY, 3 additionally every Taylor term in fw.Zt (cache_steps of them)
    int cache_kind = 0, cache_steps = 0;
    // option reuse_forward_sweep: the step budget the Jacobian's chain planned from its exact norms at the cached point (q = 0: none)
    int plan_q = 0, plan_dub = 0;
    // ... and whether the Taylor terms of the p column of that point sit in fw.Zt ([term][Kpad][npad], one type per term:
    // eval_constraint and the Hessian's forward sweep store them), p_steps + 1 of them, valid counts per block in fw.nterms_p
    bool p_terms = false;
    int p_steps = 0;
    int p_nblk = 0;           // intervals per entry of fw.nterms_p (the convergence blocks of the sweep that stored the p terms)
    // row-split cluster sweeps (dto_sweep_fused.hip): exchange slabs and arrival counters, one set per sweep buffer (the
    // Hessian's forward and adjoint sweeps may run side by side), grow-only
    double* xch[2] = {nullptr, nullptr};
    unsigned* xch_arrive[2] = {nullptr, nullptr};
    size_t xch_cap[2] = {0, 0};
    int xch_clusters[2] = {0, 0};
    // generator-stationary sweeps (dto_sweep_gs.hip): partial-norm slabs and arrival counters per sweep buffer, grow-only
    double* gs_xn[2] = {nullptr, nullptr};
    unsigned* gs_arrive[2] = {nullptr, nullptr};
    size_t gs_xn_cap[2] = {0, 0};
    int gs_groups[2] = {0, 0};
    bool small = false;       // n <= 32: fused one-workgroup-per-interval path (dto_small.hip)
    double* d_Gs = nullptr;   // compact generators for that path
    bool use_basis = false;   // A^2..A^4 from the generator subspace instead of three batched GEMMs
    BasisSet basis[3]{};      // degrees 2, 3, 4
    BasisSet basis_all{};     // every multiset of degree 0..4: one GEMM gives the factor K of the two-product Taylor form
    // Hessian pairing path: stored Taylor terms of both sweeps, E_j * forward terms, Beta-weighted adjoint sums
    bool pairing = false;
    double* EP = nullptr;
    double* Upair = nullptr;
    double* d_Btab = nullptr;
};

struct ConHost {
    KCon k{};
    int equality = 0;
    int64_t n_times_total = 0;
    int64_t row_off = 0;  // global 0-based first row
    std::vector<int64_t> times0;  // all times (0-based knots), reference order
    std::vector<int32_t> comps;
    int g_dim = 1;
    bool external = false;        // DTO_CONSTRAINT_EXTERNAL[_GLOBAL]: values/Jacobian/Hessian blocks come from dto_set_external
    bool global = false;          // ..._GLOBAL: `comps` index global_data; the single listing sits at the pseudo-knot N
    int ext_slot = -1;
    std::vector<double> jac0;     // external: Jacobian blocks at Z0 (pattern)
    std::vector<double> hess0;    // global: Hessian of sum(g) at Z0 (pattern)
    KExtTerm xk{};                // external: placement of the Hessian blocks
};

// DTO_OBJECTIVE_EXTERNAL_KNOT / _GLOBAL: placement data of a host-evaluated objective term
struct ExtObjHost {
    double weight = 1.0;
    bool global = false;
    std::vector<int32_t> comps, gcomps;
    std::vector<int64_t> times0;  // listed knots, 0-based; {N} for a GlobalObjective (no knot part)
    KExtTerm k{};
    int ext_slot = -1;
};

// one host-evaluated term's values for the coming callbacks + grow-only device staging
struct ExtSlot {
    dto_external_values v{nullptr, nullptr, nullptr};
    size_t len[3] = {0, 0, 0};   // doubles per array (all listed times)
    double* d[3] = {nullptr, nullptr, nullptr};
};

// DTO_INTEGRATOR_TIME_DEPENDENT_BILINEAR: evaluated by dto_tdb.hip into per-interval blocks, placed like an external integrator
struct TdbHost {
    KTdb k{};
    KExtInt place{};
    double *d_vals = nullptr, *d_jac = nullptr, *d_hess = nullptr, *d_scratch = nullptr;
    size_t stride = 0;
};

struct ProfRec {
    hipEvent_t a, b;
    int cat;
    double flops;
};

}  // namespace

struct dto_handle {
    std::string err;
    int device = 0;
    bool structure_only = false;  // created with device < 0: sizes, structure and shard queries only
    int64_t N = 0, K = 0;
    int z = 0, gd = 0, dt_idx = 0, D = 0;
    int eval_hessian = 1;
    int64_t n_vars = 0, n_cons = 0, n_dyn = 0, jac_nnz = 0, hess_nnz = 0;
    int64_t k_lo = 1, k_hi = 1;
    KProb P{};
    std::vector<int64_t> colptr;            // host copy
    std::vector<int64_t> con_cols, con_rows;  // constraint pattern entries sorted by (col,row), 0-based
    std::vector<int64_t> con_colstart;      // index into con_* of each column with entries (map col -> range)
    std::vector<BilHost> bil;
    std::vector<KDer> der;
    std::vector<int> integ_kind, integ_index;  // reference order -> (kind, index into bil/der)
    std::vector<int> integ_dim;
    std::vector<int64_t> integ_row_off;
    std::vector<ConHost> con;
    std::vector<KObj> obj;
    struct ObjInfo {  // host copy of what a built-in objective term touches in the Hessian (the D2H plan needs it)
        int kind, comp_off, comp_dim;
        std::vector<int32_t> comps;
        std::vector<int64_t> times;  // owned, 0-based
        // The term's listings are stored in LAYERS: layer l holds the (l+1)-th listing of every knot, so that inside one layer
        // no two listings name the same knot.  Gradient and Hessian kernels run layer by layer (one launch each; a single
        // layer unless `times` repeats a knot): contributions to one entry are added in listing order, never concurrently.
        std::vector<int64_t> layer_start;  // [n_layers + 1] offsets into the (layer-sorted) listing arrays
    };
    std::vector<ObjInfo> obj_info;
    // host-pointer entry points: only the entries that can change cross PCIe (dto_hostxfer.h); built lazily at the first
    // host-pointer Jacobian / Hessian call, option "host_xfer" = 0 keeps the plain whole-slab copy
    std::unique_ptr<HostXfer> xfer;
    XferPlan jac_plan, hess_plan;
    int xfer_cap = 0;                                         // host-pointer Jacobian: chain chunk (intervals) for the early hand-over, 0 = off
    std::function<void(int64_t, int)> on_chain_chunk;         // ... and its hook: (first local interval, count) of a finished chunk
    bool plans_built = false;
    bool raw_plans_built = false;
    // dto_bind_output_dev: a device buffer the caller passes again and again (MadNLP's value vectors in GPU mode).  Once a call
    // has written it in full ("primed"), later calls into the SAME pointer leave the call-invariant entries alone -- structural
    // zeros, the identity z_{k+1} halves: half of a Jacobian slab, 99 % of a Hessian slab -- and clear only the runs a kernel
    // accumulates into.  [0] Jacobian, [1] Hessian.
    double* bound[2] = {nullptr, nullptr};
    bool primed[2] = {false, false};
    int64_t* d_bind_start[2] = {nullptr, nullptr};
    int64_t* d_bind_len[2] = {nullptr, nullptr};
    int64_t n_bind_runs[2] = {0, 0};
    bool bind_ready[2] = {false, false};
    int host_xfer = 1;
    int xfer_check = 0;  // option "host_xfer_check": every host-pointer Jacobian / Hessian is compared with the whole device slab
    std::vector<ExtObjHost> ext_obj;
    std::vector<KExtInt> ext_int;  // DTO_INTEGRATOR_EXTERNAL, slot = index
    std::vector<TdbHost> tdb;      // DTO_INTEGRATOR_TIME_DEPENDENT_BILINEAR
    std::vector<ExtSlot> ext;      // external integrators, then constraints, then objectives, each in list order
    int n_ext_int = 0, n_ext_con = 0, n_ext_obj = 0;
    std::vector<int64_t> tail_colptr, tail_rows;  // Hessian entries in global-variable columns (CSC tail)
    int64_t hess_block_nnz = 0;
    std::vector<std::pair<int64_t, int64_t>> row_segments;  // (global start 0-based, len)
    int64_t cons_len = 0;
    dto_shard_info info{};

    // device scratch
    int64_t* d_colptr = nullptr;
    double* d_Z = nullptr;
    double* d_mu = nullptr;
    double* d_out = nullptr;  // host-API staging for the largest output
    size_t d_out_cap = 0;
    double* d_partial = nullptr;
    double* d_f = nullptr;
    double* d_bounds = nullptr;  // [2] max beta, max b1 (as uint64 bit patterns)
    int32_t* d_plan = nullptr;   // [4] {q, d_ub, tc} of a sweep planned on the device from d_bounds (launch_plan_dev)
    double* d_jac_scratch = nullptr;     // value slab for the Jacobian-vector products (lazy)
    double* d_w = nullptr;               // product input
    int64_t* d_conbase = nullptr;        // [n_vars+1] first constraint-pattern entry of each column
    int64_t* d_crow_ptr = nullptr;       // the constraint pattern in row order (J w gathers rows): [rows+1], columns, slab positions
    int64_t* d_crow_col = nullptr;
    int64_t* d_crow_pos = nullptr;
    int64_t* d_con_rows = nullptr;       // constraint-pattern rows, (col,row) order
    double* h_pinned = nullptr;  // [32]: 0-1 bounds, 2-3 chain scalars, 6 sweep stats, 16-23 hump readback, 28 deferred squaring count
    bool smax_pending = false;   // the one-launch chain's squaring count lands in h_pinned[28] behind ev_done (read by check_sweeps)
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // generator sweep runs here, concurrently with the propagator chain
    hipStream_t stream_rb = nullptr; // the chain's 96-byte readback (evaluation form, squaring counts, hump bound) leaves on this one
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_stats = nullptr, ev_chain = nullptr, ev_rb = nullptr, ev_zero = nullptr;

    bool reuse = false;          // option reuse_forward_sweep
    double* d_Zcache = nullptr;  // the Z the cached sweeps belong to
    int32_t* d_eq = nullptr;
    bool profiling = false;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;  // recycled timing events (creating them inside the timed region costs host time)
    int last_smax = 0, last_terms = 0;
    int expm_form = 0;  // option "expm_form": 0 = by cost, 2 / 3 = forced
    int overlap_sweep = 1;  // option "overlap_sweep": the Jacobian's generator sweep on a second stream next to the chain's products
    int sweep_form = 0;   // option "sweep_form": 0 = fused persistent sweep where it applies, 1 = step-per-launch form only
    int chain_form = 0;   // option "chain_form": 0 = the one-launch chain of 33..64-state integrators where it applies, 1 = batched-GEMM launches only
    int n_cu = 256;
    int chain_chunk = 0;  // option "chain_chunk": upper bound on the intervals per chain chunk (0: workspace capacity)
    int deterministic = 0;  // option "deterministic": results independent of overlap_sweep and of the entry-point family
    // deferred errors of the `*_dev` entry points (dto_engine.h, error convention): the sweep statistics of the last
    // asynchronous call are copied to pinned memory behind its kernels and looked at by the next call through the ABI
    hipEvent_t ev_done = nullptr;
    bool stats_pending = false;
    int32_t* h_stats = nullptr;  // pinned [2 * bilinear integrators][2]
    int last_form = 0;
    // multi-GPU (dto_comm.h): the knot ranges of the communicator's ranks (dto_comm_create exchanges them over RCCL,
    // dto_comm_set_ranges takes them from a caller with a transport of its own), the gather plans that follow from them,
    // and the communicator itself
    std::unique_ptr<Comm> comm;
    int comm_rank = -1;
    std::vector<std::pair<int64_t, int64_t>> rank_ranges;               // (k_lo, k_hi) per rank, 1-based inclusive
    GatherPlan gather_plan[4];                                           // DTO_VECTOR_* - 1
    std::vector<Slab> cons_segments;                                     // every rank's row segments of g ...
    std::vector<int> cons_segment_root;                                  // ... and the rank that owns each
    int64_t* d_ranges = nullptr;
    std::vector<void*> owned;  // device allocations to free

    ~dto_handle();
};

dto_handle::~dto_handle() {
    if (structure_only) return;
    (void)hipSetDevice(device);
    comm.reset();
    for (auto& r : prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto& e : ev_pool) (void)hipEventDestroy(e);
    for (void* p : owned) (void)hipFree(p);
    if (h_pinned) (void)hipHostFree(h_pinned);
    if (stream) (void)hipStreamDestroy(stream);
    if (stream2) (void)hipStreamDestroy(stream2);
    if (stream_rb) (void)hipStreamDestroy(stream_rb);
    if (ev_rb) (void)hipEventDestroy(ev_rb);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_zero) (void)hipEventDestroy(ev_zero);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (ev_stats) (void)hipEventDestroy(ev_stats);
    if (ev_chain) (void)hipEventDestroy(ev_chain);
    if (ev_done) (void)hipEventDestroy(ev_done);
    if (h_stats) (void)hipHostFree(h_stats);
}

namespace {

template <class T>
T* own(dto_handle* h, T* p) {
    h->owned.push_back(p);
    return p;
}

int fail(dto_handle* h, const std::string& m) {
    if (h) h->err = m; else g_create_error = m;
    return 1;
}

inline int pad64(int n) { return ((n + 63) / 64) * 64; }

struct ProfScope {
    dto_handle* h;
    hipStream_t st;
    bool on;
    ProfRec r{};
    ProfScope(dto_handle* h_, hipStream_t st_, int cat, double flops) : h(h_), st(st_), on(h_->profiling) {
        if (!on) return;
        r.cat = cat;
        r.flops = flops;
        auto take = [&](hipEvent_t& e) {
            if (h->ev_pool.empty()) (void)hipEventCreate(&e);
            else { e = h->ev_pool.back(); h->ev_pool.pop_back(); }
        };
        take(r.a);
        take(r.b);
        (void)hipEventRecord(r.a, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.b, st);
        h->prof.push_back(r);
    }
};
// (for the bandwidth-bound categories from CAT_ZERO on, `flops` carries the launch's algorithmic BYTES)
enum { CAT_BGEMM = 0, CAT_SWEEP = 1, CAT_OTHER = 2, CAT_BGEMM_HORNER = 3, CAT_BGEMM_SQUARE = 4, CAT_SWEEP_ADJOINT = 5,
       CAT_ZERO = 6, CAT_BUILD_A = 7, CAT_ASSEMBLY = 8, CAT_BASIS_MULTI = 9, CAT_CHAIN64 = 10 };

// ------------------------------------------------------------------------------------------
// structure
// ------------------------------------------------------------------------------------------

// number of integrator rows touching a column of knot kn (0-based): D per adjacent interval
inline int col_cnt(const dto_handle* h, int64_t kn) { return kn >= h->N ? 0 : (kn >= 1 ? 1 : 0) + (kn < h->K ? 1 : 0); }

double con_jac_value(const ConHost& c, const double* zk, int comp_i) {
    double s = 0.0;
    for (int q : c.comps) s += zk[q] * zk[q];
    const double v = zk[c.comps[comp_i]];
    return c.k.kind == DTO_CONSTRAINT_NORM_MINUS_C ? v / std::sqrt(s) : 2.0 * v;
}

void build_structure(dto_handle* h, const double* Z0) {
    const int64_t nv = h->n_vars;
    // constraint pattern = numeric Jacobian at Z0, exact zeros not stored (evaluator.jl:136)
    std::vector<std::pair<int64_t, int64_t>> ent;
    for (auto& c : h->con) {
        for (int64_t i = 0; i < c.n_times_total; ++i) {
            const int64_t kn = c.times0[i];
            const double* zk = Z0 + kn * h->z;
            for (size_t q = 0; q < c.comps.size(); ++q) {
                if (c.external) {
                    for (int r = 0; r < c.g_dim; ++r)
                        if (c.jac0[((size_t)i * c.comps.size() + q) * c.g_dim + r] != 0.0)
                            ent.emplace_back(kn * h->z + c.comps[q], c.row_off + i * c.g_dim + r);
                    continue;
                }
                const double v = con_jac_value(c, zk, (int)q);
                if (v != 0.0) ent.emplace_back(kn * h->z + c.comps[q], c.row_off + i);
            }
        }
    }
    std::sort(ent.begin(), ent.end());
    ent.erase(std::unique(ent.begin(), ent.end()), ent.end());
    h->con_cols.resize(ent.size());
    h->con_rows.resize(ent.size());
    std::vector<int32_t> extra(nv, 0);
    for (size_t i = 0; i < ent.size(); ++i) {
        h->con_cols[i] = ent[i].first;
        h->con_rows[i] = ent[i].second;
        extra[ent[i].first]++;
    }
    h->colptr.assign(nv + 1, 0);
    for (int64_t kn = 0; kn < h->N; ++kn) {
        const int64_t per = (int64_t)h->D * col_cnt(h, kn);
        for (int j = 0; j < h->z; ++j) {
            const int64_t c = kn * h->z + j;
            h->colptr[c + 1] = h->colptr[c] + per + extra[c];
        }
    }
    for (int64_t c = h->N * h->z; c < nv; ++c) h->colptr[c + 1] = h->colptr[c] + extra[c];  // global columns: only NonlinearGlobalConstraint rows
    h->jac_nnz = h->colptr[nv];
    const int64_t z = h->z;
    h->hess_block_nnz = h->N * (z * (z + 1) / 2) + h->K * z * z;  // evaluator.jl:201-202 on the block pattern
    // tail: entries whose column is a global variable (global columns follow every knot column in the CSC order).
    // GlobalObjective / GlobalKnotPointObjective mark their whole index block (global_objectives.jl:89-101, 277-300),
    // NonlinearGlobalConstraint contributes the non-zeros of its Hessian at mu = ones (evaluator.jl:166)
    std::vector<std::vector<int64_t>> rows(h->gd);
    const int64_t g0 = h->N * z;
    for (auto& e : h->ext_obj) {
        if (!e.global) continue;
        for (int gb : e.gcomps) {
            for (int64_t t : e.times0)
                if (t < h->N)
                    for (int ca : e.comps) rows[gb].push_back(t * z + ca);
            for (int ga : e.gcomps)
                if (ga <= gb) rows[gb].push_back(g0 + ga);
        }
    }
    for (auto& c : h->con) {
        if (!c.global) continue;
        const size_t ng = c.comps.size();
        for (size_t b = 0; b < ng; ++b)
            for (size_t a = 0; a < ng; ++a)
                if (c.comps[a] <= c.comps[b] && c.hess0[a + ng * b] != 0.0) rows[c.comps[b]].push_back(g0 + c.comps[a]);
    }
    h->tail_colptr.assign(h->gd + 1, 0);
    h->tail_rows.clear();
    for (int j = 0; j < h->gd; ++j) {
        std::sort(rows[j].begin(), rows[j].end());
        rows[j].erase(std::unique(rows[j].begin(), rows[j].end()), rows[j].end());
        h->tail_rows.insert(h->tail_rows.end(), rows[j].begin(), rows[j].end());
        h->tail_colptr[j + 1] = (int64_t)h->tail_rows.size();
    }
    h->hess_nnz = h->hess_block_nnz + (int64_t)h->tail_rows.size();
}

// first constraint-pattern entry of column c
inline size_t con_lower(const dto_handle* h, int64_t c) {
    return std::lower_bound(h->con_cols.begin(), h->con_cols.end(), c) - h->con_cols.begin();
}

int64_t hess_block_start(const dto_handle* h, int64_t kn) {
    const int64_t z = h->z, tri = z * (z + 1) / 2;
    return kn == 0 ? 0 : tri + (kn - 1) * (z * z + tri);
}

// value slabs of the handle that owns knots k_lo..k_hi (1-based, inclusive): positions inside the global vectors
struct ShardExtents {
    int64_t grad_lo, grad_len, jac_lo, jac_len, hess_lo, hess_len;
};
ShardExtents shard_extents(const dto_handle* h, int64_t k_lo, int64_t k_hi) {
    const int64_t kn_lo = k_lo - 1, n_knots = k_hi - k_lo + 1;
    const int64_t c_lo = kn_lo * h->z, c_hi = (kn_lo + n_knots) * h->z;
    const bool last = k_hi == h->N;  // global-variable columns and the Hessian's tail ride with the last knot
    ShardExtents e;
    e.grad_lo = c_lo;
    e.grad_len = c_hi - c_lo + (last ? h->gd : 0);
    e.jac_lo = h->colptr[c_lo];
    e.jac_len = h->colptr[last ? h->n_vars : c_hi] - e.jac_lo;
    e.hess_lo = hess_block_start(h, kn_lo);
    e.hess_len = hess_block_start(h, kn_lo + n_knots) - e.hess_lo + (last ? (int64_t)h->tail_rows.size() : 0);
    return e;
}

// rows of g that handle owns, as (global 0-based start, length) segments in the order of its local buffer: per integrator the
// rows of the owned intervals, then per constraint the listed times at owned knots (runs of consecutive listings merged)
std::vector<std::pair<int64_t, int64_t>> shard_row_segments(const dto_handle* h, int64_t k_lo, int64_t k_hi) {
    const int64_t kn_lo = k_lo - 1, n_knots = k_hi - k_lo + 1;
    const int64_t n_int = std::max<int64_t>(0, std::min<int64_t>(k_hi, h->K) - k_lo + 1);
    std::vector<std::pair<int64_t, int64_t>> seg;
    for (size_t i = 0; i < h->integ_kind.size(); ++i)
        if (n_int > 0) seg.emplace_back(h->integ_row_off[i] + kn_lo * h->integ_dim[i], n_int * h->integ_dim[i]);
    for (auto& c : h->con) {
        int64_t prev = -2;  // index of the previous owned listing of THIS constraint
        for (int64_t i = 0; i < c.n_times_total; ++i) {
            const int64_t kn = c.times0[i];
            if (kn >= h->N ? k_hi != h->N : (kn < kn_lo || kn >= kn_lo + n_knots)) continue;  // pseudo-knot N: last rank
            if (!seg.empty() && prev == i - 1 && seg.back().first + seg.back().second == c.row_off + i * c.g_dim)
                seg.back().second += c.g_dim;
            else
                seg.emplace_back(c.row_off + i * c.g_dim, c.g_dim);
            prev = i;
        }
    }
    return seg;
}

// ------------------------------------------------------------------------------------------
// sweeps and chain
// ------------------------------------------------------------------------------------------

void alloc_sweep(dto_handle* h, BilHost& b, SweepBuf& w, int T, bool with_W) {
    const int npad = b.k.npad;
    w.npad = npad;
    w.T_alloc = T;
    w.TN = (npad % 128 == 0) ? 128 : 64;
    w.nblk = w.TN;
    int64_t nint = std::max<int64_t>(h->P.n_int, 1);
    w.Kpad = (int)(((nint + w.TN - 1) / w.TN) * w.TN);
    const size_t typesz = (size_t)w.Kpad * npad;
    w.Z[0] = own(h, dalloc<double>(typesz * T));
    w.Z[1] = own(h, dalloc<double>(typesz * T));
    w.S = own(h, dalloc<double>(typesz * T));
    w.GY = own(h, dalloc<double>(typesz));
    w.W = with_W ? own(h, dalloc<double>(typesz * (b.k.m + 1))) : nullptr;
    w.scaleA = own(h, dalloc<double>((size_t)(b.k.m + 1) * w.Kpad));
    w.scaleU = own(h, dalloc<double>((size_t)(b.k.m + 1) * w.Kpad));
    w.scaleE = own(h, dalloc<double>((size_t)2 * w.Kpad));
    w.termnorm = own(h, dalloc<unsigned long long>((size_t)3 * T * w.Kpad));
    w.sumnorm = own(h, dalloc<unsigned long long>((size_t)T * w.Kpad));
    w.active = own(h, dalloc<int32_t>(w.Kpad / w.TN));
    w.stats = own(h, dalloc<int32_t>(4));
    HIP_CHECK(hipMemset(w.stats, 0, 4 * sizeof(int32_t)));
    // padding columns/rows must be finite zeros from the start
    HIP_CHECK(hipMemset(w.Z[0], 0, typesz * T * sizeof(double)));
    HIP_CHECK(hipMemset(w.Z[1], 0, typesz * T * sizeof(double)));
    HIP_CHECK(hipMemset(w.S, 0, typesz * T * sizeof(double)));
    HIP_CHECK(hipMemset(w.GY, 0, typesz * sizeof(double)));
}

SweepTypes make_types(int m, bool second_order) {
    SweepTypes ty{};
    int T = 0;
    ty.t[T++] = TypeDesc{0, {0, 0}, {0, 0}, {0, 0}};  // p
    for (int j = 0; j < m; ++j) {                    // d^j: + E_j p
        TypeDesc d{};
        d.n_extra = 1; d.gen[0] = 1 + j; d.src[0] = 0; d.mult[0] = 1.0;
        ty.t[T++] = d;
    }
    if (second_order) {
        for (int i = 0; i < m; ++i)
            for (int j = i; j < m; ++j) {  // h^{ij}: + E_i d^j + E_j d^i
                TypeDesc d{};
                if (i == j) {
                    d.n_extra = 1; d.gen[0] = 1 + i; d.src[0] = 1 + i; d.mult[0] = 2.0;
                } else {
                    d.n_extra = 2;
                    d.gen[0] = 1 + i; d.src[0] = 1 + j; d.mult[0] = 1.0;
                    d.gen[1] = 1 + j; d.src[1] = 1 + i; d.mult[1] = 1.0;
                }
                ty.t[T++] = d;
            }
    }
    ty.T = T;
    return ty;
}

struct Bounds {
    double beta, b1;
};

// enqueue only: the two doubles land in h_pinned[0..1] in stream order (read them after a later synchronisation point)
void enqueue_bounds(dto_handle* h, BilHost& b, const double* dZ, hipStream_t st) {
    HIP_CHECK(hipMemsetAsync(h->d_bounds, 0, 2 * sizeof(double), st));
    launch_norm_bounds(st, h->P, b.k, dZ, b.d_g1, b.d_n2, reinterpret_cast<unsigned long long*>(h->d_bounds));
    HIP_CHECK(hipMemcpyAsync(h->h_pinned, h->d_bounds, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
}
Bounds get_bounds(dto_handle* h, BilHost& b, const double* dZ, hipStream_t st) {
    HIP_CHECK(hipMemsetAsync(h->d_bounds, 0, 2 * sizeof(double), st));
    launch_norm_bounds(st, h->P, b.k, dZ, b.d_g1, b.d_n2, reinterpret_cast<unsigned long long*>(h->d_bounds));
    HIP_CHECK(hipMemcpyAsync(h->h_pinned, h->d_bounds, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    Bounds r{h->h_pinned[0], h->h_pinned[1]};
    return r;
}

struct SweepPlan {
    int q, d_ub;
    int tc = -1;   // first step at which the termination test runs (-1: d_ub / 2 - 1)
};
void read_hump(dto_handle* h, BilHost& b);
SweepPlan plan_hump(const BilHost& b, double beta_fallback);

SweepPlan plan_sweep(double beta) {
    SweepPlan p{1, 12};
    if (!(beta == beta) || beta > 1e6) {  // non-finite iterate: bounded work, NaN/Inf propagates to the output
        p.q = 1; p.d_ub = 30;
        return p;
    }
    static const double theta_v = tune_double("DTO_THETA_V", 9.0);  // worst-case cancellation budget e^9 ~ 1e4 on the Taylor sums (tolerance 1e-10)
    p.q = std::max(1, (int)std::ceil(beta / theta_v));
    const double br = beta / p.q;
    int t = 8;
    double term = 1.0;
    for (int i = 1; i <= t; ++i) term *= br / i;
    while (term > 1e-19 && t < 200) { ++t; term *= br / t; }
    p.d_ub = t + 6;
    return p;
}

// Taylor steps a fused sweep (already enqueued on st) actually took: waits for it.
int fused_sweep_steps(dto_handle* h, const SweepBuf& w, int d_ub, hipStream_t st) {
    int32_t* hs = reinterpret_cast<int32_t*>(h->h_pinned + 6);
    HIP_CHECK(hipMemcpyAsync(hs, w.stats, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    return std::max(1, std::min(hs[1] - 1, d_ub));
}
bool fused_sweep_applies(const dto_handle* h, const BilHost& b, const SweepBuf& w, const SweepTypes& ty, const SweepPlan& plan, bool store);
bool cluster_sweep_applies(const dto_handle* h, const BilHost& b, const SweepBuf& w, const SweepTypes& ty, const SweepPlan& plan, bool store,
                           ClusterSweepPlan& cp);
bool gs_sweep_applies(const dto_handle* h, const BilHost& b, const SweepBuf& w, const SweepTypes& ty, const SweepPlan& plan, bool store,
                      bool shared_chip, GsSweepPlan& gp);
bool ensure_bind_runs(dto_handle* h, int which);

// Does this sweep run in the 64-state generator-stationary form, which can read its step budget from device memory?  (No term
// store, no reuse: the host then needs nothing of the plan.)
bool s64_plans_itself(const dto_handle* h, const BilHost& b, const SweepBuf& w, const SweepTypes& ty) {
    static const int on = tune_int("DTO_PLAN_DEV", 1);  // A/B runs (TUNING builds)
    FusedSweepPlan fp;
    return on && !h->reuse && h->sweep_form != 1 && !w.frozen && b.k.npad == 64 && sweep_fused_plan(w.npad, b.k.m, ty, h->P.n_int, h->n_cu, fp) && fp.S64;
}

// 33..64 states: the propagator chain as one launch (dto_chain64.hip)
bool chain64_applies(const dto_handle* h, const BilHost& b) {
    static const int chain64_on = tune_int("DTO_CHAIN64", 1);  // A/B runs (TUNING builds)
    return b.k.npad == 64 && chain64_on && h->chain_form != 1 && h->P.n_int <= b.chain_cap;
}

// Returns the number of Taylor steps enqueued in the last round.  store = true keeps every term in w.Zt
// (term t of all types at Zt + t*T*Kpad*npad) instead of ping-ponging two buffers.
int run_sweep(dto_handle* h, BilHost& b, SweepBuf& w, const SweepTypes& ty, const double* dZ, const double* dmu,
              int src_kind, int transposed, const SweepPlan& plan, hipStream_t st, bool store = false,
              bool skip_init = false, bool want_steps = false, bool shared_chip = false, int prof_cat = CAT_SWEEP,
              const int32_t* plan_dev = nullptr) {
    const double flops_step = [&] {
        double segs = 0;
        for (int t = w.frozen ? w.first_type : 0; t < ty.T; ++t) segs += b.k.m + 1;  // an extra term rides in the segment of its generator
        return 2.0 * b.k.npad * (double)b.k.npad * w.Kpad * segs;
    }();
    // The termination test cannot fire early in the series: it is first run at step tc = d_ub/2 - 1 (d_ub comes from an
    // upper bound on the terms needed at this very Z, so tc is a function of Z alone and results stay reproducible; a
    // column block that would pass earlier merely adds a few terms below 1e-16 of its sum).
    static const int tc_env = tune_int("DTO_SWEEP_TC", -1);
    int tc = tc_env >= 0 ? tc_env : (plan.tc >= 0 ? plan.tc : plan.d_ub / 2 - 1);
    if (tc < 2) tc = 0;
    // Generator-stationary form (dto_sweep_gs.hip, round 4): clusters of npad / 32 workgroups with the generators resident in
    // their registers; serves what the single-workgroup form cannot fill the chip with -- single-column sweeps (eval_constraint,
    // the Hessian's forward column), short shards -- see gs_sweep_applies.
    GsSweepPlan gp;
    if (!skip_init && gs_sweep_applies(h, b, w, ty, plan, store, shared_chip, gp)) {
        const int wi = &w == &b.ad ? 1 : 0;
        const size_t need = sweep_gs_norm_doubles(gp);
        if (need > b.gs_xn_cap[wi] || gp.n_groups > b.gs_groups[wi]) {
            b.gs_xn[wi] = own(h, dalloc<double>(need));
            b.gs_arrive[wi] = own(h, dalloc<unsigned>((size_t)(gp.n_groups + 3) / 4 * 4));
            b.gs_xn_cap[wi] = need;
            b.gs_groups[wi] = gp.n_groups;
        }
        w.nblk = gp.ipw;
        HIP_CHECK(hipMemsetAsync(w.stats, 0, 4 * sizeof(int32_t), st));
        {
            ProfScope ps(h, st, prof_cat, flops_step * plan.d_ub);
            HIP_CHECK(launch_sweep_gs(st, h->P, b.k, w, ty, gp, b.gs_xn[wi], b.gs_arrive[wi], dZ, dmu, src_kind, transposed, plan.d_ub, tc,
                                      store, 1.1e-16));
        }
        if (!want_steps) return plan.d_ub;
        return fused_sweep_steps(h, w, plan.d_ub, st);
    }
    // Fused form (dto_sweep_fused.hip): the whole series in one persistent launch, a workgroup per few intervals.  The
    // step-per-launch form below remains for sweeps over frozen p terms and for the products' extra start vector.
    FusedSweepPlan fp;
    if (!skip_init && fused_sweep_applies(h, b, w, ty, plan, store) && sweep_fused_plan(w.npad, b.k.m, ty, h->P.n_int, h->n_cu, fp, shared_chip)) {
        w.nblk = fp.ipw;
        HIP_CHECK(hipMemsetAsync(w.stats, 0, 4 * sizeof(int32_t), st));
        {
            // flops of the step budget (an upper bound: workgroups leave when their columns have converged)
            ProfScope ps(h, st, prof_cat, flops_step * plan.d_ub * plan.q);
            HIP_CHECK(launch_sweep_fused(st, h->P, b.k, w, ty, fp, dZ, dmu, src_kind, transposed, plan.q, plan.d_ub, tc, store, 1.1e-16,
                                         plan_dev));
        }
        if (!want_steps) return plan.d_ub;
        return fused_sweep_steps(h, w, plan.d_ub, st);
    }
    if (plan_dev) throw HipError{"run_sweep: a device-side plan without the form that reads it"};
    // Row-split cluster form: where the single-workgroup form has too few interval groups for the chip -- short shards of 128-
    // and 256-state problems (the 250-knot share of the 2000-knot metric on 8 GPUs).  Measured per Jacobian / Hessian,
    // cluster against step per launch (tools/cluster_time.py): 256 x 250 2.12 / 2.02 against 2.22 / 2.04 ms, 128 x 250 0.74 /
    // 1.02 against 0.93 / 1.18 ms.  NOT used where it measured slower: single-column sweeps (eval_constraint 0.94 against
    // 0.62 ms at 250 knots, 1.52 against 1.45 at 2000: the rendezvous + slice exchange costs ~10 us per Taylor step, as much
    // as the step's MFMA work there) and 512 / 1024 states (23.6 against 18.7 ms, 57 against 47 ms per Jacobian: the step
    // launches tile the 2500 columns 32 wide, a cluster member is held to 16 by its LDS).  Not with frozen p terms or the
    // products' extra start vector.
    ClusterSweepPlan cp;
    if (!skip_init && cluster_sweep_applies(h, b, w, ty, plan, store, cp)) {
        const int wi = &w == &b.ad ? 1 : 0;
        const size_t need = sweep_cluster_workspace_doubles(w.npad, cp);
        if (need > b.xch_cap[wi] || cp.n_clusters > b.xch_clusters[wi]) {
            b.xch[wi] = own(h, dalloc<double>(need));
            b.xch_arrive[wi] = own(h, dalloc<unsigned>((size_t)cp.n_clusters));
            b.xch_cap[wi] = need;
            b.xch_clusters[wi] = cp.n_clusters;
        }
        w.nblk = cp.ipw;
        HIP_CHECK(hipMemsetAsync(w.stats, 0, 4 * sizeof(int32_t), st));
        {
            ProfScope ps(h, st, prof_cat, flops_step * plan.d_ub * plan.q);
            HIP_CHECK(launch_sweep_cluster(st, h->P, b.k, w, ty, cp, b.xch[wi], b.xch_arrive[wi], dZ, dmu, src_kind, transposed, plan.q,
                                           plan.d_ub, tc, store, 1.1e-16));
        }
        if (!want_steps) return plan.d_ub;
        return fused_sweep_steps(h, w, plan.d_ub, st);
    }
    w.nblk = w.TN;
    const size_t tstride = (size_t)ty.T * w.Kpad * w.npad;
    SweepBuf ws = w;
    if (store) ws.Z[0] = w.Zt;
    if (w.frozen) launch_sweep_init_tangents(st, ws, ty.T);  // scale factors and type-0 sums are those of the earlier callback
    else if (!skip_init) launch_sweep_init(st, h->P, b.k, ws, ty, dZ, dmu, src_kind, plan.q);
    int launched = 0;
    // one timed region per sweep (steps, termination tests and the gaps between them): an event pair per step costs
    // 0.2 ms per Jacobian call at 256x2000.  Its flop count covers every enqueued step, including the few that find their
    // column blocks already converged (bench.py prices the sweep by the terms actually used, dto_last_stats).
    ProfScope ps(h, st, prof_cat, 0.0);
    for (int round = 0; round < plan.q; ++round) {
        if (round > 0) launch_sweep_restart(st, w, ty.T);
        int buf = 0;
        launched = 0;
        bool pending = false;
        int slot = 0;
        // (each termination test left out before tc is one kernel and one dependent-launch gap less, ~12 us; the term-norm
        // slots the tests recycle are cleared once instead)
        for (int t = 0; t < plan.d_ub; ++t) {
            if (tc > 0 && t == tc - 1)
                HIP_CHECK(hipMemsetAsync(w.termnorm, 0, sizeof(unsigned long long) * (size_t)3 * w.T_alloc * w.Kpad, st));
            {
                ps.r.flops += flops_step;
                if (store) {
                    ws.Z[0] = w.Zt + (size_t)t * tstride;
                    ws.Z[1] = w.Zt + (size_t)(t + 1) * tstride;
                    // one stored type: split K over the generators, the next terms' slots are the scratch
                    const bool split = ty.T == 1 && (int64_t)(t + b.k.m + 4) <= (int64_t)w.dcap * (1 + b.k.m);
                    launch_sweep_step(st, b.k, ws, ty, transposed, t, 0, split ? 1 : 0);
                } else {
                    launch_sweep_step(st, b.k, w, ty, transposed, t, buf);
                }
            }
            if (t >= tc) launch_sweep_check(st, w, ty.T, t, 1.1e-16);
            buf ^= 1;
            launched = t + 1;
            // Every 4 steps the number of still-active column blocks (4 bytes) is copied back; the copy of the
            // PREVIOUS checkpoint is read before enqueueing more, so the host never waits on the GPU's current
            // work (the decision lags by 4 steps, which then cost ~5 us each as inactive blocks exit at once).
            if (t >= 3 && (t % 4) == 3 && t + 1 < plan.d_ub) {
                int32_t* hs = reinterpret_cast<int32_t*>(h->h_pinned + 6);
                if (pending) {
                    HIP_CHECK(hipEventSynchronize(h->ev_stats));
                    if (hs[slot ^ 1] == 0) break;
                }
                HIP_CHECK(hipMemcpyAsync(hs + slot, w.stats, sizeof(int32_t), hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipEventRecord(h->ev_stats, st));
                pending = true;
                slot ^= 1;
            }
        }
    }
    return launched;
}

// multisets of size r over {0..m} as sorted tuples, lexicographic
void enum_multisets(int m1, int r, std::vector<int>& cur, int start, std::vector<std::vector<int>>& out) {
    if ((int)cur.size() == r) { out.push_back(cur); return; }
    for (int i = start; i < m1; ++i) {
        cur.push_back(i);
        enum_multisets(m1, r, cur, i, out);
        cur.pop_back();
    }
}

// S_alpha = sum over distinct first letters i of alpha of G_i * S_(alpha minus i): built once at create with
// the engine's own batched GEMM (one product per (alpha, i)).
void build_basis(dto_handle* h, BilHost& b, int cap) {
    const int m1 = b.k.m + 1, npad = b.k.npad;
    const size_t nn = (size_t)npad * npad;
    std::vector<std::vector<std::vector<int>>> sets(5);
    std::vector<std::map<std::vector<int>, int>> index(5);
    for (int r = 1; r <= 4; ++r) {
        std::vector<int> cur;
        enum_multisets(m1, r, cur, 0, sets[r]);
        for (size_t a = 0; a < sets[r].size(); ++a) index[r][sets[r][a]] = (int)a;
    }
    double* tmp = own(h, dalloc<double>(nn));
    std::vector<double*> S(5, nullptr);
    S[1] = const_cast<double*>(b.k.G);
    for (int r = 2; r <= 4; ++r) {
        const int cnt = (int)sets[r].size(), cntpad = ((cnt + 15) / 16) * 16;
        S[r] = own(h, dalloc<double>(nn * cntpad));
        HIP_CHECK(hipMemsetAsync(S[r], 0, nn * cntpad * sizeof(double), h->stream));
        std::vector<int32_t> idx;
        for (int a = 0; a < cnt; ++a) {
            const std::vector<int>& al = sets[r][a];
            for (int v : al) idx.push_back(v);
            int last = -1;
            for (size_t p = 0; p < al.size(); ++p) {
                const int i = al[p];
                if (i == last) continue;  // distinct first letters only
                last = i;
                std::vector<int> rest = al;
                rest.erase(rest.begin() + p);
                const double* prev = S[r - 1] + (size_t)index[r - 1][rest] * nn;
                launch_bgemm_plain(h->stream, npad, 1, b.k.G + (size_t)i * nn, prev, tmp);
                launch_add(h->stream, S[r] + (size_t)a * nn, tmp, (int64_t)nn);
            }
        }
        BasisSet& bs = b.basis[r - 2];
        bs.r = r; bs.cnt = cnt; bs.cntpad = cntpad; bs.S = S[r];
        bs.idx = own(h, dupload(idx));
        const int cappad = ((cap + 127) / 128) * 128;
        bs.coef = own(h, dalloc<double>((size_t)cappad * cntpad));
    }
    {
        // concatenation [I | G_0..G_m | S2 | S3 | S4] for the factor K
        int cnt = 1 + m1;
        for (int r = 2; r <= 4; ++r) cnt += (int)sets[r].size();
        const int cntpad = ((cnt + 15) / 16) * 16;
        double* Sall = own(h, dalloc<double>(nn * cntpad));
        HIP_CHECK(hipMemsetAsync(Sall, 0, nn * cntpad * sizeof(double), h->stream));
        std::vector<double> eye(nn, 0.0);
        for (int i = 0; i < b.k.n; ++i) eye[(size_t)i * npad + i] = 1.0;  // identity on the un-padded block only
        HIP_CHECK(hipMemcpyAsync(Sall, eye.data(), nn * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        std::vector<int32_t> idx;
        size_t col = 1;
        for (int q = 0; q < 4; ++q) idx.push_back(-1);
        for (int r = 1; r <= 4; ++r) {
            HIP_CHECK(hipMemcpyAsync(Sall + col * nn, S[r], nn * sets[r].size() * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            col += sets[r].size();
            for (auto& al : sets[r]) {
                for (int q = 0; q < 4; ++q) idx.push_back(q < r ? al[q] : -1);
            }
        }
        BasisSet& bs = b.basis_all;
        bs.r = 4; bs.cnt = cnt; bs.cntpad = cntpad; bs.S = Sall;
        bs.idx = own(h, dupload(idx));
        const int cappad = ((cap + 127) / 128) * 128;
        bs.coef = own(h, dalloc<double>((size_t)cappad * cntpad));
    }
    HIP_CHECK(hipStreamSynchronize(h->stream));
    b.use_basis = true;
}

void alloc_chain(dto_handle* h, BilHost& b, int cap) {
    const size_t nn = (size_t)b.k.npad * b.k.npad;
    for (int i = 0; i < 9; ++i) b.chain.W[i] = own(h, dalloc<double>(nn * cap));
    b.chain.norms = own(h, dalloc<double>((size_t)cap * 4));
    b.chain.colsum = own(h, dalloc<double>((size_t)3 * cap * b.k.npad * (b.k.npad / 64)));  // [set][interval][column][64-row chunk]
    if (!b.d_hump) b.d_hump = own(h, dalloc<unsigned long long>(8));
    b.chain.coef = own(h, dalloc<double>((size_t)cap * COEF_STRIDE));
    b.chain.s = own(h, dalloc<int32_t>(cap));
    b.chain.s3 = own(h, dalloc<int32_t>(cap));
    b.chain.smax = own(h, dalloc<int32_t>(8));
    HIP_CHECK(hipMemset(b.chain.smax, 0, 8 * sizeof(int32_t)));
    b.chain.d2max = reinterpret_cast<unsigned long long*>(b.chain.smax + 2);
    b.chain_cap = cap;
}

int chunk_size(const dto_handle* h, int npad) {
    // workspace budget for the 9 chain matrices (option "chain_chunk" lowers the chunk per call)
    const double budget = 40e9;   // (1024 states x 500 knots, the configs[4] share, in ONE chunk: 37.7 GB)
    int c = (int)(budget / (9.0 * npad * (double)npad * 8.0));
    c = std::max(8, (c / 8) * 8);
    return (int)std::min<int64_t>(c, std::max<int64_t>(h->P.n_int, 1));
}

// exp(dt G(u_k)) for every owned interval; -E_k goes straight into the Jacobian slab.
// Returns max_k ||A_k^2||_1^(1/2) (exact), which bounds the growth of the Taylor terms of the sweep.
// `after_last_enqueue(d2max)` runs on the host once every kernel of the chain has been enqueued (the GPU
// is then busy with the Taylor products and squarings): the caller uses it to drive the generator sweep on a
// second stream so that both proceed concurrently.
double run_chain(dto_handle* h, BilHost& b, const double* dZ, double* vals, double b1max, hipStream_t st,
                 const std::function<void(double)>& after_last_enqueue = nullptr, const std::function<void()>& in_bubble = nullptr,
                 bool nothing_waits = false) {
    const int npad = b.k.npad;
    const int64_t nint = h->P.n_int;
    double d2max = 0.0;
    if (nint <= 0) return d2max;
    int cap = h->chain_chunk > 0 ? std::min(h->chain_chunk, b.chain_cap) : b.chain_cap;
    if (h->xfer_cap > 0 && !h->deterministic) cap = std::min(cap, h->xfer_cap);
    int s_ub = 1;
    if (b1max == b1max && b1max > THETA_16) s_ub = std::isinf(b1max) ? 60 : std::max(1, (int)std::ceil(std::log2(b1max / THETA_16)));
    s_ub = std::min(s_ub, 60);
    // 33..64 states: the whole chain in ONE launch, a workgroup per interval (dto_chain64.hip) -- no batched-GEMM launches, no
    // workspace chunks, the evaluation form chosen per interval on the device; the host reads back only what plans the sweep
    if (chain64_applies(h, b)) {
        ChainWork& w = b.chain;
        HIP_CHECK(hipMemsetAsync(w.smax, 0, 8 * sizeof(int32_t), st));
        HIP_CHECK(hipMemsetAsync(b.d_hump, 0, 8 * sizeof(unsigned long long), st));
        int want_form = 0;
        if (h->expm_form == 2 || h->expm_form == 3) want_form = h->expm_form;
        {
            // priced at three powers + three more products per interval (either form, with its squarings, is 5..7 at these norms)
            ProfScope ps(h, st, CAT_CHAIN64, 6.0 * 2.0 * 64.0 * 64.0 * 64.0 * (double)nint);
            HIP_CHECK(launch_chain64(st, h->P, b.k, dZ, vals, w.norms, w.smax, w.d2max, w.s, s_ub, want_form, h->n_cu));
        }
        if (nothing_waits) {
            // the caller has planned and enqueued its sweep already: nothing on the host depends on this launch, the call stays
            // enqueue-only; the squaring count (a diagnostic) is read with the sweep statistics at the next entry point
            HIP_CHECK(hipMemcpyAsync(h->h_pinned + 28, w.smax, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            h->smax_pending = true;
            h->last_form = 0;
            if (in_bubble) in_bubble();
            if (h->on_chain_chunk) h->on_chain_chunk(0, (int)nint);
            if (after_last_enqueue) after_last_enqueue(0.0);
            return 0.0;
        }
        launch_hump(st, h->P, b.k, dZ, b.d_g1, h->P.kn_lo, (int)nint, w.norms, b.d_hump);
        int32_t* hs = reinterpret_cast<int32_t*>(h->h_pinned + 2);
        HIP_CHECK(hipMemcpyAsync(hs, w.smax, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(h->h_pinned + 16, b.d_hump, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipEventRecord(h->ev_chain, st));
        if (in_bubble) in_bubble();
        HIP_CHECK(hipEventSynchronize(h->ev_chain));
        h->last_form = 0;   // per interval
        h->last_smax = hs[0];
        read_hump(h, b);
        {
            double dv;
            memcpy(&dv, hs + 2, sizeof(double));
            d2max = dv;
        }
        if (h->on_chain_chunk) h->on_chain_chunk(0, (int)nint);
        if (after_last_enqueue) after_last_enqueue(d2max);
        return d2max;
    }
    const double gemm_flops = 2.0 * npad * (double)npad * npad;
    h->last_smax = 0;
    // chunks of equal size (a multiple of 8 intervals) rather than full ones plus a remainder
    const int64_t nchunk = (nint + cap - 1) / cap;
    const int per = (int)std::min<int64_t>(cap, (((nint + nchunk - 1) / nchunk) + 7) / 8 * 8);
    for (int64_t c0 = 0; c0 < nint; c0 += per) {
        const int nb = (int)std::min<int64_t>(per, nint - c0);
        const int64_t int0 = h->P.kn_lo + c0;
        ChainWork& w = b.chain;
        const int nbpad = ((nb + 127) / 128) * 128;
        if (b.use_basis) {
            launch_fill(st, w.norms, (int64_t)nb * 4, INFINITY);  // ||A||_1 is not needed: alpha never exceeds d_2
            double* outs[3] = {w.W[1], w.W[2], w.W[3]};
            const size_t cs_set = (size_t)cap * npad * (npad / 64);  // every slot is written by the launch below: no clearing
            double* css[3] = {w.colsum, w.colsum + cs_set, w.colsum + 2 * cs_set};
            // (A_k as a fourth, degree-1 set of the launch below instead of k_build_A's streaming pass: measured slower,
            // 1.15 against 0.79 + 0.22 ms -- 8000 more tiles with one K panel each)
            {
                ProfScope ps(h, st, CAT_BUILD_A, 8.0 * npad * (double)npad * nb);   // one matrix written per interval
                launch_build_A(st, h->P, b.k, dZ, int0, nb, w.W[0]);
            }
            launch_basis_coef_multi(st, h->P, b.k, 3, b.basis, dZ, int0, nb, nbpad);
            {
                // A^2, A^3, A^4 in one launch (tiles interleaved: the write-bound sets overlap the MFMA-bound one)
                const double cols = b.basis[0].cntpad + b.basis[1].cntpad + b.basis[2].cntpad;
                ProfScope ps(h, st, CAT_BASIS_MULTI, 2.0 * npad * (double)npad * cols * nb);
                launch_basis_gemm_multi(st, npad, nb, nbpad, 3, b.basis, outs, css);
            }
            launch_norm_from_colsum_multi(st, npad, nb, 3, css, w.norms);
        } else {
            {
                ProfScope ps(h, st, CAT_BUILD_A, 8.0 * npad * (double)npad * nb);
                launch_build_A(st, h->P, b.k, dZ, int0, nb, w.W[0]);
            }
            { ProfScope ps(h, st, CAT_BGEMM, gemm_flops * nb); launch_bgemm_plain(st, npad, nb, w.W[0], w.W[0], w.W[1]); }
            { ProfScope ps(h, st, CAT_BGEMM, gemm_flops * nb); launch_bgemm_plain(st, npad, nb, w.W[0], w.W[1], w.W[2]); }
            { ProfScope ps(h, st, CAT_BGEMM, gemm_flops * nb); launch_bgemm_plain(st, npad, nb, w.W[1], w.W[1], w.W[3]); }
            launch_norm1(st, npad, nb, w);
        }
        HIP_CHECK(hipMemsetAsync(w.smax, 0, 8 * sizeof(int32_t), st));
        launch_expm_params(st, nb, s_ub, w);
        if (c0 == 0) HIP_CHECK(hipMemsetAsync(b.d_hump, 0, 8 * sizeof(unsigned long long), st));
        launch_hump(st, h->P, b.k, dZ, b.d_g1, int0, nb, w.norms, b.d_hump);
        // The evaluation form is decided ON THE DEVICE (k_expm_coef: three products + s3 squarings against two products + s
        // squarings, summed over the chunk) unless an option pins it, so the factor K of the first product can be
        // enqueued before the host knows the outcome; the host needs it only for the launch sequence that follows and reads
        // it back (with the squaring counts, 32 bytes) while the GPU works on K -- `in_bubble` adds more independent work.
        static const int env_form = tune_int("DTO_EXPM_FORM", 0);  // A/B runs
        int want_form = 0;
        if (env_form == 2 || env_form == 3) want_form = env_form;
        if (h->expm_form == 2 || h->expm_form == 3) want_form = h->expm_form;
        launch_expm_coef(st, nb, w, want_form);
        // ... and it leaves on a stream of its own right behind k_expm_coef, so the host learns the form while the GPU is still
        // busy with K's GEMM (0.65 ms at 256 x 2000) and has the products enqueued before that GEMM ends: no bubble
        int32_t* hs = reinterpret_cast<int32_t*>(h->h_pinned + 2);
        static const int rb_side = tune_int("DTO_RB_STREAM", 1);  // 0: the readback follows K's GEMM on the call's stream (round 2)
        hipEvent_t ev_s = h->ev_chain;
        auto readback = [&](hipStream_t rs) {
            HIP_CHECK(hipMemcpyAsync(hs, w.smax, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, rs));
            HIP_CHECK(hipMemcpyAsync(h->h_pinned + 16, b.d_hump, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, rs));
            HIP_CHECK(hipEventRecord(ev_s, rs));
        };
        if (rb_side) {
            HIP_CHECK(hipEventRecord(h->ev_rb, st));
            HIP_CHECK(hipStreamWaitEvent(h->stream_rb, h->ev_rb, 0));
            readback(h->stream_rb);
        }
        if (b.use_basis) {
            launch_basis_coef(st, h->P, b.k, b.basis_all, dZ, int0, nb, nbpad, w.coef);
            ProfScope ps(h, st, CAT_OTHER, 2.0 * npad * (double)npad * b.basis_all.cntpad * nb);
            launch_basis_gemm(st, npad, nb, nbpad, b.basis_all, w.W[5], nullptr);
        } else {
            launch_poly_h3(st, npad, nb, w);
        }
        if (!rb_side) readback(st);
        if (c0 == 0 && in_bubble) in_bubble();
        HIP_CHECK(hipEventSynchronize(ev_s));
        const int form = hs[6];
        if (form != 2 && form != 3) throw HipError{"propagator chain: the evaluation form did not come back from the device"};
        h->last_form = form;
        // Y = A^4 K -> (Y + Pa, Y + Pb) in one launch
        { ProfScope ps(h, st, CAT_BGEMM_HORNER, gemm_flops * nb); launch_bgemm_poly(st, npad, nb, w, 3, 5, 4, COEF_PA, 6, COEF_PB); }
        const SlabDest slab{h->P, b.k, int0, vals};  // intervals with s_k = 0: the last product is exp(A_k) itself
        if (form == 2) {
            // T_16 = (Y + Pa)(Y + Pb) + Pc
            ProfScope ps(h, st, CAT_BGEMM_HORNER, gemm_flops * nb);
            launch_bgemm_poly(st, npad, nb, w, 4, 6, 5, COEF_PC, -1, 0, false, &slab);
        } else {
            // (L, R) = Ya Yb + weights of Ya + polynomials, then r = L R + Pe
            { ProfScope ps(h, st, CAT_BGEMM_HORNER, gemm_flops * nb); launch_bgemm_poly(st, npad, nb, w, 4, 6, 7, COEF_L, 8, COEF_R, true); }
            { ProfScope ps(h, st, CAT_BGEMM_HORNER, gemm_flops * nb); launch_bgemm_poly(st, npad, nb, w, 7, 8, 5, COEF_PC, -1, 0, false, &slab); }
        }
        const int s_max = form == 3 ? hs[4] : hs[0];
        const int s_sum = form == 3 ? hs[5] : hs[1];
        read_hump(h, b);  // accumulated over the chunks so far; final after the last one
        {
            double dv;
            memcpy(&dv, hs + 2, sizeof(double));
            d2max = (dv == dv) ? std::max(d2max, dv) : dv;
        }
        const double sq_flops = s_max > 0 ? gemm_flops * (double)s_sum / s_max : 0.0;
        h->last_smax = std::max(h->last_smax, s_max);
        int src = 5;
        for (int it = 0; it < s_max; ++it) {
            ProfScope ps(h, st, CAT_BGEMM_SQUARE, sq_flops);
            launch_bgemm_square(st, npad, nb, w, src, src == 4 ? 5 : 4, it, h->P, b.k, int0, vals);
            src = src == 4 ? 5 : 4;
        }
        if (h->on_chain_chunk) h->on_chain_chunk(c0, nb);  // the -E_k of these intervals are final behind what is enqueued now
    }
    if (after_last_enqueue) after_last_enqueue(d2max);
    return d2max;
}

bool fused_sweep_applies(const dto_handle* h, const BilHost& b, const SweepBuf& w, const SweepTypes& ty, const SweepPlan& plan, bool store) {
    FusedSweepPlan fp;
    if (h->sweep_form == 1 || w.frozen || (store && !(w.Zt && plan.d_ub + 1 <= w.dcap))) return false;
    if (!sweep_fused_plan(w.npad, b.k.m, ty, h->P.n_int, h->n_cu, fp)) return false;
    // single-column sweeps (eval_constraint, the Hessian's forward column): only the 64-state generator-stationary form
    if (ty.T == 1) return fp.S64 && !(store && h->reuse);   // (a frozen sweep reads nterms_p in blocks of TN intervals)
    return true;
}

bool cluster_sweep_applies(const dto_handle* h, const BilHost& b, const SweepBuf& w, const SweepTypes& ty, const SweepPlan& plan, bool store,
                           ClusterSweepPlan& cp) {
    static const int on = tune_int("DTO_SWEEP_CLUSTER", 1);  // A/B runs (TUNING builds): 0 = never
    static const int big = tune_int("DTO_SWEEP_CLUSTER_BIG", 0);  // 512 and 1024 states as well
    static const int t1 = tune_int("DTO_SWEEP_CLUSTER_T1", 0);    // single-column sweeps too
    if (!on || h->sweep_form == 1 || w.frozen) return false;
    if (ty.T == 1 && !t1) return false;
    if (store && !(w.Zt && plan.d_ub + 1 <= w.dcap)) return false;
    if (store && ty.T == 1 && h->reuse) return false;
    if (w.npad > 256 && !big) return false;
    return sweep_cluster_plan(w.npad, b.k.m, ty, h->P.n_int, h->n_cu, cp);
}

// The generator-stationary form needs the whole chip to itself (one 512-register workgroup per CU, all cluster members resident):
// not beside the chain (`shared_chip`), not with sub-stepping, frozen p terms or the products' extra start vector.  It is taken
// where the single-workgroup form cannot fill the chip: single-column sweeps and sweeps the fused planner refuses (short shards);
// measured (tools/sweep_gs_probe, 256 states): p column of 2000 knots 0.77 ms against 1.38 ms for the split-K step launches,
// Jacobian sweep of 250 knots 0.80 against 1.15 ms for the row-split cluster form.
bool gs_sweep_applies(const dto_handle* h, const BilHost& b, const SweepBuf& w, const SweepTypes& ty, const SweepPlan& plan, bool store,
                      bool shared_chip, GsSweepPlan& gp) {
    static const int on = tune_int("DTO_SWEEP_GS", 1);  // A/B runs (TUNING builds): 0 = never, 2 = wherever it can run
    if (!on || h->sweep_form == 1 || w.frozen || plan.q != 1 || h->n_cu < 64) return false;
    if (shared_chip && on != 2) return false;
    if (store && !(w.Zt && plan.d_ub + 1 <= w.dcap)) return false;
    if (store && ty.T == 1 && h->reuse) return false;   // (a frozen sweep reads nterms_p in blocks of TN intervals)
    if ((size_t)ty.T * w.Kpad * w.npad * 8 >= (1ull << 31)) return false;   // 32-bit buffer offsets into a term slab
    if (!sweep_gs_plan(w.npad, b.k.m, ty, h->P.n_int, h->n_cu, gp)) return false;
    if (on == 2) return true;
    if (ty.T == 1) return true;
    FusedSweepPlan fp;
    return !sweep_fused_plan(w.npad, b.k.m, ty, h->P.n_int, h->n_cu, fp);
}

// max_k ||A_k^2||_1^(1/2), exact, for callbacks that do not run the propagator chain: A_k and A_k^2 only
// (one streaming pass + one small GEMM per chunk).  Sharper than the generator-norm bound, so the sweep
// usually needs a single round (q = 1).
// `higher`: also ||A^3||, ||A^4|| (store-less basis GEMMs) and return max_k min(d2, max(d3, d4)) -- the sharper growth
// rate of Al-Mohy & Higham's alpha_3, worth its cost only when d2 alone would force sub-stepping
double exact_d2(dto_handle* h, BilHost& b, const double* dZ, hipStream_t st, bool higher = false) {
    const int npad = b.k.npad;
    const int64_t nint = h->P.n_int;
    double d2max = 0.0;
    for (int64_t c0 = 0; c0 < nint; c0 += b.chain_cap) {
        const int nb = (int)std::min<int64_t>(b.chain_cap, nint - c0);
        const int64_t int0 = h->P.kn_lo + c0;
        ChainWork& w = b.chain;
        HIP_CHECK(hipMemsetAsync(w.smax, 0, 8 * sizeof(int32_t), st));
        launch_fill(st, w.norms, (int64_t)nb * 4, INFINITY);  // norms not computed below stay "unknown" for k_hump
        if (c0 == 0) HIP_CHECK(hipMemsetAsync(b.d_hump, 0, 8 * sizeof(unsigned long long), st));
        if (b.use_basis) {
            // ||A_k^2||_1 straight from the generator-subspace GEMM's fused column sums: neither A nor A^2 is
            // written (K = number of degree-2 products: a few GFLOP for all intervals)
            const int nbpad = ((nb + 127) / 128) * 128;
            const int nr = higher ? 3 : 1;
            for (int r = 0; r < nr; ++r) {
                double* cs = w.colsum + (size_t)r * b.chain_cap * npad * (npad / 64);
                launch_basis_coef(st, h->P, b.k, b.basis[r], dZ, int0, nb, nbpad, nullptr);
                launch_basis_gemm(st, npad, nb, nbpad, b.basis[r], nullptr, cs);
                launch_norm_from_colsum(st, npad, nb, cs, w.norms, 1 + r, higher ? nullptr : w.d2max);
            }
            if (higher) launch_beta_from_norms(st, nb, w);
        } else {
            launch_build_A(st, h->P, b.k, dZ, int0, nb, w.W[0]);
            launch_bgemm_plain(st, npad, nb, w.W[0], w.W[0], w.W[1]);
            launch_norm1_one(st, npad, nb, w, 1);
        }
        launch_hump(st, h->P, b.k, dZ, b.d_g1, int0, nb, w.norms, b.d_hump);
        int32_t* hs = reinterpret_cast<int32_t*>(h->h_pinned + 2);
        HIP_CHECK(hipMemcpyAsync(hs, w.smax, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(h->h_pinned + 16, b.d_hump, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        double dv;
        memcpy(&dv, hs + 2, sizeof(double));
        d2max = (dv == dv) ? std::max(d2max, dv) : dv;
    }
    read_hump(h, b);
    return d2max;
}

// Plan from the a-priori hump bound of k_hump: the fewest rounds whose Taylor sums cannot lose more than theta_v
// e-folds to cancellation; falls back to the growth-rate rule when no q <= 4 qualifies or the bound is not finite.
void read_hump(dto_handle* h, BilHost& b) {
    const unsigned long long* hp = reinterpret_cast<const unsigned long long*>(h->h_pinned + 16);
    b.hump_valid = true;
    for (int q = 0; q < 4; ++q) {
        double v;
        memcpy(&v, &hp[q], sizeof(double));
        b.hump_logH[q] = v;
        b.hump_kend[q] = (int)hp[4 + q];
        if (!(v == v)) b.hump_valid = false;
    }
}
SweepPlan plan_hump(const BilHost& b, double beta_fallback) {
    static const double theta_v = tune_double("DTO_THETA_V", 9.0);
    static const bool on = tune_int("DTO_HUMP_PLAN", 1) != 0;
    if (on && b.hump_valid)
        for (int q = 1; q <= 4; ++q)
            if (b.hump_logH[q - 1] <= theta_v) return SweepPlan{q, std::min(200, b.hump_kend[q - 1] + 6)};
    return plan_sweep(beta_fallback);
}

// `loose`: the caller keeps no Taylor terms (eval_constraint without reuse_forward_sweep), so a generous step budget costs nothing --
// the one-launch sweeps end by their own termination test.  Then the hump criterion is applied to the cheap bound itself
// (max_k beta^k / k! <= e^9, the same four digits plan_hump allows): at the benchmark shape the triangle-inequality bound on
// ||A^2||^(1/2) is 9.5, just past the beta <= 9 rule, and the exact norm (a store-less basis GEMM, a kernel for the hump and two
// host round trips: 0.2 of the callback's 1.1 ms) was bought only to learn what this already shows.
// The step budget the cheap generator-norm bound alone gives, where that is a single round (q = 1): no exact norm needed.
bool cheap_plan(const Bounds& bd, bool loose, SweepPlan& out) {
    if (plan_sweep(bd.beta).q == 1) { out = plan_sweep(bd.beta); return true; }  // the cheap bound already gives one round
    if (loose && bd.beta == bd.beta && bd.beta < 40.0) {
        double lh = 0.0;
        for (int k = 1; k < 200; ++k) lh = std::max(lh, k * std::log(bd.beta) - std::lgamma(k + 1.0));
        if (lh <= 9.0) {
            SweepPlan p{1, 12};
            int t = 8;
            double term = 1.0;
            for (int i = 1; i <= t; ++i) term *= bd.beta / i;
            while (term > 1e-19 && t < 200) { ++t; term *= bd.beta / t; }
            p.d_ub = t + 6;
            // the terms of the series grow up to index ~beta and fall from there: the test (two successive terms below 1.1e-16 of the
            // sum, Al-Mohy & Higham's own criterion, which they apply from the first term on) starts a few terms past the peak
            // of the BOUND -- a function of Z alone, like d_ub / 2 - 1, but not inflated by the bound's slack in the tail
            p.tc = std::min(p.d_ub / 2 - 1, std::max(2, (int)std::ceil(bd.beta) + 4));
            out = p;
            return true;
        }
    }
    return false;
}

SweepPlan plan_from(dto_handle* h, BilHost& b, const double* dZ, hipStream_t st, bool loose = false) {
    Bounds bd = get_bounds(h, b, dZ, st);
    SweepPlan cheap;
    if (cheap_plan(bd, loose, cheap)) return cheap;
    // A cheap bound far beyond one round's radius (2.5 theta_v: a function of Z alone): ||A^2|| by itself will not settle q = 1 either,
    // so the pass over A^2 alone is skipped and the norms of A^2, A^3, A^4 are taken at once (1024 x 500, beta = 28.8: the A^2-only
    // pass was 1.4 of the 6.4 ms the exact norms cost a Hessian or an eval_constraint there)
    const bool straight = b.use_basis && bd.beta == bd.beta && bd.beta > 22.5;
    double d2 = exact_d2(h, b, dZ, st, straight);
    auto plan = [&] { return plan_hump(b, d2 == d2 ? std::min(bd.beta, d2) : d2); };
    if (!straight && d2 == d2 && plan().q > 1 && b.use_basis) d2 = exact_d2(h, b, dZ, st, true);  // ||A^3||, ||A^4|| sharpen the bound
    return plan();
}

// ------------------------------------------------------------------------------------------
// callbacks (device-pointer forms)
// ------------------------------------------------------------------------------------------

// reuse_forward_sweep: is dZ bit-identical to the point the cached sweeps were computed at?  If not, dZ becomes the new
// cache point and every integrator's cache is dropped.
bool same_point(dto_handle* h, const double* dZ, hipStream_t st) {
    if (!h->reuse) return false;
    if (!h->d_Zcache) {
        h->d_Zcache = own(h, dalloc<double>((size_t)h->n_vars));
        h->d_eq = own(h, dalloc<int32_t>(1));
        HIP_CHECK(hipMemcpyAsync(h->d_Zcache, dZ, sizeof(double) * (size_t)h->n_vars, hipMemcpyDeviceToDevice, st));
        for (auto& b : h->bil) { b.cache_kind = 0; b.p_terms = false; b.plan_q = 0; }
        return false;
    }
    int32_t* flag = reinterpret_cast<int32_t*>(h->h_pinned + 24);
    *flag = 1;
    HIP_CHECK(hipMemcpyAsync(h->d_eq, flag, sizeof(int32_t), hipMemcpyHostToDevice, st));
    launch_bits_equal(st, dZ, h->d_Zcache, h->n_vars, h->d_eq);
    HIP_CHECK(hipMemcpyAsync(flag, h->d_eq, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (*flag) return true;
    HIP_CHECK(hipMemcpyAsync(h->d_Zcache, dZ, sizeof(double) * (size_t)h->n_vars, hipMemcpyDeviceToDevice, st));
    for (auto& b : h->bil) { b.cache_kind = 0; b.p_terms = false; b.plan_q = 0; }
    return false;
}

// copy one of the caller's arrays of an external term (dto_set_external) into its device staging buffer
const double* ext_upload(dto_handle* h, int slot, int which, hipStream_t st) {
    ExtSlot& e = h->ext[slot];
    const double* src = which == 0 ? e.v.values : which == 1 ? e.v.first : e.v.second;
    static const char* what[3] = {"values", "first-derivative blocks", "second-derivative blocks"};
    if (!src)
        throw HipError{"external term " + std::to_string(slot) + ": " + what[which] +
                       " were not supplied (dto_set_external) -- the engine has no host fallback for closures"};
    if (!e.d[which]) e.d[which] = own(h, dalloc<double>(e.len[which]));
    HIP_CHECK(hipMemcpyAsync(e.d[which], src, e.len[which] * sizeof(double), hipMemcpyHostToDevice, st));
    return e.d[which];
}

// blocks of a device-evaluated time-dependent bilinear integrator: the owned intervals for the defect; for the Jacobian /
// Hessian also the interval left of the first owned knot, whose z_{k+1} half lands in that knot's columns
void tdb_eval(dto_handle* h, TdbHost& t, const double* dZ, const double* dmu, int need, hipStream_t st) {
    const KProb& P = h->P;
    const int64_t lo = need == 0 ? P.kn_lo : std::max<int64_t>(0, P.kn_lo - 1);
    const int64_t hi = need == 0 ? P.kn_lo + P.n_int : std::min<int64_t>(P.K, P.kn_lo + P.n_knots);
    HIP_CHECK(launch_tdb(st, P, t.k, dZ, dmu, need, lo, hi - lo, t.d_vals, t.d_jac, t.d_hess, t.d_scratch, t.stride));
}

// One launch per layer of an objective term's listings (a single layer unless its `times` repeats a knot): within a launch no
// two listings touch the same gradient / Hessian entry, across launches the stream orders them -- fixed order of addition.
template <class F>
void for_layers(dto_handle* h, size_t i, F&& f, bool first_only = false) {
    const KObj& o = h->obj[i];
    const std::vector<int64_t>& ls = h->obj_info[i].layer_start;
    for (size_t l = 0; l + 1 < ls.size() && !(first_only && l > 0); ++l) {
        KObj ol = o;
        const int64_t i0 = ls[l];
        ol.n_times = ls[l + 1] - i0;
        ol.times += i0;
        if (ol.Qs) ol.Qs += i0;
        if (ol.last) ol.last += i0;
        if (ol.params) ol.params += i0 * ol.n_comps;
        f(ol);
    }
}

void do_objective(dto_handle* h, const double* dZ, double* df, hipStream_t st) {
    HIP_CHECK(hipMemsetAsync(df, 0, sizeof(double), st));
    for (auto& o : h->obj) launch_objective(st, h->P, o, dZ, h->d_partial, df);
    for (auto& e : h->ext_obj)
        if (e.k.n_list > 0) launch_ext_objective(st, e.k, e.weight, ext_upload(h, e.ext_slot, 0, st), df);
}

void do_gradient(dto_handle* h, const double* dZ, double* dgrad, hipStream_t st) {
    HIP_CHECK(hipMemsetAsync(dgrad, 0, sizeof(double) * (size_t)h->info.grad_len, st));
    for (size_t i = 0; i < h->obj.size(); ++i)
        for_layers(h, i, [&](const KObj& ol) { launch_gradient(st, h->P, ol, dZ, dgrad); });
    for (auto& e : h->ext_obj)
        if (e.k.n_list > 0) launch_ext_gradient(st, h->P, e.k, e.weight, ext_upload(h, e.ext_slot, 1, st), dgrad);
}

// after a stored sweep of the p column alone: its terms stay valid for later callbacks at the same point
void remember_p_terms(dto_handle* h, BilHost& b, bool stored, int steps, hipStream_t st) {
    b.p_terms = stored && h->reuse;
    b.p_steps = steps;
    b.p_nblk = b.fw.nblk;
    if (stored)
        HIP_CHECK(hipMemcpyAsync(b.fw.nterms_p, b.fw.nterms, sizeof(int32_t) * b.fw.Kpad, hipMemcpyDeviceToDevice, st));
}

void do_constraint(dto_handle* h, const double* dZ, double* dg, hipStream_t st) {
    const bool same = same_point(h, dZ, st);
    for (auto& b : h->bil) {
        if (b.small) {
            HIP_CHECK(launch_small(st, h->P, b.k, b.d_Gs, make_types(0, false), make_types(0, false), dZ, nullptr, dg, nullptr, nullptr, 1));
            continue;
        }
        if (h->P.n_int > 0) {
            if (!(same && b.cache_kind >= 1) && s64_plans_itself(h, b, b.fw, make_types(0, false))) {
                // 33..64 states: the one-launch sweep takes its step budget from the norm bound ON THE DEVICE (k_plan_dev: the
                // formulas of cheap_plan / plan_sweep) -- the call no longer waits 40 us for eight bytes
                enqueue_bounds(h, b, dZ, st);
                launch_plan_dev(st, reinterpret_cast<const unsigned long long*>(h->d_bounds), h->d_plan);
                run_sweep(h, b, b.fw, make_types(0, false), dZ, nullptr, 0, 0, SweepPlan{1, 200}, st, false, false, false, false, CAT_SWEEP, h->d_plan);
                b.cache_kind = 0;
                remember_p_terms(h, b, false, 0, st);
            } else if (!(same && b.cache_kind >= 1)) {  // else exp(A)x of this very point is still in b.fw.S
                SweepPlan plan = plan_from(h, b, dZ, st, /*loose=*/true);
                if (plan.tc >= 0 && h->reuse && b.pairing && plan.d_ub + 1 > b.fw.dcap) plan = plan_from(h, b, dZ, st);  // the term store is what limits
                SweepTypes ty = make_types(0, false);
                // with reuse on, the terms of the p column are kept: a Jacobian at this point then sweeps its tangent
                // columns alone and a Hessian needs no forward sweep at all
                const bool keep_p = h->reuse && b.pairing && plan.q == 1 && plan.d_ub + 1 <= b.fw.dcap;
                const int steps = run_sweep(h, b, b.fw, ty, dZ, nullptr, 0, 0, plan, st, keep_p);
                b.cache_kind = h->reuse ? 1 : 0;
                remember_p_terms(h, b, keep_p, steps, st);
            }
            launch_cons_bilinear(st, h->P, b.k, b.fw, dZ, dg);
        }
    }
    for (auto& d : h->der) launch_cons_derivative(st, h->P, d, dZ, dg);
    for (size_t i = 0; i < h->ext_int.size(); ++i)
        if (h->P.n_int > 0) launch_extint_cons(st, h->P, h->ext_int[i], ext_upload(h, (int)i, 0, st), dg);
    for (auto& t : h->tdb)
        if (h->P.n_int > 0) { tdb_eval(h, t, dZ, nullptr, 0, st); launch_extint_cons(st, h->P, t.place, t.d_vals, dg); }
    for (auto& c : h->con) {
        if (!c.external) launch_cons_knot(st, h->P, c.k, dZ, dg);
        else if (c.k.n_times > 0) launch_ext_cons(st, c.k, ext_upload(h, c.ext_slot, 0, st), dg);
    }
}

void do_jacobian(dto_handle* h, const double* dZ, double* dvals, hipStream_t st) {
    // fill!(∂, 0), evaluator.jl:497 -- the -E_k block of a lone bilinear integrator is skipped: the chain
    // overwrites all of it
    // (enqueued inside the chain, where it fills the GPU while the host waits for the scaling decision)
    const bool lone = h->bil.size() == 1 && h->P.n_int > 0 && !h->bil[0].small;
    // bound and primed output (dto_bind_output_dev): constants are in place, clear only the runs kernels accumulate into
    const bool keep_constants = h->bound[0] == dvals && h->primed[0] && ensure_bind_runs(h, 0);
    if (keep_constants) launch_zero_runs(st, h->d_bind_start[0], h->d_bind_len[0], h->n_bind_runs[0], dvals);
    else if (!lone) {
        ProfScope ps(h, st, CAT_ZERO, 8.0 * (double)h->info.jac_len);
        HIP_CHECK(hipMemsetAsync(dvals, 0, sizeof(double) * (size_t)h->info.jac_len, st));
    }
    h->last_terms = 0;
    const bool same = same_point(h, dZ, st);
    for (auto& b : h->bil) {
        if (b.small) {
            HIP_CHECK(launch_small(st, h->P, b.k, b.d_Gs, make_types(b.k.m, false), make_types(0, false), dZ, nullptr, nullptr, dvals, nullptr, 2));
            continue;
        }
        Bounds bd{0, 0};
        if (h->P.n_int > 0) {
            // generator-norm bounds: enqueued here, read inside the chain's own readback point (no stream sync of their
            // own); the squaring cap they used to provide is the constant 60, a NaN iterate gets one squaring
            enqueue_bounds(h, b, dZ, st);
            // The sweep (one persistent launch, no HBM traffic to speak of) runs on the second stream next to the polynomial
            // products and squarings of the chain: its 223 workgroups leave 33 CUs idle and every chain launch has a tail --
            // together 5-6 % of the call (12.3 -> 11.5 ms at 256 x 2000).  Option "overlap_sweep" = 0 runs one kernel at a
            // time (per-kernel timings mean something only then).
            // (from 512 states on the step-per-launch sweep's small launches only get in the way of the chain's big ones:
            // 18.7 -> 19.3 ms at 512 x 500, 71.1 -> 72.5 ms at 1024 x 300; below, 3-6 % faster: tools/overlap_by_size.py)
            const bool overlap = h->overlap_sweep != 0 && b.k.npad < 512;
            hipStream_t ss = overlap ? h->stream2 : st;
            if (overlap) {
                HIP_CHECK(hipEventRecord(h->ev_fork, st));  // dZ (and the zero-filled slab) are ready here
                HIP_CHECK(hipStreamWaitEvent(ss, h->ev_fork, 0));
            }
            auto sweep_with = [&](SweepPlan plan, const int32_t* plan_dev = nullptr) {
                SweepTypes ty = make_types(b.k.m, false);
                // the p column of this very point is stored (eval_constraint or a Hessian came first)
                const bool have_p = same && b.p_terms && plan.q == 1;
                // ... but where the whole sweep runs as ONE persistent launch beside the chain, sweeping all columns again is
                // cheaper than the step-per-launch form the frozen variant needs (256 x 2000: 10.9 against 12.0 ms per Jacobian);
                // the stored p terms stay valid for a Hessian at this point either way (a sweep without store leaves them alone)
                ClusterSweepPlan cp_unused;
                GsSweepPlan gp_unused;
                const bool one_launch = fused_sweep_applies(h, b, b.fw, ty, plan, false) || cluster_sweep_applies(h, b, b.fw, ty, plan, false, cp_unused) ||
                                        gs_sweep_applies(h, b, b.fw, ty, plan, false, overlap && !h->deterministic, gp_unused);
                if (have_p && !one_launch) {
                    // sweep the tangent columns alone, their inhomogeneous terms read the stored p terms
                    SweepBuf wf = b.fw;
                    wf.frozen = b.fw.Zt;
                    wf.frozen_total = b.p_steps + 1;
                    wf.first_type = 1;
                    run_sweep(h, b, wf, ty, dZ, nullptr, 0, 0, plan, ss, false);
                    launch_apply_Gu(ss, b.k, b.fw, 0, b.fw.S, b.fw.GY);
                    b.cache_kind = 2;
                    return;
                }
                // with reuse on and a Hessian to follow, keep every Taylor term so that the Hessian can skip its forward sweep
                // (not when the p terms are there already: they are all the Hessian's pairing takes from the forward sweep)
                const bool keep = h->reuse && b.pairing && plan.q == 1 && plan.d_ub + 1 <= b.fw.dcap && !have_p;
                // (option "deterministic": the sweep keeps the shape it has when it runs alone, so the bits do not depend on
                // overlap_sweep; next to the chain the 256-state sweep otherwise groups its intervals by twelve instead of nine)
                // (a short shard's sweep in the generator-stationary form wants the chip to itself for a fraction of a millisecond: it
                // follows the chain on the call's stream instead of sharing the chip with it)
                GsSweepPlan gp_j;
                const bool gs_alone = gs_sweep_applies(h, b, b.fw, ty, plan, keep, false, gp_j);
                hipStream_t sw = gs_alone ? st : ss;
                const int steps = run_sweep(h, b, b.fw, ty, dZ, nullptr, 0, 0, plan, sw, keep, false, false,
                                            /*shared_chip=*/overlap && !h->deterministic && !gs_alone, CAT_SWEEP, plan_dev);
                launch_apply_Gu(sw, b.k, b.fw, 0, b.fw.S, b.fw.GY);
                b.cache_kind = h->reuse ? (keep ? 3 : 2) : 0;
                b.cache_steps = steps;
                if (keep || plan.q > 1) b.p_terms = false;  // the store now holds every column type / the scale factors changed
            };
            // One-launch chain (33..64 states): the chain's exact norms arrive only when ALL of it is done, so a sweep planned from
            // them would run behind it.  Where the cheap bound already gives a single round, the sweep is planned from that bound
            // (as eval_constraint and the Hessian do) and enqueued FIRST, on the second stream: its workgroups and the chain's
            // share the CUs (64 x 1000: 0.42 -> 0.3x ms per Jacobian).
            bool swept = same && b.cache_kind >= 2;  // the tangent sums of this very point are still in b.fw
            bool zeroed = false;
            static const int early_on = tune_int("DTO_SWEEP_EARLY", 1);  // A/B runs (TUNING builds)
            if (!swept && early_on && chain64_applies(h, b) && s64_plans_itself(h, b, b.fw, make_types(b.k.m, false))) {
                // ... and planned on the device: the whole call is enqueue-only
                launch_plan_dev(st, reinterpret_cast<const unsigned long long*>(h->d_bounds), h->d_plan);
                HIP_CHECK(hipEventRecord(h->ev_fork, st));   // the plan is ready here
                if (overlap) HIP_CHECK(hipStreamWaitEvent(ss, h->ev_fork, 0));
                sweep_with(SweepPlan{1, 200}, h->d_plan);
                swept = true;
                if (lone && !keep_constants) {
                    // the fill behind the sweep on ITS stream: only the tangent-column writers after the join need it, and the chain
                    // (which skips nothing the fill touches) starts 28 us earlier than with the fill in front of it
                    ProfScope ps(h, ss, CAT_ZERO, 8.0 * ((double)h->info.jac_len - (double)h->P.n_int * b.k.n * b.k.n));
                    launch_jac_zero(ss, h->P, b.k, dvals);
                    zeroed = true;
                }
            }
            if (!swept && early_on && chain64_applies(h, b)) {   // (with or without overlap: the plan, hence the bits, must not depend on it)
                HIP_CHECK(hipStreamSynchronize(st));
                bd = Bounds{h->h_pinned[0], h->h_pinned[1]};
                SweepPlan early;
                if (cheap_plan(bd, /*loose=*/true, early)) {
                    sweep_with(early);
                    swept = true;
                    if (lone && !keep_constants) {
                        // (the fill no longer has a host wait to hide in: it goes behind the sweep, on the sweep's stream)
                        ProfScope ps(h, ss, CAT_ZERO, 8.0 * ((double)h->info.jac_len - (double)h->P.n_int * b.k.n * b.k.n));
                        launch_jac_zero(ss, h->P, b.k, dvals);
                        zeroed = true;
                    }
                }
            }
            const bool nothing_waits = swept && chain64_applies(h, b);
            run_chain(h, b, dZ, dvals, INFINITY, st, [&](double d2) {
                bd = Bounds{h->h_pinned[0], h->h_pinned[1]};  // copied before the chain's readback event
                // ||A^t|| <= ||A^2||^floor(t/2) ||A||^(t mod 2): the exact d2 of the chain is the sharper
                // (and still rigorous) growth rate for the sweep's step budget
                if (swept) return;
                const SweepPlan from_chain = plan_hump(b, d2 == d2 ? std::min(bd.beta, d2) : d2);
                if (h->reuse) { b.plan_q = from_chain.q; b.plan_dub = from_chain.d_ub; }  // a Hessian at this very point need not buy the norms again
                sweep_with(from_chain);
            }, [&] {
                if (lone && !keep_constants && !zeroed) {
                    // every entry but the -E_k blocks, which the chain overwrites
                    ProfScope ps(h, st, CAT_ZERO, 8.0 * ((double)h->info.jac_len - (double)h->P.n_int * b.k.n * b.k.n));
                    launch_jac_zero(st, h->P, b.k, dvals);
                }
            }, nothing_waits);
            if (overlap) {
                HIP_CHECK(hipEventRecord(h->ev_join, ss));
                HIP_CHECK(hipStreamWaitEvent(st, h->ev_join, 0));
            }
        }
        {
            // tangent columns (m + 1 per interval: u_j and dt) and the identity of the z_{k+1} half
            ProfScope ps(h, st, CAT_ASSEMBLY, 8.0 * (double)h->P.n_int * b.k.n * (b.k.m + 2 + b.k.m + 1));
            launch_jac_bilinear(st, h->P, b.k, b.fw, dvals);
        }
    }
    for (auto& d : h->der) launch_jac_derivative(st, h->P, d, dZ, dvals);
    for (size_t i = 0; i < h->ext_int.size(); ++i)
        launch_extint_jac(st, h->P, h->ext_int[i], ext_upload(h, (int)i, 1, st), dvals);
    for (auto& t : h->tdb) { tdb_eval(h, t, dZ, nullptr, 1, st); launch_extint_jac(st, h->P, t.place, t.d_jac, dvals); }
    for (auto& c : h->con) {
        if (!c.external) launch_jac_knot(st, h->P, c.k, dZ, dvals);
        else if (c.k.n_times > 0) launch_ext_jac(st, c.k, ext_upload(h, c.ext_slot, 1, st), dvals);
    }
}

void do_hessian(dto_handle* h, const double* dZ, double sigma, const double* dmu, double* dH, hipStream_t st) {
    static const int zero_beside_on = tune_int("DTO_HESS_ZERO_BESIDE", 1);  // A/B runs (TUNING builds)
    // (worth an event round only for a large slab: 5.57 -> 5.49 ms at 256 x 2000, +4 us at 64 x 1000 and 256 x 250)
    const bool zero_beside = zero_beside_on && h->overlap_sweep != 0 && h->P.n_int > 0 && !h->bil.empty() && !h->bil[0].small &&
                             (size_t)h->info.hess_len * sizeof(double) >= ((size_t)256 << 20) &&
                             !h->integ_kind.empty() && h->integ_kind[0] == DTO_INTEGRATOR_BILINEAR;
    bool zero_joined = true;
    if (h->bound[1] == dH && h->primed[1] && ensure_bind_runs(h, 1))   // bound output: structural zeros are in place
        launch_zero_runs(st, h->d_bind_start[1], h->d_bind_len[1], h->n_bind_runs[1], dH);
    else if (zero_beside) {
        // fill!(H, 0), evaluator.jl:571 -- on the second stream: the fill is HBM-bound, the sweeps that open the bilinear block
        // are MFMA-bound and write nothing into H, so the 1.7 GB (256 x 2000) are cleared underneath them; the first kernel
        // that writes H waits for it (need_zero)
        HIP_CHECK(hipEventRecord(h->ev_fork, st));  // whatever used H before on this stream is done
        HIP_CHECK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
        {
            ProfScope ps(h, h->stream2, CAT_ZERO, 8.0 * (double)h->info.hess_len);
            HIP_CHECK(hipMemsetAsync(dH, 0, sizeof(double) * (size_t)h->info.hess_len, h->stream2));
        }
        HIP_CHECK(hipEventRecord(h->ev_zero, h->stream2));
        zero_joined = false;
    } else {
        ProfScope ps(h, st, CAT_ZERO, 8.0 * (double)h->info.hess_len);
        HIP_CHECK(hipMemsetAsync(dH, 0, sizeof(double) * (size_t)h->info.hess_len, st));  // fill!(H, 0), evaluator.jl:571
    }
    auto need_zero = [&] {
        if (!zero_joined) { HIP_CHECK(hipStreamWaitEvent(st, h->ev_zero, 0)); zero_joined = true; }
    };
    const bool same = same_point(h, dZ, st);
    // integrators in reference order (evaluator.jl:574-598)
    for (size_t i = 0; i < h->integ_kind.size(); ++i) {
        if (h->integ_kind[i] == DTO_INTEGRATOR_BILINEAR) {
            BilHost& b = h->bil[h->integ_index[i]];
            if (h->P.n_int <= 0) continue;
            if (b.small) {
                need_zero();
                HIP_CHECK(launch_small(st, h->P, b.k, b.d_Gs, make_types(b.k.m, true), make_types(b.k.m, false), dZ, dmu, nullptr, nullptr, dH, 4));
                continue;
            }
            // (the budget from the cheap bound serves while the term store holds it: the sweeps end by their own test, and the
            // pairing loops run over the terms actually produced -- the exact norms cost a store-less basis GEMM and two round trips)
            // (reuse_forward_sweep, same point as the last Jacobian: the chain's exact norms have planned that call's sweep already --
            // at 1024 states the passes that establish q = 1 for a Hessian on its own are 4.9 of its 29.5 ms)
            const bool planned = same && b.plan_q > 0;
            SweepPlan plan = planned ? SweepPlan{b.plan_q, b.plan_dub} : plan_from(h, b, dZ, st, /*loose=*/b.pairing);
            if (plan.tc >= 0 && plan.d_ub + 1 > b.fw.dcap) plan = plan_from(h, b, dZ, st);
            const bool pair = b.pairing && plan.q == 1 && plan.d_ub + 1 <= b.fw.dcap;
            const int m = b.k.m, T1 = 1 + m;
            int steps_f, Tf = T1;  // Tf: types per stored forward term
            SweepTypes ty1 = make_types(m, false);
            // Pairing path with a forward sweep of its own: the adjoint sweep is one persistent launch that leaves CUs idle, the
            // forward sweep of the p column a host-driven sequence of small launches -- independent until the pairing kernels,
            // so the adjoint sweep is enqueued first and the forward sweep runs next to it on the second stream.
            const bool fwd_needed = pair && !(same && (b.cache_kind == 3 || b.p_terms));
            // (where the adjoint sweep has no single-workgroup form -- short shards -- both sweeps take the generator-stationary form,
            // one after the other; beside a fused adjoint sweep the forward column keeps its step launches, which fit into the CUs
            // that sweep leaves idle: measured 5.5 against 5.9 ms at 256 x 2000 with the forward column first and alone)
            const bool side_by_side = fwd_needed && h->overlap_sweep && fused_sweep_applies(h, b, b.ad, ty1, plan, true);
            bool adjoint_enqueued = false;
            if (side_by_side) {
                HIP_CHECK(hipEventRecord(h->ev_fork, st));  // dZ, dmu and the zeroed slab are ready here
                HIP_CHECK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
                // (the adjoint sweep in the eight-wavefront shape it takes next to the chain, leaving more CUs to the forward
                // sweep's launches: 5.9 against 5.7 ms, gpurun_out/r03t)
                run_sweep(h, b, b.ad, ty1, dZ, dmu, 1, 1, plan, st, true, false, /*want_steps=*/false, false, CAT_SWEEP_ADJOINT);
                adjoint_enqueued = true;
            }
            hipStream_t sf = side_by_side ? h->stream2 : st;
            if (pair) {
                // Pairing path: every tangent comes from the ADJOINT sweep (the (x,u) block needs those anyway); of the forward
                // sweep only the Taylor terms of the p column are used (k_hess_pair, k_hess_bilinear).
                if (same && b.cache_kind == 3) {
                    steps_f = b.cache_steps;  // the Jacobian of this very point stored every forward term
                } else if (same && b.p_terms) {
                    Tf = 1;
                    steps_f = b.p_steps;      // eval_constraint (or an earlier Hessian) stored the p terms of this point
                } else {
                    Tf = 1;
                    steps_f = run_sweep(h, b, b.fw, make_types(0, false), dZ, nullptr, 0, 0, plan, sf, true, false, false, /*shared_chip=*/side_by_side);
                    b.cache_kind = h->reuse ? 1 : 0;  // the p sums are valid, the tangent sums are not
                    b.cache_steps = steps_f;
                    remember_p_terms(h, b, true, steps_f, sf);
                }
                launch_apply_generators(sf, b.k, b.fw, 0, b.fw.Zt, b.fw.W);  // V_l = G_l x (term 0 of the p column is x)
                if (side_by_side) {
                    HIP_CHECK(hipEventRecord(h->ev_join, sf));
                    HIP_CHECK(hipStreamWaitEvent(st, h->ev_join, 0));
                }
            } else {
                steps_f = run_sweep(h, b, b.fw, make_types(m, true), dZ, nullptr, 0, 0, plan, st, false);
                launch_apply_Gu(st, b.k, b.fw, 0, b.fw.S, b.fw.GY);
                b.cache_kind = 0;
                b.p_terms = false;  // this sweep re-initialised the scale factors for its own q
                // W_j = G_j' mu from the adjoint sweep's term-0 buffer
                launch_sweep_init(st, h->P, b.k, b.ad, make_types(0, false), dZ, dmu, 1, plan.q);
                launch_apply_generators(st, b.k, b.ad, 1, b.ad.Z[0], b.ad.W);
            }
            const int steps_a = adjoint_enqueued ? fused_sweep_steps(h, b.ad, plan.d_ub, st)
                                                 : run_sweep(h, b, b.ad, ty1, dZ, dmu, 1, 1, plan, st, pair, false, /*want_steps=*/pair, false, CAT_SWEEP_ADJOINT);
            if (h->profiling && pair)
                // the fused launch was priced by its step budget; now that the terms it ran are known, price it by those: 2 npad^2
                // (m+1) generator products per column and term, (1+m) column types (the flops bench.py's roofline uses)
                for (auto it = h->prof.rbegin(); it != h->prof.rend(); ++it)
                    if (it->cat == CAT_SWEEP_ADJOINT) {
                        if (it->flops > 0.0) it->flops = 2.0 * b.k.npad * (double)b.k.npad * b.ad.Kpad * (m + 1) * T1 * steps_a * plan.q;
                        break;
                    }
            launch_apply_Gu(st, b.k, b.ad, 1, b.ad.S, b.ad.GY);
            need_zero();
            launch_hess_bilinear(st, h->P, b.k, b.fw, b.ad, dmu, dH, pair ? 0 : 1);
            if (pair) {
                // (u_i,u_j) block from the stored Taylor terms (no second-order columns): Beta-weighted sums U_a of the
                // forward p terms, G_j U_a, then dot products with the adjoint tangent terms
                const int nf = steps_f + 1, na = steps_a + 1;
                const int64_t typesz = (int64_t)b.fw.Kpad * b.k.npad;
                const int64_t cols = (int64_t)na * b.fw.Kpad;  // one type of every stored term
                SweepBuf plain = b.fw;
                if (Tf == 1) { plain.nterms = b.fw.nterms_p; plain.nblk = b.p_nblk; }  // (a frozen Jacobian sweep in between re-used fw.nterms)
                launch_pair_combine(st, plain, Tf, 1, na, nf, b.ad.nterms, b.ad.nblk, b.d_Btab, b.Upair);
                {
                    ProfScope ps(h, st, CAT_SWEEP, 2.0 * b.k.npad * (double)b.k.npad * cols * m);
                    launch_apply_generators_cols(st, b.k, b.fw, 0, b.Upair, b.EP, 1, m, cols, b.fw.Kpad, (int64_t)Tf * typesz);
                }
                launch_hess_pair(st, h->P, b.k, b.ad, na, b.EP, dH);
            }
        } else if (h->integ_kind[i] == DTO_INTEGRATOR_DERIVATIVE) {
            need_zero();
            launch_hess_derivative(st, h->P, h->der[h->integ_index[i]], dmu, dH);
        } else if (h->integ_kind[i] == DTO_INTEGRATOR_TIME_DEPENDENT_BILINEAR) {
            need_zero();
            TdbHost& t = h->tdb[h->integ_index[i]];
            tdb_eval(h, t, dZ, dmu, 2, st);
            launch_extint_hess(st, h->P, t.place, t.d_hess, dH);
        } else {  // the caller's blocks already carry mu_k (eval_hessian_of_lagrangian(integrator, traj, mu_slice))
            need_zero();
            const int e = h->integ_index[i];
            launch_extint_hess(st, h->P, h->ext_int[e], ext_upload(h, e, 2, st), dH);
        }
    }
    need_zero();
    for (auto& c : h->con) {
        if (!c.external) launch_hess_knot(st, h->P, c.k, dZ, dmu, dH);
        else if (c.k.n_times > 0)  // the caller's blocks already carry mu_i (knot_point_constraint.jl:283-291)
            launch_ext_hess(st, h->P, c.xk, 1.0, ext_upload(h, c.ext_slot, 2, st), dH);
    }
    if (sigma != 0.0) {
        // get_full_hessian of the regularizers ASSIGNS its blocks per listed time (regularizers.jl:155-163, :305-309), so a knot
        // listed twice counts once -- unlike their value and gradient, which add per listing (:86-87, :102-108): first layer only
        for (size_t i = 0; i < h->obj.size(); ++i) {
            const bool assigns = h->obj[i].kind == DTO_OBJECTIVE_QUADRATIC_REGULARIZER || h->obj[i].kind == DTO_OBJECTIVE_LINEAR_REGULARIZER;
            for_layers(h, i, [&](const KObj& ol) { launch_hess_objective(st, h->P, ol, dZ, sigma, dH); }, assigns);
        }
        for (auto& e : h->ext_obj)
            if (e.k.n_list > 0) launch_ext_hess(st, h->P, e.k, sigma * e.weight, ext_upload(h, e.ext_slot, 2, st), dH);
    }
}


// ------------------------------------------------------------------------------------------
// host-pointer hand-off plans (dto_hostxfer.h): which slab entries a callback may write a call-dependent value to
// ------------------------------------------------------------------------------------------
void build_jac_plan(dto_handle* h) {
    XferPlan& p = h->jac_plan;
    const KProb& P = h->P;
    const int z = h->z;
    p.total = h->info.jac_len;
    auto var = [&](int64_t at, int64_t n, int64_t early = -1) {
        if (n > 0) { p.start.push_back(at); p.len.push_back(n); p.early.push_back((int32_t)early); }
    };
    auto one = [&](int64_t at, double v) { p.one_pos.push_back(at); p.one_val.push_back(v); };
    // the -E_k block of a lone general-path bilinear integrator is written by the propagator chain alone (do_jacobian): final
    // with its chain chunk
    const bool lone = h->bil.size() == 1 && P.n_int > 0 && !h->bil[0].small;
    for (int64_t kn = P.kn_lo; kn < P.kn_lo + P.n_knots; ++kn) {
        const int has_prev = kn >= 1, has_own = kn < h->K;
        const int cnt = has_prev + has_own;
        for (int j = 0; j < z; ++j) {
            const int64_t c = kn * z + j;
            const int64_t base = h->colptr[c] - P.jac_lo;
            int pre = 0;
            for (size_t i = 0; i < h->integ_kind.size(); ++i) {
                const int d = h->integ_dim[i];
                const int64_t prev_at = base + (int64_t)pre * cnt, own_at = prev_at + (has_prev ? d : 0);
                const int kind = h->integ_kind[i];
                if (kind == DTO_INTEGRATOR_BILINEAR) {
                    const KBil& b = h->bil[h->integ_index[i]].k;
                    const bool xcol = j >= b.x_off && j < b.x_off + b.n;
                    // z_{k+1} half: identity on the state columns, zeros elsewhere -- constant
                    if (has_prev && xcol) one(prev_at + (j - b.x_off), 1.0);
                    // own rows: -E_k (x), the tangents (u) and -G(u) E_k x (dt) change; every other column is a structural zero
                    if (has_own && (xcol || (j >= b.u_off && j < b.u_off + b.m) || j == h->dt_idx))
                        var(own_at, d, lone && xcol && j != h->dt_idx && kn - P.kn_lo < P.n_int ? kn - P.kn_lo : -1);
                } else if (kind == DTO_INTEGRATOR_DERIVATIVE) {
                    const KDer& dd = h->der[h->integ_index[i]];
                    const bool xcol = j >= dd.x_off && j < dd.x_off + dd.d, xdcol = j >= dd.xdot_off && j < dd.xdot_off + dd.d;
                    if (has_prev && xcol) one(prev_at + (j - dd.x_off), 1.0);
                    if (has_own) {
                        if (xdcol || j == h->dt_idx) var(own_at, d);         // -dt I and -xdot
                        else if (xcol) one(own_at + (j - dd.x_off), -1.0);   // -I (a column that is x AND xdot / dt is variable)
                    }
                } else {  // host-evaluated and time-dependent bilinear integrators: dense blocks, both halves
                    if (has_prev) var(prev_at, d);
                    if (has_own) var(own_at, d);
                }
                pre += d;
            }
            var(base + (int64_t)h->D * cnt, h->colptr[c + 1] - h->colptr[c] - (int64_t)h->D * cnt);  // constraint entries
        }
    }
    if (h->k_hi == h->N) {  // global-variable columns ride with the last knot: constraint entries only
        const int64_t c0 = h->N * z;
        var(h->colptr[c0] - P.jac_lo, h->colptr[h->n_vars] - h->colptr[c0]);
    }
    // the builders above emit per column in ascending order, but a derivative integrator's -1 may precede a later one's runs
    std::vector<size_t> idx(p.one_pos.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return p.one_pos[a] < p.one_pos[b]; });
    std::vector<int64_t> op(idx.size());
    std::vector<double> ov(idx.size());
    for (size_t i = 0; i < idx.size(); ++i) { op[i] = p.one_pos[idx[i]]; ov[i] = p.one_val[idx[i]]; }
    p.one_pos.swap(op);
    p.one_val.swap(ov);
}

void build_hess_plan(dto_handle* h) {
    XferPlan& p = h->hess_plan;
    const KProb& P = h->P;
    const int z = h->z;
    p.total = h->info.hess_len;
    const bool dense_blocks = !h->ext_int.empty() || !h->tdb.empty();  // their 2z x 2z blocks touch everything
    // listed times of the knot terms, per owned knot
    std::vector<std::vector<std::pair<int, int>>> extra((size_t)P.n_knots);  // (a <= b) pairs per local knot
    auto add_pairs = [&](int64_t kn, const std::vector<int32_t>& comps) {
        if (kn < P.kn_lo || kn >= P.kn_lo + P.n_knots) return;
        auto& e = extra[(size_t)(kn - P.kn_lo)];
        for (int a : comps)
            for (int b : comps) e.emplace_back(std::min(a, b), std::max(a, b));
    };
    for (auto& oi : h->obj_info) {
        for (int64_t kn : oi.times) {
            auto& e = extra[(size_t)(kn - P.kn_lo)];
            if (oi.kind == DTO_OBJECTIVE_QUADRATIC_REGULARIZER) {
                for (int c = 0; c < oi.comp_dim; ++c) {
                    const int a = oi.comp_off + c;
                    e.emplace_back(a, a);
                    if (a < h->dt_idx) e.emplace_back(a, h->dt_idx);
                }
                e.emplace_back(h->dt_idx, h->dt_idx);
            } else if (oi.kind == DTO_OBJECTIVE_LINEAR_REGULARIZER) {
                for (int c = 0; c < oi.comp_dim; ++c)
                    if (oi.comp_off + c <= h->dt_idx) e.emplace_back(oi.comp_off + c, h->dt_idx);
            } else if (oi.kind == DTO_OBJECTIVE_KNOT_SQDIST) {
                for (int a : oi.comps) e.emplace_back(a, a);
            } else if (oi.kind == DTO_OBJECTIVE_KNOT_LOWRANK_INFIDELITY) {
                add_pairs(kn, oi.comps);
            }
        }
    }
    for (auto& e : h->ext_obj)
        for (int64_t kn : e.times0) add_pairs(kn, e.comps);
    for (auto& c : h->con)
        if (!c.global)
            for (int64_t kn : c.times0) add_pairs(kn, c.comps);
    const int64_t tri = (int64_t)z * (z + 1) / 2;
    std::vector<uint8_t> mask((size_t)z * z);
    auto var = [&](int64_t at, int64_t n) { if (n > 0) { p.start.push_back(at); p.len.push_back(n); } };
    for (int64_t kn = P.kn_lo; kn < P.kn_lo + P.n_knots; ++kn) {
        std::fill(mask.begin(), mask.end(), dense_blocks ? 1 : 0);
        auto set = [&](int a, int b) { mask[(size_t)std::min(a, b) + (size_t)z * std::max(a, b)] = 1; };
        if (!dense_blocks) {
            if (kn < h->K) {
                for (auto& bh : h->bil) {
                    const KBil& b = bh.k;
                    for (int i = 0; i < b.n; ++i) {
                        for (int j = 0; j < b.m; ++j) set(b.x_off + i, b.u_off + j);
                        set(b.x_off + i, h->dt_idx);
                    }
                    for (int i = 0; i < b.m; ++i) {
                        for (int j = 0; j < b.m; ++j) set(b.u_off + i, b.u_off + j);
                        set(b.u_off + i, h->dt_idx);
                    }
                    set(h->dt_idx, h->dt_idx);
                }
                for (auto& dd : h->der)
                    for (int i = 0; i < dd.d; ++i) set(dd.xdot_off + i, h->dt_idx);
            }
            for (auto& ab : extra[(size_t)(kn - P.kn_lo)]) set(ab.first, ab.second);
        }
        const int64_t blk0 = hess_block_start(h, kn) - P.hess_lo;
        for (int b = 0; b < z; ++b) {
            int64_t col = blk0 + (kn == 0 ? (int64_t)b * (b + 1) / 2 : (int64_t)b * z + (int64_t)b * (b + 1) / 2);
            if (kn >= 1) {
                if (dense_blocks) var(col, z);  // off-diagonal block (kn-1, kn): the cross part of interval kn-1
                col += z;
            }
            int a = 0;
            while (a <= b) {
                while (a <= b && !mask[(size_t)a + (size_t)z * b]) ++a;
                const int a0 = a;
                while (a <= b && mask[(size_t)a + (size_t)z * b]) ++a;
                var(col + a0, a - a0);
            }
        }
        (void)tri;
    }
    if (h->k_hi == h->N) var(h->hess_block_nnz - P.hess_lo, (int64_t)h->tail_rows.size());  // global-column tail
}

void build_raw_plans(dto_handle* h) {
    if (h->raw_plans_built) return;
    h->raw_plans_built = true;
    build_jac_plan(h);
    if (h->eval_hessian) build_hess_plan(h);
}

// Bound outputs: the variable runs a callback must clear itself when the full zero-fill is skipped -- every run of the plan except
// the -E_k blocks, which the propagator chain overwrites entry by entry ("early" runs).
bool ensure_bind_runs(dto_handle* h, int which) {
    if (h->bind_ready[which]) return h->n_bind_runs[which] >= 0;
    h->bind_ready[which] = true;
    build_raw_plans(h);
    const XferPlan& p = which == 0 ? h->jac_plan : h->hess_plan;
    h->n_bind_runs[which] = -1;
    if (p.total <= 0) return false;
    std::vector<int64_t> st, ln;
    int64_t var_total = 0;
    for (size_t r = 0; r < p.start.size(); ++r) {
        var_total += p.len[r];
        if (!p.early.empty() && p.early[r] >= 0) continue;
        st.push_back(p.start[r]);
        ln.push_back(p.len[r]);
    }
    if (var_total * 10 > p.total * 9) return false;  // (nearly) everything varies: the plain zero-fill is as good
    h->d_bind_start[which] = own(h, dupload(st));
    h->d_bind_len[which] = own(h, dupload(ln));
    h->n_bind_runs[which] = (int64_t)st.size();
    return true;
}

void ensure_plans(dto_handle* h) {
    if (h->plans_built) return;
    h->plans_built = true;
    if (!h->host_xfer) return;
    build_raw_plans(h);
    for (XferPlan* p : {&h->jac_plan, &h->hess_plan}) {
        if (p->total <= 0) continue;
        p->finalize(h->P.n_int, HostXfer::CHUNK_DOUBLES);
        // shipping the runs pays only when a good part of the slab stays at home
        if (p->packed_total() * 10 > p->total * 9 || p->n_runs() == 0) continue;
        p->d_start = own(h, dupload(p->pk_start));
        p->d_len = own(h, dupload(p->pk_len));
        p->d_poff = own(h, dupload(p->pk_poff));
        p->d_packed = own(h, dalloc<double>((size_t)std::max<int64_t>(p->packed_total(), 1)));
    }
    if (h->jac_plan.usable() || h->hess_plan.usable()) h->xfer.reset(new HostXfer());
}

double* staging(dto_handle* h, size_t n) {
    if (n > h->d_out_cap) {
        h->d_out = own(h, dalloc<double>(n));
        h->d_out_cap = n;
    }
    return h->d_out;
}

void drop_caches(dto_handle* h) {
    for (auto& b : h->bil) { b.cache_kind = 0; b.p_terms = false; b.plan_q = 0; }
}
void unprime(dto_handle* h) { h->primed[0] = h->primed[1] = false; }

// Enqueue, behind the kernels of an asynchronous call, the copy of every sweep's statistics to pinned memory.
void enqueue_stats(dto_handle* h, hipStream_t st) {
    if (!h->h_stats) return;
    size_t i = 0;
    bool any = false;
    for (auto& b : h->bil)
        for (SweepBuf* w : {&b.fw, &b.ad}) {
            if (w->stats) {
                HIP_CHECK(hipMemcpyAsync(h->h_stats + 2 * i, w->stats, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
                any = true;
            }
            ++i;
        }
    if (!any) return;
    HIP_CHECK(hipEventRecord(h->ev_done, st));
    h->stats_pending = true;
}
// Look at the statistics of the last call (waits for them): a sweep that ran out of its step budget is an error.
void check_sweeps(dto_handle* h, bool wait = true) {
    if (!h->stats_pending) return;
    // a call that runs no sweep of its own (objective, gradient) does not wait for the previous asynchronous call: it looks
    // only if that call has finished, and otherwise leaves the check to the next call that would overwrite the statistics
    if (!wait && hipEventQuery(h->ev_done) != hipSuccess) { (void)hipGetLastError(); return; }
    h->stats_pending = false;
    HIP_CHECK(hipEventSynchronize(h->ev_done));
    if (h->smax_pending) { h->last_smax = *reinterpret_cast<const int32_t*>(h->h_pinned + 28); h->smax_pending = false; }
    size_t i = 0;
    bool bad = false;
    for (auto& b : h->bil)
        for (SweepBuf* w : {&b.fw, &b.ad}) {
            if (w->stats) {
                h->last_terms = std::max(h->last_terms, h->h_stats[2 * i + 1]);
                if (h->h_stats[2 * i] != 0) bad = true;
            }
            ++i;
        }
    if (bad) {
        drop_caches(h);  // whatever the sweeps left behind is not a converged result
        throw HipError{"generator sweep did not converge within its step budget"};
    }
}

// Every entry point that evaluates runs through here.  ASYNC: device-pointer form on stream `st` -- device-side errors are
// reported by the NEXT call (check_sweeps at entry); BLOCKING: host-pointer form, checked before it returns; PLAIN: no sweeps.
enum { G_PLAIN = 0, G_ASYNC = 1, G_BLOCKING = 2 };
template <class F>
int guarded(dto_handle* h, F&& f, int mode = G_PLAIN, hipStream_t st = nullptr) {
    if (!h) return fail(nullptr, "null handle");
    if (h->structure_only) return fail(h, "structure-only handle (created with device < 0): no evaluation without a GPU");
    try {
        HIP_CHECK(hipSetDevice(h->device));
        check_sweeps(h, mode != G_PLAIN);  // deferred error of the previous asynchronous call, if any
        (void)hipGetLastError();    // the launches below are judged on their own
        f();
        // kernel launches report a rejected configuration through the runtime's last-error slot, not a return value
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) throw HipError{std::string("kernel launch failed: ") + hipGetErrorString(le)};
        if (mode == G_BLOCKING) { enqueue_stats(h, h->stream); check_sweeps(h); }
        else if (mode == G_ASYNC) enqueue_stats(h, st);
        return 0;
    } catch (const HipError& e) {
        drop_caches(h);
        h->stats_pending = false;
        h->smax_pending = false;
        return fail(h, e.msg);
    } catch (const std::exception& e) {
        drop_caches(h);
        h->stats_pending = false;
        return fail(h, e.what());
    }
}

// ---- multi-GPU: ranges, gather plans, collectives (dto_comm.h)
void set_ranges(dto_handle* h, int rank, int world, const int64_t* k_lo, const int64_t* k_hi) {
    if (world < 1 || rank < 0 || rank >= world) throw HipError{"dto_comm: bad rank / world"};
    std::vector<std::pair<int64_t, int64_t>> rr;
    for (int r = 0; r < world; ++r) {
        if (k_lo[r] < 1 || k_hi[r] < k_lo[r] || k_hi[r] > h->N) throw HipError{"dto_comm: a rank's knot range is out of bounds"};
        rr.emplace_back(k_lo[r], k_hi[r]);
    }
    if (rr[rank].first != h->k_lo || rr[rank].second != h->k_hi)
        throw HipError{"dto_comm: this rank's entry of the knot ranges is not the handle's own shard"};
    std::vector<Slab> sl[3];
    for (int r = 0; r < world; ++r) {
        const ShardExtents e = shard_extents(h, rr[r].first, rr[r].second);
        sl[0].push_back(Slab{e.jac_lo, e.jac_len});
        sl[1].push_back(Slab{e.hess_lo, e.hess_len});
        sl[2].push_back(Slab{e.grad_lo, e.grad_len});
    }
    h->gather_plan[0] = make_gather_plan(sl[0], h->jac_nnz);
    h->gather_plan[1] = make_gather_plan(sl[1], h->hess_nnz);
    h->gather_plan[2] = make_gather_plan(sl[2], h->n_vars);
    h->cons_segments.clear();
    h->cons_segment_root.clear();
    for (int r = 0; r < world; ++r)
        for (auto& sg : shard_row_segments(h, rr[r].first, rr[r].second)) {
            h->cons_segments.push_back(Slab{sg.first, sg.second});
            h->cons_segment_root.push_back(r);
        }
    h->rank_ranges.swap(rr);
    h->comm_rank = rank;
}

const GatherPlan& plan_of(const dto_handle* h, int vector) {
    if (h->rank_ranges.empty()) throw HipError{"no knot ranges yet: dto_comm_create / dto_comm_set_ranges first"};
    if (vector < DTO_VECTOR_JACOBIAN || vector > DTO_VECTOR_GRADIENT) throw HipError{"dto_gather: unknown vector kind"};
    return h->gather_plan[vector - 1];
}

void gather_vector(dto_handle* h, int vector, double* dbuf, hipStream_t st) {
    const GatherPlan& pl = plan_of(h, vector);
    if (!h->comm) throw HipError{"dto_gather: no communicator (dto_comm_create)"};
    if (vector == DTO_VECTOR_HESSIAN && !h->eval_hessian) throw HipError{"handle was created with eval_hessian = 0"};
    std::string e;
    if (pl.in_place) {
        e = h->comm->all_gather_in_place(dbuf, pl.n, st);
    } else {
        std::vector<int> root(pl.slabs.size());
        for (size_t r = 0; r < root.size(); ++r) root[r] = (int)r;
        e = h->comm->broadcast_slabs(dbuf, pl.slabs, root, st);
    }
    if (!e.empty()) throw HipError{e};
}

// the comm entry points touch no sweep state: errors come back at once, nothing is deferred
template <class F>
int comm_guarded(dto_handle* h, F&& f, bool needs_device = true) {
    if (!h) return fail(nullptr, "null handle");
    try {
        if (needs_device) {
            if (h->structure_only) throw HipError{"structure-only handle (created with device < 0): no collectives without a GPU"};
            HIP_CHECK(hipSetDevice(h->device));
        }
        f();
        return 0;
    } catch (const HipError& e) {
        return fail(h, e.msg);
    } catch (const std::exception& e) {
        return fail(h, e.what());
    }
}

// Option "host_xfer_check": the hand-off plans (build_jac_plan / build_hess_plan) are a second statement of which slab entries
// a callback may write; a kernel that writes outside them would be dropped silently on the host-pointer path.  With the
// option on, every such call also copies the WHOLE device slab and compares it bit for bit with what was assembled.
void check_against_slab(dto_handle* h, const double* d_slab, const double* assembled, size_t n, const char* what) {
    std::vector<double> full(n);
    HIP_CHECK(hipMemcpy(full.data(), d_slab, n * sizeof(double), hipMemcpyDeviceToHost));
    if (memcmp(full.data(), assembled, n * sizeof(double)) == 0) return;
    size_t i = 0;
    while (i < n && memcmp(&full[i], &assembled[i], sizeof(double)) == 0) ++i;
    char buf[256];
    snprintf(buf, sizeof(buf), "host_xfer_check: %s entry %zu is %.17g on the device and %.17g in the caller's vector -- a kernel "
             "wrote outside the hand-off plan", what, i, full[i], assembled[i]);
    throw HipError{buf};
}

void upload_Z(dto_handle* h, const double* Z) {
    HIP_CHECK(hipMemcpyAsync(h->d_Z, Z, sizeof(double) * (size_t)h->n_vars, hipMemcpyHostToDevice, h->stream));
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================

extern "C" {

const char* dto_last_error(const dto_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void dto_destroy(dto_handle* h) { delete h; }

int dto_create(const dto_problem_desc* d, dto_handle** out) {
    if (!d || !out) return fail(nullptr, "dto_create: null argument");
    *out = nullptr;
    if (d->abi_version != DTO_ABI_VERSION) return fail(nullptr, "dto_create: ABI version mismatch");
    if (d->N < 2) return fail(nullptr, "dto_create: need at least 2 knots");
    if (d->z < 1 || d->gd < 0) return fail(nullptr, "dto_create: bad dimensions");
    if (d->dt_idx < 0 || d->dt_idx >= d->z)
        return fail(nullptr, "dto_create: the timestep must be a trajectory component (bilinear_integrator.jl:123)");
    if (!d->Z0) return fail(nullptr, "dto_create: Z0 is required (constraint patterns are taken at Z0)");
    const bool sonly = d->device < 0;  // structure-only handle: no GPU is touched, evaluations fail
    if (!sonly) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            return fail(nullptr, "dto_create: no HIP device available (the engine has no CPU fallback)");
        if (d->device >= ndev) return fail(nullptr, "dto_create: bad device ordinal");
    }

    dto_handle* h = new dto_handle();
    try {
        h->device = d->device;
        h->structure_only = sonly;
        if (!sonly) {
            HIP_CHECK(hipSetDevice(h->device));
            HIP_CHECK(hipStreamCreate(&h->stream));
            HIP_CHECK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
            HIP_CHECK(hipStreamCreateWithFlags(&h->stream_rb, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&h->ev_rb, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&h->ev_zero, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&h->ev_stats, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&h->ev_chain, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
            HIP_CHECK(hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, h->device));
            HIP_CHECK(sweep_fused_prepare());
            HIP_CHECK(sweep_cluster_prepare());
            HIP_CHECK(sweep_gs_prepare());
            HIP_CHECK(chain64_prepare());
        }
        h->N = d->N; h->K = d->N - 1; h->z = d->z; h->gd = d->gd; h->dt_idx = d->dt_idx;
        h->eval_hessian = d->eval_hessian;
        h->n_vars = (int64_t)d->z * d->N + d->gd;
        h->k_lo = d->k_lo > 0 ? d->k_lo : 1;
        h->k_hi = d->k_hi > 0 ? d->k_hi : d->N;
        if (h->k_lo > h->k_hi || h->k_hi > h->N) throw HipError{"dto_create: bad knot shard"};

        // integrators: rows stacked in list order (evaluator.jl:211-217)
        int pre = 0;
        int64_t row = 0;
        for (int i = 0; i < d->n_integrators; ++i) {
            const dto_integrator_desc& s = d->integrators[i];
            if (s.x_dim < 1 || (s.kind != DTO_INTEGRATOR_EXTERNAL && (s.x_off < 0 || s.x_off + s.x_dim > d->z)))
                throw HipError{"integrator: bad state range"};
            h->integ_kind.push_back(s.kind);
            h->integ_dim.push_back(s.x_dim);
            h->integ_row_off.push_back(row);
            if (s.kind == DTO_INTEGRATOR_BILINEAR) {
                if (s.u_dim < 0 || s.u_dim > MAX_DRIVES) throw HipError{"bilinear integrator: supports 0..7 drives"};
                if (s.u_dim > 0 && (s.u_off < 0 || s.u_off + s.u_dim > d->z)) throw HipError{"bilinear integrator: bad control range"};
                if (!s.G) throw HipError{"bilinear integrator: G is null"};
                BilHost b;
                b.k.n = s.x_dim; b.k.m = s.u_dim; b.k.npad = pad64(s.x_dim);
                b.k.x_off = s.x_off; b.k.u_off = s.u_off; b.k.pre = pre; b.k.row_off = row;
                const int n = s.x_dim, np = b.k.npad, m1 = s.u_dim + 1;
                // (+ 16 zero columns: the fused sweep streams the generators a few k-steps ahead, past the last one)
                std::vector<double> G((size_t)m1 * np * np + 16 * (size_t)np, 0.0), GT((size_t)m1 * np * np + 16 * (size_t)np, 0.0);
                b.g1.assign(m1, 0.0);
                for (int j = 0; j < m1; ++j)
                    for (int c = 0; c < n; ++c) {
                        double cs = 0.0;
                        for (int r = 0; r < n; ++r) {
                            const double v = s.G[(size_t)j * n * n + (size_t)c * n + r];
                            G[(size_t)j * np * np + (size_t)c * np + r] = v;
                            GT[(size_t)j * np * np + (size_t)r * np + c] = v;
                            cs += std::fabs(v);
                        }
                        b.g1[j] = std::max(b.g1[j], cs);
                    }
                if (!sonly) {
                    b.k.G = own(h, dupload(G));
                    b.k.GT = own(h, dupload(GT));
                    const bool small_on = (d->flags & DTO_FLAG_GENERAL_PATH_ONLY) == 0;
                    // fused one-workgroup-per-interval path: n <= 16 with one wavefront, 17..32 with four, while the
                    // interval's matrices, generators and sweep columns fit the CU's LDS
                    const int mm_ = s.u_dim, Tf = d->eval_hessian ? 1 + mm_ + mm_ * (mm_ + 1) / 2 : 1 + mm_;
                    if (small_on && n <= 32 && Tf <= MAX_TYPES && small_lds_bytes(n, mm_, Tf, 1 + mm_) <= 150 * 1024) {
                        b.small = true;
                        // worst-case dynamic LDS of this handle's fused kernel, opted into on THIS device
                        HIP_CHECK(small_prepare(small_lds_bytes(n, mm_, Tf, 1 + mm_)));
                        b.d_Gs = own(h, dupload(std::vector<double>(s.G, s.G + (size_t)m1 * n * n)));
                    }
                }
                h->integ_index.push_back((int)h->bil.size());
                h->bil.push_back(std::move(b));
            } else if (s.kind == DTO_INTEGRATOR_DERIVATIVE) {
                if (s.u_off < 0 || s.u_off + s.x_dim > d->z) throw HipError{"derivative integrator: bad derivative range"};
                KDer k{};
                k.d = s.x_dim; k.x_off = s.x_off; k.xdot_off = s.u_off; k.pre = pre; k.row_off = row;
                h->integ_index.push_back((int)h->der.size());
                h->der.push_back(k);
            } else if (s.kind == DTO_INTEGRATOR_TIME_DEPENDENT_BILINEAR) {
                if (s.u_dim < 0 || s.u_dim > MAX_DRIVES) throw HipError{"time-dependent bilinear integrator: supports 0..7 drives"};
                if (s.u_dim > 0 && (s.u_off < 0 || s.u_off + s.u_dim > d->z)) throw HipError{"time-dependent bilinear integrator: bad control range"};
                if (s.t_off < 0 || s.t_off >= d->z) throw HipError{"time-dependent bilinear integrator: bad time component"};
                if (s.spline_order != 0 && s.spline_order != 1) throw HipError{"Unsupported spline order (0 or 1)"};
                if (!s.G || (s.n_mod > 0 && (!s.H || !s.mod_kind || !s.mod_omega))) throw HipError{"time-dependent bilinear integrator: G / H / modulation arrays are null"};
                TdbHost t;
                t.k.n = s.x_dim; t.k.m = s.u_dim; t.k.x_off = s.x_off; t.k.u_off = s.u_off; t.k.t_off = s.t_off;
                t.k.order = s.spline_order; t.k.substeps = s.substeps; t.k.nmod = s.n_mod; t.k.row_off = row;
                if (!tdb_supported(t.k)) throw HipError{"time-dependent bilinear integrator: outside the device kernel's range (1..64 states, substeps >= 1, coefficient table)"};
                for (int c = 0; c < s.n_mod; ++c)
                    if (s.mod_kind[c] != 1 && s.mod_kind[c] != 2) throw HipError{"time-dependent bilinear integrator: mod_kind is 1 (cos) or 2 (sin)"};
                t.place.d = s.x_dim; t.place.pre = pre; t.place.row_off = row;
                if (!sonly) {
                    const size_t nn = (size_t)s.x_dim * s.x_dim, m1 = (size_t)s.u_dim + 1;
                    t.k.G = own(h, dupload(std::vector<double>(s.G, s.G + m1 * nn)));
                    if (s.n_mod > 0) {
                        t.k.H = own(h, dupload(std::vector<double>(s.H, s.H + (size_t)s.n_mod * m1 * nn)));
                        t.k.mod_kind = own(h, dupload(std::vector<int32_t>(s.mod_kind, s.mod_kind + s.n_mod)));
                        t.k.mod_omega = own(h, dupload(std::vector<double>(s.mod_omega, s.mod_omega + s.n_mod)));
                    }
                }
                h->integ_index.push_back((int)h->tdb.size());
                h->tdb.push_back(t);
            } else if (s.kind == DTO_INTEGRATOR_EXTERNAL) {
                KExtInt e{};
                e.d = s.x_dim; e.pre = pre; e.row_off = row;
                h->integ_index.push_back((int)h->ext_int.size());
                h->ext_int.push_back(e);
                ExtSlot sl;
                sl.len[0] = (size_t)s.x_dim * h->K;
                sl.len[1] = (size_t)s.x_dim * 2 * d->z * h->K;
                sl.len[2] = (size_t)4 * d->z * d->z * h->K;
                h->ext.push_back(sl);
            } else {
                throw HipError{"unknown integrator kind"};
            }
            pre += s.x_dim;
            row += (int64_t)s.x_dim * h->K;
        }
        h->D = pre;
        h->n_dyn = row;
        h->n_ext_int = (int)h->ext_int.size();
        if (h->integ_kind.size() > 8) throw HipError{"at most 8 integrators"};

        // nonlinear knot constraints: rows follow the dynamics (evaluator.jl:219-223)
        for (int i = 0; i < d->n_constraints; ++i) {
            const dto_constraint_desc& s = d->constraints[i];
            if (s.kind != DTO_CONSTRAINT_NORM_MINUS_C && s.kind != DTO_CONSTRAINT_SQNORM_MINUS_C && s.kind != DTO_CONSTRAINT_EXTERNAL &&
                s.kind != DTO_CONSTRAINT_EXTERNAL_GLOBAL)
                throw HipError{"unknown constraint kind"};
            if (s.kind == DTO_CONSTRAINT_EXTERNAL_GLOBAL) {
                // NonlinearGlobalConstraint: one listing at the pseudo-knot N whose "components" are global_data entries
                if (s.n_comps < 1 || !s.comps || s.g_dim < 1 || !s.jac0 || !s.hess0)
                    throw HipError{"global constraint: comps, g_dim, jac0 and hess0 are required"};
                ConHost c;
                c.k.kind = s.kind; c.k.n_comps = s.n_comps; c.equality = s.equality;
                c.external = true; c.global = true; c.g_dim = s.g_dim;
                c.k.g_dim = c.g_dim; c.k.external = 1;
                c.comps.assign(s.comps, s.comps + s.n_comps);
                for (int q : c.comps)
                    if (q < 0 || q >= d->gd) throw HipError{"global constraint: global component out of range"};
                c.jac0.assign(s.jac0, s.jac0 + (size_t)s.g_dim * s.n_comps);
                c.hess0.assign(s.hess0, s.hess0 + (size_t)s.n_comps * s.n_comps);
                c.n_times_total = 1;
                c.times0.push_back(d->N);
                c.row_off = row;
                row += c.g_dim;
                h->con.push_back(std::move(c));
                continue;
            }
            if (s.n_comps < 1 || !s.comps || (!s.times && s.n_times > 0)) throw HipError{"constraint: bad description"};
            ConHost c;
            c.k.kind = s.kind; c.k.n_comps = s.n_comps; c.k.c = s.c;
            c.equality = s.equality;
            if (s.kind == DTO_CONSTRAINT_EXTERNAL) {
                if (s.g_dim < 1) throw HipError{"external constraint: g_dim must be >= 1"};
                if (!s.jac0) throw HipError{"external constraint: jac0 (Jacobian blocks at Z0) is required for the sparsity pattern"};
                c.external = true;
                c.g_dim = s.g_dim;
                c.jac0.assign(s.jac0, s.jac0 + (size_t)s.g_dim * s.n_comps * s.n_times);
            } else if (s.g_dim > 1) {
                throw HipError{"constraint: the built-in kinds have g_dim = 1"};
            }
            c.k.g_dim = c.g_dim; c.k.external = c.external ? 1 : 0;
            c.comps.assign(s.comps, s.comps + s.n_comps);
            for (int q : c.comps)
                if (q < 0 || q >= d->z) throw HipError{"constraint: component out of range"};
            c.n_times_total = s.n_times;
            for (int64_t t = 0; t < s.n_times; ++t) {
                if (s.times[t] < 1 || s.times[t] > d->N) throw HipError{"constraint: time out of range"};
                c.times0.push_back(s.times[t] - 1);
            }
            c.row_off = row;
            row += s.n_times * c.g_dim;
            h->con.push_back(std::move(c));
        }
        h->n_cons = row;
        for (auto& c : h->con)
            if (c.external) {
                c.ext_slot = (int)h->ext.size();  // after the external integrators
                ExtSlot e;
                e.len[0] = (size_t)c.g_dim * c.n_times_total;
                e.len[1] = (size_t)c.g_dim * c.comps.size() * c.n_times_total;
                e.len[2] = c.comps.size() * c.comps.size() * (size_t)c.n_times_total;
                h->ext.push_back(e);
            }
        h->n_ext_con = (int)h->ext.size() - h->n_ext_int;
        // host-evaluated objective terms: their slots follow, and the Global* ones shape the Hessian structure
        for (int i = 0; i < d->n_objectives; ++i) {
            const dto_objective_desc& s = d->objectives[i];
            if (s.kind != DTO_OBJECTIVE_EXTERNAL_KNOT && s.kind != DTO_OBJECTIVE_EXTERNAL_GLOBAL) continue;
            ExtObjHost e;
            e.weight = s.weight;
            e.global = s.kind == DTO_OBJECTIVE_EXTERNAL_GLOBAL;
            if (s.n_comps > 0) {
                if (!s.comps) throw HipError{"external objective: comps is null"};
                e.comps.assign(s.comps, s.comps + s.n_comps);
            }
            for (int q : e.comps)
                if (q < 0 || q >= d->z) throw HipError{"external objective: component out of range"};
            if (e.global) {
                if (s.n_gcomps < 1 || !s.gcomps) throw HipError{"global objective: gcomps are required"};
                e.gcomps.assign(s.gcomps, s.gcomps + s.n_gcomps);
                for (int q : e.gcomps)
                    if (q < 0 || q >= d->gd) throw HipError{"global objective: global component out of range"};
            } else if (e.comps.empty() || !s.times) {
                throw HipError{"external knot objective: comps and times are required"};
            }
            if (s.n_times > 0 && !s.times) throw HipError{"external objective: times is null"};
            for (int64_t t = 0; t < s.n_times; ++t) {
                if (s.times[t] < 1 || s.times[t] > d->N) throw HipError{"objective: time out of range"};
                e.times0.push_back(s.times[t] - 1);
            }
            if (e.times0.empty()) {  // GlobalObjective: the global variables alone
                if (!e.comps.empty()) throw HipError{"global objective: knot components without times"};
                e.times0.push_back(d->N);
            }
            const size_t nb = e.comps.size() + e.gcomps.size(), nl = e.times0.size();
            e.ext_slot = (int)h->ext.size();
            ExtSlot sl;
            sl.len[0] = nl; sl.len[1] = nb * nl; sl.len[2] = nb * nb * nl;
            h->ext.push_back(sl);
            h->ext_obj.push_back(std::move(e));
        }
        h->n_ext_obj = (int)h->ext_obj.size();

        // shard
        KProb& P = h->P;
        P.N = h->N; P.K = h->K; P.z = h->z; P.dt_idx = h->dt_idx; P.D = h->D;
        P.kn_lo = h->k_lo - 1;
        P.n_knots = h->k_hi - h->k_lo + 1;
        P.n_int = std::max<int64_t>(0, std::min<int64_t>(h->k_hi, h->K) - h->k_lo + 1);

        build_structure(h, d->Z0);
        if (!sonly) {
            h->d_colptr = own(h, dupload(h->colptr));
            P.colptr = h->d_colptr;
        }
        const ShardExtents ext = shard_extents(h, h->k_lo, h->k_hi);
        P.jac_lo = ext.jac_lo;
        P.hess_lo = ext.hess_lo;
        P.grad_lo = ext.grad_lo;
        dto_shard_info& I = h->info;
        I.k_lo = h->k_lo; I.k_hi = h->k_hi;
        I.n_vars = h->n_vars; I.n_cons = h->n_cons; I.jac_nnz = h->jac_nnz; I.hess_nnz = h->hess_nnz;
        I.grad_lo = ext.grad_lo; I.grad_len = ext.grad_len;
        I.jac_lo = ext.jac_lo; I.jac_len = ext.jac_len;      // global columns ride with the last knot
        I.hess_lo = ext.hess_lo; I.hess_len = ext.hess_len;
        P.tail_lo = h->k_hi == h->N ? h->hess_block_nnz - P.hess_lo : -1;
        if (!sonly) {
            P.tail_colptr = own(h, dupload(h->tail_colptr));
            P.tail_rows = own(h, dupload(h->tail_rows));
        }

        // local constraint rows: integrators first, then constraints
        int64_t lrow = 0;
        for (size_t i = 0; i < h->integ_kind.size(); ++i) {
            const int dd = h->integ_dim[i];
            if (h->integ_kind[i] == DTO_INTEGRATOR_BILINEAR) h->bil[h->integ_index[i]].k.lrow_off = lrow;
            else if (h->integ_kind[i] == DTO_INTEGRATOR_DERIVATIVE) h->der[h->integ_index[i]].lrow_off = lrow;
            else if (h->integ_kind[i] == DTO_INTEGRATOR_TIME_DEPENDENT_BILINEAR) h->tdb[h->integ_index[i]].place.lrow_off = lrow;
            else h->ext_int[h->integ_index[i]].lrow_off = lrow;
            lrow += P.n_int * dd;
        }
        for (auto& c : h->con) {
            std::vector<int64_t> times, lrows, tidx, jpos;
            std::vector<int32_t> hess_on;
            for (int64_t i = 0; i < c.n_times_total; ++i) {
                const int64_t kn = c.times0[i];
                if (kn >= h->N ? h->k_hi != h->N : (kn < P.kn_lo || kn >= P.kn_lo + P.n_knots)) continue;  // pseudo-knot N: last rank
                times.push_back(kn);
                lrows.push_back(lrow);
                lrow += c.g_dim;
                tidx.push_back(i);
                {
                    int32_t last = 1;
                    for (int64_t i2 = i + 1; i2 < c.n_times_total; ++i2)
                        if (c.times0[i2] == kn) { last = 0; break; }
                    hess_on.push_back(last);
                }
                for (size_t q = 0; q < c.comps.size(); ++q) {
                    const int64_t col = kn * h->z + c.comps[q];
                    const size_t lo = con_lower(h, col);
                    for (int r = 0; r < c.g_dim; ++r) {
                        int64_t pos = -1;
                        for (size_t e = lo; e < h->con_cols.size() && h->con_cols[e] == col; ++e)
                            if (h->con_rows[e] == c.row_off + i * c.g_dim + r) {
                                pos = h->colptr[col] + (int64_t)h->D * col_cnt(h, kn) + (int64_t)(e - lo) - P.jac_lo;
                                break;
                            }
                        jpos.push_back(pos);
                    }
                }
            }
            c.k.n_times = (int64_t)times.size();
            c.k.mu_off = c.row_off;
            {
                std::vector<int64_t> st = times;
                std::sort(st.begin(), st.end());
                c.k.repeats = std::adjacent_find(st.begin(), st.end()) != st.end() ? 1 : 0;
                std::vector<int32_t> sc = c.comps;
                std::sort(sc.begin(), sc.end());
                c.k.comp_repeats = std::adjacent_find(sc.begin(), sc.end()) != sc.end() ? 1 : 0;
            }
            if (!sonly) {
                c.k.comps = own(h, dupload(c.comps));
                c.k.times = own(h, dupload(times));
                c.k.lrow = own(h, dupload(lrows));
                c.k.tidx = own(h, dupload(tidx));
                c.k.hess_on = own(h, dupload(hess_on));
                c.k.jpos = own(h, dupload(jpos));
                if (c.external) {  // Hessian blocks: knot constraints place knot entries, the global one tail entries
                    c.xk.nc = c.global ? 0 : (int32_t)c.comps.size();
                    c.xk.ng = c.global ? (int32_t)c.comps.size() : 0;
                    c.xk.comps = c.k.comps; c.xk.gcomps = c.k.comps;
                    c.xk.times = c.k.times; c.xk.tidx = c.k.tidx;
                    c.xk.knot_on = c.k.hess_on; c.xk.count = c.k.hess_on;
                    c.xk.glob_on = h->k_hi == h->N ? 1 : 0;
                    c.xk.n_list = c.k.n_times;
                }
            }
        }
        h->row_segments = shard_row_segments(h, h->k_lo, h->k_hi);  // the local buffer is their concatenation, in this order
        h->cons_len = lrow;
        I.cons_len = lrow;
        I.n_row_segments = (int32_t)h->row_segments.size();

        // objectives
        for (int i = 0; i < d->n_objectives && !sonly; ++i) {
            const dto_objective_desc& s = d->objectives[i];
            KObj o{};
            o.kind = s.kind; o.weight = s.weight; o.D = s.D;
            std::vector<int64_t> times;
            std::vector<int64_t> layer_start;
            // stable order of the listings by layer (occurrence index of their knot); returns the permutation
            auto layer_order = [&](const std::vector<int64_t>& t) {
                std::map<int64_t, int> seen;
                std::vector<int> layer(t.size());
                int nl = 0;
                for (size_t q = 0; q < t.size(); ++q) { layer[q] = seen[t[q]]++; nl = std::max(nl, layer[q] + 1); }
                std::vector<size_t> perm(t.size());
                for (size_t q = 0; q < perm.size(); ++q) perm[q] = q;
                std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b2) { return layer[a] < layer[b2]; });
                layer_start.assign((size_t)nl + 1, 0);
                for (size_t q = 0; q < t.size(); ++q) layer_start[(size_t)layer[q] + 1]++;
                for (int l = 0; l < nl; ++l) layer_start[(size_t)l + 1] += layer_start[(size_t)l];
                if (t.empty()) layer_start.assign(1, 0);
                return perm;
            };
            auto permute = [&](auto& v, const std::vector<size_t>& perm, size_t width) {
                if (v.empty()) return;
                auto src = v;
                for (size_t q = 0; q < perm.size(); ++q)
                    for (size_t c = 0; c < width; ++c) v[q * width + c] = src[perm[q] * width + c];
            };
            if (s.kind == DTO_OBJECTIVE_MINIMUM_TIME) {
                for (int64_t kn = P.kn_lo; kn < P.kn_lo + P.n_knots; ++kn)
                    if (kn < h->K) times.push_back(kn);
                (void)layer_order(times);
            } else if (s.kind == DTO_OBJECTIVE_QUADRATIC_REGULARIZER || s.kind == DTO_OBJECTIVE_LINEAR_REGULARIZER) {
                if (s.comp_dim < 1 || s.comp_off < 0 || s.comp_off + s.comp_dim > d->z || !s.R)
                    throw HipError{"objective: bad component range"};
                o.comp_off = s.comp_off; o.comp_dim = s.comp_dim;
                o.R = own(h, dupload(std::vector<double>(s.R, s.R + s.comp_dim)));
                if (s.baseline && s.kind == DTO_OBJECTIVE_QUADRATIC_REGULARIZER) {
                    o.has_baseline = 1;
                    o.baseline = own(h, dupload(std::vector<double>(s.baseline, s.baseline + (size_t)s.comp_dim * d->N)));
                }
                if (s.times) {
                    for (int64_t t = 0; t < s.n_times; ++t) {
                        if (s.times[t] < 1 || s.times[t] > d->N) throw HipError{"objective: time out of range"};
                        const int64_t kn = s.times[t] - 1;
                        if (kn >= P.kn_lo && kn < P.kn_lo + P.n_knots) times.push_back(kn);
                    }
                } else {
                    for (int64_t kn = P.kn_lo; kn < P.kn_lo + P.n_knots; ++kn) times.push_back(kn);
                }
                permute(times, layer_order(times), 1);
            } else if (s.kind == DTO_OBJECTIVE_KNOT_SQDIST || s.kind == DTO_OBJECTIVE_KNOT_LOWRANK_INFIDELITY) {
                if (s.n_comps < 1 || !s.comps || !s.times) throw HipError{"knot objective: comps and times are required"};
                if (s.kind == DTO_OBJECTIVE_KNOT_LOWRANK_INFIDELITY) {
                    if (s.comp_dim < 1 || !s.R) throw HipError{"low-rank infidelity: the factor A (R) and its row count (comp_dim) are required"};
                    o.comp_dim = s.comp_dim;
                    o.R = own(h, dupload(std::vector<double>(s.R, s.R + (size_t)s.comp_dim * s.n_comps)));
                }
                std::vector<int32_t> comps(s.comps, s.comps + s.n_comps);
                for (int q : comps)
                    if (q < 0 || q >= d->z) throw HipError{"knot objective: component out of range"};
                std::vector<double> params, Qs;
                std::vector<int32_t> last;
                for (int64_t t = 0; t < s.n_times; ++t) {
                    if (s.times[t] < 1 || s.times[t] > d->N) throw HipError{"objective: time out of range"};
                    const int64_t kn = s.times[t] - 1;
                    if (kn < P.kn_lo || kn >= P.kn_lo + P.n_knots) continue;
                    times.push_back(kn);
                    Qs.push_back(s.Qs ? s.Qs[t] : 1.0);
                    if (s.params) params.insert(params.end(), s.params + (size_t)t * s.n_comps, s.params + (size_t)(t + 1) * s.n_comps);
                    int32_t is_last = 1;
                    for (int64_t t2 = t + 1; t2 < s.n_times; ++t2)
                        if (s.times[t2] == s.times[t]) { is_last = 0; break; }
                    last.push_back(is_last);
                }
                {
                    const std::vector<size_t> perm = layer_order(times);
                    permute(times, perm, 1);
                    permute(Qs, perm, 1);
                    permute(last, perm, 1);
                    permute(params, perm, (size_t)s.n_comps);
                }
                o.n_comps = s.n_comps;
                o.comps = own(h, dupload(comps));
                o.Qs = own(h, dupload(Qs));
                o.last = own(h, dupload(last));
                o.params = s.params ? own(h, dupload(params)) : nullptr;
            } else if (s.kind == DTO_OBJECTIVE_EXTERNAL_KNOT || s.kind == DTO_OBJECTIVE_EXTERNAL_GLOBAL) {
                continue;  // placed by the external-term kernels (set up below)
            } else {
                throw HipError{"unknown objective kind"};
            }
            o.n_times = (int64_t)times.size();
            o.times = own(h, dupload(times));
            h->obj.push_back(o);
            dto_handle::ObjInfo oi;
            oi.layer_start = layer_start;
            oi.kind = s.kind; oi.comp_off = s.comp_off; oi.comp_dim = s.comp_dim; oi.times = times;
            if (s.comps && s.n_comps > 0) oi.comps.assign(s.comps, s.comps + s.n_comps);
            h->obj_info.push_back(std::move(oi));
        }

        // host-evaluated objective terms: which listings this handle places (knot part: the knot's owner; entries in
        // global-variable columns and listings without a knot part: the rank that owns the last knot)
        const bool last_rank = h->k_hi == h->N;
        for (auto& e : h->ext_obj) {
            std::vector<int64_t> times, tix;
            std::vector<int32_t> knot_on, count;
            for (size_t i = 0; i < e.times0.size(); ++i) {
                const int64_t kn = e.times0[i];
                const bool pseudo = kn >= h->N;
                const bool own = pseudo ? last_rank : (kn >= P.kn_lo && kn < P.kn_lo + P.n_knots);
                if (!own && !(e.global && last_rank)) continue;
                int32_t on = own && !pseudo;
                if (on && !e.global)  // KnotPointObjective's gradient!/hessian! overwrite per listing: the last one wins
                    for (size_t i2 = i + 1; i2 < e.times0.size(); ++i2)
                        if (e.times0[i2] == kn) { on = 0; break; }
                times.push_back(kn);
                tix.push_back((int64_t)i);
                knot_on.push_back(on);
                count.push_back(own ? 1 : 0);
            }
            e.k.nc = (int32_t)e.comps.size(); e.k.ng = (int32_t)e.gcomps.size();
            e.k.n_list = (int64_t)times.size();
            e.k.glob_on = last_rank ? 1 : 0;
            if (!sonly) {
                e.k.comps = own(h, dupload(e.comps));
                e.k.gcomps = own(h, dupload(e.gcomps));
                e.k.times = own(h, dupload(times));
                e.k.tidx = own(h, dupload(tix));
                e.k.knot_on = own(h, dupload(knot_on));
                e.k.count = own(h, dupload(count));
            }
        }

        if (sonly) {
            *out = h;
            return 0;
        }
        // scratch
        h->d_Z = own(h, dalloc<double>(h->n_vars));
        h->d_mu = own(h, dalloc<double>(std::max<int64_t>(h->n_cons, 1)));
        h->d_partial = own(h, dalloc<double>(256));
        h->d_f = own(h, dalloc<double>(1));
        h->d_bounds = own(h, dalloc<double>(2));
        h->d_plan = own(h, dalloc<int32_t>(4));
        HIP_CHECK(hipHostMalloc((void**)&h->h_pinned, 32 * sizeof(double)));
        HIP_CHECK(hipHostMalloc((void**)&h->h_stats, sizeof(int32_t) * 4 * std::max<size_t>(h->bil.size(), 1)));
        memset(h->h_stats, 0, sizeof(int32_t) * 4 * std::max<size_t>(h->bil.size(), 1));

        // device buffers of the time-dependent bilinear integrators: blocks indexed by the global interval (like the
        // host-evaluated integrators' arrays), one scratch slab per interval this handle evaluates
        for (auto& t : h->tdb) {
            const size_t n = t.k.n, z = h->z, K = (size_t)h->K;
            t.d_vals = own(h, dalloc<double>(K * n));
            t.d_jac = own(h, dalloc<double>(K * n * 2 * z));
            if (d->eval_hessian) t.d_hess = own(h, dalloc<double>(K * 4 * z * z));
            t.stride = (std::max(tdb_scratch_doubles(t.k, 1), d->eval_hessian ? tdb_scratch_doubles(t.k, 2) : (size_t)0) + 1) & ~(size_t)1;
            t.d_scratch = own(h, dalloc<double>(t.stride * (size_t)(P.n_knots + 1)));
        }
        // per-bilinear workspaces + generator product norms (for the step-budget bounds)
        for (auto& b : h->bil) {
            const int m = b.k.m;
            if (b.small) {
                if (d->eval_hessian && 1 + m + m * (m + 1) / 2 > MAX_TYPES) throw HipError{"bilinear integrator: too many drives for second-order sweep"};
                continue;  // the fused kernel needs no workspace
            }
            const int T_fw = std::max(2 + m, d->eval_hessian ? 1 + m + m * (m + 1) / 2 : 1 + m);  // +1: exp(A)w_x column of J w
            if (T_fw > MAX_TYPES) throw HipError{"bilinear integrator: too many drives for the second-order sweep"};
            alloc_sweep(h, b, b.fw, T_fw, d->eval_hessian != 0);  // W: G_l x for the Hessian's scalar blocks
            if (d->eval_hessian) {
                alloc_sweep(h, b, b.ad, 1 + m, true);
                // pairing path: term stores for both sweeps + E_j*terms + Beta-weighted sums (skipped when they
                // would take more than 40 % of the free HBM: the second-order sweep is then used)
                // (80 terms: the step budget from the cheap norm bound of the 256 x 2000 benchmark is 66 -- with 64 the Hessian
                // bought the exact norms, a store-less basis GEMM and two round trips, only to fit its budget into the store)
                const int dcap = PAIR_DCAP, T1 = 1 + m;
                const double bytes = (double)dcap * T1 * b.fw.Kpad * b.k.npad * 8.0;
                static const bool pair_on = tune_int("DTO_HESS_PAIRING", 1) != 0;
                size_t free_b = 0, total_b = 0;
                HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
                if (pair_on && m >= 1 && bytes * (3.0 + 1.0 * m / T1) < 0.4 * (double)free_b) {
                    const size_t store = (size_t)dcap * T1 * b.fw.Kpad * b.k.npad;
                    for (SweepBuf* w : {&b.fw, &b.ad}) {
                        w->Zt = own(h, dalloc<double>(store));
                        w->dcap = dcap;
                        // one entry per convergence block; the fused sweep's blocks are as small as one interval
                        w->nterms = own(h, dalloc<int32_t>(w->Kpad));
                        HIP_CHECK(hipMemset(w->nterms, 0, sizeof(int32_t) * w->Kpad));
                        w->nterms_p = own(h, dalloc<int32_t>(w->Kpad));
                        HIP_CHECK(hipMemset(w->nterms_p, 0, sizeof(int32_t) * w->Kpad));
                    }
                    b.EP = own(h, dalloc<double>((size_t)m * dcap * b.fw.Kpad * b.k.npad));  // G_j' U_a
                    b.Upair = own(h, dalloc<double>(store));
                    std::vector<double> bt(PAIR_DCAP * PAIR_DCAP);
                    for (int a = 0; a < PAIR_DCAP; ++a)
                        for (int c = 0; c < PAIR_DCAP; ++c)
                            bt[a * PAIR_DCAP + c] = std::exp(std::lgamma(a + 1.0) + std::lgamma(c + 1.0) - std::lgamma(a + c + 2.0));
                    b.d_Btab = own(h, dupload(bt));
                    b.pairing = true;
                }
            }
            const int npad = b.k.npad;
            alloc_chain(h, b, std::max(chunk_size(h, npad), (m + 1) * (m + 1)));
            // N2[i][j] = ||G_i G_j||_1 with the engine's own batched GEMM + norm kernels
            const int m1 = m + 1, nb = m1 * m1;
            const size_t nn = (size_t)npad * npad;
            for (int i = 0; i < m1; ++i)
                for (int j = 0; j < m1; ++j) {
                    HIP_CHECK(hipMemcpyAsync(b.chain.W[0] + (size_t)(i * m1 + j) * nn, b.k.G + (size_t)i * nn, nn * 8, hipMemcpyDeviceToDevice, h->stream));
                    HIP_CHECK(hipMemcpyAsync(b.chain.W[2] + (size_t)(i * m1 + j) * nn, b.k.G + (size_t)j * nn, nn * 8, hipMemcpyDeviceToDevice, h->stream));
                }
            launch_bgemm_plain(h->stream, npad, nb, b.chain.W[0], b.chain.W[2], b.chain.W[1]);
            launch_norm1(h->stream, npad, nb, b.chain);
            std::vector<double> norms((size_t)nb * 4);
            HIP_CHECK(hipMemcpyAsync(norms.data(), b.chain.norms, norms.size() * 8, hipMemcpyDeviceToHost, h->stream));
            HIP_CHECK(hipStreamSynchronize(h->stream));
            b.n2.resize(nb);
            for (int i = 0; i < nb; ++i) b.n2[i] = norms[(size_t)i * 4 + 1];
            b.d_g1 = own(h, dupload(b.g1));
            b.d_n2 = own(h, dupload(b.n2));
            // generator-subspace powers pay off while the number of symmetrised products stays well
            // below the 3n columns the three GEMMs would process (DTO_BASIS_POWERS=0/1 overrides)
            {
                long cnt = 0;
                long c2 = (long)m1 * (m1 + 1) / 2, c3 = c2 * (m1 + 2) / 3, c4 = c3 * (m1 + 3) / 4;
                cnt = c2 + c3 + c4;
                // (npad is a multiple of 64: npad^2 is a multiple of the kernel's 128-row tiles and every wave's 64 rows of
                // vec(A^r) stay inside one matrix column, which is all k_basis_gemm's fused column sums need)
                bool want = cnt * 2 <= 3L * npad;
                { const int f = tune_int("DTO_BASIS_POWERS", -1); if (f >= 0) want = f != 0; }
                if (want) build_basis(h, b, b.chain_cap);
            }
        }
        HIP_CHECK(hipDeviceSynchronize());
    } catch (const HipError& e) {
        g_create_error = e.msg;
        delete h;
        return 1;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        delete h;
        return 1;
    }
    *out = h;
    return 0;
}

int dto_num_vars(const dto_handle* h, int64_t* out) { if (!h || !out) return 1; *out = h->n_vars; return 0; }
int dto_num_cons(const dto_handle* h, int64_t* out) { if (!h || !out) return 1; *out = h->n_cons; return 0; }
int dto_num_dynamics_cons(const dto_handle* h, int64_t* out) { if (!h || !out) return 1; *out = h->n_dyn; return 0; }
int dto_jac_nnz(const dto_handle* h, int64_t* out) { if (!h || !out) return 1; *out = h->jac_nnz; return 0; }
int dto_hess_nnz(const dto_handle* h, int64_t* out) { if (!h || !out) return 1; *out = h->hess_nnz; return 0; }
int dto_features_available(const dto_handle* h, int32_t* grad, int32_t* jac, int32_t* hess) {
    if (!h) return 1;
    if (grad) *grad = 1;
    if (jac) *jac = 1;
    if (hess) *hess = h->eval_hessian ? 1 : 0;
    return 0;
}
int dto_get_shard_info(const dto_handle* h, dto_shard_info* out) { if (!h || !out) return 1; *out = h->info; return 0; }
int dto_shard_rows(const dto_handle* h, int64_t* start1, int64_t* len) {
    if (!h || !start1 || !len) return 1;
    for (size_t i = 0; i < h->row_segments.size(); ++i) {
        start1[i] = h->row_segments[i].first + 1;
        len[i] = h->row_segments[i].second;
    }
    return 0;
}

// Cost model of one eval_constraint_jacobian per interval ("balanced by sum s_k if scaling counts vary"), host
// arithmetic on Z with the norms of the generators the handle keeps: the same growth bounds the engine plans with (k_norm_bounds:
// b1 >= ||A_k||_1 from ||G_j||_1, b2 >= ||A_k^2||_1^(1/2) from ||G_i G_j||_1 where the handle has them -- a GPU handle), turned into
// squaring counts of the cheaper evaluation form, the Taylor terms of the sweep and their flop counts.
int dto_interval_costs(const dto_handle* h, const double* Z, int64_t first, int64_t count, double* cost) {
    if (!h || !Z || !cost || first < 0 || count < 0 || first + count > h->K) return 1;
    for (int64_t i = 0; i < count; ++i) cost[i] = 0.0;
    for (const BilHost& b : h->bil) {
        const int m1 = b.k.m + 1;
        const double np = b.k.npad, gemm = 2.0 * np * np * np;
        // multisets of sizes 2..4 (and 0..4 for the factor K) over m + 1 generators: the generator-subspace GEMMs
        const double c2 = m1 * (m1 + 1) / 2.0, c3 = c2 * (m1 + 2) / 3.0, c4 = c3 * (m1 + 3) / 4.0;
        const double basis = b.small ? 0.0 : 2.0 * np * np * (2.0 * (c2 + c3 + c4) + 1 + m1);
        for (int64_t i = 0; i < count; ++i) {
            const double* zk = Z + (first + i) * h->z;
            const double dt = std::fabs(zk[h->P.dt_idx]);
            double ub[MAX_DRIVES + 1];
            ub[0] = 1.0;
            for (int j = 0; j < b.k.m; ++j) ub[j + 1] = std::fabs(zk[b.k.u_off + j]);
            double b1 = 0.0, s2 = 0.0;
            for (int a = 0; a < m1; ++a) {
                b1 += ub[a] * b.g1[a];
                if ((int)b.n2.size() == m1 * m1)
                    for (int c = 0; c < m1; ++c) s2 += ub[a] * ub[c] * b.n2[a * m1 + c];
            }
            b1 *= dt;
            double alpha = b1;
            if ((int)b.n2.size() == m1 * m1) alpha = std::min(b1, dt * std::sqrt(s2));
            if (!(alpha == alpha) || alpha > 1e6) alpha = 1e6;
            const double sq2 = alpha > THETA_16 ? std::ceil(std::log2(alpha / THETA_16)) : 0.0;
            const double sq3 = alpha > THETA_3P ? std::ceil(std::log2(alpha / THETA_3P)) : 0.0;
            const double products = std::min(2.0 + sq2, 3.0 + sq3);
            const SweepPlan sp = plan_sweep(alpha);
            const double sweep = 2.0 * np * np * m1 * m1 * (double)sp.d_ub * sp.q;
            cost[i] += b.small ? 2.0 * b.k.n * (double)b.k.n * b.k.n * (6.0 + sq2) : gemm * products + basis + sweep;
        }
    }
    // every other term kind costs O(z) per knot: a constant that keeps intervals without a bilinear integrator from counting as free
    for (int64_t i = 0; i < count; ++i) cost[i] += 64.0 * h->z;
    return 0;
}

int dto_jacobian_structure(const dto_handle* h, int64_t first, int64_t count, int64_t* rows, int64_t* cols) {
    if (!h || !rows || !cols || first < 0 || count < 0 || first + count > h->jac_nnz) return 1;
    if (count == 0) return 0;
    // column containing `first`
    int64_t c = std::upper_bound(h->colptr.begin(), h->colptr.end(), first) - h->colptr.begin() - 1;
    int64_t written = 0;
    const int64_t nzcols = h->N * h->z;
    for (; c < nzcols && written < count; ++c) {
        const int64_t kn = c / h->z;
        int64_t pos = h->colptr[c];
        auto emit = [&](int64_t row0) {
            if (pos >= first && written < count) {
                rows[written] = row0 + 1;
                cols[written] = c + 1;
                ++written;
            }
            ++pos;
        };
        // integrator rows: for each integrator, interval kn-1 then interval kn
        for (size_t i = 0; i < h->integ_kind.size(); ++i) {
            const int d = h->integ_dim[i];
            const int64_t off = h->integ_row_off[i];
            if (kn >= 1) for (int r = 0; r < d; ++r) emit(off + (kn - 1) * d + r);
            if (kn < h->K) for (int r = 0; r < d; ++r) emit(off + kn * d + r);
        }
        for (size_t e = con_lower(h, c); e < h->con_cols.size() && h->con_cols[e] == c; ++e) emit(h->con_rows[e]);
    }
    for (; c < h->n_vars && written < count; ++c) {  // global-variable columns: NonlinearGlobalConstraint rows only
        int64_t pos = h->colptr[c];
        for (size_t e = con_lower(h, c); e < h->con_cols.size() && h->con_cols[e] == c; ++e, ++pos)
            if (pos >= first && written < count) {
                rows[written] = h->con_rows[e] + 1;
                cols[written] = c + 1;
                ++written;
            }
    }
    return written == count ? 0 : 1;
}

int dto_hessian_structure(const dto_handle* h, int64_t first, int64_t count, int64_t* rows, int64_t* cols) {
    if (!h || !rows || !cols || first < 0 || count < 0 || first + count > h->hess_nnz) return 1;
    if (count == 0) return 0;
    const int64_t z = h->z, tri = z * (z + 1) / 2, blk = z * z + tri;
    int64_t kn = first < tri ? 0 : 1 + (first - tri) / blk;
    int64_t written = 0;
    for (; kn < h->N && written < count; ++kn) {
        int64_t pos = hess_block_start(h, kn);
        for (int b = 0; b < z && written < count; ++b) {
            const int64_t c = kn * z + b;
            auto emit = [&](int64_t row0) {
                if (pos >= first && written < count) {
                    rows[written] = row0 + 1;
                    cols[written] = c + 1;
                    ++written;
                }
                ++pos;
            };
            if (kn >= 1) for (int a = 0; a < z; ++a) emit((kn - 1) * z + a);
            for (int a = 0; a <= b; ++a) emit(kn * z + a);
        }
    }
    for (int j = 0; j < h->gd && written < count; ++j)  // tail: global-variable columns
        for (int64_t e = h->tail_colptr[j]; e < h->tail_colptr[j + 1] && written < count; ++e)
            if (h->hess_block_nnz + e >= first) {
                rows[written] = h->tail_rows[e] + 1;
                cols[written] = h->N * z + j + 1;
                ++written;
            }
    return written == count ? 0 : 1;
}

int dto_constraint_bounds(const dto_handle* h, double* lower, double* upper) {
    if (!h || !lower || !upper) return 1;
    for (int64_t i = 0; i < h->n_cons; ++i) { lower[i] = 0.0; upper[i] = 0.0; }
    for (auto& c : h->con)
        if (!c.equality)
            for (int64_t i = 0; i < c.n_times_total * c.g_dim; ++i) lower[c.row_off + i] = -std::numeric_limits<double>::infinity();
    return 0;
}

int dto_num_external(const dto_handle* h, int32_t* n_integrators, int32_t* n_constraints, int32_t* n_objectives) {
    if (!h) return 1;
    if (n_integrators) *n_integrators = h->n_ext_int;
    if (n_constraints) *n_constraints = h->n_ext_con;
    if (n_objectives) *n_objectives = h->n_ext_obj;
    return 0;
}
int dto_set_external(dto_handle* h, int32_t n, const dto_external_values* v) {
    if (!h) return 1;
    if (n != h->n_ext_int + h->n_ext_con + h->n_ext_obj || (n > 0 && !v))
        return fail(h, "dto_set_external: one entry per external term is required (integrators, constraints, objectives)");
    for (int i = 0; i < n && i < (int)h->ext.size(); ++i) h->ext[i].v = v[i];
    return 0;
}

// ---- device-pointer callbacks
int dto_eval_objective_dev(dto_handle* h, const double* dZ, double* df, void* stream) {
    return guarded(h, [&] { do_objective(h, dZ, df, (hipStream_t)stream); });
}
int dto_eval_gradient_dev(dto_handle* h, const double* dZ, double* dgrad, void* stream) {
    return guarded(h, [&] { do_gradient(h, dZ, dgrad, (hipStream_t)stream); });
}
int dto_eval_constraint_dev(dto_handle* h, const double* dZ, double* dg, void* stream) {
    return guarded(h, [&] { do_constraint(h, dZ, dg, (hipStream_t)stream); }, G_ASYNC, (hipStream_t)stream);
}
int dto_eval_jacobian_dev(dto_handle* h, const double* dZ, double* dvals, void* stream) {
    return guarded(h, [&] {
        do_jacobian(h, dZ, dvals, (hipStream_t)stream);
        if (h->bound[0] == dvals) h->primed[0] = true;
    }, G_ASYNC, (hipStream_t)stream);
}
int dto_eval_hessian_dev(dto_handle* h, const double* dZ, double sigma, const double* dmu, double* dvals, void* stream) {
    return guarded(h, [&] {
        if (!h->eval_hessian) throw HipError{"handle was created with eval_hessian = 0"};
        do_hessian(h, dZ, sigma, dmu, dvals, (hipStream_t)stream);
        if (h->bound[1] == dvals) h->primed[1] = true;
    }, G_ASYNC, (hipStream_t)stream);
}

// ---- host-pointer callbacks (blocking)
int dto_eval_objective(dto_handle* h, const double* Z, double* f) {
    return guarded(h, [&] {
        upload_Z(h, Z);
        do_objective(h, h->d_Z, h->d_f, h->stream);
        HIP_CHECK(hipMemcpyAsync(f, h->d_f, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}
int dto_eval_gradient(dto_handle* h, const double* Z, double* grad) {
    return guarded(h, [&] {
        upload_Z(h, Z);
        double* o = staging(h, (size_t)h->info.grad_len);
        do_gradient(h, h->d_Z, o, h->stream);
        HIP_CHECK(hipMemcpyAsync(grad, o, sizeof(double) * (size_t)h->info.grad_len, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}
int dto_eval_constraint(dto_handle* h, const double* Z, double* g) {
    return guarded(h, [&] {
        upload_Z(h, Z);
        double* o = staging(h, (size_t)h->cons_len);
        do_constraint(h, h->d_Z, o, h->stream);
        HIP_CHECK(hipMemcpyAsync(g, o, sizeof(double) * (size_t)h->cons_len, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
    }, G_BLOCKING);
}
int dto_eval_jacobian(dto_handle* h, const double* Z, double* vals) {
    return guarded(h, [&] {
        upload_Z(h, Z);
        double* o = staging(h, (size_t)h->info.jac_len);
        ensure_plans(h);
        if (h->jac_plan.usable()) {
            // constants are filled by host threads while the GPU computes; only the variable runs cross PCIe, and the -E_k
            // blocks start crossing as soon as their chain chunk is done (dto_hostxfer.h)
            const XferPlan& pl = h->jac_plan;
            struct Reset {
                dto_handle* h;
                ~Reset() { h->xfer_cap = 0; h->on_chain_chunk = nullptr; }
            } reset{h};
            h->xfer->begin(pl, o, vals);
            try {
                if (pl.n_early() > 0) {
                    static const int chunks = std::max(1, tune_int("DTO_XFER_CHUNKS", 4));
                    const int64_t nint = h->P.n_int;
                    if (nint >= 512 && chunks > 1) h->xfer_cap = (int)(((nint + chunks - 1) / chunks + 7) / 8 * 8);
                    h->on_chain_chunk = [&](int64_t c0, int nb) {
                        h->xfer->submit(pl.early_off[(size_t)c0], pl.early_off[(size_t)(c0 + nb)], h->stream);
                    };
                }
                do_jacobian(h, h->d_Z, o, h->stream);
                h->xfer->submit(pl.n_early(), pl.n_runs(), h->stream);
                h->xfer->finish();
            } catch (...) {
                h->xfer->abort();  // joins the host threads before the error leaves
                throw;
            }
            if (h->xfer_check) check_against_slab(h, o, vals, (size_t)h->info.jac_len, "Jacobian");
            return;
        }
        do_jacobian(h, h->d_Z, o, h->stream);
        HIP_CHECK(hipMemcpyAsync(vals, o, sizeof(double) * (size_t)h->info.jac_len, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
    }, G_BLOCKING);
}
int dto_eval_hessian(dto_handle* h, const double* Z, double sigma, const double* mu, double* vals) {
    return guarded(h, [&] {
        if (!h->eval_hessian) throw HipError{"handle was created with eval_hessian = 0"};
        upload_Z(h, Z);
        HIP_CHECK(hipMemcpyAsync(h->d_mu, mu, sizeof(double) * (size_t)h->n_cons, hipMemcpyHostToDevice, h->stream));
        double* o = staging(h, (size_t)h->info.hess_len);
        ensure_plans(h);
        if (h->hess_plan.usable()) {
            const XferPlan& pl = h->hess_plan;
            h->xfer->begin(pl, o, vals);
            try {
                do_hessian(h, h->d_Z, sigma, h->d_mu, o, h->stream);
                h->xfer->submit(0, pl.n_runs(), h->stream);
                h->xfer->finish();
            } catch (...) {
                h->xfer->abort();
                throw;
            }
            if (h->xfer_check) check_against_slab(h, o, vals, (size_t)h->info.hess_len, "Hessian");
            return;
        }
        do_hessian(h, h->d_Z, sigma, h->d_mu, o, h->stream);
        HIP_CHECK(hipMemcpyAsync(vals, o, sizeof(double) * (size_t)h->info.hess_len, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
    }, G_BLOCKING);
}

// Matrix-free products for handles whose bilinear integrators all take the general path: exp(A)w_x rides the
// forward sweep as an extra column type (J w), exp(A')w_k is an adjoint sweep (J' w); no value slab is formed.
static void jac_product_matrix_free(dto_handle* h, const double* dZ, const double* dw, double* dy, int transpose, hipStream_t st) {
    const int64_t n_out = transpose ? h->n_vars : h->n_cons;
    HIP_CHECK(hipMemsetAsync(dy, 0, sizeof(double) * (size_t)n_out, st));  // fill!(y, 0), evaluator.jl:416,442
    for (auto& b : h->bil) {
        if (h->P.n_int <= 0) continue;
        b.cache_kind = 0;  // the product sweeps use b.fw with their own column types
        b.p_terms = false;
        SweepPlan plan = plan_from(h, b, dZ, st);
        SweepTypes ty = make_types(b.k.m, false);
        if (!transpose) {
            const int tw = ty.T;
            ty.t[ty.T++] = TypeDesc{0, {0, 0}, {0, 0}, {0, 0}};  // exp(A) w_x: a second "p" column
            SweepBuf ws = b.fw;
            launch_sweep_init(st, h->P, b.k, ws, ty, dZ, nullptr, 0, plan.q);
            launch_sweep_set_type(st, h->P, b.k, ws, ty.T, tw, dw);
            run_sweep(h, b, b.fw, ty, dZ, nullptr, 0, 0, plan, st, false, /*skip_init=*/true);
            launch_apply_Gu(st, b.k, b.fw, 0, b.fw.S, b.fw.GY);
            launch_jv_bilinear(st, h->P, b.k, b.fw, tw, dw, dy);
        } else {
            run_sweep(h, b, b.fw, ty, dZ, nullptr, 0, 0, plan, st);
            launch_apply_Gu(st, b.k, b.fw, 0, b.fw.S, b.fw.GY);
            run_sweep(h, b, b.ad, make_types(0, false), dZ, dw, 1, 1, plan, st);
            launch_jtv_bilinear(st, h->P, b.k, b.fw, b.ad, dw, dy);
        }
    }
    for (auto& d : h->der) launch_jv_derivative(st, h->P, d, dZ, dw, dy, transpose);
    for (auto& c : h->con) launch_jv_knot(st, h->P, c.k, dZ, dw, dy, transpose);
}

static void jac_product(dto_handle* h, const double* Z, const double* w, double* y, int transpose) {
    if (h->k_lo != 1 || h->k_hi != h->N) throw HipError{"Jacobian-vector products need an unsharded handle"};
    bool mfree = h->eval_hessian != 0 || transpose == 0;  // the adjoint sweep buffers exist only with eval_hessian
    for (auto& b : h->bil) mfree = mfree && !b.small && (transpose == 0 || b.ad.S != nullptr) && b.k.m + 2 <= MAX_TYPES;
    for (auto& c : h->con) mfree = mfree && !c.external;  // external blocks are placed into the value slab
    mfree = mfree && h->ext_int.empty() && h->tdb.empty();
    static const bool mfree_on = tune_int("DTO_JV_MATRIX_FREE", 1) != 0;
    if (mfree && mfree_on) {
        const int64_t n_in = transpose ? h->n_cons : h->n_vars, n_out = transpose ? h->n_vars : h->n_cons;
        if (!h->d_w) h->d_w = own(h, dalloc<double>((size_t)std::max(h->n_vars, h->n_cons)));
        upload_Z(h, Z);
        HIP_CHECK(hipMemcpyAsync(h->d_w, w, sizeof(double) * (size_t)n_in, hipMemcpyHostToDevice, h->stream));
        double* o = staging(h, (size_t)n_out);
        jac_product_matrix_free(h, h->d_Z, h->d_w, o, transpose, h->stream);
        HIP_CHECK(hipMemcpyAsync(y, o, sizeof(double) * (size_t)n_out, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        return;
    }
    if (h->integ_kind.size() > 8) throw HipError{"Jacobian-vector products support at most 8 integrators"};
    const int64_t n_in = transpose ? h->n_cons : h->n_vars, n_out = transpose ? h->n_vars : h->n_cons;
    if (!h->d_jac_scratch) {
        h->d_jac_scratch = own(h, dalloc<double>((size_t)h->info.jac_len));
        h->d_w = own(h, dalloc<double>((size_t)std::max(h->n_vars, h->n_cons)));
        std::vector<int64_t> base((size_t)h->n_vars + 1, 0);
        for (size_t e = 0; e < h->con_cols.size(); ++e) base[(size_t)h->con_cols[e] + 1]++;
        for (int64_t c = 0; c < h->n_vars; ++c) base[(size_t)c + 1] += base[(size_t)c];
        h->d_conbase = own(h, dupload(base));
        h->d_con_rows = own(h, dupload(h->con_rows));
        // the constraint entries once more in ROW order (J w gathers every row in ascending column order: no atomics): per
        // constraint row the columns of its entries and their positions in the value slab
        const int64_t n_con_rows = h->n_cons - h->n_dyn;
        std::vector<int64_t> rptr((size_t)n_con_rows + 1, 0), rcol(h->con_rows.size()), rpos(h->con_rows.size());
        for (size_t e = 0; e < h->con_rows.size(); ++e) rptr[(size_t)(h->con_rows[e] - h->n_dyn) + 1]++;
        for (int64_t r = 0; r < n_con_rows; ++r) rptr[(size_t)r + 1] += rptr[(size_t)r];
        std::vector<int64_t> fill(rptr.begin(), rptr.end() - 1);
        for (size_t e = 0; e < h->con_rows.size(); ++e) {   // CSC order: ascending column, so every row's list ends up ascending too
            const int64_t col = h->con_cols[e], r = h->con_rows[e] - h->n_dyn;
            const int64_t kn = col / h->z;
            const int64_t at = fill[(size_t)r]++;
            rcol[(size_t)at] = col;
            rpos[(size_t)at] = h->colptr[col] + (int64_t)h->D * col_cnt(h, kn) + ((int64_t)e - base[(size_t)col]) - h->P.jac_lo;
        }
        h->d_crow_ptr = own(h, dupload(rptr));
        h->d_crow_col = own(h, dupload(rcol));
        h->d_crow_pos = own(h, dupload(rpos));
    }
    KIntegTable T{};
    T.n = (int)h->integ_kind.size();
    for (int i = 0; i < T.n; ++i) { T.d[i] = h->integ_dim[i]; T.off[i] = h->integ_row_off[i]; }
    upload_Z(h, Z);
    HIP_CHECK(hipMemcpyAsync(h->d_w, w, sizeof(double) * (size_t)n_in, hipMemcpyHostToDevice, h->stream));
    do_jacobian(h, h->d_Z, h->d_jac_scratch, h->stream);
    double* o = staging(h, (size_t)n_out);
    HIP_CHECK(hipMemsetAsync(o, 0, sizeof(double) * (size_t)n_out, h->stream));  // fill!(y, 0), evaluator.jl:416,442
    if (T.n > 0 || !h->con.empty()) {
        if (transpose) launch_jac_spmv(h->stream, h->P, T, h->d_conbase, h->d_con_rows, h->d_jac_scratch, h->d_w, o, 1, h->gd);
        else launch_jac_rowgather(h->stream, h->P, T, h->n_cons - h->n_dyn, h->d_crow_ptr, h->d_crow_col, h->d_crow_pos, h->n_dyn, h->d_jac_scratch, h->d_w, o);
    }
    HIP_CHECK(hipMemcpyAsync(y, o, sizeof(double) * (size_t)n_out, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
}
// y = J(Z) w  -- MOI.eval_constraint_jacobian_product (evaluator.jl:406-430)
int dto_eval_jacobian_product(dto_handle* h, const double* Z, const double* w, double* y) {
    return guarded(h, [&] { jac_product(h, Z, w, y, 0); }, G_BLOCKING);
}
// y = J(Z)' w -- MOI.eval_constraint_jacobian_transpose_product (evaluator.jl:432-456)
int dto_eval_jacobian_transpose_product(dto_handle* h, const double* Z, const double* w, double* y) {
    return guarded(h, [&] { jac_product(h, Z, w, y, 1); }, G_BLOCKING);
}

// ---- multi-GPU: ranges, gather plans, collectives (dto_comm.h)
int dto_comm_unique_id(void* id128) {
    if (!id128) return fail(nullptr, "dto_comm_unique_id: null argument");
    const std::string e = Comm::unique_id(id128);
    return e.empty() ? 0 : fail(nullptr, e);
}

int dto_comm_create(dto_handle* h, const void* id128, int32_t rank, int32_t world) {
    return comm_guarded(h, [&] {
        if (!id128) throw HipError{"dto_comm_create: null id"};
        if (h->comm) throw HipError{"dto_comm_create: the handle already has a communicator (dto_comm_destroy first)"};
        std::string err;
        std::unique_ptr<Comm> c = Comm::create(id128, rank, world, err);
        if (!c) throw HipError{err};
        // every rank's knot range: one 16-byte all-gather
        if (!h->d_ranges) h->d_ranges = own(h, dalloc<int64_t>(2 * 1024));
        if (world > 1023) throw HipError{"dto_comm_create: at most 1023 ranks"};
        const int64_t mine[2] = {h->k_lo, h->k_hi};
        HIP_CHECK(hipMemcpyAsync(h->d_ranges, mine, sizeof(mine), hipMemcpyHostToDevice, h->stream));
        err = c->all_gather_i64(h->d_ranges, h->d_ranges + 2, 2, h->stream);
        if (!err.empty()) throw HipError{err};
        std::vector<int64_t> all((size_t)2 * world);
        HIP_CHECK(hipMemcpyAsync(all.data(), h->d_ranges + 2, sizeof(int64_t) * all.size(), hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        std::vector<int64_t> lo(world), hi(world);
        for (int r = 0; r < world; ++r) { lo[r] = all[2 * r]; hi[r] = all[2 * r + 1]; }
        set_ranges(h, rank, world, lo.data(), hi.data());
        h->comm = std::move(c);
    });
}

int dto_comm_set_ranges(dto_handle* h, int32_t rank, int32_t world, const int64_t* k_lo, const int64_t* k_hi) {
    return comm_guarded(h, [&] {
        if (!k_lo || !k_hi) throw HipError{"dto_comm_set_ranges: null argument"};
        if (h->comm) throw HipError{"dto_comm_set_ranges: the handle's communicator already fixed the ranges"};
        set_ranges(h, rank, world, k_lo, k_hi);
    }, /*needs_device=*/false);
}

int dto_comm_destroy(dto_handle* h) {
    return comm_guarded(h, [&] {
        h->comm.reset();
        h->rank_ranges.clear();
        h->comm_rank = -1;
    }, /*needs_device=*/!(h && h->structure_only));
}

int dto_get_gather_layout(const dto_handle* h, int32_t vector, dto_gather_layout* out) {
    dto_handle* hm = const_cast<dto_handle*>(h);
    return comm_guarded(hm, [&] {
        if (!out) throw HipError{"dto_get_gather_layout: null argument"};
        dto_gather_layout L{};
        L.world = (int32_t)h->rank_ranges.size();
        if (vector == DTO_VECTOR_CONSTRAINT) {
            if (h->rank_ranges.empty()) throw HipError{"no knot ranges yet: dto_comm_create / dto_comm_set_ranges first"};
            L.total = L.padded_len = h->n_cons;
            L.own_len = h->cons_len;
        } else {
            const GatherPlan& pl = plan_of(h, vector);
            L.total = pl.total;
            L.padded_len = pl.padded_len();
            L.front_pad = pl.in_place ? pl.front : 0;
            L.own_lo = pl.slabs[h->comm_rank].lo;
            L.own_len = pl.slabs[h->comm_rank].len;
            L.in_place_all_gather = pl.in_place ? 1 : 0;
        }
        *out = L;
    }, /*needs_device=*/false);
}

int dto_gather_slabs(const dto_handle* h, int32_t vector, int64_t* lo, int64_t* len) {
    dto_handle* hm = const_cast<dto_handle*>(h);
    return comm_guarded(hm, [&] {
        if (!lo || !len) throw HipError{"dto_gather_slabs: null argument"};
        const GatherPlan& pl = plan_of(h, vector);
        for (size_t r = 0; r < pl.slabs.size(); ++r) { lo[r] = pl.slabs[r].lo; len[r] = pl.slabs[r].len; }
    }, /*needs_device=*/false);
}

int dto_gather_jacobian_dev(dto_handle* h, double* dbuf, void* stream) {
    return comm_guarded(h, [&] { gather_vector(h, DTO_VECTOR_JACOBIAN, dbuf, (hipStream_t)stream); });
}
int dto_gather_hessian_dev(dto_handle* h, double* dbuf, void* stream) {
    return comm_guarded(h, [&] { gather_vector(h, DTO_VECTOR_HESSIAN, dbuf, (hipStream_t)stream); });
}
int dto_gather_gradient_dev(dto_handle* h, double* dbuf, void* stream) {
    return comm_guarded(h, [&] { gather_vector(h, DTO_VECTOR_GRADIENT, dbuf, (hipStream_t)stream); });
}
int dto_gather_constraint_dev(dto_handle* h, const double* dg_local, double* dg_full, void* stream) {
    return comm_guarded(h, [&] {
        if (!h->comm) throw HipError{"dto_gather: no communicator (dto_comm_create)"};
        hipStream_t st = (hipStream_t)stream;
        int64_t at = 0;
        for (auto& sg : h->row_segments) {  // the local buffer is the concatenation of the rank's segments
            HIP_CHECK(hipMemcpyAsync(dg_full + sg.first, dg_local + at, sizeof(double) * (size_t)sg.second, hipMemcpyDeviceToDevice, st));
            at += sg.second;
        }
        const std::string e = h->comm->broadcast_slabs(dg_full, h->cons_segments, h->cons_segment_root, st);
        if (!e.empty()) throw HipError{e};
    });
}
int dto_allreduce_objective_dev(dto_handle* h, double* df, void* stream) {
    return comm_guarded(h, [&] {
        if (!h->comm) throw HipError{"dto_allreduce_objective_dev: no communicator (dto_comm_create)"};
        const std::string e = h->comm->all_reduce_sum(df, 1, (hipStream_t)stream);
        if (!e.empty()) throw HipError{e};
    });
}

int dto_bind_output_dev(dto_handle* h, int32_t vector, double* dptr) {
    if (!h) return fail(nullptr, "null handle");
    if (vector != DTO_VECTOR_JACOBIAN && vector != DTO_VECTOR_HESSIAN) return fail(h, "dto_bind_output_dev: the Jacobian or the Hessian value vector");
    const int w = vector == DTO_VECTOR_JACOBIAN ? 0 : 1;
    h->bound[w] = dptr;
    h->primed[w] = false;
    return 0;
}

// ---- measurement
int dto_set_option(dto_handle* h, const char* name, int64_t value) {
    if (!h || !name) return 1;
    if (std::string(name) == "reuse_forward_sweep") {
        h->reuse = value != 0;
        drop_caches(h);
        return 0;
    }
    if (std::string(name) == "overlap_sweep") {
        h->overlap_sweep = value != 0;
        return 0;
    }
    if (std::string(name) == "chain_form") {
        if (value != 0 && value != 1) return fail(h, "dto_set_option: chain_form takes 0 (one launch per call for 33..64 states) or 1 (batched-GEMM launches)");
        h->chain_form = (int)value;
        return 0;
    }
    if (std::string(name) == "sweep_form") {
        if (value != 0 && value != 1) return fail(h, "dto_set_option: sweep_form takes 0 (fused where it applies) or 1 (step per launch)");
        h->sweep_form = (int)value;
        drop_caches(h);
        return 0;
    }
    if (std::string(name) == "host_xfer") {
        if (h->plans_built) return fail(h, "dto_set_option: host_xfer must be set before the first host-pointer Jacobian / Hessian call");
        h->host_xfer = value != 0;
        return 0;
    }
    if (std::string(name) == "chain_chunk") {
        if (value < 0) return fail(h, "dto_set_option: chain_chunk must be >= 0");
        h->chain_chunk = (int)std::min<int64_t>(value, 1 << 30);
        return 0;
    }
    if (std::string(name) == "debug_bad_launch") {
#ifdef DTO_TUNING
        h->P.debug_bad_launch = value != 0;
        return 0;
#else
        return fail(h, "dto_set_option: debug_bad_launch exists in TUNING builds only (make TUNING=1: libdto_engine_t.so)");
#endif
    }
    if (std::string(name) == "host_xfer_check") {
        h->xfer_check = value != 0;
        return 0;
    }
    if (std::string(name) == "deterministic") {
        h->deterministic = value != 0;
        drop_caches(h);
        return 0;
    }
    if (std::string(name) == "expm_form") {
        if (value != 0 && value != 2 && value != 3) return fail(h, "dto_set_option: expm_form takes 0 (by cost), 2 or 3");
        h->expm_form = (int)value;
        return 0;
    }
    return fail(h, std::string("dto_set_option: unknown option ") + name);
}

int dto_profile_enable(dto_handle* h, int32_t on) {
    if (!h) return 1;
    if (on && !h->structure_only && h->ev_pool.size() < 1024) {
        (void)hipSetDevice(h->device);
        while (h->ev_pool.size() < 1024) {  // enough for ~8 calls without touching the allocator in the timed region
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) break;
            h->ev_pool.push_back(e);
        }
    }
    h->profiling = on != 0;
    return 0;
}
int dto_profile_reset(dto_handle* h) {
    if (!h) return 1;
    for (auto& r : h->prof) { h->ev_pool.push_back(r.a); h->ev_pool.push_back(r.b); }
    h->prof.clear();
    return 0;
}
int dto_profile_get(dto_handle* h, const char* name, double* ms, int64_t* launches, double* flops) {
    return guarded(h, [&] {
        int cat = -1;
        bool any_gemm = false, any_basis = false;
        if (!strcmp(name, "bgemm")) any_gemm = true;
        else if (!strcmp(name, "bgemm_horner")) cat = CAT_BGEMM_HORNER;
        else if (!strcmp(name, "bgemm_square")) cat = CAT_BGEMM_SQUARE;
        else if (!strcmp(name, "bgemm_plain")) cat = CAT_BGEMM;
        else if (!strcmp(name, "chain64")) cat = CAT_CHAIN64;
        else if (!strcmp(name, "basis")) any_basis = true;
        else if (!strcmp(name, "basis_k")) cat = CAT_OTHER;
        else if (!strcmp(name, "basis_multi")) cat = CAT_BASIS_MULTI;
        else if (!strcmp(name, "zero_fill")) cat = CAT_ZERO;
        else if (!strcmp(name, "build_A")) cat = CAT_BUILD_A;
        else if (!strcmp(name, "assembly")) cat = CAT_ASSEMBLY;
        else if (!strcmp(name, "expmv")) cat = CAT_SWEEP;
        else if (!strcmp(name, "expmv_adjoint")) cat = CAT_SWEEP_ADJOINT;
        else if (strcmp(name, "all")) throw HipError{"dto_profile_get: unknown name"};
        double tot = 0, fl = 0;
        int64_t n = 0;
        for (auto& r : h->prof) {
            if (any_gemm && r.cat != CAT_BGEMM && r.cat != CAT_BGEMM_HORNER && r.cat != CAT_BGEMM_SQUARE && r.cat != CAT_CHAIN64) continue;
            if (any_basis && r.cat != CAT_OTHER && r.cat != CAT_BASIS_MULTI) continue;
            if (!any_gemm && !any_basis && cat >= 0 && r.cat != cat) continue;
            HIP_CHECK(hipEventSynchronize(r.b));
            float t = 0;
            HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
            tot += t; fl += r.flops; ++n;
        }
        if (ms) *ms = tot;
        if (launches) *launches = n;
        if (flops) *flops = fl;
    });
}
int dto_last_stats(const dto_handle* h, int32_t* max_squarings, int32_t* expmv_terms) {
    if (!h) return 1;
    dto_handle* hm = const_cast<dto_handle*>(h);
    // the deferred error of an asynchronous call surfaces here (guarded looks at the pending statistics first); the
    // statistics themselves are then read again so that `expmv_terms` is current after blocking calls as well
    int rc = guarded(hm, [&] { HIP_CHECK(hipDeviceSynchronize()); }, G_BLOCKING);
    if (max_squarings) *max_squarings = h->last_smax;
    if (expmv_terms) *expmv_terms = h->last_terms;
    return rc;
}

}  // extern "C"
