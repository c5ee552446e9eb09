/*
 * dto_engine.h -- C ABI of the MI355X-native NLP-callback engine for DirectTrajOpt.jl.
 *
 * Drop-in boundary: the seven MOI.AbstractNLPEvaluator methods implemented by the reference's
 * `Evaluator` (src/solvers/evaluator.jl:66-98, 291-456).  A Julia shim `GPUEvaluator <:
 * MOI.AbstractNLPEvaluator` forwards each MOI method to one entry point below through `@ccall`
 * (see INTEGRATION.md).  Plain C: pointers + sizes only, no exceptions cross the boundary.
 *
 * Conventions
 *   - All floating-point data are IEEE double; all index outputs are int64, 1-BASED (Julia/MOI).
 *   - Matrices are dense COLUMN-MAJOR (Julia's native layout).
 *   - The NLP variable vector is Z = [datavec; global_data], datavec knot-major
 *     (src/solvers/evaluator.jl:230, 474-482); component offsets are 0-based inside a knot.
 *   - Return value 0 = OK, non-zero = error; text via dto_last_error().  A kernel launch the HIP runtime rejects
 *     (hipGetLastError after the call's launches) is such an error.  Errors that only show once the device has
 *     run -- a generator sweep that exhausted its step budget -- are reported by the blocking entry points at once,
 *     and by the `*_dev` entry points on the NEXT call through the ABI on that handle (any entry point, including
 *     dto_last_stats, which synchronises): that call returns non-zero without doing its own work and clears the flag.
 *   - One in-flight call per handle (solvers call back serially).
 *   - `*_dev` entry points take DEVICE pointers and a hipStream_t (passed as void*); inputs and outputs
 *     stay in HBM and the call returns with the last kernels still in flight on `stream`.  They may wait
 *     on a few 8-16 byte device-to-host readbacks in between (data-dependent launch counts: squarings,
 *     Taylor terms), so they are not graph-capturable.  The plain entry points take HOST pointers, copy
 *     in/out and block until the result is in the caller's buffer.
 *   - A handle may own a SHARD of the knot range [k_lo, k_hi] (1-based, inclusive).  All value
 *     outputs are then the shard-local contiguous slabs described by dto_shard_info; with
 *     k_lo = 1, k_hi = N the slabs are the whole vectors.
 */
#ifndef DTO_ENGINE_H
#define DTO_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DTO_ABI_VERSION 7

/* integrator kinds (src/integrators/) */
#define DTO_INTEGRATOR_BILINEAR 1   /* bilinear_integrator.jl:61-85   */
#define DTO_INTEGRATOR_DERIVATIVE 2 /* derivative_integrator.jl:26-49 */
#define DTO_INTEGRATOR_EXTERNAL 3   /* any other AbstractIntegrator (TimeDependentBilinearIntegrator,
                                       time_dependent_bilinear_integrator.jl:60-244; user integrators): evaluated
                                       on the HOST by the reference's own code, merged by the engine.  Structure is
                                       the generic one (dense x_dim x 2z block per interval, _integrators.jl:49-77);
                                       only x_dim is read from the descriptor */

#define DTO_INTEGRATOR_TIME_DEPENDENT_BILINEAR 4 /* TimeDependentBilinearIntegrator (time_dependent_bilinear_integrator.jl:
                                       60-244) for the parametrised generator family
                                         G(u, t) = sum_{j=0..m} ubar_j ( G_j + sum_c phi_c(t) H_cj ),  ubar_0 = 1,
                                         phi_c(t) = cos(omega_c t) or sin(omega_c t)
                                       (carrier-modulated drives, rotating-frame terms; the reference's own test
                                       G(a) + 0.1 cos(t) I is the case H_c0 = 0.1 I).  Evaluated ON THE DEVICE: defect
                                       x_{k+1} - Phi_k x_k of  dx/dtau = dt_k G(u(tau), t_k + tau dt_k) x,  tau in [0,1], by
                                       classical RK4 with `substeps` fixed steps (the reference integrates adaptively with
                                       Tsit5), controls held (spline_order 0) or interpolated linearly to u_{k+1} (1), with the
                                       exact first and second derivatives of that discrete map.  An arbitrary closure G(u, t)
                                       stays DTO_INTEGRATOR_EXTERNAL */

/* objective term kinds (src/objectives/) */
#define DTO_OBJECTIVE_QUADRATIC_REGULARIZER 1 /* regularizers.jl:38-167   */
#define DTO_OBJECTIVE_LINEAR_REGULARIZER 2    /* regularizers.jl:207-313  */
#define DTO_OBJECTIVE_MINIMUM_TIME 3          /* minimum_time_objective.jl:24-76 */
#define DTO_OBJECTIVE_KNOT_SQDIST 4           /* KnotPointObjective / TerminalObjective with the built-in loss
                                                 l(v, p) = ||v - p||^2 (knot_point_objectives.jl:65-243) */
#define DTO_OBJECTIVE_KNOT_LOWRANK_INFIDELITY 6 /* KnotPointObjective / TerminalObjective with the built-in loss
                                                 l(v) = |1 - ||A v||^2|, A a constant k x n_comps factor: the
                                                 coherent (ket / unitary) fidelity losses in isomorphic
                                                 coordinates, whose per-knot Hessian is the ConstantLowRankHVP
                                                 shape A' G A with G = -2 sign(1 - F) I (knot_hvp.jl:45-84).
                                                 A is passed in `R` (k x n_comps column-major), k in `comp_dim` */
#define DTO_OBJECTIVE_EXTERNAL_GLOBAL 7         /* GlobalObjective / GlobalKnotPointObjective (global_objectives.jl:35-350),
                                                 host closure over [z_t[comps]; global_data[gcomps]] per listed time
                                                 (n_times = 0: one listing of the global variables alone).  Gradient
                                                 and Hessian blocks ACCUMULATE over the listings (:270-271, :341), the
                                                 Hessian entries whose column is a global variable form the tail of
                                                 the CSC order (global columns follow all knot columns) */
#define DTO_OBJECTIVE_EXTERNAL_KNOT 5         /* KnotPointObjective / TerminalObjective with a HOST closure l: the
                                                 caller evaluates Q_i l, its gradient and Hessian per listed time
                                                 (the reference's own ForwardDiff code, knot_point_objectives.jl:
                                                 173-243) and hands the blocks over with dto_set_external; the
                                                 engine merges them at its precomputed offsets */

/* built-in g kinds for NonlinearKnotPointConstraint (knot_point_constraint.jl:27-107) */
#define DTO_CONSTRAINT_NORM_MINUS_C 1   /* g(v) = [ ||v||_2   - c ] */
#define DTO_CONSTRAINT_SQNORM_MINUS_C 2 /* g(v) = [ ||v||_2^2 - c ] */
#define DTO_CONSTRAINT_EXTERNAL_GLOBAL 4 /* NonlinearGlobalConstraint g(global_data[comps]) with g_dim outputs
                                           (global_constraint.jl:20-160): host closure; `comps` index global_data,
                                           `times` is unused; patterns = non-zeros of `jac0` (g_dim x n_comps) and
                                           of `hess0` (n_comps x n_comps, mu = ones; evaluator.jl:166) */
#define DTO_CONSTRAINT_EXTERNAL 3       /* host closure g with g_dim outputs: values, Jacobian blocks and
                                           mu-weighted Hessian blocks come from the caller (dto_set_external),
                                           evaluated with the reference's own code (knot_point_constraint.jl:
                                           235-294); the sparsity pattern is that of `jac0` */

typedef struct dto_integrator_desc {
    int32_t kind;
    int32_t x_off;  /* state component offset inside a knot (0-based) */
    int32_t x_dim;
    int32_t u_off;  /* bilinear: control offset; derivative: offset of the derivative component */
    int32_t u_dim;  /* bilinear: number of drives m; derivative: ignored (= x_dim) */
    const double* G; /* bilinear: (m+1) matrices x_dim*x_dim, column-major, G[0] = G(0) (drift),
                        G[j] = G(e_j) - G(0) (drive j); copied at create.  NULL for derivative. */
    /* DTO_INTEGRATOR_TIME_DEPENDENT_BILINEAR only (zero / NULL otherwise) */
    int32_t t_off;        /* component offset of the time variable t inside a knot (0-based) */
    int32_t spline_order; /* 0: controls held over the interval, 1: linearly interpolated to u_{k+1} */
    int32_t substeps;     /* fixed RK4 steps per interval */
    int32_t n_mod;        /* number of modulation terms c */
    const int32_t* mod_kind; /* [n_mod] 1 = cos(omega t), 2 = sin(omega t) */
    const double* mod_omega; /* [n_mod] */
    const double* H;         /* [n_mod][(m+1)] matrices x_dim*x_dim, column-major; copied at create */
} dto_integrator_desc;

typedef struct dto_objective_desc {
    int32_t kind;
    int32_t comp_off;
    int32_t comp_dim;
    int32_t reserved;
    double weight;          /* CompositeObjective weight (_objectives.jl:106-156) */
    double D;               /* MinimumTimeObjective scale */
    const double* R;        /* comp_dim weights (regularizers); LOWRANK_INFIDELITY: the factor A */
    const double* baseline; /* comp_dim x N column-major, or NULL = zeros (QuadraticRegularizer) */
    const int64_t* times;   /* 1-based knot indices, or NULL = 1:N */
    int64_t n_times;
    /* knot-point kinds (KNOT_SQDIST, KNOT_LOWRANK_INFIDELITY, EXTERNAL_KNOT): */
    const int32_t* comps;   /* knot-local component indices (0-based), vcat of var_names comps */
    int32_t n_comps;
    int32_t reserved2;
    const double* params;   /* n_comps x n_times column-major targets p_i, or NULL = zeros */
    const double* Qs;       /* n_times weights Q_i, or NULL = ones */
    const int32_t* gcomps;  /* EXTERNAL_GLOBAL: indices into global_data (0-based), vcat of global_names comps */
    int32_t n_gcomps;
    int32_t reserved3;
} dto_objective_desc;

typedef struct dto_constraint_desc {
    int32_t kind;
    int32_t equality;       /* 1: g = 0, 0: g <= 0 (src/solvers/solve.jl:54-62) */
    int32_t n_comps;
    int32_t g_dim;          /* outputs per listed time; 0 or 1 for the built-in kinds */
    const int32_t* comps;   /* knot-local component indices (0-based), vcat of var_names comps */
    double c;
    const int64_t* times;   /* 1-based knot indices (required) */
    int64_t n_times;
    const double* jac0;     /* EXTERNAL only: Jacobian blocks at Z0, [n_times] blocks g_dim x n_comps column-major;
                               entries that are exactly 0.0 are outside the pattern (evaluator.jl:136) */
    const double* hess0;    /* EXTERNAL_GLOBAL only: Hessian of sum(g) at Z0, n_comps x n_comps column-major */
} dto_constraint_desc;

/* dto_problem_desc.flags */
#define DTO_FLAG_GENERAL_PATH_ONLY 1 /* bilinear integrators with <= 32 states also take the batched-GEMM path instead of
                                        the fused one-workgroup-per-interval kernel (tests run both and compare) */

typedef struct dto_problem_desc {
    int32_t abi_version;    /* DTO_ABI_VERSION */
    int32_t device;         /* HIP device ordinal */
    int64_t N;              /* knots */
    int32_t z;              /* traj.dim: components per knot */
    int32_t gd;             /* traj.global_dim */
    int32_t dt_idx;         /* component index of the timestep inside a knot (0-based); the engine
                               requires a timestep COMPONENT (bilinear_integrator.jl:123) */
    int32_t eval_hessian;   /* Evaluator(...; eval_hessian) -> features_available (evaluator.jl:293) */
    int32_t n_integrators;
    int32_t n_objectives;
    int32_t n_constraints;
    int32_t flags;          /* DTO_FLAG_* bits, 0 by default */
    const dto_integrator_desc* integrators;
    const dto_objective_desc* objectives;
    const dto_constraint_desc* constraints;
    const double* Z0;       /* initial point, length z*N+gd: constraint sparsity patterns are taken
                               from the numeric Jacobian at Z0 (evaluator.jl:136) */
    int64_t k_lo, k_hi;     /* owned knot range, 1-based inclusive; 0,0 = whole trajectory */
} dto_problem_desc;

typedef struct dto_handle dto_handle;

typedef struct dto_shard_info {
    int64_t k_lo, k_hi;
    int64_t n_vars, n_cons, jac_nnz, hess_nnz;     /* GLOBAL sizes */
    int64_t grad_lo, grad_len;                      /* gradient slab: entries of owned knots */
    int64_t jac_lo, jac_len;                        /* Jacobian value slab (CSC order)        */
    int64_t hess_lo, hess_len;                      /* Hessian value slab (upper-tri CSC order) */
    int64_t cons_len;                               /* local constraint rows (see dto_shard_rows) */
    int32_t n_row_segments;
    int32_t reserved;
} dto_shard_info;

/* lifecycle -- replaces Evaluator(prob; eval_hessian) at ipopt_solver/solver.jl:68-69 and
 * ext/MadNLPSolverExt/solver.jl:81 */
int dto_create(const dto_problem_desc* desc, dto_handle** out);
void dto_destroy(dto_handle* h);
const char* dto_last_error(const dto_handle* h); /* h may be NULL: error of the last failed create */

/* sizes: Evaluator fields n_constraints / n_dynamics_constraints / n_nonlinear_constraints
 * (evaluator.jl:73-76) and structure lengths */
int dto_num_vars(const dto_handle* h, int64_t* out);
int dto_num_cons(const dto_handle* h, int64_t* out);
int dto_num_dynamics_cons(const dto_handle* h, int64_t* out);
int dto_jac_nnz(const dto_handle* h, int64_t* out);
int dto_hess_nnz(const dto_handle* h, int64_t* out);
int dto_features_available(const dto_handle* h, int32_t* grad, int32_t* jac, int32_t* hess); /* evaluator.jl:293-299 */
int dto_get_shard_info(const dto_handle* h, dto_shard_info* out);
/* local constraint buffer = concatenation of global row segments [start1 (1-based), len] */
int dto_shard_rows(const dto_handle* h, int64_t* start1, int64_t* len);

/* Cost model of one eval_constraint_jacobian for intervals first .. first+count-1 (0-based, GLOBAL numbering; Z is the whole NLP
 * vector, host memory): flops from the growth bound of every interval's A_k = dt_k G(u_k) -- squarings of the propagator chain,
 * Taylor terms of the sweep (bilinear_integrator.jl:81: `expv`'s work grows with ||dt G(u)|| too).  Host arithmetic only, also on
 * structure-only handles: a caller that shards knot ranges over GPUs balances them by these instead of by knot
 * count, so that on a pulse whose amplitude varies along the trajectory the slowest rank does not set the step; unequal ranges
 * are gathered in the broadcast form (dto_get_gather_layout). */
int dto_interval_costs(const dto_handle* h, const double* Z, int64_t first, int64_t count, double* cost);

/* structure -- MOI.jacobian_structure (evaluator.jl:364) / MOI.hessian_lagrangian_structure (:385).
 * Writes entries [first, first+count) (0-based position in the GLOBAL CSC order), 1-based pairs. */
int dto_jacobian_structure(const dto_handle* h, int64_t first, int64_t count, int64_t* rows, int64_t* cols);
int dto_hessian_structure(const dto_handle* h, int64_t first, int64_t count, int64_t* rows, int64_t* cols);
/* row bounds handed to the solver, src/solvers/solve.jl:30-65: lower/upper per NLP row */
int dto_constraint_bounds(const dto_handle* h, double* lower, double* upper);

/* Host-evaluated ("external") terms -- SURVEY.md §8f rank 2.  One entry per EXTERNAL term: the external
 * integrators in list order, then the external constraints in list order, then the external objectives.  All pointers are HOST
 * pointers; they are read (copied to the device) by every callback that follows until replaced, so the shim
 * fills whatever the next callback needs and calls dto_set_external first.  Arrays cover ALL listed times of the
 * term, in list order; a sharded handle reads only the blocks of its own knots.
 *   integrator: values = defects, [N-1] x x_dim;  first = Jacobian blocks, [N-1] x (x_dim x 2z col-major, columns
 *               z_k then z_{k+1});  second = Hessian blocks of mu_k' f, [N-1] x (2z x 2z col-major) -- accumulated (+=)
 *               like every integrator block (evaluator.jl:574-598), row <= col entries only
 *   constraint: values = g, [n_times] x g_dim;  first = Jacobian blocks, [n_times] x (g_dim x n_comps col-major);
 *               second = Hessian blocks of mu_i' g, [n_times] x (n_comps x n_comps col-major)
 *   global constraint: values = g [g_dim]; first = Jacobian g_dim x n_comps col-major; second = Hessian of mu' g
 *   global objective:  as "objective" below with v_i = [z_t[comps]; global_data[gcomps]] (block size n_comps + n_gcomps)
 *   objective:  values = Q_i l(v_i, p_i), [n_times];  first = Q_i grad l, [n_times] x n_comps;
 *               second = Q_i Hessian of l, [n_times] x (n_comps x n_comps col-major)
 * Semantics follow the reference: Jacobian entries are assigned, outside-pattern entries dropped
 * (evaluator.jl:491-551); gradient!/hessian! overwrite per listed time, so for a knot listed twice the later
 * block wins; only row <= col Hessian entries are kept (evaluator.jl:637). */
typedef struct dto_external_values {
    const double* values;
    const double* first;
    const double* second;
} dto_external_values;
int dto_num_external(const dto_handle* h, int32_t* n_integrators, int32_t* n_constraints, int32_t* n_objectives);
int dto_set_external(dto_handle* h, int32_t n, const dto_external_values* v);

/* host-pointer callbacks (blocking) */
int dto_eval_objective(dto_handle* h, const double* Z, double* f);                 /* evaluator.jl:304 */
int dto_eval_gradient(dto_handle* h, const double* Z, double* grad);               /* evaluator.jl:310 */
int dto_eval_constraint(dto_handle* h, const double* Z, double* g);                /* evaluator.jl:323 */
int dto_eval_jacobian(dto_handle* h, const double* Z, double* vals);               /* evaluator.jl:368 */
int dto_eval_hessian(dto_handle* h, const double* Z, double sigma, const double* mu,
                     double* vals);                                                /* evaluator.jl:389 */
/* y = J w  /  y = J' w without materialising J on the host (evaluator.jl:406-456) */
int dto_eval_jacobian_product(dto_handle* h, const double* Z, const double* w, double* y);
int dto_eval_jacobian_transpose_product(dto_handle* h, const double* Z, const double* w, double* y);

/* device-pointer callbacks (asynchronous on `stream`; outputs stay in HBM) */
int dto_eval_objective_dev(dto_handle* h, const double* dZ, double* df, void* stream);
int dto_eval_gradient_dev(dto_handle* h, const double* dZ, double* dgrad, void* stream);
int dto_eval_constraint_dev(dto_handle* h, const double* dZ, double* dg, void* stream);
int dto_eval_jacobian_dev(dto_handle* h, const double* dZ, double* dvals, void* stream);
int dto_eval_hessian_dev(dto_handle* h, const double* dZ, double sigma, const double* dmu,
                         double* dvals, void* stream);

/* ---- Multi-GPU: knot ranges sharded over the GPUs of one node, one process (or thread with its own device) per GPU
 * (BASELINE configs[3] "knot range sharded across 8 x MI355X (RCCL allgather)").  The engine owns the RCCL
 * communicator (SURVEY.md §8b "Ownership").  No callback needs a collective: every rank's Jacobian / Hessian / gradient
 * output is one contiguous slab of the global value vector.  The entry points below serve a consumer that wants the WHOLE
 * vector on every GPU (MadNLP's KKT assembly in GPU mode, ext/MadNLPSolverExt/solver.jl:81,
 * src/solvers/madnlp_solver/options.jl:12-16) and the objective's sum over the ranks.
 *
 *   1. one rank:   dto_comm_unique_id(id)                  -- hand the 128 bytes to the others (MPI, a file, a socket)
 *   2. every rank: dto_comm_create(h, id, rank, world)     -- collective; exchanges the ranks' knot ranges
 *   3. every rank: dto_get_gather_layout(h, DTO_VECTOR_JACOBIAN, &L); allocate L.padded_len doubles of device memory `buf`;
 *                  the global value vector is buf + L.front_pad, this rank's slab starts at buf + L.front_pad + L.own_lo
 *   4. per call:   dto_eval_jacobian_dev(h, dZ, buf + L.front_pad + L.own_lo, stream);
 *                  dto_gather_jacobian_dev(h, buf, stream);   -- all ranks, same order; enqueued on `stream`
 *
 * With contiguous knot ranges in rank order the gather is ONE in-place ncclAllGather on `buf` (L.in_place_all_gather = 1:
 * the first and the last rank's slabs are a boundary half-block shorter than the others, which is what front_pad and the
 * padding behind the vector absorb -- a few hundred KB on 17.6 GB); any other assignment of knots to handles (a rank's knots
 * over two handles to overlap the gather of one with the compute of the other) is served by one in-place ncclBroadcast per
 * rank inside a group call, and padded_len = total.  Padding is never read as data.
 * dto_comm_set_ranges is the same bookkeeping WITHOUT a communicator (also on structure-only handles): for callers that move
 * the slabs with a transport of their own and only want the layout. */
#define DTO_COMM_ID_BYTES 128
#define DTO_VECTOR_JACOBIAN 1
#define DTO_VECTOR_HESSIAN 2
#define DTO_VECTOR_GRADIENT 3
#define DTO_VECTOR_CONSTRAINT 4 /* g: a rank's rows are several segments of the global vector (dto_shard_rows), moved by
                                   one broadcast per segment; layout: total = n_cons, no padding, own_lo / own_len unused */
typedef struct dto_gather_layout {
    int64_t total;       /* length of the global vector */
    int64_t padded_len;  /* doubles to allocate */
    int64_t front_pad;   /* the global vector starts here inside the allocation */
    int64_t own_lo;      /* this rank's slab inside the global vector (= dto_shard_info.*_lo) */
    int64_t own_len;
    int32_t in_place_all_gather; /* 1: one ncclAllGather moves every slab; 0: one broadcast per rank */
    int32_t world;
} dto_gather_layout;
int dto_comm_unique_id(void* id128);
int dto_comm_create(dto_handle* h, const void* id128, int32_t rank, int32_t world);
int dto_comm_set_ranges(dto_handle* h, int32_t rank, int32_t world, const int64_t* k_lo, const int64_t* k_hi);
int dto_comm_destroy(dto_handle* h);
int dto_get_gather_layout(const dto_handle* h, int32_t vector, dto_gather_layout* out);
/* every rank's (lo, len) of one value vector, [world] each (VECTOR_JACOBIAN / HESSIAN / GRADIENT) */
int dto_gather_slabs(const dto_handle* h, int32_t vector, int64_t* lo, int64_t* len);
int dto_gather_jacobian_dev(dto_handle* h, double* dbuf, void* stream);
int dto_gather_hessian_dev(dto_handle* h, double* dbuf, void* stream);
int dto_gather_gradient_dev(dto_handle* h, double* dbuf, void* stream);
/* dg_full [n_cons]: the rank's local rows (dg_local, as dto_eval_constraint_dev wrote them) are copied to their global
 * positions, the other ranks' segments arrive by broadcast */
int dto_gather_constraint_dev(dto_handle* h, const double* dg_local, double* dg_full, void* stream);
/* *df := sum over the ranks of *df (the objective's per-shard partial sums, evaluator.jl:304); ncclAllReduce on `stream` */
int dto_allreduce_objective_dev(dto_handle* h, double* df, void* stream);

/* Bound outputs (device-pointer callbacks).  A solver in GPU mode hands the SAME device vector to eval_constraint_jacobian /
 * eval_hessian_lagrangian every iteration (MadNLP's callback buffers; ext/MadNLPSolverExt/solver.jl:81).  Half of a Jacobian slab
 * and ~99 % of a Hessian slab never change -- structural zeros the reference's `fill!(..., 0)` rewrites every call
 * (evaluator.jl:497, :571), the identity blocks of the z_{k+1} halves.  dto_bind_output_dev(h, DTO_VECTOR_JACOBIAN | _HESSIAN, ptr)
 * declares such a vector: the first dto_eval_*_dev call into `ptr` writes it in full, later calls into the same pointer write
 * only what can change (and clear only the runs that kernels accumulate into), PROVIDED the caller has not written the other
 * entries in between (it may read everything and overwrite the variable entries).  Calls with any other pointer, and all
 * host-pointer calls, keep the full-write semantics; ptr = NULL unbinds; a failed call re-arms the full write.
 * (-0.3 ms per Jacobian, -0.25 ms per Hessian at 256 states x 2000 knots: 1.1 GB / 1.7 GB of zero-fill less.) */
int dto_bind_output_dev(dto_handle* h, int32_t vector, double* dptr);

/* Options (name, value); unknown names are an error.
 *   "reuse_forward_sweep" (default 0): interior-point solvers evaluate g, J and H at the same point one after the
 *   other.  With this on, the engine remembers the forward generator sweep of the last callback and re-uses it when
 *   the next callback's Z is bit-identical (compared on the device, one 4-byte readback): eval_constraint after
 *   eval_jacobian then costs a copy, eval_jacobian after eval_constraint sweeps its tangent columns only, eval_hessian
 *   skips its forward sweep and -- after an eval_jacobian at that point -- takes the step budget that call's propagator chain
 *   planned from its exact norms instead of buying them again (a sixth of a Hessian at 1024 states).  Results agree to rounding
 *   either way (the constraint-only
 *   sweep sums its generator products in a different order than the Jacobian's); the benchmark never turns it on (each
 *   callback is timed cold). */
/*   "expm_form" (default 0): evaluation form of the matrix-exponential polynomial in eval_constraint_jacobian.  0 picks per
 *   call by cost: two products for the degree-16 Taylor polynomial (backward-error radius 0.78) or three products for an
 *   order-26 approximant (radius 2.82, i.e. up to two squarings fewer); 2 / 3 force one form (tests, measurements). */
/*   "host_xfer" (default 1): the host-pointer dto_eval_jacobian / dto_eval_hessian copy only the entries that can change
 *   from call to call (about half of a Jacobian slab, ~1 % of a Hessian slab) through a pinned ring and fill the constant
 *   entries (identity / zero runs) with host threads while the GPU computes; 0 copies the whole slab in one piece.  To be set
 *   before the first such call.
 *   "overlap_sweep" (default 1): eval_constraint_jacobian runs its generator sweep on a second stream next to the polynomial
 *   products and squarings of the propagator chain (5-6 % faster at 256 states x 2000 knots); 0 runs one kernel at a time,
 *   which is what per-kernel measurements (dto_profile_get, rocprofv3) need.  Results agree to rounding: bit-identical, except
 *   that next to the chain a 256-state sweep over 1536 intervals or more groups its intervals by twelve instead of nine (a
 *   different, equally fixed summation order of the generator products).
 *   "sweep_form" (default 0): 0 runs the generator sweep as one persistent launch where that form applies, 1 always one
 *   launch per Taylor step.
 *   "chain_form" (default 0): 0 runs the propagator chain of a 33..64-state integrator as ONE launch (a workgroup per interval, the
 *   evaluation form chosen per interval on the device), 1 always as batched-GEMM launches over all intervals (the form of larger
 *   integrators).  Same approximants and radii either way; results agree to rounding.
 *   "chain_chunk" (default 0 = the engine's workspace budget): at most this many intervals per chunk of the propagator
 *   chain (what a 16000-knot trajectory does by itself; tests use it to exercise the chunk loop on small problems).
 *   "host_xfer_check" (default 0): every host-pointer dto_eval_jacobian / dto_eval_hessian also copies the whole device slab and
 *   compares it bit for bit with the vector it assembled from the variable runs and the constants; a difference (a kernel that
 *   wrote an entry the hand-off plan does not list) fails the call.  The repository's GPU tests run with it on.
 *   "debug_bad_launch": TUNING builds only (libdto_engine_t.so) -- 1 gives the next callbacks' kernels an invalid launch
 *   configuration, which must come back as a non-zero return code with text (test of the error convention); the product
 *   library refuses the name. */
/*   "deterministic" (default 0).  Run to run the engine is bit-reproducible without any option: every reduction has a fixed
 *   order (column sums of the generator-subspace GEMMs per 64-row chunk, objective partial sums, listings of a term that
 *   repeats a knot layer by layer, contributions to global-variable entries in listing order; the sweeps' K order is a function
 *   of the block index).  What still depends on HOW a result is asked for is switched off by 1: the Jacobian's sweep keeps the
 *   interval grouping it has alone on the chip (else `overlap_sweep` changes the summation order at 256 states), and the
 *   host-pointer dto_eval_jacobian runs the propagator chain in the same chunks as dto_eval_jacobian_dev (else the early
 *   hand-over of -E_k caps the chunk, and the evaluation form is decided per chunk).  Cost at 256 x 2000: +0.1..0.4 ms per
 *   device-resident Jacobian, and the host-pointer Jacobian loses the overlap of its PCIe copy with the chain (23 -> 34 ms).
 *   dto_eval_jacobian_product / _transpose_product are covered since round 4: every entry of y has one writer per launch with a
 *   fixed internal order (J w gathers row by row; J' w writes per knot, listings that repeat a knot in listing order). */
int dto_set_option(dto_handle* h, const char* name, int64_t value);

/* measurement hooks: HIP-event timing of the engine's kernels on the stream they are launched on */
int dto_profile_enable(dto_handle* h, int32_t on);
int dto_profile_reset(dto_handle* h);
/* name: "bgemm" (batched f64 MFMA GEMM of the propagator chain: every template instance), its parts
 * "bgemm_horner" (the products with a fused polynomial epilogue) / "bgemm_square" / "bgemm_plain" / "chain64" (the one-launch
 * chain of a 33..64-state integrator, priced at six products per interval), "basis"
 * (generator-subspace GEMM), "expmv" (forward generator sweeps and the pairing products), "expmv_adjoint" (the Hessian's adjoint
 * sweep: its dominant kernel), "all"; "basis_multi" / "basis_k" (the two generator-subspace launches apart: A^2..A^4, and the
 * factor K), and the bandwidth-bound assembly kernels "zero_fill" (the Jacobian's / Hessian's fill!(., 0)), "build_A" (A_k from the
 * generators), "assembly" (the writers of the bilinear Jacobian's tangent columns).
 * Returns accumulated device milliseconds, launches and algorithmic FLOPs of those launches -- for the three assembly names the
 * third output is the launches' algorithmic BYTES (what they must read and write), not FLOPs. */
int dto_profile_get(dto_handle* h, const char* name, double* ms, int64_t* launches, double* flops);
/* diagnostics of the last Jacobian call: max squarings used, Taylor terms used by the tangent sweep */
int dto_last_stats(const dto_handle* h, int32_t* max_squarings, int32_t* expmv_terms);

#ifdef __cplusplus
}
#endif
#endif /* DTO_ENGINE_H */
