# DTOEngine.jl -- the reference-side binding of libdto_engine.so (include/dto_engine.h, ABI version 6).
#
# Drop this file into DirectTrajOpt.jl (e.g. src/solvers/DTOEngine.jl, `include`d from src/solvers/_solvers.jl) and
# replace the evaluator at the two swap points:
#
#     src/solvers/ipopt_solver/solver.jl:68-69   evaluator = DTOEngine.GPUEvaluator(prob; eval_hessian = options.eval_hessian)
#     ext/MadNLPSolverExt/solver.jl:81           evaluator = DTOEngine.GPUEvaluator(prob; eval_hessian = true)
#
# `GPUEvaluator <: MOI.AbstractNLPEvaluator` implements the same MOI methods as `Solvers.Evaluator`
# (src/solvers/evaluator.jl:291-456) and carries the fields other code reads off it (`trajectory`, `n_constraints`,
# `n_dynamics_constraints`, `n_nonlinear_constraints`, evaluator.jl:66-98).  Everything the engine implements natively
# runs on the GPU: BilinearIntegrator, DerivativeIntegrator, QuadraticRegularizer, LinearRegularizer,
# MinimumTimeObjective, CompositeObjective.  Terms that are Julia closures (KnotPointObjective, NonlinearKnotPoint-
# Constraint, TimeDependentBilinearIntegrator and other integrators, GlobalObjective, GlobalKnotPointObjective,
# NonlinearGlobalConstraint) are evaluated HERE with the reference's own functions (`evaluate!`, `eval_jacobian`,
# `eval_hessian_of_lagrangian`, `objective_value`, `gradient!`, `get_full_hessian`, or ForwardDiff on the closure exactly as
# those functions do) and merged by the engine at its precomputed offsets (`dto_set_external`).
#
# Beyond the MOI surface the file binds the device-resident entry points (`*_dev!`: MadNLP GPU mode,
# src/solvers/madnlp_solver/options.jl:12-16 -- value vectors never cross PCIe) and the engine's multi-GPU collectives
# (`comm_create!`, `gather_jacobian_dev!` and friends: RCCL over xGMI behind the C ABI).
#
# This file cannot be executed in the build environment of the engine (no Julia there); tests/test_julia_shim.py checks
# its struct layouts and ccall signatures against include/dto_engine.h.
module DTOEngine

using LinearAlgebra
using SparseArrays
using ForwardDiff
import MathOptInterface as MOI
using NamedTrajectories
using TrajectoryIndexingUtils: slice
using ..Problems: DirectTrajOptProblem
using ..Integrators: AbstractIntegrator, BilinearIntegrator, DerivativeIntegrator
using ..Objectives: AbstractObjective, CompositeObjective, NullObjective, QuadraticRegularizer, LinearRegularizer,
    MinimumTimeObjective, GlobalObjective, GlobalKnotPointObjective, objective_value, gradient!, get_full_hessian
using ..Constraints: AbstractNonlinearConstraint, NonlinearKnotPointConstraint, NonlinearGlobalConstraint
using ..CommonInterface: evaluate!, eval_jacobian, eval_hessian_of_lagrangian

const lib = get(ENV, "DTO_ENGINE_LIB", "libdto_engine.so")
const DTO_ABI_VERSION = Int32(7)

const DTO_INTEGRATOR_BILINEAR = Int32(1)
const DTO_INTEGRATOR_DERIVATIVE = Int32(2)
const DTO_INTEGRATOR_EXTERNAL = Int32(3)
const DTO_INTEGRATOR_TIME_DEPENDENT_BILINEAR = Int32(4)
const DTO_OBJECTIVE_QUADRATIC_REGULARIZER = Int32(1)
const DTO_OBJECTIVE_LINEAR_REGULARIZER = Int32(2)
const DTO_OBJECTIVE_MINIMUM_TIME = Int32(3)
const DTO_OBJECTIVE_EXTERNAL_KNOT = Int32(5)
const DTO_OBJECTIVE_EXTERNAL_GLOBAL = Int32(7)
const DTO_CONSTRAINT_EXTERNAL = Int32(3)
const DTO_CONSTRAINT_EXTERNAL_GLOBAL = Int32(4)
const DTO_COMM_ID_BYTES = Int32(128)
const DTO_VECTOR_JACOBIAN = Int32(1)
const DTO_VECTOR_HESSIAN = Int32(2)
const DTO_VECTOR_GRADIENT = Int32(3)
const DTO_VECTOR_CONSTRAINT = Int32(4)

# ---- plain-C structs, field for field as in include/dto_engine.h ----------------------------------------------------
struct IntegratorDesc
    kind::Int32
    x_off::Int32
    x_dim::Int32
    u_off::Int32
    u_dim::Int32
    G::Ptr{Float64}
    t_off::Int32
    spline_order::Int32
    substeps::Int32
    n_mod::Int32
    mod_kind::Ptr{Int32}
    mod_omega::Ptr{Float64}
    H::Ptr{Float64}
end
# bilinear / derivative / host-evaluated integrators leave the time-dependent fields empty
IntegratorDesc(kind, x_off, x_dim, u_off, u_dim, G) =
    IntegratorDesc(kind, x_off, x_dim, u_off, u_dim, G, Int32(0), Int32(0), Int32(0), Int32(0), Ptr{Int32}(C_NULL),
                   Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL))

struct ObjectiveDesc
    kind::Int32
    comp_off::Int32
    comp_dim::Int32
    reserved::Int32
    weight::Float64
    D::Float64
    R::Ptr{Float64}
    baseline::Ptr{Float64}
    times::Ptr{Int64}
    n_times::Int64
    comps::Ptr{Int32}
    n_comps::Int32
    reserved2::Int32
    params::Ptr{Float64}
    Qs::Ptr{Float64}
    gcomps::Ptr{Int32}
    n_gcomps::Int32
    reserved3::Int32
end

struct ConstraintDesc
    kind::Int32
    equality::Int32
    n_comps::Int32
    g_dim::Int32
    comps::Ptr{Int32}
    c::Float64
    times::Ptr{Int64}
    n_times::Int64
    jac0::Ptr{Float64}
    hess0::Ptr{Float64}
end

struct ProblemDesc
    abi_version::Int32
    device::Int32
    N::Int64
    z::Int32
    gd::Int32
    dt_idx::Int32
    eval_hessian::Int32
    n_integrators::Int32
    n_objectives::Int32
    n_constraints::Int32
    flags::Int32
    integrators::Ptr{IntegratorDesc}
    objectives::Ptr{ObjectiveDesc}
    constraints::Ptr{ConstraintDesc}
    Z0::Ptr{Float64}
    k_lo::Int64
    k_hi::Int64
end

struct ExternalValues
    values::Ptr{Float64}
    first::Ptr{Float64}
    second::Ptr{Float64}
end

struct GatherLayout
    total::Int64
    padded_len::Int64
    front_pad::Int64
    own_lo::Int64
    own_len::Int64
    in_place_all_gather::Int32
    world::Int32
end

# ---- TimeDependentBilinearIntegrator on the device ------------------------------------------------------------------
"""
The generator family the engine integrates on the GPU (DTO_INTEGRATOR_TIME_DEPENDENT_BILINEAR):

    G(u, t) = sum_{j=0..m} ubar_j ( G[:, :, j+1] + sum_c phi_c(t) H[:, :, j+1, c] ),  ubar_0 = 1,  phi_c = cos | sin (omegas[c] t)

The object is callable, so it is also the `G` the reference's constructor takes:

    fam = DTOEngine.ModulatedGenerators(G, Int32[1], [1.7], H; substeps = 32)
    B = TimeDependentBilinearIntegrator(fam, :x, :u, :t, traj; spline_order = 1)
    DTOEngine.device_family!(B, fam)        # tells the evaluator to run B on the device

(`G` is captured inside the ODE problem the integrator builds, out of reach of the struct's fields, hence the registration.)
The engine integrates with classical RK4, `substeps` fixed steps per interval, and returns the exact derivatives of that
scheme; the reference integrates adaptively (Tsit5).  Any other closure keeps running in Julia and is merged.
"""
struct ModulatedGenerators
    G::Array{Float64,3}
    kinds::Vector{Int32}      # 1 = cos, 2 = sin
    omegas::Vector{Float64}
    H::Array{Float64,4}       # n x n x (m+1) x n_mod
    substeps::Int
end
ModulatedGenerators(G, kinds, omegas, H; substeps::Int = 32) =
    ModulatedGenerators(Array{Float64,3}(G), Vector{Int32}(kinds), Vector{Float64}(omegas), Array{Float64,4}(H), substeps)
function (g::ModulatedGenerators)(u, t)
    ub = vcat(1.0, u)
    M = sum(ub[j] .* g.G[:, :, j] for j in eachindex(ub))
    for c in eachindex(g.omegas)
        phi = g.kinds[c] == 1 ? cos(g.omegas[c] * t) : sin(g.omegas[c] * t)
        M = M .+ phi .* sum(ub[j] .* g.H[:, :, j, c] for j in eachindex(ub))
    end
    return M
end
const DEVICE_FAMILIES = IdDict{Any,ModulatedGenerators}()
device_family!(B, fam::ModulatedGenerators) = (DEVICE_FAMILIES[B] = fam; B)

# ---- the evaluator --------------------------------------------------------------------------------------------------
mutable struct GPUEvaluator <: MOI.AbstractNLPEvaluator
    handle::Ptr{Cvoid}
    trajectory::NamedTrajectory
    n_variables::Int
    n_constraints::Int
    n_dynamics_constraints::Int
    n_nonlinear_constraints::Int
    eval_hessian::Bool
    jacobian_structure::Vector{Tuple{Int,Int}}
    hessian_structure::Vector{Tuple{Int,Int}}
    # host-evaluated terms, in the engine's slot order: integrators, constraints, objectives
    ext_integrators::Vector{Tuple{AbstractIntegrator,Int}}          # (integrator, first NLP row, 1-based)
    ext_constraints::Vector{Tuple{AbstractNonlinearConstraint,Int}}  # (constraint, first NLP row, 1-based)
    ext_objectives::Vector{AbstractObjective}
    ext_comps::Dict{Any,Vector{Int}}                                  # term -> knot-local component indices (1-based)
    ext_gcomps::Dict{Any,Vector{Int}}                                 # Global* term -> indices into global_data (1-based)
    staging::Vector{Vector{Float64}}                                  # keeps the blocks alive across the ccall
end

function check(ev::GPUEvaluator, rc::Integer)
    rc == 0 && return nothing
    error(unsafe_string(@ccall lib.dto_last_error(ev.handle::Ptr{Cvoid})::Cstring))
end

"""
Generators of a `BilinearIntegrator`: the struct stores only the closure `f`, which captured the user's `G`
(src/integrators/bilinear_integrator.jl:61-81), so `B.f.G` is that function.  `G` must be affine in `u`:
`G(u) = G_0 + sum_j u_j G_j`; the engine takes `G_0 = G(0)` and `G_j = G(e_j) - G(0)`.
"""
function generators(B::BilinearIntegrator, traj::NamedTrajectory)
    G = B.f.G
    m = traj.dims[B.u_name]
    G0 = Matrix{Float64}(G(zeros(m)))
    Gs = [Matrix{Float64}(G(Float64.(1:m .== j))) .- G0 for j = 1:m]
    u = randn(m)
    Gu = G0 + sum(u[j] .* Gs[j] for j = 1:m; init = zeros(size(G0)))
    isapprox(Matrix{Float64}(G(u)), Gu; rtol = 1e-12, atol = 1e-12) ||
        error("DTOEngine: G(u) of the BilinearIntegrator on :$(B.x_name) is not affine in u; keep Solvers.Evaluator for it")
    return cat(G0, Gs...; dims = 3)   # n x n x (m+1), column-major = the ABI layout
end

flatten(obj::NullObjective) = Tuple{AbstractObjective,Float64}[]
flatten(obj::AbstractObjective) = Tuple{AbstractObjective,Float64}[(obj, 1.0)]
function flatten(obj::CompositeObjective)
    out = Tuple{AbstractObjective,Float64}[]
    for (o, w) in zip(obj.objectives, obj.weights)
        o isa NullObjective && continue
        o isa CompositeObjective && error("DTOEngine: nested CompositeObjective (the reference's `+` flattens them)")
        push!(out, (o, w))
    end
    return out
end

first0(traj, name) = Int32(first(traj.components[name]) - 1)   # 0-based offset of a component inside a knot

function GPUEvaluator(prob::DirectTrajOptProblem; eval_hessian::Bool = true, device::Integer = 0, k_lo::Integer = 0, k_hi::Integer = 0)
    traj = prob.trajectory
    traj.timestep isa Symbol || error("DTOEngine: the engine needs a timestep component (bilinear_integrator.jl:123)")
    keep = Any[]   # everything the descriptors point at, alive until dto_create returns

    # integrators, in list order (their rows stack in that order, evaluator.jl:213-217)
    idescs = IntegratorDesc[]
    ext_integrators = Tuple{AbstractIntegrator,Int}[]
    row = 1
    for integ in prob.integrators
        if integ isa BilinearIntegrator
            G = generators(integ, traj)
            push!(keep, G)
            push!(idescs, IntegratorDesc(DTO_INTEGRATOR_BILINEAR, first0(traj, integ.x_name), Int32(integ.x_dim),
                                         first0(traj, integ.u_name), Int32(traj.dims[integ.u_name]), pointer(G)))
        elseif integ isa DerivativeIntegrator
            push!(idescs, IntegratorDesc(DTO_INTEGRATOR_DERIVATIVE, first0(traj, integ.x_name), Int32(integ.x_dim),
                                         first0(traj, integ.ẋ_name), Int32(integ.x_dim), Ptr{Float64}(C_NULL)))
        elseif haskey(DEVICE_FAMILIES, integ)   # TimeDependentBilinearIntegrator with a registered generator family
            fam = DEVICE_FAMILIES[integ]
            push!(keep, fam)
            push!(idescs, IntegratorDesc(DTO_INTEGRATOR_TIME_DEPENDENT_BILINEAR, first0(traj, integ.x_name), Int32(integ.x_dim),
                                         first0(traj, integ.u_name), Int32(integ.u_dim), pointer(fam.G),
                                         first0(traj, integ.t_name), Int32(integ.spline_order), Int32(fam.substeps),
                                         Int32(length(fam.omegas)), pointer(fam.kinds), pointer(fam.omegas), pointer(fam.H)))
        else   # TimeDependentBilinearIntegrator with an arbitrary closure, user integrators: evaluated here, merged by the engine
            push!(idescs, IntegratorDesc(DTO_INTEGRATOR_EXTERNAL, Int32(0), Int32(integ.x_dim), Int32(0), Int32(0), Ptr{Float64}(C_NULL)))
            push!(ext_integrators, (integ, row))
        end
        row += integ.dim
    end
    n_dyn = row - 1

    # objective terms (CompositeObjective flattened with its weights, _objectives.jl:106-156)
    odescs = ObjectiveDesc[]
    ext_objectives = AbstractObjective[]
    ext_comps = Dict{Any,Vector{Int}}()
    ext_gcomps = Dict{Any,Vector{Int}}()
    null = Ptr{Float64}(C_NULL)
    for (o, w) in flatten(prob.objective)
        if o isa QuadraticRegularizer || o isa LinearRegularizer
            R = Vector{Float64}(o.R)
            times = Vector{Int64}(o.times)
            base = o isa QuadraticRegularizer ? Matrix{Float64}(o.baseline) : zeros(0, 0)
            push!(keep, R, times, base)
            push!(odescs, ObjectiveDesc(o isa QuadraticRegularizer ? DTO_OBJECTIVE_QUADRATIC_REGULARIZER : DTO_OBJECTIVE_LINEAR_REGULARIZER,
                                        first0(traj, o.name), Int32(traj.dims[o.name]), Int32(0), w, 0.0, pointer(R),
                                        o isa QuadraticRegularizer ? pointer(base) : null, pointer(times), length(times),
                                        Ptr{Int32}(C_NULL), Int32(0), Int32(0), null, null, Ptr{Int32}(C_NULL), Int32(0), Int32(0)))
        elseif o isa MinimumTimeObjective
            push!(odescs, ObjectiveDesc(DTO_OBJECTIVE_MINIMUM_TIME, Int32(0), Int32(0), Int32(0), w, o.D, null, null,
                                        Ptr{Int64}(C_NULL), 0, Ptr{Int32}(C_NULL), Int32(0), Int32(0), null, null,
                                        Ptr{Int32}(C_NULL), Int32(0), Int32(0)))
        elseif o isa GlobalObjective || o isa GlobalKnotPointObjective
            # closures over [z_t[var_names]; global_data[global_names]] (global_objectives.jl:35-350): blocks per listing, the
            # engine ACCUMULATES them (:270-271, :341); a GlobalObjective is one listing of the global variables alone
            comps1 = o isa GlobalKnotPointObjective ? vcat([collect(traj.components[n]) for n in o.var_names]...) : Int[]
            gcomps1 = vcat([collect(traj.global_components[n]) for n in o.global_names]...)
            comps, gcomps = Int32.(comps1 .- 1), Int32.(gcomps1 .- 1)
            times = o isa GlobalKnotPointObjective ? Vector{Int64}(o.times) : Int64[]
            push!(keep, comps, gcomps, times)
            push!(odescs, ObjectiveDesc(DTO_OBJECTIVE_EXTERNAL_GLOBAL, Int32(0), Int32(0), Int32(0), w, 0.0, null, null,
                                        isempty(times) ? Ptr{Int64}(C_NULL) : pointer(times), length(times),
                                        isempty(comps) ? Ptr{Int32}(C_NULL) : pointer(comps), Int32(length(comps)), Int32(0), null, null,
                                        pointer(gcomps), Int32(length(gcomps)), Int32(0)))
            push!(ext_objectives, o)
            ext_comps[o] = comps1
            ext_gcomps[o] = gcomps1
        elseif hasproperty(o, :var_names) && hasproperty(o, :times)   # KnotPointObjective / TerminalObjective: host closure
            comps1 = vcat([collect(traj.components[n]) for n in o.var_names]...)
            comps = Int32.(comps1 .- 1)
            times = Vector{Int64}(o.times)
            push!(keep, comps, times)
            push!(odescs, ObjectiveDesc(DTO_OBJECTIVE_EXTERNAL_KNOT, Int32(0), Int32(0), Int32(0), w, 0.0, null, null,
                                        pointer(times), length(times), pointer(comps), Int32(length(comps)), Int32(0), null, null,
                                        Ptr{Int32}(C_NULL), Int32(0), Int32(0)))
            push!(ext_objectives, o)
            ext_comps[o] = comps1
        else
            error("DTOEngine: objective term $(typeof(o)) is not supported; keep Solvers.Evaluator for this problem")
        end
    end

    # nonlinear constraints: rows follow the dynamics (evaluator.jl:219-223); closures -> EXTERNAL with the pattern of the
    # Jacobian at the initial point (evaluator.jl:136)
    cdescs = ConstraintDesc[]
    ext_constraints = Tuple{AbstractNonlinearConstraint,Int}[]
    for con in prob.constraints
        con isa AbstractNonlinearConstraint || continue   # linear constraints go to MOI directly (solve.jl)
        if con isa NonlinearGlobalConstraint
            # g(global_data[global_names]) (global_constraint.jl:20-160): patterns from the Jacobian and the Hessian of sum(g)
            # at the initial point (evaluator.jl:136, :166)
            gcomps1 = vcat([collect(traj.global_components[n]) for n in con.global_names]...)
            gcomps = Int32.(gcomps1 .- 1)
            g0 = Vector{Float64}(traj.global_data[gcomps1])
            jac0 = vec(Matrix{Float64}(ForwardDiff.jacobian(con.g, g0)))
            hess0 = vec(Matrix{Float64}(ForwardDiff.hessian(x -> sum(con.g(x)), g0)))
            push!(keep, gcomps, jac0, hess0)
            push!(cdescs, ConstraintDesc(DTO_CONSTRAINT_EXTERNAL_GLOBAL, Int32(con.equality), Int32(length(gcomps)), Int32(con.dim),
                                         pointer(gcomps), 0.0, Ptr{Int64}(C_NULL), 0, pointer(jac0), pointer(hess0)))
            push!(ext_constraints, (con, row))
            ext_gcomps[con] = gcomps1
            row += con.dim
            continue
        end
        con isa NonlinearKnotPointConstraint || error("DTOEngine: constraint $(typeof(con)) is not supported")
        comps1 = vcat([collect(traj.components[n]) for n in con.var_names]...)
        comps = Int32.(comps1 .- 1)
        times = Vector{Int64}(con.times)
        J0 = eval_jacobian(con, traj)
        jac0 = knot_blocks(J0, con, comps1, traj)
        push!(keep, comps, times, jac0)
        push!(cdescs, ConstraintDesc(DTO_CONSTRAINT_EXTERNAL, Int32(con.equality), Int32(length(comps)), Int32(con.g_dim),
                                     pointer(comps), 0.0, pointer(times), length(times), pointer(jac0), null))
        push!(ext_constraints, (con, row))
        ext_comps[con] = comps1
        row += con.dim
    end

    Z0 = vcat(traj.datavec, traj.global_data)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve keep idescs odescs cdescs Z0 begin
        desc = Ref(ProblemDesc(DTO_ABI_VERSION, Int32(device), Int64(traj.N), Int32(traj.dim), Int32(traj.global_dim),
                               first0(traj, traj.timestep), Int32(eval_hessian), Int32(length(idescs)), Int32(length(odescs)),
                               Int32(length(cdescs)), Int32(0), pointer(idescs), pointer(odescs), pointer(cdescs),
                               pointer(Z0), Int64(k_lo), Int64(k_hi)))
        rc = @ccall lib.dto_create(desc::Ptr{ProblemDesc}, h::Ptr{Ptr{Cvoid}})::Cint
        rc == 0 || error(unsafe_string(@ccall lib.dto_last_error(C_NULL::Ptr{Cvoid})::Cstring))
    end
    handle = h[]
    n = Ref{Int64}(0)
    @ccall lib.dto_num_vars(handle::Ptr{Cvoid}, n::Ptr{Int64})::Cint
    n_vars = Int(n[])
    @ccall lib.dto_num_cons(handle::Ptr{Cvoid}, n::Ptr{Int64})::Cint
    n_cons = Int(n[])
    @ccall lib.dto_num_dynamics_cons(handle::Ptr{Cvoid}, n::Ptr{Int64})::Cint
    @assert Int(n[]) == n_dyn
    function structure(count_fn, fill_fn)
        cnt = Ref{Int64}(0)
        count_fn(cnt)
        rows = Vector{Int64}(undef, cnt[])
        cols = Vector{Int64}(undef, cnt[])
        fill_fn(cnt[], rows, cols) == 0 || error("DTOEngine: structure query failed")
        return collect(zip(Int.(rows), Int.(cols)))
    end
    jstruct = structure(c -> (@ccall lib.dto_jac_nnz(handle::Ptr{Cvoid}, c::Ptr{Int64})::Cint),
                        (c, r, k) -> (@ccall lib.dto_jacobian_structure(handle::Ptr{Cvoid}, 0::Int64, c::Int64, r::Ptr{Int64}, k::Ptr{Int64})::Cint))
    hstruct = structure(c -> (@ccall lib.dto_hess_nnz(handle::Ptr{Cvoid}, c::Ptr{Int64})::Cint),
                        (c, r, k) -> (@ccall lib.dto_hessian_structure(handle::Ptr{Cvoid}, 0::Int64, c::Int64, r::Ptr{Int64}, k::Ptr{Int64})::Cint))
    ev = GPUEvaluator(handle, traj, n_vars, n_cons, n_dyn, n_cons - n_dyn, eval_hessian, jstruct, hstruct,
                      ext_integrators, ext_constraints, ext_objectives, ext_comps, ext_gcomps, Vector{Float64}[])
    finalizer(e -> (@ccall lib.dto_destroy(e.handle::Ptr{Cvoid})::Cvoid), ev)
    return ev
end

# ---- host-evaluated terms: the reference's own functions produce the blocks -----------------------------------------

# _update_trajectory_cache! of the reference (evaluator.jl:474-482)
function update!(ev::GPUEvaluator, Z::AbstractVector{Float64})
    traj = ev.trajectory
    nd = traj.dim * traj.N
    traj.datavec .= @view Z[1:nd]
    traj.global_dim > 0 && (traj.global_data .= @view Z[nd+1:end])
    return traj
end

# dense per-listing blocks (g_dim x n_comps, column-major, listing after listing) of a knot constraint's sparse Jacobian
function knot_blocks(J::SparseMatrixCSC, con::NonlinearKnotPointConstraint, comps1::Vector{Int}, traj::NamedTrajectory)
    out = Vector{Float64}(undef, con.g_dim * length(comps1) * length(con.times))
    p = 0
    for (i, t) in enumerate(con.times)
        blk = Matrix(J[slice(i, con.g_dim), slice(t, comps1, traj.dim)])
        out[p+1:p+length(blk)] .= vec(blk)
        p += length(blk)
    end
    return out
end

# need: 0 values, 1 + first derivatives, 2 + second derivatives; -1: this callback does not read that family
function stage_external!(ev::GPUEvaluator, Z::AbstractVector{Float64}; con_need::Int = -1, obj_need::Int = -1, μ = nothing)
    n_ext = length(ev.ext_integrators) + length(ev.ext_constraints) + length(ev.ext_objectives)
    n_ext == 0 && return nothing
    traj = update!(ev, Z)
    z = traj.dim
    vals = fill(ExternalValues(C_NULL, C_NULL, C_NULL), n_ext)
    empty!(ev.staging)
    stage(v::Vector{Float64}) = (push!(ev.staging, v); pointer(v))
    slot = 0
    for (integ, row) in ev.ext_integrators
        slot += 1
        con_need < 0 && continue
        δ = zeros(integ.dim)
        evaluate!(δ, integ, traj)
        jac_p = Ptr{Float64}(C_NULL)
        hess_p = Ptr{Float64}(C_NULL)
        if con_need >= 1
            J = eval_jacobian(integ, traj)
            blocks = Vector{Float64}(undef, integ.x_dim * 2z * (traj.N - 1))
            for k = 1:traj.N-1
                blocks[(k-1)*integ.x_dim*2z+1:k*integ.x_dim*2z] .= vec(Matrix(J[slice(k, integ.x_dim), slice(k, 1:2z, z)]))
            end
            jac_p = stage(blocks)
        end
        if con_need >= 2
            H = eval_hessian_of_lagrangian(integ, traj, μ[row:row+integ.dim-1])
            blocks = zeros(4 * z * z * (traj.N - 1))
            for k = 1:traj.N-1
                # one 2z x 2z block per interval; the reference's matrix holds the SUM of neighbouring intervals on the
                # shared diagonal block, so take interval k's own part: evaluate the integrator's blocks one by one
                blocks[(k-1)*4z*z+1:k*4z*z] .= vec(interval_hessian(integ, traj, μ[row:row+integ.dim-1], k, H))
            end
            hess_p = stage(blocks)
        end
        vals[slot] = ExternalValues(stage(δ), jac_p, hess_p)
    end
    for (con, row) in ev.ext_constraints
        slot += 1
        con_need < 0 && continue
        if con isa NonlinearGlobalConstraint   # one listing: values g, Jacobian g_dim x n_comps, Hessian of mu' g (global_constraint.jl:102-134)
            gv = Vector{Float64}(traj.global_data[ev.ext_gcomps[con]])
            jac_p = con_need >= 1 ? stage(vec(Matrix{Float64}(ForwardDiff.jacobian(con.g, gv)))) : Ptr{Float64}(C_NULL)
            μc = con_need >= 2 ? Vector{Float64}(μ[row:row+con.dim-1]) : Float64[]
            hess_p = con_need >= 2 ? stage(vec(Matrix{Float64}(ForwardDiff.hessian(x -> μc' * con.g(x), gv)))) : Ptr{Float64}(C_NULL)
            vals[slot] = ExternalValues(stage(Vector{Float64}(con.g(gv))), jac_p, hess_p)
            continue
        end
        g = zeros(con.dim)
        evaluate!(g, con, traj)
        jac_p = Ptr{Float64}(C_NULL)
        hess_p = Ptr{Float64}(C_NULL)
        comps1 = ev.ext_comps[con]
        con_need >= 1 && (jac_p = stage(knot_blocks(eval_jacobian(con, traj), con, comps1, traj)))
        if con_need >= 2
            H = eval_hessian_of_lagrangian(con, traj, μ[row:row+con.dim-1])
            nc = length(comps1)
            blocks = Vector{Float64}(undef, nc * nc * length(con.times))
            for (i, t) in enumerate(con.times)
                r = slice(t, comps1, z)
                blocks[(i-1)*nc*nc+1:i*nc*nc] .= vec(Matrix(H[r, r]))
            end
            hess_p = stage(blocks)
        end
        vals[slot] = ExternalValues(stage(g), jac_p, hess_p)
    end
    for o in ev.ext_objectives
        slot += 1
        obj_need < 0 && continue
        if o isa GlobalObjective || o isa GlobalKnotPointObjective
            # blocks over xg = [z_t[comps]; global_data[gcomps]] per listing, differentiated as the reference does
            # (ForwardDiff on the closure, global_objectives.jl:76-79, :117, :258, :330)
            gv = Vector{Float64}(traj.global_data[ev.ext_gcomps[o]])
            listings = o isa GlobalObjective ? [(x -> o.Q * o.ℓ(x), gv)] :
                [(let i = i; x -> o.Qs[i] * o.ℓ(x, o.params[i]) end, vcat(vcat([traj[t][n] for n in o.var_names]...), gv))
                 for (i, t) in enumerate(o.times)]
            v = Float64[f(x) for (f, x) in listings]
            grad_p = obj_need >= 1 ? stage(vcat([ForwardDiff.gradient(f, x) for (f, x) in listings]...)) : Ptr{Float64}(C_NULL)
            hess_p = obj_need >= 2 ? stage(vcat([vec(ForwardDiff.hessian(f, x)) for (f, x) in listings]...)) : Ptr{Float64}(C_NULL)
            vals[slot] = ExternalValues(stage(v), grad_p, hess_p)
            continue
        end
        comps1 = ev.ext_comps[o]
        nc = length(comps1)
        nt = length(o.times)
        v = [o.Qs[i] * o.ℓ(vcat([traj[t][n] for n in o.var_names]...), o.params[i]) for (i, t) in enumerate(o.times)]
        grad_p = Ptr{Float64}(C_NULL)
        hess_p = Ptr{Float64}(C_NULL)
        if obj_need >= 1
            ∇ = zeros(ev.n_variables)
            gradient!(∇, o, traj)   # overwrites per listed time (knot_point_objectives.jl:198): the engine keeps the last listing
            grad_p = stage(vcat([∇[slice(t, comps1, z)] for t in o.times]...))
        end
        if obj_need >= 2
            H = get_full_hessian(o, traj)
            H = H + triu(H, 1)'     # the reference returns the upper triangle (knot_point_objectives.jl:242)
            blocks = Vector{Float64}(undef, nc * nc * nt)
            for (i, t) in enumerate(o.times)
                r = slice(t, comps1, z)
                blocks[(i-1)*nc*nc+1:i*nc*nc] .= vec(Matrix(H[r, r]))
            end
            hess_p = stage(blocks)
        end
        vals[slot] = ExternalValues(stage(Vector{Float64}(v)), grad_p, hess_p)
    end
    n32 = Int32(n_ext)
    GC.@preserve vals check(ev, @ccall lib.dto_set_external(ev.handle::Ptr{Cvoid}, n32::Int32, vals::Ptr{ExternalValues})::Cint)
    return nothing
end

# interval k's own 2z x 2z Hessian block of an integrator: H holds sums on the shared diagonal blocks, so the own part of
# the z_k diagonal is recovered by subtracting what interval k-1 put there.  Integrators whose blocks do not touch their
# z_{k+1} diagonal (all of the reference's) need no correction; a user integrator that does should provide this method.
function interval_hessian(integ, traj, μ, k, H)
    z = traj.dim
    return Matrix(H[slice(k, 1:2z, z), slice(k, 1:2z, z)])
end

# ---- MOI surface (evaluator.jl:291-456) -----------------------------------------------------------------------------
MOI.initialize(::GPUEvaluator, features) = nothing
MOI.features_available(ev::GPUEvaluator) = ev.eval_hessian ? [:Grad, :Jac, :Hess] : [:Grad, :Jac]
MOI.jacobian_structure(ev::GPUEvaluator) = ev.jacobian_structure
MOI.hessian_lagrangian_structure(ev::GPUEvaluator) = ev.hessian_structure

function MOI.eval_objective(ev::GPUEvaluator, Z::AbstractVector{Float64})
    stage_external!(ev, Z; obj_need = 0)
    f = Ref{Float64}(0.0)
    GC.@preserve Z check(ev, @ccall lib.dto_eval_objective(ev.handle::Ptr{Cvoid}, Z::Ptr{Float64}, f::Ptr{Float64})::Cint)
    return f[]
end

function MOI.eval_objective_gradient(ev::GPUEvaluator, ∇::AbstractVector{Float64}, Z::AbstractVector{Float64})
    stage_external!(ev, Z; obj_need = 1)
    GC.@preserve ∇ Z check(ev, @ccall lib.dto_eval_gradient(ev.handle::Ptr{Cvoid}, Z::Ptr{Float64}, ∇::Ptr{Float64})::Cint)
    return nothing
end

function MOI.eval_constraint(ev::GPUEvaluator, g::AbstractVector{Float64}, Z::AbstractVector{Float64})
    stage_external!(ev, Z; con_need = 0)
    GC.@preserve g Z check(ev, @ccall lib.dto_eval_constraint(ev.handle::Ptr{Cvoid}, Z::Ptr{Float64}, g::Ptr{Float64})::Cint)
    return nothing
end

function MOI.eval_constraint_jacobian(ev::GPUEvaluator, ∂::AbstractVector{Float64}, Z::AbstractVector{Float64})
    stage_external!(ev, Z; con_need = 1)
    GC.@preserve ∂ Z check(ev, @ccall lib.dto_eval_jacobian(ev.handle::Ptr{Cvoid}, Z::Ptr{Float64}, ∂::Ptr{Float64})::Cint)
    return nothing
end

function MOI.eval_hessian_lagrangian(ev::GPUEvaluator, H::AbstractVector{Float64}, Z::AbstractVector{Float64}, σ::Float64, μ::AbstractVector{Float64})
    stage_external!(ev, Z; con_need = 2, obj_need = σ != 0 ? 2 : -1, μ = μ)
    GC.@preserve H Z μ check(ev, @ccall lib.dto_eval_hessian(ev.handle::Ptr{Cvoid}, Z::Ptr{Float64}, σ::Float64, μ::Ptr{Float64}, H::Ptr{Float64})::Cint)
    return nothing
end

# y = J(Z) w and y = J(Z)' w (evaluator.jl:406-456): matrix-free on the device, no Jacobian crosses the bus
function MOI.eval_constraint_jacobian_product(ev::GPUEvaluator, y::AbstractVector{Float64}, Z::AbstractVector{Float64}, w::AbstractVector{Float64})
    stage_external!(ev, Z; con_need = 1)
    GC.@preserve y Z w check(ev, @ccall lib.dto_eval_jacobian_product(ev.handle::Ptr{Cvoid}, Z::Ptr{Float64}, w::Ptr{Float64}, y::Ptr{Float64})::Cint)
    return nothing
end

function MOI.eval_constraint_jacobian_transpose_product(ev::GPUEvaluator, y::AbstractVector{Float64}, Z::AbstractVector{Float64}, w::AbstractVector{Float64})
    stage_external!(ev, Z; con_need = 1)
    GC.@preserve y Z w check(ev, @ccall lib.dto_eval_jacobian_transpose_product(ev.handle::Ptr{Cvoid}, Z::Ptr{Float64}, w::Ptr{Float64}, y::Ptr{Float64})::Cint)
    return nothing
end

# row bounds handed to the solver (src/solvers/solve.jl:30-65)
function constraint_bounds(ev::GPUEvaluator)
    lo = Vector{Float64}(undef, ev.n_constraints)
    hi = Vector{Float64}(undef, ev.n_constraints)
    check(ev, @ccall lib.dto_constraint_bounds(ev.handle::Ptr{Cvoid}, lo::Ptr{Float64}, hi::Ptr{Float64})::Cint)
    return lo, hi
end

# ---- device-resident entry points (MadNLP GPU mode, src/solvers/madnlp_solver/options.jl:12-16) ---------------------------
# Pointers are DEVICE pointers (e.g. `Ptr{Float64}(UInt(pointer(A)))` of an AMDGPU.ROCArray{Float64}), `stream` the HIP
# stream of those arrays; outputs stay in HBM and the call returns with its last kernels in flight on `stream`.  Closure-based
# terms are evaluated on the host: pass the host copy of Z (and of mu) as `Z_host` / `μ_host` for problems that have any.
function eval_objective_dev!(ev::GPUEvaluator, df::Ptr{Float64}, dZ::Ptr{Float64}, stream::Ptr{Cvoid}; Z_host = nothing)
    Z_host === nothing || stage_external!(ev, Z_host; obj_need = 0)
    check(ev, @ccall lib.dto_eval_objective_dev(ev.handle::Ptr{Cvoid}, dZ::Ptr{Float64}, df::Ptr{Float64}, stream::Ptr{Cvoid})::Cint)
end
function eval_objective_gradient_dev!(ev::GPUEvaluator, d∇::Ptr{Float64}, dZ::Ptr{Float64}, stream::Ptr{Cvoid}; Z_host = nothing)
    Z_host === nothing || stage_external!(ev, Z_host; obj_need = 1)
    check(ev, @ccall lib.dto_eval_gradient_dev(ev.handle::Ptr{Cvoid}, dZ::Ptr{Float64}, d∇::Ptr{Float64}, stream::Ptr{Cvoid})::Cint)
end
function eval_constraint_dev!(ev::GPUEvaluator, dg::Ptr{Float64}, dZ::Ptr{Float64}, stream::Ptr{Cvoid}; Z_host = nothing)
    Z_host === nothing || stage_external!(ev, Z_host; con_need = 0)
    check(ev, @ccall lib.dto_eval_constraint_dev(ev.handle::Ptr{Cvoid}, dZ::Ptr{Float64}, dg::Ptr{Float64}, stream::Ptr{Cvoid})::Cint)
end
function eval_constraint_jacobian_dev!(ev::GPUEvaluator, d∂::Ptr{Float64}, dZ::Ptr{Float64}, stream::Ptr{Cvoid}; Z_host = nothing)
    Z_host === nothing || stage_external!(ev, Z_host; con_need = 1)
    check(ev, @ccall lib.dto_eval_jacobian_dev(ev.handle::Ptr{Cvoid}, dZ::Ptr{Float64}, d∂::Ptr{Float64}, stream::Ptr{Cvoid})::Cint)
end
function eval_hessian_lagrangian_dev!(ev::GPUEvaluator, dH::Ptr{Float64}, dZ::Ptr{Float64}, σ::Float64, dμ::Ptr{Float64}, stream::Ptr{Cvoid};
                                      Z_host = nothing, μ_host = nothing)
    Z_host === nothing || stage_external!(ev, Z_host; con_need = 2, obj_need = σ != 0 ? 2 : -1, μ = μ_host)
    check(ev, @ccall lib.dto_eval_hessian_dev(ev.handle::Ptr{Cvoid}, dZ::Ptr{Float64}, σ::Float64, dμ::Ptr{Float64}, dH::Ptr{Float64}, stream::Ptr{Cvoid})::Cint)
end

"Declare a device value vector (DTO_VECTOR_JACOBIAN / DTO_VECTOR_HESSIAN) that the `*_dev!` calls are handed every iteration: its call-invariant entries are then written once (include/dto_engine.h, bound outputs).  `C_NULL` unbinds."
bind_output_dev!(ev::GPUEvaluator, vector::Int32, dptr::Ptr{Float64}) =
    check(ev, @ccall lib.dto_bind_output_dev(ev.handle::Ptr{Cvoid}, vector::Int32, dptr::Ptr{Float64})::Cint)

# ---- multi-GPU: one Julia process per GPU, `k_lo` / `k_hi` at construction; the engine owns the RCCL communicator --------------
# (SURVEY.md section 8e; BASELINE configs[3]).  No callback needs a collective; these gather the per-rank value slabs for a
# consumer that wants the whole vector on every GPU, and sum the objective's per-shard partial sums.
"128 bytes from ncclGetUniqueId: call on ONE rank, hand them to the others (MPI.bcast, a file, a socket)."
function comm_unique_id()
    id = zeros(UInt8, DTO_COMM_ID_BYTES)
    rc = @ccall lib.dto_comm_unique_id(id::Ptr{Cvoid})::Cint
    rc == 0 || error(unsafe_string(@ccall lib.dto_last_error(C_NULL::Ptr{Cvoid})::Cstring))
    return id
end
"Collective over the ranks: ncclCommInitRank on this handle's device + exchange of the ranks' knot ranges (rank is 0-based)."
function comm_create!(ev::GPUEvaluator, id::Vector{UInt8}, rank::Integer, world::Integer)
    r32, w32 = Int32(rank), Int32(world)
    GC.@preserve id check(ev, @ccall lib.dto_comm_create(ev.handle::Ptr{Cvoid}, id::Ptr{Cvoid}, r32::Int32, w32::Int32)::Cint)
end
comm_destroy!(ev::GPUEvaluator) = check(ev, @ccall lib.dto_comm_destroy(ev.handle::Ptr{Cvoid})::Cint)
"How to allocate one value vector (DTO_VECTOR_*) so that ONE in-place ncclAllGather moves every rank's slab: `padded_len` doubles; the vector starts at `front_pad`, this rank's slab at `front_pad + own_lo` (0-based offsets)."
function gather_layout(ev::GPUEvaluator, vector::Int32)
    L = Ref(GatherLayout(0, 0, 0, 0, 0, Int32(0), Int32(0)))
    check(ev, @ccall lib.dto_get_gather_layout(ev.handle::Ptr{Cvoid}, vector::Int32, L::Ptr{GatherLayout})::Cint)
    return L[]
end
gather_jacobian_dev!(ev::GPUEvaluator, dbuf::Ptr{Float64}, stream::Ptr{Cvoid}) =
    check(ev, @ccall lib.dto_gather_jacobian_dev(ev.handle::Ptr{Cvoid}, dbuf::Ptr{Float64}, stream::Ptr{Cvoid})::Cint)
gather_hessian_dev!(ev::GPUEvaluator, dbuf::Ptr{Float64}, stream::Ptr{Cvoid}) =
    check(ev, @ccall lib.dto_gather_hessian_dev(ev.handle::Ptr{Cvoid}, dbuf::Ptr{Float64}, stream::Ptr{Cvoid})::Cint)
gather_gradient_dev!(ev::GPUEvaluator, dbuf::Ptr{Float64}, stream::Ptr{Cvoid}) =
    check(ev, @ccall lib.dto_gather_gradient_dev(ev.handle::Ptr{Cvoid}, dbuf::Ptr{Float64}, stream::Ptr{Cvoid})::Cint)
gather_constraint_dev!(ev::GPUEvaluator, dg_local::Ptr{Float64}, dg_full::Ptr{Float64}, stream::Ptr{Cvoid}) =
    check(ev, @ccall lib.dto_gather_constraint_dev(ev.handle::Ptr{Cvoid}, dg_local::Ptr{Float64}, dg_full::Ptr{Float64}, stream::Ptr{Cvoid})::Cint)
allreduce_objective_dev!(ev::GPUEvaluator, df::Ptr{Float64}, stream::Ptr{Cvoid}) =
    check(ev, @ccall lib.dto_allreduce_objective_dev(ev.handle::Ptr{Cvoid}, df::Ptr{Float64}, stream::Ptr{Cvoid})::Cint)

"""
Flop model of one `eval_constraint_jacobian` per interval (`dto_interval_costs`): squarings of the propagator chain and Taylor
terms of the sweep from the growth bound of every `Δt_k G(u_k)` (the work of the reference's `expv`, bilinear_integrator.jl:81,
grows with that norm as well).  `balanced_knot_ranges` turns it into contiguous knot ranges of equal cost for `world` ranks --
SURVEY.md section 8e: on a pulse whose amplitude varies along the trajectory equal knot counts leave the slowest rank setting the
step.  Ranges of unequal length are gathered in the broadcast form (`gather_layout` then reports `in_place_all_gather == 0`).
"""
function interval_costs(ev::GPUEvaluator, Z⃗::AbstractVector{Float64})
    K = ev.trajectory.N - 1
    cost = Vector{Float64}(undef, K)
    Zc = convert(Vector{Float64}, Z⃗)
    first0, cnt = Int64(0), Int64(K)
    GC.@preserve Zc cost check(ev, @ccall lib.dto_interval_costs(ev.handle::Ptr{Cvoid}, Zc::Ptr{Float64}, first0::Int64, cnt::Int64, cost::Ptr{Float64})::Cint)
    return cost
end
function balanced_knot_ranges(cost::AbstractVector{Float64}, world::Integer)
    N = length(cost) + 1
    c = vcat(cost, 0.0)                      # the last knot owns no interval
    pre = vcat(0.0, cumsum(c))
    function parts(limit)
        cuts = Tuple{Int,Int}[]; lo = 0
        for r in 1:world
            left = world - r
            hi = searchsortedlast(pre, pre[lo + 1] + limit) - 1
            hi = max(lo + 1, min(hi, N - left))
            r == world && (hi = N)
            push!(cuts, (lo + 1, hi)); lo = hi
        end
        return cuts, maximum(pre[b + 1] - pre[a] for (a, b) in cuts)
    end
    lo_l, hi_l = max(maximum(c), pre[end] / world), pre[end]
    best = parts(hi_l)
    for _ in 1:60
        mid = (lo_l + hi_l) / 2
        cand = parts(mid)
        if cand[2] <= mid * (1 + 1e-12)
            best, hi_l = cand, mid
        else
            lo_l = mid
        end
    end
    return best[1]                            # [(k_lo, k_hi)] 1-based inclusive, one per rank
end

function set_option!(ev::GPUEvaluator, name::AbstractString, value::Integer)
    v64 = Int64(value)
    check(ev, @ccall lib.dto_set_option(ev.handle::Ptr{Cvoid}, name::Cstring, v64::Int64)::Cint)
end

end # module
