"""Test helpers: convert an oracle `Problem` (oracle/dto_oracle.py) into the engine's host mirror."""
import numpy as np

import dto_amd
import dto_oracle as O


def to_engine(prob: O.Problem, closure_derivatives="numeric", tdb_on_device=True):
    """Build NamedTrajectory + DirectTrajOptProblem (host mirror) describing the same problem."""
    N, z = prob.N, prob.z
    data = prob.Z0[:z * N].reshape(N, z).T
    # one trajectory component per distinct range mentioned by the problem; leftovers become filler
    ranges = {}
    for it in prob.integrators:
        if isinstance(it, O.TimeDependentBilinearIntegrator):
            ranges[(it.x_off, it.x_dim)] = None
            ranges[(it.u_off, it.u_dim)] = None
            ranges[(it.t_off, 1)] = None
            continue
        if it.kind == "external":
            continue
        ranges[(it.x_off, it.x_dim)] = None
        if it.kind == "bilinear":
            if it.u_dim:
                ranges[(it.u_off, it.u_dim)] = None
        else:
            ranges[(it.xdot_off, it.x_dim)] = None
    for t in prob.objectives:
        if t.kind not in ("mintime", "knot_sqdist", "knot_closure", "knot_lowrank", "global_closure"):
            ranges[(t.comp_off, t.comp_dim)] = None
    ranges[(prob.dt_idx, 1)] = None
    cuts = sorted(ranges)
    comps, pos, names = {}, 0, {}
    for off, d in cuts:
        assert off >= pos, "overlapping component ranges are not representable"
        if off > pos:
            comps[f"_f{pos}"] = data[pos:off]
        names[(off, d)] = f"c{off}"
        comps[f"c{off}"] = data[off:off + d]
        pos = off + d
    if pos < z:
        comps[f"_f{pos}"] = data[pos:z]
    gdata = prob.Z0[z * N:] if prob.gd else None
    traj = dto_amd.NamedTrajectory(comps, timestep=names[(prob.dt_idx, 1)], global_data=gdata)
    assert traj.dim == z
    integ = []
    for it in prob.integrators:
        if isinstance(it, O.TimeDependentBilinearIntegrator):
            # the DEVICE integrator (csrc/dto_tdb.hip) for the same generator family; `tdb_on_device=False` routes the same
            # closure through the host-evaluated merge path instead
            fam = dto_amd.ModulatedGenerators(it.G, it.mods)
            integ.append(dto_amd.TimeDependentBilinearIntegrator(fam, names[(it.x_off, it.x_dim)], names[(it.u_off, it.u_dim)],
                                                                 names[(it.t_off, 1)], traj, spline_order=it.spline_order,
                                                                 substeps=it.substeps, on_device=tdb_on_device))
            continue
        if it.kind == "external":
            analytic = closure_derivatives == "analytic"
            integ.append(dto_amd.HostIntegrator(it.f, it.x_dim, traj, jac=it.jac if analytic else None,
                                                hess=it.hess if analytic else None))
            continue
        if it.kind == "bilinear":
            if it.u_dim == 0:
                raise NotImplementedError
            integ.append(dto_amd.BilinearIntegrator(it.G, names[(it.x_off, it.x_dim)], names[(it.u_off, it.u_dim)], traj))
        else:
            integ.append(dto_amd.DerivativeIntegrator(names[(it.x_off, it.x_dim)], names[(it.xdot_off, it.x_dim)], traj))
    obj = dto_amd.NullObjective()
    terms = []
    for i, t in enumerate(prob.objectives):
        if t.kind == "quadratic":
            o = dto_amd.QuadraticRegularizer(names[(t.comp_off, t.comp_dim)], traj, t.R, baseline=t.baseline, times=t.times1)
        elif t.kind == "linear":
            o = dto_amd.LinearRegularizer(names[(t.comp_off, t.comp_dim)], traj, t.R, times=t.times1)
        elif t.kind == "knot_sqdist":
            o = dto_amd.KnotPointObjective.__new__(dto_amd.KnotPointObjective)
            o.kind, o.var_names, o.external, o.A = "sqdist", [], False, None
            o.times = np.asarray(t.times1, dtype=np.int64)
            o.comps = np.asarray(t.comps, dtype=np.int32)
            o.Qs = np.asarray(t.Qs, dtype=np.float64)
            o.params = None if t.params is None else np.asarray(t.params, dtype=np.float64)
        elif t.kind == "global_closure":
            analytic = closure_derivatives == "analytic"
            o = dto_amd.GlobalKnotPointObjective.__new__(dto_amd.GlobalKnotPointObjective)
            o.l, o.grad, o.hess = t.l, (t.grad if analytic else None), (t.hess if analytic else None)
            o.var_names, o.global_names = [], []
            o.times = np.asarray(t.times1, dtype=np.int64)
            o.comps = np.asarray(t.comps, dtype=np.int32)
            o.gcomps = np.asarray(t.gcomps, dtype=np.int32)
            o.Qs = np.asarray(t.Qs, dtype=np.float64)
            nl = max(1, o.times.size)
            o.params = [None] * nl if t.params is None else list(t.params)
            if o.times.size == 0:  # GlobalObjective: the global variables alone
                o._listings = (lambda Zk, g, o=o: [g[o.gcomps]])
        elif t.kind == "knot_lowrank":
            o = dto_amd.KnotPointObjective.__new__(dto_amd.KnotPointObjective)
            o.kind, o.var_names, o.external = "lowrank_infidelity", [], False
            o.times = np.asarray(t.times1, dtype=np.int64)
            o.comps = np.asarray(t.comps, dtype=np.int32)
            o.Qs = np.asarray(t.Qs, dtype=np.float64)
            o.params, o.A = None, np.asarray(t.A, dtype=np.float64)
        elif t.kind == "knot_closure":
            # the host mirror differentiates the closure itself (complex step / differences) unless the test
            # asks for the analytic derivatives to be handed through (closure_derivatives="analytic")
            o = dto_amd.KnotPointObjective.__new__(dto_amd.KnotPointObjective)
            o.kind, o.var_names, o.external = t.l, [], True
            analytic = closure_derivatives == "analytic"
            o.l, o.grad, o.hess = t.l, (t.grad if analytic else None), (t.hess if analytic else None)
            o.times = np.asarray(t.times1, dtype=np.int64)
            o.comps = np.asarray(t.comps, dtype=np.int32)
            o.Qs = np.asarray(t.Qs, dtype=np.float64)
            o.params = [None] * o.times.size if t.params is None else list(t.params)
        else:
            o = dto_amd.MinimumTimeObjective(traj, D=t.D)
        terms.append(prob.w(i) * o)
    if terms:
        obj = terms[0]
        for t in terms[1:]:
            obj = obj + t
    cons = []
    for c in prob.constraints:
        # component indices are passed through a synthetic single name when they form one range
        if c.kind == "global_closure":
            analytic = closure_derivatives == "analytic"
            k = dto_amd.NonlinearGlobalConstraint.__new__(dto_amd.NonlinearGlobalConstraint)
            k.g, k.jac, k.hess = c.g, (c.jac if analytic else None), (c.hess if analytic else None)
            k.equality, k.global_names = bool(c.equality), []
            k.gcomps = np.asarray(c.gcomps, dtype=np.int32)
            k.global_dim, k.g_dim, k.dim = k.gcomps.size, c.g_dim, c.g_dim
            cons.append(k)
            continue
        k = dto_amd.NonlinearKnotPointConstraint.__new__(dto_amd.NonlinearKnotPointConstraint)
        if c.kind == "closure":
            analytic = closure_derivatives == "analytic"
            k.kind, k.var_names, k.c, k.equality, k.external = c.g, [], 0.0, bool(c.equality), True
            k.g, k.jac, k.hess = c.g, (c.jac if analytic else None), (c.hess if analytic else None)
            k.times = np.asarray(c.times1, dtype=np.int64)
            k.comps = np.asarray(c.comps, dtype=np.int32)
            k.params = [None] * k.times.size if c.params is None else list(c.params)
            k.g_dim, k.var_dim = c.g_dim, k.comps.size
            k.dim = k.g_dim * k.times.size
            cons.append(k)
            continue
        k.external = False
        k.kind, k.var_names, k.c, k.equality = c.kind, [], float(c.c), bool(c.equality)
        k.times = np.asarray(c.times1, dtype=np.int64)
        k.comps = np.asarray(c.comps, dtype=np.int32)
        k.g_dim, k.var_dim = 1, k.comps.size
        k.dim = k.times.size
        cons.append(k)
    return dto_amd.DirectTrajOptProblem(traj, obj, integ, constraints=cons)


def rel_err(a, b):
    """max |a-b| / max(1,|b|) elementwise (the tolerance form of SURVEY.md §8c)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def run_all(ev, prob_o, Z, mu, sigma=1.0, hessian=True):
    """All MOI callbacks through the engine (host-pointer C ABI)."""
    out = {}
    out["f"] = ev.eval_objective(Z)
    g = np.full(ev.shard.grad_len, np.nan); ev.eval_objective_gradient(g, Z); out["grad"] = g
    c = np.full(ev.shard.cons_len, np.nan); ev.eval_constraint(c, Z); out["cons"] = c
    j = np.full(ev.shard.jac_len, np.nan); ev.eval_constraint_jacobian(j, Z); out["jac"] = j
    if hessian:
        h = np.full(ev.shard.hess_len, np.nan); ev.eval_hessian_lagrangian(h, Z, sigma, mu); out["hess"] = h
    return out
