"""``Evaluator`` -- mirror of the reference's ``Solvers.Evaluator <: MOI.AbstractNLPEvaluator``
(src/solvers/evaluator.jl:66-98, 291-456) on top of the C ABI.  Method names follow MOI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import library_path, load_library  # noqa: F401
from .problem import (BilinearIntegrator, CompositeObjective, DerivativeIntegrator, GlobalKnotPointObjective, HostIntegrator,
                      KnotPointObjective, TimeDependentBilinearIntegrator, LinearRegularizer, MinimumTimeObjective, NonlinearGlobalConstraint,
                      NonlinearKnotPointConstraint, NullObjective, QuadraticRegularizer)


class EngineError(RuntimeError):
    pass


def _dp(a):
    return a.ctypes.data_as(capi.c_double_p)


def _ip(a):
    return a.ctypes.data_as(capi.c_int64_p)


def _out(a, n, what):
    """An output buffer the engine writes `n` doubles into: it must be a C-contiguous float64 ndarray of exactly that
    size (the engine's device-to-host copy would otherwise run past it or land in the wrong bytes)."""
    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or not a.flags["C_CONTIGUOUS"] or not a.flags["WRITEABLE"]:
        raise ValueError(f"{what}: need a writeable C-contiguous float64 ndarray")
    if a.size != n:
        raise ValueError(f"{what}: length {a.size}, the engine writes {n} (shard-local slab lengths are in Evaluator.shard)")
    return _dp(a)


def _in(a, n, what):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.size != n:
        raise ValueError(f"{what}: length {a.size}, expected {n}")
    return a


def _flatten_objective(obj):
    if obj is None or isinstance(obj, NullObjective):
        return []
    if isinstance(obj, CompositeObjective):
        out = []
        for o, w in zip(obj.objectives, obj.weights):
            if isinstance(o, NullObjective):
                continue
            if isinstance(o, CompositeObjective):
                raise ValueError("nested CompositeObjective: `+` flattens them (_objectives.jl:165-176)")
            out.append((o, w))
        return out
    return [(obj, 1.0)]


class Evaluator:
    """Evaluator(prob; eval_hessian=true) -- evaluator.jl:99-288.

    Extra keyword arguments select the device and, for multi-GPU runs, the owned knot range
    ``k_lo..k_hi`` (1-based, inclusive).  Inputs ``Z`` and ``mu`` are always the GLOBAL vectors;
    value outputs are the shard-local slabs described by ``shard`` (whole vectors when unsharded)."""

    # options applied to every new handle (dto_set_option name -> value); the repository's tests switch "host_xfer_check" on here
    default_options = {}

    def __init__(self, prob, eval_hessian=True, device=0, k_lo=0, k_hi=0, verbose=False, general_path_only=False):
        self._lib = load_library()
        self._h = capi.H()
        traj = prob.trajectory
        self.trajectory = traj
        self.eval_hessian = bool(eval_hessian)
        keep = []  # keep numpy buffers alive across dto_create

        self._ext_int = []  # (integrator, first global row)
        integ = (capi.IntegratorDesc * max(1, len(prob.integrators)))()
        for i, it in enumerate(prob.integrators):
            if isinstance(it, BilinearIntegrator):
                # column-major n x n per generator = transpose of numpy's row-major
                G = np.ascontiguousarray(np.transpose(it.G, (0, 2, 1)))
                keep.append(G)
                integ[i] = capi.IntegratorDesc(capi.INTEGRATOR_BILINEAR, it.x_off, it.x_dim, it.u_off, it.u_dim, _dp(G))
            elif isinstance(it, DerivativeIntegrator):
                integ[i] = capi.IntegratorDesc(capi.INTEGRATOR_DERIVATIVE, it.x_off, it.x_dim, it.xdot_off, it.x_dim, None)
            elif isinstance(it, TimeDependentBilinearIntegrator) and it.family is not None:
                fam = it.family
                G = np.ascontiguousarray(np.transpose(fam.G, (0, 2, 1)))  # column-major per matrix
                kinds = np.ascontiguousarray([1 if k == "cos" else 2 for k, _, _ in fam.mods], dtype=np.int32)
                omegas = np.ascontiguousarray([w for _, w, _ in fam.mods], dtype=np.float64)
                Hs = (np.ascontiguousarray(np.stack([np.transpose(Hc, (0, 2, 1)) for _, _, Hc in fam.mods]))
                      if fam.mods else np.zeros(1))
                keep += [G, kinds, omegas, Hs]
                integ[i] = capi.IntegratorDesc(capi.INTEGRATOR_TIME_DEPENDENT_BILINEAR, it.x_off, it.x_dim, it.u_off, it.u_dim, _dp(G),
                                               it.t_off, it.spline_order, it.substeps, len(fam.mods),
                                               kinds.ctypes.data_as(capi.c_int32_p) if fam.mods else None,
                                               _dp(omegas) if fam.mods else None, _dp(Hs) if fam.mods else None)
            elif isinstance(it, HostIntegrator):
                integ[i] = capi.IntegratorDesc(capi.INTEGRATOR_EXTERNAL, 0, it.x_dim, 0, 0, None)
                self._ext_int.append((it, sum(p.dim for p in prob.integrators[:i])))
            else:
                raise NotImplementedError(f"{type(it).__name__}: wrap it in HostIntegrator to have it merged")

        self._ext_con, self._ext_obj, self._ext_keep = [], [], None
        terms = _flatten_objective(prob.objective)
        objs = (capi.ObjectiveDesc * max(1, len(terms)))()
        for i, (o, w) in enumerate(terms):
            d = capi.ObjectiveDesc()
            d.weight = w
            if isinstance(o, MinimumTimeObjective):
                d.kind, d.D = capi.OBJECTIVE_MINTIME, o.D
            elif isinstance(o, (QuadraticRegularizer, LinearRegularizer)):
                d.kind = capi.OBJECTIVE_QUADRATIC if isinstance(o, QuadraticRegularizer) else capi.OBJECTIVE_LINEAR
                d.comp_off, d.comp_dim = o.comp_off, o.comp_dim
                R = np.ascontiguousarray(o.R, dtype=np.float64)
                keep.append(R)
                d.R = _dp(R)
                if getattr(o, "baseline", None) is not None:
                    b = np.ascontiguousarray(o.baseline.T, dtype=np.float64)  # column-major comp_dim x N
                    keep.append(b)
                    d.baseline = _dp(b)
                if o.times is not None:
                    t = np.ascontiguousarray(o.times, dtype=np.int64)
                    keep.append(t)
                    d.times, d.n_times = _ip(t), t.size
            elif isinstance(o, GlobalKnotPointObjective):  # incl. GlobalObjective
                d.kind = capi.OBJECTIVE_EXTERNAL_GLOBAL
                comps = np.ascontiguousarray(o.comps, dtype=np.int32)
                gcomps = np.ascontiguousarray(o.gcomps, dtype=np.int32)
                t = np.ascontiguousarray(o.times, dtype=np.int64)
                keep += [comps, gcomps, t]
                if comps.size:
                    d.comps, d.n_comps = comps.ctypes.data_as(capi.c_int32_p), comps.size
                d.gcomps, d.n_gcomps = gcomps.ctypes.data_as(capi.c_int32_p), gcomps.size
                if t.size:
                    d.times, d.n_times = _ip(t), t.size
                self._ext_obj.append(o)
            elif isinstance(o, KnotPointObjective) and o.external:
                d.kind = capi.OBJECTIVE_EXTERNAL_KNOT
                comps = np.ascontiguousarray(o.comps, dtype=np.int32)
                t = np.ascontiguousarray(o.times, dtype=np.int64)
                keep += [comps, t]
                d.comps, d.n_comps = comps.ctypes.data_as(capi.c_int32_p), comps.size
                d.times, d.n_times = _ip(t), t.size
                self._ext_obj.append(o)
            elif isinstance(o, KnotPointObjective):
                d.kind = KnotPointObjective.KINDS[o.kind]
                comps = np.ascontiguousarray(o.comps, dtype=np.int32)
                t = np.ascontiguousarray(o.times, dtype=np.int64)
                Qs = np.ascontiguousarray(o.Qs, dtype=np.float64)
                keep += [comps, t, Qs]
                d.comps, d.n_comps = comps.ctypes.data_as(capi.c_int32_p), comps.size
                d.times, d.n_times = _ip(t), t.size
                d.Qs = _dp(Qs)
                if getattr(o, "A", None) is not None:
                    A = np.ascontiguousarray(o.A.T, dtype=np.float64)  # column-major k x n_comps
                    keep.append(A)
                    d.R, d.comp_dim = _dp(A), o.A.shape[0]
                if o.params is not None:
                    P = np.ascontiguousarray(o.params, dtype=np.float64)  # rows = times: column-major n_comps x n_times
                    keep.append(P)
                    d.params = _dp(P)
            else:
                raise NotImplementedError(f"{type(o).__name__} stays on the host (closure-based objective)")
            objs[i] = d

        nl = [c for c in prob.constraints if isinstance(c, (NonlinearKnotPointConstraint, NonlinearGlobalConstraint))]
        self._nl_constraints = nl
        cons = (capi.ConstraintDesc * max(1, len(nl)))()
        for i, c in enumerate(nl):
            if isinstance(c, NonlinearGlobalConstraint):
                gcomps = np.ascontiguousarray(c.gcomps, dtype=np.int32)
                _, jac0, hess0 = c.external_blocks(None, 2, mu=np.ones(c.g_dim), g=traj.global_data)  # patterns at Z0, mu = ones
                jac0, hess0 = np.ascontiguousarray(jac0), np.ascontiguousarray(hess0)
                keep += [gcomps, jac0, hess0]
                cons[i] = capi.ConstraintDesc(capi.CONSTRAINT_EXTERNAL_GLOBAL, int(c.equality), gcomps.size, c.g_dim,
                                              gcomps.ctypes.data_as(capi.c_int32_p), 0.0, None, 0, _dp(jac0), _dp(hess0))
                self._ext_con.append(c)
                continue
            comps = np.ascontiguousarray(c.comps, dtype=np.int32)
            t = np.ascontiguousarray(c.times, dtype=np.int64)
            keep += [comps, t]
            if c.external:
                # pattern = the closure's Jacobian at Z0, as the reference takes it (evaluator.jl:136)
                Zk0 = traj.vec()[:traj.dim * traj.N].reshape(traj.N, traj.dim)
                jac0 = np.ascontiguousarray(c.external_blocks(Zk0, 1)[1])
                keep.append(jac0)
                cons[i] = capi.ConstraintDesc(capi.CONSTRAINT_EXTERNAL, int(c.equality), comps.size, c.g_dim,
                                              comps.ctypes.data_as(capi.c_int32_p), 0.0, _ip(t), t.size, _dp(jac0), None)
                self._ext_con.append(c)
                continue
            cons[i] = capi.ConstraintDesc(NonlinearKnotPointConstraint.KINDS[c.kind], int(c.equality), comps.size, 1,
                                          comps.ctypes.data_as(capi.c_int32_p), c.c, _ip(t), t.size, None, None)

        Z0 = np.ascontiguousarray(traj.vec(), dtype=np.float64)
        desc = capi.ProblemDesc(capi.DTO_ABI_VERSION, device, traj.N, traj.dim, traj.global_dim,
                                traj.components[traj.timestep][0], int(eval_hessian), len(prob.integrators),
                                len(terms), len(nl), capi.FLAG_GENERAL_PATH_ONLY if general_path_only else 0, integ, objs, cons,
                                _dp(Z0), k_lo, k_hi)
        if self._lib.dto_create(C.byref(desc), C.byref(self._h)) != 0:
            raise EngineError(self._lib.dto_last_error(None).decode())
        v = C.c_int64()
        self._lib.dto_num_vars(self._h, C.byref(v)); self.n_variables = v.value
        self._lib.dto_num_cons(self._h, C.byref(v)); self.n_constraints = v.value
        self._lib.dto_num_dynamics_cons(self._h, C.byref(v)); self.n_dynamics_constraints = v.value
        self.n_nonlinear_constraints = self.n_constraints - self.n_dynamics_constraints
        self._lib.dto_jac_nnz(self._h, C.byref(v)); self.n_jacobian_entries = v.value
        self._lib.dto_hess_nnz(self._h, C.byref(v)); self.n_hessian_entries = v.value
        self.shard = capi.ShardInfo()
        self._lib.dto_get_shard_info(self._h, C.byref(self.shard))
        for name, value in self.default_options.items():
            self.set_option(name, value)

    # ---- plumbing
    def _check(self, rc):
        if rc != 0:
            raise EngineError(self._lib.dto_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.dto_destroy(self._h)
            self._h = capi.H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # ---- MOI surface (host vectors)
    def initialize(self, features=None):  # MOI.initialize, evaluator.jl:291
        return None

    def features_available(self):  # evaluator.jl:293-299
        return ["Grad", "Jac", "Hess"] if self.eval_hessian else ["Grad", "Jac"]

    def _Z(self, Z):
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        if Z.size != self.n_variables:
            raise ValueError("Z has the wrong length")
        return Z

    def _stage_external(self, Z, con_need=-1, obj_need=-1, mu=None):
        """Evaluate the closure-based knot terms on the host (what the Julia shim does with the reference's
        ForwardDiff code) and hand the blocks to the engine: need 0 = values, 1 = + first derivatives,
        2 = + second derivatives; -1 = this callback does not read that family."""
        if not (self._ext_int or self._ext_con or self._ext_obj):
            return
        if Z is None:
            raise EngineError("closure-based terms need the host copy of Z (pass Z_host to the *_dev call)")
        traj = self.trajectory
        Zk = np.asarray(Z, dtype=np.float64)[:traj.dim * traj.N].reshape(traj.N, traj.dim)
        gdat = np.asarray(Z, dtype=np.float64)[traj.dim * traj.N:]
        ni = len(self._ext_int)
        vals = (capi.ExternalValues * (ni + len(self._ext_con) + len(self._ext_obj)))()
        keep = []
        for i, (it, row0) in enumerate(self._ext_int):  # integrators ride with the constraint-side callbacks
            if con_need < 0:
                continue
            m = None if mu is None else np.asarray(mu, dtype=np.float64)[row0:row0 + it.dim]
            for name, b in zip(("values", "first", "second"), it.external_blocks(Zk, con_need, m)):
                if b is not None:
                    b = np.ascontiguousarray(b, dtype=np.float64)
                    keep.append(b)
                    setattr(vals[i], name, _dp(b))
        row = self.n_dynamics_constraints
        rows = {}
        for c in self._nl_constraints:
            rows[id(c)] = row
            row += c.dim
        for i, c in enumerate(self._ext_con):
            if con_need < 0:
                continue
            m = None if mu is None else np.asarray(mu, dtype=np.float64)[rows[id(c)]:rows[id(c)] + c.dim]
            blocks = (c.external_blocks(Zk, con_need, m, g=gdat) if isinstance(c, NonlinearGlobalConstraint)
                      else c.external_blocks(Zk, con_need, m))
            for name, b in zip(("values", "first", "second"), blocks):
                if b is not None:
                    b = np.ascontiguousarray(b, dtype=np.float64)
                    keep.append(b)
                    setattr(vals[ni + i], name, _dp(b))
        for j, o in enumerate(self._ext_obj):
            if obj_need < 0:
                continue
            blocks = (o.external_blocks(Zk, obj_need, g=gdat) if isinstance(o, GlobalKnotPointObjective)
                      else o.external_blocks(Zk, obj_need))
            for name, b in zip(("values", "first", "second"), blocks):
                if b is not None:
                    b = np.ascontiguousarray(b, dtype=np.float64)
                    keep.append(b)
                    setattr(vals[ni + len(self._ext_con) + j], name, _dp(b))
        self._ext_keep = (vals, keep)  # the engine reads these host buffers during the next callbacks
        self._check(self._lib.dto_set_external(self._h, len(vals), vals))

    def eval_objective(self, Z):  # evaluator.jl:304
        Z = self._Z(Z)
        f = C.c_double()
        self._stage_external(Z, obj_need=0)
        self._check(self._lib.dto_eval_objective(self._h, _dp(Z), C.byref(f)))
        return f.value

    def eval_objective_gradient(self, grad, Z):  # evaluator.jl:310
        Z = self._Z(Z)
        self._stage_external(Z, obj_need=1)
        self._check(self._lib.dto_eval_gradient(self._h, _dp(Z), _out(grad, self.shard.grad_len, "grad")))

    def eval_constraint(self, g, Z):  # evaluator.jl:323
        Z = self._Z(Z)
        self._stage_external(Z, con_need=0)
        self._check(self._lib.dto_eval_constraint(self._h, _dp(Z), _out(g, self.shard.cons_len, "g")))

    def jacobian_structure(self, first=0, count=None):  # evaluator.jl:364 (1-based pairs)
        count = self.n_jacobian_entries - first if count is None else count
        r = np.empty(count, dtype=np.int64)
        c = np.empty(count, dtype=np.int64)
        self._check(self._lib.dto_jacobian_structure(self._h, first, count, _ip(r), _ip(c)))
        return r, c

    def eval_constraint_jacobian(self, vals, Z):  # evaluator.jl:368
        Z = self._Z(Z)
        self._stage_external(Z, con_need=1)
        self._check(self._lib.dto_eval_jacobian(self._h, _dp(Z), _out(vals, self.shard.jac_len, "vals")))

    def hessian_lagrangian_structure(self, first=0, count=None):  # evaluator.jl:385
        count = self.n_hessian_entries - first if count is None else count
        r = np.empty(count, dtype=np.int64)
        c = np.empty(count, dtype=np.int64)
        self._check(self._lib.dto_hessian_structure(self._h, first, count, _ip(r), _ip(c)))
        return r, c

    def eval_hessian_lagrangian(self, H, Z, sigma, mu):  # evaluator.jl:389
        Z = self._Z(Z)
        mu = _in(mu, self.n_constraints, "mu")
        self._stage_external(Z, con_need=2, obj_need=2 if sigma != 0.0 else -1, mu=mu)
        self._check(self._lib.dto_eval_hessian(self._h, _dp(Z), float(sigma), _dp(mu), _out(H, self.shard.hess_len, "H")))

    def eval_constraint_jacobian_product(self, y, Z, w):  # evaluator.jl:406 (y = J w)
        Z = self._Z(Z)
        w = _in(w, self.n_variables, "w")
        self._stage_external(Z, con_need=1)
        self._check(self._lib.dto_eval_jacobian_product(self._h, _dp(Z), _dp(w), _out(y, self.n_constraints, "y")))

    def eval_constraint_jacobian_transpose_product(self, y, Z, w):  # evaluator.jl:432 (y = J' w)
        Z = self._Z(Z)
        w = _in(w, self.n_constraints, "w")
        self._stage_external(Z, con_need=1)
        self._check(self._lib.dto_eval_jacobian_transpose_product(self._h, _dp(Z), _dp(w), _out(y, self.n_variables, "y")))

    def constraint_bounds(self):  # get_nonlinear_constraints, src/solvers/solve.jl:30-65
        lo = np.empty(self.n_constraints)
        hi = np.empty(self.n_constraints)
        self._check(self._lib.dto_constraint_bounds(self._h, _dp(lo), _dp(hi)))
        return lo, hi

    def shard_rows(self):
        n = self.shard.n_row_segments
        s = np.empty(n, dtype=np.int64)
        ln = np.empty(n, dtype=np.int64)
        self._check(self._lib.dto_shard_rows(self._h, _ip(s), _ip(ln)))
        return s, ln

    # ---- device-resident forms (pointers are integers, e.g. torch.Tensor.data_ptr())
    # (closure-based terms are evaluated on the host: pass the host copy of Z -- and of mu -- as well)
    def eval_objective_dev(self, dZ, df, stream=0, Z_host=None):
        self._stage_external(Z_host, obj_need=0)
        self._check(self._lib.dto_eval_objective_dev(self._h, dZ, df, stream))

    def eval_gradient_dev(self, dZ, dgrad, stream=0, Z_host=None):
        self._stage_external(Z_host, obj_need=1)
        self._check(self._lib.dto_eval_gradient_dev(self._h, dZ, dgrad, stream))

    def eval_constraint_dev(self, dZ, dg, stream=0, Z_host=None):
        self._stage_external(Z_host, con_need=0)
        self._check(self._lib.dto_eval_constraint_dev(self._h, dZ, dg, stream))

    def eval_jacobian_dev(self, dZ, dvals, stream=0, Z_host=None):
        self._stage_external(Z_host, con_need=1)
        self._check(self._lib.dto_eval_jacobian_dev(self._h, dZ, dvals, stream))

    def eval_hessian_dev(self, dZ, sigma, dmu, dvals, stream=0, Z_host=None, mu_host=None):
        self._stage_external(Z_host, con_need=2, obj_need=2 if sigma != 0.0 else -1, mu=mu_host)
        self._check(self._lib.dto_eval_hessian_dev(self._h, dZ, float(sigma), dmu, dvals, stream))

    # ---- multi-GPU: the engine's own collectives (RCCL over xGMI behind the C ABI; include/dto_engine.h, "Multi-GPU")
    @staticmethod
    def comm_unique_id():
        """128 bytes from ncclGetUniqueId: call on ONE rank and hand them to the others (e.g. torch.distributed's
        broadcast_object_list, MPI, a file)."""
        lib = load_library()
        buf = C.create_string_buffer(capi.COMM_ID_BYTES)
        if lib.dto_comm_unique_id(buf) != 0:
            raise EngineError(lib.dto_last_error(None).decode())
        return buf.raw

    def comm_create(self, unique_id, rank, world):
        """Collective over the ranks: ncclCommInitRank on this handle's device + exchange of the ranks' knot ranges."""
        buf = C.create_string_buffer(bytes(unique_id), capi.COMM_ID_BYTES)
        self._check(self._lib.dto_comm_create(self._h, buf, rank, world))

    def comm_set_ranges(self, rank, ranges):
        """The layout bookkeeping without a communicator (works on structure-only handles): ranges = [(k_lo, k_hi)] per rank."""
        lo = np.ascontiguousarray([a for a, _ in ranges], dtype=np.int64)
        hi = np.ascontiguousarray([b for _, b in ranges], dtype=np.int64)
        self._check(self._lib.dto_comm_set_ranges(self._h, rank, len(ranges), _ip(lo), _ip(hi)))

    def comm_destroy(self):
        self._check(self._lib.dto_comm_destroy(self._h))

    def gather_layout(self, vector):
        """dto_get_gather_layout: how to allocate one value vector (capi.VECTOR_*) so that the gather moves every slab in place."""
        L = capi.GatherLayout()
        self._check(self._lib.dto_get_gather_layout(self._h, vector, C.byref(L)))
        return L

    def gather_slabs(self, vector):
        """[(lo, len)] of every rank's slab of one value vector."""
        w = self.gather_layout(vector).world
        lo, ln = np.empty(w, dtype=np.int64), np.empty(w, dtype=np.int64)
        self._check(self._lib.dto_gather_slabs(self._h, vector, _ip(lo), _ip(ln)))
        return list(zip(lo.tolist(), ln.tolist()))

    def gather_dev(self, vector, dbuf, stream=0):
        """All ranks: fill in the other ranks' slabs of the vector allocated per gather_layout (dbuf = the allocation)."""
        fn = {capi.VECTOR_JACOBIAN: self._lib.dto_gather_jacobian_dev, capi.VECTOR_HESSIAN: self._lib.dto_gather_hessian_dev,
              capi.VECTOR_GRADIENT: self._lib.dto_gather_gradient_dev}[vector]
        self._check(fn(self._h, dbuf, stream))

    def gather_constraint_dev(self, dg_local, dg_full, stream=0):
        self._check(self._lib.dto_gather_constraint_dev(self._h, dg_local, dg_full, stream))

    def allreduce_objective_dev(self, df, stream=0):
        self._check(self._lib.dto_allreduce_objective_dev(self._h, df, stream))

    def bind_output_dev(self, vector, dptr):
        """dto_bind_output_dev: declare a device value vector (capi.VECTOR_JACOBIAN / VECTOR_HESSIAN) that later `*_dev` calls are
        handed again and again; its call-invariant entries are then written once.  dptr = 0 unbinds."""
        self._check(self._lib.dto_bind_output_dev(self._h, vector, dptr))

    def set_option(self, name, value):
        """dto_set_option: ``reuse_forward_sweep`` (solver loops evaluate g, J, H at the same point), ``expm_form``
        (0 = by cost, 2 / 3 = two- / three-product form of the Jacobian's matrix exponential)."""
        self._check(self._lib.dto_set_option(self._h, name.encode(), int(value)))

    # ---- measurement
    def profile_enable(self, on=True):
        self._check(self._lib.dto_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._check(self._lib.dto_profile_reset(self._h))

    def interval_costs(self, Z, first=0, count=None):
        """Flop model of one eval_constraint_jacobian per interval (dto_interval_costs): input of cost-balanced sharding."""
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        K = self.trajectory.N - 1
        count = K - first if count is None else count
        out = np.empty(count)
        self._check(self._lib.dto_interval_costs(self._h, Z.ctypes.data_as(capi.c_double_p), first, count, out.ctypes.data_as(capi.c_double_p)))
        return out

    def profile_get(self, name):
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        self._check(self._lib.dto_profile_get(self._h, name.encode(), C.byref(ms), C.byref(n), C.byref(fl)))
        return ms.value, n.value, fl.value

    def last_stats(self):
        a, b = C.c_int32(), C.c_int32()
        self._check(self._lib.dto_last_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value
