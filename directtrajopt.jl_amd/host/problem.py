"""Host-side mirrors of the reference's hot-path plugin types (same names and argument meaning).

Only the DESCRIPTION lives here; all arithmetic of the built-in kinds happens in the HIP engine.
Closure-based knot terms (a Python callable as the loss / g) are evaluated on the host, as the Julia shim
does with the reference's own ForwardDiff code, and merged by the engine at its precomputed offsets
(SURVEY.md §8f rank 2); TimeDependentBilinearIntegrator and the Global* kinds raise."""
from __future__ import annotations

import numpy as np


class BilinearIntegrator:
    """BilinearIntegrator(G, x, u, traj) -- src/integrators/bilinear_integrator.jl:61-85.

    ``G`` is either the reference's closure ``u -> matrix`` (must be affine in u: it is probed at 0
    and the unit vectors and checked at a random point, because closures cannot cross the C ABI) or
    an array of shape (m+1, n, n) holding [G(0), G(e_1)-G(0), ...]."""

    def __init__(self, G, x, u, traj):
        self.x_name, self.u_name = x, u
        self.x_dim = traj.dims[x]
        self.u_dim = traj.dims[u]
        self.var_dim = 2 * self.x_dim + self.u_dim + 1
        self.dim = self.x_dim * (traj.N - 1)
        self.x_off = traj.components[x][0]
        self.u_off = traj.components[u][0]
        n, m = self.x_dim, self.u_dim
        if callable(G):
            G0 = np.asarray(G(np.zeros(m)), dtype=np.float64)
            drives = [np.asarray(G(np.eye(m)[j]), dtype=np.float64) - G0 for j in range(m)]
            probe = np.random.default_rng(0).standard_normal(m)
            want = G0 + sum(probe[j] * drives[j] for j in range(m))
            got = np.asarray(G(probe), dtype=np.float64)
            if not np.allclose(got, want, rtol=1e-12, atol=1e-12 * max(1.0, np.abs(want).max())):
                raise ValueError("BilinearIntegrator: G(u) is not affine in u; only bilinear dynamics run on the device")
            self.G = np.stack([G0] + drives)
        else:
            self.G = np.asarray(G, dtype=np.float64)
        if self.G.shape != (m + 1, n, n):
            raise ValueError(f"BilinearIntegrator: generators must have shape {(m + 1, n, n)}")


class DerivativeIntegrator:
    """DerivativeIntegrator(x, xdot, traj) -- src/integrators/derivative_integrator.jl:26-49."""

    def __init__(self, x, xdot, traj):
        self.x_name, self.xdot_name = x, xdot
        self.x_dim = traj.dims[x]
        if traj.dims[xdot] != self.x_dim:
            raise ValueError("DerivativeIntegrator: x and its derivative must have equal dimension")
        self.var_dim = 3 * self.x_dim + 1
        self.dim = self.x_dim * (traj.N - 1)
        self.x_off = traj.components[x][0]
        self.xdot_off = traj.components[xdot][0]


class HostIntegrator:
    """Any other AbstractIntegrator (src/integrators/_integrators.jl:22-34), evaluated on the host and merged by the
    engine (DTO_INTEGRATOR_EXTERNAL).  ``f(zz, k) -> residual`` of the stacked knot pair ``zz = [z_k; z_{k+1}]`` for
    interval ``k`` (0-based), the form the reference differentiates (time_dependent_bilinear_integrator.jl:181-203);
    ``jac(zz, k) -> (x_dim, 2z)`` and ``hess(zz, k, mu) -> (2z, 2z)`` are optional (else host/closures.py)."""

    external = True

    def __init__(self, f, x_dim, traj, jac=None, hess=None):
        self.f, self.jac, self.hess = f, jac, hess
        self.x_dim = int(x_dim)
        self.dim = self.x_dim * (traj.N - 1)
        self.var_dim = 2 * traj.dim

    def external_blocks(self, Zk, need, mu=None):
        """values [K, d]; Jacobian blocks [K, 2z, d] (= column-major d x 2z); Hessian blocks of mu_k' f [K, 2z, 2z]."""
        from . import closures
        N, z = Zk.shape
        K, d = N - 1, self.x_dim
        vals = np.zeros((K, d))
        first = np.zeros((K, 2 * z, d)) if need >= 1 else None
        second = np.zeros((K, 2 * z, 2 * z)) if need >= 2 else None
        for k in range(K):
            zz = np.concatenate([Zk[k], Zk[k + 1]])
            vals[k] = np.asarray(self.f(zz, k), dtype=np.float64).reshape(d)
            if need >= 1:
                first[k] = closures.jacobian(lambda x, p: self.f(x, k), zz, None, d,
                                             None if self.jac is None else (lambda x, p: self.jac(x, k))).T
            if need >= 2:
                m = np.asarray(mu[k * d:(k + 1) * d], dtype=np.float64)
                Hm = closures.hessian(lambda x: m @ np.asarray(self.f(x, k)).reshape(d), zz,
                                      None if self.jac is None else (lambda x: m @ np.asarray(self.jac(x, k)).reshape(d, 2 * z)),
                                      None if self.hess is None else (lambda x: self.hess(x, k, m)))
                second[k] = Hm.T
        return vals, first, second


class ModulatedGenerators:
    """The parametrised generator family the engine integrates ON THE DEVICE (DTO_INTEGRATOR_TIME_DEPENDENT_BILINEAR):

        G(u, t) = sum_{j=0..m} ubar_j ( G_j + sum_c phi_c(t) H_cj ),   ubar_0 = 1,  phi_c = cos(omega_c t) or sin(omega_c t)

    ``G``: (m+1, n, n); ``mods``: sequence of (kind, omega, H) with kind "cos" / "sin" and H of shape (m+1, n, n).  The object
    is also the closure ``G(u, t)`` the reference's constructor takes, so the same problem description runs through the host
    path (for comparison) and the device path."""

    def __init__(self, G, mods=()):
        self.G = np.asarray(G, dtype=np.float64)
        self.mods = [(str(k), float(w), np.asarray(Hc, dtype=np.float64)) for k, w, Hc in mods]
        for k, _, Hc in self.mods:
            if k not in ("cos", "sin") or Hc.shape != self.G.shape:
                raise ValueError("modulation terms are (\"cos\" | \"sin\", omega, H) with H shaped like G")

    def __call__(self, u, t):
        ub = np.concatenate([[1.0], np.asarray(u)])
        M = np.tensordot(ub, self.G, axes=(0, 0))
        for k, w, Hc in self.mods:
            M = M + (np.cos(w * t) if k == "cos" else np.sin(w * t)) * np.tensordot(ub, Hc, axes=(0, 0))
        return M


class TimeDependentBilinearIntegrator(HostIntegrator):
    """TimeDependentBilinearIntegrator(G, x, u, t, traj; spline_order=1) --
    src/integrators/time_dependent_bilinear_integrator.jl:60-140: defect x_{k+1} - Phi_k x_k for
    dx/dtau = dt_k G(u(tau), t_k + tau dt_k) x on tau in [0, 1], controls held (order 0) or linearly interpolated to
    u_{k+1} (order 1).  ``G(u, t)`` is an arbitrary closure, so this integrator is host-evaluated and merged.  Where the
    reference integrates with adaptive Tsit5 (default tolerances) and differentiates through the solver, this mirror
    uses fixed-step RK4 (``substeps`` per interval; an analytic map, so complex-step derivatives are exact for it).

    With ``G`` a ``ModulatedGenerators`` (and ``on_device`` left on) the engine integrates on the GPU -- the same RK4 scheme
    with its exact first and second derivatives (csrc/dto_tdb.hip) -- and nothing of this class's arithmetic runs."""

    def __init__(self, G, x, u, t, traj, spline_order=1, substeps=32, on_device=True):
        if spline_order not in (0, 1):
            raise ValueError(f"Unsupported spline order: {spline_order}")
        self.family = G if (on_device and isinstance(G, ModulatedGenerators)) else None
        self.x_off, self.u_off, self.t_off = traj.components[x][0], traj.components[u][0], traj.components[t][0]
        self.G, self.spline_order, self.substeps = G, int(spline_order), int(substeps)
        self.x_name, self.u_name, self.t_name = x, u, t
        self.u_dim = traj.dims[u]
        z = traj.dim
        xc, uc = np.asarray(traj.components[x]), np.asarray(traj.components[u])
        tc, dtc = traj.components[t][0], traj.components[traj.timestep][0]

        def f(zz, k):
            xk, uk, tk, dt = zz[xc], zz[uc], zz[tc], zz[dtc]
            xk1, uk1 = zz[z + xc], zz[z + uc]
            ctrl = (lambda tau: uk) if self.spline_order == 0 else (lambda tau: uk + tau * (uk1 - uk))
            rhs = lambda tau, y: dt * (np.asarray(self.G(ctrl(tau), tk + tau * dt)) @ y)
            y, h = xk, 1.0 / self.substeps
            for i in range(self.substeps):
                tau = i * h
                k1 = rhs(tau, y)
                k2 = rhs(tau + 0.5 * h, y + 0.5 * h * k1)
                k3 = rhs(tau + 0.5 * h, y + 0.5 * h * k2)
                k4 = rhs(tau + h, y + h * k3)
                y = y + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
            return xk1 - y

        super().__init__(f, traj.dims[x], traj)


class AbstractObjective:
    def __add__(self, other):
        """`+` flattens nested composites -- src/objectives/_objectives.jl:165-176."""
        a = self if isinstance(self, CompositeObjective) else CompositeObjective([self], [1.0])
        b = other if isinstance(other, CompositeObjective) else CompositeObjective([other], [1.0])
        return CompositeObjective(a.objectives + b.objectives, a.weights + b.weights)

    def __rmul__(self, num):
        """num * obj -- _objectives.jl:178-187."""
        if isinstance(self, CompositeObjective):
            return CompositeObjective(self.objectives, [float(num) * w for w in self.weights])
        return CompositeObjective([self], [float(num)])

    __mul__ = __rmul__


class CompositeObjective(AbstractObjective):
    """_objectives.jl:106-156."""

    def __init__(self, objectives, weights):
        self.objectives, self.weights = list(objectives), [float(w) for w in weights]


class NullObjective(AbstractObjective):
    """_objectives.jl:209-230."""

    def __init__(self, traj=None):
        pass


def _times(times, N):
    if times is None:
        return None
    t = np.asarray(list(times), dtype=np.int64)
    if t.size and (t.min() < 1 or t.max() > N):
        raise ValueError("times are 1-based knot indices in 1..N")
    return t


class QuadraticRegularizer(AbstractObjective):
    """QuadraticRegularizer(name, traj, R; baseline, times) -- src/objectives/regularizers.jl:38-64."""

    def __init__(self, name, traj, R, baseline=None, times=None):
        d = traj.dims[name]
        self.name = name
        self.R = np.full(d, float(R)) if np.isscalar(R) else np.asarray(R, dtype=np.float64)
        if self.R.shape != (d,):
            raise ValueError("length(R) must equal the component dimension")
        self.baseline = None if baseline is None else np.asarray(baseline, dtype=np.float64).reshape(d, traj.N)
        self.times = _times(times, traj.N)
        self.comp_off, self.comp_dim = traj.components[name][0], d


class LinearRegularizer(AbstractObjective):
    """LinearRegularizer(name, traj, R; times) -- regularizers.jl:207-227."""

    def __init__(self, name, traj, R, times=None):
        d = traj.dims[name]
        self.name = name
        self.R = np.full(d, float(R)) if np.isscalar(R) else np.asarray(R, dtype=np.float64)
        if self.R.shape != (d,):
            raise ValueError("length(R) must equal the component dimension")
        self.times = _times(times, traj.N)
        self.comp_off, self.comp_dim = traj.components[name][0], d


class MinimumTimeObjective(AbstractObjective):
    """MinimumTimeObjective(traj; D) -- src/objectives/minimum_time_objective.jl:24-40."""

    def __init__(self, traj, D=1.0):
        self.D = float(D)


class KnotPointObjective(AbstractObjective):
    """KnotPointObjective(l, names, traj, params; times, Qs) -- src/objectives/knot_point_objectives.jl:65-121.
    J = sum_i Q_i l(z_{t_i}[names], p_i).

    ``l`` is either the name of a loss built into the engine -- ``"sqdist"``: l(v, p) = ||v - p||^2 (p = None
    means zeros, i.e. ``norm(v)^2``); ``"lowrank_infidelity"``: l(v) = |1 - ||A v||^2| with the constant factor
    ``A`` (ket / unitary infidelity in isomorphic coordinates) -- or a callable ``l(v, p) -> float`` (the reference's closure form).  A
    callable is evaluated on the host per listed time together with its gradient and Hessian (``grad(v, p)``,
    ``hess(v, p)`` if given, else differentiated numerically, host/closures.py) and the engine merges the blocks."""

    KINDS = {"sqdist": 4, "lowrank_infidelity": 6}

    def __init__(self, l, names, traj, params=None, times=None, Qs=None, grad=None, hess=None, A=None):
        names = [names] if isinstance(names, str) else list(names)
        self.external = callable(l)
        if not self.external and l not in self.KINDS:
            raise ValueError(f"unknown built-in loss {l!r}")
        self.kind, self.var_names = l, names
        self.l, self.grad, self.hess = (l, grad, hess) if self.external else (None, None, None)
        self.times = _times(range(1, traj.N + 1) if times is None else times, traj.N)
        self.comps = np.concatenate([np.asarray(traj.components[n]) for n in names]).astype(np.int32)
        nt = self.times.size
        self.Qs = np.ones(nt) if Qs is None else np.asarray(Qs, dtype=np.float64)
        if self.Qs.shape != (nt,):
            raise ValueError("Qs must have the same length as times")
        if self.external:
            self.params = [None] * nt if params is None else list(params)
            if len(self.params) != nt:
                raise ValueError("params must have the same length as times")
            return
        self.A = None
        if l == "lowrank_infidelity":  # l(v) = |1 - ||A v||^2|, knot_hvp.jl:45-84 (ConstantLowRankHVP shape)
            self.A = np.asarray(A, dtype=np.float64)
            if self.A.ndim != 2 or self.A.shape[1] != self.comps.size:
                raise ValueError("A must be a (rank, n_comps) matrix")
            params = None
        if params is None:
            self.params = None
        else:
            P = np.asarray(params, dtype=np.float64)
            if P.shape == (self.comps.size,) and nt == 1:
                P = P[None, :]
            if P.shape != (nt, self.comps.size):
                raise ValueError("params must have the same length as times")
            self.params = P

    # host evaluation of a closure-based loss: per listed time Q_i l, Q_i grad l, Q_i hess l
    def external_blocks(self, Zk, need):
        from . import closures
        nt, nc = self.times.size, self.comps.size
        vals = np.zeros(nt)
        first = np.zeros((nt, nc)) if need >= 1 else None
        second = np.zeros((nt, nc, nc)) if need >= 2 else None
        for i, t in enumerate(self.times):
            v, p, Q = Zk[t - 1, self.comps], self.params[i], self.Qs[i]
            vals[i] = Q * float(self.l(v, p))
            if need >= 1:
                first[i] = Q * closures.gradient(self.l, v, p, self.grad)
            if need >= 2:
                Hm = closures.hessian(lambda x: self.l(x, p), v,
                                      None if self.grad is None else (lambda x: self.grad(x, p)),
                                      None if self.hess is None else (lambda x: self.hess(x, p)))
                second[i] = Q * Hm.T  # column-major block = transpose in C order (symmetric anyway)
        return vals, first, second


class GlobalKnotPointObjective(AbstractObjective):
    """GlobalKnotPointObjective(l, names, global_names, traj, params; times, Qs) --
    src/objectives/global_objectives.jl:151-206: J = sum_i Q_i l([z_{t_i}[names]; global_data[global_names]], p_i).
    Host-evaluated (a closure) and merged by the engine; gradient and Hessian accumulate over the listings."""

    external = True
    is_global = True

    def __init__(self, l, names, global_names, traj, params=None, times=None, Qs=None, grad=None, hess=None):
        names = [names] if isinstance(names, str) else list(names)
        if global_names is None:
            global_names = list(traj.global_components.keys())  # auto-detect, :171-174
        global_names = [global_names] if isinstance(global_names, str) else list(global_names)
        self.l, self.grad, self.hess = l, grad, hess
        self.var_names, self.global_names = names, global_names
        self.times = _times(range(1, traj.N + 1) if times is None else times, traj.N)
        self.comps = (np.concatenate([np.asarray(traj.components[n]) for n in names]).astype(np.int32)
                      if names else np.zeros(0, dtype=np.int32))
        self.gcomps = np.concatenate([np.asarray(traj.global_components[n]) for n in global_names]).astype(np.int32)
        nt = self.times.size
        self.Qs = np.ones(nt) if Qs is None else np.asarray(Qs, dtype=np.float64)
        self.params = [None] * nt if params is None else list(params)
        if self.Qs.shape != (nt,) or len(self.params) != nt:
            raise ValueError("Qs and params must have the same length as times")

    def _listings(self, Zk, g):
        return [np.concatenate([Zk[t - 1, self.comps], g[self.gcomps]]) for t in self.times]

    def external_blocks(self, Zk, need, g=None):
        from . import closures
        vs = self._listings(Zk, g)
        nl, nb = len(vs), self.comps.size + self.gcomps.size
        vals = np.zeros(nl)
        first = np.zeros((nl, nb)) if need >= 1 else None
        second = np.zeros((nl, nb, nb)) if need >= 2 else None
        for i, v in enumerate(vs):
            p, Q = self.params[i], self.Qs[i]
            vals[i] = Q * float(self.l(v, p))
            if need >= 1:
                first[i] = Q * closures.gradient(self.l, v, p, self.grad)
            if need >= 2:
                second[i] = Q * closures.hessian(lambda x: self.l(x, p), v,
                                                 None if self.grad is None else (lambda x: self.grad(x, p)),
                                                 None if self.hess is None else (lambda x: self.hess(x, p))).T
        return vals, first, second


class GlobalObjective(GlobalKnotPointObjective):
    """GlobalObjective(l, global_names, traj; Q) -- global_objectives.jl:35-52: J = Q l(global_data[global_names])."""

    def __init__(self, l, global_names, traj, Q=1.0, grad=None, hess=None):
        super().__init__((lambda v, p: l(v)), [], global_names, traj, params=[None], times=[1], Qs=[float(Q)],
                         grad=None if grad is None else (lambda v, p: grad(v)),
                         hess=None if hess is None else (lambda v, p: hess(v)))
        self.times = np.zeros(0, dtype=np.int64)  # no knot part: one listing of the global variables alone

    def _listings(self, Zk, g):
        return [g[self.gcomps]]


class NonlinearGlobalConstraint:
    """NonlinearGlobalConstraint(g, global_names, traj; equality) --
    src/constraints/nonlinear/global_constraint.jl:20-75: g(global_data[global_names]) = 0 or <= 0.  Host-evaluated and
    merged; its Jacobian/Hessian entries live in the global-variable columns."""

    external = True
    is_global = True

    def __init__(self, g, global_names, traj, equality=True, jac=None, hess=None):
        global_names = [global_names] if isinstance(global_names, str) else list(global_names)
        self.g, self.jac, self.hess, self.equality = g, jac, hess, bool(equality)
        self.global_names = global_names
        self.gcomps = np.concatenate([np.asarray(traj.global_components[n]) for n in global_names]).astype(np.int32)
        self.global_dim = self.gcomps.size
        self.g_dim = int(np.asarray(g(traj.global_data[self.gcomps])).size)
        self.dim = self.g_dim

    def external_blocks(self, Zk, need, mu=None, g=None):
        from . import closures
        v, gd, ng = g[self.gcomps], self.g_dim, self.gcomps.size
        vals = np.asarray(self.g(v), dtype=np.float64).reshape(gd)
        first = second = None
        if need >= 1:
            first = closures.jacobian(lambda x, p: self.g(x), v, None, gd, None if self.jac is None else (lambda x, p: self.jac(x))).T
        if need >= 2:
            m = np.ones(gd) if mu is None else np.asarray(mu, dtype=np.float64)
            second = closures.hessian(lambda x: m @ np.asarray(self.g(x)).reshape(gd), v,
                                      None if self.jac is None else (lambda x: m @ np.asarray(self.jac(x)).reshape(gd, ng)),
                                      None if self.hess is None else (lambda x: self.hess(x, m))).T
        return vals, first, second


def ket_fidelity_factor(goal_iso):
    """A (2 x 2n) with ||A psi~||^2 = |<goal|psi>|^2 for iso vectors psi~ = [Re psi; Im psi]."""
    g = np.asarray(goal_iso, dtype=np.float64)
    n = g.size // 2
    return np.vstack([np.concatenate([g[:n], g[n:]]), np.concatenate([-g[n:], g[:n]])])


def TerminalObjective(l, names, traj, goal=None, Q=1.0, grad=None, hess=None, A=None):
    """TerminalObjective(l, name, traj; Q) -- knot_point_objectives.jl:123-157: the loss at the last knot."""
    if callable(l):
        return KnotPointObjective(l, names, traj, params=[goal], times=[traj.N], Qs=[float(Q)], grad=grad, hess=hess)
    if l == "lowrank_infidelity":
        return KnotPointObjective(l, names, traj, times=[traj.N], Qs=[float(Q)], A=A)
    return KnotPointObjective(l, names, traj, params=None if goal is None else np.asarray(goal, dtype=np.float64)[None, :],
                              times=[traj.N], Qs=[float(Q)])


class NonlinearKnotPointConstraint:
    """NonlinearKnotPointConstraint(g, names, traj; equality, times, params) --
    src/constraints/nonlinear/knot_point_constraint.jl:27-107.

    ``g`` is either a kind built into the engine -- ``"norm"`` (g(v) = [||v|| - c], the shape of
    test/test_snippets.jl:39-45) or ``"sqnorm"`` (g(v) = [||v||^2 - c]) -- or a callable ``g(v, p) -> array``
    (the reference's closure form; g_dim is taken from one evaluation at the trajectory, as the reference does,
    knot_point_constraint.jl:84-90).  A callable is evaluated on the host together with its Jacobian and the
    Hessian of mu_i' g (``jac(v, p)``, ``hess(v, p, mu_i)`` if given, else differentiated numerically) and the
    engine merges the blocks."""

    KINDS = {"norm": 1, "sqnorm": 2}

    def __init__(self, g, names, traj, c=0.0, equality=True, times=None, params=None, jac=None, hess=None):
        names = [names] if isinstance(names, str) else list(names)
        self.external = callable(g)
        if not self.external and g not in self.KINDS:
            raise ValueError(f"unknown built-in g kind {g!r}")
        self.kind, self.var_names, self.c, self.equality = g, names, float(c), bool(equality)
        self.times = _times(range(1, traj.N + 1) if times is None else times, traj.N)
        self.comps = np.concatenate([np.asarray(traj.components[n]) for n in names]).astype(np.int32)
        self.g_dim, self.var_dim = 1, self.comps.size
        if self.external:
            self.g, self.jac, self.hess = g, jac, hess
            self.params = [None] * self.times.size if params is None else list(params)
            if len(self.params) != self.times.size:
                raise ValueError("params must have the same length as times")
            Zk = traj.vec()[:traj.dim * traj.N].reshape(traj.N, traj.dim)
            self.g_dim = int(np.asarray(g(Zk[self.times[0] - 1, self.comps], self.params[0])).size)
        self.dim = self.g_dim * self.times.size

    # host evaluation of a closure-based g: values, Jacobian blocks (column-major g_dim x n_comps), mu-weighted Hessians
    def external_blocks(self, Zk, need, mu=None):
        from . import closures
        nt, nc, gd = self.times.size, self.comps.size, self.g_dim
        vals = np.zeros((nt, gd))
        first = np.zeros((nt, nc, gd)) if need >= 1 else None
        second = np.zeros((nt, nc, nc)) if need >= 2 else None
        for i, t in enumerate(self.times):
            v, p = Zk[t - 1, self.comps], self.params[i]
            vals[i] = np.asarray(self.g(v, p), dtype=np.float64).reshape(gd)
            if need >= 1:
                first[i] = closures.jacobian(self.g, v, p, gd, self.jac).T
            if need >= 2:
                m = np.asarray(mu[i * gd:(i + 1) * gd], dtype=np.float64)
                Hm = closures.hessian(lambda x: m @ np.asarray(self.g(x, p)).reshape(gd), v,
                                      None if self.jac is None else (lambda x: m @ np.asarray(self.jac(x, p)).reshape(gd, nc)),
                                      None if self.hess is None else (lambda x: self.hess(x, p, m)))
                second[i] = Hm.T
        return vals, first, second


class DirectTrajOptProblem:
    """DirectTrajOptProblem(traj, obj, integrators; constraints) -- src/problems.jl:50-122 (fields only;
    the linear-constraint extraction done there is solver set-up, outside the hot path)."""

    def __init__(self, trajectory, objective, integrators, constraints=()):
        self.trajectory = trajectory
        self.objective = objective
        self.integrators = [integrators] if not isinstance(integrators, (list, tuple)) else list(integrators)
        self.constraints = list(constraints)
