"""CPU oracle for the DirectTrajOpt.jl NLP-callback hot path (TEST INFRASTRUCTURE ONLY).

This file is a NumPy/SciPy *restatement* of the reference's evaluator path.  It is the checker
used by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg; nothing in
the product path (``directtrajopt.jl_amd/``) may import it.

Parity status: **values unpinned against the Julia reference** (no ``julia`` in the build
container, and the reference's own tests hold no literal expected values for this path, see
SURVEY.md §4/§8c).  The oracle is instead pinned by
  * the reference's structure-construction rules, re-enacted literally with scipy.sparse
    (pattern matrices -> vstack -> CSC walk), so the sparsity indices follow
    ``src/solvers/evaluator.jl:119-209`` by construction;
  * the one hard-coded trajectory of the reference's tests (``test/test_utils.jl:57-82``) with the
    closed-form ``DerivativeIntegrator`` answers (``tests/test_oracle_pinning.py``);
  * complex-step / high-precision (mpmath) differentiation of ``x_{k+1} - exp(dt G(u)) x_k`` for
    the bilinear first- and second-order terms (same test file).

Every function cites the reference lines it restates (paths relative to /root/reference).
Indices at this API are 0-based unless a name ends in ``1`` (1-based, Julia/MOI convention).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

# ----------------------------------------------------------------------------------------------
# Problem description (mirrors include/dto_engine.h)
# ----------------------------------------------------------------------------------------------


@dataclass
class BilinearIntegrator:
    """x_{k+1} - exp(dt_k G(u_k)) x_k = 0, G(u) = G[0] + sum_j u_j G[j].

    src/integrators/bilinear_integrator.jl:61-85 (defect at :81)."""

    x_off: int
    x_dim: int
    u_off: int
    u_dim: int
    G: np.ndarray  # (u_dim+1, n, n): drift then drives

    kind = "bilinear"


@dataclass
class DerivativeIntegrator:
    """x_{k+1} - x_k - dt_k xdot_k = 0.  src/integrators/derivative_integrator.jl:26-49 (:45)."""

    x_off: int
    x_dim: int
    xdot_off: int

    kind = "derivative"


@dataclass
class ClosureIntegrator:
    """Any other AbstractIntegrator (src/integrators/_integrators.jl:22-34), e.g. TimeDependentBilinearIntegrator
    (time_dependent_bilinear_integrator.jl:60-244): residual f(zz, k) of the stacked knot pair zz = [z_k; z_{k+1}]
    with analytic jac(zz, k) -> (x_dim, 2z) and hess(zz, k, mu) -> (2z, 2z) standing in for ForwardDiff
    (:178-244).  Structure: the generic dense block per interval (_integrators.jl:49-77)."""

    f: object
    jac: object
    hess: object
    x_dim: int

    kind = "external"


@dataclass
class TimeDependentBilinearIntegrator:
    """TimeDependentBilinearIntegrator(G, x, u, t, traj; spline_order) --
    src/integrators/time_dependent_bilinear_integrator.jl:60-132.

    Defect of interval k (the compiled residual, :123-128): x_{k+1} - y(1), where y solves on the normalised interval
        dy/dtau = G(u(tau), t_k + tau dt_k) (y dt_k),   y(0) = x_k                      (f!, :102-106)
    with the controls held, u(tau) = u_k (spline_order 0, :85-86), or interpolated linearly to the next knot,
    u(tau) = u_k + tau (u_{k+1} - u_k) (spline_order 1, :87-92).  The interval's variables are the stacked knot pair
    zz = [z_k; z_{k+1}] (x_k, u_k, t_k, dt_k from z_k; x_{k+1}, u_{k+1} from z_{k+1}: evaluate! :150-170, eval_jacobian :180-204).

    The closure G(u, t) is the generator family the engine integrates on the device (include/dto_engine.h,
    DTO_INTEGRATOR_TIME_DEPENDENT_BILINEAR):
        G(u, t) = sum_{j=0..m} ubar_j (G[j] + sum_c phi_c(t) H_c[j]),  ubar_0 = 1,  phi_c = cos(omega_c t) | sin(omega_c t)
    -- the reference's own test closure `G(a) + 0.1 cos(t) I` (:262-266) is the member mods = [("cos", 1.0, [0.1 I, 0, ...])].

    WHERE THIS RESTATEMENT DEPARTS FROM THE REFERENCE, and why: the reference integrates with adaptive Tsit5 at
    OrdinaryDiffEq's default tolerances (:126, reltol 1e-3 / abstol 1e-6) and differentiates THROUGH the adaptive solver with
    ForwardDiff (:178-244), so its own numbers carry an O(1e-3 .. 1e-6) integration error and are not reproducible to fp64
    precision by any other code.  What is well defined is the ODE.  This oracle integrates it with classical RK4 on `substeps`
    equal steps -- the scheme the engine's kernel states in its ABI -- and `tdb_flow_reference` integrates the SAME right-hand
    side with scipy's DOP853 at rtol 1e-12: tests/test_oracle_pinning.py checks that the fixed-step map converges to that
    solution at fourth order and sits inside the reference's Tsit5 tolerance at the sub-step counts the tests use.
    Derivatives: complex step for the Jacobian (exact to rounding for this analytic map), Richardson-extrapolated central
    differences of the complex-step gradient for the Hessian of mu' f (checked against 40-digit mpmath in the same test file);
    they stand in for ForwardDiff.jacobian! (:180) and ForwardDiff.hessian (:216) of the same map."""

    x_off: int
    x_dim: int
    u_off: int
    u_dim: int
    t_off: int
    G: np.ndarray                      # (u_dim+1, n, n)
    mods: list = field(default_factory=list)   # [(kind "cos"|"sin", omega, H (u_dim+1, n, n))]
    spline_order: int = 1
    substeps: int = 32
    z: int = 0                         # components per knot; set by `bind`
    dt_idx: int = 0

    kind = "external"                  # generic dense block per interval (_integrators.jl:49-77)

    def bind(self, z, dt_idx):
        self.z, self.dt_idx = z, dt_idx
        return self

    def generator(self, u, t):
        """G(u, t) of the family; works for complex arguments (complex-step differentiation) and for a leading batch axis
        (u: (..., m), t: (...,)) -> (..., n, n)."""
        u = np.asarray(u)
        t = np.asarray(t)
        ub = np.concatenate([np.ones(u.shape[:-1] + (1,), dtype=np.result_type(u, t)), u], axis=-1)
        M = np.einsum("...j,jrc->...rc", ub, self.G)
        for kind, w, Hc in self.mods:
            phi = np.cos(w * t) if kind == "cos" else np.sin(w * t)
            M = M + phi[..., None, None] * np.einsum("...j,jrc->...rc", ub, Hc)
        return M

    def _unpack(self, zz):
        z, n, m = self.z, self.x_dim, self.u_dim
        xk, uk = zz[..., self.x_off:self.x_off + n], zz[..., self.u_off:self.u_off + m]
        tk, dt = zz[..., self.t_off], zz[..., self.dt_idx]
        xk1, uk1 = zz[..., z + self.x_off:z + self.x_off + n], zz[..., z + self.u_off:z + self.u_off + m]
        return xk, uk, tk, dt, xk1, uk1

    def rhs(self, zz):
        """The reference's f! (:102-106) as rhs(tau, y) for the interval whose stacked variables are zz (a leading batch
        axis of zz and y is carried along)."""
        xk, uk, tk, dt, xk1, uk1 = self._unpack(zz)
        if self.spline_order == 0:
            ctrl = lambda tau: uk                                   # u_fn, :85-86
        elif self.spline_order == 1:
            ctrl = lambda tau: uk + tau * (uk1 - uk)                # u_fn, :87-92
        else:
            raise ValueError(f"Unsupported spline order: {self.spline_order}")   # :94
        return lambda tau, y: np.einsum("...rc,...c->...r", self.generator(ctrl(tau), tk + tau * dt), y * dt[..., None])

    def f(self, zz, k=0):
        """Residual x_{k+1} - y(1) (:123-128) with y(1) from `substeps` classical Runge-Kutta-4 steps.  zz: (2z,) or (B, 2z)."""
        zz = np.asarray(zz)
        xk, _, _, _, xk1, _ = self._unpack(zz)
        rhs = self.rhs(zz)
        y, h = xk.astype(np.result_type(zz, np.float64)), 1.0 / self.substeps
        for i in range(self.substeps):
            tau = i * h
            k1 = rhs(tau, y)
            k2 = rhs(tau + 0.5 * h, y + 0.5 * h * k1)
            k3 = rhs(tau + 0.5 * h, y + 0.5 * h * k2)
            k4 = rhs(tau + h, y + h * k3)
            y = y + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        return xk1 - y

    def jac(self, zz, k=0):
        """d f / d zz, (..., x_dim, 2z): complex step, exact to rounding (stands in for ForwardDiff.jacobian!, :180-204).
        All 2z directions run as one batch."""
        zz = np.asarray(zz, dtype=np.float64)
        nv = zz.shape[-1]
        zc = np.repeat(zz[..., None, :].astype(np.complex128), nv, axis=-2)   # (..., direction, 2z)
        zc[..., np.arange(nv), np.arange(nv)] += 1e-30j
        return np.swapaxes(self.f(zc).imag / 1e-30, -1, -2)

    def hess(self, zz, k, mu):
        """Hessian of mu' f, (2z, 2z) (stands in for ForwardDiff.hessian, :216-240): central differences of the complex-step
        gradient at two step sizes, Richardson-extrapolated (error O(h^4), h = 2e-3 relative to max(1, |zz_i|))."""
        zz = np.asarray(zz, dtype=np.float64)
        mu = np.asarray(mu, dtype=np.float64)
        nv = zz.size
        h = 2e-3 * np.maximum(1.0, np.abs(zz))
        pts = np.repeat(zz[None, None, :], 4, axis=0).repeat(nv, axis=1)         # (4 offsets, direction i, 2z)
        for q, c in enumerate((1.0, -1.0, 0.5, -0.5)):
            pts[q, np.arange(nv), np.arange(nv)] += c * h
        g = np.einsum("r,qirv->qiv", mu, self.jac(pts))                           # gradients of mu' f at every point
        d1 = (g[0] - g[1]) / (2.0 * h[:, None])
        d2 = (g[2] - g[3]) / h[:, None]
        H = (4.0 * d2 - d1) / 3.0
        return 0.5 * (H + H.T)


def tdb_flow_reference(integ, zz, rtol=1e-12, atol=1e-14):
    """y(1) of the interval's ODE (time_dependent_bilinear_integrator.jl:102-106, :123-127) by scipy's adaptive DOP853: the
    solution the reference's Tsit5 approximates to its own tolerances.  Test infrastructure for the convergence check."""
    from scipy.integrate import solve_ivp
    zz = np.asarray(zz, dtype=np.float64)
    xk = integ._unpack(zz)[0]
    sol = solve_ivp(integ.rhs(zz), (0.0, 1.0), xk, method="DOP853", rtol=rtol, atol=atol)
    return sol.y[:, -1]


@dataclass
class QuadraticRegularizer:
    """src/objectives/regularizers.jl:38-167."""

    comp_off: int
    comp_dim: int
    R: np.ndarray
    baseline: Optional[np.ndarray] = None  # (comp_dim, N) or None (= zeros)
    times1: Optional[Sequence[int]] = None  # 1-based knots, None = 1:N

    kind = "quadratic"


@dataclass
class LinearRegularizer:
    """src/objectives/regularizers.jl:207-313."""

    comp_off: int
    comp_dim: int
    R: np.ndarray
    times1: Optional[Sequence[int]] = None

    kind = "linear"


@dataclass
class MinimumTimeObjective:
    """src/objectives/minimum_time_objective.jl:24-76."""

    D: float = 1.0

    kind = "mintime"


@dataclass
class KnotSqDistObjective:
    """KnotPointObjective / TerminalObjective with l(v, p) = ||v - p||^2
    (src/objectives/knot_point_objectives.jl:65-243)."""

    comps: Sequence[int]
    times1: Sequence[int]
    Qs: Sequence[float]
    params: Optional[np.ndarray] = None  # (n_times, n_comps) or None (= zeros)

    kind = "knot_sqdist"


@dataclass
class LowRankInfidelityObjective:
    """KnotPointObjective / TerminalObjective with l(v) = |1 - ||A v||^2|, A constant (k x n_comps): the
    coherent-fidelity losses in isomorphic coordinates.  Its exact Hessian -2 sign(1 - F) A'A is the
    ConstantLowRankHVP(A, :neg2_sign) shape of src/objectives/knot_hvp.jl:45-84; placement follows
    knot_point_objectives.jl:173-243 like every KnotPointObjective."""

    comps: Sequence[int]
    times1: Sequence[int]
    Qs: Sequence[float]
    A: np.ndarray

    kind = "knot_lowrank"


def as_closure_objective(t: "LowRankInfidelityObjective"):
    """The same loss as an explicit closure term (what a user of the reference writes)."""
    A = np.asarray(t.A, dtype=np.float64)

    def l(v, p):
        return abs(1.0 - float((A @ v) @ (A @ v)))

    def grad(v, p):
        return -2.0 * np.sign(1.0 - (A @ v) @ (A @ v)) * (A.T @ (A @ v))

    def hess(v, p):
        return -2.0 * np.sign(1.0 - (A @ v) @ (A @ v)) * (A.T @ A)

    return ClosureKnotObjective(l, grad, hess, list(t.comps), list(t.times1), list(t.Qs), None)


@dataclass
class KnotConstraint:
    """NonlinearKnotPointConstraint with a built-in g (closed set).

    kind "norm":   g(v) = [ ||v||_2 - c ]      (shape of test/test_snippets.jl:39-45)
    kind "sqnorm": g(v) = [ ||v||_2^2 - c ]
    src/constraints/nonlinear/knot_point_constraint.jl:27-107, 235-294."""

    kind: str
    comps: Sequence[int]  # 0-based knot-local component indices (vcat of var_names comps)
    c: float
    times1: Sequence[int]
    equality: bool = False

    g_dim = 1


@dataclass
class ClosureKnotConstraint:
    """NonlinearKnotPointConstraint with a user closure g(v, p) of g_dim outputs
    (src/constraints/nonlinear/knot_point_constraint.jl:27-107, 235-294).  The reference differentiates g
    with ForwardDiff; the oracle takes the analytic derivatives from the test: jac(v, p) -> (g_dim, d),
    hess(v, p, mu) -> (d, d) = Hessian of mu' g."""

    g: object
    jac: object
    hess: object
    comps: Sequence[int]
    times1: Sequence[int]
    g_dim: int
    params: Optional[Sequence[object]] = None
    equality: bool = False

    kind = "closure"


@dataclass
class GlobalClosureConstraint:
    """NonlinearGlobalConstraint g(global_data[gcomps]) (src/constraints/nonlinear/global_constraint.jl:20-160) with
    analytic jac(v) -> (g_dim, ng) and hess(v, mu) -> (ng, ng) standing in for ForwardDiff (:126-158)."""

    g: object
    jac: object
    hess: object
    gcomps: Sequence[int]
    g_dim: int
    equality: bool = True

    kind = "global_closure"
    times1 = (0,)  # one row block (row counting shares the knot-constraint code)


@dataclass
class GlobalClosureObjective:
    """GlobalObjective (times1 empty: Q l(g)) / GlobalKnotPointObjective (sum_i Q_i l([z_t[comps]; g], p_i)) of
    src/objectives/global_objectives.jl:35-350, g = global_data[gcomps]; analytic grad(v, p), hess(v, p).  Unlike
    KnotPointObjective, gradient and Hessian ACCUMULATE over the listings (:270-271, :341)."""

    l: object
    grad: object
    hess: object
    comps: Sequence[int]
    gcomps: Sequence[int]
    times1: Sequence[int]
    Qs: Sequence[float]
    params: Optional[Sequence[object]] = None

    kind = "global_closure"

    def listings(self, prob):
        """(global index list, weight, param) per listing."""
        g_idx = prob.N * prob.z + np.asarray(self.gcomps, dtype=np.int64)
        if len(self.times1) == 0:
            return [(g_idx, self.Qs[0], _param(self, 0))]
        comps = np.asarray(self.comps, dtype=np.int64)
        return [(np.concatenate([(t1 - 1) * prob.z + comps, g_idx]), self.Qs[i], _param(self, i))
                for i, t1 in enumerate(self.times1)]


@dataclass
class ClosureKnotObjective:
    """KnotPointObjective / TerminalObjective with a user closure l(v, p)
    (src/objectives/knot_point_objectives.jl:65-243); analytic grad(v, p), hess(v, p) from the test."""

    l: object
    grad: object
    hess: object
    comps: Sequence[int]
    times1: Sequence[int]
    Qs: Sequence[float]
    params: Optional[Sequence[object]] = None

    kind = "knot_closure"


@dataclass
class Problem:
    N: int
    z: int  # traj.dim
    dt_idx: int  # 0-based component index of the timestep inside a knot
    integrators: List[object]
    objectives: List[object] = field(default_factory=list)  # terms
    weights: Optional[List[float]] = None  # composite weights (None: all 1)
    constraints: List[object] = field(default_factory=list)  # KnotConstraint / ClosureKnotConstraint
    gd: int = 0  # global_dim (columns exist, no hot-path term touches them)
    Z0: Optional[np.ndarray] = None  # initial point (value-dependent constraint patterns)

    @property
    def K(self):
        return self.N - 1

    @property
    def n_vars(self):
        return self.z * self.N + self.gd

    def w(self, i):
        return 1.0 if self.weights is None else float(self.weights[i])


def times0(term_times1, N):
    if term_times1 is None:
        return np.arange(N)
    return np.asarray(term_times1, dtype=np.int64) - 1


# ----------------------------------------------------------------------------------------------
# Integrators
# ----------------------------------------------------------------------------------------------


def _knot(Z, prob, k0):
    return Z[k0 * prob.z:(k0 + 1) * prob.z]


def _Gu(integ, u):
    return integ.G[0] + np.tensordot(u, integ.G[1:], axes=(0, 0))


def integrator_evaluate(integ, prob, Z):
    """evaluate! -- bilinear_integrator.jl:98-107, derivative_integrator.jl:55-64."""
    d = integ.x_dim
    out = np.zeros(d * prob.K)
    for k in range(prob.K):
        zk, zk1 = _knot(Z, prob, k), _knot(Z, prob, k + 1)
        if integ.kind == "external":
            out[k * d:(k + 1) * d] = np.asarray(integ.f(np.concatenate([zk, zk1]), k), dtype=np.float64)
            continue
        dt = zk[prob.dt_idx]
        xk = zk[integ.x_off:integ.x_off + d]
        xk1 = zk1[integ.x_off:integ.x_off + d]
        if integ.kind == "bilinear":
            u = zk[integ.u_off:integ.u_off + integ.u_dim]
            out[k * d:(k + 1) * d] = xk1 - sla.expm(dt * _Gu(integ, u)) @ xk
        else:
            xd = zk[integ.xdot_off:integ.xdot_off + d]
            out[k * d:(k + 1) * d] = xk1 - xk - dt * xd
    return out


def integrator_jacobian_structure(integ, prob):
    """get_jacobian_structure -- src/integrators/_integrators.jl:49-60."""
    d, z = integ.x_dim, prob.z
    S = sp.lil_matrix((d * prob.K, prob.n_vars))
    for k in range(prob.K):
        S[k * d:(k + 1) * d, k * z:(k + 2) * z] = 1.0
    return S.tocsc()


def bilinear_block_jacobian(integ, prob, zk):
    """Dense n x z block d(delta_k)/d(z_k) of the bilinear defect (the x_{k+1} half is +I).

    Exact derivative of bilinear_integrator.jl:81 (the reference differentiates the same expression
    with ForwardDiff at :111-131)."""
    n, m, z = integ.x_dim, integ.u_dim, prob.z
    dt = zk[prob.dt_idx]
    x = zk[integ.x_off:integ.x_off + n]
    u = zk[integ.u_off:integ.u_off + m]
    Gu = _Gu(integ, u)
    A = dt * Gu
    E = sla.expm(A)
    B = np.zeros((n, z))
    B[:, integ.x_off:integ.x_off + n] += -E
    for j in range(m):
        L = sla.expm_frechet(A, dt * integ.G[1 + j], compute_expm=False)
        B[:, integ.u_off + j] += -(L @ x)
    B[:, prob.dt_idx] += -(Gu @ (E @ x))
    return B


def integrator_jacobian(integ, prob, Z):
    """eval_jacobian -- bilinear_integrator.jl:111-131, derivative_integrator.jl:68-86.

    Returned as CSC with exact zeros NOT stored (Julia's sparse setindex! drops zeros written to
    unstored positions, which is how the ForwardDiff.jacobian! into a sparse view behaves)."""
    d, z = integ.x_dim, prob.z
    J = sp.lil_matrix((d * prob.K, prob.n_vars))
    for k in range(prob.K):
        zk = _knot(Z, prob, k)
        blk = np.zeros((d, 2 * z))
        if integ.kind == "external":
            zz = np.concatenate([zk, _knot(Z, prob, k + 1)])
            J[k * d:(k + 1) * d, k * z:(k + 2) * z] = np.asarray(integ.jac(zz, k), dtype=np.float64).reshape(d, 2 * z)
            continue
        if integ.kind == "bilinear":
            blk[:, :z] = bilinear_block_jacobian(integ, prob, zk)
        else:
            dt = zk[prob.dt_idx]
            xd = zk[integ.xdot_off:integ.xdot_off + d]
            blk[:, integ.x_off:integ.x_off + d] += -np.eye(d)
            blk[:, integ.xdot_off:integ.xdot_off + d] += -dt * np.eye(d)
            blk[:, prob.dt_idx] += -xd
        blk[:, z + integ.x_off:z + integ.x_off + d] += np.eye(d)
        J[k * d:(k + 1) * d, k * z:(k + 2) * z] = blk
    J = J.tocsc()
    J.eliminate_zeros()
    return J


def integrator_hessian_structure(prob):
    """get_hessian_of_lagrangian_structure -- src/integrators/_integrators.jl:68-77."""
    z = prob.z
    S = sp.lil_matrix((prob.n_vars, prob.n_vars))
    for k in range(prob.K):
        S[k * z:(k + 2) * z, k * z:(k + 2) * z] = 1.0
    return S.tocsc()


def _second_frechet_action(A, E1, E2, x):
    """d^2/(ds dt) exp(A + s E1 + t E2) x at 0, via the 3x3 block-triangular exponential."""
    n = A.shape[0]
    Zr = np.zeros((n, n))

    def top_right(Ea, Eb):
        M = np.block([[A, Ea, Zr], [Zr, A, Eb], [Zr, Zr, A]])
        return sla.expm(M)[:n, 2 * n:]

    return (top_right(E1, E2) + top_right(E2, E1)) @ x


def bilinear_block_hessian(integ, prob, zk, mu, skip_uu=False):
    """Dense z x z Hessian of mu' f wrt z_k for the bilinear defect (x_{k+1} rows/cols are 0).

    Exact second derivative of bilinear_integrator.jl:81 (reference: ForwardDiff.hessian,
    :135-161).  skip_uu leaves the (u_i, u_j) block at zero: its second-order Frechet terms need the exponential of
    a 3n x 3n matrix, which at n = 1024 takes minutes per pair (the large-state tests check that block against central
    differences of the Jacobian instead, the reference's own method, evaluator.jl:779-790)."""
    n, m, z = integ.x_dim, integ.u_dim, prob.z
    dt = zk[prob.dt_idx]
    x = zk[integ.x_off:integ.x_off + n]
    u = zk[integ.u_off:integ.u_off + m]
    Gu = _Gu(integ, u)
    A = dt * Gu
    E = sla.expm(A)
    H = np.zeros((z, z))
    xs = slice(integ.x_off, integ.x_off + n)
    Ls = [sla.expm_frechet(A, dt * integ.G[1 + j], compute_expm=False) for j in range(m)]
    Ex = E @ x
    for j in range(m):
        uj = integ.u_off + j
        v = -(Ls[j].T @ mu)
        H[xs, uj] += v
        H[uj, xs] += v
        # d/d(dt) d/du_j [exp(dt G) x] = G_j E x + G L(A, dt G_j) x
        val = -(mu @ (integ.G[1 + j] @ Ex + Gu @ (Ls[j] @ x)))
        H[uj, prob.dt_idx] += val
        H[prob.dt_idx, uj] += val
        for i in range(0 if skip_uu else j + 1):
            ui = integ.u_off + i
            val = -(mu @ _second_frechet_action(A, dt * integ.G[1 + i], dt * integ.G[1 + j], x))
            H[ui, uj] += val
            if i != j:
                H[uj, ui] += val
    v = -((Gu @ E).T @ mu)
    H[xs, prob.dt_idx] += v
    H[prob.dt_idx, xs] += v
    H[prob.dt_idx, prob.dt_idx] += -(mu @ (Gu @ (Gu @ Ex)))
    return H


def integrator_hessian(integ, prob, Z, mu, skip_uu=False):
    """eval_hessian_of_lagrangian -- bilinear_integrator.jl:135-161, derivative_integrator.jl:90-116."""
    d, z = integ.x_dim, prob.z
    H = sp.lil_matrix((prob.n_vars, prob.n_vars))
    for k in range(prob.K):
        zk = _knot(Z, prob, k)
        muk = mu[k * d:(k + 1) * d]
        blk = np.zeros((2 * z, 2 * z))
        if integ.kind == "external":
            zz = np.concatenate([zk, _knot(Z, prob, k + 1)])
            blk = np.asarray(integ.hess(zz, k, muk), dtype=np.float64).reshape(2 * z, 2 * z)
        elif integ.kind == "bilinear":
            blk[:z, :z] = bilinear_block_hessian(integ, prob, zk, muk, skip_uu)
        else:
            for i in range(d):
                blk[integ.xdot_off + i, prob.dt_idx] += -muk[i]
                blk[prob.dt_idx, integ.xdot_off + i] += -muk[i]
        cur = H[k * z:(k + 2) * z, k * z:(k + 2) * z].toarray()
        H[k * z:(k + 2) * z, k * z:(k + 2) * z] = cur + blk  # `.+=` at :158
    return H.tocsc()


# ----------------------------------------------------------------------------------------------
# Nonlinear knot-point constraints (built-in g kinds)
# ----------------------------------------------------------------------------------------------


def _param(term, i):
    return None if term.params is None else term.params[i]


def _g(con, v, i=0):
    if con.kind == "closure":
        return np.asarray(con.g(v, _param(con, i)), dtype=np.float64).reshape(con.g_dim)
    if con.kind == "norm":
        return np.array([np.sqrt(v @ v) - con.c])
    if con.kind == "sqnorm":
        return np.array([v @ v - con.c])
    raise ValueError(con.kind)


def _g_jac(con, v, i=0):
    if con.kind == "closure":
        return np.asarray(con.jac(v, _param(con, i)), dtype=np.float64).reshape(con.g_dim, len(v))
    if con.kind == "norm":
        return (v / np.sqrt(v @ v))[None, :]
    return (2.0 * v)[None, :]


def _g_hess(con, v, mu, i=0):
    d = len(v)
    if con.kind == "closure":
        return np.asarray(con.hess(v, _param(con, i), np.asarray(mu)), dtype=np.float64).reshape(d, d)
    if con.kind == "norm":
        r = np.sqrt(v @ v)
        return mu[0] * (np.eye(d) / r - np.outer(v, v) / r**3)
    return mu[0] * 2.0 * np.eye(d)


def _global_vals(con, prob, Z):
    return Z[prob.N * prob.z + np.asarray(con.gcomps, dtype=np.int64)]


def constraint_evaluate(con, prob, Z):
    """evaluate! -- knot_point_constraint.jl:235-247."""
    if con.kind == "global_closure":  # global_constraint.jl:96-104
        return np.asarray(con.g(_global_vals(con, prob, Z)), dtype=np.float64).reshape(con.g_dim)
    out = np.zeros(con.g_dim * len(con.times1))
    comps = np.asarray(con.comps)
    for i, t1 in enumerate(con.times1):
        v = _knot(Z, prob, t1 - 1)[comps]
        out[i * con.g_dim:(i + 1) * con.g_dim] = _g(con, v, i)
    return out


def constraint_jacobian(con, prob, Z):
    """eval_jacobian -- knot_point_constraint.jl:254-268 (exact zeros not stored)."""
    J = sp.lil_matrix((con.g_dim * len(con.times1), prob.n_vars))
    if con.kind == "global_closure":  # global_constraint.jl:111-128
        Jg = np.asarray(con.jac(_global_vals(con, prob, Z)), dtype=np.float64).reshape(con.g_dim, len(con.gcomps))
        for r in range(con.g_dim):
            for b, gc in enumerate(con.gcomps):
                if Jg[r, b] != 0.0:
                    J[r, prob.N * prob.z + gc] = Jg[r, b]
        return J.tocsc()
    comps = np.asarray(con.comps)
    for i, t1 in enumerate(con.times1):
        v = _knot(Z, prob, t1 - 1)[comps]
        Jg = _g_jac(con, v, i)
        for r in range(con.g_dim):
            for c, comp in enumerate(comps):
                if Jg[r, c] != 0.0:
                    J[i * con.g_dim + r, (t1 - 1) * prob.z + comp] = Jg[r, c]
    return J.tocsc()


def constraint_hessian(con, prob, Z, mu):
    """eval_hessian_of_lagrangian -- knot_point_constraint.jl:275-294."""
    H = sp.lil_matrix((prob.n_vars, prob.n_vars))
    if con.kind == "global_closure":  # global_constraint.jl:141-158
        Hg = np.asarray(con.hess(_global_vals(con, prob, Z), np.asarray(mu)), dtype=np.float64)
        idx = prob.N * prob.z + np.asarray(con.gcomps, dtype=np.int64)
        for a, ia in enumerate(idx):
            for b, ib in enumerate(idx):
                H[ia, ib] = Hg[a, b]
        return H.tocsc()
    comps = np.asarray(con.comps)
    for i, t1 in enumerate(con.times1):
        v = _knot(Z, prob, t1 - 1)[comps]
        Hg = _g_hess(con, v, mu[i * con.g_dim:(i + 1) * con.g_dim], i)
        base = (t1 - 1) * prob.z
        for a, ca in enumerate(comps):  # hessian! writes the whole block view: a later listing of the knot overwrites
            for b, cb in enumerate(comps):
                H[base + ca, base + cb] = Hg[a, b]
    return H.tocsc()


# ----------------------------------------------------------------------------------------------
# Objectives
# ----------------------------------------------------------------------------------------------


def _baseline(term, t0):
    if term.baseline is None:
        return np.zeros(term.comp_dim)
    return np.asarray(term.baseline)[:, t0]


def term_value(term, prob, Z):
    """objective_value -- regularizers.jl:79-91, :240-249, minimum_time_objective.jl:44-50."""
    J = 0.0
    if term.kind == "global_closure":  # global_objectives.jl:61-67, 218-235
        return sum(Q * float(term.l(Z[idx], p)) for idx, Q, p in term.listings(prob))
    if term.kind == "knot_lowrank":
        return term_value(as_closure_objective(term), prob, Z)
    if term.kind == "knot_closure":  # knot_point_objectives.jl:173-182
        comps = np.asarray(term.comps)
        for i, t1 in enumerate(term.times1):
            J += term.Qs[i] * float(term.l(_knot(Z, prob, t1 - 1)[comps], _param(term, i)))
        return J
    if term.kind == "knot_sqdist":  # knot_point_objectives.jl:173-182 (every listed time counts)
        comps = np.asarray(term.comps)
        for i, t1 in enumerate(term.times1):
            v = _knot(Z, prob, t1 - 1)[comps]
            p = np.zeros(len(comps)) if term.params is None else np.asarray(term.params)[i]
            J += term.Qs[i] * float((v - p) @ (v - p))
        return J
    if term.kind == "mintime":
        for k in range(prob.K):
            J += _knot(Z, prob, k)[prob.dt_idx]
        return term.D * J
    for t0 in times0(term.times1, prob.N):
        zk = _knot(Z, prob, t0)
        v = zk[term.comp_off:term.comp_off + term.comp_dim]
        dt = zk[prob.dt_idx]
        if term.kind == "quadratic":
            r = dt * (v - _baseline(term, t0))
            J += 0.5 * r @ (term.R * r)
        else:
            J += dt * (term.R @ v)
    return J


def term_gradient_accumulate(grad, term, prob, Z, scale=1.0):
    """gradient! (accumulating form) -- regularizers.jl:93-115, :251-271, minimum_time_objective.jl:52-66."""
    if term.kind == "global_closure":  # global_objectives.jl:69-87, 237-275 (accumulating over the listings)
        for idx, Q, p in term.listings(prob):
            np.add.at(grad, idx, scale * Q * np.asarray(term.grad(Z[idx], p), dtype=np.float64))
        return
    if term.kind == "knot_lowrank":
        return term_gradient_accumulate(grad, as_closure_objective(term), prob, Z, scale)
    if term.kind == "knot_closure":  # knot_point_objectives.jl:184-207 (per listed time: overwrite, then scale)
        comps = np.asarray(term.comps)
        tmp = np.zeros_like(grad)
        for i, t1 in enumerate(term.times1):
            v = _knot(Z, prob, t1 - 1)[comps]
            tmp[(t1 - 1) * prob.z + comps] = term.Qs[i] * np.asarray(term.grad(v, _param(term, i)), dtype=np.float64)
        grad += scale * tmp
        return
    if term.kind == "knot_sqdist":
        # gradient! overwrites the knot's view per listed time, then scales (knot_point_objectives.jl:184-207)
        comps = np.asarray(term.comps)
        tmp = np.zeros_like(grad)
        for i, t1 in enumerate(term.times1):
            v = _knot(Z, prob, t1 - 1)[comps]
            p = np.zeros(len(comps)) if term.params is None else np.asarray(term.params)[i]
            tmp[(t1 - 1) * prob.z + comps] = term.Qs[i] * 2.0 * (v - p)
        grad += scale * tmp
        return
    if term.kind == "mintime":
        for k in range(prob.K):
            grad[k * prob.z + prob.dt_idx] += scale * term.D
        return
    for t0 in times0(term.times1, prob.N):
        zk = _knot(Z, prob, t0)
        v = zk[term.comp_off:term.comp_off + term.comp_dim]
        dt = zk[prob.dt_idx]
        base = t0 * prob.z
        if term.kind == "quadratic":
            dv = v - _baseline(term, t0)
            grad[base + term.comp_off:base + term.comp_off + term.comp_dim] += scale * dt**2 * (term.R * dv)
            grad[base + prob.dt_idx] += scale * (dv @ (term.R * dv)) * dt
        else:
            grad[base + term.comp_off:base + term.comp_off + term.comp_dim] += scale * term.R * dt
            grad[base + prob.dt_idx] += scale * (term.R @ v)


def term_hessian_structure(term, prob):
    """hessian_structure -- regularizers.jl:117-140, :273-293, minimum_time_objective.jl:68-72."""
    S = sp.lil_matrix((prob.n_vars, prob.n_vars))
    if term.kind == "mintime":
        return S.tocsc()
    if term.kind == "global_closure":  # global_objectives.jl:89-101, 277-300: the whole index block
        for idx, _, _ in term.listings(prob):
            for a in idx:
                for b in idx:
                    S[a, b] = 1.0
        return S.tocsc()
    if term.kind in ("knot_sqdist", "knot_closure", "knot_lowrank"):  # knot_point_objectives.jl:209-222
        comps = np.asarray(term.comps)
        for t1 in term.times1:
            idx = (t1 - 1) * prob.z + comps
            for a in idx:
                for b in idx:
                    S[a, b] = 1.0
        return S.tocsc()
    for t0 in times0(term.times1, prob.N):
        base = t0 * prob.z
        vs = slice(base + term.comp_off, base + term.comp_off + term.comp_dim)
        di = base + prob.dt_idx
        if term.kind == "quadratic":
            S[vs, vs] = 1.0
            S[vs, di] = 1.0
            S[di, di] = 1.0
        else:
            S[vs, di] = 1.0
    return S.tocsc()


def term_full_hessian(term, prob, Z):
    """get_full_hessian -- regularizers.jl:142-167, :295-313 (setindex! order matters: the
    (v,dt) block and the (dt,dt) entry are written after the (v,v) block and overwrite it when the
    timestep is itself inside `v`; not the case for any hot-path configuration)."""
    H = sp.lil_matrix((prob.n_vars, prob.n_vars))
    if term.kind == "mintime":
        return H.tocsc()
    if term.kind == "global_closure":  # global_objectives.jl:104-125, 303-345 (`.+=` over the listings, full block)
        Hd = np.zeros((prob.n_vars, prob.n_vars)) if prob.n_vars <= 4000 else None
        for idx, Q, p in term.listings(prob):
            Hl = Q * np.asarray(term.hess(Z[idx], p), dtype=np.float64).reshape(len(idx), len(idx))
            Hd[np.ix_(idx, idx)] += Hl
        return sp.csc_matrix(Hd)
    if term.kind == "knot_lowrank":
        return term_full_hessian(as_closure_objective(term), prob, Z)
    if term.kind == "knot_closure":  # hessian! per listed time (overwrite), triu (knot_point_objectives.jl:224-243)
        comps = np.asarray(term.comps)
        for i, t1 in enumerate(term.times1):
            v = _knot(Z, prob, t1 - 1)[comps]
            Hl = term.Qs[i] * np.asarray(term.hess(v, _param(term, i)), dtype=np.float64).reshape(len(comps), len(comps))
            idx = (t1 - 1) * prob.z + comps
            for a, ia in enumerate(idx):
                for b, ib in enumerate(idx):
                    H[ia, ib] = Hl[a, b]
        return sp.triu(H.tocsc()).tocsc()
    if term.kind == "knot_sqdist":  # hessian! per listed time (overwrite), triu (knot_point_objectives.jl:224-243)
        comps = np.asarray(term.comps)
        for i, t1 in enumerate(term.times1):
            idx = (t1 - 1) * prob.z + comps
            for a in idx:
                for b in idx:
                    H[a, b] = 0.0
            for a in idx:
                H[a, a] = 2.0 * term.Qs[i]
        return sp.triu(H.tocsc()).tocsc()
    for t0 in times0(term.times1, prob.N):
        zk = _knot(Z, prob, t0)
        dt = zk[prob.dt_idx]
        base = t0 * prob.z
        vi = np.arange(base + term.comp_off, base + term.comp_off + term.comp_dim)
        di = base + prob.dt_idx
        v = zk[term.comp_off:term.comp_off + term.comp_dim]
        if term.kind == "quadratic":
            r = v - _baseline(term, t0)
            for a in range(term.comp_dim):
                H[vi[a], vi[a]] = dt**2 * term.R[a]
            for a in range(term.comp_dim):
                H[vi[a], di] = 2 * dt * term.R[a] * r[a]
            H[di, di] = r @ (term.R * r)
        else:
            for a in range(term.comp_dim):
                H[vi[a], di] = term.R[a]
    return H.tocsc()


def objective_value(prob, Z):
    """CompositeObjective -- src/objectives/_objectives.jl:111-117."""
    return sum(prob.w(i) * term_value(t, prob, Z) for i, t in enumerate(prob.objectives))


def objective_gradient(prob, Z):
    """CompositeObjective gradient! -- _objectives.jl:119-128 (engine convention: zero-filled
    first)."""
    g = np.zeros(prob.n_vars)
    for i, t in enumerate(prob.objectives):
        term_gradient_accumulate(g, t, prob, Z, scale=prob.w(i))
    return g


def objective_full_hessian(prob, Z):
    """CompositeObjective get_full_hessian -- _objectives.jl:149-156."""
    H = sp.csc_matrix((prob.n_vars, prob.n_vars))
    for i, t in enumerate(prob.objectives):
        H = H + prob.w(i) * term_full_hessian(t, prob, Z)
    return H.tocsc()


# ----------------------------------------------------------------------------------------------
# Evaluator (MOI.AbstractNLPEvaluator surface) -- src/solvers/evaluator.jl
# ----------------------------------------------------------------------------------------------


def _findnz_colmajor(M):
    """(rows, cols) of stored entries in CSC order = Julia's findnz order."""
    M = M.tocsc()
    M.sort_indices()
    cols = np.repeat(np.arange(M.shape[1]), np.diff(M.indptr))
    return M.indices.astype(np.int64), cols.astype(np.int64)


class OracleEvaluator:
    """Restates `Evaluator` (evaluator.jl:66-288) and its MOI methods (:291-456, :474-647)."""

    def __init__(self, prob: Problem, eval_hessian: bool = True):
        self.prob = prob
        self.eval_hessian = eval_hessian
        Z0 = np.asarray(prob.Z0, dtype=np.float64)
        self.n_dynamics_constraints = sum(i.x_dim * prob.K for i in prob.integrators)
        self.n_nonlinear_constraints = sum(c.g_dim * len(c.times1) for c in prob.constraints)
        self.n_constraints = self.n_dynamics_constraints + self.n_nonlinear_constraints

        # Jacobian structure, evaluator.jl:119-149
        blocks = [integrator_jacobian_structure(i, prob) for i in prob.integrators]
        blocks += [constraint_jacobian(c, prob, Z0) for c in prob.constraints]
        if blocks:
            dg = sp.vstack(blocks, format="csc")
        else:
            dg = sp.csc_matrix((0, prob.n_vars))
        self.jac_rows, self.jac_cols = _findnz_colmajor(dg)

        # Hessian structure, evaluator.jl:151-209
        H = sp.csc_matrix((prob.n_vars, prob.n_vars))
        for _ in prob.integrators:
            H = H + integrator_hessian_structure(prob)
        for c in prob.constraints:
            H = H + abs(constraint_hessian(c, prob, Z0, np.ones(c.g_dim * len(c.times1))))
        for t in prob.objectives:
            H = H + term_hessian_structure(t, prob)
        r, c = _findnz_colmajor(H)
        keep = r <= c
        self.hess_rows, self.hess_cols = r[keep], c[keep]

        # offsets, evaluator.jl:211-227
        self.integrator_offsets = np.concatenate(
            [[0], np.cumsum([i.x_dim * prob.K for i in prob.integrators])]).astype(np.int64)
        self.constraint_offsets = (self.n_dynamics_constraints + np.concatenate(
            [[0], np.cumsum([c.g_dim * len(c.times1) for c in prob.constraints])])).astype(np.int64)

        self._jmap = {(int(r), int(c)): i for i, (r, c) in enumerate(zip(self.jac_rows, self.jac_cols))}
        self._hmap = {(int(r), int(c)): i for i, (r, c) in enumerate(zip(self.hess_rows, self.hess_cols))}

    # MOI.jacobian_structure / hessian_lagrangian_structure (1-based tuples), evaluator.jl:364,385
    def jacobian_structure1(self):
        return self.jac_rows + 1, self.jac_cols + 1

    def hessian_structure1(self):
        return self.hess_rows + 1, self.hess_cols + 1

    def eval_objective(self, Z):
        return objective_value(self.prob, Z)  # evaluator.jl:304-308

    def eval_objective_gradient(self, Z):
        return objective_gradient(self.prob, Z)  # evaluator.jl:310-318

    def eval_constraint(self, Z):
        """evaluator.jl:323-362."""
        g = np.zeros(self.n_constraints)
        for i, integ in enumerate(self.prob.integrators):
            o = self.integrator_offsets[i]
            g[o:o + integ.x_dim * self.prob.K] = integrator_evaluate(integ, self.prob, Z)
        for i, con in enumerate(self.prob.constraints):
            o = self.constraint_offsets[i]
            g[o:o + con.g_dim * len(con.times1)] = constraint_evaluate(con, self.prob, Z)
        return g

    def eval_constraint_jacobian(self, Z):
        """_fill_jacobian_values! -- evaluator.jl:491-551 (assignment, unknown positions dropped)."""
        out = np.zeros(len(self.jac_rows))
        comps = [(self.integrator_offsets[i], integrator_jacobian(integ, self.prob, Z))
                 for i, integ in enumerate(self.prob.integrators)]
        comps += [(self.constraint_offsets[i], constraint_jacobian(con, self.prob, Z))
                  for i, con in enumerate(self.prob.constraints)]
        for off, M in comps:
            r, c = _findnz_colmajor(M)
            M = M.tocsc()
            M.sort_indices()
            for rr, cc, vv in zip(r, c, M.data):
                idx = self._jmap.get((int(off + rr), int(cc)))
                if idx is not None:
                    out[idx] = vv
        return out

    def eval_hessian_lagrangian(self, Z, sigma, mu, skip_uu=False):
        """_fill_hessian_values! -- evaluator.jl:560-647 (accumulation, upper triangle only).
        skip_uu: see bilinear_block_hessian (large-state tests only)."""
        out = np.zeros(len(self.hess_rows))

        def scatter(M, scale=1.0):
            r, c = _findnz_colmajor(M)
            M = M.tocsc()
            M.sort_indices()
            for rr, cc, vv in zip(r, c, M.data):
                if rr <= cc:
                    idx = self._hmap.get((int(rr), int(cc)))
                    if idx is not None:
                        out[idx] += scale * vv

        for i, integ in enumerate(self.prob.integrators):
            o = self.integrator_offsets[i]
            scatter(integrator_hessian(integ, self.prob, Z, mu[o:o + integ.x_dim * self.prob.K], skip_uu))
        for i, con in enumerate(self.prob.constraints):
            o = self.constraint_offsets[i]
            scatter(constraint_hessian(con, self.prob, Z, mu[o:o + con.g_dim * len(con.times1)]))
        if sigma != 0:
            scatter(objective_full_hessian(self.prob, Z), sigma)
        return out

    def eval_constraint_jacobian_product(self, Z, w):
        """MOI.eval_constraint_jacobian_product -- evaluator.jl:406-430 (materialise, then y[row] += w[col]*v)."""
        y = np.zeros(self.n_constraints)
        np.add.at(y, self.jac_rows, w[self.jac_cols] * self.eval_constraint_jacobian(Z))
        return y

    def eval_constraint_jacobian_transpose_product(self, Z, w):
        """MOI.eval_constraint_jacobian_transpose_product -- evaluator.jl:432-456."""
        y = np.zeros(self.prob.n_vars)
        np.add.at(y, self.jac_cols, w[self.jac_rows] * self.eval_constraint_jacobian(Z))
        return y

    def row_bounds(self):
        """get_nonlinear_constraints -- src/solvers/solve.jl:30-65."""
        lo = np.zeros(self.n_constraints)
        hi = np.zeros(self.n_constraints)
        for i, con in enumerate(self.prob.constraints):
            if not con.equality:
                o = self.constraint_offsets[i]
                lo[o:o + con.g_dim * len(con.times1)] = -np.inf
        return lo, hi


# ----------------------------------------------------------------------------------------------
# Synthetic problem generators (own counter-based RNG: Julia's Xoshiro stream is not reproducible)
# ----------------------------------------------------------------------------------------------


def philox_normal(seed: int, count: int) -> np.ndarray:
    """Deterministic N(0,1) stream (numpy Philox bit generator, fixed seed)."""
    return np.random.Generator(np.random.Philox(seed)).standard_normal(count)


def make_scaled_problem(N, n, m=4, seed=42, with_constraint=False, skew=False, extra=None):
    """Shape of the reference's scaling generator benchmark/problem_utils.jl:49-77:
    components x[n], u[m], du[m], dt; [BilinearIntegrator(G,:x,:u), DerivativeIntegrator(:u,:du)];
    QuadraticRegularizer(:u, 1.0); G ~ randn, x ~ randn, u ~ 0.1 randn, du ~ randn, dt = 0.1."""
    z = n + 2 * m + 1
    r = philox_normal(seed, (m + 1) * n * n + N * (n + 2 * m))
    G = r[:(m + 1) * n * n].reshape(m + 1, n, n).transpose(0, 2, 1).copy()  # column-major fill
    if skew:
        G = (G - G.transpose(0, 2, 1)) / np.sqrt(2.0 * n)
    rest = r[(m + 1) * n * n:]
    x = rest[:n * N].reshape(N, n).T
    u = 0.1 * rest[n * N:n * N + m * N].reshape(N, m).T
    du = rest[n * N + m * N:].reshape(N, m).T
    data = np.vstack([x, u, du, np.full((1, N), 0.1)])
    Z0 = data.T.reshape(-1).copy()  # knot-major datavec
    prob = Problem(
        N=N, z=z, dt_idx=n + 2 * m,
        integrators=[BilinearIntegrator(0, n, n, m, G), DerivativeIntegrator(n, m, n + m)],
        objectives=[QuadraticRegularizer(n, m, np.ones(m))],
        Z0=Z0)
    if with_constraint:
        prob.constraints = [KnotConstraint("norm", list(range(n, n + m)), 1.0,
                                           list(range(2, N)), equality=False)]
    return prob


def make_l1_slack_problem(N, n, m=4, seed=42):
    """BASELINE configs[4] / SURVEY section 8d "C5" on the hot path: components x[n], u[m], du[m], s_du[m], dt
    (z = n + 3m + 1); [BilinearIntegrator(G,:x,:u), DerivativeIntegrator(:u,:du)]; the nonlinear inequality
    NonlinearKnotPointConstraint(u -> [norm(u) - 1], times = 2:N-1, <= 0) of test/test_snippets.jl:39-45; objective
    QuadraticRegularizer(:u, 1.0) + LinearRegularizer(:s_du, 1e-2) -- the penalty on the slack of an L1SlackConstraint
    (src/constraints/linear/l1_slack_constraint.jl:28), whose own rows |du| <= s_du are linear and go to MOI directly,
    not through the evaluator.  Same Philox stream and fill order as make_scaled_problem, then s_du = |du| + 0.1."""
    z = n + 3 * m + 1
    r = philox_normal(seed, (m + 1) * n * n + N * (n + 2 * m))
    G = r[:(m + 1) * n * n].reshape(m + 1, n, n).transpose(0, 2, 1).copy()
    rest = r[(m + 1) * n * n:]
    x = rest[:n * N].reshape(N, n).T
    u = 0.1 * rest[n * N:n * N + m * N].reshape(N, m).T
    du = rest[n * N + m * N:].reshape(N, m).T
    data = np.vstack([x, u, du, np.abs(du) + 0.1, np.full((1, N), 0.1)])
    prob = Problem(
        N=N, z=z, dt_idx=n + 3 * m,
        integrators=[BilinearIntegrator(0, n, n, m, G), DerivativeIntegrator(n, m, n + m)],
        objectives=[QuadraticRegularizer(n, m, np.ones(m)), LinearRegularizer(n + 2 * m, m, np.full(m, 1e-2))],
        Z0=data.T.reshape(-1).copy())
    prob.constraints = [KnotConstraint("norm", list(range(n, n + m)), 1.0, list(range(2, N)), equality=False)]
    return prob


def make_readme_problem(seed=7):
    """README.md:70-92 : x[2], u[1], dt; G = [-0.1 1; -1 -0.1] + u [0 1; 1 0]; QuadraticRegularizer(:u,1)."""
    N = 50
    r = philox_normal(seed, 3 * N)
    data = np.vstack([r[:2 * N].reshape(N, 2).T, r[2 * N:].reshape(1, N), np.full((1, N), 0.1)])
    G = np.array([[[-0.1, 1.0], [-1.0, -0.1]], [[0.0, 1.0], [1.0, 0.0]]])
    return Problem(N=N, z=4, dt_idx=3,
                   integrators=[BilinearIntegrator(0, 2, 2, 1, G)],
                   objectives=[QuadraticRegularizer(2, 1, np.ones(1))],
                   Z0=data.T.reshape(-1).copy())


def pauli_generators():
    """test/test_utils.jl:121-145."""
    Gx = np.array([[0, 0, 0, 1], [0, 0, 1, 0], [0, -1, 0, 0], [-1, 0, 0, 0]], dtype=float)
    Gy = np.array([[0, -1, 0, 0], [1, 0, 0, 0], [0, 0, 0, -1], [0, 0, 1, 0]], dtype=float)
    Gz = np.array([[0, 0, 1, 0], [0, 0, 0, -1], [-1, 0, 0, 0], [0, 1, 0, 0]], dtype=float)
    return Gx, Gy, Gz


def make_standard_problem(N=10, seed=3, omega=0.1):
    """The reference's "standard problem" (test/test_snippets.jl:29-54, test/test_utils.jl:113-178):
    components x[4], u[2], du[2], ddu[2], dt; Bilinear + 2 Derivative;
    TerminalObjective(x -> norm(x - goal)^2) + QuadReg(u) + QuadReg(du) + MinimumTime; ||u|| - 1 <= 0 at
    2:N-1.  Data are seeded here (the reference's are unseeded rand/randn)."""
    Gx, Gy, Gz = pauli_generators()
    G = np.stack([omega * Gz, Gx, Gy])
    rng = np.random.Generator(np.random.Philox(seed))
    x = 2 * rng.random((4, N)) - 1
    u = 0.1 * (2 * rng.random((2, N)) - 1)
    du = rng.standard_normal((2, N))
    ddu = rng.standard_normal((2, N))
    dt = 0.1 + 0.02 * rng.random((1, N))
    data = np.vstack([x, u, du, ddu, dt])
    return Problem(
        N=N, z=11, dt_idx=10,
        integrators=[BilinearIntegrator(0, 4, 4, 2, G), DerivativeIntegrator(4, 2, 6),
                     DerivativeIntegrator(6, 2, 8)],
        objectives=[KnotSqDistObjective([0, 1, 2, 3], [N], [1.0], np.array([[0.0, 1.0, 0.0, 0.0]])),
                    QuadraticRegularizer(4, 2, np.ones(2)), QuadraticRegularizer(6, 2, np.ones(2)),
                    MinimumTimeObjective(1.0)],
        weights=[1.0, 1.0, 1.0, 1.0],
        constraints=[KnotConstraint("norm", [4, 5], 1.0, list(range(2, N)), equality=False)],
        Z0=data.T.reshape(-1).copy())


def make_tdb_problem(N=6, n=4, m=2, order=1, seed=5, substeps=16, n_mods=2, with_derivative=False):
    """A TimeDependentBilinearIntegrator problem on random data: components x[n], u[m], t, dt (+ du[m] and a
    DerivativeIntegrator(u, du) with `with_derivative`), generator family with `n_mods` carrier terms
    (cos 1.7 t and sin 0.6 t), QuadraticRegularizer(u)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, N))
    u = 0.4 * rng.standard_normal((m, N))
    t = np.cumsum(np.full(N, 0.3))[None, :]
    dt = 0.25 + 0.1 * rng.random((1, N))
    G = rng.standard_normal((m + 1, n, n)) / np.sqrt(n / 4.0)
    mods = [("cos", 1.7, 0.5 * rng.standard_normal((m + 1, n, n)) / np.sqrt(n / 4.0)),
            ("sin", 0.6, 0.5 * rng.standard_normal((m + 1, n, n)) / np.sqrt(n / 4.0))][:n_mods]
    rows = [x, u]
    z = n + m + 2 + (m if with_derivative else 0)
    if with_derivative:
        rows.append(rng.standard_normal((m, N)))
    rows += [t, dt]
    data = np.vstack(rows)
    t_off, dt_idx = z - 2, z - 1
    integ = [TimeDependentBilinearIntegrator(0, n, n, m, t_off, G, mods, order, substeps).bind(z, dt_idx)]
    if with_derivative:
        integ.append(DerivativeIntegrator(n, m, n + m))
    return Problem(N=N, z=z, dt_idx=dt_idx, integrators=integ, objectives=[QuadraticRegularizer(n, m, np.ones(m))],
                   Z0=data.T.reshape(-1).copy())


def make_tdb_reference_carrier_problem(N=10, seed=4, omega=0.1, substeps=16):
    """The reference's own TimeDependentBilinearIntegrator test (time_dependent_bilinear_integrator.jl:259-269 with the
    generator of test/test_utils.jl:113-175, `bilinear_dynamics_and_trajectory(add_time = true)`): Pauli generators,
    G(a) = omega Gz + a_1 Gx + a_2 Gy, G_td(a, t) = G(a) + 0.1 cos(t) I, components x[4], u[2], du[2], ddu[2], dt, t
    (t = cumulative times, appended by add_component), spline_order 1 (the constructor's default), dt = 0.1, N = 10.
    Data are seeded here (the reference's are unseeded rand / randn)."""
    Gx, Gy, Gz = pauli_generators()
    G = np.stack([omega * Gz, Gx, Gy])
    H = np.zeros_like(G)
    H[0] = 0.1 * np.eye(4)
    rng = np.random.Generator(np.random.Philox(seed))
    x = 2 * rng.random((4, N)) - 1
    u = 0.1 * (2 * rng.random((2, N)) - 1)
    du = rng.standard_normal((2, N))
    ddu = rng.standard_normal((2, N))
    dt = np.full((1, N), 0.1)
    t = np.concatenate([[0.0], np.cumsum(dt[0])[:-1]])[None, :]   # get_times: t_1 = 0
    data = np.vstack([x, u, du, ddu, dt, t])
    integ = [TimeDependentBilinearIntegrator(0, 4, 4, 2, 11, G, [("cos", 1.0, H)], 1, substeps).bind(12, 10),
             DerivativeIntegrator(4, 2, 6), DerivativeIntegrator(6, 2, 8)]
    return Problem(N=N, z=12, dt_idx=10, integrators=integ, objectives=[QuadraticRegularizer(4, 2, np.ones(2))],
                   Z0=data.T.reshape(-1).copy())


def make_external_integrator_problem(N=7, seed=9):
    """Standard problem with one MORE integrator evaluated outside the engine, placed between the built-in ones: a
    nonlinear two-row defect that also reads z_{k+1} components other than the state (u_{k+1}) and the interval index,
        f_r = y_{k+1,r} - y_{k,r} - dt_k ( sin(y_{k,r}) u_{k,0} + w_k y_{k+1,r} u_{k+1,1} ),   y = ddu (comps 8, 9),
    so both halves of the Jacobian block and all three parts of the 2z x 2z Hessian block (diagonal z_k, diagonal
    z_{k+1}, cross) are non-trivial.  The shape a TimeDependentBilinearIntegrator has (:178-244)."""
    prob = make_standard_problem(N=N, seed=seed)
    z, y0, u0, dti = prob.z, 8, 4, prob.dt_idx
    w = 0.1 + 0.05 * np.arange(N)

    def f(zz, k):
        yk, yk1, dt = zz[y0:y0 + 2], zz[z + y0:z + y0 + 2], zz[dti]
        return yk1 - yk - dt * (np.sin(yk) * zz[u0] + w[k] * yk1 * zz[z + u0 + 1])

    def jac(zz, k):
        yk, yk1, dt = zz[y0:y0 + 2], zz[z + y0:z + y0 + 2], zz[dti]
        J = np.zeros((2, 2 * z))
        for r in range(2):
            J[r, y0 + r] = -1.0 - dt * np.cos(yk[r]) * zz[u0]
            J[r, u0] = -dt * np.sin(yk[r])
            J[r, dti] = -(np.sin(yk[r]) * zz[u0] + w[k] * yk1[r] * zz[z + u0 + 1])
            J[r, z + y0 + r] = 1.0 - dt * w[k] * zz[z + u0 + 1]
            J[r, z + u0 + 1] = -dt * w[k] * yk1[r]
        return J

    def hess(zz, k, mu):
        yk, yk1, dt = zz[y0:y0 + 2], zz[z + y0:z + y0 + 2], zz[dti]
        H = np.zeros((2 * z, 2 * z))

        def sym(a, b, v):
            H[a, b] += v
            if a != b:
                H[b, a] += v
        for r in range(2):
            sym(y0 + r, y0 + r, mu[r] * dt * np.sin(yk[r]) * zz[u0])
            sym(y0 + r, u0, -mu[r] * dt * np.cos(yk[r]))
            sym(y0 + r, dti, -mu[r] * np.cos(yk[r]) * zz[u0])
            sym(u0, dti, -mu[r] * np.sin(yk[r]))
            sym(dti, z + y0 + r, -mu[r] * w[k] * zz[z + u0 + 1])
            sym(dti, z + u0 + 1, -mu[r] * w[k] * yk1[r])
            sym(z + y0 + r, z + u0 + 1, -mu[r] * dt * w[k])
        return H

    prob.integrators = [prob.integrators[0], ClosureIntegrator(f, jac, hess, 2)] + prob.integrators[1:]
    return prob


def make_global_problem(N=7, seed=13, gd=3):
    """Standard problem with global variables (traj.global_data, NamedTrajectories) and the three Global* kinds of the
    reference: GlobalObjective Q l(g), GlobalKnotPointObjective sum_i Q_i l([x_t; g]) listed at repeated knots, and a
    two-row NonlinearGlobalConstraint, the latter placed BETWEEN the knot constraints."""
    prob = make_standard_problem(N=N, seed=seed)
    rng = np.random.Generator(np.random.Philox(seed + 50))
    prob.gd = gd
    prob.Z0 = np.concatenate([prob.Z0, 0.5 + rng.random(gd)])

    def lg(v, p):
        return np.cosh(v[0]) + v[1] ** 2 * v[2]

    def lg_grad(v, p):
        return np.array([np.sinh(v[0]), 2.0 * v[1] * v[2], v[1] ** 2])

    def lg_hess(v, p):
        return np.array([[np.cosh(v[0]), 0.0, 0.0], [0.0, 2.0 * v[2], 2.0 * v[1]], [0.0, 2.0 * v[1], 0.0]])

    # l([x1, u0, g2, g0], p) = g2 x1^2 + sin(u0 g0) + p0 g0
    def lk(v, p):
        return v[2] * v[0] ** 2 + np.sin(v[1] * v[3]) + p[0] * v[3]

    def lk_grad(v, p):
        c = np.cos(v[1] * v[3])
        return np.array([2.0 * v[2] * v[0], v[3] * c, v[0] ** 2, v[1] * c + p[0]])

    def lk_hess(v, p):
        sn, c = np.sin(v[1] * v[3]), np.cos(v[1] * v[3])
        H = np.zeros((4, 4))
        H[0, 0] = 2.0 * v[2]
        H[0, 2] = H[2, 0] = 2.0 * v[0]
        H[1, 1] = -v[3] ** 2 * sn
        H[1, 3] = H[3, 1] = c - v[1] * v[3] * sn
        H[3, 3] = -v[1] ** 2 * sn
        return H

    def gc(v):
        return np.array([v[0] * v[1] - 0.3, np.exp(v[1]) - 2.0])

    def gc_jac(v):
        return np.array([[v[1], v[0]], [0.0, np.exp(v[1])]])

    def gc_hess(v, mu):
        return mu[0] * np.array([[0.0, 1.0], [1.0, 0.0]]) + mu[1] * np.array([[0.0, 0.0], [0.0, np.exp(v[1])]])

    kt = [2, 4, 4, N]
    prob.objectives += [GlobalClosureObjective(lg, lg_grad, lg_hess, [], [0, 1, 2], [], [1.3]),
                        GlobalClosureObjective(lk, lk_grad, lk_hess, [1, 4], [2, 0], kt, list(0.5 + rng.random(len(kt))),
                                               params=[rng.standard_normal(1) for _ in kt])]
    prob.weights = list(prob.weights) + [0.8, 1.1]
    prob.constraints = [prob.constraints[0], GlobalClosureConstraint(gc, gc_jac, gc_hess, [2, 0], 2, equality=False),
                        KnotConstraint("sqnorm", [6, 7], 0.5, [1, N - 1], equality=True)]
    return prob


def ket_fidelity_factor(goal_iso):
    """A (2 x 2n) with ||A psi~||^2 = |<goal|psi>|^2 for iso vectors psi~ = [Re psi; Im psi]."""
    g = np.asarray(goal_iso, dtype=np.float64)
    n = g.size // 2
    gr, gi = g[:n], g[n:]
    return np.vstack([np.concatenate([gr, gi]), np.concatenate([-gi, gr])])


def make_ket_problem(N=8, seed=11):
    """Standard (Pauli) problem with the TerminalObjective replaced by the ket infidelity |1 - |<goal|psi_N>|^2|
    (x[4] is the iso-vector of a qubit state), plus the same loss listed at interior knots with weights, one knot
    twice, and a state scaled past F = 1 so that both signs of (1 - F) occur."""
    prob = make_standard_problem(N=N, seed=seed)
    goal = np.array([0.6, 0.0, 0.0, 0.8])  # (0.6 + 0i, 0 + 0.8i)
    A = ket_fidelity_factor(goal)
    Zk = prob.Z0[:prob.z * N].reshape(N, prob.z)
    Zk[2, 0:4] = 1.7 * goal  # F = 1.7^2 > 1 at knot 3
    prob.objectives[0] = LowRankInfidelityObjective([0, 1, 2, 3], [N], [1.0], A)
    prob.objectives.append(LowRankInfidelityObjective([0, 1, 2, 3], [2, 3, 3, 5], [0.5, 1.5, 0.25, 2.0], A))
    prob.weights = list(prob.weights) + [0.3]
    return prob


def make_closure_problem(N=9, seed=5):
    """The standard problem plus closure-based knot terms (the reference's NonlinearKnotPointConstraint /
    KnotPointObjective take arbitrary closures, knot_point_constraint.jl:27-107, knot_point_objectives.jl:65-121):
    a 2-output g over (u0, u1, dt) listed at repeated knots, placed BETWEEN two built-in constraints, and a loss over
    an unsorted component list (x1, du0, x3).  Analytic derivatives stand in for ForwardDiff."""
    prob = make_standard_problem(N=N, seed=seed)
    rng = np.random.Generator(np.random.Philox(seed + 100))

    def g(v, p):
        return np.array([v[0] * v[1] - p[0], np.sin(5.0 * v[2]) + v[0] ** 2 - p[1]])

    def g_jac(v, p):
        return np.array([[v[1], v[0], 0.0], [2.0 * v[0], 0.0, 5.0 * np.cos(5.0 * v[2])]])

    def g_hess(v, p, mu):
        return (mu[0] * np.array([[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
                + mu[1] * np.array([[2.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, -25.0 * np.sin(5.0 * v[2])]]))

    def l(v, p):
        return np.exp(v[0]) * v[1] ** 2 + np.cos(v[2] - p[0])

    def l_grad(v, p):
        e = np.exp(v[0])
        return np.array([e * v[1] ** 2, 2.0 * e * v[1], -np.sin(v[2] - p[0])])

    def l_hess(v, p):
        e = np.exp(v[0])
        return np.array([[e * v[1] ** 2, 2.0 * e * v[1], 0.0], [2.0 * e * v[1], 2.0 * e, 0.0], [0.0, 0.0, -np.cos(v[2] - p[0])]])

    ct = [2, 3, 5, 5, N]
    closure_con = ClosureKnotConstraint(g, g_jac, g_hess, [4, 5, 10], ct, 2, params=[rng.standard_normal(2) for _ in ct],
                                        equality=False)
    prob.constraints = [prob.constraints[0], closure_con,
                        KnotConstraint("sqnorm", [6, 7], 0.5, [1, N - 1], equality=True)]
    ot = [1, 4, 4, N]
    prob.objectives.append(ClosureKnotObjective(l, l_grad, l_hess, [1, 6, 3], ot, list(0.5 + rng.random(len(ot))),
                                                params=[rng.standard_normal(1) for _ in ot]))
    prob.weights = list(prob.weights) + [0.7]
    return prob


NAMED_TRAJECTORY_TYPE_1 = np.array([
    [1.0, 0.957107, 0.853553, 0.75, 0.707107],
    [0.0, 0.103553, 0.353553, 0.603553, 0.707107],
    [0.0, 0.103553, 0.146447, 0.103553, 1.38778e-17],
    [0.0, -0.25, -0.353553, -0.25, -1.52656e-16],
    [0.0, 0.103553, 0.353553, 0.603553, 0.707107],
    [1.0, 0.75, 0.146447, -0.457107, -0.707107],
    [0.0, -0.25, -0.353553, -0.25, -1.249e-16],
    [0.0, 0.603553, 0.853553, 0.603553, 4.16334e-16],
    [0.0, -0.243953, 0.959151, -0.665253, 0.0],
    [0.0, 0.0139165, 0.668917, 0.625329, 0.0],
    [0.00393491, 0.0240775, -0.00942396, 0.00329391, 0.00941354],
    [-0.00223794, -0.0105816, 0.00328457, 0.0204239, 0.0253415],
    [0.0058186, 0.00686586, -0.00422555, 0.00442631, 0.000319156],
    [-0.00134597, -0.00120682, 0.0114915, 0.00189333, -0.0251649],
    [0.2, 0.2, 0.2, 0.2, 0.2],
])
"""The literal 15x5 data matrix of `named_trajectory_type_1` (test/test_utils.jl:57-82): a data
fixture of the reference's own tests (components U[8], a[2], da[2], dda[2], dt[1])."""


def make_type1_derivative_problem():
    """`DerivativeIntegrator(:a, :da, traj)` on named_trajectory_type_1
    (derivative_integrator.jl:118-123)."""
    data = NAMED_TRAJECTORY_TYPE_1
    return Problem(N=5, z=15, dt_idx=14, integrators=[DerivativeIntegrator(8, 2, 10)],
                   objectives=[], Z0=data.T.reshape(-1).copy())
