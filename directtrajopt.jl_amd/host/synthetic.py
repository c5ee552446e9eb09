"""Synthetic benchmark problems of BASELINE.json's configurations.

Shape of the reference's scaling generator ``make_scaled_problem``
(benchmark/problem_utils.jl:49-77): components x[n], u[m], du[m], dt (z = n+2m+1),
[BilinearIntegrator(G,:x,:u), DerivativeIntegrator(:u,:du)], QuadraticRegularizer(:u, 1.0),
G0, G_j ~ randn(n,n), x ~ randn, u ~ 0.1 randn, du ~ randn, dt = 0.1.  Julia's Xoshiro stream is not
reproducible outside Julia, so a counter-based Philox stream (seed 42) is used instead."""
from __future__ import annotations

import numpy as np

from .problem import (BilinearIntegrator, DerivativeIntegrator, DirectTrajOptProblem, LinearRegularizer,
                      NonlinearKnotPointConstraint, QuadraticRegularizer)
from .trajectory import NamedTrajectory


def scaled_problem_arrays(N, n, m=4, seed=42):
    rng = np.random.Generator(np.random.Philox(seed))
    r = rng.standard_normal((m + 1) * n * n + N * (n + 2 * m))
    G = r[:(m + 1) * n * n].reshape(m + 1, n, n).transpose(0, 2, 1).copy()  # column-major fill order
    rest = r[(m + 1) * n * n:]
    x = rest[:n * N].reshape(N, n).T
    u = 0.1 * rest[n * N:n * N + m * N].reshape(N, m).T
    du = rest[n * N + m * N:].reshape(N, m).T
    return G, x, u, du


def make_scaled_problem(N, n, m=4, seed=42):
    G, x, u, du = scaled_problem_arrays(N, n, m, seed)
    traj = NamedTrajectory({"x": x, "u": u, "du": du, "dt": np.full((1, N), 0.1)}, timestep="dt")
    integrators = [BilinearIntegrator(G, "x", "u", traj), DerivativeIntegrator("u", "du", traj)]
    J = QuadraticRegularizer("u", traj, 1.0)
    return DirectTrajOptProblem(traj, J, integrators)


def make_l1_slack_problem(N, n, m=4, seed=42):
    """BASELINE configs[4] workload on the evaluator path ("C5"): components
    x[n], u[m], du[m], s_du[m], dt (z = n + 3m + 1); [BilinearIntegrator(G,:x,:u), DerivativeIntegrator(:u,:du)];
    NonlinearKnotPointConstraint(u -> [norm(u) - 1], :u, times = 2:N-1, equality = false) as in the reference's
    test/test_snippets.jl:39-45; QuadraticRegularizer(:u, 1.0) + LinearRegularizer(:s_du, 1e-2), the penalty on the slack
    of an L1SlackConstraint (src/constraints/linear/l1_slack_constraint.jl:28 -- its own rows |du| <= s_du are linear and
    are handed to MOI once, they never reach the evaluator)."""
    G, x, u, du = scaled_problem_arrays(N, n, m, seed)
    traj = NamedTrajectory({"x": x, "u": u, "du": du, "s_du": np.abs(du) + 0.1, "dt": np.full((1, N), 0.1)}, timestep="dt")
    integrators = [BilinearIntegrator(G, "x", "u", traj), DerivativeIntegrator("u", "du", traj)]
    J = QuadraticRegularizer("u", traj, 1.0) + LinearRegularizer("s_du", traj, 1e-2)
    con = NonlinearKnotPointConstraint("norm", "u", traj, c=1.0, equality=False, times=range(2, N))
    return DirectTrajOptProblem(traj, J, integrators, constraints=[con])
