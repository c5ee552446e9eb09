"""Synthetic benchmark problems of BASELINE.json's configurations.

Shape of the reference's scaling generator ``make_scaled_problem``
(benchmark/problem_utils.jl:49-77): components x[n], u[m], du[m], dt (z = n+2m+1),
[BilinearIntegrator(G,:x,:u), DerivativeIntegrator(:u,:du)], QuadraticRegularizer(:u, 1.0),
G0, G_j ~ randn(n,n), x ~ randn, u ~ 0.1 randn, du ~ randn, dt = 0.1.  Julia's Xoshiro stream is not
reproducible outside Julia, so a counter-based Philox stream (seed 42) is used instead."""
from __future__ import annotations

import numpy as np

from .problem import BilinearIntegrator, DerivativeIntegrator, DirectTrajOptProblem, QuadraticRegularizer
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
