"""Host-side evaluation of closure-based knot terms.

In the Julia integration the reference's own ForwardDiff code produces these blocks
(knot_point_constraint.jl:235-294, knot_point_objectives.jl:173-243) and the shim passes them to
``dto_set_external``; this module is the Python mirror's stand-in for ForwardDiff: analytic derivatives
when the user supplies them, otherwise complex-step differentiation (exact to rounding for analytic NumPy
expressions) with a real finite-difference fallback.  The engine only PLACES the blocks -- no arithmetic
of the device path runs here."""
from __future__ import annotations

import warnings

import numpy as np

_H = 1e-30


def _stencil(f, v, d, h, out_dim):
    """4th-order central difference of f along direction d."""
    F = lambda t: np.asarray(f(v + t * d), dtype=np.float64).reshape(out_dim)
    return (8.0 * (F(h) - F(-h)) - (F(2 * h) - F(-2 * h))) / (12.0 * h)


def _first_derivative(f, v, out_dim):
    """d f / d v  as an (out_dim, len(v)) array: complex step, verified against a real difference along one
    random direction (closures that call abs/norm or cast to float are not analytic: their complex step is
    silently wrong), else 4th-order central differences."""
    v = np.asarray(v, dtype=np.float64)
    J = np.empty((out_dim, v.size))
    ok = True
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            for q in range(v.size):
                vc = v.astype(np.complex128)
                vc[q] += 1j * _H
                J[:, q] = np.imag(np.asarray(f(vc), dtype=np.complex128).reshape(out_dim)) / _H
    except Exception:
        ok = False
    if ok:
        d = np.random.default_rng(12345).standard_normal(v.size)
        d /= np.linalg.norm(d)
        want = _stencil(f, v, d, 1e-3 * (1.0 + np.abs(v).max()), out_dim)
        ok = bool(np.all(np.isfinite(J))) and np.linalg.norm(J @ d - want) <= 1e-6 * (1.0 + np.linalg.norm(want))
    if not ok:
        for q in range(v.size):
            e = np.zeros(v.size)
            e[q] = 1.0
            J[:, q] = _stencil(f, v, e, 1e-3 * (1.0 + abs(v[q])), out_dim)
    return J


def jacobian(g, v, p, g_dim, jac=None):
    if jac is not None:
        return np.asarray(jac(v, p), dtype=np.float64).reshape(g_dim, len(v))
    return _first_derivative(lambda x: g(x, p), v, g_dim)


def gradient(l, v, p, grad=None):
    if grad is not None:
        return np.asarray(grad(v, p), dtype=np.float64).reshape(len(v))
    return _first_derivative(lambda x: np.atleast_1d(l(x, p)), v, 1)[0]


def hessian(f, v, grad_f=None, hess=None):
    """Hessian of the scalar function f at v: analytic if given, else central differences of the (analytic or
    complex-step) gradient, symmetrised."""
    v = np.asarray(v, dtype=np.float64)
    if hess is not None:
        return np.asarray(hess(v), dtype=np.float64).reshape(v.size, v.size)
    g = grad_f if grad_f is not None else (lambda x: _first_derivative(lambda y: np.atleast_1d(f(y)), x, 1)[0])
    Hm = np.empty((v.size, v.size))
    for q in range(v.size):
        h = 1e-5 * (1.0 + abs(v[q]))
        vp, vm = v.copy(), v.copy()
        vp[q] += h
        vm[q] -= h
        Hm[:, q] = (np.asarray(g(vp), dtype=np.float64) - np.asarray(g(vm), dtype=np.float64)) / (2 * h)
    return 0.5 * (Hm + Hm.T)
