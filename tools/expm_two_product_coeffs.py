#!/usr/bin/env python3
"""Coefficients of the two-product evaluation of the degree-16 Taylor polynomial of exp (csrc/dto_kernels.h, EXPM2_*).

With B, B^2, B^3, B^4 in hand (the engine gets them from the generator subspace), Paterson-Stockmeyer
needs three more products for T_16(B) = sum_{r<=16} B^r/r!.  Two suffice:

    Y  = B^4 K(B)                       K, Pa, Pb, Pc polynomials of degree <= 4
    T_16(B) = (Y + Pa(B)) (Y + Pb(B)) + Pc(B)

(the scheme of Sastre, Ibanez, Defez & Ruiz, "Boosting the computation of the matrix exponential", 2019, adapted to free low
powers: 20 unknowns, 17 equations).  This script solves the polynomial identity with a minimum-norm Gauss-Newton iteration in
60-digit arithmetic from the guess Y + Pa ~ Y + Pb ~ exp(B/2), so that the two factors have no cancellation, checks the
identity to 1e-50, measures the rounding error of the evaluation in double against mpmath's expm, and prints the constants."""
import mpmath as mp
import numpy as np
from math import factorial

mp.mp.dps = 60
c = [mp.mpf(1) / mp.factorial(r) for r in range(17)]


def polymul(p, q):
    out = [mp.mpf(0)] * (len(p) + len(q) - 1)
    for i, a in enumerate(p):
        for j, b in enumerate(q):
            out[i + j] += a * b
    return out


def polyadd(p, q):
    n = max(len(p), len(q))
    return [(p[i] if i < len(p) else 0) + (q[i] if i < len(q) else 0) for i in range(n)]


def residual(v):
    k, a, b, pc = v[0:5], v[5:10], v[10:15], v[15:20]
    Y = [mp.mpf(0)] * 4 + list(k)
    E = polyadd(polymul(polyadd(Y, a), polyadd(Y, b)), pc)
    return [E[r] - c[r] for r in range(17)]


def solve():
    h = [mp.mpf(1) / (2 ** r) / mp.factorial(r) for r in range(9)]
    k = [h[4 + j] for j in range(5)]
    k[0] = h[4] / 2
    a = [h[j] for j in range(5)]
    a[4] = h[4] / 2
    b = list(a)
    a = [a[j] * (1 + mp.mpf("0.1") * (j > 0)) for j in range(5)]  # Pa = Pb has too few parameters: start off the diagonal
    b = [b[j] * (1 - mp.mpf("0.1") * (j > 0)) for j in range(5)]
    v = mp.matrix(k + a + b + [mp.mpf(0)] * 5)
    for _ in range(60):
        f0 = residual(list(v))
        J = mp.matrix(17, 20)
        for j in range(20):
            d = mp.mpf(10) ** (-30) * max(abs(v[j]), mp.mpf(10) ** -8)
            w = list(v)
            w[j] += d
            f1 = residual(w)
            for i in range(17):
                J[i, j] = (f1[i] - f0[i]) / d
        D = mp.diag([max(abs(v[j]), mp.mpf(10) ** -9) for j in range(20)])  # minimum norm in relative terms
        Js = J * D
        v = v - D * (Js.T * mp.lu_solve(Js * Js.T, mp.matrix(f0)))
        if max(abs(x) for x in f0) < mp.mpf(10) ** -50:
            break
    assert max(abs(x) for x in residual(list(v))) < mp.mpf(10) ** -50
    return v


def check(vd):
    k, a, b, pc = vd[0:5], vd[5:10], vd[10:15], vd[15:20]
    rng = np.random.default_rng(0)
    worst_new = worst_ps = 0.0
    for n in (8, 24):
        for kind in ("gauss", "skew", "neg", "upper"):
            for _ in range(3):
                A = rng.standard_normal((n, n))
                if kind == "skew":
                    A = A - A.T
                if kind == "neg":
                    A = -np.abs(A)
                if kind == "upper":
                    A = 3 * np.triu(A) + 0.1 * A
                A *= 0.78 / np.abs(A).sum(0).max()
                P = [np.eye(n), A]
                for _r in range(3):
                    P.append(P[-1] @ A)
                Y = P[4] @ sum(k[j] * P[j] for j in range(5))
                new = (Y + sum(a[j] * P[j] for j in range(5))) @ (Y + sum(b[j] * P[j] for j in range(5))) + sum(pc[j] * P[j] for j in range(5))
                X = sum(P[j] / factorial(12 + j) for j in range(5))
                for base in (8, 4, 0):
                    X = P[4] @ X + sum(P[j] / factorial(base + j) for j in range(4))
                E = np.array(mp.expm(mp.matrix(A.tolist()), method="taylor").tolist(), dtype=float)
                worst_new = max(worst_new, np.abs(new - E).max() / np.abs(E).max())
                worst_ps = max(worst_ps, np.abs(X - E).max() / np.abs(E).max())
    return worst_new, worst_ps


if __name__ == "__main__":
    v = solve()
    vd = [float(x) for x in v]
    wn, wp = check(vd)
    print("// max relative error at ||B||_1 = 0.78 over 24 random matrices: two-product %.1e, Paterson-Stockmeyer %.1e" % (wn, wp))
    for name, off in (("K", 0), ("A", 5), ("B", 10), ("C", 15)):
        print("constexpr double EXPM2_%s[5] = {%s};" % (name, ", ".join(repr(vd[off + j]) for j in range(5))))
