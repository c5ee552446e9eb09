"""The constants of the two polynomial forms of the matrix exponential (csrc/dto_kernels.h, EXPM2_* / EXPM3_*) are checked
on the CPU against what they claim: as exact rationals (the doubles the kernels use) the two-product form reproduces the
Taylor coefficients of exp up to degree 16, the three-product form up to order 26, to rounding; the backward-error radii in
the header do not exceed the radii recomputed from those very coefficients (tools/expm_three_product_coeffs.py)."""
import os
import re
import sys
from fractions import Fraction
from math import factorial

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "directtrajopt.jl_amd", "csrc", "dto_kernels.h")


def _constants():
    text = open(HEADER).read()
    out = {}
    for name, body in re.findall(r"constexpr double (EXPM[23]_[A-Z]+)\[5\] = \{([^}]*)\};", text):
        out[name] = [Fraction(float(x)) for x in body.split(",")]
    m = re.search(r"EXPM3_AL = ([-0-9.e]+), EXPM3_BE = ([-0-9.e]+);", text)
    out["EXPM3_AL"], out["EXPM3_BE"] = Fraction(float(m.group(1))), Fraction(float(m.group(2)))
    out["THETA_16"] = float(re.search(r"THETA_16 = ([0-9.]+);", text).group(1))
    out["THETA_3P"] = float(re.search(r"THETA_3P = ([0-9.]+);", text).group(1))
    return out


def _mul(p, q):
    r = [Fraction(0)] * (len(p) + len(q) - 1)
    for i, a in enumerate(p):
        for j, b in enumerate(q):
            r[i + j] += a * b
    return r


def _add(p, q, s=Fraction(1)):
    n = max(len(p), len(q))
    return [(p[i] if i < len(p) else 0) + s * (q[i] if i < len(q) else 0) for i in range(n)]


def test_two_product_form_is_the_degree_16_taylor_polynomial():
    c = _constants()
    Y = [Fraction(0)] * 4 + c["EXPM2_K"]
    T = _add(_mul(_add(Y, c["EXPM2_A"]), _add(Y, c["EXPM2_B"])), c["EXPM2_C"])
    assert len(T) == 17
    for r, t in enumerate(T):
        assert abs(float(t * factorial(r)) - 1.0) < 1e-12, (r, float(t * factorial(r)))


def test_three_product_form_matches_exp_to_order_26():
    c = _constants()
    Y1 = [Fraction(0)] * 4 + c["EXPM3_K"]
    Y2 = _mul(_add(Y1, c["EXPM3_A"]), _add(Y1, c["EXPM3_B"]))
    L = _add(_add(Y2, Y1, c["EXPM3_AL"]), c["EXPM3_C"])
    R = _add(_add(Y2, Y1, c["EXPM3_BE"]), c["EXPM3_D"])
    P = _add(_mul(L, R), c["EXPM3_E"])
    assert len(P) == 33
    for r in range(27):
        assert abs(float(P[r] * factorial(r)) - 1.0) < 1e-10, (r, float(P[r] * factorial(r)))
    # beyond the order the coefficients stay of the size of Taylor's (that is what keeps the radius large)
    assert all(abs(float(P[r] * factorial(r))) < 2.5 for r in range(27, 33))
    # the engine's second product leaves A^4 out of its epilogue
    assert c["EXPM3_A"][4] == 0 and c["EXPM3_C"][4] == 0 and c["EXPM3_D"][4] == 0


def test_radii_in_the_header_are_not_larger_than_the_recomputed_ones():
    import mpmath as mp
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import expm_three_product_coeffs as T3
    c = _constants()
    mp.mp.dps = 100
    t16, first = T3.theta_from_poly([mp.mpf(1) / mp.factorial(k) for k in range(17)])
    assert first == 17 and abs(float(t16) - c["THETA_16"]) < 1e-7
    v = [mp.mpf(x.numerator) / mp.mpf(x.denominator) for x in
         c["EXPM3_K"] + c["EXPM3_A"] + c["EXPM3_B"] + [c["EXPM3_AL"], c["EXPM3_BE"]] + c["EXPM3_C"] + c["EXPM3_D"] + c["EXPM3_E"]]
    coefs = T3.build(v)
    # rounding the constants to doubles leaves O(1e-16) low-order terms in log(e^-x r(x)); they are rounding, not truncation
    h_tail = [coefs[r] if r > 26 else mp.mpf(1) / mp.factorial(r) for r in range(33)]
    th, first = T3.theta_from_poly(h_tail)
    assert first == 27
    assert c["THETA_3P"] <= float(th) and float(th) - c["THETA_3P"] < 1e-3
