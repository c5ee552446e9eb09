"""GPU parity on the edge cases of the path: term variety (LinearRegularizer, baselines, time subsets,
weights, sqnorm / equality constraints, duplicate constraint times), component orders that drop the
(v,dt) Hessian cross terms, global variables, several bilinear integrators, zero-drive problems,
minimum sizes, non-finite iterates."""
import os

import numpy as np
import pytest

import dto_oracle as O
from helpers import rel_err, run_all, to_engine

pytestmark = pytest.mark.gpu


def _compare(p, Z=None, hessian=True, tol=1e-10, tol_h=1e-8):
    import dto_amd
    ev_o = O.OracleEvaluator(p)
    ev = dto_amd.Evaluator(to_engine(p), eval_hessian=hessian)
    try:
        r, c = ev.jacobian_structure(); r1, c1 = ev_o.jacobian_structure1()
        assert np.array_equal(r, r1) and np.array_equal(c, c1)
        r, c = ev.hessian_lagrangian_structure(); r1, c1 = ev_o.hessian_structure1()
        assert np.array_equal(r, r1) and np.array_equal(c, c1)
        Z = p.Z0.copy() if Z is None else Z
        mu = np.random.default_rng(7).standard_normal(ev_o.n_constraints)
        out = run_all(ev, p, Z, mu, sigma=1.3, hessian=hessian)
        assert rel_err(out["f"], ev_o.eval_objective(Z)) <= tol
        assert rel_err(out["grad"], ev_o.eval_objective_gradient(Z)) <= tol
        assert rel_err(out["cons"], ev_o.eval_constraint(Z)) <= tol
        assert rel_err(out["jac"], ev_o.eval_constraint_jacobian(Z)) <= tol
        if hessian:
            assert rel_err(out["hess"], ev_o.eval_hessian_lagrangian(Z, 1.3, mu)) <= tol_h
        return out
    finally:
        ev.close()


def test_objective_term_variety():
    N, n, m = 7, 5, 2
    p = O.make_scaled_problem(N, n, m, seed=31)
    rng = np.random.default_rng(0)
    p.objectives = [
        O.QuadraticRegularizer(n, m, np.array([0.5, 2.0]), baseline=rng.standard_normal((m, N)), times1=[1, 3, 4, 7]),
        O.LinearRegularizer(n + m, m, np.array([1e-2, 3e-2])),
        O.MinimumTimeObjective(2.5),
        O.QuadraticRegularizer(0, n, rng.random(n)),
        O.KnotSqDistObjective([0, 2, n + 1], [1, 4, 4, N], [1.0, 2.0, 0.5, 3.0], rng.standard_normal((4, 3))),  # a knot listed twice
        O.KnotSqDistObjective(list(range(n)), [N], [1.5]),  # norm(x)^2 at the last knot, no target
    ]
    p.weights = [1.0, 0.3, 2.0, 0.1, 0.7, 1.0]
    _compare(p)


def test_constraint_kinds_equality_and_duplicate_times():
    N, n, m = 6, 4, 2
    p = O.make_scaled_problem(N, n, m, seed=8)
    p.constraints = [
        O.KnotConstraint("norm", [n, n + 1], 1.0, [2, 3, 3, 5], equality=False),   # a knot listed twice
        O.KnotConstraint("sqnorm", [0, 1, 2], 0.5, [1, N], equality=True),
        O.KnotConstraint("norm", [n + m, n + m + 1, n], 2.0, [4], equality=False),  # unordered comps
    ]
    _compare(p)


def test_timestep_before_regularized_component_drops_cross_terms():
    """(v,dt) Hessian entries of the regularizers survive only if dt follows v (regularizers.jl:160,
    evaluator.jl:637): here dt is the FIRST component."""
    N, n, m = 5, 3, 2
    rng = np.random.default_rng(3)
    z = 1 + n + 2 * m
    data = np.vstack([0.1 + 0.05 * rng.random((1, N)), rng.standard_normal((n, N)), 0.2 * rng.standard_normal((m, N)),
                      rng.standard_normal((m, N))])
    G = rng.standard_normal((m + 1, n, n))
    p = O.Problem(N=N, z=z, dt_idx=0,
                  integrators=[O.DerivativeIntegrator(1 + n, m, 1 + n + m), O.BilinearIntegrator(1, n, 1 + n, m, G)],
                  objectives=[O.QuadraticRegularizer(1 + n, m, np.ones(m)), O.LinearRegularizer(1 + n + m, m, np.ones(m))],
                  Z0=data.T.reshape(-1).copy())
    _compare(p)


def test_two_bilinear_integrators_and_global_columns():
    N = 5
    rng = np.random.default_rng(11)
    # comps: x[3], y[2], u[2], dt ; globals: 3 (no hot-path term touches them, columns still exist)
    data = np.vstack([rng.standard_normal((3, N)), rng.standard_normal((2, N)), 0.3 * rng.standard_normal((2, N)),
                      np.full((1, N), 0.15)])
    p = O.Problem(N=N, z=8, dt_idx=7, gd=3,
                  integrators=[O.BilinearIntegrator(0, 3, 5, 2, rng.standard_normal((3, 3, 3))),
                               O.BilinearIntegrator(3, 2, 5, 2, rng.standard_normal((3, 2, 2)))],
                  objectives=[O.QuadraticRegularizer(5, 2, np.ones(2))],
                  Z0=np.concatenate([data.T.reshape(-1), rng.standard_normal(3)]))
    _compare(p)


def test_minimum_sizes():
    rng = np.random.default_rng(2)
    data = np.vstack([rng.standard_normal((1, 2)), rng.standard_normal((1, 2)), np.full((1, 2), 0.2)])
    p = O.Problem(N=2, z=3, dt_idx=2, integrators=[O.BilinearIntegrator(0, 1, 1, 1, rng.standard_normal((2, 1, 1)))],
                  objectives=[], Z0=data.T.reshape(-1).copy())
    _compare(p)


def test_zero_controls_and_zero_state():
    p = O.make_scaled_problem(5, 6, 2, seed=4)
    Z = p.Z0.copy()
    Z[6:8] = 0.0             # u = 0 at knot 1: A = dt G0
    Z[p.z:p.z + 6] = 0.0     # x = 0 at knot 2: every sweep column of that interval is zero
    _compare(p, Z=Z)


def test_wide_range_of_norms_in_one_call():
    """Per-interval scaling: dt spans 1e-6 .. 2, so s_k ranges from 1 to many squarings in one launch."""
    p = O.make_scaled_problem(8, 12, 2, seed=17)
    Z = p.Z0.copy()
    Z[p.dt_idx::p.z] = [1e-6, 1e-3, 0.05, 0.3, 0.8, 1.5, 2.0, 0.1]
    _compare(p, Z=Z, tol=1e-9, tol_h=1e-7)  # ||exp|| grows to ~1e6 here: tolerance relative to max(1,|ref|)


def test_non_finite_iterate_propagates_without_error():
    """Solvers may probe non-finite points; the reference would return NaNs, not throw."""
    import dto_amd
    p = O.make_scaled_problem(4, 4, 2, seed=1)
    ev = dto_amd.Evaluator(to_engine(p))
    Z = p.Z0.copy()
    Z[p.z + 1] = np.nan  # a state entry of knot 2
    g = np.zeros(ev.n_constraints); ev.eval_constraint(g, Z)
    j = np.zeros(ev.n_jacobian_entries); ev.eval_constraint_jacobian(j, Z)
    assert np.isnan(g).any() and np.isfinite(g[-2:]).all()
    assert np.isfinite(j[:p.z]).any()
    ev.close()


def test_jacobian_vector_products():
    """A3: y = J w and y = J' w (evaluator.jl:406-456; the reference compares them with the dense J,
    evaluator.jl:808-853)."""
    import dto_amd
    for p in (O.make_standard_problem(N=7), O.make_scaled_problem(6, 9, 3, seed=23, with_constraint=True)):
        ev_o = O.OracleEvaluator(p)
        ev = dto_amd.Evaluator(to_engine(p))
        rng = np.random.default_rng(5)
        w = rng.standard_normal(p.n_vars)
        v = rng.standard_normal(ev_o.n_constraints)
        y = np.full(ev_o.n_constraints, np.nan); ev.eval_constraint_jacobian_product(y, p.Z0, w)
        yt = np.full(p.n_vars, np.nan); ev.eval_constraint_jacobian_transpose_product(yt, p.Z0, v)
        assert rel_err(y, ev_o.eval_constraint_jacobian_product(p.Z0, w)) <= 1e-10
        assert rel_err(yt, ev_o.eval_constraint_jacobian_transpose_product(p.Z0, v)) <= 1e-10
        ev.close()


def test_generic_path_on_small_states():
    """n <= 16 normally takes the fused small-state kernel; DTO_FLAG_GENERAL_PATH_ONLY forces the batched-GEMM path
    (padding to 64, generator sweeps) on the same problems so both stay parity-checked."""
    import dto_amd
    for p in (O.make_standard_problem(N=7), O.make_scaled_problem(5, 9, 3, seed=5, with_constraint=True), O.make_readme_problem()):
        ev_o = O.OracleEvaluator(p)
        ev = dto_amd.Evaluator(to_engine(p), general_path_only=True)
        mu = np.random.default_rng(1).standard_normal(ev_o.n_constraints)
        out = run_all(ev, p, p.Z0, mu, sigma=0.5)
        assert rel_err(out["cons"], ev_o.eval_constraint(p.Z0)) <= 1e-10
        assert rel_err(out["jac"], ev_o.eval_constraint_jacobian(p.Z0)) <= 1e-10
        assert rel_err(out["hess"], ev_o.eval_hessian_lagrangian(p.Z0, 0.5, mu)) <= 1e-8
        assert ev.last_stats()[1] >= 1  # the generator sweep of the general path ran (the fused kernel reports 0 terms)
        ev.close()


def test_chunked_propagator_chain():
    """Option chain_chunk = 8 splits the 20 intervals of the chain into three chunks (what 16000 knots do with the default
    workspace budget): each chunk reads back its own squaring counts and picks its own polynomial form -- the time steps
    are chosen so that the chunks disagree (alpha ~ 1.5 in the first, ~ 4.5 in the others)."""
    import dto_amd
    p = O.make_scaled_problem(21, 40, 3, seed=9)
    Z = p.Z0.copy()
    dt = np.full(p.N, 0.28)
    dt[:8] = 0.09
    Z[p.dt_idx::p.z] = dt
    ref = O.OracleEvaluator(p).eval_constraint_jacobian(Z)
    ev = dto_amd.Evaluator(to_engine(p), eval_hessian=False)
    ev.set_option("chain_chunk", 8)
    for form in (0, 2, 3):
        ev.set_option("expm_form", form)
        j = np.full(ev.shard.jac_len, np.nan)
        ev.eval_constraint_jacobian(j, Z)
        assert rel_err(j, ref) <= 1e-10, (form, rel_err(j, ref))
    ev.close()


_BAD_LAUNCH_CHILD = r"""
import os, sys
root = os.environ["DTO_ROOT"]
for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
import numpy as np
import dto_amd, dto_oracle as O
from helpers import to_engine, rel_err
for general in (False, True):
    p = O.make_scaled_problem(5, 9, 2, seed=8, with_constraint=True)
    ev_o = O.OracleEvaluator(p)
    ev = dto_amd.Evaluator(to_engine(p), general_path_only=general)
    j = np.full(ev.shard.jac_len, np.nan)
    ev.set_option("debug_bad_launch", 1)
    for call in (lambda: ev.eval_constraint_jacobian(j, p.Z0), lambda: ev.eval_constraint(np.full(ev.shard.cons_len, np.nan), p.Z0)):
        try:
            call()
            raise SystemExit("the rejected launch did not come back as an error")
        except dto_amd.EngineError as e:
            assert "launch" in str(e) or "invalid" in str(e), str(e)
    ev.set_option("debug_bad_launch", 0)
    ev.eval_constraint_jacobian(j, p.Z0)
    assert rel_err(j, ev_o.eval_constraint_jacobian(p.Z0)) <= 1e-10
    ev.close()
print("child-ok")
"""


def test_rejected_kernel_launch_comes_back_as_an_error():
    """include/dto_engine.h error convention: non-zero return + text.  The TUNING build's option debug_bad_launch gives the
    callbacks' kernels a launch configuration the hardware does not have (block of 4096 threads on the general path, 512 KB of
    LDS more than a CU owns on the fused small-state path); the call must fail through the ABI, and the handle must work again
    afterwards.  The product library does not know the option (it is a test hook): it refuses the name."""
    import subprocess
    import sys
    import dto_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DTO_ROOT=root, DTO_ENGINE_LIB="libdto_engine_t.so")
    r = subprocess.run([sys.executable, "-c", _BAD_LAUNCH_CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "child-ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
    p = O.make_scaled_problem(5, 9, 2, seed=8)
    ev = dto_amd.Evaluator(to_engine(p))
    with pytest.raises(dto_amd.EngineError, match="TUNING builds only"):
        ev.set_option("debug_bad_launch", 1)
    ev.close()


def test_output_buffers_are_checked_on_the_host():
    """A float32, strided or short output array must be refused before the engine copies into it."""
    import dto_amd
    p = O.make_scaled_problem(5, 9, 2, seed=8, with_constraint=True)
    ev = dto_amd.Evaluator(to_engine(p), k_lo=2, k_hi=4)
    n = ev.shard.jac_len
    assert n < ev.n_jacobian_entries
    for bad in (np.empty(ev.n_jacobian_entries), np.empty(n, dtype=np.float32), np.empty(2 * n)[::2], np.empty(n - 1)):
        with pytest.raises(ValueError):
            ev.eval_constraint_jacobian(bad, p.Z0)
    with pytest.raises(ValueError):
        ev.eval_hessian_lagrangian(np.empty(ev.shard.hess_len), p.Z0, 1.0, np.ones(ev.n_constraints - 1))
    ok = np.empty(n)
    ev.eval_constraint_jacobian(ok, p.Z0)
    assert np.isfinite(ok).all()
    ev.close()


def test_large_norm_on_the_general_path_uses_substeps_and_many_squarings():
    """n > 16 with big time steps: the sweeps need q > 1 rounds (which also takes the Hessian off the
    pairing path, onto second-order columns) and the chain several squarings."""
    p = O.make_scaled_problem(4, 24, 3, seed=24)
    Z = p.Z0.copy()
    Z[p.dt_idx::p.z] = [2.0, 0.7, 1.3, 0.1]
    _compare(p, Z=Z, tol=1e-9, tol_h=1e-7)


@pytest.mark.parametrize("n,m", [(6, 5), (20, 6), (9, 7)])
def test_many_drives(n, m):
    """More than 4 drives: 1+m+m(m+1)/2 second-order column types (up to 36 at m = 7)."""
    _compare(O.make_scaled_problem(4, n, m, seed=n * m, with_constraint=True))


def test_matrix_free_products_on_the_general_path():
    """n > 16: J w and J' w come from the sweeps (exp(A)w_x as an extra column type / an adjoint sweep from w),
    no value slab; checked against the oracle's materialise-and-multiply."""
    import dto_amd
    for p in (O.make_scaled_problem(5, 20, 3, seed=77, with_constraint=True), O.make_scaled_problem(4, 70, 2, seed=78)):
        ev_o = O.OracleEvaluator(p)
        ev = dto_amd.Evaluator(to_engine(p))
        rng = np.random.default_rng(6)
        w = rng.standard_normal(p.n_vars)
        v = rng.standard_normal(ev_o.n_constraints)
        y = np.full(ev_o.n_constraints, np.nan); ev.eval_constraint_jacobian_product(y, p.Z0, w)
        yt = np.full(p.n_vars, np.nan); ev.eval_constraint_jacobian_transpose_product(yt, p.Z0, v)
        assert rel_err(y, ev_o.eval_constraint_jacobian_product(p.Z0, w)) <= 1e-10
        assert rel_err(yt, ev_o.eval_constraint_jacobian_transpose_product(p.Z0, v)) <= 1e-10
        ev.close()


def test_reuse_forward_sweep_between_callbacks():
    """Option reuse_forward_sweep: g, J, H evaluated one after the other at the same point (what an interior-point
    iteration does) share the forward generator sweep; every ordering, with the point changing in between, must give
    the same numbers as a handle without the option."""
    import dto_amd
    import dto_oracle as O
    from helpers import rel_err, to_engine
    p = O.make_scaled_problem(6, 40, 2, seed=17, with_constraint=True)  # general path (n > 32)
    ref = dto_amd.Evaluator(to_engine(p))
    ev = dto_amd.Evaluator(to_engine(p))
    ev.set_option("reuse_forward_sweep", 1)
    with pytest.raises(dto_amd.EngineError):
        ev.set_option("no_such_option", 1)
    rng = np.random.default_rng(0)
    Z1 = p.Z0.copy()
    Z2 = p.Z0 + 0.05 * rng.standard_normal(p.n_vars)
    mu = rng.standard_normal(ref.n_constraints)

    def calls(e, Z):
        g = np.empty(e.n_constraints); e.eval_constraint(g, Z)
        J = np.empty(e.n_jacobian_entries); e.eval_constraint_jacobian(J, Z)
        H = np.empty(e.n_hessian_entries); e.eval_hessian_lagrangian(H, Z, 0.9, mu)
        return {"g": g, "J": J, "H": H}

    try:
        want = {id(Z): calls(ref, Z) for Z in (Z1, Z2)}
        orders = [("g", "J", "H"), ("J", "g", "H"), ("H", "J", "g"), ("J", "H", "J", "g"), ("g", "g", "H", "H")]
        for order in orders:
            for Z in (Z1, Z2, Z1):
                for what in order:
                    if what == "g":
                        out = np.empty(ev.n_constraints); ev.eval_constraint(out, Z)
                    elif what == "J":
                        out = np.empty(ev.n_jacobian_entries); ev.eval_constraint_jacobian(out, Z)
                    else:
                        out = np.empty(ev.n_hessian_entries); ev.eval_hessian_lagrangian(out, Z, 0.9, mu)
                    # (the constraint-only sweep splits K by generator: same terms, different summation order)
                    assert rel_err(out, want[id(Z)][what]) <= 1e-13, (order, what)
        # products in between must not leave a stale cache behind
        y = np.empty(ev.n_constraints); ev.eval_constraint_jacobian_product(y, Z1, rng.standard_normal(p.n_vars))
        out = np.empty(ev.n_constraints); ev.eval_constraint(out, Z1)
        assert rel_err(out, want[id(Z1)]["g"]) <= 1e-13
    finally:
        ref.close(); ev.close()


@pytest.mark.parametrize("n,N", [(64, 900), (128, 700)])
def test_reuse_forward_sweep_where_the_sweep_is_one_launch(n, N):
    """The same option on shapes whose multi-column sweeps run as ONE persistent launch (fused / cluster form): a Jacobian that
    finds the p terms of its point stored sweeps all columns again (cheaper than the frozen step-per-launch form) and leaves the
    stored terms to the Hessian.  Compared with a handle that has the option off, every ordering, the point changing in between."""
    import dto_amd
    from helpers import rel_err
    p = dto_amd.host.synthetic.make_scaled_problem(N, n, 2, seed=5)
    ref = dto_amd.Evaluator(p)
    ev = dto_amd.Evaluator(p)
    ev.set_option("reuse_forward_sweep", 1)
    rng = np.random.default_rng(1)
    Z1 = p.trajectory.vec()
    Z2 = Z1 + 0.02 * rng.standard_normal(Z1.size)
    mu = rng.standard_normal(ref.n_constraints)

    def one(e, what, Z):
        if what == "g":
            out = np.empty(e.n_constraints); e.eval_constraint(out, Z)
        elif what == "J":
            out = np.empty(e.n_jacobian_entries); e.eval_constraint_jacobian(out, Z)
        else:
            out = np.empty(e.n_hessian_entries); e.eval_hessian_lagrangian(out, Z, 0.9, mu)
        return out

    try:
        want = {id(Z): {w: one(ref, w, Z) for w in ("g", "J", "H")} for Z in (Z1, Z2)}
        for order in (("g", "J", "H"), ("J", "g", "H"), ("H", "J", "g"), ("g", "H", "J", "H")):
            for Z in (Z1, Z2, Z1):
                for what in order:
                    assert rel_err(one(ev, what, Z), want[id(Z)][what]) <= 1e-12, (order, what)
    finally:
        ref.close(); ev.close()


def test_host_pointer_hand_off_ships_only_what_changes():
    """dto_hostxfer: the host-pointer Jacobian / Hessian copy the variable runs only and fill constants on the host.  The
    result must be bit-identical to the whole-slab copy (option host_xfer = 0), into buffers pre-filled with garbage, for
    problems with every kind of term (constraints, closures, foreign integrators, globals, several bilinear integrators)."""
    import dto_amd
    cases = [(O.make_scaled_problem(9, 24, 3, seed=2, with_constraint=True), "numeric"), (O.make_standard_problem(N=8), "numeric"),
             (O.make_closure_problem(), "analytic"), (O.make_external_integrator_problem(), "analytic"),
             (O.make_global_problem(), "analytic"), (O.make_type1_derivative_problem(), "numeric"), (O.make_ket_problem(), "numeric")]
    for p, how in cases:
        outs = []
        for on in (1, 0):
            ev = dto_amd.Evaluator(to_engine(p, how))
            ev.set_option("host_xfer", on)
            mu = np.random.default_rng(3).standard_normal(ev.n_constraints)
            Z = p.Z0 + 0.01
            for rep in range(2):  # the second call re-uses the ring and the plans
                j = np.full(ev.shard.jac_len, np.nan); ev.eval_constraint_jacobian(j, Z)
                h = np.full(ev.shard.hess_len, 1e300); ev.eval_hessian_lagrangian(h, Z, 0.7, mu)
            outs.append((j, h))
            ev.close()
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    # the general path hands the -E_k blocks over chain chunk by chain chunk (four chunks from 512 intervals on), whole and
    # as a shard whose first interval is not the problem's first
    p = O.make_scaled_problem(600, 40, 2, seed=5)
    for lo, hi in ((1, 600), (31, 590)):
        outs = []
        for on in (1, 0):
            ev = dto_amd.Evaluator(to_engine(p), eval_hessian=False, k_lo=lo, k_hi=hi)
            ev.set_option("host_xfer", on)
            for rep in range(2):
                j = np.full(ev.shard.jac_len, np.nan); ev.eval_constraint_jacobian(j, p.Z0 + 0.001 * rep)
            outs.append(j)
            ev.close()
        assert np.array_equal(outs[0], outs[1])
    # shards too
    p = O.make_scaled_problem(11, 8, 2, seed=3, with_constraint=True)
    for lo, hi in ((1, 4), (5, 11)):
        outs = []
        for on in (1, 0):
            ev = dto_amd.Evaluator(to_engine(p), k_lo=lo, k_hi=hi)
            ev.set_option("host_xfer", on)
            mu = np.random.default_rng(3).standard_normal(ev.n_constraints)
            j = np.full(ev.shard.jac_len, np.nan); ev.eval_constraint_jacobian(j, p.Z0)
            h = np.full(ev.shard.hess_len, np.nan); ev.eval_hessian_lagrangian(h, p.Z0, 0.7, mu)
            outs.append((j, h))
            ev.close()
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_sweep_on_the_second_stream_changes_no_bit():
    """Option overlap_sweep (default 1): the Jacobian's generator sweep runs on a second stream next to the chain's products.
    Same kernels, same summation orders: the slab must be bit-identical to the one-kernel-at-a-time run, for the fused sweep
    (64 states x 1200 knots: 134 workgroups) and the step-per-launch sweep (few intervals), whole and sharded."""
    import torch
    import dto_amd
    dev = torch.device("cuda", 0)
    for N, n, lo, hi in ((1200, 64, 0, 0), (40, 48, 0, 0), (1200, 64, 101, 1150)):
        p = O.make_scaled_problem(N, n, 3, seed=11)
        ev = dto_amd.Evaluator(to_engine(p), eval_hessian=False, k_lo=lo, k_hi=hi)
        Z = torch.from_numpy(p.Z0).to(dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        outs = []
        for on in (1, 0, 1):
            ev.set_option("overlap_sweep", on)
            o = torch.full((ev.shard.jac_len,), float("nan"), dtype=torch.float64, device=dev)
            ev.eval_jacobian_dev(Z.data_ptr(), o.data_ptr(), st)
            torch.cuda.synchronize()
            outs.append(o)
        ev.close()
        assert bool(torch.isfinite(outs[0]).all())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # the Hessian's forward sweep next to its adjoint sweep (same option): accumulated with floating-point atomics, so equal
    # to rounding rather than to the bit
    p = O.make_scaled_problem(1200, 64, 3, seed=11)
    ev = dto_amd.Evaluator(to_engine(p))
    Z = torch.from_numpy(p.Z0).to(dev)
    mu = torch.from_numpy(np.random.default_rng(2).standard_normal(ev.n_constraints)).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    outs = []
    for on in (1, 0):
        ev.set_option("overlap_sweep", on)
        o = torch.full((ev.shard.hess_len,), float("nan"), dtype=torch.float64, device=dev)
        ev.eval_hessian_dev(Z.data_ptr(), 0.9, mu.data_ptr(), o.data_ptr(), st)
        torch.cuda.synchronize()
        outs.append(o.cpu().numpy())
    ev.close()
    assert np.isfinite(outs[0]).all() and rel_err(outs[0], outs[1]) <= 1e-12


def test_host_pointer_calls_back_to_back_match_the_device_resident_ones():
    """The host-pointer Jacobian hands its -E_k blocks to a drainer thread chain chunk by chain chunk while the GPU still
    computes (dto_hostxfer.h), host threads scatter into the caller's vector.  Twelve calls in a row at changing points,
    Jacobian and Hessian alternating, each into a buffer full of NaN: every result must equal the device-resident entry
    point's, bit for bit (Jacobian; same kernels) / to rounding (Hessian: floating-point atomics)."""
    import torch
    import dto_amd
    dev = torch.device("cuda", 0)
    p = O.make_scaled_problem(700, 64, 3, seed=17)
    ev = dto_amd.Evaluator(to_engine(p))
    st = torch.cuda.current_stream(dev).cuda_stream
    rng = np.random.default_rng(9)
    mu = rng.standard_normal(ev.n_constraints)
    mu_d = torch.from_numpy(mu).to(dev)
    jd = torch.empty(ev.shard.jac_len, dtype=torch.float64, device=dev)
    hd = torch.empty(ev.shard.hess_len, dtype=torch.float64, device=dev)
    for it in range(12):
        Z = p.Z0 + 0.01 * rng.standard_normal(p.Z0.size)
        Zd = torch.from_numpy(Z).to(dev)
        j = np.full(ev.shard.jac_len, np.nan)
        ev.eval_constraint_jacobian(j, Z)
        ev.eval_jacobian_dev(Zd.data_ptr(), jd.data_ptr(), st)
        torch.cuda.synchronize()
        assert np.array_equal(j, jd.cpu().numpy()), it
        if it % 2 == 0:
            h = np.full(ev.shard.hess_len, np.nan)
            ev.eval_hessian_lagrangian(h, Z, 0.5, mu)
            ev.eval_hessian_dev(Zd.data_ptr(), 0.5, mu_d.data_ptr(), hd.data_ptr(), st)
            torch.cuda.synchronize()
            assert rel_err(h, hd.cpu().numpy()) <= 1e-12, it
    ev.close()


def test_reuse_takes_the_chains_step_budget_where_the_cheap_bound_is_not_enough():
    """Large time steps on the batched-GEMM path (96 states): the cheap norm bound asks for sub-steps, so a Hessian on its own buys the
    exact norms of A^2..A^4 for its step budget; with reuse_forward_sweep a Hessian that follows the Jacobian at the same point takes
    the budget that call's chain planned.  Same numbers either way (to rounding), against the oracle, the point changing in between."""
    import dto_amd
    p = O.make_scaled_problem(5, 96, 2, seed=96, skew=True)
    Z1 = p.Z0.copy()
    Z1[p.dt_idx::p.z] = 0.1 * np.sqrt(256.0 / 96) * 4.0 * np.array([2.0, 0.7, 3.0, 1.3, 1.0])
    Z2 = Z1.copy()
    Z2[p.dt_idx::p.z] *= 0.8
    ev_o = O.OracleEvaluator(p)
    ref = dto_amd.Evaluator(to_engine(p))
    ev = dto_amd.Evaluator(to_engine(p))
    ev.set_option("reuse_forward_sweep", 1)
    mu = np.random.default_rng(3).standard_normal(ev_o.n_constraints)
    try:
        for Z in (Z1, Z2, Z1):
            J = np.empty(ev.n_jacobian_entries); ev.eval_constraint_jacobian(J, Z)
            H = np.empty(ev.n_hessian_entries); ev.eval_hessian_lagrangian(H, Z, 0.9, mu)
            J0 = np.empty(ref.n_jacobian_entries); ref.eval_constraint_jacobian(J0, Z)
            H0 = np.empty(ref.n_hessian_entries); ref.eval_hessian_lagrangian(H0, Z, 0.9, mu)
            assert rel_err(J, J0) <= 1e-12 and rel_err(H, H0) <= 1e-11
            assert rel_err(J, ev_o.eval_constraint_jacobian(Z)) <= 1e-9
            assert rel_err(H, ev_o.eval_hessian_lagrangian(Z, 0.9, mu)) <= 1e-7
            H2 = np.empty(ev.n_hessian_entries); ev.eval_hessian_lagrangian(H2, Z, 0.9, mu)   # again, the budget still cached
            assert rel_err(H2, H0) <= 1e-11
    finally:
        ref.close(); ev.close()
