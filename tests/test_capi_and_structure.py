"""CPU checks of the boundary: the C-ABI library loads, exports every symbol include/dto_engine.h
declares, and its (GPU-free) structure builder reproduces the reference's index contract."""
import ctypes
import os
import re

import numpy as np
import pytest

import dto_amd
import dto_oracle as O
from helpers import to_engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "dto_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dto_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(engine_lib):
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(engine_lib, n), f"{n} declared in the header but not exported"
        assert n in dto_amd.capi.SYMBOLS, f"{n} has no ctypes prototype"
    assert sorted(dto_amd.capi.SYMBOLS) == names


def test_struct_sizes_match_header_layout():
    # sizes computed by hand from include/dto_engine.h (LP64): catches field drift between C and ctypes
    assert ctypes.sizeof(dto_amd.capi.IntegratorDesc) == 72
    assert ctypes.sizeof(dto_amd.capi.ObjectiveDesc) == 112
    assert ctypes.sizeof(dto_amd.capi.ConstraintDesc) == 64
    assert ctypes.sizeof(dto_amd.capi.ProblemDesc) == 96
    assert ctypes.sizeof(dto_amd.capi.ShardInfo) == 112
    assert ctypes.sizeof(dto_amd.capi.ExternalValues) == 24


def test_struct_layout_matches_the_compiled_header(tmp_path):
    """sizeof/offsetof of every ABI struct as gcc sees include/dto_engine.h vs the ctypes mirror."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {"dto_integrator_desc": dto_amd.capi.IntegratorDesc, "dto_objective_desc": dto_amd.capi.ObjectiveDesc,
               "dto_constraint_desc": dto_amd.capi.ConstraintDesc, "dto_problem_desc": dto_amd.capi.ProblemDesc,
               "dto_shard_info": dto_amd.capi.ShardInfo, "dto_external_values": dto_amd.capi.ExternalValues}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "dto_engine.h"', 'int main(void) {',
             'printf("abi %d\\n", DTO_ABI_VERSION);']
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    assert int(got["abi"]) == dto_amd.capi.DTO_ABI_VERSION
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_create_rejects_bad_input():
    p = O.make_readme_problem()
    prob = to_engine(p)
    with pytest.raises(dto_amd.EngineError):
        dto_amd.Evaluator(prob, device=-1, k_lo=3, k_hi=2)
    with pytest.raises(ValueError):
        dto_amd.NamedTrajectory({"x": np.zeros((2, 3))}, timestep=0.1)  # fixed Float timestep is rejected
    with pytest.raises(ValueError):
        traj = prob.trajectory
        dto_amd.BilinearIntegrator(lambda u: np.eye(2) * (1 + u[0] ** 2), "c0", "c2", traj)  # not affine


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dto_amd.EngineError, match="no HIP device|no CPU fallback"):
        dto_amd.Evaluator(to_engine(O.make_readme_problem()), device=0)
    ev = dto_amd.Evaluator(to_engine(O.make_readme_problem()), device=-1)
    with pytest.raises(dto_amd.EngineError, match="structure-only"):
        ev.eval_objective(np.zeros(ev.n_variables))


PROBLEMS = {
    "readme": lambda: O.make_readme_problem(),
    "standard": lambda: O.make_standard_problem(N=9),
    "type1": lambda: O.make_type1_derivative_problem(),
    "scaled_con": lambda: O.make_scaled_problem(7, 5, 3, seed=8, with_constraint=True),
    "closure": lambda: O.make_closure_problem(),
    "ket": lambda: O.make_ket_problem(),
    "external_integrator": lambda: O.make_external_integrator_problem(),
    "global": lambda: O.make_global_problem(),
}


@pytest.mark.parametrize("name", sorted(PROBLEMS))
def test_structure_bit_exact_vs_reenactment(name):
    p = PROBLEMS[name]()
    ev_o = O.OracleEvaluator(p)
    ev = dto_amd.Evaluator(to_engine(p), device=-1)
    assert (ev.n_variables, ev.n_constraints, ev.n_dynamics_constraints) == (p.n_vars, ev_o.n_constraints, ev_o.n_dynamics_constraints)
    r, c = ev.jacobian_structure()
    r1, c1 = ev_o.jacobian_structure1()
    assert np.array_equal(r, r1) and np.array_equal(c, c1)
    r, c = ev.hessian_lagrangian_structure()
    r1h, c1h = ev_o.hessian_structure1()
    assert np.array_equal(r, r1h) and np.array_equal(c, c1h)
    # ranged queries (the Julia shim pulls multi-GB structures in pieces)
    rng = np.random.default_rng(0)
    for _ in range(5):
        first = int(rng.integers(0, len(r1)))
        cnt = int(rng.integers(0, len(r1) - first + 1))
        rr, cc = ev.jacobian_structure(first, cnt)
        assert np.array_equal(rr, r1[first:first + cnt]) and np.array_equal(cc, c1[first:first + cnt])
        first = int(rng.integers(0, len(r1h)))
        cnt = int(rng.integers(0, len(r1h) - first + 1))
        rr, cc = ev.hessian_lagrangian_structure(first, cnt)
        assert np.array_equal(rr, r1h[first:first + cnt]) and np.array_equal(cc, c1h[first:first + cnt])
    lo, hi = ev.constraint_bounds()
    lo_o, hi_o = ev_o.row_bounds()
    assert np.array_equal(lo, lo_o) and np.array_equal(hi, hi_o)


def test_value_dependent_constraint_pattern():
    """Exact zeros of the constraint Jacobian at Z0 are not part of the structure (evaluator.jl:136)."""
    p = O.make_scaled_problem(6, 3, 2, seed=2, with_constraint=True)
    p.Z0[2 * p.z + 3] = 0.0  # u_1 of knot 3 is exactly zero at Z0
    ev_o = O.OracleEvaluator(p)
    ev = dto_amd.Evaluator(to_engine(p), device=-1)
    r, c = ev.jacobian_structure()
    r1, c1 = ev_o.jacobian_structure1()
    assert len(r1) == 2 * p.z * 5 * 5 + 2 * 4 - 1
    assert np.array_equal(r, r1) and np.array_equal(c, c1)


def test_sizes_at_baseline_configurations():
    """nnz formulas of BASELINE.md §2 / SURVEY.md §8a at the north-star size, without materialising."""
    n, m, N = 256, 4, 2000
    prob = dto_amd.host.synthetic.make_scaled_problem(N, n, m)
    ev = dto_amd.Evaluator(prob, device=-1)
    assert ev.n_variables == 530000 and ev.n_constraints == 519740
    assert ev.n_jacobian_entries == 275462200
    assert ev.n_hessian_entries == 210869775
    r, c = ev.jacobian_structure(275462200 - 3, 3)
    assert list(c) == [530000] * 3 and list(r) == [519738, 519739, 519740]
    r, c = ev.hessian_lagrangian_structure(0, 3)
    assert list(zip(r, c)) == [(1, 1), (1, 2), (2, 2)]


def test_shards_tile_the_value_vectors():
    p = O.make_scaled_problem(11, 4, 2, seed=3, with_constraint=True)
    full = dto_amd.Evaluator(to_engine(p), device=-1)
    for world in (2, 3, 4):
        jl = hl = gl = 0
        rows = []
        for (lo, hi) in dto_amd.distributed.shard_ranges(p.N, world):
            ev = dto_amd.Evaluator(to_engine(p), device=-1, k_lo=lo, k_hi=hi)
            s = ev.shard
            assert (s.jac_lo, s.hess_lo, s.grad_lo) == (jl, hl, gl)
            jl += s.jac_len; hl += s.hess_len; gl += s.grad_len
            st, ln = ev.shard_rows()
            assert ln.sum() == s.cons_len
            for a, b in zip(st, ln):
                rows += list(range(a, a + b))
        assert (jl, hl, gl) == (full.n_jacobian_entries, full.n_hessian_entries, full.n_variables)
        assert sorted(rows) == list(range(1, full.n_constraints + 1))


def test_host_mirror_objects():
    traj = dto_amd.NamedTrajectory({"x": np.arange(6.0).reshape(2, 3), "u": np.ones((1, 3)), "dt": np.full((1, 3), 0.1)}, "dt")
    assert traj.dim == 4 and traj.N == 3 and list(traj.components["u"]) == [2]
    assert np.array_equal(traj.vec()[:4], [0.0, 3.0, 1.0, 0.1])  # knot-major datavec
    a = dto_amd.QuadraticRegularizer("u", traj, 2.0)
    b = dto_amd.MinimumTimeObjective(traj, D=3.0)
    J = 0.5 * (a + 2 * b)
    assert isinstance(J, dto_amd.CompositeObjective) and J.weights == [0.5, 1.0]
    # closure-based knot terms are accepted (host-evaluated, engine-merged); g_dim comes from one evaluation
    c = dto_amd.NonlinearKnotPointConstraint(lambda u, p: np.array([u[0] ** 2 - 1.0, u[0]]), "u", traj, times=[1, 3])
    assert c.external and c.g_dim == 2 and c.dim == 4
    vals, jac, hess = c.external_blocks(traj.vec()[:12].reshape(3, 4), 2, mu=np.array([1.0, 2.0, 3.0, 4.0]))
    assert np.allclose(vals, [[0.0, 1.0], [0.0, 1.0]]) and np.allclose(jac[:, 0, :], [[2.0, 1.0], [2.0, 1.0]])
    assert np.allclose(hess[:, 0, 0], [2.0, 6.0], atol=1e-6)  # mu_i[0] * d2(u^2)/du2
    o = dto_amd.TerminalObjective(lambda x, goal: float(np.sum((x - goal) ** 2)), "x", traj, goal=np.array([1.0, 0.0]), Q=2.0)
    v, g, _ = o.external_blocks(traj.vec()[:12].reshape(3, 4), 1)
    assert np.allclose(v, [2.0 * ((2 - 1) ** 2 + 5 ** 2)]) and np.allclose(g, [[4.0, 20.0]])
    with pytest.raises(ValueError):
        dto_amd.NonlinearKnotPointConstraint("no-such-kind", "u", traj)


def test_external_terms_are_counted_and_required():
    """dto_num_external / dto_set_external on a structure-only handle (no GPU): slot count and argument check."""
    p = O.make_closure_problem()
    ev = dto_amd.Evaluator(to_engine(p), device=-1)
    ni, nc, no = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    assert ev._lib.dto_num_external(ev.handle, ctypes.byref(ni), ctypes.byref(nc), ctypes.byref(no)) == 0
    assert (ni.value, nc.value, no.value) == (0, 1, 1)
    ev2 = dto_amd.Evaluator(to_engine(O.make_external_integrator_problem()), device=-1)
    assert ev2._lib.dto_num_external(ev2.handle, ctypes.byref(ni), ctypes.byref(nc), ctypes.byref(no)) == 0
    assert (ni.value, nc.value, no.value) == (1, 0, 0)
    assert ev._lib.dto_set_external(ev.handle, 1, None) != 0  # needs exactly 2 entries
    vals = (dto_amd.capi.ExternalValues * 2)()
    assert ev._lib.dto_set_external(ev.handle, 2, vals) == 0
    lo, hi = ev.constraint_bounds()
    lo_o, hi_o = O.OracleEvaluator(p).row_bounds()
    assert np.array_equal(lo, lo_o) and np.array_equal(hi, hi_o)


def test_cpu_baseline_worker_uses_the_bench_problem():
    """oracle/cpu_baseline_worker.py restates the synthetic generator (it must not import the product);
    both must give the same arrays."""
    import importlib.util
    import os
    import numpy as np
    import dto_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cpu_baseline_worker", os.path.join(root, "oracle", "cpu_baseline_worker.py"))
    W = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(W)
    a = W.scaled_problem_arrays(np, 9, 5, 2, 42)
    b = dto_amd.host.synthetic.scaled_problem_arrays(9, 5, 2, 42)
    for p, q in zip(a, b):
        assert np.array_equal(p, q)
