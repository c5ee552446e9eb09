"""The C ABI from plain C: tests/c_client/readme_structure.c is compiled with gcc against include/dto_engine.h,
linked with the in-tree libdto_engine.so and run WITHOUT a GPU (structure-only handle); its printed sizes and sparsity
structure must be the oracle's."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import dto_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_client_gets_the_reference_structure(tmp_path, engine_lib):
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    libdir = os.path.join(ROOT, "directtrajopt.jl_amd")
    exe = str(tmp_path / "readme_structure")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_client", "readme_structure.c"), "-o", exe,
                    "-L", libdir, "-ldto_engine", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines()
    got = {l.split()[0]: l.split()[1:] for l in out}

    p = O.make_readme_problem()
    p.Z0 = np.tile([1.0, 0.0, 0.1, 0.1], p.N)
    ev = O.OracleEvaluator(p)
    jr, jc = ev.jacobian_structure1()
    hr, hc = ev.hessian_structure1()
    assert [int(x) for x in got["sizes"]] == [p.n_vars, ev.n_constraints, ev.n_dynamics_constraints, jr.size, hr.size]
    i = np.arange(jr.size)
    assert int(got["jac_checksum"][0]) == int((jr * (i % 7 + 1)).sum()) and int(got["jac_checksum"][1]) == int((jc * (i % 5 + 1)).sum())
    assert [int(x) for x in (got["jac_checksum"][3], got["jac_checksum"][4], got["jac_checksum"][6], got["jac_checksum"][7])] == \
        [int(jr[0]), int(jc[0]), int(jr[-1]), int(jc[-1])]
    assert got["hess_tail"] == [f"{a}:{b}" for a, b in zip(hr[-8:], hc[-8:])]
    assert got["eval_rc"][0] != "0"  # no device, no answer: the engine has no CPU fallback
