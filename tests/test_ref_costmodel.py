"""CPU: the C restatement of the reference's algorithm (oracle/dto_ref_costmodel.c -- forward-mode duals in chunks of 12
through the truncated-Taylor expv; it is the CPU baseline bench.py times) computes the same Jacobian block as the
oracle's closed forms (scipy expm / expm_frechet)."""
import os
import subprocess

import numpy as np
import pytest

import dto_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ref") / "dto_ref_costmodel")
    subprocess.run(["gcc", "-O2", "-o", out, os.path.join(ROOT, "oracle", "dto_ref_costmodel.c"), "-lm"], check=True)
    return out


def _write(p, path):
    it = p.integrators[0]
    with open(path, "wb") as f:
        np.array([it.x_dim, it.u_dim, p.N, p.z], dtype=np.int64).tofile(f)
        np.ascontiguousarray(np.transpose(it.G, (0, 2, 1))).tofile(f)
        p.Z0[:p.N * p.z].tofile(f)


@pytest.mark.parametrize("N,n,m,dt", [(4, 6, 2, 0.1), (3, 24, 3, 0.1), (3, 10, 1, 0.9)])
def test_costmodel_block_matches_the_oracle(exe, tmp_path, N, n, m, dt):
    p = O.make_scaled_problem(N, n, m, seed=N + n)
    p.Z0[p.dt_idx::p.z] = dt  # 0.9: several scaling stages (s > 1) of Algorithm 3.2
    prob, blk = str(tmp_path / "p.bin"), str(tmp_path / "b.bin")
    _write(p, prob)
    for k in range(N - 1):
        r = subprocess.run([exe, "block", prob, str(k), blk], capture_output=True, text=True, check=True)
        assert int(r.stdout) >= -(-2 * p.z // 12)  # at least one dual product per ForwardDiff chunk
        got = np.fromfile(blk).reshape(2 * p.z, n).T
        ref = np.zeros((n, 2 * p.z))
        ref[:, :p.z] = O.bilinear_block_jacobian(p.integrators[0], p, p.Z0[k * p.z:(k + 1) * p.z])
        ref[:, p.z:p.z + n] = np.eye(n)
        assert np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))) <= 1e-11


def test_costmodel_bench_mode_reports_knots_and_seconds(exe, tmp_path):
    p = O.make_scaled_problem(5, 8, 2, seed=1)
    prob = str(tmp_path / "p.bin")
    _write(p, prob)
    r = subprocess.run([exe, "bench", prob, "1", "2", "5"], capture_output=True, text=True, check=True)
    knots, secs, matvecs = r.stdout.split()
    assert int(knots) == 2 and float(secs) > 0 and int(matvecs) > 0  # intervals 1 and 3 of the four
