#!/usr/bin/env python3
"""CPU-baseline worker for bench.py's `cpu_baseline` leg (test/measurement infrastructure, never on
the product path).  One single-threaded process: evaluates the oracle's bilinear Jacobian blocks
(scipy expm + expm_frechet; bilinear_integrator.jl:111-131 restated in dto_oracle.py) for the
intervals  worker, worker+W, worker+2W, ...  of the synthetic benchmark problem until the time
budget is spent, then prints "<intervals done> <seconds>".

usage: cpu_baseline_worker.py <repo root> <n> <m> <N> <worker> <W> <budget seconds>"""
import os
import sys
import time

def scaled_problem_arrays(np, N, n, m, seed):
    """Same stream and fill order as directtrajopt.jl_amd/host/synthetic.py (checked by
    tests/test_capi_and_structure.py); restated here so the worker imports nothing of the product."""
    rng = np.random.Generator(np.random.Philox(seed))
    r = rng.standard_normal((m + 1) * n * n + N * (n + 2 * m))
    G = r[:(m + 1) * n * n].reshape(m + 1, n, n).transpose(0, 2, 1).copy()
    rest = r[(m + 1) * n * n:]
    return (G, rest[:n * N].reshape(N, n).T, 0.1 * rest[n * N:n * N + m * N].reshape(N, m).T,
            rest[n * N + m * N:].reshape(N, m).T)


def main():
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"  # before numpy loads its BLAS
    root, n, m, N, w, W, budget = sys.argv[1], *map(int, sys.argv[2:7]), float(sys.argv[7])
    sys.path.insert(0, os.path.join(root, "oracle"))
    import numpy as np
    import dto_oracle as O
    G, x, u, du = scaled_problem_arrays(np, N, n, m, 42)
    z = n + 2 * m + 1
    prob = O.Problem(N=N, z=z, dt_idx=z - 1, integrators=[O.BilinearIntegrator(0, n, n, m, G)], Z0=None)
    integ = prob.integrators[0]
    t0 = time.perf_counter()
    done = 0
    k = w
    while k < N - 1 and (time.perf_counter() - t0 < budget or done < 1):
        zk = np.concatenate([x[:, k], u[:, k], du[:, k], [0.1]])
        O.bilinear_block_jacobian(integ, prob, zk)
        done += 1
        k += W
    print(done, time.perf_counter() - t0, flush=True)


if __name__ == "__main__":
    main()
