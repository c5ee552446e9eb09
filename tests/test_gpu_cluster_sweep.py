"""GPU: the row-split cluster form of the generator sweep (csrc/dto_sweep_fused.hip, k_sweep_cluster): R workgroups share an
interval group, each computes npad / R rows of every Taylor term and the slices are exchanged through global memory with
agent-scope 8-byte atomics.  It serves what the single-workgroup form cannot fill the chip with: short shards (the 250-knot
share of the 2000-knot metric on 8 GPUs), single-column sweeps (eval_constraint, the Hessian's forward sweep).

Checked against the oracle at the usual bars AND against the step-per-launch form of the same engine (option sweep_form = 1),
for 128- and 256-state integrators, with and without sub-stepping, whole and sharded; repeated calls are bit-identical (the
exchange protocol must not let a stale slice through: every word of every callback is compared)."""
import numpy as np
import pytest

import dto_oracle as O
from helpers import rel_err, to_engine

pytestmark = pytest.mark.gpu


def _callbacks(ev, Z, mu):
    c = np.full(ev.shard.cons_len, np.nan); ev.eval_constraint(c, Z)
    j = np.full(ev.shard.jac_len, np.nan); ev.eval_constraint_jacobian(j, Z)
    h = np.full(ev.shard.hess_len, np.nan); ev.eval_hessian_lagrangian(h, Z, 0.9, mu)
    return c, j, h


@pytest.mark.parametrize("n,m,N,scale", [(100, 3, 14, 1.0), (200, 4, 12, 1.0), (256, 2, 9, 1.0), (120, 2, 10, 6.0)],
                         ids=["128-states", "256-states", "256-exact", "sub-stepped"])
def test_cluster_sweep_matches_the_oracle_and_the_step_form(n, m, N, scale):
    import dto_amd
    p = O.make_scaled_problem(N, n, m, seed=100 + n, with_constraint=True)
    if scale != 1.0:   # larger steps: ||A|| grows, the sweep runs q > 1 rounds (the sums go through the exchange as well)
        Zk = p.Z0[:p.z * N].reshape(N, p.z)
        Zk[:, p.dt_idx] *= scale
    ev_o = O.OracleEvaluator(p)
    Z = p.Z0
    mu = np.random.default_rng(5).standard_normal(ev_o.n_constraints)
    want = (ev_o.eval_constraint(Z), ev_o.eval_constraint_jacobian(Z), ev_o.eval_hessian_lagrangian(Z, 0.9, mu))
    ev = dto_amd.Evaluator(to_engine(p))
    ev_step = dto_amd.Evaluator(to_engine(p))
    ev_step.set_option("sweep_form", 1)
    try:
        got = _callbacks(ev, Z, mu)
        ref = _callbacks(ev_step, Z, mu)
        for g, w, r, tol in zip(got, want, ref, (1e-10, 1e-10, 1e-8)):
            assert rel_err(g, w) <= tol and rel_err(g, r) <= tol
        for _ in range(3):   # no stale slice: every repetition reproduces every word
            again = _callbacks(ev, Z, mu)
            for a, g in zip(again, got):
                assert np.array_equal(a, g)
    finally:
        ev.close(); ev_step.close()


def test_cluster_sweep_on_shards():
    import dto_amd
    p = O.make_scaled_problem(20, 128, 2, seed=77)
    ev_o = O.OracleEvaluator(p)
    Z = p.Z0
    want = ev_o.eval_constraint_jacobian(Z)
    got = np.full_like(want, np.nan)
    for lo, hi in dto_amd.distributed.shard_ranges(p.N, 3):
        e = dto_amd.Evaluator(to_engine(p), k_lo=lo, k_hi=hi)
        o = np.empty(e.shard.jac_len); e.eval_constraint_jacobian(o, Z)
        got[e.shard.jac_lo:e.shard.jac_lo + e.shard.jac_len] = o
        e.close()
    assert rel_err(got, want) <= 1e-10
