"""Detailed parity dump (GPU box): per-callback errors and worst entries for a few problems."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import dto_amd
import dto_oracle as O
from helpers import to_engine, rel_err

def dump(name, prob_o, Z=None, hessian=True):
    print("=====", name, flush=True)
    ev_o = O.OracleEvaluator(prob_o)
    t = time.time()
    ev = dto_amd.Evaluator(to_engine(prob_o), eval_hessian=hessian)
    print("create %.3fs n_vars %d n_cons %d jac %d hess %d" % (time.time() - t, ev.n_variables, ev.n_constraints, ev.n_jacobian_entries, ev.n_hessian_entries))
    Z = prob_o.Z0.copy() if Z is None else Z
    mu = np.random.default_rng(0).standard_normal(ev_o.n_constraints)
    jr, jc = ev.jacobian_structure(); r1, c1 = ev_o.jacobian_structure1()
    print("jac structure equal:", np.array_equal(jr, r1) and np.array_equal(jc, c1))
    hr, hc = ev.hessian_lagrangian_structure(); r1h, c1h = ev_o.hessian_structure1()
    print("hess structure equal:", np.array_equal(hr, r1h) and np.array_equal(hc, c1h))
    def show(tag, got, ref, rows=None, cols=None):
        got = np.atleast_1d(got); ref = np.atleast_1d(ref)
        e = np.abs(got - ref) / np.maximum(1, np.abs(ref))
        e = np.where(np.isnan(e), np.inf, e)
        i = int(np.argmax(e))
        extra = "" if rows is None else " (row %d col %d)" % (rows[i], cols[i])
        print("%-5s err %.3e  worst idx %d got %.15g ref %.15g%s" % (tag, e.max(), i, got[i], ref[i], extra), flush=True)
    show("f", ev.eval_objective(Z), ev_o.eval_objective(Z))
    g = np.full(ev.shard.grad_len, np.nan); ev.eval_objective_gradient(g, Z); show("grad", g, ev_o.eval_objective_gradient(Z))
    c = np.full(ev.shard.cons_len, np.nan); ev.eval_constraint(c, Z); show("cons", c, ev_o.eval_constraint(Z))
    print("stats", ev.last_stats())
    j = np.full(ev.shard.jac_len, np.nan); ev.eval_constraint_jacobian(j, Z); show("jac", j, ev_o.eval_constraint_jacobian(Z), r1, c1)
    print("stats", ev.last_stats())
    if hessian:
        h = np.full(ev.shard.hess_len, np.nan); ev.eval_hessian_lagrangian(h, Z, 0.7, mu)
        show("hess", h, ev_o.eval_hessian_lagrangian(Z, 0.7, mu), r1h, c1h)
        print("stats", ev.last_stats())
    ev.close()

if __name__ == "__main__":
    dump("type1", O.make_type1_derivative_problem())
    dump("readme", O.make_readme_problem())
    dump("standard", O.make_standard_problem(N=6))
    dump("scaled 8", O.make_scaled_problem(6, 8, 2, seed=3, with_constraint=True))
    dump("scaled 64", O.make_scaled_problem(5, 64, 4, seed=4, with_constraint=True))
    dump("scaled 128", O.make_scaled_problem(4, 128, 2, seed=5), hessian=False)
