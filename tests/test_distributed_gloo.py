"""N>1 path on CPU: world_size-2 gloo processes, one knot-range shard each.

No GPU here, so each rank's slab VALUES are produced by slicing the oracle's full vectors with the
shard metadata of a structure-only engine handle; what is under test is the host logic of the
multi-GPU path: shard ranges, slab offsets/lengths from dto_shard_info, local row segments, the
all-gather of padded slabs and the objective all-reduce."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dto_amd
import dto_oracle as O
from helpers import to_engine


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, N=9):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = O.make_scaled_problem(N, 4, 2, seed=6, with_constraint=True)
        ev_o = O.OracleEvaluator(p)
        Z = p.Z0
        mu = np.random.default_rng(1).standard_normal(ev_o.n_constraints)
        full = {"jac": ev_o.eval_constraint_jacobian(Z), "hess": ev_o.eval_hessian_lagrangian(Z, 1.0, mu),
                "grad": ev_o.eval_objective_gradient(Z), "cons": ev_o.eval_constraint(Z)}
        lo, hi = dto_amd.distributed.shard_ranges(p.N, world)[rank]
        ev = dto_amd.Evaluator(to_engine(p), device=-1, k_lo=lo, k_hi=hi)
        s = ev.shard
        ok = True
        for key, (a, n) in {"jac": (s.jac_lo, s.jac_len), "hess": (s.hess_lo, s.hess_len), "grad": (s.grad_lo, s.grad_len)}.items():
            local = torch.from_numpy(full[key][a:a + n].copy())
            lens = [None] * world
            dist.all_gather_object(lens, int(n))
            got = dto_amd.distributed.allgather_slabs(local, lens)
            ok &= bool(np.array_equal(got.numpy(), full[key]))
            # the in-place form: the rank's slab sits in its slice of the full vector, the gather fills in the rest
            vec = torch.full((full[key].size,), float("nan"), dtype=torch.float64)
            vec[a:a + n] = local
            layout = dto_amd.distributed.slab_layout(a, n)
            assert layout[rank] == (a, n) and sum(x[1] for x in layout) == full[key].size
            dto_amd.distributed.gather_slabs_inplace(vec, layout)
            ok &= bool(np.array_equal(vec.numpy(), full[key]))
            # ... and on the padded allocation, where ONE equal-size all-gather moves the unequal slabs in place
            buf, vec2 = dto_amd.distributed.alloc_gather_vector(full[key].size, layout, torch.float64, "cpu")
            if key in ("jac", "hess"):
                ok &= buf is not None and buf.numel() == world * max(x[1] for x in layout)
            vec2.fill_(float("nan"))
            vec2[a:a + n] = local
            dto_amd.distributed.gather_slabs_inplace(vec2, layout, buffer=buf)
            ok &= bool(np.array_equal(vec2.numpy(), full[key]))
            # ... and the asynchronous form (one in-place broadcast per rank, work handles waited on afterwards): what a rank with
            # two engine handles uses to send one handle's slabs while the other computes
            vec3 = torch.full((full[key].size,), float("nan"), dtype=torch.float64)
            vec3[a:a + n] = local
            for w in dto_amd.distributed.gather_slabs_async(vec3, layout):
                w.wait()
            ok &= bool(np.array_equal(vec3.numpy(), full[key]))
        # constraint rows: scatter the local buffer back through the row segments, then sum over ranks
        st, ln = ev.shard_rows()
        g = torch.zeros(ev.n_constraints, dtype=torch.float64)
        for a, b in zip(st, ln):
            g[a - 1:a - 1 + b] = torch.from_numpy(full["cons"][a - 1:a - 1 + b])
        dto_amd.distributed.allreduce_sum(g)
        ok &= bool(np.array_equal(g.numpy(), full["cons"]))
        # objective partial sums: owned knots only
        pl = O.Problem(N=p.N, z=p.z, dt_idx=p.dt_idx, integrators=p.integrators, Z0=p.Z0,
                       objectives=[O.QuadraticRegularizer(t.comp_off, t.comp_dim, t.R, times1=list(range(lo, hi + 1)))
                                   for t in p.objectives])
        f = torch.tensor([O.objective_value(pl, Z)], dtype=torch.float64)
        dto_amd.distributed.allreduce_sum(f)
        ok &= bool(abs(f.item() - ev_o.eval_objective(Z)) <= 1e-15 * max(1.0, abs(f.item())))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_padded_gather_plan_shapes():
    P = dto_amd.distributed.padded_gather_plan
    assert P([(0, 7), (7, 10), (17, 10), (27, 6)]) == (10, 3, 4)       # first and last rank one boundary half-block short
    assert P([(0, 5), (5, 8)]) == (8, 3, 0) and P([(0, 8), (8, 8)]) == (8, 0, 0)
    assert P([(0, 7), (7, 10), (17, 9), (26, 6)]) is None              # an interior rank that is shorter
    assert P([(0, 7), (8, 10), (18, 10)]) is None                      # a gap
    assert P([(0, 0), (0, 0)]) is None


import pytest


@pytest.mark.parametrize("world,N", [(2, 9), (3, 9), (8, 16)])
def test_two_rank_shards_and_gather(world, N):
    """world 8 = the node the driver's scaling run uses: eight processes, two knots each, the padded in-place all-gather with eight
    chunks, the broadcast form, the row segments and the objective all-reduce -- on CPU tensors over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, N)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]
