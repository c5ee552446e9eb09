"""GPU: configs[3]'s data path with REAL engine slabs on what one box offers -- two fresh processes (torch.distributed,
gloo) share cuda:0, each owns one knot-range shard through its own engine handle, writes its Jacobian / Hessian /
gradient slab straight into its slice of the full device vector and takes part in the in-place all-gather
(dto_amd.distributed.gather_slabs_inplace -- with backend "nccl" the same call is RCCL over xGMI).  Rank 0 compares the
gathered vectors with the oracle.  The 8-GPU scaling curve itself can only be measured by the driver."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys
root = os.environ["DTO_ROOT"]
for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
import dto_amd, dto_oracle as O
from helpers import to_engine, rel_err
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
p = O.make_scaled_problem(int(os.environ["DTO_N"]), int(os.environ["DTO_STATES"]), 3, seed=21, with_constraint=True)
lo, hi = dto_amd.distributed.shard_ranges(p.N, world)[rank]
ev = dto_amd.Evaluator(to_engine(p), k_lo=lo, k_hi=hi)
s = ev.shard
Z = torch.from_numpy(p.Z0).to(dev)
mu_h = np.random.default_rng(4).standard_normal(ev.n_constraints)
mu = torch.from_numpy(mu_h).to(dev)
st = torch.cuda.current_stream(dev).cuda_stream
full = {k: torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
        for k, n in (("jac", ev.n_jacobian_entries), ("hess", ev.n_hessian_entries), ("grad", ev.n_variables))}
ev.eval_jacobian_dev(Z.data_ptr(), full["jac"][s.jac_lo:s.jac_lo + s.jac_len].data_ptr(), st)
ev.eval_hessian_dev(Z.data_ptr(), 0.8, mu.data_ptr(), full["hess"][s.hess_lo:s.hess_lo + s.hess_len].data_ptr(), st)
ev.eval_gradient_dev(Z.data_ptr(), full["grad"][s.grad_lo:s.grad_lo + s.grad_len].data_ptr(), st)
f = torch.zeros(1, dtype=torch.float64, device=dev)
ev.eval_objective_dev(Z.data_ptr(), f.data_ptr(), st)
torch.cuda.synchronize()
for k, (a, n) in (("jac", (s.jac_lo, s.jac_len)), ("hess", (s.hess_lo, s.hess_len)), ("grad", (s.grad_lo, s.grad_len))):
    dto_amd.distributed.gather_slabs_inplace(full[k], dto_amd.distributed.slab_layout(a, n))
f = f.cpu()
dto_amd.distributed.allreduce_sum(f)
# overlapped form: the rank's knots over two handles; the first half's slabs travel (asynchronous in-place broadcasts)
# while the second half computes
over = torch.full((ev.n_jacobian_entries,), float("nan"), dtype=torch.float64, device=dev)
subs, works = [], []
for a, b in dto_amd.distributed.split_range(lo, hi, 2):
    e = dto_amd.Evaluator(to_engine(p), k_lo=a, k_hi=b)
    subs.append((e, dto_amd.distributed.slab_layout(e.shard.jac_lo, e.shard.jac_len)))
for e, lay in subs:
    e.eval_jacobian_dev(Z.data_ptr(), over[e.shard.jac_lo:e.shard.jac_lo + e.shard.jac_len].data_ptr(), st)
    works += dto_amd.distributed.gather_slabs_async(over, lay)
for w in works:
    w.wait()
torch.cuda.synchronize()
for e, _ in subs:
    e.close()
ok = True
if rank == 0:
    ev_o = O.OracleEvaluator(p)
    errs = {"jac": rel_err(full["jac"].cpu().numpy(), ev_o.eval_constraint_jacobian(p.Z0)),
            "hess": rel_err(full["hess"].cpu().numpy(), ev_o.eval_hessian_lagrangian(p.Z0, 0.8, mu_h)),
            "grad": rel_err(full["grad"].cpu().numpy(), ev_o.eval_objective_gradient(p.Z0)),
            "f": rel_err(f.item(), ev_o.eval_objective(p.Z0)),
            "jac_overlapped": rel_err(over.cpu().numpy(), ev_o.eval_constraint_jacobian(p.Z0))}
    ok = errs["jac"] <= 1e-10 and errs["jac_overlapped"] <= 1e-10 and errs["grad"] <= 1e-10 and errs["f"] <= 1e-10 and errs["hess"] <= 1e-8
    print("errs", errs)
ev.close()
dist.barrier()
dist.destroy_process_group()
print("rank-ok" if ok else "rank-FAILED")
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("states,N", [(6, 9), (40, 12)])  # fused small-state path / general path (chain + sweeps)
def test_two_ranks_gather_real_engine_slabs(states, N):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, DTO_ROOT=root, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DTO_N=str(N), DTO_STATES=str(states), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "rank-ok" in o, f"rank {rank}:\n{o}"
