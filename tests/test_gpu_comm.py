"""GPU: the engine's collectives through the C ABI (include/dto_engine.h "Multi-GPU", csrc/dto_comm.*) with REAL RCCL ranks.

A one-GPU box offers one device, and RCCL refuses two ranks of one host on the same device.  With a different NCCL_HOSTID per
process the two ranks look like two hosts and RCCL connects them over its socket transport (loopback): same communicator
set-up, same ncclAllGather / ncclBroadcast / ncclAllReduce calls on device memory as over xGMI, only the wire differs.  Each
rank owns one knot-range shard through its own engine handle, writes its slabs straight into the padded full vectors
(dto_get_gather_layout) and takes part in dto_gather_*_dev; no torch.distributed anywhere -- the 128-byte id travels through a
file.  Every rank compares all gathered vectors with the oracle.  The xGMI curve itself is the driver's 8-GPU run."""
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys, time
root = os.environ["DTO_ROOT"]
for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import dto_amd, dto_oracle as O
from dto_amd import capi
from helpers import to_engine, rel_err
rank, world, idfile = int(os.environ["DTO_RANK"]), int(os.environ["DTO_WORLD"]), os.environ["DTO_IDFILE"]

def exchange(tag):
    # the 128 bytes of ncclGetUniqueId from rank 0 to the others, through the file system
    path = idfile + tag
    if rank == 0:
        uid = dto_amd.Evaluator.comm_unique_id()
        open(path + ".tmp", "wb").write(uid)
        os.replace(path + ".tmp", path)
        return uid
    t0 = time.time()
    while not os.path.exists(path):
        assert time.time() - t0 < 120, "no id from rank 0"
        time.sleep(0.02)
    return open(path, "rb").read()

dev = torch.device("cuda", 0)
p = O.make_scaled_problem(int(os.environ["DTO_N"]), int(os.environ["DTO_STATES"]), 3, seed=21, with_constraint=True)
ev_o = O.OracleEvaluator(p)
lo, hi = dto_amd.distributed.shard_ranges(p.N, world)[rank]
ev = dto_amd.Evaluator(to_engine(p), k_lo=lo, k_hi=hi)
ev.comm_create(exchange("a"), rank, world)
Z = torch.from_numpy(p.Z0).to(dev)
mu_h = np.random.default_rng(4).standard_normal(ev.n_constraints)
mu = torch.from_numpy(mu_h).to(dev)
st = torch.cuda.current_stream(dev).cuda_stream
errs = {}

def padded(vec):
    L = ev.gather_layout(vec)
    assert L.in_place_all_gather == int(os.environ["DTO_EXPECT_INPLACE"]) and L.world == world
    buf = torch.full((L.padded_len,), float("nan"), dtype=torch.float64, device=dev)
    return L, buf, buf.data_ptr() + 8 * (L.front_pad + L.own_lo)

Lj, bj, pj = padded(capi.VECTOR_JACOBIAN)
Lh, bh, ph = padded(capi.VECTOR_HESSIAN)
Lg, bg, pg = padded(capi.VECTOR_GRADIENT)
for _ in range(2):  # twice: the communicator is reused, padding survives
    ev.eval_jacobian_dev(Z.data_ptr(), pj, st)
    ev.gather_dev(capi.VECTOR_JACOBIAN, bj.data_ptr(), st)
    ev.eval_hessian_dev(Z.data_ptr(), 0.8, mu.data_ptr(), ph, st)
    ev.gather_dev(capi.VECTOR_HESSIAN, bh.data_ptr(), st)
    ev.eval_gradient_dev(Z.data_ptr(), pg, st)
    ev.gather_dev(capi.VECTOR_GRADIENT, bg.data_ptr(), st)
g_loc = torch.empty(ev.shard.cons_len, dtype=torch.float64, device=dev)
g_full = torch.full((ev.n_constraints,), float("nan"), dtype=torch.float64, device=dev)
ev.eval_constraint_dev(Z.data_ptr(), g_loc.data_ptr(), st)
ev.gather_constraint_dev(g_loc.data_ptr(), g_full.data_ptr(), st)
f = torch.zeros(1, dtype=torch.float64, device=dev)
ev.eval_objective_dev(Z.data_ptr(), f.data_ptr(), st)
ev.allreduce_objective_dev(f.data_ptr(), st)
torch.cuda.synchronize()
cut = lambda L, b: b[L.front_pad:L.front_pad + L.total].cpu().numpy()
errs["jac"] = rel_err(cut(Lj, bj), ev_o.eval_constraint_jacobian(p.Z0))
errs["hess"] = rel_err(cut(Lh, bh), ev_o.eval_hessian_lagrangian(p.Z0, 0.8, mu_h))
errs["grad"] = rel_err(cut(Lg, bg), ev_o.eval_objective_gradient(p.Z0))
errs["cons"] = rel_err(g_full.cpu().numpy(), ev_o.eval_constraint(p.Z0))
errs["f"] = rel_err(f.item(), ev_o.eval_objective(p.Z0))
# the rank's own slab went through the collective untouched (in place): bit-identical to a private evaluation
own = torch.empty(ev.shard.jac_len, dtype=torch.float64, device=dev)
ev.eval_jacobian_dev(Z.data_ptr(), own.data_ptr(), st)
torch.cuda.synchronize()
a = Lj.front_pad + Lj.own_lo
errs["own_bits"] = 0.0 if torch.equal(own, bj[a:a + Lj.own_len]) else 1.0

# overlapped form: the rank's knots over two handles, each with its own communicator; their slabs do not tile the vector,
# so the gather is one in-place broadcast per rank inside a group call (in_place_all_gather = 0)
over = torch.full((ev.n_jacobian_entries,), float("nan"), dtype=torch.float64, device=dev)
subs = []
for i, (a_, b_) in enumerate(dto_amd.distributed.split_range(lo, hi, 2)):
    e = dto_amd.Evaluator(to_engine(p), k_lo=a_, k_hi=b_)
    e.comm_create(exchange("b%d" % i), rank, world)
    L = e.gather_layout(capi.VECTOR_JACOBIAN)
    assert L.in_place_all_gather == 0 and L.front_pad == 0 and L.padded_len == ev.n_jacobian_entries
    subs.append((e, L))
for e, L in subs:
    e.eval_jacobian_dev(Z.data_ptr(), over.data_ptr() + 8 * L.own_lo, st)
    e.gather_dev(capi.VECTOR_JACOBIAN, over.data_ptr(), st)
torch.cuda.synchronize()
errs["jac_two_handles"] = rel_err(over.cpu().numpy(), ev_o.eval_constraint_jacobian(p.Z0))
for e, _ in subs:
    e.comm_destroy()
    e.close()
ev.comm_destroy()
ev.close()
print("errs", errs)
ok = (max(errs[k] for k in ("jac", "grad", "cons", "f", "jac_two_handles")) <= 1e-10 and errs["hess"] <= 1e-8 and errs["own_bits"] == 0.0)
print("rank-ok" if ok else "rank-FAILED")
"""


def _run(world, states, N, in_place=1):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for rank in range(world):
            env = dict(os.environ, DTO_ROOT=root, DTO_RANK=str(rank), DTO_WORLD=str(world), DTO_IDFILE=os.path.join(tmp, "uid"),
                       DTO_N=str(N), DTO_STATES=str(states), DTO_EXPECT_INPLACE=str(in_place), HSA_ENABLE_IPC_MODE_LEGACY="0",
                       # one device, two RCCL ranks: each process poses as a host of its own and talks over loopback sockets
                       NCCL_HOSTID=f"dto-test-rank-{rank}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_DEBUG="WARN")
            procs.append(subprocess.Popen([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=300)[0])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "rank-ok" in o, f"rank {rank}:\n{o[-3000:]}"


def test_single_rank_communicator():
    _run(1, 6, 9)


@pytest.mark.parametrize("states,N", [(6, 9), (40, 12)])  # fused small-state path / general path (chain + sweeps)
def test_two_rccl_ranks_gather_through_the_c_abi(states, N):
    _run(2, states, N)


def test_three_rccl_ranks():
    _run(3, 6, 12)


def test_unequal_knot_counts_take_the_broadcast_form():
    # 10 knots over 3 ranks = 4 + 3 + 3: the slabs are not equally long, no padded layout exists, and the same entry points
    # move them with one in-place broadcast per rank
    _run(3, 6, 10, in_place=0)
