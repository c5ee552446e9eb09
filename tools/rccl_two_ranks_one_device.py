#!/usr/bin/env python3
"""Exploration: can two processes on ONE device form a 2-rank RCCL communicator through the engine's C ABI?
RCCL refuses duplicate devices inside one host; with a different NCCL_HOSTID per rank the ranks look like two hosts and talk
over the socket transport (loopback).  Usage: rccl_two_ranks_one_device.py <rank> <world> <idfile>"""
import os
import sys
import time

rank, world, idfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.environ.setdefault("NCCL_HOSTID", f"dto-rank-{rank}")
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
os.environ.setdefault("NCCL_IB_DISABLE", "1")
os.environ.setdefault("NCCL_DEBUG", "WARN")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import dto_amd
import dto_oracle as O
from dto_amd import capi
from helpers import to_engine, rel_err

if rank == 0:
    uid = dto_amd.Evaluator.comm_unique_id()
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 60:
            raise SystemExit("no id file")
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
dev = torch.device("cuda", 0)
p = O.make_scaled_problem(12, 40, 3, seed=21, with_constraint=True)
lo, hi = dto_amd.distributed.shard_ranges(p.N, world)[rank]
ev = dto_amd.Evaluator(to_engine(p), k_lo=lo, k_hi=hi)
print(rank, "comm_create ...", flush=True)
ev.comm_create(uid, rank, world)
print(rank, "comm ok", flush=True)
Z = torch.from_numpy(p.Z0).to(dev)
st = torch.cuda.current_stream(dev).cuda_stream
L = ev.gather_layout(capi.VECTOR_JACOBIAN)
buf = torch.full((L.padded_len,), float("nan"), dtype=torch.float64, device=dev)
ev.eval_jacobian_dev(Z.data_ptr(), buf.data_ptr() + 8 * (L.front_pad + L.own_lo), st)
ev.gather_dev(capi.VECTOR_JACOBIAN, buf.data_ptr(), st)
torch.cuda.synchronize()
full = buf[L.front_pad:L.front_pad + L.total].cpu().numpy()
err = rel_err(full, O.OracleEvaluator(p).eval_constraint_jacobian(p.Z0))
print(rank, "in_place", L.in_place_all_gather, "jac err", err, flush=True)
f = torch.zeros(1, dtype=torch.float64, device=dev)
ev.eval_objective_dev(Z.data_ptr(), f.data_ptr(), st)
ev.allreduce_objective_dev(f.data_ptr(), st)
torch.cuda.synchronize()
print(rank, "objective", f.item(), O.OracleEvaluator(p).eval_objective(p.Z0), flush=True)
ev.comm_destroy()
ev.close()
print(rank, "done", flush=True)
