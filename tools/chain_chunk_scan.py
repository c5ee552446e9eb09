"""Jacobian time at the headline shape as a function of the chain chunk (option chain_chunk: intervals whose nine chain matrices
are live at once).  A chunk of C intervals keeps 9 C x 512 KB live: C <= 48 fits the 256 MB Infinity Cache.
usage: python tools/chain_chunk_scan.py [chunk ...]      (run under rocprofv3 --kernel-trace / --pmc for per-kernel figures)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dto_amd

chunks = [int(a) for a in sys.argv[1:]] or [0, 1024, 512, 256, 128, 64, 48, 32]
prob = dto_amd.host.synthetic.make_scaled_problem(2000, 256, 4, seed=42)
ev = dto_amd.Evaluator(prob, eval_hessian=False)
dev = torch.device("cuda", 0)
Z = torch.from_numpy(prob.trajectory.vec()).to(dev)
st = torch.cuda.current_stream(dev).cuda_stream
out = torch.empty(ev.shard.jac_len, dtype=torch.float64, device=dev)
ref = None
for c in chunks:
    ev.set_option("chain_chunk", c)
    for _ in range(2):
        ev.eval_jacobian_dev(Z.data_ptr(), out.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ev.eval_jacobian_dev(Z.data_ptr(), out.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    if ref is None:
        ref = out.clone()
    print(f"chain_chunk={c}: {dt * 1e3:.3f} ms per Jacobian, max |diff| to the first setting {float((out - ref).abs().max()):.2e}", flush=True)
ev.close()
