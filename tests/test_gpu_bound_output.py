"""GPU: bound outputs (dto_bind_output_dev, include/dto_engine.h).  A solver in GPU mode hands the same device vector to
eval_constraint_jacobian / eval_hessian_lagrangian every iteration; once a call has written it in full, later calls into the same
pointer skip the zero-fill of the call-invariant entries (structural zeros, identity blocks) and clear only the runs kernels
accumulate into.  Checked bit for bit against an unbound handle: over several points, with every entry that can change poisoned
between the calls, for the general path, the small-state path, the device time-dependent integrator and host-merged terms."""
import numpy as np
import pytest

import dto_oracle as O
from helpers import to_engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("make", [lambda: O.make_scaled_problem(14, 40, 3, seed=4, with_constraint=True),
                                  lambda: O.make_standard_problem(N=10),
                                  lambda: O.make_tdb_problem(N=6, n=4, m=2, order=1, seed=5, substeps=8, with_derivative=False),
                                  lambda: O.make_scaled_problem(70, 100, 2, seed=9)],
                         ids=["general-path", "small-state+derivative", "time-dependent", "fused-sweep-128"])
def test_bound_vectors_keep_their_constants_and_track_the_point(make):
    import torch
    import dto_amd
    from dto_amd import capi
    p = make()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    ev = dto_amd.Evaluator(to_engine(p))       # bound
    ref = dto_amd.Evaluator(to_engine(p))      # unbound: every call writes everything
    try:
        rng = np.random.default_rng(0)
        Zs = [p.Z0 + 0.05 * k * rng.standard_normal(p.n_vars) for k in range(3)]
        mus = [rng.standard_normal(ev.n_constraints) for _ in range(3)]
        dZ = [torch.from_numpy(Z).to(dev) for Z in Zs]
        dmu = [torch.from_numpy(m).to(dev) for m in mus]

        def fresh(k, which):
            o = torch.full((ref.shard.jac_len if which == "jac" else ref.shard.hess_len,), float("nan"), dtype=torch.float64, device=dev)
            if which == "jac":
                ref.eval_jacobian_dev(dZ[k].data_ptr(), o.data_ptr(), st)
            else:
                ref.eval_hessian_dev(dZ[k].data_ptr(), 0.7, dmu[k].data_ptr(), o.data_ptr(), st)
            torch.cuda.synchronize()
            return o

        for which, vec in (("jac", capi.VECTOR_JACOBIAN), ("hess", capi.VECTOR_HESSIAN)):
            want = [fresh(k, which) for k in range(3)]
            varies = (want[0] != want[1]) | (want[0] != want[2])          # entries that demonstrably change from call to call
            assert 0 < int(varies.sum()) < want[0].numel()
            buf = torch.full_like(want[0], float("nan"))
            ev.bind_output_dev(vec, buf.data_ptr())
            for k in (0, 1, 2, 0, 2):
                if which == "jac":
                    ev.eval_jacobian_dev(dZ[k].data_ptr(), buf.data_ptr(), st)
                else:
                    ev.eval_hessian_dev(dZ[k].data_ptr(), 0.7, dmu[k].data_ptr(), buf.data_ptr(), st)
                torch.cuda.synchronize()
                assert torch.equal(buf, want[k]), (which, k, int((buf != want[k]).sum()))
                buf[varies] = float("nan")                                  # the caller may overwrite what varies
            # another pointer is written in full; unbinding re-arms the full write for the old one
            other = torch.full_like(want[0], float("nan"))
            if which == "jac":
                ev.eval_jacobian_dev(dZ[1].data_ptr(), other.data_ptr(), st)
            else:
                ev.eval_hessian_dev(dZ[1].data_ptr(), 0.7, dmu[1].data_ptr(), other.data_ptr(), st)
            torch.cuda.synchronize()
            assert torch.equal(other, want[1])
            ev.bind_output_dev(vec, 0)
            buf.fill_(float("nan"))
            if which == "jac":
                ev.eval_jacobian_dev(dZ[2].data_ptr(), buf.data_ptr(), st)
            else:
                ev.eval_hessian_dev(dZ[2].data_ptr(), 0.7, dmu[2].data_ptr(), buf.data_ptr(), st)
            torch.cuda.synchronize()
            assert torch.equal(buf, want[2])
    finally:
        ev.close(); ref.close()
