"""
The entry point that allocates nothing, mifwi_l2_misfit_noalloc, against the entry point it is a variant
of, mifwi_l2_misfit: same residuals and the same objective bits (both run the same two kernels; only the
home of the partial sums differs), on two calls in a row, the second with a smaller input, and within a
captured HIP graph, where a device allocation would not be allowed.
"""
import numpy as np
import pytest

from helpers import Case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_noalloc_misfit_equals_the_plan_entry_point(dtype):
    import torch
    import mifwi
    case = Case((30, 28), (10.0, 10.0), nbl=5, space_order=4, nt=421, nshots=5, nrec=23, dtype=dtype, seed=6)
    plan = case.plan(mifwi)
    rng = np.random.default_rng(8)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    for nshots in (5, 2):            # two calls in a row, the second smaller: nothing of the first may leak into it
        syn = rng.standard_normal((nshots, case.nt, 23)).astype(dtype)
        obs = rng.standard_normal((nshots, case.nt, 23)).astype(dtype)
        f_ref, r_ref = plan.l2_misfit(syn, obs, nshots=nshots)
        d_syn, d_obs = torch.from_numpy(syn).cuda(), torch.from_numpy(obs).cuda()
        d_res = torch.full((nshots, case.nt, 23), 7.0, dtype=tdt, device="cuda")
        d_f = torch.full((nshots,), -1.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        mifwi.l2_misfit_noalloc(d_syn, d_obs, d_res, d_f, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before
        assert np.array_equal(d_res.cpu().numpy(), r_ref)
        assert np.array_equal(d_f.cpu().numpy(), f_ref)
    plan.close()


def test_noalloc_misfit_inside_a_captured_graph_and_limits():
    import torch
    import mifwi
    rng = np.random.default_rng(9)
    syn = torch.from_numpy(rng.standard_normal((3, 1000)).astype(np.float32)).cuda()
    obs = torch.from_numpy(rng.standard_normal((3, 1000)).astype(np.float32)).cuda()
    res, f = torch.zeros_like(syn), torch.zeros(3, dtype=torch.float64, device="cuda")
    mifwi.l2_misfit_noalloc(syn, obs, res, f, stream=torch.cuda.current_stream().cuda_stream)   # outside capture first
    torch.cuda.synchronize()
    f_eager, r_eager = f.clone(), res.clone()
    res.zero_(), f.zero_()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            mifwi.l2_misfit_noalloc(syn, obs, res, f, stream=s.cuda_stream)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(res, r_eager) and torch.equal(f, f_eager)
    ref = 0.5 * ((syn - obs).double() ** 2).sum(1)      # the residual is formed in the data's own precision
    assert torch.allclose(f, ref, rtol=1e-12, atol=0)
    # beyond the fixed scratch array: a clear error, nothing launched
    big = torch.zeros((257, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(mifwi.MifwiError, match="256"):
        mifwi.l2_misfit_noalloc(big, big, torch.zeros_like(big), torch.zeros(257, dtype=torch.float64, device="cuda"))
    with pytest.raises(mifwi.MifwiError):
        lib = mifwi.load_library()
        rc = lib.mifwi_l2_misfit_noalloc(0, 1, 4, None, None, None, None, None)
        if rc != 0:
            raise mifwi.MifwiError(rc, lib.mifwi_last_error().decode())
