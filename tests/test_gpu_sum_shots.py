"""
mifwi_sum_shots and mifwi_precondition against the host arithmetic of the reference's fwi_obj_multi
(fwi.py:164-171,190-204): crop, per-shot mute, shot sum in float64, illumination preconditioner and mask.
Bit-exact: the device kernels round the product before the sum and add in shot order, like numpy on float64 arrays.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _host(g, il, mute, crop, precond, mask):
    grad = np.zeros(mute.shape[1:])
    illum = np.zeros(mute.shape[1:])
    for k in range(g.shape[0]):
        grad += g[k][crop] * mute[k]
        illum += il[k][crop] * mute[k]
    if precond:
        grad /= np.sqrt(illum + 1e-30)
    if mask is not None:
        grad *= mask
    return grad, illum


@pytest.mark.parametrize("shape,lo,phys,dtype", [
    ((61, 47), (10, 10), (41, 27), np.float32),
    ((61, 47), (10, 0), (41, 37), np.float32),        # free surface: no layer above the first row of the last axis
    ((23, 19, 17), (4, 4, 0), (15, 11, 13), np.float64),
    ((9, 8), (0, 0), (9, 8), np.float32),             # no layer at all
])
def test_sum_shots_and_precondition_match_numpy(shape, lo, phys, dtype):
    import torch
    from mifwi import Plan
    rng = np.random.default_rng(5)
    ns = 5
    plan = Plan(shape, (10.0,) * len(shape), (0.0,) * len(shape), 4, 8, 1.0, 1, 3, max_shots=ns, dtype=dtype)
    g = rng.standard_normal((ns,) + shape).astype(dtype)
    il = (rng.random((ns,) + shape) * 3).astype(dtype)
    mute = rng.random((ns,) + phys)
    mask = rng.random(phys)
    crop = tuple(slice(a, a + n) for a, n in zip(lo, phys))
    g_d, il_d = torch.as_tensor(g).cuda(), torch.as_tensor(il).cuda()
    mute_d, mask_d = torch.as_tensor(mute).cuda(), torch.as_tensor(mask).cuda()

    # host outputs
    gs, ils = plan.sum_shots(g_d, il_d, mute_d, lo, phys)
    want_g, want_il = _host(g, il, mute, crop, False, None)
    assert np.array_equal(gs, want_g) and np.array_equal(ils, want_il)

    # device outputs, then the preconditioner with and without its optional parts
    n = int(np.prod(phys))
    packed = torch.empty(1 + 2 * n, dtype=torch.float64, device='cuda')
    plan.sum_shots(g_d, il_d, mute_d, lo, phys, g_sum=packed[1:1 + n], il_sum=packed[1 + n:])
    assert np.array_equal(packed[1:1 + n].cpu().numpy().reshape(phys), want_g)
    for precond, use_mask in ((True, True), (True, False), (False, True), (False, False)):
        out = plan.precondition(packed[1:1 + n], packed[1 + n:] if precond else None,
                                mask_d.reshape(-1) if use_mask else None)
        want, _ = _host(g, il, mute, crop, precond, mask if use_mask else None)
        assert np.array_equal(out.reshape(phys), want), (precond, use_mask)

    # no mute table, no illumination
    gs2, none = plan.sum_shots(g_d, None, None, lo, phys)
    assert none is None
    want2 = np.zeros(phys)
    for k in range(ns):
        want2 += g[k][crop]
    assert np.array_equal(gs2, want2)
    plan.close()


def test_sum_shots_rejects_bad_windows_and_host_inputs():
    import torch
    from mifwi import Plan, MifwiError
    plan = Plan((20, 20), (10.0, 10.0), (0.0, 0.0), 4, 8, 1.0, 1, 3, max_shots=2)
    g = torch.zeros((2, 20, 20), device='cuda')
    with pytest.raises(MifwiError):
        plan.sum_shots(g, None, None, (5, 5), (16, 10))          # window runs past the grid
    with pytest.raises(MifwiError):
        plan.sum_shots(g, None, None, (-1, 0), (10, 10))
    with pytest.raises(MifwiError):
        plan.sum_shots(np.zeros((2, 20, 20), np.float32), None, None, (0, 0), (10, 10))   # host gradient
    plan.close()
