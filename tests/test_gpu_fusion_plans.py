"""The point <-> pixel index plans of csrc/fusion.hip (u2mkd_c2l_plan, u2mkd_l2c_keys / u2mkd_csr_build /
u2mkd_l2c_finish) against the torch formulation they replace (fusion._c2l_plan_torch / _l2c_plan_torch, which the
host-side tests pin against the reference's Python loops, spvcnn_swiftnet18_spformer_tsd_full.py:448-495): indices
bit-exact, weights bit-exact (the kernels repeat the fp32 operations one for one), for one and two samples, every grid
size the four fusion stages use, incl. points no camera sees and coordinates on / outside the image border."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(n_list, ncam=6, seed=0):
    g = torch.Generator().manual_seed(seed)
    pcs, masks = [], []
    for n in n_list:
        pc = torch.rand(ncam, n, 2, generator=g) * 2.4 - 1.2           # some outside (-1, 1)
        pc[0, :7] = torch.tensor([[-1.0, -1.0], [1.0, 1.0], [0.0, 0.0], [1.0, -1.0], [0.999999, 0.5], [-1.0, 0.25], [0.3, 1.0]])
        m = (pc.abs().max(-1).values < 1.0) & (torch.rand(ncam, n, generator=g) < 0.35)
        m[:, -5:] = False                                               # points no camera sees
        pcs.append(pc.cuda())
        masks.append(m.cuda())
    return pcs, masks


@pytest.mark.parametrize('n_list', [[4000], [3000, 1777]])
@pytest.mark.parametrize('hw', [(180, 320), (90, 160), (45, 80), (23, 40), (360, 640)])
def test_c2l_plan_equals_the_torch_formulation(hip, n_list, hw):
    from u2mkd_amd import fusion
    pcs, masks = _batch(n_list, seed=hw[0])
    i_t, w_t = fusion._c2l_plan_torch(pcs, masks, *hw)
    i_h, w_h = fusion._c2l_plan(pcs, masks, *hw)
    assert torch.equal(i_h, i_t)
    assert torch.equal(w_h, w_t)


@pytest.mark.parametrize('n_list', [[4000], [3000, 1777]])
@pytest.mark.parametrize('hw', [(180, 320), (90, 160), (45, 80), (23, 40), (12, 20)])
def test_l2c_plan_equals_the_torch_formulation(hip, n_list, hw):
    from u2mkd_amd import fusion
    pcs, masks = _batch(n_list, seed=hw[1])
    (fr_t, fw_t, sd_t), (bp_t, bw_t, ss_t), nd_t = fusion._l2c_plan_torch(pcs, masks, *hw)
    pcs2, masks2 = [p.clone() for p in pcs], [m.clone() for m in masks]          # (plans are cached on masks[0])
    for _ in range(2):                                                            # second call: the cached by-source grouping
        (fr_h, fw_h, sd_h), (bp_h, bw_h, ss_h), nd_h = fusion._l2c_plan(pcs2, masks2, *hw)
        assert nd_h == nd_t and torch.equal(sd_h, sd_t) and torch.equal(ss_h, ss_t)
        live_d, live_s = int(sd_t[-1]), int(ss_t[-1])
        assert live_d == live_s == int(sum(int(m.sum()) for m in masks))
        assert torch.equal(fr_h[:live_d], fr_t[:live_d]) and torch.equal(fw_h[:live_d], fw_t[:live_d])
        assert torch.equal(bp_h[:live_s], bp_t[:live_s]) and torch.equal(bw_h[:live_s], bw_t[:live_s])


@pytest.mark.parametrize('n,c,frac', [(80000, 64, 0.4), (5003, 128, 0.9), (777, 256, 0.0), (33, 32, 1.0)])
def test_select_and_mse_equals_the_torch_formulation(hip, n, c, frac):
    """kd._select_and_mse (one pass forward, one backward: csrc/fusion.hip) against torch.where + the masked MSE written with
    torch operators (tsd_full.py:489-498): the selected rows are EQUAL, the loss and the three gradients agree to fp32
    rounding of a sum over the rows; with no point in view the loss is 0 and only the select's gradient flows."""
    from u2mkd_amd import kd
    g = torch.Generator().manual_seed(n)
    gathered = torch.randn(n, c, generator=g).cuda().requires_grad_(True)
    pseudo = torch.randn(n, c, generator=g).cuda().requires_grad_(True)
    fov = (torch.rand(n, generator=g) < frac).cuda()
    w_out = torch.randn(n, c, generator=g).cuda()

    def ref(gat, pse):
        out = torch.where(fov.unsqueeze(1), gat, pse)
        return out, kd._masked_mse(pse.double(), out.detach().double(), fov)
    o1, l1 = kd._select_and_mse(gathered, pseudo, fov)
    ((o1 * w_out).sum() + 3.0 * l1).backward()
    g1 = gathered.grad.clone(), pseudo.grad.clone()
    gathered.grad = pseudo.grad = None
    o2, l2 = ref(gathered, pseudo)
    ((o2 * w_out).sum() + 3.0 * l2.float()).backward()
    assert torch.equal(o1, o2)
    assert abs(float(l1) - float(l2)) <= 2e-6 * max(1.0, abs(float(l2)))
    assert torch.equal(g1[0], gathered.grad)
    assert float((g1[1] - pseudo.grad).abs().max()) <= 2e-6 * max(1e-6, float(pseudo.grad.abs().max()))
    # reproducible: the partial sums are merged in workgroup order
    o3, l3 = kd._select_and_mse(gathered.detach(), pseudo.detach(), fov)
    assert torch.equal(l1.detach(), l3)


@pytest.mark.gpu
@pytest.mark.parametrize('hw,n_scales,c', [((23, 40), 3, 64), ((12, 20), 1, 32), ((45, 80), 4, 16), ((9, 7), 2, 8)])
def test_l2c_combine_equals_the_torch_formulation(hw, n_scales, c, monkeypatch):
    """l2c_scatter's grids combined by csrc/pixhead.hip (l2c_combine_*: one full-resolution sum over the channel-last grids + a
    scaled transpose; backward: the scaled transpose and one gather per coarse grid) against the torch formulation it replaces
    (F.interpolate(bilinear, align_corners=True) per scale, adds, a division, a layout copy): values and the gradient with
    respect to the point features."""
    from u2mkd_amd import fusion
    torch.manual_seed(3)
    ncam, n_pts = 3, 700
    coords = [(torch.rand(ncam, n_pts, 2, device='cuda') * 2.2 - 1.1), (torch.rand(ncam, 311, 2, device='cuda') * 2.2 - 1.1)]
    masks = [torch.rand(ncam, n_pts, device='cuda') < 0.6, torch.rand(ncam, 311, device='cuda') < 0.6]
    feats = torch.randn(n_pts + 311, c, device='cuda')
    wgt = torch.randn(len(masks) * ncam, c, *hw, device='cuda')
    res = {}
    for on in (True, False):
        monkeypatch.setattr(fusion, '_L2C_COMBINE', on)
        x = feats.clone().requires_grad_(True)
        out = fusion.l2c_scatter(x, coords, [m.clone() for m in masks], hw[0], hw[1], n_scales)
        assert out.shape == wgt.shape and out.is_contiguous()
        (out * wgt).sum().backward()
        res[on] = (out.detach(), x.grad.detach())
    scale = float(res[False][0].abs().max()) + 1e-12
    assert float((res[True][0] - res[False][0]).abs().max()) <= 2e-6 * max(scale, 1.0)
    gscale = float(res[False][1].abs().max()) + 1e-12
    assert float((res[True][1] - res[False][1]).abs().max()) <= 2e-5 * max(gscale, 1.0)
