"""Point <-> pixel transfer and fusion blocks of the multi-modal student (rows a14-a16;
core/models/fusion_blocks.py:9-153, 241-278 and the two Python loops of
core/models/nuscenes/spvcnn_swiftnet18_spformer_tsd_full.py:448-495).

The reference walks samples x cameras x scales in Python with ``torch.unique``,
``sparse_coo_tensor(...).to_dense()`` and boolean-mask writes (a host sync per camera).  Here
both transfers are batched tensor programs without host synchronisation:

* :func:`l2c_scatter`  -- point features -> per-camera feature maps: per scale ONE pixel-mean
  over ALL (sample, camera) pairs into a dense [B*ncam, H_c, W_c, C] grid, bilinear up-sampling,
  mean over scales;
* :func:`c2l_gather`   -- camera feature maps -> point features: bilinear sample from the ONE
  camera the reference's "later camera overwrites" rule picks for each point.
On the HIP device both are the point<->voxel kernels in disguise: the pixel mean is a CSR
segment-sum (entries grouped by destination pixel, like spvoxelize), the bilinear gather is
spdevoxelize with 4 corners, and each one's gradient is the other kind of segment-sum -- no
atomics, bitwise reproducible.  (torch's ``index_add_`` / ``grid_sample`` backward resolve pixel
collisions with float atomics: 46 ms + 152 ms of a 392 ms KD step on MI355X, profiles/.)  The
plain torch formulations of the same maths are kept as ``l2c_scatter_torch`` / ``c2l_gather_torch``
(pinned against the reference's loops by the host-side tests); the product functions run on the
HIP device only.
Fusion modules keep the reference's parameter names."""
import os

import torch
import torch.nn.functional as F
from torch import nn

from .camera import BatchNorm2d, bn_act

__all__ = ['IA_Layer', 'Atten_Fusion_Conv', 'L2CAILayer', 'L2CFusion', 'feature_gather', 'c2l_gather',
           'l2c_scatter', 'feature_fetch', 'l2c_scatter_torch', 'c2l_gather_torch']


def _rows_linear(x, weight, bias, before=None):
    """x [N, Cin] @ weight[Cout, Cin(,1)]^T + bias on the MFMA pipeline when the shapes allow (k = 1 Conv1d
    weights are [Cout, Cin, 1]).  ``before`` = the BatchNorm the result goes straight into, if any."""
    w = weight.squeeze(-1) if weight.dim() == 3 else weight
    if x.is_cuda and w.shape[1] % 4 == 0:
        from .torchsparse.nn import functional as spf
        return spf.linear(x, w, bias, bias_feeds_batchnorm=before is not None and before.training)
    return F.linear(x, w, bias)


def _rows_bn(bn, x, relu=False):
    """BatchNorm1d (+ReLU) over the rows of [N, C] (fused HIP pass on the device; SyncBatchNorm aware)."""
    if x.is_cuda and x.shape[1] % 4 == 0:
        from .torchsparse.nn import functional as spf
        return spf.batch_norm(x, bn, relu)
    y = bn(x)
    return F.relu(y) if relu else y


class IA_Layer(nn.Module):
    """fusion_blocks.py:9-46 with the reference's parameters (Conv1d / BatchNorm1d / Linear), evaluated
    on row-major [N, C] features: a k=1 Conv1d over [1, C, N] is a Linear over rows and BatchNorm1d over
    [1, C, N] is BatchNorm1d over [N, C], so none of the reference's transposes (strided copies and
    strided gradients on 80k x 256 tensors) is needed."""

    def __init__(self, channels):
        super().__init__()
        self.ic, self.pc = channels
        rc = self.pc // 4
        self.conv1 = nn.Sequential(nn.Conv1d(self.ic, self.pc, 1), nn.BatchNorm1d(self.pc), nn.ReLU(True))
        self.fc1 = nn.Sequential(nn.BatchNorm1d(self.ic), nn.ReLU(True), nn.Linear(self.ic, rc))
        self.fc2 = nn.Linear(self.pc, rc)
        self.fc3 = nn.Linear(rc, 1)

    def forward(self, img_feats, point_feats):
        """img_feats [N, ic], point_feats [N, pc] -> attention-weighted image features [N, pc]."""
        img_feats = img_feats.contiguous()
        ri = _rows_linear(_rows_bn(self.fc1[0], img_feats, relu=True), self.fc1[2].weight, self.fc1[2].bias)
        rp = _rows_linear(point_feats.contiguous(), self.fc2.weight, self.fc2.bias)
        # (fc3 has ONE output feature: rocBLAS answers its weight gradient -- a [1, rc] product over K = N rows -- with a
        # single-tile split-K kernel of ~0.3 ms; the row operator pads the output to a column block and is deterministic)
        att = torch.sigmoid(_rows_linear(torch.tanh(ri + rp), self.fc3.weight, self.fc3.bias))      # [N, 1]
        img = _rows_bn(self.conv1[1], _rows_linear(img_feats, self.conv1[0].weight, self.conv1[0].bias, before=self.conv1[1]), relu=True)
        return img * att


class Atten_Fusion_Conv(nn.Module):
    def __init__(self, inplanes_I, inplanes_P, outplanes):
        super().__init__()
        self.ai_layer = IA_Layer(channels=[inplanes_I, inplanes_P])
        self.conv1 = nn.Conv1d(inplanes_P + inplanes_P, outplanes, 1)
        self.bn1 = nn.BatchNorm1d(outplanes)

    def forward(self, point_features, img_features):
        """[N, P], [N, I] -> [N, outplanes] (fusion_blocks.py:49-68, row-major)."""
        img = self.ai_layer(img_features, point_features)
        fused = torch.cat([point_features, img], dim=1)
        return _rows_bn(self.bn1, _rows_linear(fused, self.conv1.weight, self.conv1.bias, before=self.bn1), relu=True)


class L2CAILayer(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.ic, self.pc = channels
        rc = self.ic // 4
        self.conv1 = nn.Sequential(nn.Conv2d(self.pc, self.ic, 1), BatchNorm2d(self.ic), nn.ReLU(True))
        self.fc1 = nn.Conv2d(self.ic, rc, kernel_size=1)
        self.fc2 = nn.Conv2d(self.pc, rc, kernel_size=1)
        self.fc3 = nn.Conv2d(rc, 1, kernel_size=1)

    def forward(self, img_feats, point_feats):
        att = torch.sigmoid(self.fc3(torch.tanh(self.fc1(img_feats.contiguous()) + self.fc2(point_feats.contiguous()))))
        return bn_act(self.conv1[1], self.conv1[0](point_feats), relu=True) * att


class L2CFusion(nn.Module):
    def __init__(self, inplanes_I, inplanes_P, outplanes):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes_I + inplanes_I, outplanes, kernel_size=1)
        self.bn1 = BatchNorm2d(outplanes)
        self.l2c_ai_layer = L2CAILayer(channels=[inplanes_I, inplanes_P])

    def forward(self, point_features, img_features):
        l2c = self.l2c_ai_layer(img_features, point_features)
        fused = self.bn1(self.conv1(torch.cat([img_features, l2c], dim=1)))
        return F.relu(fused), fused


def feature_gather(feature_map, xy, mode='bilinear'):
    """[B,C,H,W] sampled at xy [B,N,2] in [-1,1] (width, height) -> [B,C,N] (fusion_blocks.py:241-254)."""
    return F.grid_sample(feature_map, xy.unsqueeze(1), padding_mode='zeros', align_corners=True, mode=mode).squeeze(2)


def _last_camera(mask):
    """Index of the LAST camera that sees each point (the reference writes cameras in order,
    later ones overwrite: tsd_full.py:492-493) and whether any does.  mask bool [ncam, N]."""
    ncam = mask.shape[0]
    rank = torch.arange(1, ncam + 1, device=mask.device).view(-1, 1) * mask.to(torch.int64)
    best = rank.max(0)[0]
    return (best - 1).clamp(min=0), best > 0


class _NchwToRows(torch.autograd.Function):
    """[B, ncam, C, h, w] -> channel-last rows [B*ncam*h*w, C] and back, with a PLAIN NCHW gradient: the
    autograd of permute().reshape() would hand the camera branch a channel-last-strided gradient, and
    every backward kernel downstream (BatchNorm, adds, copies) then runs its slow strided variant."""

    @staticmethod
    def forward(ctx, fm):
        ctx.shape = fm.shape
        B, ncam, C, h, w = fm.shape
        if fm.is_cuda and fm.dtype == torch.float32 and fm.is_contiguous() and B * ncam <= 65535 and fm.numel():
            from . import _lib as L
            out = torch.empty(B * ncam * h * w, C, dtype=torch.float32, device=fm.device)
            L.call('u2mkd_transpose_batched', L.ptr(fm), L.ptr(out), B * ncam, C, h * w, L.stream())
            return out
        return fm.permute(0, 1, 3, 4, 2).contiguous().view(B * ncam * h * w, C)

    @staticmethod
    def backward(ctx, g):
        B, ncam, C, h, w = ctx.shape
        if g.is_cuda and g.dtype == torch.float32 and B * ncam <= 65535 and g.numel():
            from . import _lib as L
            g = g.contiguous()
            out = torch.empty(B, ncam, C, h, w, dtype=torch.float32, device=g.device)
            L.call('u2mkd_transpose_batched', L.ptr(g), L.ptr(out), B * ncam, h * w, C, L.stream())
            return out
        return g.view(B, ncam, h, w, C).permute(0, 1, 4, 2, 3).contiguous()


def _c2l_plan_torch(pixel_coordinates, masks, h, w):
    """idx int32 [N, 8] / weights f32 [N, 8]: the 4 bilinear corners (align_corners=True, zero
    padding; slots 4..7 unused) of every point in the feature map of the LAST camera seeing it,
    as rows of the [B*ncam*h*w, C] channel-last feature matrix.  (The torch formulation: what the HIP plan
    kernel is held against, bit for bit, in tests/test_gpu_fusion_plans.py.)"""
    ncam = masks[0].shape[0]
    idx, wts = [], []
    for b, (coord, mask) in enumerate(zip(pixel_coordinates, masks)):
        cam, seen = _last_camera(mask)
        xy = coord.gather(0, cam.view(1, -1, 1).expand(1, -1, 2)).squeeze(0).float()     # [N, 2] of the chosen camera
        x = (xy[:, 0] + 1.0) * 0.5 * (w - 1)
        y = (xy[:, 1] + 1.0) * 0.5 * (h - 1)
        x0, y0 = torch.floor(x), torch.floor(y)
        fx, fy = x - x0, y - y0
        base = (b * ncam + cam) * (h * w)
        ii, ww = [], []
        for dy, wy in ((0, 1.0 - fy), (1, fy)):
            for dx, wx in ((0, 1.0 - fx), (1, fx)):
                xi, yi = x0.long() + dx, y0.long() + dy
                ok = seen & (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
                ii.append(torch.where(ok, base + yi * w + xi, torch.full_like(xi, -1)))
                ww.append(torch.where(ok, wx * wy, torch.zeros_like(wx)))
        pad_i = torch.full_like(ii[0], -1)
        pad_w = torch.zeros_like(ww[0])
        idx.append(torch.stack(ii + [pad_i] * 4, 1))
        wts.append(torch.stack(ww + [pad_w] * 4, 1))
    return torch.cat(idx, 0).int().contiguous(), torch.cat(wts, 0).float().contiguous()


def _mask_bytes(m):
    return m.contiguous().view(torch.uint8) if m.dtype == torch.bool else m.contiguous().to(torch.uint8)


def _c2l_plan(pixel_coordinates, masks, h, w):
    """The same plan from ONE launch per sample (csrc/fusion.hip, u2mkd_c2l_plan)."""
    from . import _lib as L
    ncam = masks[0].shape[0]
    n_tot = sum(m.shape[1] for m in masks)
    dev = masks[0].device
    idx8 = torch.empty(n_tot, 8, dtype=torch.int32, device=dev)
    w8 = torch.empty(n_tot, 8, dtype=torch.float32, device=dev)
    cur = 0
    for b, (coord, mask) in enumerate(zip(pixel_coordinates, masks)):
        n = mask.shape[1]
        pc, mb = coord.contiguous().float(), _mask_bytes(mask)
        L.call('u2mkd_c2l_plan', L.ptr(pc), L.ptr(mb), ncam, n, b, h, w, L.ptr(idx8[cur:]), L.ptr(w8[cur:]), L.stream())
        cur += n
    return idx8, w8


def c2l_plan(pixel_coordinates, masks, h, w):
    """(corner rows int32 [sum N_b, 8], corner weights f32 [sum N_b, 8]) of the camera -> LiDAR gather on h x w maps, cached on
    ``masks[0]`` (rebuilt when a mask or a coordinate tensor changed)."""
    from .torchsparse.nn import functional as spf
    return spf._plan(masks[0], 'c2l_%d_%d' % (h, w), lambda: _c2l_plan(pixel_coordinates, masks, h, w),
                     *masks[1:], *pixel_coordinates)


def l2c_plan(pixel_coordinates, masks, ch, cw):
    """(forward CSR, backward CSR, pixels) of the LiDAR -> camera pixel-mean map on a ch x cw grid, cached on ``masks[0]``."""
    from .torchsparse.nn import functional as spf
    return spf._plan(masks[0], 'l2c_%d_%d' % (ch, cw), lambda: _l2c_plan(pixel_coordinates, masks, ch, cw),
                     *masks[1:], *pixel_coordinates)


def _l2c_grids(ifh, ifw, n_scales):
    cnt = 1
    for _ in range(n_scales):
        yield int(round(float(ifh) / cnt + 0.01)), int(round(float(ifw) / cnt + 0.01))
        cnt *= 2


def prefetch_plans(pixel_coordinates, masks, shapes):
    """The point <-> pixel plans of a batch for the fusion points ``shapes`` = [(map height, map width, l2c scales), ..], built
    now on the current stream (a trainer preparing the next batch: kd.StudentMSP2IFM.prefetch_plans): the gather plan of
    c2l_gather with the plan of its backward, and the pixel-mean maps of l2c_scatter, all of which depend on the batch's
    pixel coordinates and masks only."""
    from .torchsparse.nn import functional as spf
    n_img = len(masks) * masks[0].shape[0]
    for ifh, ifw, n_scales in shapes:
        idx8, w8 = c2l_plan(pixel_coordinates, masks, ifh, ifw)
        spf.devoxelize_plan(idx8, w8, n_img * ifh * ifw)
        for ch, cw in _l2c_grids(ifh, ifw, n_scales):
            l2c_plan(pixel_coordinates, masks, ch, cw)


def c2l_gather(feature_maps, pixel_coordinates, masks):
    """Camera -> LiDAR gather.  feature_maps [B, ncam, C, h, w]; per sample coordinates
    [ncam, N_b, 2] and masks [ncam, N_b].  Returns [sum N_b, C], zeros outside every camera."""
    from . import _lib
    _lib.require_cuda(feature_maps)                      # HIP device only; there is no CPU fallback
    from .torchsparse.nn import functional as spf
    B, ncam, C, h, w = feature_maps.shape
    idx8, w8 = c2l_plan(pixel_coordinates, masks, h, w)
    rows = _NchwToRows.apply(feature_maps)
    if C % 4:                                   # e.g. the 17-class logit map: pad rows to whole 16-byte segments
        rows = F.pad(rows, (0, 4 - C % 4))
    return spf.spdevoxelize(rows, idx8, w8)[:, :C]


def c2l_gather_torch(feature_maps, pixel_coordinates, masks):
    """The same gather written with torch ops only (grid_sample over every camera + a mask-priority
    select): the formulation the host-side tests pin against the reference's Python loop."""
    out = []
    for fmap, coord, mask in zip(feature_maps, pixel_coordinates, masks):
        sampled = feature_gather(fmap, coord)                       # [ncam, C, N_b]
        cam, seen = _last_camera(mask)
        pick = sampled.gather(0, cam.view(1, 1, -1).expand(1, sampled.shape[1], -1)).squeeze(0)   # [C, N_b]
        out.append((pick * seen.to(pick.dtype).unsqueeze(0)).t())
    return torch.cat(out, dim=0)


def feature_fetch(masks, pixel_coordinates, imfeats, mode='bilinear'):
    """Feature_Fetch (fusion_blocks.py:257-278) on the current device."""
    return c2l_gather(imfeats, pixel_coordinates, masks)


class _SegmentMap(torch.autograd.Function):
    """out[d] = sum_e w_e * src[row_e] over the entries e of destination d (CSR by destination);
    gradient: gsrc[r] = sum_e w_e * g[dst_e] over the entries of source row r (CSR by source)."""

    @staticmethod
    def forward(ctx, src, fwd, bwd, n_dst):
        from .torchsparse.nn import functional as spf
        ctx.bwd, ctx.n_src, ctx.in_dtype = bwd, src.shape[0], src.dtype
        erow, ew, seg = fwd
        # bf16 rows stay bf16 (bf16 storage under autocast); anything else is moved as fp32
        return spf._segment_sum(spf._rows(src, src.dtype == torch.bfloat16), erow, ew, seg, n_dst, False)

    @staticmethod
    def backward(ctx, g):
        from .torchsparse.nn import functional as spf
        erow, ew, seg = ctx.bwd
        gs = spf._segment_sum(spf._rows(g, g.dtype == torch.bfloat16), erow, ew, seg, ctx.n_src, False)
        return (gs if gs.dtype == ctx.in_dtype else gs.to(ctx.in_dtype)), None, None, None


def _flat_entries(pixel_coordinates, masks):
    """(coords [E,2], mask [E], camera slot [E], point row [E]) over all (sample, camera, point)."""
    ncam = masks[0].shape[0]
    dev = masks[0].device
    coords = torch.cat([c.reshape(-1, 2) for c in pixel_coordinates], 0)
    mask = torch.cat([m.reshape(-1) for m in masks], 0)
    slot, row = [], []
    cur = 0
    for b, m in enumerate(masks):
        n = m.shape[1]
        slot.append((torch.arange(ncam, device=dev) + b * ncam).repeat_interleave(n))
        row.append((torch.arange(n, device=dev) + cur).repeat(ncam))
        cur += n
    return coords, mask, torch.cat(slot), torch.cat(row), cur


def _l2c_plan_torch(pixel_coordinates, masks, ch, cw):
    """Both CSR forms of the (camera pixel <- point) pixel-mean map of one grid size.  (The torch formulation: what
    the HIP plan is held against in tests/test_gpu_fusion_plans.py.)"""
    from .torchsparse.nn import functional as spf
    coords, mask, slot, row, n_pts = _flat_entries(pixel_coordinates, masks)
    u = torch.floor((coords[:, 0] + 1.0) / 2 * (cw - 1.0)).long().clamp(0, cw - 1)
    v = torch.floor((coords[:, 1] + 1.0) / 2 * (ch - 1.0)).long().clamp(0, ch - 1)
    pix = (slot * ch + v) * cw + u
    n_dst = len(masks) * masks[0].shape[0] * ch * cw
    key_d = torch.where(mask, pix, torch.full_like(pix, -1)).int()
    num = spf.spcount(key_d, n_dst).clamp(min=1).float()
    w = 1.0 / num[pix]
    order_d, seg_d = spf._csr_by_destination(key_d, n_dst)
    od = order_d.long()
    key_s = torch.where(mask, row, torch.full_like(row, -1)).int()
    order_s, seg_s = spf._csr_by_destination(key_s, n_pts)
    os_ = order_s.long()
    fwd = (row[od].int().contiguous(), w[od].contiguous(), seg_d)
    bwd = (pix[os_].int().contiguous(), w[os_].contiguous(), seg_s)
    return fwd, bwd, n_dst


def _l2c_plan(pixel_coordinates, masks, ch, cw):
    """The same two lists from csrc/fusion.hip: one key launch per sample, u2mkd_csr_build by pixel (and, once per batch,
    by point: the grouping by source does not depend on the grid), one finishing launch."""
    from . import _lib as L
    from .torchsparse.nn import functional as spf
    ncam = masks[0].shape[0]
    dev = masks[0].device
    n_pts = sum(m.shape[1] for m in masks)
    e = ncam * n_pts
    n_dst = len(masks) * ncam * ch * cw
    pix = torch.empty(e, dtype=torch.int32, device=dev)
    key_d = torch.empty_like(pix)
    row = torch.empty_like(pix)

    def keys(key_s):
        row0 = e0 = 0
        for b, (coord, mask) in enumerate(zip(pixel_coordinates, masks)):
            n = mask.shape[1]
            pc, mb = coord.contiguous().float(), _mask_bytes(mask)
            L.call('u2mkd_l2c_keys', L.ptr(pc), L.ptr(mb), ncam, n, b, row0, e0, ch, cw, L.ptr(pix), L.ptr(key_d),
                   L.ptr(key_s), L.ptr(row), L.stream())
            row0 += n
            e0 += ncam * n

    def by_source():
        key_s = torch.empty_like(pix)
        keys(key_s)
        return spf._csr_by_destination(key_s, n_pts) + (True,)
    hit = masks[0].__dict__.get('_u2mkd_plans', {}).get('l2c_by_source')
    order_s, seg_s, fresh = spf._plan(masks[0], 'l2c_by_source', by_source, *masks[1:])
    if hit is not None or not fresh:
        keys(None)
    masks[0].__dict__['_u2mkd_plans']['l2c_by_source'] = (masks[0].__dict__['_u2mkd_plans']['l2c_by_source'][0], (order_s, seg_s, False))
    order_d, seg_d = spf._csr_by_destination(key_d, n_dst)
    fwd_row, bwd_pix = torch.empty_like(pix), torch.empty_like(pix)
    fwd_w = torch.empty(e, dtype=torch.float32, device=dev)
    bwd_w = torch.empty_like(fwd_w)
    L.call('u2mkd_l2c_finish', L.ptr(order_d), L.ptr(seg_d), L.ptr(order_s), L.ptr(pix), L.ptr(row), e, L.ptr(fwd_row),
           L.ptr(fwd_w), L.ptr(bwd_pix), L.ptr(bwd_w), L.stream())
    return (fwd_row, fwd_w, seg_d), (bwd_pix, bwd_w, seg_s), n_dst


# U2MKD_L2C_COMBINE=0: the grids of l2c_scatter up-sampled, added, divided and laid out NCHW by torch operations (rounds 1-5)
_L2C_COMBINE = os.environ.get('U2MKD_L2C_COMBINE', '1') != '0'


class _L2cCombine(torch.autograd.Function):
    """mean over the scales of the grids up-sampled to the first one's size, as a plain NCHW map: ``grids[s]`` channel-last rows
    [n_img * ch_s * cw_s, C] (fp32), ``sizes[s]`` = (ch_s, cw_s), sizes[0] = the map's (csrc/pixhead.hip: l2c_combine_*)."""

    @staticmethod
    def forward(ctx, n_img, sizes, *grids):
        from . import _lib as L
        n = len(grids)
        (h, w), c = sizes[0], grids[0].shape[1]
        ctx.meta = (n_img, sizes, c)
        out = torch.empty(n_img, c, h, w, dtype=torch.float32, device=grids[0].device)
        if n == 1:
            rows = grids[0]
        else:
            rows = torch.empty(n_img * h * w, c, dtype=torch.float32, device=grids[0].device)
            gp = [L.ptr(g) for g in grids] + [None] * (4 - n)
            dims = [v for s in list(sizes[1:]) + [(0, 0)] * (4 - n) for v in s]
            L.call('u2mkd_l2c_combine_forward', *gp, n, n_img, h, w, c, *dims, L.ptr(rows), L.stream())
        L.call('u2mkd_transpose_batched_scaled', L.ptr(rows), L.ptr(out), n_img, h * w, c, 1.0 / n, L.stream())
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from . import _lib as L
        n_img, sizes, c = ctx.meta
        n = len(sizes)
        h, w = sizes[0]
        g = g.contiguous()
        g_rows = torch.empty(n_img * h * w, c, dtype=torch.float32, device=g.device)      # = the full-resolution grid's gradient
        L.call('u2mkd_transpose_batched_scaled', L.ptr(g), L.ptr(g_rows), n_img, c, h * w, 1.0 / n, L.stream())
        grads = [g_rows]
        for ch, cw in sizes[1:]:
            d = torch.empty(n_img * ch * cw, c, dtype=torch.float32, device=g.device)
            L.call('u2mkd_l2c_combine_backward', L.ptr(g_rows), n_img, h, w, c, ch, cw, L.ptr(d), L.stream())
            grads.append(d)
        return (None, None) + tuple(grads)


def l2c_scatter(point_feats, pixel_coordinates, masks, ifh, ifw, n_scales):
    """LiDAR -> camera multi-scale scatter-mean (tsd_full.py:448-478).  point_feats [sum N_b, C];
    returns [B*ncam, C, ifh, ifw]: for scale s in 0..n_scales-1 the (masked) points of a camera are
    averaged per pixel of a (round(ifh/2^s + .01), round(ifw/2^s + .01)) grid, the grid is
    bilinearly up-sampled to (ifh, ifw), and the n_scales maps are averaged."""
    from . import _lib
    _lib.require_cuda(point_feats)                       # HIP device only; there is no CPU fallback
    if point_feats.shape[1] % 4:
        return l2c_scatter_torch(point_feats, pixel_coordinates, masks, ifh, ifw, n_scales)   # torch ops on the GPU
    from .torchsparse.nn import functional as spf
    B, ncam, C = len(masks), masks[0].shape[0], point_feats.shape[1]
    sizes = list(_l2c_grids(ifh, ifw, n_scales))
    if (_L2C_COMBINE and point_feats.dtype == torch.float32 and not spf.bf16_rows() and 1 <= n_scales <= 4 and B * ncam <= 65535
            and sizes[0] == (ifh, ifw)):
        grids = [_SegmentMap.apply(point_feats, *l2c_plan(pixel_coordinates, masks, ch, cw)) for ch, cw in sizes]
        if all(g.dtype == torch.float32 for g in grids):
            return _L2cCombine.apply(B * ncam, tuple(sizes), *grids)
    total = None
    for ch, cw in sizes:
        fwd, bwd, n_dst = l2c_plan(pixel_coordinates, masks, ch, cw)
        grid = _SegmentMap.apply(point_feats, fwd, bwd, n_dst).view(B * ncam, ch, cw, C).permute(0, 3, 1, 2)
        up = grid if (ch, cw) == (ifh, ifw) else F.interpolate(grid, (ifh, ifw), mode='bilinear', align_corners=True)
        total = up if total is None else total + up
    # the grids are channel-last views; hand the camera branch (NCHW convs) a plain NCHW tensor so
    # its skip additions do not mix memory formats (strided adds were 5 ms of the KD step)
    return (total / n_scales).contiguous()


def l2c_scatter_torch(point_feats, pixel_coordinates, masks, ifh, ifw, n_scales):
    """The same map written with torch ops only (index_add_ per scale): the formulation the host-side
    tests pin against the reference's Python loop, and the path for channel counts that are not
    multiples of 4.  Not used by l2c_scatter for anything the HIP kernels cover."""
    B = len(masks)
    ncam = masks[0].shape[0]
    C = point_feats.shape[1]
    dev, dt = point_feats.device, point_feats.dtype
    sizes = [m.shape[1] for m in masks]
    # flatten (sample, camera, point): camera slot id and the point's row in point_feats
    coords = torch.cat([c.reshape(-1, 2) for c in pixel_coordinates], 0)              # [sum ncam*N_b, 2]
    mask = torch.cat([m.reshape(-1) for m in masks], 0)
    slot, row = [], []
    cur = 0
    for b, n in enumerate(sizes):
        slot.append((torch.arange(ncam, device=dev) + b * ncam).repeat_interleave(n))
        row.append((torch.arange(n, device=dev) + cur).repeat(ncam))
        cur += n
    slot, row = torch.cat(slot), torch.cat(row)
    w = mask.to(dt).unsqueeze(1)
    feats = point_feats[row] * w
    total = torch.zeros(B * ncam, C, ifh, ifw, device=dev, dtype=dt)
    cnt = 1
    for _ in range(n_scales):
        ch = int(round(float(ifh) / cnt + 0.01))
        cw = int(round(float(ifw) / cnt + 0.01))
        u = torch.floor((coords[:, 0] + 1.0) / 2 * (cw - 1.0)).long().clamp(0, cw - 1)
        v = torch.floor((coords[:, 1] + 1.0) / 2 * (ch - 1.0)).long().clamp(0, ch - 1)
        pix = (slot * ch + v) * cw + u
        acc = torch.zeros(B * ncam * ch * cw, C, device=dev, dtype=dt).index_add_(0, pix, feats)
        num = torch.zeros(B * ncam * ch * cw, 1, device=dev, dtype=dt).index_add_(0, pix, w)
        grid = (acc / num.clamp(min=1.0)).view(B * ncam, ch, cw, C).permute(0, 3, 1, 2)
        total = total + F.interpolate(grid, (ifh, ifw), mode='bilinear', align_corners=True)
        cnt *= 2
    return total / n_scales
