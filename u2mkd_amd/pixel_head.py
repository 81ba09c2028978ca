"""The pixel head of the camera branch evaluated at the pixels the LiDAR points read (csrc/pixhead.hip).

Reference (core/models/nuscenes/spvcnn_swiftnet18_spformer_tsd_full.py, the `run_pix_decoder` branch of the student's
forward): ``x_pix = Feature_Fetch(masks, pixel_coordinates, classifier_pix(pix_branch.forward_up(feats, im_size)))`` --
the decoder's [b*ncam, 128, H/2, W/2] map is up-sampled to the image size (swiftnet.py forward_up), BatchNorm + ReLU +
a 1x1 convolution run over every one of the b*ncam*H*W pixels, and Feature_Fetch (fusion_blocks.py:257-278) then reads
the logits at the <= 4 bilinear corners of every LiDAR point.  Here the same numbers come from the low-resolution map:

* BatchNorm's batch statistics over the up-sampled map are two quadratures on the low-resolution map
  (`u2mkd_upbn_stats`; running statistics are updated with them exactly as nn.BatchNorm2d would);
* every (point, corner) sample gathers its up-sampled 128-vector from 4 low-resolution pixels (`u2mkd_up_plan` composes
  the two bilinear maps, the gather is the row operator `spdevoxelize` with its deterministic CSR backward), is
  normalised, rectified and multiplied by the classifier (the row operator `linear`), and the 4 corners of a point are
  blended with Feature_Fetch's weights;
* backward: the sampled rows carry their own gradient back through the gather; the BatchNorm backward's terms that
  reach EVERY up-sampled pixel (dU = c0 + c1 U) fold through the up-sampling in closed form (`u2mkd_upbn_dense_grad`).

Same result as the dense evaluation up to fp32 summation order (tests/test_gpu_pixel_head.py); 6 x 128 x 360 x 640
floats (708 MB, 4.4 GB at 900 x 1600) per tensor of the dense tail are never formed."""
import math
import os

import numpy as np
import torch

from . import _lib as L

__all__ = ['sampled_pixel_logits', 'sampled_head_applies']

_ENABLED = os.environ.get('U2MKD_SAMPLED_PIXEL_HEAD', '1') != '0'
_ROW_BN = os.environ.get('U2MKD_PIXEL_HEAD_ROW_BN', '1') != '0'      # 0: the samples' normalise + ReLU as torch element-wise kernels (A/B, tests)
_COEF = {}
_ROWS_PER_CHUNK = 16


def _rows_per_chunk(w):
    """rows of one plane a workgroup stages in LDS (with two halo rows) for the stencil kernels: <= 60 KB"""
    return max(1, min(_ROWS_PER_CHUNK, (60 * 1024 // 4) // max(w, 1) - 2))


def _interp_coefficients(n_in, n_out, device):
    """(scale, a [n_in], diagonals [3, n_in]) of W = the [n_out, n_in] matrix of F.interpolate(mode='bilinear',
    align_corners=True) along one axis, with torch's fp32 index arithmetic (src = scale * dst, lambda = src - floor):
    a = W^T 1 (column sums), diagonals = lower / main / upper diagonal of W^T W (tridiagonal: a row of W has two
    neighbouring entries).  Built once per size in fp64 on the host."""
    key = (n_in, n_out, str(device))
    if key not in _COEF:
        r = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
        src = (r * np.arange(n_out, dtype=np.float32)).astype(np.float32)
        i0 = src.astype(np.int64)
        ip = (i0 < n_in - 1).astype(np.int64)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        l0 = (np.float32(1) - l1).astype(np.float32)
        wm = np.zeros((n_out, n_in), np.float64)
        np.add.at(wm, (np.arange(n_out), i0), l0.astype(np.float64))
        np.add.at(wm, (np.arange(n_out), i0 + ip), l1.astype(np.float64))
        gram = wm.T @ wm
        diag = np.zeros((3, n_in))
        diag[1] = np.diag(gram)
        diag[0, 1:] = np.diag(gram, -1)
        diag[2, :-1] = np.diag(gram, 1)
        _COEF[key] = (float(r), torch.tensor(wm.sum(0), dtype=torch.float32, device=device),
                      torch.tensor(diag, dtype=torch.float32, device=device).contiguous())
    return _COEF[key]


class _UpsampledBatchNormReLU(torch.autograd.Function):
    """relu(batch_norm(up(x))) at sampled pixels: x [N, C, h, w] the low-resolution map, u [S, C] the up-sampled values
    at the samples (gathered from x by the caller, so their gradient flows through the gather)."""

    @staticmethod
    def forward(ctx, x, u, weight, bias, bn, size):
        from .torchsparse.nn.functional import _sync_group
        n, c, h, w = x.shape
        big_h, big_w = size
        count = n * big_h * big_w
        training = bn.training or bn.running_mean is None
        sync = _sync_group(bn) if training else None
        coef = None
        if training:
            _, a, ay = _interp_coefficients(h, big_h, x.device)
            _, b, ax = _interp_coefficients(w, big_w, x.device)
            coef = (a, b, ay, ax)
            rpc = _rows_per_chunk(w)
            chunks = (h + rpc - 1) // rpc
            partial = torch.empty(n * c, chunks, 2, dtype=torch.float32, device=x.device)
            L.call('u2mkd_upbn_stats', L.ptr(x), n, c, h, w, L.ptr(a), L.ptr(b), L.ptr(ay), L.ptr(ax), rpc,
                   L.ptr(partial), L.stream())
            sums = partial.view(n, c, chunks, 2).double().sum((0, 2))               # [C, 2], shifted by x[0, :, 0, 0]
            m1 = sums[:, 0] / count
            mean = x[0, :, 0, 0].double() + m1
            var = (sums[:, 1] / count - m1 * m1).clamp_(min=0.0)
            if sync is not None:
                # SyncBatchNorm: every rank's (mean, M2, count) in ONE all_gather of [2C+1] floats, merged in rank order
                # (Chan) -- the statistics torch.nn.SyncBatchNorm takes over the ranks' up-sampled maps
                from .torchsparse.nn.functional import _gather_rows, note_collective
                group, world = sync
                row = torch.cat([mean, var * count, torch.full((1,), float(count), dtype=torch.float64, device=x.device)]).float()
                rows = torch.empty(world, 2 * c + 1, dtype=torch.float32, device=x.device)
                note_collective('all_gather', row)
                if world > 1:
                    _gather_rows(rows, row, group)
                else:
                    rows[0] = row
                rows = rows.double()
                cnt = rows[:, 2 * c]
                count = cnt.sum()                               # (a device scalar: no host round trip)
                mean = (rows[:, :c] * cnt[:, None]).sum(0) / count
                var = ((rows[:, c:2 * c] + cnt[:, None] * (rows[:, :c] - mean) ** 2).sum(0) / count).clamp_(min=0.0)
            if bn.track_running_stats and bn.running_mean is not None:
                with torch.no_grad():
                    bn.num_batches_tracked.add_(1)
                    m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                    bn.running_mean.mul_(1.0 - m).add_(mean.to(bn.running_mean.dtype), alpha=m)
                    unbias = count / (count - 1).clamp(min=1) if torch.is_tensor(count) else count / max(count - 1, 1)
                    bn.running_var.mul_(1.0 - m).add_((var * unbias).to(bn.running_var.dtype), alpha=m)
        else:
            mean, var = bn.running_mean.double(), bn.running_var.double()
        invstd = (var + bn.eps).rsqrt().float().contiguous()
        mean = mean.float().contiguous()
        weight, bias, u = weight.contiguous(), bias.contiguous(), u.contiguous()
        # relu((u - mean) invstd weight + bias) over the [S, C] samples: the row BatchNorm's apply pass with THESE statistics
        # (one pass; as five element-wise kernels it moved 1.5 GB per step at 6 x 360x640)
        if _ROW_BN:
            y = torch.empty_like(u)
            L.call('u2mkd_bn_apply', L.ptr(u), u.shape[0], c, L.ptr(mean), L.ptr(invstd), L.ptr(weight), L.ptr(bias), 1, L.ptr(y),
                   L.stream())
        else:
            y = torch.relu((u - mean) * invstd * weight + bias)
        ctx.save_for_backward(x, u, weight, bias, invstd, mean)
        ctx.meta = (training, count, coef, sync)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, u, weight, bias, invstd, mean = ctx.saved_tensors
        training, count, coef, sync = ctx.meta
        # g = gy (y > 0); dbeta = sum g; dgamma = sum g xhat; du = g scale: the row BatchNorm's backward in its eval form (the
        # statistics' own gradient reaches every pixel of the map, below), mask and xhat re-derived from u
        gy = gy.contiguous()
        s_rows = u.shape[0]
        cc = u.shape[1]
        partial = torch.empty(max(int(L.load().u2mkd_bn_num_slabs(s_rows)), 1) * 2 * cc, dtype=torch.float32, device=u.device)
        dgamma = torch.empty(cc, dtype=torch.float32, device=u.device)
        dbeta = torch.empty(cc, dtype=torch.float32, device=u.device)
        du = torch.empty_like(u)
        if not _ROW_BN:
            xhat = (u - mean) * invstd
            g = gy * (xhat * weight + bias > 0)
            dbeta, dgamma, du = g.sum(0), (g * xhat).sum(0), g * (weight * invstd)
        elif s_rows:
            L.call('u2mkd_bn_backward', L.ptr(gy), L.ptr(u), s_rows, cc, L.ptr(mean), L.ptr(invstd), L.ptr(weight), L.ptr(bias), 1, 0,
                   L.ptr(partial), L.ptr(dgamma), L.ptr(dbeta), L.ptr(du), L.stream())
        else:
            dgamma.zero_(); dbeta.zero_()
        scale = weight * invstd
        dx = None
        if training:
            # dU = scale * (g - mean(g) - xhat * mean(g * xhat)) over ALL count pixels; g = 0 off the samples.  The two
            # mean terms reach every pixel: c0 + c1 * U, folded through the up-sampling by the dense-gradient kernel.
            sb, sg = dbeta, dgamma
            if sync is not None:
                from .torchsparse.nn.functional import note_collective
                note_collective('all_reduce', torch.empty(0, dtype=dbeta.dtype).new_empty(2 * dbeta.numel()))
            if sync is not None and sync[1] > 1:          # the two sums over every rank's pixels: ONE all_reduce of [2C]
                from .torchsparse.nn.functional import _sum_over_ranks
                both = torch.cat([dbeta, dgamma])
                _sum_over_ranks(both, sync[0])
                sb, sg = both[:dbeta.numel()], both[dbeta.numel():]
            c1 = -(scale * invstd) * (sg / count)
            c0 = -scale * (sb / count) - c1 * mean
            a, b, ay, ax = coef
            n, c, h, w = x.shape
            dx = torch.empty_like(x)
            L.call('u2mkd_upbn_dense_grad', L.ptr(x), n, c, h, w, L.ptr(a), L.ptr(b), L.ptr(ay), L.ptr(ax),
                   L.ptr(c0.contiguous()), L.ptr(c1.contiguous()), _rows_per_chunk(w), L.ptr(dx), L.stream())
        return dx, du, dgamma, dbeta, None, None


def sampled_head_applies(x, head):
    """A device map and a BatchNorm2d -> ReLU -> bias-free 1x1 convolution head (this package's BatchNorm2d, or the
    SyncBatchNorm2d DistributedDataParallel training converts it to: its statistics then take one small collective per
    pass, like every other synchronised BatchNorm).  Under autocast the tail is evaluated in fp32 from the reduced map
    (nn.BatchNorm is an fp32 operator under autocast anyway)."""
    from .camera import BatchNorm2d, SyncBatchNorm2d
    return (_ENABLED and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16, torch.float16)
            and x.shape[1] % 4 == 0 and x.shape[1] <= 1024      # (the row BatchNorm kernels' channel counts)
            and type(head.norm) in (BatchNorm2d, SyncBatchNorm2d) and head.norm.momentum is not None
            and head.conv.bias is None and head.conv.kernel_size == (1, 1))


def sampled_pixel_logits(x, head, pixel_coordinates, masks, im_size, ib, ncam):
    """``feature_fetch(masks, pixel_coordinates, head(upsample(x, im_size)).view(ib, ncam, ...))`` without the
    full-resolution tensors.  x [ib*ncam, C, h, w]; head = camera.BNReluConv(C, classes, k=1); returns
    [sum N_b, classes]."""
    from .fusion import _NchwToRows, _c2l_plan
    from .torchsparse.nn import functional as spf
    L.require_cuda(x)
    with torch.autocast('cuda', enabled=False):
        return _sampled_pixel_logits(x.float(), head, pixel_coordinates, masks, im_size, ib, ncam)


def _sampled_pixel_logits(x, head, pixel_coordinates, masks, im_size, ib, ncam):
    from .fusion import _NchwToRows, _c2l_plan
    from .torchsparse.nn import functional as spf
    big_h, big_w = im_size
    n, c, h, w = x.shape
    assert n == ib * ncam, (n, ib, ncam)
    x = x.contiguous()
    # Feature_Fetch's corners in the full-resolution map (the same plan its dense form uses) ...
    idx8f, w8f = spf._plan(masks[0], 'c2l_%d_%d' % (big_h, big_w),
                           lambda: _c2l_plan(pixel_coordinates, masks, big_h, big_w), *masks[1:], *pixel_coordinates)

    def compose():
        npts = idx8f.shape[0]
        ia = torch.empty(npts * 4, 8, dtype=torch.int32, device=x.device)
        wa = torch.empty(npts * 4, 8, dtype=torch.float32, device=x.device)
        rh, rw = _interp_coefficients(h, big_h, x.device)[0], _interp_coefficients(w, big_w, x.device)[0]
        L.call('u2mkd_up_plan', L.ptr(idx8f), npts, big_h, big_w, h, w, rh, rw, L.ptr(ia), L.ptr(wa), L.stream())
        return ia, wa
    # ... composed with the up-sampling: 4 low-resolution sources per (point, corner) sample
    ia, wa = spf._plan(masks[0], 'up_%d_%d_%d_%d' % (big_h, big_w, h, w), compose, *masks[1:], *pixel_coordinates)
    rows = _NchwToRows.apply(x.view(ib, ncam, c, h, w))
    u = spf.spdevoxelize(rows, ia, wa)                                              # [4 * points, C]
    y = _UpsampledBatchNormReLU.apply(x, u, head.norm.weight.float(), head.norm.bias.float(), head.norm, (big_h, big_w))
    z = spf.linear(y, head.conv.weight.float().view(head.conv.out_channels, c), None)      # [4 * points, classes]
    return (z.view(-1, 4, z.shape[1]) * w8f[:, :4].unsqueeze(-1)).sum(1)
