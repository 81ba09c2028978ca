"""SwiftNet-18 camera branch (row a13; core/models/image_branch/swiftnet.py:20-50, 114-341).

The dense 2D convolutions ride PyTorch-ROCm / MIOpen as the north star prescribes; the BatchNorm2d
layers (and the ReLU / residual add that follow them) run on the HIP kernels of csrc/bn2d.hip when the
map is an fp32 NCHW tensor on the device, and on torch.nn otherwise (CPU, autocast, SyncBatchNorm).
ResNet-18 encoder whose 7x7 stem has stride 1 (feature
maps are H/2 after the stem), pre-activation lateral skips, a spatial pyramid pooling
bottleneck (grids 8/4/2/1 scaled by the aspect ratio, BN momentum 0.012) and a 3-level
up-sampling decoder.  Module / parameter names equal the reference's, so its ImageNet and
U2MKD checkpoints load unchanged."""
import os

import torch
import torch.nn.functional as F
from torch import nn

__all__ = ['SwiftNetRes18', 'SwiftNetResNet', 'BNReluConv', 'BatchNorm2d', 'SyncBatchNorm2d', 'bn_act']

_HIP_BN2D = os.environ.get('U2MKD_BN2D', '1') != '0'


class _MaxPool3s2Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        from . import _lib as L
        n, c, h, w = x.shape
        oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty(n, c, oh, ow, dtype=torch.float32, device=x.device)
        code = torch.empty(n, c, oh, ow, dtype=torch.uint8, device=x.device)
        L.call('u2mkd_maxpool3s2_forward', L.ptr(x), n * c, h, w, L.ptr(y), L.ptr(code), L.stream())
        ctx.save_for_backward(code)
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from . import _lib as L
        code, = ctx.saved_tensors
        n, c, h, w = ctx.shape
        dx = torch.empty(n, c, h, w, dtype=torch.float32, device=g.device)
        L.call('u2mkd_maxpool3s2_backward', L.ptr(g.contiguous()), L.ptr(code), n * c, h, w, L.ptr(dx), L.stream())
        return dx


class MaxPool3x3s2(nn.MaxPool2d):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (the stem's pooling, swiftnet.py) on csrc/pixhead.hip's two
    kernels for fp32 device maps: a byte per output instead of torch's int64 index, a gathering backward (0.7 ms -> 0.15 ms
    on the 6 x 64 x 360 x 640 map, at the very end of the step's camera chain); the same element wins a tie."""

    def __init__(self):
        super().__init__(kernel_size=3, stride=2, padding=1)

    def forward(self, x):
        # (the kernels take planes = n * c <= 65535 and h * w < 2^31: include/u2mkd_hip.h; larger maps stay on torch's)
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and x.numel() and not torch.is_autocast_enabled()
                and x.shape[0] * x.shape[1] <= 65535 and x.shape[2] * x.shape[3] < 2 ** 31):
            return _MaxPool3s2Function.apply(x)
        return super().forward(x)


_UP_TAPS = {}
# U2MKD_UP_BILINEAR: 1 (default since round 6) = the decoder's up-samplings on csrc/pixhead.hip (gathering, run-to-run
# reproducible backward), 0 = torch's kernels (backward scatters with float atomics), auto = 1 exactly when
# torch.use_deterministic_algorithms(True) is set -- torch's backward refuses to run then.  The HIP kernels take 1.1 ms less
# kernel time per KD step (forward 135-450 us -> 14-180 us per call, backward 215 -> 131 us).  Round 4 measured the step 0.3-0.4 ms
# SLOWER with them (74.4-74.7 vs 74.1-74.4 ms) and kept torch's; in round 6 the step is bound by the GPU in both phases and the
# same switch reads 62.76 / 62.91 / 63.04 against 63.15 / 63.08 / 63.02 ms (three same-box pairs): small, in the right direction,
# and the backward is reproducible -- so they are the default now.
_UP_MODE = os.environ.get('U2MKD_UP_BILINEAR', '1')
_UP_HIP = None if _UP_MODE == 'auto' else _UP_MODE != '0'


def _up_on_hip():
    return torch.are_deterministic_algorithms_enabled() if _UP_HIP is None else _UP_HIP


def _up_taps(n_in, n_out, device):
    """Per input index of one axis: (first output index, number of outputs) that read it and their weights [n_in, 8], from
    torch's index arithmetic of F.interpolate(bilinear, align_corners=True) (src = scale * dst in fp32).  None when an
    input feeds more than 8 outputs (up-sampling factors beyond ~4: torch's kernel serves those)."""
    key = (n_in, n_out, str(device))
    if key not in _UP_TAPS:
        import numpy as np
        r = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
        src = (r * np.arange(n_out, dtype=np.float32)).astype(np.float32)
        i0 = src.astype(np.int64)
        ip = (i0 < n_in - 1).astype(np.int64)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        l0 = (np.float32(1) - l1).astype(np.float32)
        first = np.full(n_in, n_out, np.int64)
        last = np.full(n_in, -1, np.int64)
        for idx in (i0, i0 + ip):
            np.minimum.at(first, idx, np.arange(n_out))
            np.maximum.at(last, idx, np.arange(n_out))
        cnt = np.maximum(last - first + 1, 0)
        ok = int(cnt.max()) <= 8
        taps = wts = span = None
        if ok:
            w = np.zeros((n_in, 8), np.float32)
            for o in range(n_out):             # (sizes of a few hundred: once per size)
                w[i0[o], o - first[i0[o]]] += l0[o]
                w[i0[o] + ip[o], o - first[i0[o] + ip[o]]] += l1[o]
            first = np.where(cnt > 0, first, 0)
            # the widest window of outputs a tile of `tile` consecutive inputs reads (csrc/pixhead.hip's LDS form)
            span = {t: 1 << 30 if int(cnt.min()) == 0 else max(int(first[min(a + t, n_in) - 1] + cnt[min(a + t, n_in) - 1] - first[a]) for a in range(0, n_in, t))
                    for t in (16, 64)}
            taps = torch.tensor(np.stack([first, cnt], 1), dtype=torch.int32, device=device).contiguous()
            wts = torch.tensor(w, dtype=torch.float32, device=device).contiguous()
        _UP_TAPS[key] = (float(r), taps, wts, span)
    return _UP_TAPS[key]


class _UpBilinearFunction(torch.autograd.Function):
    """y = F.interpolate(x, size, bilinear, align_corners=True) [+ skip] on csrc/pixhead.hip (deterministic backward)."""

    @staticmethod
    def forward(ctx, x, skip, size):
        from . import _lib as L
        n, c, h, w = x.shape
        big_h, big_w = size
        rh, ty, wy, span_y = _up_taps(h, big_h, x.device)
        rw, tx, wx, span_x = _up_taps(w, big_w, x.device)
        y = torch.empty(n, c, big_h, big_w, dtype=torch.float32, device=x.device)
        L.call('u2mkd_up_bilinear_forward', L.ptr(x), L.ptr(skip), n * c, h, w, big_h, big_w, rh, rw, L.ptr(y), L.stream())
        tiled = int(span_y[16] <= 40 and span_x[64] <= 136 and n * c <= 65535)
        ctx.geom = (n, c, h, w, big_h, big_w, ty, wy, tx, wx, tiled)
        ctx.has_skip = skip is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from . import _lib as L
        n, c, h, w, big_h, big_w, ty, wy, tx, wx, tiled = ctx.geom
        g = g.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(n, c, h, w, dtype=torch.float32, device=g.device)
            L.call('u2mkd_up_bilinear_backward', L.ptr(g), n * c, h, w, big_h, big_w, L.ptr(ty), L.ptr(wy), L.ptr(tx), L.ptr(wx),
                   tiled, L.ptr(dx), L.stream())
        return dx, (g if ctx.has_skip and ctx.needs_input_grad[1] else None), None


def _up(x, size, skip=None):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) [+ skip]."""
    size = tuple(int(v) for v in size)
    if (_up_on_hip() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and x.numel() and not torch.is_autocast_enabled()
            and x.shape[0] * x.shape[1] <= 65535 and size[0] >= x.shape[2] and size[1] >= x.shape[3] and (skip is None or (skip.dtype == x.dtype and skip.is_contiguous()
                                                                                      and skip.shape[2:] == size))):
        if _up_taps(x.shape[2], size[0], x.device)[1] is not None and _up_taps(x.shape[3], size[1], x.device)[1] is not None:
            return _UpBilinearFunction.apply(x, skip, size)
    y = F.interpolate(x, size, mode='bilinear', align_corners=True)
    return y if skip is None else y + skip


class _BatchNorm2dFunction(torch.autograd.Function):
    """relu?(batch_norm(x) [+ res]) on csrc/bn2d.hip (statistics, normalisation, residual add and ReLU in three
    passes over the map; the backward recomputes the ReLU mask from x)."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, bn, relu):
        from . import _lib as L
        b, c, h, w = x.shape
        hw = h * w
        y = torch.empty_like(x)
        st = L.stream()
        batch_stats = bn.training or bn.running_mean is None
        if batch_stats:
            ws = torch.empty(max(L.load().u2mkd_bn2d_workspace_bytes(b, c, hw), 16), dtype=torch.uint8, device=x.device)
            mean = torch.empty(c, dtype=torch.float32, device=x.device)
            invstd = torch.empty_like(mean)
            track = bn.training and bn.running_mean is not None
            L.call('u2mkd_bn2d_train_forward', L.ptr(x), L.ptr(res), b, c, hw, L.ptr(weight), L.ptr(bias), bn.eps,
                   bn.momentum if track else 0.0, int(relu), L.ptr(bn.running_mean if track else None),
                   L.ptr(bn.running_var if track else None), L.ptr(bn.num_batches_tracked if track else None),
                   L.ptr(ws), L.ptr(mean), L.ptr(invstd), L.ptr(y), st)
        else:
            mean, invstd = bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)
            L.call('u2mkd_bn2d_eval_forward', L.ptr(x), L.ptr(res), b, c, hw, L.ptr(weight), L.ptr(bias), bn.eps, int(relu),
                   L.ptr(bn.running_mean), L.ptr(bn.running_var), L.ptr(y), st)
        ctx.save_for_backward(x, weight, bias, res, mean, invstd)
        ctx.relu, ctx.batch_stats = bool(relu), batch_stats
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib as L
        x, weight, bias, res, mean, invstd = ctx.saved_tensors
        b, c, h, w = x.shape
        hw = h * w
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if res is not None else None
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty_like(dgamma)
        ws = torch.empty(max(L.load().u2mkd_bn2d_workspace_bytes(b, c, hw), 16), dtype=torch.uint8, device=x.device)
        L.call('u2mkd_bn2d_backward', L.ptr(dy), L.ptr(x), L.ptr(res), b, c, hw, L.ptr(mean), L.ptr(invstd), L.ptr(weight),
               L.ptr(bias), int(ctx.relu), int(ctx.batch_stats), L.ptr(ws), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dx), L.ptr(dres),
               L.stream())
        return dx, (dgamma if weight is not None else None), (dbeta if bias is not None else None), dres, None, None


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters, buffers and state-dict keys) whose forward can take the ReLU and the residual
    add that follow it: ``relu?(bn(x) [+ residual])`` -- one fused HIP pass on the device, torch.nn elsewhere."""

    def forward(self, x, relu=False, residual=None):
        if (_HIP_BN2D and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and x.numel() > 0
                and not torch.is_autocast_enabled() and (self.momentum is not None or not self.training)
                and (residual is None or (residual.dtype == x.dtype and residual.shape == x.shape))):
            if residual is not None:
                residual = residual.contiguous()
            return _BatchNorm2dFunction.apply(x, self.weight, self.bias, residual, self, relu)
        y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y


class _SyncBatchNorm2dFunction(torch.autograd.Function):
    """relu?(sync_batch_norm(x) [+ res]) over the ranks of a process group on the csrc/bn2d.hip pieces: local
    (mean, M2, count) -> ONE all_gather of [2C+1] floats -> Chan merge in rank order -> normalise (+ residual, ReLU);
    backward: local sums -> ONE all_reduce of [2C] floats -> dx with the global count.  The same statistics as
    torch.nn.SyncBatchNorm (what the reference's convert_sync_batchnorm(model.model_s) makes of every BatchNorm2d of the
    camera branch, train_lc_nusc_tsd_full.py:80), with the ReLU / residual add still fused under DDP."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, bn, relu, group, world):
        from . import _lib as L
        from .torchsparse.nn.functional import _gather_rows, note_collective
        b, c, h, w = x.shape
        hw = h * w
        dev, st = x.device, L.stream()
        ws = torch.empty(max(L.load().u2mkd_bn2d_workspace_bytes(b, c, hw), 16), dtype=torch.uint8, device=dev)
        stats = torch.empty(2 * c + 1, dtype=torch.float32, device=dev)
        L.call('u2mkd_bn2d_local_stats', L.ptr(x), b, c, hw, L.ptr(ws), L.ptr(stats), st)
        note_collective('all_gather', stats)
        if world > 1:
            gathered = torch.empty(world, 2 * c + 1, dtype=torch.float32, device=dev)
            _gather_rows(gathered, stats, group)
        else:
            gathered = stats.view(1, -1)          # (a one-rank group: the row is its own gathering)
        mit = torch.empty(2 * c + 1, dtype=torch.float32, device=dev)
        mean, invstd, total = mit[:c], mit[c:2 * c], mit[2 * c:]
        track = bn.running_mean is not None
        # (the step counter is bumped inside the merge launch)
        L.call('u2mkd_bn_merge_stats_counted', L.ptr(gathered), world, c, float(bn.eps), float(bn.momentum if track else 0.0),
               L.ptr(bn.running_mean if track else None), L.ptr(bn.running_var if track else None), L.ptr(mean),
               L.ptr(invstd), L.ptr(total), L.ptr(bn.num_batches_tracked if track else None), L.stream())
        y = torch.empty_like(x)
        L.call('u2mkd_bn2d_apply', L.ptr(x), L.ptr(res), b, c, hw, L.ptr(mean), L.ptr(invstd), L.ptr(weight), L.ptr(bias),
               int(relu), L.ptr(y), L.stream())
        ctx.save_for_backward(x, weight, bias, res, mean, invstd, total)
        ctx.relu, ctx.group, ctx.world = bool(relu), group, world
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib as L
        from .torchsparse.nn.functional import _sum_over_ranks, note_collective
        x, weight, bias, res, mean, invstd, total = ctx.saved_tensors
        b, c, h, w = x.shape
        hw = h * w
        dy = dy.contiguous()
        dev = x.device
        ws = torch.empty(max(L.load().u2mkd_bn2d_workspace_bytes(b, c, hw), 16), dtype=torch.uint8, device=dev)
        both = torch.empty(2, 2 * c, dtype=torch.float32, device=dev)
        sums, local = both[0], both[1]            # `sums` into the all_reduce in place; `local` = this rank's parameter gradients (DDP averages them)
        L.call('u2mkd_bn2d_backward_local_keep', L.ptr(dy), L.ptr(x), L.ptr(res), b, c, hw, L.ptr(mean), L.ptr(invstd),
               L.ptr(weight), L.ptr(bias), int(ctx.relu), L.ptr(ws), L.ptr(sums), L.ptr(local), L.stream())
        note_collective('all_reduce', sums)
        if ctx.world > 1:
            _sum_over_ranks(sums, ctx.group)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if res is not None else None
        L.call('u2mkd_bn2d_backward_apply', L.ptr(dy), L.ptr(x), L.ptr(res), b, c, hw, L.ptr(total), L.ptr(mean),
               L.ptr(invstd), L.ptr(weight), L.ptr(bias), int(ctx.relu), L.ptr(sums), L.ptr(dx), L.ptr(dres), L.stream())
        return (dx, local[c:] if weight is not None else None, local[:c] if bias is not None else None, dres,
                None, None, None, None)


class SyncBatchNorm2d(nn.SyncBatchNorm):
    """What ``SparseSyncBatchNorm.convert_sync_batchnorm`` makes of this file's BatchNorm2d: torch.nn.SyncBatchNorm
    (same parameters, buffers, keys) whose forward takes the ReLU / residual add that follow it and runs on the HIP
    pieces with one small collective per pass; anything they do not cover (CPU, autocast, evaluation) takes torch's."""

    def forward(self, x, relu=False, residual=None):
        from .torchsparse.nn.functional import _sync_group
        sync = _sync_group(self) if self.training else None
        if (sync is not None and _HIP_BN2D and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
                and x.numel() > 0 and not torch.is_autocast_enabled() and self.momentum is not None
                and (residual is None or (residual.dtype == x.dtype and residual.shape == x.shape))):
            if residual is not None:
                residual = residual.contiguous()
            return _SyncBatchNorm2dFunction.apply(x, self.weight, self.bias, residual, self, relu, sync[0], sync[1])
        if (sync is None and _HIP_BN2D and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
                and x.numel() > 0 and not torch.is_autocast_enabled() and (self.momentum is not None or not self.training)
                and (residual is None or (residual.dtype == x.dtype and residual.shape == x.shape))):
            # nothing to synchronise with (one rank, or evaluation): the fused single-process pass of BatchNorm2d above
            # (torch.nn.SyncBatchNorm makes the same decision and calls F.batch_norm)
            if residual is not None:
                residual = residual.contiguous()
            return _BatchNorm2dFunction.apply(x, self.weight, self.bias, residual, self, relu)
        y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y


def bn_act(bn, x, relu=False, residual=None):
    """relu?(bn(x) [+ residual]) for any BatchNorm flavour (this file's fused one, or what convert_sync_batchnorm
    made of it)."""
    if isinstance(bn, (BatchNorm2d, SyncBatchNorm2d)):
        return bn(x, relu, residual)
    y = bn(x)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


class _Conv2dFunction(torch.autograd.Function):
    """``F.conv2d`` (MIOpen) whose weight gradient leaves the chain of the backward.  Every fusion stage's LiDAR gradient waits
    for the camera layer's INPUT gradient (camera -> LiDAR gather <- ResNet layer <- LiDAR -> camera scatter); the weight
    gradients of the layer's convolutions (igemm + layout transposes, ~0.25 ms each) are needed by nobody before the
    optimizer: they go to a side stream (deferred.side_for) and are joined when the backward ends."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding):
        ctx.dtypes = (x.dtype, weight.dtype)
        ctx.pid = id(weight)
        if torch.is_autocast_enabled('cuda'):
            dt = torch.get_autocast_dtype('cuda')
            x, weight = x.to(dt), weight.to(dt)
        ctx.save_for_backward(x, weight)
        ctx.conf = (stride, padding)
        with torch.autocast('cuda', enabled=False):
            return F.conv2d(x, weight, None, stride, padding)

    @staticmethod
    def backward(ctx, g):
        from . import deferred
        x, weight = ctx.saved_tensors
        stride, padding = ctx.conf
        g = g.contiguous().to(x.dtype)

        def bwd(mask):
            return torch.ops.aten.convolution_backward(g, x, weight, None, stride, padding, (1, 1), False, (0, 0), 1, mask)
        gx = bwd((True, False, False))[0].to(ctx.dtypes[0]) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            if deferred.owned(ctx.pid):
                # a second use of the weight in this pass: autograd adds the two contributions when this function returns, on
                # this stream -- the first must be complete here and this one is computed in line
                deferred.join()
                gw = bwd((False, True, False))[1].to(ctx.dtypes[1])
            else:
                side = deferred.side_for('camera_wgrad', g.device, owner=ctx.pid)
                with torch.cuda.stream(side):
                    gw = bwd((False, True, False))[1].to(ctx.dtypes[1])
                for t in (x, weight, g):
                    t.record_stream(side)
                gw.record_stream(torch.cuda.current_stream(g.device))      # allocated on the side stream, consumed on this one
        return gx, gw, None, None


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters, same state-dict keys) on _Conv2dFunction while training on the GPU."""

    def forward(self, x):
        from . import deferred
        w = self.weight
        if (deferred.enabled() and deferred.overlap_ok() and x.is_cuda and torch.is_grad_enabled() and w.requires_grad and w.is_leaf and w.grad is None and self.bias is None
                and self.groups == 1 and self.dilation == (1, 1) and self.padding_mode == 'zeros'
                and not isinstance(self.padding, str) and not torch.cuda.is_current_stream_capturing()):
            return _Conv2dFunction.apply(x, w, self.stride, self.padding)
        return super().forward(x)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        # relu(bn2(conv2(relu(bn1(conv1(x))))) + identity): the reference's ReLU is in place, so the "pre-activation
        # skip" it returns next to the activated map IS the activated map (swiftnet.py BasicBlock.forward)
        out = bn_act(self.bn1, self.conv1(x), relu=True)
        out = bn_act(self.bn2, self.conv2(out), relu=True, residual=x if self.downsample is None else self.downsample(x))
        return out, out


class BNReluConv(nn.Sequential):
    def __init__(self, cin, cout, k=3, bn_momentum=0.1):
        super().__init__()
        self.add_module('norm', BatchNorm2d(cin, momentum=bn_momentum))
        self.add_module('relu', nn.ReLU(inplace=True))
        self.add_module('conv', Conv2d(cin, cout, kernel_size=k, padding=k // 2, bias=False))

    def forward(self, x):
        return self.conv(bn_act(self.norm, x, relu=True))


class SpatialPyramidPooling(nn.Module):
    def __init__(self, cin, num_levels, bt_size, level_size, out_size, grids, bn_momentum):
        super().__init__()
        self.grids = grids
        self.spp = nn.Sequential()
        self.spp.add_module('spp_bn', BNReluConv(cin, bt_size, k=1, bn_momentum=bn_momentum))
        final = bt_size
        for i in range(num_levels):
            final += level_size
            self.spp.add_module('spp' + str(i), BNReluConv(bt_size, level_size, k=1, bn_momentum=bn_momentum))
        self.spp.add_module('spp_fuse', BNReluConv(final, out_size, k=1, bn_momentum=bn_momentum))

    def forward(self, x):
        h, w = x.shape[2:4]
        ar = w / h
        x = self.spp[0](x)
        levels = [x]
        for i in range(1, len(self.spp) - 1):
            grid = (self.grids[i - 1], max(1, round(ar * self.grids[i - 1])))
            levels.append(_up(self.spp[i](F.adaptive_avg_pool2d(x, grid)), (h, w)))
        return self.spp[-1](torch.cat(levels, 1))


class _Upsample(nn.Module):
    def __init__(self, cin, cskip, cout, k=3):
        super().__init__()
        self.bottleneck = BNReluConv(cskip, cin, k=1)
        self.blend_conv = BNReluConv(cin, cout, k=k)

    def forward(self, x, skip):
        skip = self.bottleneck(skip)
        return self.blend_conv(_up(x, skip.shape[2:4], skip))


class SwiftNetResNet(nn.Module):
    def __init__(self, layers=(2, 2, 2, 2), num_features=(128, 128, 128), spp_grids=(8, 4, 2, 1)):
        super().__init__()
        self.inplanes = 64
        self.img_cs = [64, 64, 128, 256, num_features[0]]
        self.conv1 = Conv2d(3, 64, kernel_size=7, stride=1, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = MaxPool3x3s2()
        skips = []
        self.layer1 = self._make_layer(64, layers[0])
        skips.append(self.inplanes)
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        skips.append(self.inplanes)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        skips.append(self.inplanes)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        spp_size = num_features[0]
        self.spp = SpatialPyramidPooling(self.inplanes, 3, bt_size=spp_size, level_size=spp_size // 3,
                                         out_size=num_features[0], grids=spp_grids, bn_momentum=0.024 / 2)
        ups = [_Upsample(num_features[1], skips[0], num_features[2]),
               _Upsample(num_features[0], skips[1], num_features[1]),
               _Upsample(num_features[0], skips[2], num_features[0])]
        self.upsample = nn.ModuleList(list(reversed(ups)))
        self.num_features = num_features[-1]
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False),
                                       BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    @staticmethod
    def forward_resblock(x, layers):
        skip = None
        for layer in layers:
            x, skip = layer(x)
        return x, skip

    def forward_stem(self, image):
        return self.maxpool(bn_act(self.bn1, self.conv1(image), relu=True))

    def forward_down(self, image):
        x = self.forward_stem(image)
        feats = []
        for layer in (self.layer1, self.layer2, self.layer3):
            x, skip = self.forward_resblock(x, layer)
            feats.append(skip)
        x, skip = self.forward_resblock(x, self.layer4)
        feats.append(self.spp(skip))
        return feats

    def forward_up(self, features, im_size=None):
        features = features[::-1]
        x = features[0]
        for skip, up in zip(features[1:], self.upsample):
            x = up(x, skip)
        return _up(x, im_size) if im_size is not None else x

    def forward(self, image, im_size=None):
        return self.forward_up(self.forward_down(image), im_size=im_size)


def SwiftNetRes18(num_feature=(128, 128, 128), pretrained_path=None):
    model = SwiftNetResNet((2, 2, 2, 2), num_feature)
    if pretrained_path is not None:
        model.load_state_dict(torch.load(pretrained_path, map_location='cpu'), strict=False)
    return model
