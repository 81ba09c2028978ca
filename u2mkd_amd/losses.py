"""Segmentation losses of the hot path (row a17): Lovasz-softmax + cross entropy
("lovasz" in the reference configs is this mix; core/criterions.py:40-52,
73-101, 129-146, 159-174).

The Lovasz term is evaluated for all classes at once: one batched descending
sort of the [P, C] error matrix instead of C separate sorts of length P, and no
host synchronisation (absent classes are masked, not skipped by a Python
``if fg.sum() == 0``).  Same value and gradient as the reference formulation.
"""
import os

import torch
from torch import nn
import torch.nn.functional as F

__all__ = ['lovasz_softmax_flat', 'Lovasz_softmax', 'MixLovaszCrossEntropy']


def lovasz_softmax_flat(probas: torch.Tensor, labels: torch.Tensor, valid: torch.Tensor = None) -> torch.Tensor:
    """probas [P, C], labels [P]; classes='present'.  ``valid`` (bool [P]) marks the rows that
    count; the reference compacts them first (``probas[valid]``, a host synchronisation) -- here
    ignored rows get error -1, which sorts them behind every valid row (errors >= 0), their
    foreground flag is 0 and their error enters the dot product as 0: the prefix sums of the
    valid rows, hence value and gradient, are exactly those of the compacted input.

    Works on the [C, P] transpose so that the sort and the two cumulative sums
    run along the contiguous dimension (a dim-0 cumsum of a [P, C] tensor is a
    14 ms kernel at P = 80k on MI355X; the row-wise scan is ~20 us)."""
    if probas.numel() == 0:
        return probas.sum() * 0.
    P, C = probas.shape
    pt = probas.t().contiguous()                                 # [C, P]
    fg = (labels.unsqueeze(0) == torch.arange(C, device=labels.device).unsqueeze(1))
    if valid is not None:
        fg = fg & valid.unsqueeze(0)
    fg = fg.to(probas.dtype)
    errors = (fg - pt).abs()
    if valid is not None:
        errors = torch.where(valid.unsqueeze(0), errors, errors.new_full((), -1.))
    # One flat sort instead of a row-wise one: torch answers sort(dim=1) of [17, 80 000] with ~77 merge-sort launches
    # (0.65 ms on MI355X, twice per KD step, between the forward and the backward); the composite key 4c - error in
    # float64 (exact: an integer plus a float32) orders class-major and error-descending in ONE radix sort, and block c of
    # the result holds exactly the P entries of class c.
    with torch.no_grad():
        cls = torch.arange(C, device=probas.device).unsqueeze(1)
        keys = (cls * 4).to(torch.float64) - errors.detach().to(torch.float64)
        perm = torch.sort(keys.view(-1))[1].view(C, P) - cls * P
    errors_sorted = torch.gather(errors, 1, perm)
    fg_sorted = torch.gather(fg, 1, perm)
    if C * P < (1 << 24):
        # ONE flat scan instead of C row scans (torch gives a scan along the last dimension one workgroup per row: 17
        # workgroups walking 80 000 elements each, 150 us between the forward and the backward, twice per KD step; the 1-D
        # scan is rocPRIM's device scan).  Exact: the flags are 0 / 1, every prefix an integer below 2^24.
        flat = fg_sorted.reshape(-1).cumsum(0).view(C, P)
        ends = flat[:, -1:]
        starts = torch.cat([ends.new_zeros(1, 1), ends[:-1]], 0)
        gts = ends - starts                                      # [C, 1]
        cs = flat - starts
    else:
        gts = fg_sorted.sum(1, keepdim=True)
        cs = fg_sorted.cumsum(1)
    intersection = gts - cs
    union = gts + (torch.arange(1, P + 1, device=probas.device, dtype=probas.dtype).unsqueeze(0) - cs)
    jaccard = 1. - intersection / union
    jaccard = torch.cat([jaccard[:, :1], jaccard[:, 1:] - jaccard[:, :-1]], 1)
    per_class = (errors_sorted.clamp(min=0.) * jaccard).sum(1)   # [C]
    present = (gts.squeeze(1) > 0).to(probas.dtype)
    return (per_class * present).sum() / present.sum().clamp(min=1.)


class _LovaszFunction(torch.autograd.Function):
    """lovasz_softmax_flat(probas, labels, labels != ignore_index) on csrc/lovasz.hip: the element-wise chains around the sort as
    three launches forward and one backward (the torch formulation above: ~45 + ~18 launches, twice per KD step, between the
    forward and the backward of the critical stream).  Same arithmetic per element; the final sums in a fixed order."""

    @staticmethod
    def forward(ctx, probas, labels, ignore_index):
        from . import _lib as L
        P, C = probas.shape
        probas = probas.contiguous().float()
        labels = labels.contiguous().long()
        dev = probas.device
        st = L.stream()
        errors = torch.empty(C, P, dtype=torch.float32, device=dev)
        keys = torch.empty(C, P, dtype=torch.float64, device=dev)
        L.call('u2mkd_lovasz_errors', L.ptr(probas), L.ptr(labels), int(ignore_index), P, C, L.ptr(errors), L.ptr(keys), st)
        perm = torch.sort(keys.view(-1))[1]
        fg_sorted = torch.empty(C * P, dtype=torch.int32, device=dev)
        L.call('u2mkd_lovasz_gather', L.ptr(perm), L.ptr(labels), int(ignore_index), P, C, L.ptr(fg_sorted), st)
        csum = fg_sorted.cumsum(0)                                  # int64: exact
        jgrad = torch.empty(C * P, dtype=torch.float32, device=dev)
        partial = torch.empty(int(L.load().u2mkd_lovasz_partials(P, C)), dtype=torch.float32, device=dev)
        stats = torch.empty(2 + C, dtype=torch.float32, device=dev)
        L.call('u2mkd_lovasz_terms', L.ptr(perm), L.ptr(errors), L.ptr(csum), L.ptr(fg_sorted), P, C, L.ptr(jgrad), L.ptr(partial),
               L.ptr(stats), st)
        ctx.save_for_backward(probas, labels, perm, jgrad, stats)
        ctx.ignore_index = int(ignore_index)
        return stats[0]

    @staticmethod
    def backward(ctx, g):
        from . import _lib as L
        probas, labels, perm, jgrad, stats = ctx.saved_tensors
        P, C = probas.shape
        g = g.contiguous().float().reshape(1)
        d = torch.empty_like(probas)
        L.call('u2mkd_lovasz_backward', L.ptr(g), L.ptr(stats), L.ptr(perm), L.ptr(jgrad), L.ptr(probas), L.ptr(labels),
               ctx.ignore_index, P, C, L.ptr(d), L.stream())
        return d, None, None


class Lovasz_softmax(nn.Module):
    def __init__(self, classes='present', ignore_index=0):
        super().__init__()
        assert classes == 'present', 'the reference configs use classes="present" only'
        self.ignore_index = ignore_index

    def forward(self, probas, labels):
        if (probas.is_cuda and probas.dim() == 2 and probas.dtype == torch.float32 and 0 < probas.shape[0] < (1 << 24)
                and probas.shape[1] <= 4096 and -2 ** 31 <= self.ignore_index < 2 ** 31):
            return _LovaszFunction.apply(probas, labels, self.ignore_index)
        return lovasz_softmax_flat(probas, labels, labels != self.ignore_index)


class _CrossEntropyFunction(torch.autograd.Function):
    """nn.CrossEntropyLoss(ignore_index, reduction='mean') on [P, C] fp32 logits as one pass per direction
    (csrc/lovasz.hip: ce_forward / ce_backward; torch: log_softmax + nll_loss, whose two reductions are single-workgroup
    kernels -- 57 + 76 us at 80 000 x 17, twice per KD step on the critical stream)."""

    @staticmethod
    def forward(ctx, x, labels, ignore_index):
        from . import _lib as L
        x = x.contiguous()
        labels = labels.contiguous()
        P, C = x.shape
        lse = torch.empty(P, dtype=torch.float32, device=x.device)
        partial = torch.empty(int(L.load().u2mkd_ce_partials(P)), dtype=torch.float32, device=x.device)
        stats = torch.empty(2, dtype=torch.float32, device=x.device)
        L.call('u2mkd_ce_forward', L.ptr(x), L.ptr(labels), int(ignore_index), P, C, L.ptr(lse), L.ptr(partial), L.ptr(stats), L.stream())
        ctx.save_for_backward(x, labels, lse, stats)
        ctx.ignore_index = int(ignore_index)
        return stats[0]

    @staticmethod
    def backward(ctx, g):
        from . import _lib as L
        x, labels, lse, stats = ctx.saved_tensors
        P, C = x.shape
        g = g.contiguous().float().reshape(1)
        dx = torch.empty_like(x)
        L.call('u2mkd_ce_backward', L.ptr(g), L.ptr(stats), L.ptr(x), L.ptr(lse), L.ptr(labels), ctx.ignore_index, P, C, L.ptr(dx),
               L.stream())
        return dx, None, None


class _KLDivFunction(torch.autograd.Function):
    """nn.KLDivLoss(reduction='batchmean')(log_softmax(s, 1), softmax(t[index], 1)) on [P, C] fp32 logits, the target side
    without gradient, as one pass per direction (csrc/lovasz.hip: kl_forward / kl_backward)."""

    @staticmethod
    def forward(ctx, s, t, index):
        from . import _lib as L
        s, t = s.contiguous(), t.contiguous()
        index = index.contiguous() if index is not None else None
        P, C = s.shape
        rows = torch.empty(P, 3, dtype=torch.float32, device=s.device)
        partial = torch.empty(int(L.load().u2mkd_ce_partials(P)), dtype=torch.float32, device=s.device)
        stats = torch.empty(1, dtype=torch.float32, device=s.device)
        L.call('u2mkd_kl_forward', L.ptr(s), L.ptr(t), L.ptr(index), P, C, L.ptr(rows), L.ptr(partial), L.ptr(stats), L.stream())
        ctx.save_for_backward(s, t, index, rows)
        return stats[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from . import _lib as L
        s, t, index, rows = ctx.saved_tensors
        P, C = s.shape
        g = g.contiguous().float().reshape(1)
        ds = torch.empty_like(s)
        L.call('u2mkd_kl_backward', L.ptr(g), L.ptr(s), L.ptr(t), L.ptr(index), L.ptr(rows), P, C, L.ptr(ds), L.stream())
        return ds, None, None


_FUSED_KL = os.environ.get('U2MKD_FUSED_KL', '1') != '0'      # 0: log_softmax / softmax / nn.KLDivLoss as torch runs them


def kl_div_logits(s, t, index=None, criterion=None):
    """``criterion(log_softmax(s, 1), softmax(t[index], 1))`` for an nn.KLDivLoss(reduction='batchmean') -- the KD step's `kl` term
    (t detached); on the device in one pass per direction."""
    if (_FUSED_KL and s.is_cuda and s.dim() == 2 and s.dtype == torch.float32 and t.dtype == torch.float32 and s.shape[0] > 0
            and (criterion is None or (isinstance(criterion, nn.KLDivLoss) and criterion.reduction == 'batchmean'
                                       and not criterion.log_target))
            and (index is None or (index.dtype == torch.int64 and index.shape[0] == s.shape[0]))
            and (index is not None or t.shape == s.shape) and t.shape[1] == s.shape[1]):
        return _KLDivFunction.apply(s, t.detach(), index)
    tt = t.detach() if index is None else t.detach().index_select(0, index)
    crit = criterion if criterion is not None else nn.KLDivLoss(reduction='batchmean')
    return crit(F.log_softmax(s, dim=1), F.softmax(tt, dim=1))


_FUSED_CE = os.environ.get('U2MKD_FUSED_CE', '1') != '0'      # 0: torch's log_softmax + nll_loss (the formulation the fused pass is tested against)


class MixLovaszCrossEntropy(nn.Module):
    def __init__(self, weight=None, classes='present', ignore_index=255):
        super().__init__()
        self.ignore_index = ignore_index
        self.lovasz = Lovasz_softmax(classes, ignore_index=ignore_index)
        self.ce = nn.CrossEntropyLoss(weight=weight, ignore_index=ignore_index)

    def _ce(self, x, y):
        if (_FUSED_CE and self.ce.weight is None and self.ce.label_smoothing == 0.0 and x.is_cuda and x.dim() == 2
                and x.dtype == torch.float32 and y.dtype == torch.int64 and x.shape[0] > 0 and -2 ** 31 <= self.ignore_index < 2 ** 31):
            return _CrossEntropyFunction.apply(x, y, self.ignore_index)
        return self.ce(x, y)

    def forward(self, x, y):
        return self.lovasz(F.softmax(x, 1), y) + self._ce(x, y)
