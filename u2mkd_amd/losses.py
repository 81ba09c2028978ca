"""Segmentation losses of the hot path (row a17): Lovasz-softmax + cross entropy
("lovasz" in the reference configs is this mix; core/criterions.py:40-52,
73-101, 129-146, 159-174).

The Lovasz term is evaluated for all classes at once: one batched descending
sort of the [P, C] error matrix instead of C separate sorts of length P, and no
host synchronisation (absent classes are masked, not skipped by a Python
``if fg.sum() == 0``).  Same value and gradient as the reference formulation.
"""
import torch
from torch import nn
import torch.nn.functional as F

__all__ = ['lovasz_softmax_flat', 'Lovasz_softmax', 'MixLovaszCrossEntropy']


def lovasz_softmax_flat(probas: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """probas [P, C] (rows of valid points only), labels [P]; classes='present'."""
    if probas.numel() == 0:
        return probas.sum() * 0.
    P, C = probas.shape
    fg = F.one_hot(labels, C).to(probas.dtype)                 # [P, C]
    errors = (fg - probas).abs()
    errors_sorted, perm = torch.sort(errors, 0, descending=True)
    fg_sorted = torch.gather(fg, 0, perm)
    gts = fg_sorted.sum(0, keepdim=True)                        # [1, C]
    intersection = gts - fg_sorted.cumsum(0)
    union = gts + (1. - fg_sorted).cumsum(0)
    jaccard = 1. - intersection / union
    jaccard = torch.cat([jaccard[:1], jaccard[1:] - jaccard[:-1]], 0)
    per_class = (errors_sorted * jaccard).sum(0)                # [C]
    present = (gts.squeeze(0) > 0).to(probas.dtype)
    return (per_class * present).sum() / present.sum().clamp(min=1.)


class Lovasz_softmax(nn.Module):
    def __init__(self, classes='present', ignore_index=0):
        super().__init__()
        assert classes == 'present', 'the reference configs use classes="present" only'
        self.ignore_index = ignore_index

    def forward(self, probas, labels):
        valid = labels != self.ignore_index
        return lovasz_softmax_flat(probas[valid], labels[valid])


class MixLovaszCrossEntropy(nn.Module):
    def __init__(self, weight=None, classes='present', ignore_index=255):
        super().__init__()
        self.ignore_index = ignore_index
        self.lovasz = Lovasz_softmax(classes, ignore_index=ignore_index)
        self.ce = nn.CrossEntropyLoss(weight=weight, ignore_index=ignore_index)

    def forward(self, x, y):
        return self.lovasz(F.softmax(x, 1), y) + self.ce(x, y)
