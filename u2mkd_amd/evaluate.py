"""Evaluation path: the eval branch of the reference trainers and the MeanIoU callback.

* :func:`voxel_logits_to_point_predictions` -- what ``_run_step`` does when the model is not training
  (core/nusc_trainers.py:367-418, core/spformer_trainer.py:95-117): per scene, the voxel logits are gathered back
  to the raw points through ``inverse_map`` and arg-maxed; multi-sweep inputs keep the key-frame points only.
* :class:`MeanIoU` -- core/callbacks.py:91-171: per-class seen / correct / predicted counters over an epoch,
  summed over the ranks, IoU_c = correct / (seen + predicted - correct), classes never seen count as 1 except the
  ignore label, mean over the kept classes.  The counters are three ``bincount`` calls on the device per step
  instead of 3 x num_classes host round trips, and one all_reduce of [3, num_classes] instead of 3 x num_classes."""
from __future__ import annotations

import numpy as np
import torch

__all__ = ['voxel_logits_to_point_predictions', 'MeanIoU']


def voxel_logits_to_point_predictions(logits, voxel_batch, inverse_map, inverse_batch, keyframe_mask_full=None):
    """Per-point class predictions of a batch.

    logits [Nv, C] per voxel; voxel_batch [Nv] scene of every voxel (``lidar.C[:, -1]``); inverse_map [Np] index of
    every raw point's voxel WITHIN its scene (``inverse_map.F``); inverse_batch [Np] scene of every point
    (``inverse_map.C[:, -1]``).  Equals the reference's loop ``outputs[cur_scene_pts][cur_inv].argmax(1)`` over the
    scenes, concatenated in scene order; ``keyframe_mask_full`` (multisweeps != 0) keeps the key-frame points."""
    voxel_batch = voxel_batch.long()
    inverse_batch = inverse_batch.long()
    n_scene = int(max(int(voxel_batch.max()) if voxel_batch.numel() else -1,
                      int(inverse_batch.max()) if inverse_batch.numel() else -1)) + 1
    # global row of the j-th voxel of scene s: a stable sort by scene keeps the in-scene order of boolean indexing
    order = torch.argsort(voxel_batch, stable=True)
    counts = torch.bincount(voxel_batch, minlength=n_scene)
    starts = torch.cumsum(counts, 0) - counts
    # points must come out scene by scene as well (the reference concatenates per scene)
    p_order = torch.argsort(inverse_batch, stable=True)
    rows = order[starts[inverse_batch[p_order]] + inverse_map.long()[p_order]]
    pred = logits[rows].argmax(1)
    if keyframe_mask_full is not None:
        pred = pred[keyframe_mask_full[p_order]]
    return pred


class MeanIoU:
    """core/callbacks.py:91-171 without the torchpack Callback plumbing: ``before_epoch`` / ``after_step`` /
    ``after_epoch`` carry the same arithmetic; ``after_epoch`` returns (mIoU, per-class IoUs of the kept classes)."""

    def __init__(self, num_classes: int, ignore_label: int, output_tensor: str = 'outputs', target_tensor: str = 'targets',
                 name: str = 'iou'):
        self.num_classes, self.ignore_label, self.name = num_classes, ignore_label, name
        self.output_tensor, self.target_tensor = output_tensor, target_tensor
        self.before_epoch()

    def before_epoch(self):
        self.total_seen = np.zeros(self.num_classes)
        self.total_correct = np.zeros(self.num_classes)
        self.total_positive = np.zeros(self.num_classes)
        self._dev = None

    def after_step(self, output_dict):
        outputs = torch.as_tensor(output_dict[self.output_tensor])
        targets = torch.as_tensor(output_dict[self.target_tensor]).to(outputs.device)
        keep = targets != self.ignore_label
        outputs, targets = outputs[keep].long(), targets[keep].long()
        c = self.num_classes
        upd = torch.stack([torch.bincount(targets, minlength=c)[:c],
                           torch.bincount(targets[outputs == targets], minlength=c)[:c],
                           torch.bincount(outputs.clamp(0, c - 1), minlength=c)[:c]]).double()
        self._dev = upd if self._dev is None else self._dev + upd.to(self._dev.device)

    def after_epoch(self):
        import torch.distributed as dist
        acc = self._dev if self._dev is not None else torch.zeros(3, self.num_classes, dtype=torch.float64)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
            acc = acc.to(dev)
            dist.all_reduce(acc)                       # the reference: 3 x num_classes scalar all-reduces
        acc = acc.cpu().numpy()
        self.total_seen, self.total_correct, self.total_positive = (self.total_seen + acc[0], self.total_correct + acc[1],
                                                                     self.total_positive + acc[2])
        self._dev = None
        ious = []
        for i in range(self.num_classes):
            if self.total_seen[i] == 0:
                if i == self.ignore_label:
                    continue
                ious.append(1)
            else:
                ious.append(self.total_correct[i] / (self.total_seen[i] + self.total_positive[i] - self.total_correct[i]))
        return float(np.mean(ious)), ious
