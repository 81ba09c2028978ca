"""Data-parallel plumbing of the hot path.

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU for tests).  Scenes are independent -- the batch index is
part of every hash key -- so ranks only exchange gradients (DDP bucketed
all-reduce overlapped with backward) and BatchNorm statistics (SyncBatchNorm),
exactly the two collectives of the reference (train_lc_nusc_tsd_full.py:80-84).
There is no data-path collective.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

__all__ = ['init_from_env', 'world', 'rank', 'wrap_model', 'max_over_ranks', 'scene_seed', 'shutdown']


def init_from_env(backend: str | None = None):
    """Initialise from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  A single process needs no process group."""
    world_size = int(os.environ.get('WORLD_SIZE', '1'))
    rnk = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        # (more local ranks than devices: only a gloo group can share a card -- tests on a one-GPU box; RCCL refuses)
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    force = os.environ.get('U2MKD_FORCE_DDP') == '1'     # measure the N>1 code path on one GPU
    if (world_size > 1 or force) and not dist.is_initialized():
        backend = backend or ('nccl' if use_cuda else 'gloo')
        kwargs = {}
        if backend == 'nccl':
            kwargs['device_id'] = torch.device('cuda', local)
        if force and 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', '29533'))
        dist.init_process_group(backend, rank=rnk, world_size=world_size, **kwargs)
    return rnk, world_size, local


def world() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def scene_seed(base: int, step: int = 0) -> int:
    """Scenes are sharded by rank (DistributedSampler semantics): rank r of W
    takes scene ``base + step * W + r``."""
    return base + step * world() + rank()


def _is_sync_bn(m) -> bool:
    """a BatchNorm whose batch statistics are merged over the ranks (torch's, or one of this package's forms)"""
    return isinstance(m, torch.nn.SyncBatchNorm) or 'Sync' in type(m).__name__ or bool(getattr(m, 'synchronised', False))


def wrap_model(model: torch.nn.Module, sync_bn: bool = True, bucket_cap_mb: int = 25):
    """DDP + (on GPU) SyncBatchNorm, as train_spformer.py:79-83.  Gradient buckets are
    all-reduced while backward is still running; ``gradient_as_bucket_view`` avoids a
    copy per bucket."""
    if world() == 1 and os.environ.get('U2MKD_FORCE_DDP') != '1':
        return model
    on_gpu = next(model.parameters()).is_cuda
    if sync_bn and on_gpu:
        from .lidar.point_voxel import SparseSyncBatchNorm
        model = SparseSyncBatchNorm.convert_sync_batchnorm(model)
    ids = [torch.cuda.current_device()] if on_gpu else None
    # broadcast_buffers=False: DDP's default re-broadcasts every buffer from rank 0 before each forward.  The only
    # buffers here are BatchNorm running statistics and step counters, and with every BatchNorm synchronised (or frozen:
    # the teacher) each rank computes the same values from the same all-gathered statistics in the same order, so the
    # broadcast would move hundreds of small tensors per step to overwrite them with themselves.
    # That argument needs EVERY train-mode BatchNorm under the wrap to be a synchronising one: a plain one (a custom
    # submodule, a torch fallback path) would let the ranks' running statistics drift apart -- then keep DDP's default.
    from torch.nn.modules.batchnorm import _BatchNorm
    plain = [n for n, m in model.named_modules()
             if isinstance(m, _BatchNorm) and m.training and m.track_running_stats and not _is_sync_bn(m)
             and any(p.requires_grad for p in m.parameters())]
    ddp = torch.nn.parallel.DistributedDataParallel(
        model, device_ids=ids, find_unused_parameters=False, gradient_as_bucket_view=True,
        bucket_cap_mb=bucket_cap_mb, broadcast_buffers=bool(plain))
    # The built-in reduction divides every parameter's gradient view by the world size as it becomes ready: one tiny
    # kernel per parameter per step (485 for the KD student, on the backward's stream).  The stock all-reduce hook does
    # the same division once per BUCKET before the same all-reduce.
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    ddp.register_comm_hook(state=None, hook=default_hooks.allreduce_hook)
    return ddp


def max_over_ranks(value: float) -> float:
    if world() == 1:
        return float(value)
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if world() > 1:
        dist.barrier()


def shutdown():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
