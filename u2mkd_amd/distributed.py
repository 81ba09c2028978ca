"""Data-parallel plumbing of the hot path.

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU for tests).  Scenes are independent -- the batch index is
part of every hash key -- so ranks only exchange gradients (DDP bucketed
all-reduce overlapped with backward) and BatchNorm statistics (SyncBatchNorm),
exactly the two collectives of the reference (train_lc_nusc_tsd_full.py:80-84).
There is no data-path collective.
"""
from __future__ import annotations

import contextlib
import os
import weakref

import torch
import torch.distributed as dist

__all__ = ['init_from_env', 'configure_runtime', 'multi_rank_process', 'hardware_queues', 'world', 'rank', 'wrap_model', 'BucketedGradientAverage', 'collective_counts', 'max_over_ranks', 'min_over_ranks', 'scene_seed', 'shutdown']


def init_from_env(backend: str | None = None):
    """Initialise from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  A single process needs no process group."""
    world_size = int(os.environ.get('WORLD_SIZE', '1'))
    rnk = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    configure_runtime(warn=False)      # (a launcher calls it itself before its first GPU call; here it may already be too late)
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


def multi_rank_process() -> bool:
    """True for a process that is (or is made to behave as) one rank of several: WORLD_SIZE > 1 in the launcher's environment,
    an initialised group of more than one rank, or U2MKD_FORCE_DDP=1 (the N > 1 code path on one GPU)."""
    if os.environ.get('U2MKD_FORCE_DDP') == '1':
        return True
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size() > 1
    try:
        return int(os.environ.get('WORLD_SIZE', '1')) > 1
    except ValueError:
        return False


def hardware_queues() -> int:
    """HIP hardware queues of this process as far as the environment tells (the runtime's default is 4)."""
    try:
        return int(os.environ.get('GPU_MAX_HW_QUEUES', '4'))
    except ValueError:
        return 4


def configure_runtime(warn=True):
    """Process-level HIP settings; must run before the process's first HIP call (the launchers -- bench.py,
    run_training.py -- call it on their first line; nothing is changed at ``import u2mkd_amd``): the runtime reads the
    variable at its initialisation only, so a late call warns and changes nothing.

    The number of HIP hardware queues, by the kind of process (round 5; same-box sweeps in NOTES N9.11):
    * ONE rank (the default bench line): the runtime's default, 4, is left alone.  The step keeps five streams busy -- student
      LiDAR (main), frozen teacher, camera branch, weight gradients, the next batch's geometry -- so one of them shares a
      queue; since the geometry is queued in slices between the phases of the step (train.KDStep: its host reads find their
      counts ready) that sharing costs nothing, while the slices on a queue of their own (5 / 6 / 8 queues) push the
      hardware scheduler over a cliff: 64.2 ms at 4 queues against 105-106 ms at 5, 6 and 8.
    * one rank OF SEVERAL (WORLD_SIZE > 1, or U2MKD_FORCE_DDP=1): ``GPU_MAX_HW_QUEUES=8`` and the geometry in one piece behind
      the backward (train._staged_geometry reads the same setting): the gradient reducer's and RCCL's streams need the room
      (N > 1 path at one rank: 70-73 ms at 8 queues, 76-77 ms at 4, with or without the slices)."""
    if 'GPU_MAX_HW_QUEUES' in os.environ:       # the user's choice wins
        return
    if not multi_rank_process():
        return
    if torch.cuda.is_initialized():
        if not warn:
            return
        import warnings
        warnings.warn('u2mkd_amd.distributed.configure_runtime() was called after the HIP runtime had been initialised: '
                      'GPU_MAX_HW_QUEUES keeps the runtime\'s default (4) and the streams of a multi-rank step share hardware queues '
                      '(a few percent of step time); call it before the first GPU call, or export GPU_MAX_HW_QUEUES=8',
                      RuntimeWarning, stacklevel=2)
        return
    os.environ['GPU_MAX_HW_QUEUES'] = '8'


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
    return isinstance(m, torch.nn.SyncBatchNorm)


class _ArmSink(torch.autograd.Function):
    """Identity on the wrapped module's outputs whose backward runs FIRST in a backward pass that reaches them: it books the
    reducer's end-of-backward callback even on a rank where (this step) no parameter receives a gradient, so every rank issues
    every bucket's collective (torch DDP's _DDPSink plays this role)."""

    @staticmethod
    def forward(ctx, reducer_ref, *tensors):
        ctx.reducer_ref = reducer_ref
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        red = ctx.reducer_ref()
        if red is not None:
            red._arm()
        return (None,) + grads


def _map_tensors(o, fn):
    """Apply ``fn`` to every tensor reachable through dicts, lists, tuples (named ones too) and the feature matrix of a
    SparseTensor / PointTensor (``.F``, replaced in place); containers of other kinds are returned unchanged.  (The sunk
    tensors are views made by one autograd.Function: in-place operations on a wrapped module's outputs raise, as with
    torch DDP's own sink.)"""
    if torch.is_tensor(o):
        return fn(o)
    if type(o) is dict:
        return {k: _map_tensors(v, fn) for k, v in o.items()}
    if isinstance(o, tuple) and hasattr(o, '_fields'):          # namedtuple: positional constructor
        return type(o)(*(_map_tensors(v, fn) for v in o))
    if type(o) in (list, tuple):
        return type(o)(_map_tensors(v, fn) for v in o)
    if isinstance(o, dict):                                     # (defaultdict, OrderedDict ...: keep the object, map its values)
        for k in list(o.keys()):
            o[k] = _map_tensors(o[k], fn)
        return o
    f = getattr(o, 'F', None)
    if torch.is_tensor(f):
        g = fn(f)
        if g is not None and g is not f:
            o.F = g
    return o


class BucketedGradientAverage(torch.nn.Module):
    """Data-parallel gradient averaging, the role DistributedDataParallel plays in train_lc_nusc_tsd_full.py:80-84: the
    parameters are cut into buckets in reverse registration order (~ the order their gradients appear), and as soon as
    the last gradient of a bucket has been accumulated the bucket is flattened and all-reduced (average) on a side
    stream while the backward keeps running; when the backward ends every ``p.grad`` IS its slice of the reduced buffer.

    Collectives are issued in BUCKET INDEX ORDER on every rank (bucket i only after buckets 0..i-1, as DDP's reducer does):
    a bucket that completes early waits for its predecessors, so two ranks whose gradients arrive in a different order --
    a data-dependent branch, a stage without points on one rank -- still pair the same buffers.  The end-of-backward
    callback that flushes the buckets with missing gradients (zeros for those) is booked by a sink on the module's outputs,
    i.e. by every backward pass that reaches them, and the per-step state is reset at every forward, so a backward that
    raised (out of memory, a NaN check) does not leave the reducer disarmed.  A gradient hook that fires twice for one
    parameter in a pass (re-entrant checkpointing) raises.  ``no_sync()`` accumulates local gradients without collectives.

    Why not torch's DDP: with ``gradient_as_bucket_view`` every parameter's gradient is COPIED into its bucket slice as
    it is produced -- one copy kernel per parameter per step, 485 for the KD student: +515 launches and +3.2 ms of kernel
    time per step at ONE rank (profiles/r4: the N>1 path cost 86 ms against 80 ms before any byte moved).  Here a bucket is
    flattened by one multi-tensor copy, the division by the world size is the collective's own (ReduceOp.AVG on RCCL), and
    the module's buffers are not re-broadcast (see ``wrap_model``).  Same keys as DDP in ``state_dict()`` (`module.`
    prefix: the reference's checkpoints, core/nusc_trainers.py:180,198)."""

    def __init__(self, module: torch.nn.Module, bucket_cap_mb: float = 25.0, broadcast_buffers: bool = False):
        super().__init__()
        self.module = module
        self.broadcast_buffers = bool(broadcast_buffers)
        self._world = world()
        params = [p for p in module.parameters() if p.requires_grad]
        self._on_gpu = bool(params) and params[0].is_cuda
        self._avg = dist.is_initialized() and dist.get_backend() == 'nccl'
        # every rank starts from rank 0's values (DDP's constructor does the same)
        if self._world > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, 0)
        self._buckets = []
        self._bucket_of = {}
        cap = int(bucket_cap_mb * 2 ** 20)
        cur, size = [], 0
        for p in reversed(params):
            nbytes = p.numel() * p.element_size()
            if cur and (size + nbytes > cap or cur[0].dtype != p.dtype):
                self._add_bucket(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self._add_bucket(cur)
        self._armed = False
        self._sync = True
        self._next = 0                # the next bucket (index) to launch
        self.collectives = {'count': 0, 'bytes': 0}     # all-reduces issued by the last backward pass (bench.py reports them)
        self._self_ref = weakref.ref(self)
        for p in params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    def _add_bucket(self, plist):
        flat = torch.zeros(sum(p.numel() for p in plist), dtype=plist[0].dtype, device=plist[0].device)
        views, o = [], 0
        for p in plist:
            views.append(flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        b = {'params': list(plist), 'flat': flat, 'views': views, 'pending': len(plist), 'seen': set(), 'ready': False,
             'launched': False, 'work': None, 'streams': {}, 'index': len(self._buckets)}
        for p in plist:
            self._bucket_of[p] = b
        self._buckets.append(b)

    def _reset(self):
        """Per-pass state; also what a backward pass that raised left behind (its end-of-backward callback never ran)."""
        for b in self._buckets:
            b['pending'] = len(b['params'])
            b['seen'] = set()
            b['ready'] = b['launched'] = False
            b['streams'] = {}
            b['work'] = None
            b['done'] = None
        self._armed = False
        self._next = 0

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside accumulate LOCAL gradients (no collectives), as DistributedDataParallel.no_sync."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def forward(self, *args, **kwargs):
        if self.broadcast_buffers and self._world > 1 and self.training:
            with torch.no_grad():
                for t in self.module.buffers():
                    dist.broadcast(t, 0)
        out = self.module(*args, **kwargs)
        if torch.is_grad_enabled() and self._sync:
            self._reset()
            tensors = []
            _map_tensors(out, lambda t: (tensors.append(t) if t.requires_grad else None, t)[1])
            if tensors:
                sunk = iter(_ArmSink.apply(self._self_ref, *tensors))
                out = _map_tensors(out, lambda t: next(sunk) if t.requires_grad else t)
        return out

    # ---- backward side
    def _arm(self):
        if self._sync and not self._armed:        # first event of this backward: finish when the engine is done
            self._armed = True
            self.collectives = {'count': 0, 'bytes': 0}
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)

    def _on_grad(self, p):
        if not self._sync:
            return
        self._arm()
        b = self._bucket_of[p]
        if id(p) in b['seen'] or b['launched']:
            raise RuntimeError('BucketedGradientAverage: a parameter received its gradient twice in one backward pass '
                               '(re-entrant backward / activation checkpointing is not supported); use no_sync() for '
                               'gradient accumulation over several passes')
        b['seen'].add(id(p))
        if self._on_gpu:                          # (gradients come from several streams: camera branch, weight-gradient stream)
            st = torch.cuda.current_stream()
            b['streams'][st.stream_id] = st
        b['pending'] -= 1
        if b['pending'] == 0:
            b['ready'] = True
            self._launch_ready()

    def _launch_ready(self):
        """Launch, in index order, every bucket whose predecessors have all been launched.  Not while the step's library
        kernels run in reduced precision (deferred.overlap_ok, NOTES N9): RCCL's reduction kernels would run next to MIOpen's
        bf16 backward kernels, which change the results of concurrent kernels -- the buckets then all go at the end of the
        backward (``_finish``), behind the last of them."""
        if self._on_gpu:
            from . import deferred
            if not deferred.overlap_ok():
                return
        while self._next < len(self._buckets) and self._buckets[self._next]['ready']:
            self._reduce(self._buckets[self._next])
            self._next += 1

    def _all_reduce(self, b):
        self.collectives['count'] += 1
        self.collectives['bytes'] += b['flat'].numel() * b['flat'].element_size()
        if self._world > 1:
            b['work'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, async_op=True)

    def _reduce(self, b):
        """Flatten the bucket and start its all-reduce.  No stream of our own (a sixth busy stream costs the step 50 ms,
        deferred.py).  When weight gradients are still running on their side stream (deferred.py: they are joined at the END
        of the backward) the bucket goes THERE, behind them: the stream the completing gradient arrived on -- the backward's
        critical one -- is not made to wait in the middle of the backward.  Otherwise it goes to the arriving stream.  Every
        other stream that produced one of the bucket's gradients is joined by an event recorded now: it covers what that
        stream has produced so far."""
        b['launched'] = True
        with torch.no_grad():
            if not self._on_gpu:
                grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b['params']]
                torch._foreach_copy_(b['views'], grads)
                self._all_reduce(b)
                return
            from . import deferred
            cur = torch.cuda.current_stream()
            pend = deferred.pending()
            run = pend[0] if pend else cur
            if run is not cur:
                run.wait_event(cur.record_event())
                for other in pend[1:]:
                    run.wait_event(other.record_event())
            for sid, st in b['streams'].items():
                if sid != run.stream_id:
                    run.wait_event(st.record_event())
            with torch.cuda.stream(run):
                grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b['params']]
                torch._foreach_copy_(b['views'], grads)
                self._all_reduce(b)
                b['done'] = run.record_event()

    def _finish(self):
        """End of the backward: the buckets not yet launched go now, in index order (zeros for the parameters that received
        no gradient, so every rank issues the same collectives), the caller's stream joins every bucket's copy /
        collective, and every ``p.grad`` becomes its slice of the reduced bucket."""
        for b in self._buckets[self._next:]:
            self._reduce(b)
        self._next = len(self._buckets)
        with torch.no_grad():
            for b in self._buckets:
                if self._on_gpu and b.get('done') is not None:
                    torch.cuda.current_stream().wait_event(b['done'])
                    b['done'] = None
                if b['work'] is not None:
                    b['work'].wait()              # (RCCL: orders the current stream behind the collective, no host wait)
                    b['work'] = None
                    if not self._avg:             # gloo has no averaging reduction
                        b['flat'].div_(self._world)
        for b in self._buckets:
            for p, v in zip(b['params'], b['views']):
                p.grad = v
        self._reset()


def wrap_model(model: torch.nn.Module, sync_bn: bool = True, bucket_cap_mb: int = 25):
    """Data-parallel wrap + (on GPU) SyncBatchNorm, as train_spformer.py:79-83: ``BucketedGradientAverage`` above in
    DistributedDataParallel's place (gradient buckets all-reduced while the backward is still running)."""
    if world() == 1 and os.environ.get('U2MKD_FORCE_DDP') != '1':
        return model
    on_gpu = next(model.parameters()).is_cuda
    if sync_bn and on_gpu:
        from .lidar.point_voxel import SparseSyncBatchNorm
        model = SparseSyncBatchNorm.convert_sync_batchnorm(model)
    # Buffers are NOT re-broadcast before every forward (DDP's default does): the only buffers here are BatchNorm running
    # statistics and step counters, and with every BatchNorm synchronised (or frozen: the teacher) each rank computes the
    # same values from the same all-gathered statistics in the same order -- the broadcast would move hundreds of small
    # tensors per step to overwrite them with themselves.  That argument needs EVERY train-mode BatchNorm under the wrap
    # to be a synchronising one: a plain one (a custom submodule, a torch fallback path, the CPU path) would let the
    # ranks' running statistics drift apart -- then the buffers are broadcast as DDP would.
    from torch.nn.modules.batchnorm import _BatchNorm
    plain = [n for n, m in model.named_modules()
             if isinstance(m, _BatchNorm) and m.training and m.track_running_stats and not _is_sync_bn(m)
             and any(p.requires_grad for p in m.parameters())]
    return BucketedGradientAverage(model, bucket_cap_mb=bucket_cap_mb, broadcast_buffers=bool(plain))


def collective_counts(reducer=None, reset=False):
    """Collectives this process has issued (or, at world size 1 on the forced N > 1 path, would issue) since the last reset:
    SyncBatchNorm statistics (one all_gather of [2C+1] per synchronising BatchNorm forward, one all_reduce of [2C] per
    backward -- the reference's count, core/models/utils.py:138-220 -> torch.nn.SyncBatchNorm) and, with ``reducer`` (the
    BucketedGradientAverage), the gradient buckets of its LAST backward pass."""
    from .torchsparse.nn import functional as F
    out = {k: {'count': v[0], 'bytes': v[1]} for k, v in F.COLLECTIVES.items()}
    if reducer is not None and hasattr(reducer, 'collectives'):
        out['gradient_buckets_last_pass'] = dict(reducer.collectives)
    if reset:
        for v in F.COLLECTIVES.values():
            v[0] = v[1] = 0
    return out


def max_over_ranks(value: float) -> float:
    if world() == 1:
        return float(value)
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(value: float) -> float:
    if world() == 1:
        return float(value)
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def barrier():
    if world() > 1:
        dist.barrier()


def shutdown():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
