"""hipGraph capture of the STATIC-SHAPE pieces of the training step.

The LiDAR side of the KD step has data-dependent shapes (voxel sets, kernel maps) and cannot be captured, but the
camera side is image-shaped: the SwiftNet stem + ResNet layers, the LiDAR->camera fusion convs and the pixel
decoder see the same tensor shapes every step.  They are also the part of the step with the most expensive
launches on the host (every MIOpen convolution / BatchNorm call walks its solver database before it launches:
~40-100 us of host time per call, against ~10 us for a plain kernel), and the step is host-bound
(DESIGN.md section 6).  Each such piece is captured once -- forward and backward -- into a pair of hipGraphs
(three warm-up passes on a side stream, then capture of the forward and of torch.autograd.grad over it into one
private memory pool: the scheme of torch.cuda.make_graphed_callables), and a training step replays them: ~18 graph
launches instead of ~700 kernel launches.  The captured kernels, their order and their arguments are the eager
ones, so the results are the eager results.

The OUTPUTS of a replayed piece are views of the graph's static memory: they are valid until the piece's next
replay (the next training step).  Anything kept across steps (logged features, an exponential average of an
activation) must be cloned by the holder.  Within a step the piece guards itself: a second forward of the same
piece before the first one's backward runs eagerly instead of overwriting the first one's saved activations.

Parameter gradients come out of a replay in static buffers, handed to autograd without a copy; ``.grad`` may
therefore alias such a buffer until the next ``zero_grad(set_to_none=True)``.  A backward pass that finds a
``.grad`` still aliasing its buffer (gradient accumulation, or zero_grad(set_to_none=False)) first moves that
gradient into memory of its own, so accumulating works as in eager mode.

What is NOT captured (the piece then runs eagerly, as before): evaluation / no-grad passes, autocast regions,
pieces holding a SyncBatchNorm (a collective inside a capture) and a new input signature beyond the first
``_MAX_SHAPES`` per piece.

Round 3: capture is OFF by default (``U2MKD_CAMERA_GRAPH=1`` turns it on).  With the geometry of the next batch prepared
between forward and backward the host runs ~25 ms ahead of the GPU in the steady state, so the launches a replay saves
no longer shorten the step: measured in the same process pair 78.8 ms eager against 80.0 ms replayed, and 77.9 against
79.1 ms on a second box (a replay is one indivisible node of its stream; the eager kernels interleave with the other
streams' work at kernel granularity)."""
import atexit
import os
import warnings
import weakref

import torch
from torch import nn
from torch.utils._pytree import tree_flatten, tree_unflatten

__all__ = ['StaticPiece', 'PieceCache', 'graphs_enabled']

_ENABLED = os.environ.get('U2MKD_CAMERA_GRAPH', '0') != '0'
_MAX_SHAPES = 2


def graphs_enabled():
    return _ENABLED


_LIVE = weakref.WeakSet()           # every StaticPiece that may hold captured graphs


@atexit.register
def _release_graphs():
    """Destroy the captured graphs (and their private pools) while the HIP runtime is still up: left to interpreter
    shutdown, a hipGraphExec can be destroyed after the runtime's own teardown has begun (seen once as an abort at
    the exit of an otherwise green test process)."""
    pieces = list(_LIVE)
    if not pieces:
        return
    try:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:                                           # noqa: BLE001 -- nothing left to protect at exit
        pass
    for p in pieces:
        p._records.clear()


class PieceCache(dict):
    """name -> StaticPiece of one model.  Captured graphs belong to the model instance that made them: a copy of
    the model (deepcopy, pickle) starts with an empty cache."""

    def __deepcopy__(self, memo):
        return PieceCache()

    def __reduce__(self):
        return (PieceCache, ())


class _Record:
    """One captured (forward graph, backward graph) pair and its static tensors."""
    __slots__ = ('fwd', 'bwd', 'sample', 'params', 'outs', 'spec', 'gouts', 'gins', 'pool', 'pending')


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rec, *inputs):
        for s, a in zip(rec.sample, inputs):                   # (the parameters follow the user tensors: in place already)
            if s.data_ptr() != a.data_ptr():
                s.copy_(a)
        rec.fwd.replay()
        # the static activations now belong to THIS forward until its backward has run or its graph has been dropped
        rec.pending = weakref.ref(ctx)
        ctx.rec = rec
        return tuple(o.detach() for o in rec.outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        rec = ctx.rec
        n = len(rec.sample)
        for p, b in zip(rec.params, rec.gins[n:]):
            g = p.grad
            if g is not None and b is not None and g.data_ptr() == b.data_ptr():
                p.grad = g.clone()                              # accumulating: keep the old gradient out of the replay's way
        for g, grad in zip(rec.gouts, grads):
            if g is not None and g.data_ptr() != grad.data_ptr():
                g.copy_(grad)
        rec.bwd.replay()
        rec.pending = None
        return (None,) + tuple(b.detach() if b is not None else None for b in rec.gins)


class StaticPiece:
    """``fn(*tensors) -> tensor | tuple of tensors`` over the parameters / buffers of ``mods``, replayed from a
    hipGraph when the call qualifies (module docstring) and run eagerly otherwise.  Not an nn.Module: it holds
    references to modules that already live in the model's tree and must not add state-dict keys."""

    def __init__(self, name, fn, mods):
        self.name, self.fn, self.mods = name, fn, list(mods)
        self._records = {}          # input signature -> _Record, or None after a failed capture
        _LIVE.add(self)
        self._sync_bn = any(isinstance(s, nn.SyncBatchNorm) for m in self.mods for s in m.modules())

    def _qualifies(self, args):
        if not (_ENABLED and args and args[0].is_cuda and torch.is_grad_enabled()) or self._sync_bn:
            return False
        if torch.is_autocast_enabled() or not all(m.training for m in self.mods):
            return False
        return not torch.cuda.is_current_stream_capturing()

    def __call__(self, *args):
        if not self._qualifies(args):
            return self.fn(*args)
        key = tuple((tuple(a.shape), a.dtype, a.requires_grad) for a in args)
        if key not in self._records:
            self._records[key] = self._capture(args) if len(self._records) < _MAX_SHAPES else None
        rec = self._records[key]
        if rec is None:
            return self.fn(*args)
        if rec.pending is not None and rec.pending() is not None:
            # A second forward while an earlier one still waits for its backward (two student passes summed into one
            # loss, gradient checkpointing): a replay would overwrite the saved activations and outputs of the first.
            # This call runs eagerly on memory of its own.  (A forward whose backward never runs stops counting as
            # soon as its autograd graph is released: `pending` is a weak reference to that graph's node.)
            return self.fn(*args)
        return tree_unflatten(list(_Replay.apply(rec, *args, *rec.params)), rec.spec)

    def _capture(self, args):
        # the warm-up passes must leave no trace: BatchNorm running statistics / counters are put back
        buffers = [b for m in self.mods for b in m.buffers()]
        saved = [b.detach().clone() for b in buffers]
        try:
            rec = self._capture_graphs(args)
        except Exception as e:                                  # noqa: BLE001 -- capture is an optimisation only
            warnings.warn('u2mkd_amd.graphs: capture of %r failed (%s: %s); the piece runs eagerly'
                          % (self.name, type(e).__name__, e))
            rec = None
        with torch.no_grad():
            for b, s in zip(buffers, saved):
                b.copy_(s)
        return rec

    def _capture_graphs(self, args):
        rec = _Record()
        rec.sample = tuple(a.detach().clone().requires_grad_(a.requires_grad) for a in args)
        rec.params = tuple(p for m in self.mods for p in m.parameters() if p.requires_grad)
        surface = rec.sample + rec.params
        wrt = tuple(t for t in surface if t.requires_grad)

        def backward(outs, gouts):
            return torch.autograd.grad(tuple(o for o in outs if o.requires_grad), wrt,
                                       tuple(g for g in gouts if g is not None), only_inputs=True, allow_unused=True)

        def grads_like(outs):
            return tuple(torch.zeros_like(o) if o.requires_grad else None for o in outs)

        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):                                  # solver look-ups, workspaces, lazy initialisation
                outs = tree_flatten(self.fn(*rec.sample))[0]
                gins = backward(outs, grads_like(outs))
            del outs, gins
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        rec.pool = torch.cuda.graph_pool_handle()
        rec.fwd = torch.cuda.CUDAGraph()
        # (thread_local: a helper thread may be queueing the teacher on another stream meanwhile)
        with torch.cuda.graph(rec.fwd, pool=rec.pool, capture_error_mode='thread_local'):
            outs = self.fn(*rec.sample)
        outs, rec.spec = tree_flatten(outs)
        rec.outs = tuple(outs)
        rec.gouts = grads_like(rec.outs)
        # Parameter gradients leave the backward graph through buffers OUTSIDE the pool: the pool's memory is shared
        # by both graphs (a forward replay may reuse what held a gradient), and .grad may alias these for as long as
        # the training loop likes.
        keep = [torch.zeros_like(p) for p in rec.params]
        rec.bwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(rec.bwd, pool=rec.pool, capture_error_mode='thread_local'):
            gins = list(backward(rec.outs, rec.gouts))
            first = len(gins) - len(rec.params)
            live = [(k, g) for k, g in zip(keep, gins[first:]) if g is not None]
            if live:
                torch._foreach_copy_([k for k, _ in live], [g for _, g in live])
        gins[first:] = [k if g is not None else None for k, g in zip(keep, gins[first:])]
        gins = iter(gins)
        rec.gins = tuple(next(gins) if t.requires_grad else None for t in surface)
        rec.pending = None
        return rec
