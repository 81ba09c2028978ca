"""Gradient work that is joined at the END of the backward pass instead of where it was issued.

A leaf's gradient is read by nobody before the backward ends (the optimizer, or a gradient hook), so the kernels that compute
it may run on a side stream next to the rest of the backward.  ``side_for(role, device)`` hands out the stream ordered behind
the caller's stream and books the join: an end-of-backward callback makes the stream that called ``backward()`` wait for it, so
``.grad`` is ordinary data once ``backward()`` returns; code that reads ``.grad`` DURING the backward (a gradient hook, the
accumulation of a second contribution to the same leaf) calls ``join()`` first, which orders the CURRENT stream behind the side
streams and leaves the end-of-backward join in place for the stream that called ``backward()``.

Bookkeeping is per backward pass (the autograd engine's graph-task id): a pass that raised -- out of memory, a NaN check --
never runs its final callbacks, and the next pass must book its own join instead of trusting a stale one; whatever the failed
pass left on the side streams is joined first."""
import contextlib

import torch

# Off until a trainer that knows the protocol switches it on (train.LidarStep / train.KDStep hold it for their lifetime:
# acquire() / release()): the drop-in operators used under a foreign trainer -- the reference's own, with torch's
# DistributedDataParallel, whose reducer copies gradients inside autograd hooks and knows nothing of these streams -- keep their
# gradient launches joined where they are issued.
_ENABLED = [False]
_HOLDERS = [0]


def enable(flag: bool = True):
    """Allow (or forbid) gradient launches to be joined at the end of the backward.  A caller that enables it promises that
    whatever reads ``.grad`` during the backward calls ``join()`` first."""
    _ENABLED[0] = bool(flag)


def enabled() -> bool:
    return _ENABLED[0]


def acquire():
    """A trainer's hold on the switch (train.LidarStep / KDStep constructors); ``release()`` when the trainer goes away: the
    switch is on exactly while a trainer that follows the protocol exists in the process."""
    _HOLDERS[0] += 1
    _ENABLED[0] = True


def release():
    _HOLDERS[0] = max(0, _HOLDERS[0] - 1)
    if _HOLDERS[0] == 0:
        _ENABLED[0] = False


@contextlib.contextmanager
def scope(flag: bool = True):
    """``with deferred.scope():`` -- the switch for the duration of a block (tests, a caller that drives forward + backward
    itself); leftovers of a pass that raised inside are joined on the way out."""
    old = _ENABLED[0]
    _ENABLED[0] = bool(flag)
    try:
        yield
    finally:
        _ENABLED[0] = old
        begin_step()


# ---- may this package's streams run side by side at all? ---------------------------------------------------------------------
# On MI355X a kernel that issues gfx950's double-K matrix instructions (v_mfma_f32_16x16x32_bf16 / _f16, v_mfma_f32_32x32x16_bf16)
# makes waves of OTHER streams' kernels compute wrong results (NOTES N9; tools/repro_concurrent_kernels.hip).  This package's own
# kernels use the gfx942 forms and the libraries' fp32 kernels are clean, but what torch dispatches under REDUCED-PRECISION
# autocast is not: MIOpen's bf16 convolution backward and the rocBLAS / hipBLASLt bf16 products change a concurrent
# u2mkd_ti_weights' results in 6-16 % of the rounds (tools/dbg_library_aggressors.py).  Those kernels are not ours to change, so
# while a step runs under bf16 / fp16 autocast every fork of this package (teacher stream, camera stream, deferred weight
# gradients, geometry pre-pass) stays on the caller's stream: nothing runs next to a library kernel.  The trainers announce the
# mode for the backward pass, which runs outside the autocast context (``set_reduced_precision``).
# U2MKD_OVERLAP_UNDER_AUTOCAST=1 keeps the forks (A/B runs: the price of the rule).
import os as _os

_REDUCED = [False]
_FORCE_OVERLAP = _os.environ.get('U2MKD_OVERLAP_UNDER_AUTOCAST') == '1'


def set_reduced_precision(flag: bool):
    """Trainers: the current step's library kernels run in bf16 / fp16 (autocast), forward AND backward."""
    _REDUCED[0] = bool(flag)


def overlap_ok() -> bool:
    """True while this package may put work on its side streams (see above)."""
    if _FORCE_OVERLAP:
        return True
    if _REDUCED[0]:
        return False
    return not (torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') in (torch.bfloat16, torch.float16))


_STREAMS = {}
_PENDING = {}         # (device, role) -> side stream with work booked in the pass `_TASK`
_TASK = [None]        # graph-task id of the backward pass whose end-of-backward join is booked
OWNERS = set()        # ids of the leaves whose gradient was issued on a side stream in the pass `_TASK`


# The GPU runs five streams of this package side by side without trouble (main, teacher, camera, sparse weight gradients,
# geometry); a SIXTH busy stream pushes the hardware-queue scheduler over a cliff (KD step 70 -> 122 ms, measured; round 2 met
# the same ~120 ms cliff with a priority stream).  Deferred gradient work therefore borrows a stream that is idle during the
# backward: the frozen teacher's (kd._side_stream), which only works during the forward.
_ALIAS = {'camera_wgrad': 'teacher'}
# U2MKD_GEO_STREAM: which stream carries the next batch's geometry slices.  'own' (rounds 3-5): a fifth stream -- on the
# runtime's four hardware queues it SHARES a queue with the main stream, whose barrier packets (the joins with the camera
# stream at every fusion point) then hold the slices' kernels back: the host's first size read waited ~14 ms per step for a
# slice queued 30 ms earlier (NOTES N10).  'sparse_wgrad' / 'teacher' / 'camera' put the slices on that role's stream instead
# (four streams, four queues, nobody shares): the wait disappears from the read and the same time reappears spread over
# every launch call of the step -- the step is bound by the GPU, the host is merely held wherever it runs ahead (measured: 64.4
# vs 64.2-64.5 ms in same-box pairs) -- so the default stays 'own'.
_GEO = _os.environ.get('U2MKD_GEO_STREAM', 'own')
if _GEO != 'own':
    _ALIAS['geo'] = _GEO


def stream(device_index: int, role: str) -> torch.cuda.Stream:
    """One side HIP stream per (device, role); roles in _ALIAS share another role's stream."""
    key = (device_index, _ALIAS.get(role, role))
    s = _STREAMS.get(key)
    if s is None:
        s = _STREAMS[key] = torch.cuda.Stream(device=torch.device('cuda', device_index), priority=_PRIORITY.get(key[1], 0))
    return s


# U2MKD_STREAM_PRIORITY="camera=-1,teacher=0": HIP stream priority per role (-1 = high; default 0 everywhere).  A/B runs.
_PRIORITY = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in _os.environ.get('U2MKD_STREAM_PRIORITY', '').split(',') if '=' in kv}


_FAST_WAIT = _os.environ.get('U2MKD_FAST_STREAM_WAIT', '1') != '0'


def order_behind_current(side: torch.cuda.Stream, device_index: int):
    """``side.wait_stream(current stream)`` -- ~130 times per backward pass, so through ONE C-ABI call
    (u2mkd_stream_wait_stream: an event record + a stream wait on an event kept per stream) instead of torch's three Python calls
    and a new event object (17 -> 7 us each on the host, which bounds the step).  U2MKD_FAST_STREAM_WAIT=0: torch's."""
    if _FAST_WAIT:
        from . import _lib as L
        L.call('u2mkd_stream_wait_stream', side.cuda_stream, torch._C._cuda_getCurrentRawStream(device_index))
    else:
        side.wait_stream(torch.cuda.current_stream(torch.device('cuda', device_index)))


def _graph_task():
    """Identity of the running backward pass (-1 outside one)."""
    return torch._C._current_graph_task_id()


def _drop_stale():
    """Work booked by a pass other than the running one (that pass raised: its callback never ran): the current stream waits
    for it, then the books are cleared."""
    for side in _PENDING.values():
        torch.cuda.current_stream(side.device).wait_stream(side)
    _PENDING.clear()
    OWNERS.clear()
    _TASK[0] = None


def begin_step():
    """Called by the trainers at the start of every step (and by ``scope``): nothing of an earlier pass stays booked."""
    if _PENDING or OWNERS or _TASK[0] is not None:
        _drop_stale()


def side_for(role: str, device: torch.device, owner=None) -> torch.cuda.Stream:
    """The side stream of ``role`` on ``device``, waiting for everything queued on the current stream, its join booked.
    Only callable from inside a backward pass (the callback belongs to the running autograd engine)."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    task = _graph_task()
    if _TASK[0] != task:
        _drop_stale()
        _TASK[0] = task
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
    side = stream(index, role)
    key = (index, _ALIAS.get(role, role))
    order_behind_current(side, index)
    _PENDING[key] = side
    if owner is not None:
        OWNERS.add(owner)
    return side


def owned(owner) -> bool:
    """True if ``owner``'s gradient was issued on a side stream in the RUNNING backward pass: a second contribution to the
    same leaf must ``join()`` and be computed in line (autograd adds the two as soon as the second function returns)."""
    return owner in OWNERS and _TASK[0] == _graph_task()


def pending():
    """The side streams with booked, not yet joined work (a caller may queue more work behind them there)."""
    return list(_PENDING.values())


def join():
    """The CURRENT stream waits for every side stream with booked work.  The books stay: the end-of-backward callback still
    joins them into the stream that called ``backward()`` (a hook may run on any of the backward's streams)."""
    for side in _PENDING.values():
        torch.cuda.current_stream(side.device).wait_stream(side)


def _end_of_backward():
    """Engine callback, on the thread and stream that called ``backward()``."""
    join()
    _PENDING.clear()
    OWNERS.clear()
    _TASK[0] = None
