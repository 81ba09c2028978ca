"""Gradient work that is joined at the END of the backward pass instead of where it was issued.

A leaf's gradient is read by nobody before the backward ends (the optimizer, or a gradient hook), so the kernels that compute
it may run on a side stream next to the rest of the backward.  ``side_for(role, device)`` hands out the stream ordered behind
the caller's stream and books the join: an end-of-backward callback makes the stream that called ``backward()`` wait for it, so
``.grad`` is ordinary data once ``backward()`` returns; a hook that reads ``.grad`` DURING the backward (distributed.
BucketedGradientAverage) calls ``join()`` itself."""
import torch

# Off until a trainer that knows the protocol switches it on (train.LidarStep / train.KDStep do): the drop-in operators used under
# a foreign trainer -- the reference's own, with torch's DistributedDataParallel, whose reducer copies gradients inside autograd
# hooks and knows nothing of these streams -- keep their gradient launches joined where they are issued.
_ENABLED = [False]


def enable(flag: bool = True):
    """Allow (or forbid) gradient launches to be joined at the end of the backward.  A caller that enables it promises that
    whatever reads ``.grad`` during the backward calls ``join()`` first."""
    _ENABLED[0] = bool(flag)


def enabled() -> bool:
    return _ENABLED[0]


_STREAMS = {}
_PENDING = {}
OWNERS = set()        # ids of the leaves whose gradient is still running on a side stream


# The GPU runs five streams of this package side by side without trouble (main, teacher, camera, sparse weight gradients,
# geometry); a SIXTH busy stream pushes the hardware-queue scheduler over a cliff (KD step 70 -> 122 ms, measured; round 2 met
# the same ~120 ms cliff with a priority stream).  Deferred gradient work therefore borrows a stream that is idle during the
# backward: the frozen teacher's (kd._side_stream), which only works during the forward.
_ALIAS = {'camera_wgrad': 'teacher'}


def stream(device_index: int, role: str) -> torch.cuda.Stream:
    """One side HIP stream per (device, role); roles in _ALIAS share another role's stream."""
    key = (device_index, _ALIAS.get(role, role))
    s = _STREAMS.get(key)
    if s is None:
        s = _STREAMS[key] = torch.cuda.Stream(device=torch.device('cuda', device_index))
    return s


def side_for(role: str, device: torch.device, owner=None) -> torch.cuda.Stream:
    """The side stream of ``role`` on ``device``, waiting for everything queued on the current stream, its join booked.
    Only callable from inside a backward pass (the callback belongs to the running autograd engine)."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    side = stream(index, role)
    key = (index, _ALIAS.get(role, role))
    side.wait_stream(torch.cuda.current_stream(device))
    if key not in _PENDING:
        torch.autograd.Variable._execution_engine.queue_callback(join)
    _PENDING[key] = side
    if owner is not None:
        OWNERS.add(owner)
    return side


def pending():
    """The side streams with booked, not yet joined work (a caller may queue more work behind them there)."""
    return list(_PENDING.values())


def join():
    """The current stream waits for every side stream with booked work."""
    for key, side in list(_PENDING.items()):
        torch.cuda.current_stream(side.device).wait_stream(side)
        del _PENDING[key]
    OWNERS.clear()
