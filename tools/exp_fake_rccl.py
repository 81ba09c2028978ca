"""What a BUSY communication stream does to the step at one rank: the N > 1 code path (U2MKD_FORCE_DDP=1 U2MKD_FORCE_SYNC_BN=1)
with every all_reduce / all_gather replaced by a device pass over the same tensor on a private stream, ordered like an
asynchronous RCCL collective (the stand-in for RCCL's own stream, which at world size 1 never runs a kernel) -- the closest
one GPU gets to the queue pressure of a real multi-rank step.  Settings come from the environment (GPU_MAX_HW_QUEUES,
U2MKD_STAGED_GEOMETRY).      python tools/exp_fake_rccl.py [bench.py arguments]"""
import os, sys
os.environ['U2MKD_FORCE_DDP'] = '1'
os.environ['U2MKD_FORCE_SYNC_BN'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

_comm = []


class _Work:
    def __init__(self, st):
        self.st = st

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.st)
        return True


def _on_comm(fn, t):
    if not t.is_cuda:
        return fn()
    if not _comm:
        _comm.append(torch.cuda.Stream())
    st = _comm[0]
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    t.record_stream(st)
    return _Work(st)


def all_reduce(t, op=None, group=None, async_op=False):
    w = _on_comm(lambda: t.mul_(1.0), t)          # one read + one write of the buffer, as the local part of a ring step
    if w is not None and not async_op and hasattr(w, 'wait'):
        w.wait()
    return w if async_op else None


def all_gather_into_tensor(out, t, group=None, async_op=False):
    w = _on_comm(lambda: out.view(-1)[:t.numel()].copy_(t.view(-1)), out)
    if w is not None and not async_op and hasattr(w, 'wait'):
        w.wait()
    return w if async_op else None


import bench
real_init = dist.init_process_group


def init(*a, **k):
    r = real_init(*a, **k)
    dist.all_reduce = all_reduce
    dist.all_gather_into_tensor = all_gather_into_tensor
    return r


dist.init_process_group = init
sys.argv = ['bench.py', '--no-secondary', '--no-roofline', '--no-cpu-baseline', '--steps', '20', '--warmup', '6'] + sys.argv[1:]
bench.main()
