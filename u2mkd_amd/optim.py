"""The optimizer of the shipped configurations -- ``torch.optim.SGD(lr, momentum=0.9, weight_decay=1e-4, nesterov=True)``
(core/builder.py:663-669, configs/nuscenes/default.yaml:18-23) -- with its update over all parameters of a group as ONE HIP
launch (csrc/optim.hip) instead of torch's multi-tensor path (~40 launches behind ~3.5 ms of host-side list handling per KD
step, plus ~2 ms of ``zero_grad``), in a step that is bound by the host.  Same class hierarchy, same ``param_groups`` /
``state`` / ``state_dict`` layout (``momentum_buffer`` per parameter: views of one flat buffer per group), same global step
hooks, and bit-identical parameters after every step (tests/test_gpu_optim.py): the kernel applies torch's five element-wise
operations in torch's order with torch's roundings.  Whatever the fused path does not cover (CPU parameters, dampening,
maximize, sparse or non-fp32 gradients, tensor learning rates) runs torch's own step."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib as L

__all__ = ['FusedSGD']

_ENABLED = os.environ.get('U2MKD_FUSED_SGD', '1') != '0'      # 0: torch's own step (A/B runs)
_RING = 4


class _Group:
    """Static part of one parameter group's job table."""

    def __init__(self, params, chunk):
        self.params = params
        self.ptrs = [p.data_ptr() for p in params]
        dev = params[0].device
        self.device = dev
        numel = [p.numel() for p in params]
        chunks = [(n + chunk - 1) // chunk for n in numel]
        first = np.concatenate([[0], np.cumsum(chunks)]).astype(np.int64)
        self.total_chunks = int(first[-1])
        offs = np.concatenate([[0], np.cumsum([(n + 3) // 4 * 4 for n in numel])]).astype(np.int64)       # (16-byte aligned views)
        self.flat = torch.zeros(int(offs[-1]), dtype=torch.float32, device=dev)
        self.bufs = [self.flat[int(offs[i]):int(offs[i]) + numel[i]].view_as(p) for i, p in enumerate(params)]
        self.has_buf = [False] * len(params)
        self.first_col = np.ones(len(params), dtype=np.int64)      # 1: no momentum buffer yet
        self.missing = len(params)
        n = len(params)
        self.stage = [torch.zeros(n, 6, dtype=torch.int64).pin_memory() for _ in range(_RING)]
        self.tabs = [s.numpy() for s in self.stage]
        for t in self.tabs:
            t[:, 0] = self.ptrs
            t[:, 2] = [b.data_ptr() for b in self.bufs]
            t[:, 3] = numel
            t[:, 4] = first[:-1]
            t[:, 5] = 1
        self.events = [None] * _RING
        self.table = torch.zeros(n, 6, dtype=torch.int64, device=dev)
        self.turn = 0


class FusedSGD(torch.optim.SGD):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, **kw):
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov, **kw)
        self._fused_groups = {}
        self._adopt = True
        self._contract = 1
        self._chunk = None

    # ------------------------------------------------------------------ torch.optim.Optimizer surface
    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._adopt = True               # (the loaded momentum buffers are new tensors: copied into the flat buffers at the next step)

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, '_fused_groups'):
            self._fused_groups.clear()
            self._adopt = True

    def zero_grad(self, set_to_none: bool = True):
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group['params']:
                p.grad = None

    # ------------------------------------------------------------------ the step
    @staticmethod
    def _eligible(group):
        """The group's settings are ones the kernel implements (the parameters themselves are checked when the group's job
        table is built, ``_fused``)."""
        return _ENABLED and not group.get('maximize') and not group.get('differentiable') and group['dampening'] == 0 \
            and isinstance(group['lr'], (float, int)) and len(group['params']) > 0      # (numpy.float64 from a LambdaLR is a float)

    def _fused(self, gi, group):
        """The group's job table, rebuilt when its parameter list or a parameter's storage changed; None if a parameter is not
        a contiguous fp32 tensor on one HIP device."""
        ps = group['params']
        fg = self._fused_groups.get(gi)
        if fg is None or fg.params_list is not ps or len(fg.params) != len(ps) or fg.ptrs != [p.data_ptr() for p in ps]:
            dev = ps[0].device
            if dev.type != 'cuda' or not all(p.dtype == torch.float32 and p.device == dev and p.is_contiguous() for p in ps):
                return None
            if self._chunk is None:
                self._chunk = int(L.load().u2mkd_sgd_chunk_elements())
            fg = self._fused_groups[gi] = _Group(list(ps), self._chunk)
            fg.params_list = ps
            self._adopt = True
        return fg

    def _adopt_state(self, fg):
        """Momentum buffers that exist in ``self.state`` but are not this group's views (a loaded checkpoint, a first step that
        torch's own path ran) are copied into the flat buffer and replaced by the views."""
        for i, p in enumerate(fg.params):
            st = self.state.get(p)
            buf = None if st is None else st.get('momentum_buffer')
            if buf is None:
                fg.has_buf[i] = False
            elif buf.data_ptr() != fg.bufs[i].data_ptr():
                fg.bufs[i].copy_(buf)
                st['momentum_buffer'] = fg.bufs[i]
                fg.has_buf[i] = True
            else:
                fg.has_buf[i] = True
        fg.first_col[:] = [0 if h else 1 for h in fg.has_buf]
        fg.missing = fg.has_buf.count(False)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not all(self._eligible(g) for g in self.param_groups):
            return self._torch_step(loss)
        plans = []
        f32 = torch.float32
        for gi, group in enumerate(self.param_groups):
            fg = self._fused(gi, group)
            if fg is None:
                return self._torch_step(loss)
            grads = [p.grad for p in fg.params]
            if any(g is not None and (g.dtype is not f32 or g.is_sparse or not g.is_contiguous()) for g in grads):
                return self._torch_step(loss)
            plans.append((group, fg, grads))
        if self._adopt:
            for _, fg, _ in plans:
                self._adopt_state(fg)
            self._adopt = False
        for group, fg, grads in plans:
            with torch.cuda.device(fg.device):          # (the launch goes to the current stream of the GROUP's device)
                self._launch(group, fg, grads)
        return loss

    def _launch(self, group, fg, grads):
        slot = fg.turn % _RING
        fg.turn += 1
        ev = fg.events[slot]
        if ev is not None and not ev.query():
            ev.synchronize()              # (the host is a whole ring ahead of the copy that reads this staging buffer)
        tab = fg.tabs[slot]
        tab[:, 1] = [0 if g is None else g.data_ptr() for g in grads]
        tab[:, 5] = fg.first_col
        fg.table.copy_(fg.stage[slot], non_blocking=True)
        if ev is None:
            ev = fg.events[slot] = torch.cuda.Event()
        ev.record()
        L.call('u2mkd_sgd_batch', L.ptr(fg.table), len(fg.params), fg.total_chunks, float(group['lr']), float(group['momentum']),
               float(group['weight_decay']), int(bool(group['nesterov'])), self._contract, L.stream())
        if fg.missing and group['momentum'] != 0:
            # torch: a parameter's momentum buffer comes into being with its first gradient (buf = clone(grad))
            for i, g in enumerate(grads):
                if g is not None and not fg.has_buf[i]:
                    fg.has_buf[i] = True
                    fg.first_col[i] = 0
                    fg.missing -= 1
                    self.state[fg.params[i]]['momentum_buffer'] = fg.bufs[i]

    def _torch_step(self, loss):
        """torch's own update (whatever the fused path does not cover); its buffers are adopted by the next fused step."""
        self._adopt = True
        fn = torch.optim.SGD.step
        if getattr(fn, 'hooked', False):      # (the class-level hook wrapper: this call is already inside FusedSGD's own)
            fn = fn.__wrapped__
        fn(self)
        return loss
