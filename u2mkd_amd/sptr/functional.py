"""Window attention with contextual relative position tables on the HIP kernels of
csrc/sptr.hip (rows a10-a11 of SURVEY.md §8a)."""
from __future__ import annotations

import numpy as np
import os

import torch
from torch.autograd import Function

from .. import _lib as L

__all__ = ['WindowPlan', 'get_indices_params', 'sparse_self_attention', 'window_attention']


class _Inert:
    """Placeholder for an M-sized index tensor of the reference API; only ever passed back."""

    def int(self):
        return self

    def long(self):
        return self


class WindowPlan:
    """Tokens sorted by window: ``sort_idx`` (token id at sorted position), and per sorted
    position the first position / length of its window.  Built with no host sync."""

    def __init__(self, xyz: torch.Tensor, batch: torch.Tensor, window_size):
        L.require_cuda(xyz, batch)
        xyz = xyz.contiguous().float()
        n = xyz.shape[0]
        dev = xyz.device
        w = [float(v) for v in np.asarray(window_size, dtype=np.float64).reshape(-1)]
        b32 = batch.int().contiguous()
        p4 = torch.cat([xyz, b32.view(-1, 1).float()], 1)
        lo4 = p4.amin(0).contiguous()
        hi4 = p4.amax(0).contiguous()
        keys = torch.empty(n, dtype=torch.int64, device=dev)
        L.call('u2mkd_sptr_window_keys', L.ptr(xyz), L.ptr(b32), n, L.ptr(lo4), L.ptr(hi4), w[0], w[1], w[2],
               L.ptr(keys), L.stream())
        self._finish(n, keys, lo4, w)

    def _finish(self, n, keys, lo4, w):
        dev = keys.device
        skeys, sort_idx = torch.sort(keys, stable=True)
        self.sort_idx = sort_idx.int().contiguous()
        self.wstart = torch.empty(n, dtype=torch.int32, device=dev)
        self.wlen = torch.empty(n, dtype=torch.int32, device=dev)
        L.call('u2mkd_sptr_window_ranges', L.ptr(skeys), n, L.ptr(self.wstart), L.ptr(self.wlen), L.stream())
        self.n = n
        self.window_size = w
        self.lo = lo4
        self._qc = {}

    @classmethod
    def pair(cls, xyz: torch.Tensor, batch: torch.Tensor, window_size, window_size_sphere):
        """(cubic plan, spherical plan, spherical coordinates [n, 3]) of one SphereFormer block: ``cart2sphere`` and both
        ``WindowPlan`` constructions (spherical_transformer.py:31-36, 206-213) from ONE pass over the points
        (u2mkd_sptr_plan_prepare: two launches for the coordinates, the bounds and both key arrays)."""
        L.require_cuda(xyz, batch)
        xyz = xyz.contiguous().float()
        n = xyz.shape[0]
        dev = xyz.device
        wc = [float(v) for v in np.asarray(window_size, dtype=np.float64).reshape(-1)]
        ws = [float(v) for v in np.asarray(window_size_sphere, dtype=np.float64).reshape(-1)]
        if len(wc) == 1:
            wc = wc * 3
        b32 = batch.int().contiguous()
        sphere = torch.empty(n, 3, dtype=torch.float32, device=dev)
        bounds = torch.empty(16, dtype=torch.float32, device=dev)
        keys = torch.empty(2, n, dtype=torch.int64, device=dev)
        ws_bytes = L.load().u2mkd_sptr_plan_prepare_workspace_bytes()
        work = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        L.call('u2mkd_sptr_plan_prepare', L.ptr(xyz), L.ptr(b32), n, wc[0], wc[1], wc[2], ws[0], ws[1], ws[2], L.ptr(sphere),
               L.ptr(bounds), L.ptr(keys[0]), L.ptr(keys[1]), L.ptr(work), L.stream())
        plans = []
        for i, w in enumerate((wc, ws)):
            p = cls.__new__(cls)
            p._finish(n, keys[i], bounds[8 * i:8 * i + 4], w)
            plans.append(p)
        return plans[0], plans[1], sphere

    def int(self):
        return self

    def long(self):
        return self

    def quant_coords(self, xyz: torch.Tensor, quant_size, want_radial: bool):
        """(int32 [n,3] quantised in-window coordinates, radial f32 [n] or None), sorted order."""
        q = tuple(float(v) for v in np.asarray(quant_size, dtype=np.float64).reshape(-1))
        key = (q, want_radial)
        hit = self._qc.get(key)
        if hit is None:
            xyz = xyz.contiguous().float()
            qc = torch.empty(self.n, 3, dtype=torch.int32, device=xyz.device)
            radial = torch.empty(self.n, dtype=torch.float32, device=xyz.device) if want_radial else None
            w = self.window_size
            L.call('u2mkd_sptr_quant_coords', L.ptr(xyz), L.ptr(self.sort_idx), self.n, L.ptr(self.lo), w[0], w[1],
                   w[2], q[0], q[1], q[2], L.ptr(qc), L.ptr(radial), L.stream())
            # host-known bound on the quantised coordinates of the affine axes (x, y and, in the cubic
            # branch, z): qc = floor((v mod w) / q) <= floor(w / q)
            axes = (0, 1) if want_radial else (0, 1, 2)
            span = max(int(np.floor(np.float64(w[d]) / np.float64(q[d]))) + 1 for d in axes)
            hit = (qc, radial, span)
            self._qc[key] = hit
        return hit

    @property
    def n_max(self) -> int:      # only for API compatibility (host sync); the kernels do not need it
        return int(self.wlen.max().item()) if self.n else 0


# U2MKD_SPTR_TILES: which branches run the TILE form of the forward (csrc/sptr_tiles.hip: 16 x 16 score tiles on the matrix pipe,
# relative-position terms by look-up) instead of the one-thread-per-(token, head) kernels of csrc/sptr.hip.  'sphere': the
# spherical branch, whose windows hold tens to hundreds of tokens; 'all': the cubic branch too (windows of 4-9 tokens: the tiles
# are mostly masked); '0' (default): none.  The tile form is verified (tests/test_gpu_sptr.py runs it) and was MEASURED SLOWER in
# round 6 (one wave per 16 queries, strips re-read per tile: 3.5 -> 6.5 ms of forward launches per KD step, NOTES N10.6): it stays
# as the tested starting point of that formulation, off by default.
_TILES = os.environ.get('U2MKD_SPTR_TILES', '0')


def _attention_forward(q, k, v, ld_qkv, q_scale, plan, qc, radial, tq, tk, tv, tl, qgl, split_a, n, h, d, out, ld_out, lse, st):
    """One branch's forward on row-strided operands (q, k, v: views whose first element is the branch's first head)."""
    if _TILES == 'all' or (_TILES == 'sphere' and split_a > 0):
        nbytes = L.load().u2mkd_sptr_tiles_workspace_bytes(n, h)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=out.device)
        L.call('u2mkd_sptr_attention_forward_tiles', L.ptr(q), L.ptr(k), L.ptr(v), ld_qkv, q_scale, L.ptr(plan.sort_idx),
               L.ptr(plan.wstart), L.ptr(plan.wlen), L.ptr(qc), L.ptr(radial), L.ptr(tq), L.ptr(tk), L.ptr(tv), tl, qgl, split_a,
               n, h, d, L.ptr(out), ld_out, L.ptr(lse), L.ptr(ws), nbytes, st)
        return
    L.call('u2mkd_sptr_attention_forward_strided', L.ptr(q), L.ptr(k), L.ptr(v), ld_qkv, q_scale, L.ptr(plan.sort_idx),
           L.ptr(plan.wstart), L.ptr(plan.wlen), L.ptr(qc), L.ptr(radial), L.ptr(tq), L.ptr(tk), L.ptr(tv), tl, qgl, split_a,
           n, h, d, L.ptr(out), ld_out, L.ptr(lse), st)


class WindowAttentionFunction(Function):
    @staticmethod
    def forward(ctx, q, k, v, tq, tk, tv, plan, qc, radial, qgl, split_a, qc_span=0):
        L.require_cuda(q, k, v, tq, tk, tv)
        q, k, v = (t.contiguous().float() for t in (q, k, v))
        tq, tk, tv = (t.contiguous().float() for t in (tq, tk, tv))
        n, h, d = q.shape
        tl = tq.shape[0]
        if h == 0 or n == 0:
            ctx.empty = True
            ctx.shapes = (q.shape, tq.shape)
            return torch.zeros_like(q)
        ctx.empty = False
        if tq.shape != (tl, 3, h, d) or tk.shape != tq.shape or tv.shape != tq.shape:
            raise RuntimeError(f'relative position tables must be [L,3,{h},{d}], got {tuple(tq.shape)}')
        out = torch.empty_like(q)
        lse = torch.empty(n, h, dtype=torch.float32, device=q.device)
        _attention_forward(q, k, v, h * d, 1.0, plan, qc, radial, tq, tk, tv, tl, int(qgl), float(split_a), n, h, d, out, h * d, lse,
                           L.stream())
        ctx.save_for_backward(q, k, v, out, lse, tq, tk, tv, qc, radial if radial is not None else q.new_empty(0))
        ctx.plan, ctx.qgl, ctx.split_a, ctx.has_radial = plan, int(qgl), float(split_a), radial is not None
        ctx.qc_span = int(qc_span)
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.empty:
            qs, ts = ctx.shapes
            z = dout.new_zeros(qs)
            t = dout.new_zeros(ts)
            return z, z, z, t, t, t, None, None, None, None, None, None
        q, k, v, out, lse, tq, tk, tv, qc, radial = ctx.saved_tensors
        plan = ctx.plan
        dout = dout.contiguous().float()
        n, h, d = q.shape
        tl = tq.shape[0]
        delta = torch.empty(n, h, dtype=torch.float32, device=q.device)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dtq, dtk, dtv = torch.empty_like(tq), torch.empty_like(tk), torch.empty_like(tv)
        nbytes = L.load().u2mkd_sptr_backward_workspace_bytes(n, h, tl)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=q.device)
        L.call('u2mkd_sptr_attention_backward', L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(dout), L.ptr(lse),
               L.ptr(plan.sort_idx), L.ptr(plan.wstart), L.ptr(plan.wlen), L.ptr(qc),
               L.ptr(radial) if ctx.has_radial else None, L.ptr(tq), L.ptr(tk), L.ptr(tv), tl, ctx.qgl, ctx.split_a,
               ctx.qc_span, n, h, d, L.ptr(delta), L.ptr(ws), nbytes, L.ptr(dq), L.ptr(dk), L.ptr(dv), L.ptr(dtq), L.ptr(dtk), L.ptr(dtv),
               L.stream())
        return dq, dk, dv, dtq, dtk, dtv, None, None, None, None, None, None


class PackedAttentionFunction(Function):
    """All branches of a SphereFormer attention layer on the PACKED projection output: ``qkv`` [N, 3, H, 16] (the
    reshaped output of the qkv Linear), ``scale`` applied to q inside the kernels, branch b = heads h0..h0+h with its
    own window plan and tables; returns [N, H * 16] with every branch's heads in their columns.  Equal, bit for bit,
    to window_attention on (qkv[:, 0] * scale)[:, h0:h0+h], qkv[:, 1][:, h0:h0+h], qkv[:, 2][:, h0:h0+h] per branch
    + torch.cat -- without the scale / slice / concatenate copies and, in the backward, without the zero-filled
    [N, 3, H, 16] gradient of every slice: both kernels read and write the packed layouts through row strides
    (u2mkd_sptr_attention_forward_strided / _backward_strided)."""

    @staticmethod
    def forward(ctx, qkv, scale, branches, *tables):
        L.require_cuda(qkv, *tables)
        qkv = qkv.contiguous().float()
        n, three, H, d = qkv.shape
        assert three == 3 and d == 16 and len(tables) == 3 * len(branches)
        given = tables
        tables = tuple(t.contiguous().float() for t in tables)
        # tables that ARE leaf parameters (no contiguous / cast copy in between): nobody reads their gradients before the
        # backward ends, so the slab sum that finishes them may leave the backward's chain (deferred.py)
        ctx.table_leaves = tuple(g if (g is t and g.is_leaf and g.requires_grad) else None for g, t in zip(given, tables))
        from .. import deferred
        ctx.overlap_ok = deferred.overlap_ok()
        out = torch.empty(n, H, d, dtype=torch.float32, device=qkv.device)
        lses = []
        st = L.stream()
        for b, br in enumerate(branches):
            h0, h = br['h0'], br['h']
            tq, tk, tv = tables[3 * b:3 * b + 3]
            if tq.shape != (tq.shape[0], 3, h, d) or tk.shape != tq.shape or tv.shape != tq.shape:
                raise RuntimeError(f'relative position tables must be [L,3,{h},{d}], got {tuple(tq.shape)}')
            lse = torch.empty(n, h, dtype=torch.float32, device=qkv.device)
            lses.append(lse)
            if n == 0 or h == 0:
                continue
            plan = br['plan']
            _attention_forward(qkv[:, 0, h0:], qkv[:, 1, h0:], qkv[:, 2, h0:], 3 * H * d, float(scale), plan, br['qc'], br['radial'],
                               tq, tk, tv, tq.shape[0], int(br['qgl']), float(br['split_a']), n, h, d, out[:, h0:], H * d, lse, st)
        ctx.save_for_backward(qkv, out, *lses, *tables)
        ctx.branches, ctx.scale = branches, float(scale)
        return out.view(n, H * d)

    @staticmethod
    def backward(ctx, dout):
        from .. import deferred
        branches = ctx.branches
        nb = len(branches)
        saved = ctx.saved_tensors
        qkv, out, lses, tables = saved[0], saved[1], saved[2:2 + nb], saved[2 + nb:]
        n, _, H, d = qkv.shape
        dout = dout.contiguous().float()
        dqkv = torch.empty_like(qkv)            # every (q | k | v, head) column belongs to exactly one branch
        grads = []
        st = L.stream()
        for b, br in enumerate(branches):
            h0, h = br['h0'], br['h']
            tq, tk, tv = tables[3 * b:3 * b + 3]
            dtq, dtk, dtv = torch.empty_like(tq), torch.empty_like(tk), torch.empty_like(tv)
            grads += [dtq, dtk, dtv]
            if n == 0 or h == 0:
                for t in (dtq, dtk, dtv):
                    t.zero_()
                continue
            plan = br['plan']
            tl = tq.shape[0]
            delta = torch.empty(n, h, dtype=torch.float32, device=qkv.device)
            nbytes = L.load().u2mkd_sptr_backward_workspace_bytes(n, h, tl)
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=qkv.device)
            side = _table_side(ctx, ctx.table_leaves[3 * b:3 * b + 3], qkv.device)
            late = side is not None
            L.call('u2mkd_sptr_attention_backward_strided', L.ptr(qkv[:, 0, h0:]), L.ptr(qkv[:, 1, h0:]), L.ptr(qkv[:, 2, h0:]),
                   3 * H * d, ctx.scale, L.ptr(out[:, h0:]), L.ptr(dout[:, h0 * d:]), H * d, L.ptr(lses[b]),
                   L.ptr(plan.sort_idx), L.ptr(plan.wstart), L.ptr(plan.wlen), L.ptr(br['qc']), L.ptr(br['radial']),
                   L.ptr(tq), L.ptr(tk), L.ptr(tv), tl, int(br['qgl']), float(br['split_a']), int(br['span']), n, h, d,
                   L.ptr(delta), L.ptr(ws), nbytes, L.ptr(dqkv[:, 0, h0:]), L.ptr(dqkv[:, 1, h0:]), L.ptr(dqkv[:, 2, h0:]),
                   3 * H * d, L.ptr(None if late else dtq), L.ptr(None if late else dtk), L.ptr(None if late else dtv), st)
            if late:       # the slab sum behind the kernels, on the weight-gradient side stream, joined when the backward ends
                deferred.order_behind_current(side, qkv.device.index if qkv.device.index is not None else torch.cuda.current_device())
                for t in (ws, dtq, dtk, dtv):
                    t.record_stream(side)
                L.call('u2mkd_sptr_table_reduce', L.ptr(ws), n, h, tl, float(br['split_a']), L.ptr(dtq), L.ptr(dtk), L.ptr(dtv),
                       side.cuda_stream)
        if n == 0:
            dqkv.zero_()
        return (dqkv, None, None, *grads)


_DEFER_TABLES = os.environ.get('U2MKD_DEFER_SPTR_TABLES', '1') != '0'


def _table_side(ctx, leaves, device):
    """The side stream for a branch's table-gradient sum, or None: all three tables are leaf parameters without a gradient
    yet, none of them has had a contribution in this backward pass (a shared table: autograd adds the second one as soon as
    the function returns), a trainer that follows the protocol is active and the step may fork (deferred.py)."""
    from .. import deferred
    if not (_DEFER_TABLES and deferred.enabled() and getattr(ctx, 'overlap_ok', False) and deferred.overlap_ok()):
        return None
    if any(t is None or t.grad is not None for t in leaves):
        return None
    if any(deferred.owned(id(t)) for t in leaves):
        deferred.join()
        return None
    side = deferred.side_for('sparse_wgrad', device, owner=id(leaves[0]))
    for t in leaves[1:]:
        deferred.OWNERS.add(id(t))
    return side


def packed_window_attention(qkv, scale, branches):
    """``branches``: [(h0, h, xyz, plan, quant_size, quant_grid_length, (table_q, table_k, table_v), split_a or None), ..]
    covering heads 0..H of ``qkv`` [N, 3, H, 16]; returns the concatenated attention output [N, H * 16]."""
    descs, tables = [], []
    for h0, h, xyz, plan, quant_size, qgl, tabs, split_a in branches:
        sphere = split_a is not None
        qc, radial, span = plan.quant_coords(xyz, quant_size, sphere)
        descs.append({'h0': int(h0), 'h': int(h), 'plan': plan, 'qc': qc, 'radial': radial, 'qgl': int(qgl),
                      'split_a': float(split_a) if sphere else 0.0, 'span': int(span)})
        tables += list(tabs)
    return PackedAttentionFunction.apply(qkv, float(scale), descs, *tables)


def window_attention(q, k, v, xyz, plan: WindowPlan, quant_size, quant_grid_length, table_q, table_k, table_v,
                     split_a=None):
    """softmax over each token's window of (q.k + q.Tq(rel) + k.Tk(rel)) applied to (v + Tv(rel));
    q,k,v [N,h,16] (q pre-scaled), tables [L,3,h,16]; ``split_a`` selects the spherical branch."""
    sphere = split_a is not None
    qc, radial, span = plan.quant_coords(xyz, quant_size, sphere)
    return WindowAttentionFunction.apply(q, k, v, table_q, table_k, table_v, plan, qc, radial,
                                         int(quant_grid_length), float(split_a) if sphere else 0.0, span)


def get_indices_params(xyz, batch, window_size, shift_win: bool):
    """sptr/utils.py:49-78 signature.  Returns (plan, inert, n_max, inert, inert, sort_idx)."""
    if shift_win:
        raise NotImplementedError('shift_win=True is never used by U2MKD (spherical_transformer.py:79)')
    plan = WindowPlan(xyz, batch, window_size)
    return plan, _Inert(), None, _Inert(), _Inert(), plan.sort_idx


def sparse_self_attention(query, key, value, xyz, index_0, index_0_offsets, n_max, index_1, index_1_offsets, sort_idx,
                          window_size, shift_win, pe_type='none', rel_query=False, rel_key=False, rel_value=False,
                          quant_size=None, quant_grid_length=None, relative_pos_query_table=None,
                          relative_pos_key_table=None, relative_pos_value_table=None, split_func=None):
    """sptr/modules.py:11-66 signature; ``index_0`` must be the WindowPlan of get_indices_params."""
    if not (pe_type == 'contextual' and rel_query and rel_key and rel_value):
        raise NotImplementedError('only pe_type="contextual" with rel_query=rel_key=rel_value=True is on the '
                                  'U2MKD path (core/models/nuscenes/spvcnn_spformer.py:70-72)')
    if not isinstance(index_0, WindowPlan):
        raise TypeError('index_0 must come from u2mkd_amd.sptr.get_indices_params')
    split_a = None
    if split_func is not None:
        split_a = getattr(split_func, 'keywords', {}).get('a', 0.05 * 0.25)
    return window_attention(query, key, value, xyz, index_0, quant_size, quant_grid_length, relative_pos_query_table,
                            relative_pos_key_table, relative_pos_value_table, split_a)
