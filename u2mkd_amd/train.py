"""Training-step drivers of the hot path (row a19 + the train branches of the reference's
trainers): LiDAR-only (core/spformer_trainer.py:58-94) and KD
(core/nusc_trainers.py:255-366).  One call = forward, losses, zero_grad, backward (DDP
all-reduce overlapped), SGD-nesterov step, LR-scheduler step; no ``.item()`` host syncs."""
import os
import weakref

import numpy as np
import torch

from . import deferred
from . import distributed as D
from . import kd as KD
from . import torchsparse as ts
from .losses import MixLovaszCrossEntropy
from .optim import FusedSGD

__all__ = ['cosine_schedule_with_warmup', 'make_optimizer', 'LidarStep', 'KDStep', 'TeacherWatch', 'kd_batch_to_device', 'pin_kd_batch', 'fresh_batch', 'state_dict',
           'load_state_dict', 'load_weights']


def cosine_schedule_with_warmup(k, num_epochs, batch_size, dataset_size, world):
    """core/schedulers.py:10-35: the batch is scaled by the world size, warm-up lasts
    1000 // world iterations and only exists for world > 1."""
    batch_size *= world
    warmup_iters = 0 if world == 1 else 1000 // world
    if k < warmup_iters:
        return (k + 1) / warmup_iters
    iter_per_epoch = (dataset_size + batch_size - 1) // batch_size
    return 0.5 * (1 + np.cos(np.pi * (k - warmup_iters) / (num_epochs * iter_per_epoch)))


def make_optimizer(params, lr=0.24, momentum=0.9, weight_decay=1.0e-4, name='sgd', nesterov=True, transformer_lr_scale=0.1):
    """``make_optimizer`` of core/builder.py:662-718 with the values of configs/nuscenes/default.yaml:18-23 as defaults
    (`sgd`: SGD, nesterov -- what every shipped configuration trains with).  ``params``: an iterable of parameters, or,
    for the `*_spformer` variants (SphereFormer blocks at a reduced learning rate: 0.1 x for `sgd_spformer`,
    ``transformer_lr_scale`` for `adamw_spformer`), the module itself -- the groups are split by parameter name
    ("transformer_block" in the name) exactly as the reference does."""
    if name in ('sgd_spformer', 'adamw_spformer'):
        if not isinstance(params, torch.nn.Module):
            raise TypeError(f'make_optimizer({name!r}) splits the parameters by name: pass the module')
        named = [(n, p) for n, p in params.named_parameters() if p.requires_grad]
        rest = [p for n, p in named if 'transformer_block' not in n]
        blocks = [p for n, p in named if 'transformer_block' in n]
        if name == 'sgd_spformer':
            common = dict(momentum=momentum, weight_decay=weight_decay, nesterov=nesterov)
            return FusedSGD([dict(params=rest, lr=lr, **common), dict(params=blocks, lr=lr * 0.1, **common)],
                            lr=lr, **common)
        return torch.optim.AdamW([dict(params=rest, lr=lr, weight_decay=weight_decay),
                                  dict(params=blocks, lr=lr * transformer_lr_scale, weight_decay=weight_decay)],
                                 lr=lr, weight_decay=weight_decay)
    if isinstance(params, torch.nn.Module):
        params = params.parameters()
    if name == 'sgd':
        return FusedSGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov)
    if name == 'adam':
        return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay)
    if name == 'adamw':
        return torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay)
    raise NotImplementedError(name)


def _advance_geometry(amp, geo, staged):
    """One slice of the next batch's geometry (a generator: kd.TSDFull.prepare_staged / point_voxel.prepare_geometry_staged) on
    the geometry stream, under the contexts the one-piece preparation runs in; returns the generator's result when it is
    done, else None."""
    with torch.cuda.stream(geo), amp.autocast(), torch.no_grad():
        try:
            next(staged)
        except StopIteration as done:
            return done.value
    return None


def _staged_geometry() -> bool:
    """The next batch's geometry in slices between the phases of the current step (KDStep.__call__) -- or in one piece behind
    the backward.  U2MKD_STAGED_GEOMETRY=0 / 1 decides; otherwise slices exactly when the process runs on the runtime's 4
    hardware queues (a single-rank process, distributed.configure_runtime): with more queues the slices' stream gets a queue
    of its own and the step falls off a scheduling cliff (64 -> 105 ms, NOTES N9.11)."""
    env = os.environ.get('U2MKD_STAGED_GEOMETRY')
    if env in ('0', '1'):
        return env == '1'
    return D.hardware_queues() <= 4

class _Amp:
    """``amp.autocast(enabled)`` + ``amp.GradScaler(enabled)`` of the reference trainers
    (core/nusc_trainers.py:157-158,285,362-364; core/spformer_trainer.py likewise).  amp = False / None: fp32
    (the parity path); 'fp16': the reference's mode (autocast to half + loss scaling); 'bf16': autocast to
    bfloat16, no loss scaling needed (fp32 exponent range) -- BASELINE.json configs[4].  The sparse operators
    of this package take fp32 rows (like torchsparse's fp32-only CPU backend, they up-cast their inputs), so
    under autocast the dense torch.nn layers -- the SwiftNet-18 camera branch above all -- run reduced."""

    def __init__(self, amp):
        if amp is True:
            amp = 'fp16'
        if amp not in (None, False, 'fp16', 'bf16'):
            raise ValueError(f'amp must be False, "fp16" or "bf16", got {amp!r}')
        self.dtype = {'fp16': torch.float16, 'bf16': torch.bfloat16}.get(amp)
        self.enabled = self.dtype is not None
        self.scaler = torch.amp.GradScaler('cuda', enabled=(amp == 'fp16'))

    def autocast(self):
        return torch.autocast('cuda', dtype=self.dtype or torch.float16, enabled=self.enabled)

    def backward_and_step(self, loss, opt):
        opt.zero_grad()
        self.scaler.scale(loss).backward()
        self.scaler.step(opt)
        self.scaler.update()


def _scheduler(opt, num_epochs, batch_size, dataset_size=28130):
    w = D.world()
    return torch.optim.lr_scheduler.LambdaLR(
        opt, lambda k: cosine_schedule_with_warmup(k, num_epochs, batch_size, dataset_size, w))


class LidarStep:
    """Teacher / LiDAR-only training step."""

    def __init__(self, model, num_epochs=25, batch_size=1, ignore_index=0, amp=False, criterion=None, optimizer=None,
                 scheduler=None):
        """``criterion`` / ``optimizer`` / ``scheduler``: what the reference's trainer receives from core/builder.py
        (train_spformer.py:84-95) -- the criterion as an object, the other two as callables of (module) and (optimizer)
        because the module they act on is the DDP-wrapped one built here; None = the shipped configuration's (Lovasz +
        CE, SGD nesterov 0.24, cosine_warmup)."""
        self.model = model
        self.amp = _Amp(amp)
        deferred.acquire()         # held for this trainer's lifetime: its reducer (distributed.BucketedGradientAverage) follows the protocol
        weakref.finalize(self, deferred.release)
        self.net = D.wrap_model(model, sync_bn=True)
        self.criterion = criterion if criterion is not None else MixLovaszCrossEntropy(ignore_index=ignore_index)
        self.opt = optimizer(self.net) if optimizer is not None else \
            make_optimizer([p for p in self.net.parameters() if p.requires_grad])
        self.sched = scheduler(self.opt) if scheduler is not None else _scheduler(self.opt, num_epochs, batch_size)

    def __call__(self, feats, coords, targets, keyframe_mask=None, prefetch=None):
        """One training step.  ``prefetch`` = (feats, coords) of the NEXT batch (the tensors the next call will receive):
        its geometry -- voxel set and kernel maps, every host synchronisation of a step (point_voxel.prepare_geometry) --
        is built between this step's forward and backward, and the next call issues its forward without waiting for the
        GPU (as KDStep does)."""
        deferred.begin_step()      # (nothing of a backward pass that raised stays booked)
        deferred.set_reduced_precision(self.amp.enabled)      # (library kernels in bf16 / fp16: no side streams, deferred.overlap_ok)
        queued = self.__dict__.pop('_queued', None)
        if getattr(self, '_geo_done', None) is not None:
            # whichever branch this call takes (prefetch=None on an epoch's last batch, staging switched off), the forward
            # reads kernel maps the previous call built on the geometry stream: order the main stream behind them first
            torch.cuda.current_stream().wait_event(self._geo_done)
            self._geo_done = None
        in_mod = {'lidar': ts.SparseTensor(feats, coords)}
        if queued is not None and queued[0] is feats and queued[1] is coords:
            in_mod = queued[2]
        staged = None
        if prefetch is not None and feats.is_cuda and deferred.overlap_ok() and _staged_geometry():
            # the next batch's geometry in slices on a side stream, one in front of each phase of this step (as KDStep does,
            # see there): its two size reads find their counts ready instead of waiting behind the forward's kernels
            from .lidar.point_voxel import prepare_geometry_staged
            main = torch.cuda.current_stream()
            entry = main.record_event()
            geo = KD._side_stream(feats, 'geo')
            geo.wait_event(entry)
            nf, nc = prefetch
            nxt = {'lidar': ts.SparseTensor(nf, nc)}
            staged = prepare_geometry_staged([(nxt['lidar'], self.model.pres, self.model.vres)])
            _advance_geometry(self.amp, geo, staged)
        with self.amp.autocast():
            out = self.net(in_mod)['x_vox']
            if keyframe_mask is not None:
                out, targets = out[keyframe_mask], targets[keyframe_mask]
            loss = self.criterion(out, targets)
        if staged is not None:
            _advance_geometry(self.amp, geo, staged)
            self.amp.backward_and_step(loss, self.opt)
            self.sched.step()
            nxt['_geometry'] = _advance_geometry(self.amp, geo, staged)[0]
            self._queued = (nf, nc, nxt)
            self._geo_done = geo.record_event()
            self._geo_keep = in_mod              # (this batch's geometry lives until the next call has queued its forward)
            for t in _tensors_of(nxt['_geometry']):
                t.record_stream(main)            # allocated on the side stream, read (and freed) on the caller's
            return loss.detach()
        if prefetch is not None:
            from .lidar.point_voxel import prepare_geometry
            nf, nc = prefetch
            nxt = {'lidar': ts.SparseTensor(nf, nc)}
            with self.amp.autocast(), torch.no_grad():
                nxt['_geometry'] = prepare_geometry(nxt['lidar'], self.model.pres, self.model.vres)
            self._queued = (nf, nc, nxt)
        self.amp.backward_and_step(loss, self.opt)
        self.sched.step()
        return loss.detach()

    @torch.no_grad()
    def evaluate(self, feats, coords, inverse_map, inverse_batch, targets_mapped, keyframe_mask_full=None):
        """The branch of ``_run_step`` taken when the model is not training (core/spformer_trainer.py:95-117): voxel
        logits -> per-point predictions through ``inverse_map`` (index of every raw point's voxel within its scene;
        ``inverse_batch`` = its scene), concatenated scene by scene; multi-sweep inputs keep the key-frame points.
        Returns the reference's dictionary (``outputs_vox``, ``targets``) for evaluate.MeanIoU, on the device."""
        from .evaluate import voxel_logits_to_point_predictions
        assert not self.model.training, 'evaluate() is the eval branch: call model.eval() first (core/spformer_trainer.py:119-120)'
        with self.amp.autocast():
            out = self.net({'lidar': ts.SparseTensor(feats, coords)})['x_vox']
        pred = voxel_logits_to_point_predictions(out, coords[:, -1], inverse_map, inverse_batch, keyframe_mask_full)
        order = torch.argsort(inverse_batch.long(), stable=True)
        targets = targets_mapped[order]
        if keyframe_mask_full is not None:
            targets = targets[keyframe_mask_full[order]]
        return {'outputs_vox': pred, 'targets': targets}


def kd_batch_to_device(b, device='cuda'):
    """numpy KD batch (synth.synth_kd_batch / the reference's collate schema) -> device tensors
    laid out as NuScenesLCTSDFullTrainer._prepare_input does (images -> [B, ncam, 3, H, W])."""
    s, t = b['student'], b['teacher']
    dev = torch.device(device)
    # (arrays, or the pinned host tensors of pin_kd_batch: then every copy is asynchronous on the current stream, as the
    # reference's `.cuda(non_blocking=True)` over its pinned loader output, core/nusc_trainers.py:257-279)
    f = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(a)).to(dev, non_blocking=True)
    out = {
        's_feats': f(s['feats']), 's_coords': f(s['coords']), 'targets': f(s['targets']),
        'images': f(s['images']).permute(0, 1, 4, 2, 3).contiguous(),
        'pixel_coordinates': [f(c) for c in s['pixel_coordinates']], 'masks': [f(m) for m in s['masks']],
        'fov_mask': f(s['fov_mask']), 'inds': [[f(i[0])] for i in s['inds']],
        't_feats': f(t['feats']), 't_coords': f(t['coords']), 'inverse_map': f(t['inverse_map']),
        'num_pts': list(t['num_pts']), 'num_vox_t': list(t['num_vox']),
        'keyframe_mask_full': f(t['keyframe_mask_full']) if 'keyframe_mask_full' in t else None,
    }
    # the teacher-only trainer's own labels and voxel-level key-frame mask (core/spformer_trainer.py:64-68), when the
    # batch carries them
    if 'targets' in t:
        out['targets_t'] = f(t['targets'])
    if 'keyframe_mask' in t:
        out['keyframe_mask'] = f(t['keyframe_mask'])
    return out


def pin_kd_batch(b):
    """The numpy KD batch as page-locked host tensors (what a DataLoader(pin_memory=True) hands the trainer,
    train_lc_nusc_tsd_full.py:63-78): ``kd_batch_to_device`` of the result issues asynchronous host-to-device copies."""
    def pin(v):
        if isinstance(v, np.ndarray):
            return torch.from_numpy(v).pin_memory()
        if isinstance(v, dict):
            return {k: pin(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)) and v and isinstance(v[0], (np.ndarray, list, tuple)):
            return type(v)(pin(x) for x in v)
        return v
    return pin(b)


def fresh_batch(d):
    """A new device copy of a resident KD batch: every tensor a new object (``clone``), as the host-to-device copy of a
    data loader delivers one per step (core/nusc_trainers.py:257-279).  Nothing that an earlier step cached on a batch
    tensor (point<->pixel plans hang on ``masks[0]``) can be seen by the step that receives the copy."""
    def cp(v):
        if torch.is_tensor(v):
            return v.clone()
        if isinstance(v, (list, tuple)):
            return type(v)(cp(x) for x in v)
        return v
    return {k: cp(v) for k, v in d.items()}


def _tensors_of(obj, seen=None, depth=0):
    """Every device tensor reachable from ``obj`` through attributes, dict values, lists and tuples (a prepared geometry:
    PointTensor / SparseTensor with their coordinate and kernel-map dictionaries, KernelMap objects with their tables and
    schedules, tensors' cached `_u2mkd_plans`)."""
    seen = set() if seen is None else seen
    if id(obj) in seen or depth > 14:
        return
    seen.add(id(obj))
    if torch.is_tensor(obj):
        if obj.is_cuda:
            yield obj
        for v in (obj.__dict__.get('_u2mkd_plans') or {}).values():
            yield from _tensors_of(v, seen, depth + 1)
        return
    if isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_of(v, seen, depth + 1)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_of(v, seen, depth + 1)
    elif hasattr(obj, '__dict__') and not isinstance(obj, (type, torch.nn.Module)):
        for v in vars(obj).values():
            yield from _tensors_of(v, seen, depth + 1)
    elif hasattr(obj, '__slots__'):
        for n in obj.__slots__:
            yield from _tensors_of(getattr(obj, n, None), seen, depth + 1)


class TeacherWatch:
    """Bit-reproducibility monitor of the frozen teacher.  The teacher runs under ``no_grad`` with eval-mode BatchNorm
    (core/nusc_trainers.py:285-324, tsd_full.py:590-596): its logits are a pure function of the batch, whatever the student
    does next to it on the other streams.  A forward hook keeps a clone of every step's teacher logits under the key the
    caller set (``watch.key = batch id``); ``deviating_steps()`` -- called after the run, it synchronises -- counts the steps
    whose logits differ in any bit from the FIRST step that saw the same key.  bench.py reports it for its timed run
    (``config.teacher_deviating_steps``), tools/soak.py asserts it is 0."""

    def __init__(self, model_t):
        self.key = None
        self.log = []
        self._handle = model_t.register_forward_hook(self._hook)

    def _hook(self, module, inputs, output):
        # the batch's own key where it carries one (`d['_key']`: the teacher of batch k + 1 may run inside step k, kd.teacher_ahead)
        key = inputs[0].get('_key', self.key) if inputs and isinstance(inputs[0], dict) else self.key
        if key is not None:
            self.log.append((key, output['x_vox'].detach().clone()))

    def deviating_steps(self):
        """(steps that differ from the first visit of their batch, steps compared)"""
        first, bad, compared = {}, 0, 0
        if self.log and self.log[0][1].is_cuda:
            torch.cuda.synchronize()
        for key, t in self.log:
            if key not in first:
                first[key] = t
            else:
                compared += 1
                bad += int(not torch.equal(first[key], t))
        return bad, compared

    def close(self):
        self._handle.remove()
        self.log = []


class KDStep:
    """Uni-to-multi-modal KD training step: frozen teacher forward (no grad, eval-mode BN),
    student forward/backward, the five loss terms."""

    def __init__(self, model: KD.TSDFull, num_epochs=50, batch_size=1, w_kl=1.0, w_feat=1.0, amp=False, criterion=None,
                 optimizer=None, scheduler=None):
        """``criterion``: a kd.KDCriterion (builder.make_kd_criterion); ``optimizer`` / ``scheduler``: factories of
        (module) / (optimizer), as for LidarStep (train_lc_nusc_tsd_full.py:84-93)."""
        self.model = model
        self.amp = _Amp(amp)
        deferred.acquire()         # (see LidarStep)
        weakref.finalize(self, deferred.release)
        if D.world() > 1 or os.environ.get('U2MKD_FORCE_DDP') == '1':      # (the knob: the N>1 code path on one GPU)
            from .lidar.point_voxel import SparseSyncBatchNorm
            model.model_s = SparseSyncBatchNorm.convert_sync_batchnorm(model.model_s)   # train_lc_nusc_tsd_full.py:80
        self.net = D.wrap_model(model, sync_bn=False)
        self.crit = criterion if criterion is not None else KD.KDCriterion(ignore_index=0, w_kl=w_kl, w_feat=w_feat)
        self.opt = optimizer(self.net) if optimizer is not None else \
            make_optimizer([p for p in self.net.parameters() if p.requires_grad])
        self.sched = scheduler(self.opt) if scheduler is not None else _scheduler(self.opt, num_epochs, batch_size)

    def train_mode(self):
        self.model.train()
        self.model.model_t.eval()          # core/nusc_trainers.py:203-208

    def _advance_geometry(self, geo, staged):
        return _advance_geometry(self.amp, geo, staged)

    @staticmethod
    def _in_mod(d):
        stu = {'lidar': ts.SparseTensor(d['s_feats'], d['s_coords']), 'images': d['images'],
               'pixel_coordinates': d['pixel_coordinates'], 'masks': d['masks'], 'fov_mask': d['fov_mask']}
        tea = {'lidar': ts.SparseTensor(d['t_feats'], d['t_coords'])}
        if '_key' in d:
            tea['_key'] = d['_key']          # (TeacherWatch: which resident batch this is)
        return {'student': stu, 'teacher': tea}

    def __call__(self, d, prefetch=None):
        """One training step on batch ``d``.  ``prefetch`` = the NEXT batch (the dict the next call will receive): its
        geometry (voxel sets, kernel maps: every host synchronisation of a step, kd.TSDFull.prepare) is built between
        this step's forward and backward, where the GPU queue is short; the next call then issues its whole forward
        without waiting for the GPU while this step's backward drains.  The work per batch is the same, it only moves
        one step ahead, as a data loader's prefetch does."""
        deferred.begin_step()      # (nothing of a backward pass that raised stays booked)
        deferred.set_reduced_precision(self.amp.enabled)      # (library kernels in bf16 / fp16: no side streams, deferred.overlap_ok)
        queued = self.__dict__.pop('_queued', None)
        in_mod = queued[1] if (queued is not None and queued[0] is d) else self._in_mod(d)
        entry = torch.cuda.current_stream().record_event() if d['s_feats'].is_cuda else None     # (see the geometry side stream below)
        if getattr(self, '_geo_done', None) is not None:
            torch.cuda.current_stream().wait_event(self._geo_done)
            self._geo_done = None
        staged = None
        if prefetch is not None and d['s_feats'].is_cuda and deferred.overlap_ok() and _staged_geometry():
            # The next batch's geometry in THREE slices on the side stream, one in front of each phase of this step: its two
            # host reads then fall behind the forward's and the backward's launch work (25 + 30 ms of host time) and find
            # their counts ready -- the step is bound by the host (tools/host_phases.py: host time = wall time), and the reads
            # were 5-6 ms of it sitting in hipStreamSynchronize.  Same launches, same order, same stream as the one-piece form
            # below; `entry`, `_geo_done`, `_geo_keep` and the record_stream registration as there.
            geo = KD._side_stream(d['s_feats'], 'geo')
            geo.wait_event(entry)
            staged = self.model.prepare_staged(self._in_mod(prefetch))
            self._advance_geometry(geo, staged)            # slice 1: point hashes, voxel sets (launches only)
        with self.amp.autocast():
            out = self.net(in_mod)
            ld = KD.kd_losses(out, d['targets'], d['fov_mask'], d['inverse_map'], d['inds'], d['num_pts'], d['num_vox_t'],
                              self.crit, d['keyframe_mask_full'])
        if staged is not None:
            self._advance_geometry(geo, staged)            # read 1 (ready), slice 2: the down-sampled levels
            self.amp.backward_and_step(ld['total'], self.opt)
            self.sched.step()
            after_bwd = torch.cuda.current_stream().record_event() if KD._TEACHER_AHEAD == 2 else None
            done = self._advance_geometry(geo, staged)     # read 2 (ready), slice 3: kernel maps, schedules
            self._queued = (prefetch, done)
            self._geo_done = geo.record_event()
            self._geo_keep = in_mod
            # the next batch's frozen-teacher forward, queued NOW: the GPU is still running this step's backward, the host
            # would otherwise arrive at the next step's start with the teacher's ~1 000 launches ahead of the student's
            with self.amp.autocast():
                KD.teacher_ahead(self.model, done, after=(entry, self._geo_done, after_bwd))
            users = [torch.cuda.current_stream(), KD._side_stream(d['s_feats'], 'teacher'), KD._side_stream(d['s_feats'], 'camera')]
            for key in ('student', 'teacher'):
                nxt = self._queued[1][key]
                # (the geometry, and the point <-> pixel plans kd.StudentMSP2IFM.prefetch_plans hung on the batch's masks)
                for t in _tensors_of([nxt.get('_geometry'), nxt.get('masks')]):
                    for st in users:
                        t.record_stream(st)
            return ld['total'].detach()
        if prefetch is not None and d['s_feats'].is_cuda and deferred.overlap_ok():
            # The next batch's geometry AFTER this step's backward has been issued, on a side stream: its ~1 000 small
            # kernels (hash tables, kernel maps, voxel sets) and the two host round trips run underneath the backward's
            # large kernels instead of between forward and backward with the main stream idle (same box: 77.9 -> 76.3 ms).
            # Memory: the side stream's allocations are consumed on the other streams one step later and freed there.
            # Ordering: (a) the next forward waits for `_geo_done`, (b) a batch's geometry is kept alive for one more step
            # (`_geo_keep`), (c) the side stream starts a preparation only behind `entry`, the main stream's position when
            # THIS call began; and (round 5) every tensor of the prepared geometry is registered with its reader streams
            # (record_stream, below), so the allocator itself holds a freed block until they have passed.
            self.amp.backward_and_step(ld['total'], self.opt)
            self.sched.step()
            geo = KD._side_stream(d['s_feats'], 'geo')
            geo.wait_event(entry)
            with torch.cuda.stream(geo), self.amp.autocast():
                self._queued = (prefetch, self.model.prepare(self._in_mod(prefetch)))
            self._geo_done = geo.record_event()
            self._geo_keep = in_mod
            # Round 5: the ownership argument above is no longer all there is -- every tensor of the prepared geometry is also
            # REGISTERED with the streams that will read it (the caller's and the teacher's): whenever it is freed, the
            # allocator holds its block until those streams have passed that point (~200 tensors, ~0.2 ms of host time a step)
            users = [torch.cuda.current_stream(), KD._side_stream(d['s_feats'], 'teacher'), KD._side_stream(d['s_feats'], 'camera')]
            for key in ('student', 'teacher'):
                nxt = self._queued[1][key]
                for t in _tensors_of([nxt.get('_geometry'), nxt.get('masks')]):
                    for st in users:
                        t.record_stream(st)
            return ld['total'].detach()
        if prefetch is not None:
            with self.amp.autocast():
                self._queued = (prefetch, self.model.prepare(self._in_mod(prefetch)))
        self.amp.backward_and_step(ld['total'], self.opt)
        self.sched.step()
        return ld['total'].detach()

    @torch.no_grad()
    def evaluate(self, d, s_inverse_map, s_inverse_batch, targets_mapped, label_fov, t_inverse_batch=None,
                 targets_mapped_t=None):
        """The eval branch of ``NuScenesLCTSDFullTrainer._run_step`` (core/nusc_trainers.py:367-418): the student's
        voxel logits and pixel-branch logits -> per-point predictions through the student's ``inverse_map``
        (``s_inverse_map`` [Np] voxel index within the scene, ``s_inverse_batch`` [Np] scene), with the mapped labels
        and the in-view labels; with ``debug_val`` (kd.TSDFull(debug_val=True)) also the teacher's predictions through
        ``d['inverse_map']``, key-frame points only for multi-sweep teachers.  Same dictionary keys as the
        reference, tensors stay on the device."""
        from .evaluate import voxel_logits_to_point_predictions as v2p
        assert not self.model.training, 'evaluate() is the eval branch: call model.eval() first (core/nusc_trainers.py:420-421)'
        deferred.set_reduced_precision(self.amp.enabled)
        with self.amp.autocast():
            out = self.net(self._in_mod(d))
        vb = d['s_coords'][:, -1]
        order = torch.argsort(s_inverse_batch.long(), stable=True)
        ret = {'outputs_vox': v2p(out['stu']['x_vox'], vb, s_inverse_map, s_inverse_batch),
               'outputs_pix': v2p(out['stu']['x_pix'], vb, s_inverse_map, s_inverse_batch),
               'targets': targets_mapped[order], 'targets_fov': label_fov[order]}
        if self.model.debug_val:
            kf = d.get('keyframe_mask_full')
            ret['outputs_vox_t'] = v2p(out['t']['x_vox'], d['t_coords'][:, -1], d['inverse_map'], t_inverse_batch, kf)
            order_t = torch.argsort(t_inverse_batch.long(), stable=True)
            ret['targets_t'] = targets_mapped_t[order_t] if kf is None else targets_mapped_t[order_t][kf[order_t]]
        return ret


# ------------------------------------------------------------------------------------- checkpoints
def _strip(sd):
    """DDP prefixes its keys with `module.`; the reference strips them on load (core/nusc_trainers.py:180,198)."""
    return {k.replace('module.', ''): v for k, v in sd.items()}


def state_dict(runner):
    """The trainer checkpoint of the reference, same four entries with the same meaning
    (``_state_dict``, core/nusc_trainers.py:423-429): model (keys as the reference's modules produce them: conv
    weights are ``...kernel`` [K, Cin, Cout]), scaler, optimizer, scheduler.  ``runner`` = LidarStep / KDStep."""
    return {'model': runner.net.state_dict(), 'scaler': runner.amp.scaler.state_dict(),
            'optimizer': runner.opt.state_dict(), 'scheduler': runner.sched.state_dict()}


def load_state_dict(runner, ckpt):
    """``_load_state_dict`` (core/nusc_trainers.py:431-435): resume model, loss scaler, optimizer and LR schedule."""
    own_ddp = any(k.startswith('module.') for k in runner.net.state_dict())
    model_sd = ckpt['model']
    if not own_ddp:
        model_sd = _strip(model_sd)
    runner.net.load_state_dict(model_sd)
    if 'scaler' in ckpt and ckpt['scaler']:
        runner.amp.scaler.load_state_dict(ckpt['scaler'])
    runner.opt.load_state_dict(ckpt['optimizer'])
    runner.sched.load_state_dict(ckpt['scheduler'])


def load_weights(model, weight_path=None, pretrain_weight=None, teacher_pretrain_weight=None, map_location='cpu'):
    """The three weight sources of ``_before_train`` (core/nusc_trainers.py:173-201), in the reference's order of
    precedence; ``model`` is the bare module (kd.TSDFull for the KD trainer).  Returns which one was used.

    * ``weight_path``: a trainer checkpoint -> its ``model`` entry, `module.` stripped, strict;
    * ``pretrain_weight``: ``model`` entry without the classifier heads, non-strict (fine-tuning);
    * ``teacher_pretrain_weight``: the stage-1 teacher (train_spformer.py) into ``model.model_t``, strict --
      the authors' teacher checkpoints load because module names and the ``kernel`` [K, Cin, Cout] layout are the
      reference's (tests/golden/*_keys.json)."""
    import os
    if weight_path is not None and os.path.exists(weight_path):
        sd = torch.load(weight_path, map_location=map_location, weights_only=False)
        model.load_state_dict(_strip(sd['model']))
        return 'weight_path'
    if pretrain_weight is not None and os.path.exists(pretrain_weight):
        sd = torch.load(pretrain_weight, map_location=map_location, weights_only=False)['model']
        model.load_state_dict({k: v for k, v in sd.items() if 'classifier' not in k}, strict=False)
        return 'pretrain_weight'
    if teacher_pretrain_weight is not None and os.path.exists(teacher_pretrain_weight):
        sd = torch.load(teacher_pretrain_weight, map_location=map_location, weights_only=False)['model']
        model.model_t.load_state_dict(_strip(sd), strict=True)
        return 'teacher_pretrain_weight'
    return None
