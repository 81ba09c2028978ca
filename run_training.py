"""Training entry point: what train_lc_nusc_tsd_full.py (the KD student, :27-127) and train_spformer.py (the stage-1
teacher, :27-116) do, on the HIP operators.  One script; the trainer follows `model.name` of the configuration.

    python run_training.py configs/nuscenes/train/spformer_tsd_full_ours_star.yaml --run-dir runs/kd \\
        --model.in_channel_t 4 [--weight-path ckpt.pt] [--non-dist] [--max-iters K] [--synthetic N_VOXELS]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 run_training.py CONFIG ...

Same order of business as the reference: configuration (file, recursive defaults, command-line overrides), process
group, seeding (`train.seed + rank * workers_per_gpu * num_epochs`), datasets with a DistributedSampler each, model ->
SyncBatchNorm conversion -> DDP (inside train.KDStep / train.LidarStep, train_lc_nusc_tsd_full.py:78-84), criterion /
optimizer / scheduler from u2mkd_amd.builder, the three weight sources of the trainer's `_before_train`, then per epoch:
training steps, the eval branch over the validation split with the MeanIoU callbacks (iou-vox / iou-pix / iou-vox-t
with `debug.debug_val`), and the checkpoints torchpack's savers write (`checkpoints/step-<global step>.pt`, the latest
only, and `checkpoints/max-<metric>.pt` whenever the metric improves: the file the KD configuration's `teacher_pretrain`
points to is the teacher run's `max-iou-val-vox.pt`).  `--synthetic` replaces the on-disk dataset by seeded synthetic
scenes of that many voxels (no nuScenes in the container); `--max-iters` ends every epoch early (smoke runs)."""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from u2mkd_amd import builder, distributed as D, train as T        # noqa: E402
from u2mkd_amd.evaluate import MeanIoU                               # noqa: E402

KD_MODEL = 'spvcnn_swiftnet18_spformer_tsd_full'


def log(*a):
    if D.rank() == 0:
        print('[run_training]', *a, flush=True)


# ------------------------------------------------------------------------------------------------- batches
def _cuda(x):
    return x.cuda(non_blocking=True)


def kd_eval_feed(collated):
    """The tensors of the eval branch (core/nusc_trainers.py:367-418) out of a collated loader batch."""
    s, t = collated['feed_dict_s'], collated['feed_dict_t']
    out = {'s_inverse_map': _cuda(s['inverse_map'].F.long()), 's_inverse_batch': _cuda(s['inverse_map'].C[:, -1].long()),
           'targets_mapped': _cuda(s['targets_mapped'].F.long()),
           't_inverse_batch': _cuda(t['inverse_map'].C[:, -1].long()), 'targets_mapped_t': _cuda(t['targets_mapped'].F.long())}
    fov = s.get('label_fov')                 # only with debug.debug_val (the loader's :453-456); otherwise the mapped labels
    out['label_fov'] = _cuda(fov.F.long()) if fov is not None else out['targets_mapped']
    return out


class SyntheticKD(torch.utils.data.Dataset):
    """Seeded synthetic scenes in the loader's collated schema (synth.synth_kd_batch), one scene per item."""

    def __init__(self, n_vox, length, image_hw, seed, sweeps=0):
        self.n_vox, self.length, self.hw, self.seed, self.sweeps = n_vox, length, image_hw, seed, sweeps

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        return i

    def collate_fn(self, idx):
        from u2mkd_amd.synth import synth_eval_feed, synth_kd_batch
        seed = self.seed + 7919 * int(idx[0])
        b = synth_kd_batch(self.n_vox, len(idx), seed=seed, image_hw=self.hw)
        b['student']['images'] = ((b['student']['images'] / 255.0 - 0.45) / 0.225).astype(np.float32)
        return {'_numpy': b, '_eval': synth_eval_feed(b, seed)}


def to_device(collated):
    """(training batch of train.KDStep, eval tensors) of one loader batch."""
    if '_numpy' in collated:
        d = T.kd_batch_to_device(collated['_numpy'])
        return d, {k: _cuda(torch.from_numpy(np.asarray(v))) for k, v in collated['_eval'].items()}
    from u2mkd_amd.data.nuscenes_lc import collated_to_kd_batch
    return T.kd_batch_to_device(collated_to_kd_batch(collated)), kd_eval_feed(collated)


def teacher_view(d, ev):
    """The teacher-only trainer's feed dict (core/spformer_trainer.py:58-94) out of the KD batch's teacher half."""
    kf = d.get('keyframe_mask_full')
    targets_t = d.get('targets_t')
    return dict(feats=d['t_feats'], coords=d['t_coords'], targets=targets_t, inverse_map=d['inverse_map'],
                inverse_batch=ev['t_inverse_batch'], targets_mapped=ev['targets_mapped_t'], keyframe_mask_full=kf)


def seed_worker(dataset, seed, epoch, workers, worker_id):
    """The reference's `worker_init_fn` (core/nusc_trainers.py:210-211, re-installed before every epoch): loader worker
    `worker_id` of epoch `epoch` draws its augmentations from seed + (epoch - 1) * workers + worker_id -- there through
    numpy's global state, here through the dataset copy's own generator (without it every worker would replay the parent's
    stream)."""
    s = seed + (epoch - 1) * workers + worker_id
    np.random.seed(s % (2 ** 32))
    if hasattr(dataset, 'rng'):
        dataset.rng = np.random.default_rng(s)
    return s


# ------------------------------------------------------------------------------------------------- savers
class Savers:
    """torchpack's `Saver(max_to_keep=1)` + `MaxSaver(metric)` as the reference configures them
    (train_lc_nusc_tsd_full.py:123-127): rank 0 only."""

    def __init__(self, run_dir, metrics):
        self.dir = os.path.join(run_dir, 'checkpoints')
        self.best = {m: None for m in metrics}
        self.last = None
        if D.rank() == 0:
            os.makedirs(self.dir, exist_ok=True)

    def after_epoch(self, runner, global_step, values, extra):
        if D.rank() != 0:
            return
        improved = [m for m, best in self.best.items() if m in values and (best is None or values[m] > best)]
        for m in improved:
            self.best[m] = values[m]
        # the best values travel with the state, so a resumed run does not overwrite max-*.pt with a worse epoch
        state = dict(T.state_dict(runner), best_metrics=dict(self.best), **extra)
        path = os.path.join(self.dir, 'step-%d.pt' % global_step)
        torch.save(state, path)
        if self.last and self.last != path and os.path.exists(self.last):
            os.remove(self.last)
        self.last = path
        for m in improved:
            torch.save(state, os.path.join(self.dir, 'max-%s.pt' % m.replace('/', '-')))


# ------------------------------------------------------------------------------------------------- main
def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('config', metavar='FILE')
    ap.add_argument('--run-dir', metavar='DIR', default=None)
    ap.add_argument('--weight-path', metavar='FILE', default=None)
    ap.add_argument('--non-dist', action='store_true', help='single process (the reference spells this flag inverted)')
    ap.add_argument('--max-iters', type=int, default=0, help='stop every epoch (training and validation) after K batches')
    ap.add_argument('--synthetic', type=int, default=0, metavar='N_VOXELS', help='synthetic scenes instead of dataset.root')
    ap.add_argument('--backend', default=None, help='process-group backend (default nccl = RCCL; gloo: several ranks on one GPU, tests)')
    args, opts = ap.parse_known_args(argv)

    cfg = builder.Config.load(args.config, recursive=True)
    cfg.update(opts)
    cfg.update({k: v for k, v in vars(args).items() if k != 'config'})
    D.configure_runtime()                      # (before the first HIP call of the process)
    if not torch.cuda.is_available():
        raise RuntimeError('run_training.py drives the HIP operators: no GPU is visible (there is no CPU fallback)')
    if not args.non_dist:
        D.init_from_env(args.backend)
    rank, world = D.rank(), D.world()
    run_dir = args.run_dir or os.path.join('runs', time.strftime('run-%Y%m%d-%H%M%S'))
    if rank == 0:
        os.makedirs(run_dir, exist_ok=True)
        with open(os.path.join(run_dir, 'configs.json'), 'w') as f:
            json.dump(cfg, f, indent=1, default=str)
    log('experiment "%s", world %d' % (run_dir, world))

    # seeds (train_lc_nusc_tsd_full.py:52-60)
    train_cfg = cfg.setdefault('train', builder.Config())
    if train_cfg.get('seed') is None:
        train_cfg['seed'] = torch.initial_seed() % (2 ** 32 - 1)
    workers = cfg.get('workers_per_gpu', 4)
    seed = train_cfg['seed'] + rank * workers * cfg.num_epochs
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed); torch.cuda.manual_seed(seed)

    is_kd = cfg.model.name == KD_MODEL
    # datasets and loaders (:62-76)
    if args.synthetic:
        hw = tuple(int(x * cfg.dataset.get('im_cr', 0.4)) for x in (900, 1600))
        n_items = max(cfg.batch_size * world * max(args.max_iters, 2), 8)
        dataset = {'train': SyntheticKD(args.synthetic, n_items, hw, seed), 'val': SyntheticKD(args.synthetic, n_items, hw, seed + 1)}
        workers = 0
    else:
        dataset = builder.make_dataset(cfg, rng=np.random.default_rng(seed))
    flow = {}
    epoch_box = [1]                       # read by the workers when an epoch's iterator starts them
    for split, ds in dataset.items():
        sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=(split == 'train'))
        flow[split] = torch.utils.data.DataLoader(
            ds, batch_size=cfg.batch_size, sampler=sampler, num_workers=workers, pin_memory=True, collate_fn=ds.collate_fn,
            worker_init_fn=lambda wid: seed_worker(torch.utils.data.get_worker_info().dataset, seed, epoch_box[0], workers, wid))

    # model, weights, trainer (:78-99)
    model = builder.make_model(cfg).cuda()
    used = T.load_weights(model, weight_path=args.weight_path, pretrain_weight=cfg.model.get('pretrain_weight'),
                          teacher_pretrain_weight=cfg.model.get('teacher_pretrain') if is_kd else None)
    log('weights:', used or 'random initialisation')
    make_opt = lambda net: builder.make_optimizer(cfg, net)
    make_sch = lambda opt: builder.make_scheduler(cfg, opt)
    amp = bool(cfg.get('amp_enabled', False))
    if is_kd:
        runner = T.KDStep(model, amp=amp, criterion=builder.make_kd_criterion(cfg), optimizer=make_opt, scheduler=make_sch)
        names = [('iou-vox/val', 'outputs_vox', 'targets'), ('iou-pix/val', 'outputs_pix', 'targets_fov')]
        if cfg.get('debug', {}).get('debug_val', False):
            names.append(('iou-vox-t/val', 'outputs_vox_t', 'targets_t'))
        savers = Savers(run_dir, ['iou-vox/val', 'iou-pix/val'])
    else:
        runner = T.LidarStep(model, amp=amp, criterion=builder.make_criterion(cfg), optimizer=make_opt, scheduler=make_sch)
        names = [('iou/val/vox', 'outputs_vox', 'targets')]
        savers = Savers(run_dir, ['iou/val/vox'])
    metrics = [MeanIoU(cfg.data.num_classes, cfg.data.ignore_label, out, tgt, name=n) for n, out, tgt in names]

    def batches(split, epoch):
        flow[split].sampler.set_epoch(epoch - 1)          # core/nusc_trainers.py:209
        epoch_box[0] = epoch
        for i, c in enumerate(flow[split]):
            if args.max_iters and i >= args.max_iters:
                break
            yield to_device(c)

    def train_step(cur, nxt):
        d, ev = cur
        if is_kd:
            return runner(d, prefetch=nxt[0] if nxt is not None else None)
        tv = teacher_view(d, ev)
        kf = d.get('keyframe_mask')
        return runner(tv['feats'], tv['coords'], tv['targets'], keyframe_mask=kf)

    global_step, first_epoch = 0, 1
    if used == 'weight_path' and not args.non_dist:
        # a distributed run resumes the whole trainer from `--weight-path` (model, loss scaler, optimizer, LR schedule:
        # core/nusc_trainers.py:174-177, 431-435); `--non-dist` takes the model weights only (:178-180), done above
        ckpt = torch.load(args.weight_path, map_location='cpu', weights_only=False)
        T.load_state_dict(runner, ckpt)
        # position: our key 'epoch' or torchpack's Trainer.state_dict() name 'epoch_num' (an upstream step-*.pt); without
        # either the run would silently restart at epoch 1 on an already-advanced LR schedule -- refuse that
        if 'epoch' not in ckpt and 'epoch_num' not in ckpt:
            raise KeyError('%s holds no epoch position (epoch / epoch_num): cannot resume the schedule' % args.weight_path)
        global_step = int(ckpt.get('global_step', 0))
        first_epoch = int(ckpt.get('epoch', ckpt.get('epoch_num', 0))) + 1
        savers.best.update({m: v for m, v in (ckpt.get('best_metrics') or {}).items() if m in savers.best})
        log('resumed the trainer state at epoch %d, step %d' % (first_epoch, global_step))
    history = []
    for epoch in range(first_epoch, cfg.num_epochs + 1):
        runner.train_mode() if is_kd else model.train()
        t0, losses = time.time(), []
        it = batches('train', epoch)
        cur = next(it, None)
        while cur is not None:                       # one batch ahead: the next batch's geometry is built under this step's backward
            nxt = next(it, None)
            losses.append(train_step(cur, nxt))
            global_step += 1
            cur = nxt
        mean_loss = float(torch.stack(losses).mean()) if losses else float('nan')
        log('epoch %d: %d steps, mean loss %.4f, %.1f s' % (epoch, len(losses), mean_loss, time.time() - t0))

        model.eval()
        for m in metrics:
            m.before_epoch()
        for d, ev in batches('val', epoch):
            if is_kd:
                ret = runner.evaluate(d, ev['s_inverse_map'], ev['s_inverse_batch'], ev['targets_mapped'], ev['label_fov'],
                                      ev['t_inverse_batch'], ev['targets_mapped_t'])
            else:
                tv = teacher_view(d, ev)
                ret = runner.evaluate(tv['feats'], tv['coords'], tv['inverse_map'], tv['inverse_batch'], tv['targets_mapped'],
                                      tv['keyframe_mask_full'])
            for m in metrics:
                m.after_step(ret)
        values = {m.name: m.after_epoch()[0] for m in metrics}
        log('epoch %d:' % epoch, '  '.join('%s %.3f' % kv for kv in values.items()))
        history.append(dict(epoch=epoch, loss=mean_loss, **values))
        savers.after_epoch(runner, global_step, values, {'epoch': epoch, 'epoch_num': epoch, 'local_step': len(losses),
                                                         'global_step': global_step})
        if rank == 0:
            with open(os.path.join(run_dir, 'history.json'), 'w') as f:
                json.dump(history, f, indent=1)
    D.barrier()
    D.shutdown()
    return history


if __name__ == '__main__':
    main()
