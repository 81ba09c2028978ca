"""Configuration -> objects: the branches of core/builder.py that the KD training path uses.

The reference keeps one global ``configs`` (torchpack.utils.config, an un-vendored dependency: a nested dict with
attribute access, ``load(path, recursive=True)`` = every ``default.yaml`` from the outermost directory of the path down,
then the file itself, later values over earlier ones) and every ``make_*`` reads it; its model classes read it again in
their constructors (``configs['model']['cr_t']``, ``configs['debug']['debug_val']``, ``configs['eval'][...]``:
core/models/nuscenes/spvcnn_swiftnet18_spformer_tsd_full.py:25-31,204-216,325-326,574-575).  Here the configuration is an
explicit argument and the constructors receive plain keywords, so two models of different configurations can live in one
process; the names, keys, defaults and the NotImplementedError of an unknown name are the reference's.

    cfg = Config.load('configs/nuscenes/train/spformer_tsd_full_ours_star.yaml', recursive=True)
    cfg.update(['--model.in_channel_t', '4'])                 # the command-line overrides of train_lc_nusc_tsd_full.py:33-35
    model = make_model(cfg)                                    # core/builder.py:170-622
    criterion = make_criterion_dict(cfg)                       # core/builder.py:645-660
    optimizer = make_optimizer(cfg, model)                     # core/builder.py:662-718
    scheduler = make_scheduler(cfg, optimizer)                 # core/builder.py:721-748
"""
import ast
import os

import numpy as np
import torch
from torch import nn

from . import distributed as D
from .losses import MixLovaszCrossEntropy
from .train import cosine_schedule_with_warmup, make_optimizer as _make_optimizer

__all__ = ['Config', 'make_dataset', 'make_model', 'make_kd_criterion', 'make_criterion', 'make_criterion_dict', 'make_optimizer', 'make_scheduler']


class Config(dict):
    """Nested dict with attribute access (what the reference's code does with its ``configs``: ``configs.model.cr``,
    ``configs['criterion'].get('class_weight')``, ``'cr' in configs.model``)."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None

    def __setattr__(self, key, value):
        self[key] = value

    @staticmethod
    def _wrap(value):
        if isinstance(value, dict) and not isinstance(value, Config):
            out = Config()
            out.update(value)
            return out
        return value

    def __setitem__(self, key, value):
        super().__setitem__(key, Config._wrap(value))

    def update(self, other=(), **kwargs):
        """A mapping is merged recursively (a later file refines ``model:`` without dropping the earlier keys); a list
        is command-line overrides ``--a.b value`` / ``--a.b=value`` with literal values (``1e-3``, ``[2, 2, 120]``,
        ``true``), anything that does not parse stays a string."""
        if isinstance(other, (list, tuple)) and (not other or isinstance(other[0], str)):
            other = _parse_overrides(list(other))
        for src in (dict(other), kwargs):
            for key, value in src.items():
                if isinstance(value, dict) and isinstance(self.get(key), Config):
                    self[key].update(value)
                else:
                    self[key] = value
        return self

    @classmethod
    def load(cls, path, recursive=False, into=None):
        import yaml
        cfg = cls() if into is None else into
        if not os.path.isfile(path):
            raise FileNotFoundError(path)
        paths = [path]
        if recursive:
            ext = os.path.splitext(path)[1]
            d = os.path.dirname(path)
            while True:
                paths.append(os.path.join(d, 'default' + ext))
                parent = os.path.dirname(d)
                if not d or parent == d:
                    break
                d = parent
        for p in reversed(paths):                      # outermost default first, the named file last
            if os.path.isfile(p):
                with open(p) as f:
                    cfg.update(yaml.safe_load(f) or {})
        return cfg


def _literal(text):
    low = text.lower()
    if low in ('true', 'false'):
        return low == 'true'
    if low in ('none', 'null', '~'):
        return None
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        try:
            return float(text)                         # '1.0e-4' and friends that literal_eval already takes; kept for '1e-4'
        except ValueError:
            return text


def _parse_overrides(opts):
    out = {}
    i = 0
    while i < len(opts):
        tok = opts[i]
        if not tok.startswith('--'):
            raise ValueError(f'configuration override {tok!r}: expected --key.sub value')
        if '=' in tok:
            key, value = tok[2:].split('=', 1)
            i += 1
        else:
            if i + 1 >= len(opts):
                raise ValueError(f'configuration override {tok!r} has no value')
            key, value = tok[2:], opts[i + 1]
            i += 2
        node = out
        parts = key.split('.')
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = _literal(value)
    return out


def make_dataset(cfg, dataset_name=None, tables=None, rng=None) -> dict:
    """core/builder.py:17-127, the branch of the KD path: `lc_semantic_nusc_tsd_full` -> {'train', 'val'} of the
    LiDAR + camera loader (core/datasets/lc_semantic_nusc_tsd_full.py:60-67) with the keys its constructor reads from
    the configuration (:131-158).  ``tables``: an opened data.NuScenesTables (default: `dataset.root`, version
    `dataset.version` or v1.0-trainval); the official split index files (`./data/nuscenes/nuscenes_{train,val}_official.npy`,
    :160-165) are used when they exist next to the working directory, otherwise every sample belongs to both splits.
    The teacher-only trainer's `semantic_nusc` is served by the same loader: its `feed_dict_t` is that trainer's feed
    dict (lidar, targets, targets_mapped, inverse_map, key-frame masks)."""
    from .data import nuscenes_lc as data
    cfg = Config._wrap(cfg)
    name = cfg.dataset.name if dataset_name is None else dataset_name
    if name not in ('lc_semantic_nusc_tsd_full', 'semantic_nusc'):
        raise NotImplementedError(name)
    ds = cfg.dataset
    if tables is None:
        tables = data.NuScenesTables(ds.root, ds.get('version', 'v1.0-trainval'))
    sweeps = ds.get('multisweeps', {})
    common = dict(voxel_size=ds.voxel_size, im_cr=ds.get('im_cr', 0.4), im_drop=ds.get('im_drop', 0), flip=ds.get('flip', ds.get('flip_aug', True)),
                  multisweeps=sweeps.get('num_sweeps', 0), only_past=sweeps.get('only_past', False),
                  ignore_index=cfg.criterion.ignore_index, debug=cfg.get('debug', {}).get('debug_val', False), rng=rng)
    out = {}
    for split in ('train', 'val'):
        idx_file = os.path.join('data', 'nuscenes', 'nuscenes_%s_official.npy' % split)
        select = np.load(idx_file) if os.path.isfile(idx_file) else None
        out[split] = data.LCNuScenesDataset(tables, split=split, select_idx=select, **common)
    return out


def _spformer_arguments(cfg):
    """The keyword block core/builder.py:533-554 (and :599-620, identical) computes for the SphereFormer models."""
    voxel_size = cfg.dataset.voxel_size
    voxel_size_list = list(voxel_size) if isinstance(voxel_size, (list, tuple)) else [voxel_size] * 3
    patch_size = np.array([voxel_size_list[i] * cfg.model.patch_size for i in range(3)]).astype(np.float32)
    window_size = patch_size * cfg.model.window_size
    window_size_sphere = np.array(cfg.model.window_size_sphere)
    return dict(window_size=window_size, window_size_sphere=list(cfg.model.window_size_sphere),
                quant_size=window_size / cfg.model.quant_size_scale,
                quant_size_sphere=window_size_sphere / cfg.model.quant_size_scale,
                drop_path_rate=cfg.model.drop_path_rate, window_size_scale=list(cfg.model.window_size_scale),
                a=cfg.model.a, pres=voxel_size, vres=voxel_size)


def make_model(cfg, model_name=None) -> nn.Module:
    """core/builder.py:170-622, the three branches that are not commented out there: `spvcnn` (:176-184),
    `spvcnn_spformer` (:533-554; the teacher-only trainer) and `spvcnn_swiftnet18_spformer_tsd_full` (:599-620; the KD
    step).  What the reference's constructors pick out of the global configuration is passed here."""
    from . import kd, lidar
    cfg = Config._wrap(cfg)
    if model_name is None:
        model_name = cfg.model.name
    model_cfg = cfg.model
    cr = model_cfg.cr if 'cr' in model_cfg else 1.0
    num_classes = cfg.data.num_classes
    if model_name == 'spvcnn':
        return lidar.SPVCNN(in_channel=model_cfg.in_channel, num_classes=num_classes, cr=cr,
                            pres=cfg.dataset.voxel_size, vres=cfg.dataset.voxel_size)
    if model_name == 'spvcnn_spformer':
        # core/models/nuscenes/spvcnn_spformer.py:22-30 reads model.cr / model.in_channel; the copy of the class inside the KD
        # model's file reads model.cr_t / model.in_channel_t (tsd_full.py:25-31) -- the branch below
        return lidar.SPVCNN_SPFORMER(cr=cr, in_channel=model_cfg.in_channel, num_classes=num_classes,
                                     **_spformer_arguments(cfg))
    if model_name == 'spvcnn_swiftnet18_spformer_tsd_full':
        debug = cfg.get('debug', {})
        evaluation = cfg.get('eval', {})
        return kd.TSDFull(cr=cr, cr_t=model_cfg.cr_t, in_channel=model_cfg.in_channel,
                          in_channel_t=model_cfg['in_channel_t'], num_classes=num_classes,
                          spformer=_spformer_arguments(cfg), imagenet_pretrain=model_cfg['imagenet_pretrain'],
                          run_pix_decoder=evaluation.get('run_pix_decoder', True), debug_val=debug.get('debug_val', False))
    raise NotImplementedError(model_name)


def _class_weight(cfg, device):
    w = cfg['criterion'].get('class_weight', None)
    return None if w is None else torch.tensor(w, dtype=torch.float32, device=device)


def make_criterion(cfg, device=None):
    """core/builder.py:623-642: `cross_entropy` and `lovasz` (the teacher-only trainer's).  `lc_lovasz` /
    `lc_lovasz_distill` belong to trainers outside the KD path (out of scope)."""
    cfg = Config._wrap(cfg)
    name = cfg.criterion.name
    if device is None:
        device = 'cuda' if torch.cuda.is_available() else 'cpu'
    if name == 'cross_entropy':
        return nn.CrossEntropyLoss(ignore_index=cfg.criterion.ignore_index)
    if name == 'lovasz':
        return MixLovaszCrossEntropy(weight=_class_weight(cfg, device), ignore_index=cfg.criterion.ignore_index)
    raise NotImplementedError(name)


def make_criterion_dict(cfg) -> dict:
    """core/builder.py:645-660: one criterion per name of the list `criterion.name` (ce / lovasz / kl / mse)."""
    cfg = Config._wrap(cfg)
    out = {}
    for name in cfg['criterion']['name']:
        if name == 'ce':
            out['ce'] = nn.CrossEntropyLoss(ignore_index=cfg.criterion.ignore_index)
        elif name == 'lovasz':
            out['lovasz'] = MixLovaszCrossEntropy(ignore_index=cfg.criterion.ignore_index)
        elif name == 'kl':
            out['kl'] = nn.KLDivLoss(reduction='batchmean')
        elif name == 'mse':
            out['mse'] = nn.MSELoss(reduction='mean')
        else:
            raise NotImplementedError(name)
    return out


def make_kd_criterion(cfg):
    """The same three criteria with the weights the trainer reads next to them (core/nusc_trainers.py:300-340:
    `w_kl`, `w_feat`, `mse_norm_feat`) as the module kd.kd_losses takes."""
    from .kd import KDCriterion
    cfg = Config._wrap(cfg)
    c = cfg.criterion
    return KDCriterion(ignore_index=c.ignore_index, w_kl=c.get('w_kl', 1.0), w_feat=c.get('w_feat', 1.0),
                       mse_norm_feat=c.get('mse_norm_feat', False))


def make_optimizer(cfg, model: nn.Module):
    """core/builder.py:662-718 (the five names; the arithmetic lives in train.make_optimizer)."""
    cfg = Config._wrap(cfg)
    o = cfg.optimizer
    split = o.name in ('sgd_spformer', 'adamw_spformer')
    return _make_optimizer(model if split else model.parameters(), lr=o.lr, momentum=o.get('momentum', 0.9),
                           weight_decay=o.weight_decay, name=o.name, nesterov=o.get('nesterov', True),
                           transformer_lr_scale=o.get('transformer_lr_scale', 0.1))


class PolyLR(torch.optim.lr_scheduler.LambdaLR):
    """core/schedulers.py:38-57: (1 - step / (max_iter + 1)) ** power per iteration."""

    def __init__(self, optimizer, max_iter, power=0.9, last_step=-1):
        super().__init__(optimizer, lambda s: (1 - s / (max_iter + 1)) ** power, last_step)


def make_scheduler(cfg, optimizer, world=None):
    """core/builder.py:721-748: none / cosine / cosine_warmup (per iteration, core/schedulers.py:10-35) / poly."""
    cfg = Config._wrap(cfg)
    name = cfg.scheduler.name
    if name == 'none':
        return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda epoch: 1)
    if name == 'cosine':
        return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=cfg.num_epochs)
    if name == 'cosine_warmup':
        w = D.world() if world is None else world
        ne, bs, ds = cfg.num_epochs, cfg.batch_size, cfg.data.training_size
        return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda k: cosine_schedule_with_warmup(k, ne, bs, ds, w))
    if name == 'poly':
        return PolyLR(optimizer, max_iter=cfg.num_epochs * cfg.data.training_size, power=cfg.scheduler.power)
    raise NotImplementedError(name)
