"""Generate golden vectors by importing the REFERENCE's own Python modules
(from /root/reference, build container only) over the CPU oracle operators.

    python tests/golden/make_golden.py

Outputs small .npz / .json fixtures next to this file.  The fixtures are data
(inputs are regenerated from seeds, expected outputs are stored); no reference
source travels.  What each fixture pins:

* spvcnn_cr05_4000.npz  -- logits + loss of the reference `SPVCNN` class
  (core/models/semantickitti/spvcnn.py) with `core/models/utils.py` and
  `core/models/build_blocks.py` unchanged, run over oracle.torchsparse_cpu:
  pins the MODEL WIRING of oracle.spvcnn_ref (bit-exact on CPU) and of
  u2mkd_amd.lidar.SPVCNN (1e-3 on the GPU), and the state-dict key set.
* lovasz_ce.npz -- reference `MixLovaszCrossEntropy` (core/criterions.py) values
  and gradients on seeded logits: pins oracle + u2mkd_amd.losses.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

from oracle import torchsparse_cpu as ots  # noqa: E402
from oracle import spvcnn_ref as O         # noqa: E402
from u2mkd_amd.synth import synth_batch    # noqa: E402


def import_reference():
    ots.install()                                   # `import torchsparse` -> CPU oracle
    sys.modules.setdefault('torchvision', types.ModuleType('torchvision'))
    tvt = types.ModuleType('torchvision.transforms')
    tvf = types.ModuleType('torchvision.transforms.functional')
    sys.modules.setdefault('torchvision.transforms', tvt)
    sys.modules.setdefault('torchvision.transforms.functional', tvf)
    sys.path.insert(0, REF)
    from core.models.semantickitti.spvcnn import SPVCNN     # the reference class
    from core.criterions import MixLovaszCrossEntropy
    return SPVCNN, MixLovaszCrossEntropy


def import_reference_spformer(cr):
    """The reference's SphereFormer / SPVCNN_SPFORMER modules with their un-installable imports
    stubbed: timm (DropPath, trunc_normal_), torch_scatter, torchpack configs, and
    third_party.SparseTransformer.sptr -> oracle.sptr_cpu (CPU restatement of the CUDA ops)."""
    from oracle import sptr_cpu
    from oracle.spformer_ref import DropPath
    timm = types.ModuleType('timm')
    timm_models = types.ModuleType('timm.models')
    timm_layers = types.ModuleType('timm.models.layers')
    timm_layers.DropPath = DropPath
    timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
    sys.modules.update({'timm': timm, 'timm.models': timm_models, 'timm.models.layers': timm_layers})
    tscatter = types.ModuleType('torch_scatter')
    tscatter.scatter_mean = None
    sys.modules['torch_scatter'] = tscatter
    for name in ('third_party', 'third_party.SparseTransformer'):
        sys.modules[name] = types.ModuleType(name)
    # a package path lets fusion_blocks import the reference's pure-python third_party.csrc wrapper
    # (dead CamLiFlow code with torch fallbacks); the sptr entry below still wins over the real one
    sys.modules['third_party'].__path__ = [os.path.join(REF, 'third_party')]
    sys.modules['third_party.SparseTransformer.sptr'] = sptr_cpu
    tp = types.ModuleType('torchpack')
    tpu = types.ModuleType('torchpack.utils')
    tpc = types.ModuleType('torchpack.utils.config')
    tpc.configs = {'model': {'cr': cr, 'in_channel': 4}, 'data': {'num_classes': 17}}
    sys.modules.update({'torchpack': tp, 'torchpack.utils': tpu, 'torchpack.utils.config': tpc})
    from core.models.nuscenes.spvcnn_spformer import SPVCNN_SPFORMER
    return SPVCNN_SPFORMER


def main():
    SPVCNN, MixLovaszCrossEntropy = import_reference()
    torch.manual_seed(0)
    kw = dict(cr=0.5, in_channel=4, num_classes=17, pres=0.05, vres=0.05)
    b = synth_batch(4000, 1, seed=21)
    feats, coords, labels = (torch.from_numpy(b[k]) for k in ('feats', 'coords', 'labels'))

    ref = O.fill_state_by_name(SPVCNN(**kw)).train()
    ref.dropout.p = 0.0
    out = ref({'lidar': ots.SparseTensor(feats.clone(), coords.clone())})['x_vox']
    crit = MixLovaszCrossEntropy(ignore_index=0)
    loss = crit(out, labels)
    loss.backward()
    sd = ref.state_dict()
    grads = {n: p.grad for n, p in ref.named_parameters()}
    np.savez_compressed(
        os.path.join(HERE, 'spvcnn_cr05_4000.npz'),
        logits=out.detach().numpy().astype(np.float32), loss=np.float32(loss.item()),
        grad_stem0=grads['stem.0.kernel'].numpy(), grad_cls=grads['classifier_vox.0.weight'].numpy(),
        grad_up3=grads['vox_ups.3.1.1.net.3.kernel'].numpy()[13])
    with open(os.path.join(HERE, 'spvcnn_cr05_keys.json'), 'w') as f:
        json.dump({k: list(v.shape) for k, v in sd.items()}, f, indent=0)

    g = torch.Generator().manual_seed(5)
    x = torch.randn(3000, 17, generator=g, requires_grad=True)
    y = torch.randint(0, 17, (3000,), generator=g)
    l2 = crit(x, y)
    l2.backward()
    np.savez_compressed(os.path.join(HERE, 'lovasz_ce.npz'), x=x.detach().numpy(), y=y.numpy(),
                        loss=np.float32(l2.item()), grad=x.grad.numpy())
    # ---- SPVCNN_SPFORMER (teacher), the reference class with the builder's arguments
    from oracle.spformer_ref import default_spformer_kwargs
    cr = 1.0
    SPF = import_reference_spformer(cr)
    kw = default_spformer_kwargs(cr=cr, drop_path_rate=0.0)
    for k in ('cr', 'in_channel', 'num_classes'):
        kw.pop(k)
    b = synth_batch(2000, 2, seed=33)
    feats, coords, labels = (torch.from_numpy(b[k]) for k in ('feats', 'coords', 'labels'))
    ref = O.fill_state_by_name(SPF(**kw)).train()
    ref.dropout.p = 0.0
    out = ref({'lidar': ots.SparseTensor(feats.clone(), coords.clone())})['x_vox']
    loss3 = crit(out, labels)
    loss3.backward()
    grads = {n: p.grad for n, p in ref.named_parameters()}
    blk = 'transformer_blocks.1.attn.'
    np.savez_compressed(
        os.path.join(HERE, 'spformer_cr10_4000.npz'),
        logits=out.detach().numpy().astype(np.float32), loss=np.float32(loss3.item()),
        grad_tq=grads[blk + 'relative_pos_query_table'].numpy(),
        grad_tv_sphere=grads[blk + 'relative_pos_value_table_sphere'].numpy(),
        grad_qkv=grads[blk + 'qkv.weight'].numpy())
    with open(os.path.join(HERE, 'spformer_cr10_keys.json'), 'w') as f:
        json.dump({k: list(v.shape) for k, v in ref.state_dict().items()}, f, indent=0)
    make_kd_golden(crit, seeds=FIRST_KD_SEEDS)
    print('golden written:', float(loss), float(l2), float(loss3))


import contextlib


@contextlib.contextmanager
def shifted_quantisers(k):
    """SphereFormer's hard quantisers fed with inputs moved k units in the last place: cart2sphere's two atan2-derived
    angles and the logarithm exponential_split floors (the CPU and GPU libm differ there).  A scene whose outputs
    survive +-4 holds no token within 4 ulp of a window / relative-position bin edge, so its fixture can be held to the
    north-star 1e-3 without exception."""
    import core.models.sphereformer.spherical_transformer as ST
    from oracle import sptr_ref
    real_c2s, real_split = ST.cart2sphere, ST.exponential_split

    def c2s(xyz):
        o = real_c2s(xyz)
        ang = o[:, :2]
        for _ in range(abs(k)):
            ang = torch.nextafter(ang, torch.full_like(ang, float('inf') if k > 0 else -float('inf')))
        return torch.cat([ang, o[:, 2:]], 1)
    if k:
        ST.cart2sphere = c2s
        ST.exponential_split = lambda xyz, i0, i1, rpi, a=0.05 * 0.25: sptr_ref.exponential_split(xyz, i0, i1, rpi, a, _log_ulps=k)
    try:
        yield
    finally:
        ST.cart2sphere, ST.exponential_split = real_c2s, real_split


def kd_inputs(b):
    """numpy KD batch -> (student in_mod, teacher in_mod, extras) as torch CPU tensors, with the
    reference's layouts (images already permuted to [B, ncam, 3, H, W] like _prepare_input does)."""
    s, t = b['student'], b['teacher']
    stu = {'lidar': ots.SparseTensor(torch.from_numpy(s['feats']), torch.from_numpy(s['coords'])),
           'images': torch.from_numpy(s['images']).permute(0, 1, 4, 2, 3).contiguous(),
           'pixel_coordinates': [torch.from_numpy(c) for c in s['pixel_coordinates']],
           'masks': [torch.from_numpy(m) for m in s['masks']], 'fov_mask': torch.from_numpy(s['fov_mask'])}
    tea = {'lidar': ots.SparseTensor(torch.from_numpy(t['feats']), torch.from_numpy(t['coords']))}
    return stu, tea


# candidate scenes of the first KD fixture, scanned IN ORDER from the round-2 fixture's seed (77, which sits on a quantiser edge
# and is kept as the edge fixture): the first whose reference outputs do not move when the quantiser inputs move by +-4 ulp.
# A CPU-only criterion on the reference's own outputs -- no output of the implementation under test is consulted (rounds 2-3
# scanned a list that tools/diag_kd_seed.py had pre-filtered on the GPU box: selection bias).
FIRST_KD_SEEDS = tuple(range(77, 140))
SELECTION_NOTE = ('first candidate seed, scanned in order, whose reference outputs move by < 1e-4 under +-4 ulp of the quantiser '
                  'inputs (CPU only; no output of the implementation under test consulted)')


def make_kd_golden(crit, cr=1.0, cr_t=1.0, tag='kd_cr10_3000', n_vox=1500, write_keys=True, seeds=(77,), batches=None):
    """The reference's own SPVCNN_SWIFTNET18_SPFORMER_TSD_FULL (student + teacher) and the KD loss
    arithmetic of NuScenesLCTSDFullTrainer._run_step, on CPU over the oracle operators.  (cr, cr_t) =
    (1.0, 1.0): the first fixture; (1.0, 2.0) = configs/nuscenes/train/spformer_tsd_full_ours_star.yaml:32-43
    (the shipped student / teacher); (2.0, 2.0) = spformer_tsd_full_ours_star_B.yaml:34-36."""
    import torch.nn.functional as F
    from u2mkd_amd.synth import synth_kd_batch
    from oracle.spformer_ref import default_spformer_kwargs
    cfg = sys.modules['torchpack.utils.config'].configs
    cfg['model'].update({'cr': cr, 'cr_t': cr_t, 'in_channel': 4, 'in_channel_t': 4, 'imagenet_pretrain': None})
    cfg['eval'] = {'run_pix_decoder': True, 'run_align_loss': True}
    cfg['debug'] = {'debug_val': False}
    torch.Tensor.cuda = lambda self, *a, **k: self          # Feature_Fetch hard-codes .cuda() (fusion_blocks.py:271-273)
    from core.models.nuscenes.spvcnn_swiftnet18_spformer_tsd_full import SPVCNN_SWIFTNET18_SPFORMER_TSD_FULL
    kw = default_spformer_kwargs(drop_path_rate=0.0)
    for k in ('cr', 'in_channel', 'num_classes'):
        kw.pop(k)
    model = O.fill_state_by_name(SPVCNN_SWIFTNET18_SPFORMER_TSD_FULL(**kw)).train()
    model.model_t.eval()
    model.model_s.dropout.p = 0.0
    # fixture scene: the first candidate seed whose student / teacher outputs do not move when the quantiser inputs
    # move by +-4 ulp (one candidate = the seed is taken as it is: the round-2 fixtures at the shipped widths, which
    # hold the strict gate on the GPU)
    # ``batches``: [(seed, numpy KD batch)] candidates from another source (the loader-fed fixture) instead of synth scenes
    for seed in (seeds if batches is None else [sd for sd, _ in batches]):
        b = synth_kd_batch(n_vox, 2, seed=seed, image_hw=(64, 112)) if batches is None else dict(batches)[seed]
        if batches is None and len(seeds) == 1:
            break

        def probe(k):
            with torch.no_grad(), shifted_quantisers(k):
                o = model({'student': kd_inputs(b)[0], 'teacher': kd_inputs(b)[1]})
            return [o['stu']['x_vox'].clone(), o['stu']['pts_feats'][0].clone(), o['t']['x_vox'].clone()]
        base = probe(0)
        worst = max(float((x - y).abs().max()) for k in (4, -4) for x, y in zip(probe(k), base))
        print(tag, 'seed', seed, 'output change under +-4 ulp of the quantiser inputs: %.3g' % worst, flush=True)
        if worst < 1e-4:
            break
    else:
        raise SystemExit('no candidate seed keeps every token 4 ulp away from the quantiser edges')
    stu, tea = kd_inputs(b)
    out = model({'student': stu, 'teacher': tea})
    s, t = b['student'], b['teacher']
    targets = torch.from_numpy(s['targets'])
    fov = torch.from_numpy(s['fov_mask'])
    inv_map = torch.from_numpy(t['inverse_map'])
    # core/nusc_trainers.py:288-345, verbatim arithmetic
    x_vox_t2s, feat_t2s = [], []
    cur_v = cur_p = 0
    for n_p, n_v, inds in zip(t['num_pts'], t['num_vox'], s['inds']):
        inv = inv_map[cur_p:cur_p + n_p]
        x_vox_t2s.append(out['t']['x_vox'][cur_v:cur_v + n_v][inv, :][torch.from_numpy(inds[0]), :])
        feat_t2s.append(out['t']['pts_feats'][0][cur_v:cur_v + n_v][inv, :][torch.from_numpy(inds[0]), :])
        cur_v += n_v
        cur_p += n_p
    x_vox_t2s, feat_t2s = torch.cat(x_vox_t2s), torch.cat(feat_t2s)
    x_vox, x_pix = out['stu']['x_vox'], out['stu']['x_pix']
    ce_vox = crit(x_vox, targets)
    ce_pix = crit(x_pix[fov], targets[fov])
    kl = torch.nn.KLDivLoss(reduction='batchmean')(F.log_softmax(x_vox, dim=1), F.softmax(x_vox_t2s.detach(), dim=1))
    feat = torch.nn.MSELoss()(out['stu']['pts_feats'][0], feat_t2s.detach())
    total = ce_vox + ce_pix + 1.0 * kl + sum(out['stu']['mse_loss']) + 1.0 * feat
    total.backward()
    g = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    np.savez_compressed(
        os.path.join(HERE, tag + '.npz'), seed=np.int64(seed), selection=np.array(SELECTION_NOTE if len(seeds) > 1 or batches is not None else 'fixed seed'),
        x_vox=x_vox.detach().numpy(), x_pix=x_pix.detach().numpy(), x_vox_t=out['t']['x_vox'].numpy(),
        mse=np.array([float(m) for m in out['stu']['mse_loss']], dtype=np.float32),
        pts_feats_s=out['stu']['pts_feats'][0].detach().numpy()[::16],
        feat_t2s=feat_t2s.numpy()[::16],
        losses=np.array([float(ce_vox), float(ce_pix), float(kl), float(feat), float(total)], dtype=np.float32),
        grad_l2c=g['model_s.l2c_fusion_blocks.1.conv1.weight'].numpy(),
        grad_c2l=g['model_s.c2l_fusion_blocks.2.conv1.weight'].numpy(),
        grad_layer2=g['model_s.pix_branch.layer2.0.conv1.weight'].numpy()[:8],
        grad_stem=g['model_s.stem.3.kernel'].numpy())
    if write_keys:
        with open(os.path.join(HERE, 'kd_cr10_keys.json'), 'w') as f:
            json.dump({k: list(v.shape) for k, v in model.state_dict().items()}, f, indent=0)
    print(tag, 'kd golden losses', [float(x) for x in (ce_vox, ce_pix, kl, feat, total)])


def make_teacher_multisweep_golden():
    """Stage-1 (teacher-only) training step of the reference on a MULTI-SWEEP scene (BASELINE.json configs[4]'s
    teacher input): the reference's own SPVCNN_SPFORMER (core/models/nuscenes/spvcnn_spformer.py)
    over the oracle operators on three aggregated sweeps, and the loss of core/spformer_trainer.py:80-83 --
    criterion(outputs['x_vox'][keyframe_mask], targets[keyframe_mask]): only key-frame voxels carry labels."""
    from oracle.spformer_ref import default_spformer_kwargs
    _, MixLovaszCrossEntropy = import_reference()
    cr = 1.0
    SPF = import_reference_spformer(cr)
    crit = MixLovaszCrossEntropy(ignore_index=0)
    kw = default_spformer_kwargs(cr=cr, drop_path_rate=0.0)
    for k in ('cr', 'in_channel', 'num_classes'):
        kw.pop(k)
    def forward(seed, k=0):
        import gc
        gc.collect()
        b = synth_batch(3000, 2, seed=seed, sweeps=3)
        feats, coords, labels, kf = (torch.from_numpy(b[x]) for x in ('feats', 'coords', 'labels', 'keyframe'))
        torch.manual_seed(0)
        import copy
        # a fresh copy of the builder's arguments per construction: the reference's constructor edits its
        # quant_size_sphere / window_size_sphere arguments in place
        ref = O.fill_state_by_name(SPF(**copy.deepcopy(kw))).train()
        ref.dropout.p = 0.0
        with shifted_quantisers(k):
            out = ref({'lidar': ots.SparseTensor(feats.clone(), coords.clone())})['x_vox']
        return ref, out, labels, kf

    for seed in range(50, 120):      # scanned in order; CPU-only criterion below (rounds 2-3: a list pre-filtered on the GPU box)
        with torch.no_grad():
            base = forward(seed)[1].clone()
            worst = max(float((forward(seed, k)[1] - base).abs().max()) for k in (4, -4))
        print('seed', seed, 'logit change under +-4 ulp of the angles: %.3g' % worst, flush=True)
        if worst < 1e-4:
            break
    else:
        raise SystemExit('no seed keeps every token 4 ulp away from the quantiser edges')
    ref, out, labels, kf = forward(seed)
    assert 0 < int(kf.sum()) < len(kf)
    loss = crit(out[kf], labels[kf])
    loss.backward()
    grads = {n: p.grad for n, p in ref.named_parameters()}
    blk = 'transformer_blocks.0.attn.'
    np.savez_compressed(
        os.path.join(HERE, 'teacher_multisweep_cr10_6000.npz'),
        logits=out.detach().numpy().astype(np.float32), loss=np.float32(loss.item()), n_keyframe=np.int64(kf.sum()),
        seed=np.int64(seed), selection=np.array(SELECTION_NOTE), grad_stem=grads['stem.3.kernel'].numpy(), grad_cls=grads['classifier_vox.0.weight'].numpy(),
        grad_tk=grads[blk + 'relative_pos_key_table'].numpy(), grad_up3=grads['vox_ups.3.1.1.net.3.kernel'].numpy()[13])
    print('teacher multi-sweep golden: loss', float(loss), 'key-frame voxels', int(kf.sum()), 'of', len(kf))




def make_kd_eval_golden():
    """Rows f2 of SURVEY.md 8f: the reference's own KD model class in eval mode (BatchNorm running statistics,
    `debug_val` = the teacher evaluated as well) on the scene of the first KD fixture, and the arithmetic of the eval
    branch of NuScenesLCTSDFullTrainer._run_step (core/nusc_trainers.py:367-418): per-point predictions of the voxel
    head, the pixel head and the teacher.  Stored: the three logit matrices and the three prediction vectors."""
    from u2mkd_amd.synth import synth_kd_batch, synth_eval_feed as eval_feed
    from oracle.spformer_ref import default_spformer_kwargs
    import_reference()
    import_reference_spformer(1.0)
    cfg = sys.modules['torchpack.utils.config'].configs
    cfg['model'].update({'cr': 1.0, 'cr_t': 1.0, 'in_channel': 4, 'in_channel_t': 4, 'imagenet_pretrain': None})
    cfg['eval'] = {'run_pix_decoder': True, 'run_align_loss': True}
    cfg['debug'] = {'debug_val': True}
    torch.Tensor.cuda = lambda self, *a, **k: self
    from core.models.nuscenes.spvcnn_swiftnet18_spformer_tsd_full import SPVCNN_SWIFTNET18_SPFORMER_TSD_FULL
    kw = default_spformer_kwargs(drop_path_rate=0.0)
    for k in ('cr', 'in_channel', 'num_classes'):
        kw.pop(k)
    model = O.fill_state_by_name(SPVCNN_SWIFTNET18_SPFORMER_TSD_FULL(**kw), conv2d_he=True).eval()
    seed = int(np.load(os.path.join(HERE, 'kd_cr10_3000.npz'))['seed'])       # a scene off the quantiser edges
    b = synth_kd_batch(1500, 2, seed=seed, image_hw=(64, 112))
    # the loader's image normalisation (lc_semantic_nusc_tsd_full.py: /255, ImageNet mean / std); with running
    # statistics instead of batch statistics the raw 0..255 range overflows fp32 in the randomly initialised ResNet
    b['student']['images'] = ((b['student']['images'] / 255.0 - 0.45) / 0.225).astype(np.float32)
    stu, tea = kd_inputs(b)
    with torch.no_grad():
        outputs = model({'student': stu, 'teacher': tea})
    for k in ('x_vox', 'x_pix'):
        assert bool(torch.isfinite(outputs['stu'][k]).all()), k
    f = eval_feed(b, seed)
    s_c, t_c = stu['lidar'].C, tea['lidar'].C
    inv_t = b['teacher']['inverse_map']
    o_vox, o_pix, o_t = [], [], []
    for idx in range(int(f['s_inverse_batch'].max()) + 1):                      # nusc_trainers.py:375-388, 400-409
        cur_scene_pts = (s_c[:, -1] == idx).numpy()
        cur_inv = f['s_inverse_map'][f['s_inverse_batch'] == idx]
        o_vox.append(outputs['stu']['x_vox'][cur_scene_pts][cur_inv].argmax(1))
        o_pix.append(outputs['stu']['x_pix'][cur_scene_pts][cur_inv].argmax(1))
        cur_scene_pts = (t_c[:, -1] == idx).numpy()
        cur_inv = inv_t[f['t_inverse_batch'] == idx]
        o_t.append(outputs['t']['x_vox'][cur_scene_pts][cur_inv].argmax(1))
    np.savez_compressed(
        os.path.join(HERE, 'kd_eval_cr10_3000.npz'), seed=np.int64(seed),
        x_vox=outputs['stu']['x_vox'].numpy(), x_pix=outputs['stu']['x_pix'].numpy(), x_vox_t=outputs['t']['x_vox'].numpy(),
        outputs_vox=torch.cat(o_vox).numpy(), outputs_pix=torch.cat(o_pix).numpy(), outputs_vox_t=torch.cat(o_t).numpy())
    print('kd eval golden: points', len(f['s_inverse_map']), 'teacher points', len(inv_t))

def make_loader_kd_golden():
    """Row f1 end to end: the batch comes from u2mkd_amd/data/nuscenes_lc.py (the synthetic on-disk tree of
    tests/nusc_tree.py: validation split, two samples, six cameras, images at 72 x 128, no augmentation -> the same bytes in
    the GPU test), the expected outputs from the reference's own KD model class at small widths (cr = cr_t = 0.5).  The
    tree's seed is the first of 0, 1, ... whose scene keeps every token 4 ulp away from the quantiser edges -- a CPU-only
    criterion, no implementation output is consulted."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from nusc_tree import build_tree
    from u2mkd_amd.data import nuscenes_lc as D
    _, _Crit = import_reference()
    import_reference_spformer(0.5)
    cands = []
    for tree_seed in range(4):
        with tempfile.TemporaryDirectory() as root:
            root, ver = build_tree(root, seed=tree_seed)
            ds = D.LCNuScenesDataset(D.NuScenesTables(root, ver), split='val', im_cr=0.08)
            cands.append((tree_seed, D.collated_to_kd_batch(D.collate_fn([ds[0], ds[1]]))))
    make_kd_golden(_Crit(ignore_index=0), cr=0.5, cr_t=0.5, tag='kd_loader_cr05', write_keys=False, batches=cands)


def main_kd_widths():
    """Only the KD fixtures at the shipped widths (`python tests/golden/make_golden.py kd`)."""
    _, MixLovaszCrossEntropy = import_reference()
    import_reference_spformer(1.0)
    crit = MixLovaszCrossEntropy(ignore_index=0)
    make_kd_golden(crit, cr=1.0, cr_t=2.0, tag='kd_cr10_t20_2000', n_vox=1000, write_keys=False)
    make_kd_golden(crit, cr=2.0, cr_t=2.0, tag='kd_cr20_t20_2000', n_vox=1000, write_keys=False)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'kd':
        main_kd_widths()
    elif len(sys.argv) > 1 and sys.argv[1] == 'kd_first':
        _, _Crit = import_reference()
        import_reference_spformer(1.0)
        make_kd_golden(_Crit(ignore_index=0), seeds=FIRST_KD_SEEDS)
    elif len(sys.argv) > 1 and sys.argv[1] == 'kd_eval':
        make_kd_eval_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == 'kd_loader':
        make_loader_kd_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == 'teacher_ms':
        make_teacher_multisweep_golden()
    else:
        main()
        main_kd_widths()
        make_teacher_multisweep_golden()
        make_kd_eval_golden()
        make_loader_kd_golden()
