"""Upper bounds for step-level changes, measured instead of estimated: the pipelined KD step (fresh batches, geometry
prefetch) with parts of the work switched off.  One variant per process (env / argv), wall ms per step printed.
  python tools/exp_step_bounds.py [no_cam_wgrad] [no_pix_decoder] [half_res_tail] [cached_plans] [no_cam_backward]"""
import sys, time, os; sys.path.insert(0, '.')
import torch
from u2mkd_amd import lidar, train as T, kd as KD
from u2mkd_amd.synth import synth_kd_batch

flags = set(sys.argv[1:])
torch.manual_seed(0)
sp = {k: v for k, v in lidar.spformer_kwargs().items() if k not in ('cr', 'in_channel', 'num_classes')}
model = KD.TSDFull(cr=1.0, cr_t=2.0, in_channel=4, in_channel_t=4, num_classes=17, spformer=sp,
                   run_pix_decoder='no_pix_decoder' not in flags).cuda()
if 'no_cam_wgrad' in flags:        # camera convolutions without weight gradients (their data gradients stay)
    for n, p in model.model_s.pix_branch.named_parameters():
        if p.dim() == 4:
            p.requires_grad_(False)
if 'no_cam_backward' in flags:       # the camera ENCODER's features enter the fusion detached: no gradient flows back into the ResNet layers
    pb = model.model_s.pix_branch     # (its backward chain -- MIOpen input gradients interlocked with the LiDAR branch at every fusion stage -- disappears)
    for name in ('layer1', 'layer2', 'layer3', 'layer4'):
        layer = getattr(pb, name, None)
        if layer is not None:
            real = layer.forward
            layer.forward = (lambda f: (lambda *a, **k: _detach(f(*a, **k))))(real)


def _detach(o):
    if isinstance(o, torch.Tensor):
        return o.detach()
    if isinstance(o, (tuple, list)):
        return type(o)(_detach(v) for v in o)
    return o


if 'half_res_tail' in flags:      # the pixel head evaluated on the H/2 map (no final x2 up-sampling): what the full-resolution tail costs
    pb = model.model_s.pix_branch
    fu = pb.forward_up
    pb.forward_up = lambda feats, im_size=None: fu(feats, im_size=None)
if 'no_sptr' in flags or 'no_sptr_student' in flags:      # window attention replaced by its value rows (differentiable, ~free): the upper bound of a faster attention kernel
    from u2mkd_amd import sptr as _sp
    from u2mkd_amd.lidar import sphereformer as _sf
    _real_pwa = _sp.packed_window_attention

    def _cheap(qkv, scale, branches):
        if 'no_sptr_student' in flags and not torch.is_grad_enabled():
            return _real_pwa(qkv, scale, branches)                 # (the frozen teacher keeps the real kernels)
        return qkv[:, 2].reshape(qkv.shape[0], -1) * 1.0
    _sp.packed_window_attention = _cheap
    _sf.sptr.packed_window_attention = _cheap
if 'no_lidar_bn' in flags:      # BatchNorm over voxel / point rows as identity (+ ReLU): the upper bound of folding it into its neighbours
    from u2mkd_amd.torchsparse.nn import functional as _spf
    _real_bn = _spf.batch_norm

    def _bn(x, bn, relu=False, residual=None):
        if not bn.training:
            return _real_bn(x, bn, relu, residual)
        y = x if residual is None else x + residual
        return torch.relu(y) if relu else y * 1.0
    _spf.batch_norm = _bn
    from u2mkd_amd.lidar import blocks as _bl
    _bl.spf.batch_norm = _bn
if 'no_sparse_wgrad' in flags:      # sparse-conv and point-linear weights frozen: their weight-gradient kernels (side stream) disappear
    for n, p in model.model_s.named_parameters():
        if 'pix_branch' not in n and (n.endswith('.kernel') or (n.endswith('.weight') and p.dim() == 2)):
            p.requires_grad_(False)
if 'no_teacher' in flags:           # the frozen teacher's outputs cached per resident scene: its launches and its GPU work disappear
    _real_t = model.model_t.forward
    _cache = {}

    def _t(in_mod):
        key = int(in_mod['lidar'].F.shape[0]) * 1000003 + int(in_mod['lidar'].C.shape[0])
        if key not in _cache or len(_cache) <= 4 and _cache[key][1] < 2:
            out = _real_t(in_mod)
            _cache[key] = (out, _cache.get(key, (None, 0))[1] + 1)
        return _cache[key][0]
    model.model_t.forward = _t
run = T.KDStep(model, num_epochs=50, batch_size=1)
run.train_mode()
res = [T.kd_batch_to_device(synth_kd_batch(80000, 1, seed=1234 + i, image_hw=(360, 640))) for i in range(4)]
if 'no_pix_decoder' in flags:
    orig = KD.kd_losses
    def losses(out, targets, fov, *a, **k):
        out['stu']['x_pix'] = out['stu']['x_vox']          # the loss arithmetic stays, the decoder's work goes
        return orig(out, targets, fov, *a, **k)
    KD.kd_losses = losses


def fresh(i):
    d = T.fresh_batch(res[i % 4])
    if 'cached_plans' in flags:      # the point<->pixel plans hang on masks[0]: keep the resident tensors -> plans are built once
        d['masks'], d['pixel_coordinates'], d['fov_mask'] = res[i % 4]['masks'], res[i % 4]['pixel_coordinates'], res[i % 4]['fov_mask']
    return d


def loop(steps):
    cur = fresh(0)
    for i in range(steps):
        nxt = fresh(i + 1)
        run(cur, prefetch=nxt)
        cur = nxt


loop(6)
torch.cuda.synchronize()
t0 = time.perf_counter()
loop(16)
torch.cuda.synchronize()
print('VARIANT %-40s %.2f ms/step' % (' '.join(sorted(flags)) or 'baseline', (time.perf_counter() - t0) / 16 * 1e3), flush=True)
