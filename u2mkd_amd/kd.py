"""Uni-to-multi-modal knowledge distillation: the multi-modal student, the teacher/student
wrapper and the KD training losses (rows a12-a18 of SURVEY.md §8a).

Reference: core/models/nuscenes/spvcnn_swiftnet18_spformer_tsd_full.py:197-596 (student
``SPVCNN_SWIFTNET18_SPFORMER_MSP2IFM``, wrapper ``..._TSD_FULL``) and the train branch of
``NuScenesLCTSDFullTrainer._run_step`` (core/nusc_trainers.py:255-366).  Hyper-parameters that
the reference reads from the global torchpack ``configs`` are explicit arguments here.
Module / parameter names equal the reference's (checkpoint compatible)."""
from copy import deepcopy

import os

import torch
import torch.nn.functional as F
from torch import nn

from . import torchsparse
from .camera import BNReluConv, SwiftNetRes18
from .graphs import PieceCache, StaticPiece
from .fusion import Atten_Fusion_Conv, L2CFusion, c2l_gather, feature_fetch, l2c_scatter
from .pixel_head import sampled_head_applies, sampled_pixel_logits
from .lidar.blocks import (BasicConvolutionBlock, BasicDeconvolutionBlock, FusedSequential, PointBatchNorm1d, PointLinear,
                           ResidualBlock)
from .lidar.point_voxel import (initial_voxelize, point_to_voxel, prepare_geometry, prepare_geometry_many, prepare_geometry_staged,
                                voxel_to_point)
from .torchsparse.nn import functional as spf
from .lidar.sphereformer import SphereFormer
from .lidar.spvcnn_spformer import SPVCNN_SPFORMER
from .losses import MixLovaszCrossEntropy, kl_div_logits
from .torchsparse import PointTensor
from .torchsparse import nn as spnn

__all__ = ['StudentMSP2IFM', 'TSDFull', 'teacher_to_student', 'kd_losses', 'KDCriterion']

_BASE_CHANNELS = (32, 32, 64, 128, 256, 256, 128, 96, 96)


class StudentMSP2IFM(nn.Module):
    """SPVCNN + SphereFormer LiDAR branch fused at four scales with a SwiftNet-18 camera branch
    (``SPVCNN_SWIFTNET18_SPFORMER_MSP2IFM``, tsd_full.py:197-559)."""

    def __init__(self, cr, in_channel, num_classes, window_size, window_size_sphere, quant_size, quant_size_sphere,
                 window_size_scale, drop_path_rate, a, pres, vres, imagenet_pretrain=None, run_pix_decoder=True):
        super().__init__()
        cs = [int(cr * c) for c in _BASE_CHANNELS]
        self.cs = cs
        self.pix_branch = SwiftNetRes18(num_feature=(128, 128, 128), pretrained_path=imagenet_pretrain)
        img_cs = self.pix_branch.img_cs
        self.in_channel, self.num_classes, self.out_channel = in_channel, num_classes, cs[-1]
        self.pres, self.vres = pres, vres
        self._fusion_shapes = {}      # {(image height, width): {fusion point: (map height, map width, l2c scales)}}, seen by forward

        self.stem = FusedSequential(
            spnn.Conv3d(in_channel, cs[0], kernel_size=3, stride=1), spnn.BatchNorm(cs[0]), spnn.ReLU(True),
            spnn.Conv3d(cs[0], cs[0], kernel_size=3, stride=1), spnn.BatchNorm(cs[0]), spnn.ReLU(True))
        self.vox_downs = nn.ModuleList([
            nn.Sequential(BasicConvolutionBlock(cs[i], cs[i], ks=2, stride=2, dilation=1),
                          ResidualBlock(cs[i], cs[i + 1], ks=3, stride=1, dilation=1),
                          ResidualBlock(cs[i + 1], cs[i + 1], ks=3, stride=1, dilation=1))
            for i in range(4)])

        self.window_size, self.window_size_sphere = window_size, window_size_sphere
        self.quant_size, self.quant_size_sphere = quant_size, quant_size_sphere
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, 7)]
        self.transformer_blocks = nn.ModuleList()
        for idx in range(1, 5):
            self.transformer_blocks.append(SphereFormer(
                cs[idx], cs[idx] // 16, self.window_size, self.window_size_sphere, self.quant_size,
                self.quant_size_sphere, indice_key='sphereformer{}'.format(idx + 1), drop_path=dpr[idx], a=a))
            sc, ss = window_size_scale
            self.window_size = self.window_size * sc
            self.quant_size = self.quant_size * sc
            self.window_size_sphere[0] = self.window_size_sphere[0] * ss
            self.window_size_sphere[1] = self.window_size_sphere[1] * ss
            self.quant_size_sphere[0] = self.quant_size_sphere[0] * ss      # in place
            self.quant_size_sphere[1] = self.quant_size_sphere[1] * ss

        self.c2l_fusion_blocks = nn.ModuleList(
            [Atten_Fusion_Conv(inplanes_I=img_cs[i], inplanes_P=cs[i], outplanes=cs[i]) for i in range(1, 5)])
        self.l2c_fusion_blocks = nn.ModuleList(
            [L2CFusion(inplanes_I=img_cs[i], inplanes_P=cs[i], outplanes=img_cs[i]) for i in range(1, 5)])
        self.learner = nn.ModuleList([
            FusedSequential(PointLinear(cs[i], img_cs[i]), PointBatchNorm1d(img_cs[i]), nn.ReLU(True),
                            PointLinear(img_cs[i], img_cs[i]), PointBatchNorm1d(img_cs[i]))
            for i in range(1, 5)])
        self.mse = nn.MSELoss()

        self.vox_ups = nn.ModuleList([
            nn.ModuleList([
                BasicDeconvolutionBlock(cs[i], cs[i + 1], ks=2, stride=2),
                nn.Sequential(
                    ResidualBlock(cs[i + 1] + cs[len(cs) - 2 - i], cs[i + 1], ks=3, stride=1, dilation=1),
                    ResidualBlock(cs[i + 1], cs[i + 1], ks=3, stride=1, dilation=1))])
            for i in range(4, len(cs) - 1)])
        self.classifier_vox = nn.Sequential(PointLinear(cs[8], num_classes))
        self.classifier_pix = BNReluConv(self.pix_branch.num_features, num_classes, k=1)
        self.point_transforms = nn.ModuleList([
            FusedSequential(PointLinear(cs[a_], cs[b_]), PointBatchNorm1d(cs[b_]), nn.ReLU(True))
            for a_, b_ in ((0, 4), (4, 6), (6, 8))])
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self.dropout = nn.Dropout(0.3, True)
        self.run_pix_decoder = run_pix_decoder
        self.adapt_layer = None          # attached by TSDFull (tsd_full.py:576-580)

    def _camera_stage_eager(self, x_in, idx):
        x_o, skip_o = self.pix_branch.forward_resblock(x_in, getattr(self.pix_branch, 'layer%d' % (idx + 1)))
        if idx == len(self.vox_downs) - 1:
            skip_o = self.pix_branch.spp(skip_o)
        return x_o, skip_o

    def _piece(self, name):
        """The image-shaped (static-shape) pieces of the camera side as hipGraph-replayable callables
        (graphs.StaticPiece; eager whenever a call does not qualify).  Built on first use; not modules."""
        pieces = self.__dict__.setdefault('_pieces', PieceCache())
        if name not in pieces:
            pb, last = self.pix_branch, len(self.vox_downs) - 1
            if name == 'head':
                fn, mods = (lambda im: self._camera_stage_eager(pb.forward_stem(im), 0)), [pb.conv1, pb.bn1, pb.layer1]
            elif name.startswith('stage'):
                idx = int(name[5:])
                fn = lambda x_in, idx=idx: self._camera_stage_eager(x_in, idx)                    # noqa: E731
                mods = [getattr(pb, 'layer%d' % (idx + 1))] + ([pb.spp] if idx == last else [])
            elif name.startswith('l2c'):
                blk = self.l2c_fusion_blocks[int(name[3:])]
                fn, mods = (lambda fmap, skip: blk(fmap, skip)), [blk]
            elif name == 'decoder_low':
                # the decoder up to its last blend (H/2 x W/2): the tail is evaluated at the sampled pixels (pixel_head.py)
                fn, mods = (lambda *feats: pb.forward_up(list(feats), im_size=None)), [pb.upsample]
            else:
                # the up-sampling target is part of the piece's NAME: two image sizes with equal feature-map shapes
                # (H = 359 and 360 both give H/2 = 180) must not share a captured graph
                assert name.startswith('decoder_'), name
                im_size = tuple(int(v) for v in name[8:].split('x'))
                fn = lambda *feats: self.classifier_pix(pb.forward_up(list(feats), im_size=im_size))   # noqa: E731
                mods = [pb.upsample, self.classifier_pix]
            pieces[name] = StaticPiece(name, fn, mods)
        return pieces[name]

    def _camera_stage(self, x_in, idx):
        return self._piece('head')(x_in) if idx == 0 else self._piece('stage%d' % idx)(x_in)

    def camera_head(self, in_mod):
        """SwiftNet stem + layer1 (everything of the camera branch ahead of the first fusion point) queued on the
        camera stream; returns (x_im, skip) for ``forward``."""
        im = in_mod['images']
        im = im.reshape(-1, *im.shape[2:])
        return _Fork(im, 'camera', _CAMERA_STREAM and _overlap_ok()).on_side(
            lambda: self._camera_stage(im, 0), im)

    @staticmethod
    def _tokens(coords, stride, zz):
        """(positions [n, 3], batch index [n]) of the attention tokens of the voxels ``coords``: the mean of the stride-1 voxel
        features' first three columns per voxel (tsd_full.py: ``point_to_voxel(vox_out, zz)``) -- a function of the input batch
        only.  (One copy: the plan kernels need contiguous rows.)"""
        from .lidar.point_voxel import p2v_maps
        idx_query, counts = p2v_maps(coords, stride, zz)
        return spf.spvoxelize(zz.F, idx_query, counts)[:, :3].contiguous(), coords[:, 3]

    @torch.no_grad()
    def prefetch_plans(self, in_mod):
        """Every index structure of this network's forward and backward that depends on the input batch only, built NOW on the
        current stream from the prepared geometry ``in_mod['_geometry']``: the point <-> voxel maps of every stride with their
        scatter / gather plans (core/models/utils.py:40-118 builds them at first use), the attention tokens' positions and the
        window plans of the four SphereFormer blocks, the point <-> pixel plans of the four fusion points.  In the forward they
        are ~350 small launches on the student's chain (tools/main_chain_calls.py); a trainer that prepares batch k + 1 while
        step k runs (train.KDStep) builds them there instead, and ``forward`` finds them: in the caches the lazy path fills
        (``z.idx_query``, ``z.additional_features``, spf._plan), or in ``x0._u2mkd_kd_plans`` for what the forward derives from
        tensors of its own.  Same kernels, same inputs, same results; a structure that is not found is built lazily as ever."""
        from .fusion import prefetch_plans as fusion_plans
        from .lidar.point_voxel import p2v_maps, v2p_maps
        z, x0 = in_mod['_geometry']
        if x0.C.shape[0] == 0:
            return
        zz = PointTensor(x0.F, x0.C.float())
        n_stage = len(self.vox_downs)
        strides = [x0.s] + [tuple(v * 2 ** (i + 1) for v in x0.s) for i in range(n_stage)]
        if any(x0.cmaps.get(s3) is None for s3 in strides):
            return
        tokens = []
        parts = spf._PREFETCH_PARTS
        for i, s3 in enumerate(strides):
            coords = x0.cmaps[s3]
            if 'A' in parts:
                v2p_maps(coords, s3, z, plans=True)
                p2v_maps(coords, s3, z, plans=True)
            if i > 0 and 'C' in parts:
                p2v_maps(coords, s3, zz, plans=True)
                tok = self._tokens(coords, s3, zz)
                self.transformer_blocks[i - 1].attn.plans(tok[0].float(), tok[1], quantised=True)
                tokens.append(tok)
        if 'C' in parts:
            x0.__dict__['_u2mkd_kd_plans'] = {'zz': zz, 'tokens': tokens}
        im = in_mod['images']
        shapes = self._fusion_shapes.get((int(im.shape[-2]), int(im.shape[-1])))
        if shapes is not None and len(shapes) == n_stage and 'D' in parts:
            fusion_plans(in_mod['pixel_coordinates'], in_mod['masks'], [shapes[i] for i in range(n_stage)])

    def forward(self, in_mod):
        x = in_mod['lidar']
        im = in_mod['images']                                   # [B, ncam, 3, H, W]
        ib, ncam, ic, ih, iw = im.shape
        im = im.reshape(-1, ic, ih, iw)
        pixel_coordinates, masks, fov_mask = in_mod['pixel_coordinates'], in_mod['masks'], in_mod['fov_mask']
        # The camera branch is independent of the LiDAR branch between two fusion points (stem + layer1 before the first
        # one, one ResNet layer per stage after, the decoder at the end), and it is a few LARGE MIOpen kernels where
        # the LiDAR branch is hundreds of small ones with host synchronisations in between (the voxel-set sizes):
        # it runs on a side HIP stream, queued BEFORE the LiDAR work of the same stage, so the GPU has the camera
        # kernels to run while the host waits for a torch.unique; autograd replays the same streams in the backward
        # (MIOpen's backward next to the sparse-conv backward).  U2MKD_CAMERA_STREAM=0: everything on one stream.
        fork = _Fork(im, 'camera', _CAMERA_STREAM and _overlap_ok())
        on_side, join = fork.on_side, fork.join
        n_stage = len(self.vox_downs)
        cam_stage = self._camera_stage
        cam = in_mod.get('_camera_head')          # queued already by TSDFull (ahead of the teacher's forward)
        if cam is None:
            cam = self.camera_head(in_mod)
        # points -> stride-1 voxels + all kernel maps (every down-sample sync), prepared ahead by the trainer or here
        z, x0 = in_mod.get('_geometry') or prepare_geometry(x, self.pres, self.vres)
        ahead = x0.__dict__.get('_u2mkd_kd_plans')      # (prefetch_plans: this batch's index structures, built a step ahead)
        zz = ahead['zz'] if ahead is not None else PointTensor(x0.F, x0.C.float())
        x0 = self.stem(x0)
        z0 = voxel_to_point(x0, z, nearest=False)
        vox_feats = [point_to_voxel(x0, z0)]
        img_feats, mse_loss, pts_feats = [], [], []
        x_im_ready = None
        for idx in range(n_stage):
            if idx > 0:
                # queued ahead of this stage's LiDAR kernels, ordered behind the L2C block that made x_im -- NOT behind the
                # camera -> LiDAR half of the previous fusion point that the main stream has queued since (gather, learner,
                # fusion MLPs, point_to_voxel: none of it feeds the camera branch; U2MKD_CAMERA_FORK_EARLY=0: behind all of it)
                cam = on_side(lambda: cam_stage(x_im, idx), x_im, after=x_im_ready if _CAMERA_FORK_EARLY else None)
            vox_out = self.vox_downs[idx](vox_feats[idx])
            if ahead is not None:
                coord_xyz, batch = ahead['tokens'][idx]
            else:
                coord_xyz, batch = self._tokens(vox_out.C, vox_out.s, zz)
            vox_out.F = self.transformer_blocks[idx](vox_out.F, coord_xyz, batch)
            pts_feat = voxel_to_point(vox_out, z0)
            if idx == n_stage - 1:
                pts_feats.append(self.adapt_layer(pts_feat.F))

            x_im, skip = cam
            join(x_im, skip)
            _, ifc, ifh, ifw = skip.shape
            self._fusion_shapes.setdefault((ih, iw), {})[idx] = (ifh, ifw, n_stage - idx)

            # LiDAR -> camera: multi-scale scatter-mean of the point features into every camera's map
            l2c_feat_map = l2c_scatter(pts_feat.F, pixel_coordinates, masks, ifh, ifw, n_stage - idx)
            x_im, skip = self._piece('l2c%d' % idx)(l2c_feat_map, skip)
            x_im_ready = fork.mark()
            img_feats.append(skip)

            # camera -> LiDAR: bilinear gather, later cameras overwrite; points no camera sees take the
            # learner's pseudo image feature (not detached), the MSE target is detached (Appendix C-5)
            img_feat_tensor = c2l_gather(skip.view(ib, ncam, ifc, ifh, ifw), pixel_coordinates, masks)
            pseudo = self.learner[idx](pts_feat.F)
            img_feat_tensor, mse = _select_and_mse(img_feat_tensor, pseudo, fov_mask)
            mse_loss.append(mse)
            pts_feat.F = self.c2l_fusion_blocks[idx](pts_feat.F, img_feat_tensor)
            vox_feats.append(point_to_voxel(vox_out, pts_feat))

        fused = fork.mark()           # the pixel decoder's inputs (img_feats) are complete here
        _, x1, x2, x3, x4 = vox_feats
        z1 = pts_feat
        z1.F = z1.F + self.point_transforms[0](z0.F)
        y1 = point_to_voxel(x4, z1)
        y1.F = self.dropout(y1.F)
        y1 = self.vox_ups[0][0](y1)
        y1 = self.vox_ups[0][1](torchsparse.cat([y1, x3]))
        y2 = self.vox_ups[1][0](y1)
        y2 = self.vox_ups[1][1](torchsparse.cat([y2, x2]))
        z2 = voxel_to_point(y2, z1)
        z2.F = z2.F + self.point_transforms[1](z1.F)
        y3 = point_to_voxel(y2, z2)
        y3.F = self.dropout(y3.F)
        y3 = self.vox_ups[2][0](y3)
        y3 = self.vox_ups[2][1](torchsparse.cat([y3, x1]))
        y4 = self.vox_ups[3][0](y3)
        y4 = self.vox_ups[3][1](torchsparse.cat([y4, x0]))
        z3 = voxel_to_point(y4, z2)
        z3.F = z3.F + self.point_transforms[2](z2.F)

        ret = {'x_vox': self.classifier_vox(z3.F), 'num_pts': [c.shape[1] for c in pixel_coordinates],
               'mse_loss': mse_loss, 'pts_feats': pts_feats}
        # The pixel decoder (camera side) runs next to the voxel decoder (LiDAR side) on the camera stream.  It is QUEUED
        # after the voxel decoder on purpose: autograd issues the backward in reverse creation order, so the decoder's
        # backward -- the head of the camera branch's backward chain, which every fusion stage's gradient waits for --
        # is the first thing the host issues when the backward starts, instead of following the voxel decoder's
        # few hundred launches (measured: the camera stream sat idle for the first ~6 ms of every backward).  In the
        # forward the order costs nothing: the camera stream only depends on img_feats, and in the pipelined steady
        # state the host issues the forward well ahead of the GPU (the side stream waits for the event `fused`, not for
        # the voxel decoder).
        if self.run_pix_decoder:
            sampled = sampled_head_applies(img_feats[0], self.classifier_pix)

            def pix_decoder():
                if sampled:
                    # up-sampling to the image size + BatchNorm + ReLU + classifier only at the pixels Feature_Fetch reads
                    fmap = self._piece('decoder_low')(*img_feats)
                    return sampled_pixel_logits(fmap, self.classifier_pix, pixel_coordinates, masks, (ih, iw), ib, ncam)
                if ib * ncam * self.pix_branch.num_features * ih * iw * 4 >= 2 ** 32:
                    # MIOpen's weight gradient of the 1x1 classifier is wrong once its input passes 4 GiB (6 x 128 x 900 x
                    # 1600 floats; tools/dbg_miopen_large.py): no silent wrong gradients
                    raise RuntimeError('the dense pixel head would form a %.1f GB tensor, beyond what MIOpen computes '
                                       'correctly: use the sampled head (U2MKD_SAMPLED_PIXEL_HEAD=1, the default)'
                                       % (ib * ncam * self.pix_branch.num_features * ih * iw * 4 / 1e9))
                fmap = self._piece('decoder_%dx%d' % (ih, iw))(*img_feats)
                fmap = fmap.view(ib, ncam, fmap.shape[1], fmap.shape[2], fmap.shape[3])
                return feature_fetch(masks, pixel_coordinates, fmap)
            x_pix = on_side(pix_decoder, *img_feats, after=fused)
            join(x_pix)
            ret['x_pix'] = x_pix
        return ret


class _SelectMSE(torch.autograd.Function):
    """(where(fov, gathered, pseudo), MSELoss()(pseudo[fov], gathered[fov].detach())) of one fusion stage in one pass
    each way (csrc/fusion.hip: u2mkd_select_mse_forward / _backward); the torch formulation below is ~30 element-wise
    launches per stage over [N, C]."""

    @staticmethod
    def forward(ctx, gathered, pseudo, fov):
        from . import _lib as L
        n, c = pseudo.shape
        gathered, pseudo = gathered.contiguous(), pseudo.contiguous()
        fov8 = fov.contiguous().view(torch.uint8)
        out = torch.empty_like(pseudo)
        partial = torch.empty(2 * L.load().u2mkd_select_mse_partials(), dtype=torch.float32, device=pseudo.device)
        stats = torch.empty(2, dtype=torch.float32, device=pseudo.device)
        L.call('u2mkd_select_mse_forward', L.ptr(gathered), L.ptr(pseudo), L.ptr(fov8), n, c, L.ptr(out), L.ptr(partial),
               L.ptr(stats), L.stream())
        ctx.save_for_backward(gathered, pseudo, fov8, stats)
        return out, stats[0]

    @staticmethod
    def backward(ctx, g_out, g_loss):
        from . import _lib as L
        gathered, pseudo, fov8, stats = ctx.saved_tensors
        n, c = pseudo.shape
        g_out = g_out.contiguous() if g_out is not None else None
        g_loss = g_loss.contiguous().float() if g_loss is not None else None
        d_g = torch.empty_like(gathered) if ctx.needs_input_grad[0] else None
        d_p = torch.empty_like(pseudo)
        L.call('u2mkd_select_mse_backward', L.ptr(g_out), L.ptr(g_loss), L.ptr(stats), L.ptr(gathered), L.ptr(pseudo),
               L.ptr(fov8), n, c, L.ptr(d_g), L.ptr(d_p), L.stream())
        return d_g, d_p, None


def _select_and_mse(gathered, pseudo, fov):
    """(img_feat_tensor, mse) of tsd_full.py:489-498: the learner's pseudo feature where no camera sees the point, and the
    MSE between pseudo feature and (detached) gathered feature over the points a camera does see."""
    if (pseudo.is_cuda and pseudo.dtype == torch.float32 and gathered.dtype == torch.float32 and pseudo.shape[1] % 4 == 0
            and fov.dtype == torch.bool and not torch.is_autocast_enabled()):
        return _SelectMSE.apply(gathered, pseudo, fov)
    out = torch.where(fov.unsqueeze(1), gathered, pseudo)
    return out, _masked_mse(pseudo, out.detach(), fov)


def _masked_mse(a, b, mask):
    """nn.MSELoss()(a[mask], b[mask]) without the boolean-index compaction (a host synchronisation
    per fusion stage): mean over the selected rows of the squared differences."""
    m = mask.to(a.dtype).unsqueeze(1)
    return (((a - b) ** 2) * m).sum() / (m.sum() * a.shape[1]).clamp(min=1.0)


class TSDFull(nn.Module):
    """Student + frozen teacher (``SPVCNN_SWIFTNET18_SPFORMER_TSD_FULL``, tsd_full.py:562-596).
    ``spformer`` holds the shared SphereFormer hyper-parameters (see lidar.spformer_kwargs); like
    the reference the teacher receives a deep copy (separate aliasing of quant_size_sphere)."""

    def __init__(self, cr, cr_t, in_channel, in_channel_t, num_classes, spformer: dict, imagenet_pretrain=None,
                 run_pix_decoder=True, debug_val=False):
        super().__init__()
        param_s, param_t = dict(spformer), deepcopy(spformer)
        self.model_s = StudentMSP2IFM(cr=cr, in_channel=in_channel, num_classes=num_classes,
                                      imagenet_pretrain=imagenet_pretrain, run_pix_decoder=run_pix_decoder, **param_s)
        self.model_t = SPVCNN_SPFORMER(cr=cr_t, in_channel=in_channel_t, num_classes=num_classes,
                                       return_pts_feats=True, **param_t)
        self.model_t.requires_grad_(False)
        self.num_classes = num_classes
        self.debug_val = debug_val
        self.model_s.adapt_layer = FusedSequential(
            PointLinear(self.model_s.cs[4], self.model_t.cs[4]), PointBatchNorm1d(self.model_t.cs[4]), nn.ReLU(True))

    def prepare(self, in_mod: dict):
        """Geometry of a batch for both networks (point_voxel.prepare_geometry), on the current stream: everything in a
        training step that makes the host wait for the GPU.  ``forward`` of a batch that carries the result as
        ``in_mod['student']['_geometry']`` / ``in_mod['teacher']['_geometry']`` is pure launch issue.  Called by
        train.KDStep for batch k+1 between the forward and the backward of step k: the GPU queue is short there (the
        forward is bound by the host's launch rate), so the waits are short, and the host then issues the next
        forward while step k's backward -- ~30 ms of queued kernels -- drains, instead of sitting in the next
        forward's first synchronisation for as long."""
        with torch.no_grad():
            g_s, g_t = prepare_geometry_many([(in_mod['student']['lidar'], self.model_s.pres, self.model_s.vres, 'kd_student'),
                                              (in_mod['teacher']['lidar'], self.model_t.pres, self.model_t.vres, _TEACHER_TAG)])
            in_mod['student']['_geometry'] = g_s
            in_mod['teacher']['_geometry'] = g_t
            if spf.prefetch_plans_enabled():
                self.model_s.prefetch_plans(in_mod['student'])
        return in_mod

    def prepare_staged(self, in_mod: dict):
        """``prepare`` as a generator that yields in front of each of its two host reads (point_voxel.prepare_geometry_staged):
        train.KDStep resumes it between the phases of the CURRENT step, so the reads find their counts ready.  The caller sets
        ``torch.no_grad()`` (and its stream / autocast contexts) around every ``next()``; returns ``in_mod``."""
        g_s, g_t = yield from prepare_geometry_staged([(in_mod['student']['lidar'], self.model_s.pres, self.model_s.vres, 'kd_student'),
                                                       (in_mod['teacher']['lidar'], self.model_t.pres, self.model_t.vres, _TEACHER_TAG)])
        in_mod['student']['_geometry'] = g_s
        in_mod['teacher']['_geometry'] = g_t
        if spf.prefetch_plans_enabled():
            self.model_s.prefetch_plans(in_mod['student'])
        return in_mod

    def forward(self, in_mod: dict):
        """tsd_full.py:582-596.  The frozen teacher's forward is independent of the student's until the KD losses,
        so (training, on the GPU) it runs on a side HIP stream underneath the student's forward: both are long
        chains of kernels that rarely fill all 256 CUs on their own.  The side stream is ordered after everything
        queued so far (the teacher's inputs) and the main stream waits for it before the outputs are used;
        U2MKD_TEACHER_STREAM=0 runs the reference's sequential order."""
        want_t = self.training or self.debug_val
        side = _side_stream(in_mod['teacher']['lidar'].F, 'teacher') if want_t and _TEACHER_STREAM and _overlap_ok() else None
        if side is None:
            ret = {'stu': self.model_s(in_mod['student'])}
            if want_t:
                with torch.no_grad():
                    ret['t'] = self.model_t(in_mod['teacher'])
            return ret
        main = torch.cuda.current_stream()
        stu_in = in_mod['student']
        ahead = in_mod['teacher'].pop('_teacher_out', None)
        if ahead is not None:
            # the trainer queued this batch's teacher forward one step ahead (``teacher_ahead``, train.KDStep): nothing of the
            # teacher is issued here, the host goes straight to the student
            t = ahead
            ret = {'stu': self.model_s(stu_in)}
            main.wait_stream(side)
            for v in _tensors(t):
                v.record_stream(main)
            ret['t'] = t
            return ret
        if _CAMERA_STREAM and self.training and _overlap_ok():
            # the camera head first: its large kernels run while the host queues the teacher's ~1500 small ones
            stu_in = dict(stu_in, _camera_head=self.model_s.camera_head(stu_in))
        if 'teacher_after_camera' in _DEBUG_ORDER:      # (debug, NOTES N9: the teacher starts when the camera head has finished)
            side.wait_stream(_side_stream(in_mod['teacher']['lidar'].F, 'camera'))
        side.wait_stream(main)
        with torch.cuda.stream(side), torch.no_grad():
            t = self.model_t(in_mod['teacher'])
        if 'student_after_teacher' in _DEBUG_ORDER:     # (debug: the student's LiDAR branch starts when the teacher has finished)
            main.wait_stream(side)
        ret = {'stu': self.model_s(stu_in)}
        main.wait_stream(side)
        for v in _tensors(t):
            v.record_stream(main)      # allocated on the side stream, consumed (and freed) on the main one
        ret['t'] = t
        return ret


_TEACHER_STREAM = os.environ.get('U2MKD_TEACHER_STREAM', '1') != '0'
# U2MKD_TEACHER_AHEAD: the frozen teacher's forward of batch k + 1 is ISSUED by the trainer at the end of step k, behind the
# backward's launches (train.KDStep with ``prefetch=``), instead of at the start of step k + 1: its ~1 000 launches (6-7 ms of
# host time) leave the forward phase.  1 = the teacher's kernels may start as soon as its geometry is there (next to the tail of
# the backward), 2 = they additionally wait for the end of the backward's main-stream chain, 0 = off (default).  The teacher is
# frozen (no gradient, eval-mode BatchNorm): its outputs do not depend on which side of the optimizer step they are computed.
# MEASURED (round 6, same-box pairs): 63.3-63.6 ms with it against 62.7-63.1 ms without -- single steps drop to 57-58 ms, the mean
# does not move: the step is bound by the GPU's total kernel time (86 ms over five streams), not by where or when the host
# issues it (NOTES N10).  Kept as a switch: it is the measurement that settles the question.
# (U2MKD_PREFETCH_PLANS=2: the teacher's kernel-map schedules a step ahead too; they are off the student's chain either way)
_TEACHER_TAG = 'kd_teacher' if os.environ.get('U2MKD_PREFETCH_PLANS', '0') == '2' else None
_CAMERA_FORK_EARLY = os.environ.get('U2MKD_CAMERA_FORK_EARLY', '1') != '0'
_TEACHER_AHEAD = int(os.environ.get('U2MKD_TEACHER_AHEAD', '0'))


def teacher_ahead(model, in_mod_next, after=()):
    """Queue ``model.model_t``'s forward for a PREPARED next batch (``TSDFull.prepare``: its geometry is there) on the teacher's
    side stream, behind the events ``after``; the outputs ride in ``in_mod_next['teacher']['_teacher_out']`` until
    ``TSDFull.forward`` of that batch picks them up.  Returns False (nothing queued) where the step does not fork."""
    if not (_TEACHER_AHEAD and _TEACHER_STREAM and model.training and _overlap_ok()):
        return False
    tea = in_mod_next['teacher']
    if tea.get('_geometry') is None or not tea['lidar'].F.is_cuda:
        return False
    side = _side_stream(tea['lidar'].F, 'teacher')
    for ev in after:          # (the batch's tensors, its geometry; not the backward that was queued in between)
        if ev is not None:
            side.wait_event(ev)
    with torch.cuda.stream(side), torch.no_grad():
        tea['_teacher_out'] = model.model_t(tea)
    return True


def _overlap_ok():
    """(deferred.overlap_ok: no side streams while library kernels run in reduced precision -- NOTES N9)"""
    from . import deferred
    return deferred.overlap_ok()

_DEBUG_ORDER = os.environ.get('U2MKD_DEBUG_ORDER', '')      # (tools/stale_discriminators.sh only)
_CAMERA_STREAM = os.environ.get('U2MKD_CAMERA_STREAM', '1') != '0'


# Under DDP the AccumulateGrad nodes of the camera parameters are created on the default stream while their gradients
# arrive from the camera stream: intended here (autograd inserts the stream synchronisation), so the notice is off.
if _CAMERA_STREAM and hasattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch'):
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)


class _Fork:
    """fn() on a side stream ordered after the main stream's queue (`on_side`), and the way back (`join`)."""

    def __init__(self, ref, role, enabled):
        self.main = torch.cuda.current_stream() if ref.is_cuda else None
        self.side = _side_stream(ref, role) if (enabled and ref.is_cuda) else None

    def mark(self):
        """An event on the main stream: `on_side(..., after=mark)` orders the side stream behind the work queued up to
        here only, whatever the main stream receives in between."""
        if self.side is None:
            return None
        e = torch.cuda.Event()
        e.record(self.main)
        return e

    def on_side(self, fn, *inputs, after=None):
        if self.side is None:
            return fn()
        if after is not None:
            self.side.wait_event(after)
        else:
            self.side.wait_stream(self.main)
        for t in inputs:
            t.record_stream(self.side)       # allocated on the main stream, read on the side stream
        with torch.cuda.stream(self.side):
            return fn()

    def join(self, *outs):
        if self.side is not None:
            self.main.wait_stream(self.side)
            for t in outs:
                t.record_stream(self.main)   # allocated on the side stream, used (and freed) on the main one


def _side_stream(ref, role):
    """One side HIP stream per (device, role) (the registry of deferred.py: deferred gradient work borrows the teacher's)."""
    if not ref.is_cuda:
        return None
    from . import deferred
    return deferred.stream(ref.device.index if ref.device.index is not None else torch.cuda.current_device(), role)


def _tensors(o):
    if torch.is_tensor(o):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _tensors(v)
    elif isinstance(o, (list, tuple)):
        for v in o:
            yield from _tensors(v)


def teacher_to_student(x_t, inverse_map, inds_s, num_pts, num_vox_t, keyframe_mask_full=None):
    """Re-index a per-teacher-voxel tensor to the student's voxels:
    ``x_t[inv][keyframe][inds]`` per sample (core/nusc_trainers.py:288-324), as ONE gather:
    the composed index is built from the per-sample offsets, no Python-side tensor slicing."""
    return x_t.index_select(0, teacher_to_student_index(inverse_map, inds_s, num_pts, num_vox_t, keyframe_mask_full))


def teacher_to_student_index(inverse_map, inds_s, num_pts, num_vox_t, keyframe_mask_full=None):
    """The composed row index of ``teacher_to_student`` (int64 [student voxels])."""
    idx = []
    cur_v = cur_p = 0
    for n_p, n_v, inds in zip(num_pts, num_vox_t, inds_s):
        inv = inverse_map[cur_p:cur_p + n_p]
        if keyframe_mask_full is not None:
            inv = inv[keyframe_mask_full[cur_p:cur_p + n_p]]
        idx.append(inv[inds[0]] + cur_v)
        cur_v += n_v
        cur_p += n_p
    return torch.cat(idx) if len(idx) > 1 else idx[0]


class KDCriterion(nn.Module):
    """The three criteria of configs/nuscenes/train/spformer_tsd_full_ours_star.yaml:1-9."""

    def __init__(self, ignore_index=0, w_kl=1.0, w_feat=1.0, mse_norm_feat=False):
        super().__init__()
        self.lovasz = MixLovaszCrossEntropy(ignore_index=ignore_index)
        self.kl = nn.KLDivLoss(reduction='batchmean')
        self.mse = nn.MSELoss()
        self.w_kl, self.w_feat, self.mse_norm_feat = w_kl, w_feat, mse_norm_feat


def kd_losses(outputs, targets, fov_mask, inverse_map, inds_s, num_pts, num_vox_t, crit: KDCriterion,
              keyframe_mask_full=None):
    """Loss terms and total of the KD step (core/nusc_trainers.py:288-358)."""
    # (the teacher -> student row index once for both re-indexed tensors; the logits' re-indexing happens inside the KL pass)
    t2s = teacher_to_student_index(inverse_map, inds_s, num_pts, num_vox_t, keyframe_mask_full)
    feat_t2s = outputs['t']['pts_feats'][0].index_select(0, t2s)
    x_vox, x_pix = outputs['stu']['x_vox'], outputs['stu']['x_pix']
    ld = {'ce_vox': crit.lovasz(x_vox, targets),
          # the reference compacts the rows a camera sees (`x_pix[fov_mask]`, two host synchronisations right after
          # the forward); the criterion drops ignore-labelled rows itself (mask-based Lovasz, CE ignore_index), so
          # the rows outside the field of view take the ignore label instead: same value, same gradient, no sync
          'ce_pix': crit.lovasz(x_pix, torch.where(fov_mask, targets, torch.full_like(targets, crit.lovasz.ignore_index))),
          'kl': kl_div_logits(x_vox.float(), outputs['t']['x_vox'].float(), t2s, crit.kl),
          'mse': outputs['stu']['mse_loss']}
    pts_feat_s = outputs['stu']['pts_feats'][0]
    if crit.mse_norm_feat:
        def norm(f):
            f_max, f_min = f.max(-1, keepdim=True).values, f.min(-1, keepdim=True).values
            return (f - f_min) / (f_max - f_min)
        pts_feat_s, feat_t2s = norm(pts_feat_s), norm(feat_t2s)
    ld['feat'] = crit.mse(pts_feat_s, feat_t2s.detach())
    total = ld['ce_vox'] + ld['ce_pix'] + crit.w_kl * ld['kl'] + sum(ld['mse']) + crit.w_feat * ld['feat']
    ld['total'] = total
    return ld
