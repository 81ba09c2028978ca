"""Point <-> pixel transfer and fusion blocks of the multi-modal student (rows a14-a16;
core/models/fusion_blocks.py:9-153, 241-278 and the two Python loops of
core/models/nuscenes/spvcnn_swiftnet18_spformer_tsd_full.py:448-495).

The reference walks samples x cameras x scales in Python with ``torch.unique``,
``sparse_coo_tensor(...).to_dense()`` and boolean-mask writes (a host sync per camera).  Here
both transfers are batched tensor programs without host synchronisation:

* :func:`l2c_scatter`  -- point features -> per-camera feature maps: one ``index_add_`` per
  scale over ALL (sample, camera) pairs into a dense [B*ncam, H_c, W_c, C] accumulator (pixel
  mean = sum / count), bilinear up-sampling, mean over scales;
* :func:`c2l_gather`   -- camera feature maps -> point features: one ``grid_sample`` per sample
  over all cameras, then the reference's "later camera overwrites" rule as a mask-priority
  select.
Fusion modules keep the reference's parameter names."""
import torch
import torch.nn.functional as F
from torch import nn

__all__ = ['IA_Layer', 'Atten_Fusion_Conv', 'L2CAILayer', 'L2CFusion', 'feature_gather', 'c2l_gather',
           'l2c_scatter', 'feature_fetch']


class IA_Layer(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.ic, self.pc = channels
        rc = self.pc // 4
        self.conv1 = nn.Sequential(nn.Conv1d(self.ic, self.pc, 1), nn.BatchNorm1d(self.pc), nn.ReLU(True))
        self.fc1 = nn.Sequential(nn.BatchNorm1d(self.ic), nn.ReLU(True), nn.Linear(self.ic, rc))
        self.fc2 = nn.Linear(self.pc, rc)
        self.fc3 = nn.Linear(rc, 1)

    def forward(self, img_feats, point_feats):
        img_feats = img_feats.contiguous()
        att = torch.sigmoid(self.fc3(torch.tanh(self.fc1(img_feats) + self.fc2(point_feats.contiguous()))))
        att = att.view(1, 1, -1)
        return self.conv1(img_feats.unsqueeze(0).transpose(1, 2).contiguous()) * att      # [1, pc, N]


class Atten_Fusion_Conv(nn.Module):
    def __init__(self, inplanes_I, inplanes_P, outplanes):
        super().__init__()
        self.ai_layer = IA_Layer(channels=[inplanes_I, inplanes_P])
        self.conv1 = nn.Conv1d(inplanes_P + inplanes_P, outplanes, 1)
        self.bn1 = nn.BatchNorm1d(outplanes)

    def forward(self, point_features, img_features):
        img = self.ai_layer(img_features, point_features)
        fused = torch.cat([point_features.unsqueeze(0).transpose(1, 2), img], dim=1)
        return F.relu(self.bn1(self.conv1(fused))).squeeze(0).transpose(0, 1)


class L2CAILayer(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.ic, self.pc = channels
        rc = self.ic // 4
        self.conv1 = nn.Sequential(nn.Conv2d(self.pc, self.ic, 1), nn.BatchNorm2d(self.ic), nn.ReLU(True))
        self.fc1 = nn.Conv2d(self.ic, rc, kernel_size=1)
        self.fc2 = nn.Conv2d(self.pc, rc, kernel_size=1)
        self.fc3 = nn.Conv2d(rc, 1, kernel_size=1)

    def forward(self, img_feats, point_feats):
        att = torch.sigmoid(self.fc3(torch.tanh(self.fc1(img_feats.contiguous()) + self.fc2(point_feats.contiguous()))))
        return self.conv1(point_feats) * att


class L2CFusion(nn.Module):
    def __init__(self, inplanes_I, inplanes_P, outplanes):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes_I + inplanes_I, outplanes, kernel_size=1)
        self.bn1 = nn.BatchNorm2d(outplanes)
        self.l2c_ai_layer = L2CAILayer(channels=[inplanes_I, inplanes_P])

    def forward(self, point_features, img_features):
        l2c = self.l2c_ai_layer(img_features, point_features)
        fused = self.bn1(self.conv1(torch.cat([img_features, l2c], dim=1)))
        return F.relu(fused), fused


def feature_gather(feature_map, xy, mode='bilinear'):
    """[B,C,H,W] sampled at xy [B,N,2] in [-1,1] (width, height) -> [B,C,N] (fusion_blocks.py:241-254)."""
    return F.grid_sample(feature_map, xy.unsqueeze(1), padding_mode='zeros', align_corners=True, mode=mode).squeeze(2)


def _last_camera(mask):
    """Index of the LAST camera that sees each point (the reference writes cameras in order,
    later ones overwrite: tsd_full.py:492-493) and whether any does.  mask bool [ncam, N]."""
    ncam = mask.shape[0]
    rank = torch.arange(1, ncam + 1, device=mask.device).view(-1, 1) * mask.to(torch.int64)
    best = rank.max(0)[0]
    return (best - 1).clamp(min=0), best > 0


def c2l_gather(feature_maps, pixel_coordinates, masks):
    """Camera -> LiDAR gather.  feature_maps [B, ncam, C, h, w]; per sample coordinates
    [ncam, N_b, 2] and masks [ncam, N_b].  Returns [sum N_b, C], zeros outside every camera."""
    out = []
    for fmap, coord, mask in zip(feature_maps, pixel_coordinates, masks):
        sampled = feature_gather(fmap, coord)                       # [ncam, C, N_b]
        cam, seen = _last_camera(mask)
        pick = sampled.gather(0, cam.view(1, 1, -1).expand(1, sampled.shape[1], -1)).squeeze(0)   # [C, N_b]
        out.append((pick * seen.to(pick.dtype).unsqueeze(0)).t())
    return torch.cat(out, dim=0)


def feature_fetch(masks, pixel_coordinates, imfeats, mode='bilinear'):
    """Feature_Fetch (fusion_blocks.py:257-278) on the current device."""
    return c2l_gather(imfeats, pixel_coordinates, masks)


def l2c_scatter(point_feats, pixel_coordinates, masks, ifh, ifw, n_scales):
    """LiDAR -> camera multi-scale scatter-mean (tsd_full.py:448-478).  point_feats [sum N_b, C];
    returns [B*ncam, C, ifh, ifw]: for scale s in 0..n_scales-1 the (masked) points of a camera are
    averaged per pixel of a (round(ifh/2^s + .01), round(ifw/2^s + .01)) grid, the grid is
    bilinearly up-sampled to (ifh, ifw), and the n_scales maps are averaged."""
    B = len(masks)
    ncam = masks[0].shape[0]
    C = point_feats.shape[1]
    dev, dt = point_feats.device, point_feats.dtype
    sizes = [m.shape[1] for m in masks]
    # flatten (sample, camera, point): camera slot id and the point's row in point_feats
    coords = torch.cat([c.reshape(-1, 2) for c in pixel_coordinates], 0)              # [sum ncam*N_b, 2]
    mask = torch.cat([m.reshape(-1) for m in masks], 0)
    slot, row = [], []
    cur = 0
    for b, n in enumerate(sizes):
        slot.append((torch.arange(ncam, device=dev) + b * ncam).repeat_interleave(n))
        row.append((torch.arange(n, device=dev) + cur).repeat(ncam))
        cur += n
    slot, row = torch.cat(slot), torch.cat(row)
    w = mask.to(dt).unsqueeze(1)
    feats = point_feats[row] * w
    total = torch.zeros(B * ncam, C, ifh, ifw, device=dev, dtype=dt)
    cnt = 1
    for _ in range(n_scales):
        ch = int(round(float(ifh) / cnt + 0.01))
        cw = int(round(float(ifw) / cnt + 0.01))
        u = torch.floor((coords[:, 0] + 1.0) / 2 * (cw - 1.0)).long().clamp(0, cw - 1)
        v = torch.floor((coords[:, 1] + 1.0) / 2 * (ch - 1.0)).long().clamp(0, ch - 1)
        pix = (slot * ch + v) * cw + u
        acc = torch.zeros(B * ncam * ch * cw, C, device=dev, dtype=dt).index_add_(0, pix, feats)
        num = torch.zeros(B * ncam * ch * cw, 1, device=dev, dtype=dt).index_add_(0, pix, w)
        grid = (acc / num.clamp(min=1.0)).view(B * ncam, ch, cw, C).permute(0, 3, 1, 2)
        total = total + F.interpolate(grid, (ifh, ifw), mode='bilinear', align_corners=True)
        cnt *= 2
    return total / n_scales
