"""Loop restatements of the reference's point <-> pixel transfers and KD re-indexing
(TEST INFRASTRUCTURE; rows a14, a15, a18 of SURVEY.md §8a).

l2c_loop:  core/models/nuscenes/spvcnn_swiftnet18_spformer_tsd_full.py:448-478
c2l_loop:  same file :482-495 with Feature_Gather (core/models/fusion_blocks.py:241-254)
t2s_loop:  core/nusc_trainers.py:295-324
Pinned through the golden KD vectors made by the reference's own model
(tests/golden/kd_cr10_3000.npz)."""
import torch
import torch.nn.functional as F


def l2c_loop(pts_feat, pixel_coordinates, masks, ifh, ifw, n_stage, idx):
    out = []
    cur = 0
    for mask, coord in zip(masks, pixel_coordinates):
        n = mask.size(1)
        bs = pts_feat[cur:cur + n, :]
        for co, ma in zip(coord, mask):
            l2c_f = torch.zeros(1, pts_feat.size(1), ifh, ifw, dtype=pts_feat.dtype)
            if torch.sum(ma) == 0:
                out.append(l2c_f / (n_stage - idx))
                continue
            cnt = 1
            for _ in range(idx, n_stage):
                c_ih = int(round(float(ifh) / cnt + 0.01))
                c_iw = int(round(float(ifw) / cnt + 0.01))
                u = (co[:, 0] + 1.0) / 2 * (c_iw - 1.0)
                v = (co[:, 1] + 1.0) / 2 * (c_ih - 1.0)
                uv = torch.floor(torch.stack([u, v], dim=1)).long()
                uv = torch.fliplr(uv[ma])
                uq, inv, count = torch.unique(uv, dim=0, return_inverse=True, return_counts=True)
                f2d = torch.zeros(uq.size(0), pts_feat.size(1), dtype=pts_feat.dtype)
                f2d.scatter_add_(0, inv.view(-1, 1).expand(-1, f2d.size(-1)), bs[ma])
                f2d /= count.view(-1, 1)
                tmp = torch.sparse_coo_tensor(uq.transpose(0, 1).contiguous(), f2d, size=(c_ih, c_iw, f2d.size(-1))
                                              ).to_dense().permute(2, 0, 1).contiguous().view(1, -1, c_ih, c_iw)
                l2c_f = l2c_f + F.interpolate(tmp, (ifh, ifw), mode='bilinear', align_corners=True)
                cnt *= 2
            out.append(l2c_f / (n_stage - idx))
        cur += n
    return torch.cat(out, dim=0).contiguous()


def c2l_loop(feature_maps, pixel_coordinates, masks):
    res = []
    for mask, coord, img in zip(masks, pixel_coordinates, feature_maps):
        imf = torch.zeros(mask.size(1), img.size(1), dtype=img.dtype)
        imf_list = F.grid_sample(img, coord.unsqueeze(1), padding_mode='zeros', align_corners=True,
                                 mode='bilinear').squeeze(2).permute(0, 2, 1)
        for m_i in range(mask.size(0)):
            imf[mask[m_i]] = imf_list[m_i, mask[m_i], :]
        res.append(imf)
    return torch.cat(res, dim=0)


def t2s_loop(x_t, inv_map, inds_s, num_pts, num_vox, keyframe_mask_full=None):
    out = []
    cur_v = cur_p = 0
    for n_p, n_v, inds in zip(num_pts, num_vox, inds_s):
        tmp = x_t[cur_v:cur_v + n_v]
        inv = inv_map[cur_p:cur_p + n_p]
        if keyframe_mask_full is not None:
            kfm = keyframe_mask_full[cur_p:cur_p + n_p]
            out.append(tmp[inv, :][kfm, :][inds[0], :])
        else:
            out.append(tmp[inv, :][inds[0], :])
        cur_v += n_v
        cur_p += n_p
    return torch.cat(out, dim=0)
