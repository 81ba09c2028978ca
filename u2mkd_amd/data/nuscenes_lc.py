"""LiDAR + camera nuScenes loader for the KD step.

Produces, sample by sample, what ``_LCNuScenesTSDistillFullInternal.__getitem__`` of the reference produces
(core/datasets/lc_semantic_nusc_tsd_full.py:300-434) and collates it with the same ``collate_fn`` rules
(:436-462): ``{'feed_dict_s', 'feed_dict_t', 'lidar_token'}`` with torchsparse ``SparseTensor`` payloads, so the
batch drops into ``train.KDStep`` through ``collated_to_kd_batch``.  Follows the reference step by step:

* key-frame sweep ``[N,5] f32`` -> x, y, z, intensity; lidarseg labels through ``labels_mapping`` (:72-105, 306-311);
* teacher dict: optional multi-sweep aggregation with ego-motion compensation and a 1 m close-point filter
  (``_aggregate_lidar_sweeps``, :236-298), random rotate / scale / flip in training (:173-195), voxelise by
  ``round(xyz / voxel_size)`` minus its minimum, ``sparse_quantize`` (first point of a voxel wins) (:197-233);
* per camera (training drops ``im_drop`` random cameras): lidar -> ego -> global -> ego(cam time) -> camera,
  depth > 1 m, pinhole projection, normalisation by (W-1, H-1) to [-1, 1], strict inside test (:326-378);
* student dict: its own random yaw / scale, voxelise, quantise, ``inds`` (:387-432).

The reference reads the tables through the nuscenes-devkit, rotates with pyquaternion and resizes with
torchvision; none of the three is a dependency here: ``NuScenesTables`` reads the same ``<version>/*.json``
tables, ``quat_to_rot`` is the Hamilton (w, x, y, z) rotation matrix the devkit uses, images are resized with
PIL's bilinear filter (what ``torchvision.transforms.Resize`` applies to a PIL image).  Instance-paste
augmentation (``InstAugmentationV2``, an offline object database) is out of scope.
"""
from __future__ import annotations

import json
import os
from functools import reduce

import numpy as np
import torch

from ..torchsparse import SparseTensor
from ..torchsparse.utils.collate import sparse_collate
from ..torchsparse.utils.quantize import sparse_quantize

__all__ = ['NuScenesTables', 'LCNuScenesDataset', 'collate_fn', 'collated_to_kd_batch', 'quat_to_rot',
           'transform_matrix', 'LABELS_MAPPING', 'CAM_CHANNELS']

# core/datasets/lc_semantic_nusc_tsd_full.py:72-105: lidarseg class -> the 16 training classes (0 = ignore)
LABELS_MAPPING = {1: 0, 5: 0, 7: 0, 8: 0, 10: 0, 11: 0, 13: 0, 19: 0, 20: 0, 0: 0, 29: 0, 31: 0, 9: 1, 14: 2, 15: 3,
                  16: 3, 17: 4, 18: 5, 21: 6, 2: 7, 3: 7, 4: 7, 6: 7, 12: 8, 22: 9, 23: 10, 24: 11, 25: 12, 26: 13,
                  27: 14, 28: 15, 30: 16}
_LABEL_LUT = np.zeros(256, dtype=np.int64)
for _k, _v in LABELS_MAPPING.items():
    _LABEL_LUT[_k] = _v

CAM_CHANNELS = ['CAM_FRONT_LEFT', 'CAM_FRONT', 'CAM_FRONT_RIGHT', 'CAM_BACK_LEFT', 'CAM_BACK', 'CAM_BACK_RIGHT']   # :107-108
IMAGE_SIZE = (900, 1600)                                                                                           # :110


def quat_to_rot(q) -> np.ndarray:
    """Rotation matrix of the unit quaternion (w, x, y, z) -- ``pyquaternion.Quaternion(q).rotation_matrix``."""
    w, x, y, z = (float(v) for v in q)
    n = np.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def transform_matrix(translation, rotation, inverse=False) -> np.ndarray:
    """nuscenes.utils.data_classes.transform_matrix: homogeneous 4x4 of (translation, quaternion) or its inverse."""
    tm = np.eye(4)
    r = quat_to_rot(rotation)
    t = np.asarray(translation, dtype=np.float64)
    if inverse:
        tm[:3, :3] = r.T
        tm[:3, 3] = r.T.dot(-t)
    else:
        tm[:3, :3] = r
        tm[:3, 3] = t
    return tm


class NuScenesTables:
    """The relational tables of a nuScenes release (``<dataroot>/<version>/*.json``), by token -- the part of
    ``nuscenes.NuScenes`` the loader uses: ``sample`` (with its ``data`` channel -> key-frame sample_data token map),
    ``get(table, token)``, ``dataroot``."""

    TABLES = ('sample', 'sample_data', 'ego_pose', 'calibrated_sensor', 'sensor', 'lidarseg')

    def __init__(self, dataroot: str, version: str = 'v1.0-trainval'):
        self.dataroot, self.version = dataroot, version
        self._tab, self._idx = {}, {}
        for name in self.TABLES:
            path = os.path.join(dataroot, version, name + '.json')
            if not os.path.exists(path):
                if name == 'lidarseg':       # test split: no labels
                    self._tab[name], self._idx[name] = [], {}
                    continue
                raise FileNotFoundError(path)
            with open(path) as f:
                rows = json.load(f)
            self._tab[name] = rows
            key = 'sample_data_token' if name == 'lidarseg' else 'token'     # lidarseg records are fetched by their sweep
            self._idx[name] = {r[key]: i for i, r in enumerate(rows)}
        # devkit reverse index: sample['data'][channel] = token of the key-frame sample_data of that channel
        chan = {}
        for cs in self._tab['calibrated_sensor']:
            chan[cs['token']] = self.get('sensor', cs['sensor_token'])['channel']
        for s in self._tab['sample']:
            s['data'] = {}
        for sd in self._tab['sample_data']:
            if sd.get('is_key_frame'):
                self.get('sample', sd['sample_token'])['data'][chan[sd['calibrated_sensor_token']]] = sd['token']
        self.sample = self._tab['sample']

    def get(self, table: str, token: str) -> dict:
        return self._tab[table][self._idx[table][token]]


def _rotate_points(p3n: np.ndarray, rm: np.ndarray) -> np.ndarray:
    return rm.dot(p3n)


class LCNuScenesDataset(torch.utils.data.Dataset):
    """``_LCNuScenesTSDistillFullInternal`` without torchpack / devkit: the hyper-parameters the reference reads from
    the global ``configs`` are arguments (``configs/nuscenes/default.yaml`` values as defaults).

    ``select_idx``: indices into ``tables.sample`` (the reference loads ./data/nuscenes/nuscenes_{train,val}_official.npy);
    None = all samples.  ``rng``: numpy Generator for the augmentations (the reference uses the global numpy state)."""

    def __init__(self, tables: NuScenesTables, voxel_size=0.05, split='train', im_cr=0.4, im_drop=3, flip=True,
                 multisweeps=0, only_past=False, ignore_index=0, select_idx=None, rng=None, debug=False):
        self.nusc = tables
        self.voxel_size, self.split = voxel_size, split
        self.input_image_size = [int(x * im_cr) for x in IMAGE_SIZE]                          # :133-134
        self.im_drop, self.flip_aug = im_drop, flip
        self.multisweeps, self.only_past = multisweeps, only_past
        self.ignored_labels = ignore_index
        self.debug = debug                              # configs['debug']['debug_val'] (:137): adds `label_fov` (:453-456)
        self.rng = rng or np.random.default_rng()
        self.sample = tables.sample if select_idx is None else [tables.sample[int(i)] for i in select_idx]

    def __len__(self):
        return len(self.sample)

    # ---- augmentations (:173-195)
    def _rotate_and_scale(self, pts):
        out = np.zeros_like(pts)
        theta = self.rng.uniform(0, 2 * np.pi)
        scale = self.rng.uniform(0.95, 1.05)
        rot = np.array([[np.cos(theta), np.sin(theta), 0], [-np.sin(theta), np.cos(theta), 0], [0, 0, 1]])
        out[:, :3] = np.dot(pts[:, :3], rot) * scale
        out[:, 3] = pts[:, 3]
        return out

    def _flip3d(self, pts):
        kind = int(self.rng.integers(0, 4))
        if kind == 1:
            pts[:, 0] = -pts[:, 0]
        elif kind == 2:
            pts[:, 1] = -pts[:, 1]
        elif kind == 3:
            pts[:, :2] = -pts[:, :2]
        return pts

    # ---- multi-sweep aggregation (:236-298)
    def _aggregate_lidar_sweeps(self, sample_ref, nsweeps, only_past=False):
        nusc = self.nusc
        ref_sd = nusc.get('sample_data', sample_ref['data']['LIDAR_TOP'])
        ref_pose = nusc.get('ego_pose', ref_sd['ego_pose_token'])
        ref_cs = nusc.get('calibrated_sensor', ref_sd['calibrated_sensor_token'])
        ref_time = 1e-6 * ref_sd['timestamp']
        ref_from_car = transform_matrix(ref_cs['translation'], ref_cs['rotation'], inverse=True)
        car_from_global = transform_matrix(ref_pose['translation'], ref_pose['rotation'], inverse=True)

        def walk(count, direction):
            cur, pts, ts = ref_sd, [], []
            for _ in range(count):
                if cur[direction] == '':
                    break
                cur = nusc.get('sample_data', cur[direction])
                p = np.fromfile(os.path.join(nusc.dataroot, cur['filename']), dtype=np.float32).reshape([-1, 5])[:, :4]
                close = np.logical_and(np.fabs(p[:, 0]) < 1.0, np.fabs(p[:, 1]) < 1.0)
                p = p[~close]
                pose = nusc.get('ego_pose', cur['ego_pose_token'])
                global_from_car = transform_matrix(pose['translation'], pose['rotation'])
                cs = nusc.get('calibrated_sensor', cur['calibrated_sensor_token'])
                car_from_current = transform_matrix(cs['translation'], cs['rotation'])
                tm = reduce(np.dot, [ref_from_car, car_from_global, global_from_car, car_from_current])
                xyz = np.dot(tm, np.vstack((p[:, :3].T, np.ones(p.shape[0]))))[:3, :]
                lag = ref_time - 1e-6 * cur['timestamp'] if direction == 'prev' else 1e-6 * cur['timestamp'] - ref_time
                ts.append(lag * np.ones((p.shape[0],)))
                pts.append(np.concatenate([xyz.T, p[:, 3].reshape(-1, 1)], axis=-1))
            return pts, ts

        prev_pts, prev_ts = walk(nsweeps, 'prev')
        next_pts, next_ts = ([], []) if only_past else walk(2 * nsweeps - len(prev_pts), 'next')
        return prev_pts + next_pts, prev_ts + next_ts

    # ---- teacher dict (:197-234)
    def _process_unimodal_input(self, pts, labels_, sample):
        pts = pts.copy()
        keyframe_mask = None
        if self.multisweeps != 0:
            agg_pts, agg_ts = self._aggregate_lidar_sweeps(sample, self.multisweeps, self.only_past)
            agg_ts = np.concatenate([np.zeros(shape=(pts.shape[0],))] + agg_ts, axis=0)
            pts = np.concatenate([pts] + agg_pts, axis=0)
            keyframe_mask = (agg_ts == 0)
            extra = int(np.sum(~keyframe_mask))
            labels_ = np.concatenate([labels_, np.full((extra,), self.ignored_labels, dtype=labels_.dtype)], axis=0)
        if 'train' in self.split:
            pts = self._rotate_and_scale(pts)
            if self.flip_aug:
                pts = self._flip3d(pts)
        voxel = np.round(pts[:, :3] / self.voxel_size).astype(np.int32)
        voxel -= voxel.min(0, keepdims=1)
        feat = pts.astype(np.float32)
        _, inds, inverse_map = sparse_quantize(voxel, return_index=True, return_inverse=True)
        voxel_full = voxel[inds]
        fd = {'lidar': SparseTensor(feat[inds], voxel_full), 'targets': SparseTensor(labels_[inds], voxel_full),
              'targets_mapped': SparseTensor(labels_, voxel), 'inverse_map': SparseTensor(inverse_map, voxel),
              'num_vox': voxel_full.shape[0], 'num_pts': voxel.shape[0]}
        if keyframe_mask is not None:
            fd['keyframe_mask'] = SparseTensor(keyframe_mask[inds], voxel_full)
            fd['keyframe_mask_full'] = SparseTensor(keyframe_mask, voxel)
        return fd

    def _load_image(self, path):
        from PIL import Image
        im = Image.open(path).convert('RGB')
        size = im.size                                                          # (W, H) of the sensor image
        h, w = self.input_image_size
        return np.array(im.resize((w, h), Image.BILINEAR)), size

    # ---- one sample (:300-434)
    def __getitem__(self, index):
        nusc = self.nusc
        sample = self.sample[index]
        lidar_token = sample['data']['LIDAR_TOP']
        lidar_sd = nusc.get('sample_data', lidar_token)
        pts = np.fromfile(os.path.join(nusc.dataroot, lidar_sd['filename']), dtype=np.float32).reshape([-1, 5])[:, :4]
        if self.split == 'test':
            labels_raw = np.zeros(pts.shape[0], dtype=np.int64)
        else:
            seg = np.fromfile(os.path.join(nusc.dataroot, nusc.get('lidarseg', lidar_token)['filename']), dtype=np.uint8)
            labels_raw = _LABEL_LUT[seg]
        train = 'train' in self.split
        feed_dict_t = self._process_unimodal_input(pts, labels_raw, sample)

        images, pixel_coordinates, masks = [], [], []
        valid = np.full(pts.shape[0], -1)
        drop = self.rng.choice(len(CAM_CHANNELS), self.im_drop, replace=False) if train else []
        cs_l = nusc.get('calibrated_sensor', lidar_sd['calibrated_sensor_token'])
        pose_l = nusc.get('ego_pose', lidar_sd['ego_pose_token'])
        for idx, channel in enumerate(CAM_CHANNELS):
            if train and idx in drop:
                continue
            cam_sd = nusc.get('sample_data', sample['data'][channel])
            im, (im_w, im_h) = self._load_image(os.path.join(nusc.dataroot, cam_sd['filename']))
            images.append(im)
            p = pts[:, :3].T.astype(np.float64)                                   # [3, N]
            # lidar -> ego (lidar time) -> global -> ego (camera time) -> camera   (:326-343)
            p = _rotate_points(p, quat_to_rot(cs_l['rotation'])) + np.asarray(cs_l['translation'])[:, None]
            p = _rotate_points(p, quat_to_rot(pose_l['rotation'])) + np.asarray(pose_l['translation'])[:, None]
            pose_c = nusc.get('ego_pose', cam_sd['ego_pose_token'])
            p = _rotate_points(p - np.asarray(pose_c['translation'])[:, None], quat_to_rot(pose_c['rotation']).T)
            cs_c = nusc.get('calibrated_sensor', cam_sd['calibrated_sensor_token'])
            p = _rotate_points(p - np.asarray(cs_c['translation'])[:, None], quat_to_rot(cs_c['rotation']).T)
            mask = p[2, :] > 1
            with np.errstate(divide='ignore', invalid='ignore'):
                p = np.dot(np.asarray(cs_c['camera_intrinsic'], dtype=np.float64), p)
                p = p / p[2:3, :]
            uv = p[:2, :]
            uv[0, :] = uv[0, :] / (im_w - 1.0) * 2.0 - 1.0
            uv[1, :] = uv[1, :] / (im_h - 1.0) * 2.0 - 1.0
            mask = mask & (uv[0, :] > -1) & (uv[0, :] < 1) & (uv[1, :] > -1) & (uv[1, :] < 1)
            valid[mask] = idx
            masks.append(mask)
            pixel_coordinates.append(uv.T)
        pt_with_img = valid != -1
        pixel_coordinates = np.stack(pixel_coordinates, axis=0)
        masks = np.stack(masks, axis=0)
        images = np.stack(images, axis=0)

        pts_cp = np.zeros_like(pts)
        if train:
            theta = self.rng.uniform(0, 2 * np.pi)
            scale = self.rng.uniform(0.95, 1.05)
            rot = np.array([[np.cos(theta), np.sin(theta), 0], [-np.sin(theta), np.cos(theta), 0], [0, 0, 1]])
            pts_cp[:, :3] = np.dot(pts[:, :3], rot) * scale
        else:
            pts_cp[...] = pts[...]
        pts_cp[:, 3] = pts[:, 3]
        voxel = np.round(pts_cp[:, :3] / self.voxel_size).astype(np.int32)
        voxel -= voxel.min(0, keepdims=1)
        feat = pts_cp.astype(np.float32)
        _, inds, inverse_map = sparse_quantize(voxel, return_index=True, return_inverse=True)
        voxel_full = voxel[inds]
        feed_dict_s = {
            'lidar': SparseTensor(feat[inds], voxel_full), 'targets': SparseTensor(labels_raw[inds], voxel_full),
            'targets_mapped': SparseTensor(labels_raw, voxel), 'inverse_map': SparseTensor(inverse_map, voxel),
            'images': images, 'pixel_coordinates': pixel_coordinates[:, inds, :], 'masks': masks[:, inds],
            'fov_mask': SparseTensor(pt_with_img[inds], voxel_full), 'inds': [inds], 'num_vox': voxel_full.shape[0]}
        if self.debug:                                  # labels of the points a camera sees, `ignore` elsewhere (:453-456)
            label_fov = np.full_like(labels_raw, fill_value=self.ignored_labels)
            label_fov[pt_with_img] = labels_raw[pt_with_img]
            feed_dict_s['label_fov'] = SparseTensor(label_fov, voxel)
        return {'feed_dict_s': feed_dict_s, 'feed_dict_t': feed_dict_t, 'lidar_token': lidar_token}

    collate_fn = staticmethod(lambda batch: collate_fn(batch))


def collate_fn(batch):
    """``_LCNuScenesTSDistillFullInternal.collate_fn`` (:436-462)."""
    if not isinstance(batch[0], dict):
        return batch
    out = {}
    for key in batch[0].keys():
        first = batch[0][key]
        if key == 'masks':
            out[key] = [torch.from_numpy(s[key]) for s in batch]
        elif key == 'pixel_coordinates':
            out[key] = [torch.from_numpy(s[key]).float() for s in batch]
        elif isinstance(first, SparseTensor):
            out[key] = sparse_collate([s[key] for s in batch])
        elif isinstance(first, np.ndarray):
            out[key] = torch.stack([torch.from_numpy(s[key]).float() for s in batch], dim=0)
        elif isinstance(first, torch.Tensor):
            out[key] = torch.stack([s[key] for s in batch], dim=0)
        elif isinstance(first, dict):
            out[key] = collate_fn([s[key] for s in batch])
        else:
            out[key] = [s[key] for s in batch]
    return out


def collated_to_kd_batch(c):
    """Collated loader output -> the numpy batch of ``train.kd_batch_to_device`` (the schema ``synth.synth_kd_batch``
    emits): what ``NuScenesLCTSDFullTrainer._run_step`` reads from its feed dicts (core/nusc_trainers.py:255-300)."""
    s, t = c['feed_dict_s'], c['feed_dict_t']
    npy = lambda x: x.numpy() if torch.is_tensor(x) else np.asarray(x)
    student = {'coords': npy(s['lidar'].C).astype(np.int32), 'feats': npy(s['lidar'].F).astype(np.float32),
               'targets': npy(s['targets'].F).astype(np.int64), 'images': npy(s['images']).astype(np.float32),
               'pixel_coordinates': [npy(p).astype(np.float32) for p in s['pixel_coordinates']],
               'masks': [npy(m).astype(bool) for m in s['masks']], 'fov_mask': npy(s['fov_mask'].F).astype(bool),
               'inds': [[npy(i[0]).astype(np.int64)] for i in s['inds']], 'num_vox': list(s['num_vox'])}
    teacher = {'coords': npy(t['lidar'].C).astype(np.int32), 'feats': npy(t['lidar'].F).astype(np.float32),
               'targets': npy(t['targets'].F).astype(np.int64), 'inverse_map': npy(t['inverse_map'].F).astype(np.int64),
               'num_pts': list(t['num_pts']), 'num_vox': list(t['num_vox'])}
    if 'keyframe_mask_full' in t:
        teacher['keyframe_mask_full'] = npy(t['keyframe_mask_full'].F).astype(bool)
        teacher['keyframe_mask'] = npy(t['keyframe_mask'].F).astype(bool)
    return {'student': student, 'teacher': teacher}
