"""TEST INFRASTRUCTURE (oracle): plain-loop restatement of the reference's nuScenes LiDAR + camera loader pieces that
u2mkd_amd/data/nuscenes_lc.py implements with array operations.  Only tests/ may import this.

Follows core/datasets/lc_semantic_nusc_tsd_full.py of the reference:

* ``aggregate_sweeps``  :241-310  (_aggregate_lidar_sweeps: walk the prev / next sample_data chain, drop points within
  1 m of the sensor in x AND y, move every sweep into the key frame's sensor frame through
  ref_from_car . car_from_global . global_from_car . car_from_current, positive time lags);
* ``project_points``    :344-387  (lidar -> ego(lidar time) -> global -> ego(camera time) -> camera, depth > 1 m, pinhole
  projection, normalisation by (W - 1, H - 1) to [-1, 1], strict inside test; ``valid_mask`` = last camera that sees a point);
* ``collate``           :464-486  (the recursive collate rules: masks -> list of bool tensors, pixel_coordinates -> list of
  float tensors, SparseTensor -> batch index appended as the LAST coordinate column, arrays stacked as float, the rest
  listed).

Everything is written point by point with scalar arithmetic (no broadcasting, no matrix products) on the raw tables
(the ``<version>/*.json`` files): an independent formulation of the same numbers.  The reference's own loader cannot be
imported here (nuscenes-devkit, pyquaternion and torchvision are not installed), and the reference holds no fixture
for it: parity of this restatement is by reading, the loader is pinned against it by tests/test_nuscenes_loader.py.
Quaternions are (w, x, y, z), Hamilton convention, as pyquaternion / the devkit use them.
"""
import json
import os

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------------- tables
def load_tables(dataroot, version):
    """{table: {token: record}} for the tables the loader touches (nuscenes-devkit: NuScenes.get(table, token))."""
    out = {}
    for name in ('sample', 'sample_data', 'ego_pose', 'calibrated_sensor', 'lidarseg'):
        with open(os.path.join(dataroot, version, name + '.json')) as f:
            rows = json.load(f)
        out[name] = {r['token']: r for r in rows}
    # NuScenes.__init__ decorates every sample with sample['data'][channel] = its KEY-FRAME sample_data token
    sensors = {}
    with open(os.path.join(dataroot, version, 'sensor.json')) as f:
        for r in json.load(f):
            sensors[r['token']] = r['channel']
    for smp in out['sample'].values():
        smp['data'] = {}
    for sd in out['sample_data'].values():
        if sd['is_key_frame']:
            cs = out['calibrated_sensor'][sd['calibrated_sensor_token']]
            out['sample'][sd['sample_token']]['data'][sensors[cs['sensor_token']]] = sd['token']
    return out


# ------------------------------------------------------------------------------------------ scalar geometry
def _rot(q):
    """3 x 3 rotation matrix (list of rows) of the quaternion (w, x, y, z): pyquaternion's rotation_matrix."""
    w, x, y, z = [float(v) for v in q]
    n = (w * w + x * x + y * y + z * z) ** 0.5
    w, x, y, z = w / n, x / n, y / n, z / n
    return [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]


def _apply(r, p):
    return [r[0][0] * p[0] + r[0][1] * p[1] + r[0][2] * p[2],
            r[1][0] * p[0] + r[1][1] * p[1] + r[1][2] * p[2],
            r[2][0] * p[0] + r[2][1] * p[1] + r[2][2] * p[2]]


def _apply_t(r, p):      # the transposed (= inverse) rotation
    return [r[0][0] * p[0] + r[1][0] * p[1] + r[2][0] * p[2],
            r[0][1] * p[0] + r[1][1] * p[1] + r[2][1] * p[2],
            r[0][2] * p[0] + r[1][2] * p[1] + r[2][2] * p[2]]


def _to_parent(rec, p):      # x_parent = R x + t        (transform_matrix(..., inverse=False))
    q = _apply(_rot(rec['rotation']), p)
    t = rec['translation']
    return [q[0] + t[0], q[1] + t[1], q[2] + t[2]]


def _to_child(rec, p):       # x_child = R^T (x - t)     (transform_matrix(..., inverse=True))
    t = rec['translation']
    return _apply_t(_rot(rec['rotation']), [p[0] - t[0], p[1] - t[1], p[2] - t[2]])


# --------------------------------------------------------------------------------- :241-310 sweep aggregation
def aggregate_sweeps(tb, dataroot, sample, nsweeps, only_past=False):
    """(points [M, 4] float64 in the key frame's LIDAR_TOP frame + intensity, time lags [M]) of the non-key-frame sweeps
    around ``sample``: up to ``nsweeps`` previous ones, then ``2 nsweeps - (previous found)`` following ones."""
    ref_sd = tb['sample_data'][sample['data']['LIDAR_TOP']]
    ref_pose = tb['ego_pose'][ref_sd['ego_pose_token']]
    ref_cs = tb['calibrated_sensor'][ref_sd['calibrated_sensor_token']]
    ref_time = 1e-6 * ref_sd['timestamp']
    pts, lags = [], []

    def walk(count, direction):
        found = 0
        cur = ref_sd
        for _ in range(count):
            if cur[direction] == '':
                break
            cur = tb['sample_data'][cur[direction]]
            found += 1
            raw = np.fromfile(os.path.join(dataroot, cur['filename']), dtype=np.float32).reshape(-1, 5)
            pose = tb['ego_pose'][cur['ego_pose_token']]
            cs = tb['calibrated_sensor'][cur['calibrated_sensor_token']]
            lag = ref_time - 1e-6 * cur['timestamp'] if direction == 'prev' else 1e-6 * cur['timestamp'] - ref_time
            for row in raw:
                x, y, z, inten = float(row[0]), float(row[1]), float(row[2]), float(row[3])
                if abs(x) < 1.0 and abs(y) < 1.0:              # _remove_close: BOTH |x| and |y| below 1 m
                    continue
                p = _to_parent(cs, [x, y, z])                  # sensor -> ego (sweep time)
                p = _to_parent(pose, p)                        # ego -> global
                p = _to_child(ref_pose, p)                     # global -> ego (key-frame time)
                p = _to_child(ref_cs, p)                       # ego -> key-frame sensor
                pts.append([p[0], p[1], p[2], inten])
                lags.append(lag)
        return found
    n_prev = walk(nsweeps, 'prev')
    if not only_past:
        walk(2 * nsweeps - n_prev, 'next')
    return np.asarray(pts, dtype=np.float64).reshape(-1, 4), np.asarray(lags, dtype=np.float64)


# --------------------------------------------------------------------------------------- :344-387 projection
def project_points(tb, sample, xyz, channels, image_wh=(1600, 900)):
    """(pixel_coordinates [ncam, N, 2] in [-1, 1] (width, height), masks [ncam, N], valid [N] = index into ``channels``
    of the LAST camera that sees the point or -1) of the key-frame points ``xyz`` [N, 3] (LIDAR_TOP frame)."""
    lidar_sd = tb['sample_data'][sample['data']['LIDAR_TOP']]
    cs_l = tb['calibrated_sensor'][lidar_sd['calibrated_sensor_token']]
    pose_l = tb['ego_pose'][lidar_sd['ego_pose_token']]
    n = len(xyz)
    pix = np.zeros((len(channels), n, 2), dtype=np.float64)
    masks = np.zeros((len(channels), n), dtype=bool)
    valid = np.full(n, -1)
    w, h = image_wh
    for ci, channel in enumerate(channels):
        cam_sd = tb['sample_data'][sample['data'][channel]]
        pose_c = tb['ego_pose'][cam_sd['ego_pose_token']]
        cs_c = tb['calibrated_sensor'][cam_sd['calibrated_sensor_token']]
        k = cs_c['camera_intrinsic']
        for i in range(n):
            p = [float(xyz[i][0]), float(xyz[i][1]), float(xyz[i][2])]
            p = _to_parent(cs_l, p)            # first step: lidar -> ego at the sweep's timestamp
            p = _to_parent(pose_l, p)          # second step: ego -> global
            p = _to_child(pose_c, p)           # third step: global -> ego at the image's timestamp
            p = _to_child(cs_c, p)             # fourth step: ego -> camera
            depth = p[2]
            with np.errstate(divide='ignore', invalid='ignore'):
                # fifth step: view_points(..., normalize=True) -- K p, divided by the third component
                q = [np.float64(k[r][0]) * p[0] + np.float64(k[r][1]) * p[1] + np.float64(k[r][2]) * p[2] for r in range(3)]
                u = q[0] / q[2] / (w - 1.0) * 2.0 - 1.0
                v = q[1] / q[2] / (h - 1.0) * 2.0 - 1.0
            pix[ci, i, 0], pix[ci, i, 1] = u, v
            inside = depth > 1 and u > -1 and u < 1 and v > -1 and v < 1
            masks[ci, i] = inside
            if inside:
                valid[i] = ci
    return pix, masks, valid


# ------------------------------------------------------------------------------------------ :464-486 collate
def collate(batch, is_sparse, sparse_parts):
    """The recursive collate of ``_LCNuScenesTSDistillFullInternal.collate_fn``.  ``is_sparse(x)`` tells a SparseTensor,
    ``sparse_parts(x)`` returns its (feats, coords) arrays; a collated SparseTensor is returned as the pair
    (feats tensor, coords tensor with the sample index appended as the last column) -- what sparse_collate builds."""
    first = batch[0]
    if not isinstance(first, dict):
        return batch
    out = {}
    for key in first.keys():
        vals = [smp[key] for smp in batch]
        if key == 'masks':
            out[key] = [torch.from_numpy(np.asarray(v)) for v in vals]
        elif key == 'pixel_coordinates':
            out[key] = [torch.from_numpy(np.asarray(v)).float() for v in vals]
        elif is_sparse(first[key]):
            feats, coords = [], []
            for b, v in enumerate(vals):
                f, c = sparse_parts(v)
                for row in range(len(c)):
                    coords.append([int(c[row][0]), int(c[row][1]), int(c[row][2]), b])
                feats.append(torch.as_tensor(np.asarray(f)))
            out[key] = (torch.cat(feats, 0), torch.tensor(coords, dtype=torch.int32).reshape(-1, 4))
        elif isinstance(first[key], np.ndarray):
            out[key] = torch.stack([torch.from_numpy(v).float() for v in vals], dim=0)
        elif isinstance(first[key], torch.Tensor):
            out[key] = torch.stack(vals, dim=0)
        elif isinstance(first[key], dict):
            out[key] = collate(vals, is_sparse, sparse_parts)
        else:
            out[key] = vals
    return out
