"""The synthetic on-disk nuScenes tree of the loader tests (tests/test_nuscenes_loader.py) and of the loader-fed KD golden
(tests/golden/make_golden.py): deterministic, so the fixture generator in the build container and the GPU test rebuild
the same scene."""
import json
import os

import numpy as np

from u2mkd_amd.data import nuscenes_lc as D

CAMS = D.CAM_CHANNELS
YAW = {'CAM_FRONT_LEFT': 55.0, 'CAM_FRONT': 0.0, 'CAM_FRONT_RIGHT': -55.0, 'CAM_BACK_LEFT': 110.0, 'CAM_BACK': 180.0,
       'CAM_BACK_RIGHT': -110.0}
K_CAM = [[1266.0, 0.0, 816.0], [0.0, 1266.0, 491.0], [0.0, 0.0, 1.0]]


def _quat_from_matrix(m):
    from scipy.spatial.transform import Rotation
    x, y, z, w = Rotation.from_matrix(m).as_quat()
    return [float(w), float(x), float(y), float(z)]


def _cam_rotation(yaw_deg):
    """camera axes (x right, y down, z forward) in the ego frame (x forward, y left, z up), yawed."""
    a = np.deg2rad(yaw_deg)
    fwd = np.array([np.cos(a), np.sin(a), 0.0]); left = np.array([-np.sin(a), np.cos(a), 0.0]); up = np.array([0, 0, 1.0])
    return np.stack([-left, -up, fwd], axis=1)       # columns = camera axes in ego coordinates


def build_tree(root, seed=0):
    """A two-sample nuScenes tree under `root` (tables, key-frame sweeps + two intermediate sweeps each, lidarseg labels, six
    900 x 1600 camera images per sample), everything drawn from default_rng(seed): the same bytes wherever it is built."""
    from PIL import Image
    ver = 'v1.0-mini'
    os.makedirs(os.path.join(root, ver)); os.makedirs(os.path.join(root, 'samples')); os.makedirs(os.path.join(root, 'sweeps'))
    os.makedirs(os.path.join(root, 'lidarseg'))
    rng = np.random.default_rng(seed)
    sensor = [{'token': 's_lidar', 'channel': 'LIDAR_TOP', 'modality': 'lidar'}] + \
             [{'token': 's_' + c, 'channel': c, 'modality': 'camera'} for c in CAMS]
    calib = [{'token': 'cs_lidar', 'sensor_token': 's_lidar', 'translation': [0.9, 0.0, 1.8],
              'rotation': _quat_from_matrix(np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1.0]])), 'camera_intrinsic': []}]
    for c in CAMS:
        calib.append({'token': 'cs_' + c, 'sensor_token': 's_' + c, 'translation': [1.5, 0.1, 1.5],
                      'rotation': _quat_from_matrix(_cam_rotation(YAW[c])), 'camera_intrinsic': K_CAM})
    sample, sample_data, ego_pose, lidarseg = [], [], [], []

    def pose(tok, t, yaw):
        a = np.deg2rad(yaw)
        m = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
        ego_pose.append({'token': tok, 'translation': [100.0 + 5 * t, 50.0 + 0.5 * t, 0.0], 'rotation': _quat_from_matrix(m),
                         'timestamp': int(1e6 * (1000 + t))})

    def sweep_file(name, n):
        p = np.concatenate([rng.uniform(-40, 40, (n, 2)), rng.uniform(-2, 3, (n, 1)), rng.uniform(0, 255, (n, 1)),
                            np.zeros((n, 1))], 1).astype(np.float32)
        p.tofile(os.path.join(root, name))
        return p

    lidar_chain = []
    for s in range(2):
        stok = f'sample{s}'
        sample.append({'token': stok, 'timestamp': int(1e6 * (1000 + s)), 'scene_token': 'scene0'})
        # key-frame sweep + two intermediate sweeps after it
        for j in range(3):
            tok = f'sd_lidar_{s}_{j}'
            t = s + j * 0.3
            pose('pose_' + tok, t, 3.0 * t)
            folder = 'samples' if j == 0 else 'sweeps'
            fn = f'{folder}/lidar_{s}_{j}.bin'
            sweep_file(fn, 4000 if j == 0 else 1500)
            sample_data.append({'token': tok, 'sample_token': stok, 'ego_pose_token': 'pose_' + tok,
                                'calibrated_sensor_token': 'cs_lidar', 'filename': fn, 'is_key_frame': j == 0,
                                'timestamp': int(1e6 * (1000 + t)), 'prev': '', 'next': ''})
            lidar_chain.append(tok)
            if j == 0:
                lab = rng.integers(0, 32, 4000).astype(np.uint8)
                lfn = f'lidarseg/{tok}_lidarseg.bin'
                lab.tofile(os.path.join(root, lfn))
                lidarseg.append({'token': 'ls_' + tok, 'sample_data_token': tok, 'filename': lfn})
        for c in CAMS:
            tok = f'sd_{c}_{s}'
            pose('pose_' + tok, s + 0.02, 3.0 * s + 0.1)
            fn = f'samples/{c}_{s}.png'
            Image.fromarray(rng.integers(0, 255, (900, 1600, 3), dtype=np.uint8)).save(os.path.join(root, fn))
            sample_data.append({'token': tok, 'sample_token': stok, 'ego_pose_token': 'pose_' + tok,
                                'calibrated_sensor_token': 'cs_' + c, 'filename': fn, 'is_key_frame': True,
                                'timestamp': int(1e6 * (1000 + s + 0.02)), 'prev': '', 'next': ''})
    by = {r['token']: r for r in sample_data}
    for a, b in zip(lidar_chain[:-1], lidar_chain[1:]):
        by[a]['next'], by[b]['prev'] = b, a
    for name, rows in (('sample', sample), ('sample_data', sample_data), ('ego_pose', ego_pose),
                       ('calibrated_sensor', calib), ('sensor', sensor), ('lidarseg', lidarseg)):
        with open(os.path.join(root, ver, name + '.json'), 'w') as f:
            json.dump(rows, f)
    return root, ver


