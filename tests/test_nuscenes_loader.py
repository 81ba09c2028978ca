"""f1 loader (u2mkd_amd/data/nuscenes_lc.py) on a synthetic on-disk nuScenes tree: table reader, geometry against
scipy / homogeneous-matrix restatements, the sample schema of core/datasets/lc_semantic_nusc_tsd_full.py:300-434,
its collate rules (:436-462) and the hand-over to the KD step's batch schema."""
import json
import os

import numpy as np
import pytest
import torch

from u2mkd_amd.data import nuscenes_lc as D
from u2mkd_amd.torchsparse import SparseTensor

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


def _points_seen(feed_dict_s, n):
    """Raw points inside some camera image: the voxel-level masks say so for the kept point of every voxel; a dropped
    point projects like itself, so this helper re-derives the set from `label_fov` != ignore or label == ignore."""
    lab, fov = feed_dict_s['targets_mapped'].F, feed_dict_s['label_fov'].F
    assert len(lab) == n and np.all((fov == lab) | (fov == 0))
    kept = np.zeros(n, bool)
    kept[feed_dict_s['inds'][0]] = feed_dict_s['fov_mask'].F
    assert np.all(fov[kept & (lab != 0)] == lab[kept & (lab != 0)])          # a seen kept point keeps its label
    unseen_kept = np.zeros(n, bool)
    unseen_kept[feed_dict_s['inds'][0]] = ~feed_dict_s['fov_mask'].F
    assert np.all(fov[unseen_kept] == 0)                                        # an unseen kept point is ignored
    return (fov != 0) | ((lab == 0) & kept)


@pytest.fixture(scope='module')
def tree(tmp_path_factory):
    from PIL import Image
    root = str(tmp_path_factory.mktemp('nusc'))
    ver = 'v1.0-mini'
    os.makedirs(os.path.join(root, ver)); os.makedirs(os.path.join(root, 'samples')); os.makedirs(os.path.join(root, 'sweeps'))
    os.makedirs(os.path.join(root, 'lidarseg'))
    rng = np.random.default_rng(0)
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


def test_quaternion_and_transform_match_scipy():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(1)
    for _ in range(20):
        q = rng.normal(size=4)
        want = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()
        assert np.allclose(D.quat_to_rot(q), want, atol=1e-12)
        t = rng.normal(size=3)
        m, mi = D.transform_matrix(t, q), D.transform_matrix(t, q, inverse=True)
        assert np.allclose(m.dot(mi), np.eye(4), atol=1e-12)


def test_tables_index_key_frames(tree):
    root, ver = tree
    tb = D.NuScenesTables(root, ver)
    assert len(tb.sample) == 2
    assert set(tb.sample[0]['data']) == set(CAMS) | {'LIDAR_TOP'}
    assert tb.sample[1]['data']['LIDAR_TOP'] == 'sd_lidar_1_0'
    assert tb.get('lidarseg', 'sd_lidar_0_0')['filename'].endswith('_lidarseg.bin')


def test_val_sample_schema_and_projection(tree):
    root, ver = tree
    tb = D.NuScenesTables(root, ver)
    ds = D.LCNuScenesDataset(tb, voxel_size=0.05, split='val', im_cr=0.1)
    it = ds[0]
    s, t = it['feed_dict_s'], it['feed_dict_t']
    pts = np.fromfile(os.path.join(root, 'samples/lidar_0_0.bin'), dtype=np.float32).reshape(-1, 5)[:, :4]
    n = pts.shape[0]
    # voxelisation and first-wins quantisation, no augmentation on val (:197-233, 396-432)
    vox = np.round(pts[:, :3] / 0.05).astype(np.int32); vox -= vox.min(0, keepdims=True)
    inds = s['inds'][0]
    assert isinstance(s['lidar'], SparseTensor)
    assert np.array_equal(s['lidar'].C, vox[inds]) and np.array_equal(s['lidar'].F, pts[inds])
    assert len(np.unique(vox[inds], axis=0)) == len(inds) == s['num_vox']
    assert np.array_equal(vox[inds][s['inverse_map'].F], vox)            # every point maps to its voxel
    assert t['num_pts'] == n and np.array_equal(t['lidar'].C, s['lidar'].C)   # same cloud, no multi-sweep
    lab = np.fromfile(os.path.join(root, 'lidarseg/sd_lidar_0_0_lidarseg.bin'), dtype=np.uint8)
    assert np.array_equal(s['targets_mapped'].F, np.vectorize(D.LABELS_MAPPING.__getitem__)(lab))
    assert s['images'].shape == (6, 90, 160, 3) and s['images'].dtype == np.uint8
    assert s['pixel_coordinates'].shape == (6, len(inds), 2) and s['masks'].shape == (6, len(inds))
    # projection of camera 1 (CAM_FRONT) by ONE fused homogeneous matrix (independent restatement)
    g = lambda name, tok: tb.get(name, tok)
    lsd, csd = g('sample_data', 'sd_lidar_0_0'), g('sample_data', 'sd_CAM_FRONT_0')
    chain = [D.transform_matrix(**{k: g('calibrated_sensor', csd['calibrated_sensor_token'])[k] for k in ('translation', 'rotation')}, inverse=True),
             D.transform_matrix(**{k: g('ego_pose', csd['ego_pose_token'])[k] for k in ('translation', 'rotation')}, inverse=True),
             D.transform_matrix(**{k: g('ego_pose', lsd['ego_pose_token'])[k] for k in ('translation', 'rotation')}),
             D.transform_matrix(**{k: g('calibrated_sensor', lsd['calibrated_sensor_token'])[k] for k in ('translation', 'rotation')})]
    m = chain[0] @ chain[1] @ chain[2] @ chain[3]
    pc = (m @ np.concatenate([pts[inds, :3].astype(np.float64), np.ones((len(inds), 1))], 1).T)[:3]
    with np.errstate(divide='ignore', invalid='ignore'):
        uvw = np.asarray(K_CAM) @ pc
        u = uvw[0] / uvw[2] / 1599.0 * 2 - 1
        v = uvw[1] / uvw[2] / 899.0 * 2 - 1
    want_mask = (pc[2] > 1) & (u > -1) & (u < 1) & (v > -1) & (v < 1)
    assert np.array_equal(s['masks'][1], want_mask) and want_mask.sum() > 10
    assert np.allclose(s['pixel_coordinates'][1][want_mask, 0], u[want_mask], atol=1e-9)
    assert np.allclose(s['pixel_coordinates'][1][want_mask, 1], v[want_mask], atol=1e-9)
    assert np.array_equal(s['fov_mask'].F, s['masks'].any(0))
    # debug.debug_val (:137, 453-456): the labels of the points some camera sees, `ignore` elsewhere, over ALL raw points
    assert 'label_fov' not in s
    dbg = D.LCNuScenesDataset(tb, voxel_size=0.05, split='val', im_cr=0.1, debug=True)[0]['feed_dict_s']
    seen = _points_seen(dbg, n)
    assert np.array_equal(dbg['label_fov'].F, np.where(seen, dbg['targets_mapped'].F, 0))
    assert np.array_equal(dbg['label_fov'].C, dbg['targets_mapped'].C) and seen.sum() > 10


def test_train_sample_drops_cameras_and_multisweep_masks(tree):
    root, ver = tree
    tb = D.NuScenesTables(root, ver)
    ds = D.LCNuScenesDataset(tb, split='train', im_cr=0.1, im_drop=3, multisweeps=2, only_past=False,
                             rng=np.random.default_rng(3))
    it = ds[0]
    s, t = it['feed_dict_s'], it['feed_dict_t']
    assert s['images'].shape[0] == 3 and s['masks'].shape[0] == 3
    # key frame 0 has no previous sweep: 2 * nsweeps = 4 following ones are aggregated (:293-296), close points removed
    kf = t['keyframe_mask_full'].F
    assert kf.sum() == 4000 and t['num_pts'] == len(kf) > 4000
    assert np.all(t['targets_mapped'].F[~kf] == 0)
    assert t['keyframe_mask'].F.shape[0] == t['num_vox']
    # the augmentation is a similarity: pairwise distances scale by one factor in [0.95, 1.05]
    pts = np.fromfile(os.path.join(root, 'samples/lidar_0_0.bin'), dtype=np.float32).reshape(-1, 5)[:, :3]
    inds = s['inds'][0]
    a, b = s['lidar'].F[:50, :3], pts[inds[:50]]
    r = np.linalg.norm(a[1:] - a[:-1], axis=1) / np.linalg.norm(b[1:] - b[:-1], axis=1)
    assert 0.949 < r.min() and r.max() < 1.051 and np.ptp(r) < 1e-4


def test_collate_and_kd_batch(tree):
    root, ver = tree
    tb = D.NuScenesTables(root, ver)
    ds = D.LCNuScenesDataset(tb, split='val', im_cr=0.1)
    c = D.collate_fn([ds[0], ds[1]])
    s = c['feed_dict_s']
    assert s['lidar'].C.shape[1] == 4 and set(s['lidar'].C[:, 3].tolist()) == {0, 1}        # batch index LAST
    assert s['images'].shape == (2, 6, 90, 160, 3) and s['images'].dtype == torch.float32
    assert isinstance(s['masks'], list) and s['masks'][0].dtype == torch.bool
    assert isinstance(s['inds'], list) and isinstance(s['inds'][0], list)
    assert c['lidar_token'] == ['sd_lidar_0_0', 'sd_lidar_1_0']
    kb = D.collated_to_kd_batch(c)
    st, te = kb['student'], kb['teacher']
    assert st['coords'].dtype == np.int32 and st['coords'].shape[0] == sum(st['num_vox']) == st['feats'].shape[0]
    assert len(st['pixel_coordinates']) == 2 and st['pixel_coordinates'][0].shape == (6, st['num_vox'][0], 2)
    assert te['inverse_map'].shape[0] == sum(te['num_pts'])
    # the re-index of core/nusc_trainers.py:295-324 lands every student voxel on its own teacher voxel on val
    # (same cloud, no augmentation): teacher coords at inverse_map[inds] == student coords
    off_p = off_v = 0
    for b in range(2):
        inv = te['inverse_map'][off_p:off_p + te['num_pts'][b]]
        got = te['coords'][off_v + inv[st['inds'][b][0]]][:, :3]
        lo = sum(st['num_vox'][:b])
        assert np.array_equal(got, st['coords'][lo:lo + st['num_vox'][b], :3])
        off_p += te['num_pts'][b]; off_v += te['num_vox'][b]


@pytest.mark.gpu
def test_loader_batch_drives_a_kd_step_on_the_gpu(tree):
    """f1 end to end: the loader's collated batch (two samples, six cameras, images at the network size) through
    ``collated_to_kd_batch`` / ``kd_batch_to_device`` into one KD training step on the HIP operators."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from u2mkd_amd import kd as KD, lidar, train as T
    root, ver = tree
    tb = D.NuScenesTables(root, ver)
    ds = D.LCNuScenesDataset(tb, split='val', im_cr=0.08)             # 72 x 128 images
    batch = T.kd_batch_to_device(D.collated_to_kd_batch(D.collate_fn([ds[0], ds[1]])))
    assert batch['images'].shape == (2, 6, 3, 72, 128)
    torch.manual_seed(0)
    sp = {k: v for k, v in lidar.spformer_kwargs().items() if k not in ('cr', 'in_channel', 'num_classes')}
    model = KD.TSDFull(cr=0.5, cr_t=0.5, in_channel=4, in_channel_t=4, num_classes=17, spformer=sp).cuda()
    run = T.KDStep(model, num_epochs=50, batch_size=2)
    run.train_mode()
    losses = [float(run(batch)) for _ in range(2)]
    assert all(np.isfinite(losses))
    assert all(p.grad is not None for p in model.model_s.parameters())
