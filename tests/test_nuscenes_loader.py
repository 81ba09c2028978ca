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

from nusc_tree import CAMS, K_CAM, YAW, build_tree   # noqa: E402  (tests/ is on sys.path: conftest)


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
    return build_tree(str(tmp_path_factory.mktemp('nusc')))


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


def test_loader_equals_the_plain_loop_oracle(tree):
    """u2mkd_amd/data/nuscenes_lc.py against oracle/loader_ref.py (point-by-point scalar restatement of
    core/datasets/lc_semantic_nusc_tsd_full.py:241-310, 344-387, 464-486) on the synthetic tree: aggregated sweeps, the
    camera projection with its masks, and the collate rules."""
    from oracle import loader_ref as LR
    root, ver = tree
    tb = D.NuScenesTables(root, ver)
    rt = LR.load_tables(root, ver)
    # ---- sweep aggregation: key frame 0 (no previous sweep: all 2 * nsweeps from `next`) and key frame 1 (two of each)
    ds = D.LCNuScenesDataset(tb, split='val', im_cr=0.1, multisweeps=2, only_past=False)
    for s_idx in (0, 1):
        for only_past in (False, True):
            got_p, got_t = ds._aggregate_lidar_sweeps(ds.sample[s_idx], 2, only_past)
            want_p, want_t = LR.aggregate_sweeps(rt, root, rt['sample'][f'sample{s_idx}'], 2, only_past)
            got_p = np.concatenate(got_p) if got_p else np.zeros((0, 4))
            got_t = np.concatenate(got_t) if got_t else np.zeros(0)
            assert got_p.shape == want_p.shape and got_p.shape[0] == (0 if (only_past and s_idx == 0) else got_p.shape[0])
            assert np.allclose(got_p, want_p, rtol=0, atol=1e-9) and np.allclose(got_t, want_t, rtol=0, atol=1e-12)
    assert LR.aggregate_sweeps(rt, root, rt['sample']['sample1'], 2, False)[0].shape[0] > 4000
    # ---- projection of the kept point of every voxel, all six cameras
    ds = D.LCNuScenesDataset(tb, split='val', im_cr=0.1)
    it = ds[1]
    s = it['feed_dict_s']
    pts = np.fromfile(os.path.join(root, 'samples/lidar_1_0.bin'), dtype=np.float32).reshape(-1, 5)[:, :3]
    inds = s['inds'][0]
    pix, masks, valid = LR.project_points(rt, rt['sample']['sample1'], pts[inds], D.CAM_CHANNELS)
    assert np.array_equal(s['masks'], masks) and masks.sum() > 100
    finite = np.isfinite(pix).all(-1)
    assert np.array_equal(np.isfinite(s['pixel_coordinates']).all(-1), finite)
    # (a point next to the camera plane projects to ~1e6: relative there, absolute inside the images)
    assert np.allclose(s['pixel_coordinates'][finite], pix[finite], rtol=1e-8, atol=1e-9)
    assert np.allclose(s['pixel_coordinates'][masks], pix[masks], rtol=0, atol=1e-9)
    assert np.array_equal(s['fov_mask'].F, valid != -1)
    # ---- collate
    items = [ds[0], it]
    got = D.collate_fn(items)
    want = LR.collate(items, lambda x: isinstance(x, SparseTensor), lambda x: (x.F, x.C))

    def same(a, b, path):
        if isinstance(b, tuple):                       # a collated SparseTensor: (feats, coords + batch column)
            assert torch.equal(torch.as_tensor(a.F), b[0]) and torch.equal(torch.as_tensor(a.C).int(), b[1]), path
        elif isinstance(b, dict):
            assert set(a) == set(b), path
            for k in b:
                same(a[k], b[k], path + '/' + k)
        elif isinstance(b, list):
            assert len(a) == len(b), path
            for i, (x, y) in enumerate(zip(a, b)):
                same(x, y, '%s[%d]' % (path, i))
        elif isinstance(b, torch.Tensor):
            assert a.dtype == b.dtype and torch.equal(a, b), path
        elif isinstance(b, np.ndarray):
            assert np.array_equal(a, b), path
        else:
            assert a == b, path
    same(got, want, '')
