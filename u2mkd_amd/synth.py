"""Synthetic nuScenes-shaped LiDAR scenes (SURVEY.md §8d "Synthetic scene").

Reproduces the *output schema* of the reference loader
(core/datasets/lc_semantic_nusc_tsd_full.py:415-433: ``voxel = round(xyz /
0.05) - min``, de-duplicated keeping the first point, features
``[x, y, z (m), intensity]`` un-normalised, label 0 = ignore) without the
300 GB dataset.  numpy only; deterministic in ``seed``.
"""
from __future__ import annotations

import numpy as np

__all__ = ['synth_scene', 'synth_batch', 'synth_kd_batch', 'project_to_cameras', 'CLASS_DISTRIBUTE', 'kmap_stats']

# class frequencies in the spirit of lc_semantic_nusc_tsd_full.py:114-116
# (17 classes, index 0 = ignore).
CLASS_DISTRIBUTE = np.array(
    [2.0, 0.2, 0.05, 0.6, 5.0, 0.2, 0.05, 0.3, 0.1, 1.0, 3.0, 35.0, 1.0, 8.0, 8.0, 20.0, 15.0],
    dtype=np.float64)
CLASS_DISTRIBUTE = CLASS_DISTRIBUTE / CLASS_DISTRIBUTE.sum()

VOXEL_SIZE = 0.05


def _lidar_sweep(rng: np.random.Generator, n_az: int, n_beams: int = 32):
    """One LiDAR-like sweep: 32 elevation beams in [-30, +10] deg x n_az
    azimuth steps; range = nearest of ground plane (z = -1.8 m), 64 radial wall
    sectors and a few random boxes, out to 50 m; sigma = 2 cm range noise."""
    elev = np.deg2rad(np.linspace(-30.0, 10.0, n_beams))
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False) + rng.uniform(0, 2 * np.pi / n_az)
    el, a = np.meshgrid(elev, az, indexing='ij')
    # per-ray angular jitter (beam divergence / encoder noise, ~0.03 deg): without it whole beams sit at
    # exactly representable angles, i.e. exactly ON SphereFormer's 2-degree window boundaries
    el = el + rng.normal(0.0, 5e-4, el.shape)
    a = a + rng.normal(0.0, 5e-4, a.shape)
    dx, dy, dz = np.cos(el) * np.cos(a), np.cos(el) * np.sin(a), np.sin(el)
    r = np.full(el.shape, 50.0)
    down = dz < -1e-3
    r_ground = np.where(down, -1.8 / np.where(down, dz, -1.0), np.inf)
    r = np.minimum(r, r_ground)
    # radial wall sectors (buildings): per azimuth sector a wall at a random distance
    n_sec = 64
    wall_r = rng.uniform(8.0, 45.0, n_sec)
    wall_on = rng.uniform(size=n_sec) < 0.6
    sec = ((a + np.pi) / (2 * np.pi) * n_sec).astype(np.int64) % n_sec
    r_wall = np.where(wall_on[sec], wall_r[sec] / np.maximum(np.cos(el), 1e-3), np.inf)
    wall_top = rng.uniform(2.0, 8.0, n_sec)
    z_hit = r_wall * dz
    r_wall = np.where(z_hit < wall_top[sec], r_wall, np.inf)
    r = np.minimum(r, r_wall)
    # a few boxes (vehicles): ray / axis-aligned box slab test
    for _ in range(12):
        c = np.array([rng.uniform(-30, 30), rng.uniform(-30, 30), -1.0])
        half = np.array([rng.uniform(0.8, 2.5), rng.uniform(0.8, 2.5), 0.8])
        lo, hi = c - half, c + half
        with np.errstate(divide='ignore', invalid='ignore'):
            t1 = np.stack([lo[0] / dx, lo[1] / dy, lo[2] / dz])
            t2 = np.stack([hi[0] / dx, hi[1] / dy, hi[2] / dz])
        tmin = np.nanmax(np.minimum(t1, t2), axis=0)
        tmax = np.nanmin(np.maximum(t1, t2), axis=0)
        hit = (tmax >= tmin) & (tmin > 0.5)
        r = np.where(hit, np.minimum(r, tmin), r)
    valid = r < 49.9
    r = r + rng.normal(0.0, 0.02, r.shape)
    xyz = np.stack([r * dx, r * dy, r * dz], -1)[valid]
    return xyz.astype(np.float32)


def synth_scene(n_vox: int, seed: int = 1234, sweeps: int | None = None):
    """Return a dict for one scene with exactly ``n_vox`` unique voxels:

    ``coords`` int32 [n_vox,3] (voxel = round(xyz/0.05) - min), ``feats`` f32
    [n_vox,4] = (x,y,z metres, intensity U[0,255)), ``labels`` int64 [n_vox]
    (0 = ignore), ``keyframe`` bool [n_vox] (only sweep 0 is the key frame).

    Azimuth density is the real sensor's (1085 steps x 32 beams = 34 720 rays
    per sweep); larger scenes aggregate rigidly shifted sweeps (<= 0.5 m per
    sweep) like the reference's multi-sweep loader
    (lc_semantic_nusc_tsd_full.py:241-310).  ``sweeps=None`` adds sweeps until
    ``n_vox`` unique voxels exist.
    """
    rng = np.random.default_rng(seed)
    n_az = 1085
    clouds = []
    s = 0
    while True:
        p = _lidar_sweep(rng, n_az)
        shift = np.array([0.37 * s, 0.11 * s, 0.0], dtype=np.float32)  # <= 0.5 m / sweep
        clouds.append((p + shift, np.full(len(p), s == 0)))
        s += 1
        if sweeps is not None and s < sweeps:
            continue
        xyz = np.concatenate([c[0] for c in clouds])
        kf = np.concatenate([c[1] for c in clouds])
        vox = np.round(xyz / VOXEL_SIZE).astype(np.int32)
        vox -= vox.min(0, keepdims=True)
        # sparse_quantize semantics: keep the first point of every voxel
        key64 = (vox[:, 0].astype(np.int64) << 40) | (vox[:, 1].astype(np.int64) << 20) | vox[:, 2].astype(np.int64)
        _, first = np.unique(key64, return_index=True)
        first.sort()
        if len(first) >= n_vox:
            break
        if s > 64:
            raise RuntimeError(f'synth_scene could not reach {n_vox} voxels')
    sel = first[np.sort(rng.choice(len(first), n_vox, replace=False))]
    xyz, vox, kf = xyz[sel], vox[sel], kf[sel]
    vox = vox - vox.min(0, keepdims=True)
    inten = rng.uniform(0, 255, (n_vox, 1)).astype(np.float32)
    feats = np.concatenate([xyz, inten], 1).astype(np.float32)
    labels = rng.choice(17, n_vox, p=CLASS_DISTRIBUTE).astype(np.int64)
    return {'coords': np.ascontiguousarray(vox), 'feats': feats, 'labels': labels,
            'keyframe': kf, 'sweeps': s}


def synth_batch(n_vox: int, batch: int = 1, seed: int = 1234, sweeps: int | None = None):
    """Collate ``batch`` scenes like sparse_collate (batch index LAST column)."""
    cs, fs, ls, ks, num = [], [], [], [], []
    for b in range(batch):
        s = synth_scene(n_vox, seed + b, sweeps)
        cs.append(np.concatenate([s['coords'], np.full((n_vox, 1), b, np.int32)], 1))
        fs.append(s['feats'])
        ls.append(s['labels'])
        ks.append(s['keyframe'])
        num.append(n_vox)
    return {'coords': np.concatenate(cs), 'feats': np.concatenate(fs), 'labels': np.concatenate(ls),
            'keyframe': np.concatenate(ks), 'num_vox': num}


def kmap_stats(coords4: np.ndarray):
    """N and pairs-per-voxel (k-bar) of the stride-1 k=3 map, via a set lookup
    (independent of the oracle and the HIP path; used to report bytes)."""
    c = coords4.astype(np.int64)
    key = (c[:, 3] << 54) | ((c[:, 0] + 1) << 36) | ((c[:, 1] + 1) << 18) | (c[:, 2] + 1)
    s = np.sort(key)
    pairs = 0
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                q = key + (dx << 36) + (dy << 18) + dz
                pos = np.minimum(np.searchsorted(s, q), len(s) - 1)
                pairs += int((s[pos] == q).sum())
    return len(c), pairs, pairs / max(len(c), 1)


# --------------------------------------------------------------------------- KD batch
_CAM_YAW_DEG = (55.0, 0.0, -55.0, 110.0, 180.0, -110.0)


def project_to_cameras(xyz: np.ndarray, image_hw=(900, 1600)):
    """6 pinhole cameras (yaw +55, 0, -55, +110, 180, -110 deg; fx = fy = 1266, cx = 816,
    cy = 491 on a 1600x900 sensor, mounted 1.5 m above the LiDAR origin).  Returns
    ``pixel_coordinates`` f32 [6,N,2] normalised to [-1,1] by (W-1, H-1) and ``masks`` bool
    [6,N] = depth > 1 m and strictly inside (-1,1)^2 -- the schema of
    core/datasets/lc_semantic_nusc_tsd_full.py:344-387."""
    H, W = 900, 1600
    fx = fy = 1266.0
    cx, cy = 816.0, 491.0
    n = xyz.shape[0]
    pix = np.zeros((6, n, 2), dtype=np.float32)
    masks = np.zeros((6, n), dtype=bool)
    for c, yaw in enumerate(_CAM_YAW_DEG):
        a = np.deg2rad(yaw)
        fwd = np.array([np.cos(a), np.sin(a), 0.0])
        left = np.array([-np.sin(a), np.cos(a), 0.0])
        up = np.array([0.0, 0.0, 1.0])
        p = xyz[:, :3].astype(np.float64) - np.array([0.0, 0.0, 1.5])
        depth = p @ fwd
        xr = -(p @ left)
        yd = -(p @ up)
        with np.errstate(divide='ignore', invalid='ignore'):
            u = fx * xr / depth + cx
            v = fy * yd / depth + cy
        un = u / (W - 1) * 2 - 1
        vn = v / (H - 1) * 2 - 1
        ok = (depth > 1.0) & (un > -1) & (un < 1) & (vn > -1) & (vn < 1)
        pix[c, :, 0] = np.where(ok, un, 0.0)
        pix[c, :, 1] = np.where(ok, vn, 0.0)
        masks[c] = ok
    return pix, masks


def _yaw(xyz, deg):
    a = np.deg2rad(deg)
    r = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=np.float32)
    return xyz @ r.T


def synth_kd_batch(n_vox: int, batch: int = 1, seed: int = 1234, image_hw=(360, 640), ncam: int = 6,
                   sweeps: int | None = None):
    """Student + teacher feed dicts with the collate schema of
    core/datasets/lc_semantic_nusc_tsd_full.py:436-486, numpy payloads.

    One LiDAR scene per sample.  The student sees it under one random yaw and its voxels in a
    shuffled order; the teacher sees the same key-frame points under another yaw, re-voxelised
    (so points may merge: ``inverse_map`` point -> teacher voxel).  ``inds`` maps every student
    voxel to its key-frame point, as core/nusc_trainers.py:295-324 consumes it.

    ``sweeps`` = k > 1: the TEACHER sees k aggregated sweeps (BASELINE.json configs[4]; the reference's
    ``multisweeps``, lc_semantic_nusc_tsd_full.py:198-206): ``n_vox`` then counts the voxels of the aggregate, the
    student sees the key-frame points only, the teacher dict carries ``keyframe_mask_full`` (over all points;
    ``num_pts`` = all points) and the re-index of core/nusc_trainers.py:288-324 selects the key-frame rows."""
    rng = np.random.default_rng(seed + 7919)
    H, W = image_hw
    S = {'coords': [], 'feats': [], 'targets': [], 'pixel_coordinates': [], 'masks': [], 'fov_mask': [], 'inds': [],
         'num_vox': []}
    T = {'coords': [], 'feats': [], 'targets': [], 'inverse_map': [], 'num_pts': [], 'num_vox': [], 'keyframe_mask_full': []}
    for b in range(batch):
        sc = synth_scene(n_vox, seed + b, sweeps)
        xyz_all, inten_all, labels_all = sc['feats'][:, :3], sc['feats'][:, 3:], sc['labels']
        kf = sc['keyframe'] if (sweeps or 0) > 1 else np.ones(len(xyz_all), bool)
        # the student (and the re-index `inds`) live on the key-frame points only
        xyz, inten, labels = xyz_all[kf], inten_all[kf], labels_all[kf]
        # ---- student: yaw, voxelise (unique by construction up to rounding), shuffled voxel order
        xs = _yaw(xyz, rng.uniform(-180, 180))
        vs = np.round(xs / VOXEL_SIZE).astype(np.int32)
        vs -= vs.min(0, keepdims=True)
        key = (vs[:, 0].astype(np.int64) << 40) | (vs[:, 1].astype(np.int64) << 20) | vs[:, 2].astype(np.int64)
        _, first = np.unique(key, return_index=True)
        inds = rng.permutation(first)                      # student voxel -> key-frame point
        pix, masks = project_to_cameras(xs[inds])
        S['coords'].append(np.concatenate([vs[inds], np.full((len(inds), 1), b, np.int32)], 1))
        S['feats'].append(np.concatenate([xs[inds], inten[inds]], 1).astype(np.float32))
        S['targets'].append(labels[inds])
        S['pixel_coordinates'].append(pix[:ncam])
        S['masks'].append(masks[:ncam])
        S['fov_mask'].append(masks[:ncam].any(0))
        S['inds'].append([inds.astype(np.int64)])
        S['num_vox'].append(len(inds))
        # ---- teacher: another yaw, re-voxelised; inverse_map over ALL key-frame points
        xt = _yaw(xyz_all, rng.uniform(-180, 180))
        vt = np.round(xt / VOXEL_SIZE).astype(np.int32)
        vt -= vt.min(0, keepdims=True)
        keyt = (vt[:, 0].astype(np.int64) << 40) | (vt[:, 1].astype(np.int64) << 20) | vt[:, 2].astype(np.int64)
        _, firstt, inv = np.unique(keyt, return_index=True, return_inverse=True)
        T['coords'].append(np.concatenate([vt[firstt], np.full((len(firstt), 1), b, np.int32)], 1))
        T['feats'].append(np.concatenate([xt[firstt], inten_all[firstt]], 1).astype(np.float32))
        T['targets'].append(np.where(kf, labels_all, 0)[firstt])      # non-key-frame points carry the ignore label
        T['inverse_map'].append(inv.astype(np.int64))
        T['num_pts'].append(len(xyz_all))
        T['num_vox'].append(len(firstt))
        T['keyframe_mask_full'].append(kf)
    images = rng.uniform(0, 255, (batch, ncam, H, W, 3)).astype(np.float32)
    student = {'coords': np.concatenate(S['coords']), 'feats': np.concatenate(S['feats']),
               'targets': np.concatenate(S['targets']), 'images': images,
               'pixel_coordinates': S['pixel_coordinates'], 'masks': S['masks'],
               'fov_mask': np.concatenate(S['fov_mask']), 'inds': S['inds'], 'num_vox': S['num_vox']}
    teacher = {'coords': np.concatenate(T['coords']), 'feats': np.concatenate(T['feats']),
               'targets': np.concatenate(T['targets']), 'inverse_map': np.concatenate(T['inverse_map']),
               'num_pts': T['num_pts'], 'num_vox': T['num_vox']}
    if (sweeps or 0) > 1:
        teacher['keyframe_mask_full'] = np.concatenate(T['keyframe_mask_full'])
    return {'student': student, 'teacher': teacher}


def synth_eval_feed(b, seed: int):
    """The pieces of the eval feed dict that synth_kd_batch does not carry (core/datasets/
    lc_semantic_nusc_tsd_full.py:436-486, consumed by core/nusc_trainers.py:367-418): the student's raw points ->
    voxel map (``inverse_map``: index within the scene) with their labels and in-view labels, and the scene index /
    labels of the teacher's raw points.  Random but seeded (the golden generator and the tests build the same)."""
    rng = np.random.default_rng(seed)
    s, t = b['student'], b['teacher']
    inv, ib = [], []
    for i, nv in enumerate(s['num_vox']):
        n_pts = int(1.3 * nv)
        inv.append(rng.integers(0, nv, n_pts))
        ib.append(np.full(n_pts, i))
    inv, ib = np.concatenate(inv).astype(np.int64), np.concatenate(ib).astype(np.int64)
    tb = np.concatenate([np.full(n, i) for i, n in enumerate(t['num_pts'])]).astype(np.int64)
    return {'s_inverse_map': inv, 's_inverse_batch': ib, 'targets_mapped': rng.integers(0, 17, len(inv)),
            'label_fov': rng.integers(0, 17, len(inv)), 't_inverse_batch': tb,
            'targets_mapped_t': rng.integers(0, 17, len(tb))}
