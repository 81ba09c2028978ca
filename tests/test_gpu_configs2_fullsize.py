"""BASELINE.json configs[2] AT FULL SIZE on one GPU: "SPVCNN teacher + SwiftNet18 student + KD loss (spformer_tsd_full),
synthetic 6-cam 900x1600 + 80k voxels" -- the cr 1.0 student / cr_t 2.0 teacher of
configs/nuscenes/train/spformer_tsd_full_ours_star.yaml:32-43, one 80 000-point scene, six cameras at the literal
900 x 1600, fp32 (the bench's secondary line `kd_6cam_900x1600` times this step; its headline line runs the same step at
360 x 640).  The CPU oracle cannot run this size, so the test holds size-independent properties (small-scene parity
against the reference's own classes: test_kd_path.py; per operator: test_gpu_torchsparse_ops.py; the pixel head at this
camera size: test_gpu_pixel_head.py):
  * the LiDAR side is run-to-run reproducible: the student's voxel set / kernel-map geometry bit for bit (order-deterministic
    kernels, no atomics), the frozen teacher's logits bit for bit (torch.equal over every repetition, round 5); the student's outputs, which depend on MIOpen's
    convolutions (not reproducible run to run), to rounding noise;
  * every student parameter receives a finite gradient, the frozen teacher none; every loss term is finite;
  * the teacher -> student re-index (core/nusc_trainers.py:295-324) selects exactly the rows the reference's chained
    indexings select."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_VOX, HW = 80000, (900, 1600)


@pytest.fixture(scope='module')
def world(hip):
    import os
    os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')          # (bounds MIOpen's first-call kernel search at this image size)
    from u2mkd_amd import train as T
    from u2mkd_amd.synth import synth_kd_batch
    from test_gpu_configs import _runner
    nb = synth_kd_batch(N_VOX, 1, seed=1234, image_hw=HW)
    d = T.kd_batch_to_device(nb)
    run = _runner(1.0, 2.0)
    state = {k: v.clone() for k, v in run.model.state_dict().items()}
    return nb, d, run, state


def _step(run, d):
    from test_gpu_configs4_fullsize import _step as step
    return step(run, d, False)


def test_scene_is_configs2_sized(world):
    nb, d = world[0], world[1]
    assert nb['student']['num_vox'][0] > 70000 and nb['teacher']['num_pts'][0] == N_VOX
    assert d['images'].shape == (1, 6, 3) + HW


def test_fp32_step_reproduces_and_trains_every_student_parameter(world):
    nb, d, run, state = world
    res = []
    REPS = 12
    for _ in range(REPS):         # (the first pass also settles MIOpen's solver choice)
        run.model.load_state_dict(state)
        out, ld = _step(run, d)
        res.append((out['t']['x_vox'].clone(), out['stu']['x_vox'].detach().clone(), out['stu']['x_pix'].detach().clone(),
                    {k: (torch.stack(list(v)) if isinstance(v, (list, tuple)) else v).detach().clone() for k, v in ld.items()}))
    assert res[0][0].shape[0] == nb['teacher']['num_vox'][0] and res[0][1].shape == (nb['student']['num_vox'][0], 17)
    # The frozen teacher (core/nusc_trainers.py:285-324: under no_grad, eval-mode BatchNorm): LiDAR operators only, every kernel
    # order-deterministic -- BIT-IDENTICAL logits in every repetition, next to the student's streams.  (Rounds 3-4 saw ~1 step in
    # 13 deviate here and bounded a band instead; the cause was gfx950's v_mfma_f32_16x16x32_bf16 executing next to other
    # kernels' waves -- NOTES N9, tools/repro_concurrent_kernels.hip -- and the library no longer issues that instruction.)
    for i in range(1, REPS):
        assert torch.equal(res[0][0], res[i][0]), ('teacher logits of repetition %d differ from repetition 0' % i,
                                                   float((res[0][0] - res[i][0]).abs().max()))
    # the student sits behind MIOpen's convolutions, whose outputs differ in the last places between two identical forwards
    # (DESIGN.md section 7b); ~60 layers with batch statistics carry that to the logits: stated bound = the median element
    # within 1e-3 of the logit range, at most 2 % of the elements beyond 1e-2 of it (measured: printed)
    for i, name in ((1, 'x_vox'), (2, 'x_pix')):
        a, b = res[1][i], res[2][i]
        dlt, scale = (a - b).abs(), max(1.0, float(b.abs().max()))
        far = float((dlt > 1e-2 * scale).float().mean())
        print('CONFIGS2-REPRO %s: median %.2e max %.2e of range %.1f; elements beyond 1e-2 of the range: %.4f'
              % (name, float(dlt.median()), float(dlt.max()), scale, far))
        assert float(dlt.median()) <= 1e-3 * scale and far <= 0.02, (name, float(dlt.median()), far)
    for k in res[1][3]:
        assert bool(torch.isfinite(res[1][3][k]).all()), k
        assert torch.allclose(res[1][3][k], res[2][3][k], rtol=2e-2, atol=1e-4), (k, res[1][3][k], res[2][3][k])
    for n, p in run.model.model_s.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
    assert all(p.grad is None for p in run.model.model_t.parameters())


def test_student_geometry_is_bit_reproducible(world):
    """voxel set, kernel maps and tile schedules of the student's stride-1 level built twice from the same coordinates:
    identical tensors (hash table, rulebook compaction, counting sorts: no atomics on the data path)."""
    from u2mkd_amd.torchsparse.nn import functional as F
    d = world[1]
    a = F.build_kmap(d['s_coords'], (1, 1, 1), (3, 3, 3), (1, 1, 1))
    b = F.build_kmap(d['s_coords'].clone(), (1, 1, 1), (3, 3, 3), (1, 1, 1))
    assert torch.equal(a.nbr, b.nbr)
    sa, sb = a.schedule(False), b.schedule(False)
    assert torch.equal(sa.order, sb.order) and torch.equal(sa.nbr_s, sb.nbr_s)
    assert int(sa.n_items) == int(sb.n_items) and torch.equal(sa.items[:int(sa.n_items)], sb.items[:int(sb.n_items)])
    pa, pb = a.pair_schedule(), b.pair_schedule()
    p_pad = int(pa.meta[0])                      # (the lists are allocated at their capacity: entries past P_pad are not written)
    assert torch.equal(pa.meta, pb.meta) and torch.equal(pa.pair_in[:p_pad], pb.pair_in[:p_pad]) and torch.equal(pa.pos_out, pb.pos_out)


def test_reindex_selects_the_reference_rows(world):
    """x_t[inverse_map][inds] per sample (core/nusc_trainers.py:295-324) as the product's single gather"""
    from u2mkd_amd import kd as KD
    nb, d, _, _ = world
    t, s = nb['teacher'], nb['student']
    ids = torch.arange(t['num_vox'][0], device='cuda').view(-1, 1)
    got = KD.teacher_to_student(ids, d['inverse_map'], d['inds'], d['num_pts'], d['num_vox_t'], d.get('keyframe_mask_full'))
    want = np.arange(t['num_vox'][0])[t['inverse_map']][s['inds'][0][0]]
    assert got.shape[0] == s['num_vox'][0] and np.array_equal(got.view(-1).cpu().numpy(), want)
