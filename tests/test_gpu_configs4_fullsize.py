"""BASELINE.json configs[4] AT FULL SIZE on one GPU: "Multi-sweep=4 (~300k pts/scene) SphereFormer+SwiftNet KD, bf16":
one 300 000-point scene of 9 aggregated sweeps for the cr_t 2.0 teacher (multisweeps.num_sweeps = 4 => 2*4+1 sweeps,
core/datasets/lc_semantic_nusc_tsd_full.py:241-310), its key frame for the cr 2.0 student of
configs/nuscenes/train/spformer_tsd_full_ours_star_B.yaml:34-36, six 360x640 cameras, the step under bf16 autocast with
BF16 STORAGE between the sparse operators.  The CPU oracle cannot run this size, so the test holds size-independent
properties (the per-operator and small-scene parity against the oracle / the reference goldens is in
test_gpu_bf16_rows.py, test_golden_teacher_multisweep.py, test_kd_path.py):
  * run-to-run reproducibility of the bf16 step from one state: the frozen teacher (this package's order-deterministic
    kernels throughout, SphereFormer's Linear layers included) bit for bit over ten repetitions beside the student's streams;
    the student, whose camera branch runs
    MIOpen convolutions that are not run-to-run reproducible (measured: the first module whose output differs between two
    identical forwards is pix_branch.layer2.0.conv1), within rounding noise;
  * every student parameter receives a finite fp32 gradient, the frozen teacher none;
  * the bf16 step stays within a stated bound of the fp32 step from the same state (logits and losses);
  * the teacher -> student re-index with key-frame masking (core/nusc_trainers.py:288-324) selects exactly the rows the
    reference's three chained index operations select, whatever the storage type."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_PTS, SWEEPS, HW = 300000, 9, (360, 640)


@pytest.fixture(scope='module')
def world(hip):
    from u2mkd_amd import train as T
    from u2mkd_amd.synth import synth_kd_batch
    from test_gpu_configs import _runner
    nb = synth_kd_batch(N_PTS, 1, seed=1234, image_hw=HW, sweeps=SWEEPS)
    d = T.kd_batch_to_device(nb)
    run = _runner(2.0, 2.0, amp='bf16')
    state = {k: v.clone() for k, v in run.model.state_dict().items()}
    return nb, d, run, state


def _step(run, d, amp):
    """forward + losses + backward of KDStep.__call__ (no optimizer step), keeping the outputs"""
    from u2mkd_amd import kd as KD, torchsparse as ts
    stu = {'lidar': ts.SparseTensor(d['s_feats'], d['s_coords']), 'images': d['images'],
           'pixel_coordinates': d['pixel_coordinates'], 'masks': d['masks'], 'fov_mask': d['fov_mask']}
    tea = {'lidar': ts.SparseTensor(d['t_feats'], d['t_coords'])}
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        out = run.net({'student': stu, 'teacher': tea})
        ld = KD.kd_losses(out, d['targets'], d['fov_mask'], d['inverse_map'], d['inds'], d['num_pts'], d['num_vox_t'],
                          run.crit, d['keyframe_mask_full'])
    run.opt.zero_grad()
    ld['total'].backward()
    torch.cuda.synchronize()
    return out, ld


def test_scene_is_configs4_sized(world):
    nb = world[0]
    t, s = nb['teacher'], nb['student']
    assert t['num_pts'][0] == N_PTS and t['num_vox'][0] > 250000
    kf = t['keyframe_mask_full']
    assert 15000 < int(kf.sum()) < 40000 and s['num_vox'][0] < int(kf.sum()) + 1      # the student sees the key frame only


def test_bf16_step_is_reproducible_and_trains_every_student_parameter(world):
    nb, d, run, state = world
    res = []
    REPS = 10
    for _ in range(REPS):         # (the first pass also settles MIOpen's solver choice)
        run.model.load_state_dict(state)
        out, ld = _step(run, d, True)
        res.append((out['t']['x_vox'].clone(), out['stu']['x_vox'].detach().float().clone(),
                    {k: (torch.stack(list(v)) if isinstance(v, (list, tuple)) else v).detach().float().clone() for k, v in ld.items()}))
    assert res[0][0].shape[0] == nb['teacher']['num_vox'][0]
    # teacher: every kernel on its path is order-deterministic (this package's kernels throughout, SphereFormer's Linear layers
    # included): bit-identical logits in every repetition, next to the student's streams (NOTES N9 for what rounds 3-4 saw here)
    for i in range(1, REPS):
        assert torch.equal(res[0][0], res[i][0]), ('teacher logits of repetition %d differ from repetition 0' % i,
                                                   float((res[0][0].float() - res[i][0].float()).abs().max()))
    # student: bf16 logits of two runs agree to within two bf16 steps on all but a sliver of the rows (a last-place
    # difference upstream of a rounding edge moves a value by one bf16 step), the loss terms to 3e-3 relative
    a, b = res[1][1], res[2][1]
    step = 2.0 ** -7 * float(b.abs().max())
    far = ((a - b).abs().max(1).values > 2 * step).float().mean()
    # (seen over ~15 runs of this test: 0 .. 0.3 % of the rows, once 1.0 % -- the run-to-run band of the library convolutions
    # behind the camera branch, amplified by the bf16 rounding edges; the bound states that band, it is not a parity gate)
    assert float(far) < 0.02, float(far)
    for k in res[1][2]:
        # (observed between two runs: up to 6e-4 on the KL term, which sees the teacher rows that moved by a bf16 step, and
        # 8e-4 on the deepest stage's MSE term; the others 1e-4)
        assert torch.allclose(res[1][2][k], res[2][2][k], rtol=3e-3, atol=1e-6), (k, res[1][2][k], res[2][2][k])
    assert all(bool(torch.isfinite(v).all()) for v in res[0][2].values())
    for n, p in run.model.model_s.named_parameters():
        assert p.dtype == torch.float32 and p.grad is not None and p.grad.dtype == torch.float32, n
        assert bool(torch.isfinite(p.grad).all()), n
    assert all(p.grad is None for p in run.model.model_t.parameters())


def test_bf16_step_stays_within_a_stated_bound_of_the_fp32_step(world):
    """Stated bound (bf16 has 8 significant bits; ~60 layers deep): teacher and student logits within 15 % of the
    logit range at the worst element and 2 % at the median, every loss term within 6 % (the tiny MSE terms 15 %)."""
    nb, d, run, state = world
    run.model.load_state_dict(state)
    out_b, ld_b = _step(run, d, True)
    xb_t, xb_s = out_b['t']['x_vox'].float().clone(), out_b['stu']['x_vox'].detach().float().clone()
    lb = {k: ([float(x) for x in v] if isinstance(v, (list, tuple)) else float(v)) for k, v in ld_b.items()}
    del out_b, ld_b
    run.model.load_state_dict(state)
    out_f, ld_f = _step(run, d, False)
    for name, b, f in (('teacher', xb_t, out_f['t']['x_vox'].float()), ('student', xb_s, out_f['stu']['x_vox'].detach().float())):
        dlt = (b - f).abs()
        scale = float(f.abs().max())
        assert float(dlt.max()) < 0.15 * scale and float(dlt.median()) < 0.02 * scale, \
            (name, float(dlt.max()), float(dlt.median()), scale)
    for k in ('ce_vox', 'ce_pix', 'kl', 'feat', 'total'):
        assert abs(lb[k] - float(ld_f[k])) < 0.06 * abs(float(ld_f[k])) + 1e-4, (k, lb[k], float(ld_f[k]))
    for a, b in zip(lb['mse'], ld_f['mse']):
        assert abs(a - float(b)) < 0.15 * abs(float(b)) + 1e-4, (lb['mse'], [float(x) for x in ld_f['mse']])


def test_keyframe_reindex_selects_the_reference_rows(world):
    """x_t[inverse_map][keyframe_mask][inds] (core/nusc_trainers.py:295-324) as the product's single gather: fed with
    the teacher's row ids it returns, per student voxel, exactly the ids the three chained numpy indexings give."""
    from u2mkd_amd import kd as KD
    nb, d, _, _ = world
    t, s = nb['teacher'], nb['student']
    ids = torch.arange(t['num_vox'][0], device='cuda').view(-1, 1)
    got = KD.teacher_to_student(ids, d['inverse_map'], d['inds'], d['num_pts'], d['num_vox_t'], d['keyframe_mask_full'])
    want = np.arange(t['num_vox'][0])[t['inverse_map']][t['keyframe_mask_full']][s['inds'][0][0]]
    assert got.shape[0] == s['num_vox'][0] and np.array_equal(got.view(-1).cpu().numpy(), want)
    # every selected teacher voxel holds at least one key-frame point
    vox_has_kf = np.zeros(t['num_vox'][0], bool)
    vox_has_kf[t['inverse_map'][t['keyframe_mask_full']]] = True
    assert vox_has_kf[want].all()
