"""Size-independent properties at BASELINE.json's full size (80 000 voxels), where the CPU oracle of
the whole network is too slow to be the checker: linearity and permutation equivariance of the
sparse conv on both schedules, voxelise/devoxelise round trip, run-to-run determinism of the full
SPVCNN cr=1.0 training step."""
import numpy as np
import pytest
import torch

from u2mkd_amd.synth import synth_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def F(hip):
    from u2mkd_amd.torchsparse.nn import functional as F
    return F


@pytest.fixture(scope='module')
def scene():
    return synth_batch(80000, 1)


@pytest.mark.parametrize('cin,cout', [(64, 64), (96, 96)])      # tile schedule / pair schedule
def test_conv_is_linear_and_permutation_equivariant(F, scene, cin, cout):
    coords = torch.from_numpy(scene['coords']).cuda()
    n = coords.shape[0]
    km = F.build_kmap(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    g = torch.Generator(device='cuda').manual_seed(1)
    x, y = torch.randn(n, cin, device='cuda', generator=g), torch.randn(n, cin, device='cuda', generator=g)
    w = torch.randn(27, cin, cout, device='cuda', generator=g) / (27 * cin) ** 0.5
    conv = lambda t, m=km: F.ConvolutionFunction.apply(t, w, m, False)
    a, b = 0.75, -1.5
    lhs = conv(a * x + b * y)
    rhs = a * conv(x) + b * conv(y)
    assert float((lhs - rhs).abs().max()) < 2e-5 * max(1.0, float(rhs.abs().max()))
    # permuting the voxels permutes the output rows (the map is rebuilt from the permuted coordinates)
    perm = torch.randperm(n, device='cuda', generator=g)
    km_p = F.build_kmap(coords[perm].contiguous(), (1, 1, 1), (3, 3, 3), (1, 1, 1))
    out_p = conv(x[perm].contiguous(), km_p)
    assert float((out_p - conv(x)[perm]).abs().max()) < 2e-5 * max(1.0, float(out_p.abs().max()))
    # an all-zero input gives exactly zero, a centre-only kernel is a per-voxel linear map
    assert float(conv(torch.zeros_like(x)).abs().max()) == 0.0
    wc = torch.zeros_like(w)
    wc[13] = w[13]
    out_c = F.ConvolutionFunction.apply(x, wc, km, False)
    assert float((out_c - x @ w[13]).abs().max()) < 2e-5 * max(1.0, float(out_c.abs().max()))


def test_voxelize_devoxelize_round_trip(F, scene):
    """Nearest devoxelise of the voxel means of one-point-per-voxel features returns the features."""
    coords = torch.from_numpy(scene['coords']).cuda()
    n = coords.shape[0]
    feats = torch.randn(n, 32, device='cuda')
    h = F.sphash(coords)
    uniq = torch.unique(h)
    idx = F.sphashquery(h, uniq)
    counts = F.spcount(idx.int(), len(uniq))
    assert int(counts.max()) == 1 and len(uniq) == n          # the synthetic scene has unique voxels
    vox = F.spvoxelize(feats, idx, counts)
    idx8 = torch.full((n, 8), -1, dtype=torch.int32, device='cuda')
    idx8[:, 0] = idx.int()
    w8 = torch.zeros(n, 8, device='cuda')
    w8[:, 0] = 1.0
    back = F.spdevoxelize(vox, idx8, w8)
    assert torch.equal(back, feats)


def test_full_size_training_step_is_deterministic(hip, scene):
    """Two runs of the configs[1] step from the same initial state: identical loss and logits, bit for
    bit (no atomics anywhere on the LiDAR path: ordered reductions in conv, wgrad, BatchNorm, CSR sums)."""
    from u2mkd_amd import lidar, train as T
    from u2mkd_amd import torchsparse as ts
    feats, coords, labels = (torch.from_numpy(scene[k]).cuda() for k in ('feats', 'coords', 'labels'))
    torch.manual_seed(0)
    model = lidar.SPVCNN(cr=1.0, in_channel=4, num_classes=17, pres=0.05, vres=0.05).cuda().train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    res = []
    for _ in range(2):
        model.load_state_dict(state)
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0                                      # dropout draws differ between runs by design
        run = T.LidarStep(model)
        l0 = run(feats, coords, labels)
        l1 = run(feats, coords, labels)
        with torch.no_grad():
            out = model.eval()({'lidar': ts.SparseTensor(feats, coords)})['x_vox']
        model.train()
        res.append((l0.clone(), l1.clone(), out.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[0][2], res[1][2])
    assert np.isfinite(float(res[0][1])) and float(res[0][1]) < float(res[0][0]) + 0.5
