"""Pins for oracle.ts_ref (torchsparse v1.4.0 restatement).

torchsparse itself is not available, so the restatement is pinned on
independent dense identities: F.conv3d / F.conv_transpose3d on dense grids,
fp64 gradcheck, bincount means, the manual trilinear formula.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from oracle import ts_ref as R


def _dense_grid(D, B=1):
    # coords (x,y,z,b) of a full D^3 grid, shuffled
    g = np.stack(np.meshgrid(np.arange(D), np.arange(D), np.arange(D), indexing='ij'), -1).reshape(-1, 3)
    cs = []
    for b in range(B):
        cs.append(np.concatenate([g, np.full((len(g), 1), b)], 1))
    c = np.concatenate(cs).astype(np.int32)
    rng = np.random.default_rng(0)
    return c[rng.permutation(len(c))]


def _to_dense(coords, feats, D, B=1):
    # dense tensor [B, C, D(z), H(y), W(x)]
    C = feats.shape[1]
    x = torch.zeros(B, C, D, D, D, dtype=feats.dtype)
    x[coords[:, 3], :, coords[:, 2], coords[:, 1], coords[:, 0]] = feats
    return x


def test_kernel_offsets_order():
    o3 = R.get_kernel_offsets(3)
    assert o3.shape == (27, 3) and o3.dtype == np.int32
    # odd: x fastest
    assert o3[0].tolist() == [-1, -1, -1] and o3[1].tolist() == [0, -1, -1] and o3[13].tolist() == [0, 0, 0]
    o2 = R.get_kernel_offsets(2, stride=4)
    # even: z fastest, scaled by tensor stride
    assert o2.tolist() == [[0, 0, 0], [0, 0, 4], [0, 4, 0], [0, 4, 4], [4, 0, 0], [4, 0, 4], [4, 4, 0], [4, 4, 4]]


def test_sphash_fnv_known_values():
    # FNV-1a-64 over the 4 int32 words, folded: recomputed here with Python ints
    def fnv(c):
        h = 14695981039346656037
        for v in c:
            h ^= (v & 0xFFFFFFFF)
            h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        h = (h >> 60) ^ (h & 0x0FFFFFFFFFFFFFFF)
        return h if h < (1 << 63) else h - (1 << 64)
    c = np.array([[0, 0, 0, 0], [1, 2, 3, 0], [-1, 5, 7, 1], [1023, 77, 12, 3]], dtype=np.int32)
    got = R.sphash(c)
    assert got.tolist() == [fnv(r) for r in c.tolist()]
    off = np.array([[1, 0, -1], [0, 0, 0]], dtype=np.int32)
    got2 = R.sphash(c, off)
    assert got2.shape == (2, 4)
    assert got2[1].tolist() == got.tolist()
    assert got2[0].tolist() == [fnv([r[0] + 1, r[1], r[2] - 1, r[3]]) for r in c.tolist()]


def test_hashquery_miss_and_first_wins():
    ref = np.array([5, 9, 5, 100], dtype=np.int64)
    q = np.array([[9, 5], [7, 100]], dtype=np.int64)
    assert R.sphashquery(q, ref).tolist() == [[1, 0], [-1, 3]]
    assert R.sphashquery(np.zeros((0,), np.int64), ref).shape == (0,)


def test_spcount():
    assert R.spcount(np.array([0, 2, 2, -1, 1, 2]), 4).tolist() == [1, 1, 3, 0]


@pytest.mark.parametrize('B', [1, 2])
def test_subm_k3_equals_dense_conv3d(B):
    D, Cin, Cout = 6, 5, 7
    coords = _dense_grid(D, B)
    torch.manual_seed(0)
    feats = torch.randn(len(coords), Cin, dtype=torch.float64)
    W = torch.randn(27, Cin, Cout, dtype=torch.float64)
    nbmaps, nbsizes, oc, _ = R.build_kmap(coords, 1, 3, 1)
    out = R.conv_forward(feats, W, nbmaps, nbsizes, (len(coords), len(coords)))
    # k = dz*9 + dy*3 + dx with x fastest -> view(3,3,3) is (z,y,x)
    wd = W.view(3, 3, 3, Cin, Cout).permute(4, 3, 0, 1, 2)
    ref = TF.conv3d(_to_dense(coords, feats, D, B), wd, padding=1)
    got = _to_dense(coords, out, D, B)
    assert torch.allclose(got, ref, atol=1e-10)


def test_k2s2_down_and_transposed_equal_dense():
    D, Cin, Cout = 6, 4, 3
    coords = _dense_grid(D)
    torch.manual_seed(1)
    feats = torch.randn(len(coords), Cin, dtype=torch.float64)
    W = torch.randn(8, Cin, Cout, dtype=torch.float64)
    nbmaps, nbsizes, oc, _ = R.build_kmap(coords, 1, 2, 2)
    assert len(oc) == (D // 2) ** 3
    # sorted by (b,x,y,z)
    key = [(int(c[3]), int(c[0]), int(c[1]), int(c[2])) for c in oc]
    assert key == sorted(key)
    out = R.conv_forward(feats, W, nbmaps, nbsizes, (len(coords), len(oc)))
    # even kernel: k = dx*4 + dy*2 + dz (z fastest) -> view(2,2,2) is (x,y,z)
    wd = W.view(2, 2, 2, Cin, Cout).permute(4, 3, 2, 1, 0)
    ref = TF.conv3d(_to_dense(coords, feats, D), wd, stride=2)
    oc_half = oc.copy()
    oc_half[:, :3] //= 2
    got = _to_dense(oc_half, out, D // 2)
    assert torch.allclose(got, ref, atol=1e-10)
    # transposed: reuse the down map with roles swapped
    Wt = torch.randn(8, Cout, Cin, dtype=torch.float64)
    up = R.conv_forward(out, Wt, nbmaps, nbsizes, (len(coords), len(oc)), transposed=True)
    wdt = Wt.view(2, 2, 2, Cout, Cin).permute(3, 4, 2, 1, 0)
    ref_up = TF.conv_transpose3d(ref, wdt, stride=2)
    assert torch.allclose(_to_dense(coords, up, D), ref_up, atol=1e-10)


def test_conv_backward_matches_autograd_of_dense():
    D, Cin, Cout = 4, 3, 2
    coords = _dense_grid(D)
    # drop a third of the voxels: a truly sparse cloud
    coords = coords[: int(len(coords) * 0.66)]
    torch.manual_seed(2)
    feats = torch.randn(len(coords), Cin, dtype=torch.float64, requires_grad=True)
    W = torch.randn(27, Cin, Cout, dtype=torch.float64, requires_grad=True)
    nbmaps, nbsizes, _, _ = R.build_kmap(coords, 1, 3, 1)
    g = torch.randn(len(coords), Cout, dtype=torch.float64)
    gi, gw = R.conv_backward(feats.detach(), W.detach(), g, nbmaps, nbsizes)
    # autograd through an index-based dense evaluation (submanifold: only active sites are outputs)
    x = _to_dense(coords, feats, D)
    y = TF.conv3d(x, W.view(3, 3, 3, Cin, Cout).permute(4, 3, 0, 1, 2), padding=1)
    ysp = y[0][:, coords[:, 2], coords[:, 1], coords[:, 0]].t()
    (ysp * g).sum().backward()
    assert torch.allclose(gi, feats.grad, atol=1e-10)
    assert torch.allclose(gw, W.grad, atol=1e-10)


def test_voxelize_is_bincount_mean_and_grad():
    rng = np.random.default_rng(3)
    idx = rng.integers(0, 50, 400)
    idx[::17] = -1
    f = torch.randn(400, 6, dtype=torch.float64)
    counts = R.spcount(idx, 50)
    out = R.voxelize_forward(f, idx, counts)
    ref = torch.zeros(50, 6, dtype=torch.float64)
    ok = idx >= 0
    ref.index_add_(0, torch.from_numpy(idx[ok]), f[ok])
    ref = ref / torch.from_numpy(np.maximum(counts, 1)).double().unsqueeze(1)
    assert torch.allclose(out, ref, atol=1e-12)
    g = torch.randn(50, 6, dtype=torch.float64)
    gi = R.voxelize_backward(g, idx, counts, 400)
    ref_g = torch.zeros_like(f)
    ref_g[ok] = g[idx[ok]] / torch.from_numpy(counts[idx[ok]]).double().unsqueeze(1)
    assert torch.allclose(gi, ref_g, atol=1e-12)


def test_devoxelize_identity_and_trilinear():
    # integer point coords at stride 1 -> weights (1,0,...,0), identity gather
    coords = _dense_grid(4)
    pts = torch.from_numpy(coords).float()
    off = R.get_kernel_offsets(2, 1)
    q = R.sphash(coords, off)
    idx = R.sphashquery(q, R.sphash(coords))
    w = R.calc_ti_weights(pts, idx, 1)
    assert torch.allclose(w[0], torch.ones(len(coords)))
    assert float(w[1:].abs().max()) == 0.0
    f = torch.randn(len(coords), 3)
    out = R.devoxelize_forward(f, idx.T.copy(), w.t().contiguous())
    assert torch.equal(out, f)
    # stride 2: manual trilinear formula on the interior
    vox = coords.copy()
    vox[:, :3] = vox[:, :3] // 2 * 2
    vox = np.unique(vox, axis=0).astype(np.int32)
    fv = torch.randn(len(vox), 2, dtype=torch.float64)
    p = torch.tensor([[0.5, 1.0, 1.5, 0.0]], dtype=torch.float32)
    base = np.array([[0, 0, 0, 0]], dtype=np.int32)
    q = R.sphash(base, R.get_kernel_offsets(2, 2))
    idx = R.sphashquery(q, R.sphash(vox))
    w = R.calc_ti_weights(p, idx, 2).double()
    out = R.devoxelize_forward(fv, idx.T.copy(), w.t().contiguous())
    tx, ty, tz = 0.25, 0.5, 0.75
    ref = 0
    for k, (ox, oy, oz) in enumerate(R.get_kernel_offsets(2, 2).tolist()):
        wx = tx if ox else 1 - tx
        wy = ty if oy else 1 - ty
        wz = tz if oz else 1 - tz
        ref = ref + wx * wy * wz * fv[int(idx[k, 0])]
    assert torch.allclose(out[0], ref, atol=1e-6)
    # backward == transpose of forward (adjoint identity)
    g = torch.randn(1, 2, dtype=torch.float64)
    gv = R.devoxelize_backward(g, idx.T.copy(), w.t().contiguous(), len(vox))
    assert torch.allclose((gv * fv).sum(), (out * g).sum(), atol=1e-10)


def test_sparse_quantize_keeps_unique_floor():
    pts = np.array([[0.01, 0.02, 0.0], [0.04, 0.03, 0.01], [0.11, 0.0, 0.0], [-0.01, 0, 0]])
    c, ind, inv = R.sparse_quantize(pts, 0.05, return_index=True, return_inverse=True)
    assert len(c) == 3 and (c[inv] == np.floor(pts / 0.05).astype(np.int32)).all()


def test_autograd_wrappers_gradcheck():
    from oracle import torchsparse_cpu as ts
    import oracle.torchsparse_cpu.nn.functional as F
    coords = torch.from_numpy(_dense_grid(3)[:20].copy())
    x = ts.SparseTensor(torch.randn(20, 2, dtype=torch.float64, requires_grad=True), coords, 1)
    W = torch.randn(27, 2, 3, dtype=torch.float64, requires_grad=True)

    def fn(f, w):
        x.feats = f
        return F.conv3d(x, w, 3).feats
    assert torch.autograd.gradcheck(fn, (x.feats, W), atol=1e-6)
