"""Which LIBRARY kernels disturb another stream's kernels the way gfx950's v_mfma_f32_16x16x32_bf16 does (NOTES N9)?
Victims on the main stream (u2mkd_ti_weights + an element-wise division, compared bit for bit with their results on an idle GPU),
one kind of library call looping on a side stream: MIOpen convolutions (fp32 / bf16, forward and both gradients), rocBLAS /
hipBLASLt matrix products (fp32 / bf16), this library's own conv kernels.   python tools/dbg_library_aggressors.py [rounds=200]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')
import torch
from u2mkd_amd import _lib as L
from u2mkd_amd.synth import synth_batch
from u2mkd_amd.torchsparse.nn import functional as F

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
b = synth_batch(80000, 1, seed=1234)
coords = torch.from_numpy(b['coords']).cuda()
n = coords.shape[0]
g = torch.Generator(device='cuda').manual_seed(0)
pts = (coords.float() * 0.05) / 0.05
idx_kn = torch.randint(-1, n, (8, n), device='cuda', generator=g, dtype=torch.int64)
idx_kn[idx_kn % 3 == 0] = -1
num = torch.rand(n, 16, device='cuda', generator=g) * 50 + 0.5


def victims():
    w, i8 = F.ti_weights_n8(pts, idx_kn, scale=2)
    return w, i8, (num * 0.05) / 0.05


img32 = torch.randn(6, 64, 180, 320, device='cuda', generator=g)
wt32 = torch.randn(64, 64, 3, 3, device='cuda', generator=g) / 24
img16, wt16 = img32.bfloat16(), wt32.bfloat16()
a32 = torch.randn(4096, 1024, device='cuda', generator=g)
b32 = torch.randn(1024, 1024, device='cuda', generator=g)
a16, b16 = a32.bfloat16(), b32.bfloat16()
km = F.build_kmap(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
x64 = torch.randn(n, 64, device='cuda', generator=g)
w64 = torch.randn(27, 64, 64, device='cuda', generator=g) / 40
o64 = torch.empty(n, 64, device='cuda')
sch = km.schedule(False)


def conv_bwd(x, w):
    go = torch.ones(x.shape[0], w.shape[0], x.shape[2], x.shape[3], device='cuda', dtype=x.dtype)
    return torch.ops.aten.convolution_backward(go, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, True, False))


AGG = [('nothing', lambda: None),
       ('MIOpen conv2d fp32 forward', lambda: torch.nn.functional.conv2d(img32, wt32, padding=1)),
       ('MIOpen conv2d fp32 backward (data + weights)', lambda: conv_bwd(img32, wt32)),
       ('MIOpen conv2d bf16 forward', lambda: torch.nn.functional.conv2d(img16, wt16, padding=1)),
       ('MIOpen conv2d bf16 backward (data + weights)', lambda: conv_bwd(img16, wt16)),
       ('matmul fp32 (rocBLAS / hipBLASLt)', lambda: a32 @ b32),
       ('matmul bf16 (rocBLAS / hipBLASLt)', lambda: a16 @ b16),
       ('u2mkd conv_tp 64 -> 64 (this library)', lambda: sch.run(x64, w64, True, 64, False, o64))]
torch.cuda.synchronize()
ref = victims()
torch.cuda.synchronize()
side = torch.cuda.Stream()
for name, fn in AGG:
    with torch.cuda.stream(side):
        fn()                      # (solver search / first-call set-up outside the rounds)
    torch.cuda.synchronize()
    bad_rounds, bad_elems = 0, 0
    for r in range(rounds):
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        got = [victims() for _ in range(4)]
        with torch.cuda.stream(side):
            fn()
        torch.cuda.synchronize()
        d = sum(int((a != e).sum()) for res in got for a, e in zip(res, ref))
        bad_rounds += int(d > 0)
        bad_elems += d
    print('%-48s rounds with a changed victim result: %3d of %d (%d elements)' % (name, bad_rounds, rounds, bad_elems), flush=True)
