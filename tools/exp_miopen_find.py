"""Does MIOpen's search (torch.backends.cudnn.benchmark = True) find faster convolution kernels for the camera branch than the
immediate-mode pick?  Runs the KD step's camera convolutions (SwiftNet-18, 6 x 360 x 640, forward + backward) under both
settings with a user find-db under gpurun_out/, prints progress per layer (the search compiles kernels: minutes).
  MIOPEN_USER_DB_PATH=gpurun_out/miopen_db python tools/exp_miopen_find.py [find_mode]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(os.getcwd(), 'gpurun_out', 'miopen_db'))
os.makedirs(os.environ['MIOPEN_USER_DB_PATH'], exist_ok=True)
if len(sys.argv) > 1:
    os.environ['MIOPEN_FIND_MODE'] = sys.argv[1]
import torch
import torch.nn.functional as F

# the distinct 2-D convolutions of SwiftNet-18 on 6 x 3 x 360 x 640 (core/models/image_branch/swiftnet.py): (cin, cout, k, stride, H, W)
shapes = [(3, 64, 7, 2, 360, 640)]
hw = (90, 160)
for cin, cout, stride in ((64, 64, 1), (64, 128, 2), (128, 256, 2), (256, 512, 2)):
    h, w = hw
    shapes.append((cin, cout, 3, stride, h, w))
    ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
    shapes.append((cout, cout, 3, 1, ho, wo))
    if stride != 1:
        shapes.append((cin, cout, 1, stride, h, w))
    hw = (ho, wo)
shapes += [(128, 128, 3, 1, 45, 80), (128, 128, 3, 1, 90, 160), (128, 128, 3, 1, 23, 40), (512, 128, 1, 1, 12, 20)]


def bench(benchmark):
    torch.backends.cudnn.benchmark = benchmark
    tot_f = tot_b = 0.0
    for cin, cout, k, s, h, w in shapes:
        x = torch.randn(6, cin, h, w, device='cuda', requires_grad=True)
        wt = torch.randn(cout, cin, k, k, device='cuda', requires_grad=True)
        t0 = time.perf_counter()
        y = F.conv2d(x, wt, None, s, k // 2)
        g = torch.randn_like(y)
        y.backward(g)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        for _ in range(3):
            y = F.conv2d(x, wt, None, s, k // 2); y.backward(g)
        torch.cuda.synchronize()
        e[0].record()
        for _ in range(10):
            y = F.conv2d(x, wt, None, s, k // 2)
        e[1].record()
        for _ in range(10):
            y = F.conv2d(x, wt, None, s, k // 2); y.backward(g)
        e[2].record(); torch.cuda.synchronize()
        tf = e[0].elapsed_time(e[1]) / 10
        tb = e[1].elapsed_time(e[2]) / 10 - tf
        tot_f += tf; tot_b += tb
        print('  %s conv %3d->%3d k%d s%d %3dx%3d: first call %.1f s, fwd %.3f ms, bwd %.3f ms' % ('FIND' if benchmark else 'imm ', cin, cout, k, s, h, w, first, tf, tb), flush=True)
    print('%s: forward %.2f ms, backward %.2f ms over the %d distinct convolutions' % ('benchmark=True' if benchmark else 'immediate mode', tot_f, tot_b, len(shapes)), flush=True)


bench(False)
bench(True)
