"""Host time per MIOpen convolution call (forward, input gradient, weight gradient) for the camera branch's typical layers, with
torch.backends.cudnn.benchmark off / on.  MIOPEN_FIND_MODE=FAST keeps the first-call search short.
python tools/miopen_host_cost.py [benchmark=0|1]"""
import os, sys, time
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')
import torch
torch.backends.cudnn.benchmark = len(sys.argv) > 1 and sys.argv[1] == '1'
shapes = [((6, 64, 180, 320), (64, 64, 3, 3), 1, 1), ((6, 128, 90, 160), (128, 128, 3, 3), 1, 1), ((6, 256, 45, 80), (256, 256, 3, 3), 1, 1),
          ((6, 128, 90, 160), (128, 128, 1, 1), 1, 0)]
for xs, ws, stride, pad in shapes:
    x = torch.randn(*xs, device='cuda')
    w = torch.randn(*ws, device='cuda')
    y = torch.nn.functional.conv2d(x, w, None, stride, pad)
    g = torch.randn_like(y)
    ops = {'forward': lambda: torch.nn.functional.conv2d(x, w, None, stride, pad),
           'input gradient': lambda: torch.ops.aten.convolution_backward(g, x, w, None, (stride, stride), (pad, pad), (1, 1), False, (0, 0), 1, (True, False, False)),
           'weight gradient': lambda: torch.ops.aten.convolution_backward(g, x, w, None, (stride, stride), (pad, pad), (1, 1), False, (0, 0), 1, (False, True, False))}
    out = []
    for name, fn in ops.items():
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        host = (time.perf_counter() - t0) / 50
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 50
        out.append('%s host %.0f us (gpu-bound wall %.0f us)' % (name, host * 1e6, wall * 1e6))
    print('benchmark=%s x%s w%s: %s' % (torch.backends.cudnn.benchmark, xs, ws, '; '.join(out)), flush=True)
