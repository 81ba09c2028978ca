"""How far can the host run AHEAD of the GPU on one stream?  A long spin kernel blocks the stream, then the host queues N tiny
kernels and notes when each launch call returns: the first launch that takes milliseconds instead of microseconds is where the
runtime made the host wait (a full AQL queue, the kernel-argument ring, the signal pool).
  python tools/exp_launch_lead.py [n=12000] [spin_ms=400]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from u2mkd_amd import _lib as L

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
SPIN = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
x = torch.zeros(64, device='cuda')
coords = torch.zeros(64, 4, dtype=torch.float32, device='cuda')
out = torch.zeros(64, 4, dtype=torch.int32, device='cuda')
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
cyc = 20_000_000 / e0.elapsed_time(e1)


def run(label, launch):
    torch.cuda.synchronize()
    torch.cuda._sleep(int(SPIN * cyc))
    t0 = time.perf_counter()
    ts = []
    for i in range(N):
        launch()
        ts.append(time.perf_counter())
    torch.cuda.synchronize()
    d = [ts[0] - t0] + [b - a for a, b in zip(ts, ts[1:])]
    stalls = [(i, v * 1e3) for i, v in enumerate(d) if v > 0.5e-3]
    med = sorted(d)[len(d) // 2] * 1e6
    print('%-34s median launch %.1f us; launches above 0.5 ms: %s' % (label, med, ['#%d %.1f ms' % s for s in stalls[:8]]), flush=True)
    print('   host time to queue all %d: %.1f ms (the spin is %.0f ms)' % (N, (ts[-1] - t0) * 1e3, SPIN))


run('torch add_ (1 small arg block)', lambda: x.add_(1))
st = L.stream()
run('u2mkd_floor_coords (C ABI)', lambda: L.call('u2mkd_floor_coords', L.ptr(coords), 64, 1, L.ptr(out), st))
side = torch.cuda.Stream()
evs = [torch.cuda.Event() for _ in range(64)]


def with_events():
    x.add_(1)
    evs[0].record()
    side.wait_event(evs[0])


run('add_ + event record + stream wait', with_events)
