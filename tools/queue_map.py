"""Which of the step's HIP streams share a hardware queue?  For every ordered pair (A, B) of {main, teacher, camera,
sparse_wgrad, geo} (created in the order a KD step creates them): a ~20 ms spin kernel goes to A, a tiny kernel + event to B;
if B's event completes while A is still spinning the two run on different hardware queues, otherwise B sits behind A in one
queue.  Run under different GPU_MAX_HW_QUEUES to see the mapping the runtime chooses.
    GPU_MAX_HW_QUEUES=4 python tools/queue_map.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import u2mkd_amd          # noqa: F401  (sets GPU_MAX_HW_QUEUES unless exported)
from u2mkd_amd import deferred

print('GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES'))
torch.cuda.init()
main = torch.cuda.current_stream()
roles = sys.argv[1:] or ['geo', 'camera', 'teacher', 'sparse_wgrad']      # (creation order = the order a KD step first uses them)
streams = {'main': main}
for r in roles:
    streams[r] = deferred.stream(0, r)
names = list(streams)
x = torch.zeros(1, device='cuda')
torch.cuda.synchronize()
SPIN = int(2.0e7)      # cycles of torch.cuda._sleep (~10-20 ms)
print('%-14s' % 'A \\ B' + ''.join('%-14s' % n for n in names))
for a in names:
    row = []
    for b in names:
        if a == b:
            row.append('-')
            continue
        torch.cuda.synchronize()
        with torch.cuda.stream(streams[a]):
            torch.cuda._sleep(SPIN)
            ea = torch.cuda.Event()
            ea.record()
        with torch.cuda.stream(streams[b]):
            x.add_(1)
            eb = torch.cuda.Event()
            eb.record()
        t0 = time.time()
        while not eb.query() and not ea.query() and time.time() - t0 < 2.0:
            pass
        overlapped = eb.query() and not ea.query()
        row.append('concurrent' if overlapped else 'SAME QUEUE')
        torch.cuda.synchronize()
    print('%-14s' % a + ''.join('%-14s' % v for v in row))
