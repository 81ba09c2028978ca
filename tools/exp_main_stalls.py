"""Where does the MAIN stream of the pipelined KD step wait for another stream, and for how long?  Every Python-level
``wait_event`` / ``wait_stream`` issued ON the main stream is bracketed by two HIP events; the GPU time between them is the
stall (0 when what it waits for was already done).  Aggregated by call site over the observed steps.
  python tools/exp_main_stalls.py [steps=8]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sys.argv = sys.argv[:1]
args = bench.parse()
step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
for _ in range(8):
    step()
main = torch.cuda.current_stream()
log = []
on = [False]
busy = [False]


def site():
    fr = [f for f in traceback.extract_stack(limit=24) if '/u2mkd_amd/' in f.filename]
    return ' < '.join('%s:%d %s' % (f.filename.split('/u2mkd_amd/')[-1], f.lineno, f.name) for f in reversed(fr[-3:]))


def wrap(name):
    real = getattr(torch.cuda.Stream, name)

    def f(self, other):
        if on[0] and not busy[0] and self.cuda_stream == main.cuda_stream:
            busy[0] = True
            a = torch.cuda.Event(enable_timing=True); a.record(self)
            real(self, other)
            b = torch.cuda.Event(enable_timing=True); b.record(self)
            log.append((site(), a, b))
            busy[0] = False
            return
        return real(self, other)
    setattr(torch.cuda.Stream, name, f)


wrap('wait_event')
wrap('wait_stream')
t0 = torch.cuda.Event(enable_timing=True); t0.record()
on[0] = True
for _ in range(STEPS):
    step()
on[0] = False
t1 = torch.cuda.Event(enable_timing=True); t1.record()
torch.cuda.synchronize()
print('%.2f ms per step; %d waits on the main stream per step' % (t0.elapsed_time(t1) / STEPS, len(log) / STEPS))
agg = collections.defaultdict(lambda: [0.0, 0])
for s, a, b in log:
    agg[s][0] += a.elapsed_time(b)
    agg[s][1] += 1
tot = 0.0
for s, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    tot += t
    print('%7.3f ms/step  %5.1f waits/step  %s' % (t / STEPS, c / STEPS, s))
print('total %.2f ms per step of the main stream waiting for other streams' % (tot / STEPS))
