"""Every C-level call the main thread makes during the pipelined KD step (torch operators, tensor methods, ctypes), timed with
sys.setprofile: the calls that take more than `thr` microseconds, by name -- count, total, and where in the step they sit.  The
profile hook slows the host ~2x; a call that WAITS for the GPU still stands out by its length.
  python tools/host_slow_calls.py [steps=10] [thr_us=150]"""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
THR = float(sys.argv[2]) * 1e-6 if len(sys.argv) > 2 else 150e-6
sys.argv = sys.argv[:1]
args = bench.parse()
step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
for _ in range(6):
    step()
torch.cuda.synchronize()
pc = time.perf_counter
stack = []
slow = collections.defaultdict(lambda: [0, 0.0, []])
allc = collections.defaultdict(lambda: [0, 0.0])
T0 = [0.0]


def prof(frame, event, arg):
    if event == 'c_call':
        stack.append(pc())
    elif event in ('c_return', 'c_exception'):
        if stack:
            t = stack.pop()
            d = pc() - t
            name = getattr(arg, '__qualname__', None) or getattr(arg, '__name__', repr(arg))
            a = allc[name]
            a[0] += 1; a[1] += d
            if d > THR:
                caller = '%s:%d' % (os.path.basename(frame.f_code.co_filename), frame.f_lineno)
                s = slow[(name, caller)]
                s[0] += 1; s[1] += d
                if len(s[2]) < 6:
                    s[2].append((t - T0[0]) * 1e3)


marks = []
t_begin = pc()
sys.setprofile(prof)
for _ in range(STEPS):
    T0[0] = pc()
    step()
sys.setprofile(None)
host = pc() - t_begin
torch.cuda.synchronize()
print('per step under the hook: host %.1f ms' % (host / STEPS * 1e3))
print('\ncalls above %.0f us, by total time per step:' % (THR * 1e6))
for (name, caller), (n, tot, at) in sorted(slow.items(), key=lambda kv: -kv[1][1])[:40]:
    print('  %7.2f ms/step  %5.1f/step  mean %7.0f us  %-40s %-28s at ms %s' % (tot / STEPS * 1e3, n / STEPS, tot / n * 1e6, name[:40], caller, ['%.0f' % a for a in at]))
print('\nall C calls by total time per step:')
for name, (n, tot) in sorted(allc.items(), key=lambda kv: -kv[1][1])[:30]:
    print('  %7.2f ms/step  %7.1f/step  mean %6.1f us  %s' % (tot / STEPS * 1e3, n / STEPS, tot / n * 1e6, name[:60]))
