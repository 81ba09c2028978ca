"""How far the host runs ahead of the GPU through a pipelined KD step (bench loop): host clock at the phases of the step
and, for each, the GPU's clock when it reaches the same point (an event recorded there); plus every blocking
synchronisation torch reports (set_sync_debug_mode) with its call site.   python tools/host_lead.py"""
import collections, os, sys, time, traceback, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from u2mkd_amd import lidar, train as T, kd as KD
from u2mkd_amd.synth import synth_kd_batch

torch.manual_seed(0)
sp = {k: v for k, v in lidar.spformer_kwargs().items() if k not in ('cr', 'in_channel', 'num_classes')}
model = KD.TSDFull(cr=1.0, cr_t=2.0, in_channel=4, in_channel_t=4, num_classes=17, spformer=sp).cuda()
run = T.KDStep(model, num_epochs=50, batch_size=1)
run.train_mode()
res = [T.kd_batch_to_device(synth_kd_batch(80000, 1, seed=1234 + i, image_hw=(360, 640))) for i in range(4)]
marks = []
on = [False]


def mark(label):
    if on[0]:
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        marks.append((label, time.perf_counter(), e))


amp = run.amp
orig_bs = amp.backward_and_step


def bs(loss, opt):
    mark('backward.begin')
    opt.zero_grad()
    mark('zero_grad.end')
    amp.scaler.scale(loss).backward()
    mark('backward.end')
    amp.scaler.step(opt)
    mark('opt.step.end')
    amp.scaler.update()


amp.backward_and_step = bs
orig_prep = model.prepare


def prep(*a, **k):
    mark('prepare.begin'); o = orig_prep(*a, **k); mark('prepare.end'); return o


model.prepare = prep
orig_fwd = run.net.forward if hasattr(run.net, 'forward') else None
orig_losses = KD.kd_losses if hasattr(KD, 'kd_losses') else None


def loop(steps):
    cur = T.fresh_batch(res[0])
    for i in range(steps):
        mark('step.begin')
        nxt = T.fresh_batch(res[(i + 1) % 4])
        run(cur, prefetch=nxt)
        mark('step.end')
        cur = nxt


loop(6)
torch.cuda.synchronize()
sites = collections.Counter()


def hook(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if '/u2mkd_amd/' in f.filename]
    sites[' <- '.join(f'{f.filename.split("/")[-1]}:{f.lineno}' for f in st[-3:][::-1])] += 1


warnings.showwarning = hook
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
on[0] = True
base = torch.cuda.Event(enable_timing=True); base.record(); t0 = time.perf_counter()
loop(4)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode('default')
print('%-18s %9s %9s %8s' % ('mark', 'host ms', 'gpu ms', 'lead'))
for label, th, e in marks:
    h, g = (th - t0) * 1e3, base.elapsed_time(e)
    print('%-18s %9.2f %9.2f %8.2f' % (label, h, g, g - h))
print('blocking synchronisations in 4 steps:')
for k, v in sites.most_common():
    print(v, k)
