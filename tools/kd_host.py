"""KD step: host time per step (until the Python call returns, queue still draining) against wall time per step,
the host floor on a tiny scene, the sync points of a step, and a cProfile of the host side."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from u2mkd_amd import lidar, train as T, kd as KD
from u2mkd_amd.synth import synth_kd_batch

def build(n_vox, hw=(360, 640)):
    torch.manual_seed(0)
    sp = {k: v for k, v in lidar.spformer_kwargs().items() if k not in ('cr', 'in_channel', 'num_classes')}
    model = KD.TSDFull(cr=1.0, cr_t=2.0, in_channel=4, in_channel_t=4, num_classes=17, spformer=sp).cuda()
    run = T.KDStep(model, num_epochs=50, batch_size=1)
    run.train_mode()
    d = T.kd_batch_to_device(synth_kd_batch(n_vox, 1, seed=1234, image_hw=hw))
    return run, d

def measure(run, d, steps=10, warm=4):
    for _ in range(warm): run(d)
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        a = time.perf_counter(); run(d); host += time.perf_counter() - a
    torch.cuda.synchronize()
    return host / steps * 1e3, (time.perf_counter() - t0) / steps * 1e3

run, d = build(80000)
h, w = measure(run, d)
print(f'80k scene: host {h:.1f} ms/step, wall {w:.1f} ms/step', flush=True)
if '--profile' in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): run(d)
    torch.cuda.synchronize(); pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(35)
    if '--callers' in sys.argv:
        st.sort_stats('ncalls').print_callers(r'built-in method torch\.(argsort|sort|where|arange|cat|zeros|full|empty_like|zeros_like|ones|stack|cumsum|bincount|unique|searchsorted|nonzero)|method .(int|long|float|to|clone|contiguous|fill_|copy_|zero_|index_select|masked_fill_|nonzero|sum|any|all|max|min|cumsum|repeat_interleave|expand|reshape). of')
if '--syncs' in sys.argv:
    import warnings, collections, traceback
    sites = collections.Counter()
    def hook(message, category, filename, lineno, file=None, line=None):
        st = [f for f in traceback.extract_stack() if 'u2mkd_amd' in f.filename]
        sites[' <- '.join(f'{f.filename.split("/")[-1]}:{f.lineno}' for f in st[-3:][::-1])] += 1
    warnings.showwarning = hook; warnings.simplefilter('always')
    torch.cuda.set_sync_debug_mode('warn'); run(d); torch.cuda.set_sync_debug_mode('default')
    for k, v in sites.most_common(): print(v, k)
    print('total syncs per step:', sum(sites.values()))
del run, d
torch.cuda.empty_cache()
run, d = build(1500, hw=(32, 64))
h, w = measure(run, d)
print(f'tiny scene (host floor): host {h:.1f} ms/step, wall {w:.1f} ms/step', flush=True)
