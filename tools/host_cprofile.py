"""Where the HOST spends a pipelined KD step (cProfile over a few steps, by own time).   python tools/host_cprofile.py"""
import cProfile, os, pstats, sys, io
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


def loop(steps):
    cur = T.fresh_batch(res[0])
    for i in range(steps):
        nxt = T.fresh_batch(res[(i + 1) % 4])
        run(cur, prefetch=nxt)
        cur = nxt


loop(6)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
loop(8)
pr.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumulative'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
