"""List the host<->device synchronisation points of one SPVCNN training step
(torch.cuda.set_sync_debug_mode('warn'))."""
import sys, warnings, collections, traceback; sys.path.insert(0, '.')
import torch
from u2mkd_amd import lidar, train as T
from u2mkd_amd.synth import synth_batch
b = synth_batch(80000, 1)
feats, coords, labels = (torch.from_numpy(b[k]).cuda() for k in ('feats', 'coords', 'labels'))
model = lidar.SPVCNN(cr=1.0, in_channel=4, num_classes=17, pres=0.05, vres=0.05).cuda().train()
run = T.LidarStep(model)
for _ in range(3): run(feats, coords, labels)
torch.cuda.synchronize()
sites = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if '/root/repo' in f.filename or 'u2mkd_amd' in f.filename]
    st = [f for f in st if 'sync_audit' not in f.filename]
    sites[' <- '.join(f'{f.filename.split("/")[-1]}:{f.lineno}' for f in st[-3:][::-1])] += 1
warnings.showwarning = hook
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
run(feats, coords, labels)
torch.cuda.set_sync_debug_mode('default')
for k, v in sites.most_common(): print(v, k)
print('total syncs per step:', sum(sites.values()))
