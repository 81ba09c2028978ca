import sys; sys.path.insert(0, '.')
import torch
from tests.test_gpu_spvcnn import _run_pair
n, b, cr = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
ref, out_ref, loss_ref, model, out, loss = _run_pair(n, b, cr)
print('logit err', float((out.detach().cpu() - out_ref.detach()).abs().max()), float(loss), float(loss_ref))
rg = dict(ref.named_parameters())
for name, p in model.named_parameters():
    g, gr = p.grad.detach().cpu().double(), rg[name].grad.double()
    rel = float((g - gr).abs().max() / (gr.abs().max() + 1e-12))
    if name.endswith('kernel'):
        print(f'{name:45s} {rel:.2e} {float(gr.abs().max()):.2e} {tuple(p.shape)}')
