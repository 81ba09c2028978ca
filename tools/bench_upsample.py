"""Per-call times of the decoder's bilinear up-samplings (+ skip add): torch's kernels against csrc/pixhead.hip."""
import sys; sys.path.insert(0,'.')
import torch, torch.nn.functional as F
from u2mkd_amd import camera
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(True); b=torch.cuda.Event(True); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n*1e3
for shape,size in [((6,128,12,20),(23,40)),((6,128,23,40),(45,80)),((6,128,45,80),(90,160)),((6,128,90,160),(360,640)),((6,128,90,160),(180,320))]:
    x=torch.randn(shape,device='cuda',requires_grad=True); sk=torch.randn(*shape[:2],*size,device='cuda',requires_grad=True); g=torch.randn_like(sk)
    for hip in (False,True):
        camera._UP_HIP=hip
        try:
            y=camera._up(x,size,sk)
            f=t(lambda: camera._up(x,size,sk)); b=t(lambda: torch.autograd.grad(y,[x,sk],g,retain_graph=True))
            print(shape,size,'hip' if hip else 'torch','fwd %.0f us bwd %.0f us'%(f,b), type(y.grad_fn).__name__)
        except Exception as e: print('ERR',e)
