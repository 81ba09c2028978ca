import sys; sys.path.insert(0, '.')
import torch, torch.nn as nn
torch.manual_seed(0)
def rel(a, b): return float((a.double().cpu()-b.double()).abs().max()/b.double().abs().max())
for C in (48, 96, 128, 256):
    N = 6000
    x = torch.randn(N, C) * 3 + 1; g = torch.randn(N, C) * 1e-3
    # BN
    for dev_name in ('bn',):
        bn = nn.BatchNorm1d(C); bn.weight.data.uniform_(0.8, 1.2); bn.bias.data.normal_(0, 0.05)
        xc = x.clone().double().requires_grad_(True); bnd = nn.BatchNorm1d(C).double(); bnd.load_state_dict(bn.state_dict())
        yc = bnd(xc); yc.backward(g.double())
        xg = x.clone().cuda().requires_grad_(True); bng = nn.BatchNorm1d(C).cuda(); bng.load_state_dict(bn.state_dict())
        yg = bng(xg); yg.backward(g.cuda())
        x32 = x.clone().requires_grad_(True); bn32 = nn.BatchNorm1d(C); bn32.load_state_dict(bn.state_dict()); y32 = bn32(x32); y32.backward(g)
        print(C, 'BN   fwd gpu %.1e cpu32 %.1e | dx gpu %.1e cpu32 %.1e | dw gpu %.1e cpu32 %.1e' % (rel(yg, yc), rel(y32, yc), rel(xg.grad, xc.grad), rel(x32.grad, xc.grad), rel(bng.weight.grad, bnd.weight.grad), rel(bn32.weight.grad, bnd.weight.grad)))
    lin = nn.Linear(C, 17)
    xc = x.clone().double().requires_grad_(True); ld = nn.Linear(C, 17).double(); ld.load_state_dict(lin.state_dict())
    g17 = torch.randn(N, 17)
    yc = ld(xc); yc.backward(g17.double())
    xg = x.clone().cuda().requires_grad_(True); lg = nn.Linear(C, 17).cuda(); lg.load_state_dict(lin.state_dict()); yg = lg(xg); yg.backward(g17.cuda())
    print(C, 'LIN  fwd gpu %.1e | dx %.1e | dw %.1e' % (rel(yg, yc), rel(xg.grad, xc.grad), rel(lg.weight.grad, ld.weight.grad)))
    # matmul (k=1 conv)
    w = torch.randn(C, C) / C ** 0.5
    print(C, 'MM   gpu %.1e cpu32 %.1e' % (rel(x.cuda() @ w.cuda(), x.double() @ w.double()), rel(x @ w, x.double() @ w.double())))
print(torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32, torch.get_float32_matmul_precision())
