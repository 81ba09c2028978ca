import sys; sys.path.insert(0, '.')
import numpy as np, torch
from oracle import spformer_ref as R, spvcnn_ref as O
from oracle import torchsparse_cpu as ots
from u2mkd_amd import lidar, torchsparse as ts
from u2mkd_amd.synth import synth_batch
b = synth_batch(2000, 2, seed=33)
feats, coords = torch.from_numpy(b['feats']), torch.from_numpy(b['coords'])
ref = O.fill_state_by_name(R.SPVCNN_SPFORMER(**R.default_spformer_kwargs(cr=1.0, drop_path_rate=0.0))).train(); ref.dropout.p = 0
model = lidar.SPVCNN_SPFORMER(**lidar.spformer_kwargs(cr=1.0, drop_path_rate=0.0)); model.load_state_dict(ref.state_dict()); model.cuda().train(); model.dropout.p = 0
cap_r, cap_g = {}, {}
def hook(store, name):
    def f(mod, inp, out):
        store[name] = (tuple(i.detach().cpu() if torch.is_tensor(i) else None for i in inp), out.detach().cpu() if torch.is_tensor(out) else None)
    return f
for i in range(4):
    ref.transformer_blocks[i].register_forward_hook(hook(cap_r, f'blk{i}'))
    model.transformer_blocks[i].register_forward_hook(hook(cap_g, f'blk{i}'))
    ref.transformer_blocks[i].attn.register_forward_hook(hook(cap_r, f'attn{i}'))
    model.transformer_blocks[i].attn.register_forward_hook(hook(cap_g, f'attn{i}'))
o_r = ref({'lidar': ots.SparseTensor(feats, coords)})['x_vox']
o_g = model({'lidar': ts.SparseTensor(feats.cuda(), coords.cuda())})['x_vox']
for k in sorted(cap_r):
    (ir, outr), (ig, outg) = cap_r[k], cap_g[k]
    errs = [float((a - b).abs().max()) if a is not None and a.dtype.is_floating_point else (int((a != b).sum()) if a is not None else None) for a, b in zip(ir, ig)]
    print(k, 'input errs', errs, 'out err', float((outr - outg).abs().max()), 'out max', float(outr.abs().max()))
print('logits', float((o_r.detach() - o_g.detach().cpu()).abs().max()))
for k in ('attn0', 'attn1'):
    e = (cap_r[k][1] - cap_g[k][1]).abs().max(1)[0]
    print(k, 'tokens', len(e), 'err>1e-3:', int((e > 1e-3).sum()), 'err>1e-1:', int((e > 1e-1).sum()), 'median', float(e.median()))
# same attention module, IDENTICAL inputs on both sides (oracle inputs fed to the GPU module)
(ir, outr) = cap_r['attn0']
with torch.no_grad():
    og = model.transformer_blocks[0].attn(ir[0].cuda(), ir[1].cuda(), ir[2].cuda()).cpu()
e = (outr - og).abs().max(1)[0]
print('attn0 identical inputs: err>1e-3:', int((e > 1e-3).sum()), 'max', float(e.max()), 'median', float(e.median()))
from oracle import sptr_ref as S
from u2mkd_amd import sptr
feats_in, xyz, batch = ir
att_r, att_g = ref.transformer_blocks[0].attn, model.transformer_blocks[0].attn
for name, coords, window, quant, a in (('cubic', xyz.float(), att_r.window_size, att_r.quant_size, None),
                                       ('sphere', S.cart2sphere(xyz.float()), att_r.window_size_sphere, att_r.quant_size_sphere, 0.0125)):
    c_ref = S.grid_cluster(coords, batch, np.asarray(window))
    plan = sptr.WindowPlan(coords.cuda(), batch.cuda(), window)
    order = plan.sort_idx.cpu().long(); ws = plan.wstart.cpu().long(); wl = plan.wlen.cpu().long()
    same = torch.equal(c_ref[order], c_ref[order][ws])
    cnt_ref = torch.unique(c_ref, return_counts=True)[1]
    print(name, 'partition consistent', same, 'n windows', len(cnt_ref), int((ws == torch.arange(len(ws))).sum()), 'n_max', int(cnt_ref.max()), int(wl.max()))
    qc, radial, _ = plan.quant_coords(coords.cuda(), quant, a is not None)
    wsz = torch.as_tensor(np.asarray(window)).float(); q = torch.as_tensor(np.asarray(quant)).float()
    qc_ref = torch.div((coords - coords.min(0)[0]) % wsz, q, rounding_mode='floor')[order]
    print(name, 'qc mismatches', int((qc.cpu().float() != qc_ref).any(1).sum()), 'qc max', qc.max(0)[0].tolist(), qc_ref.max(0)[0].tolist())
# per-branch outputs with identical inputs
with torch.no_grad():
    N, C = feats_in.shape
    H = att_r.num_heads
    qkv = att_r.qkv(feats_in).reshape(N, 3, H, C // H)
    q, k, v = qkv[:, 0] * att_r.scale, qkv[:, 1], qkv[:, 2]
    h1 = att_r.num_heads_brc1
    for name, sl, coords, window, quant, qgl, tabs, a in (
            ('cubic', slice(0, h1), xyz.float(), att_r.window_size, att_r.quant_size, 24,
             (att_r.relative_pos_query_table, att_r.relative_pos_key_table, att_r.relative_pos_value_table), None),
            ('sphere', slice(h1, None), S.cart2sphere(xyz.float()), att_r.window_size_sphere, att_r.quant_size_sphere, 24,
             (att_r.relative_pos_query_table_sphere, att_r.relative_pos_key_table_sphere, att_r.relative_pos_value_table_sphere), 0.0125)):
        p = S.get_indices_params(coords, batch, np.asarray(window))
        o_ref = S.sparse_self_attention(q[:, sl].contiguous(), k[:, sl].contiguous(), v[:, sl].contiguous(), coords, p[0], p[1], p[2], p[3], p[4], p[5],
                                        np.asarray(window), np.asarray(quant), qgl, tabs[0].detach(), tabs[1].detach(), tabs[2].detach(), a)
        plan = sptr.WindowPlan(coords.cuda(), batch.cuda(), window)
        o_g = sptr.window_attention(q[:, sl].cuda(), k[:, sl].cuda(), v[:, sl].cuda(), coords.cuda(), plan, quant, qgl,
                                    tabs[0].detach().cuda(), tabs[1].detach().cuda(), tabs[2].detach().cuda(), a).cpu()
        e = (o_ref - o_g).abs().flatten(1).max(1)[0]
        bad = (e > 1e-3).nonzero().squeeze(1)
        print(name, 'bad tokens', len(bad), 'max', float(e.max()))
        if len(bad):
            t = int(bad[0]); cl = S.grid_cluster(coords, batch, np.asarray(window)); mem = (cl == cl[t]).nonzero().squeeze(1)
            print('  token', t, 'window members', mem.tolist(), 'coords', coords[mem].tolist())
            rpi = S.relative_position_index(coords[p[5]], p[0], p[1], np.asarray(window), np.asarray(quant), qgl, a)
            pos = (p[5] == t).nonzero().item()
            sel = (p[0] == pos).nonzero().squeeze(1)
            print('  oracle rel rows for query', rpi[sel].tolist())
from u2mkd_amd.lidar.sphereformer import cart2sphere as c2s
sc = S.cart2sphere(xyz.float()); sg = c2s(xyz.float().cuda()).cpu()
d = (sc - sg).abs()
print('sphere coord diff max per axis', d.max(0)[0].tolist(), 'count>1e-4', (d > 1e-4).sum(0).tolist())
big = (d > 1e-3).any(1).nonzero().squeeze(1)[:5]
for t in big.tolist():
    print('  xyz', xyz[t].tolist(), 'cpu', sc[t].tolist(), 'gpu', sg[t].tolist())
w = np.asarray(att_r.window_size_sphere); qz = np.asarray(att_r.quant_size_sphere)
c1 = S.grid_cluster(sc, batch, w); c2 = S.grid_cluster(sg, batch, w)
same1 = (c1[:, None] == c1[None, :]); same2 = (c2[:, None] == c2[None, :])
print('tokens whose window membership differs:', int((same1 != same2).any(1).sum()))
wsz = torch.as_tensor(w).float(); qq = torch.as_tensor(qz).float()
q1 = torch.div((sc - sc.min(0)[0]) % wsz, qq, rounding_mode='floor'); q2 = torch.div((sg - sg.min(0)[0]) % wsz, qq, rounding_mode='floor')
print('qc mismatches per axis', (q1 != q2).sum(0).tolist(), 'mins', sc.min(0)[0].tolist(), sg.min(0)[0].tolist())
m1 = (sc - sc.min(0)[0]) % wsz; m2 = (sg - sg.min(0)[0]) % wsz
bad = (q1 != q2).any(1).nonzero().squeeze(1)[:4]
for t in bad.tolist(): print('   ', sc[t].tolist(), m1[t].tolist(), m2[t].tolist(), q1[t].tolist(), q2[t].tolist())
