"""Same-box A/B of library builds on the north-star micro-shape (SubMConv3d 64 -> 64, 80k voxels): forward / input
gradient (conv_tp alone, fragments prepared once), weight gradient incl. its slab reduce.
    python tools/ab_variants.py _v0 _v1 ...        (suffixes of u2mkd_amd/lib/libu2mkd_hip<suffix>.so, '' = the product library)
One child process per library; outputs of the first are the bitwise reference of the others."""
import os, subprocess, sys
sys.path.insert(0, '.')


def child(suffix, ref_path, reps):
    import torch
    import u2mkd_amd._lib as L
    L.LIB_PATH = L.LIB_PATH.replace('libu2mkd_hip.so', 'libu2mkd_hip%s.so' % suffix)
    from u2mkd_amd.torchsparse.nn import functional as F
    from u2mkd_amd.synth import synth_batch
    from bench import time_events
    b = synth_batch(80000, 1)
    c = torch.from_numpy(b['coords']).cuda()
    km = F.build_kmap(c, (1,) * 3, (3,) * 3, (1,) * 3)
    n = km.n_out
    sch = km.schedule(False)
    pairs, _, plan = km.pairs_plan()
    lib = L.load(); st = L.stream()
    g = torch.Generator(device='cuda').manual_seed(0)
    cin = cout = 64
    x = torch.randn(n, cin, device='cuda', generator=g); gy = torch.randn(n, cout, device='cuda', generator=g)
    w = torch.randn(27, cin, cout, device='cuda', generator=g) / (27 * cin) ** 0.5
    wf = torch.empty(2, lib.u2mkd_weight_fragments_bytes(27, cin, cout, 0), dtype=torch.uint8, device='cuda')
    L.call('u2mkd_weight_fragments', L.ptr(w), 27, cin, cout, 2, 0, L.ptr(wf), st)
    out = torch.empty(n, cout, device='cuda'); dx = torch.empty(n, cin, device='cuda'); dw = torch.empty_like(w)
    nbytes = lib.u2mkd_conv_wgrad_pairs_workspace_bytes(n, cin, cout, 27)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    xb, gyb = x.bfloat16(), gy.bfloat16()
    wfb = torch.empty(2, lib.u2mkd_weight_fragments_bytes(27, cin, cout, 3), dtype=torch.uint8, device='cuda')
    L.call('u2mkd_weight_fragments', L.ptr(w), 27, cin, cout, 2, 3, L.ptr(wfb), st)
    outb = torch.empty(n, cout, device='cuda', dtype=torch.bfloat16); dwb = torch.empty_like(w)

    def conv(a, frag, flip, o):
        L.call('u2mkd_conv_forward_tiles', L.ptr(a), n, cin, L.ptr(wf[frag]), cout, L.ptr(sch.nbr_s), L.ptr(sch.order),
               L.ptr(sch.items), L.ptr(sch.n_items), n, 27, flip, 0, L.ptr(o), st)
    fns = {
        'fwd': lambda: conv(x, 0, 0, out), 'dgrad': lambda: conv(gy, 1, 1, dx),
        'wgrad': lambda: L.call('u2mkd_conv_wgrad_pairs', L.ptr(x), cin, L.ptr(gy), cout, L.ptr(pairs), L.ptr(plan), n, 27, 0,
                                L.ptr(ws), nbytes, L.ptr(dw), st),
        'fwd_bf16': lambda: L.call('u2mkd_conv_forward_tiles_bf16', L.ptr(xb), n, cin, L.ptr(wfb[0]), cout, L.ptr(sch.nbr_s), L.ptr(sch.order),
                                   L.ptr(sch.items), L.ptr(sch.n_items), n, 27, 0, L.ptr(outb), st),
        'wgrad_bf16': lambda: L.call('u2mkd_conv_wgrad_pairs_bf16', L.ptr(xb), cin, L.ptr(gyb), cout, L.ptr(pairs), L.ptr(plan), n, 27, 0,
                                     L.ptr(ws), nbytes, L.ptr(dwb), st),
    }
    res = {}
    for _ in range(reps):
        for k, f in fns.items():
            res.setdefault(k, []).append(time_events([f], 60) * 1e3)
    outs = {'out': out, 'dx': dx, 'dw': dw, 'outb': outb.float(), 'dwb': dwb}
    cmp_ = ''
    if os.path.exists(ref_path):
        ref = torch.load(ref_path)
        cmp_ = ' '.join('%s:%s' % (k, 'same' if torch.equal(ref[k], v.cpu()) else 'DIFF %.2e' % float((ref[k] - v.cpu()).abs().max())) for k, v in outs.items())
    else:
        torch.save({k: v.cpu() for k, v in outs.items()}, ref_path)
        cmp_ = '(reference)'
    med = lambda v: sorted(v)[len(v) // 2]
    print('VARIANT %-8s ' % (suffix or '(product)') + ' '.join('%s %.1f' % (k, med(v)) for k, v in res.items())
          + ' | group %.1f us | %s' % (med(res['fwd']) + med(res['dgrad']) + med(res['wgrad']), cmp_), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == '--child':
        child(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    else:
        ref = '/tmp/ab_variants_ref.pt'
        if os.path.exists(ref):
            os.remove(ref)
        for suf in sys.argv[1:]:
            subprocess.run([sys.executable, __file__, '--child', suf if suf != "''" else '', ref, os.environ.get('AB_REPS', '3')], check=False)
