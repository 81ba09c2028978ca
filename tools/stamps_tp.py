"""Per-workgroup timeline of the 64->64 tile-pair kernel (in-kernel s_memtime / s_memrealtime stamps)."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from u2mkd_amd import _lib as L
import os
L.LIB_PATH = L.LIB_PATH.replace('libu2mkd_hip.so', 'libu2mkd_hip%s.so' % os.environ.get('AB_SUFFIX', ''))
from u2mkd_amd.torchsparse.nn import functional as F
from u2mkd_amd.synth import synth_batch

b = synth_batch(80000, 1)
c = torch.from_numpy(b['coords']).cuda()
km = F.build_kmap(c, (1,) * 3, (3,) * 3, (1,) * 3)
n = km.n_out
sch = km.schedule(False)
x = torch.randn(n, 64, device='cuda'); w = torch.randn(27, 64, 64, device='cuda') / 40
ARITH = int(sys.argv[1]) if len(sys.argv) > 1 else 0
wt = torch.empty(L.load().u2mkd_weight_fragments_bytes(27, 64, 64, ARITH), dtype=torch.uint8, device='cuda'); L.call('u2mkd_weight_fragments', L.ptr(w), 27, 64, 64, 1, ARITH, L.ptr(wt), L.stream()); o = torch.empty(n, 64, device='cuda')
tiles = (n + 63) // 64
st = torch.zeros(4 * tiles + 64, 8, dtype=torch.int64, device='cuda')
n_items = int(sch.n_items.item()); assert n_items <= 4 * tiles
for _ in range(3):
    L.call('u2mkd_debug_conv_tile_pairs_stamps', L.ptr(x), n, L.ptr(wt), L.ptr(sch.nbr_s), L.ptr(sch.order), L.ptr(sch.items), L.ptr(sch.n_items),
           n, 27, ARITH, L.ptr(o), L.ptr(st), L.stream())
torch.cuda.synchronize()
sall = st.cpu().numpy(); s = sall[:n_items]; ph = sall[4 * tiles:]
rt0 = s[:, 0].min()
start = (s[:, 0] - rt0) / 100.0          # us (100 MHz)
end = (s[:, 5] - rt0) / 100.0
setup = s[:, 2] - s[:, 1]; walk = s[:, 3] - s[:, 2]; epi = s[:, 4] - s[:, 3]; nb = s[:, 6]
print('kernel span us', end.max(), 'tiles', tiles)
print('start us percentiles', np.percentile(start, [0, 50, 80, 90, 99, 100]).round(1))
print('end   us percentiles', np.percentile(end, [0, 50, 80, 90, 99, 100]).round(1))
print('setup cycles pct', np.percentile(setup, [0, 50, 90, 100]).round(0), 'epi', np.percentile(epi, [0, 50, 90, 100]).round(0))
print('clock MHz est', ((s[:, 4] - s[:, 1]) / np.maximum(end - start, 1e-3)).mean().round(0))
for lo, hi in ((1, 4), (5, 8), (9, 16), (17, 32), (33, 48), (49, 70)):
    m = (nb >= lo) & (nb <= hi)
    if m.any():
        print(f'blocks {lo}-{hi}: tiles {m.sum()} walk cycles/block {np.median(walk[m] / nb[m]):.0f} (p90 {np.percentile(walk[m] / nb[m], 90):.0f}) '
              f'dur us {np.median(end[m] - start[m]):.1f} max {np.max(end[m] - start[m]):.1f} start med {np.median(start[m]):.1f}')
i = np.argsort(end)[-5:]
print('last finishers: blocks', nb[i], 'start', start[i].round(1), 'end', end[i].round(1), 'walk cyc/blk', (walk[i] / nb[i]).round(0))

# per-phase cycles of the heaviest tile's wave 0: issue_G | issue_B+store_G | frag reads | MFMA+RMW | barrier
nbk = int(nb[0])
d = np.diff(ph[:min(nbk, 64), :6].astype(np.int64), axis=1)
nxt = ph[1:min(nbk, 64), 0] - ph[:min(nbk, 64) - 1, 5]
print('heaviest tile blocks', nbk)
print('phase medians [issue_G, issue_B+store_G, frag-read, mfma+rmw, barrier]:', np.median(d[4:], axis=0).round(0), 'loop overhead', np.median(nxt[4:]).round(0))
print('phase p90   :', np.percentile(d[4:], 90, axis=0).round(0))
print('step total median', np.median(ph[5:min(nbk,64), 0] - ph[4:min(nbk,64)-1, 0]).round(0))
