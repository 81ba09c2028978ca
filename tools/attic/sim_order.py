"""CPU model of the conv forward's tile schedule for different row orders (no GPU).
tile = 64 sorted rows; stages = popcount(OR of the rows' 27-bit masks); list-schedule
the tiles in launch order on SLOTS workgroup slots."""
import sys; sys.path.insert(0, '.')
import heapq
import numpy as np
from oracle import ts_ref as R
from u2mkd_amd.synth import synth_batch

def masks_of(c4):
    c = c4.astype(np.int64)
    key = (c[:, 3] << 54) | ((c[:, 0] + 1) << 36) | ((c[:, 1] + 1) << 18) | (c[:, 2] + 1)
    s = np.sort(key); m = np.zeros(len(c), np.int64); k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                q = key + (dx << 36) + (dy << 18) + dz
                pos = np.minimum(np.searchsorted(s, q), len(s) - 1)
                m |= (s[pos] == q).astype(np.int64) << k; k += 1
    return m

def popc(x):
    x = x.astype(np.uint64); c = np.zeros(x.shape, np.int64)
    for i in range(27): c += ((x >> np.uint64(i)) & np.uint64(1)).astype(np.int64)
    return c

def tiles(m, order, T=64):
    ms = m[order]; n = len(ms); pad = (-n) % T
    ms = np.concatenate([ms, np.zeros(pad, np.int64)]).reshape(-1, T)
    return popc(np.bitwise_or.reduce(ms, axis=1))

def makespan(st, slots=768, setup=3.0, per=3.7):
    h = [0.0] * slots; heapq.heapify(h)
    for s in st:
        t = heapq.heappop(h); heapq.heappush(h, t + setup + per * s)
    return max(h)

def main():
    b = synth_batch(80000, 1); coords = b['coords']; ts = 1
    for lv in range(5):
        m = masks_of(np.concatenate([coords[:, :3] // ts, coords[:, 3:]], 1))
        u, inv, cnt = np.unique(m, return_inverse=True, return_counts=True)
        orders = {
            'mask': np.argsort(m, kind='stable'),
            'rare-first': np.argsort((cnt[inv].astype(np.int64) << 32) | m, kind='stable'),
            'popc-desc,mask': np.argsort(((27 - popc(m)) << 32) | m, kind='stable'),
        }
        out = [f'ts={ts} N={len(m)} masks={len(u)}']
        for nm, o in orders.items():
            st = tiles(m, o)
            out.append(f'{nm}: sum {st.sum()} max {st.max()} span {makespan(st):.0f} lpt {makespan(np.sort(st)[::-1]):.0f}')
        print(' | '.join(out), flush=True)
        coords = R.spdownsample(coords, 2, 2, ts); ts *= 2

if len(sys.argv) == 1: main()

def detail():
    b = synth_batch(80000, 1); coords = b['coords']
    m = masks_of(coords)
    u, inv, cnt = np.unique(m, return_inverse=True, return_counts=True)
    o = np.argsort((cnt[inv].astype(np.int64) << 32) | m, kind='stable')
    for T in (16, 32, 64):
        st = tiles(m, o, T)
        print(f'T={T}: tiles {len(st)} sum {st.sum()} max {st.max()} hist>{[int((st > x).sum()) for x in (4, 8, 12, 16, 20, 24)]}')
    # rare rows: sort by popcount then mask / cluster test
    rare = cnt[inv] < 16
    print('rare rows', int(rare.sum()), 'of', len(m))
    mr = m[rare]
    for nm, key in (('mask', mr), ('popc,mask', (popc(mr) << 32) | mr)):
        oo = np.argsort(key, kind='stable')
        for T in (16, 64):
            st = tiles(mr, oo, T)
            print(f'  rare {nm} T={T}: tiles {len(st)} sum {st.sum()} max {st.max()} mean {st.mean():.1f}')

if len(sys.argv) > 1: detail()

def hybrid():
    b = synth_batch(80000, 1); coords = b['coords']; ts = 1
    for lv in range(5):
        m = masks_of(np.concatenate([coords[:, :3] // ts, coords[:, 3:]], 1))
        u, inv, cnt = np.unique(m, return_inverse=True, return_counts=True)
        c = cnt[inv]
        P = int(popc(m).sum())
        for thr in (8, 16, 32, 64, 128):
            rare = c < thr
            mm = m[~rare]
            o = np.argsort((c[~rare].astype(np.int64) << 32) | mm, kind='stable')
            st = tiles(mm, o) if len(mm) else np.zeros(1, np.int64)
            pr = int(popc(m[rare]).sum())
            print(f'ts={ts} N={len(m)} thr={thr}: rare rows {int(rare.sum())} pairs {pr} ({100*pr/P:.0f}%) | main tiles {len(st)} sum {st.sum()} max {st.max()} >8:{int((st>8).sum())} >12:{int((st>12).sum())}')
        coords = R.spdownsample(coords, 2, 2, ts); ts *= 2

if len(sys.argv) > 1 and sys.argv[1] == 'h': hybrid()
