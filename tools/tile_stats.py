"""CPU-side statistics of the tile-pair schedule on the bench scene (no GPU): blocks, padding, how often
consecutive blocks share their offset, for tile heights 64 / 128 and block sizes 16 / 32."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ts_ref as R
from u2mkd_amd.synth import synth_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
b = synth_batch(n, 1, seed=1234)
nbmaps, nbsizes, oc, _ = R.build_kmap(b['coords'], 1, 3, 1)
nbmaps = np.asarray(nbmaps); nbsizes = np.asarray(nbsizes)
N = b['coords'].shape[0]
K = len(nbsizes)
P = int(nbsizes.sum())
mask = np.zeros(N, dtype=np.int64)
o = 0
for k in range(K):
    seg = nbmaps[o:o + nbsizes[k]]
    mask[seg[:, 1]] |= (1 << k)
    o += nbsizes[k]
order = np.argsort(mask.astype(np.int32), kind='stable')
ms = mask[order]
print('N', N, 'P', P, 'kbar', P / N, 'dense blocks', P / 16)
for T in (64, 128, 256):
    t = (N + T - 1) // T
    pad = np.zeros(t * T, dtype=np.int64); pad[:N] = ms
    bits = (pad.reshape(t, T, 1) >> np.arange(K)) & 1
    cnt = bits.sum(1)                        # [t, K]
    for B in (16, 32):
        blocks = (cnt + B - 1) // B
        nb = blocks.sum()
        groups = (cnt > 0).sum()
        print(f'T={T} B={B}: block-steps {nb} ({nb * B / P:.2f}x dense), offset groups {groups} '
              f'(fragment loads if reused: {groups / nb:.2f} of steps), blocks/tile mean {blocks.sum(1).mean():.1f} max {blocks.sum(1).max()}')
