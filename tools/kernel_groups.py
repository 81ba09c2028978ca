"""Group a rocprofv3 kernel_stats.csv of the KD bench by kernel family: ms and launches per step.
usage: python tools/kernel_groups.py <kernel_stats.csv> <steps incl. warm-up> [top]"""
import collections
import csv
import sys

GROUPS = (('miopen', 'miopen'), ('igemm', 'miopen'), ('winograd', 'miopen'), ('sp3asm', 'miopen'), ('gfx9', 'miopen'), ('naive_conv', 'miopen'),
          ('batched_transpose', 'miopen-transpose'), ('sptr', 'sptr'), ('conv_px3', 'conv_px3'), ('conv_tp', 'conv_tp'), ('wgrad', 'wgrad'),
          ('conv_os', 'conv_os'), ('conv_pairs', 'conv_pairs(f32)'), ('bn2d', 'bn2d'), ('bn_', 'bn'), ('pairs_gather', 'gather_sum'), ('segment_sum', 'pt-vox'),
          ('devoxelize', 'pt-vox'), ('voxelize', 'pt-vox'), ('cijk', 'rocblas'), ('rocprim', 'rocprim'), ('radix', 'rocprim'),
          ('upsample', 'upsample'), ('weight_fragments', 'fragments'), ('kmap', 'kmap/hash'), ('table_', 'kmap/hash'), ('hash', 'kmap/hash'),
          ('floor_coords', 'kmap/hash'), ('downsample_keys', 'kmap/hash'), ('unpack_keys', 'kmap/hash'), ('count_kernel', 'kmap/hash'), ('ti_weights', 'kmap/hash'),
          ('schedule', 'schedule'), ('pairs_build', 'schedule'), ('tile_', 'schedule'), ('layer_norm', 'layernorm'), ('multi_tensor', 'optimizer'),
          ('softmax', 'loss'), ('elementwise', 'torch-elementwise'), ('vectorized', 'torch-elementwise'), ('reduce_kernel', 'torch-reduce'),
          ('index', 'torch-index'), ('gather', 'torch-index'), ('scatter', 'torch-index'), ('cat', 'torch-cat'), ('fill', 'torch-fill'), ('copy', 'torch-copy'))


def grp(n):
    n = n.lower()
    for key, g in GROUPS:
        if key in n:
            return g
    return 'other'


if __name__ == '__main__':
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = int(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    calls = sum(int(r['Calls']) for r in rows)
    print('kernel time %.1f ms/step, %d launches/step (%d steps)' % (tot / 1e6 / steps, calls / steps, steps))
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        g = grp(r['Name'])
        acc[g][0] += float(r['TotalDurationNs'])
        acc[g][1] += int(r['Calls'])
    for g, (t, c) in sorted(acc.items(), key=lambda x: -x[1][0]):
        print('%-20s %7.2f ms/step %6d launches/step' % (g, t / 1e6 / steps, c / steps))
    print()
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:top]:
        print('%8.2f ms/step %6.1f  %s' % (float(r['TotalDurationNs']) / 1e6 / steps, int(r['Calls']) / steps, r['Name'][:120]))
