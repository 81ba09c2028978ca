"""Kernels of one HIP stream must not change the results of another stream's kernels (NOTES N9).

gfx950's v_mfma_f32_16x16x32_bf16 does exactly that to resident waves of other kernels (tools/repro_concurrent_kernels.hip, no
torch: u2mkd_ti_weights returns wrong weights in ~50 % of its launches next to a loop of that instruction), which is why the
library multiplies with the gfx942 form (csrc/conv_internal.h: mfma_bf16_k32).  This test runs the LIBRARY'S OWN matrix kernels
-- the tile conv (conv_tp), the wide pair conv (conv_px3) and the weight gradient (conv_wgrad_x3) -- in a loop on a side stream
and, next to them on the main stream, the kernel that showed the damage (u2mkd_ti_weights) and an element-wise division; every
result must equal, bit for bit, the one computed on an idle GPU.  A build with -DU2MKD_MFMA_GFX950_K32=1 fails it within a few
rounds (tools/ab_mfma_form.sh runs that check)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROUNDS = int(os.environ.get('U2MKD_CONCURRENCY_ROUNDS', '150'))


def test_matrix_kernels_on_a_side_stream_leave_the_main_streams_results_alone(hip):
    from u2mkd_amd.synth import synth_batch
    from u2mkd_amd.torchsparse.nn import functional as F
    b = synth_batch(80000, 1, seed=1234)
    coords = torch.from_numpy(b['coords']).cuda()
    km = F.build_kmap(coords, (1, 1, 1), (3, 3, 3), (1, 1, 1))
    n = km.n_out
    g = torch.Generator(device='cuda').manual_seed(0)
    # aggressors: the three matrix kernels of the library
    x64 = torch.randn(n, 64, device='cuda', generator=g)
    w64 = torch.randn(27, 64, 64, device='cuda', generator=g) / 40
    x128 = torch.randn(n, 128, device='cuda', generator=g)
    w128 = torch.randn(27, 128, 128, device='cuda', generator=g) / 60
    sch, ps = km.schedule(False), km.pair_schedule()
    wf128 = F._weight_layout(w128, True, True)
    o64, o128, dw = torch.empty(n, 64, device='cuda'), torch.empty(n, 128, device='cuda'), torch.empty_like(w64)
    pairs, _, plan = km.pairs_plan()
    lib = hip.load()
    nbytes = lib.u2mkd_conv_wgrad_pairs_workspace_bytes(n, 64, 64, 27)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')

    def aggressors():
        st = hip.stream()
        sch.run(x64, w64, True, 64, False, o64)                                   # conv_tp
        ps.run(x128, wf128, 128, False, o128, fragments=True)                     # conv_px3 + gather-sum
        hip.call('u2mkd_conv_wgrad_pairs', hip.ptr(x64), 64, hip.ptr(x64), 64, hip.ptr(pairs), hip.ptr(plan), n, 27, 0, hip.ptr(ws),
                 nbytes, hip.ptr(dw), st)                                          # conv_wgrad_x3 + reduce
    # victims: trilinear weights (select on a compared index, two divisions per weight) and an element-wise division
    pts = (coords.float() * 0.05) / 0.05
    idx_kn = torch.randint(-1, n, (8, n), device='cuda', generator=g, dtype=torch.int64)
    idx_kn[idx_kn % 3 == 0] = -1
    num = torch.rand(n, 16, device='cuda', generator=g) * 50 + 0.5

    def victims():
        w, i8 = F.ti_weights_n8(pts, idx_kn, scale=2)
        return w, i8, (num * 0.05) / 0.05
    torch.cuda.synchronize()
    ref = victims()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    bad = []
    for r in range(ROUNDS):
        with torch.cuda.stream(side):
            for _ in range(3):
                aggressors()
        got = [victims() for _ in range(4)]
        with torch.cuda.stream(side):
            aggressors()
        torch.cuda.synchronize()
        for j, res in enumerate(got):
            for name, a, e in zip(('ti_weights w', 'ti_weights idx', 'division'), res, ref):
                if not torch.equal(a, e):
                    bad.append((r, j, name, int((a != e).sum())))
    assert not bad, ('results changed while another stream ran the library\'s matrix kernels: %d cases, first %s' % (len(bad), bad[:5]))
