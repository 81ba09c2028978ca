"""Prototype: the frozen teacher (its geometry pre-pass + forward) in a HELPER PROCESS of the rank, one batch ahead; its outputs
come back through torch's IPC reductions.  Measures the pipelined KD step of the bench with and without it, same process order.
  python tools/exp_teacher_process.py [steps=20]"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.multiprocessing as mp


def helper(q_in, q_out, kwargs, state, resident):
    torch.cuda.set_device(0)
    from u2mkd_amd import torchsparse as ts
    from u2mkd_amd.lidar import SPVCNN_SPFORMER
    from u2mkd_amd.lidar.point_voxel import prepare_geometry
    model = SPVCNN_SPFORMER(**kwargs).cuda()
    model.load_state_dict(state)
    model.requires_grad_(False)
    model.eval()
    keep = collections.deque(maxlen=4)
    q_out.put('ready')
    with torch.no_grad():
        while True:
            i = q_in.get()
            if i is None:
                break
            th = time.perf_counter()
            feats, coords = resident[i][0].clone(), resident[i][1].clone()
            in_mod = {'lidar': ts.SparseTensor(feats, coords)}
            in_mod['_geometry'] = prepare_geometry(in_mod['lidar'], model.pres, model.vres)
            out = model(in_mod)
            torch.cuda.synchronize()
            keep.append(out)
            q_out.put((i, out['x_vox'], out['pts_feats'], time.perf_counter() - th))


if __name__ == '__main__':
    import bench
    from u2mkd_amd import kd as KD, train as T
    STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    sys.argv = sys.argv[:1]
    args = bench.parse()
    step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
    run = step.runner
    model = run.model

    def measure(label, fn, steps=STEPS):
        for _ in range(6):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print('%-40s wall %.2f ms/step  host %.2f' % (label, wall / steps * 1e3, host / steps * 1e3), flush=True)

    measure('in-process teacher (baseline)', step)

    # ---- helper process
    from u2mkd_amd import lidar
    from u2mkd_amd.synth import synth_kd_batch
    sp = {k: v for k, v in lidar.spformer_kwargs().items() if k not in ('cr', 'in_channel', 'num_classes')}
    from copy import deepcopy
    kwargs = dict(cr=args.cr_t, in_channel=4, num_classes=17, return_pts_feats=True, **deepcopy(sp))
    nb = max(1, args.batches)
    resident = [T.kd_batch_to_device(synth_kd_batch(args.voxels, 1, seed=1234 + 97 * i, image_hw=tuple(args.image_hw))) for i in range(nb)]
    shared = [(d['t_feats'], d['t_coords']) for d in resident]
    ctx = mp.get_context('spawn')
    q_in, q_out = ctx.Queue(), ctx.Queue()
    state = {k: v.detach().cpu() for k, v in model.model_t.state_dict().items()}
    proc = ctx.Process(target=helper, args=(q_in, q_out, kwargs, state, shared), daemon=True)
    proc.start()
    assert q_out.get(timeout=300) == 'ready'

    from u2mkd_amd.lidar.point_voxel import prepare_geometry_staged

    def prepare_staged(in_mod):
        (g_s,) = yield from prepare_geometry_staged([(in_mod['student']['lidar'], model.model_s.pres, model.model_s.vres)])
        in_mod['student']['_geometry'] = g_s
        in_mod['teacher']['_geometry'] = None
        return in_mod

    model.prepare_staged = prepare_staged
    pending = collections.deque()
    waits = []
    hold = collections.deque(maxlen=3)

    def forward(in_mod):
        stu_in = in_mod['student']
        stu_in = dict(stu_in, _camera_head=model.model_s.camera_head(stu_in))
        ret = {'stu': model.model_s(stu_in)}
        tg = time.perf_counter()
        i, x_vox, pts, th = q_out.get(timeout=120)
        waits.append((time.perf_counter() - tg, th))
        assert i == pending.popleft()
        hold.append((x_vox, pts))
        ret['t'] = {'x_vox': x_vox, 'pts_feats': pts}
        return ret

    model.forward = forward
    counter = [0]
    nxt = [None]

    def submit(i):
        pending.append(i)
        q_in.put(i)

    submit(0)

    def hstep():
        i = counter[0] % nb
        d = nxt[0] if nxt[0] is not None else T.fresh_batch(resident[i])
        counter[0] += 1
        j = counter[0] % nb
        submit(j)                                  # the helper starts on batch k+1 while this process runs step k
        nxt[0] = T.fresh_batch(resident[j])
        return run(d, prefetch=nxt[0])

    measure('teacher in a helper process', hstep)
    waits.clear()
    measure('teacher in a helper process (again)', hstep)
    w = waits[-20:]
    print('main blocked in get(): mean %.2f ms max %.2f; helper per batch: mean %.2f ms' % (
        sum(a for a, _ in w) / len(w) * 1e3, max(a for a, _ in w) * 1e3, sum(b for _, b in w) / len(w) * 1e3))
    q_in.put(None)
    proc.join(20)
    print('helper exit', proc.exitcode)
