"""What the N>1 code path costs at ONE rank (no communication partner): the pipelined KD step as bench.py runs it, plain
vs DistributedDataParallel + synchronising BatchNorm over a one-rank RCCL group; one variant per process.
    python tools/ddp_cost.py            (plain)
    python tools/ddp_cost.py ddp        (U2MKD_FORCE_DDP=1)
Prints the wall time per step, the launches a step issues (C-ABI calls + aten ops with device work are counted by
tools/op_census.py; here only the collectives) and what wrap_model decided."""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = (sys.argv[1:] or ['plain'])[0]        # plain | ddp | pg (process group only) | wrap (reducer only, no group)
ddp = mode == 'ddp'
if mode in ('ddp', 'pg'):
    os.environ['U2MKD_FORCE_DDP'] = '1'
import torch
import torch.distributed as dist
from u2mkd_amd import distributed as D, lidar, train as T, kd as KD
from u2mkd_amd.synth import synth_kd_batch
if mode == 'pglazy':                        # a group without an eagerly created communicator (no device_id): RCCL stays asleep
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29534')
    dist.init_process_group('nccl', rank=0, world_size=1)
else:
    D.init_from_env()
if mode == 'pg':
    os.environ.pop('U2MKD_FORCE_DDP')       # the group exists, the model stays unwrapped
torch.manual_seed(0)
sp = {k: v for k, v in lidar.spformer_kwargs().items() if k not in ('cr', 'in_channel', 'num_classes')}
model = KD.TSDFull(cr=1.0, cr_t=2.0, in_channel=4, in_channel_t=4, num_classes=17, spformer=sp).cuda()
run = T.KDStep(model, num_epochs=50, batch_size=1)
if mode == 'wrap':
    run.net = D.BucketedGradientAverage(model)
run.train_mode()
if ddp:
    net = run.net
    print('DDP: broadcast_buffers', net.broadcast_buffers, 'buckets', len(net._buckets),
          'params', sum(1 for p in net.parameters() if p.requires_grad))
res = [T.kd_batch_to_device(synth_kd_batch(80000, 1, seed=1234 + i, image_hw=(360, 640))) for i in range(4)]
calls = collections.Counter()
if ddp:
    for name in ('all_reduce', 'all_gather_into_tensor', 'all_gather', 'broadcast', '_broadcast_coalesced', 'reduce_scatter_tensor'):
        if hasattr(dist, name):
            f = getattr(dist, name)
            def wrap(*a, _f=f, _n=name, **k):
                calls[_n] += 1
                return _f(*a, **k)
            setattr(dist, name, wrap)


def loop(steps):
    cur = T.fresh_batch(res[0])
    for i in range(steps):
        nxt = T.fresh_batch(res[(i + 1) % 4])
        run(cur, prefetch=nxt)
        cur = nxt


loop(6)
torch.cuda.synchronize()
calls.clear()
t0 = time.perf_counter()
loop(16)
torch.cuda.synchronize()
print('VARIANT %-8s %.2f ms/step  collectives per step: %s' % (mode, (time.perf_counter() - t0) / 16 * 1e3,
                                                              {k: v / 16 for k, v in calls.items()}), flush=True)
if mode in ('ddp', 'pg', 'pglazy'):
    D.shutdown()
