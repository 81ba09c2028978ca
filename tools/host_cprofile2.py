"""cProfile of the pipelined bench loop (fresh batches, staged geometry), cumulative time of the trainer-level functions.
python tools/host_cprofile2.py"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = bench.parse()
step, n_pts, desc = bench.build_step(args, 0, 'kd', args.image_hw)
for _ in range(6):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(8):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
st = pstats.Stats(pr, stream=s).sort_stats('cumulative')
st.print_stats(r'(train|bench|kd|point_voxel|deferred|synth|distributed)\.py', 40)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(25)
print(s.getvalue()[:5000])
