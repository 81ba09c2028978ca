"""The reference's own, UNCHANGED model files import and construct over the HIP drop-in
(``install_as_torchsparse()`` / ``install_as_sptr()``) -- build container only: /root/reference does not exist
on the GPU box, nothing of it is stored here.  Construct-only (parameters live on the CPU; every operator of
the drop-in refuses CPU tensors): what is checked is that the package surface is complete enough for
core/models/{build_blocks,utils}.py, semantickitti/spvcnn.py, sphereformer/spherical_transformer.py and
nuscenes/spvcnn_spformer.py, and that the models come out with the reference's parameter counts
(5 449 169 / 21 777 809 / 87 073 553 for cr 0.5 / 1.0 / 2.0)."""
import os
import subprocess
import sys

import pytest

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, types
sys.path.insert(0, %(root)r)
import torch
import u2mkd_amd
u2mkd_amd.install_as_torchsparse()
u2mkd_amd.install_as_sptr()
# modules the reference imports that are neither in this image nor part of the hot path
for name in ('torchvision', 'torchvision.transforms', 'torchvision.transforms.functional', 'timm', 'timm.models',
             'torchpack', 'torchpack.utils'):
    sys.modules.setdefault(name, types.ModuleType(name))
layers = types.ModuleType('timm.models.layers')
class DropPath(torch.nn.Module):
    def __init__(self, p=0.0):
        super().__init__(); self.drop_prob = p
    def forward(self, x):
        return x
layers.DropPath, layers.trunc_normal_ = DropPath, torch.nn.init.trunc_normal_
sys.modules['timm.models.layers'] = layers
ts_ = types.ModuleType('torch_scatter')      # imported by spvcnn_spformer.py:9, never called on this path
ts_.scatter_mean = None
sys.modules['torch_scatter'] = ts_
cfg = types.ModuleType('torchpack.utils.config')
cfg.configs = {'model': {'cr': 1.0, 'in_channel': 4}, 'data': {'num_classes': 17}}
sys.modules['torchpack.utils.config'] = cfg
sys.path.insert(0, %(ref)r)

import torchsparse, torchsparse.nn as spnn
assert torchsparse.__name__ == 'u2mkd_amd.torchsparse'
from core.models.semantickitti.spvcnn import SPVCNN
from core.models.build_blocks import ResidualBlock
counts = {}
for cr in (0.5, 1.0, 2.0):
    m = SPVCNN(cr=cr, in_channel=4, num_classes=17, pres=0.05, vres=0.05)
    counts[cr] = sum(p.numel() for p in m.parameters())
    convs = [x for x in m.modules() if isinstance(x, spnn.Conv3d)]
    assert len(convs) == 49 - 7 + 7 or len(convs) > 40, len(convs)
    assert all(c.kernel.dim() in (2, 3) for c in convs)
print('COUNTS', counts[0.5], counts[1.0], counts[2.0])

import third_party.SparseTransformer.sptr as sptr
assert sptr.__name__ == 'u2mkd_amd.sptr'
from core.models.sphereformer import spherical_transformer as ST
assert ST.sparse_self_attention is sptr.sparse_self_attention and ST.get_indices_params is sptr.get_indices_params
from core.models.nuscenes.spvcnn_spformer import SPVCNN_SPFORMER
import numpy as np
kw = dict(pres=0.05, vres=0.05, window_size=np.array([0.3, 0.3, 0.3]),
          window_size_sphere=[2, 2, 120], quant_size=np.array([0.3, 0.3, 0.3]) / 24, quant_size_sphere=np.array([2, 2, 120]) / 24,
          window_size_scale=[2.0, 2.0], drop_path_rate=0.3, a=0.05 * 0.25)
t = SPVCNN_SPFORMER(**kw)
blocks = [x for x in t.modules() if isinstance(x, ST.SphereFormer)]
print('SPFORMER', len(blocks), sum(p.numel() for p in t.parameters()))
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'core', 'models')), reason='reference tree not present (GPU box)')
def test_reference_model_files_construct_over_the_dropin():
    code = CHILD % {'root': ROOT, 'ref': REF}
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = dict(line.split(' ', 1) for line in r.stdout.strip().splitlines() if ' ' in line)
    assert out['COUNTS'].split() == ['5449169', '21777809', '87073553'], out['COUNTS']
    n_blocks, n_params = (int(v) for v in out['SPFORMER'].split())
    assert n_blocks == 4 and n_params > 21777809
