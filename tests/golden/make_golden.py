"""Generate golden vectors by importing the REFERENCE's own Python modules
(from /root/reference, build container only) over the CPU oracle operators.

    python tests/golden/make_golden.py

Outputs small .npz / .json fixtures next to this file.  The fixtures are data
(inputs are regenerated from seeds, expected outputs are stored); no reference
source travels.  What each fixture pins:

* spvcnn_cr05_4000.npz  -- logits + loss of the reference `SPVCNN` class
  (core/models/semantickitti/spvcnn.py) with `core/models/utils.py` and
  `core/models/build_blocks.py` unchanged, run over oracle.torchsparse_cpu:
  pins the MODEL WIRING of oracle.spvcnn_ref (bit-exact on CPU) and of
  u2mkd_amd.lidar.SPVCNN (1e-3 on the GPU), and the state-dict key set.
* lovasz_ce.npz -- reference `MixLovaszCrossEntropy` (core/criterions.py) values
  and gradients on seeded logits: pins oracle + u2mkd_amd.losses.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

from oracle import torchsparse_cpu as ots  # noqa: E402
from oracle import spvcnn_ref as O         # noqa: E402
from u2mkd_amd.synth import synth_batch    # noqa: E402


def import_reference():
    ots.install()                                   # `import torchsparse` -> CPU oracle
    sys.modules.setdefault('torchvision', types.ModuleType('torchvision'))
    tvt = types.ModuleType('torchvision.transforms')
    tvf = types.ModuleType('torchvision.transforms.functional')
    sys.modules.setdefault('torchvision.transforms', tvt)
    sys.modules.setdefault('torchvision.transforms.functional', tvf)
    sys.path.insert(0, REF)
    from core.models.semantickitti.spvcnn import SPVCNN     # the reference class
    from core.criterions import MixLovaszCrossEntropy
    return SPVCNN, MixLovaszCrossEntropy


def import_reference_spformer(cr):
    """The reference's SphereFormer / SPVCNN_SPFORMER modules with their un-installable imports
    stubbed: timm (DropPath, trunc_normal_), torch_scatter, torchpack configs, and
    third_party.SparseTransformer.sptr -> oracle.sptr_cpu (CPU restatement of the CUDA ops)."""
    from oracle import sptr_cpu
    from oracle.spformer_ref import DropPath
    timm = types.ModuleType('timm')
    timm_models = types.ModuleType('timm.models')
    timm_layers = types.ModuleType('timm.models.layers')
    timm_layers.DropPath = DropPath
    timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
    sys.modules.update({'timm': timm, 'timm.models': timm_models, 'timm.models.layers': timm_layers})
    tscatter = types.ModuleType('torch_scatter')
    tscatter.scatter_mean = None
    sys.modules['torch_scatter'] = tscatter
    for name in ('third_party', 'third_party.SparseTransformer'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['third_party.SparseTransformer.sptr'] = sptr_cpu
    tp = types.ModuleType('torchpack')
    tpu = types.ModuleType('torchpack.utils')
    tpc = types.ModuleType('torchpack.utils.config')
    tpc.configs = {'model': {'cr': cr, 'in_channel': 4}, 'data': {'num_classes': 17}}
    sys.modules.update({'torchpack': tp, 'torchpack.utils': tpu, 'torchpack.utils.config': tpc})
    from core.models.nuscenes.spvcnn_spformer import SPVCNN_SPFORMER
    return SPVCNN_SPFORMER


def main():
    SPVCNN, MixLovaszCrossEntropy = import_reference()
    torch.manual_seed(0)
    kw = dict(cr=0.5, in_channel=4, num_classes=17, pres=0.05, vres=0.05)
    b = synth_batch(4000, 1, seed=21)
    feats, coords, labels = (torch.from_numpy(b[k]) for k in ('feats', 'coords', 'labels'))

    ref = O.fill_state_by_name(SPVCNN(**kw)).train()
    ref.dropout.p = 0.0
    out = ref({'lidar': ots.SparseTensor(feats.clone(), coords.clone())})['x_vox']
    crit = MixLovaszCrossEntropy(ignore_index=0)
    loss = crit(out, labels)
    loss.backward()
    sd = ref.state_dict()
    grads = {n: p.grad for n, p in ref.named_parameters()}
    np.savez_compressed(
        os.path.join(HERE, 'spvcnn_cr05_4000.npz'),
        logits=out.detach().numpy().astype(np.float32), loss=np.float32(loss.item()),
        grad_stem0=grads['stem.0.kernel'].numpy(), grad_cls=grads['classifier_vox.0.weight'].numpy(),
        grad_up3=grads['vox_ups.3.1.1.net.3.kernel'].numpy()[13])
    with open(os.path.join(HERE, 'spvcnn_cr05_keys.json'), 'w') as f:
        json.dump({k: list(v.shape) for k, v in sd.items()}, f, indent=0)

    g = torch.Generator().manual_seed(5)
    x = torch.randn(3000, 17, generator=g, requires_grad=True)
    y = torch.randint(0, 17, (3000,), generator=g)
    l2 = crit(x, y)
    l2.backward()
    np.savez_compressed(os.path.join(HERE, 'lovasz_ce.npz'), x=x.detach().numpy(), y=y.numpy(),
                        loss=np.float32(l2.item()), grad=x.grad.numpy())
    # ---- SPVCNN_SPFORMER (teacher), the reference class with the builder's arguments
    from oracle.spformer_ref import default_spformer_kwargs
    cr = 1.0
    SPF = import_reference_spformer(cr)
    kw = default_spformer_kwargs(cr=cr, drop_path_rate=0.0)
    for k in ('cr', 'in_channel', 'num_classes'):
        kw.pop(k)
    b = synth_batch(2000, 2, seed=33)
    feats, coords, labels = (torch.from_numpy(b[k]) for k in ('feats', 'coords', 'labels'))
    ref = O.fill_state_by_name(SPF(**kw)).train()
    ref.dropout.p = 0.0
    out = ref({'lidar': ots.SparseTensor(feats.clone(), coords.clone())})['x_vox']
    loss3 = crit(out, labels)
    loss3.backward()
    grads = {n: p.grad for n, p in ref.named_parameters()}
    blk = 'transformer_blocks.1.attn.'
    np.savez_compressed(
        os.path.join(HERE, 'spformer_cr10_4000.npz'),
        logits=out.detach().numpy().astype(np.float32), loss=np.float32(loss3.item()),
        grad_tq=grads[blk + 'relative_pos_query_table'].numpy(),
        grad_tv_sphere=grads[blk + 'relative_pos_value_table_sphere'].numpy(),
        grad_qkv=grads[blk + 'qkv.weight'].numpy())
    with open(os.path.join(HERE, 'spformer_cr10_keys.json'), 'w') as f:
        json.dump({k: list(v.shape) for k, v in ref.state_dict().items()}, f, indent=0)
    print('golden written:', float(loss), float(l2), float(loss3))


if __name__ == '__main__':
    main()
