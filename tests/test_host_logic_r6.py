"""Host-side rules added in round 6 (no GPU): which Linear weights count as "the parameter itself" for a deferred weight
gradient, the KL term's torch path, the statistics hand-over between a convolution and its BatchNorm, the teacher -> student
row index, the switches' defaults."""
import numpy as np
import torch
from torch import nn


def test_a_reshaping_view_of_a_leaf_is_traced_to_its_parameter_and_nothing_else_is():
    from u2mkd_amd.torchsparse.nn.functional import _leaf_behind_view
    conv = nn.Conv1d(8, 12, 1)
    w = conv.weight.squeeze(-1)                                   # fusion_blocks.py's k = 1 Conv1d evaluated on rows
    assert _leaf_behind_view(w) is conv.weight
    assert _leaf_behind_view(conv.weight.view(12, 8)) is conv.weight
    assert _leaf_behind_view(conv.weight) is None                 # (a leaf is handled by the caller)
    assert _leaf_behind_view(conv.weight.squeeze(-1) * 1.0) is None          # arithmetic in between: a copy
    assert _leaf_behind_view(torch.nn.functional.pad(conv.weight.squeeze(-1), (0, 4))) is None
    assert _leaf_behind_view(conv.weight.squeeze(-1).t()) is None            # same elements, another order
    assert _leaf_behind_view(conv.weight[:6].squeeze(-1)) is None            # a part of the parameter
    frozen = nn.Conv1d(8, 12, 1).requires_grad_(False)
    assert _leaf_behind_view(frozen.weight.squeeze(-1)) is None


def test_kl_term_on_the_cpu_is_torchs_formulation_with_the_row_index_applied():
    from u2mkd_amd.losses import kl_div_logits
    torch.manual_seed(0)
    s = torch.randn(50, 7, requires_grad=True)
    t = torch.randn(64, 7)
    idx = torch.randint(0, 64, (50,))
    crit = nn.KLDivLoss(reduction='batchmean')
    got = kl_div_logits(s, t, idx, crit)
    want = crit(torch.log_softmax(s, 1), torch.softmax(t[idx], 1))
    assert torch.allclose(got, want)
    got.backward()
    assert s.grad is not None and torch.isfinite(s.grad).all()


def test_statistics_describe_exactly_the_tensor_they_came_with():
    from u2mkd_amd.torchsparse.nn.functional import BnStats
    st = BnStats()
    x = torch.zeros(10, 8)
    assert not st.describes(x)                                    # nothing left by the store
    st.partial, st.slab_rows, st.of = torch.zeros(4), 32, (x.data_ptr(), 10, 8)
    assert st.describes(x)
    assert not st.describes(x[:, :4]) and not st.describes(x.clone()) and not st.describes(x.double())


def test_teacher_to_student_index_composes_inverse_map_keyframe_mask_and_student_rows():
    from u2mkd_amd.kd import teacher_to_student, teacher_to_student_index
    rng = np.random.default_rng(1)
    num_pts, num_vox_t = [40, 25], [11, 7]
    inv = torch.from_numpy(np.concatenate([rng.integers(0, v, p) for p, v in zip(num_pts, num_vox_t)]))
    kf = torch.from_numpy(rng.random(sum(num_pts)) < 0.7)
    kept = [int(kf[:40].sum()), int(kf[40:].sum())]
    inds = [(torch.from_numpy(rng.integers(0, k, 9)),) for k in kept]
    x_t = torch.randn(sum(num_vox_t), 5)
    idx = teacher_to_student_index(inv, inds, num_pts, num_vox_t, kf)
    want = torch.cat([x_t[:11][inv[:40]][kf[:40]][inds[0][0]], x_t[11:][inv[40:]][kf[40:]][inds[1][0]]])
    assert torch.equal(x_t.index_select(0, idx), want)
    assert torch.equal(teacher_to_student(x_t, inv, inds, num_pts, num_vox_t, kf), want)


def test_round_6_switch_defaults():
    from u2mkd_amd import fusion, kd, losses, pixel_head
    from u2mkd_amd.torchsparse.nn import functional as F
    assert F._PREFETCH_PLANS is False and F._CONV_BN_STATS is False           # built, measured without gain: off
    assert F._LEAF_VIEWS and fusion._L2C_COMBINE and losses._FUSED_KL and losses._FUSED_CE and pixel_head._ROW_BN
    assert kd._TEACHER_AHEAD == 0 and kd._TEACHER_TAG is None
