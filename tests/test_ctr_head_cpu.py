"""Host-side checks of cafe_amd.ctr_models (pure torch, runs without a GPU): the dot interaction in its [F, B, D] form
against the reference's list form (dlrm_s_pytorch.py:779-809), head shapes, and the DCN cross-weight quirk."""
import numpy as np
import pytest
import torch

from cafe_amd.ctr_models import CafeCTRHead, create_mlp, interact_dot


def _reference_interact(x, ly, itself):
    # dlrm_s_pytorch.py:782-809, written out
    b, d = ly[0].shape
    T = torch.cat(([x] if x is not None else []) + ly, dim=1).view((b, -1, d))
    Z = torch.bmm(T, torch.transpose(T, 1, 2))
    _, ni, nj = Z.shape
    offset = 1 if itself else 0
    li = torch.tensor([i for i in range(ni) for j in range(i + offset)])
    lj = torch.tensor([j for i in range(nj) for j in range(i + offset)])
    zflat = Z[:, li, lj]
    return torch.cat([x, zflat], dim=1) if x is not None else zflat


@pytest.mark.parametrize("dense,itself", [(True, False), (True, True), (False, False)])
def test_interaction_matches_reference_form(dense, itself):
    torch.manual_seed(0)
    F, B, D = 5, 7, 8
    emb = torch.randn(F, B, D)
    x = torch.randn(B, D) if dense else None
    want = _reference_interact(x, [emb[f] for f in range(F)], itself)
    assert torch.equal(interact_dot(x, emb, itself), want)
    assert torch.equal(interact_dot(x, [emb[f] for f in range(F)], itself), want)


@pytest.mark.parametrize("kind,dense_dim", [("dlrm", 13), ("dlrm", 0), ("wdl", 13), ("wdl", 0), ("dcn", 13), ("dcn", 0)])
def test_head_shapes_and_gradients(kind, dense_dim):
    F, B, D = 4, 6, 8
    head = CafeCTRHead(kind, F, D, dense_dim, ln_bot=[dense_dim, 16, D] if dense_dim else None, ln_top=[0, 32, 1], seed=1)
    emb = torch.randn(F, B, D, requires_grad=True)
    x = torch.rand(B, dense_dim) if dense_dim else None
    p = head(x, emb)
    assert p.shape == (B, 1) and float(p.min()) > 0 and float(p.max()) < 1
    p.sum().backward()
    assert emb.grad is not None and emb.grad.abs().sum() > 0


def test_dcn_cross_weights_follow_the_reference():
    ref_like = CafeCTRHead("dcn", 3, 4, 2)
    names = [n for n, _ in ref_like.named_parameters()]
    assert not any("cross" in n for n in names)                  # dcn.py:415-420: plain lists, the optimizer never sees them
    assert any("cross_weight_0" in n for n, _ in ref_like.named_buffers())
    trained = CafeCTRHead("dcn", 3, 4, 2, train_cross=True)
    assert sum("cross" in n for n, _ in trained.named_parameters()) == 6


def test_mlp_initialisation_is_seeded_like_the_reference():
    a = create_mlp([4, 8, 1], 1, np.random.RandomState(3))
    b = create_mlp([4, 8, 1], 1, np.random.RandomState(3))
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))
    assert isinstance(a[-1], torch.nn.Sigmoid) and isinstance(a[1], torch.nn.ReLU)     # dlrm_s_pytorch.py:243-246
    w = a[0].weight.detach().numpy()
    assert abs(w.std() - np.sqrt(2 / (4 + 8))) < 0.2
