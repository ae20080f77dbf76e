"""Thin DLRM / WDL / DCN models (cafe_amd.ctr_models) on the GPU against the same torch heads on the CPU behind the
oracle's embedding layer (tests/cpu_pipeline.py): per-step losses within 1e-5 relative, sketch decisions bit-exact,
tables within 1e-5 relative.  The heads are stock torch ops on both sides; what is compared is that the HIP
embedding path feeds and trains them exactly like the reference's order of calls does."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from cpu_pipeline import CpuPipeline  # noqa: E402

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def _close(a, b, what, rtol=RTOL):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= rtol, f"{what}: max rel err {err:.3e}"


@pytest.mark.parametrize("kind,dense_dim,per_field", [("dlrm", 13, True), ("wdl", 13, True), ("dcn", 13, False),
                                                       ("wdl", 0, False), ("dcn", 0, True), ("dlrm", 0, False)])
def test_model_losses_match_cpu_twin(kind, dense_dim, per_field, oracle_mod):
    from cafe_amd.ctr_models import CafeCTRModel
    torch.backends.cuda.matmul.allow_tf32 = False
    counts = [4000, 197, 25000, 800, 12000, 161]
    D, B, lr, thr = 16, 256, 0.5, 3
    torch.manual_seed(7)
    model = CafeCTRModel(kind, counts, D, dense_dim=dense_dim, compress_rate=0.05, hash_rate=0.3, threshold=thr,
                         alpha=1.0000005, ln_bot=[dense_dim, 32, D] if dense_dim else None, ln_top=[0, 64, 32, 1],
                         seed=5, per_field_calls=per_field)
    coll = model.embedding
    dev = coll.device
    twin = copy.deepcopy(model.head).cpu()
    cpu = CpuPipeline(oracle_mod, coll, thr, 1, 1.0000005)
    opt = torch.optim.SGD(model.head.parameters(), lr=lr)
    opt_c = torch.optim.SGD(twin.parameters(), lr=lr)
    bce = torch.nn.BCELoss(reduction="mean")
    rng = np.random.default_rng(3)
    F = len(counts)
    newly = 0
    for step in range(12):
        ids = np.stack([(rng.zipf(1.1, B) - 1) % n for n in counts]).astype(np.int64)
        ids[0] = np.maximum(ids[0], 1)
        dense = rng.random((B, dense_dim)).astype(np.float32) if dense_dim else None
        y = (rng.random((B, 1)) < 0.25).astype(np.float32)
        loss = model.train_step(torch.from_numpy(dense).to(dev) if dense_dim else None, torch.from_numpy(ids).to(dev),
                                torch.from_numpy(y).to(dev), opt, lr)
        # CPU twin, the reference's order: forward, backward, insert_grad per field, SGD
        emb = torch.from_numpy(cpu.forward(ids)).requires_grad_(True)
        opt_c.zero_grad(set_to_none=True)
        loss_c = bce(twin(torch.from_numpy(dense) if dense_dim else None, emb), torch.from_numpy(y))
        loss_c.backward()
        g = emb.grad.numpy()
        scores = np.stack([oracle_mod.ref_scores(g[f]) for f in range(F)])
        ins = cpu.backward_update(ids, g, scores, lr, per_field)
        opt_c.step()
        newly += int((ins != 0).sum())
        assert abs(loss.item() - loss_c.item()) <= RTOL * abs(loss_c.item()), f"{kind} step {step}: loss {loss.item()} vs {loss_c.item()}"
        assert np.array_equal(coll._buf["ins"].cpu().numpy().reshape(F, B), ins), f"{kind} step {step}: ins"
    # decisions bit-exact; the accumulated scores, tables and head parameters are compared at 1e-4, not 1e-5: the twin's
    # head runs on the CPU's GEMMs, the model's on rocBLAS, whose float32 results differ by ~1e-6 relative per product;
    # the difference feeds back through SGD on the head at every step, so after these steps the two parameter sets have
    # drifted by a few 1e-5 although every single step agrees to 1e-5 (the losses above are held to RTOL = 1e-5 per
    # step).  The embedding path itself is held to 1e-5 with identical gradients in test_gpu_collection.py.
    a, b = coll.sketch.state(), cpu.sk.state()
    for k in ("mval", "mdic", "mflip", "lval", "ldic", "lt", "lflip", "fifo"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), f"{kind}: sketch {k}"
    for k in ("num", "real_n", "clk", "flip", "batch_num", "resets"):
        assert a[k] == b[k], f"{kind}: sketch {k}: {a[k]} vs {b[k]}"
    _close(a["mcnt"], b["mcnt"], "main scores", 1e-4)
    _close(a["lcnt"], b["lcnt"], "LRU scores", 1e-4)
    _close(coll.weight_high.cpu().numpy(), cpu.w_hot, "hot table", 1e-4)
    for f in range(F):
        _close(coll.weight_hash[f].cpu().numpy(), cpu.w_cold[f], f"cold table {f}", 1e-4)
    for (n, p), (_, q) in zip(model.head.named_parameters(), twin.named_parameters()):
        _close(p.detach().cpu().numpy(), q.detach().numpy(), n, 1e-4)
    assert newly > 20
