"""GPU parity at the shapes of BASELINE.json's other configurations (cfg3 is test_gpu_fullsize.py):

  cfg2  26 Kaggle fields (33.8 M ids), D = 16, batch 8192, compress 0.001: the whole path for 30 steps
  cfg4  880 M ids, D = 128, compress 0.1 / hash-rate 0.7: a 24 M-slot sketch and 44 GB of tables on one GPU, 10 steps;
        every table row the last two steps touch is followed on the host
  cfg5  22 Avazu-shaped fields, no dense features, behind the WDL and DCN heads

Bars: routing, ins[] and the complete sketch state bit-exact (integers and the float scores, which both sides form
from the same inputs in the same order); gathered rows exact copies; updated table rows and losses within 1e-5
relative -- checked twice: max-norm, and ELEMENTWISE |a - b| <= 1e-5 |b| + atol.  The absolute term is the float32
rounding of the SGD updates themselves (the GPU adds the gradient rows of one table row in float32, the host
restatement in float64; an element near zero after cancellation has no relative accuracy to speak of):
  ATOL = 2e-6   one or two updates of rows touched by up to ~10^3 lookups (cfg4: ulp(10^2) * lr ~ 1e-6)
  2e-5          cfg2 / cfg5 tables followed for all steps: a cold table there has a few hundred rows, each shared by
                thousands of lookups per step, so row values reach ~10 and 30 updates of |sum| ~ 10^2..10^3 accumulate."""
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
ATOL = 2e-6


def _close(a, b, what, rtol=RTOL, atol=ATOL):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= rtol, f"{what}: max-norm rel err {err:.3e}"
    bad = np.abs(a - b) > rtol * np.abs(b) + atol
    assert not bad.any(), (f"{what}: {int(bad.sum())} of {bad.size} elements beyond {rtol} |b| + {atol}; worst "
                           f"|a-b| = {np.abs(a - b)[bad].max():.3e}")


@pytest.mark.parametrize("split_update", [1, 2], ids=["update_by_rule", "update_split"])
def test_cfg2_whole_path(split_update, oracle_mod):
    """split_update = 2 forces the two-kernel form of the update (segment sums on the library's third stream beside the
    insert, row updates behind the migration), which the library's own rule reserves for steps with 64 .. 512 MB of
    gradient (cfg3 / cfg4 shapes; test_cfg4_full_size runs it by rule)."""
    from cafe_amd.collection import CafeEmbeddingCollection
    from cafe_amd.config import workload
    wl = workload("cfg2")
    F, B, D, thr, lr = wl.num_fields, wl.batch, wl.dim, 5, 0.05
    assert (F, B, D) == (26, 8192, 16)
    coll = CafeEmbeddingCollection(wl.counts, D, wl.hotn, wl.hash_sizes, wl.offsets, threshold=thr, alpha=wl.alpha, seed=8)
    coll.sketch.set_option("split_update", split_update)
    cpu = CpuPipeline(oracle_mod, coll, thr, 1, wl.alpha)
    steps = 30
    ids_all = wl.make_ids(31, steps=steps)
    gen = torch.Generator(device="cuda").manual_seed(6)
    newly = 0
    for step in range(steps):
        ids = ids_all[step]
        out = coll.forward(torch.from_numpy(ids).cuda()).cpu().numpy()
        want = cpu.forward(ids)
        assert np.array_equal(coll._buf["que"].cpu().numpy().reshape(F, B), np.stack(cpu.que)), f"step {step}: routing"
        _close(out, want, f"step {step} forward", atol=2e-5)
        grad = torch.randn(F, B, D, device="cuda", generator=gen)
        scores, ins = coll.backward_update(grad, lr, per_field_calls=True)
        sc = scores.cpu().numpy().reshape(F, B)
        g = grad.cpu().numpy()
        if step % 10 == 0:
            for f in range(F):
                _close(sc[f], oracle_mod.ref_scores(g[f]), f"step {step} field {f} scores", atol=1e-6)
        ins_want = cpu.backward_update(ids, g, sc, lr, True)
        assert np.array_equal(ins.cpu().numpy().reshape(F, B), ins_want), f"step {step}: ins differs"
        newly += int((ins_want != 0).sum())
        if step % 10 == 9:
            _close(coll.weight_high.cpu().numpy(), cpu.w_hot, f"step {step} hot table", atol=2e-5)
            for f in range(F):
                _close(coll.weight_hash[f].cpu().numpy(), cpu.w_cold[f], f"step {step} cold table {f}", atol=2e-5)
    assert not oracle_mod.states_equal(coll.sketch.state(), cpu.sk.state())
    c = coll.sketch.counters()
    assert c["cadence_miss"] == 0 and newly > 500, (c, newly)


def _rows(t, idx):
    return t.index_select(0, torch.from_numpy(np.asarray(idx, np.int64)).to(t.device)).cpu().numpy()


def test_cfg4_full_size(oracle_mod):
    """compress 0.1 / hash-rate 0.7: hotn = 24.1 M slots (sketch state ~1 GB), 12.4 GB hot table, 31.5 GB cold tables.
    The tables do not fit a host restatement, so the last two steps follow every ROW THEY TOUCH: rows are fetched from
    the GPU before the step, the reference's migration and SGD (sk_embedding_bag.py:199-208, dlrm_s_pytorch.py:1658) are
    applied to them on the host, and the same rows are fetched again after the step."""
    from cafe_amd.collection import CafeEmbeddingCollection
    from cafe_amd.config import workload
    wl = workload("cfg4")
    F, B, D, thr, lr = wl.num_fields, wl.batch, wl.dim, wl.threshold, 0.05
    assert wl.compress_rate == 0.1 and wl.hash_rate == 0.7 and wl.hotn > 20_000_000
    coll = CafeEmbeddingCollection(wl.counts, D, wl.hotn, wl.hash_sizes, wl.offsets, threshold=thr, alpha=wl.alpha, seed=4,
                                   init_on_device=True)
    ora = oracle_mod.Oracle(wl.hotn, thr, 1, wl.alpha, 1)
    steps, table_steps = 10, 2
    ids_all = wl.make_ids(55, steps=steps)
    gen = torch.Generator(device="cuda").manual_seed(2)
    hs = np.asarray(wl.hash_sizes, np.int64)
    newly = 0
    for step in range(steps):
        ids = ids_all[step]
        gids = (ids + wl.offsets[:, None]).astype(np.int64)
        que = np.stack([ora.query(gids[f]) for f in range(F)]).astype(np.int64)
        out = coll.forward(torch.from_numpy(ids).cuda())
        assert np.array_equal(coll._buf["que"].cpu().numpy().reshape(F, B), que), f"step {step}: routing"
        follow = step >= steps - table_steps
        if follow:
            hot = que < 0
            hot_rows = np.unique((-que[hot]) % wl.hotn)
            cold_rows = [np.unique(np.concatenate([que[f][~hot[f]] % hs[f], ids[f] % hs[f]])) for f in range(F)]
            pre_hot = _rows(coll.weight_high, hot_rows)
            pre_cold = [_rows(coll.weight_hash[f], cold_rows[f]) for f in range(F)]
            got = out.cpu().numpy()
            for f in range(F):          # gathered rows are exact copies of the rows the oracle's routing selects
                src_h = pre_hot[np.searchsorted(hot_rows, (-que[f]) % wl.hotn).clip(0, len(hot_rows) - 1)]
                src_c = pre_cold[f][np.searchsorted(cold_rows[f], que[f] % hs[f]).clip(0, len(cold_rows[f]) - 1)]
                assert np.array_equal(got[f], np.where(hot[f][:, None], src_h, src_c)), f"step {step} field {f}: gather"
        grad = torch.randn(F, B, D, device="cuda", generator=gen)
        scores, ins = coll.backward_update(grad, lr, per_field_calls=True)
        sc = scores.cpu().numpy().reshape(F, B)
        ins_want = np.zeros((F, B), np.int32)
        migr = []
        for f in range(F):
            ins_want[f] = ora.insert_val(gids[f], sc[f])
            if follow and ins_want[f].any():
                qa = ora.query(gids[f]).astype(np.int64)
                for i in np.nonzero(ins_want[f])[0]:
                    if qa[i] < 0:
                        migr.append((int(-qa[i]) % wl.hotn, f, int(ids[f][i] % hs[f])))
        assert np.array_equal(ins.cpu().numpy().reshape(F, B), ins_want), f"step {step}: ins differs"
        newly += int((ins_want != 0).sum())
        if follow:
            g = grad.cpu().numpy().astype(np.float64)
            # migration targets are slots popped in this step: rows not touched by the forward's routing
            tgt = np.array(sorted({m[0] for m in migr}), np.int64)
            exp_tgt = {}
            for slot, f, crow in migr:              # in the reference's order: a later copy into the same slot wins
                exp_tgt[slot] = pre_cold[f][np.searchsorted(cold_rows[f], crow)].astype(np.float64)
            exp_hot = pre_hot.astype(np.float64)
            acc = np.zeros_like(exp_hot)
            for f in range(F):
                h = hot[f]
                np.add.at(acc, np.searchsorted(hot_rows, (-que[f][h]) % wl.hotn), g[f][h])
            # a hot row that is also a migration target takes the copied row first, then the update (the reference's order)
            for slot, row in exp_tgt.items():
                k = np.searchsorted(hot_rows, slot)
                if k < len(hot_rows) and hot_rows[k] == slot:
                    exp_hot[k] = row
            exp_hot -= lr * acc
            _close(_rows(coll.weight_high, hot_rows), exp_hot, f"step {step} hot rows")
            only_tgt = [s for s in tgt if not (np.searchsorted(hot_rows, s) < len(hot_rows) and hot_rows[np.searchsorted(hot_rows, s)] == s)]
            if only_tgt:
                _close(_rows(coll.weight_high, only_tgt), np.stack([exp_tgt[s] for s in only_tgt]), f"step {step} migrated rows")
            for f in range(F):
                c = ~hot[f]
                accc = np.zeros(pre_cold[f].shape, np.float64)
                np.add.at(accc, np.searchsorted(cold_rows[f], que[f][c] % hs[f]), g[f][c])
                _close(_rows(coll.weight_hash[f], cold_rows[f]), pre_cold[f].astype(np.float64) - lr * accc, f"step {step} cold rows {f}")
    bad = oracle_mod.states_equal(coll.sketch.state(), ora.state())
    assert not bad, bad[:3]
    c = coll.sketch.counters()
    assert c["cadence_miss"] == 0 and newly > 1000, (c, newly)


@pytest.mark.parametrize("kind", ["wdl", "dcn"])
def test_cfg5_avazu_shaped_models(kind, oracle_mod):
    """22 Avazu-shaped fields (9.4 M ids), D = 16, no dense features (ArtifactEvaluation/models.py:57-58, :87-88),
    batch 8192.  The CPU twin runs the same torch head behind the oracle's embedding layer: losses agree to 1e-5.
    The two heads' embedding gradients come from two matmul implementations; a ReLU whose input rounds to either side of
    zero makes a sample's gradient differ by a finite amount, so they are compared per element with a quota (99.9 %
    within 1e-5) and the host pipeline is then fed the GPU head's gradient: scores, decisions and table updates of the
    embedding path itself are compared on equal inputs, to the bars in this file's header."""
    from cafe_amd.config import workload
    from cafe_amd.ctr_models import CafeCTRModel
    torch.backends.cuda.matmul.allow_tf32 = False
    wl = workload("cfg5")
    F, B, D, lr, thr = wl.num_fields, wl.batch, wl.dim, 0.1, 4
    assert F == 22 and D == 16
    torch.manual_seed(3)
    model = CafeCTRModel(kind, wl.counts, D, dense_dim=0, compress_rate=wl.compress_rate, hash_rate=wl.hash_rate,
                         threshold=thr, alpha=wl.alpha, seed=9, per_field_calls=True)
    assert not model.plain_idx
    coll = model.embedding
    dev = coll.device
    twin = copy.deepcopy(model.head).cpu()
    cpu = CpuPipeline(oracle_mod, coll, thr, 1, wl.alpha)
    opt = torch.optim.SGD(model.dense_parameters(), lr=lr)
    opt_c = torch.optim.SGD(twin.parameters(), lr=lr)
    bce = torch.nn.BCELoss(reduction="mean")
    rng = np.random.default_rng(12)
    ids_all = wl.make_ids(9, steps=8)
    newly = 0
    for step in range(8):
        ids = ids_all[step]
        y = (rng.random((B, 1)) < 0.2).astype(np.float32)
        # CafeCTRModel.train_step, spelled out to keep the embedding gradient (dlrm_s_pytorch.py:1640-1658)
        opt.zero_grad(set_to_none=True)
        loss = model.loss_fn(model(None, torch.from_numpy(ids).to(dev)), torch.from_numpy(y).to(dev))
        loss.backward()
        g_gpu = model._emb_grad
        coll.backward_update(g_gpu, lr, per_field_calls=True)
        model._emb_grad = None
        opt.step()
        emb = torch.from_numpy(cpu.forward(ids)).requires_grad_(True)
        opt_c.zero_grad(set_to_none=True)
        loss_c = bce(twin(None, emb), torch.from_numpy(y))
        loss_c.backward()
        opt_c.step()
        assert abs(loss.item() - loss_c.item()) <= RTOL * abs(loss_c.item()), f"{kind} step {step}: loss {loss.item()} vs {loss_c.item()}"
        g = g_gpu.cpu().numpy()
        gc = emb.grad.numpy()
        off = np.abs(g - gc) > RTOL * np.abs(gc) + 1e-9
        assert off.mean() < 1e-3, f"{kind} step {step}: {off.mean():.2e} of the embedding gradient beyond 1e-5"
        sc = coll._buf["scores"].cpu().numpy().reshape(F, B)
        for f in range(F):
            _close(sc[f], oracle_mod.ref_scores(g[f]), f"{kind} step {step} field {f} scores", atol=1e-6)
        ins = cpu.backward_update(ids, g, sc, lr, True)
        newly += int((ins != 0).sum())
        assert np.array_equal(coll._buf["ins"].cpu().numpy().reshape(F, B), ins), f"{kind} step {step}: ins"
    assert not oracle_mod.states_equal(coll.sketch.state(), cpu.sk.state())
    _close(coll.weight_high.cpu().numpy(), cpu.w_hot, "hot table", atol=2e-5)
    for f in range(F):
        _close(coll.weight_hash[f].cpu().numpy(), cpu.w_cold[f], f"cold table {f}", atol=2e-5)
    assert newly > 100
