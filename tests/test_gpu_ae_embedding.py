"""The artifact generation's 'cafe' embedding layer (cafe_amd/ae_embedding.py) on the GPU.

1. Against steps recorded from the reference's own module (ArtifactEvaluation/embeddings/sk_embedding_bag.py driving the
   artifact's sketch.cpp, tests/golden/make_ae_module_golden.py): per-field hot tables with gradient-norm scores and
   with cafe_use_freq, and one field behind the shared sketch.  Routing after every step exact; forward output and all
   tables within 1e-5 relative.
2. The configuration the reference file cannot run on CPU (several fields sharing ONE sketch and hot table, SURVEY
   Q1) against a host pipeline built on the oracle's restatement of that sketch: routing, ins[] and sketch state exact,
   tables within 1e-5."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-5


def _close(a, b, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= RTOL, f"{what}: max rel err {err:.3e}"


@pytest.mark.parametrize("name", ["aemod_separate_three_fields", "aemod_separate_use_freq", "aemod_shared_one_field"])
def test_steps_recorded_from_the_reference_module(name):
    from cafe_amd.ae_embedding import AECafeEmbedding
    g = np.load(os.path.join(GOLD, name + ".npz"))
    counts, hs, D, B = g["counts"].tolist(), g["hash_sizes"].tolist(), int(g["dim"]), int(g["batch"])
    F, sep = len(counts), bool(g["separate"])
    hotn = g["hotn"].tolist()
    emb = AECafeEmbedding(counts, D, hotn if sep else hotn[0], hs, int(g["thr"]), float(g["decay"]), hot_separate_field=sep,
                          use_freq=bool(g["use_freq"]), offsets=g["offsets"])
    emb.set_tables([g[f"w_hot0_{k}"] for k in range(len(hotn))], [g[f"w_cold0_{f}"] for f in range(F)])
    for step in range(int(g["steps"])):
        ids = g[f"ids_{step}"]
        out = emb.forward(torch.from_numpy(ids).cuda())
        _close(out.cpu().numpy(), g[f"out_{step}"], f"step {step} forward")
        emb.backward_update(torch.from_numpy(g[f"gout_{step}"]).cuda(), float(g["lr"]))
        gids = torch.from_numpy((ids + g["offsets"][:, None]).astype(np.int32)).cuda()
        for f in range(F):
            que = emb.sketches[f if sep else 0].query(gids[f]).cpu().numpy()
            assert np.array_equal(que, g[f"que_after_{step}"][f]), f"step {step} field {f}: routing after the step"
        for k in range(len(hotn)):
            _close(emb.weight_high[k].cpu().numpy(), g[f"w_hot_{step}_{k}"], f"step {step} hot table {k}")
        for f in range(F):
            _close(emb.weight_hash[f].cpu().numpy(), g[f"w_cold_{step}_{f}"], f"step {step} cold table {f}")
    assert int(g["migrated"]) > 20


@pytest.mark.parametrize("use_freq", [False, True], ids=["grad_norm", "use_freq"])
def test_fields_sharing_one_sketch_against_the_oracle(use_freq, oracle_mod):
    from cafe_amd.ae_embedding import AECafeEmbedding, ae_sizes
    counts, D, B, thr, decay, lr = [3000, 41, 70000, 900], 16, 256, 3, 0.97, 0.1
    hotn, hs = ae_sizes(counts, D, 0.05, 0.3, False)
    assert hotn > 100
    emb = AECafeEmbedding(counts, D, hotn, hs, thr, decay, hot_separate_field=False, use_freq=use_freq, seed=3)
    F = len(counts)
    offs = emb.offsets
    ora = oracle_mod.AEOracle(hotn, thr, decay)
    w_hot = emb.weight_high[0].cpu().numpy().copy()
    w_cold = [w.cpu().numpy().copy() for w in emb.weight_hash]
    rng = np.random.default_rng(5)
    gen = torch.Generator(device="cuda").manual_seed(4)
    newly = 0
    for step in range(25):
        ids = np.stack([(rng.zipf(1.2, size=B) - 1) % c for c in counts]).astype(np.int64)
        ids[0] = np.maximum(ids[0], 1)
        gids = ids + offs[:, None]
        out = emb.forward(torch.from_numpy(ids).cuda()).cpu().numpy()
        que = np.stack([ora.query(gids[f]) for f in range(F)])
        assert np.array_equal(emb._saved[2].cpu().numpy(), que), f"step {step}: routing"
        want = np.stack([oracle_mod.ref_forward(que[f], w_hot, w_cold[f], hs[f])[0] for f in range(F)])
        _close(out, want, f"step {step} forward")
        grad = torch.randn(F, B, D, device="cuda", generator=gen)
        scores, ins = emb.backward_update(grad, lr)
        gn = grad.cpu().numpy()
        for f in range(F):
            sc = np.ones(B, np.float32) if use_freq else scores[f].cpu().numpy()
            if not use_freq:
                _close(sc, oracle_mod.ref_scores(gn[f]), f"step {step} field {f} scores")
            ins_want = ora.insert_val(gids[f], sc)
            assert np.array_equal(ins[f].cpu().numpy(), ins_want), f"step {step} field {f}: ins"
            qa = ora.query(gids[f])
            oracle_mod.ref_migrate(ins_want, qa, ids[f], w_hot, w_cold[f], hs[f])
            newly += int(((ins_want != 0) & (qa < 0)).sum())
        for f in range(F):
            oracle_mod.ref_sgd(que[f], gn[f], w_hot, w_cold[f], hs[f], lr)
    assert newly > 50, newly
    bad = oracle_mod.ae_states_equal(emb.sketches[0].state(), ora.state())
    assert not bad, bad
    _close(emb.weight_high[0].cpu().numpy(), w_hot, "hot table")
    for f in range(F):
        _close(emb.weight_hash[f].cpu().numpy(), w_cold[f], f"cold table {f}")
