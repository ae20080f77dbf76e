"""GPU parity of the whole hot path (forward gather, scores, sketch insert, migration, backward+SGD)
against the CPU oracle: sketch decisions bit-exact, gathered rows bit-exact (pure copies), scores and
updated tables within 1e-5 relative (float32 sums in a different order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-5      # BASELINE.json north_star: embeddings within 1e-5 relative fp32 tolerance


def _close(a, b, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= RTOL, f"{what}: max rel err {err:.3e}"


class _CpuPipeline:
    """The reference's per-step order on the oracle + numpy tensor restatement."""

    def __init__(self, om, coll, thr, adjust, alpha):
        self.om = om
        self.sk = om.Oracle(coll.hotn, thr, adjust, alpha, 1)
        self.w_hot = coll.weight_high.cpu().numpy().copy()
        self.w_cold = [w.cpu().numpy().copy() for w in coll.weight_hash]
        self.offsets = coll.offsets
        self.hash_sizes = coll.hash_sizes

    def forward(self, ids):
        F, B = ids.shape
        self.gids = [(ids[f] + self.offsets[f]).astype(np.int64) for f in range(F)]
        self.que = [self.sk.query(self.gids[f]) for f in range(F)]
        return np.stack([self.om.ref_forward(self.que[f], self.w_hot, self.w_cold[f], self.hash_sizes[f])[0] for f in range(F)])

    def backward_update(self, ids, grad, scores, lr, per_field):
        F, B = ids.shape
        ins_all = []
        if per_field:
            for f in range(F):
                ins = self.sk.insert_val(self.gids[f], scores[f])
                qa = self.sk.query(self.gids[f])
                self.om.ref_migrate(ins, qa, ids[f], self.w_hot, self.w_cold[f], self.hash_sizes[f])
                ins_all.append(ins)
        else:
            ins = self.sk.insert_val(np.concatenate(self.gids), scores.reshape(-1)).reshape(F, B)
            for f in range(F):
                qa = self.sk.query(self.gids[f])
                self.om.ref_migrate(ins[f], qa, ids[f], self.w_hot, self.w_cold[f], self.hash_sizes[f])
                ins_all.append(ins[f])
        # SGD on both tables with the forward's routing; hot table shared by all fields
        acc_h = np.zeros(self.w_hot.shape, np.float64)
        for f in range(F):
            que = self.que[f].astype(np.int64)
            hot = que < 0
            np.add.at(acc_h, (-que[hot]) % self.w_hot.shape[0], grad[f][hot].astype(np.float64))
            acc_c = np.zeros(self.w_cold[f].shape, np.float64)
            np.add.at(acc_c, que[~hot] % self.hash_sizes[f], grad[f][~hot].astype(np.float64))
            self.w_cold[f] -= (lr * acc_c).astype(np.float32)
        self.w_hot -= (lr * acc_h).astype(np.float32)
        return np.stack(ins_all)


@pytest.mark.parametrize("dim,per_field", [(16, True), (128, True), (16, False), (128, False), (40, False)])
def test_collection_steps_match_oracle(dim, per_field, oracle_mod):
    from cafe_amd.collection import CafeEmbeddingCollection
    counts = [3000, 40, 50000, 7, 900]
    hash_sizes = [37, 1, 211, 2, 16]
    hotn, thr, B, lr = 400, 3, 512, 0.1
    coll = CafeEmbeddingCollection(counts, dim, hotn, hash_sizes, threshold=thr, adjust_thres=1, alpha=1.0000005, seed=3)
    cpu = _CpuPipeline(oracle_mod, coll, thr, 1, 1.0000005)
    rng = np.random.default_rng(7)
    F = len(counts)
    newly = 0
    for step in range(25):
        ids = np.stack([(rng.zipf(1.2, size=B) - 1) % c for c in counts]).astype(np.int64)
        ids[0] = np.maximum(ids[0], 1)
        d_ids = torch.from_numpy(ids).cuda()
        out = coll.forward(d_ids).cpu().numpy()
        want = cpu.forward(ids)
        assert np.array_equal(coll._buf["que"].cpu().numpy().reshape(F, B), np.stack(cpu.que)), f"step {step}: routing"
        _close(out, want, f"step {step} forward")
        # the gather itself is a pure copy: bit-exact against the GPU's own tables under the oracle's routing
        gh = coll.weight_high.cpu().numpy()
        for f in range(F):
            mine, _ = oracle_mod.ref_forward(cpu.que[f], gh, coll.weight_hash[f].cpu().numpy(), hash_sizes[f])
            assert np.array_equal(out[f], mine), f"step {step} field {f}: gathered rows are not exact copies"
        grad = torch.randn(F, B, dim, device="cuda", generator=torch.Generator(device="cuda").manual_seed(step))
        scores, ins = coll.backward_update(grad, lr, per_field_calls=per_field)
        g = grad.cpu().numpy()
        sc = scores.cpu().numpy().reshape(F, B)
        for f in range(F):
            _close(sc[f], oracle_mod.ref_scores(g[f]), f"step {step} field {f} scores")
        # the oracle sketch is fed the scores the GPU computed, so decisions are compared on identical inputs
        ins_want = cpu.backward_update(ids, g, sc, lr, per_field)
        assert np.array_equal(ins.cpu().numpy().reshape(F, B), ins_want), f"step {step}: ins differs"
        newly += int(ins_want.sum())
        _close(coll.weight_high.cpu().numpy(), cpu.w_hot, f"step {step} hot table")
        for f in range(F):
            _close(coll.weight_hash[f].cpu().numpy(), cpu.w_cold[f], f"step {step} cold table {f}")
    assert newly > 50                                   # migrations really happened
    assert not oracle_mod.states_equal(coll.sketch.state(), cpu.sk.state())
