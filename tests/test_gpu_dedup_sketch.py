"""Padding / de-duplication in front of the sketch (cafe_amd/dedup_sketch.py) against the oracle fed through a
restatement of the reference loop (TOIS_revision/PinSAGE/sketch.cpp:281-332, 439-441): ins[] of every call, que[] and
the complete sketch state bit-exact.  (Not comparable with that file's own sketch, whose insert rules differ: parity
unpinned against it, see the module's header.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Restated:
    def __init__(self, ora, padding):
        self.ora, self.padding, self.vis = ora, np.uint32(padding & 0xFFFFFFFF), {}

    def start(self):
        self.vis = {}

    def insert_val(self, ids, v):
        ids = np.asarray(ids).astype(np.int64) & 0xFFFFFFFF
        take = []
        for i, x in enumerate(ids):
            if x in self.vis or x == self.padding:
                continue
            if x not in [ids[j] for j in take]:
                take.append(i)
        take = np.asarray(take, np.int64)
        got = self.ora.insert_val(ids[take], None if v is None else np.asarray(v, np.float32)[take])
        for i, g in zip(take, got):
            self.vis[int(ids[i])] = int(g)
        return np.array([0 if x == self.padding else self.vis[int(x)] for x in ids], np.int32)


@pytest.mark.parametrize("with_scores", [True, False], ids=["insert_val", "insert"])
def test_dedup_and_padding_against_the_restated_loop(with_scores, oracle_mod):
    from cafe_amd.sketch import HotSketch
    from cafe_amd.dedup_sketch import DedupSketch
    lim, thr, pad = 400, 3, -1
    sk = HotSketch(lim, thr, 1, 1.0000005, 1)
    dd = DedupSketch(sk, pad)
    ora = oracle_mod.Oracle(lim, thr, 1, 1.0000005, 1)
    ref = _Restated(ora, pad)
    rng = np.random.default_rng(3)
    for step in range(30):
        dd.batch_insert_start(); ref.start()
        for call in range(3):                                    # e.g. one call per hop of a sampled neighbourhood
            n = int(rng.integers(50, 400))
            ids = (rng.zipf(1.2, size=n) % 3000 + 1).astype(np.int64)
            ids[rng.random(n) < 0.15] = pad                      # padded positions
            v = (rng.random(n).astype(np.float32) * 3) if with_scores else None
            got = dd.batch_insert_val(torch.from_numpy(ids), None if v is None else torch.from_numpy(v))
            want = ref.insert_val(ids, v)
            assert np.array_equal(got.cpu().numpy(), want), f"step {step} call {call}: ins"
            q = dd.batch_query(torch.from_numpy(ids)).cpu().numpy()
            qw = np.where((ids & 0xFFFFFFFF) == (pad & 0xFFFFFFFF), np.int64(pad), ora.query(ids & 0xFFFFFFFF))
            assert np.array_equal(q.astype(np.int64), qw), f"step {step} call {call}: que"
    bad = oracle_mod.states_equal(sk.state(), ora.state())
    assert not bad, bad[:3]
    assert ora.state()["num"] > 20


def test_a_call_whose_ids_are_all_known_still_counts_as_a_call(oracle_mod):
    from cafe_amd.sketch import HotSketch
    from cafe_amd.dedup_sketch import DedupSketch
    sk = HotSketch(200, 2, 1, 1.0000005, 1)
    dd = DedupSketch(sk, -1)
    ora = oracle_mod.Oracle(200, 2, 1, 1.0000005, 1)
    ref = _Restated(ora, -1)
    ids = np.array([5, 5, 9, -1, 9, 12], np.int64)
    for _ in range(4):                                           # calls 2..4 insert nothing new
        got = dd.batch_insert(torch.from_numpy(ids))
        assert np.array_equal(got.cpu().numpy(), ref.insert_val(ids, None))
    assert sk.state()["batch_num"] == ora.state()["batch_num"] == 4
    assert not oracle_mod.states_equal(sk.state(), ora.state())
