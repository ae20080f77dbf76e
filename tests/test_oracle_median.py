"""The median tier (`batch_query_median`, bound by tricks/sketch_ml_embed.py:82-84) is a definition of this
repository -- no sketch.cpp of the reference exports it, so there is nothing to pin it against ("parity unpinned").
This file fixes the definition on a hand-checked case so that the oracle cannot drift silently."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import cafe_oracle as co  # noqa: E402


def test_median_known_answers():
    # init(100, 10, ...): Threshold 10, median bar 0.2 * 10 = 2.  Ids enter the LRU table with score 1 and move to
    # the main table only when the LRU score reaches the threshold or they are evicted with score >= 5
    # (sketch.cpp:179-195), so feed scores through insert_val.
    o = co.Oracle(100, 10, adjust=1, alpha=1.0)
    ids = np.array([7, 8, 9], dtype=np.uint32)
    o.insert_val(ids, np.ones(3, np.float32))                        # all three: LRU table, score 1
    assert list(o.query_median(ids)) == [7, 8, 9]                    # not in the main table -> not median
    o.insert_val(np.array([7], np.uint32), np.array([20.0], np.float32))    # 7: LRU hit, 21 >= 10 -> hot, promoted
    assert o.query([7])[0] < 0 and o.query_median([7])[0] == 7       # hot is never median


def test_median_follows_main_table_scores():
    rng = np.random.default_rng(5)
    o = co.Oracle(400, 50, adjust=1, alpha=1.02)                     # alpha > 1: flips happen, lazy decay pending
    for _ in range(200):
        ids = (rng.zipf(1.2, 256) % 3000 + 1).astype(np.uint32)
        o.insert_val(ids, rng.random(256).astype(np.float32) * 4)
    st = o.state()
    probe = np.arange(1, 3001, dtype=np.uint32)
    que, med = o.query(probe), o.query_median(probe)
    thr, gflip = float(st["thr"]), int(st["flip"])
    n1 = st["n1"]
    want = probe.astype(np.int32).copy()
    n_med = 0
    for i, x in enumerate(probe):
        if que[i] < 0:
            continue
        key = int(int(x) * 1000000007 % n1)
        for s in range(4):
            if st["mval"][key * 4 + s] == x and st["mdic"][key * 4 + s] == 0:
                c = np.float32(st["mcnt"][key * 4 + s])
                if int(st["mflip"][key]) != gflip:
                    c = np.float32(np.float64(c) / 10000.0)
                if float(c) >= 0.2 * thr:
                    want[i] = -int(x)
                    n_med += 1
                break
    assert np.array_equal(med, want)
    assert n_med > 0 and (que < 0).sum() > 0
