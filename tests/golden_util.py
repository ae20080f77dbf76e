"""Helpers to replay tests/golden/*.npz (traces recorded from the reference's own sketch.cpp)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KINDS = {0: "insert_val", 1: "insert", 2: "query"}


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def calls(g):
    off = g["offsets"]
    for k, kind in enumerate(g["kinds"]):
        a, b = int(off[k]), int(off[k + 1])
        yield k, KINDS[int(kind)], g["ids"][a:b].view(np.uint32), g["scores"][a:b], g["outs"][a:b]


def state_at(g, k):
    pre = f"st{k}_"
    st = {n[len(pre):]: g[n] for n in g if n.startswith(pre)}
    for s in ("num", "real_n", "clk", "flip", "batch_num", "n1", "n2", "lim"):
        st[s] = int(st[s])
    return st


def replay(g, sk, check_state=True, states_equal=None):
    """Feed a golden trace through `sk` (insert_val/insert/query/state methods); return list of mismatches."""
    bad = []
    dumps = set(int(x) for x in g["dump_calls"])
    for k, kind, ids, sc, out in calls(g):
        if kind == "insert_val":
            got = sk.insert_val(ids, sc)
        elif kind == "insert":
            got = sk.insert(ids)
        else:
            got = sk.query(ids)
        got = np.asarray(got)
        if not np.array_equal(got, out):
            i = int(np.nonzero(got != out)[0][0])
            bad.append(f"call {k} ({kind}): output differs first at {i}: got {got[i]} want {out[i]}")
            break
        if check_state and k in dumps:
            d = states_equal(sk.state(), state_at(g, k), where=f"call {k}: ")
            if d:
                bad.extend(d)
                break
    return bad
