"""The oracle's restatement of the earlier HotSketch generation (oracle/cafe_oracle_ae.c) against outputs recorded from
the reference's own ArtifactEvaluation/embeddings/sketch.cpp compiled in the build container (tests/golden/ae_*.npz,
tests/golden/make_ae_golden.py): ins[] and que[] of every call, bit-exact; decay() events and an exhausted free list
are part of the traces."""
import glob
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")
NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(G, "ae_*.npz")))


def replay(g, make):
    """Runs a recorded trace through sketches built by make(lim, thr, dr) -> object with insert_val / insert / query
    on numpy arrays; returns the list of mismatches and the sketches."""
    F = int(g["fields"])
    sk = [make(int(g["lim"]), int(g["thr"]), float(g["dr"])) for _ in range(F)]
    bad = []
    for c in range(int(g["calls"])):
        for f in range(F):
            ids = g[f"ids_{c}_{f}"]
            ins = sk[f].insert_val(ids, g[f"v_{c}_{f}"]) if int(g["val"]) else sk[f].insert(ids)
            if not np.array_equal(np.asarray(ins, np.int32), g[f"ins_{c}_{f}"].astype(np.int32)):
                bad.append(f"call {c} field {f}: ins")
            if not np.array_equal(np.asarray(sk[f].query(ids), np.int32), g[f"que_{c}_{f}"]):
                bad.append(f"call {c} field {f}: que")
    return bad, sk


@pytest.mark.parametrize("name", NAMES)
def test_ae_oracle_matches_reference_trace(name, oracle_mod):
    g = np.load(os.path.join(G, name + ".npz"))
    bad, sk = replay(g, oracle_mod.AEOracle)
    assert not bad, bad[:5]
    st = [s.state() for s in sk]
    if name == "ae_decay":
        assert st[0]["decays"] >= 3
    if name == "ae_dry_free_list":
        assert any(len(s["fifo"]) == 0 for s in st) or st[0]["decays"] > 0
    assert sum(int((s["dic"] != 0).sum()) for s in st) > 10


def test_ae_update_rebuilds_the_free_list(oracle_mod):
    """update() (:61-75), intended semantics: ascending slot numbers that no entry holds (the reference's own marker
    array is only cleared in its first 8 bytes, so its result depends on what the allocator returned)."""
    o = oracle_mod.AEOracle(50, 2, 0.9)
    rng = np.random.default_rng(0)
    for _ in range(10):
        o.insert((rng.zipf(1.3, size=200) % 300).astype(np.int64))
    st = o.state()
    o.update()
    used = set(int(d) for d in st["dic"] if d)
    assert o.state()["fifo"].tolist() == [i for i in range(1, 50) if i not in used]
