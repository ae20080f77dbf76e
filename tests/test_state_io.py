"""The PinSAGE-generation sketch's save_state byte layout (TOIS_revision/PinSAGE/sketch.cpp:333-402) read and written
by cafe_amd.state_io, against a file that sketch itself wrote (tests/golden/pinsage_state.bin, recorded by
tests/golden/make_pinsage_state.py from the reference's source compiled in the build container) and against what the
same sketch answered to batch_query right before saving."""
import os

import numpy as np
import pytest

from cafe_amd import state_io as sio

G = os.path.join(os.path.dirname(__file__), "golden")


def _load():
    meta = np.load(os.path.join(G, "pinsage_state.npz"))
    st = sio.read_pinsage_state(os.path.join(G, "pinsage_state.bin"), int(meta["lim"]))
    return meta, st


def test_layout_against_the_reference_file(tmp_path):
    meta, st = _load()
    lim = int(meta["lim"])
    h = st["head"]
    assert int(h["batch_num"]) == int(meta["calls"])                  # one per batch_insert_val (:316-318)
    assert 0 < int(h["num"]) < lim and int(h["real_n"]) > 0 and float(h["t"]) > 0
    # decay_importance = alpha ^ calls, divided by V = 10000 at every flip (:310-315)
    want = 1.0
    flips = 0
    for _ in range(int(meta["calls"])):
        want *= float(meta["alpha"])
        if want > 10000:
            want /= 10000; flips += 1
    assert abs(float(h["decay_importance"]) - want) <= 1e-12 * want and int(h["flip"]) == flips % 2
    # every slot 1..lim-1 is either handed out exactly once or waits in the free list
    used = np.concatenate([st["main"]["dic"].reshape(-1), st["lru"]["dic"].reshape(-1)])
    used = used[used != 0]
    assert len(np.unique(used)) == len(used)
    assert sorted(used.tolist() + st["fifo"].tolist()) == list(range(1, lim))
    assert int(h["num"]) >= len(used)
    # the reference's own answers right before the save
    assert np.array_equal(sio.pinsage_query(st, meta["probe"]), meta["que"])
    assert (meta["que"] < 0).sum() == len(used)
    # byte-exact round trip, also through the state dict of this package
    out = tmp_path / "again.bin"
    sio.write_pinsage_state(out, st)
    ref = open(os.path.join(G, "pinsage_state.bin"), "rb").read()
    assert open(out, "rb").read() == ref
    sd = sio.to_state_dict(st)
    assert sd["n1"] == int(lim * 0.9) and sd["n2"] == int(lim * 0.1) and len(sd["mval"]) == 4 * sd["n1"]
    sio.write_pinsage_state(out, sio.from_state_dict(sd, template=st))
    assert open(out, "rb").read() == ref


def test_truncated_or_foreign_files_are_rejected(tmp_path):
    meta, st = _load()
    ref = open(os.path.join(G, "pinsage_state.bin"), "rb").read()
    p = tmp_path / "short.bin"
    p.write_bytes(ref[:-2])
    with pytest.raises(ValueError):
        sio.read_pinsage_state(p, int(meta["lim"]))
    with pytest.raises(ValueError):
        sio.read_pinsage_state(os.path.join(G, "pinsage_state.bin"), int(meta["lim"]) + 40)


@pytest.mark.gpu
def test_state_file_loads_into_the_gpu_sketch():
    """The file's tables and free list through HotSketch.load_state_dict and back: every array and scalar survives, and
    the hot ids of the main table answer batch_query as in the reference (hot entries of the LRU table are a PinSAGE-
    generation feature, :28-33; tricks/sketch.cpp's query does not look there)."""
    from cafe_amd.sketch import HotSketch
    meta, st = _load()
    sd = sio.to_state_dict(st)
    sk = HotSketch(int(meta["lim"]), int(meta["thr"]), int(meta["adjust"]), float(meta["alpha"]), 1)
    sk.load_state_dict(sd)
    back = sk.state()
    for k in ("mval", "mcnt", "mdic", "mflip", "lval", "lcnt", "ldic", "lt", "lflip", "fifo"):
        assert np.array_equal(np.asarray(back[k]), np.asarray(sd[k])), k
    for k in ("num", "real_n", "clk", "flip", "batch_num"):
        assert back[k] == sd[k], k
    que = sk.query(meta["probe"].astype(np.int64)).cpu().numpy()
    main_hot = {int(v) & 0xFFFFFFFF for v, d in zip(sd["mval"], sd["mdic"]) if d}
    for x, q, want in zip(meta["probe"].tolist(), que.tolist(), meta["que"].tolist()):
        if x in main_hot:
            assert q == want
