"""GPU parity of the HotSketch kernels against the reference's own traces (tests/golden) and against the
CPU oracle on fresh seeded streams.  Bit-exact: ins[], que[], every slot, every scalar, free-list order."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import golden_names, ROOT
import golden_util as gu

pytestmark = pytest.mark.gpu


class _Np:
    """Adapter: HotSketch with numpy in/out, the interface golden_util.replay expects."""

    def __init__(self, sk):
        self.sk = sk

    def insert_val(self, ids, v):
        return self.sk.insert_val(ids, v).cpu().numpy()

    def insert(self, ids):
        return self.sk.insert(ids).cpu().numpy()

    def query(self, ids):
        return self.sk.query(ids).cpu().numpy()

    def state(self):
        return self.sk.state()


def _make(g, serial):
    from cafe_amd.sketch import HotSketch
    sk = HotSketch(int(g["lim"]), int(g["thr"]), int(g["adjust"]), float(g["alpha"]), int(g["lru_first"]))
    sk.set_option("serial_insert", 1 if serial else 0)
    return sk


@pytest.mark.parametrize("serial", [1, 0], ids=["inorder", "parallel"])
@pytest.mark.parametrize("name", golden_names())
def test_reference_trace_on_gpu(name, serial, oracle_mod):
    g = gu.load(name)
    sk = _make(g, serial)
    bad = gu.replay(g, _Np(sk), states_equal=oracle_mod.states_equal)
    assert not bad, "\n".join(bad)


def _stream(seed, n, space, a=1.08):
    rng = np.random.default_rng(seed)
    ids = ((rng.zipf(a, size=n) - 1) % space + 1).astype(np.int64)
    v = (rng.random(n, dtype=np.float32) * np.float32(2.0)).astype(np.float32)
    return ids, v


@pytest.mark.parametrize("lim,thr,batch,calls,space", [
    (2000, 4, 4096, 12, 50000),          # many ids per bucket, resets
    (13505, 3, 8192, 8, 33762577),       # cfg2-sized sketch, one field's batch
    (50000, 6, 65536, 4, 5000000),       # batches far beyond the reference's 16384 buffer
])
def test_fresh_streams_match_oracle(lim, thr, batch, calls, space, oracle_mod):
    from cafe_amd.sketch import HotSketch
    ora = oracle_mod.Oracle(lim, thr, 1, 1.0000005, 1)
    sk = HotSketch(lim, thr, 1, 1.0000005, 1)
    for c in range(calls):
        ids, v = _stream(1000 + c, batch, space)
        want = ora.insert_val(ids, v)
        got = sk.insert_val(ids, v).cpu().numpy()
        assert np.array_equal(got, want), f"call {c}: ins differs"
        assert np.array_equal(sk.query(ids).cpu().numpy(), ora.query(ids)), f"call {c}: que differs"
    assert not oracle_mod.states_equal(sk.state(), ora.state())


@pytest.mark.parametrize("bypass", [1, 0], ids=["hot_bypass", "owners_only"])
@pytest.mark.parametrize("lim,thr,batch,calls,space,a", [
    (300, 3, 2048, 30, 1500, 1.1),        # tiny table: every bucket mixes hot ids, cold ids and incoming moves
    (1200, 6, 8192, 12, 20000, 1.3),      # heavy ids (hundreds of items each) next to LRU churn
    (4000, 2, 4096, 16, 6000, 1.01),      # low threshold: promotions into buckets that hold hot slots, resets
])
def test_hot_ids_and_moves_share_buckets(lim, thr, batch, calls, space, a, bypass, oracle_mod):
    """Ids that are hot before a batch bypass the owner rounds (ordered sums only); a move that lands in their bucket
    must still see their score as of its position in the batch (sketch.cpp:244-258).  Half the scores are large so that
    evicted LRU entries qualify for a move (score >= 5, sketch.cpp:193-195)."""
    from cafe_amd.sketch import HotSketch
    ora = oracle_mod.Oracle(lim, thr, 1, 1.0000005, 1)
    sk = HotSketch(lim, thr, 1, 1.0000005, 1)
    sk.set_option("hot_bypass", bypass)
    rng = np.random.default_rng(77)
    for c in range(calls):
        ids = np.where(rng.random(batch) < 0.6, (rng.zipf(a, size=batch) - 1) % space + 1,
                       rng.integers(1, space + 1, size=batch)).astype(np.int64)
        v = (rng.random(batch, dtype=np.float32) * np.float32(2.0)).astype(np.float32)
        v[rng.random(batch) < 0.3] *= np.float32(6.0)
        want = ora.insert_val(ids, v)
        got = sk.insert_val(ids, v).cpu().numpy()
        assert np.array_equal(got, want), f"call {c}: ins differs"
        bad = oracle_mod.states_equal(sk.state(), ora.state())
        assert not bad, f"call {c}: " + "; ".join(bad[:3])
    st = ora.stats()
    assert st["evict_move"] + st["promote"] > 50 and st["main_hit"] > batch      # moves and hot hits really happened
    assert sk.counters()["inorder_calls"] <= calls // 3


def test_empty_and_tiny_batches(oracle_mod):
    from cafe_amd.sketch import HotSketch
    ora = oracle_mod.Oracle(100, 2, 1, 1.0, 1)
    sk = HotSketch(100, 2, 1, 1.0, 1)
    for ids in ([], [5], [5, 5], [0], [7, 0, 7]):
        v = np.ones(len(ids), np.float32)
        want = ora.insert_val(np.array(ids, np.int64), v)
        got = sk.insert_val(np.array(ids, np.int64), v).cpu().numpy()
        assert np.array_equal(got, want)
    assert not oracle_mod.states_equal(sk.state(), ora.state())


def test_state_roundtrip_and_file(tmp_path, oracle_mod):
    from cafe_amd.sketch import HotSketch
    g = gu.load("thr1_quirk")
    sk = _make(g, 1)
    assert not gu.replay(g, _Np(sk), check_state=False)
    st = sk.state()
    sk2 = _make(g, 1)
    sk2.load_state_dict(st)
    assert not oracle_mod.states_equal(sk2.state(), st)
    p = tmp_path / "sk.bin"
    sk.save(p)
    sk3 = _make(g, 1)
    sk3.load(p)
    assert not oracle_mod.states_equal(sk3.state(), st)
    # update(): free list rebuilt as the ascending unused slots
    sk3.rebuild_free_list()
    used = set(int(x) for x in np.concatenate([st["mdic"], st["ldic"]]) if x)
    want = np.array([i for i in range(1, int(g["lim"])) if i not in used], np.uint32)
    assert np.array_equal(sk3.state()["fifo"], want)


_ABI_CHILD = r"""
import ctypes, sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import golden_util as gu
from cafe_amd import _lib
L = _lib.load()
g = gu.load("known_answer")
i32p = ctypes.POINTER(ctypes.c_int); f32p = ctypes.POINTER(ctypes.c_float)
# bind exactly as tricks/sk_embedding_bag.py:61-73 does
L.batch_query.argtypes = [i32p, ctypes.c_int]; L.batch_query.restype = i32p
L.batch_insert.argtypes = [i32p, ctypes.c_int]; L.batch_insert.restype = i32p
L.batch_insert_val.argtypes = [i32p, f32p, ctypes.c_int]; L.batch_insert_val.restype = i32p
L.init(int(g["lim"]), int(g["thr"]), int(g["adjust"]), float(g["alpha"]))
for k, kind, ids, sc, out in gu.calls(g):
    a = np.ascontiguousarray(ids.view(np.int32)); p = a.ctypes.data_as(i32p)
    if kind == "insert_val":
        s = np.ascontiguousarray(sc); r = L.batch_insert_val(p, s.ctypes.data_as(f32p), len(a))
    elif kind == "insert":
        r = L.batch_insert(p, len(a))
    else:
        r = L.batch_query(p, len(a))
    got = np.ctypeslib.as_array(r, shape=(len(a),))
    assert np.array_equal(got, out), (k, kind, got, out)
u = np.array([7, 8, 9, 1007, 4242], np.uint32)
assert abs(L.analyse(u.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 5) - 0.8) < 1e-12
assert abs(L.analyse1(u.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 5) - 0.8) < 1e-12
L.update()
print("ABI_OK")
"""


def test_reference_abi_drop_in():
    """The g++-built sklib.so's entry points (sketch.cpp:300-327), bound the way the reference binds them."""
    r = subprocess.run([sys.executable, "-c", _ABI_CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=300)
    assert "ABI_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("option,value,reason", [("max_rounds", 1, "rounds_not_settled"), ("target_items", 16, None),
                                                 ("move_cap", 1, "workspace_limit")])
def test_batches_that_leave_the_parallel_path_stay_exact(option, value, reason, oracle_mod):
    """Too few rounds enqueued to settle a batch (max_rounds = 1), or an unusual partition size: the batch goes to the
    in-order kernel untouched (or runs with other partitions) -- never a different result -- and the reason is
    counted (cafe_sketch_fallback_reasons)."""
    from cafe_amd.sketch import HotSketch
    lim, thr, batch, space = 1200, 4, 4096, 20000
    ora = oracle_mod.Oracle(lim, thr, 1, 1.0000005, 1)
    sk = HotSketch(lim, thr, 1, 1.0000005, 1)
    sk.set_option(option, value)
    rng = np.random.default_rng(5)
    for c in range(14):
        ids = ((rng.zipf(1.2, size=batch) - 1) % space + 1).astype(np.int64)
        v = (rng.random(batch, dtype=np.float32) * np.float32(8.0)).astype(np.float32)
        want = ora.insert_val(ids, v)
        got = sk.insert_val(ids, v).cpu().numpy()
        assert np.array_equal(got, want), f"call {c}: ins differs"
    bad = oracle_mod.states_equal(sk.state(), ora.state())
    assert not bad, bad[:3]
    c = sk.counters()
    assert c["cadence_miss"] == 0
    if reason:
        assert c["inorder_calls"] > 0 and c["left_parallel_path"][reason] > 0, c
        assert c["inorder_calls"] == sum(c["left_parallel_path"].values()), c


def _ranks_in_buckets(lt):
    """Per LRU bucket: 0 for a never-used slot, else 1 + number of smaller non-zero timestamps (k_clock_renorm)."""
    t = np.asarray(lt, np.int64).reshape(-1, 4)
    nz = t != 0
    below = ((t[:, None, :] < t[:, :, None]) & nz[:, None, :]).sum(axis=2)
    return np.where(nz, below + 1, 0).astype(np.int32).reshape(-1)


@pytest.mark.parametrize("clock", [(1 << 29) - 3000, 1_000_000_000 - 2000, (1 << 31) - 9000],
                         ids=["renorm_bound", "reference_sentinel_1e9", "int_wrap"])
def test_lru_clock_lifetime(clock, oracle_mod):
    """the LRU clock only grows (one tick per main-table miss).  The reference stops finding victims at 1e9
    ticks (sketch.cpp:159) and wraps at 2^31.  The library rewrites the timestamps of every bucket to their ranks before
    the clock passes 2^29 (cafe_hip.h, cafe_sketch_clock_renorms): decisions must stay those of a reference whose clock
    is far from its sentinel.  Start both sides from the same warmed-up state, the library's copy with all timestamps
    shifted up to `clock`, and compare ins[], que[] and the whole state (timestamps by their order inside the bucket)."""
    from cafe_amd.sketch import HotSketch
    lim, thr, batch, space = 3000, 3, 4096, 40000
    ora = oracle_mod.Oracle(lim, thr, 1, 1.0000005, 1)
    rng = np.random.default_rng(11)

    def stream():
        ids = ((rng.zipf(1.15, size=batch) - 1) % space + 1).astype(np.int64)
        v = (rng.random(batch, dtype=np.float32) * np.float32(6.0)).astype(np.float32)
        return ids, v

    for _ in range(4):
        ora.insert_val(*stream())
    st = ora.state()
    shifted = dict(st)
    shift = clock - int(st["clk"])
    shifted["lt"] = np.where(st["lt"] != 0, st["lt"].astype(np.int64) + shift, 0).astype(np.int32)
    shifted["clk"] = clock
    sk = HotSketch(lim, thr, 1, 1.0000005, 1)
    sk.load_state_dict(shifted)
    for c in range(6):
        ids, v = stream()
        want = ora.insert_val(ids, v)
        got = sk.insert_val(ids, v).cpu().numpy()
        assert np.array_equal(got, want), f"call {c}: ins differs"
        assert np.array_equal(sk.query(ids).cpu().numpy(), ora.query(ids)), f"call {c}: que differs"
    a, b = sk.state(), ora.state()
    assert np.array_equal(_ranks_in_buckets(a["lt"]), _ranks_in_buckets(b["lt"]))
    a["lt"] = b["lt"]; a["clk"] = b["clk"]
    assert not oracle_mod.states_equal(a, b)
    c = sk.counters()                       # also raises if the commit kernel refused a staged entry (bad_commit)
    assert c["clock_renorms"] >= 1 and c["inorder_calls"] == 0 and c["cadence_miss"] == 0, c
    assert 0 < sk.state()["clk"] < (1 << 29)


def test_import_rejects_out_of_range_state():
    from cafe_amd.sketch import HotSketch
    sk = HotSketch(500, 2, 1, 1.0, 1)
    sk.insert_val(np.arange(1, 2000, dtype=np.int64) % 700 + 1, np.full(1999, 3.0, np.float32))
    good = sk.state()
    for key, idx, val in (("mdic", 0, 500), ("ldic", 1, -3), ("fifo", 0, 0), ("fifo", 0, 500), ("lt", 2, -1)):
        bad = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in good.items()}
        if len(bad[key]) <= idx:
            continue
        bad[key][idx] = val
        with pytest.raises(RuntimeError):
            sk.load_state_dict(bad)
    sk.load_state_dict(good)
