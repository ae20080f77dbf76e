"""GPU parity of the earlier HotSketch generation (include/cafe_hip.h section E, cafe_amd/csrc/cafe_ae.hip) against the
traces recorded from the reference's ArtifactEvaluation/embeddings/sketch.cpp (tests/golden/ae_*.npz) and against the
oracle's restatement on fresh streams: ins[], que[], every slot, the free list in pop order, num and the double `tot`,
bit-exact."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from test_oracle_ae_golden import NAMES, G, replay

pytestmark = pytest.mark.gpu


class _Np:
    def __init__(self, lim, thr, dr):
        from cafe_amd.ae_sketch import AESketch
        self.sk = AESketch(lim, thr, dr)

    def insert_val(self, ids, v):
        return self.sk.insert_val(ids.astype(np.int64), v).cpu().numpy()

    def insert(self, ids):
        return self.sk.insert(ids.astype(np.int64)).cpu().numpy()

    def query(self, ids):
        return self.sk.query(ids.astype(np.int64)).cpu().numpy()


@pytest.mark.parametrize("name", NAMES)
def test_reference_trace_on_gpu(name, oracle_mod):
    g = np.load(os.path.join(G, name + ".npz"))
    bad, sks = replay(g, _Np)
    assert not bad, bad[:5]
    # ... and the state the trace leaves equals the oracle's after the same trace
    _, ora = replay(g, oracle_mod.AEOracle)
    for a, b in zip(sks, ora):
        diff = oracle_mod.ae_states_equal(a.sk.state(), b.state())
        assert not diff, diff


@pytest.mark.parametrize("lim,thr,dr,batch,calls,space,a,val", [
    (300, 3, 0.5, 4096, 20, 4000, 1.15, True),         # decay every few calls, heavy ids (hundreds of items per bucket)
    (64, 2, 0.9, 2048, 25, 900, 1.05, False),          # the free list runs dry inside batches: in-order path
    (20000, 8, 0.99, 16384, 8, 2000000, 1.1, True),    # the artifact's largest call
    (5, 1, 0.5, 50, 30, 40, 1.5, True),                # a handful of buckets, Threshold 1
])
def test_fresh_streams_match_oracle(lim, thr, dr, batch, calls, space, a, val, oracle_mod):
    from cafe_amd.ae_sketch import AESketch
    ora = oracle_mod.AEOracle(lim, thr, dr)
    sk = AESketch(lim, thr, dr)
    rng = np.random.default_rng(lim + batch)
    for c in range(calls):
        ids = ((rng.zipf(a, size=batch) - 1) % space).astype(np.int64)
        v = (rng.random(batch, dtype=np.float32) * np.float32(3.0)).astype(np.float32) if val else None
        want = ora.insert_val(ids, v)
        got = sk.insert_val(ids, v).cpu().numpy()
        assert np.array_equal(got, want), f"call {c}: ins differs"
        assert np.array_equal(sk.query(ids).cpu().numpy(), ora.query(ids)), f"call {c}: que differs"
        if c % 5 == 4:
            diff = oracle_mod.ae_states_equal(sk.state(), ora.state())
            assert not diff, f"call {c}: {diff}"
    st = sk.state()
    assert not oracle_mod.ae_states_equal(st, ora.state())
    if lim == 300:
        assert st["decays"] >= 2
    if lim == 20000:
        assert st["serial_calls"] == 0               # the bucket owners did the work
    if lim == 64:
        assert st["serial_calls"] > 0                # the dry free list was really met


def test_empty_batch_update_and_state_roundtrip(oracle_mod):
    from cafe_amd.ae_sketch import AESketch
    ora = oracle_mod.AEOracle(50, 2, 0.9)
    sk = AESketch(50, 2, 0.9)
    rng = np.random.default_rng(0)
    assert sk.insert(np.zeros(0, np.int64)).numel() == 0
    for _ in range(10):
        ids = (rng.zipf(1.3, size=200) % 300).astype(np.int64)
        assert np.array_equal(sk.insert(ids).cpu().numpy(), ora.insert(ids))
    st = sk.state()
    sk2 = AESketch(50, 2, 0.9)
    sk2.load_state_dict(st)
    assert not oracle_mod.ae_states_equal(sk2.state(), st | {"decays": 0})
    sk.update(); ora.update()
    used = set(int(d) for d in st["dic"] if d)
    assert sk.state()["fifo"].tolist() == [i for i in range(1, 50) if i not in used] == ora.state()["fifo"].tolist()


_ABI_CHILD = r"""
import ctypes, sys, numpy as np
sys.path.insert(0, {root!r})
import cafe_amd
lib = ctypes.CDLL(cafe_amd.LIB_PATH)
# the artifact binds init / batch_query / batch_insert / batch_insert_val / update with a field index last
# (ArtifactEvaluation/embeddings/sketch.cpp:207-228); here under the cafe_ae_ref_ prefix (INTEGRATION.md)
u32p, f32p, i32p = ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
lib.cafe_ae_ref_init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]
lib.cafe_ae_ref_batch_query.argtypes = [u32p, ctypes.c_int, ctypes.c_int]; lib.cafe_ae_ref_batch_query.restype = i32p
lib.cafe_ae_ref_batch_insert.argtypes = [u32p, ctypes.c_int, ctypes.c_int]; lib.cafe_ae_ref_batch_insert.restype = i32p
lib.cafe_ae_ref_batch_insert_val.argtypes = [u32p, f32p, ctypes.c_int, ctypes.c_int]; lib.cafe_ae_ref_batch_insert_val.restype = i32p
g = np.load({golden!r})
F = int(g["fields"])
for f in range(F):
    lib.cafe_ae_ref_init(int(g["lim"]), int(g["thr"]), float(g["dr"]), f)
for c in range(int(g["calls"])):
    for f in range(F):
        ids = np.ascontiguousarray(g[f"ids_{{c}}_{{f}}"]); n = len(ids)
        v = np.ascontiguousarray(g[f"v_{{c}}_{{f}}"])
        r = lib.cafe_ae_ref_batch_insert_val(ids.ctypes.data_as(u32p), v.ctypes.data_as(f32p), n, f)
        assert np.array_equal(np.ctypeslib.as_array(r, shape=(n,)), g[f"ins_{{c}}_{{f}}"].astype(np.int32)), (c, f)
        q = lib.cafe_ae_ref_batch_query(ids.ctypes.data_as(u32p), n, f)
        assert np.array_equal(np.ctypeslib.as_array(q, shape=(n,)), g[f"que_{{c}}_{{f}}"]), (c, f)
lib.cafe_ae_ref_update(0)
print("AE_ABI_OK")
"""


def test_artifact_entry_points_on_host_pointers():
    src = _ABI_CHILD.format(root=ROOT, golden=os.path.join(G, "ae_three_fields.npz"))
    r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=300)
    assert "AE_ABI_OK" in r.stdout, r.stdout + r.stderr
