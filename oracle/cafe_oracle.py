"""ctypes front end of the CPU oracle (oracle/cafe_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (cafe_amd/) never does.  It also holds a numpy restatement of the tensor half of
the reference's embedding module (tricks/sk_embedding_bag.py), used to check the HIP gather /
backward / migration kernels.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i32p = ctypes.POINTER(ctypes.c_int32)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile liboracle.so (and, where /root/reference exists, oracle/_ref) with the committed Makefile."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, "cafe_oracle.c"), os.path.join(_HERE, "cafe_oracle_ae.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/tricks/sketch.cpp"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.cafe_oracle_create.restype = ctypes.c_void_p
        L.cafe_oracle_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]
        L.cafe_oracle_destroy.argtypes = [ctypes.c_void_p]
        L.cafe_oracle_batch_insert_val.argtypes = [ctypes.c_void_p, _u32p, _f32p, ctypes.c_int, _i32p]
        L.cafe_oracle_batch_query.argtypes = [ctypes.c_void_p, _u32p, ctypes.c_int, _i32p]
        L.cafe_oracle_batch_query_median.argtypes = [ctypes.c_void_p, _u32p, ctypes.c_int, _i32p]
        L.cafe_oracle_analyse.restype = ctypes.c_double
        L.cafe_oracle_analyse.argtypes = [ctypes.c_void_p, _u32p, ctypes.c_int]
        L.cafe_oracle_analyse1.restype = ctypes.c_double
        L.cafe_oracle_analyse1.argtypes = [ctypes.c_void_p, _u32p, ctypes.c_int]
        L.cafe_oracle_dims.argtypes = [ctypes.c_void_p, _i32p]
        L.cafe_oracle_scalars.argtypes = [ctypes.c_void_p, _f32p, _i32p, ctypes.POINTER(ctypes.c_double)]
        L.cafe_oracle_dump.argtypes = [ctypes.c_void_p, _i32p, _f32p, _i32p, _u8p, _i32p, _f32p, _i32p, _i32p, _u8p, _u32p]
        L.cafe_oracle_load.argtypes = [ctypes.c_void_p, _i32p, _f32p, _i32p, _u8p, _i32p, _f32p, _i32p, _i32p, _u8p, _u32p,
                                       ctypes.c_long, ctypes.c_float, _i32p, ctypes.c_double]
        _u32 = _u32p
        L.cafe_oracle_ae_create.restype = ctypes.c_void_p
        L.cafe_oracle_ae_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double]
        L.cafe_oracle_ae_destroy.argtypes = [ctypes.c_void_p]
        L.cafe_oracle_ae_batch_query.argtypes = [ctypes.c_void_p, _u32, ctypes.c_int, _i32p]
        L.cafe_oracle_ae_batch_insert_val.argtypes = [ctypes.c_void_p, _u32, _f32p, ctypes.c_int, _i32p]
        L.cafe_oracle_ae_update.argtypes = [ctypes.c_void_p]
        L.cafe_oracle_ae_dims.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
        L.cafe_oracle_ae_tot.restype = ctypes.c_double
        L.cafe_oracle_ae_tot.argtypes = [ctypes.c_void_p]
        L.cafe_oracle_ae_dump.argtypes = [ctypes.c_void_p, _u32, _f32p, _i32p, _u32]
        _LIB = L
    return _LIB


def _ids(a):
    return np.ascontiguousarray(np.asarray(a).astype(np.int64) & 0xFFFFFFFF, dtype=np.uint32)


class Oracle:
    """One HotSketch instance; a fresh instance equals a fresh reference process after lib.init()."""

    def __init__(self, lim, threshold, adjust=1, alpha=1.0, lru_first=1):
        self.L = _lib()
        self.h = self.L.cafe_oracle_create(int(lim), int(threshold), int(adjust), float(alpha), int(lru_first))
        self.lim = int(lim)

    def __del__(self):
        try:
            self.L.cafe_oracle_destroy(self.h)
        except Exception:
            pass

    def insert_val(self, ids, v=None):
        ids = _ids(ids)
        n = len(ids)
        ins = np.zeros(n, dtype=np.int32)
        vp = None
        if v is not None:
            v = np.ascontiguousarray(v, dtype=np.float32)
            assert len(v) == n
            vp = v.ctypes.data_as(_f32p)
        self.L.cafe_oracle_batch_insert_val(self.h, ids.ctypes.data_as(_u32p), vp, n, ins.ctypes.data_as(_i32p))
        return ins

    def insert(self, ids):
        return self.insert_val(ids, None)

    def query(self, ids):
        ids = _ids(ids)
        que = np.zeros(len(ids), dtype=np.int32)
        self.L.cafe_oracle_batch_query(self.h, ids.ctypes.data_as(_u32p), len(ids), que.ctypes.data_as(_i32p))
        return que

    def query_median(self, ids):
        """Median tier of sketch_ml_embed.py (a definition, see cafe_oracle.c): -id = medium importance."""
        ids = _ids(ids)
        out = np.zeros(len(ids), dtype=np.int32)
        self.L.cafe_oracle_batch_query_median(self.h, ids.ctypes.data_as(_u32p), len(ids), out.ctypes.data_as(_i32p))
        return out

    def analyse(self, ids):
        ids = _ids(ids)
        return self.L.cafe_oracle_analyse(self.h, ids.ctypes.data_as(_u32p), len(ids))

    def analyse1(self, ids):
        ids = _ids(ids)
        return self.L.cafe_oracle_analyse1(self.h, ids.ctypes.data_as(_u32p), len(ids))

    def state(self):
        return read_state(self.L, "cafe_oracle_", self.h)

    def load_state(self, st):
        """Continue from a state dict as state() returns it (same lim): tests that start from a prepared state."""
        a = {k: np.ascontiguousarray(st[k]) for k in ("mval", "mcnt", "mdic", "lval", "lcnt", "ldic", "lt")}
        mflip = np.ascontiguousarray(st["mflip"], dtype=np.uint8)
        lflip = np.ascontiguousarray(st["lflip"], dtype=np.uint8)
        fifo = np.ascontiguousarray(st["fifo"], dtype=np.uint32)
        ints = np.array([st["num"], st["real_n"], st["clk"], st["flip"], st["batch_num"], st.get("resets", 0)], np.int32)
        self.L.cafe_oracle_load(
            self.h, a["mval"].ctypes.data_as(_i32p), a["mcnt"].ctypes.data_as(_f32p), a["mdic"].ctypes.data_as(_i32p),
            mflip.ctypes.data_as(_u8p), a["lval"].ctypes.data_as(_i32p), a["lcnt"].ctypes.data_as(_f32p),
            a["ldic"].ctypes.data_as(_i32p), a["lt"].ctypes.data_as(_i32p), lflip.ctypes.data_as(_u8p),
            fifo.ctypes.data_as(_u32p), len(fifo), ctypes.c_float(float(st["thr"])), ints.ctypes.data_as(_i32p),
            float(st["decay"]))

    def stats(self):
        """Coverage counters: main hits, LRU hits, LRU misses, evictions moved to main, promotions,
        failed moves, slots pushed back, threshold crossings that found the free list empty."""
        out = (ctypes.c_long * 8)()
        self.L.cafe_oracle_stats(ctypes.c_void_p(self.h), out)
        names = ("main_hit", "lru_hit", "lru_miss", "evict_move", "promote", "move_fail", "push_back", "pop_empty")
        return dict(zip(names, (int(x) for x in out)))


class AEOracle:
    """The earlier HotSketch generation (ArtifactEvaluation/embeddings/sketch.cpp, class SS): one instance = one field."""

    def __init__(self, lim, threshold, decay_rate):
        self.L = _lib()
        self.h = self.L.cafe_oracle_ae_create(int(threshold), int(lim), float(decay_rate))

    def __del__(self):
        try:
            self.L.cafe_oracle_ae_destroy(self.h)
        except Exception:
            pass

    def insert_val(self, ids, v=None):
        ids = _ids(ids)
        ins = np.zeros(max(len(ids), 1), np.int32)
        vp = None
        if v is not None:
            v = np.ascontiguousarray(v, np.float32)
            vp = v.ctypes.data_as(_f32p)
        self.L.cafe_oracle_ae_batch_insert_val(self.h, ids.ctypes.data_as(_u32p), vp, len(ids), ins.ctypes.data_as(_i32p))
        return ins[:len(ids)]

    def insert(self, ids):
        return self.insert_val(ids, None)

    def query(self, ids):
        ids = _ids(ids)
        que = np.zeros(max(len(ids), 1), np.int32)
        self.L.cafe_oracle_ae_batch_query(self.h, ids.ctypes.data_as(_u32p), len(ids), que.ctypes.data_as(_i32p))
        return que[:len(ids)]

    def update(self):
        self.L.cafe_oracle_ae_update(self.h)

    def state(self):
        d = (ctypes.c_int64 * 4)()
        self.L.cafe_oracle_ae_dims(self.h, d)
        s, flen, num, decays = (int(x) for x in d)
        st = {"val": np.zeros(s * 4, np.uint32), "cnt": np.zeros(s * 4, np.float32), "dic": np.zeros(s * 4, np.int32),
              "fifo": np.zeros(max(flen, 1), np.uint32)}
        self.L.cafe_oracle_ae_dump(self.h, st["val"].ctypes.data_as(_u32p), st["cnt"].ctypes.data_as(_f32p),
                                   st["dic"].ctypes.data_as(_i32p), st["fifo"].ctypes.data_as(_u32p))
        st["fifo"] = st["fifo"][:flen]
        st["num"], st["decays"], st["tot"] = num, decays, float(self.L.cafe_oracle_ae_tot(self.h))
        return st


def ae_states_equal(a, b):
    """Bit-exact comparison of two AE-generation state dumps; list of differences."""
    bad = []
    for k in ("val", "dic", "fifo"):
        if not np.array_equal(np.asarray(a[k]), np.asarray(b[k])):
            bad.append(k)
    if not np.array_equal(np.asarray(a["cnt"], np.float32).view(np.uint32), np.asarray(b["cnt"], np.float32).view(np.uint32)):
        bad.append("cnt")
    for k in ("num", "decays"):
        if int(a[k]) != int(b[k]):
            bad.append(f"{k}: {a[k]} vs {b[k]}")
    if np.float64(a["tot"]).view(np.uint64) != np.float64(b["tot"]).view(np.uint64):
        bad.append(f"tot: {a['tot']!r} vs {b['tot']!r}")
    return bad


def read_state(L, prefix, h=None):
    """Dump every slot + scalars through <prefix>dims/scalars/dump (oracle or reference probe)."""
    args = (h,) if h is not None else ()
    dims = np.zeros(4, dtype=np.int32)
    getattr(L, prefix + "dims")(*args, dims.ctypes.data_as(_i32p))
    n1, n2, lim, flen = (int(x) for x in dims)
    thr = ctypes.c_float()
    ints = np.zeros(6, dtype=np.int32)
    decay = ctypes.c_double()
    getattr(L, prefix + "scalars")(*args, ctypes.byref(thr), ints.ctypes.data_as(_i32p), ctypes.byref(decay))
    st = {
        "mval": np.zeros(n1 * 4, np.int32), "mcnt": np.zeros(n1 * 4, np.float32), "mdic": np.zeros(n1 * 4, np.int32),
        "mflip": np.zeros(max(n1, 1), np.uint8),
        "lval": np.zeros(n2 * 4, np.int32), "lcnt": np.zeros(n2 * 4, np.float32), "ldic": np.zeros(n2 * 4, np.int32),
        "lt": np.zeros(n2 * 4, np.int32), "lflip": np.zeros(max(n2, 1), np.uint8),
        "fifo": np.zeros(max(flen, 1), np.uint32),
    }
    getattr(L, prefix + "dump")(
        *args,
        st["mval"].ctypes.data_as(_i32p), st["mcnt"].ctypes.data_as(_f32p), st["mdic"].ctypes.data_as(_i32p),
        st["mflip"].ctypes.data_as(_u8p), st["lval"].ctypes.data_as(_i32p), st["lcnt"].ctypes.data_as(_f32p),
        st["ldic"].ctypes.data_as(_i32p), st["lt"].ctypes.data_as(_i32p), st["lflip"].ctypes.data_as(_u8p),
        st["fifo"].ctypes.data_as(_u32p))
    st["mflip"] = st["mflip"][:n1]
    st["lflip"] = st["lflip"][:n2]
    st["fifo"] = st["fifo"][:flen]
    st["thr"] = np.float32(thr.value)
    st["num"], st["real_n"], st["clk"], st["flip"], st["batch_num"] = (int(x) for x in ints[:5])
    st["resets"] = int(ints[5])
    st["decay"] = float(decay.value)
    st["n1"], st["n2"], st["lim"] = n1, n2, lim
    return st


STATE_ARRAYS = ("mval", "mcnt", "mdic", "mflip", "lval", "lcnt", "ldic", "lt", "lflip", "fifo")
STATE_SCALARS = ("thr", "num", "real_n", "clk", "flip", "batch_num")


def states_equal(a, b, where=""):
    """Bit-exact comparison of two state dumps; returns a list of differences (empty = equal)."""
    bad = []
    for k in STATE_ARRAYS:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if x.shape != y.shape:
            bad.append(f"{where}{k}: shape {x.shape} vs {y.shape}")
        elif x.dtype.kind == "f":
            if not np.array_equal(x.view(np.uint32), y.view(np.uint32)):
                i = int(np.nonzero(x.view(np.uint32) != y.view(np.uint32))[0][0])
                bad.append(f"{where}{k}: first diff at {i}: {x[i]!r} vs {y[i]!r}")
        elif not np.array_equal(x, y):
            i = int(np.nonzero(x != y)[0][0])
            bad.append(f"{where}{k}: first diff at {i}: {x[i]!r} vs {y[i]!r}")
    for k in STATE_SCALARS:
        x, y = a[k], b[k]
        if k == "thr":
            x, y = np.float32(x).view(np.uint32), np.float32(y).view(np.uint32)
        if x != y:
            bad.append(f"{where}{k}: {a[k]!r} vs {b[k]!r}")
    return bad


# ------------------------------------------------------------------------------------------------
# numpy restatement of the tensor half of tricks/sk_embedding_bag.py (one index per bag)
# ------------------------------------------------------------------------------------------------

def ref_forward(que, w_hot, w_cold, hash_size):
    """sk_embedding_bag.py:121-150: que = batch_query(ids + offset); hot -> W_h[|que| % hotn],
    cold -> W_hash[|que| % hash_size]."""
    que = np.asarray(que, dtype=np.int64)
    hot = que < 0
    dic = np.abs(que)
    out = np.where(hot[:, None], w_hot[dic % w_hot.shape[0]], w_cold[dic % hash_size])
    return out.astype(np.float32), hot


def ref_forward_ml(que, qmed, w_hot, w_cold, w_med, hash_size, median_n):
    """sketch_ml_embed.py:166-197: hot ? W_h[|que| % hotn] : W_hash[|que| % hash_size] + (median ? W_med[|que| % median_n] : 0)."""
    que = np.asarray(que)
    dic = np.abs(que).astype(np.int64)
    hot = (que < 0)[:, None]
    med = (np.asarray(qmed) < 0)[:, None]
    low = w_cold[dic % hash_size] + np.where(med, w_med[dic % median_n], np.float32(0))
    return np.where(hot, w_hot[dic % w_hot.shape[0]], low).astype(np.float32)


def ref_scores(grad_out):
    """sk_embedding_bag.py:178-189 with the intended semantics: the score of a lookup is the
    L2 norm of the gradient of its own output row, scaled so a field's scores average 1."""
    g = np.asarray(grad_out, dtype=np.float32)
    gn = np.sqrt((g * g).sum(axis=1, dtype=np.float32)).astype(np.float32)
    n = np.float32(len(gn))
    return (gn * n / gn.sum(dtype=np.float32)).astype(np.float32)


def ref_migrate(ins, que_after, local_ids, w_hot, w_cold, hash_size):
    """sk_embedding_bag.py:199-208 with the Q3 fence: only entries that query() reports hot are copied.
    Source row uses the LOCAL id (sk_embedding_bag.py:207)."""
    for i in np.nonzero(np.asarray(ins))[0]:
        if que_after[i] < 0:
            w_hot[-int(que_after[i])] = w_cold[int(local_ids[i]) % hash_size]


def ref_sgd(que_fwd, grad_out, w_hot, w_cold, hash_size, lr):
    """dlrm_s_pytorch.py:1658 (torch.optim.SGD on the two sparse gradients): duplicates are summed,
    then W -= lr * g.  float64 accumulation here; tests allow 1e-5 relative."""
    que = np.asarray(que_fwd, dtype=np.int64)
    hot = que < 0
    dic = np.abs(que)
    g = np.asarray(grad_out, dtype=np.float64)
    acc_h = np.zeros(w_hot.shape, np.float64)
    acc_c = np.zeros(w_cold.shape, np.float64)
    np.add.at(acc_h, dic[hot] % w_hot.shape[0], g[hot])
    np.add.at(acc_c, dic[~hot] % hash_size, g[~hot])
    w_hot -= (lr * acc_h).astype(np.float32)
    w_cold -= (lr * acc_c).astype(np.float32)
