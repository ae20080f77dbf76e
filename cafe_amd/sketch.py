"""HotSketch handle: the device-resident sketch behind the reference's `lib` object.

The reference keeps one global sketch inside sklib.so and talks to it through ctypes
(tricks/sk_embedding_bag.py:61-73, tricks/sketch.cpp:300-313).  Here the sketch is an explicit handle
whose tables live in HBM; ids, scores and results are torch CUDA tensors (device pointers cross the
C ABI of include/cafe_hip.h, nothing else)."""
import ctypes

import numpy as np
import torch

from . import _lib

_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class HotSketch:
    """One HotSketch on the current CUDA(HIP) device.  Arguments as lib.init (sketch.cpp:309-313)."""

    def __init__(self, lim, threshold, adjust_thres=1, alpha=1.0, lru_first=1, device=None):
        if not torch.cuda.is_available():
            raise _lib.CafeError("cafe_amd.HotSketch needs a GPU: there is no CPU path in this package")
        self.L = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.lim = int(lim)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.L.cafe_sketch_create(ctypes.byref(h), int(lim), int(threshold), int(adjust_thres),
                                                 float(alpha), int(lru_first)), "cafe_sketch_create")
        self.h = h
        self.n1 = int(lim * 0.9)
        self.n2 = int(lim * 0.1)

    @classmethod
    def from_global(cls):
        """A view of the process-global sketch that lib.init() created (reference ABI, sketch.cpp:297, :309-313);
        it is not destroyed with this object."""
        self = cls.__new__(cls)
        self.L = _lib.load()
        h = self.L.cafe_global_sketch()
        if not h:
            raise _lib.CafeError("no global sketch: call lib.init(hotn, threshold, adjust, alpha) first")
        self.h = ctypes.c_void_p(h)
        self._borrowed = True
        dims = (ctypes.c_int32 * 4)()
        _lib.check(self.L.cafe_sketch_dims(self.h, dims), "cafe_sketch_dims")
        self.n1, self.n2, self.lim = int(dims[0]), int(dims[1]), int(dims[2])
        self.device = torch.device("cuda", torch.cuda.current_device())
        return self

    def close(self):
        if getattr(self, "h", None) and not getattr(self, "_borrowed", False):
            self.L.cafe_sketch_destroy(self.h)
        self.h = None

    def state_tensors(self):
        """state() as CPU tensors and Python scalars: goes into a torch checkpoint next to the model's state_dict
        and loads back with torch.load(weights_only=True).  The reference meant `sketch_buffer` for this
        (dlrm_s_pytorch.py:443) but never fills it, so its checkpoints lose the sketch."""
        out = {}
        for k, v in self.state().items():
            if isinstance(v, np.ndarray):
                out[k] = torch.from_numpy(np.ascontiguousarray(v.astype(np.int64) if v.dtype == np.uint32 else v))
            else:
                out[k] = float(v) if isinstance(v, (float, np.floating)) else int(v)
        return out

    def load_state_tensors(self, st):
        self.load_state_dict({k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in st.items()})

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers -------------------------------------------------------------------------------
    def _ids(self, ids):
        if isinstance(ids, torch.Tensor):
            if ids.dtype != torch.int32:
                ids = ids.to(torch.int32)      # wraps like numpy astype(int32), sk_embedding_bag.py:103
            return ids.to(self.device).contiguous()
        a = np.ascontiguousarray(np.asarray(ids).astype(np.int64) & 0xFFFFFFFF, dtype=np.uint32).view(np.int32)
        return torch.from_numpy(a.copy()).to(self.device)

    def set_option(self, name, value):
        _lib.check(self.L.cafe_sketch_set_option(self.h, name.encode(), int(value)), "set_option")

    def counters(self):
        out = (ctypes.c_int64 * 8)()
        self.L.cafe_sketch_counters(self.h, out)
        names = ("insert_calls", "parallel_calls", "inorder_calls", "rounds_total", "resets", "rounds_max", "cadence_miss",
                 "exact_waits")
        d = dict(zip(names, (int(x) for x in out)))
        why = (ctypes.c_int64 * 4)()
        self.L.cafe_sketch_fallback_reasons(self.h, why)
        d["left_parallel_path"] = {"workspace_limit": int(why[0]), "rounds_not_settled": int(why[1]), "free_list": int(why[2])}
        d["inner_reset_rejects"] = int(why[3])
        ren = ctypes.c_int64()
        self.L.cafe_sketch_clock_renorms(self.h, ctypes.byref(ren))
        d["clock_renorms"] = int(ren.value)
        return d

    # -- the three reference operations ---------------------------------------------------------
    def query(self, ids, out=None):
        ids = self._ids(ids)
        que = out if out is not None else torch.empty(ids.numel(), dtype=torch.int32, device=self.device)
        _lib.check(self.L.cafe_query_dev(self.h, _ptr(ids), ids.numel(), _ptr(que), _stream()), "cafe_query_dev")
        return que

    def insert_val(self, ids, scores=None, out=None):
        ids = self._ids(ids)
        n = ids.numel()
        sp = None
        if scores is not None:
            if not isinstance(scores, torch.Tensor):
                scores = torch.from_numpy(np.ascontiguousarray(scores, dtype=np.float32))
            scores = scores.to(self.device, torch.float32).contiguous()
            assert scores.numel() == n
            sp = _ptr(scores)
        ins = out if out is not None else torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        _lib.check(self.L.cafe_insert_dev(self.h, _ptr(ids), sp, n, _ptr(ins), _stream()), "cafe_insert_dev")
        return ins[:n]

    def insert(self, ids, out=None):
        return self.insert_val(ids, None, out)

    # -- state ------------------------------------------------------------------------------------
    def state(self):
        """Every slot and scalar as flat numpy arrays ([n1*4] / [n2*4] slot arrays, flip bytes, free list in pop order)."""
        dims = (ctypes.c_int32 * 4)()
        _lib.check(self.L.cafe_sketch_dims(self.h, dims), "cafe_sketch_dims")
        n1, n2, lim, flen = (int(x) for x in dims)
        st = {
            "mval": np.zeros(n1 * 4, np.int32), "mcnt": np.zeros(n1 * 4, np.float32), "mdic": np.zeros(n1 * 4, np.int32),
            "mflip": np.zeros(max(n1, 1), np.uint8),
            "lval": np.zeros(n2 * 4, np.int32), "lcnt": np.zeros(n2 * 4, np.float32), "ldic": np.zeros(n2 * 4, np.int32),
            "lt": np.zeros(n2 * 4, np.int32), "lflip": np.zeros(max(n2, 1), np.uint8),
            "fifo": np.zeros(max(flen, 1), np.uint32),
        }
        _lib.check(self.L.cafe_sketch_export(
            self.h, st["mval"].ctypes.data_as(_i32p), st["mcnt"].ctypes.data_as(_f32p), st["mdic"].ctypes.data_as(_i32p),
            st["mflip"].ctypes.data_as(_u8p), st["lval"].ctypes.data_as(_i32p), st["lcnt"].ctypes.data_as(_f32p),
            st["ldic"].ctypes.data_as(_i32p), st["lt"].ctypes.data_as(_i32p), st["lflip"].ctypes.data_as(_u8p),
            st["fifo"].ctypes.data_as(_u32p)), "cafe_sketch_export")
        thr = ctypes.c_float()
        ints = (ctypes.c_int32 * 6)()
        decay = ctypes.c_double()
        _lib.check(self.L.cafe_sketch_scalars(self.h, ctypes.byref(thr), ints, ctypes.byref(decay)), "cafe_sketch_scalars")
        st["mflip"] = st["mflip"][:n1]
        st["lflip"] = st["lflip"][:n2]
        st["fifo"] = st["fifo"][:flen]
        st["thr"] = np.float32(thr.value)
        st["num"], st["real_n"], st["clk"], st["flip"], st["batch_num"], st["resets"] = (int(x) for x in ints)
        st["decay"] = float(decay.value)
        st["n1"], st["n2"], st["lim"] = n1, n2, lim
        return st

    def load_state_dict(self, st):
        """Inverse of state(): used by checkpoint resume and by tests that start from an oracle state."""
        a = {k: np.ascontiguousarray(st[k]) for k in ("mval", "mcnt", "mdic", "mflip", "lval", "lcnt", "ldic", "lt", "lflip")}
        fifo = np.ascontiguousarray(st["fifo"], dtype=np.uint32)
        ints = (ctypes.c_int32 * 6)(int(st["num"]), int(st["real_n"]), int(st["clk"]), int(st["flip"]),
                                    int(st["batch_num"]), int(st.get("resets", 0)))
        _lib.check(self.L.cafe_sketch_import(
            self.h, a["mval"].ctypes.data_as(_i32p), a["mcnt"].ctypes.data_as(_f32p), a["mdic"].ctypes.data_as(_i32p),
            a["mflip"].astype(np.uint8).ctypes.data_as(_u8p), a["lval"].ctypes.data_as(_i32p), a["lcnt"].ctypes.data_as(_f32p),
            a["ldic"].ctypes.data_as(_i32p), a["lt"].ctypes.data_as(_i32p), a["lflip"].astype(np.uint8).ctypes.data_as(_u8p),
            fifo.ctypes.data_as(_u32p), len(fifo), ctypes.c_float(float(st["thr"])), ints, float(st["decay"])),
            "cafe_sketch_import")

    def save(self, path):
        _lib.check(self.L.cafe_sketch_save(self.h, str(path).encode()), "cafe_sketch_save")

    def load(self, path):
        _lib.check(self.L.cafe_sketch_load(self.h, str(path).encode()), "cafe_sketch_load")

    def rebuild_free_list(self):
        _lib.check(self.L.cafe_sketch_rebuild_free_list(self.h, _stream()), "cafe_sketch_rebuild_free_list")
