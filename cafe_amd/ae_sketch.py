"""The earlier HotSketch generation of the reference's artifact (ArtifactEvaluation/embeddings/sketch.cpp, class SS) on
the GPU: include/cafe_hip.h section (E).  One AESketch = one field's instance (the reference keeps ss[26])."""
import ctypes

import numpy as np
import torch

from . import _lib

_u32p = ctypes.POINTER(ctypes.c_uint32)
_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class AESketch:
    """Arguments as the artifact's init(n, Threshold, dr, idx) (sketch.cpp:207-209) without the field index."""

    def __init__(self, n, threshold, decay_rate, device=None):
        if not torch.cuda.is_available():
            raise _lib.CafeError("cafe_amd.AESketch needs a GPU: there is no CPU path in this package")
        self.L = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.n = int(n)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.L.cafe_ae_create(ctypes.byref(h), int(n), int(threshold), float(decay_rate)), "cafe_ae_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.cafe_ae_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ids(self, ids):
        if not isinstance(ids, torch.Tensor):
            ids = torch.from_numpy(np.ascontiguousarray(np.asarray(ids).astype(np.int64)))
        if ids.dtype != torch.int32:
            ids = ids.to(torch.int64).to(torch.int32)            # wraps to the uint32 bit pattern the reference sees
        return ids.to(self.device).contiguous()

    def insert_val(self, ids, scores=None):
        ids = self._ids(ids)
        n = ids.numel()
        sp = None
        if scores is not None:
            if not isinstance(scores, torch.Tensor):
                scores = torch.from_numpy(np.ascontiguousarray(scores, dtype=np.float32))
            scores = scores.to(self.device, torch.float32).contiguous()
            assert scores.numel() == n
            sp = _ptr(scores)
        ins = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        _lib.check(self.L.cafe_ae_insert_dev(self.h, _ptr(ids), sp, n, _ptr(ins), _stream()), "cafe_ae_insert_dev")
        return ins[:n]

    def insert(self, ids):
        return self.insert_val(ids, None)

    def query(self, ids):
        ids = self._ids(ids)
        que = torch.empty(max(ids.numel(), 1), dtype=torch.int32, device=self.device)
        _lib.check(self.L.cafe_ae_query_dev(self.h, _ptr(ids), ids.numel(), _ptr(que), _stream()), "cafe_ae_query_dev")
        return que[:ids.numel()]

    def update(self):
        _lib.check(self.L.cafe_ae_update(self.h, _stream()), "cafe_ae_update")

    def state(self):
        d = (ctypes.c_int64 * 5)()
        _lib.check(self.L.cafe_ae_dims(self.h, d), "cafe_ae_dims")
        s, flen, num, decays, serial = (int(x) for x in d)
        st = {"val": np.zeros(s * 4, np.uint32), "cnt": np.zeros(s * 4, np.float32), "dic": np.zeros(s * 4, np.int32),
              "fifo": np.zeros(max(flen, 1), np.uint32)}
        tot = ctypes.c_double()
        _lib.check(self.L.cafe_ae_export(self.h, st["val"].ctypes.data_as(_u32p), st["cnt"].ctypes.data_as(_f32p),
                                         st["dic"].ctypes.data_as(_i32p), st["fifo"].ctypes.data_as(_u32p), ctypes.byref(tot)),
                   "cafe_ae_export")
        st["fifo"] = st["fifo"][:flen]
        st["num"], st["decays"], st["serial_calls"], st["tot"] = num, decays, serial, float(tot.value)
        return st

    def load_state_dict(self, st):
        val = np.ascontiguousarray(st["val"], np.uint32); cnt = np.ascontiguousarray(st["cnt"], np.float32)
        dic = np.ascontiguousarray(st["dic"], np.int32); fifo = np.ascontiguousarray(st["fifo"], np.uint32)
        _lib.check(self.L.cafe_ae_import(self.h, val.ctypes.data_as(_u32p), cnt.ctypes.data_as(_f32p), dic.ctypes.data_as(_i32p),
                                         fifo.ctypes.data_as(_u32p), len(fifo), float(st["tot"]), int(st["num"])),
                   "cafe_ae_import")
