"""CafeEmbeddingCollection: all sketch fields of a model behind one set of fused HIP launches.

The reference loops over its fields in Python, per field: ids to numpy, ctypes batch_query, three H2D
copies, two embedding_bag kernels and a where (dlrm_s_pytorch.py:613-659, sk_embedding_bag.py:119-152),
and after backward a D2H sync plus a per-row Python copy loop (sk_embedding_bag.py:173-208).  Here ids
stay in HBM and the whole step is: forward (one launch), scores, sketch insert, migration, backward+SGD.

Call order and semantics are the reference's:
    forward           query with the PRE-insert sketch, gather hot/cold rows
    backward_update   scores from the output gradient -> batch_insert_val field by field -> migration of
                      newly hot ids (copies the PRE-update cold row) -> SGD on both tables
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .sketch import HotSketch, _ptr, _stream


class CafeEmbeddingCollection:
    def __init__(self, counts, dim, hotn, hash_sizes, offsets=None, threshold=500, adjust_thres=1, alpha=1.0000005,
                 device=None, seed=0, lru_first=1, init_on_device=False):
        if not torch.cuda.is_available():
            raise _lib.CafeError("cafe_amd needs a GPU: there is no CPU path in this package")
        self.L = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.counts = [int(c) for c in counts]
        self.F = len(self.counts)
        self.dim = int(dim)
        self.hotn = int(hotn)
        self.hash_sizes = [int(h) for h in hash_sizes]
        self.offsets = (np.concatenate([[0], np.cumsum(np.asarray(self.counts, np.int64))[:-1]])
                        if offsets is None else np.asarray(offsets, np.int64)).astype(np.int64)
        self.sketch = HotSketch(self.hotn, threshold, adjust_thres, alpha, lru_first, device=self.device)
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        # hot table: the reference leaves it uninitialised (dlrm_s_pytorch.py:439-442); zeros here
        self.weight_high = torch.zeros(self.hotn, self.dim, device=self.device)
        # cold tables: one allocation, per-field views; U(-sqrt(1/n), sqrt(1/n)) (sk_embedding_bag.py:115-117)
        total = sum(self.hash_sizes)
        self._cold = torch.empty(total, self.dim, device=self.device)
        self.weight_hash = []
        r0 = 0
        gd = None
        if init_on_device:          # tens of GB of tables (cfg4): draw on the GPU instead of through host memory
            gd = torch.Generator(device=self.device)
            gd.manual_seed(seed)
        for n, h in zip(self.counts, self.hash_sizes):
            w = self._cold[r0:r0 + h]
            b = float(np.sqrt(1.0 / n))
            if gd is not None:
                w.uniform_(-b, b, generator=gd)
            else:
                w.copy_((torch.rand(h, self.dim, generator=g) * 2 - 1) * b)
            self.weight_hash.append(w)
            r0 += h
        self._make_tables()
        self._buf = {}

    def _make_tables(self):
        ptrs = (ctypes.c_void_p * self.F)(*[w.data_ptr() for w in self.weight_hash])
        offs = (ctypes.c_int64 * self.F)(*[int(o) for o in self.offsets])
        hs = (ctypes.c_int32 * self.F)(*self.hash_sizes)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.L.cafe_tables_create(ctypes.byref(h), self.F, self.dim, self.hotn,
                                                 ctypes.c_void_p(self.weight_high.data_ptr()), ptrs, offs, hs),
                       "cafe_tables_create")
        self.tables = h

    def close(self):
        if getattr(self, "tables", None):
            self.L.cafe_tables_destroy(self.tables)
            self.tables = None
        self.sketch.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _get(self, name, shape, dtype):
        t = self._buf.get(name)
        if t is None or t.shape != torch.Size(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._buf[name] = t
        return t

    # ---------------------------------------------------------------------------------------------
    def _route_bufs(self, n, which):
        """Two sets of (global ids, routing) arrays, used by alternate steps: the forward named by hint_next() is queued
        while the update of the current step still reads its own set."""
        return (self._get(f"gids{which}", (n,), torch.int32), self._get(f"que{which}", (n,), torch.int32))

    def hint_next(self, local_ids, out):
        """Names the next forward(local_ids, out=out) call (include/cafe_hip.h, cafe_forward_hint_dev): the
        backward_update() of the current step queues it before waiting for the device's verdict.  `local_ids` (a data
        loader's next batch) must be complete in memory; `out` is required, it is the array the queued gather writes."""
        assert local_ids.dtype == torch.int64 and local_ids.is_cuda and local_ids.shape[0] == self.F and local_ids.is_contiguous()
        assert out is not None and out.is_contiguous()
        B = local_ids.shape[1]
        which = 1 - getattr(self, "_which", 0)
        gids, que = self._route_bufs(self.F * B, which)
        self._hint = (local_ids.data_ptr(), out.data_ptr(), B, which, local_ids)
        _lib.check(self.L.cafe_forward_hint_dev(self.sketch.h, self.tables, _ptr(local_ids), B, _ptr(gids), _ptr(que),
                                                _ptr(out), _stream()), "cafe_forward_hint_dev")

    def hint_stats(self):
        """(announced forwards found queued, queued forwards dropped because their step was rejected)"""
        o = (ctypes.c_int64 * 2)()
        _lib.check(self.L.cafe_forward_hint_stats(self.sketch.h, o), "cafe_forward_hint_stats")
        return int(o[0]), int(o[1])

    def forward(self, local_ids, out=None):
        """local_ids: [F, B] int64 on the device.  Returns [F, B, D] float32; keeps ids/routing for backward."""
        assert local_ids.dtype == torch.int64 and local_ids.is_cuda and local_ids.shape[0] == self.F
        local_ids = local_ids.contiguous()
        B = local_ids.shape[1]
        n = self.F * B
        hint = getattr(self, "_hint", None)
        self._hint = None
        if hint is not None and out is not None and hint[:3] == (local_ids.data_ptr(), out.data_ptr(), B):
            self._which = hint[3]               # the call that was announced: same arrays, found queued by the library
        else:
            self._which = 1 - getattr(self, "_which", 1)
        gids, que = self._route_bufs(n, self._which)
        self._buf["gids"], self._buf["que"] = gids, que
        if out is None:
            out = torch.empty(self.F, B, self.dim, device=self.device)
        _lib.check(self.L.cafe_forward_dev(self.sketch.h, self.tables, _ptr(local_ids), B, _ptr(gids), _ptr(que),
                                           _ptr(out), _stream()), "cafe_forward_dev")
        self._ids, self._B = local_ids, B
        return out

    def backward_update(self, grad_out, lr, per_field_calls=True):
        """grad_out: [F, B, D] gradient of the loss w.r.t. forward's output.

        per_field_calls=True: the sketch sees one batch_insert_val per field, exactly like the reference's loop over
        fields (dlrm_s_pytorch.py:734-738); the library still runs the step as one parallel insert whenever no flip
        and no reset can fall between two fields (include/cafe_hip.h, cafe_backward_update_dev).  "loop": the same
        calls issued one by one from here (test aid).  False: one call for the whole step (same id order, but decay
        and reset are evaluated once per step instead of once per field)."""
        B, n = self._B, self.F * self._B
        grad_out = grad_out.contiguous()
        gids, que = self._buf["gids"], self._buf["que"]
        scores = self._get("scores", (n,), torch.float32)
        ins = self._get("ins", (n,), torch.int32)
        st = _stream()
        if per_field_calls != "loop":
            _lib.check(self.L.cafe_backward_update_dev(self.sketch.h, self.tables, _ptr(self._ids), _ptr(gids), _ptr(que),
                                                       _ptr(grad_out), B, ctypes.c_float(lr), _ptr(scores), _ptr(ins),
                                                       1 if per_field_calls else 0, st), "cafe_backward_update_dev")
            return scores, ins
        _lib.check(self.L.cafe_scores_dev(self.tables, _ptr(grad_out), B, _ptr(scores), st), "cafe_scores_dev")
        for f in range(self.F):
            o = f * B * 4
            _lib.check(self.L.cafe_insert_dev(self.sketch.h, ctypes.c_void_p(gids.data_ptr() + o),
                                              ctypes.c_void_p(scores.data_ptr() + o), B,
                                              ctypes.c_void_p(ins.data_ptr() + o), st), "cafe_insert_dev")
            _lib.check(self.L.cafe_migrate_dev(self.sketch.h, self.tables, _ptr(self._ids), _ptr(gids), _ptr(ins), B,
                                               f, 1, st), "cafe_migrate_dev")
        _lib.check(self.L.cafe_backward_sgd_dev(self.tables, _ptr(gids), _ptr(que), _ptr(grad_out), B,
                                                ctypes.c_float(lr), st), "cafe_backward_sgd_dev")
        return scores, ins
