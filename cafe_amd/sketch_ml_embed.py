"""MLEmbeddingBag -- the three-tier ("multi-level") variant of the reference's module, tricks/sketch_ml_embed.py:27-210.

The reference file cannot run: its constructor reads `self.hash_size1` (:87, never set), `reset_parameters` touches
`self.weight_hash2` (:140, never created), `query_median` is declared without its argument but called with one
(:143, :187) and binds `lib.batch_query_median` (:82-84), which no sketch.cpp of the reference exports; no driver
imports the module.  What it evidently means is kept, and each repair is marked below:

    out[i] = hot_i ? W_h[slot_i]                                                        (:169-174, :193-197)
                   : W_hash[gid_i % hash_size] + (median_i ? W_median[gid_i % median_n] : 0)      (:175-192)

* training forward feeds the sketch by FREQUENCY before routing: `batch_insert` (score 1 per id) then `batch_query`
  (:160, :93-111); `test=True` only queries (:157-158).  There is no migration (:161-164 are commented out) and no
  `insert_grad`.
* `median_i` = `batch_query_median(gid_i) < 0` (:148-152), defined in include/cafe_hip.h: the id is not hot, sits in
  the main table without a hot slot, and its score is at least 0.2 x the sketch threshold (`threshold2 = threshold1 *
  0.2`, :66-67, with threshold1 read from the live sketch rather than the constant 500).
* repairs: `weight_hash` has `hash_size` rows (the reference's `hash_size1`); `weight_hash2` does not exist (the
  second-level table is the shared `weight_median` argument, :36, :59, :181-186).

Parity: unpinned against the reference (nothing to run or record there); the GPU path is checked against the numpy
restatement of the formula above driven by the C oracle's sketch (tests/test_gpu_ml_module.py).  No CPU path."""
import ctypes

import numpy as np
import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from . import _lib
from .sk_embedding_bag import _Timed, _p, _stream, get_sketch_time, reset_sketch_time  # noqa: F401  (same timers, :14-24)


class _RoutedGatherML(torch.autograd.Function):
    """Backward hands the three tables the sparse gradients the reference's embedding_bag(sparse=True) + where
    produce (sketch_ml_embed.py:169-197)."""

    @staticmethod
    def forward(ctx, weight_h, weight_hash, weight_median, mod, ids, test):
        ctx.mod = mod
        ctx.sizes = (weight_h.shape, weight_hash.shape, weight_median.shape)
        out = mod._launch_forward(ids, test)
        ctx.que, ctx.qmed = mod._que, mod._qmed      # this forward's routing (the module's fields follow its latest forward)
        return out

    @staticmethod
    def backward(ctx, g):
        mod = ctx.mod
        g = g.contiguous()
        que, qmed = ctx.que, ctx.qmed
        hot = (que < 0)[:, None]
        med = (~hot) & (qmed < 0)[:, None]
        dic = que.abs().to(torch.int64)
        zero = torch.zeros((), dtype=g.dtype, device=g.device)
        hs, cs, ms = ctx.sizes
        gh = torch.sparse_coo_tensor((dic % mod.hot_nums)[None], torch.where(hot, g, zero), tuple(hs))
        gc = torch.sparse_coo_tensor((dic % mod.hash_size)[None], torch.where(hot, zero, g), tuple(cs))
        gm = torch.sparse_coo_tensor((dic % mod.median_nums)[None], torch.where(med, g, zero), tuple(ms))
        return gh, gc, gm, None, None, None


class MLEmbeddingBag(nn.Module):
    def __init__(self, field_num, hotn, median_n, lib, weight_high, weight_median, device, num_categories,
                 embedding_dim, f_offset, hash_size, max_norm=None, norm_type=2.0, scale_grad_by_freq=False,
                 sparse=False):
        super().__init__()
        self.field_num = field_num
        self.lib = lib
        self.offset = int(f_offset)
        self.hash_size = int(hash_size)
        self.hot_nums = int(hotn)
        self.median_nums = int(median_n)
        self.p = 7                                                   # :55, unused there too
        self.num_categories = num_categories
        self.weight_h = weight_high
        self.weight_median = weight_median
        self.embedding_dim = int(embedding_dim)
        self.max_norm = max_norm
        self.norm_type = norm_type
        self.scale_grad_by_freq = scale_grad_by_freq
        self.device = torch.device(device)
        self.grad_norm = 0
        self.threshold1 = 500                                        # :66-67, informative: the live sketch threshold rules
        self.threshold2 = self.threshold1 * 0.2
        self.sparse = sparse
        self._L = _lib.load()                                        # raises if libcafe_hip.so is missing: no fallback
        self._sketch_h = getattr(lib, "h", None)
        self._tables = None
        self._tables_key = None
        self._que = self._qmed = self._mask = None
        self.weight_hash = Parameter(torch.empty(self.hash_size, self.embedding_dim))
        self.reset_parameters()

    def reset_parameters(self):
        b = float(np.sqrt(1 / self.num_categories))                  # :137-139
        nn.init.uniform_(self.weight_hash, -b, b)

    @property
    def query_dic(self):
        return None if self._mask is None else self._mask.cpu().numpy()         # :165

    def _sketch(self):
        if self._sketch_h is None:
            h = self._L.cafe_global_sketch()
            if not h:
                raise _lib.CafeError("MLEmbeddingBag: call lib.init(hotn, threshold, adjust, alpha) first")
            self._sketch_h = ctypes.c_void_p(h)
        return self._sketch_h

    def _device_ids(self, input):
        if not torch.cuda.is_available():
            raise _lib.CafeError("cafe_amd.MLEmbeddingBag needs a GPU: there is no CPU path")
        if not isinstance(input, torch.Tensor):
            input = torch.as_tensor(np.asarray(input))
        return input.to(self.device, torch.int64, non_blocking=True).contiguous()

    def _get_tables(self):
        wh, wc, wm = self.weight_h, self.weight_hash, self.weight_median
        for w in (wh, wc, wm):
            if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
                raise _lib.CafeError("MLEmbeddingBag: tables must be contiguous float32 tensors on the GPU")
        if wm.shape != (self.median_nums, self.embedding_dim):
            raise _lib.CafeError("MLEmbeddingBag: weight_median must be [median_n, embedding_dim]")
        key = (wh.data_ptr(), wc.data_ptr())
        if key != self._tables_key:
            self._free_tables()
            h = ctypes.c_void_p()
            ptrs = (ctypes.c_void_p * 1)(wc.data_ptr())
            offs = (ctypes.c_int64 * 1)(self.offset)
            hs = (ctypes.c_int32 * 1)(self.hash_size)
            with torch.cuda.device(wh.device):
                _lib.check(self._L.cafe_tables_create(ctypes.byref(h), 1, self.embedding_dim, self.hot_nums,
                                                      ctypes.c_void_p(wh.data_ptr()), ptrs, offs, hs), "cafe_tables_create")
            self._tables, self._tables_key = h, key
        return self._tables

    def _free_tables(self):
        if self._tables is not None:
            self._L.cafe_tables_destroy(self._tables)
            self._tables = None
            self._tables_key = None

    def __del__(self):
        try:
            self._free_tables()
        except Exception:
            pass

    def _gids(self, ids):
        return (ids + self.offset).to(torch.int32)                   # :95

    # -- reference API ------------------------------------------------------------------------------
    def insert(self, input):
        """batch_insert then batch_query (:93-111): (newly hot, hot mask, |que|) as CPU tensors."""
        ids = self._device_ids(input)
        n = ids.numel()
        gids = self._gids(ids)
        ins = torch.empty(max(n, 1), dtype=torch.int32, device=ids.device)
        que = torch.empty(max(n, 1), dtype=torch.int32, device=ids.device)
        with _Timed():
            _lib.check(self._L.cafe_insert_dev(self._sketch(), _p(gids), None, n, _p(ins), _stream()), "cafe_insert_dev")
            _lib.check(self._L.cafe_query_dev(self._sketch(), _p(gids), n, _p(que), _stream()), "cafe_query_dev")
        que = que[:n].cpu()
        return ins[:n].cpu(), que < 0, que.abs()

    def query(self, input):
        """:124-135."""
        ids = self._device_ids(input)
        n = ids.numel()
        gids = self._gids(ids)
        que = torch.empty(max(n, 1), dtype=torch.int32, device=ids.device)
        with _Timed():
            _lib.check(self._L.cafe_query_dev(self._sketch(), _p(gids), n, _p(que), _stream()), "cafe_query_dev")
        que = que[:n].cpu()
        return que < 0, que.abs()

    def query_median(self, input):
        """:143-152 with the missing argument restored: CPU bool mask of the ids of medium importance."""
        ids = self._device_ids(input)
        n = ids.numel()
        gids = self._gids(ids)
        out = torch.empty(max(n, 1), dtype=torch.int32, device=ids.device)
        with _Timed():
            _lib.check(self._L.cafe_query_median_dev(self._sketch(), _p(gids), n, _p(out), _stream()), "cafe_query_median_dev")
        return out[:n].cpu() < 0

    def _launch_forward(self, ids, test):
        n = ids.numel()
        t = self._get_tables()
        dev = ids.device
        gids = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        que = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        qmed = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        out = torch.empty(n, self.embedding_dim, device=dev)
        if n:
            with _Timed():
                if not test:                                         # :160: count the ids first, then route by the new state
                    g = self._gids(ids)
                    ins = torch.empty(n, dtype=torch.int32, device=dev)
                    _lib.check(self._L.cafe_insert_dev(self._sketch(), _p(g), None, n, _p(ins), _stream()), "cafe_insert_dev")
                    self._ins = ins
                _lib.check(self._L.cafe_forward_ml_dev(self._sketch(), t, _p(ids), n, _p(self.weight_median), self.median_nums,
                                                       _p(gids), _p(que), _p(qmed), _p(out), _stream()), "cafe_forward_ml_dev")
        self._ids, self._gids_last, self._que, self._qmed = ids, gids[:n], que[:n], qmed[:n]
        self._mask = self._que < 0
        return out

    def forward(self, input, offsets=None, per_sample_weights=None, test=False):
        """:154-199, one index per bag (offsets = arange(N))."""
        ids = self._device_ids(input)
        if offsets is not None and offsets.numel() != ids.numel():
            raise _lib.CafeError("MLEmbeddingBag: one index per bag expected (offsets = arange(len(input)))")
        if per_sample_weights is not None:
            raise _lib.CafeError("MLEmbeddingBag: per_sample_weights is not supported (the reference ignores it)")
        return _RoutedGatherML.apply(self.weight_h, self.weight_hash, self.weight_median, self, ids, bool(test))

    def extra_repr(self):
        return (f"{self.num_categories}, {self.embedding_dim}, hot={self.hot_nums}, median={self.median_nums}, "
                f"hash_size={self.hash_size}, offset={self.offset}")
