"""SKEmbeddingBag -- the reference's embedding module (tricks/sk_embedding_bag.py:26-208) on MI355X.

Same constructor, attributes and call order as the reference, so the reference drivers use it unchanged
(dlrm_s_pytorch.py:309-322 create_emb, :653-659 apply_emb, :734-738 insert_grad):

    lib = ctypes.CDLL(cafe_amd.LIB_PATH)            # instead of ./tricks/sklib.so   (dlrm_s_pytorch.py:1403)
    lib.init(hotn, threshold, adjust, alpha)         # unchanged                      (dlrm_s_pytorch.py:444-449)
    E = SKEmbeddingBag(field, hotn, lib, weight_high, device, n, m, offset, hash_size)
    out = E(ids, offsets)                            # query + hot/cold gather, ONE HIP launch
    loss.backward(); E.insert_grad(ids)              # scores -> batch_insert_val -> migration, all on the GPU
    optimizer.step()                                 # stock torch.optim.SGD on the two sparse gradients

What differs from the reference is where things run: ids are copied to HBM once, the sketch is queried by
the same kernel that gathers the rows, and insert_grad never leaves the device (the reference does a D2H
copy of the scores, a ctypes call and a Python loop of row copies, sk_embedding_bag.py:190-208).

Semantics that follow SURVEY.md rather than the reference's accidents: a lookup's score is the norm of the
gradient of its OWN output row (Q2; the reference slices an accumulated sparse gradient, which crashes on CPU
and reads another field's rows on GPU), and a newly hot id is migrated only if the sketch reports it hot
afterwards (Q3).  There is no CPU path: without a GPU and libcafe_hip.so every call raises."""
import ctypes

import numpy as np
import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from . import _lib

# ---------------------------------------------------------------------------------------------------
# sketch timers: the reference declares them but never accumulates (sk_embedding_bag.py:13-23); the
# driver prints them (dlrm_s_pytorch.py:1706).  Here they are real: GPU time of the sketch calls.
# ---------------------------------------------------------------------------------------------------
_sketch_ms = 0.0
_pending = []


def _fold(block):
    global _sketch_ms
    keep = []
    for a, b in _pending:
        if block:
            b.synchronize()
        if b.query():
            _sketch_ms += a.elapsed_time(b)
        else:
            keep.append((a, b))
    _pending[:] = keep


class _Timed:
    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        _pending.append((self.a, self.b))
        if len(_pending) > 512:
            _fold(False)


def get_sketch_time():
    """Seconds of GPU time spent in sketch calls (query+gather launches, insert, migration) since the last reset."""
    _fold(True)
    return _sketch_ms * 1e-3


def reset_sketch_time():
    global _sketch_ms
    _fold(True)
    _sketch_ms = 0.0


def sketch_state_dict():
    """Complete state of the global sketch behind `lib` (every slot, scalars, free list) as CPU tensors and Python
    scalars: what the reference's checkpoint should carry in `sketch_buffer` (dlrm_s_pytorch.py:443, never filled
    there).  Store it next to model.state_dict(); restore with load_sketch_state_dict after lib.init(...)."""
    from .sketch import HotSketch
    return HotSketch.from_global().state_tensors()


def load_sketch_state_dict(state):
    """Inverse of sketch_state_dict; makes the lib.update() the reference calls on load (:1510-1511) unnecessary
    (the free list is part of the state), though calling it stays harmless."""
    from .sketch import HotSketch
    HotSketch.from_global().load_state_tensors(state)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class _RoutedGather(torch.autograd.Function):
    """out[i] = hot_i ? W_h[slot_i] : W_hash[gid_i % hash_size]; backward hands the two tables the sparse
    gradients F.embedding_bag(sparse=True) + torch.where produce in the reference (sk_embedding_bag.py:134-150)."""

    @staticmethod
    def forward(ctx, weight_h, weight_hash, mod, ids):
        ctx.mod = mod
        ctx.sizes = (weight_h.shape, weight_hash.shape)
        out = mod._launch_forward(ids)
        # the routing and ids of THIS forward: the module's own fields follow its latest forward, which may be another
        # one (an evaluation between forward and backward, a second micro-batch)
        ctx.que, ctx.ids, ctx.gids = mod._que, mod._ids, mod._gids
        return out

    @staticmethod
    def backward(ctx, g):
        mod = ctx.mod
        g = g.contiguous()
        mod._grad_out = g                       # insert_grad scores this lookup by the norm of its own row (Q2)
        mod._grad_ids = (ctx.ids, ctx.gids)     # ... and feeds the sketch the ids this gradient belongs to
        que = ctx.que
        hot = (que < 0)[:, None]
        dic = que.abs().to(torch.int64)
        zero = torch.zeros((), dtype=g.dtype, device=g.device)
        (hs, hd), (cs, cd) = ctx.sizes
        gh = torch.sparse_coo_tensor((dic % mod.hot_nums)[None], torch.where(hot, g, zero), (hs, hd))
        gc = torch.sparse_coo_tensor((dic % mod.hash_size)[None], torch.where(hot, zero, g), (cs, cd))
        return gh, gc, None, None


class SKEmbeddingBag(nn.Module):
    def __init__(self, field_num, hotn, lib, weight_high, device, num_categories, embedding_dim, f_offset, hash_size,
                 max_norm=None, norm_type=2.0, scale_grad_by_freq=False, sparse=False):
        super().__init__()
        self.field_num = field_num
        self.lib = lib
        self.offset = int(f_offset)
        self.hash_size = int(hash_size)
        self.hot_nums = int(hotn)
        self.input_c = None                      # reference: ctypes pointer to the int32 ids; here: device tensor
        self.num_categories = num_categories
        self.weight_h = weight_high
        self.embedding_dim = int(embedding_dim)
        self.max_norm = max_norm
        self.norm_type = norm_type
        self.scale_grad_by_freq = scale_grad_by_freq
        self.device = torch.device(device)
        self.grad_norm = 0
        self.sparse = sparse
        self._L = _lib.load()                    # raises if libcafe_hip.so is missing: no fallback
        self._sketch_h = getattr(lib, "h", None)  # a cafe_amd.HotSketch works as `lib` too
        self._tables = None
        self._tables_key = None
        self._que = None
        self._mask = None
        self._grad_out = None
        self._grad_ids = None
        self.weight_hash = Parameter(torch.empty(self.hash_size, self.embedding_dim))
        self.reset_parameters()

    # -- reference API ------------------------------------------------------------------------------
    def reset_parameters(self):
        b = float(np.sqrt(1 / self.num_categories))           # sk_embedding_bag.py:115-117
        nn.init.uniform_(self.weight_hash, -b, b)

    @property
    def query_dic(self):
        """Hot mask of the last forward as a numpy bool array (sk_embedding_bag.py:130); copied on demand."""
        if getattr(self, "_query_dic", None) is not None:
            return self._query_dic
        return None if self._mask is None else self._mask.cpu().numpy()

    @query_dic.setter
    def query_dic(self, value):
        # the reference assigns this attribute itself (sk_embedding_bag.py:130); code written against it may too
        self._query_dic = value

    def _sketch(self):
        if self._sketch_h is None:
            h = self._L.cafe_global_sketch()
            if not h:
                raise _lib.CafeError("SKEmbeddingBag: call lib.init(hotn, threshold, adjust, alpha) first "
                                     "(dlrm_s_pytorch.py:444-449)")
            self._sketch_h = ctypes.c_void_p(h)
        return self._sketch_h

    def _device_ids(self, input):
        if not torch.cuda.is_available():
            raise _lib.CafeError("cafe_amd.SKEmbeddingBag needs a GPU: there is no CPU path")
        if not isinstance(input, torch.Tensor):
            input = torch.as_tensor(np.asarray(input))
        return input.to(self.device, torch.int64, non_blocking=True).contiguous()

    def _get_tables(self):
        wh, wc = self.weight_h, self.weight_hash
        if not (wh.is_cuda and wc.is_cuda):
            raise _lib.CafeError("SKEmbeddingBag: weight_high and weight_hash must live on the GPU (model.to(device))")
        if wh.dtype != torch.float32 or not wh.is_contiguous() or not wc.is_contiguous():
            raise _lib.CafeError("SKEmbeddingBag: tables must be contiguous float32")
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

    def _launch_forward(self, ids):
        n = ids.numel()
        t = self._get_tables()
        gids = torch.empty(max(n, 1), dtype=torch.int32, device=ids.device)
        que = torch.empty(max(n, 1), dtype=torch.int32, device=ids.device)
        out = torch.empty(n, self.embedding_dim, device=ids.device)
        if n:
            with _Timed():
                _lib.check(self._L.cafe_forward_dev(self._sketch(), t, _p(ids), n, _p(gids), _p(que), _p(out), _stream()),
                           "cafe_forward_dev")
        self._ids, self._gids, self._que = ids, gids[:n], que[:n]
        self._query_dic = None
        self.input_c = self._gids                 # sk_embedding_bag.py:106: the ids insert_grad will feed the sketch
        self._mask = self._que < 0
        return out

    def query(self, input):
        """(hot mask, |que|) of input + offset (sk_embedding_bag.py:101-113); CPU tensors like the reference's."""
        ids = self._device_ids(input)
        gids = (ids + self.offset).to(torch.int32)
        que = torch.empty(max(ids.numel(), 1), dtype=torch.int32, device=ids.device)
        with _Timed():
            _lib.check(self._L.cafe_query_dev(self._sketch(), _p(gids), ids.numel(), _p(que), _stream()), "cafe_query_dev")
        self.input_c = gids
        que = que[:ids.numel()].cpu()
        return que < 0, que.abs()

    def insert(self, input):
        """batch_insert (score 1 per id) then query (sk_embedding_bag.py:83-99): (newly hot, hot mask, |que|)."""
        ids = self._device_ids(input)
        n = ids.numel()
        gids = (ids + self.offset).to(torch.int32)
        ins = torch.empty(max(n, 1), dtype=torch.int32, device=ids.device)
        que = torch.empty(max(n, 1), dtype=torch.int32, device=ids.device)
        with _Timed():
            _lib.check(self._L.cafe_insert_dev(self._sketch(), _p(gids), None, n, _p(ins), _stream()), "cafe_insert_dev")
            _lib.check(self._L.cafe_query_dev(self._sketch(), _p(gids), n, _p(que), _stream()), "cafe_query_dev")
        que = que[:n].cpu()
        return ins[:n].cpu(), que < 0, que.abs()

    def forward(self, input, offsets=None, per_sample_weights=None, test=False):
        """One index per bag (offsets = arange(N), dlrm_data_pytorch.py:404-405), so pooling is a row gather.
        `test` is accepted and ignored exactly like the reference (sk_embedding_bag.py:122-129)."""
        ids = self._device_ids(input)
        if offsets is not None and offsets.numel() != ids.numel():
            raise _lib.CafeError("SKEmbeddingBag: one index per bag expected (offsets = arange(len(input)))")
        if per_sample_weights is not None:
            raise _lib.CafeError("SKEmbeddingBag: per_sample_weights is not supported (the reference ignores it)")
        return _RoutedGather.apply(self.weight_h, self.weight_hash, self, ids)

    def insert_grad(self, input):
        """sk_embedding_bag.py:173-208: importance scores from this step's gradient -> batch_insert_val ->
        copy the cold row of every id that just became hot into its hot slot.  `input` is accepted for
        signature parity; the ids are the ones the forward saw (the reference reuses self.input_c the same way)."""
        g = self._grad_out
        if g is None:
            raise _lib.CafeError("insert_grad: no gradient recorded -- call loss.backward() after forward first")
        ids, gids = self._grad_ids
        n = gids.numel()
        if n == 0:
            return
        t = self._get_tables()
        scores = torch.empty(n, dtype=torch.float32, device=g.device)
        ins = torch.empty(n, dtype=torch.int32, device=g.device)
        st = _stream()
        with _Timed():
            _lib.check(self._L.cafe_scores_dev(t, _p(g), n, _p(scores), st), "cafe_scores_dev")
            _lib.check(self._L.cafe_insert_dev(self._sketch(), _p(gids), _p(scores), n, _p(ins), st), "cafe_insert_dev")
            _lib.check(self._L.cafe_migrate_dev(self._sketch(), t, _p(ids), _p(gids), _p(ins), n, 0, 1, st),
                       "cafe_migrate_dev")
        self._scores, self._ins = scores, ins
        self._grad_out = self._grad_ids = None

    def extra_repr(self):
        return f"{self.num_categories}, {self.embedding_dim}, hot={self.hot_nums}, hash_size={self.hash_size}, offset={self.offset}"
