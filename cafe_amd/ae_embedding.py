"""The 'cafe' embedding layer of the reference's artifact generation (ArtifactEvaluation/embeddings/init_embed.py:92-121,
152-186 and embeddings/sk_embedding_bag.py) on the GPU: the earlier HotSketch (cafe_amd.AESketch, include/cafe_hip.h
section E) routing lookups between a hot table and hashed cold tables.

Knobs of the artifact (main.py:158-159):
    cafe_hot_separate_field   False: ONE sketch and ONE hot table shared by all sketch fields, ids shifted by the field's
                              offset (ss_idx = 0).  True: every field has its own sketch (ss_idx = field index), its own
                              hot table sized from its own cardinality, and still shifts ids by its offset.
    cafe_use_freq             True: the sketch counts occurrences (every score is 1) instead of gradient norms
                              (sk_embedding_bag.py:152-153).

Call order (main.py's training loop through init_embed.py:212-235): forward of every field -> backward -> insert_grad of
every field, in field order (scores -> batch_insert_val -> re-query -> copy the cold row of every newly hot lookup into
its hot row) -> optimizer step.  Here: forward(ids) and backward_update(grad_out, lr).  Sketch calls are the HIP kernels
of cafe_ae.hip, one per field as in the reference; gather, migration and SGD are the table kernels of cafe_embed.hip
driven by the routing (cafe_gather_que_dev, cafe_migrate_que_dev, cafe_backward_sgd_dev).

Differences from running the reference file, all deliberate: the score of a lookup is the norm of the gradient of ITS OWN
output row (the reference slices the shared hot table's gradient by position); a lookup that the insert
reported but that is no longer hot at the re-query is not migrated (the reference would index the hot table with a
global id); fields too small for the sketch, or whose own hot table would have no row, are not handled here
(the reference gives them plain or hashed tables)."""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from .ae_sketch import AESketch, _ptr, _stream


def ae_sizes(counts, dim, compress_rate, hash_rate, separate_field, total_ids=None):
    """hotn (one number, or one per field) and the cold tables' rows as init_embed.py:107-111, 153-156, 181 compute them."""
    share = compress_rate * (1 - hash_rate) * (dim * 4 / (dim * 4 + 48))
    if separate_field:
        hotn = [int(n * share) for n in counts]
    else:
        hotn = int((sum(counts) if total_ids is None else total_ids) * share)
    hash_sizes = [int(math.ceil(compress_rate * hash_rate * n)) for n in counts]
    return hotn, hash_sizes


class AECafeEmbedding:
    def __init__(self, counts, dim, hotn, hash_sizes, threshold, decay, hot_separate_field=False, use_freq=False,
                 offsets=None, device=None, seed=0):
        if not torch.cuda.is_available():
            raise _lib.CafeError("cafe_amd needs a GPU: there is no CPU path in this package")
        self.L = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.counts = [int(c) for c in counts]
        self.F = len(self.counts)
        self.dim = int(dim)
        self.separate = bool(hot_separate_field)
        self.use_freq = bool(use_freq)
        self.hash_sizes = [int(h) for h in hash_sizes]
        self.offsets = (np.concatenate([[0], np.cumsum(np.asarray(self.counts, np.int64))[:-1]])
                        if offsets is None else np.asarray(offsets, np.int64)).astype(np.int64)
        self.hotn = [int(h) for h in hotn] if self.separate else int(hotn)
        if self.separate and (len(self.hotn) != self.F or min(self.hotn) <= 0):
            raise _lib.CafeError("AECafeEmbedding: every field needs a hot table of its own with at least one row")
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)

        def uniform(rows, bound):
            return ((torch.rand(rows, self.dim, generator=g) * 2 - 1) * bound).to(self.device)

        # init_embed.py:113-120 / 157-163: the hot tables ARE initialised in this generation
        if self.separate:
            self.weight_high = [uniform(h, float(np.sqrt(1.0 / n))) for h, n in zip(self.hotn, self.counts)]
            self.sketches = [AESketch(h, threshold, decay, device=self.device) for h in self.hotn]
        else:
            self.weight_high = [uniform(self.hotn, float(np.sqrt(1.0 / max(self.counts))))]
            self.sketches = [AESketch(self.hotn, threshold, decay, device=self.device)]
        self.weight_hash = [uniform(h, float(np.sqrt(1.0 / n))) for h, n in zip(self.hash_sizes, self.counts)]
        self._off_dev = torch.from_numpy(self.offsets).to(self.device)[:, None]
        self._tables = []
        self._make_tables()
        self._saved = None

    # one table handle per hot table: all fields (shared) or one field each (separate)
    def _make_tables(self):
        for h in self._tables:
            self.L.cafe_tables_destroy(h)
        self._tables = []
        groups = [[f] for f in range(self.F)] if self.separate else [list(range(self.F))]
        for k, fields in enumerate(groups):
            n = len(fields)
            ptrs = (ctypes.c_void_p * n)(*[self.weight_hash[f].data_ptr() for f in fields])
            offs = (ctypes.c_int64 * n)(*[int(self.offsets[f]) for f in fields])
            hs = (ctypes.c_int32 * n)(*[self.hash_sizes[f] for f in fields])
            hotn = self.hotn[k] if self.separate else self.hotn
            h = ctypes.c_void_p()
            with torch.cuda.device(self.device):
                _lib.check(self.L.cafe_tables_create(ctypes.byref(h), n, self.dim, hotn, _ptr(self.weight_high[k]), ptrs, offs, hs),
                           "cafe_tables_create")
            self._tables.append(h)
        self._groups = groups

    def set_tables(self, weight_high, weight_hash):
        """Replace the initial tables (parity tests, checkpoints): lists of arrays/tensors shaped like the current ones."""
        for dst, src in zip(self.weight_high, weight_high):
            dst.copy_(torch.as_tensor(np.asarray(src) if not isinstance(src, torch.Tensor) else src).to(self.device))
        for dst, src in zip(self.weight_hash, weight_hash):
            dst.copy_(torch.as_tensor(np.asarray(src) if not isinstance(src, torch.Tensor) else src).to(self.device))

    def close(self):
        for h in getattr(self, "_tables", []):
            self.L.cafe_tables_destroy(h)
        self._tables = []
        for s in getattr(self, "sketches", []):
            s.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def on_load(self):
        """init_embed.py:208-210: after a checkpoint is loaded every sketch rebuilds its free list."""
        for s in self.sketches:
            s.update()

    # ---------------------------------------------------------------------------------------------
    def forward(self, local_ids):
        """local_ids [F, B] int64 on the device -> [F, B, D]; the routing is kept for backward_update."""
        assert local_ids.dtype == torch.int64 and local_ids.is_cuda and local_ids.shape[0] == self.F
        local_ids = local_ids.contiguous()
        B = local_ids.shape[1]
        gids = (local_ids + self._off_dev).to(torch.int32)           # sk_embedding_bag.py:80 (wraps like astype(np.int32))
        out = torch.empty(self.F, B, self.dim, device=self.device)
        que = torch.empty(self.F, B, dtype=torch.int32, device=self.device)
        for k, fields in enumerate(self._groups):
            lo, hi = fields[0], fields[-1] + 1
            que[lo:hi] = self.sketches[k].query(gids[lo:hi].reshape(-1)).view(hi - lo, B)
            _lib.check(self.L.cafe_gather_que_dev(self._tables[k], _ptr(que[lo:hi]), None, B, (hi - lo) * B, _ptr(out[lo:hi]),
                                                  _stream()), "cafe_gather_que_dev")
        self._saved = (local_ids, gids, que, B)
        return out

    def backward_update(self, grad_out, lr):
        """insert_grad of every field, then SGD on every table.  Returns (scores [F, B], ins [F, B])."""
        local_ids, gids, que, B = self._saved
        grad_out = grad_out.contiguous()
        st = _stream()
        scores = torch.ones(self.F, B, device=self.device)
        ins = torch.empty(self.F, B, dtype=torch.int32, device=self.device)
        for k, fields in enumerate(self._groups):
            lo, hi = fields[0], fields[-1] + 1
            if not self.use_freq:                                    # ||g|| * N / sum ||g|| per field (:155-162)
                _lib.check(self.L.cafe_scores_dev(self._tables[k], _ptr(grad_out[lo:hi]), B, _ptr(scores[lo:hi]), st),
                           "cafe_scores_dev")
            for f in fields:                                         # init_embed.py:230-233: field by field
                sk = self.sketches[k]
                ins[f] = sk.insert_val(gids[f], scores[f])
                que_after = sk.query(gids[f]).contiguous()
                fld = torch.full((B,), f - lo, dtype=torch.int32, device=self.device)
                _lib.check(self.L.cafe_migrate_que_dev(self._tables[k], _ptr(local_ids[f]), _ptr(que_after), _ptr(fld),
                                                       _ptr(ins[f]), 1, B, st), "cafe_migrate_que_dev")
        for k, fields in enumerate(self._groups):
            lo, hi = fields[0], fields[-1] + 1
            _lib.check(self.L.cafe_backward_sgd_dev(self._tables[k], _ptr(gids[lo:hi]), _ptr(que[lo:hi]), _ptr(grad_out[lo:hi]),
                                                    B, ctypes.c_float(lr), st), "cafe_backward_sgd_dev")
        return scores, ins
