"""Multi-GPU CAFE embedding: tables and sketch hash-sharded by id, one process per GPU.

The reference is single-device; this is the MI355X-native scale-out of the same hot path
(SURVEY.md 8(e)).  Every global id has one owner rank, owner = mix(gid) mod G.  The owner holds
    * an independent HotSketch with lim = hotn / G,
    * hotn / G hot rows,
    * for every field ceil(hash_size_f / G) cold rows, addressed gid mod that size,
so a shard is, by itself, exactly the single-GPU layer fed the sub-stream of lookups routed to it (in source
rank order, then original order) -- which is how parity is defined and tested per shard.

Per step (data-parallel batch, B samples per rank):
    forward   all_to_all(ids) -> owners run the fused query+gather -> all_to_all(rows) back
    backward  per-field sum of gradient norms all-reduced (scores are normalised over the GLOBAL batch)
              -> all_to_all(gradient rows, scores) -> owners run insert / migration / backward+SGD
All-to-all is the collective that fits xGMI: every pair of GPUs has its own link, so the G-1 transfers of a rank
run concurrently (ring all-reduce would be bound by one link).

The exchange logic below is plain torch + torch.distributed and also runs on CPU tensors over gloo (tests);
the owner-side compute is a backend object: HipShardBackend (libcafe_hip.so) in production."""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _lib

_MIX = 2654435761


def owner_of(gids, world):
    """Owner rank of each global id (int32 bit patterns): multiplicative mix, then mod."""
    x = gids.to(torch.int64) & 0xFFFFFFFF
    return ((x * _MIX) >> 15) % world


def shard_sizes(hotn, hash_sizes, world):
    return max(10, hotn // world), [max(1, -(-h // world)) for h in hash_sizes]


class HipShardBackend:
    """Owner-side compute on this rank's GPU through the C ABI (include/cafe_hip.h, *_gid_dev entry points)."""

    def __init__(self, counts, dim, hotn, hash_sizes, offsets, threshold, adjust_thres, alpha, seed, device):
        from .collection import CafeEmbeddingCollection
        self.coll = CafeEmbeddingCollection(counts, dim, hotn, hash_sizes, offsets, threshold=threshold,
                                            adjust_thres=adjust_thres, alpha=alpha, device=device, seed=seed)
        self.L = self.coll.L
        self.sketch = self.coll.sketch
        self.dim = dim
        self.device = self.coll.device

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr())

    @staticmethod
    def _st():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def forward_gid(self, gids, fields):
        n = gids.numel()
        que = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        rows = torch.empty(max(n, 1), self.dim, device=self.device)
        _lib.check(self.L.cafe_forward_gid_dev(self.sketch.h, self.coll.tables, self._p(gids), self._p(fields), n,
                                               self._p(que), self._p(rows), self._st()), "cafe_forward_gid_dev")
        return que[:n], rows[:n]

    def grad_norm(self, grad, batch):
        F = grad.shape[0]
        gn = torch.empty(F * batch, device=self.device)
        sums = torch.empty(F, device=self.device)
        _lib.check(self.L.cafe_grad_norm_dev(self.coll.tables, self._p(grad), batch, self._p(gn), self._p(sums), self._st()),
                   "cafe_grad_norm_dev")
        return gn, sums

    def backward_gid(self, gids, fields, que, grads, scores, lr):
        n = gids.numel()
        ins = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        st = self._st()
        _lib.check(self.L.cafe_insert_dev(self.sketch.h, self._p(gids), self._p(scores), n, self._p(ins), st), "cafe_insert_dev")
        _lib.check(self.L.cafe_migrate_gid_dev(self.sketch.h, self.coll.tables, self._p(gids), self._p(fields), self._p(ins),
                                               n, st), "cafe_migrate_gid_dev")
        _lib.check(self.L.cafe_backward_sgd_gid_dev(self.coll.tables, self._p(fields), self._p(que), self._p(grads), n,
                                                    ctypes.c_float(lr), st), "cafe_backward_sgd_gid_dev")
        return ins[:n]


class ShardedCafeEmbedding:
    def __init__(self, counts, dim, hotn, hash_sizes, offsets=None, threshold=500, adjust_thres=1, alpha=1.0000005, seed=0,
                 group=None, backend=None, device=None):
        self.group = group
        self._host_staged = dist.get_backend(group) == "gloo"
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.counts = [int(c) for c in counts]
        self.F = len(self.counts)
        self.dim = int(dim)
        self.offsets_np = (np.concatenate([[0], np.cumsum(np.asarray(self.counts, np.int64))[:-1]])
                           if offsets is None else np.asarray(offsets, np.int64)).astype(np.int64)
        self.local_hotn, self.local_hash = shard_sizes(int(hotn), [int(h) for h in hash_sizes], self.world)
        if backend is None:
            device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            backend = HipShardBackend(self.counts, self.dim, self.local_hotn, self.local_hash, self.offsets_np, threshold,
                                      adjust_thres, alpha, seed + 1000 * self.rank, device)
        self.backend = backend
        self.device = backend.device
        self.sketch = getattr(backend, "sketch", None)
        self.offsets = torch.from_numpy(self.offsets_np).to(self.device)
        self._fields = None

    # -- exchange helpers ------------------------------------------------------------------------
    def _a2a(self, send, send_splits, recv_splits):
        out = send.new_empty((int(sum(recv_splits)),) + tuple(send.shape[1:]))
        if send.is_cuda and self._host_staged:
            # gloo has no device all-to-all: stage through the host (single-GPU rehearsals and tests only;
            # on a node the backend is nccl = RCCL and buffers stay in HBM)
            h_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(h_out, send.contiguous().cpu(), output_split_sizes=list(recv_splits),
                                   input_split_sizes=list(send_splits), group=self.group)
            out.copy_(h_out)
            return out
        dist.all_to_all_single(out, send.contiguous(), output_split_sizes=list(recv_splits),
                               input_split_sizes=list(send_splits), group=self.group)
        return out

    def forward(self, local_ids, out=None):
        """local_ids [F, B] int64 (this rank's samples).  Returns [F, B, D]."""
        F, B = local_ids.shape
        n = F * B
        if self._fields is None or self._fields.numel() != n:
            self._fields = torch.arange(F, device=self.device, dtype=torch.int32).repeat_interleave(B)
        gids = (local_ids + self.offsets[:, None]).to(torch.int32).reshape(-1)      # sk_embedding_bag.py:103
        own = owner_of(gids, self.world)
        order = torch.argsort(own, stable=True)
        send_counts = torch.bincount(own, minlength=self.world)
        if send_counts.is_cuda and self._host_staged:
            send_counts = send_counts.cpu()
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        ss, rs = send_counts.tolist(), recv_counts.tolist()                          # split sizes live on the host
        packed = (self._fields[order].to(torch.int64) << 32) | (gids[order].to(torch.int64) & 0xFFFFFFFF)
        got = self._a2a(packed, ss, rs)
        r_gids = (got & 0xFFFFFFFF).to(torch.int32)          # wraps to the int32 bit pattern
        r_fields = (got >> 32).to(torch.int32)
        self.last_received = int(r_gids.numel())
        que, rows = self.backend.forward_gid(r_gids, r_fields)
        back = self._a2a(rows, rs, ss)
        if out is None:
            out = torch.empty(F, B, self.dim, device=self.device)
        out.view(n, self.dim)[order] = back
        self._saved = (B, order, ss, rs, r_gids, r_fields, que)
        return out

    def backward_update(self, grad_out, lr, per_field_calls=False):
        """grad_out [F, B, D].  Scores are normalised per field over the global batch (all ranks)."""
        B, order, ss, rs, r_gids, r_fields, que = self._saved
        F = self.F
        n = F * B
        grad_out = grad_out.contiguous()
        gn, sums = self.backend.grad_norm(grad_out, B)
        dist.all_reduce(sums, group=self.group)
        scores = gn.view(F, B) * float(B * self.world) / sums[:, None]              # sk_embedding_bag.py:189, global batch
        g_send = grad_out.view(n, self.dim)[order]
        s_send = scores.reshape(-1)[order]
        r_grads = self._a2a(g_send, ss, rs)
        r_scores = self._a2a(s_send, ss, rs)
        ins = self.backend.backward_gid(r_gids, r_fields, que, r_grads, r_scores, lr)
        self._last = (r_gids, r_fields, r_scores, ins)
        return scores.reshape(-1), ins
