"""Multi-GPU CAFE embedding: tables and sketch hash-sharded by id, one process per GPU.

The reference is single-device; this is the MI355X-native scale-out of the same hot path
(SURVEY.md 8(e)).  Every global id has one owner rank, owner = mix(gid) mod G.  The owner holds
    * an independent HotSketch with lim = hotn / G,
    * hotn / G hot rows,
    * for every field ceil(hash_size_f / G) cold rows, addressed gid mod that size,
so a shard is, by itself, exactly the single-GPU layer fed the sub-stream of lookups routed to it (in source
rank order, then original order) with ONE batch_insert_val per step -- which is how parity is defined and tested
per shard.

Per step (data-parallel batch, B samples per rank):
    route     owner of every lookup, stable order by owner, counts exchanged, ids exchanged.  It depends on the ids
              only, so `prefetch(next_ids)` queues its first half (owners, order, counts exchanged and copied to pinned
              host memory) one step ahead on the exchange stream, WITHOUT making that stream wait for the compute stream
              and without blocking the host.  torch.distributed wants the split sizes on the host: they are read at the
              next forward, from a transfer that finished during the previous step; that wait (normally zero) and the
              batches that were not prefetched at all are counted (host_syncs_in_step, host_wait_ms).
    forward   owners run the fused query+gather -> all_to_all(rows) back
    backward  per-field sums of gradient norms all-reduced (scores are normalised over the GLOBAL batch)
              -> ONE all_to_all of rows [gradient row | score | 3 pad floats] -> owners run insert / migration /
              backward+SGD straight from the received buffer (strided rows)
All-to-all is the collective that fits xGMI: every pair of GPUs has its own link, so the G-1 transfers of a rank
run concurrently (a ring all-reduce would be bound by one link).  The collectives run on a stream of their own,
ordered against the compute stream by events; forward_async() hands back a handle so that the caller's bottom MLP
runs on the compute stream while the rows travel.

The exchange logic below is plain torch + torch.distributed and also runs on CPU tensors over gloo (tests);
the owner-side compute is a backend object: HipShardBackend (libcafe_hip.so) in production."""
import contextlib
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _lib

_MIX = 2654435761
PAD = 4          # floats behind a gradient row in the packed exchange buffer: [score, 0, 0, 0] keeps rows 16-byte aligned


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

    def backward_gid(self, gids, fields, que, packed, lr):
        """packed: [n, dim + PAD] float32, a lookup's gradient row with its score at column dim (read in place).  The
        fused owner step: plan (made beside forward_gid), rounds, the device's verdict, migration, split or one-kernel
        update -- the path the single-GPU collection runs (cafe_backward_update_gid_dev)."""
        n = gids.numel()
        ins = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        _lib.check(self.L.cafe_backward_update_gid_dev(self.sketch.h, self.coll.tables, self._p(gids), self._p(fields),
                                                       self._p(que), self._p(packed), packed.shape[1], n, ctypes.c_float(lr),
                                                       self._p(ins), self._st()), "cafe_backward_update_gid_dev")
        return ins[:n]

    def pack(self, grad, gn, sums, scale, batch, order):
        """Send buffer of the backward exchange, one launch: rows [gradient | score | 0 0 0] in destination order."""
        n = order.numel()
        send = torch.empty(n, self.dim + PAD, device=self.device)
        scores = torch.empty(n, device=self.device)
        _lib.check(self.L.cafe_pack_rows_dev(self._p(grad), self._p(gn), self._p(sums), ctypes.c_float(scale), batch,
                                             self._p(order), n, self.dim, self._p(send), self._p(scores), self._st()),
                   "cafe_pack_rows_dev")
        return send, scores

    def route(self, local_ids, offsets, world):
        """Owner of every lookup, stable order by owner, ids packed for the exchange, lookups per owner: three small
        launches (cafe_route_dev) instead of a device-wide sort and a dozen elementwise kernels."""
        F, B = local_ids.shape
        n = F * B
        words = int(self.L.cafe_route_ws_words(n, world))
        ws = getattr(self, "_route_ws", None)
        if ws is None or ws.numel() < words:
            ws = self._route_ws = torch.empty(words, dtype=torch.int32, device=self.device)
        order = torch.empty(n, dtype=torch.int64, device=self.device)
        packed = torch.empty(n, dtype=torch.int64, device=self.device)
        counts = torch.empty(world, dtype=torch.int64, device=self.device)
        _lib.check(self.L.cafe_route_dev(self._p(local_ids), self._p(offsets), F, B, world, self._p(ws), self._p(order),
                                         self._p(packed), self._p(counts), self._st()), "cafe_route_dev")
        ws.record_stream(torch.cuda.current_stream())
        return order, packed, counts

    def split_ids(self, packed):
        n = packed.numel()
        gids = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        fields = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        _lib.check(self.L.cafe_split_ids_dev(self._p(packed), n, self._p(gids), self._p(fields), self._st()), "cafe_split_ids_dev")
        return gids[:n], fields[:n]

    def unpack(self, back, order, out):
        """out[order[j]] = back[j], one launch."""
        _lib.check(self.L.cafe_unpack_rows_dev(self._p(back), self._p(order), order.numel(), self.dim, self._p(out), self._st()),
                   "cafe_unpack_rows_dev")


class _Route:
    """Where the lookups of one batch go: computed from the ids alone (possibly one step ahead)."""
    __slots__ = ("ids", "B", "order", "ss", "rs", "r_gids", "r_fields", "ready", "counts_host", "counts_ev", "packed")


class _Pending:
    """Handle of forward_async(): wait() orders the compute stream behind the returning rows and hands back `out`."""

    def __init__(self, out, event):
        self.out, self.event = out, event

    def wait(self):
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
        return self.out


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
        self._route = None
        self.last_received = 0
        self.host_syncs_in_step = 0        # batches routed inside their own step (not prefetched): the step waits for the host
        self.host_syncs_total = 0          # every time the host waited for split sizes, prefetched batches included
        self.host_wait_ms = 0.0            # ... and how long in total
        # collectives on a stream of their own (device tensors only; CPU tensors over gloo run inline)
        self._xs = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    # -- streams ---------------------------------------------------------------------------------
    def _on_exchange_stream(self):
        """Context: work is queued on the exchange stream, behind everything queued on the compute stream so far."""
        if self._xs is None:
            return contextlib.nullcontext()
        self._xs.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self._xs)

    def _join(self):
        """The compute stream continues behind the exchange stream's work."""
        if self._xs is not None:
            torch.cuda.current_stream().wait_stream(self._xs)

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

    def _route_begin(self, local_ids):
        """First half of the route, all queued on the current stream, nothing waited for: owner of every lookup, stable
        order by owner, counts exchanged, both count vectors on their way to pinned host memory."""
        F, B = local_ids.shape
        n = F * B
        if hasattr(self.backend, "route") and local_ids.is_cuda:
            order, packed, send_counts = self.backend.route(local_ids.contiguous(), self.offsets, self.world)
        else:
            if self._fields is None or self._fields.numel() != n:
                self._fields = torch.arange(F, device=self.device, dtype=torch.int32).repeat_interleave(B)
            gids = (local_ids + self.offsets[:, None]).to(torch.int32).reshape(-1)      # sk_embedding_bag.py:103
            own = owner_of(gids, self.world)
            order = torch.argsort(own, stable=True)
            send_counts = torch.bincount(own, minlength=self.world)
            packed = (self._fields[order].to(torch.int64) << 32) | (gids[order].to(torch.int64) & 0xFFFFFFFF)
        if send_counts.is_cuda and self._host_staged:
            send_counts = send_counts.cpu()
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        r = _Route()
        r.ids, r.B, r.order = local_ids, B, order
        r.packed = packed
        r.counts_ev, r.ss, r.rs, r.ready, r.r_gids, r.r_fields = None, None, None, None, None, None
        if send_counts.is_cuda:
            r.counts_host = torch.empty(2, self.world, dtype=send_counts.dtype, pin_memory=True)
            r.counts_host[0].copy_(send_counts, non_blocking=True); r.counts_host[1].copy_(recv_counts, non_blocking=True)
            r.counts_ev = torch.cuda.Event(); r.counts_ev.record()
        else:
            r.ss, r.rs = send_counts.tolist(), recv_counts.tolist()
        return r

    def _route_finish(self, r):
        """Second half: the split sizes on the host (the one wait of the route; counted), then the ids to their owners."""
        if r.ss is None:
            import time
            t0 = time.perf_counter()
            r.counts_ev.synchronize()
            self.host_wait_ms += (time.perf_counter() - t0) * 1e3
            self.host_syncs_total += 1
            r.ss, r.rs = r.counts_host[0].tolist(), r.counts_host[1].tolist()
        got = self._a2a(r.packed, r.ss, r.rs)
        if hasattr(self.backend, "split_ids") and got.is_cuda:
            r.r_gids, r.r_fields = self.backend.split_ids(got)
        else:
            r.r_gids = (got & 0xFFFFFFFF).to(torch.int32)          # wraps to the int32 bit pattern
            r.r_fields = (got >> 32).to(torch.int32)
        if self._xs is not None:
            r.ready = torch.cuda.Event(); r.ready.record()
        return r

    def prefetch(self, local_ids, ids_ready=True):
        """Start the route of the NEXT batch now, on the exchange stream (the ids of a training step are known before
        the step).  ids_ready (default): the id tensor is complete in memory -- a data loader's batch -- so the exchange
        stream does not wait for the compute stream and the route runs beside the current step; False: it is the output
        of work queued on the compute stream.  The host is not blocked here; the next forward(local_ids) of the same
        tensor picks the route up."""
        if self._xs is None:
            self._route = self._route_begin(local_ids)
            return
        if not ids_ready:
            self._xs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._xs):
            self._route = self._route_begin(local_ids)

    # -- the step ----------------------------------------------------------------------------------
    def forward_async(self, local_ids, out=None):
        """local_ids [F, B] int64 (this rank's samples).  Returns a handle; handle.wait() -> [F, B, D]."""
        F, B = local_ids.shape
        n = F * B
        r = self._route
        if r is None or r.ids is not local_ids:
            self.host_syncs_in_step += 1
            r = self._route_finish(self._route_begin(local_ids))
        elif r.r_gids is None:
            if self._xs is not None:
                with torch.cuda.stream(self._xs):
                    self._route_finish(r)
                r.packed.record_stream(self._xs)
            else:
                self._route_finish(r)
        if r.ready is not None:
            torch.cuda.current_stream().wait_event(r.ready)
        self._route = None
        self.last_received = int(r.r_gids.numel())
        que, rows = self.backend.forward_gid(r.r_gids, r.r_fields)
        if out is None:
            out = torch.empty(F, B, self.dim, device=self.device)
        with self._on_exchange_stream():
            back = self._a2a(rows, r.rs, r.ss)
            if hasattr(self.backend, "unpack"):
                self.backend.unpack(back, r.order, out)
            else:
                out.view(n, self.dim)[r.order] = back
            ev = None
            if self._xs is not None:
                ev = torch.cuda.Event(); ev.record()
                for t in (rows, back, out):
                    t.record_stream(self._xs)
        self._saved = (B, r.order, r.ss, r.rs, r.r_gids, r.r_fields, que)
        return _Pending(out, ev)

    def forward(self, local_ids, out=None):
        return self.forward_async(local_ids, out).wait()

    def backward_update(self, grad_out, lr, per_field_calls=False):
        """grad_out [F, B, D].  Scores are normalised per field over the global batch (all ranks); every shard's sketch
        sees ONE batch_insert_val per step over the lookups routed to it."""
        B, order, ss, rs, r_gids, r_fields, que = self._saved
        F, D = self.F, self.dim
        n = F * B
        grad_out = grad_out.contiguous()
        gn, sums = self.backend.grad_norm(grad_out, B)
        with self._on_exchange_stream():
            dist.all_reduce(sums, group=self.group)
            if hasattr(self.backend, "pack"):
                send, scores = self.backend.pack(grad_out, gn, sums, float(B * self.world), B, order)
            else:
                scores = gn.view(F, B) * float(B * self.world) / sums[:, None]          # sk_embedding_bag.py:189, global batch
                send = torch.zeros(n, D + PAD, device=self.device)
                send[:, :D] = grad_out.view(n, D)[order]
                send[:, D] = scores.reshape(-1)[order]
            packed = self._a2a(send, ss, rs)
            if self._xs is not None:
                for t in (grad_out, gn, sums, scores, send, packed):
                    t.record_stream(self._xs)
        self._join()
        # A route queued by prefetch() is finished here (split sizes read, ids sent to their owners on the exchange stream)
        # before the owner's step makes the host wait for the device's verdict: the next forward finds its ids delivered.
        nxt = self._route
        if nxt is not None and self._xs is not None and nxt.r_gids is None:
            with torch.cuda.stream(self._xs):
                self._route_finish(nxt)
            nxt.packed.record_stream(self._xs)
        ins = self.backend.backward_gid(r_gids, r_fields, que, packed, lr)
        self._last = (r_gids, r_fields, packed[:, D], ins)
        return scores.reshape(-1), ins
