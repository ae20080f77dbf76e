"""Padding and de-duplication in front of the sketch, as the reference's graph models feed it
(TOIS_revision/PinSAGE/sketch.cpp:281-332 and :439-441): between two batch_insert_start() calls every distinct id is
inserted ONCE -- at its first occurrence, with that occurrence's score -- and every later occurrence, in the same call
or a later one, reports what the first one reported; the padding id is never inserted (ins 0) and batch_query hands it
back unchanged.  Every batch_insert / batch_insert_val call still advances the batch counter and the decay once and
makes the reset test (:291-298), whatever survives the de-duplication.

The sketch behind it is this package's HotSketch (the tricks/ generation, include/cafe_hip.h sections A-B).  The PinSAGE
file's own insert rules differ from that generation (its query also reads the LRU table, insertLRU reports slot
numbers), so this wrapper is NOT comparable with that file's outputs -- parity unpinned against it; its semantics are
checked against the oracle fed through a restatement of the loop above (tests/test_gpu_dedup_sketch.py).  The id
bookkeeping is tensor plumbing on the device; the inserts are the library's kernels."""
import torch

from .sketch import HotSketch


class DedupSketch:
    def __init__(self, sketch: HotSketch, padding_idx=-1):
        self.sketch = sketch
        self.device = sketch.device
        self.padding = int(padding_idx) & 0xFFFFFFFF
        if self.padding >= 1 << 31:
            self.padding -= 1 << 32                      # compared as the int32 bit pattern the ids are kept in
        self.batch_insert_start()

    def batch_insert_start(self):
        """:439-441: forget which ids this step has inserted."""
        self._seen = torch.empty(0, dtype=torch.int32, device=self.device)        # sorted
        self._seen_ins = torch.empty(0, dtype=torch.int32, device=self.device)

    def batch_query(self, ids):
        ids = self.sketch._ids(ids)
        que = self.sketch.query(ids)
        return torch.where(ids == self.padding, ids, que)

    def batch_insert(self, ids):
        return self.batch_insert_val(ids, None)

    def batch_insert_val(self, ids, scores=None):
        ids = self.sketch._ids(ids)
        n = ids.numel()
        if scores is not None:
            scores = torch.as_tensor(scores, dtype=torch.float32).to(self.device).contiguous()
        pos = torch.arange(n, device=self.device)
        uniq, inverse = torch.unique(ids, return_inverse=True)
        first = torch.full((uniq.numel(),), n, dtype=torch.int64, device=self.device).scatter_reduce(0, inverse, pos, "amin")
        is_first = first[inverse] == pos
        known = torch.isin(ids, self._seen) if self._seen.numel() else torch.zeros(n, dtype=torch.bool, device=self.device)
        take = is_first & ~known & (ids != self.padding)
        new_ids = ids[take]                               # first occurrences, in batch order
        new_ins = self.sketch.insert_val(new_ids, None if scores is None else scores[take]).clone()
        if new_ids.numel():
            all_ids = torch.cat([self._seen, new_ids])
            all_ins = torch.cat([self._seen_ins, new_ins])
            order = torch.argsort(all_ids)
            self._seen, self._seen_ins = all_ids[order].contiguous(), all_ins[order].contiguous()
        ins = torch.zeros(n, dtype=torch.int32, device=self.device)
        if self._seen.numel():
            at = torch.searchsorted(self._seen, ids).clamp_(max=self._seen.numel() - 1)
            hit = self._seen[at] == ids
            ins = torch.where(hit, self._seen_ins[at], ins)
        return ins
