"""Diagnostic (not collected): time the sketch insert/query at cfg3 scale."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from cafe_amd.sketch import HotSketch

lim = 563200; n = 26 * 8192
rng = np.random.default_rng(0)
sk = HotSketch(lim, 500, 1, 1.0000005, 1)
batches = []
for b in range(12):
    ids = ((rng.zipf(1.05, size=n) - 1) % 880_000_000 + 1).astype(np.int64)
    batches.append((torch.from_numpy((ids & 0xFFFFFFFF).astype(np.uint32).view(np.int32)).cuda(),
                    torch.rand(n, device="cuda") * 2))
ins = torch.empty(n, dtype=torch.int32, device="cuda")
que = torch.empty(n, dtype=torch.int32, device="cuda")
for thr_note in ("warm",):
    for it in range(3):
        for ids, sc in batches:
            sk.insert_val(ids, sc, out=ins)
torch.cuda.synchronize()
print("counters after warm", sk.counters(), flush=True)
ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
ev0.record()
for ids, sc in batches:
    sk.insert_val(ids, sc, out=ins)
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / len(batches)
print(f"insert: {ms*1000:.1f} us per batch of {n} -> {n/ms/1e6:.3f} G ids/s", flush=True)
ev0.record()
for ids, sc in batches:
    sk.query(ids, out=que)
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / len(batches)
print(f"query: {ms*1000:.1f} us per batch -> {n/ms/1e6:.3f} G ids/s", flush=True)
print(sk.counters())
