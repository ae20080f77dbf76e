"""Diagnostic (not collected): per-owner phase times of the insert rounds on cfg3.
Needs the stamp build: make -C cafe_amd/csrc clean && make -C cafe_amd/csrc EXTRA=-DCAFE_PROF"""
import ctypes, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from cafe_amd.config import workload
from cafe_amd.collection import CafeEmbeddingCollection
wl = workload("cfg3")
F, B, D = wl.num_fields, wl.batch, wl.dim
eng = CafeEmbeddingCollection(wl.counts, D, wl.hotn, wl.hash_sizes, wl.offsets, threshold=wl.threshold, alpha=wl.alpha, seed=1)
ids = [torch.from_numpy(x).cuda() for x in wl.make_ids(123, steps=8)]
grad = torch.randn(F, B, D, device="cuda")
for i in range(60):
    eng.forward(ids[i % 8]); eng.backward_update(grad, 0.1, per_field_calls=True)
torch.cuda.synchronize()
L = eng.L
buf = np.zeros(2 * 8192 * 8, np.uint64)
L.cafe_debug_read.argtypes = [ctypes.c_void_p]
L.cafe_debug_read(buf.ctypes.data)
for t, (name, phases) in enumerate([("main", ["load+sort", "phaseA", "phaseB", "writeback"]), ("lru", ["load+sort", "replay", "writeback"])]):
    b = buf.reshape(2, 8192, 8)[t].astype(np.int64)
    b = b[b[:, 0] > 0]
    if not len(b):
        continue
    t0 = b[:, 0].min()
    nh = b[:, 7] >> 32
    tot = b[:, 7] & 0xffffffff
    print("n_heavy: mean", nh.mean(), "max", nh.max())
    n = len(phases)
    print(f"== {name}: blocks {len(b)} items mean {tot.mean():.0f} max {tot.max()}")
    for k in range(n):
        d = (b[:, k + 1] - b[:, k]) / 100.0
        print(f"{phases[k]:10s} mean {d.mean():7.2f} us  p50 {np.median(d):7.2f}  p99 {np.percentile(d, 99):7.2f}  max {d.max():7.2f}")
    if t == 0:
        d = (b[:, 5] - b[:, 1]) / 100.0
        print(f"A: start->bucket loaded (wave 0) p50 {np.median(d):.2f} mean {d.mean():.2f}")
        d = (b[:, 6] - b[:, 5]) / 100.0
        print(f"A: replay of first bucket (wave 0) p50 {np.median(d):.2f} mean {d.mean():.2f}")
        d = (b[:, 2] - b[:, 6]) / 100.0
        print(f"A: first bucket done -> barrier passed p50 {np.median(d):.2f} mean {d.mean():.2f}")
    if t == 1:
        for nm, v in [("miss-loaded -> before barrier1", b[:, 4] >> 32), ("barrier1", b[:, 4] & 0xffffffff), ("rank_of", b[:, 5] >> 32), ("barrier2 + writes + barrier3", b[:, 5] & 0xffffffff)]:
            print(f"L cycles: {nm}: p50 {np.median(v):.0f} mean {v.mean():.0f} p99 {np.percentile(v, 99):.0f}")
    if False:
        for nm, x, y in [("entry->flags read", 4, 5), ("flags->first barrier (off, ldirty)", 5, 0), ("barrier->rec loaded", 0, 6), ("rec loaded->sorted (miss gather, rank)", 6, 1)]:
            dd = (b[:, y] - b[:, x]) / 100.0
            print(f"L: {nm}: p50 {np.median(dd):.2f} mean {dd.mean():.2f} p99 {np.percentile(dd, 99):.2f}")
    d = (b[:, n] - b[:, 0]) / 100.0
    print(f"block total mean {d.mean():.2f} p50 {np.median(d):.2f} max {d.max():.2f}; first start->last start {(b[:, 0].max() - t0) / 100.0:.2f}; last end {(b[:, n].max() - t0) / 100.0:.2f}")
    for o in np.argsort(-d)[:5]:
        print("  slow: items", tot[o], "heavy", nh[o], "total", d[o], "phases", [float((b[o, k + 1] - b[o, k]) / 100.0) for k in range(n)], "start", (b[o, 0] - t0) / 100.0,
              ("A parts", float((b[o, 5] - b[o, 1]) / 100.0), float((b[o, 6] - b[o, 5]) / 100.0), float((b[o, 2] - b[o, 6]) / 100.0)) if t == 0 else "")
    st = (b[:, 0] - t0) / 100.0
    for lo, hi in [(0, 1), (1, 10), (10, 20), (20, 30), (30, 40), (40, 60)]:
        sel = (st >= lo) & (st < hi)
        if sel.sum():
            print(f"  start in [{lo},{hi}) us: {sel.sum()} blocks, items mean {tot[sel].mean():.0f}, block total p50 {np.median(d[sel]):.2f} mean {d[sel].mean():.2f}")
