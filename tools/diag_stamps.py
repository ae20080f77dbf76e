"""Diagnostic (not collected): per-workgroup phase times inside the insert kernels on cfg3.
Needs the stamp build: make -C cafe_amd/csrc clean && make -C cafe_amd/csrc EXTRA=-DCAFE_PROF"""
import ctypes, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from cafe_amd.config import workload
from cafe_amd.collection import CafeEmbeddingCollection
wl = workload(sys.argv[1] if len(sys.argv) > 1 else "cfg3")
F, B, D = wl.num_fields, wl.batch, wl.dim
eng = CafeEmbeddingCollection(wl.counts, D, wl.hotn, wl.hash_sizes, wl.offsets, threshold=wl.threshold, alpha=wl.alpha, seed=1)
ids = [torch.from_numpy(x).cuda() for x in wl.make_ids(123, steps=16)]
grad = torch.randn(F, B, D, device="cuda")
out = torch.empty(F, B, D, device="cuda")
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 600):
    eng.forward(ids[i % 16], out=out); eng.hint_next(ids[(i + 1) % 16], out); eng.backward_update(grad, 0.1, per_field_calls=True)
torch.cuda.synchronize()
L = eng.L
buf = np.zeros(8 * 4096 * 8, np.uint64)
L.cafe_debug_read.argtypes = [ctypes.c_void_p]
L.cafe_debug_read(buf.ctypes.data)
buf = buf.reshape(8, 4096, 8).astype(np.int64)


def show(name, b, labels):
    b = b[b[:, 0] > 0]
    if not len(b):
        print(f"== {name}: no stamps"); return
    b = b[b[:, 0] > b[:, 0].max() - 30000]          # the latest launch only (stamps of older, larger grids stay behind)
    t0 = b[:, 0].min()
    print(f"== {name}: {len(b)} workgroups; first start -> last start {(b[:, 0].max() - t0) / 100:.2f} us")
    prev = 0
    for k, lab in labels:
        ok = b[:, k] > 0
        if not ok.any():
            continue
        d = (b[ok, k] - b[ok, prev]) / 100.0
        e = (b[ok, k] - t0) / 100.0
        print(f"   {lab:28s} n={ok.sum():5d} phase mean {d.mean():6.2f} p50 {np.median(d):6.2f} p99 {np.percentile(d, 99):6.2f} max {d.max():6.2f} | ends at p50 {np.median(e):6.2f} max {e.max():6.2f}")
        prev = k


r1 = buf[0]
role = r1[:, 7] & 0xff
tot = r1[:, 7] >> 8
for rid, rn in [(1, "hot"), (2, "lru"), (3, "main C")]:
    sel = (role == rid) & (r1[:, 0] > 0)
    if sel.any():
        print(f"-- round1 {rn}: items mean {tot[sel].mean():.0f} max {tot[sel].max()}")
        show(f"round1 {rn}", r1[sel], [(1, "sums+load"), (2, "sort"), (3, "scatter to sorted"), (4, "lane replay"), (5, "heavy/wave replay"), (6, "write-out")])
        d = (r1[sel, 6] - r1[sel, 0]) / 100.0
        o = np.argsort(-d)[:4]
        print("   slowest:", [(int(tot[sel][i]), float(d[i])) for i in o])
        for i in o[:3]:
            row = r1[sel][i]
            print("      items", int(tot[sel][i]), "phase ends (us from block start):", [round(float(row[k] - row[0]) / 100.0, 2) if row[k] > 0 else None for k in range(1, 7)])
ok = (r1[:, 0] > 0) & (r1[:, 6] > 0)
if ok.any():
    st = np.sort(r1[ok, 0]); t0 = st[0]
    ev = np.concatenate([np.stack([r1[ok, 0], np.ones(ok.sum(), np.int64)], 1), np.stack([r1[ok, 6], -np.ones(ok.sum(), np.int64)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    run = np.cumsum(ev[:, 1])
    print(f"== round1 residency: {ok.sum()} workgroups, most at once {run.max()}, kernel span {(r1[ok, 6].max() - t0) / 100:.2f} us")
    for t in (1, 2, 4, 8, 12, 16, 20, 24, 28):
        i = np.searchsorted(ev[:, 0], t0 + t * 100, side="right") - 1
        print(f"   at {t:2d} us: running {run[i]:5d}   started so far {(st <= t0 + t * 100).sum():5d}")
show("finish", buf[1], [(1, "scores out"), (2, "flags"), (3, "look-back"), (4, "ranks")])
show("commit", buf[2], [(1, "verdict"), (2, "partitions"), (3, "ticket"), (4, "close (last)")])
show("offsets", buf[3], [(1, "column sums"), (2, "prefix+ticket"), (3, "bin scans (last)"), (4, "zero+scalars (last)")])
print(eng.sketch.counters())
show("assign", buf[4], [(1, "zero LDS"), (2, "ids+probes landed"), (3, "classify+rank (wave 0)"), (4, "all waves"), (5, "wave prefix+column"), (6, "plan words out")])
