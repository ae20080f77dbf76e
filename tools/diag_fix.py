"""Diagnostic (stamp build, tools/build_prof.sh): where the fix-point kernel spends its time on a churning id stream.
    python tools/diag_fix.py [drift]"""
import ctypes, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from cafe_amd.config import workload
from cafe_amd.collection import CafeEmbeddingCollection
drift = int(sys.argv[1]) if len(sys.argv) > 1 else 7919
wl = workload("cfg3")
F, B, D = wl.num_fields, wl.batch, wl.dim
eng = CafeEmbeddingCollection(wl.counts, D, wl.hotn, wl.hash_sizes, wl.offsets, threshold=wl.threshold, alpha=wl.alpha, seed=1)
ids = [torch.from_numpy(x).cuda() for x in wl.make_ids(123, steps=16)]
counts_dev = torch.tensor(wl.counts, device="cuda", dtype=torch.int64)[:, None]
grad = torch.randn(F, B, D, device="cuda")
out = torch.empty(F, B, D, device="cuda")
L = eng.L
L.cafe_debug_read.argtypes = [ctypes.c_void_p]
for i in range(1, 700):
    if drift and i % 16 == 0:
        for t in ids:
            t.add_(drift).remainder_(counts_dev); t[0].clamp_(min=1)
    eng.forward(ids[i % 16], out=out); eng.backward_update(grad, 0.1, per_field_calls=True)
    if i > 600 and i % 16 in (3, 9):
        torch.cuda.synchronize()
        buf = np.zeros(8 * 4096 * 8, np.uint64)
        L.cafe_debug_read(buf.ctypes.data)
        buf = buf.reshape(8, 4096, 8).astype(np.int64)
        for slot, name in ((6, "round 2"), (7, "round >= 3 (last)")):
            b = buf[slot]
            b = b[b[:, 0] > 0]
            if not len(b):
                continue
            t0 = b[:, 0].min()
            ok = b[:, 1] >= b[:, 0]
            main = (b[ok, 1] - b[ok, 0]) / 100.0
            line = f"step {i} {name}: {len(b)} wgs; main half mean {main.mean():6.1f} max {main.max():6.1f} us"
            ok2 = ok & (b[:, 2] >= b[:, 1])
            if ok2.any():
                line += f"; barrier done at {(b[ok2, 2].max() - t0) / 100.0:6.1f}"
            ok3 = ok2 & (b[:, 3] >= b[:, 2])
            if ok3.any():
                lru = (b[ok3, 3] - b[ok3, 2]) / 100.0
                line += f"; lru half mean {lru.mean():6.1f} max {lru.max():6.1f}; round done at {(b[ok3, 4].max() - t0) / 100.0:6.1f}"
                ran = b[ok3, 5]
                line += f"; partitions per wg: main max {(ran & 0xffff).max()} lru max {(ran >> 16).max()} (sum {(ran & 0xffff).sum()}, {(ran >> 16).sum()})"
            print(line)
        b = buf[5]
        b = b[b[:, 1] > 0]
        if len(b):
            b = b.copy(); b[:, 0] = b[:, 1]
            d = (b[:, 1:6].max(axis=1) - b[:, 0]) / 100.0
            o = np.argsort(-d)[:5]
            for k in o:
                row = b[k]
                ph = [round(float(row[j] - row[0]) / 100.0, 1) if row[j] >= row[0] else None for j in range(1, 7)]
                print(f"   slow owner (last partition of a wg): role {row[7] & 0xff} items {row[7] >> 8} phase ends {ph}")
        print(eng.sketch.counters()["rounds_total"], flush=True)
