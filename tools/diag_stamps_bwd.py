"""Diagnostic (not collected): per-workgroup phase times of k_bwd_segments on cfg3 (stamp build)."""
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
    eng.forward(ids[i % 8]); eng.backward_update(grad, 0.1, per_field_calls=False)
torch.cuda.synchronize()
buf = np.zeros(8192 * 8, np.uint64)
eng.L.cafe_debug_read_bwd.argtypes = [ctypes.c_void_p]
eng.L.cafe_debug_read_bwd(buf.ctypes.data)
b = buf.reshape(8192, 8).astype(np.int64)
b = b[b[:, 0] > 0]
t0 = b[:, 0].min()
cnt = b[:, 7]
print("blocks", len(b), "partition size mean", cnt.mean(), "max", cnt.max())
for k, name in enumerate(["load+sort", "walk"]):
    d = (b[:, k + 1] - b[:, k]) / 100.0
    print(f"{name:10s} mean {d.mean():7.2f} p50 {np.median(d):7.2f} p99 {np.percentile(d, 99):7.2f} max {d.max():7.2f}")
d = (b[:, 2] - b[:, 0]) / 100.0
st = (b[:, 0] - t0) / 100.0
print(f"block total p50 {np.median(d):.2f} max {d.max():.2f}; last start {st.max():.2f}; last end {(b[:, 2].max() - t0) / 100.0:.2f}")
for o in np.argsort(-d)[:6]:
    print("  slow: size", cnt[o], "total", d[o], "sort", (b[o, 1] - b[o, 0]) / 100.0, "start", st[o])
for lo, hi in [(0, 1), (1, 10), (10, 20), (20, 40), (40, 80)]:
    sel = (st >= lo) & (st < hi)
    if sel.sum():
        print(f"  start in [{lo},{hi}): {sel.sum()} blocks, total p50 {np.median(d[sel]):.2f}")
