"""Diagnostic: a shard-sized sketch fed a whole step's lookups in one call (what a shard owner does)."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from cafe_amd.sketch import HotSketch
from cafe_amd.config import workload
wl = workload(sys.argv[1] if len(sys.argv) > 1 else "cfg2")
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sk = HotSketch(max(10, wl.hotn // world), wl.threshold, 1, wl.alpha, 1)
rng = np.random.default_rng(1)
for step in range(12):
    ids = wl.make_ids(step, 1)[0]
    g = (ids + wl.offsets[:, None]).reshape(-1)
    keep = rng.random(g.size) < 1.0 / world * 2 if world > 2 else np.ones(g.size, bool)
    g = g[keep]
    sc = (rng.random(g.size) * 2).astype(np.float32)
    sk.insert_val(torch.from_numpy(g).cuda(), torch.from_numpy(sc).cuda())
    c = sk.counters()
    print(step, g.size, c["parallel_calls"], c["inorder_calls"], c["left_parallel_path"], c["rounds_max"], flush=True)
