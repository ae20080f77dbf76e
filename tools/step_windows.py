"""Diagnostic: the timeline of one steady-state step as the device ran it, without a profiler attached.

Needs the stamp build in cafe_amd/libcafe_hip.so:  make -C cafe_amd/csrc clean && make -C cafe_amd/csrc EXTRA=-DCAFE_PROF
Every workgroup of the step's kernels stamps the 100 MHz wall clock on entry and exit (cafe_core.h CAFE_KWIN); the
window of a kernel is [first entry, last exit] over the workgroups of its last launch.

    python tools/step_windows.py [cfg] [steps]"""
import ctypes, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from cafe_amd.config import workload
from cafe_amd.collection import CafeEmbeddingCollection
wl = workload(sys.argv[1] if len(sys.argv) > 1 else "cfg3")
F, B, D = wl.num_fields, wl.batch, wl.dim
eng = CafeEmbeddingCollection(wl.counts, D, wl.hotn, wl.hash_sizes, wl.offsets, threshold=wl.threshold, alpha=wl.alpha, seed=1)
import os
if os.environ.get("IDS_READY", "1") == "1":
    eng.sketch.set_option("ids_ready", 1)
ids = [torch.from_numpy(x).cuda() for x in wl.make_ids(123, steps=16)]
grad = torch.randn(F, B, D, device="cuda")
out = torch.empty(F, B, D, device="cuda")
L = eng.L
import os
if os.environ.get("CAFE_IDS_READY"):
    eng.sketch.set_option("ids_ready", 1)
names = {"embed": ["k_forward", "k_grad_norm", "k_migrate_list", "k_bwd_assign", "k_bwd_offsets", "k_bwd_scatter", "k_bwd_segments", "k_bwd_apply"],
         "insert": ["k_assign", "k_offsets", "k_scatter", "k_round1", "k_fix", "k_finish", "k_commit", "k_presort"]}


def read():
    res = {}
    for tu in ("embed", "insert"):
        buf = np.zeros(8 * 4096 * 2, np.uint64)
        fn = getattr(L, "cafe_debug_windows_" + tu)
        fn.argtypes = [ctypes.c_void_p]
        assert fn(buf.ctypes.data) == 0
        buf = buf.reshape(8, 4096, 2).astype(np.int64)
        for k, nm in enumerate(names[tu]):
            res[nm] = buf[k]
    return res


n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
hint = os.environ.get("HINT", "1") == "1"


def step(i):
    eng.forward(ids[i % 16], out=out)
    if hint:
        eng.hint_next(ids[(i + 1) % 16], out)
    eng.backward_update(grad, 0.1, per_field_calls=True)


import time
for i in range(n):
    step(i)
torch.cuda.synchronize()
for rep in range(3):
    # a run of steps without a host synchronisation; every kernel's stamps are those of its last launch
    t0 = time.perf_counter()
    for i in range(40):
        step(n + i)
    n += 40
    torch.cuda.synchronize()
    period = (time.perf_counter() - t0) / 40 * 1e6
    w = read()
    # origin: the norm kernel of the last step.  With hints the forward queued behind that step (and its plan) shows up
    # behind the row updates; without, the last step's own forward lies before the origin and is listed with its offset.
    gn = w["k_grad_norm"]
    origin = gn[:, 0][gn[:, 0] > 0].min()
    rows = []
    for nm, b in w.items():
        ok = (b[:, 0] >= origin - 6000) & (b[:, 1] > 0)
        if ok.any():
            rows.append((b[ok, 0].min(), b[ok, 1].max(), nm, int(ok.sum()), np.median(b[ok, 1] - b[ok, 0]) / 100.0))
    rows.sort()
    print(f"step timeline (us from the first workgroup of k_grad_norm), {wl.name}, hint={int(hint)}, period {period:.1f} us/step:")
    for s, e, nm, cnt, med in rows:
        print(f"  {nm:16s} wgs={cnt:5d} start={(s - origin) / 100:7.1f} end={(e - origin) / 100:7.1f} dur={(e - s) / 100:6.1f}  median wg {med:6.1f}")
print(eng.sketch.counters())
