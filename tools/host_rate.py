"""Diagnostic: is the step host-bound?  Times the host's enqueue loop (no sync inside) against the GPU's wall time."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from cafe_amd.config import workload
from cafe_amd.collection import CafeEmbeddingCollection
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
wl = workload(cfg)
F, B, D = wl.num_fields, wl.batch, wl.dim
eng = CafeEmbeddingCollection(wl.counts, D, wl.hotn, wl.hash_sizes, wl.offsets, threshold=wl.threshold, alpha=wl.alpha, seed=1)
ids = [torch.from_numpy(x).cuda() for x in wl.make_ids(123, steps=16)]
grad = torch.randn(F, B, D, device="cuda")
out = torch.empty(F, B, D, device="cuda")
for i in range(600):
    eng.forward(ids[i % 16], out=out); eng.backward_update(grad, 0.1, per_field_calls=True)
torch.cuda.synchronize()
for rep in range(3):
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        eng.forward(ids[i % 16], out=out); eng.backward_update(grad, 0.1, per_field_calls=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    import ctypes
    ht = (ctypes.c_double * 4)()
    eng.L.cafe_debug_host_times.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    eng.L.cafe_debug_host_times(eng.sketch.h, ht)
    k = max(ht[3], 1)
    print(f"   inside the library per step: forward enqueue {ht[0] / k:.1f} us, update enqueue {ht[1] / k:.1f} us, verdict wait {ht[2] / k:.1f} us")
    print(f"{cfg}: host enqueue {1e6 * (t1 - t0) / n:.1f} us/step, total {1e6 * (t2 - t0) / n:.1f} us/step, drain after loop {1e3 * (t2 - t1):.2f} ms", flush=True)
print(eng.sketch.counters())
