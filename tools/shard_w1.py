"""Diagnostic (not collected): the sharded step's own overhead.  One rank, backend nccl, world size 1: the all-to-alls are
local copies, so the time per step is the sharded pipeline's host work and launches against the fused collection's
0.18 ms on the same workload.

    python tools/shard_w1.py [cfg] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from cafe_amd.config import workload
from cafe_amd.sharded import ShardedCafeEmbedding

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
wl = workload(sys.argv[1] if len(sys.argv) > 1 else "cfg3")
F, B, D = wl.num_fields, wl.batch, wl.dim
eng = ShardedCafeEmbedding(wl.counts, D, wl.hotn, wl.hash_sizes, wl.offsets, threshold=wl.threshold, alpha=wl.alpha, seed=1)
NB = int(os.environ.get("NB", "64"))
ids = [torch.from_numpy(x).cuda() for x in wl.make_ids(123, steps=NB)]
grad = torch.randn(F, B, D, device="cuda")
out = torch.empty(F, B, D, device="cuda")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300


def step(i):
    eng.forward(ids[i % NB], out=out)
    if os.environ.get("PREFETCH_LATE"):
        eng.backward_update(grad, 0.1)
        eng.prefetch(ids[(i + 1) % NB])
    else:
        eng.prefetch(ids[(i + 1) % NB])      # the next batch's ids are known: its route is queued before the update's verdict wait
        eng.backward_update(grad, 0.1)


for i in range(n):
    step(i)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(100):
        step(n + i)
    torch.cuda.synchronize()
    print(f"{wl.name}: sharded step at world size 1: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms/step", flush=True)
    n += 100
if os.environ.get("PROFILE_HOST"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(100):
        step(n + i)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
print(eng.backend.coll.sketch.counters())
dist.destroy_process_group()
