#!/usr/bin/env python3
"""bench.py -- throughput of the CAFE hot path on MI355X: embedding lookups/s for
forward (sketch query + hot/cold gather) + backward (scores, sketch insert, migration, sparse SGD).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3|cfg2|...]

One step = one pass of the hot path over one batch of synthetic Zipf(1.05) ids (ids, tables, sketch and
the upstream gradient are resident in HBM before the timed region).  N > 1 is launched by
torch.distributed.run, one rank per GPU; tables and sketch are hash-sharded by id and lookups travel by
RCCL all-to-all (cafe_amd/sharded.py).  Rank 0 prints ONE JSON line.

The default workload is cfg3 (BASELINE.json configs[2] tables: Criteo-Terabyte-shaped, ~880 M ids,
26 fields, D=128, compress 0.001) because that is the configuration the north-star target is quoted on
("at 1 GPU on Criteo-Terabyte-shaped synthetic data") and it fits one GPU; cfg2 is `--config cfg2`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (samples); default: the config's")
    ap.add_argument("--drift", type=int, default=0,
                    help="robustness runs: after every pass over the id batches all local ids move by this much (mod the "
                         "field size), so new ids keep crossing the threshold and the sketch keeps resetting")
    ap.add_argument("--id-batches", type=int, default=64, help="distinct synthetic id batches the steps cycle through")
    ap.add_argument("--preheat", type=int, default=1000, help="untimed steps that populate the sketch before warmup")
    ap.add_argument("--lr", type=float, default=0.1)
    ap.add_argument("--zipf-a", type=float, default=1.05, help="Zipf exponent of the id streams (<= 1: uniform ids; diagnostics)")
    ap.add_argument("--per-field-calls", action="store_true",
                    help="issue the reference's one batch_insert_val per field one by one (default: same sketch cadence, "
                         "run as one insert per step whenever no flip or reset can fall between two fields)")
    ap.add_argument("--one-call-per-step", action="store_true",
                    help="ONE batch_insert_val per step over all fields' ids (decay and reset test once per step)")
    ap.add_argument("--repetitions", type=int, default=20,
                    help="the timed window of --steps steps is run this many times back to back; the line reports the median "
                         "window (min and max beside it)")
    ap.add_argument("--ids-on-stream", action="store_true",
                    help="do NOT tell the library that a batch's ids are resident before forward is called (they are: the bench "
                         "holds every id batch in HBM before the timed region); the plan of the insert then waits for everything "
                         "queued on the stream instead of starting at once")
    ap.add_argument("--no-hint", action="store_true",
                    help="do NOT name the next step's id batch before the update (cafe_forward_hint_dev): the next gather is "
                         "then queued only after the host has seen the insert's verdict and returned")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the two extra measurements of the N = 1 line: the same step through 26 SKEmbeddingBag modules + "
                         "torch.optim.SGD (the reference driver's call order), and CafeCTRModel.train_step with a DLRM head")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


# --------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N=1 only): the reference's sketch (oracle/_ref/sklib.so, built from the reference's
# own sketch.cpp) or, where that is absent, the C port (oracle/liboracle.so), plus torch CPU ops for the
# tables, on a bounded sample of the same workload.  Runs in a child process: the reference needs
# MALLOC_PERTURB_ set at process start and keeps global state.
# --------------------------------------------------------------------------------------------------
def cpu_baseline_worker(args):
    import ctypes
    import torch
    from cafe_amd.config import workload
    wl = workload(args.config, args.batch or None)
    F, B, D = wl.num_fields, wl.batch, wl.dim
    ref_so = os.path.join(ROOT, "oracle", "_ref", "sklib.so")
    i32p, f32p = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)
    if os.path.exists(ref_so) and B <= 16384:
        kind = "reference"
        L = ctypes.CDLL(ref_so)
        L.init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
        L.batch_query.argtypes = [i32p, ctypes.c_int]; L.batch_query.restype = i32p
        L.batch_insert_val.argtypes = [i32p, f32p, ctypes.c_int]; L.batch_insert_val.restype = i32p
        devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1); os.dup2(devnull, 1)   # the reference prints
        L.init(wl.hotn, wl.threshold, 1, wl.alpha)

        def query(g):
            return np.ctypeslib.as_array(L.batch_query(g.ctypes.data_as(i32p), len(g)), shape=(len(g),)).copy()

        def insert(g, v):
            return np.ctypeslib.as_array(L.batch_insert_val(g.ctypes.data_as(i32p), v.ctypes.data_as(f32p), len(g)),
                                         shape=(len(g),)).copy()
    else:
        kind = "port"
        from oracle.cafe_oracle import Oracle
        sk = Oracle(wl.hotn, wl.threshold, 1, wl.alpha, 1)
        saved = None

        def query(g):
            return sk.query(g)

        def insert(g, v):
            return sk.insert_val(g, v)

    torch.manual_seed(0)
    w_hot = torch.zeros(wl.hotn, D)
    w_cold = [(torch.rand(h, D) * 2 - 1) * float(np.sqrt(1.0 / n)) for n, h in zip(wl.counts, wl.hash_sizes)]
    ids_all = wl.make_ids(123, steps=4)
    grad = torch.randn(F, B, D)
    lr = args.lr
    t0 = time.perf_counter()
    steps = 0
    sk_time = 0.0
    while True:
        ids = ids_all[steps % len(ids_all)]
        # forward (sk_embedding_bag.py:119-152)
        gids, ques, outs = [], [], []
        for f in range(F):
            g = np.ascontiguousarray((ids[f] + wl.offsets[f]).astype(np.int32))
            ts = time.perf_counter(); q = query(g); sk_time += time.perf_counter() - ts
            qt = torch.from_numpy(q.astype(np.int64))
            hot = qt < 0
            dic = qt.abs()
            out = torch.where(hot[:, None], w_hot[dic % wl.hotn], w_cold[f][dic % wl.hash_sizes[f]])
            gids.append(g); ques.append((hot, dic)); outs.append(out)
        # backward: scores, insert, migration (sk_embedding_bag.py:173-208), then SGD (dlrm_s_pytorch.py:1658)
        for f in range(F):
            gn = torch.linalg.vector_norm(grad[f], dim=1)
            sc = (gn * B / gn.sum()).numpy()
            ts = time.perf_counter(); ins = insert(gids[f], np.ascontiguousarray(sc, np.float32)); sk_time += time.perf_counter() - ts
            idx = np.nonzero(ins)[0]
            if len(idx):
                qa = query(gids[f])
                for i in idx:
                    if qa[i] < 0:
                        w_hot[-int(qa[i])] = w_cold[f][int(ids[f][i]) % wl.hash_sizes[f]]
        for f in range(F):
            hot, dic = ques[f]
            w_hot.index_add_(0, (dic[hot] % wl.hotn), grad[f][hot], alpha=-lr)
            w_cold[f].index_add_(0, (dic[~hot] % wl.hash_sizes[f]), grad[f][~hot], alpha=-lr)
        steps += 1
        if time.perf_counter() - t0 > args.cpu_seconds and steps >= 2:
            break
    dt = time.perf_counter() - t0
    if saved is not None:
        os.dup2(saved, 1)
    print(json.dumps({"value": steps * F * B / dt, "unit": "lookups/s", "cores": torch.get_num_threads(), "kind": kind,
                      "sample": f"{steps} steps of {wl.name} (batch {B}, {F} fields, D={D}) in {dt:.1f} s; sketch calls "
                                f"single-threaded ({sk_time / dt * 100:.0f}% of the time), table ops torch CPU"}))


def run_cpu_baseline(args):
    env = dict(os.environ, MALLOC_PERTURB_="165")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--config", args.config,
           "--cpu-seconds", str(args.cpu_seconds), "--lr", str(args.lr)]
    if args.batch:
        cmd += ["--batch", str(args.batch)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.cpu_seconds * 6 + 120)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:          # the baseline is a reported number, never a reason to lose the GPU result
        return {"value": None, "unit": "lookups/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"[:200]}


# --------------------------------------------------------------------------------------------------
# Extra keys of the N = 1 line (not `value`): what the same workload costs (a) through the drop-in module path -- one
# SKEmbeddingBag per field, torch sparse gradients, torch.optim.SGD, in the call order of dlrm_s_pytorch.py:613-659
# (apply_emb), :734-738 (insert_grad per field), :1658 (optimizer.step) -- and (b) as a whole training step of the thin DLRM
# driver (CafeCTRModel.train_step: bottom MLP, dot interaction, top MLP, BCE loss in the loop), with and without the next
# step's ids named ahead.  Short runs on a sketch that is still filling: orders of magnitude, not tuned figures.
# --------------------------------------------------------------------------------------------------
def run_extras(args, wl, ids_dev, grad):
    import torch
    from cafe_amd.sketch import HotSketch
    from cafe_amd.sk_embedding_bag import SKEmbeddingBag
    from cafe_amd.ctr_models import CafeCTRModel
    F, B, D = wl.num_fields, wl.batch, wl.dim
    out = {}
    nb = len(ids_dev)

    def timed(fn, warm, steps):
        for i in range(warm):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(warm + i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    try:
        lib = HotSketch(wl.hotn, wl.threshold, 1, wl.alpha, 1)
        w_high = torch.nn.Parameter(torch.zeros(wl.hotn, D, device="cuda"))
        mods = [SKEmbeddingBag(f, wl.hotn, lib, w_high, "cuda", wl.counts[f], D, int(wl.offsets[f]), wl.hash_sizes[f]).cuda()
                for f in range(F)]
        opt = torch.optim.SGD([w_high] + [m.weight_hash for m in mods], lr=args.lr)
        offs = torch.arange(B, device="cuda")
        gs = [grad[f] for f in range(F)]

        def mstep(i):
            ids = ids_dev[i % nb]
            opt.zero_grad(set_to_none=True)
            outs = [m(ids[f], offs) for f, m in enumerate(mods)]
            torch.autograd.backward(outs, gs)
            for f, m in enumerate(mods):
                m.insert_grad(ids[f])
            opt.step()
        dt = timed(mstep, 60, 40)
        out["module_path"] = {"what": f"{F} SKEmbeddingBag modules + torch.optim.SGD, one sketch call per field "
                                      "(dlrm_s_pytorch.py:613-659, 734-738, 1658)",
                              "ms_per_step": round(dt * 1e3, 3), "lookups_per_s": round(F * B / dt, 1)}
        del mods, opt, w_high
        lib.close()
    except Exception as e:          # extras never cost the main result
        out["module_path"] = {"error": repr(e)[:200]}
    try:
        # bench/criteo_terabyte.sh:20-29 and bench/criteo_kaggle.sh:20-28 (the top MLP's input width is the model's to fill in)
        ln_bot, ln_top = ([13, 512, 256, D], [0, 1024, 1024, 512, 256, 1]) if D >= 64 else ([13, 512, 256, 64, D], [0, 512, 256, 1])
        model = CafeCTRModel("dlrm", wl.counts, D, 13, wl.compress_rate, wl.hash_rate, wl.threshold, 1, wl.alpha,
                             ln_bot=ln_bot, ln_top=ln_top)
        opt = torch.optim.SGD(model.dense_parameters(), lr=args.lr)
        dense = torch.rand(B, 13, device="cuda")
        labels = (torch.rand(B, device="cuda") < 0.25).float()
        if labels.dim() != 1:
            labels = labels.reshape(-1)

        def estep(i, hint):
            model.train_step(dense, ids_dev[i % nb], labels, opt, args.lr, next_ids=ids_dev[(i + 1) % nb] if hint else None)
        lab_shape = model.forward(dense, ids_dev[0]).shape
        labels = labels.reshape(lab_shape)
        dt0 = timed(lambda i: estep(i, False), 100, 50)
        dt1 = timed(lambda i: estep(i, True), 20, 50)
        out["e2e_dlrm_train_step"] = {"what": f"CafeCTRModel('dlrm') bot {'-'.join(map(str, ln_bot))}, top {'-'.join(map(str, ln_top[1:]))}, "
                                              f"batch {B}: forward, BCE loss, backward, fused embedding update, SGD on the head",
                                      "ms_per_step": round(dt0 * 1e3, 3), "ms_per_step_next_ids_named": round(dt1 * 1e3, 3),
                                      "lookups_per_s": round(F * B / min(dt0, dt1), 1)}
    except Exception as e:
        out["e2e_dlrm_train_step"] = {"error": repr(e)[:200]}
    return out


def main():
    args = parse()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args)
        return
    import torch
    import torch.distributed as dist
    from cafe_amd.config import workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # CAFE_BENCH_REHEARSE=1: all ranks share cuda:0 and talk over gloo, to rehearse the N > 1 code path on a
    # one-GPU box (the number it prints is not a scaling measurement)
    rehearse = os.environ.get("CAFE_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    wl = workload(args.config, args.batch or None)
    F, B, D = wl.num_fields, wl.batch, wl.dim
    cadence = "loop" if args.per_field_calls else (False if args.one_call_per_step else True)

    if world == 1:
        from cafe_amd.collection import CafeEmbeddingCollection
        eng = CafeEmbeddingCollection(wl.counts, D, wl.hotn, wl.hash_sizes, wl.offsets, threshold=wl.threshold,
                                      alpha=wl.alpha, seed=1)
    else:
        from cafe_amd.sharded import ShardedCafeEmbedding
        eng = ShardedCafeEmbedding(wl.counts, D, wl.hotn, wl.hash_sizes, wl.offsets, threshold=wl.threshold,
                                   alpha=wl.alpha, seed=1)

    if world == 1 and not args.ids_on_stream:
        eng.sketch.set_option("ids_ready", 1)
    nb = args.id_batches
    ids_host = wl.make_ids(123 + rank, steps=nb, a=args.zipf_a)
    ids_dev = [torch.from_numpy(ids_host[i]).cuda() for i in range(nb)]
    grad = torch.randn(F, B, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5 + rank))
    out = torch.empty(F, B, D, device="cuda")

    counts_dev = torch.tensor(wl.counts, device="cuda", dtype=torch.int64)[:, None]
    use_hint = world == 1 and not args.no_hint

    def step(i, timed_idx=None):
        if args.drift and i and i % nb == 0:
            for t in ids_dev:
                t.add_(args.drift).remainder_(counts_dev)
                t[0].clamp_(min=1)
        eng.forward(ids_dev[i % nb], out=out)
        if use_hint and not (args.drift and (i + 1) % nb == 0):
            eng.hint_next(ids_dev[(i + 1) % nb], out)      # a data loader knows its next batch: name it before the update
        if world > 1 and not (args.drift and (i + 1) % nb == 0):
            # the next batch's ids are known: their route (owners, order, counts exchange) is queued one step ahead, before
            # the update waits for the device's verdict
            eng.prefetch(ids_dev[(i + 1) % nb])
        eng.backward_update(grad, args.lr, per_field_calls=cadence)

    it = 0
    for _ in range(args.preheat + args.warmup):
        step(it); it += 1

    def window():
        """EXACTLY --steps steps between barrier + synchronize on both sides; max over ranks."""
        nonlocal it
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(it, k); it += 1
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        w = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([w], device="cpu" if rehearse else "cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w = float(t.item())
        return w

    # the driver's --steps 20 is a ~4 ms window: one window is at the mercy of a single slow step, so the window is
    # repeated and the median reported (value, ms_per_step), min and max beside it
    reps = max(1, args.repetitions)
    windows = sorted(window() for _ in range(reps))
    dt = windows[len(windows) // 2] if reps % 2 else 0.5 * (windows[reps // 2 - 1] + windows[reps // 2])

    # roofline: the library records HIP events around its kernels on the stream they are launched on (timing mode);
    # measured in a few extra steps behind the timed region.  [0] k_forward, [1] k_bwd_segments, [2] the
    # insert chain (k_round1 .. k_commit), [3] k_grad_norm, [4] k_bwd_apply (the second kernel of the update when a step
    # runs it in two: the segment sums on the library's third stream beside the insert, the row updates behind the
    # migration; [1] + [4] is then the update's launch time)
    import ctypes
    coll = eng if world == 1 else eng.backend.coll
    L = coll.L
    kern = None
    if world == 1:
        L.cafe_tables_set_timing(coll.tables, 1)
        acc = [0.0, 0.0, 0.0, 0.0, 0.0]
        cnt = [0, 0, 0, 0, 0]
        kreps = 20
        ms = ctypes.c_float()
        for _ in range(kreps):
            step(it); it += 1
            for k in range(5):
                # a timer the last step did not record is refused ([4] only exists for split updates): each average is over
                # the steps that really ran that kernel
                if L.cafe_tables_kernel_ms(coll.tables, k, ctypes.byref(ms)) == 0:
                    acc[k] += ms.value; cnt[k] += 1
        L.cafe_tables_set_timing(coll.tables, 0)
        kern = [a / max(c, 1) for a, c in zip(acc, cnt)]
        # An event pair costs time of its own (two marker packets: the second completes a few us after the first even with
        # nothing between them).  Measured here on the same stream, median of 50 empty pairs behind a kernel, and taken
        # off every bracket: what remains is the kernel, as rocprofv3 --kernel-trace reports it (profiles/).
        pairs = []
        for _ in range(50):
            out.add_(0.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()
            torch.cuda.synchronize()
            pairs.append(e0.elapsed_time(e1))
        pair_ms = sorted(pairs)[len(pairs) // 2]
        kern = [max(k - pair_ms, 1e-6) if c else k for k, c in zip(kern, cnt)]
        split_update = cnt[4] > kreps // 2
        update_parts = {"k_bwd_segments_us": round(kern[1] * 1e3, 2), "k_bwd_apply_us": round(kern[4] * 1e3, 2)}
        kern[1] += kern[4]
    per_gpu = None
    if world > 1:
        # every rank times the two bandwidth kernels of its shard (the lookups routed to it); gathered on rank 0
        L.cafe_tables_set_timing(coll.tables, 1)
        acc = [0.0, 0.0]
        kreps = 10
        ms = ctypes.c_float()
        for _ in range(kreps):
            step(it); it += 1
            for k in range(2):
                L.cafe_tables_kernel_ms(coll.tables, k, ctypes.byref(ms))
                acc[k] += ms.value
        L.cafe_tables_set_timing(coll.tables, 0)
        mine = torch.tensor([a / kreps for a in acc] + [float(eng.last_received)], dtype=torch.float64,
                            device="cpu" if rehearse else "cuda")
        allk = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allk, mine)
        per_gpu = [[float(v) for v in t.cpu()] for t in allk]

    if rank == 0:
        lookups = F * B * world * args.steps
        value = lookups / dt
        # algorithmic bytes per lookup (no credit for duplicate ids):
        #   k_forward        id 4 + bucket probe 52 + selected row 4D + output row 4D
        #   k_bwd_segments   output gradient 4D + table row read-modify-write 8D
        #   insert chain     104: id 4, score 4, main bucket 64 read + write back 32 (k_round1 .. k_commit)
        #   k_grad_norm      output gradient 4D + norm 4: a SECOND read of the gradient, outside the 160 + 20 D minimum
        names = ["k_forward (fused sketch query + hot/cold row gather)",
                 "k_bwd_segments + k_bwd_apply (segment sums beside the insert, then SGD row updates)" if world == 1 and split_update
                 else "k_bwd_segments (segment sums + SGD row update)",
                 "insert chain (k_round1, k_fix, k_finish, k_commit)", "k_grad_norm (scores, one pass over the gradient)"]
        per_lookup = [56 + 8 * D, 12 * D, 104, 4 * D + 4]
        roof = None
        if world == 1:
            traffic = None
            tp = os.path.join(ROOT, "profiles", "traffic.json")       # PMC FETCH_SIZE/WRITE_SIZE of the same command
            tj = json.load(open(tp)) if os.path.exists(tp) else {}
            if tj.get("workload") != f"{wl.name}/b{B}":
                tj = {}

            def entry(k):
                ach = F * B * per_lookup[k] / (kern[k] * 1e-3) / 1e9
                return {"kernel": names[k], "avg_launch_us": round(kern[k] * 1e3, 2), "algorithmic_bytes_per_launch": F * B * per_lookup[k],
                        "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": tj.get(names[k].split(" ")[0])}
            dom = 0 if kern[0] >= kern[1] else 1
            e = entry(dom)
            if dom == 1 and split_update:
                e["kernel"] += f" {update_parts}"
            roof = {"bound": "hbm", "kernel": e["kernel"], "achieved": e["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": e["frac"], "traffic": e["traffic"], "avg_launch_us": e["avg_launch_us"],
                    "algorithmic_bytes_per_launch": e["algorithmic_bytes_per_launch"],
                    "other_kernel": entry(1 - dom),
                    # the step's real bottleneck is latency, not bandwidth: the insert chain moves ~22 MB in tens of us
                    "insert_chain": entry(2), "grad_norm": entry(3),
                    "event_pair_overhead_us": round(pair_ms * 1e3, 2),
                    "whole_step_frac": round(value * wl.bytes_per_lookup() / 1e9 / HBM_PEAK_GBS, 4)}
        else:
            gpus = []
            for r, (k0, k1, nrecv) in enumerate(per_gpu):
                gpus.append({"rank": r, "lookups_owned_per_step": int(nrecv),
                             "k_forward_us": round(k0 * 1e3, 2), "k_forward_GBs": round(nrecv * per_lookup[0] / (k0 * 1e-3) / 1e9, 1),
                             "k_bwd_segments_us": round(k1 * 1e3, 2),
                             "k_bwd_segments_GBs": round(nrecv * per_lookup[1] / (k1 * 1e-3) / 1e9, 1)})
            worst = max(gpus, key=lambda g: g["k_bwd_segments_us"] + g["k_forward_us"])
            dom = "k_forward" if worst["k_forward_us"] >= worst["k_bwd_segments_us"] else "k_bwd_segments"
            roof = {"bound": "hbm", "kernel": names[0 if dom == "k_forward" else 1] + f", slowest rank {worst['rank']}",
                    "achieved": worst[dom + "_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(worst[dom + "_GBs"] / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_us": worst[dom + "_us"],
                    "per_gpu": gpus,
                    "whole_step_frac": round(value * wl.bytes_per_lookup() / 1e9 / (HBM_PEAK_GBS * world), 4)}
        res = {
            "metric": "embedding lookups/sec (fwd+bwd+sketch)", "value": round(value, 1), "unit": "lookups/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32+i32",
            "data": "synthetic Zipf(1.05) ids, random-init tables, random upstream gradient",
            "config": {"workload": f"{wl.name}: {F} fields, {sum(wl.counts)} ids, D={D}, batch {B}/GPU, hotn {wl.hotn}, "
                                   f"compress {wl.compress_rate}, hash-rate {wl.hash_rate}, threshold {wl.threshold}",
                       "sketch_calls": ("one per step and shard (SURVEY 8e: a shard equals an independent sketch fed its "
                                        "sub-stream)" if world > 1 else
                                        {True: "one per field (reference cadence), run as one insert per step",
                                         "loop": "one per field, issued one by one", False: "one per step"}[cadence]),
                       "parallelism": "single GPU" if world == 1 else f"id-hash sharded x{world}, RCCL all-to-all",
                       # N = 1: every step waits once on the host for the device's verdict on the insert (polling a pinned
                       # snapshot); with next_ids_hint the next gather is queued before that wait
                       "host_syncs_in_step": 1 if world == 1 else getattr(eng, "host_syncs_in_step", 0),
                       "next_ids_hint": int(use_hint),
                       "host_waits_for_split_sizes": ({"count": eng.host_syncs_total, "ms_total": round(eng.host_wait_ms, 3)}
                                                      if world > 1 else None),
                       "ids_ready": int(world == 1 and not args.ids_on_stream),
                       "preheat_steps": args.preheat, "id_batches": nb, "drift": args.drift, "repetitions": reps,
                       "cadence_miss": (eng.sketch.counters()["cadence_miss"] if getattr(eng, "sketch", None) else None)},
            "window_ms": {"median": round(dt * 1e3, 4), "min": round(windows[0] * 1e3, 4), "max": round(windows[-1] * 1e3, 4),
                          "repetitions": reps, "steps_per_window": args.steps},
            "sketch": eng.sketch.counters() if hasattr(eng, "sketch") else None,
        }
        if roof:
            res["roofline"] = roof
        if world == 1 and not args.no_extras:
            res["extras"] = run_extras(args, wl, ids_dev, grad)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = run_cpu_baseline(args)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
