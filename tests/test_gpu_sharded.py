"""GPU parity of the multi-GPU path's owner side (the *_gid_dev entry points of include/cafe_hip.h, which take
lookups in arbitrary order with an explicit field per lookup) and of the whole sharded step with two ranks.

Only one GPU is available to the tests, so the two ranks share cuda:0 and exchange through gloo (staged via the
host); on a node the same code runs over RCCL.  Parity is per shard: every shard must equal
an independent oracle sketch fed the sub-stream routed to it."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-5


def _close(a, b, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= RTOL, f"{what}: max rel err {err:.3e}"


@pytest.mark.parametrize("dim", [16, 128])
def test_gid_entry_points_match_oracle(dim, oracle_mod):
    """forward_gid / insert / migrate_gid / backward_sgd_gid on shuffled lookups against the CPU restatement."""
    from cafe_amd.sharded import HipShardBackend
    from test_sharded_gloo import OracleShardBackend
    counts, hash_sizes, hotn, thr, lr = [3000, 40, 50000, 9], [37, 2, 211, 1], 500, 3, 0.1
    F = len(counts)
    offsets = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
    dev = torch.device("cuda", 0)
    hip = HipShardBackend(counts, dim, hotn, hash_sizes, offsets, thr, 1, 1.0000005, 5, dev)
    cpu = OracleShardBackend(oracle_mod, counts, dim, hotn, hash_sizes, offsets, thr, seed=0)
    cpu.w_cold = [w.cpu().numpy().copy() for w in hip.coll.weight_hash]          # same starting tables
    rng = np.random.default_rng(3)
    n = 1500
    for step in range(20):
        fields = rng.integers(0, F, size=n).astype(np.int32)
        local = np.array([(rng.zipf(1.2) - 1) % counts[f] for f in fields], np.int64)
        local[(fields == 0) & (local == 0)] = 1
        gids = (local + offsets[fields]).astype(np.int32)
        t_g, t_f = torch.from_numpy(gids).to(dev), torch.from_numpy(fields).to(dev)
        que, rows = hip.forward_gid(t_g, t_f)
        want_que, want_rows = cpu.forward_gid(torch.from_numpy(gids), torch.from_numpy(fields))
        assert np.array_equal(que.cpu().numpy(), want_que.numpy()), f"step {step}: routing"
        _close(rows.cpu().numpy(), want_rows.numpy(), f"step {step} rows")
        grads = torch.randn(n, dim, device=dev, generator=torch.Generator(device=dev).manual_seed(step))
        scores = (torch.rand(n, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + step)) * 2).contiguous()
        packed = torch.zeros(n, dim + 4, device=dev)            # the exchange buffer's rows: [gradient row | score | pad]
        packed[:, :dim] = grads; packed[:, dim] = scores
        ins = hip.backward_gid(t_g, t_f, que, packed, lr)
        # the CPU double applies SGD lookup by lookup in float32; sum per row in float64 here for a tighter reference
        g64 = grads.cpu().numpy().astype(np.float64)
        want_ins = cpu.sk.insert_val(gids, scores.cpu().numpy())
        qa = cpu.sk.query(gids)
        for i in np.nonzero(want_ins)[0]:
            if qa[i] < 0:
                cpu.w_hot[-int(qa[i])] = cpu.w_cold[fields[i]][int(local[i]) % hash_sizes[fields[i]]]
        acc_h = np.zeros(cpu.w_hot.shape, np.float64)
        acc_c = [np.zeros(w.shape, np.float64) for w in cpu.w_cold]
        q = want_que.numpy()
        for i in range(n):
            if q[i] < 0:
                acc_h[-q[i]] += g64[i]
            else:
                acc_c[fields[i]][q[i] % hash_sizes[fields[i]]] += g64[i]
        cpu.w_hot -= (lr * acc_h).astype(np.float32)
        for f in range(F):
            cpu.w_cold[f] -= (lr * acc_c[f]).astype(np.float32)
        assert np.array_equal(ins.cpu().numpy(), want_ins), f"step {step}: ins"
        _close(hip.coll.weight_high.cpu().numpy(), cpu.w_hot, f"step {step} hot table")
        for f in range(F):
            _close(hip.coll.weight_hash[f].cpu().numpy(), cpu.w_cold[f], f"step {step} cold table {f}")
    assert not oracle_mod.states_equal(hip.sketch.state(), cpu.sk.state())


def _rank(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from cafe_amd.sharded import ShardedCafeEmbedding, owner_of, shard_sizes
        from oracle import cafe_oracle as om
        counts, hash_sizes, dim, hotn, thr, B, lr = [3000, 40, 50000], [37, 2, 211], 16, 600, 3, 256, 0.1
        F = len(counts)
        offsets = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
        lh, _ = shard_sizes(hotn, hash_sizes, world)
        eng = ShardedCafeEmbedding(counts, dim, hotn, hash_sizes, offsets, threshold=thr, seed=1)
        ref = om.Oracle(lh, thr, 1, 1.0000005, 1)
        rng = np.random.default_rng(5 + rank)
        batches = []
        for step in range(8):
            ids = np.stack([(rng.zipf(1.2, size=B) - 1) % c for c in counts]).astype(np.int64)
            ids[0] = np.maximum(ids[0], 1)
            batches.append(torch.from_numpy(ids).cuda())
        for step in range(8):
            out = eng.forward(batches[step])
            assert out.shape == (F, B, dim) and bool(torch.isfinite(out).all())
            grad = torch.randn(F, B, dim, device="cuda", generator=torch.Generator(device="cuda").manual_seed(10 * step + rank))
            scores, ins = eng.backward_update(grad, lr)
            r_gids, r_fields, r_scores, r_ins = eng._last
            want = ref.insert_val(r_gids.cpu().numpy(), r_scores.cpu().numpy())       # the shard's own sub-stream
            assert np.array_equal(r_ins.cpu().numpy(), want), f"step {step}: per-shard ins"
            own = owner_of(r_gids, world)
            assert bool((own == rank).all()), "a lookup reached a rank that does not own it"
            if step % 2 == 0 and step + 1 < 8:
                eng.prefetch(batches[step + 1])       # routed one step ahead on the exchange stream
        assert eng.host_syncs_in_step == 4
        # every shard step ran as the fused owner step (cafe_backward_update_gid_dev): one parallel insert, none in order
        c = eng.sketch.counters()
        assert c["inorder_calls"] == 0 and c["parallel_calls"] == 8, c
        bad = om.states_equal(eng.sketch.state(), ref.state())
        assert not bad, bad
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_gpu():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}:\n{msg}"


@pytest.mark.parametrize("world", [1, 2, 3, 8, 64])
@pytest.mark.parametrize("F,B", [(26, 8192), (3, 1000), (1, 1), (5, 1025)])
def test_route_kernels_match_the_torch_route(world, F, B):
    """cafe_route_dev / cafe_split_ids_dev against the plain torch route of sharded.py (owner_of, stable argsort by
    owner, bincount, field << 32 | gid): bit-exact, ids up to the int32 wrap included."""
    from cafe_amd.sharded import HipShardBackend, owner_of
    dev = torch.device("cuda", 0)
    counts = [5000] * F
    offsets_np = (np.arange(F, dtype=np.int64) * 82_000_000)          # the last fields' ids lie beyond 2^31: negative int32 bits
    hip = HipShardBackend([10] * F, 16, 20, [3] * F, np.zeros(F, np.int64), 2, 1, 1.0, 0, dev)
    gen = torch.Generator(device=dev).manual_seed(F * 1000 + B + world)
    local = torch.randint(0, 5000, (F, B), device=dev, generator=gen)
    offsets = torch.from_numpy(offsets_np).to(dev)
    order, packed, send = hip.route(local, offsets, world)
    gids = (local + offsets[:, None]).to(torch.int32).reshape(-1)
    own = owner_of(gids, world)
    want_order = torch.argsort(own, stable=True)
    fields = torch.arange(F, device=dev, dtype=torch.int32).repeat_interleave(B)
    want_packed = (fields[want_order].to(torch.int64) << 32) | (gids[want_order].to(torch.int64) & 0xFFFFFFFF)
    assert torch.equal(send, torch.bincount(own, minlength=world)), "lookups per owner"
    assert torch.equal(order, want_order), "order"
    assert torch.equal(packed, want_packed), "packed ids"
    g2, f2 = hip.split_ids(packed)
    assert torch.equal(g2, gids[want_order]) and torch.equal(f2, fields[want_order])
    del counts
