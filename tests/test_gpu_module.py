"""GPU parity of cafe_amd.sk_embedding_bag.SKEmbeddingBag against steps recorded from the reference's own
tricks/sk_embedding_bag.py module on CPU (tests/golden/module_*.npz): same constructor, same call order
(forward of every field -> backward -> insert_grad of every field -> torch.optim.SGD.step), bound to
libcafe_hip.so the way the reference binds sklib.so (ctypes.CDLL + lib.init).

Bars: routing bit-exact, forward rows / loss / tables within 1e-5 relative (north_star tolerance)."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import module_golden_names
import golden_util as gu

pytestmark = pytest.mark.gpu
RTOL = 1e-5


import os, sys  # noqa: E402
sys.path.insert(0, os.path.dirname(__file__))
from tol import close as _tol_close, ATOL_TABLE  # noqa: E402


def _close(a, b, what):
    """max-norm AND elementwise (tests/tol.py); small tables of the reference fixtures, several steps: ATOL_TABLE"""
    _tol_close(a, b, what, RTOL, ATOL_TABLE)


def test_module_fails_loudly_without_init():
    """A module used before lib.init(...) must raise, not fall back to anything.  Run in a process of its own: whether a
    global sketch exists is process state (an earlier test of the session may have called init)."""
    import subprocess
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from cafe_amd import _lib\n"
        "from cafe_amd.sk_embedding_bag import SKEmbeddingBag\n"
        "class NoLib: pass\n"
        "assert not _lib.load().cafe_global_sketch()\n"
        "e = SKEmbeddingBag(0, 100, NoLib(), torch.zeros(100, 16, device='cuda'), 'cuda', 10, 16, 0, 3).cuda()\n"
        "try:\n"
        "    e(torch.arange(4), torch.arange(4))\n"
        "except _lib.CafeError as ex:\n"
        "    print('RAISED', ex)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert "RAISED" in r.stdout and "lib.init" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("name", module_golden_names())
def test_module_matches_reference_module(name):
    import cafe_amd
    from cafe_amd.sk_embedding_bag import SKEmbeddingBag, get_sketch_time, reset_sketch_time
    g = gu.load(name)
    counts, hs, offs = g["counts"], [int(h) for h in g["hash_sizes"]], g["offsets"]
    F, D, hotn, B, lr = len(counts), int(g["dim"]), int(g["hotn"]), int(g["batch"]), float(g["lr"])
    dev = torch.device("cuda", 0)
    lib = ctypes.CDLL(cafe_amd.LIB_PATH)                                   # dlrm_s_pytorch.py:1403
    lib.init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    lib.init(hotn, int(g["thr"]), 1, float(g["alpha"]))                    # dlrm_s_pytorch.py:444-449
    weight_high = torch.nn.Parameter(torch.zeros(hotn, D, device=dev))
    embs = [SKEmbeddingBag(f, hotn, lib, weight_high, dev, int(counts[f]), D, int(offs[f]), hs[f]) for f in range(F)]
    for f, e in enumerate(embs):
        e.to(dev)
        with torch.no_grad():
            e.weight_hash.copy_(torch.from_numpy(g[f"w_cold0_{f}"]))
    opt = torch.optim.SGD([weight_high] + [e.weight_hash for e in embs], lr=lr)
    A = torch.from_numpy(g["A"]).to(dev)
    offsets_t = torch.arange(B)
    reset_sketch_time()
    for step in range(int(g["steps"])):
        ids = g[f"ids_{step}"]
        opt.zero_grad(set_to_none=True)
        outs = [embs[f](torch.from_numpy(ids[f]), offsets_t) for f in range(F)]       # CPU ids, like the reference
        for f in range(F):
            _close(outs[f].detach().cpu().numpy(), g[f"out_{step}"][f], f"{name} step {step} field {f} forward")
        loss = sum(torch.sigmoid((outs[f] * A[f]).sum(1)).sum() for f in range(F))
        loss.backward()
        assert abs(loss.item() - float(g[f"loss_{step}"])) <= RTOL * abs(float(g[f"loss_{step}"]))
        for f in range(F):
            embs[f].insert_grad(torch.from_numpy(ids[f]))
        opt.step()
        for f in range(F):
            hot, dic = embs[f].query(torch.from_numpy(ids[f]))
            want = g[f"que_after_{step}"][f]
            assert np.array_equal(hot.numpy(), want < 0) and np.array_equal(dic.numpy(), np.abs(want)), \
                f"{name} step {step} field {f}: routing after the step"
            _close(embs[f].weight_hash.detach().cpu().numpy(), g[f"w_cold_{step}_{f}"], f"{name} step {step} cold table {f}")
        _close(weight_high.detach().cpu().numpy(), g[f"w_hot_{step}"], f"{name} step {step} hot table")
    assert get_sketch_time() > 0.0
    assert embs[0].query_dic.dtype == np.bool_ and embs[0].query_dic.shape == (B,)


def test_sketch_state_travels_with_a_checkpoint(tmp_path):
    """sketch_state_dict / load_sketch_state_dict: the global sketch through torch.save / torch.load(weights_only=True)
    and lib.update() afterwards (dlrm_s_pytorch.py:1510-1511) leave queries and later inserts unchanged."""
    import cafe_amd
    from cafe_amd.sk_embedding_bag import sketch_state_dict, load_sketch_state_dict
    from cafe_amd.sketch import HotSketch
    lib = ctypes.CDLL(cafe_amd.LIB_PATH)
    lib.init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    lib.init(500, 4, 1, 1.01)
    rng = np.random.default_rng(8)
    sk = HotSketch.from_global()
    batches = [(rng.zipf(1.2, 400) % 5000 + 1).astype(np.int32) for _ in range(60)]
    scores = [(rng.random(400) * 3).astype(np.float32) for _ in range(60)]
    for b, v in zip(batches[:40], scores[:40]):
        sk.insert_val(torch.from_numpy(b).cuda(), torch.from_numpy(v).cuda())
    path = str(tmp_path / "ck.pt")
    torch.save({"sketch": sketch_state_dict()}, path)
    for b, v in zip(batches[40:], scores[40:]):
        sk.insert_val(torch.from_numpy(b).cuda(), torch.from_numpy(v).cuda())
    want = sk.state()
    lib.init(500, 4, 1, 1.01)                                  # a fresh process would start here
    load_sketch_state_dict(torch.load(path, weights_only=True)["sketch"])
    lib.update()
    sk = HotSketch.from_global()
    for b, v in zip(batches[40:], scores[40:]):
        sk.insert_val(torch.from_numpy(b).cuda(), torch.from_numpy(v).cuda())
    got = sk.state()
    for k in ("mval", "mdic", "mcnt", "lval", "ldic", "lcnt", "lt", "fifo"):
        assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), k
    for k in ("thr", "num", "real_n", "clk", "flip", "batch_num", "decay"):
        assert got[k] == want[k], k


def test_evaluation_between_forward_and_backward_does_not_change_the_step():
    """backward and insert_grad belong to the forward that produced the gradient, not to the module's
    latest forward.  Two identical modules take the same training step; one of them evaluates another batch (of
    another size) under no_grad between its forward and its backward.  Gradients, sketch state and tables must agree."""
    import cafe_amd
    from cafe_amd.sk_embedding_bag import SKEmbeddingBag
    from cafe_amd.sketch import HotSketch
    dev = torch.device("cuda", 0)
    hotn, D, n, hs = 200, 16, 5000, 37
    rng = np.random.default_rng(3)
    train_ids = torch.from_numpy((rng.zipf(1.2, 256) % n).astype(np.int64))
    eval_ids = torch.from_numpy((rng.zipf(1.2, 100) % n).astype(np.int64))
    results = []
    for interleave in (False, True):
        sk = HotSketch(hotn, 2, 1, 1.0, 1)
        wh = torch.nn.Parameter(torch.zeros(hotn, D, device=dev))
        e = SKEmbeddingBag(0, hotn, sk, wh, dev, n, D, 0, hs).to(dev)
        with torch.no_grad():
            e.weight_hash.copy_(torch.linspace(-1, 1, hs * D, device=dev).view(hs, D))
        for step in range(3):
            out = e(train_ids, torch.arange(256))
            if interleave:
                with torch.no_grad():
                    e(eval_ids, torch.arange(100))
            (out * out).sum().backward()
            e.insert_grad(train_ids)
            with torch.no_grad():
                for p in (wh, e.weight_hash):
                    p.add_(p.grad.to_dense(), alpha=-0.05)
                    p.grad = None
        st = sk.state()
        results.append((st, wh.detach().cpu().numpy(), e.weight_hash.detach().cpu().numpy()))
    (sa, ha, ca), (sb, hb, cb) = results
    for k in ("mval", "mdic", "lval", "ldic", "lt", "fifo"):
        assert np.array_equal(np.asarray(sa[k]), np.asarray(sb[k])), k
    # (torch's sparse -> dense sums duplicate rows in no fixed order: floats agree to rounding, not bit for bit)
    for a, b, what in ((sa["mcnt"], sb["mcnt"], "main scores"), (sa["lcnt"], sb["lcnt"], "LRU scores"), (ha, hb, "hot table"),
                       (ca, cb, "cold table")):
        _close(a, b, what)
    assert (np.asarray(sa["mdic"]) != 0).sum() > 5
