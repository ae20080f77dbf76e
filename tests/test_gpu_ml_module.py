"""cafe_amd.sketch_ml_embed.MLEmbeddingBag (the three-tier module of tricks/sketch_ml_embed.py) on the GPU against
the numpy restatement of its formula driven by the C oracle's sketch.  The reference module cannot run (see the
module docstring), so there are no recorded steps of it: parity is against the oracle only ("parity unpinned").

Bars: routing (hot mask, slots, median mask) bit-exact; rows, loss and tables within 1e-5 relative."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import cafe_oracle as co  # noqa: E402

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def _close(a, b, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= RTOL, f"{what}: max rel err {err:.3e}"


@pytest.mark.parametrize("dim", [16, 128])
def test_ml_module_against_oracle(dim):
    import cafe_amd
    from cafe_amd.sketch_ml_embed import MLEmbeddingBag
    rng = np.random.default_rng(11)
    torch.manual_seed(3)
    counts = [5000, 300, 20000]
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
    offs[0] = 1                                        # ids >= 1
    hs = [37, 5, 211]
    F, B, hotn, median_n, thr, lr = 3, 512, 400, 97, 6, 0.1
    dev = torch.device("cuda", 0)
    lib = ctypes.CDLL(cafe_amd.LIB_PATH)
    lib.init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    lib.init(hotn, thr, 1, 1.05)                       # alpha 1.05: a flip every ~190 calls, decay pending in places
    o = co.Oracle(hotn, thr, 1, 1.05)
    w_hot = torch.nn.Parameter((torch.rand(hotn, dim, device=dev) - 0.5))
    w_med = torch.nn.Parameter((torch.rand(median_n, dim, device=dev) - 0.5))
    embs = [MLEmbeddingBag(f, hotn, median_n, lib, w_hot, w_med, dev, counts[f], dim, int(offs[f]), hs[f]).to(dev)
            for f in range(F)]
    opt = torch.optim.SGD([w_hot, w_med] + [e.weight_hash for e in embs], lr=lr)
    n_hot = n_med = 0
    h_hot = w_hot.detach().cpu().numpy().copy()
    h_med = w_med.detach().cpu().numpy().copy()
    h_cold = [e.weight_hash.detach().cpu().numpy().copy() for e in embs]
    A = torch.randn(F, B, dim, device=dev)
    for step in range(80):
        test = step % 10 == 9
        opt.zero_grad(set_to_none=True)
        ids = [(rng.zipf(1.15, B) - 1) % counts[f] for f in range(F)]
        outs = []
        routes = []
        for f in range(F):
            out = embs[f](torch.from_numpy(ids[f]), torch.arange(B), test=test)
            gid = (ids[f] + offs[f]).astype(np.uint32)
            if not test:
                ins = o.insert(gid)
                assert np.array_equal(embs[f]._ins.cpu().numpy(), ins), f"step {step} field {f}: ins"
            que, qmed = o.query(gid), o.query_median(gid)
            assert np.array_equal(embs[f]._que.cpu().numpy(), que), f"step {step} field {f}: que"
            assert np.array_equal(embs[f]._qmed.cpu().numpy(), qmed), f"step {step} field {f}: median"
            want = co.ref_forward_ml(que, qmed, h_hot, h_cold[f], h_med, hs[f], median_n)
            _close(out.detach().cpu().numpy(), want, f"step {step} field {f} forward")
            n_hot += int((que < 0).sum()); n_med += int(((que >= 0) & (qmed < 0)).sum())
            outs.append(out); routes.append((que, qmed))
        if step % 20 == 0:
            assert np.array_equal(embs[0].query_median(torch.from_numpy(ids[0])).numpy(),
                                  o.query_median((ids[0] + offs[0]).astype(np.uint32)) < 0)
        loss = sum((outs[f] * A[f]).sum() for f in range(F))
        loss.backward()
        opt.step()
        # the same SGD on the host tables: duplicates accumulate (sparse COO add_)
        Ah = A.cpu().numpy()
        for f in range(F):
            que, qmed = routes[f]
            dic = np.abs(que).astype(np.int64)
            hot = que < 0
            med = (~hot) & (qmed < 0)
            np.subtract.at(h_hot, dic[hot] % hotn, lr * Ah[f][hot])
            np.subtract.at(h_cold[f], dic[~hot] % hs[f], lr * Ah[f][~hot])
            np.subtract.at(h_med, dic[med] % median_n, lr * Ah[f][med])
    _close(w_hot.detach().cpu().numpy(), h_hot, "hot table")
    _close(w_med.detach().cpu().numpy(), h_med, "median table")
    for f in range(F):
        _close(embs[f].weight_hash.detach().cpu().numpy(), h_cold[f], f"cold table {f}")
    assert n_hot > 1000 and n_med > 1000, (n_hot, n_med)


def test_reference_abi_batch_query_median():
    import cafe_amd
    lib = ctypes.CDLL(cafe_amd.LIB_PATH)
    lib.init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    lib.batch_insert_val.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float), ctypes.c_int]
    lib.batch_insert_val.restype = ctypes.POINTER(ctypes.c_int)
    lib.batch_query_median.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int]      # sketch_ml_embed.py:82-84
    lib.batch_query_median.restype = ctypes.POINTER(ctypes.c_int)
    lib.init(300, 40, 1, 1.03)
    o = co.Oracle(300, 40, 1, 1.03)
    rng = np.random.default_rng(2)
    for _ in range(150):
        ids = (rng.zipf(1.2, 300) % 2500 + 1).astype(np.int32)
        v = (rng.random(300) * 5).astype(np.float32)
        lib.batch_insert_val(ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), v.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 300)
        o.insert_val(ids.astype(np.uint32), v)
    probe = np.arange(1, 2501, dtype=np.int32)
    got = np.ctypeslib.as_array(lib.batch_query_median(probe.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(probe)), (len(probe),)).copy()
    want = o.query_median(probe.astype(np.uint32))
    assert np.array_equal(got, want) and (want < 0).sum() > 0
