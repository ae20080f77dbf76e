#!/usr/bin/env python3
"""Generate module-level golden steps (tests/golden/module_*.npz) from the REFERENCE ITSELF:
the unmodified tricks/sk_embedding_bag.py::SKEmbeddingBag, imported from /root/reference, driving the
reference's own sketch.cpp (oracle/_ref/sklib.so) on CPU, in the driver's call order
(dlrm_s_pytorch.py:1640-1658): forward of every field -> loss.backward() -> insert_grad of every
field -> torch.optim.SGD.step().

Runs only in the build container.  The fixture is data only: ids, the upstream mixing tensor that
defines the loss, initial cold tables, and per step the forward output, the output gradient, the
routing after the step and both tables after the step.

Two departures from running the reference blindly, both documented in SURVEY.md 8(c):
  * weight_high is zero-filled (the reference leaves it uninitialised, dlrm_s_pytorch.py:439-442, Q9);
  * with more than one field the shared hot table's gradient is re-assembled, before insert_grad, as the
    uncoalesced sparse tensor whose value blocks are in field order (what `l = field_num * N`,
    sk_embedding_bag.py:176-181, indexes into); on CPU autograd's sparse accumulation merges the blocks
    and the unmodified slice crashes (Q1).  The re-assembled tensor is asserted equal to autograd's.

    python tests/golden/make_module_golden.py
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

SPECS = {
    # one field: the reference runs completely unmodified
    "module_one_field": dict(seed=11, counts=[5000], hash_sizes=[37], dim=16, hotn=120, thr=3, batch=96, steps=14,
                             lr=0.1, a=1.15),
    # three fields sharing the hot table, offsets not multiples of the hash sizes (Q4 visible)
    "module_three_fields": dict(seed=12, counts=[3000, 41, 70000], hash_sizes=[29, 3, 211], dim=16, hotn=300, thr=3,
                                batch=128, steps=12, lr=0.1, a=1.2),
    # D=128 rows, larger threshold: many LRU hits before a promotion
    "module_dim128": dict(seed=13, counts=[900, 20000], hash_sizes=[7, 61], dim=128, hotn=100, thr=5, batch=64, steps=8,
                          lr=0.05, a=1.3),
    # the 26 Criteo-Kaggle fields at BASELINE.json configs[1]'s sizes (compress 0.001, hash-rate 0.3, D = 16): hotn and the
    # per-field hash sizes as dlrm_s_pytorch.py:1237-1242 and :320 compute them; field cardinalities from 3 to 10 M
    "module_kaggle26": dict(seed=14, counts="kaggle", hash_sizes="kaggle", dim=16, hotn="kaggle", thr=3, batch=64, steps=5,
                            lr=0.1, a=1.1),
}
KAGGLE_COUNTS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10,
                 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]          # the reference's Kaggle count file (cfg2)


def worker(name, spec, out_path):
    import torch
    sys.path.insert(0, "/root/reference")
    from tricks.sk_embedding_bag import SKEmbeddingBag          # the reference module, unmodified

    torch.manual_seed(spec["seed"])
    torch.set_num_threads(1)
    rng = np.random.default_rng(spec["seed"])
    counts, hs, D, hotn, B = spec["counts"], spec["hash_sizes"], spec["dim"], spec["hotn"], spec["batch"]
    if counts == "kaggle":
        counts = list(KAGGLE_COUNTS)
        cr, hr = 0.001, 0.3
        hotn = int(sum(counts) * cr * (1 - hr) * (D * 4 / (D * 4 + 48)))        # dlrm_s_pytorch.py:1237-1240
        hs = [int(round(cr * hr * n + 0.51)) for n in counts]                    # dlrm_s_pytorch.py:1241, :320
    F = len(counts)
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "sklib.so"))
    lib.init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    lib.init(hotn, spec["thr"], 1, 1.0000005)
    weight_high = torch.nn.Parameter(torch.zeros(hotn, D))
    dev = torch.device("cpu")
    embs = [SKEmbeddingBag(f, hotn, lib, weight_high, dev, counts[f], D, int(offs[f]), hs[f]) for f in range(F)]
    params = [weight_high] + [e.weight_hash for e in embs]
    opt = torch.optim.SGD(params, lr=spec["lr"])
    A = torch.randn(F, B, D)                      # loss = sum_f sum_i sigmoid(<out_f[i], A_f[i]>)
    arrs = dict(counts=np.array(counts, np.int64), hash_sizes=np.array(hs, np.int32), offsets=offs, dim=D, hotn=hotn,
                thr=spec["thr"], batch=B, steps=spec["steps"], lr=np.float64(spec["lr"]), alpha=np.float64(1.0000005),
                A=A.numpy().copy())
    for f in range(F):
        arrs[f"w_cold0_{f}"] = embs[f].weight_hash.detach().numpy().copy()
    i32p = ctypes.POINTER(ctypes.c_int)
    offsets_t = torch.arange(B, dtype=torch.int64)
    for step in range(spec["steps"]):
        ids = np.stack([(rng.zipf(spec["a"], size=B) - 1) % c for c in counts]).astype(np.int64)
        ids[0] = np.maximum(ids[0], 1)
        opt.zero_grad(set_to_none=True)
        outs, gouts = [], [None] * F
        for f in range(F):
            o = embs[f](torch.from_numpy(ids[f]), offsets_t)
            o.register_hook(lambda g, f=f: gouts.__setitem__(f, g.detach().clone()))
            outs.append(o)
        loss = sum(torch.sigmoid((outs[f] * A[f]).sum(1)).sum() for f in range(F))
        loss.backward()
        if F > 1:
            masks = [torch.from_numpy(e.query_dic) for e in embs]
            que = [np.ctypeslib.as_array(lib_query(lib, ids[f] + offs[f]), shape=(B,)).copy() for f in range(F)]
            idx = torch.cat([torch.from_numpy(np.abs(que[f]).astype(np.int64) % hotn) for f in range(F)])[None, :]
            val = torch.cat([torch.where(masks[f][:, None], gouts[f], torch.zeros_like(gouts[f])) for f in range(F)])
            rebuilt = torch.sparse_coo_tensor(idx, val, (hotn, D))
            assert torch.allclose(rebuilt.to_dense(), weight_high.grad.to_dense(), rtol=1e-6, atol=1e-7)
            weight_high.grad = rebuilt
        for f in range(F):
            embs[f].insert_grad(torch.from_numpy(ids[f]))
        opt.step()
        arrs[f"ids_{step}"] = ids
        arrs[f"out_{step}"] = np.stack([o.detach().numpy() for o in outs])
        arrs[f"gout_{step}"] = np.stack([g.numpy() for g in gouts])
        arrs[f"loss_{step}"] = np.float64(loss.item())
        arrs[f"que_after_{step}"] = np.stack(
            [np.ctypeslib.as_array(lib_query(lib, ids[f] + offs[f]), shape=(B,)).copy() for f in range(F)]).astype(np.int32)
        arrs[f"w_hot_{step}"] = weight_high.detach().numpy().copy()
        for f in range(F):
            arrs[f"w_cold_{step}_{f}"] = embs[f].weight_hash.detach().numpy().copy()
    np.savez_compressed(out_path, **arrs)


def lib_query(lib, gids):
    i32p = ctypes.POINTER(ctypes.c_int)
    lib.batch_query.argtypes = [i32p, ctypes.c_int]
    lib.batch_query.restype = i32p
    a = np.ascontiguousarray(gids.astype(np.int32))
    lib_query.keep = a
    return lib.batch_query(a.ctypes.data_as(i32p), len(a))


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    only = sys.argv[1:]
    for name, spec in SPECS.items():
        if only and name not in only:
            continue
        out = os.path.join(HERE, name + ".npz")
        env = dict(os.environ, MALLOC_PERTURB_="165")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker", name, json.dumps(spec), out], env=env,
                              stdout=subprocess.DEVNULL)
        print(f"{name}: {os.path.getsize(out) / 1024:.1f} KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], json.loads(sys.argv[3]), sys.argv[4])
    else:
        main()
