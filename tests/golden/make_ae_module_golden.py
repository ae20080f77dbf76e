#!/usr/bin/env python3
"""Generate module-level golden steps of the artifact generation (tests/golden/aemod_*.npz) from the REFERENCE ITSELF:
the unmodified ArtifactEvaluation/embeddings/sk_embedding_bag.py::SKEmbeddingBag, loaded from /root/reference, driving the
artifact's own sketch.cpp (oracle/_ref/ae_sklib.so, built by `make -C oracle ref_more`) on CPU, in the call order of
init_embed.py:212-235 and main.py's loop: forward of every field -> loss.backward() -> insert_grad(input, use_freq) of
every field -> torch.optim.SGD.step().

Runs only in the build container.  The fixture is data only: ids, the mixing tensor that defines the loss, the initial
tables, and per step the forward output, the output gradient, the routing after the step and all tables after the step.

Configurations the reference file runs unmodified on CPU: every field with a hot table of its own
(cafe_hot_separate_field = True: the position slice of the hot table's gradient starts at 0), with gradient-norm scores
and with cafe_use_freq; and one field behind the shared sketch.  The generator asserts that no lookup the insert reported
has lost its slot again at the re-query (the reference would then index the hot table with a global id).

    python tests/golden/make_ae_module_golden.py
"""
import ctypes
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

SPECS = {
    "aemod_separate_three_fields": dict(seed=21, counts=[4000, 90, 60000], hash_sizes=[31, 5, 199], dim=16, hotn=[60, 12, 150],
                                        thr=3, decay=0.98, batch=128, steps=14, lr=0.1, a=1.2, separate=True, use_freq=False),
    "aemod_separate_use_freq": dict(seed=22, counts=[2500, 30000], hash_sizes=[17, 113], dim=32, hotn=[40, 90], thr=4, decay=0.95,
                                    batch=96, steps=12, lr=0.05, a=1.25, separate=True, use_freq=True),
    "aemod_shared_one_field": dict(seed=23, counts=[50000], hash_sizes=[157], dim=16, hotn=[200], thr=3, decay=0.98, batch=160,
                                   steps=12, lr=0.1, a=1.15, separate=False, use_freq=False),
}


def worker(name, spec, out_path):
    import torch
    sp = importlib.util.spec_from_file_location("ae_ref_module", "/root/reference/ArtifactEvaluation/embeddings/sk_embedding_bag.py")
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)                                   # the reference module, unmodified
    SKEmbeddingBag = mod.SKEmbeddingBag

    torch.manual_seed(spec["seed"])
    torch.set_num_threads(1)
    rng = np.random.default_rng(spec["seed"])
    counts, hs, D, B = spec["counts"], spec["hash_sizes"], spec["dim"], spec["batch"]
    F = len(counts)
    sep = spec["separate"]
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "ae_sklib.so"))
    lib.init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]
    lib.init.restype = None
    i32p = ctypes.POINTER(ctypes.c_int)
    lib.batch_query.argtypes = [i32p, ctypes.c_int, ctypes.c_int]
    lib.batch_query.restype = i32p
    dev = torch.device("cpu")
    hot = []
    for k, h in enumerate(spec["hotn"]):
        n = counts[k] if sep else max(counts)
        w = torch.nn.Parameter(torch.empty(h, D))
        torch.nn.init.uniform_(w, -np.sqrt(1 / n), np.sqrt(1 / n))          # init_embed.py:117-118 / 161-162
        lib.init(h, spec["thr"], ctypes.c_double(spec["decay"]), ctypes.c_int(k))
        hot.append(w)
    embs = [SKEmbeddingBag(f, spec["hotn"][f if sep else 0], lib, hot[f if sep else 0], dev, counts[f], D, int(offs[f]), hs[f], sep)
            for f in range(F)]
    params = hot + [e.weight_hash for e in embs]
    opt = torch.optim.SGD(params, lr=spec["lr"])
    A = torch.randn(F, B, D)
    arrs = dict(counts=np.array(counts, np.int64), hash_sizes=np.array(hs, np.int32), offsets=offs, dim=D,
                hotn=np.array(spec["hotn"], np.int64), thr=spec["thr"], decay=np.float64(spec["decay"]), batch=B,
                steps=spec["steps"], lr=np.float64(spec["lr"]), separate=int(sep), use_freq=int(spec["use_freq"]),
                A=A.numpy().copy())
    for k, w in enumerate(hot):
        arrs[f"w_hot0_{k}"] = w.detach().numpy().copy()
    for f in range(F):
        arrs[f"w_cold0_{f}"] = embs[f].weight_hash.detach().numpy().copy()

    def query(f, ids_f):
        a = np.ascontiguousarray((ids_f + offs[f]).astype(np.int32))
        return np.ctypeslib.as_array(lib.batch_query(a.ctypes.data_as(i32p), len(a), embs[f].ss_idx), shape=(len(a),)).copy()

    offsets_t = torch.arange(B, dtype=torch.int64)
    migrated = 0
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
        que_before = np.stack([query(f, ids[f]) for f in range(F)])
        for f in range(F):
            embs[f].insert_grad(torch.from_numpy(ids[f]), spec["use_freq"])
            # what the insert reported is in the library's result buffer until the next call: read through the re-query
            qa = query(f, ids[f])
            newly = (qa < 0) & (que_before[f] >= 0)
            migrated += int(newly.sum())
        opt.step()
        arrs[f"ids_{step}"] = ids
        arrs[f"out_{step}"] = np.stack([o.detach().numpy() for o in outs])
        arrs[f"gout_{step}"] = np.stack([g.numpy() for g in gouts])
        arrs[f"que_after_{step}"] = np.stack([query(f, ids[f]) for f in range(F)]).astype(np.int32)
        for k, w in enumerate(hot):
            arrs[f"w_hot_{step}_{k}"] = w.detach().numpy().copy()
        for f in range(F):
            arrs[f"w_cold_{step}_{f}"] = embs[f].weight_hash.detach().numpy().copy()
    assert migrated > 20, migrated
    arrs["migrated"] = migrated
    np.savez_compressed(out_path, **arrs)


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_more"], stdout=subprocess.DEVNULL)
    only = sys.argv[1:]
    for name, spec in SPECS.items():
        if only and name not in only:
            continue
        out = os.path.join(HERE, name + ".npz")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker", name, json.dumps(spec), out],
                              stdout=subprocess.DEVNULL)
        print(f"{name}: {os.path.getsize(out) / 1024:.1f} KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], json.loads(sys.argv[3]), sys.argv[4])
    else:
        main()
