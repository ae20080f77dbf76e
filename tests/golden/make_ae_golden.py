#!/usr/bin/env python3
"""Generate tests/golden/ae_*.npz from the REFERENCE's earlier HotSketch generation: ArtifactEvaluation/embeddings/
sketch.cpp compiled here (oracle/Makefile ref_more -> oracle/_ref/ae_sklib.so), driven through its C entry points
(init(n, Threshold, dr, idx), batch_insert, batch_insert_val, batch_query; :207-228) on seeded streams.  Per call the
fixture holds the ids, the scores and the ins[] / que[] the reference returned.  Data only.  The reference keeps no more
than 16384 results per call (:7), so batches stay below that.  Runs only in the build container.

    python tests/golden/make_ae_golden.py
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
    # scores (batch_insert_val), several decay() events (tot > lim * thr * 10), ids colliding in every bucket
    "ae_decay": dict(seed=1, lim=200, thr=3, dr=0.5, calls=60, batch=700, space=3000, a=1.2, val=True, fields=1),
    # counts (batch_insert), a free list that runs dry (more ids reach the threshold than there are slots), decay gives
    # slots back
    "ae_dry_free_list": dict(seed=2, lim=60, thr=2, dr=0.9, calls=50, batch=400, space=800, a=1.05, val=False, fields=1),
    # three instances side by side (the reference keeps one per field, ss[26]), large batches, slow decay
    "ae_three_fields": dict(seed=3, lim=5000, thr=5, dr=0.99, calls=6, batch=12000, space=400000, a=1.1, val=True, fields=3),
}


def worker(name, spec, out):
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "ae_sklib.so"))
    u32p, f32p, i32p = ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    L.init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]
    L.batch_insert.argtypes = [u32p, ctypes.c_int, ctypes.c_int]; L.batch_insert.restype = i32p
    L.batch_insert_val.argtypes = [u32p, f32p, ctypes.c_int, ctypes.c_int]; L.batch_insert_val.restype = i32p
    L.batch_query.argtypes = [u32p, ctypes.c_int, ctypes.c_int]; L.batch_query.restype = i32p
    F = spec["fields"]
    for f in range(F):
        L.init(spec["lim"], spec["thr"], spec["dr"], f)
    rng = np.random.default_rng(spec["seed"])
    arrs = dict(lim=spec["lim"], thr=spec["thr"], dr=np.float64(spec["dr"]), calls=spec["calls"], fields=F, val=int(spec["val"]))
    for c in range(spec["calls"]):
        for f in range(F):
            n = spec["batch"]
            ids = ((rng.zipf(spec["a"], size=n) - 1) % spec["space"] + f * spec["space"]).astype(np.uint32)
            if spec["val"]:
                v = (rng.random(n) * 3).astype(np.float32)
                r = L.batch_insert_val(ids.ctypes.data_as(u32p), v.ctypes.data_as(f32p), n, f)
                arrs[f"v_{c}_{f}"] = v
            else:
                r = L.batch_insert(ids.ctypes.data_as(u32p), n, f)
            arrs[f"ids_{c}_{f}"] = ids
            arrs[f"ins_{c}_{f}"] = np.ctypeslib.as_array(r, shape=(n,)).astype(np.int8)
            q = L.batch_query(ids.ctypes.data_as(u32p), n, f)
            arrs[f"que_{c}_{f}"] = np.ctypeslib.as_array(q, shape=(n,)).astype(np.int32)
    np.savez_compressed(out, **arrs)


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_more"], stdout=subprocess.DEVNULL)
    for name, spec in SPECS.items():
        out = os.path.join(HERE, name + ".npz")
        env = dict(os.environ, MALLOC_PERTURB_="165", OMP_NUM_THREADS="4")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker", name, json.dumps(spec), out], env=env,
                              stdout=subprocess.DEVNULL)
        print(f"{name}: {os.path.getsize(out) / 1024:.1f} KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], json.loads(sys.argv[3]), sys.argv[4])
    else:
        main()
