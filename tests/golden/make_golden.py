#!/usr/bin/env python3
"""Generate the golden HotSketch traces in tests/golden/*.npz from the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference): `make -C oracle ref` compiles the
reference's tricks/sketch.cpp, from where it lies, into oracle/_ref/sklib_probe.so (private state made
readable by oracle/ref_probe.cpp).  Each trace runs in its own subprocess (the reference keeps one
global sketch per process, sketch.cpp:297, and init() does not reset its other globals) with
MALLOC_PERTURB_ set (the reference constructor tests an uninitialised member, sketch.cpp:22,50).

A fixture is data only: the id / score streams fed through the reference's C ABI (sketch.cpp:300-313),
the ins[] / que[] arrays it returned, and dumps of every slot of its state.

    python tests/golden/make_golden.py            # regenerate all traces
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

KAGGLE_COUNTS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992,
                 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]   # tricks/sketchtest.py:41-45

KIND = {"insert_val": 0, "insert": 1, "query": 2}


def zipf_ids(rng, n, space, a=1.05, lo=1):
    x = (rng.zipf(a, size=n) - 1) % space
    return (x + lo).astype(np.int64)


def build_calls(spec):
    """Deterministic (kind, ids, scores) list from a trace spec."""
    rng = np.random.default_rng(spec["seed"])
    calls = []
    g = spec["gen"]
    if g == "known_answer":     # SURVEY.md 8(a) known-answer trace
        seq = [([7, 7, 7, 8, 9], [1, 1, 1, 1, 1]), ([7, 8, 8, 9, 1007], [.5, 1.5, 1, 1, 1]),
               ([9, 9, 1007, 1007, 7], [1, 1, 1, 1, 1])]
        for ids, v in seq:
            calls.append(("insert_val", np.array(ids, np.int64), np.array(v, np.float32)))
            calls.append(("query", np.array(ids, np.int64), None))
        return calls
    if g == "kaggle":           # cfg1-shaped: 26 fields x batch, global id = local + field offset
        offs = np.concatenate([[0], np.cumsum(KAGGLE_COUNTS)[:-1]])
        for step in range(spec["steps"]):
            for f, cnt in enumerate(KAGGLE_COUNTS):
                loc = (rng.zipf(1.05, size=spec["batch"]) - 1) % cnt
                if f == 0:
                    loc = np.maximum(loc, 1)
                ids = loc + offs[f]
                gn = rng.random(spec["batch"], dtype=np.float32) + np.float32(0.01)
                v = (gn * np.float32(spec["batch"]) / gn.sum(dtype=np.float32)).astype(np.float32)
                calls.append(("query", ids, None))
                calls.append(("insert_val", ids, v))
        calls.append(("query", ids, None))
        return calls
    for c in range(spec["calls"]):
        n = spec["batch"]
        if g == "zipf":
            ids = zipf_ids(rng, n, spec["space"], spec.get("a", 1.05), spec.get("lo", 1))
        elif g == "uniform":
            ids = rng.integers(spec.get("lo", 1), spec["space"] + spec.get("lo", 1), size=n).astype(np.int64)
        elif g == "mixed":      # half zipf head, half uniform tail: forces LRU churn and evictions
            ids = np.where(rng.random(n) < 0.5, zipf_ids(rng, n, spec["space"], 1.2),
                           rng.integers(1, spec["space"] + 1, size=n)).astype(np.int64)
        else:
            raise ValueError(g)
        kind = spec.get("kind", "insert_val")
        if kind == "alternate":
            kind = "insert" if c % 3 == 2 else "insert_val"
        if kind == "insert_val":
            sc = spec.get("score", "u02")
            if sc == "u02":
                v = (rng.random(n, dtype=np.float32) * np.float32(2.0)).astype(np.float32)
            elif sc == "one":
                v = np.ones(n, np.float32)
            elif sc == "heavy":   # occasional large scores: jumps over the threshold in one add
                v = (rng.random(n, dtype=np.float32) * np.float32(2.0)).astype(np.float32)
                v[rng.random(n) < 0.02] *= np.float32(50.0)
            calls.append((kind, ids, v))
        else:
            calls.append((kind, ids, None))
        if spec.get("query_every", 1) and (c % spec.get("query_every", 1) == 0):
            calls.append(("query", ids, None))
    return calls


TRACES = {
    # name: spec.  dump = "all" | "every:K" | list of call indices after which the full state is stored (-1 = last)
    "known_answer": dict(gen="known_answer", seed=0, lim=100, thr=2, adjust=1, alpha=1.0, dump="all"),
    "zipf_thr2_count": dict(gen="zipf", seed=1, lim=500, thr=2, adjust=1, alpha=1.0, calls=40, batch=64,
                            space=5000, kind="insert", dump="every:9"),
    "uniform_val_reset": dict(gen="uniform", seed=2, lim=100, thr=3, adjust=1, alpha=1.0, calls=100, batch=128,
                              space=300, kind="insert_val", score="u02", query_every=4,
                              dump=[0, 10, 40, 100, -1]),
    "flip_decay": dict(gen="zipf", seed=3, lim=600, thr=50, adjust=1, alpha=3.0, calls=60, batch=128,
                       space=20000, kind="insert_val", score="heavy", query_every=3, dump="every:7"),
    "thr1_quirk": dict(gen="zipf", seed=4, lim=400, thr=1, adjust=1, alpha=1.0, calls=40, batch=96,
                       space=3000, kind="insert_val", score="u02", query_every=2, dump="every:7"),
    "no_adjust_drain": dict(gen="zipf", seed=5, lim=120, thr=3, adjust=0, alpha=1.0, calls=60, batch=128,
                            space=4000, a=1.3, kind="alternate", score="u02", query_every=5, dump=[0, 5, 20, 40, -1]),
    "id_zero": dict(gen="uniform", seed=6, lim=200, thr=2, adjust=1, alpha=1.0, calls=30, batch=64,
                    space=300, lo=0, kind="insert_val", score="one", query_every=2, dump="every:5"),
    "evict_promote": dict(gen="mixed", seed=7, lim=800, thr=40, adjust=1, alpha=1.001, calls=100, batch=512,
                          space=2000, kind="insert_val", score="u02", query_every=10, dump=[0, 30, 60, -1]),
    "reset_mmap_ties": dict(gen="uniform", seed=9, lim=3000, thr=2, adjust=1, alpha=1.0, calls=24, batch=2048,
                            space=9000, kind="alternate", score="one", query_every=6, dump=[0, 8, 16, -1]),
    "reset_ties_lru_first": dict(gen="uniform", seed=10, lim=1500, thr=2, adjust=1, alpha=1.0, calls=24, batch=1024,
                                 space=4500, kind="alternate", score="one", query_every=6, dump=[0, 8, 16, -1],
                                 mmap_threshold=4096),   # both tables mmapped -> LRU table below the main table
    "big_batch": dict(gen="zipf", seed=8, lim=5000, thr=8, adjust=1, alpha=1.0000005, calls=8, batch=16384,
                      space=300000, kind="insert_val", score="u02", query_every=4, dump=[0, 3, -1]),
    "kaggle_cfg1": dict(gen="kaggle", seed=123, lim=13505, thr=3, adjust=1, alpha=1.0000005, steps=6, batch=128,
                        dump=[-1]),
}


def worker(spec_json, out_path):
    spec = json.loads(spec_json)
    from oracle.cafe_oracle import read_state
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "sklib_probe.so"))
    i32p, f32p = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float)
    L.init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    L.init.restype = None
    for fn in ("batch_query", "batch_insert"):
        getattr(L, fn).argtypes = [i32p, ctypes.c_int]
        getattr(L, fn).restype = i32p
    L.batch_insert_val.argtypes = [i32p, f32p, ctypes.c_int]
    L.batch_insert_val.restype = i32p
    L.init(spec["lim"], spec["thr"], spec["adjust"], spec["alpha"])
    calls = build_calls(spec)
    dump = spec["dump"]
    if dump == "all":
        dump_at = set(range(len(calls)))
    elif isinstance(dump, str) and dump.startswith("every:"):
        dump_at = set(range(0, len(calls), int(dump[6:]))) | {len(calls) - 1}
    else:
        dump_at = {d % len(calls) for d in dump}
    kinds, off, ids_all, sc_all, out_all, dumps = [], [0], [], [], [], {}
    for k, (kind, ids, v) in enumerate(calls):
        assert len(ids) <= 16384          # the reference's static result buffers, sketch.cpp:10
        idc = np.ascontiguousarray(ids.astype(np.int64).astype(np.uint32).view(np.int32))
        p = idc.ctypes.data_as(i32p)
        if kind == "insert_val":
            v = np.ascontiguousarray(v, np.float32)
            r = L.batch_insert_val(p, v.ctypes.data_as(f32p), len(idc))
        elif kind == "insert":
            r = L.batch_insert(p, len(idc))
            v = np.zeros(len(idc), np.float32)
        else:
            r = L.batch_query(p, len(idc))
            v = np.zeros(len(idc), np.float32)
        out = np.ctypeslib.as_array(r, shape=(len(idc),)).copy()
        kinds.append(KIND[kind]); off.append(off[-1] + len(idc))
        ids_all.append(idc); sc_all.append(v); out_all.append(out.astype(np.int32))
        if k in dump_at:
            dumps[k] = read_state(L, "probe_")
    arrs = dict(
        lim=spec["lim"], thr=spec["thr"], adjust=spec["adjust"], alpha=np.float64(spec["alpha"]),
        lru_first=int(L.probe_lru_first()), kinds=np.array(kinds, np.int8), offsets=np.array(off, np.int64),
        ids=np.concatenate(ids_all), scores=np.concatenate(sc_all), outs=np.concatenate(out_all),
        dump_calls=np.array(sorted(dumps), np.int32))
    for k, st in dumps.items():
        for name, val in st.items():
            arrs[f"st{k}_{name}"] = np.asarray(val)
    np.savez_compressed(out_path, **arrs)


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    only = sys.argv[1:]
    for name, spec in TRACES.items():
        if only and name not in only:
            continue
        out = os.path.join(HERE, name + ".npz")
        env = dict(os.environ, MALLOC_PERTURB_="165")
        if "mmap_threshold" in spec:
            env["MALLOC_MMAP_THRESHOLD_"] = str(spec["mmap_threshold"])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker", json.dumps(spec), out],
                              env=env, stdout=subprocess.DEVNULL)
        print(f"{name}: {os.path.getsize(out) / 1024:.1f} KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], sys.argv[3])
    else:
        main()
