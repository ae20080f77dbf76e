#!/usr/bin/env python3
"""Generate tests/golden/pinsage_state.bin (+ pinsage_state.npz) from the REFERENCE: TOIS_revision/PinSAGE/sketch.cpp
compiled here by oracle/Makefile (`make -C oracle ref_more`), driven through its C entry points (init,
batch_insert_start, batch_insert_val, batch_query, save_state; :404-446).  The .bin file is what its save_state (:333-366)
writes; the .npz holds the constructor arguments, the number of insert calls and, for a probe set of ids, what the
reference's batch_query answers right before the save -- so that a reader of the byte layout can be checked against the
reference's own view of the same state.  Data only.  Runs only in the build container.

    python tests/golden/make_pinsage_state.py
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def worker(out_bin, out_npz):
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "pinsage_sklib.so"))
    u32p, f32p, i32p = ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    L.init.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    L.batch_insert_val.argtypes = [u32p, f32p, ctypes.c_int]; L.batch_insert_val.restype = i32p
    L.batch_query.argtypes = [u32p, ctypes.c_int]; L.batch_query.restype = i32p
    L.save_state.argtypes = [ctypes.c_char_p]
    lim, thr, pad, adjust, alpha = 400, 3, -1, 1, 1.01
    L.init(lim, thr, pad, adjust, alpha)
    rng = np.random.default_rng(99)
    calls = 40
    for _ in range(calls):
        ids = ((rng.zipf(1.2, size=300) - 1) % 3000 + 1).astype(np.uint32)
        v = (rng.random(300) * 4).astype(np.float32)
        L.batch_insert_start()
        L.batch_insert_val(ids.ctypes.data_as(u32p), v.ctypes.data_as(f32p), len(ids))
    probe = np.arange(1, 3001, dtype=np.uint32)
    que = np.ctypeslib.as_array(L.batch_query(probe.ctypes.data_as(u32p), len(probe)), shape=(len(probe),)).copy()
    L.save_state(out_bin.encode())
    np.savez_compressed(out_npz, lim=lim, thr=thr, adjust=adjust, alpha=alpha, calls=calls, probe=probe, que=que.astype(np.int32))


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_more"], stdout=subprocess.DEVNULL)
    out_bin, out_npz = os.path.join(HERE, "pinsage_state.bin"), os.path.join(HERE, "pinsage_state.npz")
    env = dict(os.environ, MALLOC_PERTURB_="165")        # the reference reads memory it never initialised
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--worker", out_bin, out_npz], env=env,
                          stdout=subprocess.DEVNULL)
    print(f"pinsage_state.bin: {os.path.getsize(out_bin)} bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], sys.argv[3])
    else:
        main()
