"""Diagnostic (not collected by pytest): parallel-insert statistics on the golden traces and fresh streams."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import golden_util as gu
from conftest import golden_names
from oracle import cafe_oracle as co
from cafe_amd.sketch import HotSketch

for name in golden_names():
    g = gu.load(name)
    sk = HotSketch(int(g["lim"]), int(g["thr"]), int(g["adjust"]), float(g["alpha"]), int(g["lru_first"]))
    ora = co.Oracle(int(g["lim"]), int(g["thr"]), int(g["adjust"]), float(g["alpha"]), int(g["lru_first"]))
    bad = None
    for k, kind, ids, sc, out in gu.calls(g):
        if kind == "query":
            got = sk.query(ids).cpu().numpy(); ora.query(ids)
        else:
            v = sc if kind == "insert_val" else None
            got = sk.insert_val(ids, v).cpu().numpy(); ora.insert_val(ids, v)
        if not np.array_equal(got, out) and bad is None:
            bad = f"call {k} {kind} first diff at {int(np.nonzero(got != out)[0][0])}"
            d = co.states_equal(sk.state(), ora.state())
            bad += " | state: " + "; ".join(d[:3])
            break
    d = co.states_equal(sk.state(), ora.state()) if bad is None else []
    print(f"{name:24s} {'OK' if not bad and not d else 'BAD'} {bad or ''} {d[:2]} {sk.counters()}", flush=True)
