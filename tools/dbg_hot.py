import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from oracle import cafe_oracle as om
from cafe_amd.sketch import HotSketch
om.build()
for table in (0, 1):
    lim, thr, batch, space = 2000, 4, 4096, 50000
    ora = om.Oracle(lim, thr, 1, 1.0000005, 1)
    sk = HotSketch(lim, thr, 1, 1.0000005, 1)
    sk.set_option("hot_table", table)
    ok = True
    for c in range(6):
        rng = np.random.default_rng(1000 + c)
        ids = ((rng.zipf(1.08, size=batch) - 1) % space + 1).astype(np.int64)
        v = (rng.random(batch, dtype=np.float32) * np.float32(2.0)).astype(np.float32)
        want = ora.insert_val(ids, v)
        got = sk.insert_val(ids, v).cpu().numpy()
        bad = om.states_equal(sk.state(), ora.state())
        print("table", table, "call", c, "ins equal", np.array_equal(got, want), "state diffs", bad[:3], flush=True)
        if bad:
            a, b = sk.state(), ora.state()
            d = np.nonzero(a["mcnt"] != b["mcnt"])[0]
            print("  n diff", len(d), "first", d[:10], a["mcnt"][d[:5]], b["mcnt"][d[:5]], "dic", a["mdic"][d[:5]])
            break
