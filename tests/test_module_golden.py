"""The oracle's tensor half (numpy restatement of tricks/sk_embedding_bag.py in oracle/cafe_oracle.py) and
its C sketch, replayed against steps recorded from the reference's own SKEmbeddingBag module
(tests/golden/module_*.npz, made by tests/golden/make_module_golden.py).  CPU only.

Bars: routing after every step bit-exact; forward rows bit-exact (a gather); tables within 1e-5 relative."""
import numpy as np
import pytest

from conftest import module_golden_names
import golden_util as gu

RTOL = 1e-5


def _close(a, b, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= RTOL, f"{what}: max rel err {err:.3e}"


@pytest.mark.parametrize("name", module_golden_names())
def test_oracle_matches_reference_module(name, oracle_mod):
    om = oracle_mod
    g = gu.load(name)
    counts, hs, offs = g["counts"], [int(h) for h in g["hash_sizes"]], g["offsets"]
    F, D, hotn, B, lr = len(counts), int(g["dim"]), int(g["hotn"]), int(g["batch"]), float(g["lr"])
    sk = om.Oracle(hotn, int(g["thr"]), 1, float(g["alpha"]), 1)
    w_hot = np.zeros((hotn, D), np.float32)
    w_cold = [g[f"w_cold0_{f}"].copy() for f in range(F)]
    for step in range(int(g["steps"])):
        ids = g[f"ids_{step}"]
        gids = [ids[f] + offs[f] for f in range(F)]
        que = [sk.query(gids[f]) for f in range(F)]
        for f in range(F):
            out, _ = om.ref_forward(que[f], w_hot, w_cold[f], hs[f])
            _close(out, g[f"out_{step}"][f], f"{name} step {step} field {f} forward")
        gout = g[f"gout_{step}"]
        for f in range(F):                                    # insert_grad per field, dlrm_s_pytorch.py:734-738
            ins = sk.insert_val(gids[f], om.ref_scores(gout[f]))
            om.ref_migrate(ins, sk.query(gids[f]), ids[f], w_hot, w_cold[f], hs[f])
        acc_h = np.zeros(w_hot.shape, np.float64)             # SGD, hot table shared by the fields
        for f in range(F):
            q = que[f].astype(np.int64)
            hot = q < 0
            np.add.at(acc_h, (-q[hot]) % hotn, gout[f][hot].astype(np.float64))
            acc_c = np.zeros(w_cold[f].shape, np.float64)
            np.add.at(acc_c, q[~hot] % hs[f], gout[f][~hot].astype(np.float64))
            w_cold[f] -= (lr * acc_c).astype(np.float32)
        w_hot -= (lr * acc_h).astype(np.float32)
        for f in range(F):
            assert np.array_equal(sk.query(gids[f]), g[f"que_after_{step}"][f]), f"{name} step {step} field {f}: routing"
            _close(w_cold[f], g[f"w_cold_{step}_{f}"], f"{name} step {step} cold table {f}")
        _close(w_hot, g[f"w_hot_{step}"], f"{name} step {step} hot table")
