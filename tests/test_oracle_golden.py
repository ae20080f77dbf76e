"""The CPU oracle (oracle/cafe_oracle.c) must reproduce the reference's own traces bit for bit.

The fixtures in tests/golden/ were recorded from /root/reference/tricks/sketch.cpp itself
(tests/golden/make_golden.py); this test pins the restatement that the GPU parity tests trust."""
import numpy as np
import pytest

from conftest import golden_names
import golden_util as gu


@pytest.mark.parametrize("name", golden_names())
def test_oracle_replays_reference_trace(name, oracle_mod):
    g = gu.load(name)
    sk = oracle_mod.Oracle(int(g["lim"]), int(g["thr"]), int(g["adjust"]), float(g["alpha"]), int(g["lru_first"]))
    bad = gu.replay(g, sk, states_equal=oracle_mod.states_equal)
    assert not bad, "\n".join(bad)


def test_known_answer_table(oracle_mod):
    """init(100, 2, 1, 1.0), three calls."""
    sk = oracle_mod.Oracle(100, 2, 1, 1.0)
    assert sk.insert_val([7, 7, 7, 8, 9], [1, 1, 1, 1, 1]).tolist() == [0, 1, 0, 0, 0]
    assert sk.query([7, 7, 7, 8, 9]).tolist() == [-1, -1, -1, 8, 9]
    assert sk.insert_val([7, 8, 8, 9, 1007], [.5, 1.5, 1, 1, 1]).tolist() == [0, 1, 0, 1, 0]
    assert sk.query([7, 8, 8, 9, 1007]).tolist() == [-1, -2, -2, -3, 1007]
    assert sk.insert_val([9, 9, 1007, 1007, 7], [1, 1, 1, 1, 1]).tolist() == [0, 0, 1, 0, 0]
    assert sk.query([9, 9, 1007, 1007, 7]).tolist() == [-3, -3, -4, -4, -1]


def _replay_stats(oracle_mod, name):
    g = gu.load(name)
    sk = oracle_mod.Oracle(int(g["lim"]), int(g["thr"]), int(g["adjust"]), float(g["alpha"]), int(g["lru_first"]))
    assert not gu.replay(g, sk, check_state=False)
    return g, sk.state(), sk.stats()


def test_goldens_cover_the_hard_paths(oracle_mod):
    """The traces must actually exercise reset() under both tie orders, flips, an empty free list,
    hot slots parked in the LRU table (threshold <= 1), evictions into the main table and failed moves."""
    g, st, s = _replay_stats(oracle_mod, "uniform_val_reset")
    assert st["resets"] > 0 and st["thr"] != np.float32(3.0)
    g, st, s = _replay_stats(oracle_mod, "reset_ties_lru_first")
    assert st["resets"] > 0 and int(g["lru_first"]) == 1
    g, st, s = _replay_stats(oracle_mod, "reset_mmap_ties")
    assert st["resets"] > 0 and int(g["lru_first"]) == 0
    g, st, s = _replay_stats(oracle_mod, "flip_decay")
    assert st["flip"] == 1 and s["move_fail"] > 0 and s["push_back"] > 0 and s["evict_move"] > 0
    g, st, s = _replay_stats(oracle_mod, "no_adjust_drain")
    assert st["resets"] == 0
    g, st, s = _replay_stats(oracle_mod, "thr1_quirk")
    assert s["pop_empty"] > 0 and s["push_back"] > 0
    assert any((gu.state_at(g, int(k))["ldic"] != 0).any() for k in g["dump_calls"])
    g, st, s = _replay_stats(oracle_mod, "evict_promote")
    assert s["evict_move"] > 0 and s["promote"] > 0
