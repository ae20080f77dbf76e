import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def golden_names():
    g = os.path.join(ROOT, "tests", "golden")
    other = ("module_", "ae_", "aemod_", "loader_", "pinsage_")       # fixtures of other tests (module steps, AE generation, loader, state file)
    return sorted(f[:-4] for f in os.listdir(g) if f.endswith(".npz") and not f.startswith(other))


def module_golden_names():
    """Module-level steps recorded from the reference's SKEmbeddingBag (tests/golden/make_module_golden.py)."""
    g = os.path.join(ROOT, "tests", "golden")
    return sorted(f[:-4] for f in os.listdir(g) if f.endswith(".npz") and f.startswith("module_"))


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import cafe_oracle
    cafe_oracle.build()
    return cafe_oracle
