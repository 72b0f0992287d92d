import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# every ctx of the test processes (and of the CLI binaries they start) checks, before a coverage pass that skips its zero fill, that the arena its
# last readers were to leave clean IS all zero (round 6: self-cleaning coverage arena); the library reads the environment in pantax_hip_init
os.environ.setdefault("PANTAX_COV_ARENA_VERIFY", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def set_opt():
    """set_opt(eng, name, value): a library option (pantax_hip_set_option) for the duration of the test; value None = default.
    The library reads the environment only at pantax_hip_init, so tests switch its in-tree paths through the ctx."""
    done = []

    def _set(eng, name, value):
        eng.set_option(name, value)
        done.append((eng, name))
    yield _set
    for eng, name in done:
        try:
            if eng.ctx:
                eng.set_option(name, None)
        except Exception:   # noqa: BLE001
            pass
