import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
if str(REPO / "tests") not in sys.path:
    sys.path.insert(0, str(REPO / "tests"))
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
# tests flip the library's A/B switches and test hooks through GYP_* variables: GypsumEngine forwards them (gyp_debug_set) only
# under this opt-in -- the library itself reads no environment
os.environ.setdefault("GYP_TEST_HOOKS", "1")
# oracle worker processes inherit this: one BLAS thread each (tests/survey_worker.py says why)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def engine_factory():
    """GypsumEngine per (fs, N).  No fallback: a missing library or device is an error, not a skip."""
    from gypsum_amd.engine import GypsumEngine

    cache = {}

    def make(fs: int, n: int):
        key = (fs, n)
        if key not in cache:
            eng = GypsumEngine(0)
            eng.set_stream_format(fs, n)
            cache[key] = eng
        return cache[key]

    yield make
    for eng in cache.values():
        eng.close()
