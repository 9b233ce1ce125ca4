import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
MODELS = os.path.join(GOLDEN, "models")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_models():
    """Lazily loaded oracle models keyed by fixture file name."""
    from oracle import oracle as O
    from goldens import model_file
    cache = {}

    def get(name):
        name = model_file(name)
        if name not in cache:
            cache[name] = O.Model(os.path.join(MODELS, name))
        return cache[name]
    return get
