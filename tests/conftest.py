import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def partition_golden():
    return load_golden("partition_golden.json")


@pytest.fixture(scope="session")
def queries_golden():
    return load_golden("queries.json")


@pytest.fixture(scope="session")
def counts_golden():
    return load_golden("canonical_counts.json")
