import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")

# no pretrained VGG16 weights exist offline: the trainer tests run on the seeded-random stand-in, and say so explicitly
# (IPSR.initialize refuses to build one otherwise; tests/test_host_model.py checks the refusal)
os.environ.setdefault("IPSR_ALLOW_RANDOM_VGG", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
