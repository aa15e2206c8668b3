import gzip
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

DATA = os.path.join(HERE, "data")
GOLDEN = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_gz(name):
    with gzip.open(os.path.join(DATA, name), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def example_library_text():
    return read_gz("library.fasta.gz")


@pytest.fixture(scope="session")
def example_reads():
    names = ["sequence", "zero.sequence", "diff.sequence", "offset", "offset_clipped"]
    return {n: read_gz(n + (".fastq.gz")) for n in names}
