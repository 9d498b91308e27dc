import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def hmm():
    """One engine handle for the whole GPU session (tests run in one process).  Haplotype runs are forced on:
    left to itself (stream_mode -1) the engine sends calls as small as the tests' to the one-haplotype pass,
    and the tests are after the stream pass first of all.  Tests that want the automatic choice set -1."""
    from gatk_amd.pairhmm import HipLoglessPairHMM
    h = HipLoglessPairHMM(0)
    h.set_option("stream_mode", 1)
    yield h
    h.close()
