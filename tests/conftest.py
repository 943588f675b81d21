import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the C half of the oracle is test infrastructure: (re)build it if it did not travel
    if not os.path.exists(os.path.join(REPO, 'oracle', 'libgpe_oracle.so')):
        subprocess.check_call(['make', '-C', os.path.join(REPO, 'oracle')])


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(REPO, 'tests', 'golden')
