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


@pytest.fixture(params=['f32', 'f16x3', 'bf16x6', 'mixed', 'bf16x3'])
def math_mode(request):
    """Runs a GPU test once per arithmetic of the fused edge GEMMs (include/gpe_hip.h gpe_math_set) and restores the
    library default afterwards."""
    import gpe_amd
    prev = gpe_amd.set_math(request.param)
    # the f16x3 size gate (launches under 32768 rows run the exact kernels) is lifted: the fixtures are small and the point of
    # the parametrisation is to run the fp16-pipe kernels
    gate = gpe_amd.set_f16x3_min_rows(0)
    yield request.param
    gpe_amd.set_f16x3_min_rows(gate)
    gpe_amd.set_math(prev)
