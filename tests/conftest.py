import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def specs():
    import json
    with open(os.path.join(GOLDEN, 'g0_state_dict_keys.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def state_dicts(specs):
    import ffrnet_amd
    sd_e = ffrnet_amd.synth.synth_state_dict(specs['encoder'], seed=0)
    sd_r = ffrnet_amd.synth.synth_state_dict(specs['recnet'], seed=0)
    return sd_e, sd_r
