import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


@pytest.fixture(autouse=True)
def _process_wide_switches_off():
    """Process-wide switches a training loop may have left on (side-stream overlap, deferred final reductions -- both set by the
    optimiser / main() of an earlier test in the same process): every test starts from the defaults."""
    mods = sys.modules
    if 'vae_gp_ode_amd.vae_ops' in mods:
        mods['vae_gp_ode_amd.vae_ops'].set_deferred_reductions(False)
    if 'vae_gp_ode_amd.ops' in mods:
        mods['vae_gp_ode_amd.ops'].set_overlap(False)
    yield
