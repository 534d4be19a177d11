import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'poisson-gpfa_amd')
for path in (ROOT, PKG):
    if path not in sys.path:
        sys.path.insert(0, path)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope='session')
def c1():
    """Config-1 dataset (30 neurons, 3 latents, T=100, 20 trials) captured from the reference."""
    g = load_golden('c1_dataset.npz')
    out = {k: g[k] for k in g.files}
    out['Ys'] = [out['Y'][r].astype(np.float64) for r in range(out['Y'].shape[0])]
    out['init'] = {'C': out['init_C'].copy(), 'd': out['init_d'].copy(), 'tau': out['init_tau'].copy()}
    out['binSize'] = float(out['binSize'])
    return out


class Experiment:
    """Duck-typed stand-in for the reference's util.dataset (engine.py:32-38)."""

    def __init__(self, Ys, binSize):
        self.data = [{'Y': np.asarray(y, dtype=np.float64)} for y in Ys]
        self.ydim, self.T = self.data[0]['Y'].shape
        self.binSize = binSize
        self.trialDur = self.T * binSize
        self.numTrials = len(self.data)


@pytest.fixture(scope='session')
def c1_experiment(c1):
    return Experiment(c1['Ys'], c1['binSize'])
