import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, 'tests', 'golden')
CONFIGS = ('rp20', 'a36', 'm36', 't228', 'g30', 'd38', 'q38')       # d38, q38: rank-5 tensors (dynamic-T, T4 models)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')


class Golden(object):
    """One tests/golden/<name>.npz: tensors, f/Df values and stepper outputs captured from the reference."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
        self.meta = json.loads(bytes(self.z['meta_json']).decode())
        self.ndim = int(self.z['ndim'])

    def __getitem__(self, key):
        return self.z[key]

    def __contains__(self, key):
        return key in self.z.files


_cache = {}


def load_golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


@pytest.fixture(params=CONFIGS)
def golden(request):
    return load_golden(request.param)


@pytest.fixture(params=('rp20', 'a36', 'm36', 'g30'))
def golden_small(request):
    return load_golden(request.param)


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-300))


RK4 = dict(c=np.array([0., 0.5, 0.5, 1.]), b=np.array([1. / 6, 1. / 3, 1. / 3, 1. / 6]),
           a=np.array([[0., 0, 0, 0], [0.5, 0, 0, 0], [0, 0.5, 0, 0], [0, 0, 1., 0]]))
