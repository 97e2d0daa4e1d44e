import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def golden_names(prefix=''):
    """Without a prefix: the per-tile goldens g1_..g5_ (edge-update g6_* and directory pm_* goldens
    have their own tests and are selected by prefix)."""
    names = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith('.npz'))
    if prefix:
        return [n for n in names if n.startswith(prefix)]
    return [n for n in names if n[:3] in ('g1_', 'g2_', 'g3_', 'g4_', 'g5_')]


def load_golden(name):
    import numpy as np
    d = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    g = {k: d[k] for k in d.files}
    g['kwargs'] = dict(eval(str(g.pop('kwargs_repr'))))
    return g
