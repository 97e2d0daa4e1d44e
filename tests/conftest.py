import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def golden_names(prefix='', exclude='g6_'):
    """Per-tile goldens by default; the edge-update goldens (g6_*) have their own tests."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith('.npz') and f.startswith(prefix)
                  and not (exclude and f.startswith(exclude) and not prefix))


def load_golden(name):
    import numpy as np
    d = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    g = {k: d[k] for k in d.files}
    g['kwargs'] = dict(eval(str(g.pop('kwargs_repr'))))
    return g
