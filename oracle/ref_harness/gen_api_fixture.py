#!/usr/bin/env python
"""Capture the API surface of the reference's DEMProcessor / ProcessManager (option names and their
default values as seen on an instance, public method names, the dem_proc_kwargs whitelist) into
tests/golden/ref_api_surface.json.  Run through run.sh (imports the unmodified reference)."""
import inspect
import json
import os
import sys

from load_reference import load_reference

pydem = load_reference()
import numpy as np  # noqa: E402
from pydem.dem_processing import DEMProcessor  # noqa: E402
from pydem import process_manager  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(REPO, 'tests', 'golden', 'ref_api_surface.json')


def plain(v):
    if isinstance(v, (bool, int, str)) or v is None:
        return v
    if isinstance(v, float):
        return 'inf' if v == float('inf') else v
    return None


def main():
    dp = DEMProcessor(elev=np.arange(25, dtype=float).reshape(5, 5) + 1.0, dX=2.0, dY=3.0)
    options = {}
    for name in sorted(vars(type(dp))):
        if name.startswith('_'):
            continue
        attr = getattr(type(dp), name)
        if callable(attr) or isinstance(attr, property):
            continue
        val = getattr(dp, name)
        if isinstance(val, (bool, int, float, str)) or val is None:
            options[name] = plain(val)
    methods = sorted(n for n, f in inspect.getmembers(DEMProcessor, predicate=inspect.isfunction) if not n.startswith('_'))
    pm_methods = sorted(n for n, f in inspect.getmembers(process_manager.ProcessManager, predicate=inspect.isfunction)
                        if not n.startswith('_'))
    rec = {
        'source': 'pydem/dem_processing.py:98-258, pydem/process_manager.py (captured from an instance of the unmodified reference)',
        'demprocessor_options': options,
        'demprocessor_methods': methods,
        'processmanager_methods': pm_methods,
        'dX_after_scalar_ctor': dp.dX.tolist(), 'dY_after_scalar_ctor': dp.dY.tolist(),
        'dX2_after_scalar_ctor': np.asarray(dp.dX2).tolist(), 'dY2_after_scalar_ctor': np.asarray(dp.dY2).tolist(),
    }
    json.dump(rec, open(OUT, 'w'), indent=1, sort_keys=True)
    print(json.dumps(rec, indent=1, sort_keys=True)[:3000])


if __name__ == '__main__':
    main()
