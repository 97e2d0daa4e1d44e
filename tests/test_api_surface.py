"""API surface of the drop-in against the reference's (tests/golden/ref_api_surface.json, captured from an instance
of the unmodified reference by oracle/ref_harness/gen_api_fixture.py): every DEMProcessor option with the same default
(dem_processing.py:105-154), the scalar dX/dY normalisation of the constructor (:229-242), and the public methods of
DEMProcessor / ProcessManager that belong to the path.  No device work."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = json.load(open(os.path.join(HERE, 'golden', 'ref_api_surface.json')))

# reference members that are deliberately not part of the drop-in (outside SURVEY section 8): trait plumbing, the
# file-name helpers / loaders of the reference's on-disk layout, the multiprocessing pool
NOT_PROVIDED_DP = {'trait_names', 'get_fn', 'get_full_fn', 'load_array', 'load_direction', 'load_elevation', 'load_slope',
                   'load_uca'}
NOT_PROVIDED_PM = {'trait_names', 'queue_processes'}
# state attributes of the reference instance that are results, not options
RESULT_ATTRS = {'A', 'direction', 'done', 'flats', 'mag', 'proportion', 'section', 'twi', 'uca'}


def _dp():
    from pydem_amd import DEMProcessor
    return DEMProcessor(elev=np.arange(25, dtype=float).reshape(5, 5) + 1.0, dX=2.0, dY=3.0)


@pytest.mark.parametrize('name', sorted(set(REF['demprocessor_options']) - RESULT_ATTRS))
def test_option_exists_with_the_reference_default(name):
    dp = _dp()
    assert hasattr(dp, name), "DEMProcessor.%s missing" % name
    want = REF['demprocessor_options'][name]
    got = getattr(dp, name)
    if want == 'inf':
        assert got == float('inf')
    elif want is None:
        assert got is None
    else:
        assert type(got) in (type(want), float, int, bool) and got == want, (name, got, want)


def test_scalar_spacing_is_normalised_like_the_reference():
    dp = _dp()
    for nm in ('dX', 'dY', 'dX2', 'dY2'):
        assert np.array_equal(np.asarray(getattr(dp, nm), float), np.asarray(REF[nm + '_after_scalar_ctor'], float)), nm


def test_public_methods_of_the_path_exist():
    from pydem_amd import DEMProcessor
    from pydem_amd.process_manager import ProcessManager
    missing = [m for m in REF['demprocessor_methods'] if m not in NOT_PROVIDED_DP and not callable(getattr(DEMProcessor, m, None))]
    assert not missing, "DEMProcessor lacks %r" % missing
    missing = [m for m in REF['processmanager_methods'] if m not in NOT_PROVIDED_PM and not callable(getattr(ProcessManager, m, None))]
    assert not missing, "ProcessManager lacks %r" % missing


@pytest.mark.parametrize('kw', [dict(drain_flats=True), dict(drain_pits_spill=True)])
def test_unimplemented_drainage_alternatives_fail_loudly(kw):
    """drain_flats / drain_pits_spill only act when drain_pits is off (dem_processing.py:1094-1123); the device path
    does not implement them and must say so instead of returning the plain result."""
    from pydem_amd import DEMProcessor
    dp = DEMProcessor(elev=np.arange(25, dtype=float).reshape(5, 5) + 1.0, dX=2.0, dY=3.0, fill_flats=False,
                      drain_pits_path=False, drain_pits=False, **kw)
    dp.mag = np.ones((5, 5)); dp.direction = np.ones((5, 5)); dp.flats = np.zeros((5, 5), bool)   # skip the device stencil
    with pytest.raises(NotImplementedError):
        dp.run_uca()


@pytest.mark.parametrize('kw', [dict(dX=30.0, dY=-30.0), dict(dX=0.0, dY=30.0), dict(dX=float('nan'), dY=1.0)])
def test_non_positive_spacing_is_refused(kw):
    """Cell sizes must be finite and > 0: the device facet tests cross-multiply with them (the reference would
    silently return mirrored / infinite slopes)."""
    from pydem_amd import DEMProcessor
    with pytest.raises(ValueError):
        DEMProcessor(elev=np.arange(25, dtype=float).reshape(5, 5) + 1.0, fill_flats=False, **kw)
