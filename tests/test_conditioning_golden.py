"""Host-side elevation conditioning (pydem_amd/conditioning.py) against arrays captured from the unmodified
reference after each stage: calc_fill_pit_artifacts, calc_fill_flats, calc_pit_drain_paths
(dem_processing.py:396-579).  Bit-exact, CPU only (the conditioning is numpy/scipy code)."""
import warnings

import numpy as np
import pytest

from conftest import golden_names, load_golden
from pydem_amd import conditioning


def _cases():
    return [n for n in golden_names() + golden_names('g7_') if 'elev_drained' in load_golden(n) or 'elev_filled' in load_golden(n)]


@pytest.mark.parametrize('name', _cases())
def test_conditioning_matches_reference(name):
    g = load_golden(name)
    kw = g['kwargs']
    elev = g['in_elev'].copy()
    if kw.get('fill_flats', True):
        art = conditioning.fill_pit_artifacts(elev, kw.get('maximum_pit_area', 32.0))
        assert art.dtype == g['elev_artifacts'].dtype
        assert np.array_equal(art, g['elev_artifacts'])
        filled = conditioning.fill_flats(elev, kw.get('maximum_pit_area', 32.0))
        assert np.array_equal(filled, g['elev_filled'], equal_nan=True)
        elev = filled
    if kw.get('drain_pits_path', True):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            out, _, _ = conditioning.pit_drain_paths(np.array(elev), g['in_dX'], g['in_dY'])
        assert out.dtype == g['elev_drained'].dtype
        assert np.array_equal(out, g['elev_drained'], equal_nan=True)
