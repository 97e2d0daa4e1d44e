"""Minimal stand-in for `traittypes` (not installed). TEST INFRASTRUCTURE ONLY."""
import numpy as np
from traitlets import TraitType, _Undefined


class Array(TraitType):
    def __init__(self, default_value=_Undefined, allow_none=False, **kwargs):
        super().__init__(default_value)

    def _static_default(self):
        if self.default_value is _Undefined or self.default_value is None:
            return None
        return np.array(self.default_value)

    def coerce(self, value):
        if value is None:
            return None
        return np.asarray(value)
