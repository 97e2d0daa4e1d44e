"""Minimal stand-in for the `traitlets` package (not installed in this image).

TEST INFRASTRUCTURE ONLY -- used by oracle/ref_harness to import the unmodified
reference sources from /root/reference inside the build container so golden
vectors can be generated.  Never imported by the product package.

Semantics kept: class-level trait declarations with defaults, per-instance
deep-copied defaults, `@default(name)` dynamic defaults, `HasTraits(**kw)`
setting only declared traits (unknown keywords are dropped, as traitlets does
with a deprecation warning), `List` casting tuples to lists.
"""
import copy


class _Undefined:
    pass


class TraitType:
    _cast = None

    def __init__(self, default_value=_Undefined, *args, **kwargs):
        if default_value is _Undefined and 'default_value' in kwargs:
            default_value = kwargs['default_value']
        self.default_value = default_value
        self.name = None

    def __set_name__(self, owner, name):
        self.name = name

    def _static_default(self):
        if self.default_value is _Undefined:
            return None
        return copy.deepcopy(self.default_value)

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        d = obj.__dict__
        if self.name not in d:
            gen = type(obj)._default_generators().get(self.name)
            val = gen(obj) if gen is not None else self._static_default()
            d[self.name] = self.coerce(val)
        return d[self.name]

    def __set__(self, obj, value):
        obj.__dict__[self.name] = self.coerce(value)

    def coerce(self, value):
        return value


class Bool(TraitType):
    def _static_default(self):
        return False if self.default_value is _Undefined else self.default_value


class Int(TraitType):
    def _static_default(self):
        return 0 if self.default_value is _Undefined else self.default_value


class Float(TraitType):
    def _static_default(self):
        return 0.0 if self.default_value is _Undefined else self.default_value


class Unicode(TraitType):
    def _static_default(self):
        return '' if self.default_value is _Undefined else self.default_value


class Any(TraitType):
    pass


class Enum(TraitType):
    def __init__(self, values=None, default_value=_Undefined, **kwargs):
        super().__init__(default_value, **kwargs)
        self.values = values


class List(TraitType):
    def __init__(self, default_value=_Undefined, *args, **kwargs):
        if isinstance(default_value, TraitType):
            default_value = _Undefined
        super().__init__(default_value, **kwargs)

    def _static_default(self):
        return [] if self.default_value is _Undefined else copy.deepcopy(list(self.default_value))

    def coerce(self, value):
        if isinstance(value, tuple):
            return list(value)
        return value


class Dict(TraitType):
    def __init__(self, default_value=_Undefined, *args, **kwargs):
        if isinstance(default_value, TraitType):
            default_value = _Undefined
        kwargs.pop('key_trait', None)
        kwargs.pop('value_trait', None)
        super().__init__(default_value)

    def _static_default(self):
        return {} if self.default_value is _Undefined else copy.deepcopy(dict(self.default_value))


class Instance(TraitType):
    def __init__(self, klass=None, default_value=_Undefined, *args, **kwargs):
        if default_value is None:
            default_value = _Undefined
        super().__init__(default_value)


class _DefaultMarker:
    def __init__(self, name, func):
        self.trait_name = name
        self.func = func

    def __set_name__(self, owner, name):
        self.attr = name


def default(name):
    def deco(func):
        return _DefaultMarker(name, func)
    return deco


class HasTraits:
    @classmethod
    def _default_generators(cls):
        gens = {}
        for klass in reversed(cls.__mro__):
            for v in vars(klass).values():
                if isinstance(v, _DefaultMarker):
                    gens[v.trait_name] = v.func
        return gens

    @classmethod
    def _trait_names(cls):
        names = set()
        for klass in cls.__mro__:
            for k, v in vars(klass).items():
                if isinstance(v, TraitType):
                    names.add(k)
        return names

    def __init__(self, **kwargs):
        names = type(self)._trait_names()
        for k, v in kwargs.items():
            if k in names:
                setattr(self, k, v)

    def trait_names(self):
        return sorted(type(self)._trait_names())
