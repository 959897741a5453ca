"""Dimensional bookkeeping for model parameters (reference: qgs/params/parameter.py).

Only what the tendencies path needs is kept: a `Parameter` is a float that remembers its units and
whether the stored number is the dimensional or the nondimensional value; the conversion factor is
the product over the unit string of L^-p (metres), f0^+p (seconds) and deltap^-p (pascals)
(parameter.py:597-617).  Symbols, LaTeX and SymPy expressions of the reference are not carried.
Arithmetic between parameters returns plain floats (the reference computes `float(self) op other`
and re-wraps it; the numbers are identical).
"""
import warnings

import numpy as np


def _unit_factors(units):
    """'[m^2][s^-1]' -> [('m', 2), ('s', -1)]"""
    out = []
    for token in units.replace('][', ' ').strip('[]').split():
        name, _, power = token.partition('^')
        out.append((name, int(power) if power else 1))
    return out


class ScalingParameter(float):
    """A parameter that defines the scales themselves (never converted)."""

    def __new__(cls, value, units="", description="", dimensional=False, symbol=None):
        obj = float.__new__(cls, value)
        obj._units = units
        obj._description = description
        obj._dimensional = dimensional
        obj._symbol = symbol
        return obj

    units = property(lambda self: self._units)
    description = property(lambda self: self._description)
    dimensional = property(lambda self: self._dimensional)
    symbol = property(lambda self: self._symbol)

    def __reduce__(self):
        return (ScalingParameter, (float(self), self._units, self._description, self._dimensional, self._symbol))


class Parameter(float):
    """Float with units; `input_dimensional`/`return_dimensional` select which value is stored
    (semantics of parameter.py:473-556)."""

    def __new__(cls, value, input_dimensional=True, units="", scale_object=None, description="", symbol=None,
                return_dimensional=False):
        unscaled = False
        stored = value
        if return_dimensional:
            if not input_dimensional:
                if scale_object is None:
                    return_dimensional, unscaled = False, True
                else:
                    stored = value / cls._conversion_factor(units, scale_object)
        elif input_dimensional:
            if scale_object is None:
                return_dimensional, unscaled = True, True
            else:
                stored = value * cls._conversion_factor(units, scale_object)
        if unscaled:
            warnings.warn("Parameter configured to perform dimensional conversion but without specifying a "
                          "ScaleParams object: Conversion disabled!")
        obj = float.__new__(cls, stored)
        obj._input_dimensional = input_dimensional
        obj._return_dimensional = return_dimensional
        obj._units = units
        obj._scale_object = scale_object
        obj._description = description
        obj._symbol = symbol
        return obj

    @staticmethod
    def _conversion_factor(units, scale_object):
        factor = 1.
        for name, power in _unit_factors(units):
            if name == 'm':
                factor *= float(scale_object.L) ** (-power)
            elif name == 's':
                factor *= float(scale_object.f0) ** power
            elif name == 'Pa':
                factor *= float(scale_object.deltap) ** (-power)
        return factor

    units = property(lambda self: self._units)
    description = property(lambda self: self._description)
    symbol = property(lambda self: self._symbol)
    input_dimensional = property(lambda self: self._input_dimensional)
    return_dimensional = property(lambda self: self._return_dimensional)

    @property
    def _nondimensionalization(self):
        return 1. if self._scale_object is None else self._conversion_factor(self._units, self._scale_object)

    @property
    def dimensional_value(self):
        return float(self) if self._return_dimensional else float(self) / self._nondimensionalization

    @property
    def nondimensional_value(self):
        return float(self) * self._nondimensionalization if self._return_dimensional else float(self)

    def __reduce__(self):
        # the stored number is passed back through the constructor without conversion
        return (_rebuild_parameter, (float(self), self.__dict__.copy()))


def _rebuild_parameter(value, state):
    obj = float.__new__(Parameter, value)
    obj.__dict__.update(state)
    return obj


class ParametersArray(np.ndarray):
    """Array of parameters sharing units and conversion flags (e.g. the spectral decomposition of the
    insolation, the orography, the equilibrium temperature); reference: parameter.py `ParametersArray`.
    Stored as plain floats, converted element-wise exactly like `Parameter`."""

    def __new__(cls, values, input_dimensional=True, units="", scale_object=None, description=None, symbols=None,
                return_dimensional=False):
        vals = [float(Parameter(v, input_dimensional=input_dimensional, units=units, scale_object=scale_object,
                                return_dimensional=return_dimensional)) for v in values]
        obj = np.asarray(vals, dtype=float).view(cls)
        obj._input_dimensional = input_dimensional
        obj._return_dimensional = return_dimensional
        obj._units = units
        obj._scale_object = scale_object
        obj._description = description
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        for k in ('_input_dimensional', '_return_dimensional', '_units', '_scale_object', '_description'):
            setattr(self, k, getattr(obj, k, None))

    units = property(lambda self: self._units)

    def __setitem__(self, key, value):
        np.ndarray.__setitem__(self, key, float(value))

    def __reduce__(self):
        return (_rebuild_parray, (np.asarray(self).copy(), {k: getattr(self, k, None) for k in
                                                            ('_input_dimensional', '_return_dimensional', '_units',
                                                             '_scale_object', '_description')}))


def _rebuild_parray(values, state):
    obj = np.asarray(values, dtype=float).view(ParametersArray)
    for k, v in state.items():
        setattr(obj, k, v)
    return obj
