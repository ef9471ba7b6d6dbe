"""Time units for the hot path.  Uses astropy.units when it is installed (as the reference does,
optimization.py:7); otherwise a minimal stand-in providing what the hot path touches: ``hr``,
``Quantity.value/.unit/.to()``, indexing and ``len``."""
import numpy as np

try:  # pragma: no cover - astropy is optional
    from astropy.units import Quantity, Unit, hr, min, s, day  # noqa: F401,A004
    HAVE_ASTROPY = True
except Exception:  # astropy absent (this image): minimal stand-in
    HAVE_ASTROPY = False

    class Unit:
        __array_ufunc__ = None          # let ndarray * unit defer to Unit.__rmul__

        def __init__(self, name, in_hr):
            self.name, self.in_hr = name, float(in_hr)

        def __eq__(self, other):
            return isinstance(other, Unit) and other.name == self.name

        def __ne__(self, other):
            return not self.__eq__(other)

        def __hash__(self):
            return hash(self.name)

        def __rmul__(self, value):
            return Quantity(value, self)

        __mul__ = __rmul__

        def __repr__(self):
            return 'Unit("%s")' % self.name

        __str__ = lambda self: self.name

    class Quantity:
        __array_ufunc__ = None

        def __init__(self, value, unit):
            self.value = np.asarray(value, dtype=np.float64) if not np.isscalar(value) else float(value)
            self.unit = unit

        def to(self, unit):
            unit = _lookup(unit)
            return Quantity(np.asarray(self.value) * (self.unit.in_hr / unit.in_hr), unit)

        def __getitem__(self, key):
            return Quantity(np.asarray(self.value)[key], self.unit)

        def __len__(self):
            return len(self.value)

        def __sub__(self, other):
            return Quantity(np.asarray(self.value) - np.asarray(other.to(self.unit).value), self.unit)

        def __add__(self, other):
            return Quantity(np.asarray(self.value) + np.asarray(other.to(self.unit).value), self.unit)

        def __mul__(self, factor):
            return Quantity(np.asarray(self.value) * factor, self.unit)

        __rmul__ = __mul__

        def _cmp(self, other, op):
            return op(np.asarray(self.value), np.asarray(other.to(self.unit).value))

        def __le__(self, other):
            return self._cmp(other, np.less_equal)

        def __lt__(self, other):
            return self._cmp(other, np.less)

        def __ge__(self, other):
            return self._cmp(other, np.greater_equal)

        def __gt__(self, other):
            return self._cmp(other, np.greater)

        def __repr__(self):
            return '<Quantity %s %s>' % (self.value, self.unit)

    hr = Unit('hr', 1.0)
    min = Unit('min', 1.0 / 60.0)  # noqa: A001
    s = Unit('s', 1.0 / 3600.0)
    day = Unit('day', 24.0)
    _UNITS = {'hr': hr, 'h': hr, 'hour': hr, 'min': min, 's': s, 'day': day, 'd': day}

    def _lookup(u):
        return _UNITS[u] if isinstance(u, str) else u


def is_quantity(x):
    return isinstance(x, Quantity)


def unit_name(u):
    """'hr' for units.hr, whichever backend is in use."""
    if u is None:
        return None
    return getattr(u, 'name', None) or str(u)


def strip(x, unit=None):
    """value of a Quantity (converted to `unit` if given) or x itself."""
    if is_quantity(x):
        return np.asarray(x.to(unit).value if unit is not None else x.value)
    return x
