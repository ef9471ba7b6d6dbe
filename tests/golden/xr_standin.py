"""Named-dimension stand-in for the part of xarray that /root/reference/bhnerf/kgeo.py:91-593 uses.

Test infrastructure for tests/golden/make_golden.py only (xarray is not installed in this image): with it the
reference's own ``wave_vector``, ``spacetime_metric``, ``raise_or_lower_indices``, ``azimuthal_velocity_vector``,
``doppler_factor``, ``fluid_frame_tetrad``, ``magnetic_field_fluid_frame``, ``parallel_transport`` and the ZAMO variants
run unmodified on a geodesic ``Dataset`` and their outputs become golden vectors (fixture g12).

What is restated here are xarray's published rules for the ORDER of named dimensions, which decide the axis order of
everything those functions hand to NumPy:
  * a binary operation / ufunc on DataArrays broadcasts by dimension NAME; the result's dimensions are the ordered
    union of the operands' dimensions by first appearance, left to right (xarray.core.variable._unified_dims);
  * a plain ndarray operand is aligned POSITIONALLY with the DataArray operand's own dimensions
    (Variable._binary_op -> _broadcast_compat_data);
  * ``concat(objs, dim=<new name>)`` broadcasts the objects to the ordered union of their dimensions and puts the new
    dimension FIRST (xarray.core.variable.Variable.concat with a new dimension);
  * ``sel(dim=i)`` on an index-less dimension is positional and drops the dimension; ``transpose(..., 'mu')`` moves
    'mu' last; ``sum(dim, skipna=False)`` is a plain ``ndarray.sum``; ``np.asarray(dataarray)`` is the data in the
    object's own dimension order.
"""
import numpy as np


def _unified_dims(arrays):
    dims, sizes = [], {}
    for a in arrays:
        for d, s in zip(a.dims, a.values.shape):
            if d not in sizes:
                dims.append(d)
                sizes[d] = s
            elif sizes[d] != s:
                if sizes[d] == 1:
                    sizes[d] = s
                elif s != 1:
                    raise ValueError('dimension %r has sizes %d and %d' % (d, sizes[d], s))
    return tuple(dims), sizes


def _expand(a, dims):
    """Data of DataArray ``a`` as an ndarray whose axes follow ``dims`` (size-1 axes for the dimensions it lacks)."""
    order = [a.dims.index(d) for d in dims if d in a.dims]
    v = np.transpose(a.values, order) if order else a.values
    shape = [v.shape[[d for d in dims if d in a.dims].index(d)] if d in a.dims else 1 for d in dims]
    return v.reshape(shape)


class DataArray:
    __array_priority__ = 1000

    def __init__(self, data=np.nan, coords=None, dims=None, name=None):
        self.values = np.asarray(data)
        if dims is None:
            dims = tuple('dim_%d' % i for i in range(self.values.ndim))
        self.dims = (dims,) if isinstance(dims, str) else tuple(dims)
        assert len(self.dims) == self.values.ndim, (self.dims, self.values.shape)
        self.name = name

    # ---- NumPy protocol ------------------------------------------------------------------------------------------
    def __array__(self, dtype=None, copy=None):
        return self.values if dtype is None else self.values.astype(dtype)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != '__call__' or kwargs.get('out') is not None:
            return NotImplemented
        named = [x for x in inputs if isinstance(x, DataArray)]
        if any(isinstance(x, np.ndarray) and x.ndim > 0 for x in inputs):
            # a plain ndarray operand: NumPy broadcasting against the DataArray's data, result keeps the DataArray's dims
            assert len(named) == 1, 'ndarray mixed with several DataArrays: not needed by kgeo.py'
            dims = named[0].dims
            out = ufunc(*[x.values if isinstance(x, DataArray) else x for x in inputs], **kwargs)
            assert out.ndim == len(dims), (out.shape, dims)
            return DataArray(out, dims=dims)
        dims, _ = _unified_dims(named)
        out = ufunc(*[_expand(x, dims) if isinstance(x, DataArray) else x for x in inputs], **kwargs)
        return DataArray(out, dims=dims)

    def _bin(self, other, f, reflexive=False):
        if isinstance(other, (Dataset,)):
            return NotImplemented
        return f(other, self) if reflexive else f(self, other)

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.true_divide)
    def __rtruediv__(self, o): return self._bin(o, np.true_divide, True)
    def __pow__(self, o): return self._bin(o, np.power)
    def __rpow__(self, o): return self._bin(o, np.power, True)
    def __neg__(self): return DataArray(-self.values, dims=self.dims)
    def __float__(self): return float(self.values)
    def __len__(self): return len(self.values)

    def __iter__(self):
        for i in range(self.values.shape[0]):
            yield DataArray(self.values[i], dims=self.dims[1:])

    def __getitem__(self, key):             # positional indexing with an int / slice along the first axis only
        if isinstance(key, (int, np.integer)):
            return DataArray(self.values[key], dims=self.dims[1:])
        raise NotImplementedError(key)

    # ---- the DataArray methods kgeo.py calls -------------------------------------------------------------------------
    shape = property(lambda self: self.values.shape)
    ndim = property(lambda self: self.values.ndim)
    size = property(lambda self: self.values.size)
    data = property(lambda self: self.values)

    def sel(self, **kw):
        (dim, idx), = kw.items()
        ax = self.dims.index(dim)
        return DataArray(np.take(self.values, idx, axis=ax), dims=self.dims[:ax] + self.dims[ax + 1:])

    isel = sel

    def sum(self, dim=None, axis=None, skipna=None):
        if dim is not None:
            axis = self.dims.index(dim)
        ax = axis % self.values.ndim
        return DataArray(self.values.sum(axis=ax), dims=self.dims[:ax] + self.dims[ax + 1:])

    def clip(self, min=None, max=None):
        return DataArray(np.clip(self.values, min, max), dims=self.dims)

    def fillna(self, value):
        return DataArray(np.where(np.isnan(self.values), value, self.values), dims=self.dims)

    def conj(self):
        return DataArray(np.conj(self.values), dims=self.dims)

    def transpose(self, *dims):
        if Ellipsis in dims:
            i = dims.index(Ellipsis)
            rest = [d for d in self.dims if d not in dims]
            dims = tuple(dims[:i]) + tuple(rest) + tuple(dims[i + 1:])
        return DataArray(np.transpose(self.values, [self.dims.index(d) for d in dims]), dims=dims)


class Dataset:
    """``xr.Dataset({name: DataArray})`` with attribute access (``geos.r``, ``g_munu.tt``)."""

    def __init__(self, data_vars=None):
        object.__setattr__(self, '_vars', dict(data_vars or {}))

    def __getattr__(self, name):
        try:
            return self._vars[name]
        except KeyError:
            raise AttributeError(name)

    def __getitem__(self, name):
        return self._vars[name]

    @property
    def dims(self):
        out = {}
        for v in self._vars.values():
            out.update(zip(v.dims, v.values.shape))
        return out


def concat(objs, dim, coords=None):
    objs = [o if isinstance(o, DataArray) else DataArray(o) for o in objs]
    dims, sizes = _unified_dims(objs)
    assert dim not in dims, 'only concatenation along a NEW dimension is needed by kgeo.py'
    shape = [sizes[d] for d in dims]
    stacked = np.stack([np.broadcast_to(_expand(o, dims), shape) for o in objs], axis=0)
    return DataArray(stacked, dims=(dim,) + dims)


def full_like(other, fill_value):
    return DataArray(np.full_like(other.values, fill_value, dtype=np.result_type(other.values.dtype, type(fill_value))), dims=other.dims)


def zeros_like(other):
    return DataArray(np.zeros_like(other.values), dims=other.dims)


def dataset_from_geodesics(geos, form='image'):
    """A reference-shaped geodesic Dataset from ``bhnerf_amd.geodesics.Geodesics``.

    form='rays'   the ray-LIST form (`raytrace_ana(...).get_dataset()`): dims (pix, geo); alpha, beta, lam, eta one value per
                  ray on 'pix'.  Every broadcast in kgeo.py is then between arrays over the same two dimensions: no
                  dimension-order question arises.
    form='image'  the image-plane form (`get_dataset(num_alpha, num_beta, E, M)`, kgeo.py:61-62): 3-D fields on
                  (alpha, beta, geo), per-ray quantities INCLUDING the image coordinates alpha / beta on (alpha, beta).
                  The external kgeo package that builds this dataset is absent, so its exact coordinate layout is not
                  known; what is known is the reference's own published output: with alpha / beta as 1-D index coordinates
                  xarray's broadcasting rules would make `(geos.beta + 1j*mu) * kappa.conj()` (kgeo.py:508-509) come out on
                  (beta, alpha, geo) and rotate every pixel's polarisation by the angle of its mirror pixel -- a light curve
                  with EVPA 15 deg / polarised fraction 0.46 where the reference's notebook prints 36.9 deg / 0.634
                  (tests/test_geodesics_cpu.py::test_polarised_lightcurve_agrees...); per-ray alpha / beta reproduce the
                  published numbers, so that is the layout modelled here."""
    out = {}
    shape3 = np.asarray(geos['r']).shape
    for k, v in geos.items():
        a = np.asarray(v, dtype=np.float64)
        if form == 'rays':
            dims = ('pix', 'geo')
            if a.ndim == 3:
                out[k] = DataArray(a.reshape(-1, a.shape[-1]), dims=dims)
            elif a.ndim == 2:
                out[k] = DataArray(a.reshape(-1), dims=dims[:1])
            elif a.ndim == 0:
                out[k] = DataArray(a, dims=())
        else:
            dims = ('alpha', 'beta', 'geo')
            if a.ndim == 3 and a.shape == shape3:
                out[k] = DataArray(a, dims=dims)
            elif a.ndim == 2 and a.shape == shape3[:2]:
                out[k] = DataArray(a, dims=dims[:2])
            elif a.ndim == 0:
                out[k] = DataArray(a, dims=())
    return Dataset(out)
