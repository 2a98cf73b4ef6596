"""Dependency-free readers for ClimSim's static assets.

The reference opens these with xarray/netCDF4 (``climsim_utils/data_utils.py:46-57`` takes the
opened datasets as ctor arguments; ``baseline_models/MLP/training/HPO/baseline_v1/step2_retrain/
step2_retrain.py:187-190`` opens the norm files).  Neither xarray nor netCDF4 is a dependency of
this package, so this module provides

* ``read_cdf5``      - a NetCDF "CDF-5" (64-bit data) classic-format reader, enough for
                       ``grid_info/ClimSim_low-res_grid-info.nc``;
* ``AssetVar`` / ``AssetSet`` - a tiny mapping that quacks like the slice of ``xarray.Dataset``
                       the data_utils API touches (``ds[var].values``, ``.mean(dim=)``, ``/``,
                       ``*``, ``len``);
* ``read_netcdf`` / ``load_nc_assets`` - either NetCDF flavour (NetCDF-4 = HDF5 through
                       ``climsim_amd.hdf5``), e.g. the reference's normalisation files;
* ``load_npz_assets`` - rebuild AssetSets from the ``.npz`` bundles under ``tests/golden``.

Real xarray Datasets can be passed to ``climsim_amd.data_utils.data_utils`` as well.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterable, Mapping

import numpy as np

_NC_TYPES = {1: ("b", 1), 2: ("c", 1), 3: (">i2", 2), 4: (">i4", 4), 5: (">f4", 4), 6: (">f8", 8),
             7: ("B", 1), 8: (">u2", 2), 9: (">u4", 4), 10: (">i8", 8), 11: (">u8", 8)}
_NC_DIMENSION, _NC_VARIABLE, _NC_ATTRIBUTE = 0x0A, 0x0B, 0x0C


class _Cursor:
    def __init__(self, buf: bytes, wide: bool):
        self.buf, self.pos, self.wide = buf, 4, wide

    def u32(self) -> int:
        v = struct.unpack_from(">I", self.buf, self.pos)[0]
        self.pos += 4
        return v

    def u64(self) -> int:
        v = struct.unpack_from(">Q", self.buf, self.pos)[0]
        self.pos += 8
        return v

    def nonneg(self) -> int:          # NON_NEG: INT64 in CDF-5, INT in CDF-1/2
        return self.u64() if self.wide else self.u32()

    def name(self) -> str:
        n = self.nonneg()
        s = self.buf[self.pos:self.pos + n].decode("utf-8")
        self.pos += (n + 3) & ~3
        return s

    def values(self, nc_type: int, n: int):
        dt, sz = _NC_TYPES[nc_type]
        raw = self.buf[self.pos:self.pos + n * sz]
        self.pos += (n * sz + 3) & ~3
        if nc_type == 2:
            return raw.decode("utf-8", "replace")
        return np.frombuffer(raw, dtype=dt, count=n)

    def att_list(self) -> Dict[str, object]:
        tag, n = self.u32(), self.nonneg()
        out = {}
        if tag == 0:
            return out
        assert tag == _NC_ATTRIBUTE, "corrupt attribute list"
        for _ in range(n):
            nm = self.name()
            t = self.u32()
            cnt = self.nonneg()
            out[nm] = self.values(t, cnt)
        return out


def read_cdf5(path: str) -> Dict[str, np.ndarray]:
    """Read every variable of a classic-model NetCDF file (CDF-1, CDF-2 or CDF-5) into native
    ndarrays.  Record variables are not supported (the grid file has none)."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:3] != b"CDF" or buf[3] not in (1, 2, 5):
        raise ValueError(f"{path}: not a classic NetCDF file (magic {buf[:4]!r})")
    version = buf[3]
    cur = _Cursor(buf, wide=(version == 5))
    numrecs = cur.nonneg()
    # dim_list
    tag, n = cur.u32(), cur.nonneg()
    dims = []
    if tag != 0:
        assert tag == _NC_DIMENSION, "corrupt dimension list"
        for _ in range(n):
            nm = cur.name()
            dims.append((nm, cur.nonneg()))
    cur.att_list()
    tag, n = cur.u32(), cur.nonneg()
    out: Dict[str, np.ndarray] = {}
    out_dims: Dict[str, tuple] = {}
    recs = []          # (name, type, shape-without-record-dim, begin, vsize)
    if tag != 0:
        assert tag == _NC_VARIABLE, "corrupt variable list"
        for _ in range(n):
            nm = cur.name()
            nd = cur.nonneg()
            dimids = [cur.nonneg() for _ in range(nd)]
            cur.att_list()
            t = cur.u32()
            vsize = cur.nonneg()
            begin = cur.u64() if version != 1 else cur.u32()
            shape = tuple(dims[d][1] for d in dimids)
            out_dims[nm] = tuple(dims[d][0] for d in dimids)
            dt, sz = _NC_TYPES[t]
            if shape and shape[0] == 0:          # record variable: one slab per record
                recs.append((nm, t, shape[1:], begin, vsize))
                continue
            cnt = int(np.prod(shape)) if shape else 1
            arr = np.frombuffer(buf, dtype=dt, count=cnt, offset=begin).reshape(shape)
            out[nm] = arr.astype(arr.dtype.newbyteorder("=")) if t != 2 else arr
    recsize = sum(r[4] for r in recs)
    for nm, t, shp, begin, _ in recs:
        dt, sz = _NC_TYPES[t]
        cnt = int(np.prod(shp)) if shp else 1
        slabs = [np.frombuffer(buf, dtype=dt, count=cnt, offset=begin + r * recsize).reshape(shp)
                 for r in range(numrecs)]
        arr = np.stack(slabs) if slabs else np.zeros((0,) + shp, dtype=dt)
        out[nm] = arr.astype(arr.dtype.newbyteorder("=")) if t != 2 else arr
    # coordinate-less dimensions (ncol, lev, ilev) become index ranges, like xarray does
    for nm, ln in dims:
        if nm not in out:
            out[nm] = np.arange(ln if ln else numrecs)
            out_dims[nm] = (nm,)
    out["__dims__"] = out_dims  # type: ignore[assignment]
    return out


class AssetVar:
    """ndarray with the handful of xarray.DataArray behaviours data_utils relies on."""
    __array_priority__ = 100

    def __init__(self, values, dims: Iterable[str] = ()):
        self.values = np.asarray(values)
        self.dims = tuple(dims)

    def mean(self, dim=None):
        if dim is None:
            return AssetVar(self.values.mean())
        ax = self.dims.index(dim)
        return AssetVar(self.values.mean(axis=ax), self.dims[:ax] + self.dims[ax + 1:])

    def __len__(self):
        return len(self.values)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def _bin(self, other, op):
        o = other.values if isinstance(other, AssetVar) else other
        return AssetVar(op(self.values, o), self.dims)

    def __truediv__(self, o): return self._bin(o, np.divide)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __add__(self, o): return self._bin(o, np.add)
    def __getitem__(self, k): return self.values[k]


class AssetSet(dict):
    """``dict[str, AssetVar]``; stands in for the xarray Datasets of the reference ctor."""

    @classmethod
    def from_arrays(cls, arrays: Mapping[str, np.ndarray], dims: Mapping[str, tuple] | None = None):
        s = cls()
        for k, v in arrays.items():
            if k.startswith("__"):
                continue
            s[k] = AssetVar(v, (dims or {}).get(k, ()))
        return s


def load_grid_info(path: str) -> AssetSet:
    """``grid_info`` argument of ``data_utils`` from the CDF-5 grid file or an ``.npz`` bundle."""
    if path.endswith(".npz"):
        z = np.load(path)
        raw = {k: z[k] for k in z.files}
        dims = {k: (("ncol",) if v.shape == (384,) or k == "ncol" else
                    ("ilev",) if v.shape == (61,) else ("lev",) if v.shape == (60,) else ())
                for k, v in raw.items()}
    else:
        raw = read_cdf5(path)
        dims = raw.pop("__dims__")
    return AssetSet.from_arrays(raw, dims)


def read_netcdf(path: str) -> Dict[str, np.ndarray]:
    """Every variable of a NetCDF file, classic (CDF-1/2/5) or NetCDF-4 (= HDF5), as native ndarrays."""
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic[:3] == b"CDF":
        raw = read_cdf5(path)
        raw.pop("__dims__", None)
        return raw
    from .hdf5 import read_hdf5
    return read_hdf5(path)


def load_nc_assets(path: str) -> AssetSet:
    """One of the reference's normalisation files (``preprocessing/normalizations/{inputs,outputs}/*.nc``, HDF5) or
    any NetCDF file of scalars / per-level profiles as the mapping the ``data_utils`` ctor takes
    (the reference passes ``xr.open_dataset(path)``, ``step2_retrain.py:187-190``)."""
    raw = read_netcdf(path)
    return AssetSet.from_arrays(raw, {k: (("lev",) if np.ndim(v) == 1 else ()) for k, v in raw.items()})


def load_npz_assets(path: str, prefix: str) -> AssetSet:
    """Rebuild one of the four norm datasets (prefix ``input_mean`` / ``input_max`` /
    ``input_min`` / ``output_scale``) from a bundle written by ``tests/golden/make_golden.py``."""
    z = np.load(path)
    arrays = {k[len(prefix) + 1:]: z[k] for k in z.files if k.startswith(prefix + "/")}
    return AssetSet.from_arrays(arrays, {k: (("lev",) if v.ndim == 1 else ()) for k, v in arrays.items()})
