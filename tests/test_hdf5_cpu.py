"""Native HDF5 / NetCDF-4 reader (`climsim_amd/hdf5.py`) against files written by the HDF5 C library.

The reference opens its raw timestep files, normalisation files and `.h5` splits through xarray / netCDF4 / h5py
(climsim_utils/data_utils.py:619-640, 906-925, 1029-1035).  The fixtures under tests/golden/hdf5/ were produced by
libhdf5 1.10 (tests/golden/hdf5/make_hdf5_fixtures.c) and exercise every storage variant the reader handles; expected
values are formulas of the flat index.  Files the writer produces are checked with the library's own `h5dump` when it
is installed (the build container has it under /opt/conda/bin)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from climsim_amd.assets import AssetSet, load_nc_assets, read_netcdf
from climsim_amd.hdf5 import Hdf5File, Hdf5Unsupported, read_hdf5, write_hdf5_dataset

HERE = os.path.join(os.path.dirname(__file__), "golden", "hdf5")
LEV, NCOL = 60, 48
_i = np.arange(LEV * NCOL)
F64 = (0.25 * _i - 100.0 + 1e-9 * (_i % 7)).reshape(LEV, NCOL)
F32 = (0.5 * _i - 3.0).astype(np.float32).reshape(LEV, NCOL)


@pytest.mark.parametrize("name", ["earliest", "latest", "netcdf4_like"])
def test_fixture_every_storage_variant(name):
    d = read_hdf5(os.path.join(HERE, name + ".h5"))
    assert np.array_equal(d["state_t"], F64)                  # contiguous
    assert np.array_equal(d["state_q0001"], F64)              # chunked 16x20 (ragged edges) + shuffle + deflate
    assert np.array_equal(d["state_u"], F32)                  # chunked + fletcher32
    assert np.array_equal(d["state_v"], F32)                  # one chunk + deflate (layout v4 single-chunk index)
    assert np.array_equal(d["pbuf_ozone"], F64)               # chunked, early allocation (layout v4 implicit index)
    assert np.array_equal(d["be_f64"], F64.ravel()[:LEV]) and d["be_f64"].dtype.isnative      # big-endian source
    assert np.array_equal(d["lev"], 1000000007 * np.arange(LEV, dtype=np.int64) - 5)
    assert d["tiny"].dtype == np.int16 and np.array_equal(d["tiny"], np.arange(7) * 1000 - 3000)   # compact
    assert d["state_ps"].shape == () and float(d["state_ps"]) == 1004.64                      # scalar dataspace
    assert d["never_written"].shape == (LEV,) and not d["never_written"].any()                # no storage: fill value
    assert np.array_equal(d["extra/cam_in_LWUP"], F32.ravel()[:LEV])                          # nested group
    if name == "netcdf4_like":                                # dense link storage (fractal heap, several direct blocks)
        assert len(d) == 311
        for j in range(300):
            assert float(d[f"scalar_with_a_long_variable_name_{j:03d}"]) == 0.5 * j


def test_lazy_file_interface():
    with Hdf5File(os.path.join(HERE, "latest.h5")) as f:
        assert "state_t" in f and "nope" not in f
        assert f.shape("state_q0001") == (LEV, NCOL)
        assert sorted(f.keys()) == sorted(read_hdf5(os.path.join(HERE, "latest.h5")).keys())
        with pytest.raises(KeyError):
            f["nope"]


def test_not_hdf5_is_an_error(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"definitely not an HDF5 file" * 10)
    with pytest.raises(ValueError):
        Hdf5File(str(p))
    (tmp_path / "empty.h5").write_bytes(b"")
    with pytest.raises(ValueError):
        Hdf5File(str(tmp_path / "empty.h5"))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(0, 124), (1,), (384, 128), (7, 5, 3)])
def test_writer_round_trip(tmp_path, dtype, shape):
    rng = np.random.default_rng(3)
    a = rng.standard_normal(shape).astype(dtype)
    p = str(tmp_path / "w.h5")
    write_hdf5_dataset(p, "data", a)
    with Hdf5File(p) as f:
        assert f.keys() == ["data"]
        b = f["data"]
    assert b.dtype == a.dtype and b.shape == a.shape and np.array_equal(a, b)
    with pytest.raises(Hdf5Unsupported):
        write_hdf5_dataset(p, "data", np.arange(4))


@pytest.mark.skipif(shutil.which("h5dump") is None and not os.path.exists("/opt/conda/bin/h5dump"),
                    reason="needs the HDF5 command line tools")
def test_written_file_is_read_by_libhdf5(tmp_path):
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    a = (np.arange(12, dtype=np.float32) * 0.5 - 1).reshape(3, 4)
    p = str(tmp_path / "pred.h5")
    write_hdf5_dataset(p, "pred", a)
    out = subprocess.run([h5dump, "-d", "/pred", p], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "H5T_IEEE_F32LE" in out.stdout and "( 3, 4 )" in out.stdout
    nums = [float(t) for line in out.stdout.splitlines() if line.strip().startswith("(") and "):" in line
            for t in line.split("):")[1].replace(",", " ").split()]
    assert nums == a.ravel().tolist()


def test_data_utils_h5_paths(tmp_path):
    """`save_h5=True` splits (dataset 'data') and `load_h5_file` (dataset 'pred'), data_utils.py:906-925, 1029-1035."""
    from climsim_amd.data_utils import data_utils
    a = np.random.default_rng(0).standard_normal((768, 128)).astype(np.float32)
    write_hdf5_dataset(str(tmp_path / "p.h5"), "pred", a)
    assert np.array_equal(data_utils.load_h5_file(str(tmp_path / "p.h5")), a)
    du = data_utils.__new__(data_utils)
    du.save_npy, du.save_h5 = True, True
    du._save_array(a, str(tmp_path / "val_target"))
    assert np.array_equal(np.load(tmp_path / "val_target.npy"), a)
    assert np.array_equal(read_hdf5(str(tmp_path / "val_target.h5"))["data"], a)


def test_netcdf_dispatch_and_assets(tmp_path):
    """`read_netcdf` takes either flavour; `load_nc_assets` gives the mapping the data_utils ctor wants."""
    raw = read_netcdf(os.path.join(HERE, "netcdf4_like.h5"))
    assert np.array_equal(raw["state_t"], F64)
    ds = load_nc_assets(os.path.join(HERE, "earliest.h5"))
    assert isinstance(ds, AssetSet)
    assert float(ds["state_ps"].values) == 1004.64
    assert np.array_equal(ds["lev"].values, 1000000007 * np.arange(LEV) - 5)


# ---- the raw-file loader path on NetCDF-4 timestep files (the format ClimSim's real files have) -------------------
def ts_val(kind, v, t, l, c):
    """Formula of tests/golden/hdf5/make_hdf5_fixtures.c: variable number v of an mli (kind 0) / mlo (1) file."""
    return (1.5 if kind else 1.0) * (v + 1) + 0.03125 * l + 0.0009765625 * c + 0.25 * t + (0.001 * ((l + c) % 5) if kind else 0.0)


def nc4_tree(tmp_path):
    """Copies the two committed timestep pairs into a data_path layout; a third mli file only exists to be dropped
    by `set_filelist` (end_idx=-1, data_utils.py:742-753) and is never opened."""
    d = tmp_path / "train" / "0001-02"
    d.mkdir(parents=True)
    for f in os.listdir(HERE):
        if f.endswith(".nc"):
            shutil.copy(os.path.join(HERE, f), d / f)
    (d / "E3SM-MMF.mli.0001-02-01-02400.nc").write_bytes(b"")
    return tmp_path


def nc4_expected(du, t):
    lev, col = np.arange(60)[:, None], np.arange(384)[None, :]
    sub, div, scale = du.save_norm()
    x_raw = np.concatenate([ts_val(0, 0, t, lev, col).T, ts_val(0, 1, t, lev, col).T]
                           + [ts_val(0, 2 + i, t, 0, col[0])[:, None] for i in range(4)], axis=1)
    y_raw = np.concatenate([((ts_val(1, 0, t, lev, col) - ts_val(0, 0, t, lev, col)) / 1200).T,
                            ((ts_val(1, 1, t, lev, col) - ts_val(0, 1, t, lev, col)) / 1200).T]
                           + [ts_val(1, 2 + i, t, 0, col[0])[:, None] for i in range(8)], axis=1)
    return (x_raw - sub) / div, y_raw * scale


def test_loader_reads_netcdf4_timestep_files(lowres_assets, tmp_path):
    from test_loader_cpu import make
    du = make(lowres_assets, "pytorch", nc4_tree(tmp_path))
    assert len(du.get_filelist("train")) == 2
    pairs = list(du.load_ncdata_with_generator("train").as_numpy_iterator())
    assert len(pairs) == 2 and pairs[0][0].shape == (384, 124) and pairs[0][1].shape == (384, 128)
    for t in range(2):
        x, y = nc4_expected(du, t)
        np.testing.assert_allclose(pairs[t][0], x, rtol=1e-13)
        np.testing.assert_allclose(pairs[t][1], y, rtol=1e-13, atol=1e-300)
    du.save_as_npy("train", save_path=str(tmp_path / "npy"))
    xi = np.load(tmp_path / "npy" / "train_input.npy")
    assert xi.dtype == np.float32 and xi.shape == (768, 124)
    np.testing.assert_array_equal(xi[:384], np.float32(pairs[0][0]))
