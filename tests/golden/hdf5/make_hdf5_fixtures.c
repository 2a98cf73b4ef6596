/* Generates the HDF5 fixtures under tests/golden/hdf5/ with the HDF5 C library of the build container
 * (/opt/conda, libhdf5 1.10.x): the files exercise the storage variants climsim_amd/hdf5.py reads.
 * Expected values are a formula of the flat index (see tests/test_hdf5_cpu.py), so the files are the whole fixture.
 * E3SM-MMF.ml{i,o}.0001-02-01-{00000,01200}.nc are laid out the way the netCDF-4 library writes ClimSim's timestep
 * files: dimension-scale datasets `lev` / `ncol` (never written, CLASS/NAME attributes), every variable attached to
 * them (DIMENSION_LIST = variable-length object references, REFERENCE_LIST = compound), chunked + shuffle + deflate,
 * creation-order tracking, text and numeric attributes; values are the formulas in tests/test_hdf5_cpu.py.
 *   gcc make_hdf5_fixtures.c -I/opt/conda/include -L/opt/conda/lib -Wl,-rpath,/opt/conda/lib -lhdf5_hl -lhdf5 -o /tmp/mkh5 && /tmp/mkh5 <dir>
 */
#include <hdf5.h>
#include <hdf5_hl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LEV 60
#define NCOL 48

static double f64v(long i) { return 0.25 * (double)i - 100.0 + 1e-9 * (double)(i % 7); }
static float f32v(long i) { return (float)(0.5 * (double)i - 3.0); }

static void write_set(hid_t loc, int latest, int many) {
    static double a[LEV * NCOL];
    static float b[LEV * NCOL];
    static long long c[LEV];
    static short s[7];
    for (long i = 0; i < LEV * NCOL; ++i) { a[i] = f64v(i); b[i] = f32v(i); }
    for (long i = 0; i < LEV; ++i) c[i] = 1000000007LL * i - 5;
    for (int i = 0; i < 7; ++i) s[i] = (short)(i * 1000 - 3000);
    hsize_t d2[2] = {LEV, NCOL}, d1[1] = {LEV}, d7[1] = {7};
    hid_t sp2 = H5Screate_simple(2, d2, NULL), sp1 = H5Screate_simple(1, d1, NULL), sp7 = H5Screate_simple(1, d7, NULL);
    hid_t sc = H5Screate(H5S_SCALAR);
    hid_t ds;
    /* contiguous */
    ds = H5Dcreate2(loc, "state_t", H5T_IEEE_F64LE, sp2, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, a); H5Dclose(ds);
    /* chunked + shuffle + deflate, ragged edge chunks (16 x 20 over 60 x 48) */
    hid_t pl = H5Pcreate(H5P_DATASET_CREATE);
    hsize_t ch[2] = {16, 20};
    H5Pset_chunk(pl, 2, ch); H5Pset_shuffle(pl); H5Pset_deflate(pl, 4);
    ds = H5Dcreate2(loc, "state_q0001", H5T_IEEE_F64LE, sp2, H5P_DEFAULT, pl, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, a); H5Dclose(ds); H5Pclose(pl);
    /* chunked + fletcher32, float32, no compression */
    pl = H5Pcreate(H5P_DATASET_CREATE);
    hsize_t ch2[2] = {60, 8};
    H5Pset_chunk(pl, 2, ch2); H5Pset_fletcher32(pl);
    ds = H5Dcreate2(loc, "state_u", H5T_IEEE_F32LE, sp2, H5P_DEFAULT, pl, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, b); H5Dclose(ds); H5Pclose(pl);
    /* one chunk covering the dataset (layout v4: single-chunk index), deflate only */
    pl = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_chunk(pl, 2, d2); H5Pset_deflate(pl, 1);
    ds = H5Dcreate2(loc, "state_v", H5T_IEEE_F32LE, sp2, H5P_DEFAULT, pl, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, b); H5Dclose(ds); H5Pclose(pl);
    /* chunked, no filter, early allocation (layout v4: implicit index) */
    pl = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_chunk(pl, 2, ch); H5Pset_alloc_time(pl, H5D_ALLOC_TIME_EARLY);
    ds = H5Dcreate2(loc, "pbuf_ozone", H5T_IEEE_F64LE, sp2, H5P_DEFAULT, pl, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, a); H5Dclose(ds); H5Pclose(pl);
    /* big-endian float64 and int64 */
    ds = H5Dcreate2(loc, "be_f64", H5T_IEEE_F64BE, sp1, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, a); H5Dclose(ds);
    ds = H5Dcreate2(loc, "lev", H5T_STD_I64LE, sp1, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_LLONG, H5S_ALL, H5S_ALL, H5P_DEFAULT, c); H5Dclose(ds);
    /* compact int16 */
    pl = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_layout(pl, H5D_COMPACT);
    ds = H5Dcreate2(loc, "tiny", H5T_STD_I16LE, sp7, H5P_DEFAULT, pl, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_SHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, s); H5Dclose(ds); H5Pclose(pl);
    /* scalar, and a dataset that was never written (fill value) */
    double one = 1004.64;
    ds = H5Dcreate2(loc, "state_ps", H5T_IEEE_F64LE, sc, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, &one); H5Dclose(ds);
    ds = H5Dcreate2(loc, "never_written", H5T_IEEE_F32LE, sp1, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dclose(ds);
    /* nested group */
    hid_t g = H5Gcreate2(loc, "extra", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    ds = H5Dcreate2(g, "cam_in_LWUP", H5T_IEEE_F32LE, sp1, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, b); H5Dclose(ds);
    H5Gclose(g);
    /* many scalars: pushes a new-style group into dense (fractal heap) storage with several direct blocks */
    if (many) {
        for (int i = 0; i < 300; ++i) {
            char nm[64];
            snprintf(nm, sizeof nm, "scalar_with_a_long_variable_name_%03d", i);
            double v = 0.5 * i;
            ds = H5Dcreate2(loc, nm, H5T_IEEE_F64LE, sc, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, &v); H5Dclose(ds);
        }
    }
    (void)latest;
    H5Sclose(sp2); H5Sclose(sp1); H5Sclose(sp7); H5Sclose(sc);
}

#define TS_NCOL 384
/* value of variable number `v` at (lev l, column c) of timestep t; kind 0 = mli, 1 = mlo */
static double ts_val(int kind, int v, int t, int l, int c) {
    return (kind ? 1.5 : 1.0) * (v + 1) + 0.03125 * l + 0.0009765625 * c + 0.25 * t + (kind ? 0.001 * ((l + c) % 5) : 0.0);
}

static void text_attr(hid_t obj, const char* name, const char* value) {
    hid_t ty = H5Tcopy(H5T_C_S1); H5Tset_size(ty, strlen(value) + 1);
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t at = H5Acreate2(obj, name, ty, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(at, ty, value); H5Aclose(at); H5Sclose(sp); H5Tclose(ty);
}

static void write_timestep(const char* path, int kind, int t) {
    static const char* mli3[] = {"state_t", "state_q0001"};
    static const char* mli2[] = {"state_ps", "pbuf_SOLIN", "pbuf_LHFLX", "pbuf_SHFLX"};
    static const char* mlo2[] = {"cam_out_NETSW", "cam_out_FLWDS", "cam_out_PRECSC", "cam_out_PRECC", "cam_out_SOLS",
                                 "cam_out_SOLL", "cam_out_SOLSD", "cam_out_SOLLD"};
    hid_t fa = H5Pcreate(H5P_FILE_ACCESS);
    H5Pset_libver_bounds(fa, H5F_LIBVER_V18, H5F_LIBVER_LATEST);
    hid_t fc = H5Pcreate(H5P_FILE_CREATE);
    H5Pset_link_creation_order(fc, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
    H5Pset_attr_creation_order(fc, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
    hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, fc, fa);
    text_attr(f, "_NCProperties", "version=2,netcdf=4.8.1,hdf5=1.10.6");
    hsize_t dl[1] = {LEV}, dc[1] = {TS_NCOL}, d2[2] = {LEV, TS_NCOL};
    hid_t spl = H5Screate_simple(1, dl, NULL), spc = H5Screate_simple(1, dc, NULL), sp2 = H5Screate_simple(2, d2, NULL);
    hid_t pc = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_attr_creation_order(pc, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
    hid_t dlev = H5Dcreate2(f, "lev", H5T_IEEE_F32BE, spl, H5P_DEFAULT, pc, H5P_DEFAULT);
    H5DSset_scale(dlev, "This is a netCDF dimension but not a netCDF variable.        60");
    hid_t dcol = H5Dcreate2(f, "ncol", H5T_IEEE_F32BE, spc, H5P_DEFAULT, pc, H5P_DEFAULT);
    H5DSset_scale(dcol, "This is a netCDF dimension but not a netCDF variable.       384");
    static double a[LEV * TS_NCOL], b[TS_NCOL];
    int v = 0;
    for (int i = 0; i < 2; ++i, ++v) {
        for (int l = 0; l < LEV; ++l) for (int c = 0; c < TS_NCOL; ++c) a[l * TS_NCOL + c] = ts_val(kind, v, t, l, c);
        hid_t pl = H5Pcreate(H5P_DATASET_CREATE);
        hsize_t ch[2] = {30, 192};
        H5Pset_chunk(pl, 2, ch); H5Pset_shuffle(pl); H5Pset_deflate(pl, 6);
        H5Pset_attr_creation_order(pl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
        hid_t ds = H5Dcreate2(f, mli3[i], H5T_IEEE_F64LE, sp2, H5P_DEFAULT, pl, H5P_DEFAULT);
        H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, a);
        H5DSattach_scale(ds, dlev, 0); H5DSattach_scale(ds, dcol, 1);
        text_attr(ds, "units", i ? "kg/kg" : "K"); text_attr(ds, "long_name", i ? "Specific humidity" : "Temperature");
        H5Dclose(ds); H5Pclose(pl);
    }
    const char** names = kind ? mlo2 : mli2;
    const int n2 = kind ? 8 : 4;
    for (int i = 0; i < n2; ++i, ++v) {
        for (int c = 0; c < TS_NCOL; ++c) b[c] = ts_val(kind, v, t, 0, c);
        hid_t pl = H5Pcreate(H5P_DATASET_CREATE);
        H5Pset_attr_creation_order(pl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
        hid_t ds = H5Dcreate2(f, names[i], H5T_IEEE_F64LE, spc, H5P_DEFAULT, pl, H5P_DEFAULT);
        H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, b);
        H5DSattach_scale(ds, dcol, 0);
        text_attr(ds, "units", "W/m2");
        double fv = 9.96920996838687e+36;
        hid_t sc = H5Screate(H5S_SCALAR);
        hid_t at = H5Acreate2(ds, "_FillValue", H5T_IEEE_F64LE, sc, H5P_DEFAULT, H5P_DEFAULT);
        H5Awrite(at, H5T_NATIVE_DOUBLE, &fv); H5Aclose(at); H5Sclose(sc);
        H5Dclose(ds); H5Pclose(pl);
    }
    H5Dclose(dlev); H5Dclose(dcol); H5Pclose(pc);
    H5Sclose(spl); H5Sclose(spc); H5Sclose(sp2);
    H5Fclose(f); H5Pclose(fc); H5Pclose(fa);
}

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : ".";
    char path[1024];
    /* 1: library defaults = superblock 0, old-style groups (symbol table), object header v1, layout v3 */
    snprintf(path, sizeof path, "%s/earliest.h5", dir);
    hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    write_set(f, 0, 0); H5Fclose(f);
    /* 2: libver latest = superblock 3, object header v2, compact/dense links, layout v4 chunk indexes */
    hid_t fa = H5Pcreate(H5P_FILE_ACCESS);
    H5Pset_libver_bounds(fa, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST);
    snprintf(path, sizeof path, "%s/latest.h5", dir);
    f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, fa);
    write_set(f, 1, 0); H5Fclose(f);
    /* 3: what netCDF-4 writes: libver (V18, latest) + creation-order tracking, > 8 links -> dense storage */
    H5Pset_libver_bounds(fa, H5F_LIBVER_V18, H5F_LIBVER_LATEST);
    hid_t fc = H5Pcreate(H5P_FILE_CREATE);
    H5Pset_link_creation_order(fc, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
    snprintf(path, sizeof path, "%s/netcdf4_like.h5", dir);
    f = H5Fcreate(path, H5F_ACC_TRUNC, fc, fa);
    write_set(f, 0, 1); H5Fclose(f);
    H5Pclose(fc); H5Pclose(fa);
    for (int t = 0; t < 2; ++t) {
        snprintf(path, sizeof path, "%s/E3SM-MMF.mli.0001-02-01-%05d.nc", dir, t * 1200);
        write_timestep(path, 0, t);
        snprintf(path, sizeof path, "%s/E3SM-MMF.mlo.0001-02-01-%05d.nc", dir, t * 1200);
        write_timestep(path, 1, t);
    }
    return 0;
}
