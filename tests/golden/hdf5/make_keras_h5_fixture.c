/* Generates tests/golden/hdf5/keras_mlp_full_model.h5 and keras_mlp_weights_only.h5 with the HDF5 C library of the build
 * container (/opt/conda, libhdf5 1.10.x), laid out the way Keras 2.11 + h5py write the MLP baseline's checkpoints
 * (`ModelCheckpoint(save_weights_only=False)` -> model.save(path.h5), step2_retrain.py:253-261; `save_weights`):
 *
 *   /                      attrs: keras_version, backend, model_config  (variable-length UTF-8 strings, as h5py stores str)
 *   /model_weights         attrs: layer_names = fixed-length byte strings (numpy 'S' array), backend, keras_version
 *   /model_weights/<layer> attrs: weight_names = [b'<layer>/kernel:0', b'<layer>/bias:0']  (empty for layers without weights)
 *   /model_weights/<layer>/<layer>/kernel:0, bias:0     float32 datasets
 *   /optimizer_weights/... (present in full-model files; ignored by the importer)
 * The weights-only file has layer_names and the layer groups at the root.  Layer order = creation order of
 * step2_retrain.build_model (:95-126): input, (Dense, activation) per hidden layer, Dense(output_length), activation,
 * Dense(n_lin), Dense(n_relu), Concatenate - names are NOT sorted, so the importer must follow layer_names.
 * Model: input 8 -> Dense(128) -> LeakyReLU -> Dense(8) -> LeakyReLU -> [Dense(4) || Dense(4, relu)] (tiny on purpose).
 * Values: kernel[i][j] of weight tensor t = 0.001*(t+1)*(i*ncols + j) - 0.05*(t+1), bias[j] = 0.01*(t+1)*j - 0.02.
 *   gcc make_keras_h5_fixture.c -I/opt/conda/include -L/opt/conda/lib -Wl,-rpath,/opt/conda/lib -lhdf5 -o /tmp/mkk && /tmp/mkk <dir>
 */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void attr_vlen_str(hid_t loc, const char* name, const char* value) {
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, H5T_VARIABLE); H5Tset_cset(t, H5T_CSET_UTF8);
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(loc, name, t, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, t, &value);
    H5Aclose(a); H5Sclose(sp); H5Tclose(t);
}

static void attr_fixed_strs(hid_t loc, const char* name, const char** values, int n) {
    size_t width = 1;
    for (int i = 0; i < n; ++i) if (strlen(values[i]) > width) width = strlen(values[i]);
    char* buf = calloc((size_t)(n > 0 ? n : 1), width);
    for (int i = 0; i < n; ++i) memcpy(buf + (size_t)i * width, values[i], strlen(values[i]));   /* null-padded, numpy 'S' */
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, width); H5Tset_strpad(t, H5T_STR_NULLPAD);
    hsize_t d[1] = {(hsize_t)n};
    hid_t sp = n > 0 ? H5Screate_simple(1, d, NULL) : H5Screate(H5S_NULL);
    hid_t a = H5Acreate2(loc, name, t, sp, H5P_DEFAULT, H5P_DEFAULT);
    if (n > 0) H5Awrite(a, t, buf);
    H5Aclose(a); H5Sclose(sp); H5Tclose(t); free(buf);
}

static int tensor_index = 0;
static void dense(hid_t parent, const char* lname, int k, int n) {
    hid_t g = H5Gcreate2(parent, lname, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    char kn[64], bn[64];
    snprintf(kn, sizeof kn, "%s/kernel:0", lname); snprintf(bn, sizeof bn, "%s/bias:0", lname);
    const char* names[2] = {kn, bn};
    attr_fixed_strs(g, "weight_names", names, 2);
    hid_t g2 = H5Gcreate2(g, lname, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    float* w = malloc(sizeof(float) * (size_t)k * n);
    float* b = malloc(sizeof(float) * (size_t)n);
    const int t = tensor_index++;
    for (int i = 0; i < k * n; ++i) w[i] = (float)(0.001 * (t + 1) * i - 0.05 * (t + 1));
    for (int j = 0; j < n; ++j) b[j] = (float)(0.01 * (t + 1) * j - 0.02);
    hsize_t d2[2] = {(hsize_t)k, (hsize_t)n}, d1[1] = {(hsize_t)n};
    hid_t s2 = H5Screate_simple(2, d2, NULL), s1 = H5Screate_simple(1, d1, NULL);
    hid_t ds = H5Dcreate2(g2, "kernel:0", H5T_IEEE_F32LE, s2, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, w); H5Dclose(ds);
    ds = H5Dcreate2(g2, "bias:0", H5T_IEEE_F32LE, s1, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, b); H5Dclose(ds);
    H5Sclose(s2); H5Sclose(s1); H5Gclose(g2); H5Gclose(g); free(w); free(b);
}

static void plain(hid_t parent, const char* lname) {            /* a layer without weights */
    hid_t g = H5Gcreate2(parent, lname, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    attr_fixed_strs(g, "weight_names", NULL, 0);
    H5Gclose(g);
}

static void weights_group(hid_t loc) {
    const char* layers[9] = {"input_1", "dense", "leaky_re_lu", "dense_1", "leaky_re_lu_1", "dense_2", "dense_3", "concatenate", NULL};
    attr_fixed_strs(loc, "layer_names", layers, 8);
    attr_vlen_str(loc, "backend", "tensorflow");
    attr_vlen_str(loc, "keras_version", "2.11.0");
    tensor_index = 0;
    plain(loc, "input_1");
    dense(loc, "dense", 8, 128);
    plain(loc, "leaky_re_lu");
    dense(loc, "dense_1", 128, 8);
    plain(loc, "leaky_re_lu_1");
    dense(loc, "dense_2", 8, 4);
    dense(loc, "dense_3", 8, 4);
    plain(loc, "concatenate");
}

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : ".";
    char path[1024];
    snprintf(path, sizeof path, "%s/keras_mlp_full_model.h5", dir);
    hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    attr_vlen_str(f, "keras_version", "2.11.0");
    attr_vlen_str(f, "backend", "tensorflow");
    attr_vlen_str(f, "model_config", "{\"class_name\": \"Functional\", \"config\": {\"name\": \"model\"}}");
    hid_t g = H5Gcreate2(f, "model_weights", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    weights_group(g);
    H5Gclose(g);
    g = H5Gcreate2(f, "optimizer_weights", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    const char* on[1] = {"Adam/iter:0"};
    attr_fixed_strs(g, "weight_names", on, 1);
    hid_t g2 = H5Gcreate2(g, "Adam", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    long long it = 59130;
    hid_t sc = H5Screate(H5S_SCALAR);
    hid_t ds = H5Dcreate2(g2, "iter:0", H5T_STD_I64LE, sc, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_LLONG, H5S_ALL, H5S_ALL, H5P_DEFAULT, &it);
    H5Dclose(ds); H5Sclose(sc); H5Gclose(g2); H5Gclose(g);
    H5Fclose(f);
    snprintf(path, sizeof path, "%s/keras_mlp_weights_only.h5", dir);
    f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    weights_group(f);
    H5Fclose(f);
    return 0;
}
