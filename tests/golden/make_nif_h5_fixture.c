/* make_nif_h5_fixture.c — writes tests/golden/nif_tiny/converted.hdf5: a tiny Keras-"Functional"-style
 * HDF5 model with the layout the reference's loader walks (src/keras/Hdf5Model.cpp:62-86):
 * root attributes keras_version / backend / model_config, Dense weights under
 * /model_weights/<layer>/<layer>/{kernel:0,bias:0}. Kernels of layers 0-2 are stored as IEEE binary16
 * (a 2-byte custom float type, as Keras mixed-precision models are), layer 3 as binary32.
 * Weight value at flat index i of layer l: (((i*7 + l*13) % 61) - 30) / 256  (exact in binary16);
 * bias value: (((i*5 + l*3) % 17) - 8) / 64. tests/test_nif_assets.py recomputes these.
 *
 * Build + run (from the repo root):
 *   gcc -I/opt/conda/include tests/golden/make_nif_h5_fixture.c -L/opt/conda/lib -lhdf5 \
 *       -Wl,-rpath,/opt/conda/lib -o /tmp/mkfix && /tmp/mkfix tests/golden/nif_tiny/converted.hdf5
 */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const char* kConfig =
    "{\"class_name\": \"Functional\", \"config\": {\"name\": \"nif_tiny\", \"layers\": ["
    "{\"class_name\": \"InputLayer\", \"config\": {\"batch_input_shape\": [null, 16], \"dtype\": \"float16\", \"name\": \"input_1\"}, \"name\": \"input_1\", \"inbound_nodes\": []},"
    "{\"class_name\": \"Dense\", \"config\": {\"name\": \"dense\", \"trainable\": true, \"dtype\": \"float16\", \"units\": 32, \"activation\": \"relu\", \"use_bias\": true}, \"name\": \"dense\"},"
    "{\"class_name\": \"Dense\", \"config\": {\"name\": \"dense_1\", \"trainable\": true, \"dtype\": \"float16\", \"units\": 32, \"activation\": \"relu\", \"use_bias\": true}, \"name\": \"dense_1\"},"
    "{\"class_name\": \"Concatenate\", \"config\": {\"name\": \"concatenate\", \"axis\": -1}, \"name\": \"concatenate\"},"
    "{\"class_name\": \"Dense\", \"config\": {\"name\": \"dense_2\", \"trainable\": true, \"dtype\": \"float16\", \"units\": 32, \"activation\": \"relu\", \"use_bias\": false}, \"name\": \"dense_2\"},"
    "{\"class_name\": \"Dense\", \"config\": {\"name\": \"dense_3\", \"trainable\": true, \"dtype\": \"float32\", \"units\": 3, \"activation\": \"linear\", \"use_bias\": true}, \"name\": \"dense_3\"}"
    "]}, \"keras_version\": \"2.6.0\", \"backend\": \"tensorflow\"}";

static void check(int ok, const char* what) { if (!ok) { fprintf(stderr, "fixture writer: %s failed\n", what); exit(1); } }

static void str_attr(hid_t loc, const char* name, const char* value, int variable) {
  hid_t t = H5Tcopy(H5T_C_S1);
  hid_t s = H5Screate(H5S_SCALAR);
  if (variable) {
    H5Tset_size(t, H5T_VARIABLE);
    hid_t a = H5Acreate2(loc, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
    check(a >= 0 && H5Awrite(a, t, &value) >= 0, name);
    H5Aclose(a);
  } else {
    H5Tset_size(t, strlen(value));
    hid_t a = H5Acreate2(loc, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
    check(a >= 0 && H5Awrite(a, t, value) >= 0, name);
    H5Aclose(a);
  }
  H5Sclose(s); H5Tclose(t);
}

static hid_t half_type(void) {
  hid_t t = H5Tcopy(H5T_IEEE_F32LE);
  check(H5Tset_fields(t, 15, 10, 5, 0, 10) >= 0, "set_fields");
  check(H5Tset_size(t, 2) >= 0, "set_size");
  check(H5Tset_ebias(t, 15) >= 0, "set_ebias");
  return t;
}

static void dataset(hid_t grp, const char* name, int rank, const hsize_t* dims, const float* values, int half) {
  hid_t s = H5Screate_simple(rank, dims, NULL);
  hid_t ft = half ? half_type() : H5Tcopy(H5T_IEEE_F32LE);
  hid_t d = H5Dcreate2(grp, name, ft, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  check(d >= 0 && H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, values) >= 0, name);
  H5Dclose(d); H5Tclose(ft); H5Sclose(s);
}

int main(int argc, char** argv) {
  check(argc == 2, "usage: mkfix <out.hdf5>");
  hid_t f = H5Fcreate(argv[1], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  check(f >= 0, "create file");
  str_attr(f, "keras_version", "2.6.0", 0);
  str_attr(f, "backend", "tensorflow", 0);
  str_attr(f, "model_config", kConfig, 1);
  hid_t mw = H5Gcreate2(f, "model_weights", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  const char* names[4] = {"dense", "dense_1", "dense_2", "dense_3"};
  const hsize_t rows[4] = {16, 32, 48, 32}, cols[4] = {32, 32, 32, 3};
  const int hasBias[4] = {1, 1, 0, 1}, half[4] = {1, 1, 1, 0};
  for (int l = 0; l < 4; ++l) {
    hid_t g1 = H5Gcreate2(mw, names[l], H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    hid_t g2 = H5Gcreate2(g1, names[l], H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    const size_t n = rows[l] * cols[l];
    float* k = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) k[i] = (float)((int)((i * 7 + l * 13) % 61) - 30) / 256.f;
    const hsize_t kd[2] = {rows[l], cols[l]};
    dataset(g2, "kernel:0", 2, kd, k, half[l]);
    free(k);
    if (hasBias[l]) {
      float b[32];
      for (size_t i = 0; i < cols[l]; ++i) b[i] = (float)((int)((i * 5 + l * 3) % 17) - 8) / 64.f;
      const hsize_t bd[1] = {cols[l]};
      dataset(g2, "bias:0", 1, bd, b, half[l]);
    }
    H5Gclose(g2); H5Gclose(g1);
  }
  H5Gclose(mw);
  H5Fclose(f);
  return 0;
}
