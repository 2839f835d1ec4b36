/* nif_h5.c — the only translation unit that touches libhdf5 (C API, HDF5 1.10). Built as the small
 * plugin libmi_nif_h5.so which libmi_scene_host.so dlopens on demand, so neither the renderer nor the
 * host library carries a link-time dependency on HDF5 (it lives under /opt/conda in this image).
 *
 * Reference behaviour being provided (src/keras/Hdf5Model.cpp:62-133, which uses the HDF5 C++ API):
 *   - string attributes of the root group: keras_version, backend, model_config (:88-93);
 *   - a dataset read as binary32 (:118-131). The reference keeps binary16 kernels as 2-byte storage;
 *     here every float dataset is converted to binary32 by the library (exact for binary16 values)
 *     and the element size on file is reported so the caller knows what the model was saved as.
 */
#include <hdf5.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MI_H5_MAX_DIMS 4

typedef struct mi_h5_file { hid_t file; } mi_h5_file;

static void set_err(char* err, size_t n, const char* what, const char* name) {
  if (err && n) snprintf(err, n, "%s '%s'", what, name ? name : "");
}

int mi_h5_open(const char* path, mi_h5_file** out, char* err, size_t errlen) {
  if (!path || !out) { set_err(err, errlen, "null argument", ""); return 1; }
  H5Eset_auto2(H5E_DEFAULT, NULL, NULL); /* errors are reported through return codes, not stderr */
  hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f < 0) { set_err(err, errlen, "cannot open HDF5 file", path); return 2; }
  mi_h5_file* h = (mi_h5_file*)calloc(1, sizeof *h);
  if (!h) { H5Fclose(f); set_err(err, errlen, "out of memory", ""); return 3; }
  h->file = f;
  *out = h;
  return 0;
}

void mi_h5_close(mi_h5_file* h) {
  if (!h) return;
  H5Fclose(h->file);
  free(h);
}

void mi_h5_free(void* p) { free(p); }

/* Root-group string attribute (fixed-length or variable-length), returned as a malloc'd C string. */
int mi_h5_read_string_attr(mi_h5_file* h, const char* name, char** out, char* err, size_t errlen) {
  if (!h || !name || !out) { set_err(err, errlen, "null argument", ""); return 1; }
  hid_t a = H5Aopen(h->file, name, H5P_DEFAULT);
  if (a < 0) { set_err(err, errlen, "missing attribute", name); return 2; }
  hid_t t = H5Aget_type(a);
  int rc = 0;
  char* result = NULL;
  if (t < 0 || H5Tget_class(t) != H5T_STRING) {
    set_err(err, errlen, "attribute is not a string:", name); rc = 3;
  } else if (H5Tis_variable_str(t) > 0) {
    char* v = NULL;
    hid_t mt = H5Tcopy(H5T_C_S1);
    H5Tset_size(mt, H5T_VARIABLE);
    H5Tset_cset(mt, H5Tget_cset(t));
    if (H5Aread(a, mt, &v) < 0 || !v) { set_err(err, errlen, "cannot read attribute", name); rc = 4; }
    else {
      result = (char*)malloc(strlen(v) + 1);
      if (result) strcpy(result, v); else { set_err(err, errlen, "out of memory", ""); rc = 5; }
      H5free_memory(v);
    }
    H5Tclose(mt);
  } else {
    size_t n = H5Tget_size(t);
    result = (char*)calloc(n + 1, 1);
    if (!result) { set_err(err, errlen, "out of memory", ""); rc = 5; }
    else if (H5Aread(a, t, result) < 0) { set_err(err, errlen, "cannot read attribute", name); rc = 4; free(result); result = NULL; }
  }
  if (t >= 0) H5Tclose(t);
  H5Aclose(a);
  if (rc == 0) *out = result;
  return rc;
}

/* Float dataset → malloc'd binary32 array (row-major), its shape, and the element size on file. */
int mi_h5_read_float_dataset(mi_h5_file* h, const char* path, float** data, uint64_t dims[MI_H5_MAX_DIMS],
                             int* ndims, int* file_elem_bytes, char* err, size_t errlen) {
  if (!h || !path || !data || !dims || !ndims || !file_elem_bytes) { set_err(err, errlen, "null argument", ""); return 1; }
  hid_t d = H5Dopen2(h->file, path, H5P_DEFAULT);
  if (d < 0) { set_err(err, errlen, "missing dataset", path); return 2; }
  hid_t s = H5Dget_space(d), t = H5Dget_type(d);
  int rc = 0;
  float* buf = NULL;
  const int nd = s >= 0 ? H5Sget_simple_extent_ndims(s) : -1;
  if (nd < 0 || nd > MI_H5_MAX_DIMS) { set_err(err, errlen, "unsupported rank for dataset", path); rc = 3; }
  else if (t < 0 || H5Tget_class(t) != H5T_FLOAT) { set_err(err, errlen, "dataset is not floating point:", path); rc = 4; }
  else {
    hsize_t hd[MI_H5_MAX_DIMS] = {0, 0, 0, 0};
    if (nd > 0) H5Sget_simple_extent_dims(s, hd, NULL);
    size_t count = 1;
    for (int i = 0; i < nd; ++i) { dims[i] = (uint64_t)hd[i]; count *= (size_t)hd[i]; }
    const size_t eb = H5Tget_size(t);
    if (eb != 2 && eb != 4) { set_err(err, errlen, "Only float32 and float16 weights are supported:", path); rc = 5; }  /* Hdf5Model.cpp:111-117 */
    else {
      buf = (float*)malloc((count ? count : 1) * sizeof(float));
      if (!buf) { set_err(err, errlen, "out of memory", ""); rc = 6; }
      else if (count && H5Dread(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf) < 0) {
        set_err(err, errlen, "cannot read dataset", path); rc = 7; free(buf); buf = NULL;
      } else { *ndims = nd; *file_elem_bytes = (int)eb; }
    }
  }
  if (t >= 0) H5Tclose(t);
  if (s >= 0) H5Sclose(s);
  H5Dclose(d);
  if (rc == 0) *data = buf;
  return rc;
}

const char* mi_h5_library_version(void) {
  static char v[48];
  unsigned a = 0, b = 0, c = 0;
  H5get_libversion(&a, &b, &c);
  snprintf(v, sizeof v, "HDF5 %u.%u.%u", a, b, c);
  return v;
}
