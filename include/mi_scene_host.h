/*
 * mi_scene_host.h — C ABI of the host-side scene plumbing (CPU only, no GPU needed).
 *
 * These entry points reproduce the INPUTS the reference hands to its renderers — they are
 * the callers' side of the hot path, not the hot path itself:
 *   buildSceneDescription / makeCornellBoxScene / makePrimitiveScene  (src/app_utils.cpp:252-283,
 *                                                                      src/scene_utils.cpp:319-597)
 *   buildSceneData (array packing + BVH build driver)                 (src/app_utils.cpp:291-371)
 *   BvhBuilder::build + buildCompactBvh (node FORMAT is contractual,  (include/embree_utils/bvh.hpp:37-77,
 *     the Embree builder is not: tree topology is "parity unpinned")   src/CompactBvhBuild.cpp:5-56)
 *   initPerspectiveRayStream (no-jitter form) + zeroRgb               (src/app_utils.cpp:19-53)
 * The arrays they produce are what mi_scene_create (mi_raylib.h) and the CPU oracle consume.
 */
#ifndef MI_SCENE_HOST_H
#define MI_SCENE_HOST_H

#include "mi_raylib.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_host_scene mi_host_scene;   /* opaque; owns every array a desc points into */

/* Built-in scenes of the reference CLI (--scene box-simple | box | spheres, trace.cpp:357),
 * plus "monkey": the monkey bust alone in an open environment (BASELINE config 5).
 * `mesh_file` is the glTF-binary mesh placed on the short block for "box"
 * (assets/monkey_bust.glb in the reference, src/app_utils.cpp:264); ignored otherwise. */
int mi_host_scene_builtin(const char* scene_name, const char* mesh_file, mi_host_scene** out);

/* importScene(filename, loadNormals) (src/scene_utils.cpp:152-317, --mesh-file/--load-normals): a
 * complete scene with camera and materials from a Collada (.dae) file, camera moved to the origin,
 * materials re-interpreted with the reference's heuristics. */
int mi_host_scene_import(const char* file, int load_normals, mi_host_scene** out);

/* Build a scene from caller-provided arrays (same array contract as mi_scene_desc, but
 * bvh_nodes/max_leaf_depth are ignored and rebuilt). Used by tests with synthetic geometry. */
int mi_host_scene_from_arrays(const mi_scene_desc* geometry_only, mi_host_scene** out);

/* Fill `desc` with pointers into the host scene's storage and the reference CLI's default
 * render parameters (trace.cpp:343-366: 768x432, aa .25, path length 10, roulette depth 3,
 * 256 spp, seed 1442; fov from the scene's camera). The desc stays valid until destroy. */
int mi_host_scene_fill_desc(const mi_host_scene* scene, mi_scene_desc* desc);

void mi_host_scene_destroy(mi_host_scene* scene);

/* SAH BVH2 over axis-aligned boxes, one primitive per leaf, flattened depth-first with the
 * first child adjacent to its parent (CompactBVH2Node.hpp:60-63). lower/upper: 3 floats per
 * primitive. `nodes` must hold 2*n-1 entries. */
int mi_build_compact_bvh(const float* lower, const float* upper, const uint16_t* geom_ids,
                         const uint32_t* prim_ids, uint32_t n,
                         mi_bvh_node* nodes, uint32_t* num_nodes, uint32_t* max_leaf_depth);

/* initPerspectiveRayStream(rayStream, image, data, nullptr) + zeroRgb: window_w*window_h rays in
 * row-major window order, origin 0, un-jittered pinhole directions, u=row, v=col. */
int mi_init_ray_stream(const mi_scene_desc* desc, mi_trace_result* rays, size_t capacity);

/* scaleRgb (src/app_utils.cpp:55-59) */
void mi_scale_rgb(mi_trace_result* rays, size_t n, float scale);

/* ---- serialised scene -----------------------------------------------------------
 * The byte stream the reference's Serialiser<16> writes for a SceneRef (include/serialisation/
 * serialisation.hpp:33-53, src/IpuScene.cpp:51-53) and Deserialiser<16> aliases in place
 * (deserialisation.hpp:43-59). Layout is documented in ipu_ray_lib_amd/csrc/scene_blob.hpp. The blob
 * carries the eight scene arrays and maxLeafDepth .. samplesPerPixel; spheres, discs, rng seed, crop
 * window, pathTrace and device are not part of it.
 *   mi_scene_blob_size      bytes mi_scene_serialise will write for `desc`
 *   mi_scene_serialise      writes the blob; MI-host error if capacity is too small
 *   mi_scene_deserialise    fills `desc`'s array pointers with views INTO `blob` (which must be 16-byte
 *                           aligned and outlive the desc) and the eight scalars; other fields untouched.
 *                           A truncated blob fails with "Deserialiser encountered end of byte stream." */
size_t mi_scene_blob_size(const mi_scene_desc* desc);
int mi_scene_serialise(const mi_scene_desc* desc, uint8_t* out, size_t capacity, size_t* written);
int mi_scene_deserialise(const uint8_t* blob, size_t size, mi_scene_desc* desc, size_t* consumed);
/* Padding the format inserts before an object of alignment `align` at byte offset `offset`
 * (Serialiser::calculatePadding, Serialiser.hpp:30-39). */
uint32_t mi_blob_padding(uint32_t base_align, size_t offset, uint32_t align);

/* ---- ray-band sharding -----------------------------------------------------------
 * How a ray stream is dealt to the R replicas of a multi-GPU render and put together again; replaces the
 * round-robin batch pull of the reference's replicas (src/IpuScene.cpp:676-684, src/RayCallback.cpp:8-24).
 * The stream is cut into bands of `band` consecutive rays, band b belongs to replica b % R. One definition
 * (ipu_ray_lib_amd/csrc/ray_shard.hpp) serves these functions, mi_group_render and the Python ranks of bench.py.
 *   mi_shard_band_rays     band length for a stream of n rays rendered for a window `window_w` pixels wide: 8 rows
 *                          of the window when the stream is made of full rows, else 4096 rays
 *   mi_shard_count         rays replica r renders
 *   mi_shard_stream_index  out[k] = stream position of the k-th ray of replica r's stream
 *   mi_shard_frame_index   out[i] = position of stream ray i in the gathered buffer (replica 0's stream, then
 *                          replica 1's, ...): the de-interleave map the frame is assembled with            */
size_t mi_shard_band_rays(size_t n, uint32_t window_w);
size_t mi_shard_count(size_t n, size_t band, uint32_t replicas, uint32_t r);
int mi_shard_stream_index(size_t n, size_t band, uint32_t replicas, uint32_t r, uint64_t* out, size_t capacity);
int mi_shard_frame_index(size_t n, size_t band, uint32_t replicas, uint64_t* out);

/* ---- NIF assets ----------------------------------------------------------------
 * IpuScene::loadNifModel(assetPath) (src/IpuScene.cpp:174-187) reads <assetPath>/nif_metadata.txt
 * (NifMetaData.cpp:11-71) and <assetPath>/converted.hdf5, a Keras "Functional" model whose Dense layers
 * are stored under /model_weights/<layer>/<layer>/{kernel:0,bias:0} as float16 or float32
 * (src/keras/Hdf5Model.cpp:62-133). mi_host_nif_load does the same and hands the layers back in the
 * form mi_scene_set_nif takes (binary32 arrays; binary16 weights are widened exactly). HDF5 is reached
 * through the plugin libmi_nif_h5.so next to this library; if <assetPath>/converted.hdf5 is absent,
 * <assetPath>/nif_weights.bin is read instead: u32 numLayers, then per layer u32 rows, u32 cols,
 * u8 relu, u8 hasBias, f32 kernel[rows*cols] (Keras kernel:0 order), f32 bias[cols] if hasBias. */
typedef struct mi_host_nif mi_host_nif;

typedef struct mi_nif_desc {
  uint32_t num_layers;
  const float* const* kernels;   /* [num_layers] -> rows[i]*cols[i] floats, row-major */
  const float* const* biases;    /* [num_layers] -> cols[i] floats or NULL */
  const uint32_t* rows;
  const uint32_t* cols;
  const uint8_t* relu;           /* activation "relu" -> 1, "linear" -> 0 (NifModel.cpp:75-77, 324) */
  uint32_t embedding_dimension;  /* nif_metadata.txt */
  uint32_t hidden_size;          /* argument after --layer-size in train_command (0 if absent) */
  float max_value;
  float mean[3];                 /* eps already folded in when log_tonemap (NifMetaData.cpp:48-53) */
  int32_t log_tonemap;
  int32_t weights_are_half;      /* any kernel stored as float16 */
  const char* name;
  const char* source;            /* the weights file that was read */
} mi_nif_desc;

int mi_host_nif_load(const char* asset_path, mi_host_nif** out);
int mi_host_nif_describe(const mi_host_nif* nif, mi_nif_desc* desc);   /* pointers valid until destroy */
void mi_host_nif_destroy(mi_host_nif* nif);
const char* mi_host_nif_last_error(void);

const char* mi_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
