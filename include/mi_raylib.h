/*
 * mi_raylib.h — C ABI of the MI355X (gfx950) ray/path-trace hot path.
 *
 * This is the drop-in boundary for the device renderer of markp-gc/ipu_ray_lib:
 * what the reference reaches through `IpuScene` + `ipu_utils::GraphManager().run()`
 * (reference include/IpuScene.hpp:33-56, caller trace.cpp:270-336) is reached here
 * through five plain-C entry points. Plain pointers and sizes only; no C++ types,
 * no torch types, no exceptions cross this boundary. Every function returns an
 * int status (MI_OK == 0) and leaves a message retrievable with mi_last_error().
 *
 * POD layouts are those of the reference (probe sizes in SURVEY.md §8a):
 *   mi_trace_result == embree_utils::TraceResult   (84 B, include/embree_utils/geometry.hpp:253-259)
 *   mi_hit_record   == embree_utils::HitRecord     (64 B, geometry.hpp:226-251)
 *   mi_ray          == embree_utils::Ray           (32 B, geometry.hpp:212-224)
 *   mi_bvh_node     == CompactBVH2Node             (24 B, include/CompactBVH2Node.hpp:52-85)
 *   mi_material     == Material                    (36 B, include/Material.hpp:8-35)
 *   mi_mesh_info    == MeshInfo                    (16 B, include/Mesh.hpp:15-20)
 *   mi_geom_ref     == GeomRef                     (4 B,  include/Scene.hpp:29-34)
 * Spheres and discs cross the boundary as vptr-free PODs (the reference's host
 * Sphere/Disc objects carry vtable pointers, include/Primitives.hpp:36-82).
 */
#ifndef MI_RAYLIB_H
#define MI_RAYLIB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------- */
enum {
  MI_OK = 0,
  MI_ERR_INVALID_ARG = 1,   /* null pointer, bad size, inconsistent scene arrays */
  MI_ERR_DEVICE = 2,        /* HIP runtime error (message in mi_last_error)      */
  MI_ERR_NO_NIF = 3,        /* NIF entry point used before weights were loaded   */
  MI_ERR_IO = 4             /* file could not be read / parsed                   */
};

/* ---- POD types (layout == reference) ------------------------------------ */
typedef struct { float x, y, z; } mi_vec3;                      /* Vec3fa, 12 B, align 4 */

typedef struct {
  mi_vec3 origin; float t_min; mi_vec3 direction; float t_max;
} mi_ray;                                                        /* 32 B */

typedef struct {
  mi_ray r;
  uint32_t prim_id;          /* 0xFFFFFFFF = invalid */
  mi_vec3 normal;
  mi_vec3 throughput;
  uint16_t geom_id;          /* 0xFFFF = invalid */
  uint16_t flags;            /* MI_FLAG_ERROR | MI_FLAG_ESCAPED */
} mi_hit_record;                                                 /* 64 B */

typedef struct {
  mi_vec3 rgb;               /* path-trace: SUM over samples (caller divides by spp) */
  float u, v;                /* PixelCoord: u = image ROW, v = image COLUMN (src/app_utils.cpp:43) */
  mi_hit_record h;
} mi_trace_result;                                               /* 84 B */

#define MI_FLAG_ERROR   ((uint16_t)1)
#define MI_FLAG_ESCAPED ((uint16_t)2)
#define MI_INVALID_GEOM ((uint16_t)0xFFFF)
#define MI_INVALID_PRIM ((uint32_t)0xFFFFFFFFu)

typedef struct {
  float min_x, min_y, min_z;
  uint32_t prim_or_second_child;   /* leaf: primID; interior: index of second child (first child = this+1) */
  uint16_t dx, dy, dz;             /* IEEE binary16 bit patterns of the box extents, rounded up */
  uint16_t geom_id;                /* 0xFFFF => interior node */
} mi_bvh_node;                                                   /* 24 B, align 8 in the reference */

typedef struct {
  mi_vec3 albedo; float ior; mi_vec3 emission;
  int32_t type;                    /* 0 Diffuse, 1 Specular, 2 Refractive (Material::Type) */
  uint8_t emissive; uint8_t pad[3];
} mi_material;                                                   /* 36 B */

typedef struct {
  uint32_t first_index, first_vertex, num_triangles, num_vertices;
} mi_mesh_info;                                                  /* 16 B */

typedef struct { uint16_t index; uint8_t type; uint8_t pad; } mi_geom_ref;   /* type: 0 mesh, 1 sphere, 2 disc */

typedef struct { float x, y, z, radius; } mi_sphere;             /* radius2 = radius*radius is derived */
typedef struct { float nx, ny, nz, r, cx, cy, cz; } mi_disc;     /* r2 = r*r is derived */

/* ---- scene description: SceneRef + sphere/disc arrays + RuntimeConfig ---- */
/* Mirrors reference include/Scene.hpp:50-74 (SceneRef) field for field, plus the
 * two primitive arrays the IpuScene ctor takes separately (IpuScene.hpp:33-38).
 * All pointers are HOST pointers owned by the caller; mi_scene_create copies
 * what it needs to the device, so they may be freed after it returns.        */
typedef struct {
  const mi_geom_ref*  geometry;      uint32_t num_geometry;
  const mi_mesh_info* mesh_info;     uint32_t num_meshes;
  const uint16_t*     mesh_tris;     uint32_t num_tris;      /* 3 x u16 per triangle (Triangle, Primitives.hpp:21-25) */
  const mi_vec3*      mesh_verts;    uint32_t num_verts;
  const mi_vec3*      mesh_normals;  uint32_t num_normals;   /* 0, or == num_verts (--load-normals) */
  const uint32_t*     mat_ids;       uint32_t num_mat_ids;   /* one per geometry */
  const mi_material*  materials;     uint32_t num_materials;
  const mi_bvh_node*  bvh_nodes;     uint32_t num_nodes;
  uint32_t            max_leaf_depth;                        /* levels, root == 1 */
  const mi_sphere*    spheres;       uint32_t num_spheres;
  const mi_disc*      discs;         uint32_t num_discs;

  float    image_width, image_height;      /* FULL image size, also when a crop window is rendered */
  float    fov_radians;
  float    anti_alias_scale;
  uint32_t max_path_length;
  uint32_t roulette_start_depth;
  uint32_t samples_per_pixel;
  uint64_t rng_seed;
  int32_t  window_w, window_h, window_c, window_r;           /* CropWindow (Scene.hpp:22-27) */
  int32_t  path_trace;                                       /* bool */
  int32_t  device;                                           /* HIP device ordinal (RuntimeConfig analogue) */
} mi_scene_desc;

typedef struct mi_scene mi_scene;   /* opaque */

/* Render modes: the two trace vertices of codelets/TraceCodelets.cpp:170-316 */
enum { MI_MODE_SHADOW_TRACE = 0, MI_MODE_PATH_TRACE = 1 };

/* Replaces: IpuScene::IpuScene(...) + setRuntimeConfig (src/IpuScene.cpp:24-62, trace.cpp:297-309) */
int mi_scene_create(const mi_scene_desc* desc, mi_scene** out);

/* Same, from the reference's serialised scene: `blob` is the byte stream Serialiser<16> produced for the
 * SceneRef (IpuScene's `serialiser.bytes`, src/IpuScene.cpp:51-53; format in csrc/scene_blob.hpp), i.e.
 * what the reference broadcasts to every tile (src/IpuScene.cpp:200-216, 422-426). The eight arrays and
 * maxLeafDepth..samplesPerPixel are taken from the blob; spheres/discs, rng_seed, the crop window,
 * path_trace and device from `extras` (whose array fields are ignored). The bytes are validated like any
 * other scene and copied; no alignment requirement. */
int mi_scene_create_from_blob(const uint8_t* blob, size_t size, const mi_scene_desc* extras, mi_scene** out);

/* Replaces: IpuScene::~IpuScene */
void mi_scene_destroy(mi_scene* scene);

/* Replaces: GraphManager().run(ipuScene) -> IpuScene::build + execute (src/IpuScene.cpp:346-733),
 * host-buffer form. `rays` (n TraceResult, HOST memory) is read and overwritten in place exactly
 * as the reference overwrites the caller's ray stream (src/IpuScene.cpp:716-732). In path-trace
 * mode only rays[i].u/.v matter on input (camera rays are regenerated per sample on the device,
 * codelets/TraceCodelets.cpp:142-164); rgb comes back as the sum over samples_per_pixel samples.
 * In shadow-trace mode the given rays are traced as they are (Render.hpp:37-72).
 * `cb` (may be NULL) mirrors IpuScene::RayCallbackFn: it is called once per completed batch (see
 * mi_scene_set_ray_batch) with (user, batch_index, first_ray, ray_count), in batch order, on the calling
 * thread, while later batches are still being traced. The timed region (mi_trace_time_secs) here includes the
 * batch copies, which the pipeline overlaps with tracing. */
typedef void (*mi_ray_callback)(void* user, size_t batch_index, const mi_trace_result* rays, size_t count);
int mi_render(mi_scene* scene, int mode, mi_trace_result* rays, size_t n, mi_ray_callback cb, void* user);

/* Same operation on a DEVICE-resident ray stream (hipMalloc'ed / torch tensor memory), enqueued
 * on `hip_stream` (a hipStream_t passed as void*; NULL = the null stream). Asynchronous: returns
 * after enqueue. This is the entry bench.py times (inputs already resident in HBM).
 * Threading: a scene is thread-compatible (calls on ONE scene must not overlap in time on the host; different
 * scenes are independent). Launches of one scene enqueued on different streams own separate work counters and
 * partial-sum buffers and may execute concurrently; renders with a NIF environment share the scene's slot scratch
 * and are chained with an event, so they execute one after the other whatever streams they were enqueued on. */
int mi_render_device(mi_scene* scene, int mode, void* d_rays, size_t n, void* hip_stream);

/* Replaces: IpuScene::getTraceTimeSecs (IpuScene.hpp:55). Wall time of the last mi_render. */
double mi_trace_time_secs(const mi_scene* scene);

/* Counters accumulated by render calls since scene creation / last reset: number of
 * CompactBvh::intersect + ::occluded casts, BVH nodes visited, primitive (leaf) tests.
 * Synchronises the device. counts[0]=casts, [1]=nodes visited, [2]=leaf tests, [3]=paths. */
int mi_get_counters(mi_scene* scene, uint64_t counts[4]);
int mi_reset_counters(mi_scene* scene);

/* Diagnostics of the phase-scheduled path-trace kernel (only filled when the instrumented kernel
 * variant is selected with MI_RAYLIB_FULL_STATS=1): for each phase NODE, LEAF, SHADE, GEN the number
 * of wave-level executions and the sum of lanes active in them:
 * stats[0..7] = {node_iters, node_lanes, leaf_iters, leaf_lanes, shade_iters, shade_lanes, gen_iters, gen_lanes},
 * stats[8..11] = shader cycles summed over waves spent in {traversal loop, SHADE, GEN, whole kernel loop}.
 * No reference counterpart (the IPU has no SIMT lanes); used by DESIGN.md's occupancy table. */
int mi_get_phase_stats(mi_scene* scene, uint64_t stats[12]);
/* Scheduler bookkeeping of the path-pool kernel (kernel 3, instrumented build only): {loop iterations, refill turns,
 * lanes refilled, idle iterations, lost ring claims, traversal bursts, lanes walking at burst start, cycles in refill}. */
int mi_get_pool_stats(mi_scene* scene, uint64_t stats[8]);
/* Measurement only: enqueues, on a stream of the scene's own, ONE wave that samples the work counter of `hip_stream`'s persistent
 * launches n times, period_ticks (100-MHz ticks) apart, into d_samples (device memory, 2 n words: {s_memrealtime, counter}). Call it
 * right before mi_render_device on `hip_stream`; read the samples after a device synchronise (tools/launch_progress.py: the rate at which
 * a launch hands its work units out over its life). No reference counterpart. */
int mi_debug_launch_progress(mi_scene* scene, void* hip_stream, uint64_t* d_samples, uint32_t n, uint32_t period_ticks);
/* Diagnostics of NIF renders (only filled while the scene option "nif_timing" is 1): HIP events bracket every launch of
 * the MLP kernel on the render's stream. out[0] = milliseconds spent in MLP launches since the last call, out[1] = number
 * of launches. Synchronises the device and clears the record. tools/bench_config5.py reports the MLP's share of a frame
 * from it. No reference counterpart. */
int mi_get_nif_timing(mi_scene* scene, double out[2]);
/* The shader clock the last launch of the register-resident MLP kernel (K3a, csrc/nif_asm_kernel.hpp) ran at: out[0] = shader
 * cycles (s_memtime), out[1] = ticks of the constant 100-MHz counter (s_memrealtime) that the first wave of its first workgroup -
 * which lives as long as the launch - spent in the kernel; clock in GHz = out[0] / out[1] / 10. Both 0 when the scene's network
 * runs nif_mlp_kernel (no such record). Synchronises the device. An MFMA-dense kernel runs at the clock the chip's power
 * management leaves it, which differs from box to box: bench.py quotes this beside the kernel's time. No reference counterpart. */
int mi_get_nif_clock(mi_scene* scene, uint64_t out[2]);

/* Replaces: IpuScene::loadNifModel (src/IpuScene.cpp:174-187) with the weights handed over as
 * arrays (the file side — nif_metadata.txt + Keras-H5 — is mi_host_nif_load in mi_scene_host.h).
 * Dense layer i has kernel[i] of shape [rows[i] x cols[i]] row-major (Keras kernel:0 layout,
 * y = x·W + b) and bias[i] of cols[i] floats (NULL = no bias); relu[i] != 0 applies ReLU.
 * Where a layer's rows != current activation width, the Fourier features are re-concatenated
 * to the activations first (NifModel.cpp:306-309). Decode: y*max + mean, then exp if
 * log_tonemap (NifModel.cpp:222-246); `mean` must already have eps folded in
 * (NifMetaData.cpp:48-53). Output channels are BGR (codelets/TraceCodelets.cpp:376). */
int mi_scene_set_nif(mi_scene* scene, uint32_t num_layers,
                     const float* const* kernels, const float* const* biases,
                     const uint32_t* rows, const uint32_t* cols, const uint8_t* relu,
                     uint32_t embedding_dimension, float max_value, const float mean[3],
                     int32_t log_tonemap);

/* Replaces: IpuScene::setHdriRotation (degrees) / setMaxNifBatchSize (src/IpuScene.cpp:334-344). */
int mi_scene_set_hdri_rotation(mi_scene* scene, float degrees);
int mi_scene_set_max_nif_batch(mi_scene* scene, size_t rays_per_batch);

/* Replaces: the `raysPerWorker` constructor argument / --rays-per-worker (src/IpuScene.cpp:360-361, 110-172):
 * mi_render cuts the host ray stream into batches of this many rays (the reference: 1440 tiles x 6 workers x
 * raysPerWorker), pipelines their upload / trace / download on two HIP streams and calls the ray callback once
 * per finished batch, in batch order. 0 (default) = one batch. Results do not depend on the batch size. */
int mi_scene_set_ray_batch(mi_scene* scene, size_t rays_per_batch);

/* Kernel selection / tuning of ONE scene (no reference counterpart; the nearest is the reference's per-run
 * RuntimeConfig + codelet build flags, trace.cpp:297-309). Every scene carries its own copy: defaults, overridden
 * by the MI_RAYLIB_* environment variables as they stand when the scene is created, then by this call. Keys and the
 * values each accepts (anything else: MI_ERR_INVALID_ARG, the option keeps its value):
 *   "kernel"        0 | 1 | 2 | 3   nested-loop / phase-scheduled (default) / phase-scheduled + LDS-staged nodes / path pool
 *   "waves"         4 | 5 | 6 | 7   waves per SIMD the default kernel is built for (6: the 80-VGPR build, the default; 5 - the 96-VGPR build -,
 *                                   4 and 7 - 72 VGPRs, 21 words of LDS per lane, measured 1 % slower - only in the variants build)
 *   "merge"         0 | 1           kernel 1: SHADE and GEN served by one turn (1, the default) or by two, at five waves per SIMD - the
 *                                   default kernel up to round 3 (0: variants build only)
 *   "spec"          0 | 1           kernel 1: lanes walk on past ONE pending primitive test
 *   "full_stats"    0 | 1           instrumented kernels: node / leaf-test counters, phase occupancy
 *   "tune"          "leafAt,shadeAt,genAt[,burst,keep8,dbl,maxExtra(<=7),leafThenNode,prio,leafP,probe]"   scheduling weights of kernel 1
 *   "pool_waves"    4 | 8 | 16      kernel 3: waves per workgroup
 *   "pool_tune"     "leafAt,burst,retireAt,refillMin,shadeW,genW[,dbl,maxExtra,leafThenNode,prio]"
 *   "tiles"         0 | 1           walk row-structured streams in 8x8 pixel tiles
 *   "seg_budget_kb" N >= 1          partial-sum buffer budget per launch
 *   "nif_spl"       0..1024         NIF samples per launch, rounded up to whole segments (0 = default: 512, memory permitting - 48 B of
 *                                   slots per sample and pixel; a launch is never shorter than its longest work unit: 128 samples per launch cost
 *                                   config 5 5 % of its frame)
 *   "nif_shape"     auto | a8 | b4 | w6 | t6 | t4    which NIF MLP kernel runs (b4 = K3a's dataflow with four waves of 64 rays: 3 % slower): auto (default) = a8 where its generated body covers the network
 *                                   (the reference's 6 x 320 shape), w6 otherwise; a8 = K3a, the hand-scheduled register-resident kernel
 *                                   (csrc/nif_asm_kernel.hpp); w6 | t6 | t4 = workgroup shapes of nif_mlp_kernel; the variants build also takes
 *                                   r8 | r8s = K3r (csrc/nif_regs_kernel.hpp: measured slower; refused by the shipped library)
 *   "nif_generations" 1..4096       nif_mlp_kernel: workgroups launched per resident slot (measurement knob; default 64)
 *   "root_start"    0 | 1           a cast whose origin lies strictly inside the root's box starts at node 1 (default 1; exact either way)
 *   "say_grid"      0 | 1           print every persistent launch's grid to stderr
 *   "pin"           0 | 1           page-lock the caller's stream for the duration of mi_render
 *   "nif_overlap"   auto | 0 | 1    NIF renders trace sample batch b + 1 beside the MLP of batch b (two slot sets, a second stream). auto (the
 *                                   default) = only beside nif_mlp_kernel: K3a / K3b hold every register of their compute unit, nothing runs
 *                                   beside them, and one slot set leaves the memory for longer launches
 *   "nif_split"     0..1024         with the overlap: compute units the trace launches of batches 1.. get for themselves (two CU-masked
 *                                   streams, hipExtStreamCreateWithCUMask; the MLP and accumulate passes keep the rest). 0 = off, the default:
 *                                   the MLP is power-limited and loses as much as the hidden launch was worth (-2 ... +5 % by box)
 *   "nif_first_test" 0 | 1          NIF renders: a cast's first box test runs in the turn that sets the cast up instead of in a NODE turn
 *                                   (default 0: measured neutral)
 *   "nif_trace_wgs" 0..16           with nif_overlap: workgroups per compute unit of a trace launch that runs beside the previous batch's MLP
 *                                   (0 = all that stay resident, the default)
 *   "nif_timing"    0 | 1           bracket every MLP launch of a NIF render with HIP events (mi_get_nif_timing)
 *   "leaf_rot"      0 | 1           scenes without vertex normals: the default kernel reads primitive records pre-rotated for the cast's shear axis (default 1)
 *   "lean_hit"      0 | 1           scenes without vertex normals run the build of the default kernel that carries no barycentrics (default 1)
 *   "coords"        0 | 1           (pixel, segment) work units read the pixel's (u, v) from a compact copy of the stream gathered once
 *                                   per launch, not from the 84-byte record (default 1: a third of the HBM traffic)
 *   "cus"           0..4096         compute units the launch grids are sized for (0 = what the device reports; grids are
 *                                   units x workgroups resident per unit, asked of the runtime per kernel)
 * None of them changes a result bit. Two further keys select ARITHMETIC:
 *   "double_fallback" 0 | 1         the reference built with -DALLOW_DOUBLE_FALLBACK=1 (CMakeLists.txt:13,34-41; src/Mesh.cpp:38-51):
 *                                   edge functions that are exactly zero in binary32 are recomputed in binary64. Results are those
 *                                   of the reference's CPU path built the same way, bit for bit (default 0 = the reference default)
 *   "fast"            0 | 1         tolerance tier for plain path-trace renders of the default kernel: box test as FMAs, triangle
 *                                   test contracted, v_rcp_f32 in the cast set-up. NOT bit-exact: first hits name the same primitive
 *                                   with distance / point within 1e-6; see tests/test_gpu_parity.py (test_fast_tier_...) for the
 *                                   stated tolerance. Never the default. */
int mi_scene_set_option(mi_scene* scene, const char* key, const char* value);

/* The NIF environment evaluated stand-alone on device arrays: for i<n, bgr[i*3..] =
 * decode(MLP(fourier(u[i], v[i]))). Replaces NifModel::buildInference's execModel
 * (NifModel.cpp:249-356). d_u, d_v, d_bgr are DEVICE pointers. */
int mi_nif_infer_device(mi_scene* scene, const float* d_u, const float* d_v, float* d_bgr,
                        size_t n, void* hip_stream);

/* ---- several GPUs in one process -------------------------------------------------------
 * Replaces: the replicas of an IpuScene (RuntimeConfig.numIpus / numReplicas, trace.cpp:297-309; scene replicated per
 * device src/IpuScene.cpp:473-483; replicas pull disjoint ray batches round-robin from the one stream :676-684; every
 * batch is written back into the caller's stream :699-732 and handed to the callback, src/RayCallback.cpp:8-24).
 * mi_group_create builds one mi_scene per entry of `devices` (HIP ordinals; an ordinal may repeat - several replicas
 * then share that GPU, which is how the path is rehearsed on a one-GPU box). mi_group_render cuts the host stream into
 * ray batches (mi_group_set_ray_batch; default: one batch = the whole stream) and, for each batch: deals it in bands
 * (8 rows of the render window per band, band b to replica b % R: ipu_ray_lib_amd/csrc/ray_shard.hpp, also exported as
 * mi_shard_* by libmi_scene_host.so), uploads every replica's bands with one strided copy, traces every share on its
 * own HIP stream with no exchange while it renders, moves the finished shares to the first replica's device with ONE
 * RCCL group call (ncclSend / ncclRecv over xGMI), and copies them from there straight into their places in `rays`
 * (strided copies: no de-interleave pass, no second frame buffer). The callback is called once per batch, in batch
 * order, on the calling thread, as soon as that batch is home and while the next one is being traced. Every pixel owns
 * its RNG streams, so the result is bit-identical for any number of replicas and any batch size.
 * `transport`: 0 = RCCL as soon as more than one device takes part (peer copies otherwise), 1 = RCCL always, 2 = peer
 * copies only. RCCL is loaded with dlopen when the first group needs it.
 * mi_group_scene hands out a replica's scene for the per-scene setters (mi_scene_set_nif, mi_scene_set_option, ...),
 * which must be applied to every replica alike. */
typedef struct mi_group mi_group;
int mi_group_create(const mi_scene_desc* desc, const int32_t* devices, uint32_t num_replicas, int32_t transport, mi_group** out);
void mi_group_destroy(mi_group* group);
uint32_t mi_group_size(const mi_group* group);
mi_scene* mi_group_scene(mi_group* group, uint32_t replica);
int mi_group_set_ray_batch(mi_group* group, size_t rays_per_batch);
int mi_group_render(mi_group* group, int mode, mi_trace_result* rays, size_t n, mi_ray_callback cb, void* user);
/* The stages of mi_group_render one by one, for callers that keep the shares RESIDENT on the devices (the reference
 * keeps a batch in remote buffers between executions the same way, src/IpuScene.cpp:399-409): upload deals and copies
 * the whole stream (one batch); trace renders every share where it lies and gathers the shares on the first replica's
 * device (rgb keeps accumulating from call to call, as with mi_render_device); download copies the gathered shares into
 * `rays` (n must be the uploaded count). bench.py --gpus N times mi_group_trace: inputs resident in HBM, the RCCL
 * gather inside the timed region. mi_group_gathered_device exposes the gathered buffer (shares replica after replica;
 * offsets[r] = first record of replica r's share, up to num_offsets entries) for device-side consumers. */
int mi_group_upload(mi_group* group, const mi_trace_result* rays, size_t n);
int mi_group_trace(mi_group* group, int mode);
int mi_group_download(mi_group* group, mi_trace_result* rays, size_t n);
int mi_group_gathered_device(mi_group* group, void** d_gathered, uint64_t* offsets, uint32_t num_offsets);
double mi_group_trace_time_secs(const mi_group* group);               /* wall time of the last mi_group_render / mi_group_trace */
int mi_group_get_counters(mi_group* group, uint64_t counts[4]);       /* summed over the replicas */
int mi_group_reset_counters(mi_group* group);
/* What the last mi_group_render (or stage call) moved: info[0] = RCCL send/recv pairs, info[1] = peer copies,
 * info[2] = bands dealt, info[3] = host->device copies issued, info[4] = device->host copies issued. */
int mi_group_last_transfer(const mi_group* group, uint64_t info[5]);
/* How long the last batch's gather took on the first replica's device, in milliseconds (HIP events on its stream: from
 * "its own share is traced" to "the last share has arrived" - a slower peer's remaining trace time included). Waits for
 * that gather. And the communicator the group built: `*distinct` = number of distinct devices (= RCCL ranks when the
 * transport is RCCL), the ordinals themselves in devices[0..capacity), root first. A multi-GPU record that reports
 * distinct == 1 was a one-GPU rehearsal (src/IpuScene.cpp:676-684 spreads the batches over real replicas). */
int mi_group_last_gather_ms(mi_group* group, double* ms);
int mi_group_devices(const mi_group* group, int32_t* devices, uint32_t capacity, uint32_t* distinct);

/* Thread-local message for the last failing call on this thread. Never NULL. */
const char* mi_last_error(void);

/* Library / build identification, e.g. "mi_raylib 0.1 gfx950 contract=off". */
const char* mi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MI_RAYLIB_H */
