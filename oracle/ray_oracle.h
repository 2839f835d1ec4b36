/*
 * ray_oracle.h — CPU restatement (plain C11) of the ray-parallel hot path of
 * markp-gc/ipu_ray_lib. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the CHECKER. Nothing under ipu_ray_lib_amd/ links,
 * loads or calls it; the product path has no CPU fallback.
 *
 * Each function cites the reference file:line it restates (paths relative to the
 * reference checkout). Pinning status per function group:
 *   [REF]   checked against the reference's own sources compiled here without any
 *           stand-in header (oracle/_ref: ext/math/sincos.cpp, xoshiro.hpp,
 *           embree_utils/geometry.hpp, BxDF.hpp, geometric_sampling.hpp);
 *   [PROBE] checked against values the reference itself produced, recorded in
 *           SURVEY.md §8a/§8c (the reference files that hold these functions need
 *           Eigen::half, which this image lacks, so they cannot be built here);
 *   [UNPINNED] no reference-produced value exists: "parity unpinned".
 */
#ifndef RAY_ORACLE_H
#define RAY_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } ovec3;
typedef struct { ovec3 origin; float tMin; ovec3 direction; float tMax; } oray;             /* 32 B */
typedef struct {
  oray r; uint32_t primID; ovec3 normal; ovec3 throughput; uint16_t geomID; uint16_t flags;
} ohit;                                                                                      /* 64 B */
typedef struct { ovec3 rgb; float u, v; ohit h; } otrace;                                    /* 84 B */

typedef struct {
  float min_x, min_y, min_z;
  uint32_t link;                 /* leaf: primID, interior: secondChildIndex */
  uint16_t dx, dy, dz;           /* binary16 bit patterns */
  uint16_t geomID;               /* 0xFFFF => interior */
} onode;                                                                                     /* 24 B */

typedef struct { ovec3 albedo; float ior; ovec3 emission; int32_t type; uint8_t emissive; uint8_t pad[3]; } omaterial;
typedef struct { uint32_t firstIndex, firstVertex, numTriangles, numVertices; } omeshinfo;
typedef struct { uint16_t index; uint8_t type; uint8_t pad; } ogeomref;
typedef struct { float x, y, z, radius; } osphere;
typedef struct { float nx, ny, nz, r, cx, cy, cz; } odisc;

/* Field-for-field the same layout as mi_scene_desc (include/mi_raylib.h) so a test can
 * describe a scene once; declared separately because the oracle shares no code with the
 * product. */
typedef struct {
  const ogeomref*  geometry;   uint32_t nGeometry;
  const omeshinfo* meshInfo;   uint32_t nMeshes;
  const uint16_t*  meshTris;   uint32_t nTris;
  const ovec3*     meshVerts;  uint32_t nVerts;
  const ovec3*     meshNormals;uint32_t nNormals;
  const uint32_t*  matIDs;     uint32_t nMatIDs;
  const omaterial* materials;  uint32_t nMaterials;
  const onode*     bvhNodes;   uint32_t nNodes;
  uint32_t         maxLeafDepth;
  const osphere*   spheres;    uint32_t nSpheres;
  const odisc*     discs;      uint32_t nDiscs;
  float imageWidth, imageHeight, fovRadians, antiAliasScale;
  uint32_t maxPathLength, rouletteStartDepth, samplesPerPixel;
  uint64_t rngSeed;
  int32_t winW, winH, winC, winR;
  int32_t pathTrace;
  int32_t device;
} oscene;

enum { O_FLAG_ERROR = 1, O_FLAG_ESCAPED = 2 };
enum { O_GEOM_MESH = 0, O_GEOM_SPHERE = 1, O_GEOM_DISC = 2 };
enum { O_MAT_DIFFUSE = 0, O_MAT_SPECULAR = 1, O_MAT_REFRACTIVE = 2 };

/* traversal statistics filled by the instrumented entry points (may be NULL) */
typedef struct { uint64_t casts, nodesVisited, leafTests, paths; } ostats;

/* ---- scalar / vector building blocks (exported for known-answer tests) ---- */
float    o_half_to_float(uint16_t h);
uint16_t o_float_to_half_rne(float f);
uint16_t o_round_to_half_not_smaller(float f);          /* precision_utils.hpp:39-47 */
float    o_gamma(int i);                                 /* precision_utils.hpp:20-23 */
float    o_ray_epsilon(void);                            /* precision_utils.hpp:25 */
uint32_t o_maxi(ovec3 v);                                /* geometry.hpp:115-121 */
float    o_maxc(ovec3 v);                                /* geometry.hpp:123-125 */
void     o_sincos(float x, float* s, float* c);          /* ext/math/sincos.cpp:236-355, flg=0 */
void     o_orthonormal_system(ovec3 n, ovec3* b0, ovec3* b1);   /* geometry.hpp:147-159 */

uint64_t o_splitmix64(uint64_t z);                       /* xoshiro.hpp:22-28 */
void     o_xoshiro_seed(uint64_t s[2], uint64_t seed);   /* xoshiro.hpp:31-34 */
uint64_t o_xoshiro_next(uint64_t s[2]);                  /* xoshiro.hpp:36-46 */
void     o_xoshiro_jump(uint64_t s[2]);                  /* xoshiro.hpp:51-66 */
float    o_xoshiro_uniform01(uint64_t s[2]);             /* xoshiro.hpp:68-80 */

int  o_slab(float invDir, float origin, float slabMin, float slabMax, float* t0, float* t1);  /* CompactBVH2Node.hpp:14-50 */
int  o_node_intersect(const onode* n, ovec3 o, ovec3 invDir, float* t0, float* t1);            /* CompactBVH2Node.cpp:5-22 */

typedef struct { ovec3 o; ovec3 dir; uint32_t ix, iy, iz; float sx, sy, sz; } oshear;
void o_ray_shear(const oray* ray, oshear* out);          /* Primitives.cpp:5-22 */
/* returns t (0 = miss) and barycentrics; Mesh.cpp:6-104 with ALLOW_DOUBLE_FALLBACK=0 */
float o_intersect_triangle(ovec3 p0, ovec3 p1, ovec3 p2, const oshear* tf, float tFar, float bary[3]);
/* ALLOW_DOUBLE_FALLBACK (CMakeLists.txt:13,34-41; Mesh.cpp:38-51) as a process-wide switch of this library: 0 (default) / 1 */
void o_set_double_fallback(int on);
int o_get_double_fallback(void);
float o_sphere_intersect(const osphere* s, const oray* ray);   /* Primitives.cpp:24-47; 0 = miss */
float o_disc_intersect(const odisc* d, const oray* ray);       /* Primitives.cpp:49-67; 0 = miss */

void  o_offset_ray(oray* r, ovec3 n);                    /* Render.hpp:29-33 */
ovec3 o_pixel_to_ray_dir(float x, float y, float w, float h, float tanTheta);   /* Render.hpp:74-85 */
void  o_sample_disc_concentric(float u1, float u2, float* x, float* y);         /* geometric_sampling.hpp:8-31 */
ovec3 o_cosine_sample_hemisphere(float u1, float u2);    /* geometric_sampling.hpp:42-47 */
ovec3 o_sample_diffuse(ovec3 normal, float u1, float u2);/* BxDF.hpp:11-30 */
ovec3 o_reflect(ovec3 dir, ovec3 normal);                /* BxDF.hpp:33-37 */
float o_schlick(float cosTheta, float ri);               /* BxDF.hpp:39-46 */
ovec3 o_refract(ovec3 dir, ovec3 normal, float ndotr, float ri);   /* BxDF.hpp:48-55 */
int   o_dielectric(const oray* ray, ovec3 normal, float ri, float u1, ovec3* outDir); /* BxDF.hpp:57-75; returns refracted */
int   o_evaluate_roulette(float u1, ovec3* throughput);  /* geometric_sampling.hpp:56-63 */

/* deterministic natural log used ONLY by the per-pixel Gaussian pixel jitter (no reference
 * counterpart: the IPU uses a hardware Gaussian, the CPU path libstdc++'s normal_distribution) */
float o_logf_det(float x);
void  o_gauss2(uint64_t s[2], float* g0, float* g1);

/* ---- BVH queries (CompactBvh.hpp:33-139), instrumented ------------------- */
typedef struct { int hit; uint32_t geomID, primID; float t; ovec3 normal; } ointersection;
ointersection o_bvh_intersect(const oscene* sc, const oray* ray, ostats* st);
int           o_bvh_occluded(const oscene* sc, const oray* ray, ostats* st);

/* ---- renderers ----------------------------------------------------------- */
/* initPerspectiveRayStream without jitter (src/app_utils.cpp:19-47, gen == nullptr), then zeroRgb */
void o_init_ray_stream(const oscene* sc, otrace* rays);

/* traceShadowRay over the stream (Render.hpp:37-72, trace.cpp:246-256): light (18,257,-1060), ambient .05 */
void o_shadow_trace(const oscene* sc, otrace* rays, size_t n, int numThreads, ostats* st);

/* Tier-1 path trace: codelets/TraceCodelets.cpp:184-263 == trace.cpp:115-188 with one
 * xoshiro128** stream PER PIXEL (the scheme the GPU kernel uses): for every
 * ray, samplesPerPixel samples; rgb accumulates the sum. */
void o_path_trace_pixel_rng(const oscene* sc, otrace* rays, size_t n, int numThreads, ostats* st);

/* Tier-2 path trace: renderCPU's exact structure (trace.cpp:236-245) with ONE shared
 * generator consumed sequentially (what the reference does under OMP_NUM_THREADS=1), pixel
 * jitter from a restatement of libstdc++'s normal_distribution<float>. rgb = sum over samples. */
void o_path_trace_shared_rng(const oscene* sc, otrace* rays, size_t n, ostats* st);

/* ---- escaped rays + NIF (codelets/TraceCodelets.cpp:321-382, NifModel.cpp:186-327) ---- */
void o_escaped_uv(const otrace* rays, size_t n, float azimuthRotation, float* u, float* v);
typedef struct {
  uint32_t numLayers;
  const float* const* kernels;     /* [rows x cols] row-major */
  const float* const* biases;      /* cols or NULL */
  const uint32_t* rows; const uint32_t* cols; const uint8_t* relu;
  uint32_t embeddingDimension; float maxValue; float mean[3]; int32_t logTonemap;
  int32_t halfFeatures;            /* 1: Fourier features rounded through binary16 as on the IPU (NifModel.cpp:212-216) */
  int32_t halfWeightsActs;         /* 1: round matmul inputs to binary16 (models the fp16 MFMA path), accumulate in f32 */
} onif;
void o_nif_infer(const onif* nif, const float* u, const float* v, size_t n, float* bgr);
void o_apply_env(otrace* rays, size_t n, const float* bgr);   /* PostProcessEscapedRays */

/* Tier-1 stream definition: samples per segment for a render of samplesPerPixel */
uint32_t o_segment_samples(uint32_t samplesPerPixel);

/* Per-sample NIF path trace (src/IpuScene.cpp:571-583: Repeat(spp){trace; pre; nif; post}) with per-pixel RNG */
void o_path_trace_nif_pixel_rng(const oscene* sc, const onif* nif, float azimuthRotation,
                                otrace* rays, size_t n, int numThreads, ostats* st);

const char* o_version(void);

#ifdef __cplusplus
}
#endif
#endif
