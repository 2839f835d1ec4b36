// raylib.hip — C ABI (include/mi_raylib.h) over the gfx950 kernels: scene validation + upload,
// kernel launches, counters, timing. This file is the whole device library (libmi_raylib.so).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mi_raylib.h"
#include "ray_math.h"
#include "trace_kernels.hpp"
#include "trace_wavefront.hpp"
// MI_RAYLIB_VARIANTS=1 (libmi_raylib_variants.so, the test build): the kernel families that were built, measured and not
// made the default - LDS-staged nodes (kernel 2), the path pool (kernel 3), the speculative walk (spec), the 4-wave and
// the runtime-weights instantiations (waves, tune), the register-resident MLP kernel K3r (nif_shape r8 / r8s) - stay
// selectable and parity-tested there. The
// shipped library carries the default path, its instrumented build, the nested-loop kernel, the two arithmetic options and K3.
#ifndef MI_RAYLIB_VARIANTS
#define MI_RAYLIB_VARIANTS 0
#endif
#if MI_RAYLIB_VARIANTS
#include "trace_pool.hpp"
#include "nif_regs_kernel.hpp"      // K3r, the register-resident MLP kernel (round 4: correct, slower than K3; nif_shape r8 / r8s)
#endif
#include "nif_kernels.hpp"
#include "nif_asm_kernel.hpp"      // K3a, the hand-scheduled register-resident MLP kernel (nif_shape a8)
#include "scene_blob.hpp"

using namespace mi;

static_assert(sizeof(mi_trace_result) == 84, "TraceResult layout");
static_assert(sizeof(mi_hit_record) == 64, "HitRecord layout");
static_assert(sizeof(mi_ray) == 32, "Ray layout");
static_assert(sizeof(mi_bvh_node) == 24, "CompactBVH2Node layout");
static_assert(sizeof(mi_material) == 36, "Material layout");
static_assert(sizeof(mi_mesh_info) == 16 && sizeof(mi_geom_ref) == 4, "MeshInfo/GeomRef layout");
static_assert(offsetof(mi_trace_result, u) == 12 && offsetof(mi_trace_result, h) == 20, "TraceResult offsets");
static_assert(offsetof(mi_hit_record, prim_id) == 32 && offsetof(mi_hit_record, normal) == 36 &&
              offsetof(mi_hit_record, throughput) == 48 && offsetof(mi_hit_record, geom_id) == 60 &&
              offsetof(mi_hit_record, flags) == 62, "HitRecord offsets");

namespace {

thread_local std::string g_err;

struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };
struct ArgError : std::runtime_error { using std::runtime_error::runtime_error; };

#define HIP_CHECK(expr)                                                                      \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      throw DeviceError(std::string(#expr) + ": " + hipGetErrorString(_e));                  \
  } while (0)

template <class F>
int guarded(F&& f) {
  try { f(); g_err.clear(); return MI_OK; }
  catch (const ArgError& e) { g_err = e.what(); return MI_ERR_INVALID_ARG; }
  catch (const DeviceError& e) { g_err = e.what(); return MI_ERR_DEVICE; }
  catch (const std::exception& e) { g_err = e.what(); return MI_ERR_DEVICE; }
  // nothing crosses the C ABI (include/ipu_utils.hpp:590-595: an error becomes a status, never a crash): whatever else a
  // callback or a library below us throws is reported like a device error
  catch (...) { g_err = "unknown exception (not derived from std::exception)"; return MI_ERR_DEVICE; }
}

template <class T>
T* upload(const std::vector<T>& v) {
  if (v.empty()) return nullptr;
  T* d = nullptr;
  HIP_CHECK(hipMalloc(&d, v.size() * sizeof(T)));
  HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

const float* hostSinTable() {
  static const uint32_t bits[92] = {MI_SIN_TABLE_BITS};
  static float tbl[92];
  static bool init = false;
  if (!init) { memcpy(tbl, bits, sizeof tbl); init = true; }
  return tbl;
}

}  // namespace

// Kernel selection and tuning of ONE scene: defaults, overridden by the MI_RAYLIB_* environment variables as they
// stand when the scene is created (read once, into the scene) and by mi_scene_set_option afterwards. Nothing here is
// process-global: creating or tuning scene B never changes what scene A launches.
// Samples a NIF render traces per launch (option nif_spl; the default, memory permitting - ensureScratch). A launch of the persistent
// kernel cannot be shorter than its longest work unit - 64 samples of one pixel, ~14 ms for a pixel on config 5's mesh - and a
// 1440^2 launch of 128 samples is only ten units per lane: config 5's trace launches took 22 ms per 128 samples that way, 15 with 256
// per launch, 11 with 512 and no less with 768 or 1 024 (profiles/r05_config5_launch_ab.txt); 512 samples are 48 B x 512 per pixel, 51 GB
// of a 1440^2 frame's 288.
constexpr uint32_t kNifSplDefault = 512, kNifSplMax = 1024;
struct SceneOptions {
  bool fullStats = false;          // MI_RAYLIB_FULL_STATS / "full_stats": instrumented kernel variants (node/leaf counters, phase occupancy)
  WaveTune tune = kDefaultTune;
  int kernelChoice = 1;            // MI_RAYLIB_KERNEL / "kernel": 0 = nested-loop kernel, 1 = wavefront (global nodes); variants build: 2 = wavefront + LDS-staged nodes, 3 = path pool (trace_pool.hpp)
#if MI_RAYLIB_VARIANTS
  PoolTune poolTune;               // MI_RAYLIB_POOL_TUNE / "pool_tune": leafAt,burst,retireAt,refillMin,shadeW,genW[,dbl,maxExtra,leafThenNode,prio]
#endif
  uint32_t cus = 0;                // "cus": compute units the launch grids are sized for (0 = what the device reports)
  std::string why;                 // why the last set() returned false
  int poolWaves = 4;               // MI_RAYLIB_POOL_WAVES / "pool_waves": waves per workgroup of the path-pool kernel, 4 | 8 | 16 (400 | 800 | 1600 slots)
  int wavesPerSimd = 6;            // MI_RAYLIB_WAVES / "waves": 6 = the 80-VGPR build of the default kernel; variants build: 5 = the 96-VGPR build, 4 = the 4-wave build
  bool mergeTurns = true;          // "merge" (variants build): 0 = SHADE and GEN as two turns, five waves - the default kernel up to round 3
  bool specLeaf = false;           // MI_RAYLIB_SPEC / "spec": lanes walk on past ONE pending primitive test (trace_wavefront.hpp, SPEC)
  bool tiles = true;               // MI_RAYLIB_NO_TILES / "tiles": walk row-structured streams in 8x8 pixel tiles
  size_t segBudgetKb = (size_t)8 * 1024 * 1024;   // MI_RAYLIB_SEG_BUDGET_KB / "seg_budget_kb": partial-sum buffer budget per launch
  uint32_t nifSamplesPerLaunch = 0;               // MI_RAYLIB_NIF_SPL / "nif_spl": 0 = default (128, memory permitting)
  bool pin = true;                 // MI_RAYLIB_PIN / "pin": page-lock the caller's stream for the duration of mi_render
  uint32_t nifShape = 7;           // MI_RAYLIB_NIF_SHAPE / "nif_shape": 7 = auto (default: K3a where its generated body covers the network, else w6), 6 = a8 (K3a, nif_asm_kernel.hpp),
                                   // 0 = w6, 1 = t6, 2 = t4 (nif_mlp_kernel's workgroup shapes); 4 = r8, 5 = r8s (K3r, nif_regs_kernel.hpp: measured slower, selectable in the variants build)
  bool leafRot = true;             // "leaf_rot": scenes without vertex normals read primitive records pre-rotated for the cast's shear axis (GLeafRot: 18 selects per triangle test become 6; -2.4 %, profiles/r05_k1w_leaf_ab.txt)
  bool leanHit = true;             // "lean_hit": scenes without vertex normals run the default kernel's build that carries no barycentrics (BARY = false)
  bool coords = true;              // "coords": (pixel, segment) atoms read the pixel coordinates from a compact copy of the stream's (u, v)
  int nifOverlap = -1;             // "nif_overlap": NIF renders trace sample batch b + 1 beside the MLP of batch b (two slot sets, second stream): 1 | 0 | auto
                                   // (-1, the default: only beside nif_mlp_kernel, whose workgroups come and go - K3a / K3b hold every register of their
                                   // unit for the whole launch, nothing runs beside them, and the second slot set is memory better spent on longer launches)
  uint32_t nifTraceWgs = 0;        // "nif_trace_wgs": a NIF render's trace launch that runs beside the previous batch's MLP gets at most this many workgroups per
                                   // compute unit (0 = all that stay resident, the default: a cap of 1 paid 3 % while the trace launch queued behind its list
                                   // counter, and costs 0.7 % since it does not: profiles/r04_nif_overlap_ab.txt, r04_nif_trace_ab.txt)
  uint32_t nifSplit = 0;           // "nif_split": compute units a NIF render's trace launches of batches 1.. get for themselves (CU-masked streams: the MLP's
                                   // workgroups take every register of their unit, so a trace launch beside them otherwise only runs in their ramps and tails;
                                   // the MLP is power-limited and loses less than the units it gives up: DESIGN.md §6). 0 = no partition
  bool nifFirstTest = false;       // "nif_first_test": a NIF render's casts take their first box test in the turn that sets them up (trace_wavefront.hpp kFirstInSetup) instead of
                                   // in a NODE turn. Measured neutral (config 5: NODE turns 11.0 -> 3.4 per 64 casts, SHADE turns 1.2 -> 2.1 at lower occupancy, frame +-0.05 %,
                                   // profiles/r05_config5_launch_ab.txt): off by default, kept for A/B; results are the same either way
  bool nifTiming = false;          // "nif_timing": HIP events round every MLP launch of a NIF render (mi_get_nif_timing)
  uint32_t nifGenerations = kNifGenerations;   // MI_RAYLIB_NIF_GENERATIONS / "nif_generations": MLP workgroups launched per resident slot (nif_launch_mlp; measurement knob)
  bool rootStart = true;           // MI_RAYLIB_NO_ROOT_START / "root_start": a cast whose origin lies strictly inside the root's box starts at node 1
  bool sayGrid = false;            // MI_RAYLIB_SAY_GRID / "say_grid": print every persistent launch's grid to stderr (what the runtime said stays resident)
  // the two options that select ARITHMETIC (every other option leaves every result bit alone):
  bool doubleFallback = false;     // "double_fallback": the reference's ALLOW_DOUBLE_FALLBACK=1 build (CMakeLists.txt:13,34-41; Mesh.cpp:38-51), bit-exact to the oracle in that mode
  bool fast = false;               // "fast": the tolerance tier (FMA box / triangle tests; plain path-trace renders of the default kernel only)

  // Every key takes values from a stated domain; anything else leaves the option as it was and returns false
  // (mi_scene_set_option then reports MI_ERR_INVALID_ARG, as include/mi_raylib.h promises).
  static bool flag01(const char* v, bool& out) {
    if ((v[0] != '0' && v[0] != '1') || v[1] != '\0') return false;
    out = v[0] == '1';
    return true;
  }
  static bool number(const char* v, unsigned long long lo, unsigned long long hi, unsigned long long& out) {
    if (!*v) return false;
    char* end = nullptr;
    const unsigned long long q = strtoull(v, &end, 10);
    if (!end || *end != '\0' || v[0] == '-' || q < lo || q > hi) return false;
    out = q;
    return true;
  }
  bool set(const std::string& key, const char* v) {
    why = "unknown option or bad value";
    if (!v) return false;
    unsigned long long q = 0;
    // (an instrumented launch, a NIF launch and the double_fallback build have no tolerance-tier kernel: the combination is
    // refused where it is asked for instead of rendering another tier than the scene reports)
    if (key == "full_stats") {
      bool b = fullStats;
      if (!flag01(v, b)) return false;
      if (b && fast) { why = "full_stats cannot be combined with fast (the tolerance tier has no instrumented build)"; return false; }
      fullStats = b; return true;
    }
#if MI_RAYLIB_VARIANTS
    // (the tolerance tier is a build of the default kernel only: while it is set, nothing that selects another kernel is accepted -
    // the same refusal, whichever of the two options comes first)
    if (fast && (key == "kernel" || key == "waves" || key == "spec" || key == "merge" || key == "tune")) {
      SceneOptions probe = *this; probe.fast = false;
      if (!probe.set(key, v)) { why = probe.why; return false; }
      if (probe.kernelChoice != 1 || probe.specLeaf || probe.wavesPerSimd != 6 || !probe.mergeTurns || !(probe.tune == kDefaultTune)) {
        why = "fast is a build of the default kernel only (kernel 1, 6 waves, one SHADE / GEN turn, default weights, no spec): clear fast first"; return false;
      }
      return true;      // (a default value: nothing to change)
    }
    if (key == "kernel") { if (!number(v, 0, 3, q)) return false; kernelChoice = (int)q; return true; }
    if (key == "pool_waves") { if (!number(v, 4, 16, q) || (q != 4 && q != 8 && q != 16)) return false; poolWaves = (int)q; return true; }
    if (key == "pool_tune") {
      unsigned a, b, c, d, e, f, db = 4, mx = 5, ln = 1, pr = 1;
      if (sscanf(v, "%u,%u,%u,%u,%u,%u,%u,%u,%u,%u", &a, &b, &c, &d, &e, &f, &db, &mx, &ln, &pr) < 6) return false;
      poolTune = {a, b, c ? c : 1, d ? d : 1, e, f, db ? db : 65, mx, ln, pr};
      return true;
    }
    if (key == "waves") { if (!number(v, 4, 7, q)) return false; wavesPerSimd = (int)q; return true; }
    if (key == "spec") return flag01(v, specLeaf);
    if (key == "merge") return flag01(v, mergeTurns);
#else
    if (key == "kernel") {
      if (!number(v, 0, 3, q)) return false;
      if (q > 1) { why = "kernel 2 / 3 are not compiled into this library (the variants build, -DMI_RAYLIB_VARIANTS=1, has them)"; return false; }
      kernelChoice = (int)q; return true;
    }
    if (key == "pool_waves" || key == "pool_tune" || key == "tune") { why = "not compiled into this library (the variants build, -DMI_RAYLIB_VARIANTS=1, has it)"; return false; }
    if (key == "waves") { if (!number(v, 4, 7, q)) return false; if (q != 6) { why = "the 4-, 5- and 7-wave builds are not compiled into this library (variants build)"; return false; } return true; }
    if (key == "merge") { bool b = true; if (!flag01(v, b)) return false; if (!b) { why = "the two-turn form is not compiled into this library (variants build)"; return false; } return true; }
    if (key == "spec") { bool b = false; if (!flag01(v, b)) return false; if (b) { why = "the speculative walk is not compiled into this library (variants build)"; return false; } return true; }
#endif
    if (key == "cus") { if (!number(v, 0, 4096, q)) return false; cus = (uint32_t)q; return true; }
    if (key == "tiles") return flag01(v, tiles);
    if (key == "seg_budget_kb") { if (!number(v, 1, ~0ull >> 12, q)) return false; segBudgetKb = (size_t)q; return true; }
    if (key == "nif_spl") { if (!number(v, 0, kNifSplMax, q)) return false; nifSamplesPerLaunch = (uint32_t)q; return true; }
    if (key == "pin") return flag01(v, pin);
    if (key == "nif_timing") return flag01(v, nifTiming);
    if (key == "nif_generations") { if (!number(v, 1, 4096, q)) return false; nifGenerations = (uint32_t)q; return true; }
    if (key == "root_start") return flag01(v, rootStart);
    if (key == "say_grid") return flag01(v, sayGrid);
    if (key == "nif_overlap") { if (!strcmp(v, "auto")) { nifOverlap = -1; return true; } bool on = false; if (!flag01(v, on)) return false; nifOverlap = on ? 1 : 0; return true; }
    if (key == "nif_first_test") return flag01(v, nifFirstTest);
    if (key == "nif_split") { if (!number(v, 0, 1024, q)) return false; nifSplit = (uint32_t)q; return true; }
    if (key == "nif_trace_wgs") { if (!number(v, 0, 16, q)) return false; nifTraceWgs = (uint32_t)q; return true; }
    if (key == "coords") return flag01(v, coords);
    if (key == "lean_hit") return flag01(v, leanHit);
    if (key == "leaf_rot") return flag01(v, leafRot);
    if (key == "double_fallback") {
      bool b = doubleFallback;
      if (!flag01(v, b)) return false;
      if (b && fast) { why = "double_fallback cannot be combined with fast"; return false; }
      doubleFallback = b; return true;
    }
    if (key == "fast") {
      bool b = fast;
      if (!flag01(v, b)) return false;
      if (b && (fullStats || doubleFallback)) { why = "fast cannot be combined with full_stats or double_fallback"; return false; }
#if MI_RAYLIB_VARIANTS
      if (b && (kernelChoice != 1 || specLeaf || wavesPerSimd != 6 || !mergeTurns || !(tune == kDefaultTune))) { why = "fast is a build of the default kernel only (kernel 1, 6 waves, one SHADE / GEN turn, default weights, no spec)"; return false; }
#endif
      fast = b; return true;
    }
    if (key == "nif_shape") {
      const std::string s(v);
      if (s == "w6") nifShape = 0; else if (s == "t6") nifShape = 1; else if (s == "t4") nifShape = 2; else if (s == "a8") nifShape = 6; else if (s == "auto") nifShape = 7; else if (s == "b4") nifShape = 8;
#if MI_RAYLIB_VARIANTS
      else if (s == "r8") nifShape = 4; else if (s == "r8s") nifShape = 5;
#else
      else if (s == "r8" || s == "r8s") { why = "K3r (nif_shape r8 / r8s) is not compiled into this library (the variants build, -DMI_RAYLIB_VARIANTS=1, has it)"; return false; }
#endif
      else return false;
      return true;
    }
#if MI_RAYLIB_VARIANTS
    if (key == "tune") {      // leafAt,shadeAt,genAt[,burst,keep8,dbl,maxExtra,leafThenNode,prio,leafP,probe]
      unsigned a, b, c, dd = 48, k8 = 3, db = 4, mx = 6, ln = 1, pr = 1, lp = 40, pb = 0;
      if (sscanf(v, "%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u", &a, &b, &c, &dd, &k8, &db, &mx, &ln, &pr, &lp, &pb) < 3) return false;
      if (mx > 7 || pb > 3) return false;          // the spelled-out run of box tests has eight
      tune = {a, b, c, dd, k8, db ? db : 65, mx, ln, pr, lp ? lp : 1, pb};
      return true;
    }
#endif
    return false;
  }
  void fromEnvironment() {
    static const char* const map[][2] = {{"MI_RAYLIB_FULL_STATS", "full_stats"}, {"MI_RAYLIB_KERNEL", "kernel"}, {"MI_RAYLIB_WAVES", "waves"}, {"MI_RAYLIB_SPEC", "spec"},
                                         {"MI_RAYLIB_SEG_BUDGET_KB", "seg_budget_kb"}, {"MI_RAYLIB_NIF_SPL", "nif_spl"}, {"MI_RAYLIB_PIN", "pin"},
                                         {"MI_RAYLIB_NIF_SHAPE", "nif_shape"}, {"MI_RAYLIB_TUNE", "tune"},
                                         {"MI_RAYLIB_POOL_TUNE", "pool_tune"}, {"MI_RAYLIB_POOL_WAVES", "pool_waves"}, {"MI_RAYLIB_CUS", "cus"},
                                         {"MI_RAYLIB_NIF_OVERLAP", "nif_overlap"}, {"MI_RAYLIB_COORDS", "coords"},
                                         {"MI_RAYLIB_NIF_TRACE_WGS", "nif_trace_wgs"}, {"MI_RAYLIB_NIF_SPLIT", "nif_split"}, {"MI_RAYLIB_NIF_GENERATIONS", "nif_generations"}};
    // (an unparsable environment value is ignored: the option keeps its default. The two options that select ARITHMETIC,
    // double_fallback and fast, are deliberately not in this list: a process that says "bit-exact" must not change tier
    // because of a variable somebody exported)
    for (const auto& m : map) if (const char* e = getenv(m[0])) (void)set(m[1], e);
    if (getenv("MI_RAYLIB_NO_TILES")) tiles = false;
    if (getenv("MI_RAYLIB_NO_ROOT_START")) rootStart = false;
    if (getenv("MI_RAYLIB_SAY_GRID")) sayGrid = true;
  }
};

// What one in-flight path-trace launch owns besides the ray stream: its work counter and the partial-sum buffer of
// segmented pixels. One per HIP stream the scene has been rendered on, so launches of one scene that are enqueued on
// different streams (mi_render's two pipeline slots, or a caller's own streams through mi_render_device) never share
// them. Renders with a NIF environment additionally share the scene's slot scratch and are therefore chained with an
// event (nifDone): they may be enqueued on any streams but execute one after the other.
struct LaunchSlot {
  hipStream_t stream = nullptr;
  uint32_t* d_workCounter = nullptr;
  float* d_segPart = nullptr; size_t segPartFloats = 0;     // [segments][n][3]
  float2* d_coords = nullptr; size_t coordsCap = 0;         // the stream's pixel coordinates, compact (WaveExtras::coords)
  uint32_t* d_poolScratch = nullptr; size_t poolScratchWords = 0;   // kernel 3 (variants build): [PG_WORDS][slots of the grid]
  hipEvent_t lastWork = nullptr;                              // recorded behind everything the scene enqueued on `stream` (launchRender): what ~mi_scene waits for
};

struct mi_scene {
  int device = 0;
  int numCUs = 256;
  mi_scene_desc params{};            // scalar parameters only (pointers nulled)
  DeviceScene ds{};
  SceneOptions opt;
  std::vector<void*> allocations;
  unsigned long long* d_counters = nullptr;
  std::vector<LaunchSlot> slots;
  std::map<const void*, int> residentPerCU;      // workgroups of a kernel that stay resident on one compute unit (hipOccupancyMaxActiveBlocksPerMultiprocessor), asked once per kernel
  uint32_t cus() const { return opt.cus ? opt.cus : (uint32_t)numCUs; }
  // (auto: beside nif_mlp_kernel, and whenever the trace launches are given compute units of their own - option nif_split)
  bool nifOverlapOn() const { return opt.nifOverlap < 0 ? (opt.nifSplit > 0 || !((opt.nifShape >= 6 && opt.nifShape <= 8) && nif_asm_covers(nif.regs))) : opt.nifOverlap != 0; }
  // the scene as one launch sees it: option "root_start" decides whether the walk may start below the root
  DeviceScene view() const { DeviceScene v = ds; if (!opt.rootStart) v.rootInterior = 0; return v; }
  uint32_t residentBlocks(const void* kern, int threads, size_t ldsBytes) {
    auto it = residentPerCU.find(kern);
    if (it == residentPerCU.end()) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, ldsBytes) != hipSuccess || nb <= 0) { (void)hipGetLastError(); nb = 8; }
      it = residentPerCU.emplace(kern, nb).first;
    }
    return (uint32_t)it->second;
  }
  bool poolAttrSet[2][3] = {{false, false, false}, {false, false, false}};
  bool ldsAttrSet[2] = {false, false};   // kernel 2's dynamic-LDS opt-in (plain / instrumented build), per scene and so per device: function attributes are per device
  hipEvent_t nifDone = nullptr; bool nifPending = false;
  // mi_render's pipeline: two device batch buffers and two streams, kept between calls
  mi_trace_result* d_batch[2] = {nullptr, nullptr}; size_t batchCap[2] = {0, 0};
  hipStream_t pipeStream[2] = {nullptr, nullptr};
  double traceTimeSecs = 0.0;
  float hdriRotationDegrees = 0.f;
  size_t maxNifBatch = 0;
  size_t rayBatch = 0;               // rays per mi_render batch (0 = the whole stream in one batch)
  NifDevice nif;
  // scratch for the per-sample NIF loop
  // NIF renders: per-(sample, pixel) slots of one launch. TWO sets, so that the trace launch of sample batch b + 1 (into the
  // other set, on the render's stream) runs beside the MLP + accumulate pass of batch b (on nifAux): the MLP fills every CU
  // but its workgroups come and go in generations, and the traversal kernel takes the gaps (launch ramps and tails).
  struct NifSlots {
    float* u = nullptr; float* v = nullptr; float* bgr = nullptr; float* color = nullptr; float* tp = nullptr;
    uint32_t* index = nullptr; uint32_t* count = nullptr;
    hipEvent_t traced = nullptr, done = nullptr; bool donePending = false;
  } nifSlots[2];
  hipStream_t nifAux = nullptr;
  hipStream_t debugStream = nullptr;      // mi_debug_launch_progress
  // option "nif_split": the same pair of streams with compute-unit masks (the first numCUs - x units / the last x; a mask's bit i is
  // unit i / XCDs of XCD i % XCDs, so any multiple of the XCD count splits every XCD alike). splitUnits = x they were made for.
  hipStream_t nifSplitMlp = nullptr, nifSplitTrace = nullptr;
  uint32_t splitUnits = 0; bool splitRefused = false;
  hipEvent_t splitFirst = nullptr;
  Rng* d_rng = nullptr; size_t scratchRays = 0;
  float* d_segTotal = nullptr;                                // sample-at-a-time NIF renders: sum of the finished segments, [n][3]
  uint32_t scratchSamples = 0;                                // samples per launch the slot buffers are sized for
  uint32_t scratchAsked = 0;                                  // the MI_RAYLIB_NIF_SPL value they were sized under (0 = default)
  bool scratchOneSet = false;                                 // two slot sets did not fit the memory budget: this render runs without the overlap
  std::vector<std::pair<hipEvent_t, hipEvent_t>> nifTimes;    // option "nif_timing": events round the MLP launches since the last mi_get_nif_timing

  ~mi_scene() {
    (void)hipSetDevice(device);
    // Nothing of this scene's may still be in flight when its buffers go: mi_render_device is asynchronous and the caller may
    // destroy the scene right behind it. The wait is for THIS scene's work only - the event every launchRender records behind
    // what it enqueued on its stream (work on nifAux is always joined back into that stream before the event) - so destroying a
    // scene does not wait for other scenes' or replicas' launches on the device. (The events are the scene's own: the caller's
    // stream may be gone by now. hipFree below synchronises on its own account in this runtime; correctness does not rest on it.)
    for (LaunchSlot& l : slots) if (l.lastWork) (void)hipEventSynchronize(l.lastWork);
    if (nifAux) (void)hipStreamSynchronize(nifAux);
    for (hipStream_t q : {nifSplitMlp, nifSplitTrace, debugStream}) if (q) (void)hipStreamSynchronize(q);
    for (void* p : allocations) (void)hipFree(p);
    if (d_rng) (void)hipFree(d_rng);
    freeNifSlots();
    for (NifSlots& q : nifSlots) { if (q.count) (void)hipFree(q.count); if (q.traced) (void)hipEventDestroy(q.traced); if (q.done) (void)hipEventDestroy(q.done); }
    if (nifAux) (void)hipStreamDestroy(nifAux);
    for (hipStream_t q : {nifSplitMlp, nifSplitTrace, debugStream}) if (q) (void)hipStreamDestroy(q);
    if (splitFirst) (void)hipEventDestroy(splitFirst);
    if (d_segTotal) (void)hipFree(d_segTotal);
    for (LaunchSlot& l : slots) { if (l.lastWork) (void)hipEventDestroy(l.lastWork); if (l.d_workCounter) (void)hipFree(l.d_workCounter); if (l.d_segPart) (void)hipFree(l.d_segPart); if (l.d_coords) (void)hipFree(l.d_coords); if (l.d_poolScratch) (void)hipFree(l.d_poolScratch); }
    for (int i = 0; i < 2; ++i) { if (d_batch[i]) (void)hipFree(d_batch[i]); if (pipeStream[i]) (void)hipStreamDestroy(pipeStream[i]); }
    if (nifDone) (void)hipEventDestroy(nifDone);
    for (auto& e : nifTimes) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    nif.release();
  }
  void freeNifSlots() {
    for (NifSlots& q : nifSlots) {
      for (float** p : {&q.u, &q.v, &q.bgr, &q.color, &q.tp}) { if (*p) (void)hipFree(*p); *p = nullptr; }
      if (q.index) (void)hipFree(q.index);
      q.index = nullptr;
    }
  }
  template <class T> T* keep(T* p) { if (p) allocations.push_back((void*)p); return p; }
  // the launch slot of a stream (created on first use; a scene is thread-compatible, not thread-safe)
  LaunchSlot& slotFor(hipStream_t stream);
};

LaunchSlot& mi_scene::slotFor(hipStream_t stream) {
  for (LaunchSlot& l : slots) if (l.stream == stream) return l;
  LaunchSlot l;
  l.stream = stream;
  HIP_CHECK(hipMalloc(&l.d_workCounter, sizeof(uint32_t)));
  if (hipEventCreateWithFlags(&l.lastWork, hipEventDisableTiming) != hipSuccess) { (void)hipFree(l.d_workCounter); throw DeviceError("hipEventCreateWithFlags failed"); }
  slots.push_back(l);
  return slots.back();
}

namespace {

// Validate everything the kernels will index with, then build the device arrays.
void buildDeviceScene(mi_scene& S, const mi_scene_desc& d) {
  auto need = [](bool ok, const char* what) { if (!ok) throw ArgError(std::string("mi_scene_create: ") + what); };
  need(d.num_nodes == 0 || d.bvh_nodes, "bvh_nodes is null");
  need(d.num_nodes < (kLeafFlag >> 5), "more than 2^26 - 1 BVH nodes");
  need(d.num_geometry == 0 || d.geometry, "geometry is null");
  need(d.num_geometry <= 0xFFFF, "more than 65535 geometries (geomID is 16 bit)");
  need(d.num_mat_ids >= d.num_geometry, "All primitives must be assigned a material.");
  need(d.num_mat_ids == 0 || d.mat_ids, "mat_ids is null");
  need(d.num_materials == 0 || d.materials, "materials is null");
  need(d.num_meshes == 0 || (d.mesh_info && d.mesh_tris && d.mesh_verts), "mesh arrays are null");
  need(d.num_normals == 0 || (d.num_normals == d.num_verts && d.mesh_normals), "normals must be absent or one per vertex");
  need(d.num_spheres == 0 || d.spheres, "spheres is null");
  need(d.num_discs == 0 || d.discs, "discs is null");
  need(d.image_width > 0.f && d.image_height > 0.f, "image size must be positive");
  for (uint32_t g = 0; g < d.num_geometry; ++g) {
    const mi_geom_ref& r = d.geometry[g];
    need(r.type <= 2, "unknown geometry type");
    need(r.index < (r.type == 0 ? d.num_meshes : r.type == 1 ? d.num_spheres : d.num_discs), "geometry index out of range");
    need(d.mat_ids[g] < d.num_materials, "material index out of range");
  }
  for (uint32_t m = 0; m < d.num_meshes; ++m) {
    const mi_mesh_info& mi_ = d.mesh_info[m];
    need((uint64_t)mi_.first_index + mi_.num_triangles <= d.num_tris, "mesh triangle range out of bounds");
    need((uint64_t)mi_.first_vertex + mi_.num_vertices <= d.num_verts, "mesh vertex range out of bounds");
    for (uint32_t t = 0; t < 3 * mi_.num_triangles; ++t)
      need(d.mesh_tris[3 * (size_t)mi_.first_index + t] < mi_.num_vertices, "triangle vertex index out of range");
  }

  // Node array: box widened to six floats, link := next(i), the node that follows i's subtree in preorder (interior
  // nodes: instead of the second child's index; leaves: beside their primitive). Also checks that the array really is a
  // preorder BVH2 (first child adjacent, second child after it, every node reached).
  // (With `next` explicit the device array only has to keep "first child = i + 1". Laying the chains of first children
  // out by falling surface area puts 93 % of the box scene's node visits into the first 128 of its 8 069 nodes - in
  // preorder the first 1 024 take 38 % - and serving that prefix from LDS was measured: 4.5 % SLOWER than the L1 that
  // already holds it; the array stays in preorder.)
  const uint32_t N = d.num_nodes;
  std::vector<GNode> nodes(N);
  std::vector<uint32_t> skip(N);
  // leaves[] is indexed by NODE: a leaf node's primitive record sits at the node's own index (interior nodes leave a zero
  // record behind), so the walk never has to carry an index from the box test to the primitive test (trace_wavefront.hpp)
  std::vector<GLeaf> leaves(N);
  memset(leaves.data(), 0, leaves.size() * sizeof(GLeaf));
  std::vector<uint8_t> isLeafNode(N, 0);
  std::vector<uint32_t> geomFirstVertex(d.num_geometry, 0);
  for (uint32_t g = 0; g < d.num_geometry; ++g)
    if (d.geometry[g].type == 0) geomFirstVertex[g] = d.mesh_info[d.geometry[g].index].first_vertex;
  for (uint32_t i = N; i-- > 0;) {
    const mi_bvh_node& n = d.bvh_nodes[i];
    if (n.geom_id != MI_INVALID_GEOM) skip[i] = i + 1;
    else {
      const uint32_t second = n.prim_or_second_child;
      need(i + 1 < N && second > i + 1 && second < N, "BVH is not a depth-first BVH2 (bad second child index)");
      need(skip[i + 1] == second, "BVH is not in depth-first order (first child's subtree must end at the second child)");
      skip[i] = skip[second];
    }
  }
  need(N == 0 || skip[0] == N, "BVH root does not span the node array");
  for (uint32_t i = 0; i < N; ++i) {
    const mi_bvh_node& n = d.bvh_nodes[i];
    // the box test's fast form assumes finite slab products (trace_wavefront.hpp): a NaN / inf box is a malformed scene
    need(std::isfinite(n.min_x) && std::isfinite(n.min_y) && std::isfinite(n.min_z), "BVH node bounds are not finite");
    need((n.dx & 0x7C00u) != 0x7C00u && (n.dy & 0x7C00u) != 0x7C00u && (n.dz & 0x7C00u) != 0x7C00u, "BVH node extents are not finite");
    GNode g;
    g.minx = n.min_x; g.miny = n.min_y; g.minz = n.min_z;
    g.maxx = n.min_x + half_bits_to_float(n.dx);                  // CompactBVH2Node.cpp:8-10, one rounded add each
    g.maxy = n.min_y + half_bits_to_float(n.dy);
    g.maxz = n.min_z + half_bits_to_float(n.dz);
    g.link = skip[i] << 5;                                         // (byte offsets: trace_kernels.hpp GNode)
    g.hit = (i + 1) << 5;                                          // interior: the first child
    if (n.geom_id != MI_INVALID_GEOM) {
      need(n.geom_id < d.num_geometry, "leaf geomID out of range");
      const mi_geom_ref& r = d.geometry[n.geom_id];
      GLeaf L;
      memset(&L, 0, sizeof L);
      if (r.type == 0) {
        const mi_mesh_info& mi_ = d.mesh_info[r.index];
        need(n.prim_or_second_child < mi_.num_triangles, "leaf primID out of range");
        const size_t base = 3 * ((size_t)mi_.first_index + n.prim_or_second_child);
        for (int k = 0; k < 3; ++k) {
          const mi_vec3& p = d.mesh_verts[mi_.first_vertex + d.mesh_tris[base + k]];
          L.f[3 * k] = p.x; L.f[3 * k + 1] = p.y; L.f[3 * k + 2] = p.z;
        }
        L.type = LEAF_TRI | ((uint32_t)n.geom_id << 16); L.primID = n.prim_or_second_child; L.triBase = (uint32_t)base;
        const f3 p0 = mk(L.f[0], L.f[1], L.f[2]), p1 = mk(L.f[3], L.f[4], L.f[5]), p2 = mk(L.f[6], L.f[7], L.f[8]);
        const f3 fn = normalized(cross(p1 - p0, p2 - p0));                        // Mesh.hpp:112-114
        L.n[0] = fn.x; L.n[1] = fn.y; L.n[2] = fn.z;
      } else if (r.type == 1) {
        const mi_sphere& s = d.spheres[r.index];
        L.f[0] = s.x; L.f[1] = s.y; L.f[2] = s.z; L.f[3] = s.radius; L.f[4] = s.radius * s.radius;   // Primitives.hpp:44
        L.type = LEAF_SPHERE | ((uint32_t)n.geom_id << 16); L.primID = 0;                                                           // Primitives.cpp:45
      } else {
        const mi_disc& c = d.discs[r.index];
        L.f[0] = c.nx; L.f[1] = c.ny; L.f[2] = c.nz; L.f[3] = c.cx; L.f[4] = c.cy; L.f[5] = c.cz; L.f[6] = c.r * c.r;
        L.type = LEAF_DISC | ((uint32_t)n.geom_id << 16); L.primID = 0;
      }
      L.matIndex = d.mat_ids[n.geom_id];
      g.hit = (skip[i] << 5) | kLeafFlag;                          // leaf: stop in front of link (= i + 1)
      leaves[i] = L;
      isLeafNode[i] = 1;
    }
    nodes[i] = g;
  }

  DeviceScene& ds = S.ds;
  ds.nodes = S.keep(upload(nodes)); ds.numNodes = N;
  ds.leaves = S.keep(upload(leaves)); ds.numLeaves = (uint32_t)leaves.size();
  {
    // the primitive-test part of every record once per shear axis (GLeafRot, trace_kernels.hpp): a triangle's vertices with their
    // components rotated so that component kz comes last - (kx, ky, kz) = (kz + 1, kz + 2, kz) mod 3, permute_kz - other records as they are
    std::vector<GLeafRot> rot(leaves.size());
    memset(rot.data(), 0, rot.size() * sizeof(GLeafRot));
    for (size_t i = 0; i < leaves.size(); ++i)
      for (uint32_t kz = 0; kz < 3; ++kz) {
        GLeafBlock& B = rot[i].b[kz];
        B.type = leaves[i].type;
        for (int q = 0; q < 9; ++q) B.f[q] = leaves[i].f[q];
        if (leaf_kind(leaves[i]) == LEAF_TRI) {
          const uint32_t kx = (kz + 1) % 3, ky = (kz + 2) % 3;
          for (int v = 0; v < 3; ++v) { B.f[3 * v] = leaves[i].f[3 * v + kx]; B.f[3 * v + 1] = leaves[i].f[3 * v + ky]; B.f[3 * v + 2] = leaves[i].f[3 * v + kz]; }
        }
      }
    ds.leavesRot = S.keep(upload(rot));
  }
  ds.matIDs = S.keep(upload(std::vector<uint32_t>(d.mat_ids, d.mat_ids + d.num_mat_ids)));
  ds.materials = S.keep(upload(std::vector<mi_material>(d.materials, d.materials + d.num_materials)));
  ds.numMaterials = d.num_materials;
  ds.hasNormals = d.num_normals ? 1u : 0u;
  ds.leafNormals = nullptr;
  if (ds.hasNormals) {
    std::vector<float> ln(9 * leaves.size(), 0.f);
    for (size_t k = 0; k < leaves.size(); ++k) {
      const GLeaf& L = leaves[k];
      if (!isLeafNode[k] || leaf_kind(L) != LEAF_TRI) continue;
      const uint32_t fv = geomFirstVertex[leaf_geom(L)];
      for (int c = 0; c < 3; ++c) {
        const size_t at = (size_t)fv + d.mesh_tris[L.triBase + c];
        need(at < d.num_normals, "vertex normal index out of range");
        const mi_vec3& nn = d.mesh_normals[at];
        ln[9 * k + 3 * c] = nn.x; ln[9 * k + 3 * c + 1] = nn.y; ln[9 * k + 3 * c + 2] = nn.z;
      }
    }
    ds.leafNormals = S.keep(upload(ln));
    ds.meshTris = S.keep(upload(std::vector<uint16_t>(d.mesh_tris, d.mesh_tris + 3 * (size_t)d.num_tris)));
    ds.meshNormals = S.keep(upload(std::vector<mi_vec3>(d.mesh_normals, d.mesh_normals + d.num_normals)));
    ds.geomFirstVertex = S.keep(upload(geomFirstVertex));
  }
  ds.rootInterior = 0;
  if (N > 1) {      // (N > 1: the root is an interior node - checked above: a leaf root spans one node; option "root_start" = 0 clears the flag per launch)
    ds.rootLoX = nodes[0].minx; ds.rootHiX = nodes[0].maxx; ds.rootLoY = nodes[0].miny; ds.rootHiY = nodes[0].maxy;
    ds.rootLoZ = nodes[0].minz; ds.rootHiZ = nodes[0].maxz; ds.rootInterior = node_is_leaf(nodes[0]) ? 0u : 1u;
  }
  ds.imageWidth = d.image_width; ds.imageHeight = d.image_height;
  float s, c;
  sincos_deg_table(d.fov_radians / 2.f, hostSinTable(), s, c);   // codelets/TraceCodelets.cpp:147-149
  ds.tanTheta = s / c;
  ds.antiAliasScale = d.anti_alias_scale;
  ds.maxPathLength = d.max_path_length; ds.rouletteStartDepth = d.roulette_start_depth;
  ds.samplesPerPixel = d.samples_per_pixel;
  ds.rngSeed = d.rng_seed;
  HIP_CHECK(hipMalloc(&S.d_counters, 32 * sizeof(unsigned long long)));
  S.keep(S.d_counters);
  HIP_CHECK(hipMemset(S.d_counters, 0, 32 * sizeof(unsigned long long)));
  ds.counters = S.d_counters;
}

// Slots per pixel per launch in NIF renders: 48 B each (u, v, bgr, colour, throughput, list entry), and TWO sets of them when
// the render has more than one sample batch and the overlap is on (nif_overlap). A launch holds whole segments (ray_math.h
// segment_samples); more samples per launch mean more (pixel, segment) atoms per lane and fewer launch tails: 128 by
// default, fewer when n x samples x 48 B x sets would pass the budget - 32 GiB or half of what the device has free (plus
// what this scene already holds), whichever is less (the 1440^2 frame at 128 samples and two sets takes 25.5 GB) -, never
// less than one segment; if two sets of one segment do not fit, the render runs with one (no overlap). Option "nif_spl"
// overrides the sample count (1..128, rounded up to whole segments). None of this changes a result bit (the accumulate
// pass replays the reference's order whatever the batching: test_nif_render_sample_batching_is_order_exact).
// slot-mode (NIF) launches: at most this many workgroups, each wave of which may leave kEnvChunk - 1 padded entries in the escaped-slot list
constexpr uint32_t kMaxSlotWorkgroups = 4096;

void ensureScratch(mi_scene& S, size_t n) {
  {
    const uint32_t asked = S.opt.nifSamplesPerLaunch;
    const uint32_t segLen = segment_samples(S.ds.samplesPerPixel);
    uint32_t v = (asked >= 1 && asked <= kNifSplMax) ? asked : kNifSplDefault;
    v = ((v + segLen - 1) / segLen) * segLen;
    v = std::min(v, std::max(segLen, ((S.ds.samplesPerPixel + segLen - 1) / segLen) * segLen));      // no more than the render has
    const bool haveTwo = S.nifSlots[1].u != nullptr;
    auto setsFor = [&](uint32_t vv) { return (S.nifOverlapOn() && S.ds.samplesPerPixel > vv) ? 2u : 1u; };
    constexpr uint64_t kSlotBytes = 48;
    uint64_t budget = (uint64_t)64 << 30;
    {
      size_t freeB = 0, totalB = 0;
      if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
        const uint64_t held = (uint64_t)S.scratchRays * S.scratchSamples * kSlotBytes * (haveTwo ? 2u : 1u);
        budget = std::min<uint64_t>(budget, ((uint64_t)freeB + held) / 2);
      } else (void)hipGetLastError();
    }
    if (!(asked >= 1 && asked <= kNifSplMax))
      while (v > segLen && (uint64_t)n * v * kSlotBytes * setsFor(v) > budget) v -= segLen;
    const bool oneSetOnly = setsFor(v) == 2u && (uint64_t)n * v * kSlotBytes * 2u > budget;      // two sets of the smallest launch do not fit
    // keep what is there when it still fits this stream and the request has not changed
    if (S.scratchRays >= n && S.scratchAsked == asked && S.scratchSamples >= segLen && S.scratchSamples % segLen == 0 &&
        (haveTwo || oneSetOnly || !(S.nifOverlapOn() && S.ds.samplesPerPixel > S.scratchSamples))) return;
    S.scratchAsked = asked;
    S.scratchOneSet = oneSetOnly;
    if (v != S.scratchSamples || (!haveTwo && !oneSetOnly && S.nifOverlapOn() && S.ds.samplesPerPixel > v)) { S.scratchSamples = v; S.scratchRays = 0; }     // slot buffers are sized for n x v (x two sets)
  }
  if (S.scratchRays >= n) return;
  if (S.nifPending) { HIP_CHECK(hipDeviceSynchronize()); S.nifPending = false; }      // an earlier NIF render may still read the old buffers
  if (S.d_rng) (void)hipFree(S.d_rng);
  if (S.d_segTotal) (void)hipFree(S.d_segTotal);
  S.freeNifSlots();
  S.d_rng = nullptr; S.d_segTotal = nullptr; S.scratchRays = 0;
  const size_t slots = n * S.scratchSamples;
  HIP_CHECK(hipMalloc(&S.d_rng, n * sizeof(Rng)));
  HIP_CHECK(hipMalloc(&S.d_segTotal, 3 * n * sizeof(float)));
  // the second set only where it is used: renders of more than one sample batch with the overlap on
  const int sets = (S.nifOverlapOn() && S.ds.samplesPerPixel > S.scratchSamples && !S.scratchOneSet) ? 2 : 1;
  for (int k = 0; k < sets; ++k) {
    mi_scene::NifSlots& q = S.nifSlots[k];
    HIP_CHECK(hipMalloc(&q.u, slots * sizeof(float)));
    HIP_CHECK(hipMalloc(&q.v, slots * sizeof(float)));
    HIP_CHECK(hipMalloc(&q.bgr, 3 * slots * sizeof(float)));
    HIP_CHECK(hipMalloc(&q.color, 3 * slots * sizeof(float)));
    HIP_CHECK(hipMalloc(&q.tp, 3 * slots * sizeof(float)));
    // (every wave of a slot-mode launch pads the list to a whole chunk: trace_wavefront.hpp pushEscaped, kMaxSlotWorkgroups)
    HIP_CHECK(hipMalloc(&q.index, (slots + (size_t)kMaxSlotWorkgroups * 4u * kEnvChunk) * sizeof(uint32_t)));
    if (!q.count) HIP_CHECK(hipMalloc(&q.count, sizeof(uint32_t)));
    if (!q.traced) HIP_CHECK(hipEventCreateWithFlags(&q.traced, hipEventDisableTiming));
    if (!q.done) HIP_CHECK(hipEventCreateWithFlags(&q.done, hipEventDisableTiming));
  }
  S.scratchRays = n;
}

[[maybe_unused]] constexpr uint32_t kLdsBudgetBytes = 160 * 1024 - 2048 - 23 * 1024 * 4;     // 160 KiB per CU minus the static allocations (sin table, material cache, 23 cold-state words x 1024 threads)

// Work indices are 32 bit and every wave takes them from the launch's counter in chunks of fetchChunk (64): a wave
// that finds the counter past the end still adds its chunk, so the counter may overshoot the item count by
// (waves of the grid) x fetchChunk. Launches stay below 2^32 minus that headroom (2048 workgroups x 16 waves x 64,
// rounded up), so the counter never wraps.
constexpr uint64_t kMaxWorkItems = 0xFFFFFFFFull - ((uint64_t)1 << 22);

template <bool STATS>
void launchWavefront(mi_scene& S, mi_trace_result* d_rays, uint32_t cnt, hipStream_t stream, const WaveExtras& ex = WaveExtras{}, uint32_t wgCap = 0, uint32_t unitsHere = 0) {
  LaunchSlot& slot = S.slotFor(stream);
  const uint32_t units = unitsHere ? unitsHere : S.cus();      // (a CU-masked stream: the units of its mask)
  uint32_t* workCounter = slot.d_workCounter;
  const DeviceScene dsv = S.view();
  // Streams are walked in 8x8 pixel tiles of window-width rows (a whole window, a batch of it, or one rank's
  // 8-row bands are all sequences of full rows); the walk is only a work ORDER, any stream stays correct.
  const uint32_t w = (uint32_t)S.params.window_w;
  const uint32_t tileW = (S.opt.tiles && w >= 8 && (w % 8) == 0 && cnt >= 8u * w) ? w : 0u;
  const bool plain = ex.slotColor == nullptr;      // NIF launches leave slots instead of partial sums (kernels 1 and 3 take them; kernel 2 falls through to kernel 1)
  // Pixels with more than segment_samples(spp) samples are traced as (pixel, segment) work atoms (ray_math.h); every
  // stream has its own partial-sum buffer (LaunchSlot).
  const uint32_t segLen = segment_samples(S.ds.samplesPerPixel);
  const uint32_t segments = (S.ds.samplesPerPixel + segLen - 1) / segLen;
  const bool segmented = plain && segments > 1;
  // One launch covers as many segments of every pixel as the partial-sum budget holds (8 GiB, option "seg_budget_kb";
  // the bench frame needs 0.4 GB); longer renders run as several launches whose combine passes continue
  // the running sum in segment order, so the cut never shows in the result.
  uint32_t perLaunch = segments;
  if (segmented) {
    const size_t budgetFloats = S.opt.segBudgetKb * (size_t)(1024 / sizeof(float));
    const uint64_t byBudget = std::max<uint64_t>(1, budgetFloats / ((uint64_t)3 * cnt));
    const uint64_t byIndex = std::max<uint64_t>(1, kMaxWorkItems / cnt);           // work indices are 32-bit
    perLaunch = (uint32_t)std::min<uint64_t>(segments, std::min(byBudget, byIndex));
    const size_t need = (size_t)3 * cnt * perLaunch;
    if (slot.segPartFloats < need) {
      if (slot.d_segPart) { HIP_CHECK(hipStreamSynchronize(stream)); (void)hipFree(slot.d_segPart); }
      slot.d_segPart = nullptr; slot.segPartFloats = 0;
      HIP_CHECK(hipMalloc(&slot.d_segPart, need * sizeof(float)));
      slot.segPartFloats = need;
    }
  }
  const uint32_t launches = segmented ? (segments + perLaunch - 1) / perLaunch : 1u;
  // (pixel, segment) atoms fetch a pixel's coordinates once per segment: from a compact copy, gathered here
  const float2* coords = nullptr;
  if ((segmented || !plain) && S.opt.coords) {
    if (slot.coordsCap < cnt) {
      if (slot.d_coords) { HIP_CHECK(hipStreamSynchronize(stream)); (void)hipFree(slot.d_coords); }
      slot.d_coords = nullptr; slot.coordsCap = 0;
      HIP_CHECK(hipMalloc(&slot.d_coords, (size_t)cnt * sizeof(float2)));
      slot.coordsCap = cnt;
    }
    hipLaunchKernelGGL(pixel_coords_kernel, dim3((cnt + 255) / 256), dim3(256), 0, stream, d_rays, cnt, slot.d_coords);
    coords = slot.d_coords;
  }
  for (uint32_t l = 0; l < launches; ++l) {
    const uint32_t segBase = l * perLaunch;
    WaveExtras exs = ex;
    exs.coords = coords;
    if (segmented) { exs.segPart = slot.d_segPart; exs.segments = std::min(perLaunch, segments - segBase); exs.segBase = segBase; }
    HIP_CHECK(hipMemsetAsync(workCounter, 0, sizeof(uint32_t), stream));
    const uint64_t items = (uint64_t)cnt * ((segmented || !plain) ? exs.segments : 1u);     // work atoms of this launch
    if (items > kMaxWorkItems) throw ArgError("mi_render: too many work items for one launch (cut the stream with mi_scene_set_ray_batch)");
    // Grid: persistent workgroups, as many as stay resident (compute units x workgroups per unit, asked of the runtime for
    // the kernel about to be launched and remembered per scene), never more than the launch has work for.
    auto grid = [&](auto kern, uint32_t threads, size_t ldsBytes) {
      uint32_t perUnit = S.residentBlocks(reinterpret_cast<const void*>(kern), (int)threads, ldsBytes);
      if (wgCap) perUnit = std::min(perUnit, wgCap);      // (NIF renders: a trace launch that runs beside the previous batch's MLP)
      uint64_t wgs = std::min<uint64_t>((items + threads - 1) / threads, (uint64_t)units * perUnit);
      if (!plain) wgs = std::min<uint64_t>(wgs, kMaxSlotWorkgroups);      // (the escaped-slot list has padding for that many)
      return (uint32_t)wgs;
    };
    auto go = [&](auto kern) {
      if (S.opt.sayGrid) fprintf(stderr, "mi_raylib: grid %u workgroups = %u units x %u resident\n", grid(kern, 256, 0), units, S.residentBlocks(reinterpret_cast<const void*>(kern), 256, 0));
      hipLaunchKernelGGL(kern, dim3(grid(kern, 256, 0)), dim3(256), 0, stream, dsv, d_rays, cnt, workCounter, 0u, S.opt.tune, tileW, exs);
    };
    if (S.opt.doubleFallback) {
      // the ALLOW_DOUBLE_FALLBACK=1 variant: the phase-scheduled kernel's 4-wave build with the binary64 edge functions compiled in
      go(path_trace_wavefront_kernel<STATS, false, 256, 4, false, 2, false, true>);
    } else if (S.opt.fast) {
      // the tolerance tier (never the default): FMA box and triangle tests. Plain, un-instrumented launches of the default kernel only;
      // mi_scene_set_option refuses the combinations it has no build for, a NIF render is refused here.
      if (!plain || STATS) throw ArgError("mi_render: the tolerance tier (option fast) has no NIF / instrumented build; clear the option for this render");
      go(path_trace_wavefront_kernel<false, false, 256, 6, false, 0, true, false, true>);
#if MI_RAYLIB_VARIANTS
    } else if (S.opt.kernelChoice == 3 && S.ds.samplesPerPixel <= kPoolMaxSamples && S.ds.maxPathLength <= kPoolMaxBounces) {
      // path pool (trace_pool.hpp): persistent, exactly as many workgroups as stay resident; a workgroup of W waves
      // owns 100 W path slots
      const uint32_t W = (uint32_t)S.opt.poolWaves, pwg = 100u * W;
      const uint32_t blocks = (uint32_t)std::min<uint64_t>((items + pwg - 1) / pwg, (uint64_t)S.cus() * (16u / W));
      const uint32_t stride = blocks * pwg;
      const size_t need = (size_t)PG_WORDS * stride;
      if (slot.poolScratchWords < need) {
        if (slot.d_poolScratch) { HIP_CHECK(hipStreamSynchronize(stream)); (void)hipFree(slot.d_poolScratch); }
        slot.d_poolScratch = nullptr; slot.poolScratchWords = 0;
        HIP_CHECK(hipMalloc(&slot.d_poolScratch, need * sizeof(uint32_t)));
        slot.poolScratchWords = need;
      }
      const size_t ldsBytes = pool_lds_bytes(pwg);
      auto goPool = [&](auto kern, int which) {
        if (!S.poolAttrSet[STATS ? 1 : 0][which]) {
          HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
          S.poolAttrSet[STATS ? 1 : 0][which] = true;
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * W), ldsBytes, stream, S.view(), d_rays, cnt, workCounter, S.opt.poolTune, tileW, exs, slot.d_poolScratch, stride);
      };
      if (W == 4) goPool(path_trace_pool_kernel<STATS, 4, 400, 4>, 0);
      else if (W == 8) goPool(path_trace_pool_kernel<STATS, 8, 800, 4>, 1);
      else goPool(path_trace_pool_kernel<STATS, 16, 1600, 4>, 2);
    } else if (plain && S.opt.kernelChoice == 2 && S.ds.numNodes > 0) {
      // one 1024-thread workgroup per CU shares one LDS copy of the first nodes of the (preorder) array
      const uint32_t ldsNodes = std::min<uint32_t>(S.ds.numNodes, kLdsBudgetBytes / (uint32_t)sizeof(GNode));
      const size_t ldsBytes = (size_t)ldsNodes * sizeof(GNode);
      auto kern = path_trace_wavefront_kernel<STATS, true, 1024>;
      if (!S.ldsAttrSet[STATS ? 1 : 0]) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudgetBytes));
        S.ldsAttrSet[STATS ? 1 : 0] = true;
      }
      const uint32_t blocks = (uint32_t)std::min<uint64_t>((items + 1023) / 1024, S.cus());
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), ldsBytes, stream, S.view(), d_rays, cnt, workCounter, ldsNodes, S.opt.tune, tileW, exs);
    } else if (S.opt.specLeaf) {
      if (!STATS) go(path_trace_wavefront_kernel<false, false, 256, 5, true>);
      else go(path_trace_wavefront_kernel<STATS, false, 256, 4, true>);
    } else if (!STATS && !(S.opt.tune == kDefaultTune)) {
      // runtime weights (tune sweeps, knock-in probes): scalar spills, a few per cent slower than the compiled-in weights
      if (S.opt.wavesPerSimd == 6) go(path_trace_wavefront_kernel<false, false, 256, 6>);
      else go(path_trace_wavefront_kernel<false, false, 256, 5>);
    } else if (!STATS && !S.opt.mergeTurns) {
      // SHADE and GEN as two turns, five waves per SIMD: the default kernel up to round 3, kept for A/B and parity
      if (plain) go(path_trace_wavefront_kernel<false, false, 256, 5, false, 0, true, false, false, false>);
      else go(path_trace_wavefront_kernel<false, false, 256, 5, false, 1, true, false, false, false>);
    } else if (!STATS && S.opt.wavesPerSimd == 7 && plain) {
      go(path_trace_wavefront_kernel<false, false, 256, 7, false, 0, true>);                   // 72 VGPRs, 21 words of LDS per lane: seven waves per SIMD
    } else if (!STATS && S.opt.wavesPerSimd == 5) {
      if (plain) go(path_trace_wavefront_kernel<false, false, 256, 5, false, 0, true>);        // the 96-VGPR build of today's form
      else go(path_trace_wavefront_kernel<false, false, 256, 5, false, 1, true>);
#endif
    } else if (!STATS && S.opt.wavesPerSimd == 6) {
      // the default path: SHADE and GEN in one turn, 80 VGPRs without a spill, six waves per SIMD (profiles/r04_k1w_merged_turn_ab.txt);
      // the default scheduling weights are compiled into the two instantiations (plain renders / NIF slots)
      // the builds of the default form: a scene WITHOUT vertex normals takes the one that carries no barycentrics (option "lean_hit")
      // and reads the primitive records pre-rotated for the cast's shear axis (option "leaf_rot"); both default on, both exact (every
      // byte the oracle's either way), the other settings kept for A/B. With vertex normals the pre-rotated form spills seven
      // registers at 80 and loses 8 % (test_scene.dae: 292 against 271 ms per 1000 spp, profiles/r05_k1w_leaf_ab.txt): not built.
      const bool lean = S.opt.leanHit && !S.ds.hasNormals;
      const bool rot = lean && S.opt.leafRot && S.ds.leavesRot != nullptr;
#define MI_K1W(SL, BA, RO) path_trace_wavefront_kernel<false, false, 256, 6, false, SL, true, false, false, true, BA, RO>
      if (plain) { if (rot) go(MI_K1W(0, false, true)); else if (lean) go(MI_K1W(0, false, false)); else go(MI_K1W(0, true, false)); }
      else { if (rot) go(MI_K1W(1, false, true)); else if (lean) go(MI_K1W(1, false, false)); else go(MI_K1W(1, true, false)); }
#undef MI_K1W
    } else {
      go(path_trace_wavefront_kernel<STATS, false, 256>);               // the instrumented build (and, in the variants build, option waves = 4)
    }
    if (segmented) hipLaunchKernelGGL(segment_combine_kernel, dim3((cnt + 255) / 256), dim3(256), 0, stream, d_rays, cnt, exs.segments, slot.d_segPart, segBase ? 1u : 0u);
  }
}

// Option "nif_split": the two CU-masked streams of a NIF render (made once per value; 0 = none, also when the runtime refuses
// masked streams - said once on stderr: the render then shares the chip between its launches as without the option).
// Returns the units the trace stream owns.
uint32_t splitStreams(mi_scene& S) {
  const uint32_t N = (uint32_t)S.numCUs;
  uint32_t x = S.opt.nifSplit;
  if (x == 0 || S.splitRefused) return 0;
  if (x > N / 2) x = N / 2;
  if (S.splitUnits == x) return x;
  for (hipStream_t* q : {&S.nifSplitMlp, &S.nifSplitTrace}) if (*q) { (void)hipStreamSynchronize(*q); (void)hipStreamDestroy(*q); *q = nullptr; }
  S.splitUnits = 0;
  std::vector<uint32_t> lo((N + 31) / 32, 0u), hi((N + 31) / 32, 0u);
  for (uint32_t i = 0; i < N; ++i) (i < N - x ? lo : hi)[i / 32] |= 1u << (i % 32);
  const hipError_t e0 = hipExtStreamCreateWithCUMask(&S.nifSplitMlp, (uint32_t)lo.size(), lo.data());
  const hipError_t e1 = e0 == hipSuccess ? hipExtStreamCreateWithCUMask(&S.nifSplitTrace, (uint32_t)hi.size(), hi.data()) : e0;
  if (e1 != hipSuccess) {
    (void)hipGetLastError();
    if (S.nifSplitMlp) { (void)hipStreamDestroy(S.nifSplitMlp); S.nifSplitMlp = nullptr; }
    S.nifSplitTrace = nullptr; S.splitRefused = true;
    fprintf(stderr, "mi_raylib: option nif_split: no compute-unit masked streams here (%s); the render's launches share the chip\n", hipGetErrorString(e1));
    return 0;
  }
  if (!S.splitFirst) HIP_CHECK(hipEventCreateWithFlags(&S.splitFirst, hipEventDisableTiming));
  S.splitUnits = x;
  return x;
}

void launchRender(mi_scene& S, int mode, mi_trace_result* d_rays, size_t n, hipStream_t stream) {
  if (n == 0) return;
  if (n > 0xFFFFFFFFull) throw ArgError("mi_render: more than 2^32-1 rays in one call");
  const uint32_t cnt = (uint32_t)n;
  const dim3 block(256), grid((cnt + 255) / 256);
  if (mode == MI_MODE_SHADOW_TRACE) {
    const f3 light = mk(18.f, 257.f, -1060.f);          // trace.cpp:247, src/IpuScene.cpp:447
    if (S.opt.doubleFallback) hipLaunchKernelGGL((shadow_trace_kernel<false, true>), grid, block, 0, stream, S.view(), d_rays, cnt, .05f, light);
    else if (S.opt.fullStats) hipLaunchKernelGGL(shadow_trace_kernel<true>, grid, block, 0, stream, S.view(), d_rays, cnt, .05f, light);
    else hipLaunchKernelGGL(shadow_trace_kernel<false>, grid, block, 0, stream, S.view(), d_rays, cnt, .05f, light);
  } else if (mode == MI_MODE_PATH_TRACE) {
    // (the ALLOW_DOUBLE_FALLBACK=1 variant is compiled into the phase-scheduled kernel and the shadow kernel: it always takes them)
    // (the phase-scheduled kernel keeps a path's bounce count in 30 bits)
    const bool nested = (S.opt.kernelChoice == 0 || S.ds.maxPathLength >= (1u << 30)) && !S.opt.doubleFallback;
    if (!S.nif.loaded() && !nested && S.ds.samplesPerPixel >= 1 && S.ds.maxPathLength >= 1) {
      // sample loop inside the kernel (src/IpuScene.cpp:441), phase-scheduled persistent form
      if (S.opt.fullStats) launchWavefront<true>(S, d_rays, cnt, stream);
      else launchWavefront<false>(S, d_rays, cnt, stream);
    } else if (!S.nif.loaded()) {
      // sample loop inside the kernel (src/IpuScene.cpp:441: vertexSampleCount = samplesPerPixel)
      if (S.opt.fullStats) hipLaunchKernelGGL(path_trace_kernel<true>, grid, block, 0, stream, S.view(), d_rays, cnt, 0u, S.ds.samplesPerPixel, (Rng*)nullptr);
      else hipLaunchKernelGGL(path_trace_kernel<false>, grid, block, 0, stream, S.view(), d_rays, cnt, 0u, S.ds.samplesPerPixel, (Rng*)nullptr);
    } else {
      // Repeat(spp){ trace 1 sample; uv pre-pass; NIF; env post-pass }  (src/IpuScene.cpp:571-583)
      // NIF renders of one scene share its slot scratch: whatever streams they are enqueued on, each waits for the
      // previous one (and a scratch re-allocation waits for the device)
      if (!S.nifDone) HIP_CHECK(hipEventCreateWithFlags(&S.nifDone, hipEventDisableTiming));
      if (S.nifPending) HIP_CHECK(hipStreamWaitEvent(stream, S.nifDone, 0));
      ensureScratch(S, n);
      const float radians = (S.hdriRotationDegrees / 360.f) * (float)(2.0 * M_PI);   // src/IpuScene.cpp:644
      const bool wave = !nested && S.ds.maxPathLength >= 1;
      if (wave) {
        // persistent phase-scheduled kernel, several samples per launch; every path leaves a slot (WaveExtras), the
        // MLP runs on the compacted escaped slots, and a per-pixel pass adds everything in the reference's order
        if ((uint64_t)cnt * S.scratchSamples > kMaxWorkItems) throw ArgError("mi_render: ray batch too large for a NIF render (cut it with mi_scene_set_ray_batch)");
        const uint32_t segLen = segment_samples(S.ds.samplesPerPixel), segShift = segment_shift(S.ds.samplesPerPixel);
        // Sample batch b: trace its slots on `stream` (set b & 1), then MLP over the compacted escaped slots + the accumulate
        // pass - on nifAux when there are two sets, so that the trace launch of batch b + 1 runs beside them. The accumulate
        // passes run in batch order on one stream and touch only rgb; the trace launch only reads the pixel coordinates and
        // writes the hit record of the same TraceResults (other dwords), so the two never meet. A set is traced into again
        // only when the MLP + accumulate that read it are done; the render's stream ends behind the last of them.
        const bool two = S.nifOverlapOn() && S.nifSlots[1].u != nullptr;
        if (two && !S.nifAux) HIP_CHECK(hipStreamCreateWithFlags(&S.nifAux, hipStreamNonBlocking));
        // Option "nif_split" = x: the trace launches of batches 1.. run on x compute units of their own and the MLP + accumulate
        // passes on the other numCUs - x (two CU-masked streams), instead of sharing the chip. Batch 0's trace launch has the
        // device to itself and stays on the render's stream, which the masked trace stream follows (an event) and which
        // follows both again when the render ends. Only the SCHEDULE changes: same kernels, same buffers, same order per buffer.
        const uint32_t splitX = (two && S.opt.cus == 0 && S.ds.samplesPerPixel > 2u * S.scratchSamples) ? splitStreams(S) : 0u;
        hipStream_t mlpStream = splitX ? S.nifSplitMlp : two ? S.nifAux : stream;
        // (the MLP keeps its grid of one workgroup per unit of the whole chip: a mask that leaves the shader engines unequal - any x that
        // is not a multiple of their number - still gets the same share of workgroups per engine from the dispatcher, and with a grid
        // of only numCUs - x some units of the fuller engines would idle. The workgroups that find no unit free start when the first
        // ones leave, and leave at once: passes are drawn, nif_asm_kernel.hpp)
        const uint32_t mlpUnits = S.cus();
        uint32_t b = 0;
        mi_scene::NifSlots* pendQ = nullptr; uint32_t pendSc = 0, pendSegBase = 0;      // (nif_split: the batch whose accumulate pass is still to be enqueued)
        auto flushAccumulate = [&](hipStream_t on) {
          if (!pendQ) return;
          HIP_CHECK(hipStreamWaitEvent(on, pendQ->done, 0));
          hipLaunchKernelGGL(nif_accumulate_kernel, grid, block, 0, on, d_rays, cnt, pendSc, segShift, pendSegBase, pendQ->color, pendQ->tp, pendQ->u, pendQ->bgr);
          pendQ = nullptr;
        };
        try {
        for (uint32_t s0 = 0; s0 < S.ds.samplesPerPixel; s0 += S.scratchSamples, ++b) {
          mi_scene::NifSlots& q = S.nifSlots[two ? (b & 1u) : 0u];
          const uint32_t sc = std::min<uint32_t>(S.scratchSamples, S.ds.samplesPerPixel - s0);
          hipStream_t ts = (splitX && b > 0) ? S.nifSplitTrace : stream;          // this batch's trace launch
          if (splitX && b == 1) { HIP_CHECK(hipEventRecord(S.splitFirst, stream)); HIP_CHECK(hipStreamWaitEvent(ts, S.splitFirst, 0)); }
          if (two && q.donePending) { HIP_CHECK(hipStreamWaitEvent(ts, q.done, 0)); q.donePending = false; }
          HIP_CHECK(hipMemsetAsync(q.count, 0, sizeof(uint32_t), ts));
          WaveExtras ex;
          ex.sampleCount = sc; ex.segments = (sc + segLen - 1) / segLen; ex.segBase = s0 / segLen;     // (pixel, segment) atoms
          ex.u = q.u; ex.v = q.v; ex.slotColor = q.color; ex.slotTp = q.tp;
          ex.index = q.index; ex.count = q.count; ex.azimuthRotation = radians;
          ex.firstInSetup = S.opt.nifFirstTest ? 1u : 0u;
          const uint32_t wgCap = (two && b > 0 && !splitX) ? S.opt.nifTraceWgs : 0u;       // batch 0 has the device to itself
          const uint32_t unitsHere = (splitX && b > 0) ? splitX : 0u;
          if (S.opt.fullStats) launchWavefront<true>(S, d_rays, cnt, ts, ex, wgCap, unitsHere);
          else launchWavefront<false>(S, d_rays, cnt, ts, ex, wgCap, unitsHere);
          if (two) { HIP_CHECK(hipEventRecord(q.traced, ts)); HIP_CHECK(hipStreamWaitEvent(mlpStream, q.traced, 0)); }
          if (splitX) flushAccumulate(ts);
          std::pair<hipEvent_t, hipEvent_t> tm{nullptr, nullptr};
          if (S.opt.nifTiming) { HIP_CHECK(hipEventCreate(&tm.first)); HIP_CHECK(hipEventCreate(&tm.second)); S.nifTimes.push_back(tm); HIP_CHECK(hipEventRecord(tm.first, mlpStream)); }
          nif_launch_mlp(S.nif, q.u, q.v, q.index, q.count, cnt * sc, q.bgr, nullptr, mlpStream, true, S.opt.nifShape, mlpUnits, S.opt.nifGenerations);
          if (tm.second) HIP_CHECK(hipEventRecord(tm.second, mlpStream));
          if (splitX) {
            // the accumulate pass of this batch goes BEHIND the next batch's trace launch on the trace stream (which idles there until
            // the MLP is done; the set is traced into again only after it, by stream order): the MLP stream runs MLPs back to back
            HIP_CHECK(hipEventRecord(q.done, mlpStream));
            pendQ = &q; pendSc = sc; pendSegBase = ex.segBase;
          } else {
            hipLaunchKernelGGL(nif_accumulate_kernel, grid, block, 0, mlpStream, d_rays, cnt, sc, segShift, ex.segBase, q.color, q.tp, q.u, q.bgr);
            if (two) { HIP_CHECK(hipEventRecord(q.done, mlpStream)); q.donePending = true; }
          }
        }
        if (splitX) {
          flushAccumulate(b > 1 ? S.nifSplitTrace : stream);
          if (b > 1) { HIP_CHECK(hipEventRecord(S.splitFirst, S.nifSplitTrace)); HIP_CHECK(hipStreamWaitEvent(stream, S.splitFirst, 0)); }
        }
        if (two) for (mi_scene::NifSlots& q : S.nifSlots) if (q.donePending) { HIP_CHECK(hipStreamWaitEvent(stream, q.done, 0)); q.donePending = false; }
        } catch (...) {
          // a launch or a HIP call failed with MLP / accumulate passes queued on nifAux: they write the caller's rgb, so the
          // error must not return while they run behind the caller's stream (mi_render_device) - wait for them here
          if (two) { (void)hipStreamSynchronize(S.nifAux); for (mi_scene::NifSlots& q : S.nifSlots) q.donePending = false; }
          if (splitX) { (void)hipStreamSynchronize(S.nifSplitTrace); (void)hipStreamSynchronize(S.nifSplitMlp); }
          throw;
        }
      } else {
        const uint32_t segLen = segment_samples(S.ds.samplesPerPixel);
        for (uint32_t s = 0; s < S.ds.samplesPerPixel; ++s) {
          // a new segment: the finished ones move to the running total, rgb restarts from zero
          if (s != 0 && s % segLen == 0) hipLaunchKernelGGL(nif_segment_roll_kernel, grid, block, 0, stream, d_rays, S.d_segTotal, cnt, s / segLen, 0u);
          if (S.opt.fullStats) hipLaunchKernelGGL(path_trace_kernel<true>, grid, block, 0, stream, S.view(), d_rays, cnt, s, 1u, S.d_rng);
          else hipLaunchKernelGGL(path_trace_kernel<false>, grid, block, 0, stream, S.view(), d_rays, cnt, s, 1u, S.d_rng);
          nif_env_pass(S.nif, d_rays, cnt, radians, S.nifSlots[0].u, S.nifSlots[0].v, S.nifSlots[0].bgr, S.maxNifBatch, stream, S.opt.nifShape, S.cus(), S.opt.nifGenerations);
        }
        if (S.ds.samplesPerPixel > segLen) hipLaunchKernelGGL(nif_segment_roll_kernel, grid, block, 0, stream, d_rays, S.d_segTotal, cnt, 0u, 1u);
      }
      HIP_CHECK(hipEventRecord(S.nifDone, stream));
      S.nifPending = true;
    }
  } else {
    throw ArgError("mi_render: unknown render mode");
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipEventRecord(S.slotFor(stream).lastWork, stream));      // (~mi_scene waits for it)
}

// One wave that samples the work counter of `hip_stream`'s persistent launches `n` times, `period_ticks` (100-MHz ticks) apart, into
// d_samples as pairs {s_memrealtime, counter}: how fast a launch hands its work units out over its life - the ramp at its start, the
// moment the queue runs empty, the drain behind it (tools/launch_progress.py). It runs on a stream of the scene's own beside the launch it
// watches (K1w leaves every SIMD room for it), ends after n samples whatever happens, and changes nothing it looks at.
__global__ void __launch_bounds__(64) launch_progress_kernel(const uint32_t* counter, unsigned long long* samples, uint32_t n, uint32_t periodTicks) {
  if (threadIdx.x != 0) return;
  unsigned long long next = __builtin_amdgcn_s_memrealtime();
  for (uint32_t i = 0; i < n; ++i) {
    unsigned long long now;
    uint32_t spins = 0;
    do { __builtin_amdgcn_s_sleep(32); now = __builtin_amdgcn_s_memrealtime(); } while (now < next && ++spins < (1u << 22));
    samples[2 * i] = now;
    samples[2 * i + 1] = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    next += periodTicks;
  }
}

}  // namespace

extern "C" {

const char* mi_last_error(void) { return g_err.c_str(); }

const char* mi_version(void) {
#if MI_RAYLIB_VARIANTS
  return "mi_raylib 0.1 gfx950 fp-contract=off (bit-exact to the reference CPU path) +variants";
#else
  return "mi_raylib 0.1 gfx950 fp-contract=off (bit-exact to the reference CPU path)";
#endif
}

int mi_scene_create(const mi_scene_desc* desc, mi_scene** out) {
  if (!desc || !out) { g_err = "mi_scene_create: null argument"; return MI_ERR_INVALID_ARG; }
  *out = nullptr;
  mi_scene* S = nullptr;
  const int rc = guarded([&] {
    int count = 0;
    HIP_CHECK(hipGetDeviceCount(&count));
    if (count <= 0) throw DeviceError("no HIP device available (the product path has no CPU fallback)");
    if (desc->device < 0 || desc->device >= count) throw ArgError("mi_scene_create: device ordinal out of range");
    HIP_CHECK(hipSetDevice(desc->device));
    S = new mi_scene;
    S->device = desc->device;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, desc->device) == hipSuccess && cus > 0) S->numCUs = cus; }
    S->params = *desc;
    buildDeviceScene(*S, *desc);
    // the caller's arrays are not referenced after this point
    S->params.geometry = nullptr; S->params.mesh_info = nullptr; S->params.mesh_tris = nullptr; S->params.mesh_verts = nullptr;
    S->params.mesh_normals = nullptr; S->params.mat_ids = nullptr; S->params.materials = nullptr; S->params.bvh_nodes = nullptr;
    S->params.spheres = nullptr; S->params.discs = nullptr;
    S->opt.fromEnvironment();      // read once, into this scene (SceneOptions)
  });
  if (rc != MI_OK) { delete S; return rc; }
  *out = S;
  return MI_OK;
}

int mi_scene_create_from_blob(const uint8_t* blob, size_t size, const mi_scene_desc* extras, mi_scene** out) {
  if (!blob || !extras || !out) { g_err = "mi_scene_create_from_blob: null argument"; return MI_ERR_INVALID_ARG; }
  *out = nullptr;
  mi_scene_desc d = *extras;                 // spheres/discs, rng seed, crop window, pathTrace, device
  std::vector<uint64_t> aligned;             // only used when the caller's bytes are not 16-byte aligned
  const int rc = guarded([&] {
    if ((uintptr_t)blob % 16) {
      aligned.resize((size + 23) / 8);
      uint8_t* a = (uint8_t*)aligned.data();
      a += (16 - (uintptr_t)a % 16) % 16;
      std::memcpy(a, blob, size);
      blob = a;
    }
    try { mi::blob::deserialiseScene(blob, size, d, 16); }          // views into the bytes; create copies them
    catch (const std::runtime_error& e) { throw ArgError(std::string("mi_scene_create_from_blob: ") + e.what()); }
  });
  if (rc != MI_OK) return rc;
  return mi_scene_create(&d, out);
}

void mi_scene_destroy(mi_scene* scene) { delete scene; }

int mi_render_device(mi_scene* scene, int mode, void* d_rays, size_t n, void* hip_stream) {
  if (!scene || (!d_rays && n)) { g_err = "mi_render_device: null argument"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    HIP_CHECK(hipSetDevice(scene->device));
    launchRender(*scene, mode, (mi_trace_result*)d_rays, n, (hipStream_t)hip_stream);
  });
}

int mi_render(mi_scene* scene, int mode, mi_trace_result* rays, size_t n, mi_ray_callback cb, void* user) {
  if (!scene || (!rays && n)) { g_err = "mi_render: null argument"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    HIP_CHECK(hipSetDevice(scene->device));
    if (n == 0) { scene->traceTimeSecs = 0.0; return; }
    // Ray batches (src/IpuScene.cpp:110-172, 585-618): the stream is cut into batches of rayBatch rays that
    // flow through a two-slot pipeline — upload, trace and download of batch b+1 overlap the download and
    // the callback of batch b — each slot on its own HIP stream. Every pixel owns its RNG stream, so the
    // result does not depend on the batch size. The NIF path shares scratch buffers and runs one slot.
    const size_t batch = (scene->rayBatch && scene->rayBatch < n) ? scene->rayBatch : n;
    const size_t numBatches = (n + batch - 1) / batch;
    const int slots = (numBatches > 1 && !scene->nif.loaded()) ? 2 : 1;
    // The two device batch buffers and the two streams belong to the scene and are kept between calls (a preview
    // loop of short renders would otherwise pay an allocation, two stream creations and their tear-down per call).
    for (int i = 0; i < slots; ++i) {
      if (scene->batchCap[i] < batch) {
        if (scene->d_batch[i]) { HIP_CHECK(hipDeviceSynchronize()); (void)hipFree(scene->d_batch[i]); }
        scene->d_batch[i] = nullptr; scene->batchCap[i] = 0;
        HIP_CHECK(hipMalloc(&scene->d_batch[i], batch * sizeof(mi_trace_result)));
        scene->batchCap[i] = batch;
      }
      if (!scene->pipeStream[i]) HIP_CHECK(hipStreamCreateWithFlags(&scene->pipeStream[i], hipStreamNonBlocking));
    }
    mi_trace_result* const* d = scene->d_batch;
    hipStream_t const* st = scene->pipeStream;
    // The caller's stream is page-locked for the duration of the call, so the copies are real DMA transfers that
    // overlap the kernels of the other slot (pageable copies go through a staging buffer and serialise). Memory that
    // cannot be registered (or option "pin" = 0) just takes the pageable route; memory the caller has already
    // page-locked (hipHostMalloc / hipHostRegister / torch pin_memory) is used as it is.
    bool pinned = false;
    if (scene->opt.pin && n * sizeof(mi_trace_result) >= (size_t)1 << 20) {
      hipPointerAttribute_t attr{};
      const bool known = hipPointerGetAttributes(&attr, rays) == hipSuccess && attr.type == hipMemoryTypeHost;
      if (!known) {
        (void)hipGetLastError();
        pinned = hipHostRegister(rays, n * sizeof(mi_trace_result), hipHostRegisterDefault) == hipSuccess;
        if (!pinned) (void)hipGetLastError();
      }
    }
    auto cleanup = [&] { if (pinned) (void)hipHostUnregister(rays); };
    try {
      const auto t0 = std::chrono::steady_clock::now();
      auto finish = [&](size_t b) {
        HIP_CHECK(hipStreamSynchronize(st[b % slots]));
        const size_t first = b * batch, cnt = std::min(batch, n - first);
        if (cb) cb(user, b, rays + first, cnt);              // RayCallback::fetch, src/RayCallback.cpp:8-24
      };
      for (size_t b = 0; b < numBatches; ++b) {
        const int i = (int)(b % slots);
        if (b >= (size_t)slots) finish(b - slots);
        const size_t first = b * batch, cnt = std::min(batch, n - first);
        HIP_CHECK(hipMemcpyAsync(d[i], rays + first, cnt * sizeof(mi_trace_result), hipMemcpyHostToDevice, st[i]));
        launchRender(*scene, mode, d[i], cnt, st[i]);
        HIP_CHECK(hipMemcpyAsync(rays + first, d[i], cnt * sizeof(mi_trace_result), hipMemcpyDeviceToHost, st[i]));
      }
      for (size_t b = (numBatches > (size_t)slots ? numBatches - slots : 0); b < numBatches; ++b) finish(b);
      scene->traceTimeSecs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } catch (...) { (void)hipDeviceSynchronize(); cleanup(); throw; }
    cleanup();
  });
}

int mi_scene_set_option(mi_scene* scene, const char* key, const char* value) {
  if (!scene || !key || !value) { g_err = "mi_scene_set_option: null argument"; return MI_ERR_INVALID_ARG; }
  if (!scene->opt.set(key, value)) { g_err = std::string("mi_scene_set_option: ") + scene->opt.why + ": " + key + "=" + value; return MI_ERR_INVALID_ARG; }
  return MI_OK;
}

double mi_trace_time_secs(const mi_scene* scene) { return scene ? scene->traceTimeSecs : 0.0; }

int mi_get_counters(mi_scene* scene, uint64_t counts[4]) {
  if (!scene || !counts) { g_err = "mi_get_counters: null argument"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    HIP_CHECK(hipSetDevice(scene->device));
    HIP_CHECK(hipDeviceSynchronize());
    unsigned long long h[4];
    HIP_CHECK(hipMemcpy(h, scene->d_counters, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) counts[i] = h[i];
  });
}

int mi_get_phase_stats(mi_scene* scene, uint64_t stats[12]) {
  if (!scene || !stats) { g_err = "mi_get_phase_stats: null argument"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    HIP_CHECK(hipSetDevice(scene->device));
    HIP_CHECK(hipDeviceSynchronize());
    unsigned long long h[16];
    HIP_CHECK(hipMemcpy(h, scene->d_counters, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < 12; ++i) stats[i] = h[4 + i];
  });
}

int mi_get_pool_stats(mi_scene* scene, uint64_t stats[8]) {      // (zeros in a library built without the path-pool kernel)
  if (!scene || !stats) { g_err = "mi_get_pool_stats: null argument"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    HIP_CHECK(hipSetDevice(scene->device));
    HIP_CHECK(hipDeviceSynchronize());
    unsigned long long h[32];
    HIP_CHECK(hipMemcpy(h, scene->d_counters, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) stats[i] = h[16 + i];
  });
}

int mi_debug_launch_progress(mi_scene* scene, void* hip_stream, uint64_t* d_samples, uint32_t n, uint32_t period_ticks) {
  if (!scene || !d_samples || n == 0 || n > (1u << 20) || period_ticks == 0 || period_ticks > 100000000u) { g_err = "mi_debug_launch_progress: bad argument"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    HIP_CHECK(hipSetDevice(scene->device));
    LaunchSlot& slot = scene->slotFor((hipStream_t)hip_stream);
    if (!scene->debugStream) HIP_CHECK(hipStreamCreateWithFlags(&scene->debugStream, hipStreamNonBlocking));
    hipLaunchKernelGGL(launch_progress_kernel, dim3(1), dim3(64), 0, scene->debugStream, slot.d_workCounter, reinterpret_cast<unsigned long long*>(d_samples), n, period_ticks);
    HIP_CHECK(hipGetLastError());
  });
}

int mi_get_nif_timing(mi_scene* scene, double out[2]) {
  if (!scene || !out) { g_err = "mi_get_nif_timing: null argument"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    HIP_CHECK(hipSetDevice(scene->device));
    HIP_CHECK(hipDeviceSynchronize());
    out[0] = 0.0; out[1] = (double)scene->nifTimes.size();
    for (auto& e : scene->nifTimes) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) out[0] += ms;
      (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
    }
    scene->nifTimes.clear();
  });
}

int mi_get_nif_clock(mi_scene* scene, uint64_t out[2]) {
  if (!scene || !out) { g_err = "mi_get_nif_clock: null argument"; return MI_ERR_INVALID_ARG; }
  if (!scene->nif.loaded()) { g_err = "mi_get_nif_clock: no NIF model loaded"; return MI_ERR_NO_NIF; }
  return guarded([&] {
    HIP_CHECK(hipSetDevice(scene->device));
    HIP_CHECK(hipDeviceSynchronize());
    unsigned long long h[2] = {0, 0};
    HIP_CHECK(hipMemcpy(h, scene->nif.d_clock, sizeof h, hipMemcpyDeviceToHost));
    out[0] = h[0]; out[1] = h[1];
  });
}

int mi_reset_counters(mi_scene* scene) {
  if (!scene) { g_err = "mi_reset_counters: null argument"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    HIP_CHECK(hipSetDevice(scene->device));
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemset(scene->d_counters, 0, 32 * sizeof(unsigned long long)));
  });
}

int mi_scene_set_nif(mi_scene* scene, uint32_t num_layers, const float* const* kernels, const float* const* biases,
                     const uint32_t* rows, const uint32_t* cols, const uint8_t* relu,
                     uint32_t embedding_dimension, float max_value, const float mean[3], int32_t log_tonemap) {
  if (!scene || !kernels || !rows || !cols || !relu || !mean) { g_err = "mi_scene_set_nif: null argument"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    HIP_CHECK(hipSetDevice(scene->device));
    try { scene->nif.load(num_layers, kernels, biases, rows, cols, relu, embedding_dimension, max_value, mean, log_tonemap); }
    catch (const std::invalid_argument& e) { throw ArgError(e.what()); }
  });
}

int mi_scene_set_hdri_rotation(mi_scene* scene, float degrees) {
  if (!scene) { g_err = "null scene"; return MI_ERR_INVALID_ARG; }
  scene->hdriRotationDegrees = degrees;
  return MI_OK;
}

int mi_scene_set_max_nif_batch(mi_scene* scene, size_t rays_per_batch) {
  if (!scene) { g_err = "null scene"; return MI_ERR_INVALID_ARG; }
  scene->maxNifBatch = rays_per_batch;
  return MI_OK;
}

int mi_scene_set_ray_batch(mi_scene* scene, size_t rays_per_batch) {
  if (!scene) { g_err = "null scene"; return MI_ERR_INVALID_ARG; }
  scene->rayBatch = rays_per_batch;
  return MI_OK;
}

int mi_nif_infer_device(mi_scene* scene, const float* d_u, const float* d_v, float* d_bgr, size_t n, void* hip_stream) {
  if (!scene || ((!d_u || !d_v || !d_bgr) && n)) { g_err = "mi_nif_infer_device: null argument"; return MI_ERR_INVALID_ARG; }
  if (!scene->nif.loaded()) { g_err = "mi_nif_infer_device: no NIF model loaded"; return MI_ERR_NO_NIF; }
  return guarded([&] {
    HIP_CHECK(hipSetDevice(scene->device));
    nif_infer(scene->nif, d_u, d_v, d_bgr, n, scene->maxNifBatch, (hipStream_t)hip_stream, scene->opt.nifShape, scene->cus(), scene->opt.nifGenerations);
    HIP_CHECK(hipGetLastError());
  });
}

}  // extern "C"

#include "group_render.hpp"

#if MI_NIF_STAMPS
// diagnostic build only: reads and clears the NIF kernel's stamp sums (nif_kernels.hpp)
extern "C" int mi_debug_nif_stamps(unsigned long long* out8) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(mi::nif_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(mi::nif_stamps), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

