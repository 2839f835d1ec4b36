// trace_cli.cpp — `trace`: the reference CLI's surface (trace.cpp:338-544) on the MI355X renderer.
//
// Same flags, same defaults, same output naming (<prefix>_<vis>_gpu.exr instead of _ipu.exr). The CPU
// and Embree renderers of the reference are not part of this tool: the GPU path is what it drives
// (as with the reference's --ipu-only); parity against the CPU algorithm lives in tests/ with the
// oracle. Images: OpenEXR (uncompressed scanline, float32 B/G/R like cv::imwrite of CV_32FC3) and PFM.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <regex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/mi_scene_host.h"
#include "IpuScene.hpp"

namespace {

struct Options {
  std::string outprefix = "out", crop, meshFile, nifHdri, scene = "box", visualise = "rgb", renderMode = "path-trace", logLevel = "info";
  uint32_t gpus = 1, replicas = 0;
  int transport = 0;
  size_t raysPerWorker = 1, maxNifBatch = 0;
  int width = 768, height = 432;
  float antiAlias = .25f, hdriRotation = 0.f, availableMemoryProportion = 0.6f;
  bool loadNormals = false, gpuOnly = false, rayCallback = false;
  uint32_t maxPathLength = 10, rouletteStartDepth = 3, samples = 256;
  uint64_t seed = 1442;
};

const char* kHelp =
    "  --help                              Show command help.\n"
    "  -o [ --outprefix ] arg (=out)       Set the output filename prefix.\n"
    "  --gpus arg (=1)                     Select number of GPUs (alias: --ipus). The ray stream is dealt to them in bands of 8\n"
    "                                      image rows; the frame is assembled on the first GPU with one RCCL gather.\n"
    "  --replicas arg (=gpus)              Scene replicas the stream is dealt to (replica i runs on GPU i mod gpus).\n"
    "  --gather arg (=auto)                How the replicas' results reach the first GPU: auto | rccl | copy.\n"
    "  --rays-per-worker arg (=1)          With --ipu-ray-callback: the ray batch the callback is called for is\n"
    "                                      1440 x 6 x this many rays (one IPU's batch); no effect otherwise.\n"
    "  -w [ --width ] arg (=768)           Set rendered image width.\n"
    "  -h [ --height ] arg (=432)          Set rendered image height.\n"
    "  --crop arg                          Window of the image to render: wxh+c+r.\n"
    "  --anti-alias arg (=0.25)            Width of anti-aliasing noise distribution in pixels.\n"
    "  --mesh-file arg                     A Collada (.dae) scene with camera to render instead of a built-in scene, or the\n"
    "                                      glTF-binary mesh placed in the built-in box scene (default assets/monkey_bust.glb).\n"
    "  --nif-hdri arg                      Path to the 'assets.extra' directory of a NIF model (nif_metadata.txt + converted.hdf5 | nif_weights.bin).\n"
    "  --hdri-rotation arg (=0)            Azimuthal rotation for HDRI environment map (degrees).\n"
    "  --load-normals                      Load (and interpolate) vertex normals of a mesh file.\n"
    "  --scene arg (=box)                  One of the built in scenes [box-simple, box, spheres].\n"
    "  --visualise arg (=rgb)              One of [rgb, normal, hitpoint, tfar, color, id].\n"
    "  --render-mode arg (=path-trace)     One of [shadow-trace, path-trace].\n"
    "  --max-path-length arg (=10)         Max path length for path tracing.\n"
    "  --roulette-start-depth arg (=3)     Path length after which rays can be randomly terminated.\n"
    "  --samples arg (=256)                Number of samples per pixel for path tracing.\n"
    "  --seed arg (=1442)                  RNG seed.\n"
    "  --available-memory-proportion arg   Accepted for compatibility (no effect).\n"
    "  --max-nif-batch-size arg (=0)       Maximum batch-size for the NIF neural network (0 = whole ray batch).\n"
    "  --ipu-only / --gpu-only             Accepted for compatibility (this tool only renders on the GPU).\n"
    "  --ipu-ray-callback                  Receive results through the partial-result callback.\n"
    "  --log-level arg (=info)             One of 'trace','debug','info','warn','err','critical','off'.\n";

int logRank(const std::string& l) {
  static const std::map<std::string, int> m = {{"trace", 0}, {"debug", 1}, {"info", 2}, {"warn", 3}, {"err", 4}, {"critical", 5}, {"off", 6}};
  auto it = m.find(l);
  if (it == m.end()) throw std::runtime_error("Invalid log-level: '" + l + "'");
  return it->second;
}
int g_level = 2;
void logf(int lvl, const char* tag, const char* fmt, ...) {
  if (lvl < g_level) return;
  va_list ap; va_start(ap, fmt);
  std::fprintf(stderr, "[%s] ", tag); std::vfprintf(stderr, fmt, ap); std::fprintf(stderr, "\n");
  va_end(ap);
}

Options parse(int argc, char** argv) {
  Options o;
  auto need = [&](int& i) -> std::string { if (i + 1 >= argc) throw std::runtime_error(std::string("missing value for ") + argv[i]); return argv[++i]; };
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--help") { std::printf("%s", kHelp); throw std::runtime_error("Show help"); }
    else if (a == "-o" || a == "--outprefix") o.outprefix = need(i);
    else if (a == "--gpus" || a == "--ipus") o.gpus = (uint32_t)std::stoul(need(i));
    else if (a == "--replicas") o.replicas = (uint32_t)std::stoul(need(i));
    else if (a == "--gather") { const std::string g = need(i); if (g == "auto") o.transport = 0; else if (g == "rccl") o.transport = 1; else if (g == "copy") o.transport = 2; else throw std::runtime_error("the argument for option '--gather' is invalid"); }
    else if (a == "--rays-per-worker") o.raysPerWorker = std::stoul(need(i));
    else if (a == "-w" || a == "--width") o.width = std::stoi(need(i));
    else if (a == "-h" || a == "--height") o.height = std::stoi(need(i));
    else if (a == "--crop") o.crop = need(i);
    else if (a == "--anti-alias") o.antiAlias = std::stof(need(i));
    else if (a == "--mesh-file") o.meshFile = need(i);
    else if (a == "--nif-hdri") o.nifHdri = need(i);
    else if (a == "--hdri-rotation") o.hdriRotation = std::stof(need(i));
    else if (a == "--load-normals") o.loadNormals = true;
    else if (a == "--scene") o.scene = need(i);
    else if (a == "--visualise") o.visualise = need(i);
    else if (a == "--render-mode") o.renderMode = need(i);
    else if (a == "--max-path-length") o.maxPathLength = (uint32_t)std::stoul(need(i));
    else if (a == "--roulette-start-depth") o.rouletteStartDepth = (uint32_t)std::stoul(need(i));
    else if (a == "--samples") o.samples = (uint32_t)std::stoul(need(i));
    else if (a == "--seed") o.seed = std::stoull(need(i));
    else if (a == "--available-memory-proportion") o.availableMemoryProportion = std::stof(need(i));
    else if (a == "--max-nif-batch-size") o.maxNifBatch = std::stoul(need(i));
    else if (a == "--ipu-only" || a == "--gpu-only") o.gpuOnly = true;
    else if (a == "--ipu-ray-callback") o.rayCallback = true;
    else if (a == "--log-level") o.logLevel = need(i);
    else throw std::runtime_error("unrecognised option '" + a + "'");
  }
  static const char* vis[] = {"rgb", "normal", "hitpoint", "tfar", "color", "id"};
  if (std::find_if(std::begin(vis), std::end(vis), [&](const char* v) { return o.visualise == v; }) == std::end(vis))
    throw std::runtime_error("the argument for option '--visualise' is invalid");                       // trace.cpp:398-403
  if (o.renderMode != "shadow-trace" && o.renderMode != "path-trace")
    throw std::runtime_error("the argument for option '--render-mode' is invalid");                     // :405-410
  if (o.meshFile.empty() && o.loadNormals)
    throw std::runtime_error("Option 'load-normals' is not valid without the 'mesh-file' option");      // :412-414
  if (o.renderMode == "path-trace" && o.visualise != "rgb")
    throw std::runtime_error("Running path-tracing without visualise=rgb is not advised.");             // app_utils.cpp:244-246
  return o;
}

// parseCropString, src/app_utils.cpp:212-233
bool parseCrop(const std::string& s, int32_t out[4]) {
  if (s.empty()) return false;
  std::smatch m;
  if (!std::regex_search(s, m, std::regex("(\\d+)x(\\d+)\\+(\\d+)\\+(\\d+)")) || m.size() != 5)
    throw std::runtime_error("Badly formatted string used for --crop.");
  for (int i = 0; i < 4; ++i) out[i] = std::atoi(m.str(i + 1).c_str());
  return true;
}

// visualiseHits, src/app_utils.cpp:61-127. Returns the image as B,G,R float triples (OpenCV order).
unsigned visualise(const std::vector<mi_trace_result>& rays, const mi_scene_desc& d, const std::string& mode, int width, int height, std::vector<float>& bgr) {
  bgr.assign((size_t)width * height * 3, 0.f);
  unsigned hits = 0;
  for (const auto& tr : rays) {
    const auto& h = tr.h;
    float v[3] = {0, 0, 0};
    const bool valid = h.geom_id != MI_INVALID_GEOM;
    if (mode == "rgb") { v[0] = tr.rgb.z; v[1] = tr.rgb.y; v[2] = tr.rgb.x; }
    else if (mode == "tfar") { v[0] = v[1] = v[2] = h.r.t_max; }
    else if (valid) {
      if (mode == "id") { v[0] = (float)(h.geom_id + 1); v[1] = (float)(h.prim_id + 1); v[2] = (float)(d.mat_ids[h.geom_id] + 1); }
      else if (mode == "normal") { v[0] = h.normal.z; v[1] = h.normal.y; v[2] = h.normal.x; }
      else if (mode == "color") { const auto& c = d.materials[d.mat_ids[h.geom_id]].albedo; v[0] = c.z; v[1] = c.y; v[2] = c.x; }
      else if (mode == "hitpoint") { v[0] = h.r.origin.z; v[1] = h.r.origin.y; v[2] = h.r.origin.x; }
    }
    const int row = (int)tr.u, col = (int)tr.v;
    if (row >= 0 && row < height && col >= 0 && col < width) memcpy(&bgr[((size_t)row * width + col) * 3], v, sizeof v);
    if (valid) ++hits;
  }
  return hits;
}

void writePfm(const std::string& path, const std::vector<float>& bgr, int w, int h) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot write " + path);
  std::fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
  std::vector<float> row((size_t)w * 3);
  for (int y = h - 1; y >= 0; --y) {           // PFM stores bottom row first, RGB
    for (int x = 0; x < w; ++x) for (int c = 0; c < 3; ++c) row[(size_t)x * 3 + c] = bgr[((size_t)y * w + x) * 3 + (2 - c)];
    std::fwrite(row.data(), sizeof(float), row.size(), f);
  }
  std::fclose(f);
}

// Minimal OpenEXR 2.0 writer: single part, scanline, no compression, float channels B, G, R.
void writeExr(const std::string& path, const std::vector<float>& bgr, int w, int h) {
  std::vector<uint8_t> out;
  auto put = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; out.insert(out.end(), b, b + n); };
  auto putStr = [&](const char* s) { put(s, strlen(s) + 1); };
  auto putI = [&](int32_t v) { put(&v, 4); };
  auto putF = [&](float v) { put(&v, 4); };
  auto attr = [&](const char* name, const char* type, int32_t size) { putStr(name); putStr(type); putI(size); };
  putI(20000630); putI(2);
  attr("channels", "chlist", 3 * 18 + 1);
  for (const char* ch : {"B", "G", "R"}) { putStr(ch); putI(2 /*FLOAT*/); uint8_t z[4] = {0, 0, 0, 0}; put(z, 4); putI(1); putI(1); }
  out.push_back(0);
  attr("compression", "compression", 1); out.push_back(0);
  attr("dataWindow", "box2i", 16); putI(0); putI(0); putI(w - 1); putI(h - 1);
  attr("displayWindow", "box2i", 16); putI(0); putI(0); putI(w - 1); putI(h - 1);
  attr("lineOrder", "lineOrder", 1); out.push_back(0);
  attr("pixelAspectRatio", "float", 4); putF(1.f);
  attr("screenWindowCenter", "v2f", 8); putF(0.f); putF(0.f);
  attr("screenWindowWidth", "float", 4); putF(1.f);
  out.push_back(0);
  const uint64_t lineBytes = 8 + (uint64_t)w * 3 * 4;
  uint64_t off = out.size() + (uint64_t)h * 8;
  for (int y = 0; y < h; ++y) { put(&off, 8); off += lineBytes; }
  std::vector<float> plane((size_t)w);
  for (int y = 0; y < h; ++y) {
    putI(y); putI(w * 3 * 4);
    for (int c = 0; c < 3; ++c) {   // channels in alphabetical order B, G, R == our storage order
      for (int x = 0; x < w; ++x) plane[x] = bgr[((size_t)y * w + x) * 3 + c];
      put(plane.data(), (size_t)w * 4);
    }
  }
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot write " + path);
  std::fwrite(out.data(), 1, out.size(), f);
  std::fclose(f);
}

}  // namespace

int main(int argc, char** argv) {
  Options args;
  try {
    args = parse(argc, argv);
    g_level = logRank(args.logLevel);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "[info] Exiting after: %s.\n", e.what());
    return EXIT_FAILURE;
  }
  logf(0, "trace", "HitRecord size: %zu TraceResult size: %zu CompactBVH2Node size: %zu Ray size: %zu", sizeof(mi_hit_record),
       sizeof(mi_trace_result), sizeof(mi_bvh_node), sizeof(mi_ray));

  // ===== Scene setup (buildSceneDescription + buildSceneData, trace.cpp:452-456) =====
  mi_host_scene* host = nullptr;
  std::string mesh = args.meshFile;
  if (mesh.empty()) mesh = "assets/monkey_bust.glb";
  const bool importWholeScene = args.meshFile.size() > 4 && args.meshFile.substr(args.meshFile.size() - 4) == ".dae";
  const int rcScene = importWholeScene ? mi_host_scene_import(args.meshFile.c_str(), args.loadNormals ? 1 : 0, &host)   // importScene, app_utils.cpp:274-275
                                       : mi_host_scene_builtin(args.scene.c_str(), mesh.c_str(), &host);
  if (rcScene != MI_OK) {
    std::fprintf(stderr, "[error] %s\n", mi_host_last_error());
    return EXIT_FAILURE;
  }
  mi_scene_desc sceneRef;
  mi_host_scene_fill_desc(host, &sceneRef);
  int32_t crop[4] = {args.width, args.height, 0, 0};
  parseCrop(args.crop, crop);
  logf(2, "info", "Rendering window: width: %d, height: %d, start col: %d, start row: %d", crop[0], crop[1], crop[2], crop[3]);
  sceneRef.image_width = (float)args.width; sceneRef.image_height = (float)args.height;
  sceneRef.anti_alias_scale = args.antiAlias;
  sceneRef.max_path_length = args.maxPathLength; sceneRef.roulette_start_depth = args.rouletteStartDepth;
  sceneRef.samples_per_pixel = args.samples; sceneRef.rng_seed = args.seed;
  sceneRef.window_w = crop[0]; sceneRef.window_h = crop[1]; sceneRef.window_c = crop[2]; sceneRef.window_r = crop[3];
  sceneRef.path_trace = (args.renderMode == "path-trace") ? 1 : 0;
  logf(1, "debug", "BVH nodes: %u, max leaf depth: %u, triangles: %u", sceneRef.num_nodes, sceneRef.max_leaf_depth, sceneRef.num_tris);

  // ===== renderIPU, trace.cpp:270-336 =====
  std::vector<mi_trace_result> rayStream((size_t)crop[0] * crop[1]);
  mi_init_ray_stream(&sceneRef, rayStream.data(), rayStream.size());
  mi::IpuScene::RayCallbackFn cb = [](std::size_t idx, const std::vector<mi_trace_result>&) { logf(1, "debug", "Application callback received batch %zu", idx); };
  std::vector<mi_sphere> spheres(sceneRef.spheres, sceneRef.spheres + sceneRef.num_spheres);
  std::vector<mi_disc> discs(sceneRef.discs, sceneRef.discs + sceneRef.num_discs);
  mi::IpuScene gpuScene(spheres, discs, sceneRef, rayStream, args.raysPerWorker, args.rayCallback ? &cb : nullptr);
  mi::RuntimeConfig rc; rc.numGpus = args.gpus; rc.numReplicas = args.replicas ? args.replicas : args.gpus; rc.transport = args.transport;
  gpuScene.setRuntimeConfig(rc);
  if (!args.nifHdri.empty()) gpuScene.loadNifModel(args.nifHdri);
  gpuScene.setHdriRotation(args.hdriRotation);
  gpuScene.setAvailableMemoryProportion(args.availableMemoryProportion);
  gpuScene.setMaxNifBatchSize(args.maxNifBatch);
  logf(2, "info", "GPU Rendering started.");
  const int status = gpuScene.run();
  logf(2, "info", "GPU Rendering finished.");
  if (status != EXIT_SUCCESS) { mi_host_scene_destroy(host); return status; }
  if (sceneRef.path_trace) mi_scale_rgb(rayStream.data(), rayStream.size(), 1.f / (float)sceneRef.samples_per_pixel);
  const double secs = gpuScene.getTraceTimeSecs();
  const double castsPerRay = sceneRef.path_trace ? sceneRef.samples_per_pixel : 1;
  logf(2, "info", "GPU time: %g", secs);
  logf(2, "info", "GPU %s per second: %g", sceneRef.path_trace ? "paths" : "rays", rayStream.size() * castsPerRay / secs);
  logf(2, "info", "GPU ray casts per second: %g", (double)gpuScene.rayCasts() / secs);
  {
    uint64_t moved[5];
    gpuScene.lastTransfer(moved);
    if (moved[2]) logf(2, "info", "Replicas: %u on %u GPU(s); %llu bands dealt in %llu strided uploads, gathered with %llu RCCL send/recv pairs and %llu peer copies", rc.numReplicas, rc.numGpus,
                       (unsigned long long)moved[2], (unsigned long long)moved[3], (unsigned long long)moved[0], (unsigned long long)moved[1]);
  }

  std::vector<float> image;
  const unsigned hitCount = visualise(rayStream, sceneRef, args.visualise, args.width, args.height, image);
  const std::string prefix = args.outprefix + "_" + args.visualise + "_";
  try {
    writeExr(prefix + "gpu.exr", image, args.width, args.height);
    writePfm(prefix + "gpu.pfm", image, args.width, args.height);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "[error] %s\n", e.what());
    mi_host_scene_destroy(host);
    return EXIT_FAILURE;
  }
  logf(1, "debug", "GPU hit count: %u", hitCount);
  logf(2, "info", "Done.");
  mi_host_scene_destroy(host);
  return EXIT_SUCCESS;
}
