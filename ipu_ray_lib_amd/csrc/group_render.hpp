// group_render.hpp — one process, several GPUs: the multi-device renderer behind mi_group_* (include/mi_raylib.h).
// Included at the end of raylib.hip (it uses mi_scene and launchRender of that translation unit).
//
// Reference: an IpuScene with numReplicas > 1 replicates the scene on every device (src/IpuScene.cpp:473-483), lets
// the replicas pull disjoint ray batches round-robin from the one host stream (:676-684) and writes every batch back
// into the caller's stream (:699-732). Here:
//   * one mi_scene per replica, each on its device (several replicas may share a device: that is how the path is
//     rehearsed on a one-GPU box);
//   * the stream is dealt in bands (ray_shard.hpp: 8 window rows per band, band b to replica b % R); every replica
//     uploads its bands, traces them on its own HIP stream - no exchange while the frame renders;
//   * at frame end ONE RCCL group call moves every replica's finished stream to the root device over xGMI
//     (ncclSend on the replica's stream / ncclRecv on the root's stream, point to point: each peer uses its own
//     direct link to the root; a ring collective would be bound by one link), a de-interleave kernel restores stream
//     order, and the frame leaves the root in one download.
// RCCL is reached through dlopen (librccl.so.1): the library has no link-time dependency on it, a process that never
// builds a group never loads it, and a host process that already carries an RCCL (torch) shares that copy.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <map>

#include "ray_shard.hpp"

namespace mi {

// frame[i] = gathered[offset[replica(i)] + pos(i)] for every 84-byte record, one thread per dword
__global__ void __launch_bounds__(256) deinterleave_kernel(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ frame, size_t n, size_t band,
                                                           uint32_t replicas, const unsigned long long* __restrict__ offsets) {
  constexpr size_t W = sizeof(mi_trace_result) / 4;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * W) return;
  const size_t i = idx / W, w = idx - i * W;
  uint32_t r; size_t pos;
  shard::locate(band, replicas, i, r, pos);
  frame[idx] = gathered[((size_t)offsets[r] + pos) * W + w];
}

struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) commInitAll = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
  decltype(&ncclGroupStart) groupStart = nullptr;
  decltype(&ncclGroupEnd) groupEnd = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGetErrorString) errorString = nullptr;
  bool load(std::string& why) {
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so"}) { lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) { why = std::string("cannot load RCCL: ") + dlerror(); return false; }
    auto sym = [&](const char* s) { void* p = dlsym(lib, s); if (!p) why = std::string("RCCL symbol missing: ") + s; return p; };
    commInitAll = (decltype(commInitAll))sym("ncclCommInitAll"); commDestroy = (decltype(commDestroy))sym("ncclCommDestroy");
    groupStart = (decltype(groupStart))sym("ncclGroupStart"); groupEnd = (decltype(groupEnd))sym("ncclGroupEnd");
    send = (decltype(send))sym("ncclSend"); recv = (decltype(recv))sym("ncclRecv"); errorString = (decltype(errorString))sym("ncclGetErrorString");
    return commInitAll && commDestroy && groupStart && groupEnd && send && recv && errorString;
  }
};

}  // namespace mi

struct mi_group {
  struct Replica {
    mi_scene* scene = nullptr;
    int device = 0, rank = 0;                  // rank = index of the device in `commDevices`
    mi_trace_result* d_share = nullptr; size_t shareCap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
  };
  std::vector<Replica> replicas;
  std::vector<int> commDevices;                // distinct devices, root first
  std::vector<ncclComm_t> comms;               // one per distinct device (ncclCommInitAll)
  mi::RcclApi rccl;
  bool useRccl = false;
  mi_trace_result* d_gather = nullptr; mi_trace_result* d_frame = nullptr; size_t frameCap = 0;      // on the root device
  unsigned long long* d_offsets = nullptr;
  size_t rayBatch = 0;
  double traceTimeSecs = 0.0;
  // what the last render did (for tests and logs)
  uint64_t lastRcclMessages = 0, lastCopyMessages = 0, lastBands = 0;

  ~mi_group() {
    for (size_t i = 0; i < comms.size(); ++i) if (comms[i]) { (void)hipSetDevice(commDevices[i]); (void)rccl.commDestroy(comms[i]); }
    for (Replica& r : replicas) {
      (void)hipSetDevice(r.device);
      if (r.d_share && r.d_share != d_gather) (void)hipFree(r.d_share);
      if (r.stream) (void)hipStreamDestroy(r.stream);
      if (r.done) (void)hipEventDestroy(r.done);
      delete r.scene;
    }
    if (!replicas.empty()) (void)hipSetDevice(replicas[0].device);
    if (d_gather) (void)hipFree(d_gather);
    if (d_frame) (void)hipFree(d_frame);
    if (d_offsets) (void)hipFree(d_offsets);
  }
};

namespace {

#define RCCL_CHECK(g, expr)                                                                                   \
  do {                                                                                                        \
    ncclResult_t _r = (expr);                                                                                 \
    if (_r != ncclSuccess) throw DeviceError(std::string(#expr) + ": " + (g).rccl.errorString(_r));          \
  } while (0)

void groupRender(mi_group& G, int mode, mi_trace_result* rays, size_t n, mi_ray_callback cb, void* user) {
  const uint32_t R = (uint32_t)G.replicas.size();
  mi_group::Replica& root = G.replicas[0];
  if (n == 0) { G.traceTimeSecs = 0.0; return; }
  const size_t band = shard::band_rays(n, (uint32_t)std::max(root.scene->params.window_w, 0));
  std::vector<size_t> count(R), offset(R + 1, 0);
  for (uint32_t r = 0; r < R; ++r) { count[r] = shard::replica_count(n, band, R, r); offset[r + 1] = offset[r] + count[r]; }

  // ---- buffers: every replica's stream on its device; on the root the gathered streams and the assembled frame ----
  HIP_CHECK(hipSetDevice(root.device));
  if (G.frameCap < n) {
    HIP_CHECK(hipDeviceSynchronize());
    if (G.d_gather) (void)hipFree(G.d_gather);
    if (G.d_frame) (void)hipFree(G.d_frame);
    G.d_gather = G.d_frame = nullptr; G.frameCap = 0;
    HIP_CHECK(hipMalloc(&G.d_gather, n * sizeof(mi_trace_result)));
    HIP_CHECK(hipMalloc(&G.d_frame, n * sizeof(mi_trace_result)));
    G.frameCap = n;
  }
  if (!G.d_offsets) HIP_CHECK(hipMalloc(&G.d_offsets, 64 * sizeof(unsigned long long)));
  {
    unsigned long long h[64];
    for (uint32_t r = 0; r < R; ++r) h[r] = offset[r];
    HIP_CHECK(hipMemcpyAsync(G.d_offsets, h, R * sizeof(unsigned long long), hipMemcpyHostToDevice, root.stream));
    HIP_CHECK(hipStreamSynchronize(root.stream));        // (h is a stack array)
  }
  for (uint32_t r = 0; r < R; ++r) {
    mi_group::Replica& P = G.replicas[r];
    if (r == 0) { P.d_share = G.d_gather; P.shareCap = G.frameCap; continue; }      // the root replica traces in place: its stream is the head of the gathered buffer
    if (P.shareCap < count[r]) {
      HIP_CHECK(hipSetDevice(P.device));
      if (P.d_share) { HIP_CHECK(hipStreamSynchronize(P.stream)); (void)hipFree(P.d_share); }
      P.d_share = nullptr; P.shareCap = 0;
      HIP_CHECK(hipMalloc(&P.d_share, std::max<size_t>(count[r], 1) * sizeof(mi_trace_result)));
      P.shareCap = count[r];
    }
  }

  bool pinned = false;
  if (root.scene->opt.pin && n * sizeof(mi_trace_result) >= (size_t)1 << 20) {
    hipPointerAttribute_t attr{};
    const bool known = hipPointerGetAttributes(&attr, rays) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!known) {
      (void)hipGetLastError();
      pinned = hipHostRegister(rays, n * sizeof(mi_trace_result), hipHostRegisterPortable) == hipSuccess;
      if (!pinned) (void)hipGetLastError();
    }
  }
  auto cleanup = [&] { if (pinned) (void)hipHostUnregister(rays); };
  try {
    const auto t0 = std::chrono::steady_clock::now();
    // ---- deal the bands, trace: no exchange while the frame renders ----
    G.lastBands = 0;
    for (uint32_t r = 0; r < R; ++r) {
      mi_group::Replica& P = G.replicas[r];
      HIP_CHECK(hipSetDevice(P.device));
      size_t k = 0;
      for (size_t b = r; b * band < n; b += R, ++k) {
        const size_t first = b * band, len = std::min(band, n - first);
        HIP_CHECK(hipMemcpyAsync(P.d_share + k * band, rays + first, len * sizeof(mi_trace_result), hipMemcpyHostToDevice, P.stream));
        ++G.lastBands;
      }
      launchRender(*P.scene, mode, P.d_share, count[r], P.stream);
      if (r != 0) HIP_CHECK(hipEventRecord(P.done, P.stream));
    }
    // ---- the ONE collective of the frame: every replica's stream to the root ----
    G.lastRcclMessages = G.lastCopyMessages = 0;
    if (R > 1) {
      if (G.useRccl) {
        // Inside the group call every communicator (= device) uses ONE stream: that of the device's first replica,
        // which first waits for the device's other replicas to finish tracing.
        std::vector<hipStream_t> commStream(G.commDevices.size(), nullptr);
        for (uint32_t r = 0; r < R; ++r) {
          mi_group::Replica& P = G.replicas[r];
          if (!commStream[P.rank]) { commStream[P.rank] = P.stream; continue; }
          HIP_CHECK(hipSetDevice(P.device));
          HIP_CHECK(hipStreamWaitEvent(commStream[P.rank], P.done, 0));
        }
        RCCL_CHECK(G, G.rccl.groupStart());
        for (uint32_t r = 1; r < R; ++r) {
          mi_group::Replica& P = G.replicas[r];
          if (count[r] == 0) continue;
          const size_t bytes = count[r] * sizeof(mi_trace_result);
          RCCL_CHECK(G, G.rccl.send(P.d_share, bytes, ncclUint8, root.rank, G.comms[P.rank], commStream[P.rank]));
          RCCL_CHECK(G, G.rccl.recv(G.d_gather + offset[r], bytes, ncclUint8, P.rank, G.comms[root.rank], root.stream));
          ++G.lastRcclMessages;
        }
        RCCL_CHECK(G, G.rccl.groupEnd());
      } else {
        HIP_CHECK(hipSetDevice(root.device));
        for (uint32_t r = 1; r < R; ++r) {
          mi_group::Replica& P = G.replicas[r];
          if (count[r] == 0) continue;
          HIP_CHECK(hipStreamWaitEvent(root.stream, P.done, 0));
          HIP_CHECK(hipMemcpyPeerAsync(G.d_gather + offset[r], root.device, P.d_share, P.device, count[r] * sizeof(mi_trace_result), root.stream));
          ++G.lastCopyMessages;
        }
      }
    }
    // ---- stream order again, then home ----
    HIP_CHECK(hipSetDevice(root.device));
    const mi_trace_result* result = G.d_gather;
    if (R > 1) {
      const size_t words = n * (sizeof(mi_trace_result) / 4);
      hipLaunchKernelGGL(deinterleave_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, root.stream, reinterpret_cast<const uint32_t*>(G.d_gather),
                         reinterpret_cast<uint32_t*>(G.d_frame), n, band, R, G.d_offsets);
      HIP_CHECK(hipGetLastError());
      result = G.d_frame;
    }
    HIP_CHECK(hipMemcpyAsync(rays, result, n * sizeof(mi_trace_result), hipMemcpyDeviceToHost, root.stream));
    HIP_CHECK(hipStreamSynchronize(root.stream));
    G.traceTimeSecs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    root.scene->traceTimeSecs = G.traceTimeSecs;
    // RayCallback::fetch (src/RayCallback.cpp:8-24): one call per ray batch, in batch order, once the frame is home
    if (cb) {
      const size_t batch = (G.rayBatch && G.rayBatch < n) ? G.rayBatch : n;
      for (size_t b = 0, first = 0; first < n; ++b, first += batch) cb(user, b, rays + first, std::min(batch, n - first));
    }
  } catch (...) {
    for (auto& P : G.replicas) { (void)hipSetDevice(P.device); (void)hipDeviceSynchronize(); }
    cleanup();
    throw;
  }
  cleanup();
}

}  // namespace

extern "C" {

int mi_group_create(const mi_scene_desc* desc, const int32_t* devices, uint32_t num_replicas, int32_t transport, mi_group** out) {
  if (!desc || !devices || !out || num_replicas == 0 || num_replicas > 64) { g_err = "mi_group_create: bad argument (1..64 replicas)"; return MI_ERR_INVALID_ARG; }
  *out = nullptr;
  mi_group* G = new mi_group;
  const int rc = guarded([&] {
    std::map<int, int> rankOf;
    for (uint32_t r = 0; r < num_replicas; ++r) {
      mi_scene_desc d = *desc;
      d.device = devices[r];
      mi_scene* s = nullptr;
      if (mi_scene_create(&d, &s) != MI_OK) throw ArgError(std::string("mi_group_create: replica ") + std::to_string(r) + ": " + g_err);
      mi_group::Replica P;
      P.scene = s; P.device = devices[r];
      if (!rankOf.count(P.device)) { rankOf[P.device] = (int)G->commDevices.size(); G->commDevices.push_back(P.device); }
      P.rank = rankOf[P.device];
      HIP_CHECK(hipSetDevice(P.device));
      HIP_CHECK(hipStreamCreateWithFlags(&P.stream, hipStreamNonBlocking));
      HIP_CHECK(hipEventCreateWithFlags(&P.done, hipEventDisableTiming));
      G->replicas.push_back(P);
    }
    // transport: 0 = automatic (RCCL as soon as more than one device takes part), 1 = RCCL always (replicas that share
    // the root's device then send to themselves: the one-GPU rehearsal of the collective), 2 = peer copies only
    G->useRccl = num_replicas > 1 && (transport == 1 || (transport == 0 && G->commDevices.size() > 1));
    if (G->useRccl) {
      std::string why;
      if (!G->rccl.load(why)) throw DeviceError(why);
      G->comms.assign(G->commDevices.size(), nullptr);
      RCCL_CHECK(*G, G->rccl.commInitAll(G->comms.data(), (int)G->commDevices.size(), G->commDevices.data()));
    } else if (G->commDevices.size() > 1) {
      for (size_t i = 1; i < G->commDevices.size(); ++i) {
        HIP_CHECK(hipSetDevice(G->commDevices[0]));
        int can = 0;
        HIP_CHECK(hipDeviceCanAccessPeer(&can, G->commDevices[0], G->commDevices[i]));
        if (can) { const hipError_t e = hipDeviceEnablePeerAccess(G->commDevices[i], 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(e); (void)hipGetLastError(); }
      }
    }
  });
  if (rc != MI_OK) { delete G; return rc; }
  *out = G;
  return MI_OK;
}

void mi_group_destroy(mi_group* group) { delete group; }

uint32_t mi_group_size(const mi_group* group) { return group ? (uint32_t)group->replicas.size() : 0u; }

mi_scene* mi_group_scene(mi_group* group, uint32_t replica) {
  return (group && replica < group->replicas.size()) ? group->replicas[replica].scene : nullptr;
}

int mi_group_set_ray_batch(mi_group* group, size_t rays_per_batch) {
  if (!group) { g_err = "null group"; return MI_ERR_INVALID_ARG; }
  group->rayBatch = rays_per_batch;
  return MI_OK;
}

int mi_group_render(mi_group* group, int mode, mi_trace_result* rays, size_t n, mi_ray_callback cb, void* user) {
  if (!group || (!rays && n)) { g_err = "mi_group_render: null argument"; return MI_ERR_INVALID_ARG; }
  return guarded([&] { groupRender(*group, mode, rays, n, cb, user); });
}

double mi_group_trace_time_secs(const mi_group* group) { return group ? group->traceTimeSecs : 0.0; }

int mi_group_get_counters(mi_group* group, uint64_t counts[4]) {
  if (!group || !counts) { g_err = "mi_group_get_counters: null argument"; return MI_ERR_INVALID_ARG; }
  for (int i = 0; i < 4; ++i) counts[i] = 0;
  for (auto& P : group->replicas) {
    uint64_t c[4];
    const int rc = mi_get_counters(P.scene, c);
    if (rc != MI_OK) return rc;
    for (int i = 0; i < 4; ++i) counts[i] += c[i];
  }
  return MI_OK;
}

int mi_group_last_transfer(const mi_group* group, uint64_t info[3]) {
  if (!group || !info) { g_err = "mi_group_last_transfer: null argument"; return MI_ERR_INVALID_ARG; }
  info[0] = group->lastRcclMessages; info[1] = group->lastCopyMessages; info[2] = group->lastBands;
  return MI_OK;
}

}  // extern "C"
