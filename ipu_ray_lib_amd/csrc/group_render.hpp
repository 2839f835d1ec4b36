// group_render.hpp — one process, several GPUs: the multi-device renderer behind mi_group_* (include/mi_raylib.h).
// Included at the end of raylib.hip (it uses mi_scene and launchRender of that translation unit).
//
// Reference: an IpuScene with numReplicas > 1 replicates the scene on every device (src/IpuScene.cpp:473-483), lets
// the replicas pull disjoint ray batches round-robin from the one host stream (:676-684), writes every batch back
// into the caller's stream (:699-732) and hands it to the callback as soon as it is home (src/RayCallback.cpp:8-24).
// Here:
//   * one mi_scene per replica, each on its device (several replicas may share a device: that is how the path is
//     rehearsed on a one-GPU box);
//   * a BATCH of the stream (the whole stream unless mi_group_set_ray_batch cut it) is dealt in bands
//     (ray_shard.hpp: 8 window rows per band, band b to replica b % R). A replica's bands sit one stride apart in the
//     host stream, so its share goes up as ONE strided copy (hipMemcpy2DAsync: rows = bands) plus at most one short
//     tail band; every replica traces its share on its own HIP stream - no exchange while the batch renders;
//   * then ONE RCCL group call moves every replica's finished share to the root device over xGMI (ncclSend on the
//     replica's stream / ncclRecv on the root's stream, point to point: each peer uses its own direct link to the
//     root; a ring collective would be bound by one link);
//   * the gathered shares leave the root with one strided copy per replica straight into their places in the
//     caller's stream - the copy engine restores stream order, there is no de-interleave pass and no second frame
//     buffer on the root;
//   * the callback of batch b runs on the calling thread while batch b + 1 is being traced.
// The three stages are also exported one by one (mi_group_upload / mi_group_trace / mi_group_download): a caller that
// keeps the shares resident renders frame after frame with no host traffic, which is what bench.py times.
// RCCL is reached through dlopen (librccl.so.1): the library has no link-time dependency on it, a process that never
// builds a group never loads it, and a host process that already carries an RCCL (torch) shares that copy.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <map>
#include <memory>

#include "ray_shard.hpp"

namespace mi {

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
    mi_trace_result* d_share = nullptr; size_t shareCap = 0;      // replica 0: the head of d_gather (never freed through here)
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;                 // the replica's share is traced
  };
  // how one batch of the stream is dealt: replica r renders count[r] rays, its share lands at offset[r] of the gathered buffer
  struct Plan { size_t n = 0, band = 0; std::vector<size_t> count, offset; };
  std::vector<Replica> replicas;
  std::vector<int> commDevices;                // distinct devices, root first
  std::vector<ncclComm_t> comms;               // one per distinct device (ncclCommInitAll)
  std::vector<hipEvent_t> sent;                // per distinct device: its sends of the last gather have left
  mi::RcclApi rccl;
  bool useRccl = false;
  mi_trace_result* d_gather = nullptr; size_t gatherCap = 0;      // on the root device: the shares, replica after replica
  hipEvent_t gathered = nullptr;               // root stream: the last gather (and, when one followed, its download) is done
  hipEvent_t gatherBegin = nullptr, gatherEnd = nullptr;      // root stream, timing: round the last batch's group call / peer copies
  bool gatherPending = false;
  Plan resident;                               // what mi_group_upload left on the devices (n = 0: nothing)
  bool residentTraced = false;                 // ... and whether it has been traced and gathered since
  size_t rayBatch = 0;
  double traceTimeSecs = 0.0;
  // what the last render did (for tests and logs)
  uint64_t lastRcclMessages = 0, lastCopyMessages = 0, lastBands = 0, lastUploadCopies = 0, lastDownloadCopies = 0;

  ~mi_group() {
    for (size_t i = 0; i < comms.size(); ++i) if (comms[i]) { (void)hipSetDevice(commDevices[i]); (void)rccl.commDestroy(comms[i]); }
    for (size_t i = 0; i < sent.size(); ++i) if (sent[i]) { (void)hipSetDevice(commDevices[i]); (void)hipEventDestroy(sent[i]); }
    for (size_t r = 0; r < replicas.size(); ++r) {
      Replica& P = replicas[r];
      (void)hipSetDevice(P.device);
      if (r != 0 && P.d_share) (void)hipFree(P.d_share);
      if (P.stream) (void)hipStreamDestroy(P.stream);
      if (P.done) (void)hipEventDestroy(P.done);
      delete P.scene;
    }
    if (!replicas.empty()) (void)hipSetDevice(replicas[0].device);
    if (d_gather) (void)hipFree(d_gather);
    if (gathered) (void)hipEventDestroy(gathered);
    if (gatherBegin) (void)hipEventDestroy(gatherBegin);
    if (gatherEnd) (void)hipEventDestroy(gatherEnd);
  }
};

namespace {

#define RCCL_CHECK(g, expr)                                                                                   \
  do {                                                                                                        \
    ncclResult_t _r = (expr);                                                                                 \
    if (_r != ncclSuccess) throw DeviceError(std::string(#expr) + ": " + (g).rccl.errorString(_r));          \
  } while (0)

constexpr size_t kRec = sizeof(mi_trace_result);

mi_group::Plan groupPlan(const mi_group& G, size_t n) {
  const uint32_t R = (uint32_t)G.replicas.size();
  mi_group::Plan pl;
  pl.n = n;
  pl.band = shard::band_rays(n, (uint32_t)std::max(G.replicas[0].scene->params.window_w, 0));
  pl.count.resize(R); pl.offset.assign(R + 1, 0);
  for (uint32_t r = 0; r < R; ++r) { pl.count[r] = shard::replica_count(n, pl.band, R, r); pl.offset[r + 1] = pl.offset[r] + pl.count[r]; }
  return pl;
}

// every replica's share buffer on its device; on the root the gathered shares (the root replica traces in place at
// their head)
void groupEnsureBuffers(mi_group& G, const mi_group::Plan& pl) {
  mi_group::Replica& root = G.replicas[0];
  HIP_CHECK(hipSetDevice(root.device));
  if (G.gatherCap < pl.n) {
    HIP_CHECK(hipDeviceSynchronize());
    if (G.d_gather) (void)hipFree(G.d_gather);
    G.d_gather = nullptr; G.gatherCap = 0; root.d_share = nullptr; root.shareCap = 0;
    G.resident.n = 0;
    HIP_CHECK(hipMalloc(&G.d_gather, pl.n * kRec));
    G.gatherCap = pl.n;
  }
  root.d_share = G.d_gather; root.shareCap = G.gatherCap;
  for (size_t r = 1; r < G.replicas.size(); ++r) {
    mi_group::Replica& P = G.replicas[r];
    if (P.shareCap >= pl.count[r] && P.d_share) continue;
    HIP_CHECK(hipSetDevice(P.device));
    if (P.d_share) { HIP_CHECK(hipDeviceSynchronize()); (void)hipFree(P.d_share); }
    P.d_share = nullptr; P.shareCap = 0;
    G.resident.n = 0;
    HIP_CHECK(hipMalloc(&P.d_share, std::max<size_t>(pl.count[r], 1) * kRec));
    P.shareCap = std::max<size_t>(pl.count[r], 1);
  }
}

// Replica r's bands r, r + R, ... of a batch that starts at `rays`: full bands as the rows of one strided copy, a
// short last band (only the batch's last band can be short) as a copy of its own. toDevice: host -> P's share on
// `stream`; otherwise gathered share on the root -> host.
uint64_t groupCopyShare(const mi_group::Plan& pl, uint32_t R, uint32_t r, mi_trace_result* rays, mi_trace_result* d_share, bool toDevice, hipStream_t stream) {
  const size_t B = shard::num_bands(pl.n, pl.band);
  const size_t mine = (B > r) ? (B - 1 - r) / R + 1 : 0;
  if (mine == 0) return 0;
  const size_t lastLen = pl.n - (B - 1) * pl.band;                          // rays in the batch's last band
  const bool ownsShortLast = ((B - 1) % R == r) && lastLen < pl.band;
  const size_t full = ownsShortLast ? mine - 1 : mine;
  uint64_t copies = 0;
  mi_trace_result* h0 = rays + (size_t)r * pl.band;
  if (full) {
    if (toDevice) HIP_CHECK(hipMemcpy2DAsync(d_share, pl.band * kRec, h0, (size_t)R * pl.band * kRec, pl.band * kRec, full, hipMemcpyHostToDevice, stream));
    else HIP_CHECK(hipMemcpy2DAsync(h0, (size_t)R * pl.band * kRec, d_share, pl.band * kRec, pl.band * kRec, full, hipMemcpyDeviceToHost, stream));
    ++copies;
  }
  if (ownsShortLast) {
    mi_trace_result* h = rays + (B - 1) * pl.band; mi_trace_result* dv = d_share + full * pl.band;
    if (toDevice) HIP_CHECK(hipMemcpyAsync(dv, h, lastLen * kRec, hipMemcpyHostToDevice, stream));
    else HIP_CHECK(hipMemcpyAsync(h, dv, lastLen * kRec, hipMemcpyDeviceToHost, stream));
    ++copies;
  }
  return copies;
}

// A share may only be overwritten or traced on once the gather that read it has left the device (the gathered buffer
// itself is only touched on the root's stream, which orders its download against the next trace and receive).
void groupWaitGather(mi_group& G, mi_group::Replica& P) {
  if (!G.gatherPending) return;
  HIP_CHECK(hipStreamWaitEvent(P.stream, G.gathered, 0));
  if (G.useRccl && G.sent[P.rank]) HIP_CHECK(hipStreamWaitEvent(P.stream, G.sent[P.rank], 0));
}

void groupStageUpload(mi_group& G, const mi_group::Plan& pl, mi_trace_result* rays) {
  const uint32_t R = (uint32_t)G.replicas.size();
  G.lastBands += shard::num_bands(pl.n, pl.band);
  for (uint32_t r = 0; r < R; ++r) {
    mi_group::Replica& P = G.replicas[r];
    HIP_CHECK(hipSetDevice(P.device));
    groupWaitGather(G, P);
    G.lastUploadCopies += groupCopyShare(pl, R, r, rays, P.d_share, true, P.stream);
  }
}

void groupStageTrace(mi_group& G, const mi_group::Plan& pl, int mode) {
  for (size_t r = 0; r < G.replicas.size(); ++r) {
    mi_group::Replica& P = G.replicas[r];
    HIP_CHECK(hipSetDevice(P.device));
    groupWaitGather(G, P);
    launchRender(*P.scene, mode, P.d_share, pl.count[r], P.stream);
    if (r != 0) HIP_CHECK(hipEventRecord(P.done, P.stream));
  }
}

// the ONE collective of a batch: every replica's share to the root
void groupStageGather(mi_group& G, const mi_group::Plan& pl) {
  const uint32_t R = (uint32_t)G.replicas.size();
  mi_group::Replica& root = G.replicas[0];
  // (timing events on the root's stream: `gatherBegin` fires when the root's own share is traced, `gatherEnd` when the
  // last share has arrived - the gather as the frame sees it, a slower peer's remaining trace time included)
  HIP_CHECK(hipSetDevice(root.device));
  HIP_CHECK(hipEventRecord(G.gatherBegin, root.stream));
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
      try {
        for (uint32_t r = 1; r < R; ++r) {
          mi_group::Replica& P = G.replicas[r];
          if (pl.count[r] == 0) continue;
          const size_t bytes = pl.count[r] * kRec;
          RCCL_CHECK(G, G.rccl.send(P.d_share, bytes, ncclUint8, root.rank, G.comms[P.rank], commStream[P.rank]));
          RCCL_CHECK(G, G.rccl.recv(G.d_gather + pl.offset[r], bytes, ncclUint8, P.rank, G.comms[root.rank], root.stream));
          ++G.lastRcclMessages;
        }
      } catch (...) { (void)G.rccl.groupEnd(); throw; }         // never leave the group call open
      RCCL_CHECK(G, G.rccl.groupEnd());
      for (size_t i = 0; i < G.commDevices.size(); ++i) {
        if (!commStream[i]) continue;
        HIP_CHECK(hipSetDevice(G.commDevices[i]));
        if (!G.sent[i]) HIP_CHECK(hipEventCreateWithFlags(&G.sent[i], hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(G.sent[i], commStream[i]));
      }
    } else {
      HIP_CHECK(hipSetDevice(root.device));
      for (uint32_t r = 1; r < R; ++r) {
        mi_group::Replica& P = G.replicas[r];
        if (pl.count[r] == 0) continue;
        HIP_CHECK(hipStreamWaitEvent(root.stream, P.done, 0));
        HIP_CHECK(hipMemcpyPeerAsync(G.d_gather + pl.offset[r], root.device, P.d_share, P.device, pl.count[r] * kRec, root.stream));
        ++G.lastCopyMessages;
      }
    }
  }
  HIP_CHECK(hipSetDevice(root.device));
  HIP_CHECK(hipEventRecord(G.gatherEnd, root.stream));
  HIP_CHECK(hipEventRecord(G.gathered, root.stream));
  G.gatherPending = true;
}

// the gathered shares home, each into its bands of the caller's stream
void groupStageDownload(mi_group& G, const mi_group::Plan& pl, mi_trace_result* rays) {
  const uint32_t R = (uint32_t)G.replicas.size();
  mi_group::Replica& root = G.replicas[0];
  HIP_CHECK(hipSetDevice(root.device));
  // (root stream: behind the gather that filled the buffer and ahead of whatever the root traces or receives next)
  for (uint32_t r = 0; r < R; ++r) G.lastDownloadCopies += groupCopyShare(pl, R, r, rays, G.d_gather + pl.offset[r], false, root.stream);
}

void groupSyncAll(mi_group& G) {
  for (auto& P : G.replicas) { (void)hipSetDevice(P.device); HIP_CHECK(hipStreamSynchronize(P.stream)); }
}

void groupResetLog(mi_group& G) { G.lastRcclMessages = G.lastCopyMessages = G.lastBands = G.lastUploadCopies = G.lastDownloadCopies = 0; }

struct PinGuard {
  void* p = nullptr;
  PinGuard(const mi_group& G, void* rays, size_t bytes) {
    if (!G.replicas[0].scene->opt.pin || bytes < ((size_t)1 << 20)) return;
    hipPointerAttribute_t attr{};
    const bool known = hipPointerGetAttributes(&attr, rays) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (known) return;
    (void)hipGetLastError();
    if (hipHostRegister(rays, bytes, hipHostRegisterPortable) == hipSuccess) p = rays; else (void)hipGetLastError();
  }
  ~PinGuard() { if (p) (void)hipHostUnregister(p); }
};

void groupRender(mi_group& G, int mode, mi_trace_result* rays, size_t n, mi_ray_callback cb, void* user) {
  if (n == 0) { G.traceTimeSecs = 0.0; return; }
  mi_group::Replica& root = G.replicas[0];
  // Ray batches (src/IpuScene.cpp:110-172, 585-618; RayCallback::fetch, src/RayCallback.cpp:8-24): each batch is dealt,
  // traced, gathered and brought home on its own; the callback of a batch runs while the next one is in flight.
  const size_t batch = (G.rayBatch && G.rayBatch < n) ? G.rayBatch : n;
  const size_t numBatches = (n + batch - 1) / batch;
  {
    // buffers for every batch of this render: all batches but the last have `batch` rays; the last one is shorter and -
    // when only one of the two sizes is made of whole window rows - may be dealt in bands of another size, so a replica
    // can get MORE rays of the short batch than of a full one
    mi_group::Plan need = groupPlan(G, batch);
    const size_t lastSize = n - (numBatches - 1) * batch;
    if (lastSize != batch) {
      const mi_group::Plan last = groupPlan(G, lastSize);
      for (size_t r = 0; r < need.count.size(); ++r) need.count[r] = std::max(need.count[r], last.count[r]);
    }
    groupEnsureBuffers(G, need);
  }
  G.resident.n = 0;                    // the shares are about to be overwritten
  PinGuard pin(G, rays, n * kRec);
  groupResetLog(G);
  try {
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<hipEvent_t> home(2, nullptr);
    HIP_CHECK(hipSetDevice(root.device));
    for (auto& e : home) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    auto finish = [&](size_t b) {
      HIP_CHECK(hipEventSynchronize(home[b & 1]));
      const size_t first = b * batch;
      if (cb) cb(user, b, rays + first, std::min(batch, n - first));
    };
    try {
      for (size_t b = 0; b < numBatches; ++b) {
        const size_t first = b * batch, cnt = std::min(batch, n - first);
        const mi_group::Plan pl = groupPlan(G, cnt);
        groupStageUpload(G, pl, rays + first);
        groupStageTrace(G, pl, mode);
        groupStageGather(G, pl);
        groupStageDownload(G, pl, rays + first);
        HIP_CHECK(hipSetDevice(root.device));
        HIP_CHECK(hipEventRecord(home[b & 1], root.stream));
        if (b >= 1) finish(b - 1);
      }
      finish(numBatches - 1);
      groupSyncAll(G);
    } catch (...) { for (auto& e : home) (void)hipEventDestroy(e); throw; }
    for (auto& e : home) (void)hipEventDestroy(e);
    G.traceTimeSecs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    root.scene->traceTimeSecs = G.traceTimeSecs;
  } catch (...) {
    for (auto& P : G.replicas) { (void)hipSetDevice(P.device); (void)hipDeviceSynchronize(); }
    throw;
  }
}

}  // namespace

extern "C" {

int mi_group_create(const mi_scene_desc* desc, const int32_t* devices, uint32_t num_replicas, int32_t transport, mi_group** out) {
  if (!desc || !devices || !out || num_replicas == 0 || num_replicas > 64) { g_err = "mi_group_create: bad argument (1..64 replicas)"; return MI_ERR_INVALID_ARG; }
  *out = nullptr;
  mi_group* G = nullptr;
  const int rc = guarded([&] {
    G = new mi_group;
    std::map<int, int> rankOf;
    for (uint32_t r = 0; r < num_replicas; ++r) {
      mi_scene_desc d = *desc;
      d.device = devices[r];
      mi_scene* s = nullptr;
      if (mi_scene_create(&d, &s) != MI_OK) throw ArgError(std::string("mi_group_create: replica ") + std::to_string(r) + ": " + g_err);
      // the group owns the replica from here on: whatever fails below, ~mi_group releases it
      G->replicas.emplace_back();
      mi_group::Replica& P = G->replicas.back();
      P.scene = s; P.device = devices[r];
      if (!rankOf.count(P.device)) { rankOf[P.device] = (int)G->commDevices.size(); G->commDevices.push_back(P.device); }
      P.rank = rankOf[P.device];
      HIP_CHECK(hipSetDevice(P.device));
      HIP_CHECK(hipStreamCreateWithFlags(&P.stream, hipStreamNonBlocking));
      HIP_CHECK(hipEventCreateWithFlags(&P.done, hipEventDisableTiming));
    }
    G->sent.assign(G->commDevices.size(), nullptr);
    HIP_CHECK(hipSetDevice(G->replicas[0].device));
    HIP_CHECK(hipEventCreateWithFlags(&G->gathered, hipEventDisableTiming));
    HIP_CHECK(hipEventCreate(&G->gatherBegin));
    HIP_CHECK(hipEventCreate(&G->gatherEnd));
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
  if (rc != MI_OK) { const std::string keep = g_err; delete G; g_err = keep; return rc; }
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

int mi_group_upload(mi_group* group, const mi_trace_result* rays, size_t n) {
  if (!group || !rays || n == 0) { g_err = "mi_group_upload: null argument or empty stream"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    mi_group& G = *group;
    const mi_group::Plan pl = groupPlan(G, n);
    groupEnsureBuffers(G, pl);
    G.resident.n = 0;
    PinGuard pin(G, const_cast<mi_trace_result*>(rays), n * kRec);
    groupResetLog(G);
    try {
      groupStageUpload(G, pl, const_cast<mi_trace_result*>(rays));       // (only read: toDevice)
      groupSyncAll(G);
    } catch (...) { for (auto& P : G.replicas) { (void)hipSetDevice(P.device); (void)hipDeviceSynchronize(); } throw; }
    G.resident = pl; G.residentTraced = false;
  });
}

int mi_group_trace(mi_group* group, int mode) {
  if (!group) { g_err = "mi_group_trace: null group"; return MI_ERR_INVALID_ARG; }
  if (group->resident.n == 0) { g_err = "mi_group_trace: no resident ray stream (call mi_group_upload first)"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    mi_group& G = *group;
    const auto t0 = std::chrono::steady_clock::now();
    G.lastRcclMessages = G.lastCopyMessages = 0;
    try {
      groupStageTrace(G, G.resident, mode);
      groupStageGather(G, G.resident);
      groupSyncAll(G);
      G.residentTraced = true;
    } catch (...) { for (auto& P : G.replicas) { (void)hipSetDevice(P.device); (void)hipDeviceSynchronize(); } throw; }
    G.traceTimeSecs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    G.replicas[0].scene->traceTimeSecs = G.traceTimeSecs;
  });
}

int mi_group_download(mi_group* group, mi_trace_result* rays, size_t n) {
  if (!group || !rays) { g_err = "mi_group_download: null argument"; return MI_ERR_INVALID_ARG; }
  if (group->resident.n == 0 || group->resident.n != n) { g_err = "mi_group_download: the stream size differs from the resident one"; return MI_ERR_INVALID_ARG; }
  if (!group->residentTraced) { g_err = "mi_group_download: nothing has been traced since the upload"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    mi_group& G = *group;
    PinGuard pin(G, rays, n * kRec);
    G.lastDownloadCopies = 0;
    try {
      groupStageDownload(G, G.resident, rays);
      groupSyncAll(G);
    } catch (...) { for (auto& P : G.replicas) { (void)hipSetDevice(P.device); (void)hipDeviceSynchronize(); } throw; }
  });
}

int mi_group_gathered_device(mi_group* group, void** d_gathered, uint64_t* offsets, uint32_t num_offsets) {
  if (!group || !d_gathered) { g_err = "mi_group_gathered_device: null argument"; return MI_ERR_INVALID_ARG; }
  if (group->resident.n == 0) { g_err = "mi_group_gathered_device: no resident ray stream"; return MI_ERR_INVALID_ARG; }
  *d_gathered = group->d_gather;
  for (uint32_t r = 0; offsets && r < num_offsets && r < group->resident.offset.size(); ++r) offsets[r] = group->resident.offset[r];
  return MI_OK;
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

int mi_group_reset_counters(mi_group* group) {
  if (!group) { g_err = "mi_group_reset_counters: null group"; return MI_ERR_INVALID_ARG; }
  for (auto& P : group->replicas) { const int rc = mi_reset_counters(P.scene); if (rc != MI_OK) return rc; }
  return MI_OK;
}

int mi_group_last_gather_ms(mi_group* group, double* ms) {
  if (!group || !ms) { g_err = "mi_group_last_gather_ms: null argument"; return MI_ERR_INVALID_ARG; }
  if (!group->gatherPending) { g_err = "mi_group_last_gather_ms: nothing has been gathered yet"; return MI_ERR_INVALID_ARG; }
  return guarded([&] {
    HIP_CHECK(hipSetDevice(group->replicas[0].device));
    HIP_CHECK(hipEventSynchronize(group->gatherEnd));
    float f = 0.f;
    HIP_CHECK(hipEventElapsedTime(&f, group->gatherBegin, group->gatherEnd));
    *ms = f;
  });
}

int mi_group_devices(const mi_group* group, int32_t* devices, uint32_t capacity, uint32_t* distinct) {
  if (!group || !distinct) { g_err = "mi_group_devices: null argument"; return MI_ERR_INVALID_ARG; }
  *distinct = (uint32_t)group->commDevices.size();
  for (uint32_t i = 0; devices && i < capacity && i < group->commDevices.size(); ++i) devices[i] = group->commDevices[i];
  return MI_OK;
}

int mi_group_last_transfer(const mi_group* group, uint64_t info[5]) {
  if (!group || !info) { g_err = "mi_group_last_transfer: null argument"; return MI_ERR_INVALID_ARG; }
  info[0] = group->lastRcclMessages; info[1] = group->lastCopyMessages; info[2] = group->lastBands;
  info[3] = group->lastUploadCopies; info[4] = group->lastDownloadCopies;
  return MI_OK;
}

}  // extern "C"
