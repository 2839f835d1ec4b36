// ray_shard.hpp — how a ray stream is dealt to the replicas of a multi-GPU render, and how the frame is put
// together again. Header-only, host and device: the C ABI of libmi_scene_host.so (mi_shard_*, used by
// the Python ranks of bench.py through ipu_ray_lib_amd/sharding.py), the single-process renderer mi_group_render in
// libmi_raylib.so (whose strided share copies follow this dealing) all share these few functions, so there is one definition of who
// renders what.
//
// Reference: the replicas of an IpuScene pull disjoint ray batches round-robin from one stream
// (src/IpuScene.cpp:676-684; batch index = receiveIndex + replica, src/RayCallback.cpp:8-24) and the scene is
// replicated on every device (src/IpuScene.cpp:473-483). Here the batch is a BAND of `band` consecutive rays -
// 8 rows of the render window when the stream is made of full rows (a band is then one row of the kernel's 8x8 work
// tiles) - and band b belongs to replica b % R. Rays are independent and every pixel owns its RNG streams, so the
// assembled frame does not depend on R.
#pragma once

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MI_SHARD_HD __host__ __device__ inline
#else
#define MI_SHARD_HD inline
#endif

namespace mi::shard {

constexpr size_t kBandRows = 8;
constexpr size_t kBandRaysUnstructured = 4096;

// rays per band for a stream of n rays rendered for a window `window_w` pixels wide
MI_SHARD_HD size_t band_rays(size_t n, uint32_t window_w) {
  if (window_w > 0 && n % window_w == 0) return kBandRows * (size_t)window_w;
  return kBandRaysUnstructured;
}
MI_SHARD_HD size_t num_bands(size_t n, size_t band) { return band ? (n + band - 1) / band : 0; }
// rays replica r renders (only the stream's last band can be short)
MI_SHARD_HD size_t replica_count(size_t n, size_t band, uint32_t replicas, uint32_t r) {
  const size_t B = num_bands(n, band);
  if (B == 0 || r >= replicas) return 0;
  const size_t mine = (B > r) ? (B - 1 - r) / replicas + 1 : 0;          // bands r, r + R, ...
  if (mine == 0) return 0;
  const bool ownsLast = (B - 1) % replicas == r;
  return mine * band - (ownsLast ? (B * band - n) : 0);
}
// stream position i -> (replica, position in that replica's stream)
MI_SHARD_HD void locate(size_t band, uint32_t replicas, size_t i, uint32_t& replica, size_t& pos) {
  const size_t b = i / band;
  replica = (uint32_t)(b % replicas);
  pos = (b / replicas) * band + (i - b * band);
}

}  // namespace mi::shard
