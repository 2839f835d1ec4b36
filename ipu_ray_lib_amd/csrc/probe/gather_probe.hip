// gather_probe.hip — microbenchmark of the ONE memory operation K1w's walk is made of: a lane fetches the 32-byte
// device node (trace_kernels.hpp GNode) at an index that depends on the node it fetched before. No box test, no
// shading: only the gathers, with K1w's launch shape (256-thread workgroups, W workgroups per CU kept resident) and
// K1w's access shape (two 16-byte loads per node from one address register, N of a wave's 64 lanes active, the other
// lanes masked). It answers "what node-gather rate can this access shape attain on a CU?" for three data paths:
//   path 0  every node from global memory (vector L1 / TA path: global_load_dwordx4 x 2)
//   path 1  every node from an LDS copy (ds_read_b128 x 2; indices folded into the staged prefix)
//   path 2  mixed: node < P from LDS, the rest from global (both paths in one step, as a hot-prefix kernel has them)
// and two index sequences:
//   walk 0  uniformly random over the array (a dependent chain through the loaded words + a per-lane LCG)
//   walk 1  walk shaped: "descend" (i + 1) with probability 1/2 at interior nodes, otherwise the node's own `link`
//           (the skip pointer the real walk follows), back to the root at the end of the array - the locality of
//           a real traversal of THIS tree without its arithmetic.
// Not part of the product library: built by tools/gather_probe.py into build/probe/.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

namespace {

struct __attribute__((aligned(16))) Node32 { float minx, maxx, miny, maxy, minz, maxz; uint32_t link, leaf; };

template <int PATH, int WALK>
__global__ void __launch_bounds__(256) gather_probe_kernel(const Node32* __restrict__ nodes, uint32_t numNodes, uint32_t steps, uint32_t activeLanes,
                                                           uint32_t ldsNodes, uint32_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  if (PATH != 0) {
    const uint4* src = reinterpret_cast<const uint4*>(nodes);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (uint32_t k = threadIdx.x; k < ldsNodes * 2; k += blockDim.x) dst[k] = src[k];
    __syncthreads();
  }
  const uint32_t lane = threadIdx.x & 63u;
  // the active lanes are scattered over the wave the way a half-empty traversal step's are
  const bool active = ((lane * 37u) & 63u) < activeLanes;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t rnd = gid * 2654435761u + 12345u;
  uint32_t idx = __umulhi(rnd, numNodes);
  uint32_t acc = 0;
  if (active) {
    for (uint32_t s = 0; s < steps; ++s) {
      uint4 a, b;
      if (PATH == 1) {
        const uint32_t j = idx & (ldsNodes - 1u);          // (path 1 wants a power-of-two prefix)
        const uint4* p = reinterpret_cast<const uint4*>(lds + ((size_t)j << 5));
        a = p[0]; b = p[1];
      } else if (PATH == 2 && idx < ldsNodes) {
        const uint4* p = reinterpret_cast<const uint4*>(lds + ((size_t)idx << 5));
        a = p[0]; b = p[1];
      } else {
        const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(nodes) + ((size_t)idx << 5));
        a = p[0]; b = p[1];
      }
      acc += (a.x ^ a.y ^ a.z ^ a.w) + (b.x ^ b.y ^ b.z ^ b.w);      // every word is used: the loads stay two full 16-byte fetches
      rnd = rnd * 1664525u + 1013904223u;
      uint32_t next;
      if (WALK == 0) next = __umulhi((rnd ^ b.z ^ a.y) * 2654435761u, numNodes);  // b.z = link: the next index depends on the loaded node
      else next = ((rnd >> 31) && b.w == 0xFFFFFFFFu) ? idx + 1u : b.z;      // b.w = leaf marker
      idx = next >= numNodes ? 0u : next;
    }
  }
  if (acc == 0x9e3779b9u) sink[gid & 1023u] = acc;      // keeps the loads alive; practically never taken
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return -1.0; } } while (0)

template <int PATH, int WALK>
double run(const Node32* d_nodes, uint32_t numNodes, uint32_t steps, uint32_t activeLanes, uint32_t ldsNodes, uint32_t blocks, size_t ldsBytes, uint32_t* d_sink, int reps) {
  auto k = gather_probe_kernel<PATH, WALK>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
  struct Events {      // destroyed on every path out of this function
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } ev;
  CK(hipEventCreate(&ev.a)); CK(hipEventCreate(&ev.b));
  hipEvent_t e0 = ev.a, e1 = ev.b;
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), ldsBytes, 0, d_nodes, numNodes, steps, activeLanes, ldsNodes, d_sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), ldsBytes, 0, d_nodes, numNodes, steps, activeLanes, ldsNodes, d_sink);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

}  // namespace

// nodes: numNodes x 32 bytes (host). wgPerCU: resident 256-thread workgroups per CU (the LDS request is sized so that
// exactly that many fit: 5 = K1w's occupancy). Returns the average launch time in ms (< 0 on error).
extern "C" double gp_run(const void* nodes, uint32_t numNodes, int path, int walk, uint32_t steps, uint32_t activeLanes, uint32_t ldsNodes,
                          uint32_t wgPerCU, int reps, uint32_t* outBlocks) {
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  struct Buffers {     // freed on every path out of this function, the error returns included
    Node32* nodes = nullptr; uint32_t* sink = nullptr;
    ~Buffers() { if (nodes) (void)hipFree(nodes); if (sink) (void)hipFree(sink); }
  } buf;
  CK(hipMalloc(&buf.nodes, (size_t)numNodes * 32)); CK(hipMalloc(&buf.sink, 4096));
  Node32* const d_nodes = buf.nodes; uint32_t* const d_sink = buf.sink;
  CK(hipMemcpy(d_nodes, nodes, (size_t)numNodes * 32, hipMemcpyHostToDevice));
  // an LDS request of 160 KiB / (W + 1/2), rounded down to whole KiB: W workgroups fit a CU with room to spare, W + 1 do not
  const size_t ldsBytes = (size_t)(2 * 160 * 1024) / (2 * wgPerCU + 1) / 1024 * 1024;
  if (path != 0 && (size_t)ldsNodes * 32 > ldsBytes) { std::fprintf(stderr, "ldsNodes does not fit\n"); return -1.0; }
  const uint32_t blocks = (uint32_t)cus * wgPerCU;
  if (outBlocks) *outBlocks = blocks;
  double ms = -1.0;
  switch (path * 2 + walk) {
    case 0: ms = run<0, 0>(d_nodes, numNodes, steps, activeLanes, ldsNodes, blocks, ldsBytes, d_sink, reps); break;
    case 1: ms = run<0, 1>(d_nodes, numNodes, steps, activeLanes, ldsNodes, blocks, ldsBytes, d_sink, reps); break;
    case 2: ms = run<1, 0>(d_nodes, numNodes, steps, activeLanes, ldsNodes, blocks, ldsBytes, d_sink, reps); break;
    case 3: ms = run<1, 1>(d_nodes, numNodes, steps, activeLanes, ldsNodes, blocks, ldsBytes, d_sink, reps); break;
    case 4: ms = run<2, 0>(d_nodes, numNodes, steps, activeLanes, ldsNodes, blocks, ldsBytes, d_sink, reps); break;
    case 5: ms = run<2, 1>(d_nodes, numNodes, steps, activeLanes, ldsNodes, blocks, ldsBytes, d_sink, reps); break;
    default: break;
  }
  return ms;
}
