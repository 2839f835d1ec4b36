// mfma_shape_probe.hip — K3's k-loop as a SKELETON, in the two MFMA shapes gfx950 offers for f16, to decide whether a
// v_mfma_f32_32x32x16_f16 formulation of the NIF MLP (half the MFMA issues and, per flop, half the B-fragment reads of the
// 16x16x32 one) can beat the shipped v_mfma_f32_16x16x32_f16 kernel. The skeleton keeps what a
// k-loop costs - per k-step the weight fragments stream from an L2-resident buffer (coalesced 1 KiB wave loads), the
// activation fragments are read from an LDS image with ds_read_b128, the accumulators stay in registers - and drops
// what does not depend on the shape (Fourier features, the epilogue's convert + store, decode). Random operands (the
// clock a kernel holds depends on the data: MI355X_MICROARCH.md, DVFS give-back). Compiler-scheduled on both sides.
//   shape 0: 16x16x32, 4 waves, wave = 5 feature tiles x 6 ray tiles  (80 x 96)  = K3 "w6" as shipped
//   shape 1: 32x32x16, 5 waves, wave = 2 feature tiles x 3 ray tiles  (64 x 96)  = the only even split of 320 features
//   shape 2: 32x32x16, 4 waves, wave = 3,3,2,2 feature tiles x 3 ray tiles       = 4 waves, uneven
// Layer = 320 x 320, `layers` of them per pass, 96 rays per workgroup, two workgroups per CU (72 KiB of LDS each).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

namespace {
constexpr uint32_t kFeat = 320, kRays = 96, kK = 320, kSteps = kK / 32;

// shape 0
__global__ void __launch_bounds__(256, 2) probe_16x16x32(const h8* __restrict__ w, uint32_t layers, uint32_t passes, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t k = threadIdx.x; k < kSteps * kRays * 64 / 16; k += blockDim.x) reinterpret_cast<uint4*>(lds)[k] = reinterpret_cast<const uint4*>(w)[k];
  __syncthreads();
  f4 acc[5][6];
  for (int a = 0; a < 5; ++a) for (int m = 0; m < 6; ++m) acc[a][m] = (f4){0.f, 0.f, 0.f, 0.f};
  for (uint32_t p = 0; p < passes; ++p)
    for (uint32_t l = 0; l < layers; ++l) {
      const h8* wl = w + (size_t)l * (kFeat / 16) * kSteps * 64;
      for (uint32_t ks = 0; ks < kSteps; ++ks) {
        h8 wf[5], xf[6];
#pragma unroll
        for (int a = 0; a < 5; ++a) wf[a] = wl[((wave + 4 * a) * kSteps + ks) * 64 + lane];
#pragma unroll
        for (int m = 0; m < 6; ++m) xf[m] = *reinterpret_cast<const h8*>(lds + ks * (kRays * 64) + m * 1024 + lane * 16);
#pragma unroll
        for (int m = 0; m < 6; ++m)
#pragma unroll
          for (int a = 0; a < 5; ++a) acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[m], acc[a][m], 0, 0, 0);
      }
      __syncthreads();
    }
  float s = 0.f;
  for (int a = 0; a < 5; ++a) for (int m = 0; m < 6; ++m) s += acc[a][m][0] + acc[a][m][1] + acc[a][m][2] + acc[a][m][3];
  if (s == 1234.5f) sink[threadIdx.x] = s;
}

// shapes 1, 2: TN feature tiles of 32 per wave (the wave's first tile = tile0), 3 ray tiles of 32
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 2) probe_32x32x16(const h8* __restrict__ w, uint32_t layers, uint32_t passes, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t k = threadIdx.x; k < kSteps * kRays * 64 / 16; k += blockDim.x) reinterpret_cast<uint4*>(lds)[k] = reinterpret_cast<const uint4*>(w)[k];
  __syncthreads();
  // 10 feature tiles of 32: 5 waves x 2, or 4 waves x {3, 3, 2, 2}
  const uint32_t tn = (WAVES == 5) ? 2u : (wave < 2 ? 3u : 2u);
  const uint32_t tile0 = (WAVES == 5) ? 2u * wave : (wave < 2 ? 3u * wave : 6u + 2u * (wave - 2u));
  f16v acc[3][3];
  for (int a = 0; a < 3; ++a) for (int m = 0; m < 3; ++m) for (int q = 0; q < 16; ++q) acc[a][m][q] = 0.f;
  for (uint32_t p = 0; p < passes; ++p)
    for (uint32_t l = 0; l < layers; ++l) {
      const h8* wl = w + (size_t)l * (kFeat / 32) * (2 * kSteps) * 64;
      for (uint32_t kh = 0; kh < 2 * kSteps; ++kh) {          // k-steps of 16
        h8 wf[3], xf[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) if (a < 2 || tn == 3) wf[a] = wl[((tile0 + a) * (2 * kSteps) + kh) * 64 + lane];
#pragma unroll
        for (int m = 0; m < 3; ++m) xf[m] = *reinterpret_cast<const h8*>(lds + (kh >> 1) * (kRays * 64) + (kh & 1) * 3072 + m * 1024 + lane * 16);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0], xf[m], acc[0][m], 0, 0, 0);
          acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1], xf[m], acc[1][m], 0, 0, 0);
          if (WAVES == 4 && tn == 3) acc[2][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[2], xf[m], acc[2][m], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  float s = 0.f;
  for (int a = 0; a < 3; ++a) for (int m = 0; m < 3; ++m) for (int q = 0; q < 16; ++q) s += acc[a][m][q];
  if (s == 1234.5f) sink[threadIdx.x] = s;
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return -1.0; } } while (0)
}  // namespace

// returns the average launch time in ms; *flops = floating-point operations of one launch
extern "C" double msp_run(int shape, uint32_t layers, uint32_t passes, int reps, double* flops, uint32_t* blocksOut) {
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const size_t halves = (size_t)layers * kFeat * kK + 64 * 1024;
  _Float16* d_w = nullptr; float* d_sink = nullptr;
  CK(hipMalloc(&d_w, halves * 2)); CK(hipMalloc(&d_sink, 4096));
  {
    // random halves in [-1, 1): xorshift on the host
    _Float16* h = (_Float16*)malloc(halves * 2);
    uint32_t x = 2463534242u;
    for (size_t i = 0; i < halves; ++i) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; h[i] = (_Float16)((float)(x & 0xFFFF) / 32768.f - 1.f); }
    CK(hipMemcpy(d_w, h, halves * 2, hipMemcpyHostToDevice));
    free(h);
  }
  const uint32_t blocks = (uint32_t)cus * 2;
  const size_t ldsBytes = 72 * 1024;
  if (blocksOut) *blocksOut = blocks;
  if (flops) *flops = (double)blocks * passes * layers * 2.0 * kFeat * kRays * kK;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (shape == 0) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe_16x16x32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
      hipLaunchKernelGGL(probe_16x16x32, dim3(blocks), dim3(256), ldsBytes, 0, reinterpret_cast<const h8*>(d_w), layers, passes, d_sink);
    } else if (shape == 1) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe_32x32x16<5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
      hipLaunchKernelGGL(probe_32x32x16<5>, dim3(blocks), dim3(320), ldsBytes, 0, reinterpret_cast<const h8*>(d_w), layers, passes, d_sink);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe_32x32x16<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
      hipLaunchKernelGGL(probe_32x32x16<4>, dim3(blocks), dim3(256), ldsBytes, 0, reinterpret_cast<const h8*>(d_w), layers, passes, d_sink);
    }
  };
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipFree(d_w); (void)hipFree(d_sink);
  return ms / reps;
}
