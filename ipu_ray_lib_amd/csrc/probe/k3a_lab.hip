// k3a_lab.hip — measurement tooling (tools/k3a_lab.py; not part of the product library): ONE placement of K3a's generated body
// (-DMI_NIF_ASM_BODY_INC=... names the text) beside K3 in a library of its own, so that several placements and timing-only
// knock-outs can be timed in interleaved rounds in one process on one box (cdna_hip_programming.md §5.4 rule 24).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../nif_asm_kernel.hpp"

using namespace mi;
#ifndef MI_LAB_WHICH
#define MI_LAB_WHICH 2u      // 2 = K3a's shape (8 waves x 2 ray tiles), 4 = K3b's (4 waves x 4): -DMI_LAB_WHICH=4u
#endif

struct Lab {
  NifDevice nif;
  float* u = nullptr; float* v = nullptr; float* out = nullptr; uint32_t n = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};

extern "C" void* lab_create(uint32_t numLayers, const float* const* kernels, const float* const* biases, const uint32_t* rows, const uint32_t* cols,
                            const uint8_t* relu, uint32_t embed, float maxValue, const float* mean, int32_t logTonemap, const float* hu, const float* hv, uint32_t n) {
  Lab* L = new Lab;
  try { L->nif.load(numLayers, kernels, biases, rows, cols, relu, embed, maxValue, mean, logTonemap); } catch (const std::exception& e) { std::fprintf(stderr, "lab: %s\n", e.what()); delete L; return nullptr; }
  L->n = n;
  if (hipMalloc(&L->u, n * 4) != hipSuccess || hipMalloc(&L->v, n * 4) != hipSuccess || hipMalloc(&L->out, (size_t)n * 12) != hipSuccess) return nullptr;
  (void)hipMemcpy(L->u, hu, n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(L->v, hv, n * 4, hipMemcpyHostToDevice);
  (void)hipEventCreate(&L->e0); (void)hipEventCreate(&L->e1);
  return L;
}

// which: 0 = K3 (w6), 1 = K3a (this library's placement). Returns the average launch time in ms, < 0 on error.
extern "C" double lab_time(void* h, int which, int reps, int numCUs) {
  Lab* L = (Lab*)h;
  if (which == 1 && !nif_asm_covers(L->nif.regs)) return -2.0;
  auto go = [&] {
    if (which == 1) nif_asm_launch(L->nif.regs, L->u, L->v, nullptr, nullptr, L->n, L->out, nullptr, 0, false, (uint32_t)numCUs, nullptr, MI_LAB_WHICH);
    else nif_launch_mlp(L->nif, L->u, L->v, nullptr, nullptr, L->n, L->out, nullptr, 0, false, 0, (uint32_t)numCUs);
  };
  try {
    go();
    if (hipDeviceSynchronize() != hipSuccess) return -1.0;
    (void)hipEventRecord(L->e0, 0);
    for (int r = 0; r < reps; ++r) go();
    (void)hipEventRecord(L->e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) return -1.0;
  } catch (const std::exception& e) { std::fprintf(stderr, "lab: %s\n", e.what()); return -1.0; }
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, L->e0, L->e1);
  return ms / reps;
}

// diagnostic builds (-DMI_K3A_STAMP): {shader cycles, 100-MHz ticks, shader cycles inside the bodies, passes} of workgroup 0's wave 0 in the last launch
extern "C" int lab_stamps(unsigned long long* out4) {
#if defined(MI_K3A_STAMP)
  return hipMemcpyFromSymbol(out4, HIP_SYMBOL(mi::k3a_stamp), 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
#else
  (void)out4; return 1;
#endif
}

extern "C" int lab_result(void* h, float* out) { Lab* L = (Lab*)h; return hipMemcpy(out, L->out, (size_t)L->n * 12, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }

extern "C" void lab_destroy(void* h) {
  Lab* L = (Lab*)h;
  if (!L) return;
  (void)hipFree(L->u); (void)hipFree(L->v); (void)hipFree(L->out); (void)hipEventDestroy(L->e0); (void)hipEventDestroy(L->e1);
  L->nif.release();
  delete L;
}
