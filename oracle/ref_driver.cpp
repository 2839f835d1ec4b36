// ref_driver.cpp — extern "C" doorways into the REFERENCE's own code, for validating the
// oracle restatement and generating golden vectors (tests/golden/gen_ref_vectors.py).
//
// This file contains no ray-tracing logic of its own: every function forwards to a function
// defined in the reference checkout, compiled from where it lies (see oracle/Makefile, target
// `ref`). Only reference files that compile in this image without any stand-in header are
// reachable: ext/math/sincos.cpp, include/xoshiro.hpp, include/embree_utils/geometry.hpp,
// include/geometric_sampling.hpp, include/BxDF.hpp. Everything that includes
// include/precision_utils.hpp needs Eigen::half (absent here) and is NOT built.
// Outputs go to oracle/_ref/ (git-ignored); never shipped as product.

#include <cstdint>
#include <cstring>

#include <embree_utils/geometry.hpp>
#include <xoshiro.hpp>
#include <BxDF.hpp>          // pulls geometric_sampling.hpp and math/sincos.hpp

using embree_utils::Vec3fa;

extern "C" {

void ref_sincos(float x, float* s, float* c) { sincos(x, *s, *c); }

uint32_t ref_maxi(float x, float y, float z) { return Vec3fa(x, y, z).maxi(); }
float ref_maxc(float x, float y, float z) { return Vec3fa(x, y, z).maxc(); }

void ref_normalized(const float* v, float* out) {
  Vec3fa r = Vec3fa(v[0], v[1], v[2]).normalized();
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float ref_dot(const float* a, const float* b) { return Vec3fa(a[0], a[1], a[2]).dot(Vec3fa(b[0], b[1], b[2])); }
void ref_cross(const float* a, const float* b, float* out) {
  Vec3fa r = Vec3fa(a[0], a[1], a[2]).cross(Vec3fa(b[0], b[1], b[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

void ref_orthonormal_system(const float* n, float* b0, float* b1) {
  auto [x, y, z] = Vec3fa(n[0], n[1], n[2]).orthonormalSystem();
  b0[0] = x.x; b0[1] = x.y; b0[2] = x.z;
  b1[0] = y.x; b1[1] = y.y; b1[2] = y.z;
}

uint64_t ref_splitmix64(uint64_t z) { return xoshiro::splitmix64(z); }
void ref_xoshiro_seed(uint64_t* s, uint64_t seed) {
  xoshiro::State st; xoshiro::seed(st, seed); s[0] = st[0]; s[1] = st[1];
}
uint64_t ref_xoshiro_next(uint64_t* s) {
  xoshiro::State st{s[0], s[1]}; uint64_t r = xoshiro::next128ss(st); s[0] = st[0]; s[1] = st[1]; return r;
}
void ref_xoshiro_jump(uint64_t* s) {
  xoshiro::State st{s[0], s[1]}; xoshiro::jump(st); s[0] = st[0]; s[1] = st[1];
}
float ref_xoshiro_uniform01(uint64_t* s) {
  xoshiro::State st{s[0], s[1]}; float r = xoshiro::uniform_0_1(st); s[0] = st[0]; s[1] = st[1]; return r;
}

void ref_sample_disc_concentric(float u1, float u2, float* x, float* y) {
  auto p = sampleDiscConcentric(u1, u2); *x = p.first; *y = p.second;
}
void ref_cosine_sample_hemisphere(float u1, float u2, float* out) {
  Vec3fa r = cosineSampleHemisphere(u1, u2); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_sample_diffuse(const float* n, float u1, float u2, float* out) {
  Vec3fa r = sampleDiffuse(Vec3fa(n[0], n[1], n[2]), u1, u2); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_reflect(const float* d, const float* n, float* out) {
  Vec3fa r = reflect(Vec3fa(d[0], d[1], d[2]), Vec3fa(n[0], n[1], n[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float ref_schlick(float cosTheta, float ri) { return schlick(cosTheta, ri); }
void ref_refract(const float* d, const float* n, float ndotr, float ri, float* out) {
  Vec3fa r = refract(Vec3fa(d[0], d[1], d[2]), Vec3fa(n[0], n[1], n[2]), ndotr, ri);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int ref_dielectric(const float* o, const float* d, const float* n, float ri, float u1, float* out) {
  embree_utils::Ray ray(Vec3fa(o[0], o[1], o[2]), Vec3fa(d[0], d[1], d[2]));
  auto res = dielectric(ray, Vec3fa(n[0], n[1], n[2]), ri, u1);
  out[0] = res.first.x; out[1] = res.first.y; out[2] = res.first.z;
  return res.second ? 1 : 0;
}
int ref_evaluate_roulette(float u1, float* tp) {
  Vec3fa t(tp[0], tp[1], tp[2]);
  bool stop = evaluateRoulette(u1, t);
  tp[0] = t.x; tp[1] = t.y; tp[2] = t.z;
  return stop ? 1 : 0;
}

// struct layout facts of the reference types that do compile here
void ref_layout(uint32_t* out) {
  out[0] = sizeof(Vec3fa); out[1] = alignof(Vec3fa);
  out[2] = sizeof(embree_utils::Ray); out[3] = sizeof(embree_utils::HitRecord);
  out[4] = sizeof(embree_utils::TraceResult);
  out[5] = offsetof(embree_utils::TraceResult, p); out[6] = offsetof(embree_utils::TraceResult, h);
  out[7] = offsetof(embree_utils::HitRecord, primID); out[8] = offsetof(embree_utils::HitRecord, normal);
  out[9] = offsetof(embree_utils::HitRecord, throughput); out[10] = offsetof(embree_utils::HitRecord, geomID);
  out[11] = offsetof(embree_utils::HitRecord, flags);
}

} // extern "C"
