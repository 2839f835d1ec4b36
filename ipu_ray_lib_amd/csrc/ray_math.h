// ray_math.h — the product's ray-math core, shared by the host plumbing (g++) and the gfx950
// kernels (hipcc). Every function is a single-rounding-per-operation binary32 computation: the
// library is built with -ffp-contract=off and no fast-math so that host, device and the
// reference's CPU path agree bit for bit.
//
// Reference counterparts are cited per function (paths relative to the reference checkout).
#pragma once

#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define MI_HD __host__ __device__ __forceinline__
#else
#define MI_HD inline
#endif

namespace mi {

struct f3 { float x, y, z; };

MI_HD f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
MI_HD f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
MI_HD f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
MI_HD f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
MI_HD f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
MI_HD f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
MI_HD f3 abs3(f3 a) { return mk(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
// embree_utils/geometry.hpp:135-143 — evaluation order is part of the contract
MI_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MI_HD float sqnorm(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
MI_HD f3 normalized(f3 a) { return a * (1.f / sqrtf(sqnorm(a))); }
MI_HD f3 cross(f3 a, f3 v) { return mk(a.y * v.z - a.z * v.y, a.z * v.x - a.x * v.z, a.x * v.y - a.y * v.x); }
MI_HD float comp(f3 a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// geometry.hpp:115-125. Named maxi/maxc upstream; they select the SMALLEST component and
// every caller depends on that.
// Written as two compare / select pairs: "x < y ? (x < z ? x : z) : (y < z ? y : z)" is "m = x < y ? x : y; m < z ? m : z" -
// the SAME comparisons on the same operands in either branch (NaN behaviour included), but hipcc evaluates all three
// compares of the nested form (the triangle test takes four of these per primitive).
MI_HD uint32_t min_index(f3 v) {
  const bool xy = v.x < v.y;
  const float m = xy ? v.x : v.y;
  return (m < v.z) ? (xy ? 0u : 1u) : 2u;
}
MI_HD float min_comp(f3 v) {
  const float m = (v.x < v.y) ? v.x : v.y;
  return (m < v.z) ? m : v.z;
}

// precision_utils.hpp:18-25
constexpr float kMachineEps = 5.9604644775390625e-08f;   // 2^-24
MI_HD constexpr float gamma_n(int i) { return (kMachineEps * (float)i) / (1.f - kMachineEps * (float)i); }
constexpr float kRayEpsilon = kMachineEps * 1500.f;
constexpr float kSlabScale = 1.f + 2.f * gamma_n(3);      // CompactBVH2Node.hpp:42
constexpr float kInf = __builtin_huge_valf();

// ---- binary16 ---------------------------------------------------------------------------------
MI_HD float half_bits_to_float(uint16_t h) {
#if defined(__HIP_DEVICE_COMPILE__)
  _Float16 v;
  __builtin_memcpy(&v, &h, 2);
  return (float)v;                                        // v_cvt_f32_f16, exact
#else
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16, ex = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
  if (ex == 0) {
    if (man == 0) bits = sign;
    else {                                                // subnormal: renormalise
      int sh = 0;
      while (!(man & 0x400u)) { man <<= 1; ++sh; }
      bits = sign | ((uint32_t)(113 - sh) << 23) | ((man & 0x3FFu) << 13);
    }
  } else if (ex == 31) bits = sign | 0x7F800000u | (man << 13);
  else bits = sign | ((ex + 112u) << 23) | (man << 13);
  float f; memcpy(&f, &bits, 4); return f;
#endif
}

#if !defined(__HIP_DEVICE_COMPILE__)
// float -> half, round to nearest even (host only: used when packing BVH nodes)
inline uint16_t float_to_half_bits(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
  x &= 0x7FFFFFFFu;
  if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (x > 0x7F800000u ? 0x200u : 0u));
  if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);
  if (x < 0x33000001u) return sign;
  const int e = (int)(x >> 23) - 127;
  const uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
  uint32_t q, rem, halfway;
  if (e < -14) {
    const int shift = 13 + (-14 - e);
    q = m >> shift; rem = m & ((1u << shift) - 1u); halfway = 1u << (shift - 1);
  } else {
    q = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3FFu); rem = m & 0x1FFFu; halfway = 0x1000u;
  }
  if (rem > halfway || (rem == halfway && (q & 1u))) ++q;
  return (uint16_t)(sign | q);
}
// precision_utils.hpp:39-47
inline uint16_t half_not_smaller(float f) {
  uint16_t h = float_to_half_bits(f);
  if (half_bits_to_float(h) < f) h = (uint16_t)(h + 1);
  return h;
}
#endif

// ---- sincos: ext/math/sincos.cpp:236-355 (ACC5, ABSERR, MOD360, flg=0) ---------------------------
// `tbl` = 92 floats, sin(i degrees); lives in LDS on the device, static storage on the host.
MI_HD void sincos_deg_table(float x, const float* tbl, float& s, float& c) {
  x = x * (float)(180.0 / 3.14159265358979323846264338327950288);
  const bool neg = x < 0.f;
  if (neg) x = -x;
  x = x - 360.f * floorf(x / 360.f);
  int ix = (int)(x + .5f);
  const float z = x - (float)ix;
  bool sneg = false, cneg = false;
  if (ix > 180) { sneg = true; cneg = true; ix -= 180; }
  if (ix > 90) { cneg = !cneg; ix = 180 - ix; }
  float sx = tbl[ix];
  if (sneg) sx = -sx;
  float cx = tbl[90 - ix];
  if (cneg) cx = -cx;
  const float sz = 1.74531263774940077459e-2f * z;
  const float cz = 1.f - 1.52307909153324666207e-4f * z * z;
  float y = sx * cz + cx * sz;
  if (neg) y = -y;
  s = y;
  c = cx * cz - sx * sz;
}

// ---- xoroshiro128** / splitmix64: include/xoshiro.hpp:18-80 ------------------------------------------
struct Rng { uint64_t s0, s1; };
MI_HD uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
MI_HD uint64_t splitmix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
MI_HD void rng_seed(Rng& r, uint64_t seed) { r.s0 = splitmix64(seed); r.s1 = splitmix64(r.s0); }
MI_HD uint64_t rng_next(Rng& r) {
  const uint64_t a = r.s0;
  uint64_t b = r.s1;
  const uint64_t out = rotl64(a * 5, 7) * 9;
  b ^= a;
  r.s0 = rotl64(a, 24) ^ b ^ (b << 16);
  r.s1 = rotl64(b, 37);
  return out;
}
// xoshiro.hpp:68-80: [1,2) double minus one, narrowed with round-to-nearest (may be exactly 1.0f)
MI_HD float rng_uniform01(Rng& r) {
  const uint64_t bits = (0x3FFull << 52) | (rng_next(r) >> 12);
  double d;
  __builtin_memcpy(&d, &bits, 8);
  return (float)(d - 1.0);
}

// Saturating float -> u32 (== v_cvt_u32_f32), so host and device agree on odd pixel coords.
MI_HD uint32_t f2u_sat(float f) {
  if (!(f > 0.f)) return 0u;
  if (f >= 4294967296.f) return 0xFFFFFFFFu;
  return (uint32_t)f;
}
// Per-pixel stream: seeded from the user seed and FULL-image (row, col).
// A pixel's samples are cut into segments of segment_samples(samplesPerPixel); segment j has its own stream
// (j = 0: the plain per-pixel seed) and its own partial rgb sum, added in segment order. The work
// atom of the persistent kernel is (pixel, segment): with pixel x all-samples atoms a 1440^2 x 1000 spp frame gives
// every lane only 6 atoms, and the drain at the end of the frame cost a third of the throughput. The length is a
// function of the render's sample count alone (never of the batch, crop or GPU count): about sixteen segments per
// pixel - samplesPerPixel / 16 rounded up to a power of two - but no shorter than 4 samples (every atom costs a
// fetch, a seed and a partial sum) and no longer than 64 (atoms enough for every lane's drain to be short).
// Measured on the 1440^2 box frame: 16 spp 8.1e9 -> 10.0e9 casts/s, 64 spp 9.0e9 -> 11.5e9 against one atom per
// pixel; 1000 spp 13.1e9 with 64-sample segments against 12.8e9 with 16-sample ones.
constexpr uint32_t kSegmentSamplesMin = 4, kSegmentSamplesMax = 64, kSegmentsPerPixel = 16;
MI_HD uint32_t segment_shift(uint32_t samplesPerPixel) {
  const uint32_t want = (samplesPerPixel + kSegmentsPerPixel - 1) / kSegmentsPerPixel;          // ceil(spp / 16)
  const uint32_t shift = want <= 1u ? 0u : 32u - (uint32_t)__builtin_clz(want - 1u);              // ceil(log2(want))
  return shift < 2u ? 2u : shift > 6u ? 6u : shift;                                               // 4 ... 64 samples
}
MI_HD uint32_t segment_samples(uint32_t samplesPerPixel) { return 1u << segment_shift(samplesPerPixel); }
MI_HD void rng_seed_pixel_segment(Rng& r, uint64_t userSeed, float row, float col, uint32_t segment) {
  const uint64_t pix = ((uint64_t)f2u_sat(row) << 32) | (uint64_t)f2u_sat(col);
  rng_seed(r, (userSeed ^ ((pix + 1ull) * 0x9e3779b97f4a7c15ull)) ^ ((uint64_t)segment * 0xd1b54a32d192ed03ull));
}
MI_HD void rng_seed_pixel(Rng& r, uint64_t userSeed, float row, float col) { rng_seed_pixel_segment(r, userSeed, row, col, 0u); }

// Deterministic ln(x), x normal and positive (pixel-jitter Box-Muller only).
MI_HD float log_det(float x) {
  uint32_t b;
  __builtin_memcpy(&b, &x, 4);
  int e = (int)((b >> 23) & 0xFFu) - 126;
  const uint32_t mb = (b & 0x007FFFFFu) | 0x3F000000u;
  float m;
  __builtin_memcpy(&m, &mb, 4);
  if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
  else { m = m - 1.0f; }
  float z = m * m;
  float y = 7.0376836292e-2f;
  y = y * m + -1.1514610310e-1f;
  y = y * m + 1.1676998740e-1f;
  y = y * m + -1.2420140846e-1f;
  y = y * m + 1.4249322787e-1f;
  y = y * m + -1.6668057665e-1f;
  y = y * m + 2.0000714765e-1f;
  y = y * m + -2.4999993993e-1f;
  y = y * m + 3.3333331174e-1f;
  y = y * m * z;
  const float fe = (float)e;
  y = y + -2.12194440e-4f * fe;
  y = y + -0.5f * z;
  z = m + y;
  z = z + 0.693359375f * fe;
  return z;
}
// Two N(0,1) variates per pixel sample; stands in for the IPU's hardware f32v2grand
// (codelets/TraceCodelets.cpp:158).
MI_HD void rng_gauss2(Rng& r, const float* sinTbl, float& g0, float& g1) {
  const float ua = rng_uniform01(r);
  const float ub = rng_uniform01(r);
  float w = 1.f - ua;
  if (w < 2.98023223876953125e-08f) w = 2.98023223876953125e-08f;
  const float rad = sqrtf(-2.f * log_det(w));
  float sn, cs;
  sincos_deg_table(6.283185307179586f * ub, sinTbl, sn, cs);
  g0 = rad * cs;
  g1 = rad * sn;
}

// ---- camera: Render.hpp:74-85 --------------------------------------------------------------------
MI_HD f3 pixel_to_ray_dir(float x, float y, float w, float h, float tanTheta) {
  const float aspect = w / h;
  x = (x / w) - .5f;
  y = (y / h) - .5f;
  return normalized(mk(2.f * x * aspect * tanTheta, -2.f * y * tanTheta, -1.f));
}

// ---- Render.hpp:29-33 -----------------------------------------------------------------------------
MI_HD f3 offset_origin(f3 origin, f3 dir, f3 n) {
  const float m = (1.f + min_comp(abs3(origin))) * kRayEpsilon * copysignf(1.f, dot(n, dir));
  return origin + n * m;
}

// ---- sampling / BxDFs: geometric_sampling.hpp:8-63, BxDF.hpp:11-75, geometry.hpp:147-159 -------------
MI_HD void sample_disc_concentric(float u1, float u2, const float* sinTbl, float& ox, float& oy) {
  const float ux = 2.f * u1 - 1.f, uy = 2.f * u2 - 1.f;
  if (ux == 0.f && uy == 0.f) { ox = ux; oy = uy; return; }
  const float piby4 = (float)(3.14159265358979323846264338327950288 / 4.0);
  const float piby2 = (float)(3.14159265358979323846264338327950288 / 2.0);
  float r, th;
  if (fabsf(ux) > fabsf(uy)) { r = ux; th = piby4 * (uy / ux); }
  else { r = uy; th = piby2 - piby4 * (ux / uy); }
  float s, c;
  sincos_deg_table(th, sinTbl, s, c);
  ox = r * c; oy = r * s;
}
MI_HD f3 cosine_sample_hemisphere(float u1, float u2, const float* sinTbl) {
  float x, y;
  sample_disc_concentric(u1, u2, sinTbl, x, y);
  const float z = sqrtf(fmaxf(0.f, 1.f - x * x - y * y));
  return mk(x, y, z);
}
MI_HD f3 sample_diffuse(f3 n, float u1, float u2, const float* sinTbl) {
  f3 xb;
  const f3 a = abs3(n), sq = n * n;
  if (a.x > a.y) { const float il = 1.f / sqrtf(sq.x + sq.z); xb = mk(-n.z * il, 0.f, n.x * il); }
  else { const float il = 1.f / sqrtf(sq.y + sq.z); xb = mk(0.f, n.z * il, -n.y * il); }
  const f3 yb = cross(n, xb);
  const f3 wi = cosine_sample_hemisphere(u1, u2, sinTbl);
  return mk(dot(mk(xb.x, yb.x, n.x), wi), dot(mk(xb.y, yb.y, n.y), wi), dot(mk(xb.z, yb.z, n.z), wi));
}
MI_HD f3 reflect_dir(f3 d, f3 n) {
  const float cosTheta = dot(d, n);
  return normalized(d - n * (cosTheta * 2.f));
}
MI_HD float schlick(float cosTheta, float ri) {
  float r0 = (1.f - ri) / (1.f + ri);
  r0 = r0 * r0;
  const float base = 1.f - cosTheta;
  const float base2 = base * base;
  const float base5 = base2 * base * base2;
  return r0 + (1.f - r0) * base5;
}
MI_HD f3 refract_dir(f3 dir, f3 n, float ndotr, float ri) {
  const float cosTheta = -ndotr;
  const f3 rPerp = (dir + n * cosTheta) * ri;
  const f3 rPar = n * -sqrtf(fabsf(1.f - sqnorm(rPerp)));
  return rPerp + rPar;
}
MI_HD bool dielectric(f3 dir, f3 n, float ri, float u1, f3& out) {
  if (dot(n, dir) > 0.f) n = -n; else ri = 1.f / ri;
  const float ndotr = dot(n, dir);
  const float cost1 = -ndotr;
  const float cost2 = 1.f - ri * ri * (1.f - cost1 * cost1);
  if (cost2 > 0.f && u1 > schlick(cost1, ri)) { out = refract_dir(dir, n, ndotr, ri); return true; }
  out = reflect_dir(dir, n);
  return false;
}
// geometric_sampling.hpp:56-63 — survival probability is the smallest throughput channel
MI_HD bool roulette_stop(float u1, f3& tp) {
  const float p = min_comp(tp);
  if (p == 0.f || u1 > p) return true;
  tp = tp * (1.f / p);
  return false;
}

// sin(i degrees) table of ext/math/sincos.cpp:139-233 as binary32 bit patterns.
#define MI_SIN_TABLE_BITS \
  0x00000000u, 0x3c8ef859u, 0x3d0ef2c6u, 0x3d565e3au, 0x3d8edc7bu, 0x3db27eb6u, 0x3dd61305u, 0x3df996a2u, \
  0x3e0e8365u, 0x3e20305bu, 0x3e31d0d4u, 0x3e43636fu, 0x3e54e6cdu, 0x3e665992u, 0x3e77ba60u, 0x3e8483eeu, \
  0x3e8d2057u, 0x3e95b1beu, 0x3e9e377au, 0x3ea6b0dfu, 0x3eaf1d44u, 0x3eb77c01u, 0x3ebfcc6fu, 0x3ec80de9u, \
  0x3ed03fc9u, 0x3ed8616cu, 0x3ee0722fu, 0x3ee87171u, 0x3ef05e94u, 0x3ef838f7u, 0x3f000000u, 0x3f03d989u, \
  0x3f07a8cau, 0x3f0b6d77u, 0x3f0f2744u, 0x3f12d5e8u, 0x3f167918u, 0x3f1a108du, 0x3f1d9bfeu, 0x3f211b24u, \
  0x3f248dbbu, 0x3f27f37cu, 0x3f2b4c25u, 0x3f2e9772u, 0x3f31d522u, 0x3f3504f3u, 0x3f3826a7u, 0x3f3b39ffu, \
  0x3f3e3ebdu, 0x3f4134a6u, 0x3f441b7du, 0x3f46f30au, 0x3f49bb13u, 0x3f4c7360u, 0x3f4f1bbdu, 0x3f51b3f3u, \
  0x3f543bceu, 0x3f56b31du, 0x3f5919aeu, 0x3f5b6f51u, 0x3f5db3d7u, 0x3f5fe714u, 0x3f6208dau, 0x3f641901u, \
  0x3f66175eu, 0x3f6803cau, 0x3f69de1du, 0x3f6ba635u, 0x3f6d5becu, 0x3f6eff20u, 0x3f708fb2u, 0x3f720d81u, \
  0x3f737871u, 0x3f74d063u, 0x3f76153fu, 0x3f7746eau, 0x3f78654du, 0x3f797051u, 0x3f7a67e2u, 0x3f7b4bebu, \
  0x3f7c1c5cu, 0x3f7cd925u, 0x3f7d8235u, 0x3f7e1781u, 0x3f7e98fdu, 0x3f7f069eu, 0x3f7f605cu, 0x3f7fa62fu, \
  0x3f7fd814u, 0x3f7ff605u, 0x3f800000u, 0x3f7ff605u

}  // namespace mi
