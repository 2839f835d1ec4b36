/*
 * ray_oracle.c — CPU restatement of the ipu_ray_lib hot path. TEST INFRASTRUCTURE ONLY
 * (see ray_oracle.h for who may use it and for the pinning legend).
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math -fopenmp -fPIC -shared
 * No -march flag: like the reference's host build, every float operation is a single IEEE
 * binary32 operation with no fused multiply-add, so results are reproducible bit for bit
 * on any x86-64 host and by the -ffp-contract=off HIP build.
 */
#define _GNU_SOURCE          /* sched_getaffinity / CPU_COUNT */
#include "ray_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__linux__)
#include <sched.h>
#endif

/* ------------------------------------------------------------------------- */
/* small helpers                                                             */
/* ------------------------------------------------------------------------- */
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline ovec3 V(float x, float y, float z) { ovec3 r = {x, y, z}; return r; }
static inline ovec3 vadd(ovec3 a, ovec3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline ovec3 vsub(ovec3 a, ovec3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline ovec3 vmul(ovec3 a, ovec3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline ovec3 vscale(ovec3 a, float f) { return V(a.x * f, a.y * f, a.z * f); }
static inline ovec3 vneg(ovec3 a) { return V(-a.x, -a.y, -a.z); }
static inline ovec3 vabs(ovec3 a) { return V(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
/* geometry.hpp:139 — left-to-right sum of products */
static inline float vdot(ovec3 a, ovec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* geometry.hpp:135 */
static inline float vsqnorm(ovec3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
/* geometry.hpp:137 — normalized(): multiply by reciprocal of the root */
static inline ovec3 vnormalized(ovec3 a) { return vscale(a, 1.f / sqrtf(vsqnorm(a))); }
/* geometry.hpp:141-143 */
static inline ovec3 vcross(ovec3 a, ovec3 v) {
  return V(a.y * v.z - a.z * v.y, a.z * v.x - a.x * v.z, a.x * v.y - a.y * v.x);
}
static inline float vget(ovec3 a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline ovec3 vpermute(ovec3 a, uint32_t ix, uint32_t iy, uint32_t iz) {
  return V(vget(a, ix), vget(a, iy), vget(a, iz));
}

const char* o_version(void) { return "ray_oracle 0.1 (C11 restatement, test infrastructure only)"; }

/* ------------------------------------------------------------------------- */
/* binary16 <-> binary32                                                      */
/* ------------------------------------------------------------------------- */
/* Exact widening of an IEEE binary16 bit pattern (what `(float)half` does in
 * CompactBVH2Node.cpp:8,12,14). */
float o_half_to_float(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t ex = (h >> 10) & 0x1Fu;
  uint32_t man = h & 0x3FFu;
  if (ex == 0) {
    if (man == 0) return bits2f(sign);
    /* subnormal half: value = man * 2^-24 */
    float v = (float)man * 5.9604644775390625e-08f;
    return sign ? -v : v;
  }
  if (ex == 31) return bits2f(sign | 0x7F800000u | (man << 13));
  return bits2f(sign | ((ex + 112u) << 23) | (man << 13));
}

/* Round-to-nearest-even narrowing, the conversion `(half)f` performs (precision_utils.hpp:41). */
uint16_t o_float_to_half_rne(float f) {
  uint32_t x = f2bits(f);
  uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
  uint32_t absx = x & 0x7FFFFFFFu;
  if (absx >= 0x7F800000u) {                       /* inf / nan */
    return (uint16_t)(sign | 0x7C00u | ((absx > 0x7F800000u) ? 0x200u : 0u));
  }
  if (absx >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);   /* rounds to >= 65520 -> inf */
  if (absx < 0x33000001u) return sign;             /* <= 2^-25 rounds to zero (ties-to-even) */
  int32_t e = (int32_t)(absx >> 23) - 127;
  uint32_t m = (absx & 0x7FFFFFu) | 0x800000u;     /* 24-bit significand */
  if (e < -14) {
    /* subnormal result: value = m * 2^(e-23); unit = 2^-24 */
    int shift = (-14 - e) + 13;                    /* bits to drop */
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) q += 1;
    return (uint16_t)(sign | q);
  }
  uint32_t q = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3FFu);
  uint32_t rem = m & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q += 1;   /* carry may bump the exponent: correct */
  return (uint16_t)(sign | q);
}

/* precision_utils.hpp:28-47: round to half, then step one ulp up if that came out smaller */
uint16_t o_round_to_half_not_smaller(float f) {
  uint16_t h = o_float_to_half_rne(f);
  float ff = o_half_to_float(h);
  if (ff < f) h = (uint16_t)(h + 1);
  return h;
}

/* precision_utils.hpp:18-25 — all arithmetic in binary32 */
static const float kMachineEps = 5.9604644775390625e-08f;       /* epsilon * .5f = 2^-24 */
float o_gamma(int i) {
  const float ni = kMachineEps * (float)i;
  return ni / (1.f - ni);
}
float o_ray_epsilon(void) { return kMachineEps * 1500.f; }

/* geometry.hpp:115-125: called "maxi"/"maxc" but selects the SMALLEST component */
uint32_t o_maxi(ovec3 v) {
  if (v.x < v.y) return v.x < v.z ? 0u : 2u;
  return v.y < v.z ? 1u : 2u;
}
float o_maxc(ovec3 v) { return vget(v, o_maxi(v)); }

/* ------------------------------------------------------------------------- */
/* sincos: ext/math/sincos.cpp:236-355 with ACC5, ABSERR, MOD360, flg = 0   [REF] */
/* ------------------------------------------------------------------------- */
/* sin(i degrees), i = 0..91, as binary32 bit patterns of the reference's decimal table
 * (sincos.cpp:139-233); entry 91 repeats entry 89. */
static const uint32_t kSinTblBits[92] = {
  0x00000000u, 0x3c8ef859u, 0x3d0ef2c6u, 0x3d565e3au, 0x3d8edc7bu, 0x3db27eb6u,
  0x3dd61305u, 0x3df996a2u, 0x3e0e8365u, 0x3e20305bu, 0x3e31d0d4u, 0x3e43636fu,
  0x3e54e6cdu, 0x3e665992u, 0x3e77ba60u, 0x3e8483eeu, 0x3e8d2057u, 0x3e95b1beu,
  0x3e9e377au, 0x3ea6b0dfu, 0x3eaf1d44u, 0x3eb77c01u, 0x3ebfcc6fu, 0x3ec80de9u,
  0x3ed03fc9u, 0x3ed8616cu, 0x3ee0722fu, 0x3ee87171u, 0x3ef05e94u, 0x3ef838f7u,
  0x3f000000u, 0x3f03d989u, 0x3f07a8cau, 0x3f0b6d77u, 0x3f0f2744u, 0x3f12d5e8u,
  0x3f167918u, 0x3f1a108du, 0x3f1d9bfeu, 0x3f211b24u, 0x3f248dbbu, 0x3f27f37cu,
  0x3f2b4c25u, 0x3f2e9772u, 0x3f31d522u, 0x3f3504f3u, 0x3f3826a7u, 0x3f3b39ffu,
  0x3f3e3ebdu, 0x3f4134a6u, 0x3f441b7du, 0x3f46f30au, 0x3f49bb13u, 0x3f4c7360u,
  0x3f4f1bbdu, 0x3f51b3f3u, 0x3f543bceu, 0x3f56b31du, 0x3f5919aeu, 0x3f5b6f51u,
  0x3f5db3d7u, 0x3f5fe714u, 0x3f6208dau, 0x3f641901u, 0x3f66175eu, 0x3f6803cau,
  0x3f69de1du, 0x3f6ba635u, 0x3f6d5becu, 0x3f6eff20u, 0x3f708fb2u, 0x3f720d81u,
  0x3f737871u, 0x3f74d063u, 0x3f76153fu, 0x3f7746eau, 0x3f78654du, 0x3f797051u,
  0x3f7a67e2u, 0x3f7b4bebu, 0x3f7c1c5cu, 0x3f7cd925u, 0x3f7d8235u, 0x3f7e1781u,
  0x3f7e98fdu, 0x3f7f069eu, 0x3f7f605cu, 0x3f7fa62fu, 0x3f7fd814u, 0x3f7ff605u,
  0x3f800000u, 0x3f7ff605u,
};
static inline float sintbl(int i) { return bits2f(kSinTblBits[i]); }

void o_sincos(float x, float* s, float* c) {
  /* radians -> degrees with the double-evaluated, float-rounded constant (sincos.cpp:240) */
  x = x * (float)(180.0 / 3.14159265358979323846264338327950288);
  int xsign = 1;
  if (x < 0.f) { xsign = -1; x = -x; }
  x = x - 360.f * floorf(x / 360.f);                /* MOD360 */
  int ix = (int)(x + .5f);                           /* nearest whole degree (truncating conversion) */
  float z = x - (float)ix;                           /* residual in [-0.5, 0.5] */
  int ssign, csign;
  if (ix <= 180) { ssign = 1; csign = 1; }
  else { ssign = -1; csign = -1; ix -= 180; }
  if (ix > 90) { csign = -csign; ix = 180 - ix; }
  float sx = sintbl(ix);
  if (ssign < 0) sx = -sx;
  float cx = sintbl(90 - ix);
  if (csign < 0) cx = -cx;
  /* ACC5 + ABSERR residual polynomials (sincos.cpp:305-310) */
  float sz = 1.74531263774940077459e-2f * z;
  float cz = 1.f - 1.52307909153324666207e-4f * z * z;
  float y = sx * cz + cx * sz;
  if (xsign < 0) y = -y;
  *s = y;
  *c = cx * cz - sx * sz;
}

/* geometry.hpp:147-159: returns (v2, n x v2); third basis vector is n itself */
void o_orthonormal_system(ovec3 n, ovec3* b0, ovec3* b1) {
  ovec3 v2;
  const ovec3 a = vabs(n);
  const ovec3 sq = vmul(n, n);
  if (a.x > a.y) {
    float invLen = 1.f / sqrtf(sq.x + sq.z);
    v2 = V(-n.z * invLen, 0.f, n.x * invLen);
  } else {
    float invLen = 1.f / sqrtf(sq.y + sq.z);
    v2 = V(0.f, n.z * invLen, -n.y * invLen);
  }
  *b0 = v2;
  *b1 = vcross(n, v2);
}

/* ------------------------------------------------------------------------- */
/* xoroshiro128** + splitmix64: include/xoshiro.hpp:18-80                [REF] */
/* ------------------------------------------------------------------------- */
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

uint64_t o_splitmix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
void o_xoshiro_seed(uint64_t s[2], uint64_t seed) {
  s[0] = o_splitmix64(seed);
  s[1] = o_splitmix64(s[0]);
}
uint64_t o_xoshiro_next(uint64_t s[2]) {
  const uint64_t s0 = s[0];
  uint64_t s1 = s[1];
  const uint64_t result = rotl64(s0 * 5, 7) * 9;
  s1 ^= s0;
  s[0] = rotl64(s0, 24) ^ s1 ^ (s1 << 16);
  s[1] = rotl64(s1, 37);
  return result;
}
void o_xoshiro_jump(uint64_t s[2]) {
  static const uint64_t kJump[2] = {0xdf900294d8f554a5ull, 0x170865df4b3201fcull};
  uint64_t a = 0, b = 0;
  for (int i = 0; i < 2; ++i)
    for (int bit = 0; bit < 64; ++bit) {
      if (kJump[i] & (1ull << bit)) { a ^= s[0]; b ^= s[1]; }
      o_xoshiro_next(s);
    }
  s[0] = a; s[1] = b;
}
/* xoshiro.hpp:68-80: 52 random mantissa bits -> double in [1,2) - 1.0, narrowed to float
 * (round-to-nearest: the result can be exactly 1.0f) */
float o_xoshiro_uniform01(uint64_t s[2]) {
  uint64_t x = o_xoshiro_next(s);
  uint64_t bits = (0x3FFull << 52) | (x >> 12);
  double d; memcpy(&d, &bits, 8);
  return (float)(d - 1.0);
}

/* ------------------------------------------------------------------------- */
/* Deterministic ln(x) for the Box-Muller pixel jitter (no reference counterpart).
 * Cephes-style: x = m*2^e, m in [sqrt(.5), sqrt(2)), degree-8 minimax in (m-1). Only
 * + - * on binary32, so it is bit-reproducible on the GPU. Domain: normal positive x. */
/* ------------------------------------------------------------------------- */
float o_logf_det(float x) {
  uint32_t b = f2bits(x);
  int e = (int)((b >> 23) & 0xFFu) - 126;
  float m = bits2f((b & 0x007FFFFFu) | 0x3F000000u);           /* [0.5, 1) */
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

/* Two N(0,1) variates from two uniforms of the per-pixel stream (Box-Muller; angle through
 * the reference's own sincos). Stands in for __builtin_ipu_f32v2grand
 * (codelets/TraceCodelets.cpp:158). */
void o_gauss2(uint64_t s[2], float* g0, float* g1) {
  const float ua = o_xoshiro_uniform01(s);
  const float ub = o_xoshiro_uniform01(s);
  float w = 1.f - ua;
  if (w < 2.98023223876953125e-08f) w = 2.98023223876953125e-08f;   /* 2^-25: ua rounded up to 1 */
  const float r = sqrtf(-2.f * o_logf_det(w));
  float sn, cs;
  o_sincos(6.283185307179586f * ub, &sn, &cs);
  *g0 = r * cs;
  *g1 = r * sn;
}

/* ------------------------------------------------------------------------- */
/* AABB slab test: CompactBVH2Node.hpp:14-50 (host branch)              [PROBE] */
/* ------------------------------------------------------------------------- */
int o_slab(float invDir, float origin, float slabMin, float slabMax, float* t0, float* t1) {
  float tmin = (slabMin - origin) * invDir;
  float tmax = (slabMax - origin) * invDir;
  if (tmin > tmax) { float tmp = tmin; tmin = tmax; tmax = tmp; }
  tmax *= 1.f + 2.f * o_gamma(3);
  *t0 = tmin > *t0 ? tmin : *t0;
  *t1 = tmax < *t1 ? tmax : *t1;
  if (*t0 > *t1) return 0;
  return 1;
}

/* CompactBVH2Node.cpp:5-22: max = min + (float)half, axis by axis with early out */
int o_node_intersect(const onode* n, ovec3 o, ovec3 inv, float* t0, float* t1) {
  float max_x = n->min_x + o_half_to_float(n->dx);
  if (o_slab(inv.x, o.x, n->min_x, max_x, t0, t1)) {
    float max_y = n->min_y + o_half_to_float(n->dy);
    if (o_slab(inv.y, o.y, n->min_y, max_y, t0, t1)) {
      float max_z = n->min_z + o_half_to_float(n->dz);
      if (o_slab(inv.z, o.z, n->min_z, max_z, t0, t1)) return 1;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* primitives                                                                 */
/* ------------------------------------------------------------------------- */
/* Primitives.cpp:5-22. "maxi" picks the smallest SIGNED component as the shear axis. [PROBE] */
void o_ray_shear(const oray* ray, oshear* tf) {
  tf->o = ray->origin;
  ovec3 d = ray->direction;
  tf->iz = o_maxi(d);
  tf->ix = tf->iz + 1; if (tf->ix == 3) tf->ix = 0;
  tf->iy = tf->ix + 1; if (tf->iy == 3) tf->iy = 0;
  d = vpermute(d, tf->ix, tf->iy, tf->iz);
  tf->dir = d;
  tf->sx = -d.x / d.z;
  tf->sy = -d.y / d.z;
  tf->sz = 1.f / d.z;
}

/* The reference's compile-time variant ALLOW_DOUBLE_FALLBACK (CMakeLists.txt:13,34-41: a -D flag of the whole build,
 * default 0) as a process-wide switch of this test library: 0 = the default build, 1 = Mesh.cpp:38-51 compiled in. */
static int g_double_fallback = 0;
void o_set_double_fallback(int on) { g_double_fallback = on ? 1 : 0; }
int o_get_double_fallback(void) { return g_double_fallback; }

/* Mesh.cpp:6-104; ALLOW_DOUBLE_FALLBACK == 0 is the reference default (CMakeLists.txt:13). [PROBE]
 * Returns t (0.f == miss); bary always receives b0,b1,b2 when the determinant test passed. */
float o_intersect_triangle(ovec3 p0, ovec3 p1, ovec3 p2, const oshear* tf, float tFar, float bary[3]) {
  bary[0] = bary[1] = bary[2] = 0.f;
  ovec3 p0t = vsub(p0, tf->o), p1t = vsub(p1, tf->o), p2t = vsub(p2, tf->o);
  p0t = vpermute(p0t, tf->ix, tf->iy, tf->iz);
  p1t = vpermute(p1t, tf->ix, tf->iy, tf->iz);
  p2t = vpermute(p2t, tf->ix, tf->iy, tf->iz);
  p0t.x += tf->sx * p0t.z;  p0t.y += tf->sy * p0t.z;
  p1t.x += tf->sx * p1t.z;  p1t.y += tf->sy * p1t.z;
  p2t.x += tf->sx * p2t.z;  p2t.y += tf->sy * p2t.z;
  float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
  float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
  float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
  if (g_double_fallback) {
    /* Mesh.cpp:38-51: fall back to double precision for edge cases */
    if ((e0 == 0.0f || e1 == 0.0f || e2 == 0.0f)) {
      double p2txp1ty = (double)p2t.x * (double)p1t.y;
      double p2typ1tx = (double)p2t.y * (double)p1t.x;
      e0 = (float)(p2typ1tx - p2txp1ty);
      double p0txp2ty = (double)p0t.x * (double)p2t.y;
      double p0typ2tx = (double)p0t.y * (double)p2t.x;
      e1 = (float)(p0typ2tx - p0txp2ty);
      double p1txp0ty = (double)p1t.x * (double)p0t.y;
      double p1typ0tx = (double)p1t.y * (double)p0t.x;
      e2 = (float)(p1typ0tx - p1txp0ty);
    }
  }
  if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return 0.f;
  float det = e0 + e1 + e2;
  if (det == 0) return 0.f;
  p0t.z *= tf->sz;  p1t.z *= tf->sz;  p2t.z *= tf->sz;
  float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
  if (det < 0.f && (tScaled >= 0.f || tScaled < tFar * det)) return 0.f;
  else if (det > 0.f && (tScaled <= 0.f || tScaled > tFar * det)) return 0.f;
  float invDet = 1 / det;
  float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
  float t = tScaled * invDet;
  /* conservative error bound on t; every "maxc" below is the reference's min-selecting maxc() */
  float maxZt = o_maxc(vabs(V(p0t.z, p1t.z, p2t.z)));
  float deltaZ = o_gamma(3) * maxZt;
  float maxXt = o_maxc(vabs(V(p0t.x, p1t.x, p2t.x)));
  float maxYt = o_maxc(vabs(V(p0t.y, p1t.y, p2t.y)));
  float deltaX = o_gamma(5) * (maxXt + maxZt);
  float deltaY = o_gamma(5) * (maxYt + maxZt);
  float deltaE = 2 * (o_gamma(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
  float maxE = o_maxc(vabs(V(e0, e1, e2)));
  float deltaT = 3 * (o_gamma(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * fabsf(invDet);
  bary[0] = b0; bary[1] = b1; bary[2] = b2;
  if (t <= deltaT) return 0.f;
  return t;
}

/* Primitives.cpp:24-47 [PROBE]. 0.f == Intersection::Failed() */
float o_sphere_intersect(const osphere* s, const oray* ray) {
  const float radius2 = s->radius * s->radius;                    /* Primitives.hpp:44 */
  ovec3 f = vsub(V(s->x, s->y, s->z), ray->origin);
  float rd2 = 1.f / vsqnorm(ray->direction);
  float tca = vdot(f, ray->direction) * rd2;
  if (tca < 0.f) return 0.f;
  ovec3 l = vsub(f, vscale(ray->direction, tca));
  float l2 = vsqnorm(l);
  if (l2 > radius2) return 0.f;
  float td = sqrtf(radius2 - l2) * rd2;
  float t0 = tca - td, t1 = tca + td;
  if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
  if (t0 < ray->tMin) {
    t0 = t1;
    if (t0 < ray->tMin) return 0.f;
  }
  return t0;
}

/* Primitives.cpp:49-67 [PROBE] */
float o_disc_intersect(const odisc* d, const oray* ray) {
  const ovec3 c = V(d->cx, d->cy, d->cz), n = V(d->nx, d->ny, d->nz);
  const float r2 = d->r * d->r;                                   /* Primitives.hpp:70 */
  float angle = vdot(n, ray->direction);
  if (angle != 0.f) {
    float dd = fabsf(vdot(c, n));
    float t = -(vdot(n, ray->origin) + dd) / angle;
    if (t > kMachineEps) {
      ovec3 hp = vadd(ray->origin, vscale(ray->direction, t));
      float d2 = vsqnorm(vsub(hp, c));
      if (d2 < r2) return t;
    }
  }
  return 0.f;
}

/* Render.hpp:29-33 [PROBE] */
void o_offset_ray(oray* r, ovec3 n) {
  const float m = (1.f + o_maxc(vabs(r->origin))) * o_ray_epsilon() * copysignf(1.f, vdot(n, r->direction));
  r->origin = vadd(r->origin, vscale(n, m));
}

/* Render.hpp:74-85 [PROBE] */
ovec3 o_pixel_to_ray_dir(float x, float y, float w, float h, float tanTheta) {
  const float aspect = w / h;
  x = (x / w) - .5f;
  y = (y / h) - .5f;
  return vnormalized(V(2.f * x * aspect * tanTheta, -2.f * y * tanTheta, -1.f));
}

/* geometric_sampling.hpp:8-31 [REF] */
void o_sample_disc_concentric(float u1, float u2, float* ox, float* oy) {
  float ux = 2.f * u1 - 1.f, uy = 2.f * u2 - 1.f;
  if (ux == 0.f && uy == 0.f) { *ox = ux; *oy = uy; return; }
  float r, th;
  const float piby4 = (float)(3.14159265358979323846264338327950288 / 4.0);
  const float piby2 = (float)(3.14159265358979323846264338327950288 / 2.0);
  if (fabsf(ux) > fabsf(uy)) { r = ux; th = piby4 * (uy / ux); }
  else { r = uy; th = piby2 - piby4 * (ux / uy); }
  float s, c;
  o_sincos(th, &s, &c);
  *ox = r * c; *oy = r * s;
}

/* geometric_sampling.hpp:42-47 [REF] */
ovec3 o_cosine_sample_hemisphere(float u1, float u2) {
  float x, y;
  o_sample_disc_concentric(u1, u2, &x, &y);
  float z = sqrtf(fmaxf(0.f, 1.f - x * x - y * y));
  return V(x, y, z);
}

/* BxDF.hpp:11-30 [REF] */
ovec3 o_sample_diffuse(ovec3 n, float u1, float u2) {
  ovec3 xb, yb;
  o_orthonormal_system(n, &xb, &yb);
  const ovec3 wi = o_cosine_sample_hemisphere(u1, u2);
  return V(vdot(V(xb.x, yb.x, n.x), wi), vdot(V(xb.y, yb.y, n.y), wi), vdot(V(xb.z, yb.z, n.z), wi));
}

/* BxDF.hpp:33-37 [REF] */
ovec3 o_reflect(ovec3 d, ovec3 n) {
  float cosTheta = vdot(d, n);
  return vnormalized(vsub(d, vscale(n, cosTheta * 2.f)));
}

/* BxDF.hpp:39-46 [REF] */
float o_schlick(float cosTheta, float ri) {
  float r0 = (1.f - ri) / (1.f + ri);
  r0 = r0 * r0;
  float base = 1.f - cosTheta;
  float base2 = base * base;
  float base5 = base2 * base * base2;
  return r0 + (1.f - r0) * base5;
}

/* BxDF.hpp:48-55 [REF] */
ovec3 o_refract(ovec3 dir, ovec3 n, float ndotr, float ri) {
  const float cosTheta = -ndotr;
  ovec3 rPerp = vscale(vadd(dir, vscale(n, cosTheta)), ri);
  ovec3 rPar = vscale(n, -sqrtf(fabsf(1.f - vsqnorm(rPerp))));
  return vadd(rPerp, rPar);
}

/* BxDF.hpp:57-75 [REF] */
int o_dielectric(const oray* ray, ovec3 n, float ri, float u1, ovec3* out) {
  if (vdot(n, ray->direction) > 0.f) n = vneg(n);
  else ri = 1.f / ri;
  const float ndotr = vdot(n, ray->direction);
  const float cost1 = -ndotr;
  const float cost2 = 1.f - ri * ri * (1.f - cost1 * cost1);
  if (cost2 > 0.f && u1 > o_schlick(cost1, ri)) { *out = o_refract(ray->direction, n, ndotr, ri); return 1; }
  *out = o_reflect(ray->direction, n);
  return 0;
}

/* geometric_sampling.hpp:56-63 [REF]: p is the SMALLEST throughput channel (maxc quirk) */
int o_evaluate_roulette(float u1, ovec3* tp) {
  const float p = o_maxc(*tp);
  if (p == 0.f || u1 > p) return 1;
  *tp = vscale(*tp, 1.f / p);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* leaf dispatch: primLookup + Primitive::intersect (codelets/TraceCodelets.cpp:127-140,
 * Mesh.hpp:88-121)                                                            */
/* ------------------------------------------------------------------------- */
typedef struct { float t; uint32_t primID; ovec3 normal; int isSphere, isDisc; uint32_t idx; } oleafhit;

static oleafhit leaf_intersect(const oscene* sc, uint32_t geomID, uint32_t primID, const oray* ray) {
  oleafhit r; memset(&r, 0, sizeof r);
  const ogeomref g = sc->geometry[geomID];
  r.idx = g.index;
  if (g.type == O_GEOM_MESH) {
    const omeshinfo mi = sc->meshInfo[g.index];
    const uint16_t* tri = sc->meshTris + 3 * (size_t)(mi.firstIndex + primID);
    const ovec3* verts = sc->meshVerts + mi.firstVertex;
    const ovec3 p0 = verts[tri[0]], p1 = verts[tri[1]], p2 = verts[tri[2]];
    oshear tf; o_ray_shear(ray, &tf);                      /* recomputed per leaf, Mesh.hpp:89 */
    float bary[3];
    const float inf = INFINITY;
    float t = o_intersect_triangle(p0, p1, p2, &tf, inf, bary);
    r.t = inf; r.primID = 0xFFFFFFFFu;                     /* Intersection(inf, nullptr) */
    if (t > 0.f && t < inf) {
      r.t = t; r.primID = primID;
      /* computeNormal, Mesh.hpp:107-121 */
      const int hasNormals = (sc->nNormals != 0);          /* "if scene has normals assume every mesh has", trace.cpp:205-210 */
      if (!hasNormals) r.normal = vnormalized(vcross(vsub(p1, p0), vsub(p2, p0)));
      else {
        const ovec3* nrm = sc->meshNormals + mi.firstVertex;
        ovec3 acc = vadd(vadd(vscale(nrm[tri[0]], bary[0]), vscale(nrm[tri[1]], bary[1])), vscale(nrm[tri[2]], bary[2]));
        r.normal = vnormalized(acc);
      }
    }
  } else if (g.type == O_GEOM_SPHERE) {
    r.isSphere = 1;
    r.t = o_sphere_intersect(&sc->spheres[g.index], ray);
    r.primID = (r.t != 0.f) ? 0u : 0xFFFFFFFFu;
  } else {
    r.isDisc = 1;
    r.t = o_disc_intersect(&sc->discs[g.index], ray);
    r.primID = (r.t != 0.f) ? 0u : 0xFFFFFFFFu;
  }
  return r;
}

/* ------------------------------------------------------------------------- */
/* CompactBvh::intersect / ::occluded (CompactBvh.hpp:33-139)       [UNPINNED vs reference
 * outputs: no reference-produced traversal result exists; pinned structurally by the
 * brute-force cross-check in tests/test_oracle_pins.py::test_bvh_queries_against_brute_force] */
/* ------------------------------------------------------------------------- */
#define O_MAX_STACK 128

ointersection o_bvh_intersect(const oscene* sc, const oray* ray, ostats* st) {
  uint32_t stack[O_MAX_STACK];
  uint32_t sp = 0;
  stack[sp++] = 0;
  const ovec3 inv = V(1.f / ray->direction.x, 1.f / ray->direction.y, 1.f / ray->direction.z);
  ointersection best; memset(&best, 0, sizeof best);
  best.t = ray->tMax; best.primID = 0xFFFFFFFFu; best.hit = 0;
  oleafhit bestLeaf; memset(&bestLeaf, 0, sizeof bestLeaf);
  uint64_t nv = 0, lt = 0;
  while (sp) {
    const uint32_t cur = stack[--sp];
    const onode* node = &sc->bvhNodes[cur];
    float t0 = ray->tMin, t1 = best.t;
    ++nv;
    if (o_node_intersect(node, ray->origin, inv, &t0, &t1)) {
      if (node->geomID != 0xFFFFu) {
        ++lt;
        oleafhit lh = leaf_intersect(sc, node->geomID, node->link, ray);
        if (lh.t > ray->tMin && lh.t < best.t) {
          best.hit = 1; best.geomID = node->geomID; best.primID = lh.primID; best.t = lh.t;
          bestLeaf = lh;
        }
      } else {
        stack[sp++] = node->link;      /* second child */
        stack[sp++] = cur + 1;         /* first child is visited first */
      }
    }
  }
  if (st) { st->casts += 1; st->nodesVisited += nv; st->leafTests += lt; }
  if (best.hit) {
    /* Primitive::normal(i, hitPoint), resolved here so callers need no primitive pointer.
     * Sphere: (point - centre).normalized() with point = origin + dir*t (Render.hpp:21-22). */
    if (bestLeaf.isSphere) {
      const osphere* s = &sc->spheres[bestLeaf.idx];
      ovec3 p = vadd(ray->origin, vscale(ray->direction, best.t));
      best.normal = vnormalized(vsub(p, V(s->x, s->y, s->z)));
    } else if (bestLeaf.isDisc) {
      const odisc* d = &sc->discs[bestLeaf.idx];
      best.normal = V(d->nx, d->ny, d->nz);
    } else best.normal = bestLeaf.normal;
  }
  return best;
}

int o_bvh_occluded(const oscene* sc, const oray* ray, ostats* st) {
  uint32_t stack[O_MAX_STACK];
  uint32_t sp = 0;
  stack[sp++] = 0;
  const ovec3 inv = V(1.f / ray->direction.x, 1.f / ray->direction.y, 1.f / ray->direction.z);
  uint64_t nv = 0, lt = 0;
  int result = 0;
  while (sp) {
    const uint32_t cur = stack[--sp];
    const onode* node = &sc->bvhNodes[cur];
    float t0 = ray->tMin, t1 = ray->tMax;
    ++nv;
    if (o_node_intersect(node, ray->origin, inv, &t0, &t1)) {
      if (node->geomID != 0xFFFFu) {
        ++lt;
        oleafhit lh = leaf_intersect(sc, node->geomID, node->link, ray);
        if (lh.t > ray->tMin && lh.t < ray->tMax) { result = 1; break; }
      } else {
        stack[sp++] = node->link;
        stack[sp++] = cur + 1;
      }
    }
  }
  if (st) { st->casts += 1; st->nodesVisited += nv; st->leafTests += lt; }
  return result;
}

/* Render.hpp:15-23 */
static void update_hit(const ointersection* i, ohit* hit) {
  hit->geomID = (uint16_t)i->geomID;
  hit->primID = i->primID;
  hit->r.tMax = i->t;
  hit->r.origin = vadd(hit->r.origin, vscale(hit->r.direction, i->t));
  hit->normal = i->normal;
}

/* geometry.hpp:236-242: HitRecord(origin, dir); throughput left as it was */
static void hit_record_init(ohit* h, ovec3 origin, ovec3 dir) {
  h->r.origin = origin; h->r.tMin = 0.f; h->r.direction = dir; h->r.tMax = INFINITY;
  h->primID = 0xFFFFFFFFu;
  h->normal = V(0.f, 0.f, 1.f);
  h->geomID = 0xFFFFu;
  h->flags = 0;
}

static float fov_tan_theta(const oscene* sc) {
  float s, c;
  o_sincos(sc->fovRadians / 2.f, &s, &c);
  return s / c;
}

/* src/app_utils.cpp:19-47 with gen == nullptr, then zeroRgb (app_utils.cpp:49-53) */
void o_init_ray_stream(const oscene* sc, otrace* rays) {
  const float tanTheta = fov_tan_theta(sc);
  size_t i = 0;
  for (uint32_t r = (uint32_t)sc->winR; r < (uint32_t)(sc->winR + sc->winH); ++r)
    for (uint32_t c = (uint32_t)sc->winC; c < (uint32_t)(sc->winC + sc->winW); ++c) {
      float pu = (float)r, pv = (float)c;
      ovec3 d = o_pixel_to_ray_dir(pv, pu, sc->imageWidth, sc->imageHeight, tanTheta);
      hit_record_init(&rays[i].h, V(0.f, 0.f, 0.f), d);
      rays[i].h.throughput = V(0.f, 0.f, 0.f);   /* uninitialised in the reference; zero keeps fixtures stable */
      rays[i].u = (float)r; rays[i].v = (float)c;
      rays[i].rgb = V(0.f, 0.f, 0.f);
      ++i;
    }
}

/* Render.hpp:37-72 */
static void trace_shadow_ray(const oscene* sc, otrace* result, float ambient, ovec3 lightPos, ostats* st) {
  ohit* hit = &result->h;
  ointersection isect = o_bvh_intersect(sc, &hit->r, st);
  if (isect.hit) {
    update_hit(&isect, hit);
    const omaterial* mat = &sc->materials[sc->matIDs[hit->geomID]];
    oray shadow = hit->r;
    ovec3 lightOffset = vsub(lightPos, shadow.origin);
    shadow.direction = vnormalized(lightOffset);
    o_offset_ray(&shadow, hit->normal);
    shadow.tMin = 0.f;
    shadow.tMax = sqrtf(vsqnorm(lightOffset));
    ovec3 color = vscale(mat->albedo, ambient);
    if (!o_bvh_occluded(sc, &shadow, st))
      color = vadd(color, vscale(mat->albedo, vdot(shadow.direction, hit->normal)));
    result->rgb = color;
  } else {
    hit->flags |= O_FLAG_ESCAPED;
  }
}

void o_shadow_trace(const oscene* sc, otrace* rays, size_t n, int numThreads, ostats* st) {
  const ovec3 light = V(18.f, 257.f, -1060.f);      /* trace.cpp:247 */
  ostats tot = {0, 0, 0, 0};
  if (numThreads < 1) numThreads = 1;
#pragma omp parallel num_threads(numThreads)
  {
    ostats loc = {0, 0, 0, 0};
#pragma omp for schedule(dynamic, 256)
    for (long long i = 0; i < (long long)n; ++i) trace_shadow_ray(sc, &rays[i], .05f, light, &loc);
#pragma omp critical
    { tot.casts += loc.casts; tot.nodesVisited += loc.nodesVisited; tot.leafTests += loc.leafTests; }
  }
  tot.paths = n;
  if (st) { st->casts += tot.casts; st->nodesVisited += tot.nodesVisited; st->leafTests += tot.leafTests; st->paths += tot.paths; }
}

/* ------------------------------------------------------------------------- */
/* path tracing: codelets/TraceCodelets.cpp:198-260 == trace.cpp:115-188      */
/* ------------------------------------------------------------------------- */
typedef float (*uniform_fn)(void* ctx);

static void path_trace_one(const oscene* sc, otrace* result, uniform_fn uni, void* ctx, ostats* st) {
  ohit* hit = &result->h;
  hit->throughput = V(1.f, 1.f, 1.f);
  ovec3 color = V(0.f, 0.f, 0.f);
  for (uint32_t i = 0; i < sc->maxPathLength; ++i) {
    o_offset_ray(&hit->r, hit->normal);
    hit->r.tMin = 0.f;
    hit->r.tMax = INFINITY;
    ointersection isect = o_bvh_intersect(sc, &hit->r, st);
    if (isect.hit) {
      update_hit(&isect, hit);
      const omaterial* mat = &sc->materials[sc->matIDs[hit->geomID]];
      if (mat->emissive) color = vadd(color, vmul(hit->throughput, mat->emission));
      if (mat->type == O_MAT_DIFFUSE) {
        const float u1 = uni(ctx);
        const float u2 = uni(ctx);
        hit->r.direction = o_sample_diffuse(hit->normal, u1, u2);
        hit->throughput = vmul(hit->throughput, mat->albedo);
      } else if (mat->type == O_MAT_SPECULAR) {
        hit->r.direction = o_reflect(hit->r.direction, hit->normal);
        hit->throughput = vmul(hit->throughput, mat->albedo);
      } else if (mat->type == O_MAT_REFRACTIVE) {
        const float u1 = uni(ctx);
        ovec3 dir;
        const int refracted = o_dielectric(&hit->r, hit->normal, mat->ior, u1, &dir);
        hit->r.direction = dir;
        if (refracted) hit->throughput = vmul(hit->throughput, mat->albedo);
      } else {
        result->rgb = vscale(result->rgb, NAN);
        hit->flags |= O_FLAG_ERROR;
      }
    } else {
      hit->flags |= O_FLAG_ESCAPED;
      break;
    }
    if (i > sc->rouletteStartDepth) {
      const float u1 = uni(ctx);
      if (o_evaluate_roulette(u1, &hit->throughput)) break;
    }
  }
  result->rgb = vadd(result->rgb, color);
  if (st) st->paths += 1;
}

static float uni_state(void* ctx) { return o_xoshiro_uniform01((uint64_t*)ctx); }

/* float pixel coordinate -> unsigned with the saturating semantics of the GPU's v_cvt_u32_f32 */
static uint32_t f2u_sat(float f) {
  if (!(f > 0.f)) return 0u;
  if (f >= 4294967296.f) return 0xFFFFFFFFu;
  return (uint32_t)f;
}

/* Per-pixel stream seed: one xoroshiro128** state per image pixel, derived
 * from the user seed and the pixel's (row, col) in FULL-image coordinates, so the image does
 * not depend on crop windows, batch sizes or the number of GPUs. */
/* Tier-1 stream definition: a pixel's samples are cut into SEGMENTS of o_segment_samples(spp) -
 * ceil(spp / 16) rounded up to a power of two within 4 ... 64, a function of the render's sample count alone; segment j has its own stream,
 * seeded from (seed, row, col, j) - j = 0 is the plain per-pixel seed - and its own partial rgb sum; the pixel's rgb
 * is ((rgb_in + segment 0) + segment 1) + ... in segment order, segment 0 accumulating onto rgb_in directly. Up to
 * one segment's worth of samples this is one stream and one running sum. */
uint32_t o_segment_samples(uint32_t samplesPerPixel) {
  const uint32_t want = (samplesPerPixel + 15u) / 16u;      /* about sixteen segments per pixel ...           */
  uint32_t len = 4u;                                        /* ... of a power-of-two length from 4 ...        */
  while (len < want && len < 64u) len *= 2u;                /* ... to 64 samples                              */
  return len;
}
static void pixel_stream_seed_segment(uint64_t s[2], uint64_t rngSeed, float pu, float pv, uint32_t segment) {
  const uint64_t pix = ((uint64_t)f2u_sat(pu) << 32) | (uint64_t)f2u_sat(pv);
  o_xoshiro_seed(s, (rngSeed ^ ((pix + 1ull) * 0x9e3779b97f4a7c15ull)) ^ ((uint64_t)segment * 0xd1b54a32d192ed03ull));
}
static void pixel_stream_seed(uint64_t s[2], uint64_t rngSeed, float pu, float pv) {
  pixel_stream_seed_segment(s, rngSeed, pu, pv, 0u);
}

/* sampleCameraRays, codelets/TraceCodelets.cpp:142-164, one ray */
static void sample_camera_ray(const oscene* sc, otrace* r, float tanTheta, uint64_t s[2]) {
  float g0, g1;
  o_gauss2(s, &g0, &g1);
  const float prow = r->u + sc->antiAliasScale * g0;
  const float pcol = r->v + sc->antiAliasScale * g1;
  ovec3 d = o_pixel_to_ray_dir(pcol, prow, sc->imageWidth, sc->imageHeight, tanTheta);
  hit_record_init(&r->h, V(0.f, 0.f, 0.f), d);
}

void o_path_trace_pixel_rng(const oscene* sc, otrace* rays, size_t n, int numThreads, ostats* st) {
  const float tanTheta = fov_tan_theta(sc);
  ostats tot = {0, 0, 0, 0};
  if (numThreads < 1) numThreads = 1;
#pragma omp parallel num_threads(numThreads)
  {
    ostats loc = {0, 0, 0, 0};
#pragma omp for schedule(dynamic, 64)
    for (long long i = 0; i < (long long)n; ++i) {
      otrace* r = &rays[i];
      uint64_t s[2];
      uint32_t segment = 0;
      const uint32_t segLen = o_segment_samples(sc->samplesPerPixel);
      for (uint32_t first = 0; first < sc->samplesPerPixel || first == 0; first += segLen, ++segment) {
        const uint32_t last = sc->samplesPerPixel - first < segLen ? sc->samplesPerPixel : first + segLen;
        pixel_stream_seed_segment(s, sc->rngSeed, r->u, r->v, segment);
        const ovec3 total = r->rgb;
        if (segment > 0) r->rgb = V(0.f, 0.f, 0.f);
        for (uint32_t smp = first; smp < last; ++smp) {
          sample_camera_ray(sc, r, tanTheta, s);
          path_trace_one(sc, r, uni_state, s, &loc);
        }
        if (segment > 0) r->rgb = vadd(total, r->rgb);
        if (sc->samplesPerPixel == 0) break;
      }
    }
#pragma omp critical
    { tot.casts += loc.casts; tot.nodesVisited += loc.nodesVisited; tot.leafTests += loc.leafTests; tot.paths += loc.paths; }
  }
  if (st) { st->casts += tot.casts; st->nodesVisited += tot.nodesVisited; st->leafTests += tot.leafTests; st->paths += tot.paths; }
}

/* ---- tier 2: one shared generator, libstdc++ normal_distribution<float> ---- */
typedef struct { uint64_t s[2]; int haveSaved; float saved; } shared_rng;

/* libstdc++ (GCC 11) generate_canonical<float,24>(urng) for a 64-bit URNG: one draw,
 * float(x) / 2^64, clamped below 1 (bits/random.tcc) */
static float canonical_float(shared_rng* g) {
  const uint64_t x = o_xoshiro_next(g->s);
  float ret = (float)x / 18446744073709551616.f;
  if (ret >= 1.f) ret = nextafterf(1.f, 0.f);
  return ret;
}
/* libstdc++ normal_distribution<float>::operator() (Marsaglia polar), bits/random.tcc */
static float normal_float(shared_rng* g, float mean, float stddev) {
  float ret;
  if (g->haveSaved) { g->haveSaved = 0; ret = g->saved; }
  else {
    float x, y, r2;
    do {
      x = (float)((double)(2.0f * canonical_float(g)) - 1.0);
      y = (float)((double)(2.0f * canonical_float(g)) - 1.0);
      r2 = x * x + y * y;
    } while (r2 > 1.0f || r2 == 0.0f);
    const float mult = sqrtf(-2 * logf(r2) / r2);
    g->saved = x * mult; g->haveSaved = 1;
    ret = y * mult;
  }
  return ret * stddev + mean;
}
static float uni_shared(void* ctx) { return o_xoshiro_uniform01(((shared_rng*)ctx)->s); }

void o_path_trace_shared_rng(const oscene* sc, otrace* rays, size_t n, ostats* st) {
  const float tanTheta = fov_tan_theta(sc);
  shared_rng g; o_xoshiro_seed(g.s, sc->rngSeed); g.haveSaved = 0; g.saved = 0.f;   /* app_utils.cpp:238 */
  ostats loc = {0, 0, 0, 0};
  for (uint32_t smp = 0; smp < sc->samplesPerPixel; ++smp) {
    /* initPerspectiveRayStream(..., &sampler): a fresh distribution object per pass (app_utils.cpp:30) */
    g.haveSaved = 0;
    for (size_t i = 0; i < n; ++i) {
      float pu = rays[i].u, pv = rays[i].v;
      pu += normal_float(&g, 0.f, sc->antiAliasScale);
      pv += normal_float(&g, 0.f, sc->antiAliasScale);
      ovec3 d = o_pixel_to_ray_dir(pv, pu, sc->imageWidth, sc->imageHeight, tanTheta);
      hit_record_init(&rays[i].h, V(0.f, 0.f, 0.f), d);
    }
    for (size_t i = 0; i < n; ++i) path_trace_one(sc, &rays[i], uni_shared, &g, &loc);
  }
  if (st) { st->casts += loc.casts; st->nodesVisited += loc.nodesVisited; st->leafTests += loc.leafTests; st->paths += loc.paths; }
}

/* ------------------------------------------------------------------------- */
/* escaped rays + NIF                                                         */
/* ------------------------------------------------------------------------- */
/* PreProcessEscapedRays, codelets/TraceCodelets.cpp:321-358 */
void o_escaped_uv(const otrace* rays, size_t n, float azimuthRotation, float* u, float* v) {
  const float twoPi = (float)(2.0 * 3.14159265358979323846264338327950288);
  const float invPi = (float)(1.0 / 3.14159265358979323846264338327950288);
  const float inv2Pi = (float)(1.0 / (2.0 * 3.14159265358979323846264338327950288));
  for (size_t i = 0; i < n; ++i) {
    const ohit* hit = &rays[i].h;
    if (hit->flags & O_FLAG_ESCAPED) {
      const ovec3 d = hit->r.direction;
      float theta = acosf(d.y);
      float phi = atan2f(d.z, d.x) + azimuthRotation;
      if (phi < 0.f) phi += twoPi;
      else if (phi > twoPi) phi -= twoPi;
      u[i] = theta * invPi;
      v[i] = phi * inv2Pi;
    } else { u[i] = 0.f; v[i] = 0.f; }
  }
}

static inline float round_through_half(float f) { return o_half_to_float(o_float_to_half_rne(f)); }

/* NifModel.cpp:186-246, 300-327. Features [sin(u*2^j) | sin(v*2^j) | cos(u*2^j) | cos(v*2^j)],
 * u,v normalised to 2*(uv-1). Dense(+bias)(+ReLU) chain; when a layer's row count differs from
 * the activation width the features are appended to the activations first. */
/* threads for a team nobody sized: the cores this process may run on (sched_getaffinity), not the machine's */
static int oracle_default_threads(void) {
  int n = 0;
#if defined(__linux__)
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
#endif
  if (n < 1) n = 1;
  if (n > 64) n = 64;
  return n;
}

/* the weights as the arithmetic sees them (rounded through binary16 once when the model is an fp16 one) */
static float** nif_prepare_weights(const onif* nif) {
  float** W = (float**)malloc(sizeof(float*) * nif->numLayers);
  for (uint32_t l = 0; l < nif->numLayers; ++l) {
    size_t cnt = (size_t)nif->rows[l] * nif->cols[l];
    W[l] = (float*)malloc(sizeof(float) * cnt);
    for (size_t k = 0; k < cnt; ++k) W[l][k] = nif->halfWeightsActs ? round_through_half(nif->kernels[l][k]) : nif->kernels[l][k];
  }
  return W;
}
static void nif_free_weights(const onif* nif, float** W) {
  for (uint32_t l = 0; l < nif->numLayers; ++l) free(W[l]);
  free(W);
}

/* rows with only[i] == 0 are skipped (their bgr is left alone); only == NULL: every row */
static void nif_infer_with(const onif* nif, float** W, const float* u, const float* v, size_t n, float* bgr, const uint8_t* only, int numThreads) {
  const uint32_t E = nif->embeddingDimension;
  const uint32_t F = 4 * E;
  uint32_t maxW = F;
  for (uint32_t l = 0; l < nif->numLayers; ++l) {
    if (nif->rows[l] > maxW) maxW = nif->rows[l];
    if (nif->cols[l] > maxW) maxW = nif->cols[l];
  }
  /* (never more threads than the caller asked for: on a box that grants this process 16 of its 100+ cores an unbounded
   * team oversubscribes them and spends the time spinning at its barriers) */
  if (numThreads < 1) numThreads = oracle_default_threads();
#pragma omp parallel num_threads(numThreads)
  {
    float* feat = (float*)malloc(sizeof(float) * F);
    float* x = (float*)malloc(sizeof(float) * (maxW + F));
    float* y = (float*)malloc(sizeof(float) * (maxW + F));
#pragma omp for schedule(dynamic, 4)
    for (long long i = 0; i < (long long)n; ++i) {
      if (only && !only[i]) continue;
      const float un = (u[i] - 1.f) * 2.f, vn = (v[i] - 1.f) * 2.f;      /* NifModel.cpp:203-205 */
      for (uint32_t j = 0; j < E; ++j) {
        const float coeff = (float)(1u << j);                           /* makeCoefficients, NifModel.cpp:467-473 */
        float pu = un * coeff, pv = vn * coeff;
        if (nif->halfFeatures) { pu = round_through_half(pu); pv = round_through_half(pv); }
        float su = sinf(pu), sv = sinf(pv), cu = cosf(pu), cv = cosf(pv);
        if (nif->halfFeatures) { su = round_through_half(su); sv = round_through_half(sv); cu = round_through_half(cu); cv = round_through_half(cv); }
        feat[j] = su; feat[E + j] = sv; feat[2 * E + j] = cu; feat[3 * E + j] = cv;
      }
      uint32_t width = F;
      memcpy(x, feat, sizeof(float) * F);
      for (uint32_t l = 0; l < nif->numLayers; ++l) {
        const uint32_t R = nif->rows[l], C = nif->cols[l];
        if (width != R) { memcpy(x + width, feat, sizeof(float) * F); width += F; }
        for (uint32_t c = 0; c < C; ++c) y[c] = 0.f;
        for (uint32_t r = 0; r < R; ++r) {
          const float xr = nif->halfWeightsActs ? round_through_half(x[r]) : x[r];
          const float* wrow = W[l] + (size_t)r * C;
          for (uint32_t c = 0; c < C; ++c) y[c] += xr * wrow[c];
        }
        if (nif->biases && nif->biases[l]) for (uint32_t c = 0; c < C; ++c) y[c] += nif->biases[l][c];
        if (nif->relu[l]) for (uint32_t c = 0; c < C; ++c) y[c] = y[c] > 0.f ? y[c] : 0.f;
        float* tmp = x; x = y; y = tmp;
        width = C;
      }
      for (uint32_t c = 0; c < 3; ++c) {
        float o = x[c] * nif->maxValue + nif->mean[c];                   /* NifModel.cpp:229-238 */
        if (nif->logTonemap) o = expf(o);
        bgr[3 * i + c] = o;
      }
    }
    free(feat); free(x); free(y);
  }
}

void o_nif_infer(const onif* nif, const float* u, const float* v, size_t n, float* bgr) {
  float** W = nif_prepare_weights(nif);
  nif_infer_with(nif, W, u, v, n, bgr, NULL, 0);
  nif_free_weights(nif, W);
}

/* PostProcessEscapedRays, codelets/TraceCodelets.cpp:361-382 */
void o_apply_env(otrace* rays, size_t n, const float* bgr) {
  for (size_t i = 0; i < n; ++i) {
    ohit* hit = &rays[i].h;
    if (hit->flags & O_FLAG_ESCAPED) {
      const ovec3 env = V(bgr[3 * i + 2], bgr[3 * i + 1], bgr[3 * i + 0]);
      rays[i].rgb = vadd(rays[i].rgb, vmul(hit->throughput, env));
    }
  }
}

/* src/IpuScene.cpp:571-583: Repeat(spp){ PathTrace(1 sample); PreProcess; NIF; PostProcess } */
void o_path_trace_nif_pixel_rng(const oscene* sc, const onif* nif, float azimuthRotation,
                                otrace* rays, size_t n, int numThreads, ostats* st) {
  const float tanTheta = fov_tan_theta(sc);
  uint64_t* states = (uint64_t*)malloc(sizeof(uint64_t) * 2 * n);
  float* u = (float*)malloc(sizeof(float) * n);
  float* v = (float*)malloc(sizeof(float) * n);
  float* bgr = (float*)malloc(sizeof(float) * 3 * n);
  /* Same stream definition as the plain render (o_segment_samples): a new stream at every segment boundary, the
   * finished segments' partial sums added in segment order, segment 0 accumulating onto rgb_in directly. */
  const uint32_t segLen = o_segment_samples(sc->samplesPerPixel);
  ovec3* total = (ovec3*)malloc(sizeof(ovec3) * (n ? n : 1));
  ostats tot = {0, 0, 0, 0};
  if (numThreads < 1) numThreads = 1;
  /* (the weights are prepared once for the render, and the MLP runs for the rays that escaped only: PostProcessEscapedRays
   * reads no other ray's result - the same values as evaluating every ray every sample, in a fraction of the time) */
  float** W = nif_prepare_weights(nif);
  uint8_t* escaped = (uint8_t*)malloc(n ? n : 1);
  for (uint32_t smp = 0; smp < sc->samplesPerPixel; ++smp) {
    if (smp % segLen == 0) {
      const uint32_t segment = smp / segLen;
      for (size_t i = 0; i < n; ++i) {
        pixel_stream_seed_segment(states + 2 * i, sc->rngSeed, rays[i].u, rays[i].v, segment);
        if (segment == 1) total[i] = rays[i].rgb;
        else if (segment > 1) total[i] = vadd(total[i], rays[i].rgb);
        if (segment > 0) rays[i].rgb = V(0.f, 0.f, 0.f);
      }
    }
#pragma omp parallel num_threads(numThreads)
    {
      ostats loc = {0, 0, 0, 0};
#pragma omp for schedule(dynamic, 64)
      for (long long i = 0; i < (long long)n; ++i) {
        sample_camera_ray(sc, &rays[i], tanTheta, states + 2 * i);
        path_trace_one(sc, &rays[i], uni_state, states + 2 * i, &loc);
      }
#pragma omp critical
      { tot.casts += loc.casts; tot.nodesVisited += loc.nodesVisited; tot.leafTests += loc.leafTests; tot.paths += loc.paths; }
    }
    o_escaped_uv(rays, n, azimuthRotation, u, v);
    for (size_t i = 0; i < n; ++i) escaped[i] = (rays[i].h.flags & O_FLAG_ESCAPED) ? 1 : 0;
    nif_infer_with(nif, W, u, v, n, bgr, escaped, numThreads);
    o_apply_env(rays, n, bgr);
  }
  nif_free_weights(nif, W); free(escaped);
  if (sc->samplesPerPixel > segLen)
    for (size_t i = 0; i < n; ++i) rays[i].rgb = vadd(total[i], rays[i].rgb);
  if (st) { st->casts += tot.casts; st->nodesVisited += tot.nodesVisited; st->leafTests += tot.leafTests; st->paths += tot.paths; }
  free(states); free(u); free(v); free(bgr); free(total);
}
