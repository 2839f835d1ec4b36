"""Pins the CPU oracle (oracle/ray_oracle.c) before anything is compared against it.

Three kinds of pin, strongest first:
  [REF]    golden vectors produced by the reference's own L0 code (tests/golden/ref_l0_vectors.npz,
           generator tests/golden/gen_ref_vectors.py) and, where oracle/_ref exists, live calls;
  [PROBE]  values the reference itself printed, recorded in SURVEY.md §7/§8a/§8c;
  [XCHECK] independent mathematics (numpy float16, float64 Moller-Trumbore, brute force over all
           primitives) for the functions whose reference files cannot be compiled in this image.
All comparisons are bit-exact unless a tolerance is written next to them.
"""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

import ipu_ray_lib_amd as irl
import oracle_lib as ol
from oracle_lib import Vec3, Ray, Shear, Sphere, Disc

GOLD = np.load(Path(__file__).parent / "golden" / "ref_l0_vectors.npz")
f32 = C.c_float


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def v3(a):
    return Vec3(float(a[0]), float(a[1]), float(a[2]))


@pytest.fixture(scope="module")
def o():
    return ol.lib()


# ------------------------------------------------------------------------------------------ [REF]
def test_sincos_matches_reference_golden(o):
    xs = GOLD["sincos_x"]
    s = np.zeros_like(xs); c = np.zeros_like(xs)
    for i, x in enumerate(xs):
        a, b = f32(), f32()
        o.o_sincos(float(x), C.byref(a), C.byref(b))
        s[i], c[i] = a.value, b.value
    assert np.array_equal(bits(s), bits(GOLD["sincos_s"]))
    assert np.array_equal(bits(c), bits(GOLD["sincos_c"]))
    # SURVEY.md §2 row 3 probe: differs from libm in the last digits
    a, b = f32(), f32()
    o.o_sincos(float(np.float32(np.pi / 8)), C.byref(a), C.byref(b))
    assert "%.9f %.9f" % (a.value, b.value) == "0.382683396 0.923879564"


def test_maxi_maxc_select_the_minimum(o):
    v = GOLD["maxi_v"]
    got_i = np.array([o.o_maxi(v3(x)) for x in v], dtype=np.uint32)
    got_c = np.array([o.o_maxc(v3(x)) for x in v], dtype=np.float32)
    assert np.array_equal(got_i, GOLD["maxi_i"])
    assert np.array_equal(bits(got_c), bits(GOLD["maxi_c"]))
    assert o.o_maxi(Vec3(1, 2, 3)) == 0 and o.o_maxc(Vec3(1, 2, 3)) == 1.0   # SURVEY §7 probe


def test_xoshiro_matches_reference_golden(o):
    for i, sd in enumerate(GOLD["xo_seeds"]):
        st = (C.c_uint64 * 2)()
        o.o_xoshiro_seed(st, int(sd))
        assert (st[0], st[1]) == tuple(int(x) for x in GOLD["xo_states"][i])
        assert [o.o_xoshiro_next(st) for _ in range(16)] == [int(x) for x in GOLD["xo_next"][i]]
        o.o_xoshiro_seed(st, int(sd))
        u = np.array([o.o_xoshiro_uniform01(st) for _ in range(16)], dtype=np.float32)
        assert np.array_equal(bits(u), bits(GOLD["xo_uniform"][i]))
        o.o_xoshiro_seed(st, int(sd)); o.o_xoshiro_jump(st)
        assert (st[0], st[1]) == tuple(int(x) for x in GOLD["xo_jump"][i])
    assert [o.o_splitmix64(int(z)) for z in GOLD["sm_in"]] == [int(z) for z in GOLD["sm_out"]]
    # the two replica seeds of src/IpuScene.cpp:649-653 for --seed 1442 (SURVEY §8c probe; the probe's
    # printf evaluated its arguments right to left, so the set is what is pinned)
    st = (C.c_uint64 * 2)(); o.o_xoshiro_seed(st, 1442)
    assert {o.o_xoshiro_next(st), o.o_xoshiro_next(st)} == {0x900f7405f9373888, 0x8fb46db66246de50}


def test_sampling_and_bxdfs_match_reference_golden(o):
    u1, u2, n, d = GOLD["bx_u1"], GOLD["bx_u2"], GOLD["bx_n"], GOLD["bx_d"]
    cnt = len(u1)
    disc = np.zeros((cnt, 2), np.float32); hemi = np.zeros((cnt, 3), np.float32); diff = np.zeros((cnt, 3), np.float32)
    refl = np.zeros((cnt, 3), np.float32); ortho = np.zeros((cnt, 6), np.float32)
    for i in range(cnt):
        a, b = f32(), f32()
        o.o_sample_disc_concentric(float(u1[i]), float(u2[i]), C.byref(a), C.byref(b)); disc[i] = (a.value, b.value)
        hemi[i] = o.o_cosine_sample_hemisphere(float(u1[i]), float(u2[i])).t()
        diff[i] = o.o_sample_diffuse(v3(n[i]), float(u1[i]), float(u2[i])).t()
        refl[i] = o.o_reflect(v3(d[i]), v3(n[i])).t()
        b0, b1 = Vec3(), Vec3(); o.o_orthonormal_system(v3(n[i]), C.byref(b0), C.byref(b1)); ortho[i] = b0.t() + b1.t()
    for got, key in ((disc, "bx_disc"), (hemi, "bx_hemi"), (diff, "bx_diffuse"), (refl, "bx_reflect"), (ortho, "bx_ortho")):
        assert np.array_equal(bits(got), bits(GOLD[key])), key
    # SURVEY §8c probe: n=(0,1,0), u=(.25,.75)
    p = o.o_sample_diffuse(Vec3(0, 1, 0), 0.25, 0.75)
    assert "%.9f %.9f %.9f" % p.t() == "-0.353553385 0.866025388 0.353553385"


def test_dielectric_schlick_refract_roulette_match_reference_golden(o):
    cosT, ri = GOLD["bx_cos"], GOLD["bx_ri"]
    sch = np.array([o.o_schlick(float(a), float(b)) for a, b in zip(cosT, ri)], dtype=np.float32)
    assert np.array_equal(bits(sch), bits(GOLD["bx_schlick"]))
    n, d, u1, ior = GOLD["bx_n"], GOLD["bx_d"], GOLD["bx_u1"], GOLD["bx_ior"]
    cnt = len(u1)
    die = np.zeros((cnt, 3), np.float32); flag = np.zeros(cnt, np.int32); refr = np.zeros((cnt, 3), np.float32)
    for i in range(cnt):
        ray = Ray(Vec3(0, 0, 0), 0.0, v3(d[i]), float("inf"))
        out = Vec3()
        flag[i] = o.o_dielectric(C.byref(ray), v3(n[i]), float(ior[i]), float(u1[i]), C.byref(out)); die[i] = out.t()
        ndotr = float(np.float32(np.dot(n[i].astype(np.float64), d[i].astype(np.float64))))
        refr[i] = o.o_refract(v3(d[i]), v3(n[i]), ndotr, float(ri[i])).t()
    assert np.array_equal(flag, GOLD["bx_dielectric_refracted"])
    assert np.array_equal(bits(die), bits(GOLD["bx_dielectric"]))
    assert np.array_equal(bits(refr), bits(GOLD["bx_refract"]))
    tp, ur = GOLD["rr_tp"], GOLD["rr_u"]
    tp_out = np.zeros_like(tp); stop = np.zeros(len(ur), np.int32)
    for i in range(len(ur)):
        t = v3(tp[i]); stop[i] = o.o_evaluate_roulette(float(ur[i]), C.byref(t)); tp_out[i] = t.t()
    assert np.array_equal(stop, GOLD["rr_stop"]) and np.array_equal(bits(tp_out), bits(GOLD["rr_tp_out"]))
    # SURVEY §7 probe: (0.9,0.5,0.2), u=0.1 -> survival p is the MIN channel -> (4.5,2.5,1)
    t = Vec3(0.9, 0.5, 0.2)
    assert o.o_evaluate_roulette(0.1, C.byref(t)) == 0
    assert np.allclose(t.t(), (4.5, 2.5, 1.0), rtol=1e-6)   # tolerance: probe printed 2 digits


def test_reference_struct_layout():
    lay = [int(x) for x in GOLD["layout"]]
    # sizeof Vec3fa, alignof, Ray, HitRecord, TraceResult, offsets p, h, primID, normal, throughput, geomID, flags
    assert lay == [12, 4, 32, 64, 84, 12, 20, 32, 36, 48, 60, 62]
    assert irl.TRACE_RESULT.itemsize == 84 and irl.TRACE_RESULT.fields["u"][1] == 12 and irl.TRACE_RESULT.fields["h"][1] == 20
    assert irl.HIT.fields["primID"][1] == 32 and irl.HIT.fields["normal"][1] == 36 and irl.HIT.fields["throughput"][1] == 48
    assert irl.HIT.fields["geomID"][1] == 60 and irl.HIT.fields["flags"][1] == 62


def test_reference_material_layout_defaults_and_constructor():
    """include/Material.hpp:8-35 compiled from the reference: size, field offsets, the Type values, the default ior
    1.52 and `emissive = emission.isNonZero()`. mi_material (include/mi_raylib.h), the numpy dtype the tests use and
    the materials of the built-in scenes must have exactly those bytes (the three bytes after `emissive` are padding
    the reference never writes; they are masked)."""
    size, align, o_albedo, o_ior, o_emission, o_type, o_emissive, diffuse, specular, refractive, type_size = [int(x) for x in GOLD["material_layout"]]
    assert (size, align, type_size) == (36, 4, 4) and (diffuse, specular, refractive) == (0, 1, 2)
    M = irl.MATERIAL
    assert M.itemsize == size
    assert (M.fields["albedo"][1], M.fields["ior"][1], M.fields["emission"][1], M.fields["type"][1], M.fields["emissive"][1]) == (o_albedo, o_ior, o_emission, o_type, o_emissive)
    dflt = GOLD["material_default"].view(M)[0]
    assert dflt["ior"] == np.float32(1.52) and dflt["type"] == 0 and dflt["emissive"] == 0 and not np.any(GOLD["material_default"][:12]) and not np.any(GOLD["material_default"][16:28])
    assert list(GOLD["material_default_init"][33:]) == [False] * 3 and GOLD["material_default_init"][:33].all()      # bytes 33..35 are padding
    # Material(albedo, emission, type): the product's own constructor path (scene_builtin.cpp makeMaterial through
    # HostScene.from_arrays would need a scene; the numpy restatement below is what every test scene is built with)
    for row, want, init in zip(GOLD["material_in"], GOLD["material_bytes"], GOLD["material_init"]):
        m = np.zeros(1, M)
        m["albedo"] = tuple(row[:3]); m["emission"] = tuple(row[3:6]); m["type"] = int(row[6]); m["ior"] = 1.52
        m["emissive"] = 1 if np.any(row[3:6] != 0) else 0                                            # isNonZero: -0.0 counts as zero, 1e-30 does not
        assert np.array_equal(m.view(np.uint8)[init], want[init]), row
    # the built-in scenes' material tables are such objects: emissive <=> emission != 0, ior 1.52, types 0..2
    for name in ("box", "spheres", "monkey"):
        hs = irl.HostScene.builtin(name)               # (the arrays are views into the host scene: keep it alive)
        mats = hs.materials
        assert mats.itemsize == size and np.all(mats["ior"] == np.float32(1.52)) and set(np.unique(mats["type"])) <= {0, 1, 2}
        em = np.stack([mats["emission"][k] for k in "xyz"], 1)
        assert np.array_equal(mats["emissive"] != 0, np.any(em != 0, axis=1))
    assert np.array_equal(GOLD["nonzero_out"], [0, 0, 1, 1, 1])


def test_reference_constructors_and_hit_constants(o):
    """Ray(o, d), HitRecord(o, d), TraceResult(hit, uv), PixelCoord() and the HitRecord constants
    (include/embree_utils/geometry.hpp:199-259) compiled from the reference, against the records the product's host
    code (mi_init_ray_stream) and the oracle (o_init_ray_stream) start every render from. HitRecord(o, d) does not
    initialise `throughput` (geometry.hpp:236-242): those 12 bytes are excluded."""
    assert [int(x) for x in GOLD["hit_constants"]] == [irl.FLAG_ERROR, irl.FLAG_ESCAPED, irl.INVALID_GEOM, irl.INVALID_PRIM] == [1, 2, 0xFFFF, 0xFFFFFFFF]
    assert np.all(np.isneginf(GOLD["pixelcoord_default"]))
    keep = np.ones(84, bool); keep[20 + 48:20 + 60] = False                    # TraceResult.h.throughput
    for od, uv, rb, hb, tb in zip(GOLD["ctor_od"], GOLD["ctor_uv"], GOLD["ray_bytes"], GOLD["hit_bytes"], GOLD["trace_bytes"]):
        t = np.zeros(1, irl.TRACE_RESULT)
        t["u"], t["v"] = np.float32(uv[0]), np.float32(uv[1])                 # PixelCoord(u32, u32) converts to float
        t["h"]["r"]["origin"] = tuple(od[:3]); t["h"]["r"]["direction"] = tuple(od[3:])
        t["h"]["r"]["tMin"] = 0.0; t["h"]["r"]["tMax"] = np.inf
        t["h"]["primID"] = irl.INVALID_PRIM; t["h"]["geomID"] = irl.INVALID_GEOM; t["h"]["flags"] = 0
        t["h"]["normal"] = (0.0, 0.0, 1.0)
        got = t.view(np.uint8).reshape(-1)
        assert np.array_equal(got[keep], tb[keep])
        assert np.array_equal(got[20:52], rb) and np.array_equal(got[20:84][keep[20:]], hb[keep[20:]])
    # the records a render starts from: host library and oracle, every field but the camera direction / pixel
    s = irl.HostScene.builtin("box-simple"); s.desc.set_image(24, 16)
    a = s.init_ray_stream()
    b = np.zeros(a.size, irl.TRACE_RESULT); o.o_init_ray_stream(C.byref(s.desc), b.ctypes.data)
    assert a.tobytes() == b.tobytes()
    ref = GOLD["trace_bytes"][0].view(irl.TRACE_RESULT)[0]
    for rec in (a[0], a[-1]):
        assert rec["h"]["primID"] == ref["h"]["primID"] and rec["h"]["geomID"] == ref["h"]["geomID"] and rec["h"]["flags"] == ref["h"]["flags"]
        assert rec["h"]["normal"] == ref["h"]["normal"] and rec["h"]["r"]["tMin"] == ref["h"]["r"]["tMin"] and rec["h"]["r"]["tMax"] == ref["h"]["r"]["tMax"]
        assert tuple(rec["rgb"]) == tuple(ref["rgb"]) == (0.0, 0.0, 0.0)


def test_reference_permute_abs_and_bounds():
    """Vec3fa::permute / abs and Bounds3d (geometry.hpp:126-197) compiled from the reference: permute(ix, iy, iz)
    is (c[ix], c[iy], c[iz]) - the axis rotation the ray shear and the triangle test apply as
    permute(kx, ky, kz) with kx = (kz + 1) % 3, ky = (kx + 1) % 3 (src/Primitives.cpp:5-22, src/Mesh.cpp:12-20),
    which the kernels spell as selects (csrc/trace_kernels.hpp permute_kz); an empty Bounds3d is (+inf, -inf), union is
    component-wise min / max, centroid is (max + min) * 0.5f."""
    for v, outs, ab in zip(GOLD["permute_v"], GOLD["permute_out"], GOLD["abs_out"]):
        for k in range(27):
            assert np.array_equal(bits(outs[k]), bits(v[[k // 9, (k // 3) % 3, k % 3]]))
        for kz in range(3):
            kx = (kz + 1) % 3; ky = (kx + 1) % 3
            r1, r2 = kz == 0, kz == 1                                     # permute_kz's selects
            sel = np.array([v[1] if r1 else (v[2] if r2 else v[0]), v[2] if r1 else (v[0] if r2 else v[1]), v[0] if r1 else (v[1] if r2 else v[2])], np.float32)
            assert np.array_equal(bits(sel), bits(outs[kx * 9 + ky * 3 + kz]))
        assert np.array_equal(bits(ab), bits(np.abs(v)))
    assert np.array_equal(GOLD["bounds_default"], np.array([np.inf] * 3 + [-np.inf] * 3, np.float32))
    lo, hi = GOLD["bounds_lo"], GOLD["bounds_hi"]
    umin = np.minimum(lo[:, 0], lo[:, 1]); umax = np.maximum(hi[:, 0], hi[:, 1])
    assert np.array_equal(bits(GOLD["bounds_union"][:, :3]), bits(umin)) and np.array_equal(bits(GOLD["bounds_union"][:, 3:6]), bits(umax))
    assert np.array_equal(bits(GOLD["bounds_union"][:, 6:]), bits((umax + umin) * np.float32(0.5)))


@pytest.mark.skipif(ol.ref_lib() is None, reason="oracle/_ref only exists where /root/reference was")
def test_oracle_against_live_reference_dense(o):
    r = ol.ref_lib()
    rng = np.random.default_rng(7)
    for x in rng.uniform(-30, 30, 20000).astype(np.float32):
        a, b, c, d = f32(), f32(), f32(), f32()
        o.o_sincos(float(x), C.byref(a), C.byref(b)); r.ref_sincos(float(x), C.byref(c), C.byref(d))
        assert (bits([a.value]), bits([b.value])) == (bits([c.value]), bits([d.value]))
    for _ in range(5000):
        n = rng.normal(size=3); n /= np.linalg.norm(n); n = n.astype(np.float32)
        u1, u2 = (float(np.float32(x)) for x in rng.random(2))
        out = (f32 * 3)(); r.ref_sample_diffuse((f32 * 3)(*[float(x) for x in n]), u1, u2, out)
        assert np.array_equal(bits(o.o_sample_diffuse(v3(n), u1, u2).t()), bits(list(out)))


# ---------------------------------------------------------------------------------------- [PROBE]
def test_probe_values_from_survey(o):
    assert "%.9g" % o.o_ray_epsilon() == "8.94069672e-05"                      # §8a row a8
    d = o.o_pixel_to_ray_dir(0.0, 0.0, 1440.0, 1440.0, float(np.tan(np.float32(np.pi / 8))))
    s, c = f32(), f32(); o.o_sincos(float(np.float32(np.float32(np.pi / 4) / np.float32(2))), C.byref(s), C.byref(c))
    d = o.o_pixel_to_ray_dir(0.0, 0.0, 1440.0, 1440.0, s.value / c.value)
    assert "%.9f %.9f %.9f" % d.t() == "-0.357406706 0.357406706 -0.862856209"  # §8a row a14
    assert [o.o_round_to_half_not_smaller(x) for x in (0.1, 548.8, 1e-8)] == [0x2E67, 0x604A, 0x0001]   # §8c
    assert o.o_half_to_float(0x604A) == 549.0
    ray = Ray(Vec3(0, 0, 0), 0.0, v3(np.array([.1, .2, .9]) / np.linalg.norm([.1, .2, .9])), float("inf"))
    sh = Shear(); o.o_ray_shear(C.byref(ray), C.byref(sh))
    assert (sh.ix, sh.iy, sh.iz) == (1, 2, 0)                                   # §8c: d∝(.1,.2,.9) -> perm (1,2,0)
    assert np.allclose((sh.sx, sh.sy, sh.sz), (-2.0, -9.0, 9.2736), rtol=2e-5)  # tolerance: probe printed 5 digits
    r2 = Ray(Vec3(100, -200, 300), 0.0, Vec3(0, 0, 1), float("inf"))
    o.o_offset_ray(C.byref(r2), Vec3(0, 0, 1))
    assert np.isclose(r2.origin.z - 300.0, 101 * o.o_ray_epsilon(), rtol=2e-3)  # §7: moves by 101 eps, not 301 eps


# --------------------------------------------------------------------------------------- [XCHECK]
def test_half_conversions_against_numpy_float16(o):
    allh = np.arange(0x10000, dtype=np.uint16)
    want = allh.view(np.float16).astype(np.float32)
    got = np.array([o.o_half_to_float(int(h)) for h in allh], dtype=np.float32)
    finite = np.isfinite(want)
    assert np.array_equal(bits(got[finite]), bits(want[finite]))
    assert np.all(np.isnan(got[np.isnan(want)])) and np.array_equal(got[np.isinf(want)], want[np.isinf(want)])
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(0, 70000, 20000), 10.0 ** rng.uniform(-9, 5, 20000), np.abs(want[finite][::7]).astype(np.float64),   # extents are non-negative
                        
                         [0, 65504, 65519.99, 65520, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8]]).astype(np.float32)
    with np.errstate(over="ignore"):
        rne = xs.astype(np.float16).view(np.uint16)
    got_rne = np.array([o.o_float_to_half_rne(float(x)) for x in xs], dtype=np.uint16)
    assert np.array_equal(got_rne, rne)
    up = np.array([o.o_round_to_half_not_smaller(float(x)) for x in xs], dtype=np.uint16)
    upf = up.view(np.float16).astype(np.float32)
    ok = xs <= 65504
    assert np.all(upf[ok] >= xs[ok])                       # never smaller
    prev = (up[ok] - 1).astype(np.uint16).view(np.float16).astype(np.float32)
    assert np.all((prev < xs[ok]) | (up[ok] == 0))         # and the tightest such half


def test_slab_test_semantics(o):
    inf = float("inf")
    def slab(inv, org, lo, hi, t0=0.0, t1=inf):
        a, b = f32(t0), f32(t1)
        return o.o_slab(inv, org, lo, hi, C.byref(a), C.byref(b)), a.value, b.value
    ok, t0, t1 = slab(1.0, 0.0, 2.0, 4.0)
    assert ok == 1 and t0 == 2.0 and t1 == np.float32(4.0) * np.float32(1 + 2 * o.o_gamma(3))
    assert slab(-1.0, 0.0, 2.0, 4.0)[0] == 0                # box behind the ray
    assert slab(inf, 1.0, 2.0, 4.0)[0] == 1                 # axis-parallel ray: (+inf,+inf) interval passes like the reference
    assert slab(inf, 3.0, 2.0, 4.0)[0] == 1                 # inside the slab: (-inf, +inf)
    ok, t0, t1 = slab(inf, 2.0, 2.0, 4.0)                   # origin exactly on the plane: 0*inf = NaN is ignored
    assert ok == 1 and t0 == 0.0 and t1 == inf
    assert slab(1.0, 0.0, 2.0, 4.0, 0.0, 1.0)[0] == 0       # pruned by the closest hit so far (t1 = -inf probe)
    assert slab(1.0, 0.0, 2.0, 4.0, 0.0, -inf)[0] == 0


def _moller_trumbore(p0, p1, p2, org, d):
    e1, e2 = p1 - p0, p2 - p0
    h = np.cross(d, e2); a = e1.dot(h)
    if abs(a) < 1e-14:
        return None
    f = 1.0 / a; s = org - p0; u = f * s.dot(h)
    q = np.cross(s, e1); v = f * d.dot(q)
    if u < 0 or v < 0 or u + v > 1:
        return None
    t = f * e2.dot(q)
    return t if t > 0 else None


def test_triangle_intersection_against_float64(o):
    rng = np.random.default_rng(11)
    hits = 0
    for _ in range(4000):
        p = rng.uniform(-5, 5, (3, 3)).astype(np.float32)
        org = rng.uniform(-8, 8, 3).astype(np.float32)
        tgt = (p[0] * .3 + p[1] * .3 + p[2] * .4) + rng.normal(scale=2.0, size=3)
        d = (tgt - org); d = (d / np.linalg.norm(d)).astype(np.float32)
        ray = Ray(v3(org), 0.0, v3(d), float("inf")); sh = Shear(); o.o_ray_shear(C.byref(ray), C.byref(sh))
        bary = (f32 * 3)()
        t = o.o_intersect_triangle(v3(p[0]), v3(p[1]), v3(p[2]), C.byref(sh), float("inf"), bary)
        ref = _moller_trumbore(*(p.astype(np.float64)), org.astype(np.float64), d.astype(np.float64))
        if ref is not None and ref > 1e-3:
            # Hits must agree with exact geometry. Tolerance 2e-5 * t / |d[kz]|: the reference shears along the
            # SMALLEST direction component (maxi() quirk, Primitives.cpp:9), amplifying float32 rounding by 1/|d[kz]|.
            b = np.array(list(bary))
            if t != 0.0:
                hits += 1
                assert abs(t - ref) <= 2e-5 * ref / abs(float(d[sh.iz]))
                assert abs(b.sum() - 1.0) < 1e-5
        elif ref is None and t != 0.0:
            b = np.array(list(bary))
            assert b.min() > -1e-4   # only edge-grazing rays may disagree
    assert hits > 500


def test_sphere_and_disc_semantics(o):
    s = Sphere(0, 0, -10, 2.0)
    ray = Ray(Vec3(0, 0, 0), 0.0, Vec3(0, 0, -1), float("inf"))
    assert o.o_sphere_intersect(C.byref(s), C.byref(ray)) == 8.0            # SURVEY §8c probe t=8
    inside = Ray(Vec3(0, 0, -10), 0.0, Vec3(0, 0, -1), float("inf"))
    assert o.o_sphere_intersect(C.byref(s), C.byref(inside)) == 2.0          # t0<tMin -> far root
    behind = Ray(Vec3(0, 0, -11), 0.0, Vec3(0, 0, -1), float("inf"))         # inside, centre behind: tca<0 -> miss (§8a-bis 10)
    assert o.o_sphere_intersect(C.byref(s), C.byref(behind)) == 0.0
    d = Disc(0, 0, 1, 3.0, 0, 0, -5)
    assert o.o_disc_intersect(C.byref(d), C.byref(ray)) == 5.0               # SURVEY §8c probe t=5
    off = Ray(Vec3(4, 0, 0), 0.0, Vec3(0, 0, -1), float("inf"))
    assert o.o_disc_intersect(C.byref(d), C.byref(off)) == 0.0
    par = Ray(Vec3(0, 0, 0), 0.0, Vec3(1, 0, 0), float("inf"))
    assert o.o_disc_intersect(C.byref(d), C.byref(par)) == 0.0


def test_logdet_and_gauss(o):
    xs = np.concatenate([np.linspace(2.0 ** -25, 1.0, 5000), 2.0 ** -np.arange(0, 25.0)]).astype(np.float32)
    got = np.array([o.o_logf_det(float(x)) for x in xs])
    assert np.max(np.abs(got - np.log(xs.astype(np.float64)))) < 2e-6 * np.max(np.abs(np.log(xs.astype(np.float64))))  # abs tol 3.5e-5
    st = (C.c_uint64 * 2)(); o.o_xoshiro_seed(st, 99)
    g = np.zeros((20000, 2), np.float32)
    for i in range(len(g)):
        a, b = f32(), f32(); o.o_gauss2(st, C.byref(a), C.byref(b)); g[i] = (a.value, b.value)
    assert abs(g.mean()) < 0.02 and abs(g.std() - 1.0) < 0.02 and abs(np.corrcoef(g[:, 0], g[:, 1])[0, 1]) < 0.03


@pytest.mark.parametrize("name", ["box-simple", "box", "spheres"])
def test_bvh_queries_against_brute_force(o, name):
    """CompactBvh::intersect/occluded restatement vs testing EVERY primitive (no BVH): same closest t,
    and the same primitive unless two primitives tie exactly."""
    scene = irl.HostScene.builtin(name)
    d = scene.desc
    d.set_image(64, 64)
    rays = scene.init_ray_stream()
    prim_list = []
    for g, ref in enumerate(scene.geometry):
        if ref["type"] == 0:
            prim_list += [(g, p) for p in range(scene.mesh_info[ref["index"]]["numTriangles"])]
        else:
            prim_list.append((g, 0))
    # brute force through single-leaf pseudo scenes is expensive in Python: use the oracle's own leaf tests
    verts, tris, info = scene.verts, scene.tris, scene.mesh_info
    rng = np.random.default_rng(5)
    pick = rng.choice(rays.size, 150, replace=False)
    for i in pick:
        r = rays[i]["h"]["r"]
        ray = Ray(Vec3(*[float(r["origin"][k]) for k in "xyz"]), 0.0, Vec3(*[float(r["direction"][k]) for k in "xyz"]), float("inf"))
        got = o.o_bvh_intersect(C.byref(d), C.byref(ray), None)
        sh = Shear(); o.o_ray_shear(C.byref(ray), C.byref(sh))
        best_t = np.inf
        for g, p in prim_list:
            ref = scene.geometry[g]
            if ref["type"] == 0:
                mi = info[ref["index"]]
                tri = tris[mi["firstIndex"] + p]
                pv = [verts[mi["firstVertex"] + int(k)] for k in tri]
                bary = (f32 * 3)()
                t = o.o_intersect_triangle(*[Vec3(float(q["x"]), float(q["y"]), float(q["z"])) for q in pv], C.byref(sh), float("inf"), bary)
            elif ref["type"] == 1:
                sp = scene.spheres[ref["index"]]
                t = o.o_sphere_intersect(C.byref(Sphere(*[float(sp[k]) for k in ("x", "y", "z", "radius")])), C.byref(ray))
            else:
                dc = scene.discs[ref["index"]]
                t = o.o_disc_intersect(C.byref(Disc(*[float(dc[k]) for k in ("nx", "ny", "nz", "r", "cx", "cy", "cz")])), C.byref(ray))
            if t > 0.0 and t < best_t:
                best_t = t
        if np.isinf(best_t):
            assert got.hit == 0
        else:
            assert got.hit == 1 and got.t == np.float32(best_t)


def test_double_fallback_switch_restates_mesh_cpp_38_51(o):
    """ALLOW_DOUBLE_FALLBACK=1 (CMakeLists.txt:13,34-41; src/Mesh.cpp:38-51) as a switch of the oracle: when a binary32
    edge function is exactly zero, all three are recomputed from binary64 products and differences. Checked against a
    numpy restatement of those lines on rays that pass through a triangle edge up to rounding (p2.xy = fl(k * p1.xy),
    ray (0,0,0) -> (0,0,-1): no permutation, no shear, the edge functions are those of the vertex coordinates)."""
    import oracle_lib as ol
    rng = np.random.default_rng(5)
    ray = Ray(v3((0, 0, 0)), 0.0, v3((0, 0, -1)), float("inf")); sh = Shear(); o.o_ray_shear(C.byref(ray), C.byref(sh))
    assert (sh.ix, sh.iy, sh.iz) == (0, 1, 2) and sh.sx == 0 and sh.sy == 0
    took_branch = changed = 0
    for _ in range(400):
        p1 = (rng.uniform(0.5, 2.0, 2) * rng.choice([-1, 1], 2)).astype(np.float32)
        p2 = (p1 * np.float32(-rng.uniform(0.5, 2.0))).astype(np.float32)
        p0 = rng.uniform(-3, 3, 2).astype(np.float32)
        P = [(float(q[0]), float(q[1]), -4.0) for q in (p0, p1, p2)]
        bary = (f32 * 3)()
        t0 = o.o_intersect_triangle(v3(P[0]), v3(P[1]), v3(P[2]), C.byref(sh), float("inf"), bary)
        with ol.double_fallback():
            t1 = o.o_intersect_triangle(v3(P[0]), v3(P[1]), v3(P[2]), C.byref(sh), float("inf"), bary)
        assert o.o_get_double_fallback() == 0
        # numpy: binary32 edge functions, then the binary64 branch
        x = np.array([q[0] for q in P], np.float32); y = np.array([q[1] for q in P], np.float32)
        e = [np.float32(np.float32(x[1] * y[2]) - np.float32(y[1] * x[2])), np.float32(np.float32(x[2] * y[0]) - np.float32(y[2] * x[0])),
             np.float32(np.float32(x[0] * y[1]) - np.float32(y[0] * x[1]))]
        if any(v == 0 for v in e):
            took_branch += 1
            X, Y = x.astype(np.float64), y.astype(np.float64)
            e64 = [np.float32(Y[2] * X[1] - X[2] * Y[1]), np.float32(Y[0] * X[2] - X[0] * Y[2]), np.float32(Y[1] * X[0] - X[1] * Y[0])]
            miss32 = (min(e) < 0) and (max(e) > 0)
            miss64 = (min(e64) < 0) and (max(e64) > 0)
            assert (t0 == 0.0) == bool(miss32 or sum(e) == 0) or t0 == 0.0      # (a hit may still be rejected further down)
            if miss64:
                assert t1 == 0.0
            if (t0 == 0.0) != (t1 == 0.0):
                changed += 1
        else:
            assert t0 == t1
    assert took_branch > 100 and changed > 0
