#!/usr/bin/env python3
"""Generates tests/golden/ref_l0_vectors.npz from the REFERENCE's own code.

Run in the build container only (needs /root/reference): `make -C oracle ref` compiles the
reference's Eigen-free L0 sources where they lie (ext/math/sincos.cpp, include/xoshiro.hpp,
include/embree_utils/geometry.hpp, include/geometric_sampling.hpp, include/BxDF.hpp, include/Material.hpp) behind the
thin extern "C" driver oracle/ref_driver.cpp. This script drives that library with seeded inputs
and stores inputs + outputs as DATA. No reference source text is stored.
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib  # noqa: E402

f32 = C.c_float


def arr3(v):
    return (f32 * 3)(*[float(x) for x in v])


def blob_case_desc(irl, cnt, case):
    """A SceneDesc whose eight arrays have the given element counts and seeded random bytes (serialisation does not
    interpret them), scalars derived from the case number. Returns (desc, keep-alive arrays)."""
    rng = np.random.default_rng(1000 + case)
    sizes = (4, 16, 6, 12, 12, 4, 36, 24)
    arrs = [rng.integers(0, 256, max(c, 1) * sz, dtype=np.uint8) for c, sz in zip(cnt, sizes)]
    d = irl.SceneDesc()
    for (ptr, num), a, c in zip((("geometry", "num_geometry"), ("mesh_info", "num_meshes"), ("mesh_tris", "num_tris"), ("mesh_verts", "num_verts"),
                                  ("mesh_normals", "num_normals"), ("mat_ids", "num_mat_ids"), ("materials", "num_materials"), ("bvh_nodes", "num_nodes")), arrs, cnt):
        setattr(d, ptr, a.ctypes.data); setattr(d, num, c)
    d.max_leaf_depth = 3 + case; d.image_width = 100.5 + case; d.image_height = 64.25 * (case + 1); d.fov_radians = 0.7853981852531433
    d.anti_alias_scale = 0.25 + case / 16; d.max_path_length = 10 + case; d.roulette_start_depth = case % 5; d.samples_per_pixel = 1000 + case
    return d, arrs


def main():
    r = oracle_lib.ref_lib()
    if r is None:
        raise SystemExit("oracle/_ref/libref_l0.so missing: run `make -C oracle ref` where /root/reference exists")
    rng = np.random.default_rng(20260410)
    out = {}

    # sincos: dense sweep over [-4pi, 4pi] plus special points
    xs = np.concatenate([np.linspace(-4 * np.pi, 4 * np.pi, 20001), rng.uniform(-50, 50, 5000),
                         [0.0, np.pi / 8, np.pi / 4, np.pi / 2, np.pi, 2 * np.pi, 1e-8, -1e-8, 1e-3, 100.0, -1000.0, np.pi / 4 / 2]]).astype(np.float32)
    s = np.zeros_like(xs); c = np.zeros_like(xs)
    for i, x in enumerate(xs):
        a, b = f32(), f32()
        r.ref_sincos(float(x), C.byref(a), C.byref(b))
        s[i], c[i] = a.value, b.value
    out.update(sincos_x=xs, sincos_s=s, sincos_c=c)

    # maxi / maxc incl. ties, negatives, zeros
    v = np.concatenate([rng.normal(size=(2000, 3)), rng.integers(-2, 3, size=(500, 3)).astype(np.float64),
                        [[1, 2, 3], [3, 2, 1], [2, 1, 3], [1, 1, 1], [0, 0, 0], [-1, -2, -3], [1, 1, 0], [0, 1, 0]]]).astype(np.float32)
    mi = np.array([r.ref_maxi(float(a), float(b), float(cc)) for a, b, cc in v], dtype=np.uint32)
    mc = np.array([r.ref_maxc(float(a), float(b), float(cc)) for a, b, cc in v], dtype=np.float32)
    out.update(maxi_v=v, maxi_i=mi, maxi_c=mc)

    # xoshiro: seeds -> state, 16 outputs, 16 uniforms, jump
    seeds = np.array([0, 1, 1442, 2**63, 2**64 - 1, 123456789, 0xDEADBEEF], dtype=np.uint64)
    states = np.zeros((len(seeds), 2), dtype=np.uint64)
    nexts = np.zeros((len(seeds), 16), dtype=np.uint64)
    unis = np.zeros((len(seeds), 16), dtype=np.float32)
    jumped = np.zeros((len(seeds), 2), dtype=np.uint64)
    for i, sd in enumerate(seeds):
        st = (C.c_uint64 * 2)()
        r.ref_xoshiro_seed(st, int(sd)); states[i] = (st[0], st[1])
        for k in range(16):
            nexts[i, k] = r.ref_xoshiro_next(st)
        r.ref_xoshiro_seed(st, int(sd))
        for k in range(16):
            unis[i, k] = r.ref_xoshiro_uniform01(st)
        r.ref_xoshiro_seed(st, int(sd)); r.ref_xoshiro_jump(st); jumped[i] = (st[0], st[1])
    sm_in = rng.integers(0, 2**63, size=64, dtype=np.uint64)
    sm_out = np.array([r.ref_splitmix64(int(z)) for z in sm_in], dtype=np.uint64)
    out.update(xo_seeds=seeds, xo_states=states, xo_next=nexts, xo_uniform=unis, xo_jump=jumped, sm_in=sm_in, sm_out=sm_out)

    # sampling / BxDFs
    n = 3000
    u1 = rng.random(n).astype(np.float32); u2 = rng.random(n).astype(np.float32)
    u1[:6] = [0.5, 0.5, 0.0, 1.0, 0.25, 0.75]; u2[:6] = [0.5, 0.25, 0.0, 1.0, 0.75, 0.25]
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True); nrm = nrm.astype(np.float32)
    nrm[:4] = [[0, 1, 0], [1, 0, 0], [0, 0, 1], [0, 0, -1]]
    dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True); dirs = dirs.astype(np.float32)
    disc = np.zeros((n, 2), np.float32); hemi = np.zeros((n, 3), np.float32); diff = np.zeros((n, 3), np.float32)
    refl = np.zeros((n, 3), np.float32); ortho = np.zeros((n, 6), np.float32)
    for i in range(n):
        a, b = f32(), f32()
        r.ref_sample_disc_concentric(float(u1[i]), float(u2[i]), C.byref(a), C.byref(b)); disc[i] = (a.value, b.value)
        o = (f32 * 3)(); r.ref_cosine_sample_hemisphere(float(u1[i]), float(u2[i]), o); hemi[i] = list(o)
        r.ref_sample_diffuse(arr3(nrm[i]), float(u1[i]), float(u2[i]), o); diff[i] = list(o)
        r.ref_reflect(arr3(dirs[i]), arr3(nrm[i]), o); refl[i] = list(o)
        b0, b1 = (f32 * 3)(), (f32 * 3)(); r.ref_orthonormal_system(arr3(nrm[i]), b0, b1); ortho[i] = list(b0) + list(b1)
    out.update(bx_u1=u1, bx_u2=u2, bx_n=nrm, bx_d=dirs, bx_disc=disc, bx_hemi=hemi, bx_diffuse=diff, bx_reflect=refl, bx_ortho=ortho)

    cosT = rng.uniform(-1, 1, n).astype(np.float32); ri = rng.choice([1.52, 1 / 1.52, 1.33, 2.4], n).astype(np.float32)
    sch = np.array([r.ref_schlick(float(a), float(b)) for a, b in zip(cosT, ri)], dtype=np.float32)
    die = np.zeros((n, 3), np.float32); die_flag = np.zeros(n, np.int32); refr = np.zeros((n, 3), np.float32)
    ior = rng.choice([1.52, 1.33, 1.0, 2.4], n).astype(np.float32)
    for i in range(n):
        o = (f32 * 3)()
        die_flag[i] = r.ref_dielectric(arr3([0, 0, 0]), arr3(dirs[i]), arr3(nrm[i]), float(ior[i]), float(u1[i]), o); die[i] = list(o)
        ndotr = float(np.float32(np.dot(nrm[i].astype(np.float64), dirs[i].astype(np.float64))))
        r.ref_refract(arr3(dirs[i]), arr3(nrm[i]), ndotr, float(ri[i]), o); refr[i] = list(o)
    out.update(bx_cos=cosT, bx_ri=ri, bx_schlick=sch, bx_ior=ior, bx_dielectric=die, bx_dielectric_refracted=die_flag, bx_refract=refr)

    tp = rng.random((n, 3)).astype(np.float32); tp[:3] = [[0.9, 0.5, 0.2], [0, 0.5, 0.5], [1, 1, 1]]
    ur = rng.random(n).astype(np.float32); ur[0] = 0.1
    tp_out = np.zeros_like(tp); stop = np.zeros(n, np.int32)
    for i in range(n):
        t = arr3(tp[i]); stop[i] = r.ref_evaluate_roulette(float(ur[i]), t); tp_out[i] = list(t)
    out.update(rr_tp=tp, rr_u=ur, rr_tp_out=tp_out, rr_stop=stop)

    lay = (C.c_uint32 * 12)(); r.ref_layout(lay)
    out["layout"] = np.array(list(lay), dtype=np.uint32)

    # ---- Material, constructors, constants, permute / abs / bounds (everything else that compiles here) ----
    def raw(fn, size, *args):
        """Bytes an object holds after construction over 0x00- and over 0xFF-filled memory; bytes that differ were
        never written by the constructor (padding, or members the constructor leaves uninitialised)."""
        a = (C.c_uint8 * size)(); b = (C.c_uint8 * size)()
        fn(*args, 0x00, a); fn(*args, 0xFF, b)
        a = np.frombuffer(bytes(a), np.uint8); b = np.frombuffer(bytes(b), np.uint8)
        return a.copy(), (a == b)

    ml = (C.c_uint32 * 11)(); r.ref_material_layout(ml)
    out["material_layout"] = np.array(list(ml), dtype=np.uint32)
    msize = int(ml[0])
    md, md_init = raw(r.ref_material_default, msize)
    out.update(material_default=md, material_default_init=md_init)
    mats_in = np.zeros((40, 7), np.float32)
    mats_in[:, :3] = rng.random((40, 3)); mats_in[:, 3:6] = rng.random((40, 3)) * (rng.random((40, 1)) < 0.5); mats_in[:, 6] = rng.integers(0, 3, 40)
    mats_in[0, 3:6] = 0; mats_in[1, 3:6] = [0, 0, 1e-30]; mats_in[2, 3:6] = [-0.0, 0, 0]
    mats_bytes = np.zeros((40, msize), np.uint8); mats_init = np.zeros((40, msize), bool)
    for i, m in enumerate(mats_in):
        mats_bytes[i], mats_init[i] = raw(r.ref_material_make, msize, arr3(m[:3]), arr3(m[3:6]), int(m[6]))
    out.update(material_in=mats_in, material_bytes=mats_bytes, material_init=mats_init)

    od = rng.normal(size=(16, 6)).astype(np.float32); uv = rng.integers(0, 5000, (16, 2)).astype(np.uint32)
    ray_b = np.zeros((16, 32), np.uint8); ray_i = np.zeros((16, 32), bool)
    hit_b = np.zeros((16, 64), np.uint8); hit_i = np.zeros((16, 64), bool)
    tr_b = np.zeros((16, 84), np.uint8); tr_i = np.zeros((16, 84), bool)
    for i in range(16):
        ray_b[i], ray_i[i] = raw(r.ref_ray_ctor, 32, arr3(od[i, :3]), arr3(od[i, 3:]))
        hit_b[i], hit_i[i] = raw(r.ref_hitrecord_ctor, 64, arr3(od[i, :3]), arr3(od[i, 3:]))
        tr_b[i], tr_i[i] = raw(lambda o, d, f, b: r.ref_traceresult_ctor(o, d, int(uv[i, 0]), int(uv[i, 1]), f, b), 84, arr3(od[i, :3]), arr3(od[i, 3:]))
    out.update(ctor_od=od, ctor_uv=uv, ray_bytes=ray_b, ray_init=ray_i, hit_bytes=hit_b, hit_init=hit_i, trace_bytes=tr_b, trace_init=tr_i)
    pc = (f32 * 2)(); r.ref_pixelcoord_default(pc); out["pixelcoord_default"] = np.array(list(pc), np.float32)
    hc = (C.c_uint32 * 4)(); r.ref_hit_constants(hc); out["hit_constants"] = np.array(list(hc), np.uint32)

    pv = rng.normal(size=(8, 3)).astype(np.float32); pv[0] = [-0.0, 2.5, -3.5]
    perm = np.zeros((8, 27, 3), np.float32); ab = np.zeros((8, 3), np.float32); nz = np.zeros(8, np.int32)
    for i in range(8):
        for k in range(27):
            o = (f32 * 3)(); r.ref_permute(arr3(pv[i]), k // 9, (k // 3) % 3, k % 3, o); perm[i, k] = list(o)
        o = (f32 * 3)(); r.ref_abs(arr3(pv[i]), o); ab[i] = list(o)
    nzv = np.array([[0, 0, 0], [-0.0, 0, 0], [0, 1e-45, 0], [0, 0, 1], [np.nan, 0, 0]], np.float32)
    nz = np.array([r.ref_is_non_zero(arr3(v)) for v in nzv], np.int32)
    out.update(permute_v=pv, permute_out=perm, abs_out=ab, nonzero_v=nzv, nonzero_out=nz)
    bd = (f32 * 6)(); r.ref_bounds_default(bd); out["bounds_default"] = np.array(list(bd), np.float32)
    bl = rng.normal(size=(32, 2, 3)).astype(np.float32); bh = bl + rng.random((32, 2, 3)).astype(np.float32) * 3
    bu = np.zeros((32, 9), np.float32)
    for i in range(32):
        o = (f32 * 9)()
        r.ref_bounds_union((f32 * 6)(*bl[i, 0], *bh[i, 0]), (f32 * 6)(*bl[i, 1], *bh[i, 1]), o); bu[i] = list(o)
    out.update(bounds_lo=bl, bounds_hi=bh, bounds_union=bu)

    # ---- the serialised scene as the reference's own Deserialiser<16> reads it (deserialisation.hpp:31-59) ----
    # Small synthetic scenes whose array counts put every array on every residue of its alignment; the bytes are
    # written by the PRODUCT's writer (mi_scene_serialise) and walked by the reference's reader: offsets, counts,
    # scalars and bytes consumed are the golden values; the blobs themselves are stored so that the test can hand the
    # very same bytes to the product's reader (and check that the product's writer still produces them).
    import ipu_ray_lib_amd as irl
    pad = (C.c_uint32 * 256)(); r.ref_padding_table(pad)
    out["blob_padding_table"] = np.array(list(pad), np.uint32).reshape(4, 64)
    blobs, walks, scal, used, counts = [], [], [], [], []
    brng = np.random.default_rng(77)
    for case in range(24):
        cnt = [int(x) for x in brng.integers(0, 7, 8)]
        if case == 0: cnt = [0] * 8
        if case == 1: cnt = [1] * 8
        d, keep = blob_case_desc(irl, cnt, case)
        blob = irl.serialise_scene(d)
        w, sc, u = oracle_lib.ref_walk_scene_blob(blob)
        assert u == blob.size, (case, u, blob.size)
        blobs.append(blob.copy()); walks.append(w); scal.append(sc); used.append(u); counts.append(cnt)
    out["blob_case_counts"] = np.array(counts, np.uint32)
    out["blob_case_sizes"] = np.array([b.size for b in blobs], np.uint32)
    out["blob_case_bytes"] = np.concatenate(blobs)
    out["blob_case_walk"] = np.array(walks, np.uint64)
    out["blob_case_scalars"] = np.array(scal, np.uint32)
    # truncated streams: the reader must report the end of the byte stream for every proper prefix of case 5
    b5 = blobs[5]
    trunc = []
    for cut in range(0, b5.size):
        part = irl.aligned_bytes(max(cut, 1))[:cut]; part[:] = b5[:cut]
        trunc.append(oracle_lib.ref_walk_scene_blob(part)[2] if cut else oracle_lib.ref_walk_scene_blob(irl.aligned_bytes(16)[:0])[2])
    out["blob_truncation_result"] = np.array(trunc, np.int64)

    dst = Path(__file__).with_name("ref_l0_vectors.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, dst.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
