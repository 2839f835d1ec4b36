"""ctypes doorway to oracle/libray_oracle.so — the CHECKER. Imported by tests, smoke() and
bench.py's cpu_baseline leg only; nothing in ipu_ray_lib_amd imports this."""
import ctypes as C
from pathlib import Path

import numpy as np

from ipu_ray_lib_amd import SceneDesc, TRACE_RESULT

ROOT = Path(__file__).resolve().parent.parent
ORACLE_SO = ROOT / "oracle" / "libray_oracle.so"
REF_SO = ROOT / "oracle" / "_ref" / "libref_l0.so"

f32, u32, u64 = C.c_float, C.c_uint32, C.c_uint64
pf = C.POINTER(C.c_float)


class Vec3(C.Structure):
    _fields_ = [("x", f32), ("y", f32), ("z", f32)]

    def t(self):
        return (self.x, self.y, self.z)


class Ray(C.Structure):
    _fields_ = [("origin", Vec3), ("tMin", f32), ("direction", Vec3), ("tMax", f32)]


class Shear(C.Structure):
    _fields_ = [("o", Vec3), ("dir", Vec3), ("ix", u32), ("iy", u32), ("iz", u32), ("sx", f32), ("sy", f32), ("sz", f32)]


class Sphere(C.Structure):
    _fields_ = [("x", f32), ("y", f32), ("z", f32), ("radius", f32)]


class Disc(C.Structure):
    _fields_ = [("nx", f32), ("ny", f32), ("nz", f32), ("r", f32), ("cx", f32), ("cy", f32), ("cz", f32)]


class Node(C.Structure):
    _fields_ = [("min_x", f32), ("min_y", f32), ("min_z", f32), ("link", u32),
                ("dx", C.c_uint16), ("dy", C.c_uint16), ("dz", C.c_uint16), ("geomID", C.c_uint16)]


class Stats(C.Structure):
    _fields_ = [("casts", u64), ("nodesVisited", u64), ("leafTests", u64), ("paths", u64)]

    def as_dict(self):
        return {"casts": self.casts, "nodes_visited": self.nodesVisited, "leaf_tests": self.leafTests, "paths": self.paths}


class Intersection(C.Structure):
    _fields_ = [("hit", C.c_int), ("geomID", u32), ("primID", u32), ("t", f32), ("normal", Vec3)]


class Nif(C.Structure):
    _fields_ = [("numLayers", u32), ("kernels", C.POINTER(C.c_void_p)), ("biases", C.POINTER(C.c_void_p)),
                ("rows", C.POINTER(u32)), ("cols", C.POINTER(u32)), ("relu", C.POINTER(C.c_uint8)),
                ("embeddingDimension", u32), ("maxValue", f32), ("mean", f32 * 3), ("logTonemap", C.c_int32),
                ("halfFeatures", C.c_int32), ("halfWeightsActs", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not ORACLE_SO.exists():
        raise RuntimeError("oracle/libray_oracle.so is not built: make -C oracle")
    o = C.CDLL(str(ORACLE_SO))
    o.o_version.restype = C.c_char_p
    o.o_half_to_float.argtypes = [C.c_uint16]; o.o_half_to_float.restype = f32
    o.o_float_to_half_rne.argtypes = [f32]; o.o_float_to_half_rne.restype = C.c_uint16
    o.o_round_to_half_not_smaller.argtypes = [f32]; o.o_round_to_half_not_smaller.restype = C.c_uint16
    o.o_gamma.argtypes = [C.c_int]; o.o_gamma.restype = f32
    o.o_ray_epsilon.restype = f32
    o.o_maxi.argtypes = [Vec3]; o.o_maxi.restype = u32
    o.o_maxc.argtypes = [Vec3]; o.o_maxc.restype = f32
    o.o_sincos.argtypes = [f32, pf, pf]
    o.o_orthonormal_system.argtypes = [Vec3, C.POINTER(Vec3), C.POINTER(Vec3)]
    o.o_splitmix64.argtypes = [u64]; o.o_splitmix64.restype = u64
    o.o_xoshiro_seed.argtypes = [C.POINTER(u64), u64]
    o.o_xoshiro_next.argtypes = [C.POINTER(u64)]; o.o_xoshiro_next.restype = u64
    o.o_xoshiro_jump.argtypes = [C.POINTER(u64)]
    o.o_xoshiro_uniform01.argtypes = [C.POINTER(u64)]; o.o_xoshiro_uniform01.restype = f32
    o.o_slab.argtypes = [f32, f32, f32, f32, pf, pf]; o.o_slab.restype = C.c_int
    o.o_node_intersect.argtypes = [C.POINTER(Node), Vec3, Vec3, pf, pf]; o.o_node_intersect.restype = C.c_int
    o.o_ray_shear.argtypes = [C.POINTER(Ray), C.POINTER(Shear)]
    o.o_intersect_triangle.argtypes = [Vec3, Vec3, Vec3, C.POINTER(Shear), f32, pf]; o.o_intersect_triangle.restype = f32
    o.o_set_double_fallback.argtypes = [C.c_int]; o.o_set_double_fallback.restype = None
    o.o_sphere_intersect.argtypes = [C.POINTER(Sphere), C.POINTER(Ray)]; o.o_sphere_intersect.restype = f32
    o.o_disc_intersect.argtypes = [C.POINTER(Disc), C.POINTER(Ray)]; o.o_disc_intersect.restype = f32
    o.o_offset_ray.argtypes = [C.POINTER(Ray), Vec3]
    o.o_pixel_to_ray_dir.argtypes = [f32, f32, f32, f32, f32]; o.o_pixel_to_ray_dir.restype = Vec3
    o.o_sample_disc_concentric.argtypes = [f32, f32, pf, pf]
    o.o_cosine_sample_hemisphere.argtypes = [f32, f32]; o.o_cosine_sample_hemisphere.restype = Vec3
    o.o_sample_diffuse.argtypes = [Vec3, f32, f32]; o.o_sample_diffuse.restype = Vec3
    o.o_reflect.argtypes = [Vec3, Vec3]; o.o_reflect.restype = Vec3
    o.o_schlick.argtypes = [f32, f32]; o.o_schlick.restype = f32
    o.o_refract.argtypes = [Vec3, Vec3, f32, f32]; o.o_refract.restype = Vec3
    o.o_dielectric.argtypes = [C.POINTER(Ray), Vec3, f32, f32, C.POINTER(Vec3)]; o.o_dielectric.restype = C.c_int
    o.o_evaluate_roulette.argtypes = [f32, C.POINTER(Vec3)]; o.o_evaluate_roulette.restype = C.c_int
    o.o_logf_det.argtypes = [f32]; o.o_logf_det.restype = f32
    o.o_gauss2.argtypes = [C.POINTER(u64), pf, pf]
    o.o_bvh_intersect.argtypes = [C.POINTER(SceneDesc), C.POINTER(Ray), C.POINTER(Stats)]; o.o_bvh_intersect.restype = Intersection
    o.o_bvh_occluded.argtypes = [C.POINTER(SceneDesc), C.POINTER(Ray), C.POINTER(Stats)]; o.o_bvh_occluded.restype = C.c_int
    o.o_init_ray_stream.argtypes = [C.POINTER(SceneDesc), C.c_void_p]
    o.o_shadow_trace.argtypes = [C.POINTER(SceneDesc), C.c_void_p, C.c_size_t, C.c_int, C.POINTER(Stats)]
    o.o_path_trace_pixel_rng.argtypes = [C.POINTER(SceneDesc), C.c_void_p, C.c_size_t, C.c_int, C.POINTER(Stats)]
    o.o_path_trace_shared_rng.argtypes = [C.POINTER(SceneDesc), C.c_void_p, C.c_size_t, C.POINTER(Stats)]
    o.o_escaped_uv.argtypes = [C.c_void_p, C.c_size_t, f32, C.c_void_p, C.c_void_p]
    o.o_nif_infer.argtypes = [C.POINTER(Nif), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    o.o_apply_env.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    o.o_path_trace_nif_pixel_rng.argtypes = [C.POINTER(SceneDesc), C.POINTER(Nif), f32, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(Stats)]
    _lib = o
    return o


def ref_lib():
    """The reference's own Eigen-free L0 code (only present where /root/reference was)."""
    if not REF_SO.exists():
        return None
    r = C.CDLL(str(REF_SO))
    r.ref_sincos.argtypes = [f32, pf, pf]
    r.ref_maxi.argtypes = [f32, f32, f32]; r.ref_maxi.restype = u32
    r.ref_maxc.argtypes = [f32, f32, f32]; r.ref_maxc.restype = f32
    r.ref_normalized.argtypes = [pf, pf]
    r.ref_dot.argtypes = [pf, pf]; r.ref_dot.restype = f32
    r.ref_cross.argtypes = [pf, pf, pf]
    r.ref_orthonormal_system.argtypes = [pf, pf, pf]
    r.ref_splitmix64.argtypes = [u64]; r.ref_splitmix64.restype = u64
    r.ref_xoshiro_seed.argtypes = [C.POINTER(u64), u64]
    r.ref_xoshiro_next.argtypes = [C.POINTER(u64)]; r.ref_xoshiro_next.restype = u64
    r.ref_xoshiro_jump.argtypes = [C.POINTER(u64)]
    r.ref_xoshiro_uniform01.argtypes = [C.POINTER(u64)]; r.ref_xoshiro_uniform01.restype = f32
    r.ref_sample_disc_concentric.argtypes = [f32, f32, pf, pf]
    r.ref_cosine_sample_hemisphere.argtypes = [f32, f32, pf]
    r.ref_sample_diffuse.argtypes = [pf, f32, f32, pf]
    r.ref_reflect.argtypes = [pf, pf, pf]
    r.ref_schlick.argtypes = [f32, f32]; r.ref_schlick.restype = f32
    r.ref_refract.argtypes = [pf, pf, f32, f32, pf]
    r.ref_dielectric.argtypes = [pf, pf, pf, f32, f32, pf]; r.ref_dielectric.restype = C.c_int
    r.ref_evaluate_roulette.argtypes = [f32, pf]; r.ref_evaluate_roulette.restype = C.c_int
    r.ref_layout.argtypes = [C.POINTER(u32)]
    pb = C.POINTER(C.c_uint8)
    r.ref_material_layout.argtypes = [C.POINTER(u32)]
    r.ref_material_default.argtypes = [C.c_uint8, pb]
    r.ref_material_make.argtypes = [pf, pf, u32, C.c_uint8, pb]
    r.ref_ray_ctor.argtypes = [pf, pf, C.c_uint8, pb]
    r.ref_hitrecord_ctor.argtypes = [pf, pf, C.c_uint8, pb]
    r.ref_traceresult_ctor.argtypes = [pf, pf, u32, u32, C.c_uint8, pb]
    r.ref_pixelcoord_default.argtypes = [pf]
    r.ref_hit_constants.argtypes = [C.POINTER(u32)]
    r.ref_permute.argtypes = [pf, u32, u32, u32, pf]
    r.ref_abs.argtypes = [pf, pf]
    r.ref_is_non_zero.argtypes = [pf]; r.ref_is_non_zero.restype = C.c_int
    r.ref_bounds_default.argtypes = [pf]
    r.ref_bounds_union.argtypes = [pf, pf, pf]
    if hasattr(r, "ref_walk_scene_blob"):
        r.ref_walk_scene_blob.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        r.ref_walk_scene_blob.restype = C.c_long
        r.ref_padding_table.argtypes = [C.POINTER(C.c_uint32)]
    return r


def shadow_trace(desc: SceneDesc, rays: np.ndarray, threads: int = 8):
    assert rays.dtype == TRACE_RESULT
    st = Stats()
    lib().o_shadow_trace(C.byref(desc), rays.ctypes.data, rays.size, threads, C.byref(st))
    return st


def path_trace_pixel_rng(desc: SceneDesc, rays: np.ndarray, threads: int = 8):
    assert rays.dtype == TRACE_RESULT
    st = Stats()
    lib().o_path_trace_pixel_rng(C.byref(desc), rays.ctypes.data, rays.size, threads, C.byref(st))
    return st


def path_trace_shared_rng(desc: SceneDesc, rays: np.ndarray):
    assert rays.dtype == TRACE_RESULT
    st = Stats()
    lib().o_path_trace_shared_rng(C.byref(desc), rays.ctypes.data, rays.size, C.byref(st))
    return st


def make_nif(kernels, biases, relu, embedding_dimension, max_value, mean, log_tonemap=True,
             half_features=True, half_weights_acts=False):
    """Returns (Nif struct, keepalive list)."""
    n = len(kernels)
    ks = [np.ascontiguousarray(k, dtype=np.float32) for k in kernels]
    bs = [None if b is None else np.ascontiguousarray(b, dtype=np.float32) for b in biases]
    kp = (C.c_void_p * n)(*[k.ctypes.data for k in ks])
    bp = (C.c_void_p * n)(*[(b.ctypes.data if b is not None else None) for b in bs])
    rows = (u32 * n)(*[k.shape[0] for k in ks])
    cols = (u32 * n)(*[k.shape[1] for k in ks])
    rl = (C.c_uint8 * n)(*[1 if r else 0 for r in relu])
    nif = Nif()
    nif.numLayers = n
    nif.kernels = C.cast(kp, C.POINTER(C.c_void_p)); nif.biases = C.cast(bp, C.POINTER(C.c_void_p))
    nif.rows = C.cast(rows, C.POINTER(u32)); nif.cols = C.cast(cols, C.POINTER(u32)); nif.relu = C.cast(rl, C.POINTER(C.c_uint8))
    nif.embeddingDimension = embedding_dimension
    nif.maxValue = max_value
    nif.mean = (f32 * 3)(*[float(m) for m in mean])
    nif.logTonemap = 1 if log_tonemap else 0
    nif.halfFeatures = 1 if half_features else 0
    nif.halfWeightsActs = 1 if half_weights_acts else 0
    return nif, [ks, bs, kp, bp, rows, cols, rl]


def ref_walk_scene_blob(blob):
    """The reference's Deserialiser<16> (include/serialisation/Deserialiser.hpp, compiled from the checkout into
    oracle/_ref) walking a serialised scene: ((offset, count) x 8 arrays, the eight scalars as raw u32, bytes consumed),
    consumed = -1 when it ran off the end. `blob`: a 16-byte-aligned uint8 numpy array."""
    r = ref_lib()
    assert blob.ctypes.data % 16 == 0
    out = (C.c_uint64 * 16)(); sc = (C.c_uint32 * 8)()
    used = r.ref_walk_scene_blob(blob.ctypes.data, blob.size, out, sc)
    return [int(x) for x in out], [int(x) for x in sc], int(used)


class double_fallback:
    """with oracle_lib.double_fallback(): ... - the oracle as the reference's ALLOW_DOUBLE_FALLBACK=1 build (Mesh.cpp:38-51)."""
    def __enter__(self):
        lib().o_set_double_fallback(1)
    def __exit__(self, *a):
        lib().o_set_double_fallback(0)
