"""ipu_ray_lib_amd — MI355X-native ray/path-trace hot path behind the reference's IpuScene surface.

Python here is plumbing only: ctypes bindings over the two C-ABI shared libraries

* ``libmi_scene_host.so``  (include/mi_scene_host.h) — CPU-side scene construction, BVH build,
  ray-stream initialisation: the callers' side of the hot path;
* ``libmi_raylib.so``      (include/mi_raylib.h)     — the gfx950 HIP kernels (shadow trace,
  path trace, NIF MLP) and nothing else.

There is NO CPU fallback for the device library: :func:`device_lib` raises if it is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
REPO_ROOT = PKG_DIR.parent
DEFAULT_MESH = REPO_ROOT / "assets" / "monkey_bust.glb"

# --------------------------------------------------------------------------------------------
# POD layouts (== include/mi_raylib.h == the reference's structs)
# --------------------------------------------------------------------------------------------
VEC3 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
RAY = np.dtype([("origin", VEC3), ("tMin", "<f4"), ("direction", VEC3), ("tMax", "<f4")])
HIT = np.dtype([("r", RAY), ("primID", "<u4"), ("normal", VEC3), ("throughput", VEC3),
                ("geomID", "<u2"), ("flags", "<u2")])
TRACE_RESULT = np.dtype([("rgb", VEC3), ("u", "<f4"), ("v", "<f4"), ("h", HIT)])
BVH_NODE = np.dtype([("min_x", "<f4"), ("min_y", "<f4"), ("min_z", "<f4"), ("link", "<u4"),
                     ("dx", "<u2"), ("dy", "<u2"), ("dz", "<u2"), ("geomID", "<u2")])
MATERIAL = np.dtype([("albedo", VEC3), ("ior", "<f4"), ("emission", VEC3), ("type", "<i4"),
                     ("emissive", "u1"), ("pad", "u1", (3,))])
MESH_INFO = np.dtype([("firstIndex", "<u4"), ("firstVertex", "<u4"), ("numTriangles", "<u4"), ("numVertices", "<u4")])
GEOM_REF = np.dtype([("index", "<u2"), ("type", "u1"), ("pad", "u1")])
SPHERE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("radius", "<f4")])
DISC = np.dtype([("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"), ("r", "<f4"), ("cx", "<f4"), ("cy", "<f4"), ("cz", "<f4")])

assert TRACE_RESULT.itemsize == 84 and HIT.itemsize == 64 and RAY.itemsize == 32
assert BVH_NODE.itemsize == 24 and MATERIAL.itemsize == 36 and MESH_INFO.itemsize == 16 and GEOM_REF.itemsize == 4

FLAG_ERROR, FLAG_ESCAPED = 1, 2
INVALID_GEOM, INVALID_PRIM = 0xFFFF, 0xFFFFFFFF
MODE_SHADOW_TRACE, MODE_PATH_TRACE = 0, 1

MI_OK = 0


class SceneDesc(C.Structure):
    """mi_scene_desc (include/mi_raylib.h); the CPU oracle's ``oscene`` has the same layout."""
    _fields_ = [
        ("geometry", C.c_void_p), ("num_geometry", C.c_uint32),
        ("mesh_info", C.c_void_p), ("num_meshes", C.c_uint32),
        ("mesh_tris", C.c_void_p), ("num_tris", C.c_uint32),
        ("mesh_verts", C.c_void_p), ("num_verts", C.c_uint32),
        ("mesh_normals", C.c_void_p), ("num_normals", C.c_uint32),
        ("mat_ids", C.c_void_p), ("num_mat_ids", C.c_uint32),
        ("materials", C.c_void_p), ("num_materials", C.c_uint32),
        ("bvh_nodes", C.c_void_p), ("num_nodes", C.c_uint32),
        ("max_leaf_depth", C.c_uint32),
        ("spheres", C.c_void_p), ("num_spheres", C.c_uint32),
        ("discs", C.c_void_p), ("num_discs", C.c_uint32),
        ("image_width", C.c_float), ("image_height", C.c_float),
        ("fov_radians", C.c_float), ("anti_alias_scale", C.c_float),
        ("max_path_length", C.c_uint32), ("roulette_start_depth", C.c_uint32),
        ("samples_per_pixel", C.c_uint32),
        ("rng_seed", C.c_uint64),
        ("window_w", C.c_int32), ("window_h", C.c_int32), ("window_c", C.c_int32), ("window_r", C.c_int32),
        ("path_trace", C.c_int32),
        ("device", C.c_int32),
    ]

    def set_image(self, width: int, height: int, crop=None):
        """--width/--height/--crop of the reference CLI (trace.cpp:475-479)."""
        self.image_width, self.image_height = float(width), float(height)
        if crop is None:
            crop = (width, height, 0, 0)
        self.window_w, self.window_h, self.window_c, self.window_r = crop
        return self

    @property
    def num_rays(self) -> int:
        return int(self.window_w) * int(self.window_h)


class NifDesc(C.Structure):
    """mi_nif_desc (include/mi_scene_host.h)."""
    _fields_ = [("num_layers", C.c_uint32), ("kernels", C.POINTER(C.POINTER(C.c_float))),
                ("biases", C.POINTER(C.POINTER(C.c_float))), ("rows", C.POINTER(C.c_uint32)),
                ("cols", C.POINTER(C.c_uint32)), ("relu", C.POINTER(C.c_uint8)),
                ("embedding_dimension", C.c_uint32), ("hidden_size", C.c_uint32), ("max_value", C.c_float),
                ("mean", C.c_float * 3), ("log_tonemap", C.c_int32), ("weights_are_half", C.c_int32),
                ("name", C.c_char_p), ("source", C.c_char_p)]


class RaylibError(RuntimeError):
    pass


_host = None
_device = {}


def _load(path: Path) -> C.CDLL:
    if not path.exists():
        raise RaylibError(f"{path.name} is not built (run `python -c 'import __graft_entry__ as g; g.build()'` at the repo root)")
    return C.CDLL(str(path))


def host_lib() -> C.CDLL:
    """libmi_scene_host.so — CPU-only scene plumbing."""
    global _host
    if _host is None:
        # (MI_SCENE_HOST_LIB: another build of the same library, e.g. the AddressSanitizer / UBSan build of tools/sanitize_host.sh)
        lib = _load(Path(os.environ["MI_SCENE_HOST_LIB"]) if os.environ.get("MI_SCENE_HOST_LIB") else PKG_DIR / "libmi_scene_host.so")
        lib.mi_host_last_error.restype = C.c_char_p
        lib.mi_host_scene_builtin.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        lib.mi_host_scene_import.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        lib.mi_host_scene_from_arrays.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        lib.mi_host_scene_fill_desc.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        lib.mi_host_scene_destroy.argtypes = [C.c_void_p]
        lib.mi_host_scene_destroy.restype = None
        lib.mi_build_compact_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                             C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.mi_init_ray_stream.argtypes = [C.POINTER(SceneDesc), C.c_void_p, C.c_size_t]
        lib.mi_scale_rgb.argtypes = [C.c_void_p, C.c_size_t, C.c_float]
        lib.mi_scale_rgb.restype = None
        lib.mi_scene_blob_size.argtypes = [C.POINTER(SceneDesc)]
        lib.mi_scene_blob_size.restype = C.c_size_t
        lib.mi_scene_serialise.argtypes = [C.POINTER(SceneDesc), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        lib.mi_scene_deserialise.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(SceneDesc), C.POINTER(C.c_size_t)]
        lib.mi_blob_padding.argtypes = [C.c_uint32, C.c_size_t, C.c_uint32]
        lib.mi_blob_padding.restype = C.c_uint32
        lib.mi_shard_band_rays.argtypes = [C.c_size_t, C.c_uint32]
        lib.mi_shard_band_rays.restype = C.c_size_t
        lib.mi_shard_count.argtypes = [C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32]
        lib.mi_shard_count.restype = C.c_size_t
        lib.mi_shard_stream_index.argtypes = [C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
        lib.mi_shard_frame_index.argtypes = [C.c_size_t, C.c_size_t, C.c_uint32, C.c_void_p]
        lib.mi_host_nif_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.mi_host_nif_describe.argtypes = [C.c_void_p, C.POINTER(NifDesc)]
        lib.mi_host_nif_destroy.argtypes = [C.c_void_p]
        lib.mi_host_nif_destroy.restype = None
        lib.mi_host_nif_last_error.restype = C.c_char_p
        _host = lib
    return _host


def device_lib(variants: bool = False) -> C.CDLL:
    """libmi_raylib.so — the HIP kernels. Raises (never falls back) when it is not built.
    variants=True: libmi_raylib_variants.so, the test build of the same sources with -DMI_RAYLIB_VARIANTS=1, which also
    carries the kernel families that were measured and not made the default (options kernel 2 / 3, spec, waves, tune, pool_*)."""
    global _device
    if variants not in _device:
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 (same SONAME as the
        # system one libmi_raylib.so links to). Importing torch FIRST makes both resolve to the same
        # already-loaded runtime; the other order leaves torch unable to see the GPU afterwards.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        # (MI_RAYLIB_LIB: another build of the same library, for A/B timing of two builds on one box)
        if variants:
            # (MI_RAYLIB_VARIANTS_LIB: another build of the variants library, e.g. the timing-only builds of tools/k3r_knockouts.sh)
            lib = _load(Path(os.environ["MI_RAYLIB_VARIANTS_LIB"]) if os.environ.get("MI_RAYLIB_VARIANTS_LIB") else PKG_DIR / "libmi_raylib_variants.so")
        else:
            lib = _load(Path(os.environ["MI_RAYLIB_LIB"]) if os.environ.get("MI_RAYLIB_LIB") else PKG_DIR / "libmi_raylib.so")
        lib.mi_last_error.restype = C.c_char_p
        lib.mi_version.restype = C.c_char_p
        lib.mi_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        lib.mi_scene_create_from_blob.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        lib.mi_scene_destroy.argtypes = [C.c_void_p]
        lib.mi_scene_destroy.restype = None
        lib.mi_render.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.mi_render_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.mi_trace_time_secs.argtypes = [C.c_void_p]
        lib.mi_trace_time_secs.restype = C.c_double
        lib.mi_get_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        lib.mi_reset_counters.argtypes = [C.c_void_p]
        lib.mi_get_phase_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        lib.mi_scene_set_nif.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_int32]
        lib.mi_scene_set_hdri_rotation.argtypes = [C.c_void_p, C.c_float]
        lib.mi_scene_set_max_nif_batch.argtypes = [C.c_void_p, C.c_size_t]
        lib.mi_scene_set_ray_batch.argtypes = [C.c_void_p, C.c_size_t]
        lib.mi_nif_infer_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.mi_group_create.argtypes = [C.POINTER(SceneDesc), C.c_void_p, C.c_uint32, C.c_int32, C.POINTER(C.c_void_p)]
        lib.mi_group_destroy.argtypes = [C.c_void_p]
        lib.mi_group_destroy.restype = None
        lib.mi_group_size.argtypes = [C.c_void_p]
        lib.mi_group_size.restype = C.c_uint32
        lib.mi_group_scene.argtypes = [C.c_void_p, C.c_uint32]
        lib.mi_group_scene.restype = C.c_void_p
        lib.mi_group_set_ray_batch.argtypes = [C.c_void_p, C.c_size_t]
        lib.mi_group_render.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.mi_group_trace_time_secs.argtypes = [C.c_void_p]
        lib.mi_group_trace_time_secs.restype = C.c_double
        lib.mi_group_get_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        lib.mi_group_last_transfer.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        lib.mi_group_last_gather_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        lib.mi_group_devices.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.mi_group_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.mi_group_trace.argtypes = [C.c_void_p, C.c_int]
        lib.mi_group_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.mi_group_gathered_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint32]
        lib.mi_group_reset_counters.argtypes = [C.c_void_p]
        lib.mi_get_pool_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        lib.mi_debug_launch_progress.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.mi_scene_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        lib.mi_get_nif_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        lib.mi_get_nif_clock.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        _device[variants] = lib
    return _device[variants]


def _check_host(status: int):
    if status != MI_OK:
        raise RaylibError(f"host scene call failed ({status}): {host_lib().mi_host_last_error().decode()}")


def _check_dev(status: int):
    if status != MI_OK:
        raise RaylibError(f"mi_raylib call failed ({status}): {device_lib().mi_last_error().decode()}")


class HostScene:
    """Scene arrays + CompactBVH built on the host (buildSceneDescription + buildSceneData,
    reference src/app_utils.cpp:252-371). ``desc`` is the SceneRef analogue handed to renderers."""

    def __init__(self, handle: C.c_void_p):
        self._h = handle
        self.desc = SceneDesc()
        _check_host(host_lib().mi_host_scene_fill_desc(self._h, C.byref(self.desc)))

    @classmethod
    def builtin(cls, name: str = "box", mesh_file: os.PathLike | str | None = None) -> "HostScene":
        mesh = str(mesh_file if mesh_file is not None else DEFAULT_MESH)
        h = C.c_void_p()
        _check_host(host_lib().mi_host_scene_builtin(name.encode(), mesh.encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def import_file(cls, path, load_normals: bool = False) -> "HostScene":
        """importScene(): --mesh-file / --load-normals of the reference CLI."""
        h = C.c_void_p()
        _check_host(host_lib().mi_host_scene_import(str(path).encode(), 1 if load_normals else 0, C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, geometry_desc: SceneDesc) -> "HostScene":
        h = C.c_void_p()
        _check_host(host_lib().mi_host_scene_from_arrays(C.byref(geometry_desc), C.byref(h)))
        return cls(h)

    def _view(self, ptr, count, dtype):
        if not ptr or not count:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (count * dtype.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=count)

    @property
    def nodes(self):
        return self._view(self.desc.bvh_nodes, self.desc.num_nodes, BVH_NODE)

    @property
    def verts(self):
        return self._view(self.desc.mesh_verts, self.desc.num_verts, VEC3)

    @property
    def tris(self):
        return self._view(self.desc.mesh_tris, self.desc.num_tris * 3, np.dtype("<u2")).reshape(-1, 3)

    @property
    def mesh_info(self):
        return self._view(self.desc.mesh_info, self.desc.num_meshes, MESH_INFO)

    @property
    def geometry(self):
        return self._view(self.desc.geometry, self.desc.num_geometry, GEOM_REF)

    @property
    def materials(self):
        return self._view(self.desc.materials, self.desc.num_materials, MATERIAL)

    @property
    def mat_ids(self):
        return self._view(self.desc.mat_ids, self.desc.num_mat_ids, np.dtype("<u4"))

    @property
    def spheres(self):
        return self._view(self.desc.spheres, self.desc.num_spheres, SPHERE)

    @property
    def discs(self):
        return self._view(self.desc.discs, self.desc.num_discs, DISC)

    def init_ray_stream(self) -> np.ndarray:
        """initPerspectiveRayStream (no jitter) + zeroRgb for the desc's window."""
        rays = np.zeros(self.desc.num_rays, dtype=TRACE_RESULT)
        _check_host(host_lib().mi_init_ray_stream(C.byref(self.desc), rays.ctypes.data, rays.size))
        return rays

    def close(self):
        if self._h:
            host_lib().mi_host_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def aligned_bytes(n: int, align: int = 16) -> np.ndarray:
    """A writable uint8 array of n bytes whose first byte is `align`-aligned."""
    raw = np.zeros(n + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n]


def serialise_scene(desc: SceneDesc) -> np.ndarray:
    """Serialiser<16> << SceneRef (src/IpuScene.cpp:51-53): the scene as one 16-byte-aligned byte stream."""
    lib = host_lib()
    n = lib.mi_scene_blob_size(C.byref(desc))
    out = aligned_bytes(n)
    written = C.c_size_t()
    _check_host(lib.mi_scene_serialise(C.byref(desc), out.ctypes.data, n, C.byref(written)))
    assert written.value == n
    return out


def deserialise_scene(blob: np.ndarray, into: SceneDesc | None = None) -> SceneDesc:
    """Deserialiser<16> >> SceneRef: array views INTO `blob` (keep it alive) + the eight scalars."""
    d = into if into is not None else SceneDesc()
    used = C.c_size_t()
    _check_host(host_lib().mi_scene_deserialise(blob.ctypes.data, blob.size, C.byref(d), C.byref(used)))
    d._blob = blob        # keep-alive
    d._blob_bytes_used = used.value
    return d


class NifAssets:
    """NIF model read from an 'assets.extra' directory: nif_metadata.txt + converted.hdf5 (Keras H5) or
    nif_weights.bin — what IpuScene::loadNifModel reads (src/IpuScene.cpp:174-187). Arrays are copies."""

    def __init__(self, asset_path):
        lib = host_lib()
        h = C.c_void_p()
        if lib.mi_host_nif_load(os.fspath(asset_path).encode(), C.byref(h)) != 0:
            raise RaylibError(lib.mi_host_nif_last_error().decode())
        try:
            d = NifDesc()
            if lib.mi_host_nif_describe(h, C.byref(d)) != 0:
                raise RaylibError(lib.mi_host_nif_last_error().decode())
            self.kernels, self.biases, self.relu = [], [], []
            for i in range(d.num_layers):
                r, c = int(d.rows[i]), int(d.cols[i])
                self.kernels.append(np.ctypeslib.as_array(d.kernels[i], shape=(r, c)).copy())
                self.biases.append(np.ctypeslib.as_array(d.biases[i], shape=(c,)).copy() if d.biases[i] else None)
                self.relu.append(bool(d.relu[i]))
            self.embedding_dimension = int(d.embedding_dimension)
            self.hidden_size = int(d.hidden_size)
            self.max_value = float(d.max_value)
            self.mean = np.array(list(d.mean), dtype=np.float32)
            self.log_tonemap = bool(d.log_tonemap)
            self.weights_are_half = bool(d.weights_are_half)
            self.name = d.name.decode()
            self.source = d.source.decode()
        finally:
            lib.mi_host_nif_destroy(h)


class IpuScene:
    """Mirror of the reference's ``IpuScene`` driver object (include/IpuScene.hpp:22-56) over the
    C ABI: construct from a scene description, optionally load a NIF, ``run`` a ray stream."""

    def __init__(self, desc: SceneDesc, variants: bool = False):
        self._lib = device_lib(variants)
        self._h = C.c_void_p()
        self.desc = desc
        self._check(self._lib.mi_scene_create(C.byref(desc), C.byref(self._h)))

    def _check(self, status: int):
        if status != MI_OK:
            raise RaylibError(f"mi_raylib call failed ({status}): {self._lib.mi_last_error().decode()}")

    # -- reference API names -------------------------------------------------------------
    @classmethod
    def from_blob(cls, blob: np.ndarray, extras: SceneDesc, variants: bool = False) -> "IpuScene":
        """Scene from the reference's serialised byte stream + the fields that are not part of it."""
        self = cls.__new__(cls)
        self._lib = device_lib(variants)
        self._h = C.c_void_p()
        self.desc = extras
        b = np.ascontiguousarray(blob, dtype=np.uint8)
        self._check(self._lib.mi_scene_create_from_blob(b.ctypes.data, b.size, C.byref(extras), C.byref(self._h)))
        return self

    def setHdriRotation(self, degrees: float):
        self._check(self._lib.mi_scene_set_hdri_rotation(self._h, float(degrees)))

    def setMaxNifBatchSize(self, rays_per_batch: int):
        self._check(self._lib.mi_scene_set_max_nif_batch(self._h, int(rays_per_batch)))

    def setRayBatch(self, rays_per_batch: int):
        self._check(self._lib.mi_scene_set_ray_batch(self._h, int(rays_per_batch)))

    def set_option(self, key: str, value) -> "IpuScene":
        """Kernel selection / tuning of THIS scene (mi_scene_set_option); never changes a result bit."""
        self._check(self._lib.mi_scene_set_option(self._h, key.encode(), str(value).encode()))
        return self

    def getTraceTimeSecs(self) -> float:
        return float(self._lib.mi_trace_time_secs(self._h))

    def loadNifModel(self, asset_path) -> bool:
        """IpuScene::loadNifModel (src/IpuScene.cpp:174-187): log and return False on any failure."""
        try:
            a = NifAssets(asset_path)
            self.setNif(a.kernels, a.biases, a.relu, a.embedding_dimension, a.max_value, a.mean, a.log_tonemap)
            return True
        except RaylibError as e:
            print(f"[error] {e}", file=sys.stderr)
            return False

    def setNif(self, kernels, biases, relu, embedding_dimension, max_value, mean, log_tonemap=True):
        """Weights as arrays (what loadNifModel ends up calling)."""
        n = len(kernels)
        ks = [np.ascontiguousarray(k, dtype=np.float32) for k in kernels]
        bs = [None if b is None else np.ascontiguousarray(b, dtype=np.float32) for b in biases]
        kp = (C.c_void_p * n)(*[k.ctypes.data for k in ks])
        bp = (C.c_void_p * n)(*[(b.ctypes.data if b is not None else None) for b in bs])
        rows = np.array([k.shape[0] for k in ks], dtype=np.uint32)
        cols = np.array([k.shape[1] for k in ks], dtype=np.uint32)
        rl = np.array([1 if r else 0 for r in relu], dtype=np.uint8)
        mean_a = np.ascontiguousarray(mean, dtype=np.float32)
        self._check(self._lib.mi_scene_set_nif(self._h, n, kp, bp, rows.ctypes.data, cols.ctypes.data, rl.ctypes.data,
                                              int(embedding_dimension), float(max_value), mean_a.ctypes.data,
                                              1 if log_tonemap else 0))

    def run(self, rays: np.ndarray, mode: int | None = None, callback=None) -> np.ndarray:
        """GraphManager().run(ipuScene): trace the HOST ray stream in place. `callback(batch_index, first, count)`
        mirrors IpuScene::RayCallbackFn (one call per finished ray batch, see setRayBatch)."""
        assert rays.dtype == TRACE_RESULT and rays.flags["C_CONTIGUOUS"]
        if mode is None:
            mode = MODE_PATH_TRACE if self.desc.path_trace else MODE_SHADOW_TRACE
        cb = None
        if callback is not None:
            base = rays.ctypes.data
            proto = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t)
            cb = proto(lambda user, idx, ptr, cnt: callback(idx, (ptr - base) // TRACE_RESULT.itemsize, cnt))
        self._check(self._lib.mi_render(self._h, mode, rays.ctypes.data, rays.size, cb, None))
        return rays

    def run_device(self, d_rays_ptr: int, n: int, mode: int, stream: int = 0):
        """Trace a DEVICE-resident ray stream (e.g. a torch uint8 tensor's data_ptr) asynchronously."""
        self._check(self._lib.mi_render_device(self._h, mode, C.c_void_p(d_rays_ptr), n, C.c_void_p(stream)))

    def nif_infer_device(self, d_u: int, d_v: int, d_bgr: int, n: int, stream: int = 0):
        self._check(self._lib.mi_nif_infer_device(self._h, C.c_void_p(d_u), C.c_void_p(d_v), C.c_void_p(d_bgr), n, C.c_void_p(stream)))

    def counters(self) -> dict:
        c = (C.c_uint64 * 4)()
        self._check(self._lib.mi_get_counters(self._h, c))
        return {"casts": c[0], "nodes_visited": c[1], "leaf_tests": c[2], "paths": c[3]}

    def phase_stats(self) -> dict:
        c = (C.c_uint64 * 12)()
        self._check(self._lib.mi_get_phase_stats(self._h, c))
        names = ("node", "leaf", "shade", "gen")
        out = {n: {"iters": c[2 * i], "lanes": c[2 * i + 1]} for i, n in enumerate(names)}
        out["cycles"] = {"traverse": c[8], "shade": c[9], "gen": c[10], "total": c[11]}
        return out

    def pool_stats(self) -> dict:
        c = (C.c_uint64 * 8)()
        self._check(self._lib.mi_get_pool_stats(self._h, c))
        return dict(zip(("loops", "refill_turns", "refill_lanes", "idle", "lost_claims", "bursts", "burst_lanes", "refill_cycles"), [int(x) for x in c]))

    def launch_progress(self, d_samples: int, n: int, period_ticks: int, stream: int = 0) -> None:
        """mi_debug_launch_progress: one wave samples the work counter of `stream`'s persistent launches n times, period_ticks
        (100-MHz ticks) apart, into 2 n uint64 of device memory at d_samples. Call right before run_device on `stream`."""
        self._check(self._lib.mi_debug_launch_progress(self._h, C.c_void_p(stream), C.c_void_p(d_samples), n, period_ticks))

    def nif_timing(self) -> dict:
        """Milliseconds in MLP launches (and their number) since the last call; needs set_option("nif_timing", 1)."""
        c = (C.c_double * 2)()
        self._check(self._lib.mi_get_nif_timing(self._h, c))
        return {"mlp_ms": float(c[0]), "launches": int(c[1])}

    def nif_clock_ghz(self):
        """The shader clock the last K3a launch ran at (mi_get_nif_clock), None when the network runs nif_mlp_kernel."""
        c = (C.c_uint64 * 2)()
        self._check(self._lib.mi_get_nif_clock(self._h, c))
        return (c[0] / c[1] * 0.1) if c[1] else None

    def reset_counters(self):
        self._check(self._lib.mi_reset_counters(self._h))

    def close(self):
        if self._h:
            self._lib.mi_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _BorrowedScene(IpuScene):
    """A replica's scene handle inside an IpuGroup (owned by the group: never destroyed from here)."""

    def __init__(self, handle, desc, lib=None):
        self._lib = lib if lib is not None else device_lib()
        self._h = C.c_void_p(handle)
        self.desc = desc

    def close(self):
        self._h = C.c_void_p()


TRANSPORT_AUTO, TRANSPORT_RCCL, TRANSPORT_COPY = 0, 1, 2


class IpuGroup:
    """An IpuScene with numReplicas > 1 (RuntimeConfig, trace.cpp:297-309) in ONE process: a scene replica per entry
    of `devices` (ordinals may repeat), the host ray stream dealt to them in 8-row bands, one RCCL gather to the first
    replica's device at frame end (mi_group_* in include/mi_raylib.h)."""

    def __init__(self, desc: SceneDesc, devices, transport: int = TRANSPORT_AUTO, variants: bool = False):
        self._lib = device_lib(variants)
        self._h = C.c_void_p()
        self.desc = desc
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        self._check(self._lib.mi_group_create(C.byref(desc), dv.ctypes.data, dv.size, int(transport), C.byref(self._h)))

    _check = IpuScene._check

    def scenes(self):
        return [_BorrowedScene(self._lib.mi_group_scene(self._h, i), self.desc, self._lib) for i in range(self._lib.mi_group_size(self._h))]

    def setRayBatch(self, rays_per_batch: int):
        self._check(self._lib.mi_group_set_ray_batch(self._h, int(rays_per_batch)))

    # The NIF model and its settings go to EVERY replica, as the reference streams the weights to every replica of the
    # replicated graph (src/IpuScene.cpp:535) - what mi::IpuScene::configure does in the C++ host (csrc/host/IpuScene.hpp).
    def setNif(self, kernels, biases, relu, embedding_dimension, max_value, mean, log_tonemap=True):
        for sc in self.scenes():
            sc.setNif(kernels, biases, relu, embedding_dimension, max_value, mean, log_tonemap)

    def loadNifModel(self, asset_path) -> bool:
        return all(sc.loadNifModel(asset_path) for sc in self.scenes())

    def setHdriRotation(self, degrees: float):
        for sc in self.scenes():
            sc.setHdriRotation(degrees)

    def set_option(self, key: str, value) -> "IpuGroup":
        for sc in self.scenes():
            sc.set_option(key, value)
        return self

    def run(self, rays: np.ndarray, mode: int | None = None, callback=None) -> np.ndarray:
        assert rays.dtype == TRACE_RESULT and rays.flags["C_CONTIGUOUS"]
        if mode is None:
            mode = MODE_PATH_TRACE if self.desc.path_trace else MODE_SHADOW_TRACE
        cb = None
        if callback is not None:
            base = rays.ctypes.data
            proto = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t)
            cb = proto(lambda user, idx, ptr, cnt: callback(idx, (ptr - base) // TRACE_RESULT.itemsize, cnt))
        self._check(self._lib.mi_group_render(self._h, mode, rays.ctypes.data, rays.size, cb, None))
        return rays

    def getTraceTimeSecs(self) -> float:
        return float(self._lib.mi_group_trace_time_secs(self._h))

    def counters(self) -> dict:
        c = (C.c_uint64 * 4)()
        self._check(self._lib.mi_group_get_counters(self._h, c))
        return {"casts": c[0], "nodes_visited": c[1], "leaf_tests": c[2], "paths": c[3]}

    def last_transfer(self) -> dict:
        c = (C.c_uint64 * 5)()
        self._check(self._lib.mi_group_last_transfer(self._h, c))
        return {"rccl_messages": c[0], "peer_copies": c[1], "bands": c[2], "upload_copies": c[3], "download_copies": c[4]}

    def last_gather_ms(self) -> float:
        """Milliseconds the last batch's gather took as the first replica's device saw it (HIP events on its stream)."""
        ms = C.c_double()
        self._check(self._lib.mi_group_last_gather_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def devices(self):
        """The distinct device ordinals of the group's communicator, root first (RCCL ranks when the transport is RCCL)."""
        n = C.c_uint32()
        buf = np.zeros(64, np.int32)
        self._check(self._lib.mi_group_devices(self._h, buf.ctypes.data, buf.size, C.byref(n)))
        return [int(x) for x in buf[:n.value]]

    # -- the stages one by one: the shares stay resident on the devices between calls --
    def upload(self, rays: np.ndarray):
        assert rays.dtype == TRACE_RESULT and rays.flags["C_CONTIGUOUS"]
        self._check(self._lib.mi_group_upload(self._h, rays.ctypes.data, rays.size))

    def trace(self, mode: int = MODE_PATH_TRACE):
        self._check(self._lib.mi_group_trace(self._h, mode))

    def download(self, rays: np.ndarray) -> np.ndarray:
        assert rays.dtype == TRACE_RESULT and rays.flags["C_CONTIGUOUS"]
        self._check(self._lib.mi_group_download(self._h, rays.ctypes.data, rays.size))
        return rays

    def gathered_device(self):
        """(device pointer of the gathered shares on the first replica's device, first record of every replica's share)"""
        ptr = C.c_void_p()
        n = self._lib.mi_group_size(self._h)
        off = (C.c_uint64 * (n + 1))()
        self._check(self._lib.mi_group_gathered_device(self._h, C.byref(ptr), off, n + 1))
        return ptr.value, [int(x) for x in off]

    def reset_counters(self):
        self._check(self._lib.mi_group_reset_counters(self._h))

    def close(self):
        if self._h:
            self._lib.mi_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
