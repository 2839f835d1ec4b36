"""The reference's serialised-scene wire format: Serialiser<16> << SceneRef /
Deserialiser<16> >> SceneRef (include/serialisation/*.hpp, src/IpuScene.cpp:51-53).

The reference's serialiser cannot be compiled in this image (boost::alignment and Eigen::half are absent), so
the format is pinned here (a) by the properties the reference's own unit tests assert (tests/test.cpp:38-237:
padding = distance to the next multiple of alignof(T) counted from BaseAlign + offset; a CompactBVH2Node stays
24 bytes; vector/ArrayRef + trailing scalar round trip; in-place aliasing; end-of-stream error) and (b) by an
independent numpy packer of the documented layout."""
import ctypes as C
import struct

import numpy as np
import pytest

import ipu_ray_lib_amd as irl


def _pad(base, offset, align):
    rem = (base + offset) % align
    return (align - rem) % align


def test_padding_rule_matches_reference_unit_test_property():
    """testBasicType<T>(1024) (tests/test.cpp:38-66): after `i` leading bytes, an object of alignment A is written
    with pad = A - (BaseAlign + i) % A (0 when already aligned), for float/half/int32/uint16/int8 alignments."""
    lib = irl.host_lib()
    for base in (1, 2, 4, 8, 16):
        for align in (1, 2, 4, 8, 16):
            for off in range(0, 70):
                assert lib.mi_blob_padding(base, off, align) == _pad(base, off, align)
    assert lib.mi_blob_padding(16, 5, 4) == 3 and lib.mi_blob_padding(16, 6, 2) == 0 and lib.mi_blob_padding(2, 0, 4) == 2


def _numpy_pack(desc, host):
    """Independent packer: u32 count, pad to element alignment, raw element bytes; then the eight scalars."""
    out = bytearray()

    def arr(a, align):
        out.extend(struct.pack("<I", len(a)))
        out.extend(b"\0" * _pad(16, len(out), align))
        out.extend(np.ascontiguousarray(a).tobytes())

    def scalar(fmt, v):
        out.extend(b"\0" * _pad(16, len(out), 4))
        out.extend(struct.pack(fmt, v))

    arr(host.geometry, 2); arr(host.mesh_info, 4); arr(host.tris.reshape(-1, 3), 2); arr(host.verts, 4)
    normals = np.frombuffer(C.string_at(desc.mesh_normals, desc.num_normals * 12), np.float32) if desc.num_normals else np.zeros(0, np.float32)
    arr(normals.reshape(-1, 3), 4); arr(host.mat_ids, 4); arr(host.materials, 4); arr(host.nodes, 4)
    scalar("<I", desc.max_leaf_depth); scalar("<f", desc.image_width); scalar("<f", desc.image_height)
    scalar("<f", desc.fov_radians); scalar("<f", desc.anti_alias_scale); scalar("<I", desc.max_path_length)
    scalar("<I", desc.roulette_start_depth); scalar("<I", desc.samples_per_pixel)
    return bytes(out)


@pytest.mark.parametrize("name", ["box-simple", "box", "spheres"])
def test_serialised_scene_bytes_match_independent_packer(name):
    s = irl.HostScene.builtin(name)
    s.desc.set_image(320, 200); s.desc.samples_per_pixel = 7; s.desc.max_path_length = 9
    blob = irl.serialise_scene(s.desc)
    assert blob.ctypes.data % 16 == 0
    assert blob.tobytes() == _numpy_pack(s.desc, s)
    # the node array is raw 24-byte records (testCompactBvhNode, tests/test.cpp:112-143: stays compact)
    nodes = s.nodes
    assert nodes.tobytes() in blob.tobytes() and nodes.itemsize == 24


def test_deserialise_aliases_in_place_and_round_trips():
    s = irl.HostScene.import_file(irl.REPO_ROOT / "assets" / "test_scene.dae", load_normals=True)
    d = s.desc
    d.set_image(640, 480); d.rng_seed = 99; d.samples_per_pixel = 3
    blob = irl.serialise_scene(d)
    out = irl.SceneDesc()
    out.rng_seed = 1234; out.window_w = 5                       # not part of the blob: must survive
    irl.deserialise_scene(blob, out)
    assert out._blob_bytes_used == blob.size
    lo, hi = blob.ctypes.data, blob.ctypes.data + blob.size
    for field, count, esz in (("geometry", "num_geometry", 4), ("mesh_info", "num_meshes", 16), ("mesh_tris", "num_tris", 6),
                              ("mesh_verts", "num_verts", 12), ("mesh_normals", "num_normals", 12), ("mat_ids", "num_mat_ids", 4),
                              ("materials", "num_materials", 36), ("bvh_nodes", "num_nodes", 24)):
        n = getattr(out, count)
        assert n == getattr(d, count), field
        p_out = C.cast(getattr(out, field), C.c_void_p).value
        p_in = C.cast(getattr(d, field), C.c_void_p).value
        assert lo <= p_out and p_out + n * esz <= hi, field     # a view into the blob, not a copy
        assert C.string_at(p_out, n * esz) == C.string_at(p_in, n * esz), field
    for f in ("max_leaf_depth", "image_width", "image_height", "fov_radians", "anti_alias_scale", "max_path_length",
              "roulette_start_depth", "samples_per_pixel"):
        assert getattr(out, f) == getattr(d, f), f
    assert out.rng_seed == 1234 and out.window_w == 5 and out.num_spheres == 0
    assert d.num_normals == d.num_verts > 0


def test_odd_triangle_count_pads_the_next_count():
    """One triangle = 6 bytes of u16: the following u32 count needs 2 bytes of padding."""
    tri = np.array([[0, 1, 2]], np.uint16)
    verts = np.array([[0, 0, -5], [1, 0, -5], [0, 1, -5]], np.float32)
    g = irl.SceneDesc()
    geom = np.zeros(1, irl.GEOM_REF); mesh = np.zeros(1, irl.MESH_INFO); mesh["numTriangles"] = 1; mesh["numVertices"] = 3
    mats = np.zeros(1, irl.MATERIAL); mat_ids = np.zeros(1, np.uint32)
    g.geometry = geom.ctypes.data; g.num_geometry = 1
    g.mesh_info = mesh.ctypes.data; g.num_meshes = 1
    g.mesh_tris = tri.ctypes.data; g.num_tris = 1
    g.mesh_verts = verts.ctypes.data; g.num_verts = 3
    g.mat_ids = mat_ids.ctypes.data; g.num_mat_ids = 1
    g.materials = mats.ctypes.data; g.num_materials = 1
    s = irl.HostScene.from_arrays(g)
    blob = irl.serialise_scene(s.desc).tobytes()
    # geometry: 4 + 4 ; meshInfo: 4 + 16 ; tris: 4 + 6 -> offset 38, then 2 pad bytes, then the vertex count (3)
    assert blob[28:32] == struct.pack("<I", 1) and blob[38:40] == b"\0\0" and blob[40:44] == struct.pack("<I", 3)
    out = irl.deserialise_scene(irl.serialise_scene(s.desc))
    assert out.num_tris == 1 and out.num_verts == 3 and out.num_nodes == 1


def test_truncated_and_misaligned_blobs_are_rejected():
    s = irl.HostScene.builtin("box-simple")
    blob = irl.serialise_scene(s.desc)
    lib = irl.host_lib()
    for cut in (0, 3, 4, 11, blob.size // 2, blob.size - 1):
        part = irl.aligned_bytes(max(cut, 1))[:cut]
        part[:] = blob[:cut]
        with pytest.raises(irl.RaylibError, match="Deserialiser encountered end of byte stream"):   # Deserialiser.hpp:74
            irl.deserialise_scene(part)
    shifted = irl.aligned_bytes(blob.size + 1)[1:]
    shifted[:] = blob
    with pytest.raises(irl.RaylibError, match="16-byte aligned"):
        irl.deserialise_scene(shifted)
    small = np.zeros(16, np.uint8)
    assert lib.mi_scene_serialise(C.byref(s.desc), small.ctypes.data, small.size, None) != 0
    assert lib.mi_scene_blob_size(C.byref(s.desc)) == blob.size
