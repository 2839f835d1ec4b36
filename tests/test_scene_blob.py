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


# alignof() of every element type the SceneRef serialiser writes, each taken from the reference declaration it cites
# (Serialiser::write(const T*, n) pads an array to alignof(T), include/serialisation/Serialiser.hpp:33-60).
REF_ALIGN = {
    "GeomRef": 2,           # include/Scene.hpp:29-34          u16 index + u8 type + u8 pad
    "MeshInfo": 4,          # include/Mesh.hpp:15-20           4 x u32
    "Triangle": 2,          # include/Primitives.hpp:21-25     packed, aligned(alignof(uint16_t))
    "Vec3fa": 4,            # include/embree_utils/geometry.hpp:25-27   VEC3_ALIGN 4
    "u32": 4,
    "Material": 4,          # include/Material.hpp:8-35        Vec3fa, float, Vec3fa, enum, bool
    "CompactBVH2Node": 8,   # include/CompactBVH2Node.hpp:52-53  __attribute__((aligned(8)))
}


def _numpy_pack(desc, host, layout=None):
    """Independent packer: u32 count, pad to element alignment, raw element bytes; then the eight scalars.
    `layout`, when given, receives {array name: (offset of the count, pad bytes, offset of the first element)}."""
    out = bytearray()

    def arr(name, a, elem):
        out.extend(b"\0" * _pad(16, len(out), REF_ALIGN["u32"]))      # `s << size` is a u32 write (serialisation.hpp:21-32)
        at = len(out)
        out.extend(struct.pack("<I", len(a)))
        pad = _pad(16, len(out), REF_ALIGN[elem])
        out.extend(b"\0" * pad)
        if layout is not None:
            layout[name] = (at, pad, len(out))
        out.extend(np.ascontiguousarray(a).tobytes())

    def scalar(fmt, v):
        out.extend(b"\0" * _pad(16, len(out), 4))
        out.extend(struct.pack(fmt, v))

    arr("geometry", host.geometry, "GeomRef"); arr("meshInfo", host.mesh_info, "MeshInfo")
    arr("meshTris", host.tris.reshape(-1, 3), "Triangle"); arr("meshVerts", host.verts, "Vec3fa")
    normals = np.frombuffer(C.string_at(desc.mesh_normals, desc.num_normals * 12), np.float32) if desc.num_normals else np.zeros(0, np.float32)
    arr("meshNormals", normals.reshape(-1, 3), "Vec3fa"); arr("matIDs", host.mat_ids, "u32")
    arr("materials", host.materials, "Material"); arr("bvhNodes", host.nodes, "CompactBVH2Node")
    scalar("<I", desc.max_leaf_depth); scalar("<f", desc.image_width); scalar("<f", desc.image_height)
    scalar("<f", desc.fov_radians); scalar("<f", desc.anti_alias_scale); scalar("<I", desc.max_path_length)
    scalar("<I", desc.roulette_start_depth); scalar("<I", desc.samples_per_pixel)
    return bytes(out)


def _soup_scene(num_tris, num_materials):
    """A small triangle soup with a chosen triangle and material count (both move the node array's offset)."""
    rng = np.random.default_rng(num_tris * 31 + num_materials)
    verts = (rng.random((3 * num_tris, 3), np.float32) * 4 - 2).astype(np.float32)
    verts[:, 2] -= 6
    tri = np.arange(3 * num_tris, dtype=np.uint16).reshape(-1, 3)
    g = irl.SceneDesc()
    geom = np.zeros(1, irl.GEOM_REF); mesh = np.zeros(1, irl.MESH_INFO)
    mesh["numTriangles"] = num_tris; mesh["numVertices"] = 3 * num_tris
    mats = np.zeros(num_materials, irl.MATERIAL); mat_ids = np.zeros(1, np.uint32)
    g.geometry = geom.ctypes.data; g.num_geometry = 1
    g.mesh_info = mesh.ctypes.data; g.num_meshes = 1
    g.mesh_tris = tri.ctypes.data; g.num_tris = num_tris
    g.mesh_verts = verts.ctypes.data; g.num_verts = 3 * num_tris
    g.mat_ids = mat_ids.ctypes.data; g.num_mat_ids = 1
    g.materials = mats.ctypes.data; g.num_materials = num_materials
    s = irl.HostScene.from_arrays(g)
    s._keep = (geom, mesh, tri, verts, mats, mat_ids)
    return s


def test_node_array_is_padded_to_the_reference_alignment_of_8():
    """CompactBVH2Node is __attribute__((aligned(8))) (include/CompactBVH2Node.hpp:52-53), so the serialiser puts
    4 pad bytes between the node count and the first node whenever the count ends on an offset that is 4 mod 8
    (Serialiser.hpp:33-60), and the deserialiser skips them (deserialisation.hpp:31-38). Both cases are built here
    (a 36-byte Material more or less flips the residue) and checked byte by byte."""
    seen = set()
    for num_materials in (1, 2, 3, 4):
        s = _soup_scene(5, num_materials)
        layout = {}
        want = _numpy_pack(s.desc, s, layout)
        blob = irl.serialise_scene(s.desc)
        count_at, pad, first = layout["bvhNodes"]
        assert pad == (8 - (16 + count_at + 4) % 8) % 8 and pad in (0, 4)
        assert first % 8 == 0                                       # (the blob itself is 16-byte aligned)
        seen.add(pad)
        got = blob.tobytes()
        assert got == want
        assert len(got) == irl.host_lib().mi_scene_blob_size(C.byref(s.desc))
        assert got[count_at:count_at + 4] == struct.pack("<I", s.nodes.size)
        assert got[count_at + 4:first] == b"\0" * pad
        assert got[first:first + 24 * s.nodes.size] == s.nodes.tobytes()
        out = irl.deserialise_scene(blob)
        assert C.cast(out.bvh_nodes, C.c_void_p).value == blob.ctypes.data + first
        assert out.num_nodes == s.nodes.size and out.max_leaf_depth == s.desc.max_leaf_depth
        assert out.samples_per_pixel == s.desc.samples_per_pixel
    assert seen == {0, 4}, "both residues of the node array's offset must be exercised"


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


# ------------------------------------------------------------------------------------------------------
# Pinned by the reference's own reader: Deserialiser<16> (include/serialisation/Deserialiser.hpp:14-89) compiled from
# the checkout (oracle/ref_driver.cpp, ref_walk_scene_blob: deserialiseArrayRef's calls, deserialisation.hpp:31-59,
# one by one) walked 24 blobs written by mi_scene_serialise; tests/golden/gen_ref_vectors.py stored the blobs and what
# the reference reader found in them. (GeomRef / MeshInfo / Triangle / CompactBVH2Node cannot be compiled here - Eigen -
# and stand in that walk as PODs with the alignment their declarations state; Vec3fa, Material and u32 are the
# reference's own types; the padding rule, the count encoding and the order are its running code throughout.)
# ------------------------------------------------------------------------------------------------------
GOLD = np.load(__import__("pathlib").Path(__file__).parent / "golden" / "ref_l0_vectors.npz")
FIELDS = (("geometry", "num_geometry"), ("mesh_info", "num_meshes"), ("mesh_tris", "num_tris"), ("mesh_verts", "num_verts"),
          ("mesh_normals", "num_normals"), ("mat_ids", "num_mat_ids"), ("materials", "num_materials"), ("bvh_nodes", "num_nodes"))


def _gen_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_ref_vectors", __import__("pathlib").Path(__file__).parent / "golden" / "gen_ref_vectors.py")
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def _product_walk(blob):
    out = irl.deserialise_scene(blob)
    walk = []
    for ptr, num in FIELDS:
        n = getattr(out, num)
        p = C.cast(getattr(out, ptr), C.c_void_p).value
        # (a null pointer is only handed out for an EMPTY normals array; its offset is then not observable)
        walk += [None if p is None else p - blob.ctypes.data, n]
    sc = np.array([out.max_leaf_depth, 0, 0, 0, 0, out.max_path_length, out.roulette_start_depth, out.samples_per_pixel], np.uint32)
    sc[1:5] = np.array([out.image_width, out.image_height, out.fov_radians, out.anti_alias_scale], np.float32).view(np.uint32)
    return walk, [int(x) for x in sc], out._blob_bytes_used


def test_padding_rule_against_the_reference_deserialiser():
    """Deserialiser<16>::calculatePadding<T>() at every offset 0..63 for alignments 1, 2, 4 (Vec3fa) and 8."""
    table = GOLD["blob_padding_table"]
    lib = irl.host_lib()
    for k, align in enumerate((1, 2, 4, 8)):
        for off in range(64):
            assert lib.mi_blob_padding(16, off, align) == table[k, off], (align, off)


def test_blob_reader_and_writer_against_the_reference_deserialiser():
    gen = _gen_module()
    sizes = GOLD["blob_case_sizes"]; data = GOLD["blob_case_bytes"]; starts = [0] + [int(x) for x in np.cumsum(sizes.astype(np.int64))]
    node_residues = set()
    for case, cnt in enumerate(GOLD["blob_case_counts"]):
        gold = data[starts[case]:starts[case + 1]]
        # the product's writer still produces the bytes the reference reader was given ...
        d, keep = gen.blob_case_desc(irl, [int(x) for x in cnt], case)
        mine = irl.serialise_scene(d)
        assert mine.tobytes() == gold.tobytes(), f"case {case}: the writer's bytes changed - regenerate the goldens with the reference reader"
        # ... and the product's reader finds in them what the reference's reader found
        blob = irl.aligned_bytes(gold.size); blob[:] = gold
        walk, sc, used = _product_walk(blob)
        want = [int(x) for x in GOLD["blob_case_walk"][case]]
        for k in range(8):
            assert walk[2 * k + 1] == want[2 * k + 1], (case, FIELDS[k][1])
            if walk[2 * k] is not None:
                assert walk[2 * k] == want[2 * k], (case, FIELDS[k][0], walk[2 * k], want[2 * k])
        assert sc == [int(x) for x in GOLD["blob_case_scalars"][case]], case
        assert used == gold.size == sizes[case]
        count_at = want[12] + want[13] * 36                # the node count follows the materials (4-aligned already)
        node_residues.add((want[14] - (count_at + 4), want[14] % 8))
    assert node_residues == {(0, 0), (4, 0)}              # the node array always starts 8-aligned, after 0 or 4 pad bytes: both cases are in the set


def test_truncated_blob_against_the_reference_deserialiser():
    """Every proper prefix of a blob makes the reference reader throw 'Deserialiser encountered end of byte stream.'
    (golden: -1 for all of them); the product's reader must refuse exactly those."""
    sizes = GOLD["blob_case_sizes"]; data = GOLD["blob_case_bytes"]; starts = [0] + [int(x) for x in np.cumsum(sizes.astype(np.int64))]
    b5 = data[starts[5]:starts[6]]
    res = GOLD["blob_truncation_result"]
    assert res.size == b5.size and (res == -1).all()
    for cut in range(0, b5.size):
        part = irl.aligned_bytes(max(cut, 1))[:cut]; part[:] = b5[:cut]
        with pytest.raises(irl.RaylibError, match="Deserialiser encountered end of byte stream"):
            irl.deserialise_scene(part)


def test_real_scene_blobs_against_the_live_reference_deserialiser():
    """Where oracle/_ref is built (the container that holds the reference checkout): the box scene's and
    test_scene.dae's blobs through the reference reader, live."""
    import oracle_lib
    r = oracle_lib.ref_lib()
    if r is None or not hasattr(r, "ref_walk_scene_blob"):
        pytest.skip("oracle/_ref not built here (needs the reference checkout)")
    for s in (irl.HostScene.builtin("box"), irl.HostScene.import_file(irl.REPO_ROOT / "assets" / "test_scene.dae", load_normals=True)):
        blob = irl.serialise_scene(s.desc)
        want, wsc, wused = oracle_lib.ref_walk_scene_blob(blob)
        walk, sc, used = _product_walk(blob)
        assert [w for w in walk if w is not None] == [w for w, g in zip(want, walk) if g is not None]
        assert sc == wsc and used == wused == blob.size
