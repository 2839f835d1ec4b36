"""CPU tests of the host plumbing and of the C-ABI surface (no GPU compute is attempted).

* built-in scenes match the reference's inventory (SURVEY.md §8 head: box = 6 Cornell meshes +
  Cylinder + Suzanne + 2 spheres + 1 disc => 4035 leaves / 8069 nodes, matIDs {4,0,1,2,0,5,0,0,3,7,6});
* the BVH array obeys the CompactBVH2Node contract (src/CompactBvhBuild.cpp:5-56);
* the un-jittered ray stream equals the oracle's restatement of initPerspectiveRayStream;
* both shared libraries load and export every symbol their header declares.
"""
import ctypes as C
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

import ipu_ray_lib_amd as irl
import oracle_lib as ol

ROOT = Path(__file__).resolve().parent.parent


def half(a):
    return np.asarray(a, dtype=np.uint16).view(np.float16).astype(np.float32)


@pytest.mark.parametrize("name,nodes,tris,geoms", [("box-simple", 63, 32, 6), ("box", 8069, 4032, 11), ("spheres", 11, 0, 6)])
def test_builtin_scene_inventory(name, nodes, tris, geoms):
    s = irl.HostScene.builtin(name)
    assert (s.desc.num_nodes, s.desc.num_tris, s.desc.num_geometry) == (nodes, tris, geoms)
    if name == "box":
        assert list(s.mat_ids) == [4, 0, 1, 2, 0, 5, 0, 0, 3, 7, 6]
        assert [int(m["numTriangles"]) for m in s.mesh_info] == [2, 6, 2, 2, 10, 10, 64, 3936]
        assert s.desc.num_materials == 8 and s.desc.num_spheres == 2 and s.desc.num_discs == 1
        assert np.float32(s.desc.fov_radians) == np.float32(np.pi / 4)
        light = s.materials[4]
        assert light["emissive"] == 1 and np.allclose([light["emission"][k] for k in "xyz"], [13.333333, 7.6949024, 1.7976471])
        # scene is camera-at-origin, looking down -z: all geometry in front of the camera
        assert s.verts["z"].max() < 0
    if name == "spheres":
        assert np.float32(s.desc.fov_radians) == np.float32(np.pi / 2) and s.desc.num_spheres == 5


@pytest.mark.parametrize("name", ["box-simple", "box", "spheres"])
def test_compact_bvh_contract(name):
    s = irl.HostScene.builtin(name)
    n = s.nodes
    N = len(n)
    leaf = n["geomID"] != 0xFFFF
    assert leaf.sum() * 2 - 1 == N                                  # full binary tree, one primitive per leaf
    lo = np.stack([n["min_x"], n["min_y"], n["min_z"]], 1)
    ext = np.stack([half(n["dx"]), half(n["dy"]), half(n["dz"])], 1)
    hi = lo + ext
    # preorder walk: first child adjacent, second child = node after the first child's subtree
    depth_max = 0
    seen = np.zeros(N, bool)
    stack = [(0, 1)]
    end = {}
    order = []
    while stack:
        i, dpt = stack.pop()
        seen[i] = True; order.append(i); depth_max = max(depth_max, dpt)
        if not leaf[i]:
            second = int(n["link"][i])
            assert i + 1 < second < N
            stack.append((second, dpt + 1)); stack.append((i + 1, dpt + 1))
    assert seen.all() and order == list(range(N))                   # depth-first order IS array order
    assert depth_max == s.desc.max_leaf_depth
    # every interior box contains both children (half extents are rounded UP, so up to float rounding of min+ext)
    for i in np.nonzero(~leaf)[0]:
        for c in (i + 1, int(n["link"][i])):
            # each node rounds its own extents up to binary16 (11 significant bits), so allow one half-ulp of the larger box
            assert np.all(lo[c] >= lo[i]) and np.all(hi[c] <= hi[i] + ext[i] * 2.0 ** -9 + 1e-3)
    # leaves: each (geomID, primID) exactly once, and the leaf box is the primitive's box with extents rounded up
    pairs = set()
    for i in np.nonzero(leaf)[0]:
        g, p = int(n["geomID"][i]), int(n["link"][i])
        assert (g, p) not in pairs
        pairs.add((g, p))
        ref = s.geometry[g]
        if ref["type"] == 0:
            mi = s.mesh_info[ref["index"]]
            tri = s.tris[mi["firstIndex"] + p]
            pv = np.array([[s.verts[mi["firstVertex"] + int(k)][c] for c in "xyz"] for k in tri], dtype=np.float32)
            assert np.array_equal(lo[i], pv.min(0))
            d = pv.max(0) - pv.min(0)
            want = np.array([ol.lib().o_round_to_half_not_smaller(float(x)) for x in d], dtype=np.uint16)
            assert np.array_equal(np.array([n["dx"][i], n["dy"][i], n["dz"][i]], dtype=np.uint16), want)
    expected = sum(int(s.mesh_info[r["index"]]["numTriangles"]) if r["type"] == 0 else 1 for r in s.geometry)
    assert len(pairs) == expected


def test_collada_import_config3_scene():
    """BASELINE config 3: assets/test_scene.dae --load-normals. SURVEY.md §8d: 10 meshes, 8 474 triangles =>
    16 947 nodes, 9 materials incl. 2 emissive (shininess 10 / 350 as emission factor), glass by name,
    reflectivity => specular, camera xfov 45 degrees, everything in front of the camera."""
    s = irl.HostScene.import_file(ROOT / "assets" / "test_scene.dae", load_normals=True)
    d = s.desc
    assert (d.num_meshes, d.num_tris, d.num_nodes, d.num_materials) == (10, 8474, 16947, 9)
    assert d.num_normals == d.num_verts > 0
    assert np.float32(d.fov_radians) == np.float32(np.deg2rad(np.float32(45.0)))
    m = s.materials
    assert sorted(int(x) for x in m["type"]) == [0, 0, 0, 0, 1, 1, 2, 2, 2]
    em = m[m["emissive"] == 1]
    assert len(em) == 2
    assert np.allclose(sorted(float(e["emission"]["x"]) for e in em), [10.0, 274.26], rtol=1e-3)
    assert s.verts["z"].max() < 0
    nl = np.sqrt(sum(np.frombuffer(s._view(d.mesh_normals, d.num_normals, irl.VEC3)[k].tobytes(), np.float32) ** 2 for k in "xyz"))
    assert np.allclose(nl, 1.0, atol=1e-5)
    s2 = irl.HostScene.import_file(ROOT / "assets" / "test_scene.dae", load_normals=False)
    assert s2.desc.num_normals == 0 and s2.desc.num_tris == 8474
    h = C.c_void_p()
    assert irl.host_lib().mi_host_scene_import(str(irl.DEFAULT_MESH).encode(), 0, C.byref(h)) != 0
    assert b"No camera found" in irl.host_lib().mi_host_last_error()       # scene_utils.cpp:177-180


def test_collada_import_nif_demo_scene():
    """assets/hdri_test.dae, the reference's NIF demo geometry (6 meshes / 5 656 triangles, an open
    scene lit by the environment only): mesh and triangle inventory, one BVH leaf per triangle, the reference's material
    heuristics (src/scene_utils.cpp:236-289: glass by name -> refractive, reflectivity -> specular, no emitters), a camera
    in the file, everything in front of it, optional per-vertex normals."""
    s = irl.HostScene.import_file(ROOT / "assets" / "hdri_test.dae", load_normals=False)
    d = s.desc
    assert (d.num_meshes, d.num_tris, d.num_geometry, d.num_nodes) == (6, 5656, 6, 2 * 5656 - 1)
    assert [int(t) for t in s.mesh_info["numTriangles"]] == [480, 1152, 12, 12, 3936, 64]
    assert d.num_normals == 0 and d.num_spheres == 0 and d.num_discs == 0
    m = s.materials
    assert d.num_materials == 5 and not m["emissive"].any() and sorted(int(x) for x in m["type"]) == [0, 1, 1, 2, 2]
    assert [int(x) for x in s.mat_ids[:6]] == [0, 1, 2, 3, 3, 4]
    assert 0.5 < d.fov_radians < 1.5 and s.verts["z"].max() < 0
    leaves = s.nodes[s.nodes["geomID"] != irl.INVALID_GEOM]
    assert leaves.size == 5656 and d.max_leaf_depth >= 13
    n = irl.HostScene.import_file(ROOT / "assets" / "hdri_test.dae", load_normals=True)
    assert n.desc.num_tris == 5656 and n.desc.num_normals == n.desc.num_verts > 0


def test_monkey_scene_config5():
    s = irl.HostScene.builtin("monkey")
    assert (s.desc.num_meshes, s.desc.num_tris, s.desc.num_spheres, s.desc.num_discs) == (2, 4000, 0, 0)
    box = irl.HostScene.builtin("box")
    # the bust sits exactly where the box scene has it
    assert np.array_equal(s.verts.view(np.uint8), box.verts[64:].view(np.uint8))


def test_ray_stream_matches_oracle_restatement():
    s = irl.HostScene.builtin("box-simple")
    for (w, h, crop) in [(512, 512, None), (1440, 1440, (37, 19, 700, 300)), (768, 432, None)]:
        s.desc.set_image(w, h, crop)
        got = s.init_ray_stream()
        want = np.zeros_like(got)
        ol.lib().o_init_ray_stream(C.byref(s.desc), want.ctypes.data)
        assert got.tobytes() == want.tobytes()
    # PixelCoord.u = ROW, .v = COLUMN in full-image coordinates even under --crop
    s.desc.set_image(1440, 1440, (4, 3, 100, 200))
    r = s.init_ray_stream()
    assert r.size == 12 and r["u"][0] == 200 and r["v"][0] == 100 and r["u"][-1] == 202 and r["v"][-1] == 103
    assert np.all(np.isinf(r["h"]["r"]["tMax"])) and np.all(r["h"]["geomID"] == 0xFFFF) and np.all(r["h"]["normal"]["z"] == 1)


def test_host_error_behaviour():
    h = C.c_void_p()
    lib = irl.host_lib()
    assert lib.mi_host_scene_builtin(b"no-such-scene", None, C.byref(h)) == 1        # MI_ERR_INVALID_ARG
    assert b"Invalid scene selection" in lib.mi_host_last_error()
    assert lib.mi_host_scene_builtin(b"box", b"/nonexistent/file.glb", C.byref(h)) == 4   # MI_ERR_IO
    assert lib.mi_host_scene_builtin(None, None, C.byref(h)) == 1


def _tiny_glb(path, nodes, indices, n_verts=3, normals=None):
    """A minimal glTF-binary file: one mesh, POSITION (+ optional NORMAL) and u16 indices; `nodes` is the JSON node list."""
    import json, struct
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]][:n_verts], np.float32).tobytes()
    idx = np.array(indices, np.uint16).tobytes()
    idx += b"\0" * (-len(idx) % 4)
    views = [{"buffer": 0, "byteOffset": 0, "byteLength": len(pos)}, {"buffer": 0, "byteOffset": len(pos), "byteLength": len(idx)}]
    accessors = [{"bufferView": 0, "componentType": 5126, "count": n_verts, "type": "VEC3"}, {"bufferView": 1, "componentType": 5123, "count": len(indices), "type": "SCALAR"}]
    attrs = {"POSITION": 0}
    blob = pos + idx
    if normals is not None:
        nb = np.array(normals, np.float32).tobytes()
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(nb)})
        accessors.append({"bufferView": 2, "componentType": 5126, "count": len(normals), "type": "VEC3"})
        attrs["NORMAL"] = 2
        blob += nb
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": nodes, "meshes": [{"primitives": [{"attributes": attrs, "indices": 1}]}],
           "buffers": [{"byteLength": len(blob)}], "bufferViews": views, "accessors": accessors}
    js = json.dumps(doc).encode(); js += b" " * (-len(js) % 4)
    body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(blob), 0x004E4942) + blob
    path.write_bytes(struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body)
    return path


def test_glb_importer_rejects_malformed_files(tmp_path):
    """Input hardening of the glTF-binary reader: indices are range-checked against the vertex count before they are
    narrowed to the reference's 16-bit Triangle indices (include/Primitives.hpp:21-25), a node that is its own
    ancestor is an error instead of endless recursion, and a NORMAL accessor must match POSITION."""
    lib = irl.host_lib()

    def load(path, normals=False):
        h = C.c_void_p()
        rc = lib.mi_host_scene_builtin(b"box", str(path).encode(), C.byref(h)) if not normals else lib.mi_host_scene_builtin(b"monkey", str(path).encode(), C.byref(h))
        msg = lib.mi_host_last_error().decode()
        if rc == 0:
            lib.mi_host_scene_destroy(h)
        return rc, msg

    ok = _tiny_glb(tmp_path / "ok.glb", [{"mesh": 0}], [0, 1, 2])
    assert load(ok)[0] == 0
    rc, msg = load(_tiny_glb(tmp_path / "idx.glb", [{"mesh": 0}], [0, 1, 3]))
    assert rc != 0 and "out of range" in msg
    rc, msg = load(_tiny_glb(tmp_path / "cycle.glb", [{"children": [1]}, {"mesh": 0, "children": [0]}], [0, 1, 2]))
    assert rc != 0 and "cycle" in msg
    rc, msg = load(_tiny_glb(tmp_path / "child.glb", [{"mesh": 0, "children": [7]}], [0, 1, 2]))
    assert rc != 0 and "out of range" in msg


def test_bvh_builder_generic_boxes():
    rng = np.random.default_rng(1)
    for n in (1, 2, 3, 17, 500):
        lo = rng.uniform(-100, 100, (n, 3)).astype(np.float32)
        hi = lo + rng.uniform(0, 5, (n, 3)).astype(np.float32)
        if n == 17:
            lo[:] = lo[0]; hi[:] = hi[0]                          # all boxes identical: degenerate SAH
        gid = np.zeros(n, np.uint16); pid = np.arange(n, dtype=np.uint32)
        nodes = np.zeros(2 * n - 1, dtype=irl.BVH_NODE)
        cnt, depth = C.c_uint32(), C.c_uint32()
        rc = irl.host_lib().mi_build_compact_bvh(lo.ctypes.data, hi.ctypes.data, gid.ctypes.data, pid.ctypes.data, n,
                                                 nodes.ctypes.data, C.byref(cnt), C.byref(depth))
        assert rc == 0 and cnt.value == 2 * n - 1
        leaves = nodes[nodes["geomID"] != 0xFFFF]
        assert sorted(leaves["link"].tolist()) == list(range(n))
        assert depth.value >= int(np.ceil(np.log2(n))) + 1
    # an extent above the largest finite half is an error, as in the reference (CompactBvhBuild.cpp:15-18)
    lo = np.zeros((2, 3), np.float32); hi = np.array([[1, 1, 1], [70000, 1, 1]], np.float32)
    nodes = np.zeros(3, dtype=irl.BVH_NODE); cnt, depth = C.c_uint32(), C.c_uint32()
    rc = irl.host_lib().mi_build_compact_bvh(lo.ctypes.data, hi.ctypes.data, np.zeros(2, np.uint16).ctypes.data,
                                             np.arange(2, dtype=np.uint32).ctypes.data, 2, nodes.ctypes.data, C.byref(cnt), C.byref(depth))
    assert rc != 0 and b"fp16" in irl.host_lib().mi_host_last_error()


def _declared_functions(header: Path):
    text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)) - {"mi_ray_callback"})


def test_device_library_exports_every_declared_symbol():
    """libmi_raylib.so must load on a box without a GPU and export exactly what include/mi_raylib.h
    declares; no compute call is made here."""
    lib = irl.device_lib()
    names = _declared_functions(ROOT / "include" / "mi_raylib.h")
    assert {"mi_scene_create", "mi_scene_destroy", "mi_render", "mi_render_device", "mi_trace_time_secs", "mi_get_counters",
            "mi_scene_set_nif", "mi_scene_set_hdri_rotation", "mi_scene_set_max_nif_batch", "mi_nif_infer_device",
            "mi_last_error", "mi_version"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"libmi_raylib.so does not export {n}"
    assert b"gfx950" in lib.mi_version() and b"+variants" not in lib.mi_version()
    # the test build of the same sources (-DMI_RAYLIB_VARIANTS=1: the measured-and-rejected kernel families) keeps the same ABI
    var = irl.device_lib(variants=True)
    for n in names:
        assert hasattr(var, n), f"libmi_raylib_variants.so does not export {n}"
    assert b"+variants" in var.mi_version()
    hl = irl.host_lib()
    for n in _declared_functions(ROOT / "include" / "mi_scene_host.h"):
        assert hasattr(hl, n), f"libmi_scene_host.so does not export {n}"


def test_device_library_fails_loudly_without_gpu_or_on_bad_input():
    import torch
    s = irl.HostScene.builtin("box-simple")
    if not torch.cuda.is_available():
        with pytest.raises(irl.RaylibError):
            irl.IpuScene(s.desc)                                   # no silent CPU fallback
        with pytest.raises(irl.RaylibError):
            irl.IpuGroup(s.desc, [0, 0])                           # nor for the multi-replica renderer
    h = C.c_void_p()
    assert irl.device_lib().mi_scene_create(None, C.byref(h)) == 1  # MI_ERR_INVALID_ARG
    assert b"null" in irl.device_lib().mi_last_error()
    dev = np.zeros(1, np.int32)
    assert irl.device_lib().mi_group_create(C.byref(s.desc), dev.ctypes.data, 0, 0, C.byref(h)) == 1      # no replicas
    assert irl.device_lib().mi_group_create(C.byref(s.desc), dev.ctypes.data, 65, 0, C.byref(h)) == 1     # too many
    assert irl.device_lib().mi_scene_set_option(None, b"kernel", b"1") == 1


def test_product_does_not_reach_into_the_oracle():
    """The oracle is a checker: nothing under ipu_ray_lib_amd/ may mention it."""
    for p in (ROOT / "ipu_ray_lib_amd").rglob("*"):
        if p.suffix in {".py", ".h", ".hpp", ".hip", ".cpp"}:
            txt = p.read_text()
            assert "ray_oracle" not in txt and "oracle_lib" not in txt and "libray_oracle" not in txt, p


def test_headers_are_plain_c99(tmp_path):
    """The drop-in boundary is a C ABI: both public headers must compile as strict C99 on their own."""
    src = tmp_path / "hdr.c"
    src.write_text('#include "mi_raylib.h"\n#include "mi_scene_host.h"\nint main(void) { return (int)sizeof(mi_trace_result) - 84; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", str(ROOT / "include"), "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_nif_activation_image_layout_is_conflict_free():
    """csrc/nif_kernels.hpp nif_x_byte: the k-chunk-major, piece-swapped LDS image of K3. Re-stated here (the formula is
    read out of the header, so the two cannot drift apart) and enumerated against the LDS rules of
    MI355X_MICROARCH.md: every 16-lane group of a ds_read_b128 B-fragment read hits 16 different 4-bank slots of the
    64-bank row, the epilogue's ds_write_b64 (16 consecutive lanes, 32-bank rule) is at most 2-way, and the map is a
    bijection onto [0, rows * columns * 2)."""
    src = (Path(irl.__file__).parent / "csrc" / "nif_kernels.hpp").read_text()
    m = re.search(r"const uint32_t sw = \(0x([0-9a-fA-F]+)u >> \(2u \* \(\(ray >> 2\) & 3u\)\)\) & 3u;", src)
    assert m, "nif_x_byte changed: update this test"
    packed = int(m.group(1), 16)
    assert "return (col >> 5) * (ROWS * 64u) + ray * 64u + ((((col >> 3) & 3u) ^ sw) << 4) + (col & 7u) * 2u;" in src

    def xbyte(rows, ray, col):
        sw = (packed >> (2 * ((ray >> 2) & 3))) & 3
        return (col >> 5) * (rows * 64) + ray * 64 + ((((col >> 3) & 3) ^ sw) << 4) + (col & 7) * 2

    read_groups = [[*range(0, 4), *range(12, 16), *range(20, 28)], [*range(4, 12), *range(16, 20), *range(28, 32)],
                   [*range(32, 36), *range(44, 48), *range(52, 60)], [*range(36, 44), *range(48, 52), *range(60, 64)]]
    for rows in (96, 128, 192):
        for ks in range(12):
            for mt in range(rows // 16):
                for grp in read_groups:            # B fragment: lane l reads 16 bytes of ray 16m + (l & 15), columns 32ks + 8(l >> 4)..
                    banks = {}
                    for lane in grp:
                        dw = xbyte(rows, 16 * mt + (lane & 15), 32 * ks + 8 * (lane >> 4)) // 4
                        for d in range(4):
                            banks.setdefault((dw + d) % 64, set()).add(dw + d)
                    assert max(len(v) for v in banks.values()) == 1, (rows, ks, mt)
        for nt in range(20):
            for mt in range(rows // 16):
                for g0 in range(0, 64, 16):        # D fragment store: lane l writes 8 bytes of ray 16m + (l & 15), columns 16nt + 4(l >> 4)..
                    banks = {}
                    for lane in range(g0, g0 + 16):
                        dw = xbyte(rows, 16 * mt + (lane & 15), 16 * nt + 4 * (lane >> 4)) // 4
                        for d in range(2):
                            banks.setdefault((dw + d) % 32, set()).add(dw + d)
                    assert max(len(v) for v in banks.values()) <= 2, (rows, nt, mt)
        cols = 384
        seen = {xbyte(rows, r, c) for r in range(rows) for c in range(cols)}
        assert len(seen) == rows * cols and min(seen) == 0 and max(seen) == rows * cols * 2 - 2
        # the compile-time offsets of the k-loop: a ray tile is 1 KiB on, a k-step one chunk (rows x 64 B) on
        for lane in range(64):
            base = xbyte(rows, lane & 15, 8 * (lane >> 4))
            for ks in range(12):
                for mt in range(rows // 16):
                    assert xbyte(rows, 16 * mt + (lane & 15), 32 * ks + 8 * (lane >> 4)) == base + ks * rows * 64 + mt * 1024


def test_builds_are_not_started_where_they_must_not_be(monkeypatch, tmp_path):
    """MI_NO_BUILD=1 (profiled runs: a compiler child would be an exec after the profiler's preload initialised the GPU) turns a
    stale binary into an error instead of a build, and the ranks of a multi-process job that do not build wait for the
    building one (wait_built) - neither ever compiles."""
    import __graft_entry__ as ge
    ge.build_cpu()                                   # fresh by now (conftest built everything): a no-op
    ge.wait_built(5)                                 # ... so the waiting ranks return at once
    monkeypatch.setenv("MI_NO_BUILD", "1")
    with pytest.raises(ge.StaleBinary):
        ge._run(["true"])
    monkeypatch.setattr(ge, "_stale", lambda *a, **k: True)
    with pytest.raises(ge.StaleBinary):
        ge.wait_built(0.3)
    with pytest.raises(ge.StaleBinary):
        ge.build_host()                              # stale + MI_NO_BUILD: refused, nothing is run
