"""Scene importers on user-shaped files (the reference goes through assimp, src/scene_utils.cpp:58-317: Triangulate, one
aiMesh per material group, 32-bit face indices handed to 16-bit `Triangle`s, include/Primitives.hpp:21-25): Collada
<polylist> / <polygons> with n-gons, several primitive groups with their own materials, glb meshes with several
primitives and 8 / 32-bit indices, and meshes of more than 65 536 vertices (split). Small synthetic files, written here."""
import json
import struct

import numpy as np
import pytest

import ipu_ray_lib_amd as irl
import oracle_lib as ol


def _collada(prims_xml, positions, normals=None, materials=("matA", "matB"), z_up=False):
    """A minimal Collada 1.4 document: one geometry (`prims_xml` = its primitive elements), two lambert materials, a camera
    at (0, 0, 6) looking down -z with a 40 degree horizontal field of view."""
    pos = " ".join(f"{v:.9g}" for p in positions for v in p)
    nrm_src = ""
    if normals is not None:
        nrm = " ".join(f"{v:.9g}" for p in normals for v in p)
        nrm_src = (f'<source id="g-nrm"><float_array id="g-nrm-a" count="{3 * len(normals)}">{nrm}</float_array><technique_common>'
                   f'<accessor source="#g-nrm-a" count="{len(normals)}" stride="3"><param name="X" type="float"/><param name="Y" type="float"/>'
                   f'<param name="Z" type="float"/></accessor></technique_common></source>')
    effects = "".join(f'<effect id="{m}-fx"><profile_COMMON><technique sid="common"><lambert><diffuse><color>{0.2 + 0.5 * i} 0.6 {0.8 - 0.5 * i} 1</color></diffuse>'
                      f'{"<emission><color>0.9 0.9 0.9 1</color></emission>" if i == 1 else ""}</lambert></technique></profile_COMMON></effect>' for i, m in enumerate(materials))
    mats = "".join(f'<material id="{m}" name="{m}"><instance_effect url="#{m}-fx"/></material>' for m in materials)
    binds = "".join(f'<instance_material symbol="{m}-sym" target="#{m}"/>' for m in materials)
    return f'''<?xml version="1.0" encoding="utf-8"?>
<COLLADA xmlns="http://www.collada.org/2005/11/COLLADASchema" version="1.4.1">
<asset><up_axis>{"Z_UP" if z_up else "Y_UP"}</up_axis></asset>
<library_cameras><camera id="cam"><optics><technique_common><perspective><xfov>40</xfov><aspect_ratio>1</aspect_ratio><znear>0.1</znear><zfar>100</zfar></perspective></technique_common></optics></camera></library_cameras>
<library_effects>{effects}</library_effects>
<library_materials>{mats}</library_materials>
<library_geometries><geometry id="g"><mesh>
<source id="g-pos"><float_array id="g-pos-a" count="{3 * len(positions)}">{pos}</float_array><technique_common><accessor source="#g-pos-a" count="{len(positions)}" stride="3"><param name="X" type="float"/><param name="Y" type="float"/><param name="Z" type="float"/></accessor></technique_common></source>
{nrm_src}
<vertices id="g-vtx"><input semantic="POSITION" source="#g-pos"/></vertices>
{prims_xml}
</mesh></geometry></library_geometries>
<library_visual_scenes><visual_scene id="scene">
<node id="camnode"><matrix>1 0 0 0 0 1 0 0 0 0 1 6 0 0 0 1</matrix><instance_camera url="#cam"/></node>
<node id="gnode"><matrix>1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1</matrix><instance_geometry url="#g"><bind_material><technique_common>{binds}</technique_common></bind_material></instance_geometry></node>
</visual_scene></library_visual_scenes>
<scene><instance_visual_scene url="#scene"/></scene>
</COLLADA>'''


def _mesh_triangles(scene):
    """Every triangle of the scene as (material index, sorted tuple of its three corner positions): what rendering depends on."""
    out = []
    verts = np.stack([scene.verts["x"], scene.verts["y"], scene.verts["z"]], 1)
    for g, info in enumerate(scene.mesh_info):
        tri = scene.tris[info["firstIndex"]:info["firstIndex"] + info["numTriangles"]].astype(np.int64) + int(info["firstVertex"])
        for t in tri:
            out.append((int(scene.mat_ids[g]), tuple(sorted(tuple(np.round(verts[i], 6)) for i in t))))
    return sorted(out)


# a unit-ish "house": a quad floor, a pentagon wall and a triangle, in two material groups
POS = [(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0), (0, 1.8, 0), (-1, -1, -1), (1, -1, -1), (1.5, 0.2, -0.5)]


def test_collada_polylist_and_polygons_are_cut_into_fans(tmp_path):
    """<polylist> with a quad and a pentagon, and <polygons> with a quad and a triangle, each group with its own material,
    against the same faces written out as <triangles> fans: the same triangles with the same materials, group by group."""
    poly = ('<polylist material="matA-sym" count="2"><input semantic="VERTEX" source="#g-vtx" offset="0"/><vcount>4 5</vcount><p>0 1 2 3  0 1 2 4 3</p></polylist>'
            '<polygons material="matB-sym" count="2"><input semantic="VERTEX" source="#g-vtx" offset="0"/><p>5 6 1 0</p><p>1 6 7</p></polygons>')
    fans = ('<triangles material="matA-sym" count="5"><input semantic="VERTEX" source="#g-vtx" offset="0"/><p>0 1 2  0 2 3   0 1 2  0 2 4  0 4 3</p></triangles>'
            '<triangles material="matB-sym" count="3"><input semantic="VERTEX" source="#g-vtx" offset="0"/><p>5 6 1  5 1 0  1 6 7</p></triangles>')
    (tmp_path / "poly.dae").write_text(_collada(poly, POS))
    (tmp_path / "fans.dae").write_text(_collada(fans, POS))
    a = irl.HostScene.import_file(tmp_path / "poly.dae")
    b = irl.HostScene.import_file(tmp_path / "fans.dae")
    assert a.desc.num_meshes == 2 and a.desc.num_tris == 8 and list(a.mat_ids[:2]) == [0, 1]
    assert _mesh_triangles(a) == _mesh_triangles(b)
    assert a.nodes.tobytes() == b.nodes.tobytes()                      # same triangles in the same order: the same BVH
    assert a.materials["emissive"][1] == 1 and a.materials["emissive"][0] == 0
    # holes are refused by name, a degenerate two-corner "polygon" is dropped like assimp's SortByPType drops lines
    holes = '<polygons material="matA-sym" count="1"><input semantic="VERTEX" source="#g-vtx" offset="0"/><ph><p>0 1 2 3</p><h>4 5 6</h></ph></polygons>'
    (tmp_path / "holes.dae").write_text(_collada(holes, POS))
    with pytest.raises(irl.RaylibError, match="holes"):
        irl.HostScene.import_file(tmp_path / "holes.dae")
    line = '<polylist material="matA-sym" count="2"><input semantic="VERTEX" source="#g-vtx" offset="0"/><vcount>2 3</vcount><p>0 1  0 1 2</p></polylist>'
    (tmp_path / "line.dae").write_text(_collada(line, POS))
    assert irl.HostScene.import_file(tmp_path / "line.dae").desc.num_tris == 1


def test_collada_polylist_with_per_corner_normals(tmp_path):
    """Per-corner normal indices (offset 1) through a quad: a position used with two different normals becomes two vertices,
    and --load-normals hands out one normal per vertex."""
    nrm = [(0, 0, 1), (0, 1, 0)]
    poly = ('<polylist material="matA-sym" count="2"><input semantic="VERTEX" source="#g-vtx" offset="0"/><input semantic="NORMAL" source="#g-nrm" offset="1"/>'
            '<vcount>4 3</vcount><p>0 0 1 0 2 0 3 0   3 1 2 1 4 1</p></polylist>')
    (tmp_path / "n.dae").write_text(_collada(poly, POS, normals=nrm))
    s = irl.HostScene.import_file(tmp_path / "n.dae", load_normals=True)
    assert s.desc.num_tris == 3 and s.desc.num_verts == 7 and s.desc.num_normals == 7      # corners 2 and 3 appear with both normals
    plain = irl.HostScene.import_file(tmp_path / "n.dae", load_normals=False)
    assert plain.desc.num_verts == 5 and plain.desc.num_normals == 0


def _grid(nx, ny):
    xs, ys = np.meshgrid(np.linspace(-2, 2, nx), np.linspace(-2, 2, ny), indexing="xy")
    pos = np.stack([xs.ravel(), ys.ravel(), 0.2 * np.sin(3 * xs.ravel()) * np.cos(2 * ys.ravel())], 1).astype(np.float32)
    quads = []
    for j in range(ny - 1):
        for i in range(nx - 1):
            a = j * nx + i
            quads.append((a, a + 1, a + nx + 1, a + nx))
    return pos, np.array(quads, np.int64)


def test_collada_mesh_of_more_than_65536_vertices_is_split(tmp_path):
    """A 300 x 226 grid of quads (67 800 vertices, 67 275 quads = 134 550 triangles) in ONE <polylist>: `Triangle` indices are
    16 bit (include/Primitives.hpp:21-25), so the importer cuts the group into meshes of at most 65 536 vertices, every
    triangle kept (as position triples), the pieces sharing the group's material."""
    pos, quads = _grid(300, 226)
    p = " ".join(str(int(v)) for v in quads.ravel())
    poly = (f'<polylist material="matB-sym" count="{len(quads)}"><input semantic="VERTEX" source="#g-vtx" offset="0"/>'
            f'<vcount>{" ".join(["4"] * len(quads))}</vcount><p>{p}</p></polylist>')
    (tmp_path / "big.dae").write_text(_collada(poly, pos))
    s = irl.HostScene.import_file(tmp_path / "big.dae")
    info = s.mesh_info
    assert s.desc.num_meshes == 2 and s.desc.num_tris == 2 * len(quads)
    assert int(info["numVertices"].max()) <= 65536 and int(info["numVertices"].sum()) >= len(pos)
    assert list(s.mat_ids[:2]) == [1, 1]
    # every triangle survives with its positions: compare centroids (camera to the origin: a translation by -6 in z; the view
    # matrix's x / z flips and the reference's handedness swap, scene_utils.cpp:300-309, cancel for this camera)
    verts = np.stack([s.verts["x"], s.verts["y"], s.verts["z"]], 1)
    cent = []
    for g in range(2):
        tri = s.tris[info["firstIndex"][g]:info["firstIndex"][g] + info["numTriangles"][g]].astype(np.int64) + int(info["firstVertex"][g])
        assert tri.max() < int(info["firstVertex"][g]) + int(info["numVertices"][g])
        cent.append(verts[tri].mean(1))
    got = np.concatenate(cent)
    world = np.stack([pos[:, 0], pos[:, 1], pos[:, 2] - 6.0], 1)
    want = np.concatenate([world[quads[:, [0, 1, 2]]].mean(1), world[quads[:, [0, 2, 3]]].mean(1)])
    key = lambda a: a[np.lexsort(np.round(a, 4).T)]
    assert np.allclose(key(got), key(want), atol=2e-4)
    assert s.desc.num_nodes == 2 * s.desc.num_tris - 1


def _glb(meshes):
    """A glTF-binary file with one node per entry of `meshes`; an entry is a list of primitives (positions float32 [n, 3],
    indices array whose dtype picks the component type, or None)."""
    bin_, views, accessors, gmeshes = bytearray(), [], [], []
    def add(data, target=None):
        while len(bin_) % 4: bin_.append(0)
        views.append({"buffer": 0, "byteOffset": len(bin_), "byteLength": len(data)}); bin_.extend(data)
        return len(views) - 1
    for prims in meshes:
        gp = []
        for pos, idx in prims:
            pos = np.ascontiguousarray(pos, np.float32)
            accessors.append({"bufferView": add(pos.tobytes()), "componentType": 5126, "count": len(pos), "type": "VEC3",
                              "min": pos.min(0).tolist(), "max": pos.max(0).tolist()})
            prim = {"attributes": {"POSITION": len(accessors) - 1}, "mode": 4}
            if idx is not None:
                ct = {np.dtype("u1"): 5121, np.dtype("<u2"): 5123, np.dtype("<u4"): 5125}[idx.dtype]
                accessors.append({"bufferView": add(np.ascontiguousarray(idx).tobytes()), "componentType": ct, "count": int(idx.size), "type": "SCALAR"})
                prim["indices"] = len(accessors) - 1
            gp.append(prim)
        gmeshes.append({"primitives": gp})
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": list(range(len(meshes)))}],
           "nodes": [{"mesh": i, "translation": [0.0, 0.0, float(i)]} for i in range(len(meshes))],
           "meshes": gmeshes, "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": len(bin_)}]}
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    while len(bin_) % 4: bin_.append(0)
    body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(bin_), 0x004E4942) + bytes(bin_)
    return struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body


def test_glb_with_several_primitives_and_index_widths(tmp_path):
    """A glb whose first mesh has TWO primitives (8-bit and 32-bit indices) and whose second has one without indices: in the
    open 'monkey' scene (every imported mesh gets the white material) three meshes come in, every triangle in place; in the
    built-in box scene - whose material list is the reference's fixed eleven entries, src/scene_utils.cpp:537-544 - a file
    with other than two meshes is refused with the reference's message."""
    quad = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)], np.float32)
    tri = np.array([(0, 0, 1), (1, 0, 1), (0, 1, 1)], np.float32)
    path = tmp_path / "multi.glb"
    path.write_bytes(_glb([[(quad, np.array([0, 1, 2, 2, 3, 0], "u1")), (quad + 2.0, np.array([0, 1, 2, 0, 2, 3], "<u4"))], [(tri, None)]]))
    s = irl.HostScene.builtin("monkey", mesh_file=path)
    assert s.desc.num_meshes == 3
    assert list(s.mesh_info["numTriangles"]) == [2, 2, 1] and list(s.mesh_info["numVertices"]) == [4, 4, 3]
    assert [tuple(t) for t in s.tris] == [(0, 1, 2), (2, 3, 0), (0, 1, 2), (0, 2, 3), (0, 1, 2)]
    with pytest.raises(irl.RaylibError, match="All primitives must be assigned a material"):
        irl.HostScene.builtin("box", mesh_file=path)
    two = tmp_path / "two.glb"
    two.write_bytes(_glb([[(quad, np.array([0, 1, 2, 2, 3, 0], "u1")), (quad + 2.0, np.array([0, 1, 2, 0, 2, 3], "<u4"))]]))
    box = irl.HostScene.builtin("box", mesh_file=two)            # one mesh, two primitives: the places of the cylinder and the monkey
    assert box.desc.num_meshes == 8 and list(box.mesh_info["numTriangles"][-2:]) == [2, 2]


def test_glb_primitive_of_more_than_65536_vertices_is_split(tmp_path):
    pos, quads = _grid(300, 226)
    tris = np.concatenate([quads[:, [0, 1, 2]], quads[:, [0, 2, 3]]]).astype("<u4")
    path = tmp_path / "big.glb"
    path.write_bytes(_glb([[(pos, tris.ravel())]]))
    s = irl.HostScene.builtin("monkey", mesh_file=path)
    info = s.mesh_info
    # (the triangles run over the whole grid twice - first halves of the quads, then second halves - so the greedy cut makes three pieces)
    assert s.desc.num_meshes >= 2 and s.desc.num_tris == len(tris) and int(info["numVertices"].max()) <= 65536
    assert int(info["numVertices"].sum()) >= len(pos) and s.desc.num_nodes == 2 * len(tris) - 1


@pytest.mark.gpu
def test_imported_polylist_scene_renders_like_the_oracle(tmp_path):
    """GPU parity on an imported n-gon scene: a lit room cut from quads and pentagons (one emissive material group),
    path-traced through the C ABI and through the oracle on the same arrays, every TraceResult byte."""
    pos = [(-2, -2, -2), (2, -2, -2), (2, 2, -2), (-2, 2, -2), (-2, -2, 2), (2, -2, 2), (2, 2, 2), (-2, 2, 2), (0, 2.6, -2), (0, 2.6, 2),
           (-0.7, 1.95, -0.7), (0.7, 1.95, -0.7), (0.7, 1.95, 0.7), (-0.7, 1.95, 0.7)]
    poly = ('<polylist material="matA-sym" count="5"><input semantic="VERTEX" source="#g-vtx" offset="0"/><vcount>5 4 4 4 4</vcount>'
            '<p>0 1 2 8 3   0 4 5 1   0 3 7 4   1 5 6 2   3 8 9 7</p></polylist>'
            '<polygons material="matB-sym" count="1"><input semantic="VERTEX" source="#g-vtx" offset="0"/><p>10 11 12 13</p></polygons>')
    (tmp_path / "room.dae").write_text(_collada(poly, pos))
    s = irl.HostScene.import_file(tmp_path / "room.dae")
    d = s.desc
    d.set_image(96, 96); d.samples_per_pixel = 24; d.path_trace = 1
    assert d.num_tris == 3 + 2 * 4 + 2
    got = s.init_ray_stream(); want = got.copy()
    dev = irl.IpuScene(d); dev.run(got, irl.MODE_PATH_TRACE); dev.close()
    ol.path_trace_pixel_rng(d, want, 16)
    assert got.tobytes() == want.tobytes()
    rgb = np.stack([got["rgb"][k] for k in "xyz"], 1)
    assert rgb.sum() > 0 and (got["h"]["primID"] != irl.INVALID_PRIM).mean() > 0.5
