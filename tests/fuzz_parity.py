"""Differential campaign: the HIP path against the CPU oracle on seeded random cases, for a time budget.

    python tests/fuzz_parity.py [seconds=240] [seed=1] [size_scale=1] [nif]

A fixed slice of both campaigns is collected by pytest (tests/test_gpu_parity.py::test_fuzz_campaign_slice, -m gpu);
the script form runs for a time budget. Every case draws
a scene (built-in scenes, or a random triangle soup with nasty triangles: zero-area, needle, axis-aligned and
duplicated/coplanar ones, with or without vertex normals, plus spheres and discs), render parameters (image size
incl. ragged widths, crop window, 1..900 samples so that both segment lengths and their boundaries are crossed, seed, jitter, path length,
roulette depth), a render mode, a kernel variant and (one case in five) the ALLOW_DOUBLE_FALLBACK=1 build on both sides, renders it with the library and with the oracle, and compares
every byte of every TraceResult. The first mismatch stops the run with the case's parameters (exit code 1).

With a fourth argument `nif` the cases are renders with a NIF environment (random small MLP, random samples per
launch and HDRI rotation): the batched persistent-kernel form must equal the literal per-sample loop
(MI_RAYLIB_KERNEL=0) bit for bit - both run the same MLP kernel, so there is no tolerance - whatever the launch
partition.
"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import ipu_ray_lib_amd as irl  # noqa: E402
import oracle_lib as ol  # noqa: E402


def soup(rng):
    n_tris = int(rng.integers(8, 900))
    with_normals = bool(rng.integers(0, 2))
    centers = rng.uniform(-10, 10, (n_tris, 3)); centers[:, 2] -= 40
    verts = centers.repeat(3, 0) + rng.normal(scale=float(rng.choice([0.3, 1.5, 6.0])), size=(3 * n_tris, 3))
    tri = verts.reshape(n_tris, 3, 3)
    kinds = rng.integers(0, 12, n_tris)
    for i in np.nonzero(kinds == 0)[0]:                    # zero area: two equal corners
        tri[i, 2] = tri[i, 1]
    for i in np.nonzero(kinds == 1)[0]:                    # needle
        tri[i, 2] = tri[i, 1] + 1e-4 * (tri[i, 0] - tri[i, 1])
    for i in np.nonzero(kinds == 2)[0]:                    # axis-aligned plane (lies on its bounding box faces)
        tri[i, :, int(rng.integers(0, 3))] = np.float32(tri[i, 0, int(rng.integers(0, 3))])
    for i in np.nonzero(kinds == 3)[0]:                    # duplicate of another triangle: exact ties
        tri[i] = tri[int(rng.integers(0, n_tris))]
    for i in np.nonzero(kinds == 4)[0]:                    # coplanar overlap with the previous triangle
        if i:
            a, b, c = tri[i - 1]
            tri[i] = [a + 0.1 * (b - a), b, c + 0.2 * (a - c)]
    verts = tri.reshape(-1, 3).astype(np.float32)
    half = n_tris // 2
    tris = np.concatenate([np.arange(3 * half).reshape(-1, 3), np.arange(3 * (n_tris - half)).reshape(-1, 3)]).astype(np.uint16)
    v = np.zeros(len(verts), dtype=irl.VEC3); v["x"], v["y"], v["z"] = verts[:, 0], verts[:, 1], verts[:, 2]
    nrm = np.zeros(len(verts) if with_normals else 0, dtype=irl.VEC3)
    if with_normals:
        nn = rng.normal(size=(len(verts), 3)); nn /= np.linalg.norm(nn, axis=1, keepdims=True)
        nrm["x"], nrm["y"], nrm["z"] = nn[:, 0], nn[:, 1], nn[:, 2]
    info = np.zeros(2, dtype=irl.MESH_INFO)
    info[0] = (0, 0, half, 3 * half); info[1] = (half, 3 * half, n_tris - half, 3 * (n_tris - half))
    ns, nd = int(rng.integers(0, 4)), int(rng.integers(0, 3))
    sph = np.zeros(ns, dtype=irl.SPHERE)
    for i in range(ns):
        sph[i] = (*rng.uniform(-8, 8, 2), -40 + rng.uniform(-8, 8), rng.uniform(0.2, 5))
    dsc = np.zeros(nd, dtype=irl.DISC)
    for i in range(nd):
        nv = rng.normal(size=3); nv /= np.linalg.norm(nv)
        if rng.random() < 0.5:
            nv = np.eye(3)[int(rng.integers(0, 3))]
        dsc[i] = (*nv, rng.uniform(2, 30), *rng.uniform(-10, 10, 2), -40 + rng.uniform(-12, 12))
    n_geom = 2 + ns + nd
    mats = np.zeros(4, dtype=irl.MATERIAL)
    for i, (alb, em, ty) in enumerate([((.7, .6, .5), (0, 0, 0), 0), ((.9, .9, .9), (0, 0, 0), 1), ((.8, .9, 1.), (0, 0, 0), 2), ((.5, .5, .5), (3, 2, 1), 0)]):
        mats[i]["albedo"] = alb; mats[i]["emission"] = em; mats[i]["type"] = ty; mats[i]["ior"] = float(rng.choice([1.0, 1.33, 1.52, 2.4])); mats[i]["emissive"] = int(any(em))
    mat_ids = rng.integers(0, 4, n_geom).astype(np.uint32)
    g = irl.SceneDesc()
    keep = [v, nrm, tris, info, sph, dsc, mats, mat_ids]
    g.mesh_info, g.num_meshes = info.ctypes.data, 2
    g.mesh_tris, g.num_tris = tris.ctypes.data, n_tris
    g.mesh_verts, g.num_verts = v.ctypes.data, len(v)
    g.mesh_normals, g.num_normals = (nrm.ctypes.data if with_normals else None), len(nrm)
    g.mat_ids, g.num_mat_ids = mat_ids.ctypes.data, n_geom
    g.materials, g.num_materials = mats.ctypes.data, 4
    g.spheres, g.num_spheres = (sph.ctypes.data if ns else None), ns
    g.discs, g.num_discs = (dsc.ctypes.data if nd else None), nd
    g.fov_radians = float(rng.uniform(0.3, 1.4))
    hs = irl.HostScene.from_arrays(g)
    hs._keep = keep
    return hs, f"soup({n_tris} tris, normals={with_normals}, {ns} spheres, {nd} discs)"


def differing(a, b):
    ab = a.view(np.uint8).reshape(a.size, -1); bb = b.view(np.uint8).reshape(b.size, -1)
    return np.nonzero((ab != bb).any(axis=1))[0]


def nif_weights(rng, hidden, embed, layers):
    F = 4 * embed
    dims = [(F, hidden)] + [((hidden + F) if l == layers // 2 else hidden, hidden) for l in range(1, layers)] + [(hidden, 3)]
    ks = [(rng.normal(size=d) * np.sqrt(2.0 / d[0])).astype(np.float16).astype(np.float32) for d in dims]
    bs = [(rng.normal(size=d[1]) * 0.05).astype(np.float32) for d in dims]
    return ks, bs, [1] * (len(dims) - 1) + [0]


class Mismatch(AssertionError):
    pass


def nif_campaign(budget, seed, max_cases=None, replay_case=None):
    """replay_case: draw the random parameters of every case but render only that one - five times with each batched
    kernel against the literal loop, reporting per kernel - to look at a mismatch a long campaign found."""
    rng = np.random.default_rng(seed)
    builtins = {n: irl.HostScene.builtin(n) for n in ("box-simple", "box", "spheres", "monkey")}
    t_end = time.time() + budget
    case = rays_total = 0
    while time.time() < t_end and (max_cases is None or case < max_cases):
        case += 1
        name = str(rng.choice(list(builtins))); s = builtins[name]; d = s.desc
        w = int(rng.integers(1, 12)) * 8 + int(rng.integers(0, 2)) * int(rng.integers(1, 8))
        h = int(rng.integers(1, 10)) * 8 + int(rng.integers(0, 2)) * int(rng.integers(1, 8))
        d.set_image(w, h)
        spp = int(rng.choice([1, 2, 4, 5, 63, 64, 65, 100, 128, 129, 257, 513, 705])) if rng.random() < 0.5 else int(rng.integers(1, 60))
        if w * h * spp > 4e5:
            spp = max(1, int(4e5 // (w * h)))
        d.samples_per_pixel = spp; d.path_trace = 1
        d.rng_seed = int(rng.integers(0, 2**63)); d.anti_alias_scale = float(rng.choice([0.0, 0.25, 1.0]))
        d.max_path_length = int(rng.integers(1, 12)); d.roulette_start_depth = int(rng.integers(0, 6))
        hidden = int(rng.choice([32, 64, 128, 256, 320])); layers = int(rng.integers(2, 7))
        ks, bs, relu = nif_weights(rng, hidden, 12, layers)
        spl = str(rng.choice(["1", "16", "32", "48", "64", "128", "256", "512", ""]))
        rot = float(rng.uniform(-180, 180))
        # schedule options of the batched form (results must not depend on them): the next batch's trace launch beside the MLP or behind it,
        # on compute units of its own (8 leaves the shader engines unequal), the cast's first box test in a NODE turn
        sched = {"nif_overlap": str(rng.choice(["auto", "0", "1"])), "nif_split": str(rng.choice(["0", "0", "8", "32"])), "nif_first_test": str(rng.choice(["0", "0", "1"]))}
        desc = (f"case {case} (seed {seed}): {name} {w}x{h} spp={spp} rngseed={d.rng_seed} aa={d.anti_alias_scale} len={d.max_path_length} "
                f"roulette={d.roulette_start_depth} mlp={layers}x{hidden} spl={spl or 'default'} rot={rot:.2f} {sched}")

        def render(kernel):
            dev = irl.IpuScene(d, variants=kernel not in ("0", "1")).set_option("kernel", kernel)      # (kernel 3 lives in the variants build)
            if spl:
                dev.set_option("nif_spl", spl)
            if kernel != "0":
                for k, v in sched.items():
                    dev.set_option(k, v)
            dev.setNif(ks, bs, relu, 12, 3.43, np.array([-2.35, -2.26, -1.96], np.float32), True)
            dev.setHdriRotation(rot)
            rays = s.init_ray_stream()
            dev.run(rays, irl.MODE_PATH_TRACE)
            dev.close()
            return rays

        pick = str(rng.choice(["1", "3"]))
        if replay_case is not None:
            if case < replay_case:
                continue
            print(desc, "campaign kernel", pick, flush=True)
            literal = render("0")
            print("literal loop repeatable:", differing(render("0"), literal).size == 0, flush=True)
            for kern in ("1", "3"):
                for rep in range(5):
                    bad = differing(render(kern), literal)
                    print(f"  kernel {kern} run {rep}: {bad.size} TraceResults differ" + (f", first {int(bad[0])}" if bad.size else ""), flush=True)
            return case, 0
        literal, batched = render("0"), render(pick)
        bad = differing(batched, literal)
        if bad.size:
            i = int(bad[0])
            raise Mismatch(f"MISMATCH {desc}\n {bad.size}/{batched.size} TraceResults differ; first at {i}:\n batched {batched[i]}\n literal {literal[i]}")
        rays_total += batched.size
        if case % 20 == 0:
            print(f"{case} NIF cases, {rays_total} TraceResults identical; last: {desc}", flush=True)
    print(f"OK: {case} NIF cases, {rays_total} TraceResults, batched form bit-identical to the literal per-sample loop (seed {seed}, {budget:.0f} s)")
    return case, rays_total


def campaign(budget=240.0, seed=1, scale=1, max_cases=None):
    """image edges up to 160 x scale pixels"""
    rng = np.random.default_rng(seed)
    threads = os.cpu_count() or 8
    builtins = {n: irl.HostScene.builtin(n) for n in ("box-simple", "box", "spheres")}
    t_end = time.time() + budget
    case = 0
    rays_total = 0
    while time.time() < t_end and (max_cases is None or case < max_cases):
        case += 1
        if rng.random() < 0.55:
            s, what = soup(rng)
        else:
            name = str(rng.choice(list(builtins))); s, what = builtins[name], name
        d = s.desc
        w = int(rng.integers(1, 20 * scale)) * 8 + int(rng.integers(0, 2)) * int(rng.integers(1, 8))
        h = int(rng.integers(1, 16 * scale)) * 8 + int(rng.integers(0, 2)) * int(rng.integers(1, 8))
        crop = None
        if rng.random() < 0.4:
            cw, ch = int(rng.integers(1, w + 1)), int(rng.integers(1, h + 1))
            crop = (cw, ch, int(rng.integers(0, w - cw + 1)), int(rng.integers(0, h - ch + 1)))
        d.set_image(w, h, crop)
        spp = int(rng.choice([1, 2, 3, 4, 5, 17, 63, 64, 65, 128, 129, 150, 256, 257, 511, 512, 513, 705, 900])) if rng.random() < 0.5 else int(rng.integers(1, 40))
        if w * h * spp > 1.2e6 * scale:
            spp = max(1, int(1.2e6 * scale // (w * h)))
        d.samples_per_pixel = spp
        d.rng_seed = int(rng.integers(0, 2**63))
        d.anti_alias_scale = float(rng.choice([0.0, 0.25, 1.0, 3.0]))
        d.max_path_length = int(rng.integers(0, 14))
        d.roulette_start_depth = int(rng.integers(0, 7))
        mode = irl.MODE_PATH_TRACE if rng.random() < 0.8 else irl.MODE_SHADOW_TRACE
        d.path_trace = 1 if mode == irl.MODE_PATH_TRACE else 0
        kernel = str(rng.choice(["0", "1", "1", "3", "3", "2"]))
        waves = str(rng.choice(["4", "5", "6", "7"]))
        batch = int(rng.integers(1, 4000)) if rng.random() < 0.25 else 0
        df = int(rng.random() < 0.2)          # the reference's ALLOW_DOUBLE_FALLBACK=1 build, on both sides (Mesh.cpp:38-51)
        merge = int(rng.integers(0, 4) != 0)  # one case in four: SHADE and GEN as two turns (the form of round 3)
        desc = (f"case {case} (seed {seed}): {what} {w}x{h} crop={crop} spp={spp} rngseed={d.rng_seed} aa={d.anti_alias_scale} "
                f"len={d.max_path_length} roulette={d.roulette_start_depth} mode={mode} kernel={kernel} waves={waves} merge={merge} batch={batch} double_fallback={df}")
        spec = int(rng.integers(0, 2))
        # the shipped library when the case draws its default path (or the nested-loop kernel), the variants build of the same
        # sources otherwise: both are under the campaign
        use_variants = kernel not in ("0", "1") or waves != "6" or spec == 1 or merge == 0
        dev = irl.IpuScene(d, variants=use_variants).set_option("kernel", kernel).set_option("waves", waves).set_option("spec", spec).set_option("double_fallback", df)
        # (round 5's two builds of the default form for scenes without vertex normals: drawn off one case in four each)
        dev.set_option("lean_hit", int(rng.integers(0, 4) != 0)).set_option("leaf_rot", int(rng.integers(0, 4) != 0))
        if use_variants:
            dev.set_option("merge", merge)
        got = s.init_ray_stream()
        if rng.random() < 0.3:
            for k in "xyz":
                got["rgb"][k] = rng.random(got.size).astype(np.float32)
        want = got.copy()
        dev.setRayBatch(batch)
        dev.run(got, mode)
        ol.lib().o_set_double_fallback(df)
        try:
            if mode == irl.MODE_PATH_TRACE:
                ol.path_trace_pixel_rng(d, want, threads)
            else:
                ol.shadow_trace(d, want, threads)
        finally:
            ol.lib().o_set_double_fallback(0)
        dev.close()
        bad = differing(got, want)
        if bad.size:
            i = int(bad[0])
            raise Mismatch(f"MISMATCH {desc}\n {bad.size}/{got.size} TraceResults differ; first at {i}:\n got  {got[i]}\n want {want[i]}")
        rays_total += got.size
        if case % 20 == 0:
            print(f"{case} cases, {rays_total} TraceResults identical; last: {desc}", flush=True)
    print(f"OK: {case} cases, {rays_total} TraceResults, all bit-identical to the oracle (seed {seed}, {budget:.0f} s)")
    return case, rays_total


def main():
    try:
        if len(sys.argv) > 4 and sys.argv[4] == "nif":
            nif_campaign(float(sys.argv[1]), int(sys.argv[2]), replay_case=int(sys.argv[5]) if len(sys.argv) > 5 else None)
        else:
            campaign(float(sys.argv[1]) if len(sys.argv) > 1 else 240.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                     int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    except Mismatch as e:
        print(e, flush=True)
        sys.exit(1)


if __name__ == "__main__":
    main()
