#!/usr/bin/env python3
"""Box tests and primitive tests per cast of a scene's BVH, counted by the CPU oracle (no GPU): the figures the host
builder's options are judged by (csrc/host/bvh_sah.cpp, scene_builtin.cpp).

    [MI_BVH_PRESPLIT=..] [MI_BVH_REINSERT=..] [MI_BVH_ORDER=..] python tools/bvh_eval.py [scene|file ...] [--size 160] [--spp 8]

Prints nodes, leaves, depth, box tests / cast (V), primitive tests / cast (T) and the VALU estimate 26 V + 224 T (the
static instruction counts of K1w's box-test and triangle-test steps)."""
import argparse, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import ipu_ray_lib_amd as irl
import oracle_lib as ol


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scenes", nargs="*", default=["box"])
    ap.add_argument("--size", type=int, default=160); ap.add_argument("--spp", type=int, default=8)
    a = ap.parse_args()
    for name in a.scenes:
        s = irl.HostScene.import_file(name, load_normals=True) if Path(name).suffix else irl.HostScene.builtin(name)
        d = s.desc
        d.set_image(a.size, a.size); d.samples_per_pixel = a.spp; d.path_trace = 1
        rays = s.init_ray_stream()
        st = ol.path_trace_pixel_rng(d, rays, 8)
        v, t = st.nodesVisited / st.casts, st.leafTests / st.casts
        leaves = int((s.nodes["geomID"] != 0xFFFF).sum())
        print(f"{name:28s} nodes {len(s.nodes):7d} leaves {leaves:7d} depth {d.bvh_max_depth if hasattr(d, 'bvh_max_depth') else '-'}  "
              f"V {v:6.3f}  T {t:5.3f}  26V+224T {26 * v + 224 * t:7.1f}  casts/path {st.casts / st.paths:.3f}")


if __name__ == "__main__":
    main()
