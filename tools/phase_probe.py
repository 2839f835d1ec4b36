"""Lane occupancy of the persistent kernel's turns (instrumented build, MI_RAYLIB_FULL_STATS=1).

    python tools/phase_probe.py [scene] [edge] [spp] [option=value:option=value...]

Prints, per phase, the number of wave turns, the lanes that were in that phase when the turn ran, the mean
occupancy (lanes / 64 turns) and the share of the kernel's cycle counters spent in traversal / shading / ray
generation. The instrumented build is the 4-waves-per-SIMD kernel with rolled NODE steps: shares, not rates.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ipu_ray_lib_amd as irl  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "box"
    edge = int(sys.argv[2]) if len(sys.argv) > 2 else 720
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    s = irl.HostScene.builtin(name)
    s.desc.set_image(edge, edge)
    s.desc.path_trace = 1
    s.desc.samples_per_pixel = spp
    dev = irl.IpuScene(s.desc, variants=len(sys.argv) > 4).set_option("full_stats", 1)      # (extra options: the variants build)
    if len(sys.argv) > 4:
        for kv in sys.argv[4].split(":"):
            k, v = kv.split("=", 1)
            dev.set_option(k, v)
    rays = s.init_ray_stream()
    dev.run(rays, irl.MODE_PATH_TRACE)
    c, p = dev.counters(), dev.phase_stats()
    print(f"{name} {edge}x{edge} x {spp} spp: {c['casts']} casts, {c['nodes_visited'] / c['casts']:.2f} nodes and "
          f"{c['leaf_tests'] / c['casts']:.2f} primitive tests per cast, {c['casts'] / c['paths']:.2f} casts per path")
    for k in ("node", "leaf", "shade", "gen"):
        it, ln = p[k]["iters"], p[k]["lanes"]
        print(f"  {k:5s} turns {it:12d}  lanes {ln:14d}  occupancy {ln / (64.0 * it) if it else 0.0:6.3f}  turns per 64 casts {64.0 * it / c['casts']:7.2f}")
    cy = p["cycles"]
    tot = float(cy["total"]) or 1.0
    print("  cycles: traverse %.3f  shade %.3f  gen %.3f" % (cy["traverse"] / tot, cy["shade"] / tot, cy["gen"] / tot))
    q = dev.pool_stats()      # (the instrumented default kernel leaves two counters there: primitive-test turns that hold a sphere or disc lane, and those lanes)
    if p["leaf"]["iters"]:
        print(f"  primitive-test turns with a sphere / disc lane: {q['loops'] / p['leaf']['iters']:.3f} of them, {q['refill_turns'] / max(q['loops'], 1):.2f} such lanes per such turn")
    ps = dev.pool_stats()
    if ps["bursts"]:      # (only the path-pool kernel leaves these)
        wc = c["casts"] / 64.0
        print("  pool: per 64 casts: loops %.2f  refill turns %.2f (%.1f lanes each)  bursts %.2f (%.1f lanes at start)  idle %.2f  lost claims %.3f; refill cycles %.3f" % (
            ps["loops"] / wc, ps["refill_turns"] / wc, ps["refill_lanes"] / max(ps["refill_turns"], 1), ps["bursts"] / wc,
            ps["burst_lanes"] / max(ps["bursts"], 1), ps["idle"] / wc, ps["lost_claims"] / wc, ps["refill_cycles"] / tot))
    dev.close()


if __name__ == "__main__":
    main()
