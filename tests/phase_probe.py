import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["MI_RAYLIB_FULL_STATS"] = "1"
import numpy as np, ipu_ray_lib_amd as irl
s = irl.HostScene.builtin("box"); d = s.desc; d.set_image(720, 720); d.samples_per_pixel = 16
dev = irl.IpuScene(d); rays = s.init_ray_stream(); dev.run(rays, irl.MODE_PATH_TRACE)
c = dev.counters(); p = dev.phase_stats(); print(c); 
cyc = p.pop("cycles")
tot = sum(v["iters"] for v in p.values())
print("cycle shares: traverse %.2f shade %.2f gen %.2f other(vote,fetch) %.2f" % tuple([cyc[k] / cyc["total"] for k in ("traverse", "shade", "gen")] + [1 - (cyc["traverse"] + cyc["shade"] + cyc["gen"]) / cyc["total"]]))
for k, v in p.items():
    print(k, "iters/cast*64 %.2f" % (v["iters"] * 64 / c["casts"]), "avg lanes %.1f" % (v["lanes"] / max(v["iters"], 1)), "share of iters %.2f" % (v["iters"] / tot))
print("trace time", dev.getTraceTimeSecs())
