#!/usr/bin/env python3
"""Copies the outputs of tools/prof_k1w.sh + `python bench.py > gpurun_out/bench_final.json` from gpurun_out/ into
profiles/ (kernel stats, one PMC row per counter, the HBM summary bench.py reads, the bench line)."""
import csv, glob, json, shutil
from pathlib import Path
R = Path(__file__).resolve().parent.parent
shutil.copy(R / "gpurun_out/k1wprof/stats/k1w_kernel_stats.csv", R / "profiles/r01_final_kernel_stats.csv")
d = json.loads((R / "gpurun_out/bench_final.json").read_text())
(R / "profiles/r01_final_bench.json").write_text(json.dumps(d, indent=1))
rows, hdr, agg = [], None, {}
for f in sorted(glob.glob(str(R / "gpurun_out/k1wprof/g*/**/*counter_collection.csv"), recursive=True)):
    seen = set()
    for row in csv.DictReader(open(f)):
        hdr = list(row.keys())
        if "path_trace_wavefront_kernel<false" not in row["Kernel_Name"]:
            continue
        if row["Counter_Name"] not in seen:
            seen.add(row["Counter_Name"]); rows.append(row)
        if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            agg.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
rows.sort(key=lambda r: r["Counter_Name"])
with open(R / "profiles/r01_final_pmc.csv", "w", newline="") as o:
    w = csv.DictWriter(o, fieldnames=hdr); w.writeheader(); w.writerows(rows)
fs = sum(agg["FETCH_SIZE"]) / len(agg["FETCH_SIZE"]); ws = sum(agg["WRITE_SIZE"]) / len(agg["WRITE_SIZE"])
p = R / "profiles/r01_pmc_hbm_summary.json"
s = json.loads(p.read_text()); s["FETCH_SIZE_KB"] = fs; s["WRITE_SIZE_KB"] = ws; s["hbm_bytes_per_launch"] = int((2 * fs + ws) * 1024)
p.write_text(json.dumps(s, indent=1))
stats = list(csv.DictReader(open(R / "profiles/r01_final_kernel_stats.csv")))
k = next(r for r in stats if "path_trace_wavefront_kernel<false" in r["Name"])
print(f"bench {d['value']:.4e} casts/s, {d['ms_per_step']:.2f} ms/step; rocprofv3 stats {float(k['AverageNs']) / 1e6:.2f} ms avg over {k['Calls']} launches; "
      f"FETCH {fs:.0f} KB WRITE {ws:.0f} KB -> {s['hbm_bytes_per_launch'] / 1e9:.2f} GB per launch; {len(rows)} PMC rows")
