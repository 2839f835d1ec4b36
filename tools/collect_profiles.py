#!/usr/bin/env python3
"""Turns the outputs of tools/prof_rNN.sh (gpurun_out/prof_rNN/) into the files under profiles/ (rNN = the round tag,
first argument, default r03):
  r02_<set>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary (sets c2, c3, spp16, nif, c5)
  r02_<set>_pmc.csv              one row per counter of the set's dominant kernel (mean over the timed launches)
  r02_pmc_summary.json           what bench.py quotes as offline-measured (HBM bytes per launch, VALU figures)
  r02_bench.json                 the un-profiled bench line of the same build
  r02_config5.json               config 5 at 1440^2 x 4000 spp on one GPU + the kernel shares of a profiled 1024-spp run"""
import csv, glob, json, shutil, subprocess, sys
from pathlib import Path
R = Path(__file__).resolve().parent.parent
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
SRC = R / "gpurun_out" / f"prof_{TAG}"
DST = R / "profiles"
COMMIT = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=R).stdout.strip()
CUS, SIMDS = 256, 1024


def pmc(tag, kernel_substr):
    agg = {}
    for f in sorted(glob.glob(str(SRC / tag / "g*" / "**" / "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            if kernel_substr in row["Kernel_Name"]:
                agg.setdefault(row["Counter_Name"], []).append((float(row["Counter_Value"]), row))
    out = {}
    rows = []
    for name, vals in sorted(agg.items()):
        use = vals[1:] if len(vals) > 1 else vals          # the first launch of a run is the warm-up
        mean = sum(v for v, _ in use) / len(use)
        out[name] = mean
        r = dict(use[-1][1]); r["Counter_Value"] = f"{mean:.6f}"; r["Launches_Averaged"] = len(use)
        rows.append(r)
    if rows:
        with open(DST / f"{TAG}_{tag}_pmc.csv", "w", newline="") as o:
            w = csv.DictWriter(o, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    return out


def stats(tag):
    src = SRC / tag / "stats"
    f = next(iter(glob.glob(str(src / "**" / "*kernel_stats.csv"), recursive=True)), None)
    if not f:
        return []
    shutil.copy(f, DST / f"{TAG}_{tag}_kernel_stats.csv")
    return list(csv.DictReader(open(f)))


# the node-gather probe: table + PMC rows at saturation
PROBE = {}
probe_lines = []
if (SRC / "gather_probe.json").exists():
    shutil.copy(SRC / "gather_probe.json", DST / f"{TAG}_gather_probe.json")
    shutil.copy(SRC / "gather_probe.txt", DST / f"{TAG}_gather_probe.txt")
for name in ("tree35", "uni35", "tree64", "lds35"):
    d = SRC / f"probe_{name}"
    if not d.exists():
        continue
    agg = {}
    for f in sorted(glob.glob(str(d / "g*" / "pmc_counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            if "gather_probe" in row["Kernel_Name"]:
                agg.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    st_f = next(iter(glob.glob(str(d / "stats" / "*kernel_stats.csv"))), None)
    if not agg or not st_f:
        continue
    k = [r for r in csv.DictReader(open(st_f)) if "gather_probe" in r["Name"]][0]
    c = {n: sum(v[1:]) / max(len(v[1:]), 1) for n, v in agg.items()}
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    lanes = int(name[-2:]); gathers = 256 * 6 * 4 * lanes * 20000        # 256 CUs x 6 resident workgroups (K1w's launch shape; tools/prof_r04.sh) x 4 waves x lanes x steps
    PROBE[name] = {"avg_ms": float(k["AverageNs"]) / 1e6, "clock_ghz": cyc / (float(k["AverageNs"]) * 1e-9) / 1e9,
                   "l1_accesses_per_clk_per_cu": c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / CUS / cyc, "ta_busy": c.get("TA_TA_BUSY_sum", 0) / CUS / cyc,
                   "lane_gathers_per_clk_per_cu": gathers / CUS / cyc, "l1_accesses_per_lane_gather": c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / gathers,
                   "l1_hit_rate": 1 - c.get("TCP_TCC_READ_REQ_sum", 0) / max(c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 1), 1), "counters": c}
    probe_lines.append(f"probe {name}: {PROBE[name]['avg_ms']:.3f} ms, clock {PROBE[name]['clock_ghz']:.2f} GHz, lane-gathers {PROBE[name]['lane_gathers_per_clk_per_cu']:.3f} /clk/CU, "
                       f"L1 accesses {PROBE[name]['l1_accesses_per_clk_per_cu']:.3f} /clk/CU ({PROBE[name]['l1_accesses_per_lane_gather']:.2f} per lane-gather), TA busy {PROBE[name]['ta_busy']:.3f}, L1 hit rate {PROBE[name]['l1_hit_rate']:.4f}")
if PROBE:
    (DST / f"{TAG}_gather_probe_pmc.json").write_text(json.dumps({"source_commit": COMMIT, "note": "rocprofv3 --pmc passes of tools/gather_probe.py --only path,walk,lanes,6 (tools/prof_gather_probe.sh); lanes in the set's name", "sets": PROBE}, indent=1))
    (DST / f"{TAG}_gather_probe_pmc.txt").write_text("\n".join(probe_lines) + "\n")

bench = json.loads((SRC / "bench.json").read_text())
(DST / f"{TAG}_bench.json").write_text(json.dumps(bench, indent=1))
report = list(probe_lines)
for tag, kern in (("c2", "path_trace_wavefront_kernel<false"), ("c3", "path_trace_wavefront_kernel<false"), ("spp16", "path_trace_wavefront_kernel<false"), ("nif", "nif_asm_kernel" if TAG >= "r05" else "nif_mlp_kernel"), ("nif_w6", "nif_mlp_kernel")):
    if tag == "nif_w6" and not (SRC / tag).exists():
        continue
    c = pmc(tag, kern)
    st = stats(tag)
    k = next((r for r in st if kern.split("<")[0] in r["Name"] and ("<false" in r["Name"] or "nif" in r["Name"])), None)
    line = f"{tag}: "
    if k:
        line += f"{kern.split('<')[0]} avg {float(k['AverageNs']) / 1e6:.3f} ms over {k['Calls']} launches ({k['Percentage']} % of GPU time); "
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0                                  # the counter sums the 8 XCDs
        line += f"VALU busy {c['SQ_INSTS_VALU'] * 2 / (SIMDS * cycles):.3f}, "
        if "SQ_THREAD_CYCLES_VALU" in c:
            line += f"lanes active {c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_ACTIVE_INST_VALU']):.3f}, "
        if "TA_TA_BUSY_sum" in c:
            line += f"TA busy {c['TA_TA_BUSY_sum'] / CUS / cycles:.3f}, "
        if "SQ_WAIT_INST_ANY" in c:
            line += f"waiting {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}, "
    if "FETCH_SIZE" in c:
        line += f"HBM {(2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / 1e9:.2f} GB per launch"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0
        line += (f"MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (SIMDS * cycles):.3f}, waiting {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}, shader clock during the launch "
                 f"{cycles / (float(k['AverageNs']) * 1e-9) / 1e9:.2f} GHz (GRBM_GUI_ACTIVE / 8 / time), LDS instructions per MFMA {c.get('SQ_INSTS_LDS', 0) / max(c.get('SQ_INSTS_MFMA', 1), 1):.3f}, "
                 f"L2 read requests {c.get('TCP_TCC_READ_REQ_sum', 0):.3e}")
    report.append(line)
    if tag == "c2":
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0
        casts = bench["value"] * bench["ms_per_step"] * 1e-3
        summary = {
            "workload": ["box", 1440, 1440, 1000, 1],
            "kernel": "path_trace_wavefront_kernel<false,false,256,6,false,0,true> (SHADE and GEN in one turn; + segment_combine_kernel, not included)",
            "source": f"rocprofv3 --pmc, one group per pass (tools/prof_{TAG}.sh: python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras; profiles/{TAG}_c2_pmc.csv), mean over the timed launches of a pass",
            "source_commit": COMMIT,
            "l1_accesses_per_clk_per_cu": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / CUS / cycles,
            "l1_accesses_per_cast": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / casts,
            "ta_busy": c["TA_TA_BUSY_sum"] / CUS / cycles,
            "probe_l1_accesses_per_clk_per_cu": PROBE.get("tree35", {}).get("l1_accesses_per_clk_per_cu"),
            "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
            "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE counts 128-B requests at 64 B -> doubled; WRITE_SIZE exact. The ray-record reads are 20 B out of every 84-B record, an access width the guide calls uncalibrated, so the doubled figure is an upper bound.",
            "hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
            "algorithmic_hbm_bytes_per_launch": int(2073600 * (84 + 84)),        # a TraceResult in and out once per pixel per frame
            "helper_kernels_note": "per frame, beside this kernel: pixel_coords_kernel (the compact (u, v) copy the (pixel, segment) atoms fetch from) 0.19 GB, segment_combine_kernel 0.48 GB (gpurun_out/r4q PMC passes)",
            "valu": {"insts_per_cast": c["SQ_INSTS_VALU"] / casts, "salu_insts_per_cast": c["SQ_INSTS_SALU"] / casts,
                     "busy": c["SQ_INSTS_VALU"] * 2 / (SIMDS * cycles), "lanes_active": c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]),
                     "useful_lane_ops_frac": c["SQ_INSTS_VALU"] * 2 / (SIMDS * cycles) * c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]),
                     "ta_busy": c["TA_TA_BUSY_sum"] / CUS / cycles, "wave_cycles_waiting": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                     "l1_hit_rate": 1 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"],
                     "note": "a wave64 VALU instruction occupies a SIMD for 2 cycles (MI355X_MICROARCH.md): busy = SQ_INSTS_VALU x 2 / (1024 SIMDs x shader cycles); lanes_active = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)"},
        }
        (DST / f"{TAG}_pmc_summary.json").write_text(json.dumps(summary, indent=1))

# config 5
c5 = json.loads((SRC / "c5_full.json").read_text().strip().splitlines()[0])
st = stats("c5")
tot = sum(float(r["TotalDurationNs"]) for r in st) or 1.0
shares = {r["Name"].split("(")[0][:60]: round(float(r["TotalDurationNs"]) / tot, 4) for r in st if float(r["TotalDurationNs"]) / tot > 0.002}
mlp = next((r for r in st if "nif_asm_kernel" in r["Name"] or "nif_mlp_kernel" in r["Name"]), None)
c5["ms_per_frame_4000spp"] = c5["ms_per_sample"] * 4000
c5["kernel_time_shares_1024spp_profiled"] = shares
if mlp:
    c5["nif_mlp_share"] = round(float(mlp["TotalDurationNs"]) / tot, 4)
c5["k3_mfma_frac_of_2.5PF"] = bench["nif"]["roofline"]["frac"]
c5["note"] = f"tools/bench_config5.py 4000 on one MI355X (device-resident stream, synthetic NIF weights of the reference's shape); shares from rocprofv3 --kernel-trace --stats of the same tool at 1024 spp (two launches of 512 samples) (profiles/{TAG}_c5_kernel_stats.csv)"
(DST / f"{TAG}_config5.json").write_text(json.dumps(c5, indent=1))
report.append(f"c5: {c5['ms_per_sample']:.3f} ms per sample, {c5['ms_per_frame_4000spp'] / 1e3:.2f} s per 4000-spp frame; MLP share {c5.get('nif_mlp_share')}")
if (SRC / "variants.txt").exists():
    shutil.copy(SRC / "variants.txt", DST / f"{TAG}_variants_ab.txt")
report.append(f"bench: {bench['value']:.4e} casts/s, {bench['ms_per_step']:.2f} ms/step, parity {bench['parity_checked_pixels']} px / {bench['parity_mismatches']} mismatches")
(DST / f"{TAG}_profile_summary.txt").write_text("\n".join(report) + "\n")
print("\n".join(report))
