#!/usr/bin/env python3
"""tools/collect_profiles.py TAG — turns the gpurun_out/ of tools/gpu_final.sh into the committed summaries:
profiles/<TAG>_<wl>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), profiles/<TAG>_<wl>_pmc_per_launch.csv
(counters averaged per launch and kernel), profiles/traffic.json (HBM bytes per launch, see its _note) and
profiles/<TAG>_bench_<wl>.json (the bench lines)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
# timer name (HtProfScope) of every kernel = the keys of bench.py's kernel_ms_per_step
timer = {"k_gray_linear": "gray", "k_gray_rows": "gray", "k_resample": "resample", "k_resample_bands": "resample", "k_resample_tail": "resample", "k_resample_tail_f64": "resample", "k_pyramid_frame": "resample",
         "k_scan_tiles": "scan_tiles", "k_scan_deep": "scan_deep", "k_scan_deep_lds": "scan_deep", "k_cs_track_fused": "cs_track", "k_cs_hist": "cs_hist",
         "k_cs_meanshift": "cs_meanshift", "k_cs_meanshift_cluster": "cs_meanshift", "k_cs_lut": "cs_lut", "k_cs_init": "cs_init", "k_cs_init_rows": "cs_init"}
traffic = {"_note": "HBM bytes from rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes, tools/gpu_pmc.sh): bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.  "
           "On gfx950 FETCH_SIZE tallies 128-B requests as 64 B (MI355X_MICROARCH.md, HBM section); calibrated on k_gray_linear, whose traffic is known exactly "
           "(reads W*H*4, writes W*H per frame): gray_check.  per_step: bytes per detect step and bench timer name = every launch's OWN counters summed over the "
           "launches the timer covers (resample = the k_resample launches + the tail kernel); per_launch: mean per launch and kernel; "
           "valu_per_step: SQ_INSTS_VALU wave instructions of all kernels of a step (valu_per_launch: mean per launch and kernel)."}
ks5 = glob.glob(os.path.join(G, "prof_c5", "**", "*kernel_stats.csv"), recursive=True)  # C5 (8 x 1080p feeds): kernel stats only
if ks5:
    shutil.copy(ks5[0], os.path.join(P, f"{tag}_c5_kernel_stats.csv"))
for wl in ("c2", "c4", "c3"):
    ks = glob.glob(os.path.join(G, f"prof_{wl}", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(P, f"{tag}_{wl}_kernel_stats.csv"))
    rows_out = []
    per = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(G, f"pmc_{wl}_*"))):
        fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not fs:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(fs[0])):
            m = re.search(r"(k_\w+)", r["Kernel_Name"])
            if not m:
                continue
            agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[m.group(1)].add(r["Dispatch_Id"])
        for k, v in agg.items():
            for c, x in sorted(v.items()):
                rows_out.append((k, len(disp[k]), c, round(x / len(disp[k]), 1)))
                per[k][c] = x / len(disp[k])
    if rows_out:
        with open(os.path.join(P, f"{tag}_{wl}_pmc_per_launch.csv"), "w") as fh:
            fh.write("kernel,launches,counter,value_per_launch\n")
            for r in rows_out:
                fh.write(",".join(map(str, r)) + "\n")
    launches = {r[0]: r[1] for r in rows_out}
    steps = launches.get("k_gray_linear") or launches.get("k_gray_rows") or 0
    pl, ps = {}, collections.defaultdict(float)
    for k, v in per.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v and k in timer and steps:
            b = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
            pl[k] = round(b)
            ps[timer[k]] += b * launches[k] / steps
    # SQ_INSTS_VALU (wave instructions) of every kernel of a step: the numerator of bench.py's valu_issue_frac
    valu = sum(v["SQ_INSTS_VALU"] * launches[k] / steps for k, v in per.items() if "SQ_INSTS_VALU" in v and steps)
    if ps:
        traffic[wl] = {"per_step": {k: round(v) for k, v in ps.items()}, "per_launch": pl, "valu_per_step": round(valu) if valu else None,
                       "valu_per_launch": {k: round(v["SQ_INSTS_VALU"]) for k, v in per.items() if "SQ_INSTS_VALU" in v}}
for name in ("tile_timeline_c2.txt", "tile_timeline_c4.txt", "rs_phases_c2.txt", "rs_phases_c4.txt"):  # shader-clock phase timelines (tools/gpu_tile_timeline.py, gpu_rs_phases.py)
    src = os.path.join(G, name)
    if os.path.exists(src):
        txt = open(src).read()
        if "Traceback" in txt or "Error" in txt or len(txt.strip().splitlines()) < 3:  # round 2 committed tracebacks of stale builds as "timelines"
            print(f"REFUSED {name}: not a timeline (traceback / error / empty)")
            continue
        shutil.copy(src, os.path.join(P, f"{tag}_{name}"))
for name in ("cs_step.txt", "kernel_times.txt", "host_post.txt", "one_frame_trace.txt", "launch_chain.txt", "cs_wall.txt", "ab_bands.txt", "pmc_ab.txt"):
    src = os.path.join(G, name)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, f"{tag}_{name}"))
for name in ("default", "driver", "driver_sub", "c5"):
    bj = os.path.join(G, f"bench_{name}.json")
    if os.path.exists(bj):
        shutil.copy(bj, os.path.join(P, f"{tag}_bench_{name}.json"))
builds = {}
for wl in ("c2", "c4", "c3"):
    bj = os.path.join(G, f"pmc_build_{wl}.json")
    if os.path.exists(bj) and wl in traffic:
        builds[wl] = json.load(open(bj))
if builds:
    vals = list(builds.values())
    if any(v != vals[0] for v in vals):
        raise SystemExit(f"the PMC passes of this gpurun_out/ were taken on different builds: {builds}")
    traffic["_build"] = vals[0]  # code-object fingerprint (benchlib/fingerprint.py) of the library the counters were measured on
else:
    print("WARNING: no pmc_build_*.json next to the counter passes: traffic.json carries no _build and bench.py will call it stale")
if len(traffic) > 1:
    nf = {"c2": (256, 320, 240), "c4": (128, 1280, 720), "c3": (256, 320, 240)}
    traffic["gray_check"] = {wl: {"measured": traffic[wl]["per_step"].get("gray"), "known": nf[wl][0] * nf[wl][1] * nf[wl][2] * 5} for wl in traffic if wl in nf}
    json.dump(traffic, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print(sorted(os.listdir(P)))
