"""The ONE JSON line of bench.py: a compact headline (the contract's keys + a few scalars, < 4 KB) — the full record
tree goes to a side file.  Round 4's line carried every sub-record inline, grew to 21 KB, and the driver could not
parse it (BENCH_r04.json: parsed = null)."""
import json
import os

from .common import ROOT

LINE_CAP = 4000  # hard cap in bytes (asserted); the target is < 2 KB so that even a 2 000-character tail holds it whole
SAMPLE_CAP = 200


def short(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def compact_roofline(r):
    if not r:
        return None
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_step",
            "kernel_ms_per_step")
    out = {k: r.get(k) for k in keep}
    out["traffic_source"] = "profiles/traffic.json (PMC pass, not this run)" if r.get("traffic") else None
    if r.get("traffic_stale"):
        out["traffic_stale"] = True
        out["traffic_source"] = "profiles/traffic.json (PMC pass of ANOTHER build: stale)"
    co = r.get("co_dominant") or {}
    if co:
        out["co_dominant"] = {k: v["frac"] for k, v in co.items()}
    return out


def compact_cpu(c):
    if not c:
        return None
    out = {k: c.get(k) for k in ("value", "unit", "cores", "kind")}
    out["sample"] = short(c.get("sample", ""), SAMPLE_CAP)
    return out


def compact_config(cfg):
    keep = ("frames_per_gpu", "frames_total", "streams_per_gpu", "feeds_per_gpu", "width", "height",
            "batches_in_flight", "steps_in_flight", "track_calls_per_step")
    out = {"workload": short(cfg.get("workload", ""), 140)}
    out.update({k: cfg[k] for k in keep if k in cfg})
    if cfg.get("parallelism"):
        out["parallelism"] = short(cfg["parallelism"], 90)
    return out


def get(d, *path):
    for p in path:
        if not isinstance(d, dict) or d.get(p) is None:
            return None
        d = d[p]
    return d


def compose(metric, prim, sub, world, extra=None):
    """headline line (dict) from the primary record and the sub-records"""
    line = {"metric": metric, "value": prim["value"], "unit": prim.get("unit", "frames/s"), "n_gpus": world,
            "ranks": world, "steps": prim["steps"], "warmup": prim["warmup"], "ms_per_step": prim["ms_per_step"],
            "ms_per_step_min": prim.get("ms_per_step_min"), "ms_per_step_max": prim.get("ms_per_step_max"),
            "rounds": prim.get("rounds"), "higher_is_better": True, "scaling": prim["scaling"], "vs_baseline": None,
            "dtype": "u8", "data": "synthetic", "config": compact_config(prim["config"]),
            "roofline": compact_roofline(prim.get("roofline")), "cpu_baseline": compact_cpu(prim.get("cpu_baseline")),
            "launched_by_bench": os.environ.get("HT_BENCH_LAUNCHED") == "1"}
    if world > 1:
        line["cpu_baseline_note"] = "N = 1 only"
    scal = {
        "vs_cpu": prim.get("vs_cpu"),
        # the C port of the same arithmetic, one thread / one thread per host core (SURVEY.md §8(d)'s optional second line)
        "cpu_port_1core_value": get(prim, "cpu_baseline_port", "value"),
        "cpu_port_all_cores_value": get(prim, "cpu_baseline_port", "all_cores", "value"),
        "cpu_port_all_cores": get(prim, "cpu_baseline_port", "all_cores", "cores"),
        "path_hbm_frac": prim.get("path_hbm_frac"), "wall_hbm_frac": prim.get("wall_hbm_frac"),
        "valu_issue_frac": get(prim, "valu_issue", "frac_wall"),
        "traffic_stale": prim.get("traffic_stale") or None,  # only when true: the PMC constants belong to another build (valu_issue_frac is then absent)
        "device_ms_per_step": prim.get("device_ms_per_step"),
        "depth1_ms_per_step": get(prim, "depth1", "ms_per_step"),
        "pcie_inclusive_value": get(prim, "pcie_inclusive", "value"),
        "pcie_h2d_gbs": get(prim, "pcie_inclusive", "h2d_gbs"),
        "allgather_verified": prim.get("allgather_verified"),
        "rank_ms_per_step_min": prim.get("rank_ms_per_step_min"),
        "rank_ms_per_step_max": prim.get("rank_ms_per_step_max"),
        "parity_exact": prim.get("parity_exact"),
    }
    c4 = sub.get("c4_1gpu") or sub.get("c4") or {}
    if c4:
        scal.update({
            "value_720p": c4.get("value"), "ms_per_step_720p": c4.get("ms_per_step"),
            "roofline_frac_720p": get(c4, "roofline", "frac"), "roofline_kernel_720p": get(c4, "roofline", "kernel"),
            "path_hbm_frac_720p": c4.get("path_hbm_frac"), "wall_hbm_frac_720p": c4.get("wall_hbm_frac"),
            "valu_issue_frac_720p": get(c4, "valu_issue", "frac_wall"),
            "traffic_stale_720p": c4.get("traffic_stale") or None,
            "depth1_ms_per_step_720p": get(c4, "depth1", "ms_per_step"),
            "pcie_inclusive_value_720p": get(c4, "pcie_inclusive", "value"),
            "cpu_baseline_720p": get(c4, "cpu_baseline", "value"),
            "cpu_port_all_cores_value_720p": get(c4, "cpu_baseline_port", "all_cores", "value"),
            "north_star_720p_vs_reference_js": c4.get("vs_cpu"),  # target: >= 30x on 1280x720 detect at 1 GPU
            "allgather_verified_720p": c4.get("allgather_verified"),
        })
    if sub.get("c4_strong"):
        scal["c4_strong_value"] = sub["c4_strong"].get("value")
    if sub.get("c2_large"):
        scal["c2_large_value"] = sub["c2_large"].get("value")
    if sub.get("c3"):
        scal.update({"c3_value": sub["c3"].get("value"), "c3_roofline_frac": get(sub["c3"], "roofline", "frac")})
    c5 = sub.get("c5") or {}
    if c5:
        scal.update({"c5_value": c5.get("value"), "c5_feeds_per_gpu": get(c5, "config", "feeds_per_gpu"),
                     "c5_one_feed_value": get(c5, "one_feed", "value"),
                     "c5_pcie_inclusive_value": get(c5, "pcie_inclusive", "value"),
                     "c5_latency_p50_ms": get(c5, "latency_ms", "p50"), "c5_latency_p99_ms": get(c5, "latency_ms", "p99"),
                     "c5_track_step_device_ms": get(c5, "device_ms", "track_step"),
                     "c5_vs_cpu": c5.get("vs_cpu")})
    par = [f"{k} {v['parity_exact']}" + (f" + best faces {v['parity_detect_exact']}" if v.get("parity_detect_exact") else "")
           for k, v in (("c3", sub.get("c3")), ("c5", c5)) if v and v.get("parity_exact")]
    if par:
        scal["parity_exact"] = "; ".join(par) + " vs oracle, in this run"
    lat = sub.get("latency_1frame") or {}
    for k, v in lat.items():
        scal[f"latency_1frame_{k}_ms"] = v.get("p50_ms")
    js = sub.get("js_host") or {}
    if js:
        scal["js_host_device_batch_value"] = get(js, "batch_device", "frames_per_s")
        scal["js_host_c5_value"] = get(js, "c5", "resident", "frames_per_s")
    g = sub.get("gather_n1") or {}
    for nm in ("c2", "c4"):
        if get(g, nm, "exchange_cost_frac") is not None:
            scal[f"exchange_cost_frac_{nm}"] = g[nm]["exchange_cost_frac"]
    if g.get("rccl_init_s") is not None:
        scal["rccl_init_s"] = g["rccl_init_s"]
    line.update({k: v for k, v in scal.items() if v is not None})
    if extra:
        line.update({k: v for k, v in extra.items() if v is not None})
    return line


def sub_file_path():
    """where the full record tree goes: next to bench.py — and into gpurun_out/ when that exists (pulled back)"""
    paths = [os.path.join(ROOT, "bench_sub.json")]
    g = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(g):
        paths.append(os.path.join(g, "bench_sub.json"))
    return paths


def emit(line, full, out, write_sub=True):
    """Writes the full tree to the side file(s) and prints the compact line (asserted under the cap) to `out`."""
    if write_sub:
        wrote = []
        for p in sub_file_path():
            try:
                with open(p, "w") as fh:
                    json.dump(full, fh, indent=1)
                wrote.append(os.path.relpath(p, ROOT))
            except OSError:
                pass
        line["sub_file"] = wrote[0] if wrote else None
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt.encode()) >= LINE_CAP:  # never lose the line: drop optional scalars from the end until it fits
        must = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
        for k in [k for k in reversed(list(line)) if k not in must]:
            line.pop(k)
            txt = json.dumps(line, separators=(",", ":"))
            if len(txt.encode()) < LINE_CAP:
                break
    assert len(txt.encode()) < LINE_CAP, f"bench line is {len(txt.encode())} B (cap {LINE_CAP})"
    print(txt, file=out, flush=True)
    return txt
