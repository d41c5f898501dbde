"""bench.py's one JSON line must stay a compact headline: round 4's 21 KB line could not be parsed by the driver
(BENCH_r04.json: parsed = null).  A synthetic full-size record tree — every sub-record present, long notes and
samples — must compose to < 4 KB and keep the contract's keys; the side file receives the whole tree."""
import io
import json

from benchlib import line as bl

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _roof(kernel):
    return dict(bound="hbm", kernel=kernel, co_dominant={"scan_tiles": dict(kernel_ms_per_step=0.12345, frac=0.30612, traffic=113255456)},
                achieved=2345.67, peak=8000.0, unit="GB/s", frac=0.29321, traffic=197301906, kernel_ms_per_step=0.12912, launches_per_step=4.0,
                avg_launch_ms=0.03228, dominant="largest device time per step (sum of its launches)", algorithmic_bytes_per_frame=1183190,
                frames_per_step=256, device_copy_gbs=5301.2, frac_of_device_copy=0.44, traffic_source="x" * 80)


def _cpu():
    return dict(value=40.712, unit="frames/s", cores=1, kind="reference", sample="first 64 of the 256 frames " + "s" * 400, host_cpus=256,
                cpu_model="AMD EPYC 9575F 64-Core Processor")


def _detect(value, ms):
    return {"value": value, "unit": "frames/s", "steps": 20, "warmup": 5, "rounds": 25, "ms_per_step": ms, "ms_per_step_min": ms * 0.98,
            "ms_per_step_max": ms * 1.02, "scaling": "weak",
            "config": {"workload": "C2: 256 x 320x240 RGBA frames per GPU, " + "w" * 200, "frames_per_gpu": 256, "frames_total": 256,
                       "batches_in_flight": 3, "width": 320, "height": 240, "unique_frames": 256, "frame_mix": "m" * 60, "parallelism": "p" * 200},
            "roofline": _roof("resample"), "cpu_baseline": _cpu(),
            "cpu_baseline_port": dict(value=321.5, unit="frames/s", cores=1, kind="port", sample="p" * 200,
                                      all_cores=dict(value=41234.5, unit="frames/s", cores=256, kind="port", sample="q" * 200)),
            "vs_cpu": 27400.1, "path_hbm_frac": 0.11712, "wall_hbm_frac": 0.16512,
            "valu_issue": dict(frac_wall=0.6712, frac_device=0.4812), "device_ms_per_step": 0.31812, "rank_ms_per_step_min": ms, "rank_ms_per_step_max": ms,
            "depth1": dict(ms_per_step=0.3412, value=750000.1), "pcie_inclusive": dict(value=181234.5, h2d_gbs=55.61), "allgather_verified": True,
            "kernel_ms_per_step": {k: 0.1 for k in ("gray", "resample", "scan_tiles", "scan_deep")}, "kernel_rooflines": {"gray": {"x": 1}},
            "stage_in": list(range(17)), "hits_per_step": 1392, "faces_per_step": 85}


def _tree():
    prim = _detect(1116234.56, 0.2294)
    c5 = {"value": 120034.5, "config": {"feeds_per_gpu": 8, "workload": "C5"}, "one_feed": {"value": 23412.1}, "pcie_inclusive": {"value": 6655.2},
          "latency_ms": {"p50": 1.2612, "p99": 1.4512}, "device_ms": {"track_step": 0.0631}, "vs_cpu": 3685.2, "parity_exact": "232/232",
          "parity_detect_exact": "16/16", "parity_note": "n" * 400, "roofline": _roof("resample"), "cpu_baseline": _cpu()}
    sub = {"c4_1gpu": _detect(115123.4, 1.112), "c4_strong": _detect(108812.3, 9.41), "c2_large": _detect(1175123.0, 0.871),
           "c3": {"value": 6881234.5, "roofline": _roof("cs_track"), "parity_exact": "480/480", "cpu_baseline": _cpu()}, "c5": c5,
           "latency_1frame": {"320x240": {"p50_ms": 0.0912}, "1280x720": {"p50_ms": 0.1112}, "1920x1080": {"p50_ms": 0.1312}},
           "js_host": {"batch_device": {"frames_per_s": 1048123.4}, "c5": {"resident": {"frames_per_s": 121400.2}}, "tracker": {"what": "t" * 300}},
           "gather_n1": {"rccl_init_s": 1.23, "c2": {"exchange_cost_frac": 0.0201}, "c4": {"exchange_cost_frac": 0.0012}}}
    return prim, sub


def test_full_size_line_is_a_compact_headline(tmp_path, monkeypatch):
    prim, sub = _tree()
    line = bl.compose("frames/sec full-cascade detect at 320x240", prim, sub, 1, dict(rccl_init_s=None, device_copy_gbs=5301.2, bench_wall_s=48.2))
    monkeypatch.setattr(bl, "sub_file_path", lambda: [str(tmp_path / "bench_sub.json")])
    monkeypatch.setattr(bl, "ROOT", str(tmp_path))
    out = io.StringIO()
    txt = bl.emit(line, dict(primary=prim, sub=sub), out)
    assert out.getvalue().strip() == txt and "\n" not in txt
    assert len(txt.encode()) < bl.LINE_CAP, len(txt)
    assert len(txt.encode()) < 3000, len(txt)  # what a full default run actually produces stays well below the cap
    got = json.loads(txt)
    for k in CONTRACT:
        assert k in got, k
    assert got["roofline"]["frac"] == 0.29321 and got["roofline"]["kernel"] == "resample" and got["roofline"]["traffic_source"]
    assert got["cpu_baseline"]["kind"] == "reference" and len(got["cpu_baseline"]["sample"]) <= bl.SAMPLE_CAP
    assert got["value_720p"] == 115123.4 and got["north_star_720p_vs_reference_js"] == 27400.1 and got["c3_value"] == 6881234.5
    assert got["c5_value"] == 120034.5 and got["valu_issue_frac"] == 0.6712 and got["depth1_ms_per_step"] == 0.3412
    assert got["cpu_port_1core_value"] == 321.5 and got["cpu_port_all_cores_value"] == 41234.5 and got["cpu_port_all_cores"] == 256
    assert got["cpu_port_all_cores_value_720p"] == 41234.5
    assert got["parity_exact"].startswith("c3 480/480; c5 232/232 + best faces 16/16")
    assert "sub" not in got and got["sub_file"] == "bench_sub.json"
    side = json.load(open(tmp_path / "bench_sub.json"))
    assert side["sub"]["c5"]["parity_note"] == "n" * 400  # nothing is lost: the tree is in the side file


def test_oversized_optional_scalars_are_dropped_not_the_line():
    prim, sub = _tree()
    line = bl.compose("m", prim, sub, 8, {f"extra_{i}": "x" * 100 for i in range(60)})
    out = io.StringIO()
    txt = bl.emit(line, {}, out, write_sub=False)
    got = json.loads(txt)
    assert len(txt.encode()) < bl.LINE_CAP
    for k in CONTRACT:
        assert k in got, k
    assert got["n_gpus"] == 8 and got["cpu_baseline_note"] == "N = 1 only"


def test_pmc_constants_are_tied_to_the_build(tmp_path):
    """profiles/traffic.json's counters are constants of the build they were measured on (round-5 verdict: nothing tied them to the library
    being timed).  tools/gpu_pmc.sh records the code-object fingerprint of the profiled library, collect_profiles.py stores it as `_build`,
    and load_pmc compares it with the library bench.py loads: same build -> fresh; another pyramid code object -> ['pyramid']; a file
    without a fingerprint -> stale (None).  A stale record flags the line and drops valu_issue_frac."""
    import os

    from benchlib import common, fingerprint

    lib = fingerprint.default_lib()
    if not os.path.exists(lib):
        import pytest

        pytest.skip("library not built")
    now = fingerprint.code_objects(lib)
    assert set(now) == {"pyramid", "scan", "camshift"} and all(len(v) == 16 for v in now.values())
    doc = {"c2": {"per_step": {"gray": 1, "resample": 2}, "valu_per_step": 3}, "c3": {"per_step": {"cs_track": 5}, "valu_per_step": 7}}

    def load(build, wl="c2"):
        p = tmp_path / "traffic.json"
        p.write_text(json.dumps(dict(doc, **({"_build": build} if build is not None else {}))))
        return common.load_pmc(wl, lib=lib, path=str(p))

    per, valu, stale = load(now)
    assert per == {"gray": 1, "resample": 2} and valu == 3 and stale == []
    assert load(dict(now, pyramid="0" * 16))[2] == ["pyramid"]
    assert load(dict(now, camshift="0" * 16))[2] == []  # C2's counters do not depend on the camshift code object ...
    assert load(dict(now, camshift="0" * 16), "c3")[2] == ["camshift"]  # ... C3's do
    assert load(None)[2] is None
    # the committed file: whatever its state, load_pmc answers without raising and names the stale units
    per, valu, stale = common.load_pmc("c2")
    assert stale is None or isinstance(stale, list)
    # a stale record in the line: flagged, valu_issue_frac absent
    prim, sub = _tree()
    prim["roofline"]["traffic_stale"] = True
    prim["traffic_stale"] = True
    prim.pop("valu_issue")
    line = bl.compose("m", prim, sub, 1, {})
    assert line["traffic_stale"] is True and line["roofline"]["traffic_stale"] is True and "stale" in line["roofline"]["traffic_source"]
    assert "valu_issue_frac" not in line and "valu_issue_frac_720p" in line and "traffic_stale_720p" not in line
