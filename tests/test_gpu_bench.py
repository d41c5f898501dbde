"""bench.py itself on the GPU: the launcher path (torch.distributed.run -> RANK / WORLD_SIZE -> init_process_group("nccl") ->
all-gather of the best-face records through RCCL -> verification) with ONE rank on one GPU — everything of the N > 1 path that
one GPU can execute — and the shape of the one JSON line (compact, the contract's keys, the side file)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _run(args, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "HT_BENCH_STUB"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # stdout carries the ONE line and nothing else
    return lines[0], json.loads(lines[0])


def test_launcher_path_with_rccl_on_one_gpu():
    """`bench.py --gpus 1 --force-launcher --workload c4`: torchrun starts the rank, the rank builds its RCCL group, every timed
    step all-gathers its 128 best-face records through ncclAllGather, rank 0 verifies the gathered table — the N > 1 code path,
    world size 1."""
    txt, line = _run(["--gpus", "1", "--force-launcher", "--workload", "c4", "--no-sub", "--steps", "10", "--warmup", "3",
                      "--cpu-seconds", "0", "--rounds", "3"])
    assert line["n_gpus"] == 1 and line["ranks"] == 1 and line["launched_by_bench"] is True
    assert line["allgather_verified"] is True and line["rccl_init_s"] > 0
    assert line["config"]["frames_per_gpu"] == 128 and line["config"]["width"] == 1280
    assert line["rank_ms_per_step_min"] <= line["rank_ms_per_step_max"]
    assert 20_000 < line["value"] < 1_000_000 and line["roofline"]["frac"] > 0.05
    assert len(txt) < 4000


def test_default_line_is_compact_and_complete(tmp_path):
    """the driver's command shape (`--gpus 1 --steps K --warmup W`, all sub-records) with a short CPU budget: the line stays under
    4 KB with every contract key, roofline and cpu_baseline of the headline, the scalars of the sub-records, parity of its own run;
    the full tree is in bench_sub.json."""
    txt, line = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "1.5"], timeout=900)
    assert len(txt.encode()) < 4000, len(txt)
    for k in CONTRACT:
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["dtype"] == "u8"
    r, c = line["roofline"], line["cpu_baseline"]
    assert r["bound"] == "hbm" and 0.05 < r["frac"] < 1.0 and r["peak"] == 8000.0 and r["avg_launch_ms"] > 0
    # the PMC constants are tied to the build they were measured on (benchlib/fingerprint.py): a library built from other kernel sources
    # than the committed counter pass makes the line say so and drop the figure derived from them
    stale = bool(line.get("traffic_stale"))
    assert r["traffic"] and ("stale" in r["traffic_source"] if stale else "not this run" in r["traffic_source"])
    assert bool(r.get("traffic_stale")) == stale and (("valu_issue_frac" in line) != stale)
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and len(c["sample"]) <= 200
    for k in ("value_720p", "ms_per_step_720p", "north_star_720p_vs_reference_js", "c3_value", "c5_value", "path_hbm_frac",
              "wall_hbm_frac", "depth1_ms_per_step", "depth1_ms_per_step_720p", "pcie_inclusive_value",
              "pcie_inclusive_value_720p", "latency_1frame_320x240_ms", "latency_1frame_1280x720_ms", "exchange_cost_frac_c2",
              "rccl_init_s", "parity_exact", "bench_wall_s", "sub_file"):
        assert k in line, k
    assert line["north_star_720p_vs_reference_js"] >= 30  # the north star's own target at 1 GPU
    assert "c3 " in line["parity_exact"] and "c5 " in line["parity_exact"]
    for part in line["parity_exact"].split(" vs ")[0].replace(" + best faces", ";").split(";"):
        a, b = part.strip().split()[-1].split("/")
        assert a == b, line["parity_exact"]  # every call of the run's own parity pass exact
    side = json.load(open(os.path.join(ROOT, line["sub_file"])))
    assert side["sub"]["c4_1gpu"]["cpu_baseline"]["kind"] in ("reference", "port")
    assert side["sub"]["c4_1gpu"]["pcie_inclusive"]["h2d_gbs"] > 5
    assert side["sub"]["gather_n1"]["c2"]["allgather_verified"] is True
    assert side["primary"]["depth1"]["ms_per_step"] >= side["primary"]["ms_per_step"] * 0.9
