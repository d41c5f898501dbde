"""Shared pieces of bench.py: constants, the rank environment, timed rounds, the dominant-kernel roofline."""
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
# VALU issue peak: 256 CUs x 4 SIMDs, one wave64 VALU instruction per SIMD per 4 cycles, 2.4 GHz (MI355X_MICROARCH.md)
VALU_PEAK_WAVE_INSTS_PER_S = 1024 * 2.4e9 / 4
DEFAULT_STEPS = {"c2": 1000, "c4": 300, "c3": 20, "c5": 300}
SUB_STEPS = {"c4": 100, "c4_strong": 24, "c3": 12, "c5": 90}
GEOM = {"c2": (320, 240, 256), "c3": (320, 240, 256), "c4": (1280, 720, 128)}
WORKLOAD_TEXT = {
    "c2": "C2: 256 x 320x240 RGBA frames per GPU, full BBF cascade detect (interval 5) incl. grouping + best face "
          "per frame on the host",
    "c3": "C3: 256 streams of 320x240 per GPU (one moving face each): detect once, initTracker, then 60 camshift "
          "track() calls; every processed frame counts",
    "c4": "C4: 1280x720 frames, full cascade detect incl. grouping + best face per frame, all-gather of best-face "
          "rects for N > 1",
}
TRAFFIC_SOURCE = "profiles/traffic.json (rocprofv3 PMC pass of the same build — code-object fingerprint checked —, not this run)"


class Env:
    """one rank: torch, its process group, and the barrier + synchronize fence of the timing contract"""

    def __init__(self, torch, dist, rank, world, local, stub=False):
        self.torch, self.dist, self.rank, self.world, self.local, self.stub = torch, dist, rank, world, local, stub
        self.dev = "cpu" if stub else "cuda"

    def fence(self):
        if self.world > 1:
            self.dist.barrier()
        if not self.stub:
            self.torch.cuda.synchronize()

    def max_over_ranks(self, dt):
        if self.world > 1:
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def gather_scalar(self, v):
        """every rank's value of a scalar, on every rank (outside timed regions)"""
        if self.world == 1:
            return [float(v)]
        t = self.torch.tensor([float(v)], dtype=self.torch.float64, device=self.dev)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def timed_rounds(self, run_block, steps, rounds=0, target_s=0.4):
        """R back-to-back timed blocks of `steps` steps each, every block bracketed by barrier + synchronize on both
        sides and max-reduced over the ranks.  rounds == 0: R from the first block so that the timed work is ~target_s
        (3..25; every rank derives the same R from the same max-reduced time).  Returns the block times in seconds
        (max over ranks); this rank's own block times stay in self.last_own."""
        dts, own = [], []
        r = 0
        while True:
            self.fence()
            t0 = time.perf_counter()
            run_block(steps)
            self.fence()
            mine = time.perf_counter() - t0
            own.append(mine)
            dts.append(self.max_over_ranks(mine))
            r += 1
            if rounds <= 0:
                rounds = int(min(25, max(3, -(-target_s // max(dts[0], 1e-6)))))
            if r >= rounds:
                self.last_own = own
                return dts


def round_stats(dts, steps):
    """median block -> the reported time; min / max -> the spread"""
    med = float(np.median(dts))
    return med, dict(rounds=len(dts), ms_per_step=round(med / steps * 1e3, 4),
                     ms_per_step_min=round(min(dts) / steps * 1e3, 4), ms_per_step_max=round(max(dts) / steps * 1e3, 4))


def dominant_roofline(per_step_ms, launches_per_step, bytes_per_step, extra=None):
    """SURVEY.md §8(d) roofline of the DOMINANT kernel = the one with the largest device time per step (sum of its
    launches): achieved = algorithmic bytes of a step / that kernel's time per step (for a kernel with one launch per
    step this is bytes per launch / average launch duration)."""
    dom = max(per_step_ms, key=per_step_ms.get)
    ach = bytes_per_step / (per_step_ms[dom] * 1e-3) / 1e9
    # kernels within 5 % of the dominant one's time per step are named with it (C2: resample's launches and scan_tiles
    # trade places from run to run)
    co = {k: dict(kernel_ms_per_step=round(v, 5), frac=round(bytes_per_step / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 5))
          for k, v in per_step_ms.items() if k != dom and v >= 0.95 * per_step_ms[dom]}
    r = dict(bound="hbm", kernel=dom, co_dominant=co, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
             frac=round(ach / HBM_PEAK_GBS, 5), traffic=None, kernel_ms_per_step=round(per_step_ms[dom], 5),
             launches_per_step=round(launches_per_step[dom], 2),
             avg_launch_ms=round(per_step_ms[dom] / max(launches_per_step[dom], 1e-9), 5),
             dominant="largest device time per step (sum of its launches)")
    if extra:
        r.update(extra)
    return r


def device_copy_ceiling(torch):
    """what a kernel that only reads and writes HBM reaches on this box, measured in the same run."""
    buf = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(buf)
    dst.copy_(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(8):
        dst.copy_(buf)
    e1.record()
    torch.cuda.synchronize()
    gbs = 2.0 * buf.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del buf, dst
    return gbs


def load_pmc(workload, lib=None, path=None):
    """profiles/traffic.json — the committed rocprofv3 counter passes (tools/collect_profiles.py), per detect STEP:
    `per_step` = HBM bytes per bench timer name (every launch's own FETCH_SIZE / WRITE_SIZE summed; `resample` = the
    pyramid generation launches + the tail kernel), `valu_per_step` = SQ_INSTS_VALU wave instructions of all kernels.
    The file records the fingerprint of the code objects it was measured on (`_build`, benchlib/fingerprint.py); the
    third return value lists the translation units of `workload` whose code in the library being timed NOW differs from
    that build ([] = the counters belong to this build; None = the file carries no fingerprint: treated as stale)."""
    from .fingerprint import stale_units

    try:
        with open(path or os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            doc = json.load(fh)
        j = doc.get(workload, {})
        return j.get("per_step", {}), j.get("valu_per_step"), stale_units(doc.get("_build"), workload, lib)
    except Exception:
        return {}, None, None


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port
