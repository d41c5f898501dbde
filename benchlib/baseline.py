"""CPU baselines (rank 0, N = 1 only): bounded samples of the same frames on the box's host cores.
 * "reference": the UNMODIFIED reference JS, single-threaded Node (its own execution model), from oracle/_ref
 * "port":      the plain-C oracle restatement (oracle/ht_oracle.c), 1 thread
This module and the parity checks after the timed regions are the only places bench.py touches oracle/."""
import json
import os
import shutil
import subprocess
import tempfile
import time

import numpy as np

from .common import ROOT

REF_GZ = os.path.join(ROOT, "oracle", "_ref", "headtrackr_ref.js.gz")
REF_BENCH = os.path.join(ROOT, "oracle", "ref_bench.js")


def _node_ref(raw, args, seconds):
    """oracle/ref_bench.js on `raw` (frames written to a temp file): the JSON object it prints, or raises"""
    node = shutil.which("node")
    if not node or not os.path.exists(REF_GZ):
        raise RuntimeError("node or oracle/_ref missing")
    with tempfile.NamedTemporaryFile(suffix=".raw") as tf:
        np.ascontiguousarray(raw).tofile(tf.name)
        r = subprocess.run([node, REF_BENCH, tf.name] + [str(x) for x in args], capture_output=True, text=True,
                           timeout=seconds * 6 + 120)
    return json.loads(r.stdout.strip().splitlines()[-1])


def port_all_cores(frames, blob, seconds):
    """SURVEY.md §8(d)'s optional second line: the C port on every host core this process may use — one thread per core
    (ctypes drops the GIL inside ho_detect_raw, the port keeps no global state), frames dealt out round-robin, each
    thread stops taking frames once the budget has run out.  The reference itself is single-threaded JS; this is what
    its arithmetic would deliver if somebody parallelised it over frames."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import ht_oracle as ho

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    nf = len(frames)
    t0 = time.perf_counter()

    def worker(tid):
        done, i = 0, tid
        while done < 1 or time.perf_counter() - t0 < seconds:
            ho.detect_raw(frames[i % nf], blob)
            done, i = done + 1, i + cores
        return done

    try:
        with ThreadPoolExecutor(max_workers=cores) as ex:
            done = sum(ex.map(worker, range(cores)))
        dt = time.perf_counter() - t0
        return dict(value=round(done / dt, 2), unit="frames/s", cores=cores, kind="port",
                    sample=f"{done} detect calls on the first {min(nf, done)} frames in {dt:.1f} s, one thread per core")
    except Exception as e:  # a baseline leg never costs the line
        return dict(error=f"{type(e).__name__}: {e}")


def cpu_detect_baseline(frames, W, H, blob, seconds):
    """(reference-JS record or the port standing in, port record) for ccv.grayscale + ccv.detect_objects(..., 5, 1)"""
    from oracle import ht_oracle as ho

    nf = len(frames)
    ho.detect_raw(frames[0], blob)  # warm
    t0 = time.perf_counter()
    done = 0
    while done < nf and (done < 2 or time.perf_counter() - t0 < seconds / 2):
        ho.detect_raw(frames[done], blob)
        done += 1
    cdt = time.perf_counter() - t0
    port = dict(value=round(done / cdt, 3), unit="frames/s", cores=1, kind="port", host_cpus=os.cpu_count(),
                sample=f"first {done} of the {nf} {W}x{H} frames of this workload, oracle/ht_oracle.c detect "
                       "(gray+pyramid+scan), 1 thread")
    port["all_cores"] = port_all_cores(frames, blob, seconds / 3)
    try:
        ns = min(nf, 64)
        j = _node_ref(frames[:ns], [ns, W, H, seconds], seconds)
        cpu = dict(value=round(j["fps"], 3), unit="frames/s", cores=1, kind="reference", host_cpus=j["cpus"],
                   cpu_model=j["cpu_model"],
                   sample=f"first {j['frames']} of the {nf} {W}x{H} frames of this workload: unmodified reference JS "
                          f"(ccv.grayscale + ccv.detect_objects(..., 5, 1)) on oracle/canvas_shim.js, {j['node']} single "
                          f"thread, median {j['ms_median']:.1f} ms/frame, {100 * j['shim_fraction']:.0f}% of it inside "
                          "the canvas shim")
    except Exception as e:  # the port baseline stands in
        cpu = dict(port, note=f"reference JS baseline unavailable: {e}")
    return cpu, port


def cpu_camshift_baseline(versions, rect, W, H, seconds):
    """camshift.Tracker.initTracker + track() (camshift.js:198-312) of ONE stream on its moving frames: the unmodified
    reference JS (kind "reference") or, without Node / the bundle, the C port."""
    nv = len(versions)
    try:
        j = _node_ref(versions, [nv, W, H, seconds, "camshift"] + [int(v) for v in rect], seconds)
        return dict(value=round(j["fps"], 3), unit="track() calls/s", cores=1, kind="reference", host_cpus=j["cpus"],
                    cpu_model=j["cpu_model"],
                    sample=f"{j['calls']} camshift.Tracker.track() calls of one {W}x{H} stream (initTracker on rect "
                           f"{list(map(int, rect))}, its {nv} moving frames in turn): unmodified reference JS on "
                           f"oracle/canvas_shim.js, {j['node']} single thread, median {j['ms_median']:.2f} ms/call")
    except Exception as e:
        note = f"reference JS baseline unavailable: {e}"
    from oracle import ht_oracle as ho

    st = ho.cs_init(versions[0], *[int(v) for v in rect], calc_angles=True)
    t0 = time.perf_counter()
    calls = 0
    while calls < 8 or time.perf_counter() - t0 < seconds / 2:
        ho.cs_track(st, versions[(calls + 1) % nv])
        calls += 1
    return dict(value=round(calls / (time.perf_counter() - t0), 3), unit="track() calls/s", cores=1, kind="port",
                sample=f"{calls} track() calls of one {W}x{H} stream, oracle/ht_oracle.c, 1 thread",
                host_cpus=os.cpu_count(), note=note)
