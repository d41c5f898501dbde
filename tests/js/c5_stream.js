'use strict';
/* The C5 loop from the JavaScript host (north_star: "Host code stays JavaScript (Node)"; /root/reference/src/main.js:168-180 ->
 * facetrackr.js:97-108,185-217 for K frame-synchronous 1920x1080 feeds): ccv.DeviceBatch.detectStep / trackStep.
 *     node tests/js/c5_stream.js parity <uniq.raw> <nuniq> <feeds> <steps> <out.json>    every step's results -> out.json (checked against
 *                                                                                        the oracle by tests/test_gpu_c5.py)
 *     node tests/js/c5_stream.js bench  <uniq.raw> <nuniq> <feeds> <seconds>             -> one JSON line: frames/s resident and with pinned
 *                                                                                        (hostAlloc) ingest every step, latency percentiles
 * uniq.raw: the camera's nuniq distinct frames (headtrackr_amd/synth.py stream_feed_frames); feed f at step k shows frame
 * (k % nuniq + 7 f) % nuniq — bench.py's C5 input. */
const fs = require('fs');
const path = require('path');
const root = path.join(__dirname, '..', '..');
const headtrackr = require(path.join(root, 'headtrackr_amd', 'js', 'headtrackr.js'));

const mode = process.argv[2], file = process.argv[3], nuniq = +process.argv[4], K = +process.argv[5];
const W = 1920, H = 1080, fbytes = W * H * 4, sbytes = K * fbytes;
const now = function () { return Number(process.hrtime.bigint()) / 1e6; };
const pct = function (v, q) { if (!v.length) return null; const s = v.slice().sort(function (a, b) { return a - b; }); return +s[Math.min(s.length - 1, Math.floor(q / 100 * s.length))].toFixed(4); };
const frameIndex = function (k, f) { return (k % nuniq + 7 * f) % nuniq; };

const fd = fs.openSync(file, 'r');
const b = new headtrackr.ccv.DeviceBatch(W, H, K, { depth: 1, sets: nuniq });
/* pinned ring of every step's batch (bench: the ingest source; parity: the staging buffer of the device upload) */
const pinned = headtrackr.hostAlloc(mode === 'bench' ? nuniq * sbytes : sbytes);
for (let k = 0; k < nuniq; k++) {
  const view = mode === 'bench' ? pinned.subarray(k * sbytes, (k + 1) * sbytes) : pinned;
  for (let f = 0; f < K; f++) fs.readSync(fd, view, f * fbytes, fbytes, frameIndex(k, f) * fbytes);
  b.upload(view, k);
}
fs.closeSync(fd);

const isDetect = function (i) { return i % 30 === 0; };
const step = function (i, set) { return isDetect(i) ? b.detectStep(set) : b.trackStep(set, true); };
/* the same steps with TWO track steps outstanding (bench.py's resident loop): a track step is enqueued before the previous one is collected;
 * a detect step is enqueued right behind them, then the pipeline is drained, then its best faces come back to JS (which floors them and calls initTracker).  onResult(i, r) in step order. */
const pipelined = function (from, to, setOf, onResult) {
  const pend = [];
  const drain1 = function () { const j = pend.shift(); onResult(j, b.trackCollect()); };
  for (let i = from; i < to; i++) {
    if (isDetect(i)) { b.detectStepEnqueue(setOf(i)); while (pend.length) drain1(); onResult(i, b.detectStepFinish()); }
    else { b.trackEnqueue(setOf(i), true); pend.push(i); if (pend.length > 1) drain1(); }
  }
  while (pend.length) drain1();
};

if (mode === 'parity') {
  const steps = +process.argv[6], outFile = process.argv[7];
  const res = [];
  pipelined(0, steps, function (i) { return i % nuniq; }, function (i, r) {
    res.push(isDetect(i) ? { step: i, best: Array.from(r.best), rects: Array.from(r.rects) } : { step: i, track: Array.from(r) });
  });
  res.sort(function (x, y) { return x.step - y.step; });
  fs.writeFileSync(outFile, JSON.stringify({ steps: res, graph_launches: b.graphLaunches(), feeds: K }));
  b.destroy();
  headtrackr.hostFree(pinned);
  process.stdout.write(JSON.stringify({ ok: true, steps: steps }) + '\n', function () { headtrackr.exitNow(0); });
} else {
  const seconds = +process.argv[6] || 2;
  const out = { feeds: K, width: W, height: H, node: process.version };
  for (let i = 0; i < 31; i++) step(i, i % nuniq); /* warm-up: one cycle + the second cycle's detect (captured graph) */
  { /* frames resident in HBM (the bench contract's definition) */
    let i = 0;
    const t0 = now();
    while (i < 60 || now() - t0 < seconds * 400) { pipelined(i, i + 30, function (j) { return j % nuniq; }, function () {}); i += 30; }
    const dt = (now() - t0) / 1e3;
    out.resident = { frames_per_s: +(i * K / dt).toFixed(1), ms_per_step: +(dt / i * 1e3).toFixed(4), steps: i, track_steps_outstanding: 2 };
  }
  { /* every step's frames host -> GPU from pinned memory, double-buffered: step i+1 crosses PCIe while step i is processed */
    let i = 0;
    b.ingest(pinned.subarray(0, sbytes)); b.swap();
    const t0 = now();
    while (i < 60 || now() - t0 < seconds * 400) {
      const nx = (i + 1) % nuniq;
      b.ingest(pinned.subarray(nx * sbytes, (nx + 1) * sbytes));
      step(i, -1);
      b.swap();
      i++;
    }
    const dt = (now() - t0) / 1e3;
    out.pcie_inclusive = { frames_per_s: +(i * K / dt).toFixed(1), ms_per_step: +(dt / i * 1e3).toFixed(4), steps: i, h2d_gbs: +(i * sbytes / dt / 1e9).toFixed(2) };
  }
  { /* latency of one time step strictly in turn incl. PCIe: ingest, swap, process, results in JS */
    const det = [], trk = [];
    for (let i = 0; i < 93; i++) {
      const a = now();
      b.ingest(pinned.subarray((i % nuniq) * sbytes, (i % nuniq + 1) * sbytes)); b.swap();
      step(i, -1);
      (isDetect(i) ? det : trk).push(now() - a);
    }
    out.latency_ms = { detect_p50: pct(det, 50), track_p50: pct(trk, 50), track_p99: pct(trk, 99), samples: det.length + trk.length };
  }
  out.detect_graph_replays = b.graphLaunches();
  out.what = 'ccv.DeviceBatch.detectStep / trackStep from Node: ' + K + ' frame-synchronous 1080p feeds as one batch per step, detect + initTracker every 30th step, camshift.track otherwise';
  b.destroy();
  headtrackr.hostFree(pinned);
  process.stdout.write(JSON.stringify(out) + '\n', function () { headtrackr.exitNow(0); });
}
