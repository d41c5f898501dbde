'use strict';
/*
 * ref_bench.js — TEST / MEASUREMENT INFRASTRUCTURE.  Times the UNMODIFIED reference bundle (single-threaded Node, the
 * reference's own execution model) on raw RGBA frames, for bench.py's cpu_baseline leg.
 *
 *   node oracle/ref_bench.js <frames.raw> <n> <w> <h> <seconds>                       ccv.grayscale + ccv.detect_objects
 *   node oracle/ref_bench.js <frames.raw> <n> <w> <h> <seconds> camshift <x> <y> <rw> <rh>
 *        camshift.Tracker: initTracker(frame 0, rect) once, then track() on frames 1, 2, ..., 0, 1, ... (one stream)
 *
 * The bundle is read from oracle/_ref/headtrackr_ref.js.gz, which `make -C oracle _ref` (run by __graft_entry__.build()
 * where /root/reference exists) produces from /root/reference/headtrackr.js; the directory is git-ignored, nothing of
 * the reference is committed.  Canvas = oracle/canvas_shim.js (time spent inside the shim is reported separately:
 * in a browser those calls are native code, SURVEY.md §8d).
 */
const fs = require('fs');
const path = require('path');
const zlib = require('zlib');
const Module = require('module');
const shim = require('./canvas_shim.js');

const gz = path.join(__dirname, '_ref', 'headtrackr_ref.js.gz');
global.document = shim.makeDocument();
global.window = global;
const m = new Module('headtrackr_ref', null);
m.paths = [];
m._compile(zlib.gunzipSync(fs.readFileSync(gz)).toString('utf8'), 'headtrackr_ref.js');
const headtrackr = m.exports;

const file = process.argv[2], n = +process.argv[3], w = +process.argv[4], h = +process.argv[5], budget = +process.argv[6] || 10;
const fd = fs.openSync(file, 'r');
const fbytes = w * h * 4;
function frame(i) { const b = Buffer.alloc(fbytes); fs.readSync(fd, b, 0, fbytes, (i % n) * fbytes); return new shim.Canvas(w, h).loadRGBA(b); }
function detect(c) { return headtrackr.ccv.detect_objects(headtrackr.ccv.grayscale(c), headtrackr.cascade, 5, 1); }

if (process.argv[7] === 'camshift') { /* camshift.js:198-312 */
  const rect = new headtrackr.camshift.Rectangle(+process.argv[8], +process.argv[9], +process.argv[10], +process.argv[11]);
  const frames = [];
  for (let i = 0; i < n; i++) frames.push(frame(i));
  const tr = new headtrackr.camshift.Tracker({ calcAngles: true });
  tr.initTracker(frames[0], rect);
  for (let i = 0; i < 5; i++) tr.track(frames[(i + 1) % n]); /* JIT warm-up */
  tr.initTracker(frames[0], rect);
  const times = [];
  const t0 = process.hrtime.bigint();
  let calls = 0;
  while (calls < 8 || Number(process.hrtime.bigint() - t0) / 1e9 < budget) {
    const a = process.hrtime.bigint();
    tr.track(frames[(calls + 1) % n]);
    times.push(Number(process.hrtime.bigint() - a) / 1e6);
    calls++;
  }
  const total = times.reduce(function (s, v) { return s + v; }, 0);
  times.sort(function (a, b) { return a - b; });
  const to = tr.getTrackObj();
  console.log(JSON.stringify({ calls: calls, fps: calls / (total / 1e3), ms_median: times[times.length >> 1], ms_min: times[0],
    last: [to.x, to.y, to.width, to.height], node: process.version, cpus: require('os').cpus().length, cpu_model: require('os').cpus()[0].model }));
  process.exit(0);
}

for (let i = 0; i < Math.min(3, n); i++) detect(frame(i)); /* JIT warm-up */
shim.stats.enabled = true; shim.stats.shimNs = 0n;
const times = [];
let faces = 0;
const t0 = process.hrtime.bigint();
let i = 0;
while (i < n && (i < 4 || Number(process.hrtime.bigint() - t0) / 1e9 < budget)) {
  const c = frame(i);
  const a = process.hrtime.bigint();
  faces += detect(c).length;
  times.push(Number(process.hrtime.bigint() - a) / 1e6);
  i++;
}
const total = times.reduce(function (s, v) { return s + v; }, 0);
times.sort(function (a, b) { return a - b; });
console.log(JSON.stringify({ frames: i, fps: i / (total / 1e3), ms_median: times[times.length >> 1], ms_min: times[0],
  shim_fraction: Number(shim.stats.shimNs) / 1e6 / total, faces: faces, node: process.version, cpus: require('os').cpus().length,
  cpu_model: require('os').cpus()[0].model }));
