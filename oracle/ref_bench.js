'use strict';
/*
 * ref_bench.js — TEST / MEASUREMENT INFRASTRUCTURE.  Times the UNMODIFIED reference bundle (single-threaded Node, the
 * reference's own execution model) on raw RGBA frames, for bench.py's cpu_baseline leg.
 *
 *   node oracle/ref_bench.js <frames.raw> <n> <w> <h> <seconds>
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
