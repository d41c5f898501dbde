'use strict';
/* GPU parity of the JavaScript host (headtrackr_amd/js/headtrackr.js -> N-API -> C ABI -> HIP) against the golden
 * vectors recorded from the unmodified reference JS.  Driven by tests/test_gpu_js.py, which writes the synthetic frames
 * as raw RGBA files:   node tests/js/parity_gpu.js <job.json>   -> one JSON line {ok, checked, errors}.
 * The test bodies read like calls into the reference: ccv.grayscale + ccv.detect_objects on a canvas,
 * new camshift.Tracker().initTracker/track, new facetrackr.Tracker().init/track/getTrackingObject. */
const fs = require('fs');
const path = require('path');
const root = path.join(__dirname, '..', '..');
const headtrackr = require(path.join(root, 'headtrackr_amd', 'js', 'headtrackr.js'));
const { Canvas } = require(path.join(root, 'headtrackr_amd', 'js', 'canvas.js'));

const job = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const base = path.dirname(path.resolve(process.argv[2]));
const out = { ok: true, checked: 0, errors: [] };
function check(cond, msg) { out.checked++; if (!cond) { out.ok = false; if (out.errors.length < 20) out.errors.push(msg); } }
function canvasOf(file, w, h) { return new Canvas(w, h).setFrame(fs.readFileSync(path.resolve(base, file))); }

const CRC = (function () { const t = new Int32Array(256); for (let n = 0; n < 256; n++) { let c = n; for (let k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320 ^ (c >>> 1)) : (c >>> 1); t[n] = c; } return t; })();
function crc32(buf) { let c = -1; for (let i = 0; i < buf.length; i++) c = CRC[(c ^ buf[i]) & 0xFF] ^ (c >>> 8); return (c ^ -1) >>> 0; }

/* document shim for facetrackingEvent (the reference dispatches DOM events, facetrackr.js:112-125) */
const listeners = {};
global.document = {
  createEvent: function () { return { initEvent: function (t) { this.type = t; } }; },
  dispatchEvent: function (e) { (listeners[e.type] || []).forEach(function (f) { f(e); }); },
  addEventListener: function (t, f) { (listeners[t] = listeners[t] || []).push(f); }
};

job.detect.forEach(function (cs) {
  const g = cs.golden;
  const canvas = canvasOf(cs.frame, cs.w, cs.h);
  check(headtrackr.getWhitebalance(canvas) === g.whitebalance, cs.name + ': whitebalance');
  const gray = headtrackr.ccv.grayscale(canvas);
  check(gray === canvas, cs.name + ': grayscale returns its argument');
  check(crc32(canvas.pixels) === g.gray_rgba_crc, cs.name + ': gray RGBA bytes');
  const raw = headtrackr.ccv.detect_objects(canvas, headtrackr.cascade, cs.interval, 0);
  check(raw.length === g.raw.length, cs.name + ': raw count ' + raw.length + ' vs ' + g.raw.length);
  for (let i = 0; i < Math.min(raw.length, g.raw.length); i++) {
    ['x', 'y', 'width', 'height', 'confidence', 'neighbor'].forEach(function (k) { check(raw[i][k] === g.raw[i][k], cs.name + ': raw[' + i + '].' + k); });
  }
  const grouped = headtrackr.ccv.detect_objects(canvas, headtrackr.cascade, cs.interval, g.min_neighbors);
  check(grouped.length === g.grouped.length, cs.name + ': grouped count');
  for (let i = 0; i < Math.min(grouped.length, g.grouped.length); i++) {
    ['x', 'y', 'width', 'height', 'confidence', 'neighbors'].forEach(function (k) { check(grouped[i][k] === g.grouped[i][k], cs.name + ': grouped[' + i + '].' + k); });
  }
});

function near(a, b, tol) { return Math.abs(a - b) <= tol; }
job.camshift.forEach(function (cs) {
  const g = cs.golden;
  const tracker = new headtrackr.camshift.Tracker({ calcAngles: g.calcAngles });
  tracker.initTracker(canvasOf(cs.frames[0], cs.w, cs.h), new headtrackr.camshift.Rectangle(g.rect[0], g.rect[1], g.rect[2], g.rect[3]));
  g.calls.forEach(function (call, i) {
    tracker.track(canvasOf(cs.frames[call.frame], cs.w, cs.h));
    const sw = tracker.getSearchWindow(), to = tracker.getTrackObj();
    check(near(sw.x, call.sw[0], 1) && near(sw.y, call.sw[1], 1) && near(sw.width, call.sw[2], 5) && near(sw.height, call.sw[3], 5), cs.name + ' call ' + i + ': search window');
    check(near(to.x, call.x, 1) && near(to.y, call.y, 1), cs.name + ' call ' + i + ': centre');
    check(near(to.width, call.width, 4) && near(to.height, call.height, 4), cs.name + ' call ' + i + ': size');
    if (call.angle === null) check(Number.isNaN(to.angle), cs.name + ' call ' + i + ': NaN angle');
    else { let d = Math.abs(to.angle - call.angle); d = Math.min(d, Math.abs(d - Math.PI)); check(d <= 0.5 * Math.PI / 180, cs.name + ' call ' + i + ': angle'); }
  });
  if (g.backprojection_crc !== undefined) { /* debug getters, rebuilt on the host on demand */
    const bp = tracker.getBackProjectionImg();
    check(crc32(bp.data) === g.backprojection_crc, cs.name + ': getBackProjectionImg bytes');
    const pdf = tracker.getPdf();
    g.pdf_samples.forEach(function (p) { check(pdf[p[0]][p[1]] === p[2], cs.name + ': getPdf[' + p[0] + '][' + p[1] + ']'); });
  }
  tracker.release();
});

job.facetrackr.forEach(function (cs) {
  const g = cs.golden;
  const events = [];
  listeners.facetrackingEvent = [function (e) { events.push(e); }];
  const canvas = new Canvas(cs.w, cs.h);
  const ft = new headtrackr.facetrackr.Tracker(Object.assign({}, g.params));
  ft.init(canvas);
  g.calls.forEach(function (call, i) {
    canvas.setFrame(fs.readFileSync(path.resolve(base, cs.frames[i])));
    ft.track();
    const t = ft.getTrackingObject();
    check(t.detection === call.detection, cs.name + ' call ' + i + ': detection ' + t.detection + ' vs ' + call.detection);
    check(t.confidence === call.confidence, cs.name + ' call ' + i + ': confidence');
    const tol = call.detection === 'CS' ? 1 : 0, tols = call.detection === 'CS' ? 4 : 0;
    check(near(t.x, call.x, tol) && near(t.y, call.y, tol) && near(t.width, call.width, tols) && near(t.height, call.height, tols), cs.name + ' call ' + i + ': rect');
  });
  check(events.length === g.events.length, cs.name + ': facetrackingEvent count ' + events.length + ' vs ' + g.events.length);
  ft.release();
});

/* headtrackr.Tracker facade: one step() per frame == the reference's track() body (facetrackr -> Smoother -> headposition) */
(job.pipeline || []).forEach(function (cs) {
  const g = cs.golden;
  const statuses = [];
  const canvas = new Canvas(cs.w, cs.h);
  const tr = new headtrackr.Tracker(Object.assign({ whitebalancing: g.whitebalancing !== false, onEvent: function (t, e) { if (t === 'headtrackrStatus') statuses.push(e.status); } }, g.params));
  tr.init(canvas, canvas);
  g.calls.forEach(function (call, i) {
    statuses.length = 0;
    canvas.setFrame(fs.readFileSync(path.resolve(base, cs.frames[i])));
    const r = tr.step();
    check(JSON.stringify(statuses) === JSON.stringify(call.status), cs.name + ' frame ' + i + ': status ' + JSON.stringify(statuses) + ' vs ' + JSON.stringify(call.status));
    check(r.face.detection === call.detection, cs.name + ' frame ' + i + ': detection');
    const tol = call.detection === 'CS' ? 1.5 : 0; /* smoothed camshift output: +-1 px budget of the track itself */
    check(near(r.face.x, call.x, tol) && near(r.face.y, call.y, tol) && near(r.face.width, call.width, 4) && near(r.face.height, call.height, 4), cs.name + ' frame ' + i + ': face');
    if (call.head === null) check(r.head === null, cs.name + ' frame ' + i + ': no head position expected');
    else check(r.head !== null && near(r.head.x, call.head[0], 0.5) && near(r.head.y, call.head[1], 0.5) && near(r.head.z, call.head[2], 2.5), cs.name + ' frame ' + i + ': head ' + JSON.stringify(r.head) + ' vs ' + JSON.stringify(call.head));
  });
  check(near(tr.getFOV(), g.fov, 1.0), cs.name + ': fov');
  tr.stop(); /* releases the facade's face tracker (and its camshift slot) */
});

/* device-slot hygiene (not in the reference): 1000 lost-track / redetect cycles of the facade's tracker replacement must not
 * grow the camshift reservation — every replaced facetrackr.Tracker hands its slot back */
(function () {
  const pool = headtrackr.camshift._pool;
  const canvas = new Canvas(64, 48);
  let ft = new headtrackr.facetrackr.Tracker({ whitebalancing: false });
  ft.init(canvas);
  const reserved0 = pool.reserved, next0 = pool.next;
  for (let i = 0; i < 1000; i++) {
    ft.release();
    ft = new headtrackr.facetrackr.Tracker({ whitebalancing: false });
    ft.init(canvas);
  }
  check(pool.reserved === reserved0 && pool.next === next0, 'camshift slots leaked: reserved ' + reserved0 + ' -> ' + pool.reserved + ', next ' + next0 + ' -> ' + pool.next);
  ft.release();
  check(pool.free.length === pool.next, 'every camshift slot is back in the pool after the tests (' + pool.free.length + ' of ' + pool.next + ')');
})();

/* batch entry point (async): same frames in one call == per-frame results */
(async function () {
  if (job.detect.length) {
    const same = job.detect.filter(function (c) { return c.w === 320 && c.h === 240 && c.interval === 5; }).slice(0, 6);
    const n = same.length, buf = new Uint8Array(n * 320 * 240 * 4);
    same.forEach(function (c, i) { buf.set(fs.readFileSync(path.resolve(base, c.frame)), i * 320 * 240 * 4); });
    const res = await headtrackr.ccv.detect_objects_batch(buf, n, 320, 240, headtrackr.cascade, 5, 1);
    same.forEach(function (c, i) {
      const gg = c.golden.min_neighbors === 1 ? c.golden.grouped : null;
      if (!gg) return;
      check(res[i].length === gg.length, c.name + ': batch grouped count');
      for (let k = 0; k < Math.min(res[i].length, gg.length); k++) check(res[i][k].x === gg[k].x && res[i][k].confidence === gg[k].confidence, c.name + ': batch grouped[' + k + ']');
    });
  }
  console.log(JSON.stringify(out));
})().catch(function (e) { out.ok = false; out.errors.push('exception: ' + e.stack); console.log(JSON.stringify(out)); });
