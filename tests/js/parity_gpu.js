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
    /* north_star tolerance: +-1 px, +-0.5 deg.  Sizes are multiples of 4 (`<< 2`, camshift.js:240-241) and the window size is
     * floor(1.1 * size): +-1 px means EQUAL for them */
    check(near(sw.x, call.sw[0], 1) && near(sw.y, call.sw[1], 1) && sw.width === call.sw[2] && sw.height === call.sw[3], cs.name + ' call ' + i + ': search window');
    check(near(to.x, call.x, 1) && near(to.y, call.y, 1), cs.name + ' call ' + i + ': centre');
    check(to.width === call.width && to.height === call.height, cs.name + ' call ' + i + ': size');
    out.cs_total = (out.cs_total || 0) + 1;
    if (sw.x === call.sw[0] && sw.y === call.sw[1] && to.x === call.x && to.y === call.y) out.cs_exact = (out.cs_exact || 0) + 1;
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
    const tol = call.detection === 'CS' ? 1 : 0;
    check(near(t.x, call.x, tol) && near(t.y, call.y, tol) && t.width === call.width && t.height === call.height, cs.name + ' call ' + i + ': rect');
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
    check(near(r.face.x, call.x, tol) && near(r.face.y, call.y, tol) && near(r.face.width, call.width, 1e-9) && near(r.face.height, call.height, 1e-9), cs.name + ' frame ' + i + ': face');
    if (call.head === null) check(r.head === null, cs.name + ' frame ' + i + ': no head position expected');
    else check(r.head !== null && near(r.head.x, call.head[0], 0.5) && near(r.head.y, call.head[1], 0.5) && near(r.head.z, call.head[2], 2.5), cs.name + ' frame ' + i + ': head ' + JSON.stringify(r.head) + ' vs ' + JSON.stringify(call.head));
  });
  check(near(tr.getFOV(), g.fov, 1.0), cs.name + ': fov');
  tr.stop(); /* releases the facade's face tracker (and its camshift slot) */
});

/* headtrackr.Tracker against the reference's OWN main.js loop (tests/golden/debug.json, recorded by driving the unmodified
 * headtrackr.Tracker frame by frame): status events per frame, head position, and the debug overlay — the stroke calls made on
 * the debug canvas (main.js:199-219) and, while every argument so far matched exactly, the canvas pixels (CRC under the declared
 * 1-pixel rasterisation, back-projection blit of facetrackr.js:194-196 included). */
(job.mainjs || []).forEach(function (cs) {
  const g = cs.golden;
  const statuses = [], heads = [];
  const video = new Canvas(cs.w, cs.h), canvas = new Canvas(cs.w, cs.h), debug = new Canvas(cs.w, cs.h);
  const dctx = debug.getContext('2d'), strokes = [];
  ['translate', 'rotate'].forEach(function (op) { const f = dctx[op]; dctx[op] = function () { strokes.push([op].concat(Array.prototype.slice.call(arguments))); return f.apply(dctx, arguments); }; });
  const sr = dctx.strokeRect; dctx.strokeRect = function (x, y, w, h) { strokes.push(['strokeRect', dctx.strokeStyle, x, y, w, h]); return sr.call(dctx, x, y, w, h); };
  const tr = new headtrackr.Tracker(Object.assign({ debug: debug, onEvent: function (t, e) { if (t === 'headtrackrStatus') statuses.push(e.status); if (t === 'headtrackingEvent') heads.push([e.x, e.y, e.z]); } }, g.params));
  tr.init(video, canvas);
  let exact = true;
  g.calls.forEach(function (call, i) {
    statuses.length = 0; heads.length = 0; strokes.length = 0;
    video.setFrame(fs.readFileSync(path.resolve(base, cs.frames[i])));
    tr.step();
    check(JSON.stringify(statuses) === JSON.stringify(call.status), cs.name + ' frame ' + i + ': status ' + JSON.stringify(statuses) + ' vs ' + JSON.stringify(call.status));
    if (call.head === null) check(heads.length === 0, cs.name + ' frame ' + i + ': no head event expected');
    else check(heads.length > 0 && near(heads[heads.length - 1][0], call.head[0], 0.5) && near(heads[heads.length - 1][1], call.head[1], 0.5) && near(heads[heads.length - 1][2], call.head[2], 2.5), cs.name + ' frame ' + i + ': head');
    check(strokes.length === call.strokes.length, cs.name + ' frame ' + i + ': ' + strokes.length + ' stroke calls vs ' + call.strokes.length);
    for (let k = 0; k < Math.min(strokes.length, call.strokes.length); k++) {
      const a = strokes[k], b = call.strokes[k];
      check(a[0] === b[0] && a.length === b.length, cs.name + ' frame ' + i + ' stroke ' + k + ': ' + a[0] + ' vs ' + b[0]);
      for (let q = 1; q < Math.min(a.length, b.length); q++) {
        if (typeof b[q] === 'string') { check(a[q] === b[q], cs.name + ' frame ' + i + ' stroke ' + k + ': style'); continue; }
        if (b[q] === null) { check(Number.isNaN(a[q]), cs.name + ' frame ' + i + ' stroke ' + k + ': NaN expected'); continue; }
        const tol = a[0] === 'rotate' ? 0.5 * Math.PI / 180 : (a[0] === 'translate' ? 1 : 0); /* camshift budget: +-1 px, +-0.5 deg; sizes are multiples of 4: equal */
        check(near(a[q], b[q], tol), cs.name + ' frame ' + i + ' stroke ' + k + ' arg ' + q + ': ' + a[q] + ' vs ' + b[q]);
        if (a[0] === 'rotate' ? !near(a[q], b[q], 1e-9) : a[q] !== b[q]) exact = false; /* Math.atan2 vs libm atan2: last-bit differences of the angle */
      }
    }
    if (exact) check(crc32(debug.pixels) === call.debug_crc, cs.name + ' frame ' + i + ': debug canvas pixels');
  });
  check(exact, cs.name + ': every stroke argument matched the reference exactly');
  check(near(tr.getFOV(), g.fov, 1.0), cs.name + ': fov');
  tr.stop();
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
  /* frame-sharded batch over every visible GPU + RCCL all-gather of the best-face rects (BASELINE.json configs[3] for a JS host):
   * same rect lists as the single-GPU call, and the gathered table == facetrackr's choice per frame.  On a 1-GPU box this runs
   * with one rank (no RCCL call; the RCCL path itself is forced in tests/test_gpu_shapes.py). */
  if (job.detect.length) {
    const same = job.detect.filter(function (c) { return c.w === 320 && c.h === 240 && c.interval === 5; }).slice(0, 7);
    const n = same.length, buf = new Uint8Array(n * 320 * 240 * 4);
    same.forEach(function (c, i) { buf.set(fs.readFileSync(path.resolve(base, c.frame)), i * 320 * 240 * 4); });
    const ndev = headtrackr.deviceCount();
    check(ndev >= 1, 'deviceCount() = ' + ndev);
    const devices = [];
    for (let d = 0; d < ndev; d++) devices.push(d);
    const one = await headtrackr.ccv.detect_objects_batch(buf, n, 320, 240, headtrackr.cascade, 5, 1);
    const res = await headtrackr.ccv.detect_objects_batch(buf, n, 320, 240, headtrackr.cascade, 5, 1, { devices: devices, gather: true });
    check(res.length === n && res.best && res.best.length === n, 'sharded batch: result shape');
    for (let i = 0; i < n; i++) {
      check(JSON.stringify(res[i]) === JSON.stringify(one[i]), 'sharded batch: frame ' + i + ' differs from the single-GPU result');
      let b;
      one[i].forEach(function (r) { if (b === undefined || r.confidence > b.confidence) b = r; });
      if (b === undefined) check(res.best[i] === null, 'gathered best of frame ' + i + ' should be null');
      else check(res.best[i] !== null && res.best[i].x === b.x && res.best[i].y === b.y && res.best[i].width === b.width && res.best[i].confidence === b.confidence &&
        res.best[i].neighbors === b.neighbors, 'gathered best of frame ' + i);
    }
  }
  /* ---- the pipelined path from Node (DeviceBatch: frames resident in HBM; detectEnqueue / collectBest(requeue) over several
   * contexts; whitebalance fused into the gray pass; camshift call sequences) against the same golden vectors --------------- */
  if (job.detect.length) {
    const same = job.detect.filter(function (c) { return c.w === 320 && c.h === 240 && c.interval === 5 && c.golden.min_neighbors === 1; });
    const n = same.length, fb = 320 * 240 * 4, buf = new Uint8Array(n * fb);
    same.forEach(function (c, i) { buf.set(fs.readFileSync(path.resolve(base, c.frame)), i * fb); });
    const b = new headtrackr.ccv.DeviceBatch(320, 240, n, { depth: 3 });
    b.upload(buf);
    const lists = b.detect(1);
    same.forEach(function (c, i) {
      const gg = c.golden.grouped;
      check(lists[i].length === gg.length, 'DeviceBatch.detect ' + c.name + ': grouped count');
      for (let k = 0; k < Math.min(lists[i].length, gg.length); k++) {
        ['x', 'y', 'width', 'height', 'confidence', 'neighbors'].forEach(function (q) { check(lists[i][k][q] === gg[k][q], 'DeviceBatch.detect ' + c.name + ': grouped[' + k + '].' + q); });
      }
    });
    const r = b.detectBest(7, 1); /* 7 batches over 3 contexts: enqueue x3, then collect + re-enqueue, the last ones only collected */
    check(r.best.length === 6 * n && r.batches === 7, 'DeviceBatch.detectBest: result shape');
    same.forEach(function (c, i) {
      let bb; /* facetrackr's choice (facetrackr.js:157-165) among the golden grouped rects */
      c.golden.grouped.forEach(function (g) { if (bb === undefined || g.confidence > bb.confidence) bb = g; });
      const o = 6 * i;
      if (bb === undefined) check(r.best[o + 5] === 0 && r.best[o + 4] === -10000, 'DeviceBatch.detectBest ' + c.name + ': no face');
      else check(r.best[o] === bb.x && r.best[o + 1] === bb.y && r.best[o + 2] === bb.width && r.best[o + 3] === bb.height && r.best[o + 4] === bb.confidence && r.best[o + 5] === bb.neighbors,
        'DeviceBatch.detectBest ' + c.name + ': best face');
    });
    const wb = b.whitebalance();
    same.forEach(function (c, i) { check(wb[i] === c.golden.whitebalance, 'DeviceBatch.whitebalance ' + c.name); });
    b.destroy();
  }
  job.camshift.forEach(function (cs) { /* one stream, one frame set per distinct frame, every golden call in ONE trackSequence */
    const g = cs.golden;
    const b = new headtrackr.ccv.DeviceBatch(cs.w, cs.h, 1, { depth: 1, sets: cs.frames.length });
    cs.frames.forEach(function (f, k) { b.upload(new Uint8Array(fs.readFileSync(path.resolve(base, f))), k); });
    b.initTrackers(new Int32Array(g.rect), 0);
    const r = b.trackSequence(g.calls.map(function (c) { return c.frame; }), g.calcAngles, true);
    check(r.length === 9 * g.calls.length, cs.name + ' (sequence): result shape');
    g.calls.forEach(function (call, i) {
      const o = 9 * i;
      check(r[o] === call.x && r[o + 1] === call.y && r[o + 2] === call.width && r[o + 3] === call.height, cs.name + ' (sequence) call ' + i + ': track object');
      check(r[o + 5] === call.sw[0] && r[o + 6] === call.sw[1] && r[o + 7] === call.sw[2] && r[o + 8] === call.sw[3], cs.name + ' (sequence) call ' + i + ': search window');
      if (call.angle === null) check(Number.isNaN(r[o + 4]), cs.name + ' (sequence) call ' + i + ': NaN angle');
      else { let d = Math.abs(r[o + 4] - call.angle); d = Math.min(d, Math.abs(d - Math.PI)); check(d <= 1e-6, cs.name + ' (sequence) call ' + i + ': angle'); }
    });
    b.destroy();
  });
  { /* double-buffered ingest from pinned memory: uploadAsync + swapFrames under a running detect == the plain call */
    const c0 = job.detect.find(function (c) { return c.w === 320 && c.h === 240 && c.interval === 5 && c.golden.grouped.length > 0; });
    if (c0) {
      const A = require(path.join(root, 'headtrackr_amd', 'js', 'headtrackr_hip.node'));
      const pin = headtrackr.hostAlloc(2 * 320 * 240 * 4);
      const f0 = fs.readFileSync(path.resolve(base, c0.frame));
      pin.set(f0, 0); pin.fill(110, 320 * 240 * 4); /* frame 1: flat gray, no face */
      const pk = require(path.join(root, 'headtrackr_amd', 'js', 'cascade_pack.js'));
      const hnd = A.createContext({ cascade: pk.packCascade(headtrackr.cascade), interval: 5, device: 0 });
      A.setGeometry(hnd, 320, 240, 1, null);
      A.uploadAsync(hnd, pin.subarray(0, 320 * 240 * 4), 1); A.swapFrames(hnd);
      A.detectEnqueue(hnd, A.INPUT_RGBA);
      A.uploadAsync(hnd, pin.subarray(320 * 240 * 4), 1); /* next frame crosses PCIe while frame 0 is scanned */
      const r0 = A.collectBest(hnd, 1, -1);
      A.swapFrames(hnd);
      A.detectEnqueue(hnd, A.INPUT_RGBA);
      const r1 = A.collectBest(hnd, 1, -1);
      let bb;
      c0.golden.grouped.forEach(function (g) { if (bb === undefined || g.confidence > bb.confidence) bb = g; });
      check(r0.best[0] === bb.x && r0.best[4] === bb.confidence && r0.best[5] === bb.neighbors, 'uploadAsync/swapFrames: frame 0 best face');
      check(r1.best[5] === 0 && r1.hits === 0, 'uploadAsync/swapFrames: frame 1 has no face');
      check(A.framesBound(hnd) === 1 && A.framesEnqueued(hnd) === 0, 'framesBound / framesEnqueued');
      A.destroy(hnd);
      headtrackr.hostFree(pin);
    }
  }
  out.cs_parity = (out.cs_exact || 0) + '/' + (out.cs_total || 0);
  if (job.cpu_mock) out.addon_calls = require(path.join(root, 'headtrackr_amd', 'js', 'headtrackr_hip.node')).calls; /* tests/js/parity_cpu.js */
  console.log(JSON.stringify(out));
})().catch(function (e) { out.ok = false; out.errors.push('exception: ' + e.stack); console.log(JSON.stringify(out)); });
