'use strict';
/*
 * ref_harness.js — TEST INFRASTRUCTURE.  Runs the UNMODIFIED reference bundle (/root/reference/headtrackr.js,
 * or $HT_REFERENCE_JS) on top of oracle/canvas_shim.js and dumps golden vectors as JSON.
 *
 *   node oracle/ref_harness.js job.json out.json
 *
 * job.json: { "cases": [ {name, kind, w, h, ...} ] } with raw RGBA frames in files (see tests/golden/make_golden.py).
 *   kind "detect":     {frame, interval?, ops:["gray","pyramid","raw","grouped","whitebalance"]}
 *                      gray/pyramid = CRC32 of byte 0 of every pixel of each pyramid canvas, in creation order
 *                      (ccv.js:117-147: levels 1..5, levels 6..38, then for i=12..38 variants 1,2,3)
 *   kind "camshift":   {frames:[...], rect:[x,y,w,h], calcAngles} -> per track() call searchWindow + trackObj
 *   kind "facetrackr": {frames:[...], params:{...}} -> per track() call the tracking object
 * Only this file and tests/golden/make_golden.py touch the reference; nothing here ships with the product.
 */
const fs = require('fs');
const path = require('path');
const shim = require('./canvas_shim.js');

const refPath = process.env.HT_REFERENCE_JS || '/root/reference/headtrackr.js';
global.document = shim.makeDocument();
global.window = global;                       /* the bundle's UMD wrapper falls back to `this`/window */
const headtrackr = require(refPath);

const CRC_TABLE = (function () {
  const t = new Int32Array(256);
  for (let n = 0; n < 256; n++) {
    let c = n;
    for (let k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320 ^ (c >>> 1)) : (c >>> 1);
    t[n] = c;
  }
  return t;
})();
function crc32Channel0(buf) {             /* CRC32 (zlib polynomial) over byte 0 of every RGBA pixel */
  let c = -1;
  for (let i = 0; i < buf.length; i += 4) c = CRC_TABLE[(c ^ buf[i]) & 0xFF] ^ (c >>> 8);
  return (c ^ -1) >>> 0;
}
function crc32All(buf) {
  let c = -1;
  for (let i = 0; i < buf.length; i++) c = CRC_TABLE[(c ^ buf[i]) & 0xFF] ^ (c >>> 8);
  return (c ^ -1) >>> 0;
}

function loadCanvas(file, w, h) {
  const bytes = fs.readFileSync(file);
  if (bytes.length !== w * h * 4) throw new Error(file + ': expected ' + (w * h * 4) + ' bytes, got ' + bytes.length);
  return new shim.Canvas(w, h).loadRGBA(bytes);
}
function copyCanvas(c) {
  const d = new shim.Canvas(c.width, c.height);
  d._buf.set(c._buf);
  return d;
}
function rectOut(r) {
  const o = { x: r.x, y: r.y, width: r.width, height: r.height, confidence: r.confidence };
  if (r.neighbors !== undefined) o.neighbors = r.neighbors;
  if (r.neighbor !== undefined) o.neighbor = r.neighbor;
  return o;
}

function runDetect(cs, base) {
  const out = { name: cs.name, kind: 'detect', w: cs.w, h: cs.h };
  const interval = cs.interval === undefined ? 5 : cs.interval;
  const ops = cs.ops || ['gray', 'pyramid', 'raw', 'grouped'];
  const input = loadCanvas(path.resolve(base, cs.frame), cs.w, cs.h);
  out.input_crc = crc32All(input._buf);
  if (ops.indexOf('whitebalance') >= 0) out.whitebalance = headtrackr.getWhitebalance(input);
  const gray = headtrackr.ccv.grayscale(copyCanvas(input));
  if (ops.indexOf('gray') >= 0) {
    out.gray_crc = crc32Channel0(gray._buf);
    out.gray_rgba_crc = crc32All(gray._buf);
  }
  if (ops.indexOf('raw') >= 0 || ops.indexOf('pyramid') >= 0) {
    shim.stats.trackCreated = true; shim.stats.created = [];
    const seq = headtrackr.ccv.detect_objects(copyCanvas(gray), headtrackr.cascade, interval, 0);
    shim.stats.trackCreated = false;
    if (ops.indexOf('pyramid') >= 0) {
      out.pyramid = shim.stats.created.map(function (c) { return { w: c.width, h: c.height, crc: crc32Channel0(c._buf) }; });
    }
    shim.stats.created = [];
    if (ops.indexOf('raw') >= 0) out.raw = seq.map(rectOut);
  }
  if (ops.indexOf('grouped') >= 0) {
    const mn = cs.min_neighbors === undefined ? 1 : cs.min_neighbors;
    out.min_neighbors = mn;
    out.grouped = headtrackr.ccv.detect_objects(copyCanvas(gray), headtrackr.cascade, interval, mn).map(rectOut);
  }
  return out;
}

function runCamshift(cs, base) {
  const out = { name: cs.name, kind: 'camshift', w: cs.w, h: cs.h, rect: cs.rect, calcAngles: cs.calcAngles !== false, calls: [] };
  const tr = new headtrackr.camshift.Tracker({ calcAngles: cs.calcAngles !== false });
  const first = loadCanvas(path.resolve(base, cs.frames[0]), cs.w, cs.h);
  tr.initTracker(first, new headtrackr.camshift.Rectangle(cs.rect[0], cs.rect[1], cs.rect[2], cs.rect[3]));
  const start = cs.track_first ? 0 : 1;   /* track_first: also track on the init frame */
  for (let i = start; i < cs.frames.length; i++) {
    const c = loadCanvas(path.resolve(base, cs.frames[i]), cs.w, cs.h);
    const reps = cs.repeat || 1;
    for (let r = 0; r < reps; r++) {
      tr.track(c);
      const sw = tr.getSearchWindow(), to = tr.getTrackObj();
      out.calls.push({ frame: i, sw: [sw.x, sw.y, sw.width, sw.height], x: to.x, y: to.y, width: to.width, height: to.height, angle: to.angle });
    }
  }
  /* debug getters (camshift.js:172-196): the back-projection of the last tracked frame */
  const bp = tr.getBackProjectionImg();
  out.backprojection_crc = crc32All(bp.data);
  const pdf = tr.getPdf();
  out.pdf_samples = [[0, 0], [cs.w >> 1, cs.h >> 1], [cs.w - 1, cs.h - 1], [cs.rect[0] + 3, cs.rect[1] + 3]].map(function (p) {
    const x = Math.min(Math.max(p[0], 0), cs.w - 1), y = Math.min(Math.max(p[1], 0), cs.h - 1);
    return [x, y, pdf[x][y]];
  });
  return out;
}

function runFacetrackr(cs, base) {
  const out = { name: cs.name, kind: 'facetrackr', w: cs.w, h: cs.h, params: cs.params || {}, calls: [] };
  const params = Object.assign({}, cs.params || {});
  const events = [];
  const listener = function (e) { events.push({ x: e.x, y: e.y, width: e.width, height: e.height, angle: e.angle, confidence: e.confidence, detection: e.detection }); };
  document.addEventListener('facetrackingEvent', listener);
  const canvas = new shim.Canvas(cs.w, cs.h);
  const ft = new headtrackr.facetrackr.Tracker(params);
  ft.init(canvas);
  for (let i = 0; i < cs.frames.length; i++) {
    canvas.loadRGBA(fs.readFileSync(path.resolve(base, cs.frames[i])));
    ft.track();
    const t = ft.getTrackingObject();
    out.calls.push({ frame: i, x: t.x, y: t.y, width: t.width, height: t.height, angle: t.angle, confidence: t.confidence, detection: t.detection });
  }
  document.removeEventListener('facetrackingEvent', listener);
  out.events = events;
  return out;
}

/* headtrackr.Smoother (smoother.js:13-88): feed positions, record what smooth() returns */
function runSmoother(cs) {
  const out = { name: cs.name, kind: 'smoother', alpha: cs.alpha, interval: cs.interval, positions: cs.positions, calls: [] };
  const sm = new headtrackr.Smoother(cs.alpha, cs.interval);
  cs.positions.forEach(function (p, i) {
    const pos = { x: p[0], y: p[1], z: p[2], width: p[3], height: p[4] };
    if (!sm.initialized && i >= (cs.init_at || 0)) sm.init(pos);
    const r = sm.smooth(pos);
    out.calls.push(r === false ? null : [r.x, r.y, r.z, r.width, r.height]);
  });
  return out;
}

/* headtrackr.headposition.Tracker (headposition.js:35-201) */
function runHeadposition(cs) {
  const out = { name: cs.name, kind: 'headposition', camw: cs.camw, camh: cs.camh, params: cs.params || {}, faces: cs.faces, calls: [], events: [] };
  const listener = function (e) { out.events.push([e.x, e.y, e.z]); };
  document.addEventListener('headtrackingEvent', listener);
  const f0 = cs.faces[0];
  const hp = new headtrackr.headposition.Tracker({ x: f0[0], y: f0[1], width: f0[2], height: f0[3] }, cs.camw, cs.camh, Object.assign({}, cs.params || {}));
  out.fov = hp.getFOV();
  cs.faces.forEach(function (f) {
    const r = hp.track({ x: f[0], y: f[1], width: f[2], height: f[3] });
    out.calls.push([r.x, r.y, r.z]);
  });
  document.removeEventListener('headtrackingEvent', listener);
  return out;
}

/* The per-frame body of headtrackr.Tracker's loop (main.js:168-305) composed from the reference's own objects, without
 * the webcam / DOM plumbing: facetrackr -> lost-track handling -> Smoother -> headposition.  Records status messages,
 * the (smoothed) face object and the head position per frame. */
function runPipeline(cs, base) {
  const params = Object.assign({ smoothing: true, retryDetection: true, detectionInterval: 20, cameraOffset: 11.5, calcAngles: false, headPosition: true }, cs.params || {});
  const out = { name: cs.name, kind: 'pipeline', w: cs.w, h: cs.h, params: cs.params || {}, calls: [] };
  const canvas = new shim.Canvas(cs.w, cs.h);
  let facetracker, headposition, faceFound = false, firstRun = true, fov = 0;
  const headDiagonal = [];
  const smoother = new headtrackr.Smoother(0.35, params.detectionInterval + 15);
  const headEvents = [];
  const hl = function (e) { headEvents.push([e.x, e.y, e.z]); };
  document.addEventListener('headtrackingEvent', hl);
  for (let i = 0; i < cs.frames.length; i++) {
    const status = [];
    canvas.loadRGBA(fs.readFileSync(path.resolve(base, cs.frames[i])));
    if (facetracker === undefined) {
      facetracker = new headtrackr.facetrackr.Tracker({ debug: false, calcAngles: params.calcAngles, whitebalancing: cs.whitebalancing !== false });
      facetracker.init(canvas);
    }
    facetracker.track();
    let faceObj = facetracker.getTrackingObject();
    if (faceObj.detection === 'WB') status.push('whitebalance');
    if (firstRun && faceObj.detection === 'VJ') status.push('detecting');
    let head = null;
    const nh = headEvents.length;
    if (!(faceObj.confidence === 0) && faceObj.detection === 'CS') {
      if (faceObj.width === 0 || faceObj.height === 0) {
        status.push('redetecting');
        facetracker = new headtrackr.facetrackr.Tracker({ whitebalancing: false, debug: false, calcAngles: params.calcAngles });
        facetracker.init(canvas);
        faceFound = false;
        headposition = undefined;
      } else {
        if (!faceFound) { status.push('found'); faceFound = true; }
        if (params.smoothing) {
          if (!smoother.initialized) smoother.init(faceObj);
          faceObj = smoother.smooth(faceObj);
        }
        if (headposition === undefined && params.headPosition) {
          let stable = false;
          const headdiag = Math.sqrt(faceObj.width * faceObj.width + faceObj.height * faceObj.height);
          if (headDiagonal.length < 6) headDiagonal.push(headdiag);
          else {
            headDiagonal.splice(0, 1); headDiagonal.push(headdiag);
            if ((Math.max.apply(null, headDiagonal) - Math.min.apply(null, headDiagonal)) < 5) stable = true;
          }
          if (stable) {
            if (firstRun) {
              headposition = new headtrackr.headposition.Tracker(faceObj, cs.w, cs.h, { distance_from_camera_to_screen: params.cameraOffset });
              fov = headposition.getFOV();
              firstRun = false;
            } else {
              headposition = new headtrackr.headposition.Tracker(faceObj, cs.w, cs.h, { fov: fov, distance_from_camera_to_screen: params.cameraOffset });
            }
            headposition.track(faceObj);
          }
        } else if (params.headPosition) {
          headposition.track(faceObj);
        }
      }
    }
    if (headEvents.length > nh) head = headEvents[headEvents.length - 1];
    out.calls.push({ frame: i, status: status, detection: faceObj.detection, x: faceObj.x, y: faceObj.y, width: faceObj.width, height: faceObj.height,
      angle: faceObj.angle, confidence: faceObj.confidence, head: head });
  }
  document.removeEventListener('headtrackingEvent', hl);
  out.fov = fov;
  return out;
}

/* The reference's own headtrackr.Tracker (main.js:35-379), unmodified, driven frame by frame: init(video, canvas, false) skips
 * getUserMedia, the "video" is a shim canvas that already plays (currentTime > 0), and window.setTimeout is replaced by a hook
 * that parks the loop's callback (main.js:302-304) so that the harness can swap the video frame and fire it.  Records per frame:
 * status events, the tracking object, the head position event, and — with a debug canvas — the stroke calls the reference made
 * on it (main.js:199-219) plus the CRC of its pixels under the declared rasterisation. */
function runMainJs(cs, base) {
  const out = { name: cs.name, kind: 'mainjs', w: cs.w, h: cs.h, params: cs.params || {}, calls: [] };
  const video = new shim.Canvas(cs.w, cs.h), canvas = new shim.Canvas(cs.w, cs.h), debug = new shim.Canvas(cs.w, cs.h);
  video.currentTime = 1; video.paused = false; video.ended = false; video.addEventListener = function () {}; video.style = {};
  let parked = null;
  const realSetTimeout = global.setTimeout, realClear = global.clearTimeout;
  global.setTimeout = function (fn) { parked = fn; return 1; };
  global.clearTimeout = function () { parked = null; };
  const status = [], head = [], faceEv = [];
  const sl = function (e) { status.push(e.status); }, hl = function (e) { head.push([e.x, e.y, e.z]); }, fl = function (e) { faceEv.push(e.detection); };
  document.addEventListener('headtrackrStatus', sl);
  document.addEventListener('headtrackingEvent', hl);
  document.addEventListener('facetrackingEvent', fl);
  try {
    const params = Object.assign({ ui: false, debug: cs.debug ? debug : false }, cs.params || {});
    const tr = new headtrackr.Tracker(params);
    tr.init(video, canvas, false);
    for (let i = 0; i < cs.frames.length; i++) {
      status.length = 0; head.length = 0; debug._calls.length = 0;
      video.loadRGBA(fs.readFileSync(path.resolve(base, cs.frames[i])));
      if (i === 0) tr.start(); else { const fn = parked; parked = null; if (fn) fn(); }
      out.calls.push({ frame: i, status: status.slice(), trackerStatus: tr.status, head: head.length ? head[head.length - 1] : null,
        strokes: debug._calls.map(function (c) { return c.slice(); }), debug_crc: cs.debug ? crc32All(debug._buf) : null });
    }
    out.fov = tr.getFOV();
    tr.stop();
  } finally {
    global.setTimeout = realSetTimeout; global.clearTimeout = realClear;
    document.removeEventListener('headtrackrStatus', sl);
    document.removeEventListener('headtrackingEvent', hl);
    document.removeEventListener('facetrackingEvent', fl);
  }
  return out;
}

function main() {
  const jobFile = process.argv[2], outFile = process.argv[3];
  const job = JSON.parse(fs.readFileSync(jobFile, 'utf8'));
  const base = path.dirname(path.resolve(jobFile));
  const res = { reference_rev: headtrackr.rev, node: process.version, scale6: Math.pow(2, 1 / 6),
    scale6_pows: [0, 1, 2, 3, 4, 5].map(function (i) { return Math.pow(Math.pow(2, 1 / 6), i); }), cases: [] };
  job.cases.forEach(function (cs) {
    const t0 = process.hrtime.bigint();
    let r;
    if (cs.kind === 'detect') r = runDetect(cs, base);
    else if (cs.kind === 'camshift') r = runCamshift(cs, base);
    else if (cs.kind === 'facetrackr') r = runFacetrackr(cs, base);
    else if (cs.kind === 'smoother') r = runSmoother(cs);
    else if (cs.kind === 'headposition') r = runHeadposition(cs);
    else if (cs.kind === 'pipeline') r = runPipeline(cs, base);
    else if (cs.kind === 'mainjs') r = runMainJs(cs, base);
    else throw new Error('unknown case kind ' + cs.kind);
    r.gen = cs.gen;                         /* how make_golden.py synthesised the input (echoed for the tests) */
    res.cases.push(r);
    process.stderr.write(cs.name + ': ' + Number((process.hrtime.bigint() - t0) / 1000000n) + ' ms\n');
  });
  fs.writeFileSync(outFile, JSON.stringify(res));
}
main();
