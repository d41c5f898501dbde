'use strict';
/* tests/js/mock_addon.js — TEST INFRASTRUCTURE: the product addon's entry points (csrc/ht_napi.cc) implemented on the CPU oracle
 * (tests/js/oracle_addon.node = oracle/ht_oracle.c), so that everything ABOVE the addon interface — headtrackr_amd/js/headtrackr.js and
 * tracker.js, unchanged — runs against the reference-JS golden vectors on a box without a GPU (tests/js/parity_cpu.js): seq construction,
 * grouping, the WB -> VJ -> CS state machine, Smoother / headposition, the main.js loop and its debug overlay, and the host side of
 * ccv.DeviceBatch / detect_objects_batch (frame sets, binding, the enqueue / collect / re-enqueue order, the result rings).
 * Argument lists and result shapes follow ht_napi.cc.  Device memory is a Uint8Array: an upload copies, a bind is a view (what the
 * kernels would read at execution time).  What the mock cannot say anything about is the kernels — they are checked on the GPU by the
 * same test files without the mock. */
const path = require('path');
const oracle = require(path.join(__dirname, 'oracle_addon.node'));

const CS_CALC_ANGLES_OFFSET = 4096 * 4 + 4 * 4 + 5 * 8; /* ho_cs_state.calc_angles */
const TRACK_RING = 4; /* enqueue-only track steps that may be outstanding (ht_camshift_track_batch with out == NULL) */
const calls = {}; /* entry point -> number of calls (the tests assert that the facade really went through the addon interface) */
function count(name) { calls[name] = (calls[name] || 0) + 1; }

/* csrc/ht_napi.cc bind_host_frames / ht_detect_batch: a frame of another size re-builds the geometry natively, with level sizes from
 * libm instead of the V8-computed ones a JS host is supposed to pass (headtrackr.js levelDims) — allowed, but counted: the facade should
 * always have announced the size through setGeometry */
function implicitGeometry(c, w, h, n) {
  if (w !== c.w || h !== c.h || n > c.maxBatch) { count('implicitGeometry'); c.w = w; c.h = h; c.maxBatch = Math.max(n, 1); c.frames = null; c.n = 0; }
}
function need(c) { if (!c || c.kind !== 'ctx' || c.destroyed) throw new TypeError('mock addon: expected a live context'); return c; }
function needDev(d) { if (!d || d.kind !== 'dev' || !d.buf) throw new TypeError('mock addon: expected a live device buffer'); return d; }
function bound(c) { if (!c.frames || c.n < 1) throw new Error('mock addon: no frames bound'); return c.frames; }
function fbytes(c) { return c.w * c.h * 4; }
function frameOf(frames, c, f) { return frames.subarray(f * c.stride, f * c.stride + fbytes(c)); }
function hostCopy(c, data, n, w, h) {
  implicitGeometry(c, w, h, n);
  c.frames = Uint8Array.from(data.subarray(0, n * w * h * 4)); c.n = n; c.stride = w * h * 4;
}

function detectFrames(c, frames, n, stride, flags) { /* raw hits of n frames in (frame, scale, q, y, x) order + per-frame counts */
  const per = [];
  let total = 0;
  for (let f = 0; f < n; f++) { per.push(oracle.detectRaw(frames.subarray(f * stride, f * stride + fbytes(c)), c.w, c.h, flags & 1, c.cascade, c.interval)); total += per[f].sum.length; }
  const out = { frame: new Uint32Array(total), x: new Uint16Array(total), y: new Uint16Array(total), scale: new Uint8Array(total), q: new Uint8Array(total),
    sum: new Float64Array(total), counts: new Uint32Array(n) };
  let k = 0;
  per.forEach(function (h, f) {
    out.counts[f] = h.sum.length;
    for (let i = 0; i < h.sum.length; i++, k++) { out.frame[k] = f; out.x[k] = h.x[i]; out.y[k] = h.y[i]; out.scale[k] = h.scale[i]; out.q[k] = h.q[i]; out.sum[k] = h.sum[i]; }
  });
  return out;
}
function enqueue(c, flags) { bound(c); c.enqueued = { flags: flags, frames: c.frames, n: c.n, stride: c.stride }; }
function takeEnqueued(c) { if (!c.enqueued) throw new Error('mock addon: nothing enqueued'); const e = c.enqueued; c.enqueued = null; return e; }
function trackAll(c, n, first, calcAngles) {
  const frames = bound(c), out = new Float64Array(9 * n);
  for (let i = 0; i < n; i++) {
    const st = c.cs[first + i];
    if (!st) throw new Error('mock addon: track on a slot without initTracker');
    new DataView(st.buffer).setInt32(CS_CALC_ANGLES_OFFSET, calcAngles ? 1 : 0, true); /* the product takes calcAngles per track call */
    out.set(oracle.csTrack(st, frameOf(frames, c, i), c.w, c.h), 9 * i);
  }
  return out;
}

const mock = {
  abiVersion: 2, INPUT_RGBA: 0, INPUT_GRAY_IN_R: 1, DETECT_WHITEBALANCE: 32, calls: calls,
  createContext: function (o) {
    count('createContext');
    return { kind: 'ctx', cascade: Uint8Array.from(o.cascade), interval: o.interval, frames: null, n: 0, stride: 0, w: 0, h: 0, maxBatch: 0, cs: [], ring: [], back: null };
  },
  destroy: function (c) { count('destroy'); if (c && c.kind === 'ctx') c.destroyed = true; },
  deviceCount: function () { return 1; },
  exitNow: function (code) { process.exit(code | 0); },
  info: function (c) { need(c); return { levels: 0, windowsPerFrame: 0, pyramidBytesPerFrame: 0 }; },
  setGeometry: function (c, w, h, batch, dims) {
    count('setGeometry'); need(c);
    if (dims !== null && dims !== undefined && (!(dims instanceof Int32Array) || dims[0] !== w || dims[1] !== h)) throw new Error('mock addon: level sizes must start with the frame size');
    c.w = w; c.h = h; c.maxBatch = batch; c.frames = null; c.n = 0; c.enqueued = null;
  },
  /* ---- host frames --------------------------------------------------------------------------------------------------------------- */
  upload: function (c, data, n, w, h) { count('upload'); need(c); hostCopy(c, data, n, w, h); },
  grayscale: function (c, data, n, w, h) { count('grayscale'); need(c); for (let f = 0; f < n; f++) oracle.grayscale(data.subarray(f * w * h * 4, (f + 1) * w * h * 4), w, h); },
  detect: function (c, data, n, w, h, flags) { count('detect'); need(c); hostCopy(c, data, n, w, h); return detectFrames(c, c.frames, n, c.stride, flags); },
  detectAsync: function (c, data, n, w, h, flags) { count('detectAsync'); try { return Promise.resolve(mock.detect(c, data, n, w, h, flags)); } catch (e) { return Promise.reject(e); } },
  whitebalance: function (c, data, n, w, h) { count('whitebalance'); need(c); hostCopy(c, data, n, w, h); return mock.whitebalanceBound(c, n); },
  hostAlloc: function (bytes) { count('hostAlloc'); return new Uint8Array(bytes); },
  hostFree: function (arr) { count('hostFree'); if (!(arr instanceof Uint8Array)) throw new TypeError('mock addon: hostFree(array from hostAlloc)'); },
  uploadAsync: function (c, pinned, n) { count('uploadAsync'); need(c); if (n > c.maxBatch) throw new Error('mock addon: uploadAsync beyond the batch size'); c.back = { frames: Uint8Array.from(pinned.subarray(0, n * fbytes(c))), n: n }; },
  swapFrames: function (c) { count('swapFrames'); need(c); if (!c.back) throw new Error('mock addon: swapFrames without uploadAsync'); c.frames = c.back.frames; c.n = c.back.n; c.stride = fbytes(c); c.back = null; },
  /* ---- device buffers ------------------------------------------------------------------------------------------------------------ */
  deviceAlloc: function (c, bytes) { count('deviceAlloc'); need(c); return { kind: 'dev', buf: new Uint8Array(bytes), owner: c }; },
  deviceFree: function (c, d) { count('deviceFree'); need(c); needDev(d); d.buf = null; },
  deviceUpload: function (c, d, off, src) { count('deviceUpload'); need(c); needDev(d); if (off + src.length > d.buf.length) throw new RangeError('mock addon: outside the device buffer'); d.buf.set(src, off); },
  bindDevice: function (c, d, off, n, stride) {
    count('bindDevice'); need(c); needDev(d);
    if (n > c.maxBatch || stride < fbytes(c) || off + n * stride > d.buf.length) throw new RangeError('mock addon: bindDevice outside the geometry / the buffer');
    c.frames = d.buf.subarray(off, off + n * stride); c.n = n; c.stride = stride;
  },
  framesBound: function (c) { need(c); return c.n; },
  framesEnqueued: function (c) { need(c); return c.enqueued ? c.enqueued.n : 0; },
  graphLaunches: function (c) { need(c); return 0; },
  /* ---- detect -------------------------------------------------------------------------------------------------------------------- */
  detectEnqueue: function (c, flags) { count('detectEnqueue'); need(c); enqueue(c, flags); },
  detectCollect: function (c) { count('detectCollect'); need(c); const e = takeEnqueued(c); return detectFrames(c, e.frames, e.n, e.stride, e.flags); },
  collectBest: function (c, minNeighbors, requeueFlags) { /* ht_detect_collect_best(_requeue): best face per frame of the ENQUEUED batch */
    count('collectBest'); need(c);
    const e = takeEnqueued(c), best = new Float64Array(6 * e.n);
    let hits = 0;
    if (e.flags & mock.DETECT_WHITEBALANCE) c.wbOf = { frames: e.frames, n: e.n, stride: e.stride };
    for (let f = 0; f < e.n; f++) {
      const r = oracle.bestFace(e.frames.subarray(f * e.stride, f * e.stride + fbytes(c)), c.w, c.h, e.flags & 1, c.cascade, c.interval, minNeighbors === undefined ? 1 : minNeighbors);
      best.set(r.subarray(0, 6), 6 * f); hits += r[6];
    }
    if (requeueFlags !== undefined && requeueFlags >= 0) enqueue(c, requeueFlags);
    return { best: best, hits: hits };
  },
  whitebalanceBound: function (c, n) {
    count('whitebalanceBound'); need(c);
    const frames = bound(c), out = new Float64Array(n);
    for (let f = 0; f < n; f++) out[f] = oracle.whitebalance(frameOf(frames, c, f), c.w, c.h);
    return out;
  },
  detectWhitebalance: function (c, n) { /* of the batch last COLLECTED that was enqueued with DETECT_WHITEBALANCE */
    count('detectWhitebalance'); need(c);
    if (!c.wbOf || n > c.wbOf.n) throw new Error('mock addon: no collected batch carried DETECT_WHITEBALANCE');
    const out = new Float64Array(n);
    for (let f = 0; f < n; f++) out[f] = oracle.whitebalance(c.wbOf.frames.subarray(f * c.wbOf.stride, f * c.wbOf.stride + fbytes(c)), c.w, c.h);
    return out;
  },
  allgatherBest: function (ctxs, recs, per) { /* one rank per context; every rank's table = the concatenation */
    count('allgatherBest');
    const out = new Float64Array(6 * ctxs.length * per);
    recs.forEach(function (r, k) { out.set(r.subarray(0, 6 * per), 6 * per * k); });
    return out;
  },
  /* ---- camshift ------------------------------------------------------------------------------------------------------------------ */
  camshiftReserve: function (c, n) { count('camshiftReserve'); need(c); while (c.cs.length < n) c.cs.push(null); },
  camshiftInitBound: function (c, n, first, rects) {
    count('camshiftInitBound'); need(c);
    if (!(rects instanceof Int32Array) || rects.length < 4 * n || first + n > c.cs.length) throw new TypeError('mock addon: camshiftInitBound(ctx, n, first, Int32Array rects[4n]) on reserved slots');
    const frames = bound(c);
    if (n > c.n) throw new Error('mock addon: more streams than bound frames');
    for (let i = 0; i < n; i++) {
      const st = new Uint8Array(oracle.csStateBytes);
      oracle.csInit(st, frameOf(frames, c, i), c.w, c.h, rects[4 * i], rects[4 * i + 1], rects[4 * i + 2], rects[4 * i + 3], 1);
      c.cs[first + i] = st;
    }
  },
  camshiftTrackBound: function (c, n, first, calcAngles, fetch) {
    count('camshiftTrackBound'); need(c);
    const r = trackAll(c, n, first, calcAngles);
    if (fetch !== false) return r;
    if (c.ring.length >= TRACK_RING) throw new Error('mock addon: more than ' + TRACK_RING + ' enqueue-only track steps outstanding');
    c.ring.push(r);
    return undefined;
  },
  camshiftTrackCollect: function (c, n) { /* the OLDEST outstanding enqueue-only step */
    count('camshiftTrackCollect'); need(c);
    if (!c.ring.length) throw new Error('mock addon: no enqueue-only track step outstanding');
    const r = c.ring.shift();
    if (r.length !== 9 * n) throw new Error('mock addon: camshiftTrackCollect with another stream count than the step');
    return r;
  },
  camshiftTrackSequence: function (c, first, n, calcAngles, d, offs, stride, outAll, fetch) {
    count('camshiftTrackSequence'); need(c); needDev(d);
    if (!(offs instanceof Float64Array) || !offs.length) throw new TypeError('mock addon: camshiftTrackSequence needs Float64Array byte offsets');
    const keep = { frames: c.frames, n: c.n, stride: c.stride }, all = new Float64Array(9 * n * offs.length);
    for (let k = 0; k < offs.length; k++) {
      if (offs[k] + n * stride > d.buf.length) throw new RangeError('mock addon: a call\'s frames lie outside the device buffer');
      c.frames = d.buf.subarray(offs[k], offs[k] + n * stride); c.n = n; c.stride = stride;
      all.set(trackAll(c, n, first, calcAngles), 9 * n * k);
    }
    c.frames = keep.frames; c.n = keep.n; c.stride = keep.stride;
    c.seq = { all: all, n: n, ncalls: offs.length };
    if (fetch === false) return undefined;
    return outAll ? all : all.slice(9 * n * (offs.length - 1));
  },
  camshiftSequenceCollect: function (c, n, ncalls, outAll) {
    count('camshiftSequenceCollect'); need(c);
    if (!c.seq || c.seq.n !== n || c.seq.ncalls !== ncalls) throw new Error('mock addon: camshiftSequenceCollect does not match the pending sequence');
    const s = c.seq; c.seq = null;
    return outAll ? s.all : s.all.slice(9 * n * (ncalls - 1));
  }
};

/* make `require('.../headtrackr_hip.node')` return the mock, whether or not the product addon has been built (call before loading
 * headtrackr.js) */
mock.install = function () {
  const Module = require('module');
  const load = Module._load;
  Module._load = function (request, parent, isMain) {
    if (/headtrackr_hip\.node$/.test(request)) return mock;
    return load.apply(this, arguments);
  };
  return mock;
};

module.exports = mock;
