'use strict';
/* tests/js/mock_addon.js — TEST INFRASTRUCTURE: the product addon's SINGLE-FRAME entry points (what ccv.grayscale / detect_objects /
 * getWhitebalance / camshift.Tracker / facetrackr.Tracker / headtrackr.Tracker of headtrackr_amd/js use) implemented on the CPU oracle
 * (tests/js/oracle_addon.node = oracle/ht_oracle.c).  install() puts it where headtrackr.js will `require('./headtrackr_hip.node')`, so the
 * facade's host logic — seq construction, grouping, the WB -> VJ -> CS state machine, Smoother / headposition, the main.js loop and its
 * debug overlay — runs against the reference-JS golden vectors on a box without a GPU (tests/js/parity_cpu.js).
 * Argument lists and result shapes follow csrc/ht_napi.cc; a "device copy" of the bound frame is a copy of the bytes, as on the GPU.
 * The batch / pipelined entry points (DeviceBatch, detectAsync, allgatherBest, ...) are NOT mocked: they exist only to drive the GPU. */
const path = require('path');
const oracle = require(path.join(__dirname, 'oracle_addon.node'));

const CS_CALC_ANGLES_OFFSET = 4096 * 4 + 4 * 4 + 5 * 8; /* ho_cs_state.calc_angles */
const calls = {}; /* entry point -> number of calls (the tests assert that the facade really went through the addon interface) */
function count(name) { calls[name] = (calls[name] || 0) + 1; }

function packHits(h, n) {
  const k = h.sum.length;
  return { frame: new Uint32Array(k), x: Uint16Array.from(h.x), y: Uint16Array.from(h.y), scale: Uint8Array.from(h.scale), q: Uint8Array.from(h.q),
    sum: h.sum, counts: Uint32Array.from([k].concat(new Array(Math.max(0, n - 1)).fill(0))) };
}
/* csrc/ht_napi.cc bind_host_frames / ht_detect_batch: a frame of another size re-builds the geometry natively, with level sizes from
 * libm instead of the V8-computed ones a JS host is supposed to pass (headtrackr.js levelDims) — allowed, but counted: the facade should
 * always have announced the size through setGeometry */
function implicitGeometry(c, w, h) { if (w !== c.w || h !== c.h) { count('implicitGeometry'); c.w = w; c.h = h; c.frame = null; } }
function need(c) { if (!c || c.destroyed) throw new Error('mock addon: destroyed or missing context'); return c; }
function bound(c) { if (!c.frame) throw new Error('mock addon: no frames bound'); return c.frame; }

const mock = {
  abiVersion: 2, INPUT_RGBA: 0, INPUT_GRAY_IN_R: 1, DETECT_WHITEBALANCE: 32, calls: calls,
  createContext: function (o) { count('createContext'); return { cascade: Uint8Array.from(o.cascade), interval: o.interval, frame: null, w: 0, h: 0, cs: [] }; },
  destroy: function (c) { count('destroy'); c.destroyed = true; },
  deviceCount: function () { return 1; },
  setGeometry: function (c, w, h, batch, dims) {
    count('setGeometry'); need(c);
    if (!(dims instanceof Int32Array) || dims[0] !== w || dims[1] !== h) throw new Error('mock addon: level sizes must start with the frame size');
    c.w = w; c.h = h; c.frame = null;
  },
  upload: function (c, data, n, w, h) {
    count('upload'); need(c);
    if (n !== 1) throw new Error('mock addon: one frame per call on the drop-in path');
    implicitGeometry(c, w, h);
    c.frame = Uint8Array.from(data.subarray(0, w * h * 4));
  },
  grayscale: function (c, data, n, w, h) { count('grayscale'); need(c); oracle.grayscale(data, w, h); },
  detect: function (c, data, n, w, h, flags) {
    count('detect'); need(c);
    if (n !== 1) throw new Error('mock addon: one frame per call on the drop-in path');
    implicitGeometry(c, w, h);
    c.frame = Uint8Array.from(data.subarray(0, w * h * 4)); /* ht_detect_batch uploads */
    return packHits(oracle.detectRaw(c.frame, w, h, flags & 1, c.cascade, c.interval), n);
  },
  detectEnqueue: function (c, flags) { count('detectEnqueue'); need(c); bound(c); c.enqueued = { flags: flags, frame: c.frame }; },
  detectCollect: function (c) {
    count('detectCollect'); need(c);
    if (!c.enqueued) throw new Error('mock addon: nothing enqueued');
    const e = c.enqueued; c.enqueued = null;
    return packHits(oracle.detectRaw(e.frame, c.w, c.h, e.flags & 1, c.cascade, c.interval), 1);
  },
  whitebalance: function (c, data, n, w, h) {
    count('whitebalance'); need(c);
    implicitGeometry(c, w, h);
    c.frame = Uint8Array.from(data.subarray(0, w * h * 4));
    return Float64Array.from([oracle.whitebalance(c.frame, w, h)]);
  },
  whitebalanceBound: function (c, n) { count('whitebalanceBound'); need(c); return Float64Array.from([oracle.whitebalance(bound(c), c.w, c.h)]); },
  camshiftReserve: function (c, n) { count('camshiftReserve'); need(c); while (c.cs.length < n) c.cs.push(null); },
  camshiftInitBound: function (c, n, first, rect) {
    count('camshiftInitBound'); need(c);
    if (n !== 1 || !(rect instanceof Int32Array) || first >= c.cs.length) throw new Error('mock addon: camshiftInitBound(ctx, 1, reserved slot, Int32Array)');
    const st = new Uint8Array(oracle.csStateBytes);
    oracle.csInit(st, bound(c), c.w, c.h, rect[0], rect[1], rect[2], rect[3], 1);
    c.cs[first] = st;
  },
  camshiftTrackBound: function (c, n, first, calcAngles, fetch) {
    count('camshiftTrackBound'); need(c);
    const st = c.cs[first];
    if (n !== 1 || !st) throw new Error('mock addon: camshiftTrackBound on a slot without initTracker');
    new DataView(st.buffer).setInt32(CS_CALC_ANGLES_OFFSET, calcAngles ? 1 : 0, true); /* the product takes calcAngles per track call */
    const r = oracle.csTrack(st, bound(c), c.w, c.h);
    return fetch === false ? undefined : r;
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
