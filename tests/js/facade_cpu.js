'use strict';
/* CPU-side checks of the JavaScript facade (no GPU): exports, cascade round trip, and the host-side grouping
 * (ccv.array_group + averaging + nesting filter) against the reference-JS golden vectors.  Prints one JSON line. */
const path = require('path');
const fs = require('fs');
const root = path.join(__dirname, '..', '..');
const ht = require(path.join(root, 'headtrackr_amd', 'js', 'headtrackr.js'));
/* optional arguments: other files of the same shapes (tools/cpu_soak_reference.py passes the reference's output on random cases) */
const golden = JSON.parse(fs.readFileSync(process.argv[2] || path.join(root, 'tests', 'golden', 'detect.json'), 'utf8'));
const out = { ok: true, errors: [] };
function check(cond, msg) { if (!cond) { out.ok = false; out.errors.push(msg); } }

['rev', 'cascade', 'ccv', 'camshift', 'facetrackr', 'getWhitebalance'].forEach(function (k) { check(ht[k] !== undefined, 'missing export ' + k); });
['grayscale', 'array_group', 'detect_objects'].forEach(function (k) { check(typeof ht.ccv[k] === 'function', 'missing ccv.' + k); });
['Histogram', 'Moments', 'Rectangle', 'Tracker', 'TrackObj'].forEach(function (k) { check(typeof ht.camshift[k] === 'function', 'missing camshift.' + k); });
['Tracker', 'TrackObj'].forEach(function (k) { check(typeof ht.facetrackr[k] === 'function', 'missing facetrackr.' + k); });
const c = ht.cascade;
check(c.count === 16 && c.width === 24 && c.height === 24 && c.stage_classifier.length === 16, 'cascade shape');
check(c.stage_classifier[15].count === 564 && c.stage_classifier[0].alpha[1] === 2.879683, 'cascade data');

/* grouping: raw seq -> grouped must equal what the reference produced */
const mod = require('module');
golden.cases.forEach(function (cs) {
  const seq = cs.raw.map(function (r) { return { x: r.x, y: r.y, width: r.width, height: r.height, neighbor: 1, confidence: r.confidence }; });
  /* reuse the facade's private grouping through detect_objects' tail: exposed for tests as ccv._group */
  if (cs.hits) { /* raw hits in index form (the shape the addon returns; here from the oracle): the facade's own seq construction */
    const built = ht.ccv._hitsToSeq(cs.hits, 0, cs.hits.x.length, ht.cascade, cs.interval === undefined ? 5 : cs.interval);
    check(built.length === cs.raw.length, cs.name + ': seq length ' + built.length + ' != ' + cs.raw.length);
    for (let i = 0; i < Math.min(built.length, cs.raw.length); i++) {
      ['x', 'y', 'width', 'height', 'confidence'].forEach(function (k) { check(built[i][k] === cs.raw[i][k], cs.name + ': seq[' + i + '].' + k); });
    }
    out.seq_checked = (out.seq_checked || 0) + built.length;
  }
  const got = ht.ccv._group(seq, cs.min_neighbors);
  check(got.length === cs.grouped.length, cs.name + ': grouped count ' + got.length + ' != ' + cs.grouped.length);
  for (let i = 0; i < Math.min(got.length, cs.grouped.length); i++) {
    ['x', 'y', 'width', 'height', 'confidence', 'neighbors'].forEach(function (k) {
      check(got[i][k] === cs.grouped[i][k], cs.name + ': grouped[' + i + '].' + k);
    });
  }
});
/* host post-processing (SURVEY.md §8f) against the reference-JS vectors: Smoother and headposition are pure math */
const post = JSON.parse(fs.readFileSync(process.argv[3] || path.join(root, 'tests', 'golden', 'post.json'), 'utf8'));
post.cases.forEach(function (cs) {
  if (cs.kind === 'smoother') {
    const sm = new ht.Smoother(cs.alpha, cs.interval);
    const first = cs.calls.findIndex(function (c) { return c !== null; });
    cs.positions.forEach(function (p, i) {
      const pos = { x: p[0], y: p[1], z: p[2], width: p[3], height: p[4] };
      if (!sm.initialized && i >= first) sm.init(pos);
      const r = sm.smooth(pos);
      if (cs.calls[i] === null) check(r === false, cs.name + ': call ' + i + ' should return false');
      else check(r !== false && [r.x, r.y, r.z, r.width, r.height].every(function (v, k) { return v === cs.calls[i][k]; }), cs.name + ': call ' + i);
    });
  } else if (cs.kind === 'headposition') {
    const f0 = cs.faces[0];
    const hp = new ht.headposition.Tracker({ x: f0[0], y: f0[1], width: f0[2], height: f0[3] }, cs.camw, cs.camh, Object.assign({}, cs.params));
    check(hp.getFOV() === cs.fov, cs.name + ': fov ' + hp.getFOV() + ' vs ' + cs.fov);
    cs.faces.forEach(function (f, i) {
      const r = hp.track({ x: f[0], y: f[1], width: f[2], height: f[3] });
      check(r.x === cs.calls[i][0] && r.y === cs.calls[i][1] && r.z === cs.calls[i][2], cs.name + ': track ' + i + ' ' + JSON.stringify([r.x, r.y, r.z]) + ' vs ' + JSON.stringify(cs.calls[i]));
    });
  }
});
['Smoother', 'Tracker'].forEach(function (k) { check(typeof ht[k] === 'function', 'missing ' + k); });
check(typeof ht.headposition.Tracker === 'function' && typeof ht.headposition.TrackObj === 'function', 'missing headposition');

/* the addon must load and expose the C-ABI wrappers even without a GPU */
try {
  const addon = require(path.join(root, 'headtrackr_amd', 'js', 'headtrackr_hip.node'));
  ['createContext', 'detect', 'detectAsync', 'grayscale', 'whitebalance', 'camshiftInit', 'camshiftTrack', 'deviceCount', 'allgatherBest'].forEach(function (k) {
    check(typeof addon[k] === 'function', 'addon.' + k);
  });
  /* the pipelined path: every C-ABI export the throughput numbers are made of has a JS name (INTEGRATION.md lists the pairs) */
  ['hostAlloc', 'hostFree', 'deviceAlloc', 'deviceFree', 'deviceUpload', 'upload', 'bindDevice', 'uploadAsync', 'swapFrames', 'detectEnqueue', 'detectCollect', 'collectBest',
    'detectWhitebalance', 'whitebalanceBound', 'camshiftReserve', 'camshiftInitBound', 'camshiftTrackBound', 'camshiftTrackCollect', 'camshiftTrackSequence',
    'camshiftSequenceCollect', 'framesBound', 'framesEnqueued', 'graphLaunches', 'setGeometry', 'info', 'destroy'].forEach(function (k) {
    check(typeof addon[k] === 'function', 'addon.' + k);
  });
  check(addon.DETECT_WHITEBALANCE === 32 && addon.INPUT_GRAY_IN_R === 1, 'addon flag constants');
  check(typeof ht.ccv.DeviceBatch === 'function' && typeof ht.hostAlloc === 'function' && typeof ht.ccv.detect_objects_batch === 'function', 'batch entry points of the facade');
  out.abi = addon.abiVersion;
} catch (e) { check(false, 'addon load: ' + e.message); }
console.log(JSON.stringify(out));
