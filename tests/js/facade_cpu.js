'use strict';
/* CPU-side checks of the JavaScript facade (no GPU): exports, cascade round trip, and the host-side grouping
 * (ccv.array_group + averaging + nesting filter) against the reference-JS golden vectors.  Prints one JSON line. */
const path = require('path');
const fs = require('fs');
const root = path.join(__dirname, '..', '..');
const ht = require(path.join(root, 'headtrackr_amd', 'js', 'headtrackr.js'));
const golden = JSON.parse(fs.readFileSync(path.join(root, 'tests', 'golden', 'detect.json'), 'utf8'));
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
  const got = ht.ccv._group(seq, cs.min_neighbors);
  check(got.length === cs.grouped.length, cs.name + ': grouped count ' + got.length + ' != ' + cs.grouped.length);
  for (let i = 0; i < Math.min(got.length, cs.grouped.length); i++) {
    ['x', 'y', 'width', 'height', 'confidence', 'neighbors'].forEach(function (k) {
      check(got[i][k] === cs.grouped[i][k], cs.name + ': grouped[' + i + '].' + k);
    });
  }
});
/* the addon must load and expose the C-ABI wrappers even without a GPU */
try {
  const addon = require(path.join(root, 'headtrackr_amd', 'js', 'headtrackr_hip.node'));
  ['createContext', 'detect', 'detectAsync', 'grayscale', 'whitebalance', 'camshiftInit', 'camshiftTrack'].forEach(function (k) {
    check(typeof addon[k] === 'function', 'addon.' + k);
  });
  out.abi = addon.abiVersion;
} catch (e) { check(false, 'addon load: ' + e.message); }
console.log(JSON.stringify(out));
