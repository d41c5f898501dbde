'use strict';
/* tools/pack_cascade.js — derive headtrackr_amd/data/cascade.bin (HTCB blob of the trained face cascade)
 * from the reference bundle's `headtrackr.cascade` data object (/root/reference/src/cascade.js:19).
 * Usage: node tools/pack_cascade.js [/root/reference/headtrackr.js] [out.bin]
 * The output is model DATA (classifier weights) in this repo's own packed format; no reference code is copied. */
const path = require('path');
const fs = require('fs');
const { packCascade, unpackCascade } = require('../headtrackr_amd/js/cascade_pack.js');
const refPath = process.argv[2] || '/root/reference/headtrackr.js';
const out = process.argv[3] || path.join(__dirname, '..', 'headtrackr_amd', 'data', 'cascade.bin');
const ref = require(refPath);
const blob = packCascade(ref.cascade);
/* round-trip check against the source object */
const rt = unpackCascade(blob);
const norm = function (c) {
  return JSON.stringify({ count: c.count, width: c.width, height: c.height, s: c.stage_classifier.map(function (s) {
    return { count: s.count, threshold: s.threshold, alpha: s.alpha, feature: s.feature.map(function (f) {
      const z = function (a, zz) { return a.map(function (v, i) { return zz[i] >= 0 ? v : 0; }); };
      return { size: f.size, px: z(f.px, f.pz), py: z(f.py, f.pz), pz: f.pz, nx: z(f.nx, f.nz), ny: z(f.ny, f.nz), nz: f.nz }; }) }; }) });
};
if (norm(rt) !== norm(ref.cascade)) throw new Error('cascade round-trip mismatch');
fs.writeFileSync(out, blob);
console.log('wrote', out, blob.length, 'bytes;', rt.count, 'stages');
