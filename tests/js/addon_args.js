'use strict';
/* CPU-side robustness of the PRODUCT addon's argument handling (csrc/ht_napi.cc), no GPU: every export is called with too few and with
 * wrong arguments (numbers, strings, plain objects, typed arrays of the wrong kind, functions) — each call must end in a JavaScript
 * exception or a return value, never in a crash, and apart from destroy() (idempotent by contract) and deviceCount() no entry point may
 * swallow a malformed call silently.  Without a GPU createContext itself must fail loudly.  Prints one JSON line. */
const path = require('path');
const A = require(path.join(__dirname, '..', '..', 'headtrackr_amd', 'js', 'headtrackr_hip.node'));
const bad = [undefined, null, 0, -1, 1e30, NaN, 'x', {}, [], new Uint8Array(4), new Int32Array(4), new Float64Array(2), function () {}, true];
const out = { ok: true, errors: [], calls: 0, threw: 0, silent: {} };
function check(c, msg) { if (!c) { out.ok = false; if (out.errors.length < 20) out.errors.push(msg); } }
for (const name of Object.keys(A)) {
  if (typeof A[name] !== 'function' || name === 'exitNow') continue; /* exitNow leaves the process by design */
  for (let argc = 0; argc <= 8; argc++) {
    for (let trial = 0; trial < (argc === 0 ? 1 : 40); trial++) {
      const args = [];
      for (let i = 0; i < argc; i++) args.push(bad[(trial * 7 + i * 3 + argc) % bad.length]);
      out.calls++;
      try {
        const r = A[name].apply(null, args);
        if (r && typeof r.then === 'function') r.catch(function () {});
        out.silent[name] = (out.silent[name] || 0) + 1;
      } catch (e) {
        out.threw++;
        check(e instanceof Error && typeof e.message === 'string' && e.message.length > 0, name + ': exception without a message');
      }
    }
  }
}
Object.keys(out.silent).forEach(function (k) { check(k === 'destroy' || k === 'deviceCount', k + ' accepted malformed arguments silently (' + out.silent[k] + ' calls)'); });
console.log(JSON.stringify(out));
