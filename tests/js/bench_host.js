'use strict';
/* The JavaScript host measured (north_star: "Host code stays JavaScript (Node) calling HIP through a thin C-ABI N-API addon").
 * Driven by bench.py (js_host sub-record) and by tests/test_js_host.py:
 *     node tests/js/bench_host.js <seconds> <c2_frames.raw> <n> <track_frames.raw> <nt>     -> one JSON line
 *   c2_frames.raw     n RGBA frames of 320x240 (the C2 mix): detect throughput from Node
 *       batch_host    ccv.detect_objects_batch on host frames (pageable Uint8Array -> PCIe every call), full result lists
 *       batch_device  ccv.DeviceBatch.detectBest: frames resident in HBM, 2 batches in flight (enqueue / collect-best / re-enqueue),
 *                     best face per frame — the loop bench.py's headline times, from JavaScript
 *   track_frames.raw  nt RGBA frames of 320x240 with one drifting face: per-call latency of the DROP-IN
 *                     facetrackr.Tracker.track() (facetrackr.js:67-126; the author's budget is ~15 ms per step, main.js:51,163):
 *                     a fresh tracker every 30 frames (main.js:230-238 does that on "lost"), so both the VJ step (+ initTracker)
 *                     and the CS steps are sampled; the unmodified reference JS (oracle/_ref) runs the same calls on the same frames.
 */
const fs = require('fs');
const path = require('path');
const root = path.join(__dirname, '..', '..');
const headtrackr = require(path.join(root, 'headtrackr_amd', 'js', 'headtrackr.js'));
const { Canvas } = require(path.join(root, 'headtrackr_amd', 'js', 'canvas.js'));

const seconds = +process.argv[2] || 2, c2file = process.argv[3], n = +process.argv[4], tfile = process.argv[5], nt = +process.argv[6];
const W = 320, H = 240, fbytes = W * H * 4;
const now = function () { return Number(process.hrtime.bigint()) / 1e6; };
const pct = function (v, q) { if (!v.length) return null; const s = v.slice().sort(function (a, b) { return a - b; }); return +s[Math.min(s.length - 1, Math.floor(q / 100 * s.length))].toFixed(4); };
const out = { node: process.version, cpus: require('os').cpus().length, cpu_model: require('os').cpus()[0].model, width: W, height: H };

(async function () {
  /* ---- detect throughput ---------------------------------------------------------------------------------------------------- */
  const frames = new Uint8Array(fs.readFileSync(c2file).buffer.slice(0, n * fbytes));
  {
    await headtrackr.ccv.detect_objects_batch(frames, n, W, H); /* warm-up: context, geometry, JIT */
    let calls = 0, faces = 0;
    const t0 = now();
    while (calls < 2 || now() - t0 < seconds * 250) {
      const r = await headtrackr.ccv.detect_objects_batch(frames, n, W, H);
      faces = r.reduce(function (s, x) { return s + (x.length ? 1 : 0); }, 0);
      calls++;
    }
    const dt = (now() - t0) / 1e3;
    out.batch_host = { frames_per_s: +(calls * n / dt).toFixed(1), ms_per_batch: +(dt / calls * 1e3).toFixed(3), batch: n, calls: calls, frames_with_faces: faces,
      what: 'ccv.detect_objects_batch(frames, n, w, h): pageable host frames cross PCIe every call; full grouped rect lists in JS' };
  }
  {
    const b = new headtrackr.ccv.DeviceBatch(W, H, n, process.env.HT_JS_DEPTH ? { depth: +process.env.HT_JS_DEPTH } : {});
    b.upload(frames);
    b.detectBest(12); /* warm-up */
    let batches = 0, r = null;
    const t0 = now();
    while (batches < 24 || now() - t0 < seconds * 500) { r = b.detectBest(48); batches += 48; }
    const dt = (now() - t0) / 1e3;
    let faces = 0;
    for (let f = 0; f < n; f++) if (r.best[6 * f + 5] > 0) faces++;
    out.batch_device = { frames_per_s: +(batches * n / dt).toFixed(1), ms_per_batch: +(dt / batches * 1e3).toFixed(4), batch: n, batches: batches, in_flight: b.depth,
      frames_with_faces: faces, hits_last_batch: r.hits,
      what: 'ccv.DeviceBatch.detectBest: frames resident in HBM, detectEnqueue + collectBest(requeue) over ' + b.depth + ' contexts; best face per frame' };
    b.destroy();
  }

  /* ---- facetrackr.Tracker.track() latency ------------------------------------------------------------------------------------- */
  const tbuf = fs.readFileSync(tfile);
  const tframe = function (i) { return tbuf.subarray((i % nt) * fbytes, (i % nt + 1) * fbytes); };
  const runTracker = function (ht, makeCanvas, setFrame, budgetMs, minCalls) {
    const vj = [], cs = [];
    let canvas = makeCanvas(), ft = null, i = 0, last = null, mark = null;
    const t0 = now();
    while (i < minCalls || now() - t0 < budgetMs) {
      if (i % 30 === 0) { if (ft && ft.release) ft.release(); ft = new ht.facetrackr.Tracker({ whitebalancing: false, calcAngles: true, sendEvents: false }); ft.init(canvas); }
      setFrame(canvas, tframe(i));
      const a = now();
      ft.track();
      const d = now() - a;
      last = ft.getTrackingObject();
      (last.detection === 'VJ' ? vj : cs).push(d);
      if (i === 59) mark = [last.x, last.y, last.width, last.height, last.detection]; /* both hosts reach call 59: same frames, same state machine */
      i++;
    }
    if (ft && ft.release) ft.release();
    return { calls: i, vj_calls: vj.length, cs_calls: cs.length, vj_p50_ms: pct(vj, 50), vj_max_ms: pct(vj, 100), cs_p50_ms: pct(cs, 50), cs_p99_ms: pct(cs, 99),
      p50_ms: pct(vj.concat(cs), 50), p99_ms: pct(vj.concat(cs), 99), last: [last.x, last.y, last.width, last.height, last.detection], after_60_calls: mark };
  };
  runTracker(headtrackr, function () { return new Canvas(W, H); }, function (c, f) { c.setFrame(f); }, 0, 61); /* warm-up */
  out.tracker = runTracker(headtrackr, function () { return new Canvas(W, H); }, function (c, f) { c.setFrame(f); }, seconds * 250, 120);
  out.tracker.what = 'drop-in facetrackr.Tracker({whitebalancing:false, calcAngles:true}).track() on a 320x240 canvas, fresh tracker every 30 frames: VJ step = upload + detect + initTracker, CS step = upload + camshift; frame uploaded once per call';

  /* the unmodified reference on the same frames (oracle/_ref, built by `make -C oracle _ref` where /root/reference exists) */
  const gz = path.join(root, 'oracle', '_ref', 'headtrackr_ref.js.gz');
  if (fs.existsSync(gz)) {
    const shim = require(path.join(root, 'oracle', 'canvas_shim.js'));
    const Module = require('module');
    global.document = shim.makeDocument();
    global.window = global;
    const m = new Module('headtrackr_ref', null);
    m.paths = [];
    m._compile(require('zlib').gunzipSync(fs.readFileSync(gz)).toString('utf8'), 'headtrackr_ref.js');
    const ref = m.exports;
    const mk = function () { return new shim.Canvas(W, H); };
    const set = function (c, f) { c.loadRGBA(Buffer.from(f)); };
    runTracker(ref, mk, set, 0, 31);
    out.tracker_reference_js = runTracker(ref, mk, set, seconds * 250, 60);
    out.tracker_reference_js.what = 'unmodified reference JS, same calls on the same frames, single-threaded Node on oracle/canvas_shim.js';
    out.tracker.same_result_as_reference = JSON.stringify(out.tracker.after_60_calls) === JSON.stringify(out.tracker_reference_js.after_60_calls);
    out.tracker.vs_reference_p50 = +(out.tracker_reference_js.p50_ms / out.tracker.p50_ms).toFixed(1);
  }
  process.stdout.write(JSON.stringify(out) + '\n', function () { headtrackr.exitNow(0); }); /* see ht_napi.cc exitNow: no runtime teardown */
})().catch(function (e) { console.log(JSON.stringify({ error: String(e && e.stack || e) })); process.exit(1); });
