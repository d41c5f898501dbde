'use strict';
/*
 * headtrackr.js (MI355X edition) — JavaScript host of the HIP detect / track hot path.
 *
 * Exposes the reference's per-frame API under the reference's names so that code written against
 * auduno/headtrackr's `headtrackr.ccv`, `headtrackr.cascade`, `headtrackr.camshift`, `headtrackr.facetrackr` and
 * `headtrackr.getWhitebalance` runs unchanged, with the pixel work done by libheadtrackr_hip.so through the N-API addon
 * (headtrackr_hip.node, C ABI in include/headtrackr_hip.h).  Reference citations are /root/reference/src/<file>:<line>.
 *
 *   ccv.grayscale(canvas)                               ccv.js:22-32     -> ht_grayscale_batch
 *   ccv.detect_objects(canvas,cascade,interval,min_nb)  ccv.js:109-333   -> ht_detect_batch (pyramid + cascade scan);
 *                                                                            seq construction + grouping stay in JS
 *   ccv.array_group(seq,gfunc)                          ccv.js:34-107    (host, O(n^2) on a few dozen rects)
 *   camshift.Tracker / Histogram / Moments / Rectangle / TrackObj          camshift.js:49-378 -> ht_camshift_*
 *   facetrackr.Tracker / TrackObj                       facetrackr.js:37-255  (state machine WB -> VJ -> CS)
 *   getWhitebalance(canvas)                             whitebalance.js:5-30 -> ht_whitebalance_batch
 * plus batch entry points that the single-frame browser API has no room for:
 *   ccv.detect_objects_batch(frames, n, w, h, cascade, interval, min_neighbors, opts)  -> Promise<Array<Array<rect>>>
 *   new ccv.DeviceBatch(w, h, n, opts)   frames resident in HBM, detect batches pipelined over several contexts (enqueue /
 *                                        collect-best / re-enqueue), fused whitebalance, camshift call sequences — the JS form
 *                                        of the loop bench.py times; every C-ABI export has a JS name (INTEGRATION.md)
 *   hostAlloc(bytes)                     pinned host memory for frames that cross PCIe every call
 *
 * "canvas" is anything with width, height and getContext('2d') -> {getImageData, putImageData, drawImage,
 * createImageData}; ./canvas.js provides one for Node.  Failure conventions are the reference's: empty arrays,
 * confidence -10000, width == height == 0 — plus exceptions only for misuse of the native layer (no GPU, bad cascade).
 */
const path = require('path');
const fs = require('fs');
const pack = require('./cascade_pack.js');

let native = null;
function addon() {
  if (!native) {
    try {
      native = require('./headtrackr_hip.node');
    } catch (e) {
      throw new Error('headtrackr_hip.node could not be loaded (' + e.message + '); build it with `python -m headtrackr_amd.build`. ' +
        'There is no JavaScript fallback for the detection / tracking kernels.');
    }
  }
  return native;
}

const headtrackr = {};
headtrackr.rev = 2; /* main.js:13 */

/* ---- cascade ---------------------------------------------------------------------------------------------------- */

let builtinCascade = null;
Object.defineProperty(headtrackr, 'cascade', { /* cascade.js:19: the trained face cascade, rebuilt from data/cascade.bin */
  enumerable: true,
  get: function () {
    if (!builtinCascade) builtinCascade = pack.unpackCascade(fs.readFileSync(path.join(__dirname, '..', 'data', 'cascade.bin')));
    return builtinCascade;
  }
});

/* one native context per (cascade object, interval, GPU); contexts own device memory, so they are cached */
const contexts = new WeakMap();
function contextFor(cascade, interval, device) {
  if (device === undefined) device = headtrackr.device | 0;
  let perCascade = contexts.get(cascade);
  if (!perCascade) { perCascade = new Map(); contexts.set(cascade, perCascade); }
  const key = interval + '@' + device;
  let c = perCascade.get(key);
  if (!c) {
    c = { handle: addon().createContext({ cascade: pack.packCascade(cascade), interval: interval, device: device }), w: 0, h: 0, batch: 0, device: device };
    perCascade.set(key, c);
  }
  return c;
}
headtrackr.device = 0; /* HIP device ordinal used by the single-frame (drop-in) entry points */
headtrackr.deviceCount = function () { return addon().deviceCount(); }; /* GPUs the `devices` option of the batch entry points can name */

/* pyramid level sizes exactly as ccv.js:110-127 computes them (Math.pow / Math.floor in V8), handed to the native
 * side so that no libm difference can move a level boundary */
function levelDims(w, h, cascade, interval) {
  const scale = Math.pow(2, 1 / (interval + 1));
  const next = interval + 1;
  const upto = Math.floor(Math.log(Math.min(cascade.width, cascade.height)) / Math.log(scale));
  const n = upto + next * 2;
  const d = new Int32Array(2 * n);
  d[0] = w; d[1] = h;
  for (let i = 1; i <= interval; i++) {
    d[2 * i] = Math.floor(w / Math.pow(scale, i));
    d[2 * i + 1] = Math.floor(h / Math.pow(scale, i));
  }
  for (let i = next; i < n; i++) {
    d[2 * i] = Math.floor(d[2 * (i - next)] / 2);
    d[2 * i + 1] = Math.floor(d[2 * (i - next) + 1] / 2);
  }
  return d;
}

function ensureGeometry(c, w, h, batch, cascade, interval) {
  if (c.w !== w || c.h !== h || c.batch < batch) {
    addon().setGeometry(c.handle, w, h, batch, levelDims(w, h, cascade, interval));
    c.w = w; c.h = h; c.batch = batch;
    c.boundImg = null;
  }
}

/* One frame, several consumers: facetrackr.Tracker.track() reads its canvas once and the state machine may run more than one
 * device routine on that frame (VJ detection followed by camshift.initTracker, facetrackr.js:97-108).  The ImageData object is
 * uploaded once (ht_upload_frames) and stays bound; the *Bound addon entry points then work on it.  Every other entry point that
 * binds frames of its own drops the marker. */
function bindFrame(c, img, cascade, interval) {
  if (c.boundImg === img) return;
  ensureGeometry(c, img.width, img.height, 1, cascade, interval);
  addon().upload(c.handle, img.data, 1, img.width, img.height);
  c.boundImg = img;
}
/* The marker lives for ONE public call: a host canvas may hand out the same ImageData object again with refreshed pixels (zero-copy
 * video wrappers do), so object identity says nothing across calls — every public entry point that used bindFrame drops it on return. */
function unbindFrames() {
  const per = contexts.get(headtrackr.cascade);
  if (per) per.forEach(function (c) { c.boundImg = null; });
}
headtrackr.hostAlloc = function (bytes) { return addon().hostAlloc(bytes); }; /* Uint8Array over pinned host memory */
/* leave the process NOW: live contexts are destroyed, stdout / stderr flushed, then _exit(code) — no runtime teardown (see ht_napi.cc) */
headtrackr.exitNow = function (code) { addon().exitNow(code | 0); };
headtrackr.hostFree = function (arr) { addon().hostFree(arr); }; /* explicit: the addon's handles carry no GC finalizers (see ht_napi.cc) */

/* ---- ccv ------------------------------------------------------------------------------------------------------------ */

headtrackr.ccv = {};

headtrackr.ccv.grayscale = function (canvas) { /* ccv.js:22-32, in place, returns the canvas */
  const ctx = canvas.getContext('2d');
  const img = ctx.getImageData(0, 0, canvas.width, canvas.height);
  if (canvas.width > 0 && canvas.height > 0) {
    const c = contextFor(headtrackr.cascade, 5);
    addon().grayscale(c.handle, img.data, 1, canvas.width, canvas.height);
    c.boundImg = null;
  }
  ctx.putImageData(img, 0, 0);
  return canvas;
};

/* union-find grouping with rank + path compression; class numbers in first-seen order (ccv.js:34-107) */
headtrackr.ccv.array_group = function (seq, gfunc) {
  const n = seq.length;
  const parent = new Int32Array(n).fill(-1), rank = new Int32Array(n);
  const rootOf = function (i) { while (parent[i] !== -1) i = parent[i]; return i; };
  const compress = function (i, root) { while (parent[i] !== -1) { const t = i; i = parent[i]; parent[t] = root; } };
  for (let i = 0; i < n; i++) {
    if (!seq[i]) continue;
    let root = rootOf(i);
    for (let j = 0; j < n; j++) {
      if (i === j || !seq[j] || !gfunc(seq[i], seq[j])) continue;
      const root2 = rootOf(j);
      if (root2 === root) continue;
      if (rank[root] > rank[root2]) {
        parent[root2] = root;
      } else {
        parent[root] = root2;
        if (rank[root] === rank[root2]) rank[root2]++;
        root = root2;
      }
      compress(j, root);
      compress(i, root);
    }
  }
  const index = new Array(n);
  let cat = 0;
  for (let i = 0; i < n; i++) {
    let j = -1;
    if (seq[i]) {
      const r = rootOf(i);
      if (rank[r] >= 0) rank[r] = ~cat++;
      j = ~rank[r];
    }
    index[i] = j;
  }
  return { index: index, cat: cat };
};

/* raw hits (index form) -> the reference's `seq` (ccv.js:227-234), scale_x by repeated multiplication (ccv.js:244-245) */
function hitsToSeq(hits, from, to, cascade, interval) {
  const scale = Math.pow(2, 1 / (interval + 1));
  const s = [1];
  const seq = [];
  for (let k = from; k < to; k++) {
    const i = hits.scale[k], q = hits.q[k];
    while (s.length <= i) s.push(s[s.length - 1] * scale);
    seq.push({ x: (hits.x[k] * 4 + (q & 1) * 2) * s[i], y: (hits.y[k] * 4 + (q >> 1) * 2) * s[i],
      width: cascade.width * s[i], height: cascade.height * s[i], neighbor: 1, confidence: hits.sum[k] });
  }
  return seq;
}

/* ccv.js:249-332: grouping, per-class mean + 0.5, nested-rectangle filter */
function groupSeq(seq, min_neighbors) {
  if (!(min_neighbors > 0)) return seq;
  const result = headtrackr.ccv.array_group(seq, function (r1, r2) {
    const distance = Math.floor(r1.width * 0.25 + 0.5);
    return r2.x <= r1.x + distance && r2.x >= r1.x - distance && r2.y <= r1.y + distance && r2.y >= r1.y - distance &&
      r2.width <= Math.floor(r1.width * 1.5 + 0.5) && Math.floor(r2.width * 1.5 + 0.5) >= r1.width;
  });
  const comps = [];
  for (let i = 0; i <= result.cat; i++) comps.push({ neighbors: 0, x: 0, y: 0, width: 0, height: 0, confidence: 0 });
  for (let i = 0; i < seq.length; i++) {
    const r = seq[i], c = comps[result.index[i]];
    if (c.neighbors === 0) c.confidence = r.confidence;
    ++c.neighbors;
    c.x += r.x; c.y += r.y; c.width += r.width; c.height += r.height;
    c.confidence = Math.max(c.confidence, r.confidence);
  }
  const seq2 = [];
  for (let i = 0; i < result.cat; i++) {
    const c = comps[i], n = c.neighbors;
    if (n >= min_neighbors) {
      seq2.push({ x: (c.x * 2 + n) / (2 * n), y: (c.y * 2 + n) / (2 * n), width: (c.width * 2 + n) / (2 * n),
        height: (c.height * 2 + n) / (2 * n), neighbors: n, confidence: c.confidence });
    }
  }
  const out = [];
  for (let i = 0; i < seq2.length; i++) {
    const r1 = seq2[i];
    let keep = true;
    for (let j = 0; j < seq2.length && keep; j++) {
      const r2 = seq2[j], distance = Math.floor(r2.width * 0.25 + 0.5);
      if (i !== j && r1.x >= r2.x - distance && r1.y >= r2.y - distance && r1.x + r1.width <= r2.x + r2.width + distance &&
          r1.y + r1.height <= r2.y + r2.height + distance && (r2.neighbors > Math.max(3, r1.neighbors) || r1.neighbors < 3)) keep = false;
    }
    if (keep) out.push(r1);
  }
  return out;
}

headtrackr.ccv._group = groupSeq; /* exposed for tests */
headtrackr.ccv._hitsToSeq = hitsToSeq; /* likewise */

/* ccv.js:109: `canvas` is expected to be gray already (byte 0 of each pixel is what the detector reads) */
headtrackr.ccv.detect_objects = function (canvas, cascade, interval, min_neighbors) {
  const w = canvas.width, h = canvas.height;
  const img = canvas.getContext('2d').getImageData(0, 0, w, h);
  canvas.data = img.data; /* the reference hangs the pixels on the caller's canvas (ccv.js:115) */
  if (!(w > 0 && h > 0)) return [];
  const c = contextFor(cascade, interval);
  ensureGeometry(c, w, h, 1, cascade, interval);
  const hits = addon().detect(c.handle, img.data, 1, w, h, addon().INPUT_GRAY_IN_R);
  c.boundImg = null;
  return groupSeq(hitsToSeq(hits, 0, hits.sum.length, cascade, interval), min_neighbors);
};

/* fused colour path used by facetrackr: ccv.grayscale + ccv.detect_objects without the intermediate canvas copy */
headtrackr.ccv.detect_objects_rgba = function (rgba, w, h, cascade, interval, min_neighbors) {
  if (!(w > 0 && h > 0)) return [];
  const c = contextFor(cascade, interval);
  ensureGeometry(c, w, h, 1, cascade, interval);
  const hits = addon().detect(c.handle, rgba, 1, w, h, addon().INPUT_RGBA);
  c.boundImg = null;
  return groupSeq(hitsToSeq(hits, 0, hits.sum.length, cascade, interval), min_neighbors);
};

/* the same on an ImageData that bindFrame() has (or will have) put on the device: no second upload (facetrackr's VJ step) */
function detectBoundImg(img, cascade, interval, min_neighbors) {
  const c = contextFor(cascade, interval);
  bindFrame(c, img, cascade, interval);
  addon().detectEnqueue(c.handle, addon().INPUT_RGBA);
  const hits = addon().detectCollect(c.handle);
  return groupSeq(hitsToSeq(hits, 0, hits.sum.length, cascade, interval), min_neighbors);
}

/* contiguous block of frames owned by `rank` (sizes differ by at most one) — the sharding of BASELINE.json configs[3] */
function shardRange(total, rank, world) {
  const base = Math.floor(total / world), rem = total % world;
  const start = rank * base + Math.min(rank, rem);
  return [start, start + base + (rank < rem ? 1 : 0)];
}

/* facetrackr's choice among the grouped rects of one frame (facetrackr.js:157-165: strict '>', first maximum wins) */
function bestOf(rects) {
  let best;
  for (let i = 0; i < rects.length; i++) if (best === undefined || rects[i].confidence > best.confidence) best = rects[i];
  return best;
}

/* n RGBA frames (one Uint8Array of n*w*h*4 bytes) -> Promise of n result lists; runs on the libuv pool.
 * opts.devices = [0, 1, ...]: the frames are block-sharded over these GPUs (one native context and one pool job per GPU, all in
 * flight together); opts.gather: after the detection every GPU's best-face rect per frame (facetrackr.js:147-175) is
 * all-gathered over RCCL / xGMI so that every GPU holds the whole batch's bounding boxes; the gathered table comes back as
 * result.best = [{x, y, width, height, confidence, neighbors} | null per frame]. */
headtrackr.ccv.detect_objects_batch = function (frames, n, w, h, cascade, interval, min_neighbors, opts) {
  cascade = cascade || headtrackr.cascade;
  interval = interval === undefined ? 5 : interval;
  min_neighbors = min_neighbors === undefined ? 1 : min_neighbors;
  opts = opts || {};
  const devices = (opts.devices && opts.devices.length) ? opts.devices : [headtrackr.device | 0];
  const world = Math.min(devices.length, n);
  const fbytes = w * h * 4;
  const jobs = [];
  for (let r = 0; r < world; r++) {
    const span = shardRange(n, r, world), cnt = span[1] - span[0];
    const c = contextFor(cascade, interval, devices[r]);
    ensureGeometry(c, w, h, cnt, cascade, interval);
    c.boundImg = null;
    const view = frames.subarray(span[0] * fbytes, span[1] * fbytes);
    jobs.push(addon().detectAsync(c.handle, view, cnt, w, h, addon().INPUT_RGBA).then(function (hits) {
      const out = [];
      let k = 0;
      for (let f = 0; f < cnt; f++) {
        out.push(groupSeq(hitsToSeq(hits, k, k + hits.counts[f], cascade, interval), min_neighbors));
        k += hits.counts[f];
      }
      return { ctx: c, rects: out };
    }));
  }
  return Promise.all(jobs).then(function (parts) {
    let all = [];
    parts.forEach(function (p) { all = all.concat(p.rects); });
    if (opts.gather) {
      const per = Math.ceil(n / world);
      const recs = parts.map(function (p) {
        const a = new Float64Array(6 * per); /* padding rows stay zero */
        p.rects.forEach(function (rects, f) {
          const b = bestOf(rects);
          if (b) a.set([b.x, b.y, b.width, b.height, b.confidence, b.neighbors === undefined ? 1 : b.neighbors], 6 * f);
          else a[6 * f + 4] = -10000; /* facetrackr.js:239 */
        });
        return a;
      });
      const g = addon().allgatherBest(parts.map(function (p) { return p.ctx.handle; }), recs, per);
      const best = [];
      for (let r = 0; r < world; r++) {
        const span = shardRange(n, r, world);
        for (let f = 0; f < span[1] - span[0]; f++) {
          const o = 6 * (r * per + f);
          best.push(g[o + 5] > 0 ? { x: g[o], y: g[o + 1], width: g[o + 2], height: g[o + 3], confidence: g[o + 4], neighbors: g[o + 5] } : null);
        }
      }
      all.best = best;
    }
    return all;
  });
};

/* ---- device-resident batches: the pipelined path -------------------------------------------------------------------------
 * new ccv.DeviceBatch(w, h, n, {cascade, interval, device, depth, sets}):
 *   `sets` frame sets of n RGBA frames each live in ONE device buffer (HBM); `depth` native contexts (own HIP streams, own pyramid
 *   arenas) take detect batches in turn so that `depth` batches are in flight while the host groups the previous one
 *   (ht_detect_enqueue + ht_detect_collect_best_requeue: the C2 / C4 loop of bench.py, from JavaScript).
 *     upload(frames, set = 0)              host -> HBM once (frames: Uint8Array of n*w*h*4 bytes)
 *     detectBest(batches, min_neighbors, set, flags) -> {best: Float64Array(6 n) [x,y,width,height,confidence,neighbors] of the
 *                                           last batch, hits, batches}; neighbors 0 / confidence -10000 = no face (facetrackr.js:239)
 *     detect(min_neighbors, set)           -> Array<Array<rect>>: exactly ccv.detect_objects' result per frame (parity path)
 *     whitebalance(set)                    -> Float64Array(n): getWhitebalance per frame, fused into a detect batch's gray pass
 *     detectStep(set) (= detectStepEnqueue + detectStepFinish) / trackStep(set) / trackEnqueue(set) + trackCollect() / ingest(pinned) / swap()   K frame-synchronous live feeds, one time step per call (below)
 *     initTrackers(rects, set) / trackSequence(sets[], calcAngles, outAll) -> Float64Array(9 n [* calls]): n camshift streams,
 *                                           one track() per listed frame set, ONE host call (ht_camshift_track_sequence)
 *     destroy() */
headtrackr.ccv.DeviceBatch = function (w, h, n, opts) {
  opts = opts || {};
  const cascade = opts.cascade || headtrackr.cascade, interval = opts.interval === undefined ? 5 : opts.interval;
  const device = opts.device === undefined ? (headtrackr.device | 0) : opts.device;
  /* batches in flight: 2.  (The Python host gains 6 % from a third batch at 256 x 320x240; this loop, whose calls drain the pipeline every
   * `batches` batches, loses 12 %: 1.03 M frames/s at 2, 0.90 M at 3 with 48-batch calls — pass {depth: 3} for long calls.) */
  const depth = Math.max(1, opts.depth || 2), sets = Math.max(1, opts.sets || 1);
  const A = addon(), fbytes = w * h * 4, setBytes = n * fbytes;
  const blob = pack.packCascade(cascade), dims = levelDims(w, h, cascade, interval);
  const ctxs = [];
  for (let i = 0; i < depth; i++) {
    const hnd = A.createContext({ cascade: blob, interval: interval, device: device });
    A.setGeometry(hnd, w, h, n, dims);
    ctxs.push(hnd);
  }
  const dev = A.deviceAlloc(ctxs[0], sets * setBytes);
  let bound = -1, trackers = false;
  const bind = function (set) { if (bound !== set) { ctxs.forEach(function (c) { A.bindDevice(c, dev, set * setBytes, n, fbytes); }); bound = set; } };
  this.width = w; this.height = h; this.frames = n; this.depth = depth;
  this.upload = function (frames, set) {
    if (frames.length < setBytes) throw new RangeError('DeviceBatch.upload: need n*w*h*4 bytes');
    A.deviceUpload(ctxs[0], dev, (set || 0) * setBytes, frames.subarray(0, setBytes));
  };
  this.detectBest = function (batches, min_neighbors, set, flags) {
    if (!(batches >= 1)) throw new RangeError('DeviceBatch.detectBest: batches must be >= 1');
    bind(set || 0);
    flags = flags === undefined ? A.INPUT_RGBA : flags;
    min_neighbors = min_neighbors === undefined ? 1 : min_neighbors;
    let started = Math.min(depth, batches), r = null;
    for (let i = 0; i < started; i++) A.detectEnqueue(ctxs[i], flags);
    for (let i = 0; i < batches; i++) { /* collect batch i; its context re-enqueues inside the call while batches remain */
      const more = started < batches;
      r = A.collectBest(ctxs[i % depth], min_neighbors, more ? flags : -1);
      if (more) started++;
    }
    r.batches = batches;
    return r;
  };
  this.detect = function (min_neighbors, set) {
    bind(set || 0);
    A.detectEnqueue(ctxs[0], A.INPUT_RGBA);
    const hits = A.detectCollect(ctxs[0]), out = [];
    let k = 0;
    for (let f = 0; f < n; f++) { out.push(groupSeq(hitsToSeq(hits, k, k + hits.counts[f], cascade, interval), min_neighbors)); k += hits.counts[f]; }
    return out;
  };
  this.whitebalance = function (set) {
    bind(set || 0);
    A.detectEnqueue(ctxs[0], A.INPUT_RGBA | A.DETECT_WHITEBALANCE);
    A.collectBest(ctxs[0], 1, -1);
    return A.detectWhitebalance(ctxs[0], n);
  };
  this.initTrackers = function (rects, set) { /* rects: Int32Array [x, y, width, height] per stream (camshift.js:198-211) */
    bind(set || 0);
    if (!trackers) { A.camshiftReserve(ctxs[0], n); trackers = true; }
    A.camshiftInitBound(ctxs[0], n, 0, rects);
  };
  this.trackSequence = function (setList, calcAngles, outAll) {
    const offs = new Float64Array(setList.length);
    for (let k = 0; k < setList.length; k++) offs[k] = setList[k] * setBytes;
    return A.camshiftTrackSequence(ctxs[0], 0, n, calcAngles ? 1 : 0, dev, offs, fbytes, !!outAll, true);
  };
  /* K frame-synchronous feeds, one time step at a time (the K-feed form of the reference's loop, main.js:168-180 -> facetrackr.js:97-108,
   * 185-217): the n frames of `set` are the feeds' frames of this step.
   *   detectStep(set, min_neighbors) -> Float64Array(6 n) best face per feed (as detectBest) — and camshift.initTracker on its floor()ed
   *                                     rect (facetrackr.js:101-106); a feed without a face gets the centre half of the frame
   *   trackStep(set, calcAngles)     -> Float64Array(9 n): camshift.track per feed (enqueue-only launch + collect) */
  /*   ingest(pinned) / swap()        live ingest: the NEXT step's n frames cross PCIe from a hostAlloc() view on the copy stream while the
   *                                  current step is processed (ht_upload_frames_async / ht_swap_frames); after swap() pass set = -1 */
  const bind0 = function (set) { if (set >= 0) A.bindDevice(ctxs[0], dev, set * setBytes, n, fbytes); bound = -1; };
  this.ingest = function (pinned) { A.uploadAsync(ctxs[0], pinned, n); };
  this.swap = function () { A.swapFrames(ctxs[0]); bound = -1; };
  /*   detectStepEnqueue(set) / detectStepFinish(min_neighbors)   the two halves of detectStep: a streaming host enqueues the detect of
   *                                  step i right behind the track steps still in flight, collects THOSE (trackCollect), and only then
   *                                  waits for the best faces — the GPU does not idle while the host drains its pipeline */
  this.detectStepEnqueue = function (set) {
    bind0(set === undefined ? 0 : set);
    A.detectEnqueue(ctxs[0], A.INPUT_RGBA);
  };
  this.detectStep = function (set, min_neighbors) {
    this.detectStepEnqueue(set);
    return this.detectStepFinish(min_neighbors);
  };
  this.detectStepFinish = function (min_neighbors) {
    const r = A.collectBest(ctxs[0], min_neighbors === undefined ? 1 : min_neighbors, -1);
    const rects = new Int32Array(4 * n);
    for (let f = 0; f < n; f++) {
      const ok = r.best[6 * f + 5] > 0 && r.best[6 * f + 4] > -10;
      const v = ok ? [r.best[6 * f], r.best[6 * f + 1], r.best[6 * f + 2], r.best[6 * f + 3]] : [w >> 2, h >> 2, w >> 1, h >> 1];
      for (let k = 0; k < 4; k++) rects[4 * f + k] = Math.floor(v[k]);
    }
    if (!trackers) { A.camshiftReserve(ctxs[0], n); trackers = true; }
    A.camshiftInitBound(ctxs[0], n, 0, rects);
    r.rects = rects;
    return r;
  };
  this.trackStep = function (set, calcAngles) {
    bind0(set === undefined ? 0 : set);
    A.camshiftTrackBound(ctxs[0], n, 0, calcAngles === false ? 0 : 1, false);
    return A.camshiftTrackCollect(ctxs[0], n);
  };
  /*   trackEnqueue(set, calcAngles) / trackCollect()   the two halves of trackStep: up to 4 track steps may be outstanding (the library
   *                                  keeps their results in a ring of pinned slots; trackCollect returns the OLDEST), so a streaming host
   *                                  enqueues step i + 1 before it waits for step i — the search window that links them lives on the GPU */
  this.trackEnqueue = function (set, calcAngles) {
    bind0(set === undefined ? 0 : set);
    A.camshiftTrackBound(ctxs[0], n, 0, calcAngles === false ? 0 : 1, false);
  };
  this.trackCollect = function () { return A.camshiftTrackCollect(ctxs[0], n); };
  this.graphLaunches = function () { return ctxs.reduce(function (s, c) { return s + A.graphLaunches(c); }, 0); };
  /* the frame buffer is shared by all `depth` contexts: the others go first (ht_device_free refuses while they have it bound) */
  this.destroy = function () { for (let i = ctxs.length - 1; i >= 1; i--) A.destroy(ctxs[i]); A.deviceFree(ctxs[0], dev); A.destroy(ctxs[0]); ctxs.length = 0; };
};

/* ---- whitebalance ----------------------------------------------------------------------------------------------------- */

headtrackr.getWhitebalance = function (canvas) { /* whitebalance.js:5-30 */
  const img = canvas.getContext('2d').getImageData(0, 0, canvas.width, canvas.height);
  if (!(img.width > 0 && img.height > 0)) return NaN; /* 0/0 in the reference */
  const c = contextFor(headtrackr.cascade, 5);
  /* announce the size like every other entry point: a frame of another size would make the addon re-build the geometry on its own, with
   * level sizes from libm instead of V8's, behind the back of the cache in `c` (found by tests/js/parity_cpu.js) */
  ensureGeometry(c, img.width, img.height, 1, headtrackr.cascade, 5);
  c.boundImg = null;
  return addon().whitebalance(c.handle, img.data, 1, img.width, img.height)[0];
};

/* ---- camshift ----------------------------------------------------------------------------------------------------------- */

headtrackr.camshift = {};

headtrackr.camshift.Histogram = function (imgdata) { /* camshift.js:49-72 (host-side; used by the debug getters) */
  this.size = 4096;
  const bins = new Uint32Array(4096);
  for (let x = 0, il = imgdata.length; x < il; x += 4) bins[256 * (imgdata[x] >> 4) + 16 * (imgdata[x + 1] >> 4) + (imgdata[x + 2] >> 4)] += 1;
  this.getBin = function (index) { return bins[index]; };
};

headtrackr.camshift.Moments = function (data, x, y, w, h, second) { /* camshift.js:79-120 (host-side, for API completeness) */
  this.m00 = 0; this.m01 = 0; this.m10 = 0; this.m11 = 0; this.m02 = 0; this.m20 = 0;
  for (let i = x; i < w; i++) {
    const col = data[i], vx = i - x;
    for (let j = y; j < h; j++) {
      const val = col[j], vy = j - y;
      this.m00 += val; this.m01 += vy * val; this.m10 += vx * val;
      if (second) { this.m11 += vx * vy * val; this.m02 += vy * vy * val; this.m20 += vx * vx * val; }
    }
  }
  this.invM00 = 1 / this.m00;
  this.xc = this.m10 * this.invM00;
  this.yc = this.m01 * this.invM00;
  this.mu00 = this.m00; this.mu01 = 0; this.mu10 = 0;
  if (second) {
    this.mu20 = this.m20 - this.m10 * this.xc;
    this.mu02 = this.m02 - this.m01 * this.yc;
    this.mu11 = this.m11 - this.m01 * this.xc;
  }
};

headtrackr.camshift.Rectangle = function (x, y, w, h) { /* camshift.js:127-141 */
  this.x = x; this.y = y; this.width = w; this.height = h;
  this.clone = function () { return new headtrackr.camshift.Rectangle(this.x, this.y, this.width, this.height); };
};

headtrackr.camshift.TrackObj = function () { /* camshift.js:362-378 */
  this.height = 0; this.width = 0; this.angle = 0; this.x = 0; this.y = 0;
  this.clone = function () {
    const c = new headtrackr.camshift.TrackObj();
    c.height = this.height; c.width = this.width; c.angle = this.angle; c.x = this.x; c.y = this.y;
    return c;
  };
};

/* every camshift.Tracker owns one device-side stream slot of a shared context */
const csPool = { ctx: null, next: 0, free: [], reserved: 0 };
function csSlot() {
  if (!csPool.ctx) csPool.ctx = contextFor(headtrackr.cascade, 5);
  const slot = csPool.free.length ? csPool.free.pop() : csPool.next++;
  if (csPool.next > csPool.reserved) { /* grow geometrically: a reservation re-allocates and copies every tracker's state */
    csPool.reserved = Math.max(csPool.next, 2 * csPool.reserved, 4);
    addon().camshiftReserve(csPool.ctx.handle, csPool.reserved);
  }
  return slot;
}
headtrackr.camshift._pool = csPool; /* exposed for tests */

headtrackr.camshift.Tracker = function (params) { /* camshift.js:148-354 */
  if (params === undefined) params = {};
  if (params.calcAngles === undefined) params.calcAngles = true;
  const slot = csSlot();
  let searchWindow = null, trackObj = null, lastFrame = null, modelRect = null, modelFrame = null, canvasCtx = null;

  this.getSearchWindow = function () { return searchWindow.clone(); };
  this.getTrackObj = function () { return trackObj.clone(); };

  /* the work of initTracker / track on an ImageData; the frame is uploaded unless the caller already has (bindFrame) */
  this._initImg = function (img, trackedArea, ctx2d) {
    canvasCtx = ctx2d;
    const rect = new Int32Array([trackedArea.x, trackedArea.y, trackedArea.width, trackedArea.height]);
    if (img.width > 0 && img.height > 0) {
      bindFrame(csPool.ctx, img, headtrackr.cascade, 5);
      addon().camshiftInitBound(csPool.ctx.handle, 1, slot, rect);
    }
    modelFrame = img; modelRect = trackedArea.clone();
    searchWindow = trackedArea.clone();
    trackObj = new headtrackr.camshift.TrackObj();
  };
  this._trackImg = function (img) {
    if (img.width === 0 || img.height === 0) return;
    lastFrame = img;
    bindFrame(csPool.ctx, img, headtrackr.cascade, 5);
    const r = addon().camshiftTrackBound(csPool.ctx.handle, 1, slot, params.calcAngles ? 1 : 0, true);
    trackObj.x = r[0]; trackObj.y = r[1]; trackObj.width = r[2]; trackObj.height = r[3]; trackObj.angle = r[4];
    searchWindow.x = r[5]; searchWindow.y = r[6]; searchWindow.width = r[7]; searchWindow.height = r[8];
  };

  this.initTracker = function (canvas, trackedArea) { /* camshift.js:198-211 */
    const ctx2d = canvas.getContext('2d');
    try { this._initImg(ctx2d.getImageData(0, 0, canvas.width, canvas.height), trackedArea, ctx2d); } finally { unbindFrames(); }
  };

  this.track = function (canvas) { /* camshift.js:213-259 */
    try { this._trackImg(canvas.getContext('2d').getImageData(0, 0, canvas.width, canvas.height)); } finally { unbindFrames(); }
  };

  /* debug getters: the back-projection is never materialised on the device (only these two functions can observe it),
   * so it is rebuilt here on demand from the last frame (camshift.js:172-196, 314-353) */
  this.getPdf = function () {
    if (!lastFrame || !modelFrame) return undefined;
    const w = lastFrame.width, h = lastFrame.height, d = lastFrame.data;
    const mr = modelRect, mw = modelFrame.width, mh = modelFrame.height, md = modelFrame.data;
    const model = new Uint32Array(4096);
    for (let y = mr.y; y < mr.y + mr.height; y++) {
      for (let x = mr.x; x < mr.x + mr.width; x++) {
        if (x >= 0 && x < mw && y >= 0 && y < mh) {
          const p = (y * mw + x) * 4;
          model[256 * (md[p] >> 4) + 16 * (md[p + 1] >> 4) + (md[p + 2] >> 4)]++;
        } else model[0]++;
      }
    }
    const cur = new headtrackr.camshift.Histogram(d);
    const weights = new Float64Array(4096);
    for (let i = 0; i < 4096; i++) weights[i] = cur.getBin(i) !== 0 ? Math.min(model[i] / cur.getBin(i), 1) : 0;
    const data = [];
    for (let x = 0; x < w; x++) {
      const col = [];
      for (let y = 0; y < h; y++) {
        const p = (y * w + x) * 4;
        col.push(weights[256 * (d[p] >> 4) + 16 * (d[p + 1] >> 4) + (d[p + 2] >> 4)]);
      }
      data[x] = col;
    }
    return data;
  };

  this.getBackProjectionImg = function () {
    const pdf = this.getPdf();
    const w = lastFrame.width, h = lastFrame.height;
    const img = canvasCtx.createImageData(w, h), out = img.data;
    for (let x = 0; x < w; x++) {
      for (let y = 0; y < h; y++) {
        const v = Math.floor(255 * pdf[x][y]), p = (y * w + x) * 4;
        out[p] = v; out[p + 1] = v; out[p + 2] = v; out[p + 3] = 255;
      }
    }
    return img;
  };

  /* not in the reference: returns the device slot (idempotent).  facetrackr.Tracker.release() calls it when the facade
   * replaces a tracker on "redetecting" / stop(), so a long-running feed that loses its face keeps one slot. */
  let released = false;
  this.release = function () { if (!released) { released = true; csPool.free.push(slot); } };
};

/* ---- facetrackr ------------------------------------------------------------------------------------------------------------ */

headtrackr.facetrackr = {};

headtrackr.facetrackr.TrackObj = function () { /* facetrackr.js:233-255 */
  this.height = 0; this.width = 0; this.angle = 0; this.x = 0; this.y = 0;
  this.confidence = -10000; this.detection = ''; this.time = 0;
  this.clone = function () {
    const c = new headtrackr.facetrackr.TrackObj();
    c.height = this.height; c.width = this.width; c.angle = this.angle; c.x = this.x; c.y = this.y;
    c.confidence = this.confidence; c.detection = this.detection; c.time = this.time;
    return c;
  };
};

function now() { return (new Date()).getTime(); }

headtrackr.facetrackr.Tracker = function (params) { /* facetrackr.js:37-228 */
  if (!params) params = {};
  if (params.sendEvents === undefined) params.sendEvents = true;
  if (params.whitebalancing === undefined) params.whitebalancing = true;
  if (params.debug === undefined || params.debug.tagName !== 'CANVAS') params.debug = false;
  if (params.calcAngles === undefined) params.calcAngles = false;
  let state = params.whitebalancing ? 'WB' : 'VJ';
  let input = null, current = null, cs = null;
  const confidenceThreshold = -10; /* facetrackr.js:57 */
  const wbWindow = [], wbLength = 15; /* facetrackr.js:58-59 */

  this.init = function (inputcanvas) {
    input = inputcanvas;
    if (cs) cs.release(); /* a second init() reuses nothing of the first (facetrackr.js:61-65 builds a new camshift.Tracker) */
    cs = new headtrackr.camshift.Tracker({ calcAngles: params.calcAngles });
  };

  function detectVJ(img) { /* facetrackr.js:133-182; the canvas copy + grayscale are fused into the device path */
    const start = now();
    const comp = (img.width > 0 && img.height > 0) ? detectBoundImg(img, headtrackr.cascade, 5, 1) : [];
    const diff = now() - start;
    let best;
    for (let i = 0; i < comp.length; i++) if (best === undefined || comp[i].confidence > best.confidence) best = comp[i];
    const result = new headtrackr.facetrackr.TrackObj();
    if (best !== undefined) {
      result.width = best.width; result.height = best.height; result.x = best.x; result.y = best.y; result.confidence = best.confidence;
    }
    result.time = diff;
    result.detection = 'VJ';
    return result;
  }

  function detectCS(img) { /* facetrackr.js:185-217 */
    const start = now();
    cs._trackImg(img);
    const r = cs.getTrackObj();
    if (params.debug) params.debug.getContext('2d').putImageData(cs.getBackProjectionImg(), 0, 0);
    const result = new headtrackr.facetrackr.TrackObj();
    result.width = r.width; result.height = r.height; result.x = r.x; result.y = r.y; result.angle = r.angle;
    result.confidence = 1;
    result.time = now() - start;
    result.detection = 'CS';
    return result;
  }

  function checkWB(img) { /* facetrackr.js:220-227 */
    const result = new headtrackr.facetrackr.TrackObj();
    if (img.width > 0 && img.height > 0) {
      const c = contextFor(headtrackr.cascade, 5);
      bindFrame(c, img, headtrackr.cascade, 5);
      result.wb = addon().whitebalanceBound(c.handle, 1)[0];
    } else result.wb = NaN; /* 0/0 in the reference */
    result.detection = 'WB';
    return result;
  }

  this.track = function () { /* facetrackr.js:67-126 */
    /* the frame is read from the canvas and uploaded ONCE per call; whatever runs on it (whitebalance, detection, initTracker,
     * track) shares that copy — the reference calls getImageData in each of them (whitebalance.js:12, facetrackr.js:143-149,
     * camshift.js:206,218), here that would be one PCIe transfer each */
    const ctx2d = input.getContext('2d');
    const img = ctx2d.getImageData(0, 0, input.width, input.height);
    let result;
    try {
      if (state === 'WB') result = checkWB(img);
      else if (state === 'VJ') result = detectVJ(img);
      else result = detectCS(img);
      if (result.detection === 'VJ' && result.confidence > confidenceThreshold) { /* facetrackr.js:97-108 */
        state = 'CS';
        cs._initImg(img, new headtrackr.camshift.Rectangle(Math.floor(result.x), Math.floor(result.y),
          Math.floor(result.width), Math.floor(result.height)), ctx2d);
      }
    } finally { unbindFrames(); } /* the uploaded copy is shared within this call only */

    if (result.detection === 'WB') { /* facetrackr.js:79-95 */
      if (wbWindow.length >= wbLength) wbWindow.pop();
      wbWindow.unshift(result.wb);
      if (wbWindow.length === wbLength && Math.max.apply(null, wbWindow) - Math.min.apply(null, wbWindow) < 2) state = 'VJ';
    }
    current = result;
    if (result.detection === 'CS' && params.sendEvents) { /* facetrackr.js:112-125 */
      const hasDoc = typeof document !== 'undefined' && document.createEvent;
      const evt = hasDoc ? document.createEvent('Event') : { type: 'facetrackingEvent' };
      if (hasDoc) evt.initEvent('facetrackingEvent', true, true);
      evt.height = result.height; evt.width = result.width; evt.angle = result.angle; evt.x = result.x; evt.y = result.y;
      evt.confidence = result.confidence; evt.detection = result.detection; evt.time = result.time;
      if (hasDoc) document.dispatchEvent(evt);
      if (params.onEvent) params.onEvent('facetrackingEvent', evt); /* Node hosts without a DOM */
    }
  };

  this.getTrackingObject = function () { return current.clone(); };

  this.release = function () { if (cs) { cs.release(); cs = null; } }; /* not in the reference: frees the camshift device slot */
};

require('./tracker.js')(headtrackr); /* Smoother, headposition, Tracker facade (host post-processing) */

module.exports = headtrackr;
