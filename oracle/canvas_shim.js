'use strict';
/*
 * canvas_shim.js — TEST INFRASTRUCTURE (oracle side). Not part of the product path.
 *
 * The reference (auduno/headtrackr) never does its own image resampling: every pyramid level is made by
 * the host browser's canvas-2D drawImage (/root/reference/src/ccv.js:121,128,135,140,145), whose filter
 * is implementation-defined and unpinned (no browser/version named anywhere in the reference, and the
 * reference ships no tests).  This file therefore *declares* the canvas semantics that this repo's
 * oracle, C restatement (oracle/ht_oracle.c) and HIP kernels (headtrackr_amd/csrc/pyramid.hip) all
 * implement bit-identically.  "Parity unpinned" at the drawImage boundary; pinned everywhere else by
 * executing the unmodified reference JS on top of this shim (oracle/ref_harness.js).
 *
 * Declared resampler  (drawImage src rect (sx,sy,sw,sh) -> dst rect (dx,dy,dw,dh), all integers):
 *   rx = sw/dw, ry = sh/dh                                   (IEEE-754 binary64, one division each)
 *   for dst column i:  fx = (i + 0.5) * rx - 0.5 ; fx = min(max(fx, 0), sw-1)
 *                      x0 = floor(fx) ; tx = fx - x0 ; x1 = min(x0 + 1, sw - 1)
 *   rows likewise (fy, y0, ty, y1)
 *   per channel:       top = s[y0][x0]*(1-tx) + s[y0][x1]*tx
 *                      bot = s[y1][x0]*(1-tx) + s[y1][x1]*tx
 *                      v   = top*(1-ty) + bot*ty              (binary64, no fused multiply-add)
 *                      dst = Uint8ClampedArray store of v     (clamp to [0,255], round half to even)
 *   i.e. centre-aligned bilinear, edge-clamped to the source rect; exact 2:1 halving = 2x2 box mean;
 *   1:1 is an exact copy.  Sources are treated as opaque (channels resampled independently and
 *   written, not alpha-composited).  A call with any of sw,sh,dw,dh <= 0 draws nothing.
 * Stroking (debug canvas only, /root/reference/src/main.js:199-219): strokeStyle '#RRGGBB', strokeRect, translate, rotate.  The
 * reference leaves the rasterisation to the browser (anti-aliased, implementation-defined); declared here as: the four corners
 * are mapped through the current transform (translate / rotate compose like the canvas spec's), every edge is drawn as a
 * 1-pixel line without anti-aliasing — n = max(ceil(|dx|), ceil(|dy|), 1) steps, pixel (round(x), round(y)) with Math.round,
 * colour = strokeStyle, alpha 255, clipped to the canvas.  Every stroke call is also appended to canvas._calls as
 * [op, args...], which is what the golden vectors pin (the pixels are pinned only relative to this declaration).
 * Other semantics: resizing a canvas clears it to transparent black; getImageData outside the canvas
 * reads transparent black (used by camshift.initTracker, /root/reference/src/camshift.js:206).
 */

const stats = { shimNs: 0n, enabled: false, created: [] };

function tic() { return stats.enabled ? process.hrtime.bigint() : 0n; }
function toc(t0) { if (stats.enabled) stats.shimNs += process.hrtime.bigint() - t0; }

function ImageData(w, h, data) {
  this.width = w;
  this.height = h;
  this.data = data || new Uint8ClampedArray(Math.max(0, w) * Math.max(0, h) * 4);
}

function Context2D(canvas) {
  this.canvas = canvas;
  this.strokeStyle = '#000000';
  this._m = [1, 0, 0, 1, 0, 0]; /* a b c d e f: x' = a x + c y + e, y' = b x + d y + f */
}

Context2D.prototype.translate = function (tx, ty) {
  const m = this._m;
  m[4] += m[0] * tx + m[2] * ty; m[5] += m[1] * tx + m[3] * ty;
  this.canvas._calls.push(['translate', tx, ty]);
};
Context2D.prototype.rotate = function (ang) {
  const m = this._m, c = Math.cos(ang), s = Math.sin(ang);
  const a = m[0] * c + m[2] * s, b = m[1] * c + m[3] * s, cc = m[2] * c - m[0] * s, d = m[3] * c - m[1] * s;
  m[0] = a; m[1] = b; m[2] = cc; m[3] = d;
  this.canvas._calls.push(['rotate', ang]);
};
Context2D.prototype.strokeRect = function (x, y, w, h) {
  const cv = this.canvas, m = this._m, col = parseInt(String(this.strokeStyle).slice(1), 16);
  cv._calls.push(['strokeRect', this.strokeStyle, x, y, w, h]);
  const P = [[x, y], [x + w, y], [x + w, y + h], [x, y + h]].map(function (p) { return [m[0] * p[0] + m[2] * p[1] + m[4], m[1] * p[0] + m[3] * p[1] + m[5]]; });
  for (let e = 0; e < 4; e++) {
    const p = P[e], q = P[(e + 1) & 3], dx = q[0] - p[0], dy = q[1] - p[1];
    const n = Math.max(Math.ceil(Math.abs(dx)), Math.ceil(Math.abs(dy)), 1);
    for (let i = 0; i <= n; i++) {
      const px = Math.round(p[0] + dx * i / n), py = Math.round(p[1] + dy * i / n);
      if (px >= 0 && py >= 0 && px < cv._w && py < cv._h) {
        const o = (py * cv._w + px) * 4;
        cv._buf[o] = (col >> 16) & 255; cv._buf[o + 1] = (col >> 8) & 255; cv._buf[o + 2] = col & 255; cv._buf[o + 3] = 255;
      }
    }
  }
};

Context2D.prototype.createImageData = function (w, h) {
  return new ImageData(w | 0, h | 0);
};

Context2D.prototype.getImageData = function (x, y, w, h) {
  const t0 = tic();
  x |= 0; y |= 0; w |= 0; h |= 0;
  const c = this.canvas, cw = c._w, ch = c._h, src = c._buf;
  const out = new ImageData(w, h);
  if (w > 0 && h > 0) {
    const dst = out.data;
    if (x === 0 && y === 0 && w === cw && h === ch) {
      dst.set(src);
    } else {
      const x0 = Math.max(x, 0), x1 = Math.min(x + w, cw);
      const y0 = Math.max(y, 0), y1 = Math.min(y + h, ch);
      if (x1 > x0) {
        for (let yy = y0; yy < y1; yy++) {
          const s = (yy * cw + x0) * 4;
          dst.set(src.subarray(s, s + (x1 - x0) * 4), ((yy - y) * w + (x0 - x)) * 4);
        }
      }
    }
  }
  toc(t0);
  return out;
};

Context2D.prototype.putImageData = function (img, dx, dy) {
  const t0 = tic();
  dx |= 0; dy |= 0;
  const c = this.canvas, cw = c._w, ch = c._h, dst = c._buf;
  const w = img.width, h = img.height, src = img.data;
  if (dx === 0 && dy === 0 && w === cw && h === ch) {
    dst.set(src);
  } else {
    const x0 = Math.max(dx, 0), x1 = Math.min(dx + w, cw);
    const y0 = Math.max(dy, 0), y1 = Math.min(dy + h, ch);
    if (x1 > x0) {
      for (let yy = y0; yy < y1; yy++) {
        const s = ((yy - dy) * w + (x0 - dx)) * 4;
        dst.set(src.subarray(s, s + (x1 - x0) * 4), (yy * cw + x0) * 4);
      }
    }
  }
  toc(t0);
};

/* drawImage(src,dx,dy) | drawImage(src,dx,dy,dw,dh) | drawImage(src,sx,sy,sw,sh,dx,dy,dw,dh) */
Context2D.prototype.drawImage = function (src, a1, a2, a3, a4, a5, a6, a7, a8) {
  const t0 = tic();
  let sx, sy, sw, sh, dx, dy, dw, dh;
  if (arguments.length === 3) {
    sx = 0; sy = 0; sw = src.width; sh = src.height; dx = a1; dy = a2; dw = sw; dh = sh;
  } else if (arguments.length === 5) {
    sx = 0; sy = 0; sw = src.width; sh = src.height; dx = a1; dy = a2; dw = a3; dh = a4;
  } else {
    sx = a1; sy = a2; sw = a3; sh = a4; dx = a5; dy = a6; dw = a7; dh = a8;
  }
  sx |= 0; sy |= 0; sw |= 0; sh |= 0; dx |= 0; dy |= 0; dw |= 0; dh |= 0;
  if (sw > 0 && sh > 0 && dw > 0 && dh > 0) {
    resample(src._buf, src._w, src._h, sx, sy, sw, sh, this.canvas._buf, this.canvas._w, this.canvas._h, dx, dy, dw, dh);
  }
  toc(t0);
};

function resample(S, SW, SH, sx, sy, sw, sh, D, DW, DH, dx, dy, dw, dh) {
  const rx = sw / dw, ry = sh / dh;
  /* per-column taps */
  const X0 = new Int32Array(dw), X1 = new Int32Array(dw), TX = new Float64Array(dw);
  for (let i = 0; i < dw; i++) {
    let fx = (i + 0.5) * rx - 0.5;
    if (fx < 0) fx = 0;
    if (fx > sw - 1) fx = sw - 1;
    const x0 = Math.floor(fx);
    X0[i] = (sx + x0) * 4;
    X1[i] = (sx + Math.min(x0 + 1, sw - 1)) * 4;
    TX[i] = fx - x0;
  }
  for (let j = 0; j < dh; j++) {
    const oy = dy + j;
    if (oy < 0 || oy >= DH) continue;
    let fy = (j + 0.5) * ry - 0.5;
    if (fy < 0) fy = 0;
    if (fy > sh - 1) fy = sh - 1;
    const y0 = Math.floor(fy);
    const ty = fy - y0, uy = 1 - ty;
    const r0 = (sy + y0) * SW * 4, r1 = (sy + Math.min(y0 + 1, sh - 1)) * SW * 4;
    for (let i = 0; i < dw; i++) {
      const ox = dx + i;
      if (ox < 0 || ox >= DW) continue;
      const tx = TX[i], ux = 1 - tx, a = X0[i], b = X1[i];
      const o = (oy * DW + ox) * 4;
      for (let c = 0; c < 4; c++) {
        const top = S[r0 + a + c] * ux + S[r0 + b + c] * tx;
        const bot = S[r1 + a + c] * ux + S[r1 + b + c] * tx;
        D[o + c] = top * uy + bot * ty; /* Uint8ClampedArray: clamp + round-half-even */
      }
    }
  }
}

function Canvas(w, h) {
  this.tagName = 'CANVAS';
  this._w = (w | 0) > 0 ? (w | 0) : 0;
  this._h = (h | 0) > 0 ? (h | 0) : 0;
  this._buf = new Uint8ClampedArray(this._w * this._h * 4);
  this._ctx = null;
  this._calls = []; /* stroke-call log (see the header) */
}
Object.defineProperty(Canvas.prototype, 'width', {
  get: function () { return this._w; },
  set: function (v) { this._w = (v | 0) > 0 ? (v | 0) : 0; this._buf = new Uint8ClampedArray(this._w * this._h * 4); }
});
Object.defineProperty(Canvas.prototype, 'height', {
  get: function () { return this._h; },
  set: function (v) { this._h = (v | 0) > 0 ? (v | 0) : 0; this._buf = new Uint8ClampedArray(this._w * this._h * 4); }
});
Canvas.prototype.getContext = function (kind) {
  if (!this._ctx) this._ctx = new Context2D(this);
  return this._ctx;
};
/* load raw RGBA bytes (w*h*4) into the canvas */
Canvas.prototype.loadRGBA = function (bytes) { this._buf.set(bytes); return this; };

/* minimal `document` for the reference's event plumbing (facetrackr.js:114-124, headposition.js:183-188) */
function makeDocument() {
  const listeners = {};
  return {
    createElement: function (tag) {
      if (String(tag).toLowerCase() !== 'canvas') throw new Error('shim: only <canvas> is supported');
      const c = new Canvas(300, 150);
      if (stats.trackCreated) stats.created.push(c);
      return c;
    },
    createEvent: function () {
      return { initEvent: function (type) { this.type = type; } };
    },
    addEventListener: function (type, fn) { (listeners[type] = listeners[type] || []).push(fn); },
    removeEventListener: function (type, fn) {
      const l = listeners[type] || []; const i = l.indexOf(fn); if (i >= 0) l.splice(i, 1);
    },
    dispatchEvent: function (evt) { (listeners[evt.type] || []).slice().forEach(function (fn) { fn(evt); }); return true; }
  };
}

module.exports = { Canvas: Canvas, ImageData: ImageData, makeDocument: makeDocument, resample: resample, stats: stats };
