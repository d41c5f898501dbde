'use strict';
/*
 * canvas.js — a minimal RGBA canvas for Node hosts of headtrackr.js (Node has no <canvas>).
 * Provides exactly what the per-frame path needs from a 2-D context: getImageData, putImageData, createImageData and
 * drawImage (same-size copies, and scaled copies with this project's declared resampler: centre-aligned bilinear in
 * binary64, round-half-even store — see DESIGN.md "pyramid resampler").  Frames are plain Uint8ClampedArray RGBA.
 */
function ImageDataLike(w, h) {
  this.width = w; this.height = h;
  this.data = new Uint8ClampedArray(Math.max(w, 0) * Math.max(h, 0) * 4);
}

function Ctx(canvas) { this.canvas = canvas; }
Ctx.prototype.createImageData = function (w, h) { return new ImageDataLike(w | 0, h | 0); };
Ctx.prototype.getImageData = function (x, y, w, h) {
  x |= 0; y |= 0; w |= 0; h |= 0;
  const c = this.canvas, out = new ImageDataLike(w, h);
  for (let r = Math.max(y, 0); r < Math.min(y + h, c.height); r++) {
    const x0 = Math.max(x, 0), x1 = Math.min(x + w, c.width);
    if (x1 > x0) out.data.set(c.pixels.subarray((r * c.width + x0) * 4, (r * c.width + x1) * 4), ((r - y) * w + (x0 - x)) * 4);
  }
  return out;
};
Ctx.prototype.putImageData = function (img, dx, dy) {
  dx |= 0; dy |= 0;
  const c = this.canvas;
  for (let r = Math.max(dy, 0); r < Math.min(dy + img.height, c.height); r++) {
    const x0 = Math.max(dx, 0), x1 = Math.min(dx + img.width, c.width);
    if (x1 > x0) c.pixels.set(img.data.subarray(((r - dy) * img.width + (x0 - dx)) * 4, ((r - dy) * img.width + (x1 - dx)) * 4), (r * c.width + x0) * 4);
  }
};
Ctx.prototype.drawImage = function (src) {
  const a = arguments;
  let sx = 0, sy = 0, sw = src.width, sh = src.height, dx, dy, dw = sw, dh = sh;
  if (a.length === 3) { dx = a[1]; dy = a[2]; } else if (a.length === 5) { dx = a[1]; dy = a[2]; dw = a[3]; dh = a[4]; } else {
    sx = a[1]; sy = a[2]; sw = a[3]; sh = a[4]; dx = a[5]; dy = a[6]; dw = a[7]; dh = a[8];
  }
  sx |= 0; sy |= 0; sw |= 0; sh |= 0; dx |= 0; dy |= 0; dw |= 0; dh |= 0;
  if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return;
  const S = src.pixels, SW = src.width, D = this.canvas.pixels, DW = this.canvas.width, DH = this.canvas.height;
  const rx = sw / dw, ry = sh / dh;
  for (let j = 0; j < dh; j++) {
    const oy = dy + j;
    if (oy < 0 || oy >= DH) continue;
    const fy = Math.min(Math.max((j + 0.5) * ry - 0.5, 0), sh - 1), y0 = Math.floor(fy), ty = fy - y0;
    const r0 = (sy + y0) * SW, r1 = (sy + Math.min(y0 + 1, sh - 1)) * SW;
    for (let i = 0; i < dw; i++) {
      const ox = dx + i;
      if (ox < 0 || ox >= DW) continue;
      const fx = Math.min(Math.max((i + 0.5) * rx - 0.5, 0), sw - 1), x0 = Math.floor(fx), tx = fx - x0;
      const ca = sx + x0, cb = sx + Math.min(x0 + 1, sw - 1), o = (oy * DW + ox) * 4;
      for (let c = 0; c < 4; c++) {
        const top = S[(r0 + ca) * 4 + c] * (1 - tx) + S[(r0 + cb) * 4 + c] * tx;
        const bot = S[(r1 + ca) * 4 + c] * (1 - tx) + S[(r1 + cb) * 4 + c] * tx;
        D[o + c] = top * (1 - ty) + bot * ty;
      }
    }
  }
};

function Canvas(w, h) {
  this.tagName = 'CANVAS';
  let cw = w | 0, ch = h | 0;
  this.pixels = new Uint8ClampedArray(cw * ch * 4);
  const self = this;
  Object.defineProperty(this, 'width', { get: function () { return cw; }, set: function (v) { cw = Math.max(v | 0, 0); self.pixels = new Uint8ClampedArray(cw * ch * 4); } });
  Object.defineProperty(this, 'height', { get: function () { return ch; }, set: function (v) { ch = Math.max(v | 0, 0); self.pixels = new Uint8ClampedArray(cw * ch * 4); } });
  const ctx = new Ctx(this);
  this.getContext = function () { return ctx; };
}
/* load one RGBA frame (Buffer / typed array of width*height*4 bytes) */
Canvas.prototype.setFrame = function (rgba) { this.pixels.set(rgba); return this; };

module.exports = { Canvas: Canvas };
