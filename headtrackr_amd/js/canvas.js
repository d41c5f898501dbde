'use strict';
/*
 * canvas.js — a minimal RGBA canvas for Node hosts of headtrackr.js (Node has no <canvas>).
 * Provides exactly what the per-frame path needs from a 2-D context: getImageData, putImageData, createImageData and
 * drawImage (same-size copies, and scaled copies with this project's declared resampler: centre-aligned bilinear in
 * binary64, round-half-even store — see DESIGN.md "pyramid resampler").  Frames are plain Uint8ClampedArray RGBA.
 * For the debug overlay of headtrackr.Tracker (main.js:199-219) it also strokes rectangles under translate / rotate:
 * 1-pixel aliased edges as declared in oracle/canvas_shim.js (a browser canvas would anti-alias; the reference does not care).
 */
function ImageDataLike(w, h) {
  this.width = w; this.height = h;
  this.data = new Uint8ClampedArray(Math.max(w, 0) * Math.max(h, 0) * 4);
}

function Ctx(canvas) { this.canvas = canvas; this.strokeStyle = '#000000'; this.xf = [1, 0, 0, 1, 0, 0]; }
Ctx.prototype.translate = function (tx, ty) { const m = this.xf; m[4] += m[0] * tx + m[2] * ty; m[5] += m[1] * tx + m[3] * ty; };
Ctx.prototype.rotate = function (angle) {
  const m = this.xf, co = Math.cos(angle), si = Math.sin(angle);
  this.xf = [m[0] * co + m[2] * si, m[1] * co + m[3] * si, m[2] * co - m[0] * si, m[3] * co - m[1] * si, m[4], m[5]];
};
Ctx.prototype.strokeRect = function (x, y, w, h) {
  const cv = this.canvas, m = this.xf, rgb = parseInt(String(this.strokeStyle).slice(1), 16);
  const map = function (px, py) { return [m[0] * px + m[2] * py + m[4], m[1] * px + m[3] * py + m[5]]; };
  const corners = [map(x, y), map(x + w, y), map(x + w, y + h), map(x, y + h)];
  for (let e = 0; e < 4; e++) {
    const from = corners[e], to = corners[(e + 1) % 4], dx = to[0] - from[0], dy = to[1] - from[1];
    const steps = Math.max(Math.ceil(Math.abs(dx)), Math.ceil(Math.abs(dy)), 1);
    for (let i = 0; i <= steps; i++) {
      const px = Math.round(from[0] + dx * i / steps), py = Math.round(from[1] + dy * i / steps);
      if (px < 0 || py < 0 || px >= cv.width || py >= cv.height) continue;
      const o = (py * cv.width + px) * 4;
      cv.pixels[o] = (rgb >> 16) & 255; cv.pixels[o + 1] = (rgb >> 8) & 255; cv.pixels[o + 2] = rgb & 255; cv.pixels[o + 3] = 255;
    }
  }
};
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
