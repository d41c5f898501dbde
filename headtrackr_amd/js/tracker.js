'use strict';
/*
 * tracker.js — host-side post-processing and the `headtrackr.Tracker` facade (SURVEY.md §8f).  Pure scalar math on five
 * numbers per frame: it stays on the host by design; the pixel work below it (facetrackr -> ccv / camshift) runs on the
 * GPU through headtrackr.js.  Reference: /root/reference/src/smoother.js, headposition.js, main.js.
 *
 *   headtrackr.Smoother(alpha, interval)                       smoother.js:13-88
 *   headtrackr.headposition.Tracker / TrackObj                 headposition.js:35-217
 *   headtrackr.Tracker(params)  {init, start, stop, stopStream, getFOV, status}   main.js:35-379
 *
 * The facade replaces main.js's webcam plumbing (getUserMedia, <video>, DOM overlay) by a frame source: `video` is any
 * object the canvas can drawImage() from — e.g. a ./canvas.js Canvas whose pixels the application updates — and
 * `step()` runs exactly one iteration of main.js's track() body, so servers can drive it from their own loop;
 * start()/stop() keep the reference's setTimeout loop.  Status / tracking events go to `document.dispatchEvent` when a
 * `document` exists (browser, tests) and always to the optional `params.onEvent(type, event)` callback.
 */
module.exports = function install(headtrackr) {
  function emit(type, fields, onEvent) {
    let evt = null;
    if (typeof document !== 'undefined' && document.createEvent) {
      evt = document.createEvent('Event');
      evt.initEvent(type, true, true);
    } else {
      evt = { type: type };
    }
    Object.keys(fields).forEach(function (k) { evt[k] = fields[k]; });
    if (typeof document !== 'undefined' && document.dispatchEvent) document.dispatchEvent(evt);
    if (onEvent) onEvent(type, evt);
    return evt;
  }

  /* ---- Smoother ------------------------------------------------------------------------------------------------
   * The reference declares double exponential smoothing, but `sp2 = sp` (smoother.js:28) makes both levels the SAME
   * array and predict() reads `this.interpolate` on the wrong `this` (smoother.js:65), so what actually runs is:
   *     s <- alpha*p + (1-alpha)*s ;  s <- alpha*s + (1-alpha)*s ;  out = (2 + r)*s - (1 + r)*s ,  r = alpha*step/(1-alpha)
   * with step = floor(elapsed/interval) = 0 in practice.  Restated as such (same operations, same rounding). */
  headtrackr.Smoother = function (alpha, interval) {
    let s = null;
    let last = new Date();
    this.initialized = false;
    this.interpolate = false; /* present in the reference's API; never honoured there either */
    this.init = function (p) {
      this.initialized = true;
      s = [p.x, p.y, p.z, p.width, p.height];
    };
    this.smooth = function (pos) {
      if (!this.initialized) return false;
      const p = [pos.x, pos.y, pos.z, pos.width, pos.height];
      for (let i = 0; i < 5; i++) {
        s[i] = alpha * p[i] + (1 - alpha) * s[i];
        s[i] = alpha * s[i] + (1 - alpha) * s[i];
      }
      last = new Date();
      const step = ((new Date()) - last) / interval >> 0;
      const ratio = (alpha * step) / (1 - alpha), a = 2 + ratio, b = 1 + ratio;
      pos.x = a * s[0] - b * s[0];
      pos.y = a * s[1] - b * s[1];
      pos.z = a * s[2] - b * s[2];
      pos.width = a * s[3] - b * s[3];
      pos.height = a * s[4] - b * s[4];
      return pos;
    };
  };

  /* ---- headposition ----------------------------------------------------------------------------------------------
   * Pinhole estimate of the head position (cm, relative to the screen centre) from the face rectangle, assuming a
   * 16 x 19 cm head and, when no fov is given, a viewer 60 cm away at initialisation (headposition.js:35-201). */
  headtrackr.headposition = {};
  headtrackr.headposition.TrackObj = function (x, y, z) {
    this.x = x; this.y = y; this.z = z;
    this.clone = function () { return new headtrackr.headposition.TrackObj(this.x, this.y, this.z); };
  };
  headtrackr.headposition.Tracker = function (face, camwidth, camheight, params) {
    if (!params) params = {};
    const edgecorrection = params.edgecorrection === undefined ? true : params.edgecorrection;
    this.camheight_cam = camheight;
    this.camwidth_cam = camwidth;
    const headW = 16, headH = 19;
    const smallAngle = Math.atan(headW / headH);
    const diagCm = Math.sqrt((headW * headW) + (headH * headH));
    const sinA = Math.sin(smallAngle), cosA = Math.cos(smallAngle), tanA = Math.tan(smallAngle);
    let diagCam = Math.sqrt((face.width * face.width) + (face.height * face.height));
    let fov;
    if (params.fov === undefined) {
      const headWidthCam = sinA * diagCam;
      const camWidthCm = (this.camwidth_cam / headWidthCam) * headW;
      const dist = params.distance_to_screen === undefined ? 60 : params.distance_to_screen;
      fov = Math.atan((camWidthCm / 2) / dist) * 2;
    } else {
      fov = params.fov * Math.PI / 180;
    }
    const tanFov = 2 * Math.tan(fov / 2);
    let x, y, z;

    this.track = function (f) {
      const w = f.width, h = f.height;
      let fx = f.x, fy = f.y;
      const full = function () { return Math.sqrt((w * w) + (h * h)); };
      if (edgecorrection) {
        const margin = 11;
        const left = fx - (w / 2), right = this.camwidth_cam - (fx + (w / 2));
        const top = fy - (h / 2), bottom = this.camheight_cam - (fy + (h / 2));
        const onV = (left < margin || right < margin), onH = (top < margin || bottom < margin);
        if (onH && onV) { /* corner: keep the previous diagonal, headposition.js:114-130 */
          fx = (left < margin) ? w - (diagCam * sinA / 2) : fx - (w / 2) + (diagCam * sinA / 2);
          fy = (top < margin) ? h - (diagCam * cosA / 2) : fy - (h / 2) + (diagCam * cosA / 2);
        } else if (onH) { /* top / bottom edge: trust the width, headposition.js:131-146 */
          const d = (top < margin) ? top : bottom;
          const ow = d / margin, ew = (margin - d) / margin;
          const part = ow * (h / 2) + ew * ((w / tanA) / 2);
          fy = (top < margin) ? h - part : fy - (h / 2) + part;
          diagCam = ew * (w / sinA) + ow * full();
        } else if (onV) { /* left / right edge: trust the height, headposition.js:147-161 */
          const d = (left < margin) ? left : right;
          const ow = d / margin, ew = (margin - d) / margin;
          diagCam = ew * (h / cosA) + ow * full();
          const part = (left < margin) ? ow * (w / 2) + (ew) * (h * tanA / 2) : ow * (w / 2) + ew * (h * tanA / 2);
          fx = (left < margin) ? w - part : fx - (w / 2) + part;
        } else {
          diagCam = full();
        }
      } else {
        diagCam = full();
      }
      z = (diagCm * this.camwidth_cam) / (tanFov * diagCam);
      x = -((fx / this.camwidth_cam) - 0.5) * z * tanFov;
      y = -((fy / this.camheight_cam) - 0.5) * z * tanFov * (this.camheight_cam / this.camwidth_cam);
      y = y + (params.distance_from_camera_to_screen === undefined ? 11.5 : params.distance_from_camera_to_screen);
      emit('headtrackingEvent', { x: x, y: y, z: z }, params.onEvent);
      return new headtrackr.headposition.TrackObj(x, y, z);
    };
    this.getTrackerObj = function () { return new headtrackr.headposition.TrackObj(x, y, z); };
    this.getFOV = function () { return fov * 180 / Math.PI; };
  };

  /* ---- Tracker facade --------------------------------------------------------------------------------------------- */
  headtrackr.Tracker = function (params) {
    params = Object.assign({ smoothing: true, retryDetection: true, detectionInterval: 20, cameraOffset: 11.5, calcAngles: false,
      headPosition: true, whitebalancing: true }, params || {});
    let video = null, canvas = null, ctx = null, smoother = null, facetracker, headposition, timer = null;
    let fov = 0, run = false, faceFound = false, firstRun = true, detectionStart;
    /* debug overlay (main.js:42-50): a canvas that receives the back-projection (facetrackr.js:194-196) and the face boxes */
    if (params.debug === undefined || !params.debug || params.debug.tagName !== 'CANVAS') params.debug = false;
    const debugContext = params.debug ? params.debug.getContext('2d') : null;
    const diagonals = [];
    const self = this;
    this.status = '';
    this.stream = undefined;
    this.initialized = false;

    function status(msg) { self.status = msg; emit('headtrackrStatus', { status: msg }, params.onEvent); }

    this.init = function (videoSource, canvasElement) { /* main.js:99-166 without getUserMedia */
      video = videoSource; canvas = canvasElement; ctx = canvas.getContext('2d');
      smoother = new headtrackr.Smoother(0.35, params.detectionInterval + 15);
      this.initialized = true;
      return true;
    };

    /* one iteration of main.js:168-305; returns {face, head} for callers that do not use events */
    this.step = function () {
      if (video && video !== canvas) ctx.drawImage(video, 0, 0, canvas.width, canvas.height);
      if (facetracker === undefined) {
        facetracker = new headtrackr.facetrackr.Tracker({ debug: params.debug, calcAngles: params.calcAngles, whitebalancing: params.whitebalancing, onEvent: params.onEvent });
        facetracker.init(canvas);
      }
      facetracker.track();
      let face = facetracker.getTrackingObject();
      let head = null;
      if (face.detection === 'WB') status('whitebalance');
      if (firstRun && face.detection === 'VJ') status('detecting');
      if (!(face.confidence === 0)) {
        if (face.detection === 'VJ') {
          if (detectionStart === undefined) detectionStart = (new Date()).getTime();
          if (((new Date()).getTime() - detectionStart) > 5000) status('hints');
          if (debugContext) { /* detected face on the debug canvas, main.js:199-203 */
            debugContext.strokeStyle = '#0000CC';
            debugContext.strokeRect(face.x, face.y, face.width, face.height);
          }
        }
        if (face.detection === 'CS') {
          detectionStart = undefined;
          if (debugContext) { /* tracked face, rotated about its centre, main.js:211-219 */
            debugContext.translate(face.x, face.y);
            debugContext.rotate(face.angle - (Math.PI / 2));
            debugContext.strokeStyle = '#00CC00';
            debugContext.strokeRect((-(face.width / 2)) >> 0, (-(face.height / 2)) >> 0, face.width, face.height);
            debugContext.rotate((Math.PI / 2) - face.angle);
            debugContext.translate(-face.x, -face.y);
          }
          this.status = 'tracking';
          if (face.width === 0 || face.height === 0) { /* lost: zero mass in camshift, main.js:230-248 */
            if (params.retryDetection) {
              status('redetecting');
              facetracker.release(); /* the lost tracker's camshift slot goes back to the pool */
              facetracker = new headtrackr.facetrackr.Tracker({ whitebalancing: false, debug: params.debug, calcAngles: params.calcAngles, onEvent: params.onEvent });
              facetracker.init(canvas);
              faceFound = false;
              headposition = undefined;
            } else {
              status('lost');
              this.stop();
            }
          } else {
            if (!faceFound) { status('found'); faceFound = true; }
            if (params.smoothing) {
              if (!smoother.initialized) smoother.init(face);
              face = smoother.smooth(face);
            }
            if (headposition === undefined && params.headPosition) {
              /* wait for a stable head diagonal (6 samples within 5 px) before fixing the field of view, main.js:263-291 */
              const d = Math.sqrt(face.width * face.width + face.height * face.height);
              let stable = false;
              if (diagonals.length < 6) diagonals.push(d);
              else {
                diagonals.splice(0, 1); diagonals.push(d);
                stable = (Math.max.apply(null, diagonals) - Math.min.apply(null, diagonals)) < 5;
              }
              if (stable) {
                const hp = { distance_from_camera_to_screen: params.cameraOffset, onEvent: params.onEvent };
                if (firstRun) {
                  if (params.fov !== undefined) hp.fov = params.fov;
                  headposition = new headtrackr.headposition.Tracker(face, canvas.width, canvas.height, hp);
                  fov = headposition.getFOV();
                  firstRun = false;
                } else {
                  hp.fov = fov;
                  headposition = new headtrackr.headposition.Tracker(face, canvas.width, canvas.height, hp);
                }
                head = headposition.track(face);
              }
            } else if (params.headPosition) {
              head = headposition.track(face);
            }
          }
        }
      }
      return { face: face, head: head };
    };

    const loop = function () {
      self.step();
      if (run) timer = setTimeout(loop, params.detectionInterval);
    };
    this.start = function () { /* main.js:307-345: needs something on the canvas first */
      if (!this.initialized) return false;
      run = true;
      loop();
      return true;
    };
    this.stop = function () { /* main.js:347-355 */
      if (timer) clearTimeout(timer);
      run = false;
      status('stopped');
      if (facetracker !== undefined) facetracker.release();
      facetracker = undefined;
      faceFound = false;
      return true;
    };
    this.stopStream = function () { if (this.stream !== undefined && this.stream.stop) this.stream.stop(); };
    this.getFOV = function () { return fov; };
  };
};
