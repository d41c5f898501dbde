/*
 * tests/js/oracle_addon.c — TEST INFRASTRUCTURE: the CPU oracle (oracle/ht_oracle.c) as a Node addon, so that the JavaScript
 * facade's HOST logic (facetrackr state machine, headtrackr.Tracker loop, debug overlay) can run on a box without a GPU:
 * tests/js/mock_addon.js implements the product addon's entry points on top of these six functions and
 * tests/js/parity_cpu.js replays the reference-JS golden vectors through headtrackr_amd/js/headtrackr.js with it.
 * Nothing under headtrackr_amd/ knows about this file; the product addon (csrc/ht_napi.cc) has no CPU path.
 *
 * Built by tests/test_js_host.py:  gcc -O2 -fPIC -shared -ffp-contract=off -I/usr/include/node oracle_addon.c -o oracle_addon.node -lm
 * (the oracle is compiled INTO the addon with the flags of oracle/Makefile: every binary64 operation rounds like the reference JS).
 */
#include "../../oracle/ht_oracle.c"

#include <node_api.h>

#define ARGS(n)                                                                   \
    size_t argc = (n);                                                            \
    napi_value argv[(n)];                                                         \
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < (n)) { \
        napi_throw_error(env, NULL, "oracle_addon: wrong argument count");        \
        return NULL;                                                              \
    }

static int get_u8(napi_env env, napi_value v, uint8_t **p, size_t *len) {
    napi_typedarray_type t;
    napi_value ab;
    size_t off;
    void *data;
    if (napi_get_typedarray_info(env, v, &t, len, &data, &ab, &off) != napi_ok || (t != napi_uint8_array && t != napi_uint8_clamped_array)) {
        napi_throw_error(env, NULL, "oracle_addon: Uint8Array / Uint8ClampedArray expected");
        return 0;
    }
    *p = (uint8_t *)data;
    return 1;
}

static int get_i32(napi_env env, napi_value v, int32_t *out) { return napi_get_value_int32(env, v, out) == napi_ok; }

static napi_value typed(napi_env env, napi_typedarray_type t, size_t n, size_t elem, void **data) {
    napi_value ab, arr;
    napi_create_arraybuffer(env, n * elem, data, &ab);
    napi_create_typedarray(env, t, n, ab, 0, &arr);
    return arr;
}

/* detectRaw(rgba, w, h, grayInR, blob, interval) -> {scale, q, x, y: Int32Array, sum: Float64Array} in emission order (ccv.js:178-246) */
static napi_value DetectRaw(napi_env env, napi_callback_info info) {
    ARGS(6);
    uint8_t *rgba, *blob;
    size_t len, blen;
    int32_t w, h, gray, interval;
    if (!get_u8(env, argv[0], &rgba, &len) || !get_i32(env, argv[1], &w) || !get_i32(env, argv[2], &h) || !get_i32(env, argv[3], &gray) ||
        !get_u8(env, argv[4], &blob, &blen) || !get_i32(env, argv[5], &interval))
        return NULL;
    if (len < (size_t)w * h * 4) {
        napi_throw_error(env, NULL, "oracle_addon.detectRaw: frame shorter than w * h * 4");
        return NULL;
    }
    int64_t cap = 1 << 16;
    ho_hit *hits = (ho_hit *)malloc(sizeof(ho_hit) * (size_t)cap);
    int64_t n = ho_detect_raw(rgba, w, h, gray, blob, blen, interval, hits, cap, NULL);
    if (n < 0 || n > cap) {
        free(hits);
        napi_throw_error(env, NULL, "oracle_addon.detectRaw: ho_detect_raw failed");
        return NULL;
    }
    void *ps, *pq, *px, *py, *pu;
    napi_value out, vs = typed(env, napi_int32_array, (size_t)n, 4, &ps), vq = typed(env, napi_int32_array, (size_t)n, 4, &pq),
                    vx = typed(env, napi_int32_array, (size_t)n, 4, &px), vy = typed(env, napi_int32_array, (size_t)n, 4, &py),
                    vu = typed(env, napi_float64_array, (size_t)n, 8, &pu);
    for (int64_t i = 0; i < n; i++) {
        ((int32_t *)ps)[i] = hits[i].scale;
        ((int32_t *)pq)[i] = hits[i].q;
        ((int32_t *)px)[i] = hits[i].x;
        ((int32_t *)py)[i] = hits[i].y;
        ((double *)pu)[i] = hits[i].sum;
    }
    free(hits);
    napi_create_object(env, &out);
    napi_set_named_property(env, out, "scale", vs);
    napi_set_named_property(env, out, "q", vq);
    napi_set_named_property(env, out, "x", vx);
    napi_set_named_property(env, out, "y", vy);
    napi_set_named_property(env, out, "sum", vu);
    return out;
}

/* grayscale(rgba, w, h): ccv.grayscale in place (ccv.js:22-32) */
static napi_value Grayscale(napi_env env, napi_callback_info info) {
    ARGS(3);
    uint8_t *rgba;
    size_t len;
    int32_t w, h;
    if (!get_u8(env, argv[0], &rgba, &len) || !get_i32(env, argv[1], &w) || !get_i32(env, argv[2], &h)) return NULL;
    if (len >= (size_t)w * h * 4) ho_grayscale_rgba(rgba, w, h);
    return NULL;
}

/* whitebalance(rgba, w, h) -> number (whitebalance.js:5-30) */
static napi_value Whitebalance(napi_env env, napi_callback_info info) {
    ARGS(3);
    uint8_t *rgba;
    size_t len;
    int32_t w, h;
    napi_value out;
    if (!get_u8(env, argv[0], &rgba, &len) || !get_i32(env, argv[1], &w) || !get_i32(env, argv[2], &h)) return NULL;
    napi_create_double(env, ho_whitebalance(rgba, w, h), &out);
    return out;
}

/* csInit(state /Uint8Array of csStateBytes/, rgba, w, h, rx, ry, rw, rh, calcAngles): camshift.Tracker.initTracker (camshift.js:198-211) */
static napi_value CsInit(napi_env env, napi_callback_info info) {
    ARGS(9);
    uint8_t *st, *rgba;
    size_t slen, len;
    int32_t w, h, r[4], ca;
    if (!get_u8(env, argv[0], &st, &slen) || !get_u8(env, argv[1], &rgba, &len) || !get_i32(env, argv[2], &w) || !get_i32(env, argv[3], &h) ||
        !get_i32(env, argv[4], &r[0]) || !get_i32(env, argv[5], &r[1]) || !get_i32(env, argv[6], &r[2]) || !get_i32(env, argv[7], &r[3]) ||
        !get_i32(env, argv[8], &ca))
        return NULL;
    if (slen < sizeof(ho_cs_state) || len < (size_t)w * h * 4) {
        napi_throw_error(env, NULL, "oracle_addon.csInit: buffer too small");
        return NULL;
    }
    ho_cs_init((ho_cs_state *)st, rgba, w, h, r[0], r[1], r[2], r[3], ca);
    return NULL;
}

/* csTrack(state, rgba, w, h) -> Float64Array [x, y, width, height, angle, sw.x, sw.y, sw.width, sw.height] (camshift.js:213-259) */
static napi_value CsTrack(napi_env env, napi_callback_info info) {
    ARGS(4);
    uint8_t *st, *rgba;
    size_t slen, len;
    int32_t w, h;
    if (!get_u8(env, argv[0], &st, &slen) || !get_u8(env, argv[1], &rgba, &len) || !get_i32(env, argv[2], &w) || !get_i32(env, argv[3], &h)) return NULL;
    if (slen < sizeof(ho_cs_state) || len < (size_t)w * h * 4) {
        napi_throw_error(env, NULL, "oracle_addon.csTrack: buffer too small");
        return NULL;
    }
    ho_cs_state *s = (ho_cs_state *)st;
    ho_cs_track(s, rgba, w, h);
    void *p;
    napi_value out = typed(env, napi_float64_array, 9, 8, &p);
    double *d = (double *)p;
    d[0] = s->x, d[1] = s->y, d[2] = s->width, d[3] = s->height, d[4] = s->angle;
    for (int i = 0; i < 4; i++) d[5 + i] = s->sw[i];
    return out;
}

/* bestFace(rgba, w, h, grayInR, blob, interval, minNeighbors) -> Float64Array [x, y, width, height, confidence, neighbors, raw hits]:
 * ccv.detect_objects(..., minNeighbors) + facetrackr's choice (facetrackr.js:157-165: strict '>', first maximum wins); no face: zeros,
 * confidence -10000, neighbors 0 (facetrackr.js:239) */
static napi_value BestFace(napi_env env, napi_callback_info info) {
    ARGS(7);
    uint8_t *rgba, *blob;
    size_t len, blen;
    int32_t w, h, gray, interval, mn;
    if (!get_u8(env, argv[0], &rgba, &len) || !get_i32(env, argv[1], &w) || !get_i32(env, argv[2], &h) || !get_i32(env, argv[3], &gray) ||
        !get_u8(env, argv[4], &blob, &blen) || !get_i32(env, argv[5], &interval) || !get_i32(env, argv[6], &mn))
        return NULL;
    if (len < (size_t)w * h * 4) {
        napi_throw_error(env, NULL, "oracle_addon.bestFace: frame shorter than w * h * 4");
        return NULL;
    }
    ho_cascade c;
    int64_t cap = 1 << 16;
    ho_hit *hits = (ho_hit *)malloc(sizeof(ho_hit) * (size_t)cap);
    int64_t n = ho_detect_raw(rgba, w, h, gray, blob, blen, interval, hits, cap, NULL);
    if (n < 0 || n > cap || ho_parse_cascade(blob, blen, &c) != 0) {
        free(hits);
        napi_throw_error(env, NULL, "oracle_addon.bestFace: ho_detect_raw failed");
        return NULL;
    }
    ho_rect *seq = (ho_rect *)malloc(sizeof(ho_rect) * (size_t)(n + 1)), *grp = (ho_rect *)malloc(sizeof(ho_rect) * (size_t)(n + 1));
    ho_hits_to_rects(hits, n, interval, (int)c.width, (int)c.height, seq);
    int ng = mn > 0 ? ho_group(seq, (int)n, mn, grp) : 0;
    void *p;
    napi_value out = typed(env, napi_float64_array, 7, 8, &p);
    double *d = (double *)p;
    d[0] = d[1] = d[2] = d[3] = d[5] = 0, d[4] = -10000.0, d[6] = (double)n;
    for (int k = 0; k < ng; k++)
        if (k == 0 || grp[k].confidence > d[4])
            d[0] = grp[k].x, d[1] = grp[k].y, d[2] = grp[k].width, d[3] = grp[k].height, d[4] = grp[k].confidence, d[5] = grp[k].neighbors;
    free(hits);
    free(seq);
    free(grp);
    return out;
}

static napi_value Init(napi_env env, napi_value exports) {
    const struct {
        const char *name;
        napi_callback fn;
    } fns[] = {{"detectRaw", DetectRaw}, {"grayscale", Grayscale}, {"whitebalance", Whitebalance}, {"csInit", CsInit}, {"csTrack", CsTrack}, {"bestFace", BestFace}};
    for (size_t i = 0; i < sizeof(fns) / sizeof(fns[0]); i++) {
        napi_value fn;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &fn) != napi_ok) return NULL;
        napi_set_named_property(env, exports, fns[i].name, fn);
    }
    napi_value v;
    napi_create_int32(env, (int32_t)sizeof(ho_cs_state), &v);
    napi_set_named_property(env, exports, "csStateBytes", v);
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
