// ht_napi.cc — thin N-API (raw node_api.h, no node-addon-api) shim over the C ABI in include/headtrackr_hip.h.
// It contains no algorithm: argument unpacking, one C-ABI call, result packing.  The JavaScript facade
// (headtrackr_amd/js/headtrackr.js) builds the reference's API (headtrackr.ccv / camshift / facetrackr) on top of it.
//
//   createContext({device, interval, cascade:<Buffer HTCB>, hitCapacity, queueCapacity}) -> ctx (external)
//   destroy(ctx)
//   setGeometry(ctx, w, h, maxBatch, Int32Array levelDims | null)
//   detect(ctx, Uint8Array rgba, n, w, h, flags)        -> {frame,x,y,scale,q,sum, counts}   (sync; the drop-in path)
//   detectAsync(ctx, rgba, n, w, h, flags)              -> Promise of the same              (napi_create_async_work)
//   grayscale(ctx, Uint8Array rgba, n, w, h)            -> undefined (in place)
//   whitebalance(ctx, Uint8Array rgba, n, w, h)         -> Float64Array(n)
//   camshiftReserve(ctx, nstreams)
//   camshiftInit(ctx, rgba, n, w, h, first, Int32Array rects[4n])
//   camshiftTrack(ctx, rgba, n, w, h, first, calcAngles) -> Float64Array(9n): x,y,width,height,angle,swx,swy,sww,swh
//   info(ctx) -> {levels, windowsPerFrame, pyramidBytesPerFrame}
//   deviceCount() -> number of visible GPUs
//   allgatherBest([ctx0, ctx1, ...], [Float64Array(6*f) per rank: x,y,width,height,confidence,neighbors], framesPerRank)
//        -> Float64Array(6 * nranks * framesPerRank): every rank's best-face rects after the RCCL all-gather (ht_allgather_best_faces)
//
// The pipelined path (what the throughput numbers are made of; every C-ABI export has a JS name):
//   hostAlloc(bytes) -> Uint8Array over PINNED host memory (ht_host_alloc)         deviceAlloc(ctx, bytes) -> device buffer (external)
//   deviceUpload(ctx, dev, byteOffset, Uint8Array)      deviceFree(ctx, dev)
//   upload(ctx, rgba, n, w, h)            ht_upload_frames: bind host frames once, then any number of *Bound calls on them
//   bindDevice(ctx, dev, byteOffset, n)   ht_bind_frames_device
//   uploadAsync(ctx, rgba, n) / swapFrames(ctx)          double-buffered ingest (rgba should come from hostAlloc)
//   detectEnqueue(ctx, flags)             ht_detect_enqueue           detectCollect(ctx) -> hits object (ht_detect_collect)
//   collectBest(ctx, minNeighbors, requeueFlags = -1) -> {best: Float64Array(6 n), hits}   ht_detect_collect_best(_requeue)
//   detectWhitebalance(ctx, n) -> Float64Array(n)        whitebalanceBound(ctx, n) -> Float64Array(n)
//   camshiftInitBound(ctx, n, first, Int32Array rects)   camshiftTrackBound(ctx, n, first, calcAngles, fetch = true) -> Float64Array(9n) | undefined
//   camshiftTrackCollect(ctx, n) -> Float64Array(9n)
//   camshiftTrackSequence(ctx, first, n, calcAngles, dev, Float64Array byteOffsets[ncalls], frameStride, outAll, fetch) -> Float64Array | undefined
//   camshiftSequenceCollect(ctx, n, ncalls, outAll) -> Float64Array
//   framesBound(ctx), framesEnqueued(ctx), graphLaunches(ctx)
#include <node_api.h>

#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <unistd.h>

#include "headtrackr_hip.h"

namespace {

#define NAPI_OK(call)                                                  \
    do {                                                               \
        if ((call) != napi_ok) {                                       \
            napi_throw_error(env, nullptr, "N-API call failed: " #call); \
            return nullptr;                                            \
        }                                                              \
    } while (0)

napi_value throw_ht(napi_env env, ht_ctx *ctx, ht_status st, const char *where) {
    std::string msg = std::string(where) + ": status " + std::to_string(st) + ": " + ht_last_error(ctx);
    napi_throw_error(env, nullptr, msg.c_str());
    return nullptr;
}

// What the JS side holds (a napi external): the context plus the lock that serialises every use of it.  An ht_ctx is not
// thread-safe (include/headtrackr_hip.h) but detectAsync() runs on a libuv pool thread while the JS thread may call any
// synchronous entry point on the same cached context (headtrackr.js shares one context per cascade): every entry point
// takes `mu` for the duration of its C-ABI calls, so overlapping calls run one after the other, in lock-acquisition order.
// The two kinds of handles the addon gives out are napi externals; a tag in front tells them apart, so that a device buffer passed where a
// context is expected (or the other way round) is a TypeError, not a reinterpretation of the other struct's bytes.  Neither kind is ever freed.
constexpr uint32_t SLOT_TAG = 0x4c535448u, DEVBUF_TAG = 0x42445448u;  // "HTSL", "HTDB"
struct Slot {
    uint32_t tag = SLOT_TAG;
    ht_ctx *ctx = nullptr;
    std::recursive_mutex mu;
    napi_env env = nullptr;  // the environment (main thread or a worker_threads Worker) that created the context: its cleanup hook destroys it
};

bool get_slot(napi_env env, napi_value v, Slot **out) {
    void *p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || static_cast<Slot *>(p)->tag != SLOT_TAG) {
        napi_throw_type_error(env, nullptr, "expected a headtrackr_hip context");
        return false;
    }
    *out = static_cast<Slot *>(p);
    return true;
}

// Locks the slot and yields its context; throws (and returns false) when the context was destroyed.
struct Locked {
    std::unique_lock<std::recursive_mutex> lk;
    ht_ctx *ctx = nullptr;
};
bool lock_ctx(napi_env env, napi_value v, Locked *out) {
    Slot *s = nullptr;
    if (!get_slot(env, v, &s)) return false;
    out->lk = std::unique_lock<std::recursive_mutex>(s->mu);
    out->ctx = s->ctx;
    if (!out->ctx) {
        out->lk.unlock();
        napi_throw_error(env, nullptr, "context was destroyed");
        return false;
    }
    return true;
}

// fewer arguments than the entry point needs: a TypeError, not a silent `undefined` (found by tests/js/addon_args.js)
bool too_few(napi_env env, size_t argc, size_t need) {
    if (argc >= need) return false;
    napi_throw_type_error(env, nullptr, ("headtrackr_hip: " + std::to_string(need) + " arguments expected, " + std::to_string(argc) + " given").c_str());
    return true;
}

bool get_i32(napi_env env, napi_value v, int32_t *out) { return napi_get_value_int32(env, v, out) == napi_ok; }

// Uint8Array / Uint8ClampedArray / Buffer -> pointer + length
bool get_bytes(napi_env env, napi_value v, uint8_t **data, size_t *len) {
    bool is_ta = false;
    if (napi_is_typedarray(env, v, &is_ta) == napi_ok && is_ta) {
        napi_typedarray_type t;
        size_t n;
        void *p;
        napi_value ab;
        size_t off;
        if (napi_get_typedarray_info(env, v, &t, &n, &p, &ab, &off) != napi_ok) return false;
        if (t != napi_uint8_array && t != napi_uint8_clamped_array && t != napi_int8_array) return false;
        *data = static_cast<uint8_t *>(p);
        *len = n;
        return true;
    }
    bool is_buf = false;
    if (napi_is_buffer(env, v, &is_buf) == napi_ok && is_buf) {
        void *p;
        if (napi_get_buffer_info(env, v, &p, len) != napi_ok) return false;
        *data = static_cast<uint8_t *>(p);
        return true;
    }
    return false;
}

// Every live Slot, so that the environment's cleanup hook can destroy the contexts while the HIP runtime is still up: finalizers
// of externals may run during environment teardown or not at all, and a context destroyed after the runtime's own static
// destructors crashes the process at exit.
std::mutex g_slots_mu;
std::vector<Slot *> g_slots;

// Cleanup hook of ONE environment (arg = its napi_env): a Worker that exits destroys the contexts IT created, not the main thread's.
// arg == nullptr (exitNow: the whole process is leaving): every environment's contexts.
void env_cleanup(void *arg) {
    std::lock_guard<std::mutex> lk(g_slots_mu);
    for (Slot *s : g_slots) {
        if (arg && s->env != static_cast<napi_env>(arg)) continue;
        std::lock_guard<std::recursive_mutex> l2(s->mu);
        if (s->ctx) ht_destroy(s->ctx);
        s->ctx = nullptr;
    }
}

// hostAlloc registry: hostFree releases exactly the pointers hostAlloc returned, once (a subarray, a second view or a foreign
// Uint8Array must never reach hipHostFree), and detaches the ArrayBuffer so that JS cannot touch the unmapped pages afterwards
std::mutex g_host_mu;
std::vector<std::pair<void *, size_t>> g_host_allocs;

// NO finalizers on the handles this addon hands to JavaScript.  Node 12 runs finalizers that are still pending while it tears the
// environment down, through N-API's own phantom-callback wrapper, and that wrapper crashes inside libnode (SIGSEGV at exit in
// GlobalHandles::InvokeSecondPassPhantomCallbacks -> libnode, seen in one of two runs of tests/js/bench_host.js whatever the callback
// did).  Native resources are therefore released explicitly — destroy(ctx), deviceFree(ctx, buf), hostFree(arr) — and, for whatever is
// still alive at exit, by the environment's cleanup hook (env_cleanup), which is an ordinary callback, not a finalizer.  A handle that
// is dropped without destroy() keeps its GPU memory until the process exits.
napi_value CreateContext(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ht_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.interval = 5;
    napi_value v;
    bool has;
    int32_t i;
    uint8_t *blob = nullptr;
    size_t blob_len = 0;
    if (napi_has_named_property(env, argv[0], "device", &has) == napi_ok && has) {
        NAPI_OK(napi_get_named_property(env, argv[0], "device", &v));
        if (get_i32(env, v, &i)) cfg.device = i;
    }
    if (napi_has_named_property(env, argv[0], "interval", &has) == napi_ok && has) {
        NAPI_OK(napi_get_named_property(env, argv[0], "interval", &v));
        if (get_i32(env, v, &i)) cfg.interval = i;
    }
    if (napi_has_named_property(env, argv[0], "hitCapacity", &has) == napi_ok && has) {
        NAPI_OK(napi_get_named_property(env, argv[0], "hitCapacity", &v));
        if (get_i32(env, v, &i)) cfg.hit_capacity = (uint32_t)i;
    }
    if (napi_has_named_property(env, argv[0], "queueCapacity", &has) == napi_ok && has) {
        NAPI_OK(napi_get_named_property(env, argv[0], "queueCapacity", &v));
        if (get_i32(env, v, &i)) cfg.queue_capacity = (uint32_t)i;
    }
    std::string options;  // ht_config.options: "key=value,..." schedule selectors (tests, A/B runs)
    if (napi_has_named_property(env, argv[0], "options", &has) == napi_ok && has) {
        NAPI_OK(napi_get_named_property(env, argv[0], "options", &v));
        size_t len = 0;
        if (napi_get_value_string_utf8(env, v, nullptr, 0, &len) == napi_ok) {
            options.resize(len + 1);
            NAPI_OK(napi_get_value_string_utf8(env, v, &options[0], len + 1, &len));
            options.resize(len);
            cfg.options = options.c_str();
        }
    }
    NAPI_OK(napi_get_named_property(env, argv[0], "cascade", &v));
    if (!get_bytes(env, v, &blob, &blob_len)) {
        napi_throw_type_error(env, nullptr, "createContext: `cascade` must be a Buffer/Uint8Array holding an HTCB blob");
        return nullptr;
    }
    ht_ctx *ctx = nullptr;
    ht_status st = ht_create(&cfg, blob, blob_len, &ctx);
    if (st != HT_OK) return throw_ht(env, nullptr, st, "ht_create");
    Slot *slot = new Slot();
    slot->ctx = ctx;
    slot->env = env;
    {
        std::lock_guard<std::mutex> lk(g_slots_mu);
        g_slots.push_back(slot);
    }
    napi_value ext;
    NAPI_OK(napi_create_external(env, slot, nullptr, nullptr, &ext));  // no finalizer, see release_slot
    return ext;
}

napi_value Destroy(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    void *p = nullptr;
    if (argc >= 1 && napi_get_value_external(env, argv[0], &p) == napi_ok && p && static_cast<Slot *>(p)->tag == SLOT_TAG) {
        Slot *slot = static_cast<Slot *>(p);
        std::lock_guard<std::recursive_mutex> lk(slot->mu);  // waits for an asynchronous job in flight on this context
        if (slot->ctx) ht_destroy(slot->ctx);
        slot->ctx = nullptr;
    }
    return nullptr;
}

napi_value SetGeometry(napi_env env, napi_callback_info info) {
    size_t argc = 5;
    napi_value argv[5];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    if (!lock_ctx(env, argv[0], &L)) return nullptr;
    ht_ctx *ctx = L.ctx;
    int32_t w, h, nb;
    if (!get_i32(env, argv[1], &w) || !get_i32(env, argv[2], &h) || !get_i32(env, argv[3], &nb)) {
        napi_throw_type_error(env, nullptr, "setGeometry(ctx, w, h, maxBatch, levelDims)");
        return nullptr;
    }
    const int32_t *dims = nullptr;
    int32_t nlev = 0;
    bool is_ta = false;
    if (argc > 4 && napi_is_typedarray(env, argv[4], &is_ta) == napi_ok && is_ta) {
        napi_typedarray_type t;
        size_t n;
        void *p;
        napi_value ab;
        size_t off;
        NAPI_OK(napi_get_typedarray_info(env, argv[4], &t, &n, &p, &ab, &off));
        if (t != napi_int32_array || (n & 1)) {
            napi_throw_type_error(env, nullptr, "levelDims must be an Int32Array [w0,h0,w1,h1,...]");
            return nullptr;
        }
        dims = static_cast<const int32_t *>(p);
        nlev = (int32_t)(n / 2);
    }
    ht_status st = ht_set_geometry(ctx, w, h, nb, dims, nlev);
    if (st != HT_OK) return throw_ht(env, ctx, st, "ht_set_geometry");
    return nullptr;
}

// ---- detect ----------------------------------------------------------------------------------------------------

struct DetectJob {
    Slot *slot = nullptr;
    uint8_t *rgba = nullptr;
    int32_t n = 0, w = 0, h = 0;
    uint32_t flags = 0;
    std::vector<ht_hit> hits;
    std::vector<uint32_t> counts;
    uint32_t total = 0;
    ht_status st = HT_OK;
    std::string err;
    napi_async_work work = nullptr;
    napi_deferred deferred = nullptr;
    napi_ref rgba_ref = nullptr;
    napi_ref ctx_ref = nullptr;  // keeps the context external (and so the Slot) alive until the job completed
};

void run_detect(DetectJob *j) {
    std::lock_guard<std::recursive_mutex> lk(j->slot->mu);
    ht_ctx *ctx = j->slot->ctx;
    if (!ctx) {
        j->st = HT_ERR_STATE;
        j->err = "context was destroyed";
        return;
    }
    uint32_t cap = 4096;
    for (int attempt = 0; attempt < 3; attempt++) {
        j->hits.resize(cap);
        j->counts.assign((size_t)j->n, 0);
        j->st = ht_detect_batch(ctx, j->rgba, j->n, j->w, j->h, (size_t)j->w * j->h * 4, j->flags, j->hits.data(), cap, j->counts.data(), &j->total);
        if (j->st == HT_ERR_CAPACITY && j->total > cap) {  // caller buffer too small: retry with the exact size
            cap = j->total;
            continue;
        }
        break;
    }
    if (j->st != HT_OK) j->err = ht_last_error(ctx);
}

napi_value pack_hits(napi_env env, const DetectJob &j) {
    napi_value obj;
    NAPI_OK(napi_create_object(env, &obj));
    const size_t n = j.total;
    struct Col {
        const char *name;
        napi_typedarray_type type;
        size_t esz;
    } cols[] = {{"frame", napi_uint32_array, 4}, {"x", napi_uint16_array, 2}, {"y", napi_uint16_array, 2},
                {"scale", napi_uint8_array, 1},  {"q", napi_uint8_array, 1},   {"sum", napi_float64_array, 8}};
    for (const Col &c : cols) {
        napi_value ab, ta;
        void *p = nullptr;
        NAPI_OK(napi_create_arraybuffer(env, n * c.esz, &p, &ab));
        for (size_t i = 0; i < n; i++) {
            const ht_hit &h = j.hits[i];
            if (c.name[0] == 'f') static_cast<uint32_t *>(p)[i] = h.frame;
            else if (c.name[0] == 'x') static_cast<uint16_t *>(p)[i] = h.x;
            else if (c.name[0] == 'y') static_cast<uint16_t *>(p)[i] = h.y;
            else if (c.name[0] == 's' && c.name[1] == 'c') static_cast<uint8_t *>(p)[i] = h.scale;
            else if (c.name[0] == 'q') static_cast<uint8_t *>(p)[i] = h.q;
            else static_cast<double *>(p)[i] = h.sum;
        }
        NAPI_OK(napi_create_typedarray(env, c.type, n, ab, 0, &ta));
        NAPI_OK(napi_set_named_property(env, obj, c.name, ta));
    }
    napi_value ab, ta;
    void *p = nullptr;
    NAPI_OK(napi_create_arraybuffer(env, j.counts.size() * 4, &p, &ab));
    if (!j.counts.empty()) std::memcpy(p, j.counts.data(), j.counts.size() * 4);
    NAPI_OK(napi_create_typedarray(env, napi_uint32_array, j.counts.size(), ab, 0, &ta));
    NAPI_OK(napi_set_named_property(env, obj, "counts", ta));
    return obj;
}

bool parse_detect_args(napi_env env, napi_callback_info info, DetectJob *j, napi_value *rgba_val, napi_value *ctx_val = nullptr) {
    size_t argc = 6;
    napi_value argv[6];
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < 5) {
        napi_throw_type_error(env, nullptr, "detect(ctx, rgba, n, w, h, flags)");
        return false;
    }
    if (!get_slot(env, argv[0], &j->slot)) return false;
    if (ctx_val) *ctx_val = argv[0];
    size_t len = 0;
    int32_t fl = 0;
    if (!get_bytes(env, argv[1], &j->rgba, &len) || !get_i32(env, argv[2], &j->n) || !get_i32(env, argv[3], &j->w) || !get_i32(env, argv[4], &j->h)) {
        napi_throw_type_error(env, nullptr, "detect(ctx, rgba, n, w, h, flags): bad argument");
        return false;
    }
    if (argc > 5) get_i32(env, argv[5], &fl);
    j->flags = (uint32_t)fl;
    if (j->n <= 0 || j->w <= 0 || j->h <= 0 || len < (size_t)j->n * j->w * j->h * 4) {
        napi_throw_range_error(env, nullptr, "detect: rgba buffer smaller than n*w*h*4");
        return false;
    }
    if (rgba_val) *rgba_val = argv[1];
    return true;
}

napi_value Detect(napi_env env, napi_callback_info info) {
    DetectJob j;
    if (!parse_detect_args(env, info, &j, nullptr)) return nullptr;
    run_detect(&j);
    if (j.st != HT_OK) {
        napi_throw_error(env, nullptr, ("ht_detect_batch: status " + std::to_string(j.st) + ": " + j.err).c_str());
        return nullptr;
    }
    return pack_hits(env, j);
}

void detect_execute(napi_env, void *data) { run_detect(static_cast<DetectJob *>(data)); }

void detect_complete(napi_env env, napi_status, void *data) {
    DetectJob *j = static_cast<DetectJob *>(data);
    if (j->st == HT_OK) {
        napi_value v = pack_hits(env, *j);
        napi_resolve_deferred(env, j->deferred, v);
    } else {
        napi_value msg, err;
        std::string m = "ht_detect_batch: status " + std::to_string(j->st) + ": " + j->err;
        napi_create_string_utf8(env, m.c_str(), NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, nullptr, msg, &err);
        napi_reject_deferred(env, j->deferred, err);
    }
    napi_delete_reference(env, j->rgba_ref);
    napi_delete_reference(env, j->ctx_ref);
    napi_delete_async_work(env, j->work);
    delete j;
}

napi_value DetectAsync(napi_env env, napi_callback_info info) {
    DetectJob *j = new DetectJob();
    napi_value rgba_val, ctx_val;
    if (!parse_detect_args(env, info, j, &rgba_val, &ctx_val)) {
        delete j;
        return nullptr;
    }
    napi_value promise, name;
    NAPI_OK(napi_create_promise(env, &j->deferred, &promise));
    NAPI_OK(napi_create_reference(env, rgba_val, 1, &j->rgba_ref));  // keep the frame buffer alive while the GPU works
    NAPI_OK(napi_create_reference(env, ctx_val, 1, &j->ctx_ref));
    NAPI_OK(napi_create_string_utf8(env, "headtrackr_hip.detect", NAPI_AUTO_LENGTH, &name));
    NAPI_OK(napi_create_async_work(env, nullptr, name, detect_execute, detect_complete, j, &j->work));
    NAPI_OK(napi_queue_async_work(env, j->work));
    return promise;
}

// ---- grayscale / whitebalance -------------------------------------------------------------------------------------

struct FrameArgs {
    Locked L;  // the context stays locked for the lifetime of the argument block = the whole entry point
    ht_ctx *ctx;
    uint8_t *rgba;
    int32_t n, w, h;
};

bool parse_frames(napi_env env, napi_value *argv, FrameArgs *a) {
    size_t len = 0;
    if (!lock_ctx(env, argv[0], &a->L)) return false;
    a->ctx = a->L.ctx;
    if (!get_bytes(env, argv[1], &a->rgba, &len) || !get_i32(env, argv[2], &a->n) || !get_i32(env, argv[3], &a->w) || !get_i32(env, argv[4], &a->h) ||
        a->n <= 0 || a->w <= 0 || a->h <= 0 || len < (size_t)a->n * a->w * a->h * 4) {
        napi_throw_type_error(env, nullptr, "expected (ctx, Uint8Array rgba, n, w, h, ...) with rgba.length >= n*w*h*4");
        return false;
    }
    return true;
}

ht_status bind_host_frames(const FrameArgs &a) {
    ht_status st = ht_set_geometry(a.ctx, a.w, a.h, a.n, nullptr, 0);  // no-op when unchanged
    if (st != HT_OK) return st;
    return ht_upload_frames(a.ctx, a.rgba, a.n, (size_t)a.w * a.h * 4);
}

napi_value Grayscale(napi_env env, napi_callback_info info) {
    size_t argc = 5;
    napi_value argv[5];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    FrameArgs a;
    if (too_few(env, argc, 5) || !parse_frames(env, argv, &a)) return nullptr;
    ht_status st = ht_grayscale_batch(a.ctx, a.rgba, a.n, a.w, a.h, (size_t)a.w * a.h * 4);
    if (st != HT_OK) return throw_ht(env, a.ctx, st, "ht_grayscale_batch");
    return nullptr;
}

napi_value Whitebalance(napi_env env, napi_callback_info info) {
    size_t argc = 5;
    napi_value argv[5];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    FrameArgs a;
    if (too_few(env, argc, 5) || !parse_frames(env, argv, &a)) return nullptr;
    ht_status st = bind_host_frames(a);
    if (st != HT_OK) return throw_ht(env, a.ctx, st, "ht_upload_frames");
    napi_value ab, ta;
    void *p = nullptr;
    NAPI_OK(napi_create_arraybuffer(env, (size_t)a.n * 8, &p, &ab));
    st = ht_whitebalance_batch(a.ctx, static_cast<double *>(p), a.n);
    if (st != HT_OK) return throw_ht(env, a.ctx, st, "ht_whitebalance_batch");
    NAPI_OK(napi_create_typedarray(env, napi_float64_array, (size_t)a.n, ab, 0, &ta));
    return ta;
}

// ---- camshift ---------------------------------------------------------------------------------------------------

napi_value CamshiftReserve(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    int32_t n;
    if (!lock_ctx(env, argv[0], &L)) return nullptr;
    ht_ctx *ctx = L.ctx;
    if (!get_i32(env, argv[1], &n)) {
        napi_throw_type_error(env, nullptr, "camshiftReserve(ctx, nstreams)");
        return nullptr;
    }
    ht_status st = ht_camshift_reserve(ctx, n);
    if (st != HT_OK) return throw_ht(env, ctx, st, "ht_camshift_reserve");
    return nullptr;
}

napi_value CamshiftInit(napi_env env, napi_callback_info info) {
    size_t argc = 7;
    napi_value argv[7];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    FrameArgs a;
    if (too_few(env, argc, 7) || !parse_frames(env, argv, &a)) return nullptr;
    int32_t first;
    napi_typedarray_type t;
    size_t n;
    void *p;
    napi_value ab;
    size_t off;
    if (!get_i32(env, argv[5], &first) || napi_get_typedarray_info(env, argv[6], &t, &n, &p, &ab, &off) != napi_ok || t != napi_int32_array ||
        n < (size_t)a.n * 4) {
        napi_throw_type_error(env, nullptr, "camshiftInit(ctx, rgba, n, w, h, first, Int32Array rects[4n])");
        return nullptr;
    }
    ht_status st = bind_host_frames(a);
    if (st != HT_OK) return throw_ht(env, a.ctx, st, "ht_upload_frames");
    st = ht_camshift_init_batch(a.ctx, first, a.n, static_cast<const ht_cs_rect *>(p));
    if (st != HT_OK) return throw_ht(env, a.ctx, st, "ht_camshift_init_batch");
    return nullptr;
}

napi_value CamshiftTrack(napi_env env, napi_callback_info info) {
    size_t argc = 7;
    napi_value argv[7];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    FrameArgs a;
    if (too_few(env, argc, 7) || !parse_frames(env, argv, &a)) return nullptr;
    int32_t first, calc;
    if (!get_i32(env, argv[5], &first) || !get_i32(env, argv[6], &calc)) {
        napi_throw_type_error(env, nullptr, "camshiftTrack(ctx, rgba, n, w, h, first, calcAngles)");
        return nullptr;
    }
    ht_status st = bind_host_frames(a);
    if (st != HT_OK) return throw_ht(env, a.ctx, st, "ht_upload_frames");
    std::vector<ht_cs_trackobj> out((size_t)a.n);
    st = ht_camshift_track_batch(a.ctx, first, a.n, calc, out.data());
    if (st != HT_OK) return throw_ht(env, a.ctx, st, "ht_camshift_track_batch");
    napi_value ab, ta;
    void *p = nullptr;
    NAPI_OK(napi_create_arraybuffer(env, (size_t)a.n * 9 * 8, &p, &ab));
    double *d = static_cast<double *>(p);
    for (int i = 0; i < a.n; i++) {
        const ht_cs_trackobj &o = out[i];
        double *r = d + 9 * i;
        r[0] = o.x, r[1] = o.y, r[2] = o.width, r[3] = o.height, r[4] = o.angle;
        r[5] = o.sw_x, r[6] = o.sw_y, r[7] = o.sw_width, r[8] = o.sw_height;
    }
    NAPI_OK(napi_create_typedarray(env, napi_float64_array, (size_t)a.n * 9, ab, 0, &ta));
    return ta;
}

napi_value Info(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    if (!lock_ctx(env, argv[0], &L)) return nullptr;
    ht_ctx *ctx = L.ctx;
    napi_value obj, v;
    NAPI_OK(napi_create_object(env, &obj));
    NAPI_OK(napi_create_int32(env, ht_num_levels(ctx), &v));
    NAPI_OK(napi_set_named_property(env, obj, "levels", v));
    NAPI_OK(napi_create_double(env, (double)ht_windows_per_frame(ctx), &v));
    NAPI_OK(napi_set_named_property(env, obj, "windowsPerFrame", v));
    NAPI_OK(napi_create_double(env, (double)ht_pyramid_bytes_per_frame(ctx), &v));
    NAPI_OK(napi_set_named_property(env, obj, "pyramidBytesPerFrame", v));
    return obj;
}

napi_value DeviceCount(napi_env env, napi_callback_info) {
    napi_value v;
    NAPI_OK(napi_create_int32(env, ht_device_count(), &v));
    return v;
}

napi_value AllgatherBest(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    uint32_t nctx = 0, nbest = 0;
    int32_t per = 0;
    if (argc < 3 || napi_get_array_length(env, argv[0], &nctx) != napi_ok || napi_get_array_length(env, argv[1], &nbest) != napi_ok || nctx == 0 ||
        nctx != nbest || !get_i32(env, argv[2], &per) || per <= 0) {
        napi_throw_type_error(env, nullptr, "allgatherBest([ctx...], [Float64Array...], framesPerRank)");
        return nullptr;
    }
    std::vector<Locked> locks(nctx);  // every rank's context stays locked for the exchange (slots are distinct: no lock-order issue
                                      // as long as callers pass the contexts in the same order, which headtrackr.js does)
    std::vector<ht_ctx *> ctxs(nctx);
    std::vector<std::vector<ht_rect>> rects(nctx, std::vector<ht_rect>((size_t)per));
    std::vector<const ht_rect *> ptrs(nctx);
    for (uint32_t i = 0; i < nctx; i++) {
        napi_value cv, bv;
        NAPI_OK(napi_get_element(env, argv[0], i, &cv));
        NAPI_OK(napi_get_element(env, argv[1], i, &bv));
        Slot *slot = nullptr;
        if (!get_slot(env, cv, &slot)) return nullptr;
        for (uint32_t k = 0; k < i; k++)
            if (locks[k].lk.mutex() == &slot->mu) {
                napi_throw_error(env, nullptr, "allgatherBest: the same context was passed twice (one context per GPU)");
                return nullptr;
            }
        if (!lock_ctx(env, cv, &locks[i])) return nullptr;
        ctxs[i] = locks[i].ctx;
        napi_typedarray_type t;
        size_t n;
        void *p;
        napi_value ab;
        size_t off;
        if (napi_get_typedarray_info(env, bv, &t, &n, &p, &ab, &off) != napi_ok || t != napi_float64_array || n < (size_t)per * 6) {
            napi_throw_type_error(env, nullptr, "allgatherBest: every rank needs a Float64Array of 6 * framesPerRank numbers");
            return nullptr;
        }
        const double *d = static_cast<const double *>(p);
        for (int f = 0; f < per; f++) {
            ht_rect &r = rects[i][(size_t)f];
            r.x = d[6 * f], r.y = d[6 * f + 1], r.width = d[6 * f + 2], r.height = d[6 * f + 3], r.confidence = d[6 * f + 4];
            r.neighbors = (int32_t)d[6 * f + 5];
            r.reserved = 0;
        }
        ptrs[i] = rects[i].data();
    }
    std::vector<ht_rect> out((size_t)nctx * per);
    ht_status st = ht_allgather_best_faces(ctxs.data(), (int32_t)nctx, ptrs.data(), per, out.data());
    if (st != HT_OK) return throw_ht(env, ctxs[0], st, "ht_allgather_best_faces");
    napi_value ab, ta;
    void *p = nullptr;
    NAPI_OK(napi_create_arraybuffer(env, out.size() * 6 * 8, &p, &ab));
    double *d = static_cast<double *>(p);
    for (size_t k = 0; k < out.size(); k++) {
        d[6 * k] = out[k].x, d[6 * k + 1] = out[k].y, d[6 * k + 2] = out[k].width, d[6 * k + 3] = out[k].height, d[6 * k + 4] = out[k].confidence;
        d[6 * k + 5] = out[k].neighbors;
    }
    NAPI_OK(napi_create_typedarray(env, napi_float64_array, out.size() * 6, ab, 0, &ta));
    return ta;
}

// ---- pipelined path ------------------------------------------------------------------------------------------------------------

napi_value HostAlloc(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    double bytes = 0;
    if (argc < 1 || napi_get_value_double(env, argv[0], &bytes) != napi_ok || !(bytes >= 1) || bytes > 1e12) {
        napi_throw_type_error(env, nullptr, "hostAlloc(bytes)");
        return nullptr;
    }
    void *p = nullptr;
    ht_status st = ht_host_alloc((size_t)bytes, &p);
    if (st != HT_OK) return throw_ht(env, nullptr, st, "ht_host_alloc");
    napi_value ab, ta;
    if (napi_create_external_arraybuffer(env, p, (size_t)bytes, nullptr, nullptr, &ab) != napi_ok) {  // no finalizer: hostFree(arr)
        ht_host_free(p);
        napi_throw_error(env, nullptr, "hostAlloc: napi_create_external_arraybuffer failed");
        return nullptr;
    }
    NAPI_OK(napi_create_typedarray(env, napi_uint8_array, (size_t)bytes, ab, 0, &ta));
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        g_host_allocs.emplace_back(p, (size_t)bytes);
    }
    return ta;
}

// a device buffer: freed explicitly (deviceFree) or, at the latest, when the JS handle is collected — through the context it was
// allocated on, which the handle keeps alive
struct DevBuf {
    uint32_t tag = DEVBUF_TAG;
    Slot *slot = nullptr;  // Slots are never freed (a few bytes per context): a JS handle may outlive destroy()
    void *ptr = nullptr;
    size_t bytes = 0;
};
bool get_devbuf(napi_env env, napi_value v, DevBuf **out) {
    void *p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || static_cast<DevBuf *>(p)->tag != DEVBUF_TAG || !static_cast<DevBuf *>(p)->ptr ||
        !static_cast<DevBuf *>(p)->slot->ctx) {
        napi_throw_type_error(env, nullptr, "expected a live device buffer (deviceAlloc) of a live context");
        return false;
    }
    *out = static_cast<DevBuf *>(p);
    return true;
}

// hostFree(Uint8Array from hostAlloc): releases the pinned memory; the array must not be used afterwards
napi_value HostFree(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    napi_typedarray_type t;
    size_t len = 0, off = 0;
    void *p = nullptr;
    napi_value ab;
    bool is_ta = false;
    if (argc < 1 || napi_is_typedarray(env, argv[0], &is_ta) != napi_ok || !is_ta || napi_get_typedarray_info(env, argv[0], &t, &len, &p, &ab, &off) != napi_ok || !p) {
        napi_throw_type_error(env, nullptr, "hostFree(Uint8Array returned by hostAlloc)");
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        auto it = g_host_allocs.end();
        if (off == 0)
            for (auto i = g_host_allocs.begin(); i != g_host_allocs.end(); ++i)
                if (i->first == p && i->second == len) it = i;
        if (it == g_host_allocs.end()) {  // a subarray / second view / foreign array / already freed: nothing is released
            napi_throw_error(env, nullptr, "hostFree: not a live hostAlloc() array (pass the array hostAlloc returned, whole, once)");
            return nullptr;
        }
        g_host_allocs.erase(it);
    }
    ht_host_free(p);
    (void)napi_detach_arraybuffer(env, ab);  // every view now has length 0: no use-after-free from JavaScript
    return nullptr;
}

// exitNow(code): destroys every live context (stream synchronised, device memory freed) and leaves with _exit — no atexit handlers,
// no static destructors, no environment teardown.  For hosts that are done: a Node 12 process that used HIP from libuv pool threads
// (detectAsync) has been seen to crash with SIGSEGV inside the runtime's own exit path after a complete, correct run.
napi_value ExitNow(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    int32_t code = 0;
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) == napi_ok && argc > 0) get_i32(env, argv[0], &code);
    env_cleanup(nullptr);
    fflush(stdout);
    fflush(stderr);
    _exit(code);
    return nullptr;
}

napi_value DeviceAlloc(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    double bytes = 0;
    if (too_few(env, argc, 2) || !lock_ctx(env, argv[0], &L)) return nullptr;
    if (napi_get_value_double(env, argv[1], &bytes) != napi_ok || !(bytes >= 1) || bytes > 2.5e11) {
        napi_throw_type_error(env, nullptr, "deviceAlloc(ctx, bytes)");
        return nullptr;
    }
    DevBuf *d = new DevBuf();
    ht_status st = ht_device_alloc(L.ctx, (size_t)bytes, &d->ptr);
    if (st != HT_OK) {
        delete d;
        return throw_ht(env, L.ctx, st, "ht_device_alloc");
    }
    get_slot(env, argv[0], &d->slot);
    d->bytes = (size_t)bytes;
    napi_value ext;
    NAPI_OK(napi_create_external(env, d, nullptr, nullptr, &ext));  // no finalizer: freed by deviceFree or with its context
    return ext;
}

napi_value DeviceFree(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    DevBuf *d = nullptr;
    if (too_few(env, argc, 2) || !lock_ctx(env, argv[0], &L) || !get_devbuf(env, argv[1], &d)) return nullptr;
    ht_status st = ht_device_free(L.ctx, d->ptr);
    if (st != HT_OK) return throw_ht(env, L.ctx, st, "ht_device_free");  // e.g. the wrong context, or still bound elsewhere: the handle stays valid
    d->ptr = nullptr;
    return nullptr;
}

bool get_offset(napi_env env, napi_value v, size_t *out) {
    double d = 0;
    if (napi_get_value_double(env, v, &d) != napi_ok || !(d >= 0) || d > 2.5e11) return false;
    *out = (size_t)d;
    return true;
}

napi_value DeviceUpload(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    DevBuf *d = nullptr;
    uint8_t *src = nullptr;
    size_t len = 0, off = 0;
    if (too_few(env, argc, 4) || !lock_ctx(env, argv[0], &L) || !get_devbuf(env, argv[1], &d)) return nullptr;
    if (!get_offset(env, argv[2], &off) || !get_bytes(env, argv[3], &src, &len) || off + len > d->bytes) {
        napi_throw_range_error(env, nullptr, "deviceUpload(ctx, dev, byteOffset, Uint8Array): outside the device buffer");
        return nullptr;
    }
    ht_status st = ht_device_upload(L.ctx, static_cast<char *>(d->ptr) + off, src, len);
    if (st != HT_OK) return throw_ht(env, L.ctx, st, "ht_device_upload");
    return nullptr;
}

napi_value Upload(napi_env env, napi_callback_info info) {
    size_t argc = 5;
    napi_value argv[5];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    FrameArgs a;
    if (too_few(env, argc, 5) || !parse_frames(env, argv, &a)) return nullptr;
    ht_status st = bind_host_frames(a);
    if (st != HT_OK) return throw_ht(env, a.ctx, st, "ht_upload_frames");
    return nullptr;
}

napi_value BindDevice(napi_env env, napi_callback_info info) {
    size_t argc = 5;
    napi_value argv[5];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    DevBuf *d = nullptr;
    size_t off = 0, stride = 0;
    int32_t n = 0;
    if (too_few(env, argc, 5) || !lock_ctx(env, argv[0], &L) || !get_devbuf(env, argv[1], &d)) return nullptr;
    if (!get_offset(env, argv[2], &off) || !get_i32(env, argv[3], &n) || !get_offset(env, argv[4], &stride) || n <= 0 || off + (size_t)n * stride > d->bytes) {
        napi_throw_range_error(env, nullptr, "bindDevice(ctx, dev, byteOffset, n, frameStride): outside the device buffer");
        return nullptr;
    }
    ht_status st = ht_bind_frames_device(L.ctx, static_cast<char *>(d->ptr) + off, n, stride);
    if (st != HT_OK) return throw_ht(env, L.ctx, st, "ht_bind_frames_device");
    return nullptr;
}

napi_value UploadAsync(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    uint8_t *src = nullptr;
    size_t len = 0;
    int32_t n = 0;
    if (too_few(env, argc, 3) || !lock_ctx(env, argv[0], &L)) return nullptr;
    if (!get_bytes(env, argv[1], &src, &len) || !get_i32(env, argv[2], &n) || n <= 0 || len % (size_t)n) {
        napi_throw_type_error(env, nullptr, "uploadAsync(ctx, Uint8Array rgba (n frames, ideally from hostAlloc), n)");
        return nullptr;
    }
    ht_status st = ht_upload_frames_async(L.ctx, src, n, len / (size_t)n);  // rgba must stay untouched until swapFrames
    if (st != HT_OK) return throw_ht(env, L.ctx, st, "ht_upload_frames_async");
    return nullptr;
}

napi_value SwapFrames(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    if (too_few(env, argc, 1) || !lock_ctx(env, argv[0], &L)) return nullptr;
    ht_status st = ht_swap_frames(L.ctx);
    if (st != HT_OK) return throw_ht(env, L.ctx, st, "ht_swap_frames");
    return nullptr;
}

napi_value DetectEnqueue(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    int32_t fl = 0;
    if (too_few(env, argc, 1) || !lock_ctx(env, argv[0], &L)) return nullptr;
    if (argc > 1) get_i32(env, argv[1], &fl);
    ht_status st = ht_detect_enqueue(L.ctx, (uint32_t)fl);
    if (st != HT_OK) return throw_ht(env, L.ctx, st, "ht_detect_enqueue");
    return nullptr;
}

napi_value DetectCollect(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    if (too_few(env, argc, 1) || !lock_ctx(env, argv[0], &L)) return nullptr;
    DetectJob j;
    j.n = ht_frames_enqueued(L.ctx);  // the batch in flight, not whatever is bound by now
    j.hits.resize(1u << 16);
    j.counts.assign((size_t)j.n, 0);
    j.st = ht_detect_collect(L.ctx, j.hits.data(), (uint32_t)j.hits.size(), j.counts.data(), &j.total);
    if (j.st == HT_ERR_CAPACITY && j.total > j.hits.size()) {
        napi_throw_error(env, nullptr, "detectCollect: more than 65536 raw hits in one batch; use collectBest or smaller batches");
        return nullptr;
    }
    if (j.st != HT_OK) return throw_ht(env, L.ctx, j.st, "ht_detect_collect");
    return pack_hits(env, j);
}

napi_value CollectBest(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    int32_t mn = 1, rq = -1;
    if (too_few(env, argc, 1) || !lock_ctx(env, argv[0], &L)) return nullptr;
    if (argc > 1) get_i32(env, argv[1], &mn);
    if (argc > 2) get_i32(env, argv[2], &rq);
    const int32_t n = ht_frames_enqueued(L.ctx);
    std::vector<ht_rect> best((size_t)(n > 0 ? n : 1));
    uint32_t total = 0;
    ht_status st = rq >= 0 ? ht_detect_collect_best_requeue(L.ctx, mn, best.data(), &total, (uint32_t)rq) : ht_detect_collect_best(L.ctx, mn, best.data(), &total);
    if (st != HT_OK) return throw_ht(env, L.ctx, st, "ht_detect_collect_best");
    napi_value obj, ab, ta, v;
    void *p = nullptr;
    NAPI_OK(napi_create_object(env, &obj));
    NAPI_OK(napi_create_arraybuffer(env, (size_t)n * 6 * 8, &p, &ab));
    double *d = static_cast<double *>(p);
    for (int i = 0; i < n; i++) {
        const ht_rect &r = best[(size_t)i];
        d[6 * i] = r.x, d[6 * i + 1] = r.y, d[6 * i + 2] = r.width, d[6 * i + 3] = r.height, d[6 * i + 4] = r.confidence, d[6 * i + 5] = r.neighbors;
    }
    NAPI_OK(napi_create_typedarray(env, napi_float64_array, (size_t)n * 6, ab, 0, &ta));
    NAPI_OK(napi_set_named_property(env, obj, "best", ta));
    NAPI_OK(napi_create_uint32(env, total, &v));
    NAPI_OK(napi_set_named_property(env, obj, "hits", v));
    return obj;
}

napi_value f64_result(napi_env env, const double *src, size_t n) {
    napi_value ab, ta;
    void *p = nullptr;
    NAPI_OK(napi_create_arraybuffer(env, n * 8, &p, &ab));
    if (n) std::memcpy(p, src, n * 8);
    NAPI_OK(napi_create_typedarray(env, napi_float64_array, n, ab, 0, &ta));
    return ta;
}

napi_value wb_common(napi_env env, napi_callback_info info, bool fused) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    int32_t n = 0;
    if (too_few(env, argc, 2) || !lock_ctx(env, argv[0], &L)) return nullptr;
    if (!get_i32(env, argv[1], &n) || n <= 0) {
        napi_throw_type_error(env, nullptr, "(ctx, n)");
        return nullptr;
    }
    std::vector<double> out((size_t)n);
    ht_status st = fused ? ht_detect_whitebalance(L.ctx, out.data(), n) : ht_whitebalance_batch(L.ctx, out.data(), n);
    if (st != HT_OK) return throw_ht(env, L.ctx, st, fused ? "ht_detect_whitebalance" : "ht_whitebalance_batch");
    return f64_result(env, out.data(), out.size());
}
napi_value DetectWhitebalance(napi_env env, napi_callback_info info) { return wb_common(env, info, true); }
napi_value WhitebalanceBound(napi_env env, napi_callback_info info) { return wb_common(env, info, false); }

napi_value trackobjs_result(napi_env env, const std::vector<ht_cs_trackobj> &out) {
    std::vector<double> d(out.size() * 9);
    for (size_t i = 0; i < out.size(); i++) {
        const ht_cs_trackobj &o = out[i];
        double *r = d.data() + 9 * i;
        r[0] = o.x, r[1] = o.y, r[2] = o.width, r[3] = o.height, r[4] = o.angle;
        r[5] = o.sw_x, r[6] = o.sw_y, r[7] = o.sw_width, r[8] = o.sw_height;
    }
    return f64_result(env, d.data(), d.size());
}

napi_value CamshiftInitBound(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    int32_t n = 0, first = 0;
    napi_typedarray_type t;
    size_t len;
    void *p;
    napi_value ab;
    size_t off;
    if (too_few(env, argc, 4) || !lock_ctx(env, argv[0], &L)) return nullptr;
    if (!get_i32(env, argv[1], &n) || !get_i32(env, argv[2], &first) || n <= 0 || napi_get_typedarray_info(env, argv[3], &t, &len, &p, &ab, &off) != napi_ok ||
        t != napi_int32_array || len < (size_t)n * 4) {
        napi_throw_type_error(env, nullptr, "camshiftInitBound(ctx, n, first, Int32Array rects[4n])");
        return nullptr;
    }
    ht_status st = ht_camshift_init_batch(L.ctx, first, n, static_cast<const ht_cs_rect *>(p));
    if (st != HT_OK) return throw_ht(env, L.ctx, st, "ht_camshift_init_batch");
    return nullptr;
}

napi_value CamshiftTrackBound(napi_env env, napi_callback_info info) {
    size_t argc = 5;
    napi_value argv[5];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    int32_t n = 0, first = 0, calc = 1;
    bool fetch = true;
    if (too_few(env, argc, 4) || !lock_ctx(env, argv[0], &L)) return nullptr;
    if (!get_i32(env, argv[1], &n) || !get_i32(env, argv[2], &first) || !get_i32(env, argv[3], &calc) || n <= 0) {
        napi_throw_type_error(env, nullptr, "camshiftTrackBound(ctx, n, first, calcAngles, fetch)");
        return nullptr;
    }
    if (argc > 4) napi_get_value_bool(env, argv[4], &fetch);
    std::vector<ht_cs_trackobj> out((size_t)n);
    ht_status st = ht_camshift_track_batch(L.ctx, first, n, calc, fetch ? out.data() : nullptr);
    if (st != HT_OK) return throw_ht(env, L.ctx, st, "ht_camshift_track_batch");
    if (!fetch) return nullptr;
    return trackobjs_result(env, out);
}

napi_value CamshiftTrackCollect(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    int32_t n = 0;
    if (too_few(env, argc, 2) || !lock_ctx(env, argv[0], &L)) return nullptr;
    if (!get_i32(env, argv[1], &n) || n <= 0) {
        napi_throw_type_error(env, nullptr, "camshiftTrackCollect(ctx, n)");
        return nullptr;
    }
    std::vector<ht_cs_trackobj> out((size_t)n);
    ht_status st = ht_camshift_track_collect(L.ctx, n, out.data());
    if (st != HT_OK) return throw_ht(env, L.ctx, st, "ht_camshift_track_collect");
    return trackobjs_result(env, out);
}

napi_value CamshiftTrackSequence(napi_env env, napi_callback_info info) {
    size_t argc = 9;
    napi_value argv[9];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    DevBuf *d = nullptr;
    int32_t first = 0, n = 0, calc = 1;
    size_t stride = 0;
    bool out_all = false, fetch = true;
    napi_typedarray_type t;
    size_t ncalls;
    void *p;
    napi_value ab;
    size_t off;
    if (too_few(env, argc, 7) || !lock_ctx(env, argv[0], &L)) return nullptr;
    if (!get_i32(env, argv[1], &first) || !get_i32(env, argv[2], &n) || !get_i32(env, argv[3], &calc) || !get_devbuf(env, argv[4], &d)) return nullptr;
    if (napi_get_typedarray_info(env, argv[5], &t, &ncalls, &p, &ab, &off) != napi_ok || t != napi_float64_array || ncalls == 0 || ncalls > 100000 ||
        !get_offset(env, argv[6], &stride) || n <= 0) {
        napi_throw_type_error(env, nullptr, "camshiftTrackSequence(ctx, first, n, calcAngles, dev, Float64Array byteOffsets, frameStride, outAll, fetch)");
        return nullptr;
    }
    if (argc > 7) napi_get_value_bool(env, argv[7], &out_all);
    if (argc > 8) napi_get_value_bool(env, argv[8], &fetch);
    std::vector<const void *> ptrs(ncalls);
    for (size_t k = 0; k < ncalls; k++) {
        const double o = static_cast<const double *>(p)[k];
        if (!(o >= 0) || (size_t)o + (size_t)n * stride > d->bytes) {
            napi_throw_range_error(env, nullptr, "camshiftTrackSequence: a call's frames lie outside the device buffer");
            return nullptr;
        }
        ptrs[k] = static_cast<const char *>(d->ptr) + (size_t)o;
    }
    std::vector<ht_cs_trackobj> out(fetch ? (size_t)n * (out_all ? ncalls : 1) : 0);
    ht_status st = ht_camshift_track_sequence(L.ctx, first, n, calc, ptrs.data(), (int32_t)ncalls, stride, fetch ? out.data() : nullptr, out_all ? 1 : 0);
    if (st != HT_OK) return throw_ht(env, L.ctx, st, "ht_camshift_track_sequence");
    if (!fetch) return nullptr;
    return trackobjs_result(env, out);
}

napi_value CamshiftSequenceCollect(napi_env env, napi_callback_info info) {
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    int32_t n = 0, ncalls = 0;
    bool out_all = false;
    if (too_few(env, argc, 3) || !lock_ctx(env, argv[0], &L)) return nullptr;
    if (!get_i32(env, argv[1], &n) || !get_i32(env, argv[2], &ncalls) || n <= 0 || ncalls <= 0) {
        napi_throw_type_error(env, nullptr, "camshiftSequenceCollect(ctx, n, ncalls, outAll)");
        return nullptr;
    }
    if (argc > 3) napi_get_value_bool(env, argv[3], &out_all);
    std::vector<ht_cs_trackobj> out((size_t)n * (out_all ? (size_t)ncalls : 1));
    ht_status st = ht_camshift_sequence_collect(L.ctx, n, ncalls, out_all ? 1 : 0, out.data());
    if (st != HT_OK) return throw_ht(env, L.ctx, st, "ht_camshift_sequence_collect");
    return trackobjs_result(env, out);
}

napi_value ctx_counter(napi_env env, napi_callback_info info, int which) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    Locked L;
    if (too_few(env, argc, 1) || !lock_ctx(env, argv[0], &L)) return nullptr;
    napi_value v;
    const double x = which == 0 ? (double)ht_frames_bound(L.ctx) : which == 1 ? (double)ht_frames_enqueued(L.ctx) : (double)ht_graph_launches(L.ctx);
    NAPI_OK(napi_create_double(env, x, &v));
    return v;
}
napi_value FramesBound(napi_env env, napi_callback_info info) { return ctx_counter(env, info, 0); }
napi_value FramesEnqueued(napi_env env, napi_callback_info info) { return ctx_counter(env, info, 1); }
napi_value GraphLaunches(napi_env env, napi_callback_info info) { return ctx_counter(env, info, 2); }

napi_value Init(napi_env env, napi_value exports) {
    napi_add_env_cleanup_hook(env, env_cleanup, env);
    struct {
        const char *name;
        napi_callback fn;
    } fns[] = {{"createContext", CreateContext}, {"destroy", Destroy},         {"setGeometry", SetGeometry},
               {"detect", Detect},               {"detectAsync", DetectAsync}, {"grayscale", Grayscale},
               {"whitebalance", Whitebalance},   {"camshiftReserve", CamshiftReserve},
               {"camshiftInit", CamshiftInit},   {"camshiftTrack", CamshiftTrack}, {"info", Info},
               {"deviceCount", DeviceCount},     {"allgatherBest", AllgatherBest},
               {"exitNow", ExitNow},
               {"hostAlloc", HostAlloc},         {"hostFree", HostFree},       {"deviceAlloc", DeviceAlloc}, {"deviceFree", DeviceFree}, {"deviceUpload", DeviceUpload},
               {"upload", Upload},               {"bindDevice", BindDevice},   {"uploadAsync", UploadAsync}, {"swapFrames", SwapFrames},
               {"detectEnqueue", DetectEnqueue}, {"detectCollect", DetectCollect}, {"collectBest", CollectBest},
               {"detectWhitebalance", DetectWhitebalance}, {"whitebalanceBound", WhitebalanceBound},
               {"camshiftInitBound", CamshiftInitBound}, {"camshiftTrackBound", CamshiftTrackBound}, {"camshiftTrackCollect", CamshiftTrackCollect},
               {"camshiftTrackSequence", CamshiftTrackSequence}, {"camshiftSequenceCollect", CamshiftSequenceCollect},
               {"framesBound", FramesBound},     {"framesEnqueued", FramesEnqueued}, {"graphLaunches", GraphLaunches}};
    for (auto &f : fns) {
        napi_value fn;
        if (napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &fn) != napi_ok) return nullptr;
        if (napi_set_named_property(env, exports, f.name, fn) != napi_ok) return nullptr;
    }
    napi_value v;
    napi_create_int32(env, ht_abi_version(), &v);
    napi_set_named_property(env, exports, "abiVersion", v);
    napi_create_int32(env, HT_INPUT_GRAY_IN_R, &v);
    napi_set_named_property(env, exports, "INPUT_GRAY_IN_R", v);
    napi_create_int32(env, HT_INPUT_RGBA, &v);
    napi_set_named_property(env, exports, "INPUT_RGBA", v);
    napi_create_int32(env, HT_DETECT_WHITEBALANCE, &v);
    napi_set_named_property(env, exports, "DETECT_WHITEBALANCE", v);
    napi_create_int32(env, HT_SCAN_STATS, &v);
    napi_set_named_property(env, exports, "SCAN_STATS", v);
    return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
