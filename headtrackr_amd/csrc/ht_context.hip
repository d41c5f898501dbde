// ht_context.hip — context lifetime, cascade tables, pyramid geometry, frame binding, result collection and the
// host-side post-processing (rect conversion + grouping) of libheadtrackr_hip.so.
//
// Reference behaviour restated here (paths under /root/reference/src/):
//   geometry            ccv.js:110-147      (scale, scale_upto, level sizes, variant planes)
//   hits -> seq rects   ccv.js:227-234,244-245
//   grouping            ccv.js:34-107 (array_group), 249-332 (averaging, nested-rect filter)
#include <sched.h>
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <new>

#include "ht_internal.h"
#include "ht_hostpost.h"

static thread_local std::string g_create_err;

// roctx (see ht_internal.h)
#include <dlfcn.h>
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        void *h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (!push || !pop) push = nullptr, pop = nullptr;
        }
    }
};
Roctx &roctx() {
    static Roctx r;
    return r;
}
}  // namespace
HtRange::HtRange(const char *name) : on(roctx().push != nullptr) {
    if (on) roctx().push(name);
}
HtRange::~HtRange() {
    if (on) roctx().pop();
}

// every live context of the process: ht_device_free looks for OTHER contexts that still have frames bound inside the buffer
#include <atomic>
#include <mutex>
static std::mutex g_live_mu;
static std::vector<ht_ctx *> g_live;
// ht_device_alloc buffers whose owning context was destroyed while ANOTHER live context still had frames bound inside them (a batch
// host shares one frame buffer between its pipelined contexts): kept alive here and freed by the ht_destroy after which no live
// context is bound inside them any more — never under a context that would read freed HBM on its next enqueue.
struct HtOrphan {
    void *p;
    size_t bytes;
    int device;
    bool orphan;  // false: released by its own context's ht_destroy
};
static std::vector<HtOrphan> g_orphans;
static std::atomic<int> g_orphan_count{0};  // g_orphans.size(), readable without the lock: the frame-binding calls look at it first

// g_live_mu held.  Reads the other contexts' d_frames without their lock: a context being re-bound concurrently with the destruction
// or ht_device_free of the buffer it is bound to is a caller error (include/headtrackr_hip.h, "Lifetime of shared frame buffers").
static bool bound_by_live_context(const ht_ctx *except, int device, const void *p, size_t bytes) {
    const uint8_t *pb = static_cast<const uint8_t *>(p);
    for (const ht_ctx *o : g_live)
        if (o != except && o->device == device && o->d_frames && o->d_frames >= pb && o->d_frames < pb + bytes) return true;
    return false;
}

// A context's frames moved (bind / upload / swap / free): orphans that no live context is bound inside any more are released here as
// well, not only by some later ht_destroy — a host that re-binds its last binder elsewhere and never destroys anything would keep
// the HBM for the life of the process.  One relaxed load when there are no orphans (the streaming hosts bind per step).
static void sweep_orphans(ht_ctx *c) {
    if (g_orphan_count.load(std::memory_order_relaxed) == 0) return;
    std::vector<HtOrphan> release;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        for (size_t i = 0; i < g_orphans.size();) {
            if (!bound_by_live_context(nullptr, g_orphans[i].device, g_orphans[i].p, g_orphans[i].bytes)) {
                release.push_back(g_orphans[i]);
                g_orphans.erase(g_orphans.begin() + (long)i);
            } else {
                i++;
            }
        }
        g_orphan_count.store((int)g_orphans.size(), std::memory_order_relaxed);
    }
    for (auto &a : release) {  // whatever the context that just moved away (or anybody else) still had enqueued against it has to be through
        (void)hipSetDevice(a.device);
        (void)hipDeviceSynchronize();
        (void)hipFree(a.p);
    }
    if (!release.empty() && c) (void)hipSetDevice(c->device);
}

static inline HtPostCfg post_cfg(const ht_ctx *c) { return HtPostCfg{c->interval, c->cw, c->ch}; }

// workers for a batch of `frames` frames holding `hits` raw hits: option host_threads, or (auto) up to 7 when the batch is worth it
static int ht_host_workers(const ht_ctx *c, int frames, uint32_t hits) {
    if (c->host_threads == 0) return 0;
    if (c->host_threads > 0) return c->host_threads;
    if (frames < 32 || hits < 256) return 0;  // a live feed's frame or two: the hand-off would cost more than the work
    // auto: half of this process's share of the host cores — the cores its affinity mask allows, divided by the GPUs of the node (a
    // one-process-per-GPU job runs one such pool per GPU: 8 ranks x 7 spinning workers must not pin 64 cores of a small host)
    static const int share = [] {
        cpu_set_t set;
        CPU_ZERO(&set);
        int cpus = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
        int ngpu = 1;
        if (hipGetDeviceCount(&ngpu) != hipSuccess || ngpu < 1) ngpu = 1;
        return std::max(0, std::min(7, cpus / (2 * ngpu) - 1));
    }();
    return share;
}


ht_status ht_fail(ht_ctx *ctx, ht_status st, const std::string &msg) {
    if (ctx)
        ctx->err = msg;
    else
        g_create_err = msg;
    return st;
}

HtProfScope::HtProfScope(ht_ctx *c, const char *name, hipStream_t on) : ctx(c), stream(on ? on : c->stream) {
    if (!ctx->profiling) return;
    for (size_t i = 0; i < ctx->timers.size(); i++)
        if (ctx->timers[i].name == name) idx = (int)i;
    if (idx < 0) {
        ctx->timers.emplace_back();
        ctx->timers.back().name = name;
        idx = (int)ctx->timers.size() - 1;
    }
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
        idx = -1;
        return;
    }
    (void)hipEventRecord(a, stream);
}
HtProfScope::~HtProfScope() {
    if (idx < 0) return;
    (void)hipEventRecord(b, stream);
    ctx->timers[idx].pending.emplace_back(a, b);
    ctx->timers[idx].launches++;
}

extern "C" int32_t ht_abi_version(void) { return HT_ABI_VERSION; }

extern "C" const char *ht_last_error(const ht_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

// ---------------------------------------------------------------------------------------------------------
// cascade

static bool parse_blob(const uint8_t *blob, size_t len, ht_ctx *c, std::string &why) {
    if (!blob || len < 32 || std::memcmp(blob, "HTCB", 4) != 0) {
        why = "cascade blob: bad magic";
        return false;
    }
    uint32_t h[8];
    std::memcpy(h, blob, 32);
    if (h[1] != 1 || h[6] != HT_MAXPTS) {
        why = "cascade blob: unsupported version";
        return false;
    }
    c->nstages = h[2];
    c->cw = h[3];
    c->ch = h[4];
    c->nfeat = h[5];
    if (c->nstages == 0 || c->nstages > 63 || c->cw < 4 || c->ch < 4 || c->cw > 64 || c->ch > 64) {
        why = "cascade blob: unsupported stage count or window size";
        return false;
    }
    if (len != 32 + (size_t)c->nstages * sizeof(HtBlobStage) + (size_t)c->nfeat * sizeof(HtBlobFeature)) {
        why = "cascade blob: truncated";
        return false;
    }
    c->h_stages.resize(c->nstages);
    c->h_feats.resize(c->nfeat);
    std::memcpy(c->h_stages.data(), blob + 32, c->nstages * sizeof(HtBlobStage));
    std::memcpy(c->h_feats.data(), blob + 32 + c->nstages * sizeof(HtBlobStage), c->nfeat * sizeof(HtBlobFeature));
    uint32_t first = 0;
    for (uint32_t j = 0; j < c->nstages; j++) {
        if (c->h_stages[j].first != first || first + c->h_stages[j].count > c->nfeat) {
            why = "cascade blob: inconsistent stage table";
            return false;
        }
        first += c->h_stages[j].count;
    }
    for (uint32_t k = 0; k < c->nfeat; k++) {
        const HtBlobFeature &f = c->h_feats[k];
        // the reference reads slot 0 of both polarities unconditionally (ccv.js:191-192)
        if (f.size == 0 || f.size > HT_MAXPTS || f.pz[0] < 0 || f.nz[0] < 0) {
            why = "cascade blob: feature without a valid first point";
            return false;
        }
        for (int q = 0; q < f.size; q++) {
            const int lim[3] = {(int)c->cw, (int)c->cw / 2, (int)c->cw / 4};
            const int limy[3] = {(int)c->ch, (int)c->ch / 2, (int)c->ch / 4};
            if (f.pz[q] > 2 || f.nz[q] > 2 ||
                (f.pz[q] >= 0 && (f.px[q] < 0 || f.py[q] < 0 || f.px[q] >= lim[f.pz[q]] || f.py[q] >= limy[f.pz[q]])) ||
                (f.nz[q] >= 0 && (f.nx[q] < 0 || f.ny[q] < 0 || f.nx[q] >= lim[f.nz[q]] || f.ny[q] >= limy[f.nz[q]]))) {
                why = "cascade blob: feature point outside the window";
                return false;
            }
        }
    }
    return true;
}

// alpha * 1e8 is an exact integer iff the decimal literal had <= 8 fractional digits: then (double)k / 1e8 == alpha
static bool as_decimal8(double v, int64_t *out) {
    double s = v * 1e8;
    if (!(std::fabs(s) < 9.0e15)) return false;
    int64_t k = (int64_t)std::llround(s);
    if ((double)k / 1e8 != v) return false;
    *out = k;
    return true;
}

static ht_status upload_cascade(ht_ctx *c) {
    std::vector<HtDeepFeature> deep(c->nfeat);
    std::vector<HtDevStage> st(c->nstages);
    c->decimal_alphas = true;
    for (uint32_t k = 0; k < c->nfeat; k++) {
        const HtBlobFeature &f = c->h_feats[k];
        HtDeepFeature &d = deep[k];
        int np = 0, nn = 0;
        for (int q = 0; q < f.size; q++) {
            if (f.pz[q] >= 0) {
                d.px[np] = f.px[q];
                d.py[np] = f.py[q];
                d.pz[np] = f.pz[q];
                np++;
            }
            if (f.nz[q] >= 0) {
                d.nx[nn] = f.nx[q];
                d.ny[nn] = f.ny[q];
                d.nz[nn] = f.nz[q];
                nn++;
            }
        }
        for (int q = np; q < HT_MAXPTS; q++) d.px[q] = d.px[0], d.py[q] = d.py[0], d.pz[q] = d.pz[0];
        for (int q = nn; q < HT_MAXPTS; q++) d.nx[q] = d.nx[0], d.ny[q] = d.ny[0], d.nz[q] = d.nz[0];
        d.a0 = f.alpha[0];
        d.a1 = f.alpha[1];
        d.a0i = d.a1i = 0;
        if (!as_decimal8(d.a0, &d.a0i) || !as_decimal8(d.a1, &d.a1i)) c->decimal_alphas = false;
    }
    for (uint32_t j = 0; j < c->nstages; j++) {
        st[j].first = c->h_stages[j].first;
        st[j].count = c->h_stages[j].count;
        st[j].threshold = c->h_stages[j].threshold;
        st[j].thri = 0;
        st[j].pad = 0;
        if (!as_decimal8(st[j].threshold, &st[j].thri)) c->decimal_alphas = false;
        uint32_t mp = 1;
        for (uint32_t k = 0; k < st[j].count; k++) {
            const HtBlobFeature &f = c->h_feats[st[j].first + k];
            uint32_t np = 0, nn = 0;
            for (int q = 0; q < f.size; q++) np += f.pz[q] >= 0, nn += f.nz[q] >= 0;
            mp = std::max(mp, std::max(np, nn));
        }
        st[j].maxpts = mp;
        // The integer decision "S < thri  <=>  the reference's binary64 sum < threshold" (off an exact tie) needs the
        // rounding error of the reference's SEQUENTIAL sum to stay below half the 1e-8 grid: |err| <= count * 2^-52 *
        // sum|alpha|.  True for the trained cascade (alphas O(1): bound ~1e-11); a custom cascade with huge alphas takes
        // the sequential binary64 path everywhere instead of silently diverging from ccv.js:186-222.
        double sabs = 0.0;
        for (uint32_t k = 0; k < st[j].count; k++) {
            const HtBlobFeature &f = c->h_feats[st[j].first + k];
            sabs += std::max(std::fabs(f.alpha[0]), std::fabs(f.alpha[1]));
        }
        if (!((double)st[j].count * 2.220446049250313e-16 * (sabs + std::fabs(st[j].threshold)) < 0.5e-8)) c->decimal_alphas = false;
    }
    HT_HIP(c, hipMalloc(&c->d_deep_feats, deep.size() * sizeof(HtDeepFeature)));
    HT_HIP(c, hipMalloc(&c->d_stages, st.size() * sizeof(HtDevStage)));
    HT_HIP(c, hipMemcpy(c->d_deep_feats, deep.data(), deep.size() * sizeof(HtDeepFeature), hipMemcpyHostToDevice));
    HT_HIP(c, hipMemcpy(c->d_stages, st.data(), st.size() * sizeof(HtDevStage), hipMemcpyHostToDevice));
    return ht_scan_tile_tables(c);
}

// ---------------------------------------------------------------------------------------------------------
// ht_config.options: "key=value,key=value".  Every key selects among schedules with identical results; the three that do not
// (results incomplete by design: timing experiments) exist only with -DHT_DEBUG_KNOBS.
static bool apply_options(ht_ctx *c, const std::string &opts, std::string &why) {
    size_t pos = 0;
    while (pos < opts.size()) {
        size_t end = opts.find(',', pos);
        if (end == std::string::npos) end = opts.size();
        std::string kv = opts.substr(pos, end - pos);
        pos = end + 1;
        while (!kv.empty() && (kv.front() == ' ')) kv.erase(kv.begin());
        while (!kv.empty() && (kv.back() == ' ')) kv.pop_back();
        if (kv.empty()) continue;
        const size_t eq = kv.find('=');
        const std::string key = kv.substr(0, eq), val = eq == std::string::npos ? "1" : kv.substr(eq + 1);
        char *endp = nullptr;
        const long long v = std::strtoll(val.c_str(), &endp, 10);
        if (val.empty() || (endp && *endp)) {
            why = "option '" + key + "': value is not an integer";
            return false;
        }
        const int iv = (int)std::max<long long>(std::min<long long>(v, 1ll << 30), -(1ll << 30));
        if (key == "rs_rpt") { if (iv >= 1 && iv <= HT_RS_MAX_PASSES) c->rs_rpt = iv; }
        else if (key == "rs_tailtable") c->tail_table = iv < 0 ? 0 : (iv > 2 ? 2 : (int)iv), c->tail_table_forced = true;
        else if (key == "rs_tailcap") c->rs_tailcap = (uint64_t)std::max<long long>(v, 0), c->rs_tailcap_forced = true;
        else if (key == "rs_notail") c->rs_notail = iv != 0;
        else if (key == "rs_nofast") c->rs_nofast = iv != 0;
        else if (key == "rs_nosort") c->rs_nosort = iv != 0;
        else if (key == "rs_bands") c->rs_bands = iv != 0;
        else if (key == "rs_gennames") c->rs_gennames = iv != 0;
        else if (key == "rs_minwg") c->rs_min_wgs = std::max(1, iv);
        else if (key == "rs_k") c->dbg_rs_k = iv;
        else if (key == "rs_group") { if (iv >= 1 && iv <= 64) c->rs_group = iv, c->rs_group_forced = true; }
        else if (key == "early_scan") c->early_scan = iv != 0;
        else if (key == "force_exact") c->dbg_force_exact = iv;
        else if (key == "deep_bias") c->deep_bias = (uint32_t)std::max(0, iv);
        else if (key == "deep_v") c->dbg_deep_v = iv;
        else if (key == "cs_flags") c->cs_flags = iv != 0;
        else if (key == "cs_sync_ring") c->cs_sync_ring = iv != 0;
        else if (key == "fp_sparse") c->fp_sparse = iv != 0;
        else if (key == "deep_grid") c->deep_grid = std::max(1, iv);
        else if (key == "split") c->opt_split = std::max(1, iv);
        else if (key == "cs_fused_min") c->cs_fused_min_streams = std::max(1, iv);
        else if (key == "cs_fused_nt") c->cs_fused_nt = iv;
        else if (key == "cs_keep_hist") c->cs_keep_hist = iv != 0;
        else if (key == "cs_seq_fused") c->cs_seq_fused = iv != 0;
        else if (key == "cs_cluster") c->cs_cluster = iv != 0;
        else if (key == "cs_cluster_min_px") c->cs_cluster_min_px = (uint32_t)std::max(0, iv);
        else if (key == "cs_region") c->cs_region_cap = std::min(40960, std::max(0, iv));
        else if (key == "cs_barrier_budget") c->cs_barrier_budget = std::max(1ll, v);
        else if (key == "graph_max_frames") c->graph_max_frames = std::max(0, iv);
        else if (key == "host_threads") c->host_threads = std::min(64, std::max(0, iv));
        else if (key == "force_rccl") c->force_rccl = iv != 0;
#ifdef HT_DEBUG_KNOBS
        else if (key == "stop_stage") c->dbg_stop_stage = iv;                        // the tile kernel stops before this stage
        else if (key == "cs_iters") c->dbg_cs_iters = std::min(10, std::max(0, iv));   // mean-shift iterations (camshift.js:284 has 10)
        else if (key == "rs_maxgen") c->rs_maxgen = iv;                                // pyramid generations built
#endif
        else {
            why = "unknown option '" + key + "'";
            return false;
        }
    }
    return true;
}

extern "C" ht_status ht_create(const ht_config *cfg, const void *cascade_blob, size_t cascade_len, ht_ctx **out) {
    // ABI 1 callers pass the struct without its last member (options)
    if (!cfg || !out || (cfg->struct_size != sizeof(ht_config) && cfg->struct_size != offsetof(ht_config, options)))
        return ht_fail(nullptr, HT_ERR_INVALID, "ht_create: bad config");
    *out = nullptr;
    if (cfg->interval < 1 || cfg->interval > 12) return ht_fail(nullptr, HT_ERR_INVALID, "ht_create: interval must be 1..12");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return ht_fail(nullptr, HT_ERR_NO_DEVICE, "ht_create: no HIP device visible (this library has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return ht_fail(nullptr, HT_ERR_INVALID, "ht_create: device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return ht_fail(nullptr, HT_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return ht_fail(nullptr, HT_ERR_NO_DEVICE, std::string("ht_create: device is ") + prop.gcnArchName + ", this build is gfx950-only");
    ht_ctx *c = new (std::nothrow) ht_ctx();
    if (!c) return ht_fail(nullptr, HT_ERR_NOMEM, "ht_create: out of host memory");
    c->device = cfg->device;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->interval = cfg->interval;
    c->next = cfg->interval + 1;
    if (cfg->hit_capacity) c->hit_capacity = cfg->hit_capacity;
    c->queue_capacity_cfg = cfg->queue_capacity;
    std::string why;
    if (!parse_blob((const uint8_t *)cascade_blob, cascade_len, c, why)) {
        delete c;
        return ht_fail(nullptr, HT_ERR_INVALID, "ht_create: " + why);
    }
    ht_status st = HT_OK;
    auto bail = [&](ht_status s) {
        g_create_err = c->err;
        ht_destroy(c);
        return s;
    };
    if (hipSetDevice(c->device) != hipSuccess) {
        c->err = "hipSetDevice failed";
        return bail(HT_ERR_HIP);
    }
    if (cfg->stream) {
        c->stream = (hipStream_t)cfg->stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            c->err = "hipStreamCreate failed";
            return bail(HT_ERR_HIP);
        }
        c->own_stream = true;
    }
    // per-context options (schedule selectors for tests / A-B runs): from the config string only — this library reads no environment
    {
        std::string opts = (cfg->struct_size >= offsetof(ht_config, options) + sizeof(const char *) && cfg->options) ? cfg->options : "";
#ifdef HT_DEBUG_KNOBS  // instrumented builds (tools/build_alt.py) may also be steered from the shell
        if (const char *e = std::getenv("HT_OPTIONS")) opts += std::string(opts.empty() ? "" : ",") + e;
#endif
        std::string why;
        if (!apply_options(c, opts, why)) {
            c->err = "ht_create: " + why;
            return bail(HT_ERR_INVALID);
        }
    }
    if (c->early_scan) {
        if (hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_early_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_early_done, hipEventDisableTiming) != hipSuccess) {
            c->err = "hipStreamCreate (aux) failed";
            return bail(HT_ERR_HIP);
        }
    }
    c->builtin_cascade = ht_scan_is_builtin_cascade((const uint8_t *)cascade_blob, cascade_len) && c->interval >= 1;
    // stages [0, split) always run in the tile kernel: the generated straight-line stages for the built-in cascade
    c->split_stage = std::min<uint32_t>(c->builtin_cascade ? 8u : 4u, c->nstages);
    if (c->opt_split > 0)  // option split: hand-off stage (<= 8 for the generated stage code)
        c->split_stage = std::min<uint32_t>((uint32_t)c->opt_split, std::min<uint32_t>(c->builtin_cascade ? 8u : c->nstages, c->nstages));
    if ((st = upload_cascade(c)) != HT_OK) return bail(st);
    if ((st = ht_scan_pack_deep(c)) != HT_OK) return bail(st);
    if (hipHostMalloc(reinterpret_cast<void **>(&c->h_pinned), sizeof(HtCounters) + (size_t)HT_PINNED_HITS * sizeof(ht_hit), hipHostMallocDefault) != hipSuccess ||
        hipMalloc(&c->d_stats, sizeof(unsigned long long) * 64 * HT_STAT_SHARDS) != hipSuccess ||
        // counters and hits in ONE allocation, counters first: ht_detect_collect fetches both with a single copy
        hipMalloc(reinterpret_cast<void **>(&c->d_counters), sizeof(HtCounters) + (size_t)c->hit_capacity * sizeof(ht_hit)) != hipSuccess) {
        c->err = "hipMalloc(hits) failed";
        return bail(HT_ERR_NOMEM);
    }
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live.push_back(c);
    }
    c->d_hits = reinterpret_cast<ht_hit *>(reinterpret_cast<uint8_t *>(c->d_counters) + sizeof(HtCounters));
    *out = c;
    return HT_OK;
}

static void destroy_graphs(ht_ctx *c);
static void free_geometry(ht_ctx *c) {
    destroy_graphs(c);  // captured launch sequences hold this geometry's pointers
    if (c->d_levels) (void)hipFree(c->d_levels), c->d_levels = nullptr;
    if (c->d_arena) (void)hipFree(c->d_arena), c->d_arena = nullptr;
    if (c->d_scales) (void)hipFree(c->d_scales), c->d_scales = nullptr;
    if (c->d_queue) (void)hipFree(c->d_queue), c->d_queue = nullptr;
    for (auto p : c->d_gen_blocks)
        if (p) (void)hipFree(p);
    c->d_gen_blocks.clear();
    if (c->d_tile_recs) (void)hipFree(c->d_tile_recs), c->d_tile_recs = nullptr;
    if (c->d_tail_jobs) (void)hipFree(c->d_tail_jobs), c->d_tail_jobs = nullptr;
    if (c->d_tail_prefix) (void)hipFree(c->d_tail_prefix), c->d_tail_prefix = nullptr;
    if (c->d_tail_taps) (void)hipFree(c->d_tail_taps), c->d_tail_taps = nullptr;
    if (c->d_tail_taps_fast) (void)hipFree(c->d_tail_taps_fast), c->d_tail_taps_fast = nullptr;
    if (c->d_tail_tapref) (void)hipFree(c->d_tail_tapref), c->d_tail_tapref = nullptr;
    c->tail_first_gen = 0;
    c->h_gens.clear();
    c->gen_blocks.clear();
    c->h_scales.clear();
}

extern "C" void ht_destroy(ht_ctx *c) {
    if (!c) return;
    std::vector<HtOrphan> release;  // buffers nobody is bound to any more: this context's own and orphans of earlier destroys
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live.erase(std::remove(g_live.begin(), g_live.end(), c), g_live.end());
        for (auto &a : c->user_allocs) {
            // same rule as ht_device_free: never free HBM under a live context that has frames bound inside it — the buffer outlives
            // its owner as an orphan instead
            if (bound_by_live_context(c, c->device, a.first, a.second)) g_orphans.push_back(HtOrphan{a.first, a.second, c->device, true});
            else release.push_back(HtOrphan{a.first, a.second, c->device, false});
        }
        c->user_allocs.clear();
        for (size_t i = 0; i < g_orphans.size();) {
            if (!bound_by_live_context(nullptr, g_orphans[i].device, g_orphans[i].p, g_orphans[i].bytes)) {
                release.push_back(g_orphans[i]);
                g_orphans.erase(g_orphans.begin() + (long)i);
            } else {
                i++;
            }
        }
        g_orphan_count.store((int)g_orphans.size(), std::memory_order_relaxed);
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream), (void)hipStreamDestroy(c->aux_stream);
    if (c->ev_early_ready) (void)hipEventDestroy(c->ev_early_ready);
    if (c->ev_early_done) (void)hipEventDestroy(c->ev_early_done);
    free_geometry(c);
    if (c->d_tile_feats) (void)hipFree(c->d_tile_feats);
    if (c->d_deep_feats) (void)hipFree(c->d_deep_feats);
    if (c->d_patch_feats) (void)hipFree(c->d_patch_feats);
    if (c->d_packed_feats) (void)hipFree(c->d_packed_feats);
    if (c->d_fp_feats) (void)hipFree(c->d_fp_feats);
    if (c->d_stages) (void)hipFree(c->d_stages);
    if (c->d_frames_own) (void)hipFree(c->d_frames_own);
    if (c->d_frames_back) (void)hipFree(c->d_frames_back);
    if (c->ev_copy_done) (void)hipEventDestroy(c->ev_copy_done);
    if (c->ev_front_free) (void)hipEventDestroy(c->ev_front_free);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->d_counters) (void)hipFree(c->d_counters);  // (d_hits lives in the same allocation)
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->h_wb_pinned) (void)hipHostFree(c->h_wb_pinned);
    if (c->d_cs_err) (void)hipFree(c->d_cs_err);
    if (c->h_cs_err) (void)hipHostFree(c->h_cs_err);
    if (c->h_cs_err_direct) (void)hipHostFree(c->h_cs_err_direct);
    for (auto &sl : c->cs_ring) {
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.h_flag) (void)hipHostFree(sl.h_flag);
        if (sl.ev) (void)hipEventDestroy(sl.ev);
    }
    ht_cluster_gate_forget(c);
    if (c->h_cs_rects) (void)hipHostFree(c->h_cs_rects);
    if (c->ev_cs_rects) (void)hipEventDestroy(c->ev_cs_rects);
    if (c->d_cs) (void)hipFree(c->d_cs);
    if (c->d_cs_hist) (void)hipFree(c->d_cs_hist);
    if (c->d_cs_out) (void)hipFree(c->d_cs_out);
    if (c->d_cs_seq_out) (void)hipFree(c->d_cs_seq_out);
    if (c->d_cs_lut) (void)hipFree(c->d_cs_lut);
    if (c->d_cs_parts) (void)hipFree(c->d_cs_parts);
    if (c->d_gather) (void)hipFree(c->d_gather);
    for (auto &a : release) {
        if (a.orphan) {  // whatever a context that has meanwhile re-bound elsewhere still had enqueued against it has to be through
            (void)hipSetDevice(a.device);
            (void)hipDeviceSynchronize();
        }
        (void)hipFree(a.p);
    }
    (void)hipSetDevice(c->device);
    for (auto &t : c->timers)
        for (auto &p : t.pending) (void)hipEventDestroy(p.first), (void)hipEventDestroy(p.second);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// ---------------------------------------------------------------------------------------------------------
// geometry, ccv.js:110-147


static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// rs_tap (ht_pyramid.hip) on the host: the same binary64 operations in the same order (this file is compiled with
// -ffp-contract=off like the kernels, so nothing is fused)
static HtTap ht_host_tap(int i, double r, int s, int origin) {
    double f = ((double)i + 0.5) * r;
    f = f + (-0.5);
    f = f < 0.0 ? 0.0 : f;
    const double fmax = (double)(s - 1);
    f = f > fmax ? fmax : f;
    const double af = std::floor(f);
    HtTap tp;
    tp.a = origin + (int)af;
    tp.b = origin + std::min((int)af + 1, s - 1);
    tp.t = f - af;
    tp.u = 1.0 - tp.t;
    return tp;
}

// Builds every table / allocation of one geometry.  On any failure the caller (ht_set_geometry) frees what was built and
// leaves the context without a geometry, so a retry (e.g. with a smaller max_batch after HT_ERR_NOMEM) starts clean.
static ht_status set_geometry_impl(ht_ctx *c, int32_t width, int32_t height, int32_t max_batch, const int32_t *level_dims, int n, int upto) {
    const int next = c->next;
    c->W = width;
    c->H = height;
    c->max_batch = max_batch;
    c->nlevels = n;
    c->upto = upto;

    uint64_t off = 0;
    c->pyr_bytes = 0;
    for (int i = 0; i < n; i++) {
        HtDevLevel &L = c->h_levels[i];
        if (level_dims) {  // validated by ht_set_geometry
            L.w = level_dims[2 * i];
            L.h = level_dims[2 * i + 1];
        } else if (i == 0) {
            L.w = width;
            L.h = height;
        } else if (i <= c->interval) {  // ccv.js:119-120
            L.w = (int)std::floor((double)width / ht_scale_pow(c->interval, i));
            L.h = (int)std::floor((double)height / ht_scale_pow(c->interval, i));
        } else {  // ccv.js:126-127
            L.w = c->h_levels[i - next].w / 2;
            L.h = c->h_levels[i - next].h / 2;
        }
        L.stride = (int)align_up((uint64_t)L.w, 4);
        for (int s = 0; s < 4; s++) {
            if (s == 0 || i >= 2 * next) {  // ccv.js:131
                if (off > 0xfffffff0ull) return ht_fail(c, HT_ERR_INVALID, "ht_set_geometry: frame too large");
                L.off[s] = (uint32_t)off;
                off = align_up(off + (uint64_t)L.stride * L.h, 256);
                c->pyr_bytes += (uint64_t)L.w * L.h;
            } else {
                L.off[s] = 0xffffffffu;
            }
        }
    }
    c->arena_stride = align_up(off + 256, 256);

    // resample jobs by dependency generation (generation 0 = the gray plane itself)
    std::vector<int> gen(n, 0);
    int ngen = 1;
    for (int i = 1; i < n; i++) {
        gen[i] = (i <= c->interval) ? 1 : gen[i - next] + 1;
        ngen = std::max(ngen, gen[i] + 1);
    }
    c->h_gens.assign(ngen, {});
    auto add_job = [&](int g, int src, int dst, int slot, int sx, int sy, int sw, int sh, int dw, int dh) {
        const HtDevLevel &S = c->h_levels[src], &D = c->h_levels[dst];
        if (D.w <= 0 || D.h <= 0) return;
        HtResampleJob j;
        std::memset(&j, 0, sizeof(j));
        j.src_off = S.off[0];
        j.dst_off = D.off[slot];
        j.src_stride = S.stride;
        j.dst_stride = D.stride;
        j.sx = sx, j.sy = sy, j.sw = sw, j.sh = sh;
        j.dw = dw, j.dh = dh;
        j.cw = D.w, j.ch = D.h;
        if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) {  // nothing is drawn: the canvas stays transparent black
            j.dw = j.dh = 0;
            j.sw = j.sh = 1;
            j.rx = j.ry = 1;
        } else {
            j.rx = (double)sw / (double)dw;
            j.ry = (double)sh / (double)dh;
        }
        c->h_gens[g].push_back(j);
    };
    for (int i = 1; i < n; i++) {
        const HtDevLevel &D = c->h_levels[i];
        if (i <= c->interval) {  // ccv.js:121
            add_job(gen[i], 0, i, 0, 0, 0, c->h_levels[0].w, c->h_levels[0].h, D.w, D.h);
        } else {  // ccv.js:128
            const HtDevLevel &S = c->h_levels[i - next];
            add_job(gen[i], i - next, i, 0, 0, 0, S.w, S.h, D.w, D.h);
            if (i >= 2 * next) {  // ccv.js:135,140,145
                add_job(gen[i], i - next, i, 1, 1, 0, S.w - 1, S.h, D.w - 2, D.h);
                add_job(gen[i], i - next, i, 2, 0, 1, S.w, S.h - 1, D.w, D.h - 2);
                add_job(gen[i], i - next, i, 3, 1, 1, S.w - 1, S.h - 1, D.w - 2, D.h - 2);
            }
        }
    }
    c->d_gen_blocks.assign(ngen, nullptr);
    c->gen_blocks.assign(ngen, 0);
    for (int g = 1; g < ngen; g++) {
        // k_resample tile records: 64 columns x np passes of 16 rows.  np is bounded by the LDS source window (the rows
        // a tile touches: ~16 np ry + 3 <= HT_RS_SRC_ROWS; the kernel falls back to HBM taps if a tile still does not fit)
        // and by rs_rpt; a canvas of P = ceil(ch / 16) passes is then cut into ceil(P / np) tiles of near-equal pass counts.
        std::vector<HtResampleJob> tiles;
        for (auto &j : c->h_gens[g]) {
            int npmax = 1;
            for (int t = 2; t <= std::min(c->rs_rpt, HT_RS_MAX_PASSES); t++)
                if ((int)std::ceil(16.0 * t * j.ry) + 3 <= HT_RS_SRC_ROWS) npmax = t;
            const int passes = (j.ch + 15) / 16, nby = (passes + npmax - 1) / npmax, nbx = (j.cw + 63) / 64;
            int pass0 = 0;
            for (int y = 0; y < nby; y++) {
                const int np = passes / nby + (y < passes % nby ? 1 : 0);
                for (int x = 0; x < nbx; x++) {
                    HtResampleJob t = j;
                    t.bx = (uint16_t)x, t.pass0 = (uint16_t)pass0, t.np = (uint16_t)np;
                    // bit 0: exact 2:1 in both directions (2x2 box mean, see the BOX rows of k_resample); option rs_nofast keeps
                    // every pixel on the declared binary64 sequence (A/B and cross-check)
                    t.pad = c->rs_nofast ? 2 : (uint16_t)((j.dw > 0 && j.sw == 2 * j.dw && j.sh == 2 * j.dh) ? 1 : 0);
                    const int X0 = 64 * x, Y0 = 16 * pass0, ncols = std::min(64, j.dw - X0), nrows = std::min(16 * np, j.dh - Y0);
                    if (ncols > 0 && nrows > 0) {  // the source extent k_resample stages into LDS (same expressions as in the kernel)
                        t.ex_xa = ht_host_tap(X0, j.rx, j.sw, j.sx).a & ~15;
                        t.ex_ya = ht_host_tap(Y0, j.ry, j.sh, j.sy).a;
                        t.ex_sw16 = (ht_host_tap(X0 + ncols - 1, j.rx, j.sw, j.sx).b - t.ex_xa) / 16 + 1;
                        t.ex_sh = ht_host_tap(Y0 + nrows - 1, j.ry, j.sh, j.sy).b - t.ex_ya + 1;
                        // k_resample_bands: the source rows of each wavefront's quarter of the tile (rows beyond the drawn ones read the
                        // last drawn row's taps, as in the kernel)
                        bool fit = t.ex_sw16 * 16 <= 160;
                        for (int w = 0; w < 4; w++) {
                            const int r0 = std::min(4 * np * w, nrows - 1), r1 = std::min(4 * np * (w + 1) - 1, nrows - 1);
                            const int bya = ht_host_tap(Y0 + r0, j.ry, j.sh, j.sy).a - t.ex_ya;
                            const int bsh = ht_host_tap(Y0 + r1, j.ry, j.sh, j.sy).b - (t.ex_ya + bya) + 1;
                            if (bya < 0 || bya > 255 || bsh < 1 || bsh > HT_RSB_ROWS) fit = false;
                            t.band_ya4 |= (uint32_t)(bya & 0xff) << (8 * w), t.band_sh4 |= (uint32_t)(bsh & 0xff) << (8 * w);
                        }
                        if (fit) t.pad |= 4;
                    }
                    tiles.push_back(t);
                }
                pass0 += np;
            }
        }
        // launch order = source order: tiles of different drawImage calls that read the same rows of the same source
        // plane (levels 1..6 all read level 0; the four variants of a level read the same parent) run back to back on
        // an XCD, so the source band is fetched from HBM once and then served by that XCD's L2
        if (!c->rs_nosort)
            std::stable_sort(tiles.begin(), tiles.end(), [](const HtResampleJob &a, const HtResampleJob &b) {
                if (a.src_off != b.src_off) return a.src_off < b.src_off;
                const int ya = (int)(16.0 * a.pass0 * a.ry), yb = (int)(16.0 * b.pass0 * b.ry);
                if (ya / 32 != yb / 32) return ya < yb;
                return a.bx < b.bx;
            });
        c->gen_blocks[g] = (uint32_t)tiles.size();
        if (tiles.empty()) continue;
        HT_HIP(c, hipMalloc(&c->d_gen_blocks[g], tiles.size() * sizeof(HtResampleJob)));
        HT_HIP(c, hipMemcpy(c->d_gen_blocks[g], tiles.data(), tiles.size() * sizeof(HtResampleJob), hipMemcpyHostToDevice));
    }

    HT_HIP(c, hipMalloc(&c->d_levels, sizeof(HtDevLevel) * HT_MAX_LEVELS));
    HT_HIP(c, hipMemcpy(c->d_levels, c->h_levels, sizeof(HtDevLevel) * n, hipMemcpyHostToDevice));
    if (hipMalloc(&c->d_arena, c->arena_stride * (uint64_t)max_batch) != hipSuccess)
        return ht_fail(c, HT_ERR_NOMEM, "ht_set_geometry: hipMalloc(pyramid arena) failed");
    HT_HIP(c, hipMemset(c->d_arena, 0, c->arena_stride * (uint64_t)max_batch));

    // tail plan: from the first generation g0 on which every generation has <= HT_TAIL_MAX_JOBS jobs and all of them
    // together <= tail_cap destination pixels per frame, one workgroup per frame does the rest of the pyramid in one launch
    // (k_resample_tail) instead of one nearly empty launch per generation.
    constexpr int HT_SMALL_BATCH = 48;
    // Small batches (a live feed's frame, the 8 feeds of a streaming step) are latency chains, not throughput: the tail kernel is ONE
    // workgroup per frame walking its generations behind barriers — 21 us for 17 k pixels of a single 320x240 frame, the longest kernel
    // of the call —, while a k_resample_bands launch of the same generation is 4.5 us on the otherwise idle chip.  rocprofv3 kernel
    // trace of single-frame calls (tools/gpu_one_frame_trace.sh, round 6): cap 32 768 -> 4 000 pixels takes 74.1 -> 63.5 us off the
    // device span at 320x240 (generation 4 as a launch, generations 5 - 7 in the tail) and 92.9 -> 79.3 us at 1920x1080 (no tail at all);
    // batches that fill the chip keep the large cap (C2: cap 4 000 costs +3 % on the pyramid, no tail at all +19 %).  Where it ends, pipelined
    // (three batches of 320x240 in flight / two of 1280x720, small plan against large): 24 frames +11 %, 32 +10.6 %, 48 +9.5 %, 64 +-0 %, 128 +-0 %;
    // 720p: 16 frames +2 %, 32 +3 %.
    const uint64_t tail_cap = c->rs_tailcap_forced ? c->rs_tailcap : (max_batch <= HT_SMALL_BATCH ? 4000u : c->rs_tailcap);
    // which tail kernel: measured (3 batches in flight), the table-driven binary32 tail (68 VGPRs, 35 KB LDS) is worth +4-5 % at
    // 128 x 720p but costs 3 % at 256 x 320x240, where its grid puts a 1024-thread workgroup on EVERY CU and its footprint keeps
    // the other batches' kernels from sharing them; the round-1 binary64 tail (41 VGPRs) is kept for batches that cover the chip.
    // Larger caps (generation 3 of C2 = 54 k pixels in the tail) lose with either kernel.
    // ... and for a handful of frames: 7.6 us against the table form's 10.3 for generations 5 - 7 of a single 320x240 frame (the same trace)
    if (!c->tail_table_forced) c->tail_table = (max_batch <= 128 && max_batch > HT_SMALL_BATCH) ? 1 : 0;
    c->tail_first_gen = 0;
    if (c->d_tail_jobs) (void)hipFree(c->d_tail_jobs), c->d_tail_jobs = nullptr;
    if (c->d_tail_prefix) (void)hipFree(c->d_tail_prefix), c->d_tail_prefix = nullptr;
    if (!c->rs_notail) {
        int g0 = ngen;
        uint64_t px = 0;
        for (int g = ngen - 1; g >= 1; g--) {
            uint64_t gp = 0;
            for (auto &j : c->h_gens[g]) gp += (uint64_t)j.cw * j.ch;
            if (c->h_gens[g].size() > (size_t)HT_TAIL_MAX_JOBS || px + gp > tail_cap) break;
            px += gp;
            g0 = g;
        }
        if (ngen - g0 >= 2 && ngen - g0 <= HT_TAIL_MAX_GENS) {
            std::vector<HtResampleJob> tj;
            std::vector<uint32_t> pref;
            HtTailGens &T = c->h_tail;
            std::memset(&T, 0, sizeof(T));
            T.ngen = ngen - g0;
            for (int g = g0; g < ngen; g++) {
                T.job_begin[g - g0] = (int32_t)tj.size();
                uint32_t groups = 0;
                for (auto &j : c->h_gens[g]) {
                    tj.push_back(j);
                    pref.push_back(groups);
                    groups += (uint32_t)((j.cw + 3) / 4) * (uint32_t)j.ch;
                }
                T.groups[g - g0] = groups;
            }
            T.job_begin[T.ngen] = (int32_t)tj.size();
            if (!tj.empty()) {
                HT_HIP(c, hipMalloc(&c->d_tail_jobs, tj.size() * sizeof(HtResampleJob)));
                HT_HIP(c, hipMemcpy(c->d_tail_jobs, tj.data(), tj.size() * sizeof(HtResampleJob), hipMemcpyHostToDevice));
                HT_HIP(c, hipMalloc(&c->d_tail_prefix, pref.size() * sizeof(uint32_t)));
                HT_HIP(c, hipMemcpy(c->d_tail_prefix, pref.data(), pref.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                // tap tables: the geometry is the same for every frame, so the taps are computed once here
                std::vector<HtTap> taps;
                std::vector<HtTapFast> fast;
                std::vector<HtTailTapRef> refs;
                const bool nofast = c->rs_nofast;
                size_t jidx = 0;
                for (auto &j : tj) {
                    for (int g = 0; g <= T.ngen; g++)
                        if ((size_t)T.job_begin[g] == jidx) T.tap_begin[g] = (uint32_t)taps.size();
                    jidx++;
                    HtTailTapRef r;
                    r.col = (uint32_t)taps.size();
                    const int ncol = std::max(j.dw, 1), nrow = std::max(j.dh, 1);
                    for (int i = 0; i < ncol + 3; i++) taps.push_back(ht_host_tap(std::min(i, ncol - 1), j.rx, j.sw, j.sx));
                    r.row = (uint32_t)taps.size();
                    for (int i = 0; i < nrow; i++) taps.push_back(ht_host_tap(i, j.ry, j.sh, j.sy));
                    r.mode = nofast ? 2u : ((j.dw > 0 && j.sw == 2 * j.dw && j.sh == 2 * j.dh) ? 1u : 0u);
                    r.pad = 0;
                    refs.push_back(r);
                }
                T.tap_begin[T.ngen] = (uint32_t)taps.size();
                fast.resize(taps.size());
                for (size_t i = 0; i < taps.size(); i++) fast[i].a = taps[i].a, fast[i].tf = (float)taps[i].t;
                HT_HIP(c, hipMalloc(&c->d_tail_taps, taps.size() * sizeof(HtTap)));
                HT_HIP(c, hipMemcpy(c->d_tail_taps, taps.data(), taps.size() * sizeof(HtTap), hipMemcpyHostToDevice));
                HT_HIP(c, hipMalloc(&c->d_tail_taps_fast, fast.size() * sizeof(HtTapFast)));
                HT_HIP(c, hipMemcpy(c->d_tail_taps_fast, fast.data(), fast.size() * sizeof(HtTapFast), hipMemcpyHostToDevice));
                HT_HIP(c, hipMalloc(&c->d_tail_tapref, refs.size() * sizeof(HtTailTapRef)));
                HT_HIP(c, hipMemcpy(c->d_tail_tapref, refs.data(), refs.size() * sizeof(HtTailTapRef), hipMemcpyHostToDevice));
                c->tail_first_gen = g0;
            }
        }
    }

    ht_status st = ht_scan_plan_tiles(c);
    if (st != HT_OK) return st;
    // early scan plan: scale i needs levels i, i + next, i + 2 next; the leading scales whose last plane is finished after
    // generation 2 (interval 5: scale 0 = ~30 % of the windows) can start while generations 3.. are still being built
    c->early_gen = 0;
    c->early_tiles = 0;
    if (c->early_scan && c->aux_stream && ngen > 3 && (c->tail_first_gen == 0 || c->tail_first_gen > 2)) {
        uint32_t tiles = 0;
        for (auto &S : c->h_scales) {
            if (gen[S.l2] > 2) break;
            tiles += (uint32_t)(S.ntx * S.nty);
        }
        if (tiles > 0 && tiles < c->tiles_per_frame) c->early_gen = 2, c->early_tiles = tiles;
    }

    // survivor queue between the tile kernel and the deep kernel: 1/8 of all windows unless configured
    uint64_t qc = c->queue_capacity_cfg ? c->queue_capacity_cfg : std::max<uint64_t>(1u << 16, c->windows_per_frame * (uint64_t)max_batch / 8);
    qc = std::min<uint64_t>(qc, 1ull << 28);
    c->queue_capacity = (uint32_t)qc;
    if (hipMalloc(&c->d_queue, (size_t)qc * sizeof(HtQueueEntry) + HT_DEEP_CTR_BYTES) != hipSuccess)  // + the deep kernel's work counters
        return ht_fail(c, HT_ERR_NOMEM, "ht_set_geometry: hipMalloc(survivor queue) failed");
    return HT_OK;
}

extern "C" ht_status ht_set_geometry(ht_ctx *c, int32_t width, int32_t height, int32_t max_batch, const int32_t *level_dims,
                                     int32_t nlevels_in) {
    if (!c) return HT_ERR_INVALID;
    if (width <= 0 || height <= 0 || width > 16384 || height > 16384 || max_batch <= 0)
        return ht_fail(c, HT_ERR_INVALID, "ht_set_geometry: width/height must be 1..16384 and max_batch > 0");
    HT_HIP(c, hipSetDevice(c->device));
    const int upto = (int)std::floor(std::log((double)std::min(c->cw, c->ch)) / std::log(ht_scale_of(c->interval)));  // ccv.js:112
    const int n = upto + c->next * 2;  // ccv.js:113
    if (n > HT_MAX_LEVELS) return ht_fail(c, HT_ERR_INVALID, "ht_set_geometry: too many pyramid levels");
    if (level_dims && nlevels_in != n) return ht_fail(c, HT_ERR_INVALID, "ht_set_geometry: level_dims has the wrong number of levels");
    if (level_dims)  // everything that can be rejected is rejected before the current geometry is torn down
        for (int i = 0; i < n; i++) {
            const int lw = level_dims[2 * i], lh = level_dims[2 * i + 1];
            if (lw < 0 || lh < 0 || lw > width || lh > height || (i == 0 && (lw != width || lh != height)))
                return ht_fail(c, HT_ERR_INVALID, "ht_set_geometry: bad level_dims");
        }
    if (c->W == width && c->H == height && c->max_batch >= max_batch && c->nlevels == n && !level_dims) return HT_OK;
    HT_HIP(c, hipStreamSynchronize(c->stream));
    if (c->copy_stream) HT_HIP(c, hipStreamSynchronize(c->copy_stream));
    free_geometry(c);
    // frames bound / uploaded for the old geometry (incl. a pending ht_upload_frames_async) do not survive a size change
    c->nframes = c->enq_nframes = 0;
    c->wb_collected_n = -1;
    c->back_n = 0;
    c->d_frames = nullptr;
    c->enqueued = false;
    const ht_status st = set_geometry_impl(c, width, height, max_batch, level_dims, n, upto);
    if (st != HT_OK) {  // no half-built geometry: the early-out above must not fire on a retry
        (void)hipGetLastError();  // a failed hipMalloc leaves a sticky error that the next launch check would report
        free_geometry(c);
        c->W = c->H = c->max_batch = c->nlevels = c->upto = 0;
    }
    return st;
}

extern "C" int32_t ht_num_levels(const ht_ctx *c) { return c ? c->nlevels : 0; }
extern "C" uint64_t ht_windows_per_frame(const ht_ctx *c) { return c ? c->windows_per_frame : 0; }
extern "C" uint64_t ht_pyramid_bytes_per_frame(const ht_ctx *c) { return c ? c->pyr_bytes : 0; }

extern "C" ht_status ht_plane(const ht_ctx *c, int32_t level, int32_t slot, ht_plane_info *out) {
    if (!c || !out || level < 0 || level >= c->nlevels || slot < 0 || slot > 3) return HT_ERR_INVALID;
    const HtDevLevel &L = c->h_levels[level];
    out->width = L.w;
    out->height = L.h;
    out->stride = L.stride;
    out->present = L.off[slot] != 0xffffffffu;
    out->offset = out->present ? L.off[slot] : 0;
    return HT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// frames

extern "C" ht_status ht_upload_frames(ht_ctx *c, const uint8_t *host_rgba, int32_t n, size_t frame_stride) {
    if (!c) return HT_ERR_INVALID;
    HtRange range("ht_upload_frames");
    if (c->W == 0) return ht_fail(c, HT_ERR_STATE, "ht_upload_frames: call ht_set_geometry first");
    const size_t fbytes = (size_t)c->W * c->H * 4;
    if (!host_rgba || n <= 0 || n > c->max_batch || frame_stride < fbytes)
        return ht_fail(c, HT_ERR_INVALID, "ht_upload_frames: bad frame count or stride");
    HT_HIP(c, hipSetDevice(c->device));
    const size_t need = fbytes * (size_t)n;
    if (c->d_frames_own_bytes < need) {
        HT_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_frames_own) (void)hipFree(c->d_frames_own);
        c->d_frames_own = nullptr;
        c->d_frames_own_bytes = 0;
        if (hipMalloc(&c->d_frames_own, need) != hipSuccess) return ht_fail(c, HT_ERR_NOMEM, "ht_upload_frames: hipMalloc failed");
        c->d_frames_own_bytes = need;
    }
    if (frame_stride == fbytes) {
        HT_HIP(c, hipMemcpyAsync(c->d_frames_own, host_rgba, need, hipMemcpyHostToDevice, c->stream));
    } else {
        HT_HIP(c, hipMemcpy2DAsync(c->d_frames_own, fbytes, host_rgba, frame_stride, fbytes, (size_t)n, hipMemcpyHostToDevice, c->stream));
    }
    c->d_frames = c->d_frames_own;
    c->frame_stride = fbytes;
    c->nframes = n;
    sweep_orphans(c);
    return HT_OK;
}

extern "C" ht_status ht_upload_frames_async(ht_ctx *c, const uint8_t *host_rgba, int32_t n, size_t frame_stride) {
    if (!c) return HT_ERR_INVALID;
    if (c->W == 0) return ht_fail(c, HT_ERR_STATE, "ht_upload_frames_async: call ht_set_geometry first");
    const size_t fbytes = (size_t)c->W * c->H * 4;
    if (!host_rgba || n <= 0 || n > c->max_batch || frame_stride < fbytes)
        return ht_fail(c, HT_ERR_INVALID, "ht_upload_frames_async: bad frame count or stride");
    HT_HIP(c, hipSetDevice(c->device));
    if (!c->copy_stream) {
        HT_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        HT_HIP(c, hipEventCreateWithFlags(&c->ev_copy_done, hipEventDisableTiming));
        HT_HIP(c, hipEventCreateWithFlags(&c->ev_front_free, hipEventDisableTiming));
    }
    const size_t need = fbytes * (size_t)n;
    if (c->d_frames_back_bytes < need) {
        HT_HIP(c, hipStreamSynchronize(c->copy_stream));
        if (c->d_frames_back) (void)hipFree(c->d_frames_back);
        c->d_frames_back = nullptr;
        c->d_frames_back_bytes = 0;
        if (hipMalloc(&c->d_frames_back, need) != hipSuccess) return ht_fail(c, HT_ERR_NOMEM, "ht_upload_frames_async: hipMalloc failed");
        c->d_frames_back_bytes = need;
    }
    // the back buffer was the front buffer until the last swap: kernels enqueued before that swap may still read it
    HT_HIP(c, hipStreamWaitEvent(c->copy_stream, c->ev_front_free, 0));
    if (frame_stride == fbytes) {
        HT_HIP(c, hipMemcpyAsync(c->d_frames_back, host_rgba, need, hipMemcpyHostToDevice, c->copy_stream));
    } else {
        HT_HIP(c, hipMemcpy2DAsync(c->d_frames_back, fbytes, host_rgba, frame_stride, fbytes, (size_t)n, hipMemcpyHostToDevice, c->copy_stream));
    }
    HT_HIP(c, hipEventRecord(c->ev_copy_done, c->copy_stream));
    c->back_n = n;
    return HT_OK;
}

extern "C" ht_status ht_swap_frames(ht_ctx *c) {
    if (!c) return HT_ERR_INVALID;
    if (c->back_n <= 0) return ht_fail(c, HT_ERR_STATE, "ht_swap_frames: no ht_upload_frames_async pending");
    HT_HIP(c, hipSetDevice(c->device));
    HT_HIP(c, hipEventRecord(c->ev_front_free, c->stream));            // everything enqueued so far used the old front buffer
    HT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_copy_done, 0));      // later kernels wait for the copy, the host does not
    std::swap(c->d_frames_own, c->d_frames_back);
    std::swap(c->d_frames_own_bytes, c->d_frames_back_bytes);
    c->d_frames = c->d_frames_own;
    c->frame_stride = (size_t)c->W * c->H * 4;
    c->nframes = c->back_n;
    c->back_n = 0;
    sweep_orphans(c);
    return HT_OK;
}

extern "C" ht_status ht_bind_frames_device(ht_ctx *c, const void *dev_rgba, int32_t n, size_t frame_stride) {
    if (!c) return HT_ERR_INVALID;
    if (c->W == 0) return ht_fail(c, HT_ERR_STATE, "ht_bind_frames_device: call ht_set_geometry first");
    if (!dev_rgba || n <= 0 || n > c->max_batch || frame_stride < (size_t)c->W * c->H * 4 || (frame_stride & 3) ||
        ((uintptr_t)dev_rgba & 3))
        return ht_fail(c, HT_ERR_INVALID, "ht_bind_frames_device: bad pointer, frame count or stride (4-byte alignment required)");
    c->d_frames = (const uint8_t *)dev_rgba;
    c->frame_stride = frame_stride;
    c->nframes = n;
    sweep_orphans(c);
    return HT_OK;
}

extern "C" int32_t ht_frames_bound(const ht_ctx *c) { return c ? c->nframes : 0; }
extern "C" int32_t ht_frames_enqueued(const ht_ctx *c) { return (c && c->enqueued) ? c->enq_nframes : 0; }

extern "C" ht_status ht_host_alloc(size_t bytes, void **out) {
    if (!out || bytes == 0) return HT_ERR_INVALID;
    *out = nullptr;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return ht_fail(nullptr, HT_ERR_NOMEM, "ht_host_alloc: hipHostMalloc failed");
    }
    return HT_OK;
}
extern "C" void ht_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}
extern "C" ht_status ht_device_alloc(ht_ctx *c, size_t bytes, void **out) {
    if (!c || !out || bytes == 0) return HT_ERR_INVALID;
    *out = nullptr;
    HT_HIP(c, hipSetDevice(c->device));
    if (hipMalloc(out, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return ht_fail(c, HT_ERR_NOMEM, "ht_device_alloc: hipMalloc failed");
    }
    c->user_allocs.emplace_back(*out, bytes);
    return HT_OK;
}
extern "C" ht_status ht_device_free(ht_ctx *c, void *p) {
    if (!c) return HT_ERR_INVALID;
    if (!p) return HT_OK;
    HT_HIP(c, hipSetDevice(c->device));
    HT_HIP(c, hipStreamSynchronize(c->stream));  // nothing enqueued on this context may still read it
    auto it = std::find_if(c->user_allocs.begin(), c->user_allocs.end(), [p](const std::pair<void *, size_t> &a) { return a.first == p; });
    if (it == c->user_allocs.end()) return ht_fail(c, HT_ERR_INVALID, "ht_device_free: not a live ht_device_alloc buffer of this context");
    const uint8_t *pb = static_cast<const uint8_t *>(p);
    {   // a buffer that other contexts of this device have bound (ht_bind_frames_device: a batch host shares one frame buffer between
        // its pipelined contexts) is NOT freed under them: their next enqueue would read freed HBM.  Rebind or destroy them first.
        std::lock_guard<std::mutex> lk(g_live_mu);
        if (bound_by_live_context(c, c->device, pb, it->second))
            return ht_fail(c, HT_ERR_STATE, "ht_device_free: another live context still has frames bound inside this buffer (rebind or destroy it first)");
    }
    if (c->d_frames && c->d_frames >= pb && c->d_frames < pb + it->second) {  // frames bound inside it
        c->d_frames = nullptr, c->nframes = 0;
        destroy_graphs(c);  // their keys hold the freed pointer: an allocation that reuses the address must not replay them
    }
    c->user_allocs.erase(it);
    HT_HIP(c, hipFree(p));
    sweep_orphans(c);
    return HT_OK;
}
extern "C" ht_status ht_device_upload(ht_ctx *c, void *dst_dev, const void *src_host, size_t bytes) {
    if (!c || !dst_dev || !src_host) return HT_ERR_INVALID;
    if (bytes == 0) return HT_OK;
    HT_HIP(c, hipSetDevice(c->device));
    HT_HIP(c, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
    HT_HIP(c, hipStreamSynchronize(c->stream));
    return HT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// detect

// 4 u64 per frame (R, G, B channel sums + pad), two regions of max_batch frames: the batch in flight (fused into its gray pass) and the
// stand-alone ht_whitebalance_batch; plus the pinned staging the in-flight batch's sums are copied to by ht_detect_collect
static ht_status wb_scratch(ht_ctx *c) {
    const size_t region = sizeof(unsigned long long) * 4 * (size_t)std::max(c->max_batch, 1), need = 2 * region;
    if (c->d_scratch_bytes < need) {
        HT_HIP(c, hipStreamSynchronize(c->stream));
        destroy_graphs(c);  // captured detect sequences bake the d_scratch pointer into their gray / whitebalance nodes
        if (c->d_scratch) (void)hipFree(c->d_scratch);
        c->d_scratch = nullptr;
        c->d_scratch_bytes = 0;
        if (hipMalloc(&c->d_scratch, need) != hipSuccess) return ht_fail(c, HT_ERR_NOMEM, "whitebalance scratch: hipMalloc failed");
        c->d_scratch_bytes = need;
    }
    if (c->h_wb_pinned_bytes < region) {
        HT_HIP(c, hipStreamSynchronize(c->stream));
        if (c->h_wb_pinned) (void)hipHostFree(c->h_wb_pinned);
        c->h_wb_pinned = nullptr;
        c->h_wb_pinned_bytes = 0;
        if (hipHostMalloc(reinterpret_cast<void **>(&c->h_wb_pinned), region, hipHostMallocDefault) != hipSuccess)
            return ht_fail(c, HT_ERR_NOMEM, "whitebalance staging: hipHostMalloc failed");
        c->h_wb_pinned_bytes = region;
    }
    return HT_OK;
}
static unsigned long long *wb_standalone_region(ht_ctx *c) {
    return reinterpret_cast<unsigned long long *>(c->d_scratch) + 4 * (size_t)std::max(c->max_batch, 1);
}

// the stream work of one detect batch: counters, (whitebalance sums,) gray, pyramid generations, cascade scan
static ht_status detect_enqueue_body(ht_ctx *c, uint32_t flags) {
    HT_HIP(c, hipMemsetAsync(c->d_counters, 0, sizeof(HtCounters), c->stream));
    c->early_launched = false;
    if (flags & HT_SCAN_STATS) HT_HIP(c, hipMemsetAsync(c->d_stats, 0, sizeof(unsigned long long) * 64 * HT_STAT_SHARDS, c->stream));
    ht_status st;
    c->wb_fused = false;
    const bool wb = (flags & HT_DETECT_WHITEBALANCE) != 0;
    if (wb) {  // channel sums ride along with the gray pass (no second read of the frames)
        HT_HIP(c, hipMemsetAsync(c->d_scratch, 0, sizeof(unsigned long long) * 4 * (size_t)c->nframes, c->stream));
        c->wb_fused = (c->W & 3) == 0;  // the linear gray kernel carries the sums; odd widths take the separate pass below
    }
    st = ht_launch_pyramid(c, flags);
    c->wb_fused = false;
    if (st != HT_OK) return st;
    if (wb && (c->W & 3) != 0 && (st = ht_launch_whitebalance(c, c->d_scratch, false)) != HT_OK) return st;
    return ht_launch_scan(c, flags);
}

static void destroy_graphs(ht_ctx *c) {
    for (auto &g : c->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    c->graphs.clear();
}

extern "C" uint64_t ht_graph_launches(const ht_ctx *c) { return c ? c->graph_launches : 0; }

extern "C" ht_status ht_detect_enqueue(ht_ctx *c, uint32_t flags) {
    if (!c) return HT_ERR_INVALID;
    HtRange range("ht_detect_enqueue");
    if (!c->d_frames || c->nframes <= 0) return ht_fail(c, HT_ERR_STATE, "ht_detect_enqueue: no frames bound");
    HT_HIP(c, hipSetDevice(c->device));
    ht_status st;
    if ((flags & HT_DETECT_WHITEBALANCE) && (st = wb_scratch(c)) != HT_OK) return st;  // may allocate and synchronise: never inside a capture
    // Small batches are launch-bound (one 1080p frame: ~0.13 ms of kernels inside a ~0.38 ms sequence of ~10 dependent launches):
    // the second enqueue of the same (frames, count, flags) captures the sequence into a hipGraph, later ones replay it.  The first
    // one always runs plainly (it also sets the function attributes a capture must not).
    const bool graph_ok = c->graph_max_frames > 0 && c->nframes <= c->graph_max_frames && c->own_stream && !c->profiling && !c->early_scan &&
                          !(flags & HT_SCAN_STATS);
    bool done = false;
    if (graph_ok) {
        HtDetectGraph *g = nullptr;
        for (auto &e : c->graphs)
            if (e.frames == c->d_frames && e.frame_stride == c->frame_stride && e.nframes == c->nframes && e.flags == flags) g = &e;
        if (!g) {
            if (c->graphs.size() >= 8) {  // a streaming host alternates between two frame buffers; 8 keys are plenty
                HtDetectGraph &old = c->graphs.front();
                if (old.exec) (void)hipGraphExecDestroy(old.exec);
                if (old.graph) (void)hipGraphDestroy(old.graph);
                c->graphs.erase(c->graphs.begin());
            }
            c->graphs.emplace_back();
            g = &c->graphs.back();
            g->frames = c->d_frames, g->frame_stride = c->frame_stride, g->nframes = c->nframes, g->flags = flags;
        }
        if (!g->exec && g->seen == 1) {
            ht_capture_mark(c, true);
            if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                st = detect_enqueue_body(c, flags);
                hipGraph_t graph = nullptr;
                const hipError_t e = hipStreamEndCapture(c->stream, &graph);
                ht_capture_mark(c, false);
                if (st == HT_OK && e == hipSuccess && graph && hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    g->graph = graph;
                } else {  // not capturable here: keep launching plainly
                    if (graph) (void)hipGraphDestroy(graph);
                    g->exec = nullptr;
                    g->seen = 1 << 30;
                    (void)hipGetLastError();
                }
            } else {
                ht_capture_mark(c, false);
                (void)hipGetLastError();
                g->seen = 1 << 30;
            }
        }
        if (g->exec) {
            HT_HIP(c, hipGraphLaunch(g->exec, c->stream));
            c->graph_launches++;
            c->early_launched = false;
            done = true;
        } else if (g->seen < (1 << 30)) {
            g->seen++;
        }
    }
    if (!done && (st = detect_enqueue_body(c, flags)) != HT_OK) return st;
    c->stats_enqueued = (flags & HT_SCAN_STATS) != 0;
    c->wb_enqueued = (flags & HT_DETECT_WHITEBALANCE) != 0;
    c->enqueued = true;
    c->enq_nframes = c->nframes;  // ht_detect_collect reports THIS batch even if other frames were bound / swapped in meanwhile
    return HT_OK;
}

static inline bool hit_less(const ht_hit &a, const ht_hit &b) {  // emission order, ccv.js:154,178,181-182
    if (a.frame != b.frame) return a.frame < b.frame;
    if (a.scale != b.scale) return a.scale < b.scale;
    if (a.q != b.q) return a.q < b.q;
    if (a.y != b.y) return a.y < b.y;
    return a.x < b.x;
}

extern "C" ht_status ht_detect_collect(ht_ctx *c, ht_hit *hits, uint32_t cap, uint32_t *counts, uint32_t *total) {
    if (!c) return HT_ERR_INVALID;
    HtRange range("ht_detect_collect");
    if (!c->enqueued) return ht_fail(c, HT_ERR_STATE, "ht_detect_collect: nothing enqueued");
    HT_HIP(c, hipSetDevice(c->device));
    // counters + the first HT_PINNED_HITS hits in one go (pinned host memory), one synchronisation per batch
    // (speculative: as many hits as a batch of this size usually has — 64 per frame, at least 256 —, not the whole staging buffer:
    // a live feed's single frame would otherwise wait for a 196 KB copy it almost never needs; the rare excess is fetched below)
    // ... and, once a batch of this context has been collected, 1.5 x what that one had: a 256-frame C2 batch has ~1.4 k hits, not 16 k)
    uint32_t spec = std::max<uint32_t>(256u, 64u * (uint32_t)std::max(c->enq_nframes, 1));
    if (c->spec_hint) spec = std::min(spec, std::max<uint32_t>(256u, c->spec_hint + c->spec_hint / 2 + 64u));
    spec = std::min<uint32_t>(spec, std::min<uint32_t>(HT_PINNED_HITS, c->hit_capacity));
    HT_HIP(c, hipMemcpyAsync(c->h_pinned, c->d_counters, sizeof(HtCounters) + (size_t)spec * sizeof(ht_hit), hipMemcpyDeviceToHost, c->stream));  // one copy: they are contiguous
    // the whitebalance sums of THIS batch travel with its counters: a re-enqueue below (or any later enqueue) zeroes and refills the
    // device sums, ht_detect_whitebalance reports the snapshot of the batch collected last
    const bool wb_snap = c->wb_enqueued && c->h_wb_pinned;
    if (wb_snap)
        HT_HIP(c, hipMemcpyAsync(c->h_wb_pinned, c->d_scratch, sizeof(unsigned long long) * 4 * (size_t)c->enq_nframes, hipMemcpyDeviceToHost, c->stream));
    HT_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(&c->h_counters, c->h_pinned, sizeof(HtCounters));
    c->enqueued = false;
    if (wb_snap) {
        c->h_wb_sums.assign(c->h_wb_pinned, c->h_wb_pinned + 4 * (size_t)c->enq_nframes);
        c->wb_collected_n = c->enq_nframes;
    } else {
        c->wb_collected_n = -1;
    }
    std::memset(c->h_stage_in, 0, sizeof(c->h_stage_in));
    if (c->stats_enqueued) {
        std::vector<unsigned long long> sh((size_t)64 * HT_STAT_SHARDS);
        HT_HIP(c, hipMemcpy(sh.data(), c->d_stats, sh.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int r = 0; r < HT_STAT_SHARDS; r++)
            for (int j = 0; j < 64; j++) c->h_stage_in[j] += sh[(size_t)r * 64 + j];
    }
    const uint32_t found = c->h_counters.nhits;
    c->spec_hint = found;
    if (total) *total = found;
    const uint32_t nfr = (uint32_t)c->enq_nframes;
    if (counts) std::memset(counts, 0, sizeof(uint32_t) * (size_t)nfr);
    if (found > c->hit_capacity)
        return ht_fail(c, HT_ERR_CAPACITY, "ht_detect_collect: more raw hits than ht_config.hit_capacity; results incomplete");
    std::vector<ht_hit> &tmp = c->h_raw_hits;  // context scratch: no allocation, no zero-fill per batch
    tmp.resize(found);
    if (found) {
        const uint32_t have = std::min(found, spec);
        std::memcpy(tmp.data(), c->h_pinned + sizeof(HtCounters), (size_t)have * sizeof(ht_hit));
        if (found > have)
            HT_HIP(c, hipMemcpy(tmp.data() + have, c->d_hits + have, (size_t)(found - have) * sizeof(ht_hit), hipMemcpyDeviceToHost));
    }
    // every raw hit of this batch is in host memory: the next batch of the same frames buffer may start while the host sorts and
    // groups this one (ht_detect_collect_best_requeue) — the GPU never runs one batch short during the host's post-processing
    if (c->requeue_flags >= 0) {
        const uint32_t fl = (uint32_t)c->requeue_flags;
        c->requeue_flags = -1;
        ht_status rq = ht_detect_enqueue(c, fl);
        if (rq != HT_OK) return rq;
    }
    // emission order (frame, scale, q, y, x).  The frame is the major key and a frame has few hits: a counting sort by frame
    // straight into the destination, then each frame's handful ordered by one packed 48-bit key — a 256-frame batch's 1.4 k hits
    // took ~0.1 ms of the host's 0.25 ms per batch in one std::sort with the five-field comparator (profiles/r03_host_post.txt)
    const uint32_t ncopy = std::min(found, cap);
    ht_hit *dst = hits;
    if (found && (!hits || cap < found)) {
        c->h_ordered_hits.resize(found);
        dst = c->h_ordered_hits.data();
    }
    bool bucketed = found > 0 && nfr > 0;
    if (bucketed) {
        // emission order (frame, scale, q, y, x).  The frame is the major key and a frame has few hits: a counting sort by frame straight
        // into the destination, then each frame's handful ordered by one packed 48-bit key (ht_hostpost.h) — a 256-frame batch's 1.4 k
        // hits took ~0.1 ms of the host's 0.25 ms per batch in one std::sort with the five-field comparator (profiles/r03_host_post.txt)
        bucketed = ht_post_bucket_by_frame(tmp.data(), found, nfr, dst, c->h_frame_start, counts);
        if (bucketed) {
            const uint32_t *se = c->h_frame_start.data();
            if (c->collect_best_follows) {
                c->h_sort_deferred = true;  // ht_detect_collect_best orders every frame inside its own per-frame pass
            } else {
                auto sort_frames = [&](int f0, int f1) {
                    for (int f = f0; f < f1; f++) ht_post_sort_frame(dst, f ? se[f - 1] : 0u, se[f]);
                };
                const int nw = ht_host_workers(c, (int)nfr, found);
                if (nw > 0) HtPool::get().run((int)nfr, 16, nw, sort_frames);
                else sort_frames(0, (int)nfr);
            }
        }
    }
    if (found && !bucketed) {
        std::sort(tmp.begin(), tmp.end(), hit_less);
        if (counts) {
            std::memset(counts, 0, sizeof(uint32_t) * (size_t)nfr);
            for (uint32_t i = 0; i < found; i++)
                if (tmp[i].frame < nfr) counts[tmp[i].frame]++;
        }
        std::memcpy(dst, tmp.data(), (size_t)found * sizeof(ht_hit));
    }
    if (found && dst != hits && hits && ncopy) std::memcpy(hits, dst, (size_t)ncopy * sizeof(ht_hit));
    if (found > cap) return ht_fail(c, HT_ERR_CAPACITY, "ht_detect_collect: caller buffer too small for all hits");
    return HT_OK;
}

extern "C" ht_status ht_detect_batch(ht_ctx *c, const uint8_t *host_rgba, int32_t n, int32_t width, int32_t height, size_t frame_stride,
                                     uint32_t flags, ht_hit *hits, uint32_t cap, uint32_t *counts, uint32_t *total) {
    if (!c) return HT_ERR_INVALID;
    ht_status st;
    if (c->W != width || c->H != height || c->max_batch < n)
        if ((st = ht_set_geometry(c, width, height, n, nullptr, 0)) != HT_OK) return st;
    if ((st = ht_upload_frames(c, host_rgba, n, frame_stride)) != HT_OK) return st;
    if ((st = ht_detect_enqueue(c, flags)) != HT_OK) return st;
    return ht_detect_collect(c, hits, cap, counts, total);
}

extern "C" ht_status ht_pyramid_readback(ht_ctx *c, int32_t frame, int32_t level, int32_t slot, uint8_t *out, size_t cap) {
    if (!c || !out) return HT_ERR_INVALID;
    if (frame < 0 || frame >= c->max_batch || level < 0 || level >= c->nlevels || slot < 0 || slot > 3)
        return ht_fail(c, HT_ERR_INVALID, "ht_pyramid_readback: index out of range");
    const HtDevLevel &L = c->h_levels[level];
    if (L.off[slot] == 0xffffffffu) return ht_fail(c, HT_ERR_INVALID, "ht_pyramid_readback: plane does not exist");
    if (cap < (size_t)L.w * L.h) return ht_fail(c, HT_ERR_CAPACITY, "ht_pyramid_readback: buffer too small");
    if (L.w == 0 || L.h == 0) return HT_OK;
    HT_HIP(c, hipSetDevice(c->device));
    HT_HIP(c, hipStreamSynchronize(c->stream));
    HT_HIP(c, hipMemcpy2D(out, L.w, c->d_arena + (uint64_t)frame * c->arena_stride + L.off[slot], L.stride, L.w, L.h, hipMemcpyDeviceToHost));
    return HT_OK;
}

extern "C" ht_status ht_stage_counts(ht_ctx *c, uint64_t *counts, int32_t n) {
    if (!c || !counts || n < (int32_t)c->nstages + 1) return HT_ERR_INVALID;
    for (uint32_t j = 0; j <= c->nstages; j++) counts[j] = c->h_stage_in[j];
    for (int32_t j = (int32_t)c->nstages + 1; j < std::min<int32_t>(n, 64); j++) counts[j] = c->h_stage_in[j];  // raw counter row (timeline builds)
    return HT_OK;
}

extern "C" ht_status ht_grayscale_batch(ht_ctx *c, uint8_t *host_rgba, int32_t n, int32_t width, int32_t height, size_t frame_stride) {
    if (!c || !host_rgba || n <= 0 || width <= 0 || height <= 0 || frame_stride < (size_t)width * height * 4) return HT_ERR_INVALID;
    HT_HIP(c, hipSetDevice(c->device));
    const size_t fbytes = (size_t)width * height * 4;
    uint8_t *d = nullptr;
    if (hipMalloc(&d, fbytes * n) != hipSuccess) return ht_fail(c, HT_ERR_NOMEM, "ht_grayscale_batch: hipMalloc failed");
    ht_status st = HT_OK;
    hipError_t e = hipMemcpy2DAsync(d, fbytes, host_rgba, frame_stride, fbytes, n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const int W0 = c->W, H0 = c->H;
        c->W = width, c->H = height;  // the in-place kernel only needs the pixel count
        st = ht_launch_gray_inplace(c, d, n, fbytes);
        c->W = W0, c->H = H0;
    }
    if (e == hipSuccess && st == HT_OK) e = hipMemcpy2DAsync(host_rgba, frame_stride, d, fbytes, fbytes, n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return ht_fail(c, HT_ERR_HIP, std::string("ht_grayscale_batch: ") + hipGetErrorString(e));
    return st;
}

// per-frame channel sums -> getWhitebalance values (whitebalance.js:14-26)
static void wb_values(const ht_ctx *c, const unsigned long long *sums, double *out, int32_t n) {
    const double imagesize = (double)c->W * (double)c->H;  // whitebalance.js:14
    for (int i = 0; i < n; i++) {
        // r, g, b are sums of integers < 2^53: exactly what the reference's double accumulation holds (whitebalance.js:17-21)
        const double avgr = (double)sums[4 * i] / imagesize, avgg = (double)sums[4 * i + 1] / imagesize, avgb = (double)sums[4 * i + 2] / imagesize;
        out[i] = (avgr + avgg + avgb) / 3;  // whitebalance.js:23-26
    }
}

extern "C" ht_status ht_whitebalance_batch(ht_ctx *c, double *out, int32_t n) {
    if (!c || !out) return HT_ERR_INVALID;
    if (!c->d_frames || n <= 0 || n > c->nframes) return ht_fail(c, HT_ERR_STATE, "ht_whitebalance_batch: no frames bound");
    HT_HIP(c, hipSetDevice(c->device));
    ht_status st = wb_scratch(c);
    if (st != HT_OK) return st;
    unsigned long long *d_sums = wb_standalone_region(c);  // never the region a batch in flight accumulates into
    if ((st = ht_launch_whitebalance(c, reinterpret_cast<double *>(d_sums), true)) != HT_OK) return st;
    std::vector<unsigned long long> sums((size_t)n * 4);
    HT_HIP(c, hipMemcpyAsync(sums.data(), d_sums, sizeof(unsigned long long) * 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HT_HIP(c, hipStreamSynchronize(c->stream));
    wb_values(c, sums.data(), out, n);
    return HT_OK;
}

extern "C" ht_status ht_detect_whitebalance(ht_ctx *c, double *out, int32_t n) {
    if (!c || !out) return HT_ERR_INVALID;
    if (c->wb_collected_n < 0)
        return ht_fail(c, HT_ERR_STATE, "ht_detect_whitebalance: the batch collected last was not enqueued with HT_DETECT_WHITEBALANCE (call after ht_detect_collect)");
    if (n <= 0 || n > c->wb_collected_n) return ht_fail(c, HT_ERR_INVALID, "ht_detect_whitebalance: n exceeds the collected batch");
    wb_values(c, c->h_wb_sums.data(), out, n);  // host snapshot: no device work, no wait for a batch that was re-enqueued meanwhile
    return HT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// host post-processing

extern "C" ht_status ht_hits_to_rects(const ht_ctx *c, const ht_hit *hits, uint32_t n, ht_rect *out) {
    if (!c || (n && (!hits || !out))) return HT_ERR_INVALID;
    const HtPostCfg cfg = post_cfg(c);
    double sx[HT_MAX_LEVELS];
    ht_post_level_scales(cfg, sx);
    return ht_post_hits_to_rects(cfg, sx, hits, n, out);
}

extern "C" ht_status ht_group_rects(const ht_rect *seq, uint32_t n, int32_t min_neighbors, ht_rect *out, uint32_t *nout) {
    return ht_post_group_rects(seq, n, min_neighbors, out, nout);
}

extern "C" ht_status ht_best_faces(const ht_ctx *c, const ht_hit *hits, const uint32_t *counts, int32_t nframes, int32_t min_neighbors,
                                   ht_rect *best) {
    if (!c || !counts || !best || nframes < 0) return HT_ERR_INVALID;
    const HtPostCfg cfg = post_cfg(c);
    double sx[HT_MAX_LEVELS];
    ht_post_level_scales(cfg, sx);  // once per batch, not once per frame
    size_t k = 0;
    for (int f = 0; f < nframes; f++) {
        const ht_status st = ht_post_best_face(cfg, sx, hits ? hits + k : nullptr, counts[f], min_neighbors, &best[f]);
        if (st != HT_OK) return st;
        k += counts[f];
    }
    return HT_OK;
}

extern "C" ht_status ht_detect_collect_best(ht_ctx *c, int32_t min_neighbors, ht_rect *best, uint32_t *total_hits) {
    if (!c || !best) return HT_ERR_INVALID;
    HtRange range("ht_detect_collect_best");
    if (!c->enqueued) return ht_fail(c, HT_ERR_STATE, "ht_detect_collect_best: nothing enqueued");
    const int nfr = c->enq_nframes;
    // the context's own buffers: nothing but the per-frame rects crosses the ABI (a batch server calls this once per batch)
    if (c->h_collect_hits.size() < (size_t)c->hit_capacity) c->h_collect_hits.resize(c->hit_capacity);
    c->h_collect_counts.resize((size_t)std::max(nfr, 1));
    uint32_t total = 0;
    c->collect_best_follows = true;  // the hits arrive bucketed by frame; each frame's ordering happens in the per-frame pass below
    c->h_sort_deferred = false;
    ht_status st = ht_detect_collect(c, c->h_collect_hits.data(), c->hit_capacity, c->h_collect_counts.data(), &total);
    c->collect_best_follows = false;
    if (total_hits) *total_hits = total;
    if (st != HT_OK) return st;
    if (!c->h_sort_deferred) return ht_best_faces(c, c->h_collect_hits.data(), c->h_collect_counts.data(), nfr, min_neighbors, best);
    // One pass per frame — order its hits (emission order: scale, q, y, x), seq rects, grouping, best face — dealt out to the host pool
    // (ht_hostpost.h): frames are independent and write their own slots, byte-identical to the single-threaded order
    // (tests/test_host_post.py runs the same code under ASan with 0 and 7 workers).  A C2 batch took 0.144 ms of one core here, more
    // than half of the 0.25 ms the GPU needs for it (profiles/r03_host_post.txt).
    return ht_post_frames(post_cfg(c), c->h_collect_hits.data(), c->h_frame_start.data(), nfr, min_neighbors, ht_host_workers(c, nfr, total), best);
}

extern "C" ht_status ht_detect_collect_best_requeue(ht_ctx *c, int32_t min_neighbors, ht_rect *best, uint32_t *total_hits, uint32_t next_flags) {
    if (!c || !best) return HT_ERR_INVALID;
    if (!c->enqueued) return ht_fail(c, HT_ERR_STATE, "ht_detect_collect_best_requeue: nothing enqueued");
    c->requeue_flags = (int64_t)next_flags;
    const ht_status st = ht_detect_collect_best(c, min_neighbors, best, total_hits);
    c->requeue_flags = -1;
    return st;
}

// ---------------------------------------------------------------------------------------------------------
// measurement

extern "C" ht_status ht_profile(ht_ctx *c, int32_t on) {
    if (!c) return HT_ERR_INVALID;
    c->profiling = on != 0;
    return HT_OK;
}

extern "C" ht_status ht_kernel_times(ht_ctx *c, ht_kernel_time *out, int32_t *n, int32_t reset) {
    if (!c || !n) return HT_ERR_INVALID;
    HT_HIP(c, hipSetDevice(c->device));
    HT_HIP(c, hipStreamSynchronize(c->stream));
    for (auto &t : c->timers) {
        for (auto &p : t.pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) t.ms += ms;
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
        t.pending.clear();
    }
    const int cap = *n;
    int k = 0;
    for (auto &t : c->timers) {
        if (out && k < cap) {
            std::memset(&out[k], 0, sizeof(ht_kernel_time));
            std::strncpy(out[k].name, t.name.c_str(), sizeof(out[k].name) - 1);
            out[k].ms = t.ms;
            out[k].launches = t.launches;
        }
        k++;
    }
    // which form of k_cs_track_fused the launches took (chosen per launch, ht_camshift.hip): counted with profiling on or off
    static const char *const form_names[2] = {"cs_fused_launches_1024", "cs_fused_launches_512"};
    for (int f = 0; f < 2; f++) {
        if (!c->cs_fused_launches[f]) continue;
        if (out && k < cap) {
            std::memset(&out[k], 0, sizeof(ht_kernel_time));
            std::strncpy(out[k].name, form_names[f], sizeof(out[k].name) - 1);
            out[k].launches = c->cs_fused_launches[f];
        }
        k++;
    }
    *n = k;
    if (reset) c->timers.clear(), c->cs_fused_launches[0] = c->cs_fused_launches[1] = 0;
    return HT_OK;
}

extern "C" void *ht_stream(const ht_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" ht_status ht_synchronize(ht_ctx *c) {
    if (!c) return HT_ERR_INVALID;
    HT_HIP(c, hipSetDevice(c->device));
    HT_HIP(c, hipStreamSynchronize(c->stream));
    return HT_OK;
}
