/*
 * headtrackr_hip.h — C ABI of libheadtrackr_hip.so: the MI355X (gfx950) implementation of headtrackr's per-frame
 * detect / track hot path.
 *
 * The reference (auduno/headtrackr) has no FFI; its boundary is the set of JavaScript functions that
 * facetrackr.Tracker calls once per frame.  Every entry point below names the reference interface it replaces
 * (paths are /root/reference/src/...).  The N-API shim (headtrackr_amd/csrc/ht_napi.cc) and the JS facade
 * (headtrackr_amd/js/headtrackr.js) bind exactly these symbols; INTEGRATION.md shows the binding.
 *
 * Conventions: plain C, no exceptions; every function returns ht_status (0 = OK) and records a message readable
 * with ht_last_error(); the caller owns every buffer it passes; a ctx is bound to one GPU and is not thread-safe
 * (one call in flight per ctx).  Host buffers may be pageable; *_device entry points take device pointers and never
 * copy.  All work is enqueued on one HIP stream (the caller's, if given in ht_config.stream).
 */
#ifndef HEADTRACKR_HIP_H
#define HEADTRACKR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HT_ABI_VERSION 2
#define HT_MAX_LEVELS 96

typedef int32_t ht_status;
enum {
    HT_OK = 0,
    HT_ERR_INVALID = -1,   /* bad argument / unsupported geometry */
    HT_ERR_HIP = -2,       /* a HIP runtime call failed (message has the HIP error string) */
    HT_ERR_NOMEM = -3,     /* host or device allocation failed */
    HT_ERR_CAPACITY = -4,  /* an output did not fit the caller's buffer (results truncated, counts are exact) */
    HT_ERR_NO_DEVICE = -5, /* no usable gfx950 device */
    HT_ERR_STATE = -6      /* call sequence error (e.g. detect before frames were bound) */
};

typedef struct ht_ctx ht_ctx;

/* flags for ht_detect_* */
enum {
    HT_INPUT_RGBA = 0,      /* colour frame: ccv.grayscale (ccv.js:22-32) is fused into the pyramid build */
    HT_INPUT_GRAY_IN_R = 1, /* byte 0 of each pixel is already gray: exactly what ccv.detect_objects reads (ccv.js:115,171) */
    HT_SCAN_FUSED_TAIL = 0, /* default scan schedule */
    HT_SCAN_NO_SPLIT = 2,   /* run every cascade stage in the tile kernel (no second "deep" kernel); debugging / A-B */
    HT_SCAN_SIMPLE = 4,     /* one thread per window straight from HBM (slow reference kernel); debugging / A-B */
    HT_SCAN_GENERIC = 8,    /* table-driven stage code even for the built-in cascade (no generated straight-line stages) */
    HT_SCAN_STATS = 16,     /* also count the windows entering every stage (ht_stage_counts); costs a few atomics per workgroup */
    HT_DETECT_WHITEBALANCE = 32 /* the gray pass also accumulates getWhitebalance's channel sums (whitebalance.js:5-30): the frame is
                                 * read once for both; fetch the values with ht_detect_whitebalance after ht_detect_collect */
};

typedef struct ht_config {
    uint32_t struct_size;   /* = sizeof(ht_config) */
    int32_t device;         /* HIP device ordinal */
    int32_t interval;       /* ccv.detect_objects `interval` (facetrackr.js:148 passes 5) */
    uint32_t hit_capacity;  /* max raw hits kept per batch (0 = default 1<<20) */
    void *stream;           /* hipStream_t to enqueue on; NULL = library-owned non-blocking stream */
    uint32_t queue_capacity;/* survivors handed from the tile kernel to the deep kernel per batch (0 = auto) */
    uint32_t flags;         /* reserved, 0 */
    const char *options;    /* NULL, or "key=value,key=value,...": per-context schedule selectors for tests and A/B measurements
                             * (ABI 2; a caller that passes the ABI-1 struct_size has none).  The library reads NO environment
                             * variable: what a context computes depends on its arguments alone.  Every key below selects among
                             * schedules that produce IDENTICAL results; an unknown key fails ht_create with HT_ERR_INVALID.
                             *   cs_fused_min=N       streams per call from which camshift runs as one launch (default 192)
                             *   cs_fused_nt=512|1024 threads per workgroup of that launch (0 = per launch: 512 — two workgroups per CU — when it has more
                             *                        streams than the device has CUs or another context of the device that tracks on this path has
                             *                        work in flight at launch time, else 1024)
                             *   cs_seq_fused=0|1     track sequences inside one launch (1)       cs_keep_hist=1   keep histograms for ht_camshift_debug_hist
                             *   cs_cluster=0|1, cs_cluster_min_px=N, cs_region=N                 cluster / LDS-region paths of the few-stream schedule
                             *   cs_barrier_budget=N  shader-clock cycles a cluster exchange may wait before the call fails with HT_ERR_STATE
                             *   cs_flags=0|1         enqueue-only track calls of the cluster path are completed by marks in the pinned slot (1) or an event
                             *   cs_sync_ring=0|1     a synchronous track call takes the enqueue-only route and collects at once (1) or copies back + synchronises (0)
                             *   fp_sparse=0|1        tile kernel: sparse stages one lane per (window, feature) pair when <= 256 pairs are left (1)
                             *   graph_max_frames=N   batches up to N frames replay a captured hipGraph (256; 0 = never)
                             *   split=S, deep_bias=B, deep_v=2|4, deep_grid=N                    tile kernel -> deep kernel hand-off (grid kept >= 16 wavefronts)
                             *   force_exact=1        every integer stage decision re-run on the sequential binary64 path
                             *   rs_bands=0|1         pyramid generations by k_resample_bands (1: LDS-DMA into wave-private source bands) or k_resample (0)
                             *   early_scan=1, rs_rpt, rs_minwg, rs_k, rs_group, rs_tailtable, rs_tailcap, rs_notail, rs_nofast, rs_nosort, rs_gennames
                             *   host_threads=N       worker threads of the host post-processing (0 = single-threaded)
                             *   force_rccl=1         ht_allgather_* goes through RCCL even with one rank
                             * Keys that make results incomplete by design (stop_stage, cs_iters, rs_maxgen) exist only in builds
                             * compiled with -DHT_DEBUG_KNOBS (tools/build_alt.py); the product library rejects them. */
} ht_config;

/* One raw detection = one element of ccv.detect_objects' `seq` (ccv.js:227-234) in index form:
 * rect = { x:(4*x+2*(q&1))*s, y:(4*y+2*(q>>1))*s, width:cw*s, height:ch*s, confidence:sum }, s = scale^i. */
typedef struct ht_hit {
    uint32_t frame;
    uint16_t x, y;     /* window index on the quarter-resolution plane */
    uint8_t scale;     /* i, ccv.js:154 */
    uint8_t q;         /* half-pixel phase, ccv.js:151-152,178 */
    uint16_t reserved0;
    uint32_t reserved1;
    double sum;        /* last stage's sum (binary64, accumulated in the reference's order) */
} ht_hit;

typedef struct ht_rect { /* element of detect_objects' result (ccv.js:228-233 raw, 297-302 grouped) */
    double x, y, width, height, confidence;
    int32_t neighbors;
    int32_t reserved;
} ht_rect;

typedef struct ht_plane_info {
    int32_t width, height; /* canvas size of the pyramid level */
    int32_t stride;        /* bytes per row in the device arena */
    int32_t present;       /* 0 if this (level, slot) does not exist */
    uint64_t offset;       /* byte offset inside one frame's arena */
} ht_plane_info;

typedef struct ht_cs_rect { int32_t x, y, width, height; } ht_cs_rect;

/* camshift.Tracker's persistent per-stream state (camshift.js:153-160): lives on the device, one per stream. */
typedef struct ht_cs_trackobj { /* camshift.TrackObj, camshift.js:362-378 (+ the search window, camshift.js:162-165) */
    double x, y, width, height, angle;
    int32_t sw_x, sw_y, sw_width, sw_height;
} ht_cs_trackobj;

typedef struct ht_kernel_time { /* per-kernel device time of the last ht_detect_* call when profiling is on */
    char name[32];
    double ms;
    uint32_t launches;
    uint32_t reserved;
} ht_kernel_time;

/* ---- lifetime ------------------------------------------------------------------------------------------ */

/* Creates a context on cfg->device holding the cascade (an "HTCB" blob, see headtrackr_amd/js/cascade_pack.js)
 * = the `cascade` argument of ccv.detect_objects (ccv.js:109; data: cascade.js:19). */
ht_status ht_create(const ht_config *cfg, const void *cascade_blob, size_t cascade_len, ht_ctx **out);
/* Lifetime of shared frame buffers: a buffer of ht_device_alloc that OTHER live contexts still have frames bound inside
 * (ht_bind_frames_device) is not freed by its owner's ht_destroy — it stays alive and is released by the ht_destroy after which no
 * live context is bound inside it.  The recommended order is still binders first, owner last.  A context must not be re-bound
 * concurrently (another thread) with the destruction / ht_device_free of the buffer it is bound to. */
void ht_destroy(ht_ctx *ctx);
/* Message of the last failure on ctx (ctx == NULL: last failure of ht_create on this thread). Never NULL. */
const char *ht_last_error(const ht_ctx *ctx);
int32_t ht_abi_version(void);

/* ---- geometry ------------------------------------------------------------------------------------------ */

/* Fixes frame size and batch capacity; (re)allocates the pyramid arena.  level_dims (optional, 2*nlevels int32:
 * w0,h0,w1,h1,...) lets a JavaScript host pass the sizes V8 computed with Math.pow/Math.floor (ccv.js:119-120,
 * 126-127); NULL = computed here (identical for interval 5, see oracle/ht_oracle.c HO_V8_SCALE6_POW). */
ht_status ht_set_geometry(ht_ctx *ctx, int32_t width, int32_t height, int32_t max_batch, const int32_t *level_dims,
                          int32_t nlevels);
int32_t ht_num_levels(const ht_ctx *ctx);
ht_status ht_plane(const ht_ctx *ctx, int32_t level, int32_t slot, ht_plane_info *out);
uint64_t ht_windows_per_frame(const ht_ctx *ctx);   /* sliding windows scanned per frame */
uint64_t ht_pyramid_bytes_per_frame(const ht_ctx *ctx); /* sum of w*h over all planes (gray bytes) */

/* ---- frames -------------------------------------------------------------------------------------------- */

/* Copies n RGBA frames (frame_stride bytes apart, rows packed) from host memory into the ctx's device buffer. */
ht_status ht_upload_frames(ht_ctx *ctx, const uint8_t *host_rgba, int32_t n, size_t frame_stride);
/* Double-buffered ingest for live feeds (the reference's per-frame video -> canvas copy, main.js:170):
 * ht_upload_frames_async copies the NEXT frames from pinned host memory into the ctx's back buffer on a separate copy
 * stream, so the copy overlaps the kernels working on the current frames; ht_swap_frames makes the back buffer current
 * (the compute stream waits for the copy, no host synchronisation).  host_rgba must stay valid until the swap. */
ht_status ht_upload_frames_async(ht_ctx *ctx, const uint8_t *host_rgba, int32_t n, size_t frame_stride);
ht_status ht_swap_frames(ht_ctx *ctx);
/* Uses frames already resident in device memory (no copy; must stay valid until the results were collected). */
ht_status ht_bind_frames_device(ht_ctx *ctx, const void *dev_rgba, int32_t n, size_t frame_stride);

/* Frames currently bound (ht_upload_frames / ht_swap_frames / ht_bind_frames_device) and frames of the batch enqueued last: the
 * collect calls report on the ENQUEUED batch — size counts[] / best[] with ht_frames_enqueued, not with what is bound by then. */
int32_t ht_frames_bound(const ht_ctx *ctx);
int32_t ht_frames_enqueued(const ht_ctx *ctx);

/* Memory for hosts that have no HIP binding of their own (the Node addon): pinned host buffers — frames in them cross PCIe at link
 * speed and may be handed to ht_upload_frames_async — and device buffers for frames that stay resident in HBM across calls
 * (ht_bind_frames_device, ht_camshift_track_sequence).  ht_device_upload copies host -> device and returns when src may be reused. */
ht_status ht_host_alloc(size_t bytes, void **out);
void ht_host_free(void *p);
ht_status ht_device_alloc(ht_ctx *ctx, size_t bytes, void **out);
/* Frees a buffer of ht_device_alloc on the context that allocated it.  Fails with HT_ERR_STATE (nothing freed) while ANOTHER live
 * context still has frames bound inside it (ht_bind_frames_device): rebind or destroy that context first. */
ht_status ht_device_free(ht_ctx *ctx, void *p);
ht_status ht_device_upload(ht_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);

/* ---- detect: ccv.grayscale + ccv.detect_objects (ccv.js:22-32, 109-246) ---------------------------------- */

/* Enqueues gray -> pyramid -> cascade scan for the bound frames on the stream and returns immediately. */
ht_status ht_detect_enqueue(ht_ctx *ctx, uint32_t flags);
/* Waits for the enqueued work, copies the raw hits back, sorted in the reference's emission order
 * (frame, scale, q, y, x) (ccv.js:154,178,181-182).  counts[f] = hits of frame f (counts may be NULL);
 * *total = all hits found.  HT_ERR_CAPACITY if total > cap (first cap hits in order are returned). */
ht_status ht_detect_collect(ht_ctx *ctx, ht_hit *hits, uint32_t cap, uint32_t *counts, uint32_t *total);
/* Convenience: set geometry if needed + upload + enqueue + collect, for host-resident frames. */
ht_status ht_detect_batch(ht_ctx *ctx, const uint8_t *host_rgba, int32_t n, int32_t width, int32_t height,
                          size_t frame_stride, uint32_t flags, ht_hit *hits, uint32_t cap, uint32_t *counts,
                          uint32_t *total);
/* Test hook: copies one pyramid plane of one frame back, rows packed (width*height bytes). */
ht_status ht_pyramid_readback(ht_ctx *ctx, int32_t frame, int32_t level, int32_t slot, uint8_t *out, size_t cap);
/* Scan statistics of the last collected batch (only if it was enqueued with HT_SCAN_STATS): windows that entered
 * stage j, j = 0..nstages (nstages = full survivors). */
ht_status ht_stage_counts(ht_ctx *ctx, uint64_t *counts, int32_t n);

/* ccv.grayscale drop-in on host RGBA frames, in place (R=G=B=gray, A untouched; ccv.js:22-32). */
ht_status ht_grayscale_batch(ht_ctx *ctx, uint8_t *host_rgba, int32_t n, int32_t width, int32_t height,
                             size_t frame_stride);
/* headtrackr.getWhitebalance (whitebalance.js:5-30) for the bound frames. */
ht_status ht_whitebalance_batch(ht_ctx *ctx, double *out, int32_t n);
/* The same values for the first n frames of the batch last enqueued with HT_DETECT_WHITEBALANCE (fused into the gray pass:
 * facetrackr's white-balance gate, facetrackr.js:79-95, and its detection read the frame once). Call after ht_detect_collect. */
ht_status ht_detect_whitebalance(ht_ctx *ctx, double *out, int32_t n);

/* ---- host-side post-processing of raw hits (O(n^2) on tens of rects; stays on the CPU by design) ---------- */

/* seq elements from hits (ccv.js:228-233, scale_x by repeated multiplication ccv.js:244-245). */
ht_status ht_hits_to_rects(const ht_ctx *ctx, const ht_hit *hits, uint32_t n, ht_rect *out);
/* ccv.array_group + averaging + nested-rect filter (ccv.js:34-107, 249-332). *nout <= n. */
ht_status ht_group_rects(const ht_rect *seq, uint32_t n, int32_t min_neighbors, ht_rect *out, uint32_t *nout);

/* facetrackr.Tracker.doVJDetection's selection for a whole batch (facetrackr.js:147-175): per frame, group the raw hits
 * with min_neighbors and keep the rect with the highest confidence (strict '>', first wins).  best[f].neighbors == 0 and
 * confidence == -10000 when frame f has no detection (facetrackr.js:239). counts[f] = raw hits of frame f, hits sorted as
 * ht_detect_collect returns them. */
ht_status ht_best_faces(const ht_ctx *ctx, const ht_hit *hits, const uint32_t *counts, int32_t nframes, int32_t min_neighbors,
                        ht_rect *best);

/* ht_detect_collect + ht_best_faces in one call for batch hosts: waits for the enqueued batch, sorts and groups its raw hits and
 * writes one rect per frame of the batch (facetrackr.js:147-175); the hits stay in the context.  *total_hits = raw hits found. */
ht_status ht_detect_collect_best(ht_ctx *ctx, int32_t min_neighbors, ht_rect *best, uint32_t *total_hits);
/* The same, and as soon as the batch's raw hits are in host memory the NEXT batch of the currently bound frames is enqueued with
 * next_flags (ht_detect_enqueue) — before this batch is sorted and grouped, so the GPU is not one batch short while the host
 * post-processes.  A streaming host that swaps in new frames first (ht_swap_frames) gets them in that next batch. */
ht_status ht_detect_collect_best_requeue(ht_ctx *ctx, int32_t min_neighbors, ht_rect *best, uint32_t *total_hits, uint32_t next_flags);

/* ---- camshift: camshift.Tracker (camshift.js:148-354), one tracker per stream ----------------------------- */

/* (Re)allocates per-stream tracker state for n streams. */
ht_status ht_camshift_reserve(ht_ctx *ctx, int32_t nstreams);
/* initTracker (camshift.js:198-211) for streams [first, first+n) using bound frames [0, n) and one rect each. */
ht_status ht_camshift_init_batch(ht_ctx *ctx, int32_t first, int32_t n, const ht_cs_rect *rects);
/* track (camshift.js:213-312) for streams [first, first+n) on bound frames [0, n); out[n] (may be NULL: enqueue only).  With out != NULL the call
 * returns the track objects; when no enqueue-only call is outstanding it takes the same route internally (pinned slot + completion marks, no
 * device-to-host copy, no stream synchronisation: option cs_sync_ring) — the results are the same either way. */
ht_status ht_camshift_track_batch(ht_ctx *ctx, int32_t first, int32_t n, int32_t calc_angles, ht_cs_trackobj *out);
/* Results of the OLDEST outstanding ht_camshift_track_batch that was enqueued with out == NULL (same n): waits for that call only and
 * copies its track objects.  Up to 4 enqueue-only calls may be outstanding per context (a fifth fails with HT_ERR_STATE): their kernels
 * write the track objects into a ring of pinned host slots, so a streaming host enqueues the track() of frame i + 1 before it waits for
 * frame i (the search window that links them lives on the device), and a host that serves several feeds on several contexts enqueues every
 * feed's track() first and collects afterwards (the reference's loop, main.js:168-180, is one feed; this is its K-feed form).
 * ht_camshift_reserve drops uncollected results. */
ht_status ht_camshift_track_collect(ht_ctx *ctx, int32_t n, ht_cs_trackobj *out);
/* ncalls successive track() calls (camshift.js:213-220 called once per video frame, main.js:168-180) for streams
 * [first, first+n) in ONE host call: call k uses the n device-resident frames at dev_frames[k] (frame_stride bytes apart;
 * same geometry as ht_set_geometry).  A stream's calls are sequentially dependent (its search window), so they are
 * enqueued back to back on the ctx stream with no host round trip in between.  out (may be NULL: enqueue only) receives
 * the track objects of the LAST call (out_all == 0, n entries) or of every call (out_all != 0, ncalls*n entries, call-major). */
ht_status ht_camshift_track_sequence(ht_ctx *ctx, int32_t first, int32_t n, int32_t calc_angles, const void *const *dev_frames,
                                     int32_t ncalls, size_t frame_stride, ht_cs_trackobj *out, int32_t out_all);
/* Results of the last ht_camshift_track_sequence that was enqueued with out == NULL (same n / ncalls / out_all): waits for it and
 * copies the track objects.  Lets a host overlap the tracking of one batch of streams with other work on another context
 * (the reference's main.js:168-180 consumes the track object of frame k while the camera already delivers frame k+1). */
ht_status ht_camshift_sequence_collect(ht_ctx *ctx, int32_t n, int32_t ncalls, int32_t out_all, ht_cs_trackobj *out);
/* Measurement hook (B_track = 4*W*H + 4*sum of window areas): per stream, the pixels read by the window
 * moment passes (camshift.js:79-120 called from camshift.js:284-306) and the number of track() calls since the last reset. */
ht_status ht_camshift_stats(ht_ctx *ctx, int32_t first, int32_t n, uint64_t *window_pixels, uint64_t *calls, int32_t reset);
/* Test hook: one stream's model histogram (camshift.js:206-208) and the full-frame histogram of its last track() call
 * (camshift.js:268), 4096 bins each (camshift.Histogram, camshift.js:49-72).  Either pointer may be NULL. */
ht_status ht_camshift_debug_hist(ht_ctx *ctx, int32_t stream, uint32_t *model, uint32_t *current);

/* ---- multi-GPU: fixed-size result records, all-gathered over RCCL/xGMI ------------------------------------ */

/* Single-process helper for hosts that drive several GPUs from one process (the Node addon): ctxs[i] are contexts
 * on distinct devices; records_dev[i] points to nranks*bytes_per_rank bytes of device memory on ctxs[i]'s device
 * with rank i's own records already at offset i*bytes_per_rank.  Performs one ncclAllGather per rank in a group. */
ht_status ht_allgather_records(ht_ctx *const *ctxs, int32_t nranks, void *const *records_dev, size_t bytes_per_rank);
/* The exchange step of the frame-sharded path for a single-process host (BASELINE.json: "RCCL all-gather of bounding boxes"):
 * best[i] = rank i's ht_best_faces output for its frames_per_rank frames (host memory; pad short ranks with zero rects).
 * Uploads every rank's rects into its own GPU's slot, runs ht_allgather_records, reads every rank's table back, requires
 * them to be identical and returns the table (nranks*frames_per_rank rects, rank-major) — what facetrackr.Tracker.doVJDetection
 * (facetrackr.js:147-175) would have produced for every frame of the batch, now known on every GPU. */
ht_status ht_allgather_best_faces(ht_ctx *const *ctxs, int32_t nranks, const ht_rect *const *best, int32_t frames_per_rank,
                                  ht_rect *gathered);
/* Number of HIP devices visible to the process (0 if none); the `devices` option of the JS batch entry points indexes them. */
int32_t ht_device_count(void);

/* ---- measurement --------------------------------------------------------------------------------------- */

/* on != 0: bracket every kernel of subsequent ht_detect_* / camshift calls with HIP events on the ctx stream. */
ht_status ht_profile(ht_ctx *ctx, int32_t on);
/* Device times accumulated since profiling was switched on (or last reset); *n in: capacity, out: entries.  Profiling on or off, the
 * entries cs_fused_launches_1024 / cs_fused_launches_512 (ms = 0) count the single-launch camshift kernel's launches per form since the
 * last reset (the form is chosen per launch, option cs_fused_nt). */
ht_status ht_kernel_times(ht_ctx *ctx, ht_kernel_time *out, int32_t *n, int32_t reset);
void *ht_stream(const ht_ctx *ctx); /* the hipStream_t the ctx enqueues on */
/* ht_detect_enqueue calls served by replaying a captured hipGraph (batches of <= 256 frames, option graph_max_frames: the ~10 dependent launches of a detect
 * sequence are captured once per (frames pointer, count, flags) and replayed with one hipGraphLaunch). */
uint64_t ht_graph_launches(const ht_ctx *ctx);
ht_status ht_synchronize(ht_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* HEADTRACKR_HIP_H */
