// ht_internal.h — shared host/device definitions of libheadtrackr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <string>
#include <vector>

#include "headtrackr_hip.h"

#define HT_MAXPTS 8

// ---------------------------------------------------------------------------------------------------------
// Cascade, host view of the "HTCB" blob (headtrackr_amd/js/cascade_pack.js) == headtrackr.cascade (cascade.js:19)
struct HtBlobFeature {
    uint8_t size, pad[7];
    int8_t px[HT_MAXPTS], py[HT_MAXPTS], pz[HT_MAXPTS];
    int8_t nx[HT_MAXPTS], ny[HT_MAXPTS], nz[HT_MAXPTS];
    double alpha[2];
};
struct HtBlobStage {
    uint32_t count, first;
    double threshold;
};
static_assert(sizeof(HtBlobFeature) == 72 && sizeof(HtBlobStage) == 16, "HTCB layout");

// ---------------------------------------------------------------------------------------------------------
// Device-side cascade tables.
//
// Tile kernel: every point of every feature as a byte offset into the workgroup's LDS tile, relative to the
// window base (see ht_scan.hip "unified-base layout"); read with uniform (scalar) loads, one feature at a time.
struct alignas(64) HtTileFeature {
    uint32_t po[HT_MAXPTS / 2];  // positive-point offsets, two u16 per word (low half first); valid ones first, count = np
    uint32_t no[HT_MAXPTS / 2];  // negative-point offsets, count = nn
    uint32_t a[4];               // alpha[2k] (lo,hi words), alpha[2k+1] (lo,hi)  (ccv.js:194,219)
    uint32_t np, nn;
    uint32_t pad[2];
};
static_assert(sizeof(HtTileFeature) == 64, "HtTileFeature");

// Deep kernel: coordinate form, one feature per lane.  Slots >= np / nn repeat slot 0 (min/max are idempotent).
struct alignas(16) HtDeepFeature {
    uint8_t px[HT_MAXPTS], py[HT_MAXPTS], pz[HT_MAXPTS];  // each array is read as one 64-bit word on the device
    uint8_t nx[HT_MAXPTS], ny[HT_MAXPTS], nz[HT_MAXPTS];
    int64_t a0i, a1i;  // alpha * 1e8 as exact integers (valid when the cascade is "decimal", see ht_context.hip)
    double a0, a1;
};
static_assert(sizeof(HtDeepFeature) == 80, "HtDeepFeature");

// Deep kernel: offsets into the per-wavefront window patch in LDS (24x24 + 12x12 + 6x6 bytes, see ht_scan.hip).
struct alignas(16) HtPatchFeature {
    uint16_t poff[HT_MAXPTS];  // slots >= np repeat slot 0
    uint16_t noff[HT_MAXPTS];
    int64_t a0i, a1i;          // alpha * 1e8 as exact integers
    double a0, a1;
};
static_assert(sizeof(HtPatchFeature) == 64, "HtPatchFeature");

// Deep kernel, LDS-resident form: 32-byte record, the whole tail of the cascade (stages >= split) is copied into LDS once
// per workgroup.  Usable when every feature has <= 5 points per polarity and |alpha * 1e8| < 2^31 (decimal cascade).
struct alignas(16) HtPackedFeature {
    uint16_t off[10];  // p0..p4, n0..n4 patch offsets (unused slots repeat slot 0 of their polarity)
    int32_t a0i, a1i;  // alpha * 1e8; the binary64 alpha is recovered exactly as (double)a / 1e8
    uint32_t pad;
};
static_assert(sizeof(HtPackedFeature) == 32, "HtPackedFeature");

struct HtDevStage {
    uint32_t first, count;
    uint32_t maxpts;  // max(np, nn) over the stage's features
    uint32_t pad;
    double threshold;
    int64_t thri;  // threshold * 1e8
};

// ---------------------------------------------------------------------------------------------------------
// Pyramid geometry (ccv.js:110-147), device copy.
struct HtDevLevel {
    int32_t w, h, stride;
    uint32_t off[4];  // byte offset of slot 0..3 inside one frame's arena (0xffffffff = absent)
};

// One resample job = one canvas of the pyramid (ccv.js:121,128,135,140,145).
// k_resample_tail argument: generations [first, first + ngen) as ranges into the tail job table
constexpr int HT_TAIL_MAX_GENS = 8;
constexpr int HT_TAIL_MAX_JOBS = 32;  // per generation (6 levels x 4 variants = 24 in the reference's pyramid)
struct HtTailGens {
    int32_t ngen;
    int32_t job_begin[HT_TAIL_MAX_GENS + 1];
    uint32_t groups[HT_TAIL_MAX_GENS];  // 4-pixel groups (canvas width rounded up to 4, times canvas height) per generation
    uint32_t tap_begin[HT_TAIL_MAX_GENS + 1];  // range of the generation's entries in the tap tables (its jobs' taps are contiguous)
};
constexpr int HT_TAIL_LDS_TAPS = 4096;  // compact taps of one generation kept in LDS by k_resample_tail (32 KB)

// One tap of the declared resampler (oracle/canvas_shim.js): destination coordinate i of a drawImage call reads source samples a
// and b = min(a + 1, s - 1) (absolute, incl. the source rect origin) with weights u = 1 - t and t.  Computed on the device by
// rs_tap (ht_pyramid.hip) and, for the tail kernel's jobs, once on the host by ht_host_tap — the same binary64 operations.
struct HtTap {
    double t, u;
    int32_t a, b;
};
// k_resample_tail reads its taps from tables built by the host: the compact form {a, (float)t} for the binary32 estimate (column
// tables are padded by 3 entries so that 4 consecutive ones can always be loaded), the full form for the rare binary64 fallback
struct HtTapFast {
    int32_t a;
    float tf;
};
struct HtTailTapRef {
    uint32_t col, row;  // first column / row tap of the job in both tables
    uint32_t mode;      // bit 0: exact 2:1 in both directions (integer 2x2 box mean), bit 1: binary64 everywhere (option rs_nofast)
    uint32_t pad;
};

// One drawImage call (host job list) and, with the tile fields filled in, one k_resample workgroup (device tile table).
constexpr int HT_RS_SRC_ROWS = 75;   // k_resample LDS window: source rows per tile (3 staging passes of 25 rows; 16 * 4 * 1.1225 + 3 = 74.8 still fits)
constexpr int HT_RS_MAX_PASSES = 4;  // k_resample: 16-row passes per tile at most
struct HtResampleJob {
    uint32_t src_off, dst_off;
    int32_t src_stride, dst_stride;
    int32_t sx, sy, sw, sh;  // source rect
    int32_t dw, dh;          // destination rect (at 0,0)
    int32_t cw, ch;          // destination canvas size (pixels outside dw x dh are written 0)
    uint16_t bx, pass0;      // tile record only: tile column (64 px), first 16-row pass
    uint16_t np, pad;        // tile record only: number of 16-row passes (<= HT_RS_MAX_PASSES)
    double rx, ry;           // sw/dw, sh/dh computed on the host (binary64 division)
    // tile record only: the source rectangle the tile's taps touch, from the host (ht_host_tap: the same binary64 operations as
    // rs_tap).  ex_sw16 == 0: not filled in, the kernel derives it (four rs_tap evaluations, ~1 400 cycles at the top of every
    // workgroup before its first load can be issued)
    int32_t ex_xa, ex_ya;    // first source column (rounded down to 16) / row
    int32_t ex_sw16, ex_sh;  // 16-byte chunks per source row, source rows
    // k_resample_bands: wavefront w of the tile owns destination rows [4 np w, 4 np (w + 1)) and the source rows they touch — byte w of
    // band_ya4 = first row relative to ex_ya, byte w of band_sh4 = rows; pad bit 2 says that all four fit a band (<= HT_RSB_ROWS rows)
    uint32_t band_ya4, band_sh4;
};
constexpr int HT_RSB_ROWS = 24;  // k_resample_bands: source rows per wavefront band (4 KB of the 160-byte LDS pitch, one spare row)

// One scan scale (ccv.js:154-160) and its tiling.
struct HtScanScale {
    int32_t l0, l1, l2;   // levels i, i+next, i+2*next
    int32_t qw, qh;       // windows per row / column on the quarter-resolution plane (ccv.js:155-156)
    int32_t tw2, th2;     // tile size in half-window steps X', Y' (X' = 2x+dx, Y' = 2y+dy)
    int32_t ntx, nty;     // tiles per row / column
    uint32_t tile_begin;  // first tile of this scale in the per-frame tile list
    uint32_t div_magic;   // ceil(2^20 / tw2): id / tw2 == (id * magic) >> 20 for id < 4096
    uint32_t win_begin;   // first window of this scale in the flat per-frame window index (simple kernel)
};

// Direct block -> work lookup (one 8-byte load per workgroup instead of a serial scan over a prefix table).
struct HtBlockRef {
    uint16_t item;  // resample: job index in the generation; scan: index into the scale table
    uint16_t bx, by;  // tile column / row (resample: by = first 16-row pass of the tile)
    uint16_t pad;     // resample: number of 16-row passes in this tile; scan: unused
};

// Everything a k_scan_tiles workgroup needs to know about its tile in one 64-byte record = ONE scalar load after the tile index is
// known (it used to be a chain of three dependent lookups — tile -> scale -> three level records — in front of the tile's first
// HBM load; a workgroup holds its 26 KB of LDS while it waits).
struct alignas(64) HtTileRec {
    uint32_t off0, off1, off2[4];  // byte offsets inside a frame's arena: level i, level i+6, the four variants of level i+12
    uint32_t sh0, sh1, sh2;        // stride | height << 16 of the three levels
    uint32_t origin;               // X0 | Y0 << 16: tile origin in half-window steps
    uint32_t size;                 // tw | th << 16: half-window steps of the tile that hold windows (clipped to the scale)
    uint32_t tw2_l0;               // tile pitch tw2 (window id = Y' * tw2 + X') | the scale's level index << 16
    uint32_t div_magic;            // ceil(2^20 / tw2)
    uint32_t strip_magic;          // ceil(2^24 / (4 * th)): stage 0 walks the tile in strips of 4 window pairs (see k_scan_tiles)
    uint32_t pad[2];
};
static_assert(sizeof(HtTileRec) == 64, "HtTileRec");

// Survivor handed from the tile kernel to the deep kernel.  The tile kernel holds the scale's plane offsets and strides in scalar
// registers (its tile record), so the entry carries the window's three plane origins ready-made: the deep kernel used to look the
// scale's three level records up behind the entry — a third dependent memory round trip in front of every window's patch load.
struct alignas(16) HtQueueEntry {
    uint32_t frame;
    uint16_t x, y;
    uint8_t scale, q;
    uint16_t stage;        // first stage the deep kernel has to run
    uint32_t o0, o1, o2;   // byte offsets of the window origin inside the frame's arena: level i (ccv.js:180,235), i + 6 (236), variant q of i + 12 (237)
    uint16_t s0, s1, s2;   // row strides of the three planes
    uint16_t pad;
};
static_assert(sizeof(HtQueueEntry) == 32, "HtQueueEntry");

// k_scan_deep_lds hands its queue entries out through HT_DEEP_CTRS counters, one per 256-byte line, in the 4 KB behind the queue's
// last entry (zeroed by the first workgroup of every k_scan_tiles launch): same-address atomics retire at ~90 per us, one counter for
// the 6 103 windows of a C2 batch made the launch 0.10 ms long whatever the grid.
#define HT_DEEP_CTRS 16
#define HT_DEEP_CTR_BYTES (HT_DEEP_CTRS * 256)
#define HT_PINNED_HITS 8192
#define HT_STAT_SHARDS 256  // rows of 64 u64 counters; a workgroup adds to row (blockIdx & 255)

// Device counters block (zeroed before each batch).
struct HtCounters {
    uint32_t nhits;        // hits appended (may exceed capacity)
    uint32_t nqueue;       // survivors appended to the deep queue (may exceed capacity)
    uint32_t queue_inline; // survivors that did not fit the queue and were finished inside the tile kernel
    uint32_t pad;
};

// camshift per-stream device state (camshift.js:153-160)
struct alignas(16) HtCsState {
    uint32_t model[4096];  // _modelHist
    int32_t sw[4];         // _searchWindow
    double x, y, width, height, angle;  // _trackObj
    unsigned long long win_px;          // measurement: pixels visited by the window moment passes since the last reset
    unsigned long long calls;           // measurement: track() calls since the last reset
};

// A captured detect sequence (memsets + gray + pyramid generations + scan kernels: ~10 dependent launches) replayed with one
// hipGraphLaunch.  Keyed by everything the kernels' arguments depend on besides the geometry (which owns the cache).
// INVARIANT: a captured graph bakes in d_arena, d_counters, d_hits, d_queue, d_stats, the tile / job tables and d_scratch.  The first
// four live as long as the context; the geometry's allocations are only freed by free_geometry(), d_scratch only by wb_scratch() — both
// call destroy_graphs() first.  Anything else that reallocates a buffer a detect kernel reads must do the same.
struct HtDetectGraph {
    const uint8_t *frames = nullptr;
    size_t frame_stride = 0;
    int nframes = 0;
    uint32_t flags = 0;
    int seen = 0;                  // plain enqueues of this key so far (the sequence is captured on the second one)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

// ---------------------------------------------------------------------------------------------------------
struct HtKernelTimer {
    std::string name;
    double ms = 0;
    uint32_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct ht_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // early scan (opt-in, option early_scan=1): the scales whose three planes exist after pyramid generation `early_gen` are
    // scanned on a second stream while the main stream builds the remaining (small, latency-bound) generations.  Measured with
    // 3 batches in flight: +2.5 % at 128 x 720p, -12 % at 256 x 320x240 (the other batches already fill those gaps; the extra
    // concurrency only adds contention), so it is off by default; it shortens the latency of a single batch in flight.
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_early_ready = nullptr, ev_early_done = nullptr;
    bool early_scan = false;
    int early_gen = 0;             // generation after which the early tiles may start (0 = none for this geometry)
    uint32_t early_tiles = 0;      // tiles per frame of the early scales (a prefix of the tile list)
    bool early_launched = false;   // this batch's early part was launched (ht_launch_scan then only does the rest)
    std::string err;

    // cascade
    int interval = 5, next = 6;
    uint32_t cw = 24, ch = 24, nstages = 0, nfeat = 0;
    bool decimal_alphas = false;  // all alphas / thresholds are k * 1e-8 exactly -> integer decisions allowed
    std::vector<HtBlobStage> h_stages;
    std::vector<HtBlobFeature> h_feats;
    HtTileFeature *d_tile_feats = nullptr;
    HtDeepFeature *d_deep_feats = nullptr;
    HtPatchFeature *d_patch_feats = nullptr;
    HtPackedFeature *d_packed_feats = nullptr;  // features of stages >= split_stage (index 0 = first feature of that stage)
    // every feature in the packed 32-byte form with TILE offsets (off[] relative to a window's LDS base, a1i = alpha[2k+1] * 1e8): the
    // tile kernel's feature-parallel sparse phase reads one record per lane (nullptr: cascade not decimal / more than 5 points)
    HtPackedFeature *d_fp_feats = nullptr;
    bool fp_sparse = true;  // option fp_sparse=0: the sparse stages always run as four feature slices (A/B)
    uint32_t packed_count = 0, packed_first = 0; // number of packed features / global index of the first one (0 count = unusable)
    bool builtin_cascade = false;  // blob == the cascade ht_cascade_gen.inc was generated from
    uint32_t deep_bias = 1;        // tile kernel hands survivors to the deep kernel when n*bias*ceil(count/64) <= count
                                   // (measured on C2/C4: split 8 + bias 0..1 is the optimum, profiles/r01_sweeps.txt)
    int dbg_stop_stage = -1, dbg_force_exact = 0, dbg_deep_v = 4, deep_grid = 192;  // ht_config.options, parsed once in ht_create
    HtDevStage *d_stages = nullptr;
    uint32_t split_stage = 4;  // stages [0, split) in the tile kernel, [split, nstages) in the deep kernel
    int opt_split = 0;         // option split (0 = default)

    // geometry
    int W = 0, H = 0, max_batch = 0, nlevels = 0, upto = 0;
    HtDevLevel h_levels[HT_MAX_LEVELS];
    HtDevLevel *d_levels = nullptr;
    uint64_t arena_stride = 0;  // bytes per frame
    uint64_t pyr_bytes = 0, windows_per_frame = 0;
    uint8_t *d_arena = nullptr;
    std::vector<std::vector<HtResampleJob>> h_gens;  // generation g: jobs that only depend on generations < g
    std::vector<HtResampleJob *> d_gen_blocks;  // per generation: k_resample tile records (job + tile position)
    HtTileRec *d_tile_recs = nullptr;         // per-frame tile list (same for every frame of a batch)
    std::vector<uint32_t> gen_blocks;
    std::vector<HtScanScale> h_scales;
    HtScanScale *d_scales = nullptr;
    uint32_t tiles_per_frame = 0;
    // k_resample_tail: the last generations (tiny levels) in ONE launch, one workgroup per frame
    int tail_first_gen = 0;                  // first generation handled by the tail kernel (0 = none)
    HtResampleJob *d_tail_jobs = nullptr;    // jobs of generations >= tail_first_gen, generation by generation
    uint32_t *d_tail_prefix = nullptr;       // per job: 4-pixel groups of the jobs before it in its generation
    HtTap *d_tail_taps = nullptr;            // tap tables of the tail jobs (full form) ...
    HtTapFast *d_tail_taps_fast = nullptr;   // ... and compact form
    HtTailTapRef *d_tail_tapref = nullptr;   // per tail job: where its taps start
    HtTailGens h_tail;                       // per generation: job range and group count (kernel argument)
    bool deep_attr_set = false;              // k_scan_deep_lds: > 64 KB dynamic LDS enabled on this context's device
    int tail_table = 1;  // k_resample_tail with host tap tables: 1 = compact taps in LDS, 2 = taps from L2 / small footprint; 0: the round-1 binary64 tail (option rs_tailtable)
    bool tail_table_forced = false;  // option rs_tailtable given: ht_set_geometry keeps it instead of choosing by batch size
    bool rs_nofast = false, rs_nosort = false, rs_notail = false, rs_gennames = false;  // options of the same names (A/B, cross-checks)
    uint64_t rs_tailcap = 32768;     // destination pixels per frame the tail kernel takes at most (option rs_tailcap; batches <= 48 frames: 4 000 unless the option is given)
    bool rs_tailcap_forced = false;
    int rs_maxgen = 1 << 30;         // HT_DEBUG_KNOBS builds only (results stale): pyramid generations built
    bool force_rccl = false;         // option force_rccl: ht_allgather_* runs RCCL even with one rank
    int host_threads = -1;           // option host_threads: workers of the host post-processing (-1 = auto, 0 = none)
    int rs_min_wgs = 1536;  // ... but never fewer workgroups per launch than this (option rs_minwg; round 4, three batches in flight at C2: 1024 / 1536 / 2048 / 4096 -> 0.2288 / 0.2283 / 0.2303 / 0.2365 ms per step)
    int dbg_rs_k = 0;  // option rs_k: frames per k_resample workgroup, forced (any value)
    int rs_group = 8;  // pyramid generation kernels: frames per workgroup at most (option rs_group).  Frames of >= 400 k pixels take 4 unless the option is
                       // given: with k_resample_bands at 128 x 720p, K = 8 / 4 / 3 / 2 / 1 -> resample 0.480 / 0.456 / 0.455 / 0.465 / 0.529 ms per step,
                       // wall (two in flight) 1.1114 / 1.1040 / 1.1103 / 1.1295 / 1.2585; at 256 x 320x240 8 stays best (wall 0.2184 vs 0.2218)
    bool rs_group_forced = false;
    int rs_rpt = 4;  // k_resample: destination rows per thread (tile = 64 x 16*rs_rpt)
    bool rs_bands = true;  // option rs_bands=0: the pyramid generations run k_resample (register-staged tile, two barriers per frame) instead of k_resample_bands (A/B)

    // frames
    uint8_t *d_frames_own = nullptr;
    size_t d_frames_own_bytes = 0;
    uint8_t *d_frames_back = nullptr;  // ht_upload_frames_async target; ht_swap_frames exchanges it with d_frames_own
    size_t d_frames_back_bytes = 0;
    int back_n = 0;                    // frames in the back buffer (0 = nothing pending)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copy_done = nullptr, ev_front_free = nullptr;
    const uint8_t *d_frames = nullptr;
    size_t frame_stride = 0;
    int nframes = 0;
    // small batches (a live feed = 1 frame) are launch-bound: ~10 dependent launches cost more than their kernels.  Their sequence is
    // captured into a hipGraph per (frames pointer, count, flags) and replayed.  0 disables (option graph_max_frames)
    // set (under the cluster gate's lock: ht_capture_mark) while this context's stream is between hipStreamBeginCapture and EndCapture:
    // another context's fused_threads() must not hipStreamQuery a capturing stream (the query fails and may invalidate the capture)
    std::atomic<bool> capturing{false};
    int graph_max_frames = 256;  // round 4: replaying a 256-frame C2 batch instead of launching its 9 kernels: 0.2253 -> 0.2239 ms per step at three in flight (16 until then: only launch-bound small batches)
    std::vector<HtDetectGraph> graphs;
    uint64_t graph_launches = 0;  // measurement: enqueues served by a graph replay
    int64_t requeue_flags = -1;   // >= 0: ht_detect_collect enqueues the next batch (these flags) as soon as the raw hits are on the host
    int enq_nframes = 0;  // frames of the batch enqueued last (what ht_detect_collect reports on)

    // scan outputs
    uint32_t hit_capacity = 1u << 20, queue_capacity = 0, queue_capacity_cfg = 0;
    ht_hit *d_hits = nullptr;
    HtQueueEntry *d_queue = nullptr;
    HtCounters *d_counters = nullptr;   // device block [HtCounters][hit_capacity x ht_hit]: d_hits points behind the counters
    uint32_t spec_hint = 0;             // raw hits of the batch collected last (sizes the next speculative read-back)
    HtCounters h_counters;
    uint8_t *h_pinned = nullptr;  // pinned staging: [HtCounters][HT_PINNED_HITS x ht_hit], one D2H + one sync per batch
    unsigned long long *d_stats = nullptr;      // [HT_STAT_SHARDS][64], only touched with HT_SCAN_STATS
    unsigned long long h_stage_in[64] = {0};   // windows that entered stage j ([nstages] = full survivors), last collected batch
    std::vector<ht_hit> h_collect_hits;        // ht_detect_collect_best: sorted raw hits of the last batch
    std::vector<uint32_t> h_collect_counts;    // ... and their per-frame counts
    std::vector<ht_hit> h_raw_hits;            // ht_detect_collect: the batch's raw hits in arrival order (scratch, kept between calls)
    std::vector<ht_hit> h_ordered_hits;        // ... and in emission order when the caller's buffer cannot take them directly
    std::vector<uint32_t> h_frame_start;       // ... bucket offsets of the counting sort by frame
    bool stats_enqueued = false;
    bool enqueued = false;
    bool collect_best_follows = false;  // set by ht_detect_collect_best around its ht_detect_collect: the per-frame ordering is left to its own per-frame pass
    bool h_sort_deferred = false;       // ... and this says that it was

    // whitebalance / grayscale scratch
    double *d_scratch = nullptr;
    size_t d_scratch_bytes = 0;
    bool wb_fused = false;     // set around ht_launch_pyramid: the gray kernel also accumulates the channel sums into d_scratch
    bool wb_enqueued = false;  // the batch enqueued last carried HT_DETECT_WHITEBALANCE: ht_detect_collect snapshots its sums
    // d_scratch holds two regions of 4 u64 per frame: [0, max_batch) the sums of the batch in flight (fused gray pass),
    // [max_batch, 2 max_batch) the stand-alone ht_whitebalance_batch — so neither can zero the other's sums
    unsigned long long *h_wb_pinned = nullptr;       // pinned staging of the in-flight batch's sums (copied with the counters)
    size_t h_wb_pinned_bytes = 0;
    std::vector<unsigned long long> h_wb_sums;       // channel sums of the batch COLLECTED last (what ht_detect_whitebalance reports)
    int wb_collected_n = -1;                         // frames in h_wb_sums; -1: the collected batch carried no HT_DETECT_WHITEBALANCE

    // camshift
    int cs_streams = 0;
    HtCsState *d_cs = nullptr;
    uint32_t *d_cs_hist = nullptr;  // per-stream current-frame histogram (4096 bins)
    ht_cs_trackobj *d_cs_out = nullptr;
    double *d_cs_lut = nullptr, *d_cs_parts = nullptr;  // cluster mean-shift: per-stream weight LUT, partial-sum exchange slots
    bool cs_cluster = true;                              // option cs_cluster=0 disables the cluster path
    uint32_t cs_cluster_min_px = 10000;                  // frames at least this large take it (option cs_cluster_min_px).  Round 6, rocprofv3 kernel trace of
                                                         // track() calls in turn (tools/gpu_cs_one_stream_trace.sh), device us per call, cluster / one workgroup per
                                                         // stream: 320x240 x 1: 19.1 / 19.1, x 8: 20.0 / 20.6; 480x360 x 1: 18.6 / 24.7, x 16: 20.5 / 29.8; 640x480 x 1:
                                                         // 20.8 / 31.2, x 8: 20.5 / 35.1; 1280x720 x 8: 22.4 / 40.9 — and its calls are completed by marks in the pinned
                                                         // slot instead of an event: wall per synchronous call at 320x240 21.5 / 30.2 us (tools/gpu_cs_wall.py).  The
                                                         // threshold was 400 k pixels until then: VGA and the reference's own 320x240 canvas took the slower path
    ht_cs_trackobj *d_cs_seq_out = nullptr;  // ht_camshift_track_sequence: [calls][streams] results, one D2H at the end
    size_t cs_seq_cap = 0;
    // the sequence enqueued with out == NULL that ht_camshift_sequence_collect may fetch (n == 0: none pending)
    int cs_seq_pending_n = 0, cs_seq_pending_calls = 0, cs_seq_pending_all = 0;
    // enqueue-only track calls (ht_camshift_track_batch with out == NULL): the kernels write their track objects straight into a pinned
    // host slot of this ring, an event marks the slot complete; ht_camshift_track_collect takes the oldest.  Up to HT_CS_RING calls may be
    // outstanding, so a streaming host enqueues step i + 1 before it waits for step i.
    static constexpr int HT_CS_RING = 4;
    struct HtCsSlot {
        ht_cs_trackobj *h_out = nullptr;  // pinned, cs_ring_streams objects
        hipEvent_t ev = nullptr;
        int n = 0;
        // completion without an event (option cs_flags, cluster path): the kernel stores `seq` into h_flag[stream] (pinned, system scope,
        // after the stream's track object); the collect call polls the n words.  An event record is a barrier packet of its own on the
        // stream — two of them per step (this one and the cluster gate's) were the 10 us between a step's last and the next step's
        // first kernel (rocprofv3 kernel trace, LABLOG.md round 5).
        uint32_t *h_flag = nullptr;
        uint32_t seq = 0;  // 0: this slot's call is marked by the event
    };
    ht_cs_rect *h_cs_rects = nullptr;  // pinned staging of ht_camshift_init_batch's rects
    int h_cs_rects_cap = 0;
    hipEvent_t ev_cs_rects = nullptr;  // its copy to the device has been issued and completed
    uint32_t cs_flag_seq = 0;
    bool cs_flags = true;
    bool cs_sync_ring = true;  // option cs_sync_ring=0: a synchronous track call copies its results back and synchronises the stream (A/B)
    HtCsSlot cs_ring[HT_CS_RING];
    int cs_ring_head = 0, cs_ring_count = 0, cs_ring_streams = 0;
    uint32_t *h_cs_err_direct = nullptr;  // pinned word the cluster kernel itself sets when a barrier times out (read with a ring slot: no copy)
    uint32_t *d_cs_err = nullptr;      // device word: a cluster barrier ran out of its cycle budget (k_cs_meanshift_cluster)
    uint32_t *h_cs_err = nullptr;      // pinned copy, fetched with every result read-back
    long long cs_barrier_budget = 1ll << 28;  // shader-clock cycles a workgroup waits at one cluster barrier (option cs_barrier_budget)
    int num_cus = 256;                 // hipDeviceProp_t::multiProcessorCount: sizes the cluster of k_cs_meanshift_cluster
    uint32_t cs_fused_launches[2] = {0, 0};  // k_cs_track_fused launches in the 1024- / 512-thread form since the last ht_kernel_times(reset): reported there as
                                            // the pseudo-timers cs_fused_launches_1024 / _512 (ms = 0), profiling on or off
    int cs_fused_nt = 0;             // option cs_fused_nt=512|1024: threads per workgroup of k_cs_track_fused (0: chosen per launch, ht_camshift.hip fused_threads)
    int cs_fused_min_streams = 192;  // >= this many streams per call: k_cs_track_fused (option cs_fused_min)
    bool cs_seq_attr_set = false;
    bool cs_seq_fused = true;      // option cs_seq_fused=0: ht_camshift_track_sequence launches one kernel per call (A/B)
    int dbg_cs_iters = 10;            // option cs_iters: mean-shift iterations at most (camshift.js:284 has 10; anything else = wrong results)
    bool cs_keep_hist = false;
    bool cs_attr_set = false;         // > 64 KB dynamic LDS enabled for the camshift kernels on this context's device
    int cs_region_cap = 40960;        // pixels of the LDS-cached search region (option cs_region=0 disables it)        // option cs_keep_hist: the fused kernel also writes its histogram for ht_camshift_debug_hist
    int cs_last_first = 0, cs_last_n = 0, cs_last_chunks = 0;  // layout of d_cs_hist after the last track call (debug read-back)
    const uint32_t *cs_last_hist = nullptr;                    // ... and which half of d_cs_hist it used

    std::vector<std::pair<void *, size_t>> user_allocs;  // ht_device_alloc buffers still alive (pointer, bytes): freed by ht_destroy at the latest

    // multi-GPU exchange buffer (ht_allgather_best_faces)
    void *d_gather = nullptr;
    size_t d_gather_bytes = 0;

    // profiling
    bool profiling = false;
    std::vector<HtKernelTimer> timers;
};

// roctx ranges around the host-side phases of a batch: libroctx64 is looked up once with dlopen; without it (or without a
// profiler attached) a range costs one predictable branch.  rocprofv3 --marker-trace shows "ht_detect_enqueue", "ht_detect_collect", ...
struct HtRange {
    explicit HtRange(const char *name);
    ~HtRange();
    bool on;
};

// error helpers -------------------------------------------------------------------------------------------
ht_status ht_fail(ht_ctx *ctx, ht_status st, const std::string &msg);
#define HT_HIP(ctx, call)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return ht_fail((ctx), HT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));           \
    } while (0)

// profiling scope: records a start/stop event pair around kernel launches when ctx->profiling
struct HtProfScope {
    ht_ctx *ctx;
    hipStream_t stream;
    int idx = -1;
    hipEvent_t a = nullptr, b = nullptr;
    HtProfScope(ht_ctx *c, const char *name, hipStream_t on = nullptr);  // on == nullptr: the context's main stream
    ~HtProfScope();
};

// implemented in the .hip files ---------------------------------------------------------------------------
void ht_cluster_gate_forget(const ht_ctx *ctx);             // ht_camshift.hip
void ht_capture_mark(ht_ctx *ctx, bool on);                 // ht_camshift.hip: ctx->capturing, ordered against fused_threads' stream queries
ht_status ht_launch_pyramid(ht_ctx *ctx, uint32_t flags);   // ht_pyramid.hip
ht_status ht_launch_scan(ht_ctx *ctx, uint32_t flags);      // ht_scan.hip
ht_status ht_launch_scan_early(ht_ctx *ctx, uint32_t flags); // ht_scan.hip: called by ht_launch_pyramid after generation early_gen
ht_status ht_scan_tile_tables(ht_ctx *ctx);                 // ht_scan.hip: LDS-offset feature table
ht_status ht_scan_plan_tiles(ht_ctx *ctx);
bool ht_scan_is_builtin_cascade(const uint8_t *blob, size_t len);  // ht_scan.hip
ht_status ht_scan_pack_deep(ht_ctx *ctx);                           // ht_scan.hip: LDS-resident table for stages >= split_stage                  // ht_scan.hip: per-scale tiling for the geometry
ht_status ht_launch_gray_inplace(ht_ctx *ctx, uint8_t *d_rgba, int n, size_t stride);  // ht_pyramid.hip
ht_status ht_launch_whitebalance(ht_ctx *ctx, double *d_out, bool zero);                 // ht_pyramid.hip
