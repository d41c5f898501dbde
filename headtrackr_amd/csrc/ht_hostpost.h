// ht_hostpost.h — the host side of a detect batch's post-processing: raw hits -> emission order -> seq rects -> ccv's grouping ->
// facetrackr's best face per frame, and the worker pool that deals a batch's frames out.  Plain C++ (no HIP): libheadtrackr_hip.so
// compiles it as part of ht_context.hip, and tests/host/hostpost_harness.cc compiles the SAME file with g++ -fsanitize=address,undefined
// so that the pointer arithmetic below runs under the sanitizers in the CPU test suite.
//
// Reference behaviour restated here (paths under /root/reference/src/):
//   hits -> seq rects   ccv.js:227-234,244-245
//   grouping            ccv.js:34-107 (array_group), 249-332 (averaging, nested-rect filter)
//   best face           facetrackr.js:147-175
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "headtrackr_hip.h"

struct HtPostCfg {
    int interval;      // ccv.detect_objects `interval`
    uint32_t cw, ch;   // cascade window
};

// V8's Math.pow(Math.pow(2, 1/6), i) for i = 0..5: glibc's pow() is one ulp off for i = 4, and Math.floor(W / pow)
// (ccv.js:119-120) must see the same divisor the JavaScript reference saw.
static const uint64_t kV8Scale6Pow[6] = {0x3ff0000000000000ULL, 0x3ff1f59ac3c7d6c0ULL, 0x3ff428a2f98d728bULL,
                                         0x3ff6a09e667f3bcdULL, 0x3ff965fea53d6e3eULL, 0x3ffc823e074ec12bULL};
static inline double bits2d(uint64_t u) {
    double d;
    std::memcpy(&d, &u, 8);
    return d;
}
static inline double ht_scale_of(int interval) { return interval == 5 ? bits2d(kV8Scale6Pow[1]) : std::pow(2.0, 1.0 / (interval + 1)); }
static inline double ht_scale_pow(int interval, int i) {
    return (interval == 5 && i >= 0 && i <= 5) ? bits2d(kV8Scale6Pow[i]) : std::pow(ht_scale_of(interval), (double)i);
}

inline void ht_post_level_scales(const HtPostCfg &cfg, double *sx) {
    const double scale = ht_scale_of(cfg.interval);  // ccv.js:110
    sx[0] = 1;                                       // ccv.js:150
    for (int i = 1; i < HT_MAX_LEVELS; i++) sx[i] = sx[i - 1] * scale;  // ccv.js:244-245 (repeated multiplication)
}

inline ht_status ht_post_hits_to_rects(const HtPostCfg &cfg, const double *sx, const ht_hit *hits, uint32_t n, ht_rect *out) {
    for (uint32_t k = 0; k < n; k++) {
        const ht_hit &h = hits[k];
        if (h.scale >= HT_MAX_LEVELS) return HT_ERR_INVALID;
        const double s = sx[h.scale];
        out[k].x = (double)(h.x * 4 + (h.q & 1) * 2) * s;   // ccv.js:228
        out[k].y = (double)(h.y * 4 + (h.q >> 1) * 2) * s;  // ccv.js:229
        out[k].width = (double)cfg.cw * s;                  // ccv.js:230
        out[k].height = (double)cfg.ch * s;                 // ccv.js:231
        out[k].confidence = h.sum;                          // ccv.js:233
        out[k].neighbors = 1;                               // ccv.js:232
        out[k].reserved = 0;
    }
    return HT_OK;
}

namespace {
struct Node {
    int parent, rank;
};
}  // namespace


inline ht_status ht_post_group_rects(const ht_rect *seq, uint32_t n, int32_t min_neighbors, ht_rect *out, uint32_t *nout) {
    if (!nout || (n && (!seq || !out))) return HT_ERR_INVALID;
    *nout = 0;
    if (n == 0) return HT_OK;
    // union-find with rank and path compression, visiting pairs in the reference's order (ccv.js:41-89).  The pair test
    // (ccv.js:252-261) only needs per-rectangle values: the three floor() terms and the x / y intervals are formed once per
    // rectangle, not once per ordered pair — the same doubles, compared the same way.  Scratch vectors live per thread: a batch
    // calls this once per frame with hits.
    thread_local std::vector<Node> node;
    thread_local std::vector<double> w15;
    node.assign(n, Node{-1, 0});
    w15.resize(n);
    for (uint32_t i = 0; i < n; i++) w15[i] = std::floor(seq[i].width * 1.5 + 0.5);
    auto find_root = [&](int i) {
        while (node[i].parent != -1) i = node[i].parent;
        return i;
    };
    auto compress = [&](int i, int root) {
        while (node[i].parent != -1) {
            const int t = i;
            i = node[i].parent;
            node[t].parent = root;
        }
    };
    for (uint32_t i = 0; i < n; i++) {
        int root = find_root((int)i);
        const ht_rect &r1 = seq[i];
        const double distance = std::floor(r1.width * 0.25 + 0.5);
        const double xh = r1.x + distance, xl = r1.x - distance, yh = r1.y + distance, yl = r1.y - distance, w1 = r1.width, w15i = w15[i];
        for (uint32_t j = 0; j < n; j++) {
            const ht_rect &r2 = seq[j];
            if (!(r2.x <= xh && r2.x >= xl && r2.y <= yh && r2.y >= yl && r2.width <= w15i && w15[j] >= w1) || i == j) continue;
            const int root2 = find_root((int)j);
            if (root2 == root) continue;
            if (node[root].rank > node[root2].rank) {
                node[root2].parent = root;
            } else {
                node[root].parent = root2;
                if (node[root].rank == node[root2].rank) node[root2].rank++;
                root = root2;
            }
            compress((int)j, root);
            compress((int)i, root);
        }
    }
    // class ids in first-seen order (ccv.js:90-105)
    thread_local std::vector<int> idx;
    thread_local std::vector<ht_rect> comps, seq2;
    idx.resize(n);
    int ncomp = 0;
    for (uint32_t i = 0; i < n; i++) {
        const int r = find_root((int)i);
        if (node[r].rank >= 0) node[r].rank = ~ncomp++;
        idx[i] = ~node[r].rank;
    }
    comps.assign((size_t)ncomp, ht_rect{0, 0, 0, 0, 0, 0, 0});
    for (uint32_t i = 0; i < n; i++) {  // ccv.js:274-289
        ht_rect &cp = comps[idx[i]];
        if (cp.neighbors == 0) cp.confidence = seq[i].confidence;
        ++cp.neighbors;
        cp.x += seq[i].x;
        cp.y += seq[i].y;
        cp.width += seq[i].width;
        cp.height += seq[i].height;
        cp.confidence = std::max(cp.confidence, seq[i].confidence);
    }
    seq2.clear();
    for (int i = 0; i < ncomp; i++) {  // ccv.js:293-303
        const int nn = comps[i].neighbors;
        if (nn >= min_neighbors) {
            ht_rect r;
            r.x = (comps[i].x * 2 + nn) / (2 * nn);
            r.y = (comps[i].y * 2 + nn) / (2 * nn);
            r.width = (comps[i].width * 2 + nn) / (2 * nn);
            r.height = (comps[i].height * 2 + nn) / (2 * nn);
            r.neighbors = nn;
            r.reserved = 0;
            r.confidence = comps[i].confidence;
            seq2.push_back(r);
        }
    }
    uint32_t k = 0;
    for (size_t i = 0; i < seq2.size(); i++) {  // ccv.js:307-330
        const ht_rect &r1 = seq2[i];
        bool keep = true;
        for (size_t j = 0; j < seq2.size() && keep; j++) {
            const ht_rect &r2 = seq2[j];
            const double distance = std::floor(r2.width * 0.25 + 0.5);
            if (i != j && r1.x >= r2.x - distance && r1.y >= r2.y - distance && r1.x + r1.width <= r2.x + r2.width + distance &&
                r1.y + r1.height <= r2.y + r2.height + distance && (r2.neighbors > std::max(3, r1.neighbors) || r1.neighbors < 3))
                keep = false;
        }
        if (keep) out[k++] = r1;
    }
    *nout = k;
    return HT_OK;
}

// one frame of facetrackr.Tracker.doVJDetection's selection (facetrackr.js:147-175): seq rects, ccv's grouping, strict-'>' arg-max
inline ht_status ht_post_best_face(const HtPostCfg &cfg, const double *sx, const ht_hit *hits, uint32_t n, int32_t min_neighbors, ht_rect *best) {
    thread_local std::vector<ht_rect> seq, grouped;
    ht_rect r = {0, 0, 0, 0, -10000.0, 0, 0};  // facetrackr.TrackObj defaults, facetrackr.js:233-241
    if (n) {
        if (!hits) return HT_ERR_INVALID;
        seq.resize(n);
        grouped.resize(n);
        ht_status st = ht_post_hits_to_rects(cfg, sx, hits, n, seq.data());
        if (st != HT_OK) return st;
        uint32_t ng = n;
        if (min_neighbors > 0) {
            if ((st = ht_post_group_rects(seq.data(), n, min_neighbors, grouped.data(), &ng)) != HT_OK) return st;
        } else {
            grouped = seq;
        }
        for (uint32_t i = 0; i < ng; i++)  // facetrackr.js:157-165
            if (i == 0 || grouped[i].confidence > r.confidence) r = grouped[i];
    }
    *best = r;
    return HT_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Host worker pool for the per-frame post-processing of a batch (hit ordering, seq rects, ccv's grouping, facetrackr's selection:
// ccv.js:249-332, facetrackr.js:157-165).  Frames are independent and every frame writes its own output slots, so the results do not
// depend on how the frames are dealt out.  One pool per process, created on first use, shared by all contexts (a caller holds it for the
// duration of one batch); workers spin briefly for the next batch — a batch server collects one every ~0.25 ms — and then sleep.
class HtPool {
public:
    static HtPool &get() {
        static HtPool *p = new HtPool();  // never destroyed: worker threads must not be joined from a static destructor at exit
        return *p;
    }
    // runs fn(begin, end) over [0, n) in chunks of `chunk` on up to `nthreads` workers + the calling thread; returns when all are done.
    // The worker count may differ from batch to batch.  A generation and its participant count travel in ONE atomic word (`state_` =
    // generation << 8 | participants): a worker decides from the very value that announced the batch whether it takes part, so a worker
    // outside batch g can neither be counted in nor touch fn_ / n_ / chunk_ of batch g + 1 by accident, and every participant of batch g
    // has acknowledged (pending_) before run() returns — nothing of a batch is read after its run() has returned.
    void run(int n, int chunk, int nthreads, const std::function<void(int, int)> &fn) {
        if (n <= 0) return;
        nthreads = std::min(std::min(nthreads, kMaxWorkers), (n + chunk - 1) / chunk - 1);
        if (nthreads <= 0) {
            fn(0, n);
            return;
        }
        std::lock_guard<std::mutex> user(user_mu_);  // one batch at a time
        ensure(nthreads);
        fn_ = &fn, n_ = n, chunk_ = chunk;  // published by the release store of state_ below, read by participants only
        next_.store(0, std::memory_order_relaxed);
        pending_.store(nthreads, std::memory_order_relaxed);
        const uint64_t gen = (state_.load(std::memory_order_relaxed) >> 8) + 1;
        {
            std::lock_guard<std::mutex> lk(mu_);
            state_.store(gen << 8 | (uint64_t)nthreads, std::memory_order_release);
        }
        cv_.notify_all();
        work(&fn, n, chunk);
        while (pending_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    }
    static constexpr int kMaxWorkers = 64;

private:
    void work(const std::function<void(int, int)> *fn, int n, int chunk) {
        for (;;) {
            const int b = next_.fetch_add(chunk, std::memory_order_relaxed);
            if (b >= n) break;
            (*fn)(b, std::min(n, b + chunk));
        }
    }
    void ensure(int nthreads) {
        while ((int)th_.size() < nthreads) {
            const int id = (int)th_.size();
            th_.emplace_back([this, id] { loop(id); });
            th_.back().detach();
        }
    }
    static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        asm volatile("yield" ::: "memory");
#else
        std::this_thread::yield();
#endif
    }
    void loop(int id) {
        uint64_t seen = 0;  // generation this worker has dealt with
        for (;;) {
            // poll for ~50 us (a batch server collects a batch every ~0.25 ms), by the clock, then sleep on the condition variable
            uint64_t st = state_.load(std::memory_order_acquire);
            if ((st >> 8) == seen) {
                const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(50);
                for (int spin = 0; (st >> 8) == seen; spin++) {
                    cpu_relax();
                    st = state_.load(std::memory_order_acquire);
                    if ((spin & 63) == 63 && std::chrono::steady_clock::now() >= t_end) break;
                }
            }
            if ((st >> 8) == seen) {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return (state_.load(std::memory_order_acquire) >> 8) != seen; });
                st = state_.load(std::memory_order_acquire);
            }
            seen = st >> 8;
            if (id < (int)(st & 0xff)) {  // participant of exactly this generation: run() waits for the acknowledgement below
                work(fn_, n_, chunk_);
                pending_.fetch_sub(1, std::memory_order_release);
            }
        }
    }
    std::mutex user_mu_, mu_;
    std::condition_variable cv_;
    std::vector<std::thread> th_;
    std::atomic<uint64_t> state_{0};  // generation << 8 | participating workers of that generation
    std::atomic<int> next_{0}, pending_{0};
    const std::function<void(int, int)> *fn_ = nullptr;
    int n_ = 0, chunk_ = 1;
};

// workers for a batch of `frames` frames holding `hits` raw hits: option host_threads, or (auto) up to 7 when the batch is worth it

// emission order inside one frame (ccv.js:154,178,181-182): scale, q, y, x as one packed key
inline void ht_post_sort_frame(ht_hit *hits, uint32_t b, uint32_t e) {
    auto key = [](const ht_hit &h) { return ((uint64_t)h.scale << 40) | ((uint64_t)h.q << 32) | ((uint64_t)h.y << 16) | (uint64_t)h.x; };
    if (e - b > 1) std::sort(hits + b, hits + e, [&](const ht_hit &p, const ht_hit &q) { return key(p) < key(q); });
}

// Counting sort of a batch's raw hits (arrival order) by frame, straight into dst.  On return end[f] is the END of frame f's hits in dst
// (its begin is end[f - 1], 0 for f == 0); counts (optional) gets the per-frame counts.  false: a frame index outside [0, nfr) — never
// produced by the kernels — nothing usable was written.
inline bool ht_post_bucket_by_frame(const ht_hit *raw, uint32_t found, uint32_t nfr, ht_hit *dst, std::vector<uint32_t> &end, uint32_t *counts) {
    end.assign((size_t)nfr + 1, 0u);
    for (uint32_t i = 0; i < found; i++) {
        if (raw[i].frame >= nfr) return false;
        end[raw[i].frame + 1]++;
    }
    if (counts) std::memcpy(counts, end.data() + 1, sizeof(uint32_t) * (size_t)nfr);
    for (uint32_t f = 0; f < nfr; f++) end[f + 1] += end[f];
    for (uint32_t i = 0; i < found; i++) dst[end[raw[i].frame]++] = raw[i];  // end[f] finishes as the END of frame f
    return true;
}

// The per-frame pass of a batch: order frame f's hits, then its best face — frames dealt out to `nworkers` pool threads + the caller.
// Frames are independent and write their own slots: the result is byte-identical for every nworkers.
inline ht_status ht_post_frames(const HtPostCfg &cfg, ht_hit *hits, const uint32_t *end, int nfr, int32_t min_neighbors, int nworkers, ht_rect *best) {
    double sx[HT_MAX_LEVELS];
    ht_post_level_scales(cfg, sx);
    std::atomic<int> err{HT_OK};
    auto frames = [&](int f0, int f1) {
        for (int f = f0; f < f1; f++) {
            const uint32_t b = f ? end[f - 1] : 0u, e = end[f];
            ht_post_sort_frame(hits, b, e);
            const ht_status s1 = ht_post_best_face(cfg, sx, hits + b, e - b, min_neighbors, &best[f]);
            if (s1 != HT_OK) err.store(s1);
        }
    };
    if (nworkers > 0) HtPool::get().run(nfr, 8, nworkers, frames);
    else frames(0, nfr);
    return (ht_status)err.load();
}
