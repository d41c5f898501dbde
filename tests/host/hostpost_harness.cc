// Host-only harness around headtrackr_amd/csrc/ht_hostpost.h — the SAME code libheadtrackr_hip.so runs after every detect batch
// (counting sort by frame, per-frame ordering, seq rects, ccv's grouping, facetrackr's best face, the worker pool) — built by
// tests/test_host_post.py with  g++ -fsanitize=address,undefined -fno-sanitize-recover=all  and driven with seeded raw-hit sets:
//     hostpost_harness <in.bin> <out.bin> <nworkers[,nworkers...]> [repeat]
// (a comma list of worker counts is cycled over the repeats: the pool's participant count changes from batch to batch inside one
// process — the case ThreadSanitizer is run on, tests/test_host_post.py)
// in.bin : u32 nfr, u32 found, i32 min_neighbors, i32 interval, then found x ht_hit (24 B, arrival order: frames interleaved)
// out.bin: u32 ok, found x ht_hit (emission order), nfr x u32 counts, nfr x ht_rect (best face per frame)
// A frame index outside the batch must be reported (ok = 0), not written through.  TEST INFRASTRUCTURE: not part of the product.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ht_hostpost.h"

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    uint32_t hdr[4];
    if (std::fread(hdr, 4, 4, f) != 4) return 2;
    const uint32_t nfr = hdr[0], found = hdr[1];
    const int32_t min_neighbors = (int32_t)hdr[2];
    const HtPostCfg cfg = {(int)hdr[3], 24u, 24u};
    std::vector<ht_hit> raw(found);
    if (found && std::fread(raw.data(), sizeof(ht_hit), found, f) != found) return 2;
    std::fclose(f);
    std::vector<int> workers;
    for (const char *p = argv[3]; *p;) {
        workers.push_back(std::atoi(p));
        while (*p && *p != ',') p++;
        if (*p == ',') p++;
    }
    if (workers.empty()) return 2;
    const int repeat = argc > 4 ? std::atoi(argv[4]) : 1;
    std::vector<ht_hit> dst(found);  // exactly `found` entries: an off-by-one in the bucket offsets is a heap overflow ASan reports
    std::vector<uint32_t> end, counts(nfr);
    std::vector<ht_rect> best(nfr);
    uint32_t ok = 1;
    for (int r = 0; r < repeat && ok; r++) {  // several batches through the same pool: the hand-off is exercised more than once
        ok = ht_post_bucket_by_frame(raw.data(), found, nfr, dst.data(), end, counts.data()) ? 1u : 0u;
        const int nworkers = workers[(size_t)r % workers.size()];
        if (ok && ht_post_frames(cfg, dst.data(), end.data(), (int)nfr, min_neighbors, nworkers, best.data()) != HT_OK) ok = 0;
    }
    f = std::fopen(argv[2], "wb");
    if (!f) return 2;
    std::fwrite(&ok, 4, 1, f);
    if (ok) {
        if (found) std::fwrite(dst.data(), sizeof(ht_hit), found, f);
        if (nfr) std::fwrite(counts.data(), 4, nfr, f), std::fwrite(best.data(), sizeof(ht_rect), nfr, f);
    }
    std::fclose(f);
    return 0;
}
