/*
 * ht_oracle.c — TEST INFRASTRUCTURE.  Plain-C, single-threaded CPU restatement of headtrackr's per-frame
 * detect/track hot path, used only as the checker (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
 * leg).  Nothing in the product path (headtrackr_amd/) links, imports or calls this file.
 *
 * Pinning: every function below is checked against golden vectors produced by executing the UNMODIFIED
 * reference JS (oracle/ref_harness.js on /root/reference/headtrackr.js) — tests/test_oracle_golden.py.
 * The one boundary the reference itself does not pin is the browser's canvas drawImage resampler
 * ("parity unpinned" there, see oracle/canvas_shim.js); ho_resample() restates the resampler DECLARED in
 * that shim, bit for bit.
 *
 * All line citations are into /root/reference/src/.
 * Build: see oracle/Makefile (-O2 -ffp-contract=off: every double op must round exactly like JS).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HO_MAXPTS 8
#define HO_MAX_LEVELS 96 /* scale_upto + 2*(interval+1), generous */

/* ------------------------------------------------------------------------------------------------ */
/* cascade blob ("HTCB", headtrackr_amd/js/cascade_pack.js) == headtrackr.cascade, cascade.js:19      */

typedef struct {
    uint8_t size, pad[7];
    int8_t px[HO_MAXPTS], py[HO_MAXPTS], pz[HO_MAXPTS];
    int8_t nx[HO_MAXPTS], ny[HO_MAXPTS], nz[HO_MAXPTS];
    double alpha[2];
} ho_feature;

typedef struct {
    uint32_t count, first;
    double threshold;
} ho_stage;

typedef struct {
    uint32_t nstages, width, height, nfeat;
    const ho_stage *stages;
    const ho_feature *features;
} ho_cascade;

static int ho_parse_cascade(const uint8_t *blob, size_t len, ho_cascade *c) {
    if (len < 32 || memcmp(blob, "HTCB", 4) != 0) return -1;
    const uint32_t *h = (const uint32_t *)blob;
    if (h[1] != 1 || h[6] != HO_MAXPTS) return -1;
    c->nstages = h[2];
    c->width = h[3];
    c->height = h[4];
    c->nfeat = h[5];
    if (len != 32 + (size_t)c->nstages * sizeof(ho_stage) + (size_t)c->nfeat * sizeof(ho_feature)) return -1;
    c->stages = (const ho_stage *)(blob + 32);
    c->features = (const ho_feature *)(blob + 32 + (size_t)c->nstages * sizeof(ho_stage));
    return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* ccv.grayscale, ccv.js:22-32: data[R]=data[G]=data[B] = R*0.3 + G*0.59 + B*0.11 (binary64, left to      */
/* right), stored through Uint8ClampedArray = clamp + round-half-to-even.  Alpha untouched.            */

static inline uint8_t ho_clamp_u8(double v) {
    if (!(v > 0.0)) return 0; /* also NaN -> 0 */
    if (v > 255.0) return 255;
    return (uint8_t)nearbyint(v); /* default rounding mode: ties to even */
}

void ho_grayscale_rgba(uint8_t *rgba, int w, int h) {
    size_t n = (size_t)w * (size_t)h;
    for (size_t i = 0; i < n; i++) {
        uint8_t *p = rgba + 4 * i;
        uint8_t g = ho_clamp_u8((double)p[0] * 0.3 + (double)p[1] * 0.59 + (double)p[2] * 0.11);
        p[0] = p[1] = p[2] = g;
    }
}

/* gray plane (1 byte/pixel) from RGBA — same arithmetic, planar output */
void ho_gray_plane(const uint8_t *rgba, int w, int h, uint8_t *plane) {
    size_t n = (size_t)w * (size_t)h;
    for (size_t i = 0; i < n; i++) {
        const uint8_t *p = rgba + 4 * i;
        plane[i] = ho_clamp_u8((double)p[0] * 0.3 + (double)p[1] * 0.59 + (double)p[2] * 0.11);
    }
}

/* headtrackr.getWhitebalance, whitebalance.js:5-30 */
double ho_whitebalance(const uint8_t *rgba, int w, int h) {
    double r = 0, g = 0, b = 0;
    size_t n = (size_t)w * (size_t)h;
    for (size_t i = 0; i < n; i++) {
        r += rgba[4 * i];
        g += rgba[4 * i + 1];
        b += rgba[4 * i + 2];
    }
    double avgr = r / (double)n, avgg = g / (double)n, avgb = b / (double)n;
    return (avgr + avgg + avgb) / 3;
}

/* ------------------------------------------------------------------------------------------------ */
/* The declared drawImage resampler (oracle/canvas_shim.js), on one channel.                           */
/* src rect (sx,sy,sw,sh) of a plane with row stride sstride -> dst rect (0,0,dw,dh), stride dstride. */

void ho_resample(const uint8_t *src, int sstride, int sx, int sy, int sw, int sh, uint8_t *dst, int dstride, int dw,
                 int dh) {
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return;
    const double rx = (double)sw / (double)dw, ry = (double)sh / (double)dh;
    for (int j = 0; j < dh; j++) {
        double fy = ((double)j + 0.5) * ry - 0.5;
        if (fy < 0) fy = 0;
        if (fy > (double)(sh - 1)) fy = (double)(sh - 1);
        const double y0f = floor(fy);
        const int y0 = (int)y0f, y1 = (y0 + 1 < sh - 1) ? y0 + 1 : sh - 1;
        const double ty = fy - y0f, uy = 1.0 - ty;
        const uint8_t *r0 = src + (size_t)(sy + y0) * sstride + sx, *r1 = src + (size_t)(sy + y1) * sstride + sx;
        for (int i = 0; i < dw; i++) {
            double fx = ((double)i + 0.5) * rx - 0.5;
            if (fx < 0) fx = 0;
            if (fx > (double)(sw - 1)) fx = (double)(sw - 1);
            const double x0f = floor(fx);
            const int x0 = (int)x0f, x1 = (x0 + 1 < sw - 1) ? x0 + 1 : sw - 1;
            const double tx = fx - x0f, ux = 1.0 - tx;
            const double top = (double)r0[x0] * ux + (double)r0[x1] * tx;
            const double bot = (double)r1[x0] * ux + (double)r1[x1] * tx;
            dst[(size_t)j * dstride + i] = ho_clamp_u8(top * uy + bot * ty);
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* Pyramid geometry, ccv.js:110-147.                                                                 */

typedef struct {
    int32_t w, h;       /* canvas size of this level (all 4 slots share it) */
    int64_t off[4];     /* byte offset of slot 0..3 in the arena; -1 = slot absent */
} ho_level;

/* V8's Math.pow(Math.pow(2,1/6), i), i = 0..5 (bit patterns recorded in the golden JSON files, key "scale6_pows");
 * glibc's pow() differs from V8 in the last bit for i = 4, so the default interval uses the table. */
static const uint64_t HO_V8_SCALE6_POW[6] = {0x3ff0000000000000ULL, 0x3ff1f59ac3c7d6c0ULL, 0x3ff428a2f98d728bULL,
                                             0x3ff6a09e667f3bcdULL, 0x3ff965fea53d6e3eULL, 0x3ffc823e074ec12bULL};

static double ho_u64_as_double(uint64_t u) {
    double d;
    memcpy(&d, &u, 8);
    return d;
}

double ho_scale(int interval) { /* ccv.js:110 */
    if (interval == 5) return ho_u64_as_double(HO_V8_SCALE6_POW[1]);
    return pow(2.0, 1.0 / (double)(interval + 1));
}

static double ho_scale_pow(int interval, int i) { /* Math.pow(scale, i), ccv.js:119-120 */
    if (interval == 5 && i >= 0 && i <= 5) return ho_u64_as_double(HO_V8_SCALE6_POW[i]);
    return pow(ho_scale(interval), (double)i);
}

int ho_scale_upto(int cw, int ch, int interval) { /* ccv.js:112 */
    int m = cw < ch ? cw : ch;
    return (int)floor(log((double)m) / log(ho_scale(interval)));
}

/* Fills levels[0..n) (sizes + arena offsets, rows packed with stride == w); returns n, *arena_bytes = total. */
int ho_pyramid_layout(int w, int h, int interval, int cw, int ch, ho_level *levels, int64_t *arena_bytes) {
    const int next = interval + 1;
    const int upto = ho_scale_upto(cw, ch, interval);
    const int n = upto + next * 2;
    if (n > HO_MAX_LEVELS) return -1;
    int64_t off = 0;
    for (int i = 0; i < n; i++) {
        if (i == 0) {
            levels[i].w = w;
            levels[i].h = h;
        } else if (i <= interval) { /* ccv.js:119-120 */
            levels[i].w = (int)floor((double)w / ho_scale_pow(interval, i));
            levels[i].h = (int)floor((double)h / ho_scale_pow(interval, i));
        } else { /* ccv.js:126-127 */
            levels[i].w = levels[i - next].w / 2;
            levels[i].h = levels[i - next].h / 2;
        }
        for (int s = 0; s < 4; s++) {
            if (s == 0 || i >= next * 2) { /* variants exist for i >= 2*next, ccv.js:131 */
                levels[i].off[s] = off;
                off += (int64_t)levels[i].w * levels[i].h;
            } else {
                levels[i].off[s] = -1;
            }
        }
    }
    *arena_bytes = off;
    return n;
}

/* Builds every plane.  arena[levels[0].off[0]] must already hold the w*h gray plane (= byte 0 of the canvas
 * handed to detect_objects).  Unwritten bytes of the variant planes are 0 (transparent black, ccv.js:135-145:
 * the destination rect is 2 px narrower / shorter than the canvas). */
void ho_pyramid_build(uint8_t *arena, const ho_level *L, int n, int interval) {
    const int next = interval + 1;
    for (int i = 1; i <= interval && i < n; i++) /* ccv.js:117-123 */
        ho_resample(arena + L[0].off[0], L[0].w, 0, 0, L[0].w, L[0].h, arena + L[i].off[0], L[i].w, L[i].w, L[i].h);
    for (int i = next; i < n; i++) { /* ccv.js:124-130 */
        const ho_level *s = &L[i - next];
        ho_resample(arena + s->off[0], s->w, 0, 0, s->w, s->h, arena + L[i].off[0], L[i].w, L[i].w, L[i].h);
    }
    for (int i = next * 2; i < n; i++) { /* ccv.js:131-147 */
        const ho_level *s = &L[i - next];
        const uint8_t *sp = arena + s->off[0];
        const size_t bytes = (size_t)L[i].w * L[i].h;
        memset(arena + L[i].off[1], 0, bytes);
        memset(arena + L[i].off[2], 0, bytes);
        memset(arena + L[i].off[3], 0, bytes);
        ho_resample(sp, s->w, 1, 0, s->w - 1, s->h, arena + L[i].off[1], L[i].w, L[i].w - 2, L[i].h);
        ho_resample(sp, s->w, 0, 1, s->w, s->h - 1, arena + L[i].off[2], L[i].w, L[i].w, L[i].h - 2);
        ho_resample(sp, s->w, 1, 1, s->w - 1, s->h - 1, arena + L[i].off[3], L[i].w, L[i].w - 2, L[i].h - 2);
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* Cascade scan, ccv.js:150-246.                                                                      */

typedef struct {
    int32_t scale; /* i  (ccv.js:154) */
    int32_t q;     /* half-pixel phase, dx = q&1, dy = q>>1 (ccv.js:151-152,178) */
    int32_t x, y;  /* window index on the quarter-resolution plane (ccv.js:181-182) */
    double sum;    /* last stage's sum = confidence (ccv.js:233) */
} ho_hit;

/* one feature: fires iff min(valid positive pixels) > max(valid negative pixels); ccv.js:189-220 with the
 * early exits removed (they never change the outcome: the loop keeps pmin > nmax as an invariant) */
static inline int ho_feature_fires(const ho_feature *f, const uint8_t *const pl[3], const int st[3], const int ox[3],
                                   const int oy[3]) {
    int pmin = 256, nmax = -1;
    for (int k = 0; k < f->size; k++) {
        int z = f->pz[k];
        if (z >= 0) {
            int p = pl[z][(size_t)(oy[z] + f->py[k]) * st[z] + ox[z] + f->px[k]];
            if (p < pmin) pmin = p;
        }
        z = f->nz[k];
        if (z >= 0) {
            int v = pl[z][(size_t)(oy[z] + f->ny[k]) * st[z] + ox[z] + f->nx[k]];
            if (v > nmax) nmax = v;
        }
    }
    return pmin > nmax;
}

/* Scans one pyramid; appends hits in the reference's emission order (scale, q, y, x).  Returns the number of
 * hits found (may exceed cap; only the first cap are stored).  stage_pass (optional, nstages+1 counters):
 * stage_pass[j] += windows that entered stage j (j = nstages: full survivors) — for roofline bookkeeping. */
int64_t ho_scan(const uint8_t *arena, const ho_level *L, int n, int interval, const ho_cascade *c, ho_hit *out,
                int64_t cap, int64_t *stage_pass) {
    const int next = interval + 1;
    const int upto = n - next * 2;
    int64_t nh = 0;
    for (int i = 0; i < upto; i++) {
        const ho_level *l0 = &L[i], *l1 = &L[i + next], *l2 = &L[i + next * 2];
        const int qw = l2->w - (int)(c->width / 4), qh = l2->h - (int)(c->height / 4); /* ccv.js:155-156 */
        if (qw <= 0 || qh <= 0) continue;
        const int st[3] = {l0->w, l1->w, l2->w};
        for (int q = 0; q < 4; q++) { /* ccv.js:178 */
            const int dx = q & 1, dy = q >> 1;
            const uint8_t *pl[3] = {arena + l0->off[0], arena + l1->off[0], arena + l2->off[q]};
            for (int y = 0; y < qh; y++) {
                for (int x = 0; x < qw; x++) {
                    const int ox[3] = {4 * x + 2 * dx, 2 * x + dx, x}; /* ccv.js:180,235-237 */
                    const int oy[3] = {4 * y + 2 * dy, 2 * y + dy, y};
                    double sum = 0;
                    int flag = 1;
                    for (uint32_t j = 0; j < c->nstages; j++) { /* ccv.js:185-226 */
                        const ho_stage *sg = &c->stages[j];
                        const ho_feature *f = c->features + sg->first;
                        if (stage_pass) stage_pass[j]++;
                        sum = 0;
                        for (uint32_t k = 0; k < sg->count; k++)
                            sum += f[k].alpha[ho_feature_fires(&f[k], pl, st, ox, oy)]; /* sequential binary64 adds */
                        if (sum < sg->threshold) {
                            flag = 0;
                            break;
                        }
                    }
                    if (flag) {
                        if (stage_pass) stage_pass[c->nstages]++;
                        if (nh < cap) {
                            out[nh].scale = i;
                            out[nh].q = q;
                            out[nh].x = x;
                            out[nh].y = y;
                            out[nh].sum = sum;
                        }
                        nh++;
                    }
                }
            }
        }
    }
    return nh;
}

/* rectangles of the raw hits, ccv.js:228-233,244-245: scale_x is built by repeated multiplication */
typedef struct {
    double x, y, width, height, confidence;
    int32_t neighbors, pad;
} ho_rect;

void ho_hits_to_rects(const ho_hit *hits, int64_t nh, int interval, int cw, int ch, ho_rect *out) {
    double sx[HO_MAX_LEVELS];
    const double scale = ho_scale(interval);
    sx[0] = 1;
    for (int i = 1; i < HO_MAX_LEVELS; i++) sx[i] = sx[i - 1] * scale;
    for (int64_t k = 0; k < nh; k++) {
        const ho_hit *h = &hits[k];
        const double s = sx[h->scale];
        out[k].x = (double)(h->x * 4 + (h->q & 1) * 2) * s;
        out[k].y = (double)(h->y * 4 + (h->q >> 1) * 2) * s;
        out[k].width = (double)cw * s;
        out[k].height = (double)ch * s;
        out[k].confidence = h->sum;
        out[k].neighbors = 1;
        out[k].pad = 0;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* ccv.array_group (ccv.js:34-107) with detect_objects' similarity predicate (ccv.js:252-261), then the   */
/* per-class averaging (ccv.js:262-303) and nested-rectangle filter (ccv.js:305-330).                   */

static int ho_similar(const ho_rect *r1, const ho_rect *r2) { /* ccv.js:252-261 */
    double distance = floor(r1->width * 0.25 + 0.5);
    return r2->x <= r1->x + distance && r2->x >= r1->x - distance && r2->y <= r1->y + distance &&
           r2->y >= r1->y - distance && r2->width <= floor(r1->width * 1.5 + 0.5) &&
           floor(r2->width * 1.5 + 0.5) >= r1->width;
}

/* returns number of output rects (<= n); out must hold n entries */
int ho_group(const ho_rect *seq, int n, int min_neighbors, ho_rect *out) {
    if (n <= 0) return 0;
    int *parent = (int *)malloc(sizeof(int) * n), *rank = (int *)calloc(n, sizeof(int));
    int *idx = (int *)malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) parent[i] = -1;
    for (int i = 0; i < n; i++) { /* ccv.js:41-89 */
        int root = i;
        while (parent[root] != -1) root = parent[root];
        for (int j = 0; j < n; j++) {
            if (i != j && ho_similar(&seq[i], &seq[j])) {
                int root2 = j;
                while (parent[root2] != -1) root2 = parent[root2];
                if (root2 != root) {
                    if (rank[root] > rank[root2]) {
                        parent[root2] = root;
                    } else {
                        parent[root] = root2;
                        if (rank[root] == rank[root2]) rank[root2]++;
                        root = root2;
                    }
                    int temp, node2 = j;
                    while (parent[node2] != -1) {
                        temp = node2;
                        node2 = parent[node2];
                        parent[temp] = root;
                    }
                    node2 = i;
                    while (parent[node2] != -1) {
                        temp = node2;
                        node2 = parent[node2];
                        parent[temp] = root;
                    }
                }
            }
        }
    }
    int class_idx = 0; /* ccv.js:90-105: class ids in first-seen order */
    for (int i = 0; i < n; i++) {
        int node1 = i;
        while (parent[node1] != -1) node1 = parent[node1];
        if (rank[node1] >= 0) rank[node1] = ~class_idx++;
        idx[i] = ~rank[node1];
    }
    const int ncomp = class_idx;
    ho_rect *comps = (ho_rect *)calloc(ncomp + 1, sizeof(ho_rect));
    for (int i = 0; i < n; i++) { /* ccv.js:274-289 */
        ho_rect *cp = &comps[idx[i]];
        if (cp->neighbors == 0) cp->confidence = seq[i].confidence;
        ++cp->neighbors;
        cp->x += seq[i].x;
        cp->y += seq[i].y;
        cp->width += seq[i].width;
        cp->height += seq[i].height;
        cp->confidence = cp->confidence > seq[i].confidence ? cp->confidence : seq[i].confidence;
    }
    ho_rect *seq2 = (ho_rect *)malloc(sizeof(ho_rect) * (ncomp + 1));
    int n2 = 0;
    for (int i = 0; i < ncomp; i++) { /* ccv.js:293-303 */
        int nn = comps[i].neighbors;
        if (nn >= min_neighbors) {
            seq2[n2].x = (comps[i].x * 2 + nn) / (2 * nn);
            seq2[n2].y = (comps[i].y * 2 + nn) / (2 * nn);
            seq2[n2].width = (comps[i].width * 2 + nn) / (2 * nn);
            seq2[n2].height = (comps[i].height * 2 + nn) / (2 * nn);
            seq2[n2].neighbors = nn;
            seq2[n2].pad = 0;
            seq2[n2].confidence = comps[i].confidence;
            n2++;
        }
    }
    int nout = 0;
    for (int i = 0; i < n2; i++) { /* ccv.js:307-330 */
        const ho_rect *r1 = &seq2[i];
        int flag = 1;
        for (int j = 0; j < n2; j++) {
            const ho_rect *r2 = &seq2[j];
            double distance = floor(r2->width * 0.25 + 0.5);
            if (i != j && r1->x >= r2->x - distance && r1->y >= r2->y - distance &&
                r1->x + r1->width <= r2->x + r2->width + distance &&
                r1->y + r1->height <= r2->y + r2->height + distance &&
                (r2->neighbors > (3 > r1->neighbors ? 3 : r1->neighbors) || r1->neighbors < 3)) {
                flag = 0;
                break;
            }
        }
        if (flag) out[nout++] = *r1;
    }
    free(parent);
    free(rank);
    free(idx);
    free(comps);
    free(seq2);
    return nout;
}

/* ------------------------------------------------------------------------------------------------ */
/* One-call detect = ccv.grayscale + ccv.detect_objects(canvas, cascade, interval, 0) on one RGBA frame. */
/* gray_in_r != 0: the frame is already gray (byte 0 is used as is, like detect_objects itself does).  */

int64_t ho_detect_raw(const uint8_t *rgba, int w, int h, int gray_in_r, const uint8_t *blob, size_t blob_len,
                      int interval, ho_hit *out, int64_t cap, int64_t *stage_pass) {
    ho_cascade c;
    if (ho_parse_cascade(blob, blob_len, &c) != 0) return -1;
    ho_level L[HO_MAX_LEVELS];
    int64_t bytes = 0;
    int n = ho_pyramid_layout(w, h, interval, (int)c.width, (int)c.height, L, &bytes);
    if (n < 0) return -1;
    uint8_t *arena = (uint8_t *)malloc((size_t)bytes + 16);
    if (!arena) return -1;
    if (gray_in_r) {
        for (size_t i = 0, m = (size_t)w * h; i < m; i++) arena[i] = rgba[4 * i];
    } else {
        ho_gray_plane(rgba, w, h, arena);
    }
    ho_pyramid_build(arena, L, n, interval);
    int64_t nh = ho_scan(arena, L, n, interval, &c, out, cap, stage_pass);
    free(arena);
    return nh;
}

/* pyramid only (for plane-by-plane parity): caller provides arena of *arena_bytes (query with arena == NULL) */
int ho_pyramid(const uint8_t *rgba, int w, int h, int gray_in_r, int interval, int cw, int ch, ho_level *levels,
               uint8_t *arena, int64_t *arena_bytes) {
    int n = ho_pyramid_layout(w, h, interval, cw, ch, levels, arena_bytes);
    if (n < 0 || !arena) return n;
    if (gray_in_r) {
        for (size_t i = 0, m = (size_t)w * h; i < m; i++) arena[i] = rgba[4 * i];
    } else {
        ho_gray_plane(rgba, w, h, arena);
    }
    ho_pyramid_build(arena, levels, n, interval);
    return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* camshift, camshift.js.                                                                             */

typedef struct {
    int32_t model[4096];   /* _modelHist bins, camshift.js:208 */
    int32_t sw[4];         /* _searchWindow x,y,width,height, camshift.js:209 */
    double x, y, width, height, angle; /* _trackObj, camshift.js:362-367 */
    int32_t calc_angles, pad;
} ho_cs_state;

static int32_t ho_toint32(double v) { /* ECMAScript ToInt32 (used by >>0 and <<2) */
    if (!isfinite(v)) return 0;
    double t = trunc(v);
    double m = fmod(t, 4294967296.0);
    if (m < 0) m += 4294967296.0;
    return (int32_t)(uint32_t)m;
}

static inline int ho_bin(const uint8_t *p) { /* camshift.js:63-66 */
    return 256 * (p[0] >> 4) + 16 * (p[1] >> 4) + (p[2] >> 4);
}

/* camshift.Tracker.initTracker, camshift.js:198-211; getImageData outside the canvas = transparent black */
void ho_cs_init(ho_cs_state *s, const uint8_t *rgba, int w, int h, int rx, int ry, int rw, int rh, int calc_angles) {
    memset(s, 0, sizeof(*s));
    s->calc_angles = calc_angles;
    for (int y = ry; y < ry + rh; y++)
        for (int x = rx; x < rx + rw; x++) {
            if (x >= 0 && x < w && y >= 0 && y < h)
                s->model[ho_bin(rgba + 4 * ((size_t)y * w + x))]++;
            else
                s->model[0]++;
        }
    s->sw[0] = rx;
    s->sw[1] = ry;
    s->sw[2] = rw;
    s->sw[3] = rh;
}

typedef struct {
    double m00, m01, m10, m11, m02, m20, invM00, xc, yc, mu20, mu02, mu11;
} ho_moments;

/* camshift.Moments, camshift.js:79-120 — pdf[x][y] looked up on the fly through the weights table; the
 * summation order (x outer, y inner) is the reference's */
static void ho_moments_calc(ho_moments *m, const uint8_t *rgba, int W, const double *weights, int x, int y, int w,
                            int h, int second) {
    memset(m, 0, sizeof(*m));
#ifdef HO_MOMENTS_ROW_MAJOR /* NOT the reference's order: tools/cpu_cs_order_check.py builds this variant to ask whether a case's result depends on the
                             * summation order at all (a track object that changes under it cannot be reproduced by ANY parallel reduction) */
    for (int j = y; j < h; j++)
        for (int i = x; i < w; i++) {
            double vx = (double)(i - x), vy = (double)(j - y);
            double val = weights[ho_bin(rgba + 4 * ((size_t)j * W + i))];
            m->m00 += val;
            m->m01 += vy * val;
            m->m10 += vx * val;
            if (second) {
                m->m11 += vx * vy * val;
                m->m02 += vy * vy * val;
                m->m20 += vx * vx * val;
            }
        }
    if (0)
#endif
#ifdef HO_MOMENTS_TWO_ACCUMULATORS /* the reference's loops with even and odd rows summed apart and added at the end: the smallest step towards a tree */
    {
        ho_moments p[2];
        memset(p, 0, sizeof(p));
        for (int i = x; i < w; i++)
            for (int j = y; j < h; j++) {
                double vx = (double)(i - x), vy = (double)(j - y);
                double val = weights[ho_bin(rgba + 4 * ((size_t)j * W + i))];
                ho_moments *q = &p[(j - y) & 1];
                q->m00 += val;
                q->m01 += vy * val;
                q->m10 += vx * val;
                if (second) {
                    q->m11 += vx * vy * val;
                    q->m02 += vy * vy * val;
                    q->m20 += vx * vx * val;
                }
            }
        m->m00 = p[0].m00 + p[1].m00, m->m01 = p[0].m01 + p[1].m01, m->m10 = p[0].m10 + p[1].m10;
        m->m11 = p[0].m11 + p[1].m11, m->m02 = p[0].m02 + p[1].m02, m->m20 = p[0].m20 + p[1].m20;
    }
    if (0)
#endif
#ifdef HO_MOMENTS_REVERSED /* the reference's loops walked backwards (last column first, bottom row first): another order no reduction is obliged to avoid */
    for (int i = w - 1; i >= x; i--)
        for (int j = h - 1; j >= y; j--) {
            double vx = (double)(i - x), vy = (double)(j - y);
            double val = weights[ho_bin(rgba + 4 * ((size_t)j * W + i))];
            m->m00 += val;
            m->m01 += vy * val;
            m->m10 += vx * val;
            if (second) {
                m->m11 += vx * vy * val;
                m->m02 += vy * vy * val;
                m->m20 += vx * vx * val;
            }
        }
    if (0)
#endif
    for (int i = x; i < w; i++) {
        double vx = (double)(i - x);
        for (int j = y; j < h; j++) {
            double val = weights[ho_bin(rgba + 4 * ((size_t)j * W + i))];
            double vy = (double)(j - y);
            m->m00 += val;
            m->m01 += vy * val;
            m->m10 += vx * val;
            if (second) {
                m->m11 += vx * vy * val;
                m->m02 += vy * vy * val;
                m->m20 += vx * vx * val;
            }
        }
    }
    m->invM00 = 1 / m->m00;
    m->xc = m->m10 * m->invM00;
    m->yc = m->m01 * m->invM00;
    m->mu20 = m->mu02 = m->mu11 = NAN; /* undefined in JS when !second */
    if (second) {
        m->mu20 = m->m20 - m->m10 * m->xc;
        m->mu02 = m->m02 - m->m01 * m->yc;
        m->mu11 = m->m11 - m->m01 * m->xc;
    }
}

/* camshift.Tracker.track -> camShift -> meanShift, camshift.js:213-312 */
void ho_cs_track(ho_cs_state *s, const uint8_t *rgba, int w, int h) {
    if (w == 0 || h == 0) return; /* camshift.js:219 */
    int32_t cur[4096];
    double weights[4096];
    memset(cur, 0, sizeof(cur));
    for (size_t i = 0, n = (size_t)w * h; i < n; i++) cur[ho_bin(rgba + 4 * i)]++; /* camshift.js:268 */
    for (int i = 0; i < 4096; i++) {                                                 /* camshift.js:314-330 */
        if (cur[i] != 0) {
            double p = (double)s->model[i] / (double)cur[i];
            weights[i] = p < 1 ? p : 1;
        } else {
            weights[i] = 0;
        }
    }
    ho_moments m;
    memset(&m, 0, sizeof(m));
    int prevx = s->sw[0], prevy = s->sw[1];
    for (int it = 0; it < 10; it++) { /* camshift.js:284-306 */
        int wadx = s->sw[0] > 0 ? s->sw[0] : 0;
        int wady = s->sw[1] > 0 ? s->sw[1] : 0;
        int wadw = (wadx + s->sw[2] < w) ? wadx + s->sw[2] : w;
        int wadh = (wady + s->sw[3] < h) ? wady + s->sw[3] : h;
        ho_moments_calc(&m, rgba, w, weights, wadx, wady, wadw, wadh, it == 9);
        s->sw[0] += ho_toint32(m.xc - (double)s->sw[2] / 2);
        s->sw[1] += ho_toint32(m.yc - (double)s->sw[3] / 2);
        if (s->sw[0] == prevx && s->sw[1] == prevy) {
            ho_moments_calc(&m, rgba, w, weights, wadx, wady, wadw, wadh, 1);
            break;
        } else {
            prevx = s->sw[0];
            prevy = s->sw[1];
        }
    }
    s->sw[0] = s->sw[0] < w ? s->sw[0] : w; /* camshift.js:308-309 */
    if (s->sw[0] < 0) s->sw[0] = 0;
    s->sw[1] = s->sw[1] < h ? s->sw[1] : h;
    if (s->sw[1] < 0) s->sw[1] = 0;

    const double a = m.mu20 * m.invM00, c = m.mu02 * m.invM00; /* camshift.js:230-231 */
    if (s->calc_angles) {                                       /* camshift.js:233-245 */
        const double b = m.mu11 * m.invM00;
        const double d = a + c;
        const double e = sqrt((4 * b * b) + ((a - c) * (a - c)));
        s->width = (double)(int32_t)((uint32_t)ho_toint32(sqrt((d - e) * 0.5)) << 2);
        s->height = (double)(int32_t)((uint32_t)ho_toint32(sqrt((d + e) * 0.5)) << 2);
        s->angle = atan2(2 * b, a - c + e);
        if (s->angle < 0) s->angle = s->angle + 3.141592653589793;
    } else { /* camshift.js:247-249 */
        s->width = (double)(int32_t)((uint32_t)ho_toint32(sqrt(a)) << 2);
        s->height = (double)(int32_t)((uint32_t)ho_toint32(sqrt(c)) << 2);
        s->angle = 3.141592653589793 / 2;
    }
    { /* camshift.js:253-254 (uses the OLD window size) */
        double cx = (double)s->sw[0] + (double)s->sw[2] / 2, cy = (double)s->sw[1] + (double)s->sw[3] / 2;
        cx = cx < (double)w ? cx : (double)w;
        cy = cy < (double)h ? cy : (double)h;
        s->x = floor(cx > 0 ? cx : 0);
        s->y = floor(cy > 0 ? cy : 0);
    }
    s->sw[2] = (int32_t)floor(1.1 * s->width); /* camshift.js:257-258 */
    s->sw[3] = (int32_t)floor(1.1 * s->height);
}

/* struct sizes for the ctypes binding */
int ho_sizeof_level(void) { return (int)sizeof(ho_level); }
int ho_sizeof_hit(void) { return (int)sizeof(ho_hit); }
int ho_sizeof_rect(void) { return (int)sizeof(ho_rect); }
int ho_sizeof_cs_state(void) { return (int)sizeof(ho_cs_state); }
