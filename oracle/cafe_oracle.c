/*
 * cafe_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked or called by the product path).
 *
 * A plain-C, explicit-state CPU restatement of the HotSketch algorithm the reference keeps in
 * tricks/sketch.cpp.  It exists so that the HIP kernels can be checked bit-for-bit on machines
 * where /root/reference is absent (the GPU box).  Parity status: PINNED -- tests/test_oracle_*.py
 * replay golden traces generated from the reference's own sketch.cpp (oracle/_ref, built by
 * oracle/Makefile) and compare every output and every slot of the state.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Differences from the reference that are deliberate:
 *   - the state lives in a handle (the reference keeps process globals, sketch.cpp:9-13,297);
 *     a fresh handle equals a fresh reference process.
 *   - no 16384-entry limit on a call (the reference writes into static ins[]/que[], sketch.cpp:10).
 *   - the reference's reset() orders equal scores by heap address (sketch.cpp:79,96); the caller
 *     states which table sits lower in memory (lru_first), see cafe_oracle_create.
 *   - reset() with fewer than lim candidates is undefined behaviour in the reference
 *     (vec[l - lim] with a negative index, sketch.cpp:112); here nothing loses its slot, every
 *     candidate without a slot gets one while the free list lasts, and Threshold is kept.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define W 4                 /* slots per bucket, both tables: sketch.cpp:7 (m1 = m2 = 4) */
#define DECAY_V 10000.0     /* sketch.cpp:8 */

typedef struct cafe_oracle {
    int lim, n1, n2, adjust, lru_first;
    float thr;              /* sketch.cpp:19 */
    double alpha, decay;    /* sketch.cpp:9,11 */
    int batch_num, flip;    /* sketch.cpp:12,13 */
    int num, real_n, clk;   /* sketch.cpp:17 (num, real_n, t) */
    int resets;
    long stats[8];          /* coverage counters for the tests, not part of the algorithm */
    /* main table, sketch.cpp:23-28, kept as structure of arrays [n1][W] */
    int32_t *mval, *mdic; float *mcnt; uint8_t *mflip;
    /* LRU filter table, sketch.cpp:29-35, [n2][W] */
    int32_t *lval, *ldic, *lt; float *lcnt; uint8_t *lflip;
    /* free hot slots, FIFO: sketch.cpp:37,55-56 */
    uint32_t *fifo; long fifo_cap, fifo_head, fifo_len;
} cafe_oracle_t;

/* ---- free list (std::queue<uint32_t> in the reference) ---- */
static void fifo_push(cafe_oracle_t *o, uint32_t v) {
    if (o->fifo_len == o->fifo_cap) {           /* cannot happen: at most lim-1 slots exist */
        fprintf(stderr, "cafe_oracle: free list overflow\n"); abort();
    }
    o->fifo[(o->fifo_head + o->fifo_len) % o->fifo_cap] = v;
    o->fifo_len++;
}
static uint32_t fifo_pop(cafe_oracle_t *o) {
    uint32_t v = o->fifo[o->fifo_head];
    o->fifo_head = (o->fifo_head + 1) % o->fifo_cap;
    o->fifo_len--;
    return v;
}

/* sketch.cpp:38-43 */
static int hash_main(const cafe_oracle_t *o, uint32_t x) { return (int)((long long)x * 1000000007LL % o->n1); }
static int hash_lru(const cafe_oracle_t *o, uint32_t x) { return (int)((long long)x * 1000000007LL % o->n2); }

/* sketch.cpp:44-70 (constructor) and 309-313 (init) */
cafe_oracle_t *cafe_oracle_create(int lim, int threshold, int adjust, double alpha, int lru_first) {
    cafe_oracle_t *o = (cafe_oracle_t *)calloc(1, sizeof(*o));
    o->lim = lim; o->adjust = adjust; o->alpha = alpha; o->lru_first = lru_first;
    o->thr = (float)threshold;
    o->decay = 1.0; o->batch_num = 0; o->flip = 0;
    o->num = 0; o->real_n = 0; o->clk = 0; o->resets = 0;
    o->n1 = (int)(lim * 0.9);
    o->n2 = (int)(lim * 0.1);
    size_t a = (size_t)o->n1 * W, b = (size_t)o->n2 * W;
    o->mval = calloc(a ? a : 1, 4); o->mdic = calloc(a ? a : 1, 4); o->mcnt = calloc(a ? a : 1, 4);
    o->mflip = calloc(o->n1 ? o->n1 : 1, 1);
    o->lval = calloc(b ? b : 1, 4); o->ldic = calloc(b ? b : 1, 4); o->lt = calloc(b ? b : 1, 4);
    o->lcnt = calloc(b ? b : 1, 4); o->lflip = calloc(o->n2 ? o->n2 : 1, 1);
    o->fifo_cap = lim > 1 ? lim : 1;
    o->fifo = calloc(o->fifo_cap, 4);
    o->fifo_head = 0; o->fifo_len = 0;
    for (int i = 1; i < lim; ++i) fifo_push(o, (uint32_t)i);   /* slot 0 is never handed out */
    return o;
}

void cafe_oracle_destroy(cafe_oracle_t *o) {
    if (!o) return;
    free(o->mval); free(o->mdic); free(o->mcnt); free(o->mflip);
    free(o->lval); free(o->ldic); free(o->lt); free(o->lcnt); free(o->lflip);
    free(o->fifo); free(o);
}

/* lazy decay, sketch.cpp:130-140: float divided by a double constant, rounded back to float */
static void sync_main(cafe_oracle_t *o, int key) {
    if (o->mflip[key] == (uint8_t)o->flip) return;
    o->mflip[key] = (uint8_t)o->flip;
    for (int i = 0; i < W; ++i) o->mcnt[key * W + i] = (float)((double)o->mcnt[key * W + i] / DECAY_V);
}
static void sync_lru(cafe_oracle_t *o, int key) {
    if (o->lflip[key] == (uint8_t)o->flip) return;
    o->lflip[key] = (uint8_t)o->flip;
    for (int i = 0; i < W; ++i) o->lcnt[key * W + i] = (float)((double)o->lcnt[key * W + i] / DECAY_V);
}

/* sketch.cpp:118-128: main table only; an entry without a hot slot reads as cold */
static int32_t query_one(const cafe_oracle_t *o, uint32_t x) {
    int key = hash_main(o, x);
    for (int i = 0; i < W; ++i)
        if ((uint32_t)o->mval[key * W + i] == x && o->mdic[key * W + i]) return -o->mdic[key * W + i];
    return (int32_t)x;
}

/* sketch.cpp:232-259: move an entry (from the LRU table) into the main table */
static int promote(cafe_oracle_t *o, uint32_t x, float c, int32_t dic) {
    int key = hash_main(o, x);
    sync_main(o, key);
    int32_t *val = o->mval + key * W, *d = o->mdic + key * W; float *cnt = o->mcnt + key * W;
    int lo = -1; float lo_cnt = 1e9f;
    for (int i = 0; i < W; ++i) {
        if (val[i] == 0) { val[i] = (int32_t)x; cnt[i] = c; d[i] = dic; return 1; }
        if (cnt[i] < lo_cnt) { lo_cnt = cnt[i]; lo = i; }
    }
    /* all four scores >= 1e9 leaves lo == -1 in the reference too (out-of-bounds there); fence it */
    if (lo < 0) lo = 0;
    if (!d[lo]) {                 /* space-saving replacement of the weakest non-hot entry */
        cnt[lo] += c; val[lo] = (int32_t)x; d[lo] = dic; return 1;
    }
    o->stats[5]++;
    if (dic != 0) { fifo_push(o, (uint32_t)dic); o->stats[6]++; }   /* the weakest entry is hot: give the slot back */
    return 0;
}

/* sketch.cpp:161-210 */
static int insert_lru(cafe_oracle_t *o, uint32_t x, float c) {
    int key = hash_lru(o, x), victim = -1, t_min = 1000000000, got = 0;
    sync_lru(o, key);
    int32_t *val = o->lval + key * W, *d = o->ldic + key * W, *t = o->lt + key * W;
    float *cnt = o->lcnt + key * W;
    for (int i = 0; i < W; ++i) {
        if ((uint32_t)val[i] == x) {
            o->stats[1]++;
            t[i] = ++o->clk;
            cnt[i] += c;
            if (cnt[i] >= o->thr && cnt[i] - c < o->thr) o->real_n++;
            if (cnt[i] >= o->thr && o->fifo_len > 0 && !d[i]) {
                d[i] = (int32_t)fifo_pop(o); got = 1; o->num++;
            }
            if (cnt[i] >= o->thr && o->fifo_len == 0 && !d[i]) o->stats[7]++;
            if (cnt[i] >= o->thr) {
                o->stats[4]++;
                if (promote(o, (uint32_t)val[i], cnt[i], d[i]) == 0) got = 0;
                val[i] = 0; t[i] = 0; cnt[i] = 0; d[i] = 0;
            }
            return got;
        }
        if (t_min > t[i]) { t_min = t[i]; victim = i; }
    }
    if (victim < 0) victim = 0;   /* unreachable while the clock stays below 1e9 */
    o->stats[2]++;
    if (d[victim] || cnt[victim] >= 5) o->stats[3]++;
    if (d[victim] || cnt[victim] >= 5) promote(o, (uint32_t)val[victim], cnt[victim], d[victim]);
    val[victim] = (int32_t)x;
    t[victim] = ++o->clk;
    cnt[victim] = 1;              /* not c: sketch.cpp:198 */
    d[victim] = 0;
    if (cnt[victim] >= o->thr && cnt[victim] - c < o->thr) o->real_n++;
    if (cnt[victim] >= o->thr && o->fifo_len > 0 && !d[victim]) {
        d[victim] = (int32_t)fifo_pop(o); got = 1; o->num++;
    }
    return got;
}

/* sketch.cpp:211-231 */
static int insert_one(cafe_oracle_t *o, uint32_t x, float c) {
    int key = hash_main(o, x);
    sync_main(o, key);
    int32_t *val = o->mval + key * W, *d = o->mdic + key * W; float *cnt = o->mcnt + key * W;
    for (int i = 0; i < W; ++i) {
        if ((uint32_t)val[i] == x) {
            o->stats[0]++;
            cnt[i] += c;
            float now = cnt[i];
            int got = 0;
            if (now >= o->thr && now - c < o->thr) o->real_n++;
            if (now >= o->thr && !d[i] && o->fifo_len == 0) o->stats[7]++;
            if (now >= o->thr && !d[i] && o->fifo_len > 0) {
                d[i] = (int32_t)fifo_pop(o); got = 1; o->num++;
            }
            return got;
        }
    }
    return insert_lru(o, x, c);
}

/* sketch.cpp:76-116 */
typedef struct { float score; long order; int32_t *dic; } cand_t;
static int cand_cmp(const void *pa, const void *pb) {
    const cand_t *a = pa, *b = pb;
    if (a->score < b->score) return -1;
    if (a->score > b->score) return 1;
    return (a->order > b->order) - (a->order < b->order);
}
static void reset_threshold(cafe_oracle_t *o) {
    o->resets++;
    size_t cap = ((size_t)o->n1 + o->n2) * W, l = 0;
    cand_t *vec = malloc((cap ? cap : 1) * sizeof(cand_t));
    long main_base = o->lru_first ? (long)o->n2 * W : 0;
    long lru_base = o->lru_first ? 0 : (long)o->n1 * W;
    for (int key = 0; key < o->n1; ++key) {
        sync_main(o, key);
        for (int i = 0; i < W; ++i)
            if (o->mcnt[key * W + i] >= o->thr || o->mdic[key * W + i]) {
                vec[l].score = o->mcnt[key * W + i]; vec[l].order = main_base + (long)key * W + i;
                vec[l].dic = &o->mdic[key * W + i]; ++l;
            }
    }
    for (int key = 0; key < o->n2; ++key) {
        sync_lru(o, key);
        for (int i = 0; i < W; ++i)
            if (o->lcnt[key * W + i] >= o->thr || o->ldic[key * W + i]) {
                vec[l].score = o->lcnt[key * W + i]; vec[l].order = lru_base + (long)key * W + i;
                vec[l].dic = &o->ldic[key * W + i]; ++l;
            }
    }
    qsort(vec, l, sizeof(cand_t), cand_cmp);
    long cut = (long)l - o->lim;          /* vec[0..cut] lose their slot */
    for (long i = 0; i <= cut; ++i)
        if (*vec[i].dic != 0) { fifo_push(o, (uint32_t)*vec[i].dic); *vec[i].dic = 0; }
    for (long i = cut + 1 < 0 ? 0 : cut + 1; i < (long)l; ++i)
        if (*vec[i].dic == 0 && o->fifo_len > 0) *vec[i].dic = (int32_t)fifo_pop(o);
    if (cut >= 0) o->thr = vec[cut].score;
    o->real_n = o->lim;
    free(vec);
}

/* sketch.cpp:281-296 (v != NULL) and 266-280 (v == NULL: every score is 1) */
void cafe_oracle_batch_insert_val(cafe_oracle_t *o, const uint32_t *ids, const float *v, int len, int32_t *ins) {
    o->batch_num++;
    o->decay *= o->alpha;
    if (o->decay > DECAY_V) {
        o->decay /= DECAY_V;
        o->thr = (float)((double)o->thr / DECAY_V);
        o->flip ^= 1;
    }
    if ((double)o->real_n > o->lim * 1.2 && o->adjust) reset_threshold(o);
    for (int i = 0; i < len; ++i) ins[i] = insert_one(o, ids[i], v ? v[i] : 1.0f);
}

/* sketch.cpp:260-265 */
void cafe_oracle_batch_query(const cafe_oracle_t *o, const uint32_t *ids, int len, int32_t *que) {
    for (int i = 0; i < len; ++i) que[i] = query_one(o, ids[i]);
}

/* Median tier of tricks/sketch_ml_embed.py.  The module binds lib.batch_query_median (:82-84) and reads "< 0" as
 * "medium importance" (:148-152), with threshold2 = 0.2 * threshold1 (:66-67); no sketch.cpp in the reference exports
 * the function, so this is a DEFINITION, not a restatement (parity unpinned): an id reads -id when query() does not
 * report it hot and the main table holds it without a hot slot at a score -- pending lazy decay applied, read-only --
 * of at least 0.2 * Threshold; +id otherwise. */
void cafe_oracle_batch_query_median(const cafe_oracle_t *o, const uint32_t *ids, int len, int32_t *out) {
    for (int i = 0; i < len; ++i) {
        uint32_t x = ids[i];
        int32_t r = (int32_t)x;
        if (query_one(o, x) >= 0) {
            int key = hash_main(o, x);
            for (int s = 0; s < W; ++s) {
                if ((uint32_t)o->mval[key * W + s] != x || o->mdic[key * W + s]) continue;
                float c = o->mcnt[key * W + s];
                if (o->mflip[key] != (uint8_t)o->flip) c = (float)((double)c / DECAY_V);
                if ((double)c >= 0.2 * (double)o->thr) r = -(int32_t)x;
                break;
            }
        }
        out[i] = r;
    }
}

/* sketch.cpp:314-320 */
double cafe_oracle_analyse(const cafe_oracle_t *o, const uint32_t *ids, int len) {
    int hot = 0;
    for (int i = 0; i < len; ++i) if (query_one(o, ids[i]) < 0) ++hot;
    return 1.0 * hot / len;
}
/* sketch.cpp:321-327 with Query, sketch.cpp:142-150 */
double cafe_oracle_analyse1(const cafe_oracle_t *o, const uint32_t *ids, int len) {
    int present = 0;
    for (int i = 0; i < len; ++i) {
        int key = hash_main(o, ids[i]);
        for (int s = 0; s < W; ++s)
            if ((uint32_t)o->mval[key * W + s] == ids[i]) { ++present; break; }
    }
    return 1.0 * present / len;
}

/* ---- state access for the parity tests ---- */
void cafe_oracle_stats(const cafe_oracle_t *o, long *out) { memcpy(out, o->stats, sizeof(o->stats)); }
void cafe_oracle_dims(const cafe_oracle_t *o, int32_t *out /* n1, n2, lim, fifo_len */) {
    out[0] = o->n1; out[1] = o->n2; out[2] = o->lim; out[3] = (int32_t)o->fifo_len;
}
/* scalars: thr, num, real_n, clk, flip, batch_num, resets ; decay separately */
void cafe_oracle_scalars(const cafe_oracle_t *o, float *thr, int32_t *ints /* 6 */, double *decay) {
    *thr = o->thr; ints[0] = o->num; ints[1] = o->real_n; ints[2] = o->clk; ints[3] = o->flip;
    ints[4] = o->batch_num; ints[5] = o->resets; *decay = o->decay;
}
void cafe_oracle_dump(const cafe_oracle_t *o, int32_t *mval, float *mcnt, int32_t *mdic, uint8_t *mflip,
                      int32_t *lval, float *lcnt, int32_t *ldic, int32_t *lt, uint8_t *lflip, uint32_t *fifo) {
    size_t a = (size_t)o->n1 * W, b = (size_t)o->n2 * W;
    memcpy(mval, o->mval, a * 4); memcpy(mcnt, o->mcnt, a * 4); memcpy(mdic, o->mdic, a * 4);
    memcpy(mflip, o->mflip, o->n1);
    memcpy(lval, o->lval, b * 4); memcpy(lcnt, o->lcnt, b * 4); memcpy(ldic, o->ldic, b * 4);
    memcpy(lt, o->lt, b * 4); memcpy(lflip, o->lflip, o->n2);
    for (long i = 0; i < o->fifo_len; ++i) fifo[i] = o->fifo[(o->fifo_head + i) % o->fifo_cap];
}

/* Inverse of cafe_oracle_dump + cafe_oracle_scalars: lets a test continue from a prepared state (e.g. an LRU clock
   close to the reference's 1e9 sentinel, sketch.cpp:159).  Test infrastructure; no counterpart in the reference. */
void cafe_oracle_load(cafe_oracle_t *o, const int32_t *mval, const float *mcnt, const int32_t *mdic, const uint8_t *mflip,
                      const int32_t *lval, const float *lcnt, const int32_t *ldic, const int32_t *lt, const uint8_t *lflip,
                      const uint32_t *fifo, long fifo_len, float thr, const int32_t *ints /* 6 */, double decay) {
    size_t a = (size_t)o->n1 * W, b = (size_t)o->n2 * W;
    memcpy(o->mval, mval, a * 4); memcpy(o->mcnt, mcnt, a * 4); memcpy(o->mdic, mdic, a * 4);
    memcpy(o->mflip, mflip, o->n1);
    memcpy(o->lval, lval, b * 4); memcpy(o->lcnt, lcnt, b * 4); memcpy(o->ldic, ldic, b * 4);
    memcpy(o->lt, lt, b * 4); memcpy(o->lflip, lflip, o->n2);
    o->fifo_head = 0; o->fifo_len = fifo_len < o->fifo_cap ? fifo_len : o->fifo_cap;
    for (long i = 0; i < o->fifo_len; ++i) o->fifo[i] = fifo[i];
    o->thr = thr; o->num = ints[0]; o->real_n = ints[1]; o->clk = ints[2]; o->flip = ints[3];
    o->batch_num = ints[4]; o->resets = ints[5]; o->decay = decay;
}
