/* cafe_oracle_ae.c -- CPU restatement of the earlier HotSketch generation the reference's artifact ships
 * (ArtifactEvaluation/embeddings/sketch.cpp:9-151, class SS, and its C entry points :207-228).
 *
 * TEST INFRASTRUCTURE ONLY: only tests/ and the smoke check may use it; the product path (libcafe_hip.so) never does.
 * Pinned by tests/golden/ae_*.npz: ins[] / que[] of every call recorded from that file compiled in the build container
 * (oracle/Makefile ref_more, tests/golden/make_ae_golden.py), decay() events and an exhausted free list included.
 *
 * Structure of that generation: `lim` buckets of four slots kept sorted by score (descending), bucket = id % lim; an id
 * whose score reaches Threshold takes a slot number from a FIFO free list; when the scores inserted since the last
 * decay exceed lim * Threshold * 10, every score is multiplied by decay_rate and entries that fall below Threshold give
 * their slot back.  One instance per field (ss[26]).
 *
 * Two places where the reference is undefined and this restatement fixes a behaviour:
 *  - Bucket::val is never initialised (:39-42 clear cnt and dic only); it is only read behind cnt != 0, so its initial
 *    content never matters.  Zero here.
 *  - update() (:61-75) clears its marker array with memset(v, 0, sizeof(v)), i.e. 8 bytes: the rest is whatever
 *    operator new returned.  The intended semantics (every slot number that no entry holds goes back to the free list,
 *    ascending) is what cafe_oracle_ae_update implements; goldens do not call update(). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define W 4

typedef struct {
    int s, k, lim, num;
    double tot, decay_rate;
    uint32_t *val; float *cnt; int32_t *dic;
    uint32_t *fifo; long cap, head, len;
    long decays;
} cafe_ae_t;

static void q_push(cafe_ae_t *o, uint32_t v) { o->fifo[(o->head + o->len) % o->cap] = v; o->len++; }
static uint32_t q_pop(cafe_ae_t *o) { uint32_t v = o->fifo[o->head]; o->head = (o->head + 1) % o->cap; o->len--; return v; }

/* :26-43 */
cafe_ae_t *cafe_oracle_ae_create(int threshold, int lim, double decay_rate) {
    cafe_ae_t *o = calloc(1, sizeof(*o));
    o->k = threshold; o->lim = lim; o->s = lim; o->decay_rate = decay_rate;
    size_t n = (size_t)(lim > 0 ? lim : 1) * W;
    o->val = calloc(n, 4); o->cnt = calloc(n, 4); o->dic = calloc(n, 4);
    o->cap = lim > 1 ? lim : 1;
    o->fifo = calloc(o->cap, 4);
    for (int i = 1; i < lim; ++i) q_push(o, (uint32_t)i);
    return o;
}
void cafe_oracle_ae_destroy(cafe_ae_t *o) {
    if (!o) return;
    free(o->val); free(o->cnt); free(o->dic); free(o->fifo); free(o);
}

static int hash(const cafe_ae_t *o, uint32_t v) { return (int)(v % (uint32_t)o->s); }       /* :20-22 */

/* :44-52 */
static int query(const cafe_ae_t *o, uint32_t v) {
    int key = hash(o, v);
    for (int i = 0; i < W; ++i)
        if (o->cnt[key * W + i] != 0 && o->val[key * W + i] == v)
            if (o->dic[key * W + i]) return -o->dic[key * W + i];
    return (int)v;
}

/* :76-88 */
static void decay(cafe_ae_t *o) {
    for (int key = 0; key < o->s; ++key)
        for (int i = 0; i < W; ++i) {
            size_t j = (size_t)key * W + i;
            if (o->dic[j] && (double)o->cnt[j] * o->decay_rate < (double)o->k) { q_push(o, (uint32_t)o->dic[j]); o->dic[j] = 0; }
            o->cnt[j] = (float)((double)o->cnt[j] * o->decay_rate);
        }
    o->tot = 0;
    o->decays++;
}

/* :89-124 */
static int insert(cafe_ae_t *o, uint32_t v, float c) {
    o->tot += c;
    int key = hash(o, v), id = 0;
    uint32_t *val = o->val + (size_t)key * W; float *cnt = o->cnt + (size_t)key * W; int32_t *dic = o->dic + (size_t)key * W;
    for (int i = 0; i < W; ++i)
        if (cnt[i] && val[i] == v) {
            cnt[i] += c;
            if (cnt[i] >= o->k && o->len > 0 && !dic[i]) { dic[i] = (int32_t)q_pop(o); id = 1; ++o->num; }
            while (i && cnt[i] > cnt[i - 1]) {
                float tc = cnt[i]; cnt[i] = cnt[i - 1]; cnt[i - 1] = tc;
                uint32_t tv = val[i]; val[i] = val[i - 1]; val[i - 1] = tv;
                int32_t td = dic[i]; dic[i] = dic[i - 1]; dic[i - 1] = td;
                --i;
            }
            return id;
        }
    for (int i = 0; i < W; ++i)
        if (cnt[i] == 0) { cnt[i] = c; val[i] = v; return 0; }
    if (!dic[3]) { cnt[3] += c; val[3] = v; }
    return 0;
}

/* :125-151 */
void cafe_oracle_ae_batch_query(const cafe_ae_t *o, const uint32_t *data, int len, int32_t *que) {
    for (int i = 0; i < len; ++i) que[i] = query(o, data[i]);
}
void cafe_oracle_ae_batch_insert_val(cafe_ae_t *o, const uint32_t *data, const float *v, int len, int32_t *ins) {
    if (o->tot > (double)(1ll * o->s * o->k * 10)) decay(o);
    for (int i = 0; i < len; ++i) ins[i] = insert(o, data[i], v ? v[i] : 1.0f);
}

/* :61-75, intended semantics (see the header) */
void cafe_oracle_ae_update(cafe_ae_t *o) {
    o->head = 0; o->len = 0;
    uint8_t *used = calloc((size_t)(o->lim > 0 ? o->lim : 1), 1);
    for (size_t j = 0; j < (size_t)o->s * W; ++j)
        if (o->dic[j] > 0 && o->dic[j] < o->lim) used[o->dic[j]] = 1;
    for (int i = 1; i < o->lim; ++i)
        if (!used[i]) q_push(o, (uint32_t)i);
    free(used);
}

/* state access for the parity tests */
void cafe_oracle_ae_dims(const cafe_ae_t *o, int64_t *out /* s, fifo_len, num, decays */) {
    out[0] = o->s; out[1] = o->len; out[2] = o->num; out[3] = o->decays;
}
double cafe_oracle_ae_tot(const cafe_ae_t *o) { return o->tot; }
void cafe_oracle_ae_dump(const cafe_ae_t *o, uint32_t *val, float *cnt, int32_t *dic, uint32_t *fifo) {
    size_t n = (size_t)o->s * W;
    /* val of an empty slot is undefined in the reference: reported as 0 */
    for (size_t j = 0; j < n; ++j) { val[j] = o->cnt[j] != 0 ? o->val[j] : 0; cnt[j] = o->cnt[j]; dic[j] = o->dic[j]; }
    for (long i = 0; i < o->len; ++i) fifo[i] = o->fifo[(o->head + i) % o->cap];
}
