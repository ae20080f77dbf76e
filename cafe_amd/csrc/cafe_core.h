// cafe_core.h -- HBM layout of the HotSketch and the per-bucket transition functions shared by the
// in-order kernel and the parallel (bucket-owner) kernels.
//
// Behaviour follows the reference's tricks/sketch.cpp (cited per function); the code is organised
// differently: a bucket is loaded into registers, a transition function mutates that register copy,
// and everything that is global and order dependent in the reference (free-list pops and pushes, the
// LRU clock, the real_n / num counters) goes through a context object `C`, so that the in-order
// kernel can apply it directly while the parallel kernels record placeholders that a later
// prefix-sum pass resolves in batch order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cafe {

constexpr int kSlots = 4;               // sketch.cpp:7   (m1 = m2 = 4)
constexpr double kDecayV = 10000.0;     // sketch.cpp:8
constexpr long long kHashMul = 1000000007LL;   // sketch.cpp:39,42

// ---- HBM layout --------------------------------------------------------------------------------
// Main bucket: 64 B, 64-B aligned, so a probe is one aligned half cache line; val|dic are the first
// 32 B (all a query needs), scores follow.  The reference's AoS bucket is 52 B (sketch.cpp:23-28).
struct __attribute__((aligned(64))) MainBucket {
    int32_t val[kSlots];
    int32_t dic[kSlots];
    float cnt[kSlots];
    uint32_t flip;
    uint32_t pad[3];
};
// LRU-filter bucket (sketch.cpp:29-35): 128 B, one cache line.
struct __attribute__((aligned(128))) LruBucket {
    int32_t val[kSlots];
    int32_t dic[kSlots];
    float cnt[kSlots];
    int32_t t[kSlots];
    uint32_t flip;
    uint32_t pad[15];
};
static_assert(sizeof(MainBucket) == 64, "layout");
static_assert(sizeof(LruBucket) == 128, "layout");

// Device-resident scalars of one sketch (sketch.cpp:16-19 members + free-list ring indices).
struct DevState {
    float thr;
    int32_t num, real_n, clk;
    uint32_t fifo_head, fifo_len;
    int32_t flip, resets;
    // scratch written by kernels of the current call
    int32_t maxpart, fallback, rounds;               // maxpart: largest partition of the last parallel insert
    int32_t cadence_miss;    // cumulative: call boundaries at which sketch.cpp:291 would have reset but the host had not
    int32_t pad[4];          // [0] cumulative: batches given to the in-order kernel, [1] rounds in total, [2] largest round
                             // count, [3] largest hot-bypass partition of the last parallel insert
    int32_t bad_commit;      // cumulative: staged entries the commit kernel refused (index out of range) -- always 0
    int32_t why[3];          // cumulative, batches given to the in-order kernel (or rejected): [0] a workspace limit,
                             // [1] the rounds that were enqueued did not settle it, [2] the free list changed state
    uint32_t rejected_seq;   // call number of the last insert that was rejected (0: none).  The kernels queued behind an
                             // insert whose verdict the host has not seen compare it with their own call number; unlike
                             // `fallback` it is never cleared, so a plan made ahead for the NEXT call cannot hide a rejection
    uint32_t pad2[3];
};

// What the kernel that closes an insert leaves in pinned host memory: the host reads it lagging, without a sync.
constexpr int kMaxSeg = 64;
struct Mirror {
    DevState st;
    uint32_t seq;       // sequence number of the insert call this snapshot closes
    uint32_t n;         // ids of that call
    uint32_t cross;     // threshold crossings (real_n growth) of that call
    uint32_t pops;      // free-list pops of that call
    uint32_t rejected;  // 1: a call issued "to be verified" was not applied (tables and scalars untouched)
    uint32_t reason;    // why it was rejected: 1 a workspace limit, 2 rounds did not settle, 3 free list changed state,
                        // 4 sketch.cpp:291 fires on an inner boundary (the step has to be issued call by call)
    uint32_t moves;     // moves out of the LRU table in round 1
    uint32_t got;       // newly hot ids (ins[] != 0)
};

__host__ __device__ inline int hash_main(uint32_t x, int n1) { return (int)((long long)x * kHashMul % n1); }
__host__ __device__ inline int hash_lru(uint32_t x, int n2) { return (int)((long long)x * kHashMul % n2); }

// Exact x * 1000000007 mod n without a 64-bit division: Barrett reduction with m = floor((2^64-1)/n).
// For p < 2^62 the quotient estimate is at most 2 short, so two conditional subtractions finish it.
struct FastMod {
    uint64_t m;
    uint32_t n;
};
inline FastMod make_fastmod(int n) {
    FastMod f;
    f.n = (uint32_t)n;
    f.m = n > 0 ? 0xFFFFFFFFFFFFFFFFull / (uint64_t)n : 0;
    return f;
}
__device__ inline int fast_hash(uint32_t x, FastMod f) {
    uint64_t p = (uint64_t)x * (uint64_t)kHashMul;
    uint64_t q = __umul64hi(p, f.m);
    uint64_t r = p - q * f.n;
    if (r >= f.n) r -= f.n;
    if (r >= f.n) r -= f.n;
    return (int)r;
}

// sketch.cpp:130-140: float score / double constant, rounded back to float.
__device__ inline float decay_once(float c) { return (float)((double)c / kDecayV); }

// ---- register form of a bucket ---------------------------------------------------------------------
// Named scalar members on purpose: a bucket copy indexed with a runtime slot number is placed in scratch
// memory by hipcc (seen in the ISA: scratch_load/store with a VGPR index), which turns every step of a
// bucket owner's replay loop into memory round trips.  Slot access goes through select chains instead.
struct MainR {
    int32_t v0, v1, v2, v3;     // val
    int32_t d0, d1, d2, d3;     // dic
    float c0, c1, c2, c3;       // cnt
    uint32_t flip;
};
struct LruR {
    int32_t v0, v1, v2, v3;
    int32_t d0, d1, d2, d3;
    float c0, c1, c2, c3;
    int32_t t0, t1, t2, t3;
    uint32_t flip;
};

__device__ inline int32_t sel4(int32_t a, int32_t b, int32_t c, int32_t d, int i) { return i == 0 ? a : i == 1 ? b : i == 2 ? c : d; }
__device__ inline float sel4f(float a, float b, float c, float d, int i) { return i == 0 ? a : i == 1 ? b : i == 2 ? c : d; }
// value selects, not conditional stores: hipcc merges conditional stores to neighbouring members into one
// store through a selected address, which again forces the bucket copy into scratch memory
#define CAFE_SET4(a0, a1, a2, a3, i, v)                                    \
    do {                                                                   \
        const int i_ = (i);                                                \
        const auto v_ = (v);                                               \
        a0 = i_ == 0 ? v_ : a0; a1 = i_ == 1 ? v_ : a1;                    \
        a2 = i_ == 2 ? v_ : a2; a3 = i_ == 3 ? v_ : a3;                    \
    } while (0)

__device__ inline void load_main(MainR &b, const MainBucket *src) {
    const int4 *p = reinterpret_cast<const int4 *>(src);
    const int4 v = p[0], d = p[1], c = p[2];
    b.v0 = v.x; b.v1 = v.y; b.v2 = v.z; b.v3 = v.w;
    b.d0 = d.x; b.d1 = d.y; b.d2 = d.z; b.d3 = d.w;
    b.c0 = __int_as_float(c.x); b.c1 = __int_as_float(c.y); b.c2 = __int_as_float(c.z); b.c3 = __int_as_float(c.w);
    b.flip = src->flip;
}
// `tag` goes to pad[0]: unused in the tables, the bucket index in the staging copies of the parallel insert
// `keep` goes to pad[1] (staging copies only): mask of the slots whose score the commit must leave alone
__device__ inline void store_main(MainBucket *dst, const MainR &b, int32_t tag = 0, int32_t keep = 0) {
    int4 *p = reinterpret_cast<int4 *>(dst);
    p[0] = make_int4(b.v0, b.v1, b.v2, b.v3);
    p[1] = make_int4(b.d0, b.d1, b.d2, b.d3);
    p[2] = make_int4(__float_as_int(b.c0), __float_as_int(b.c1), __float_as_int(b.c2), __float_as_int(b.c3));
    p[3] = make_int4((int)b.flip, tag, keep, 0);
}
// slots that sit hot in a bucket whose lazy decay is not pending: their items went through the hot bypass of the
// parallel insert, which owns their scores
__device__ inline int32_t hot_keep_mask(const MainR &b, uint32_t gflip) {
    if (b.flip != gflip) return 0;
    return (b.d0 ? 1 : 0) | (b.d1 ? 2 : 0) | (b.d2 ? 4 : 0) | (b.d3 ? 8 : 0);
}
__device__ inline void load_lru(LruR &b, const LruBucket *src) {
    const int4 *p = reinterpret_cast<const int4 *>(src);
    const int4 v = p[0], d = p[1], c = p[2], t = p[3];
    b.v0 = v.x; b.v1 = v.y; b.v2 = v.z; b.v3 = v.w;
    b.d0 = d.x; b.d1 = d.y; b.d2 = d.z; b.d3 = d.w;
    b.c0 = __int_as_float(c.x); b.c1 = __int_as_float(c.y); b.c2 = __int_as_float(c.z); b.c3 = __int_as_float(c.w);
    b.t0 = t.x; b.t1 = t.y; b.t2 = t.z; b.t3 = t.w;
    b.flip = src->flip;
}
__device__ inline void store_lru(LruBucket *dst, const LruR &b, int32_t tag = 0) {
    int4 *p = reinterpret_cast<int4 *>(dst);
    p[0] = make_int4(b.v0, b.v1, b.v2, b.v3);
    p[1] = make_int4(b.d0, b.d1, b.d2, b.d3);
    p[2] = make_int4(__float_as_int(b.c0), __float_as_int(b.c1), __float_as_int(b.c2), __float_as_int(b.c3));
    p[3] = make_int4(b.t0, b.t1, b.t2, b.t3);
    p[4] = make_int4((int)b.flip, tag, 0, 0);
}

template <class B>
__device__ inline void sync_flip(B &b, uint32_t gflip) {
    if (b.flip != gflip) {
        b.flip = gflip;
        b.c0 = decay_once(b.c0); b.c1 = decay_once(b.c1); b.c2 = decay_once(b.c2); b.c3 = decay_once(b.c3);
    }
}

// A move of an entry from the LRU table into the main table (the reference's Insert(), sketch.cpp:232-259),
// requested by the LRU transition and carried out on the target main bucket by whoever owns it.
struct MoveReq {
    int32_t kind;       // 0 none, 1 promotion of the inserted id itself, 2 eviction of another id
    uint32_t x;
    float cnt;
    int32_t dic;
};

// first slot whose id equals x, or -1 (sketch.cpp:214-215, 120-121: the scan stops at the first match)
template <class B>
__device__ inline int find_slot(const B &b, uint32_t x) {
    int hit = -1;
    if ((uint32_t)b.v3 == x) hit = 3;
    if ((uint32_t)b.v2 == x) hit = 2;
    if ((uint32_t)b.v1 == x) hit = 1;
    if ((uint32_t)b.v0 == x) hit = 0;
    return hit;
}

// sketch.cpp:211-229, the main-table half of insert().  Returns -1 on a miss, else the value of ins[i].
template <class C>
__device__ inline int main_probe(MainR &b, uint32_t x, float c, float thr, C &ctx) {
    const int i = find_slot(b, x);
    if (i < 0) return -1;
    const float now = sel4f(b.c0, b.c1, b.c2, b.c3, i) + c;
    int32_t dic = sel4(b.d0, b.d1, b.d2, b.d3, i);
    int got = 0;
    if (now >= thr && now - c < thr) ctx.crossed();
    if (now >= thr && !dic && ctx.can_pop()) { dic = ctx.pop(); got = 1; }
    CAFE_SET4(b.c0, b.c1, b.c2, b.c3, i, now);
    CAFE_SET4(b.d0, b.d1, b.d2, b.d3, i, dic);
    return got;
}

// sketch.cpp:232-259 Insert(): returns 1 if the entry was placed, 0 if it was dropped.
template <class C>
__device__ inline int main_move(MainR &b, uint32_t x, float c, int32_t dic, C &ctx) {
    // the reference scans slots in order: the first empty slot takes the entry; the running minimum only
    // matters if no slot is empty (strict <, first index wins ties; start value 1e9)
    int empty = -1;
    if (b.v3 == 0) empty = 3;
    if (b.v2 == 0) empty = 2;
    if (b.v1 == 0) empty = 1;
    if (b.v0 == 0) empty = 0;
    if (empty >= 0) {
        CAFE_SET4(b.v0, b.v1, b.v2, b.v3, empty, (int32_t)x);
        CAFE_SET4(b.c0, b.c1, b.c2, b.c3, empty, c);
        CAFE_SET4(b.d0, b.d1, b.d2, b.d3, empty, dic);
        return 1;
    }
    int lo = -1;
    float lo_cnt = 1e9f;
    if (b.c0 < lo_cnt) { lo_cnt = b.c0; lo = 0; }
    if (b.c1 < lo_cnt) { lo_cnt = b.c1; lo = 1; }
    if (b.c2 < lo_cnt) { lo_cnt = b.c2; lo = 2; }
    if (b.c3 < lo_cnt) { lo_cnt = b.c3; lo = 3; }
    if (lo < 0) { lo = 0; lo_cnt = b.c0; }     // all scores >= 1e9: out of bounds in the reference; fenced like the oracle
    if (!sel4(b.d0, b.d1, b.d2, b.d3, lo)) {
        CAFE_SET4(b.c0, b.c1, b.c2, b.c3, lo, lo_cnt + c);
        CAFE_SET4(b.v0, b.v1, b.v2, b.v3, lo, (int32_t)x);
        CAFE_SET4(b.d0, b.d1, b.d2, b.d3, lo, dic);
        return 1;
    }
    if (dic != 0) ctx.push(dic);
    return 0;
}

// sketch.cpp:161-210 insertLRU(), split in the order the reference executes it so that the caller can
// carry out the call into the main table (Insert(), a different bucket) at the reference's point in time:
//   hit : lru_hit()  [tick, add, cross, pop]  -> caller performs mv (kind 1) -> ins = got && moved
//   miss: lru_evict() fills mv (kind 2)       -> caller performs mv          -> lru_place() [tick, pop]

// Returns the hit slot (0..3) or -(victim+1): victim = first slot with the smallest timestamp (sketch.cpp:190-191).
__device__ inline int lru_find(const LruR &b, uint32_t x) {
    const int hit = find_slot(b, x);
    if (hit >= 0) return hit;
    int victim = -1, t_min = 1000000000;
    if (t_min > b.t0) { t_min = b.t0; victim = 0; }
    if (t_min > b.t1) { t_min = b.t1; victim = 1; }
    if (t_min > b.t2) { t_min = b.t2; victim = 2; }
    if (t_min > b.t3) { t_min = b.t3; victim = 3; }
    if (victim < 0) victim = 0;
    return -(victim + 1);
}

// sketch.cpp:165-188.  Returns the provisional ins[i]; cleared by the caller if a kind-1 move returns 0.
template <class C>
__device__ inline int lru_hit(LruR &b, int i, float c, float thr, C &ctx, MoveReq &mv) {
    int got = 0;
    mv.kind = 0;
    const int32_t t = ctx.tick();
    const float now = sel4f(b.c0, b.c1, b.c2, b.c3, i) + c;
    int32_t dic = sel4(b.d0, b.d1, b.d2, b.d3, i);
    int32_t val = sel4(b.v0, b.v1, b.v2, b.v3, i);
    float keep = now;
    int32_t tt = t;
    if (now >= thr && now - c < thr) ctx.crossed();
    if (now >= thr && ctx.can_pop() && !dic) { dic = ctx.pop(); got = 1; }
    if (now >= thr) {
        mv.kind = 1; mv.x = (uint32_t)val; mv.cnt = now; mv.dic = dic;
        val = 0; tt = 0; keep = 0.0f; dic = 0;
    }
    CAFE_SET4(b.v0, b.v1, b.v2, b.v3, i, val);
    CAFE_SET4(b.t0, b.t1, b.t2, b.t3, i, tt);
    CAFE_SET4(b.c0, b.c1, b.c2, b.c3, i, keep);
    CAFE_SET4(b.d0, b.d1, b.d2, b.d3, i, dic);
    return got;
}

// sketch.cpp:193-195: the evicted entry is moved to the main table if it is hot or has score >= 5.
__device__ inline void lru_evict(const LruR &b, int victim, MoveReq &mv) {
    mv.kind = 0;
    const int32_t dic = sel4(b.d0, b.d1, b.d2, b.d3, victim);
    const float cnt = sel4f(b.c0, b.c1, b.c2, b.c3, victim);
    if (dic || cnt >= 5.0f) {
        mv.kind = 2; mv.x = (uint32_t)sel4(b.v0, b.v1, b.v2, b.v3, victim); mv.cnt = cnt; mv.dic = dic;
    }
}

// sketch.cpp:196-209.  Returns ins[i].
template <class C>
__device__ inline int lru_place(LruR &b, int victim, uint32_t x, float c, float thr, C &ctx) {
    int got = 0;
    const int32_t t = ctx.tick();
    int32_t dic = 0;
    if (1.0f >= thr && 1.0f - c < thr) ctx.crossed();
    if (1.0f >= thr && ctx.can_pop()) { dic = ctx.pop(); got = 1; }
    CAFE_SET4(b.v0, b.v1, b.v2, b.v3, victim, (int32_t)x);
    CAFE_SET4(b.t0, b.t1, b.t2, b.t3, victim, t);
    CAFE_SET4(b.c0, b.c1, b.c2, b.c3, victim, 1.0f);          // not c: sketch.cpp:198
    CAFE_SET4(b.d0, b.d1, b.d2, b.d3, victim, dic);
    return got;
}

// ---- the batch in order, one lane (sketch.cpp:292-294 with insert() :211-231 inlined) -----------------------------
struct SerialCtx {
    uint32_t *fifo;
    uint32_t cap, head, len;
    int32_t num, real_n, clk;
    __device__ bool can_pop() const { return len > 0; }
    __device__ int32_t pop() {
        uint32_t v = fifo[head];
        head = head + 1 == cap ? 0 : head + 1;
        --len;
        ++num;
        return (int32_t)v;
    }
    __device__ void push(int32_t d) {
        uint32_t pos = head + len;
        if (pos >= cap) pos -= cap;
        fifo[pos] = (uint32_t)d;
        ++len;
    }
    __device__ int32_t tick() { return ++clk; }
    __device__ void crossed() { ++real_n; }
};

// score of batch position i: vals[i] (1 when vals == nullptr, batch_insert), scaled by seg / fsum[i / seg] when the
// caller passes per-field sums (sk_embedding_bag.py:189: gn * N / sum(gn)).  The same expression everywhere a score is
// formed, so that every kernel sees the same float.
struct ScoreSrc {
    const float *vals;
    const float *fsum;      // per-field sums (global or LDS), or nullptr
    int seg;
    int stride = 1;         // floats between two scores (a score column inside packed rows: cafe_backward_update_gid_dev)
    __device__ float at(uint32_t i) const {
        float c = vals ? vals[(size_t)i * (size_t)stride] : 1.0f;
        if (fsum) c = c * (float)seg / fsum[i / (uint32_t)seg];
        return c;
    }
};

// Sum of one field's partial sums (<= 64 of them), formed by the 64 lanes of a wave in a fixed tree order: every kernel
// that needs a field's sum forms it this way, so they all see the same float.  All lanes of the wave must call it.
constexpr int kMaxNormParts = 64;
__device__ inline float field_sum_tree(const float *__restrict__ psum, int nparts, int f, bool live) {
    const int j = threadIdx.x & 63;
    float v = (live && j < nparts) ? psum[(size_t)f * nparts + j] : 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ inline int serial_insert_range(MainBucket *mt, LruBucket *lt, int n1, int n2, float thr, uint32_t gflip, SerialCtx &ctx,
                                          const int32_t *__restrict__ ids, const ScoreSrc &sc, int n, int32_t *__restrict__ ins,
                                          int seg, double reset_at) {
    int missed = 0;
    for (int i = 0; i < n; ++i) {
        if (seg > 0 && i > 0 && i % seg == 0 && (double)ctx.real_n > reset_at) missed = 1;
        uint32_t x = (uint32_t)ids[i];
        float c = sc.at((uint32_t)i);
        int k = hash_main(x, n1);
        MainR b;
        load_main(b, &mt[k]);
        sync_flip(b, gflip);
        int r = main_probe(b, x, c, thr, ctx);
        store_main(&mt[k], b);           // also persists the lazy decay on a miss (sketch.cpp:213)
        if (r >= 0) { ins[i] = r; continue; }
        int k2 = hash_lru(x, n2);
        LruR l;
        load_lru(l, &lt[k2]);
        sync_flip(l, gflip);
        MoveReq mv;
        int f = lru_find(l, x), got;
        if (f >= 0) {
            got = lru_hit(l, f, c, thr, ctx, mv);
        } else {
            lru_evict(l, -f - 1, mv);
        }
        if (mv.kind) {
            int km = hash_main(mv.x, n1);
            MainR m;
            load_main(m, &mt[km]);
            sync_flip(m, gflip);
            int ok = main_move(m, mv.x, mv.cnt, mv.dic, ctx);
            store_main(&mt[km], m);
            if (mv.kind == 1 && !ok) got = 0;
        }
        if (f < 0) got = lru_place(l, -f - 1, x, c, thr, ctx);
        store_lru(&lt[k2], l);
        ins[i] = got;
    }
    return missed;
}

// sketch.cpp:118-128
__device__ inline int32_t query_bucket(const int32_t val[kSlots], const int32_t dic[kSlots], uint32_t x) {
#pragma unroll
    for (int i = 0; i < kSlots; ++i)
        if ((uint32_t)val[i] == x && dic[i]) return -dic[i];
    return (int32_t)x;
}

// Median tier of tricks/sketch_ml_embed.py (:66-67, :82-84, :148-152) -- a definition, the reference has no such
// function in any sketch.cpp: -x when query() does not report x hot and the main table holds it without a hot slot
// at a score (pending lazy decay applied, read-only) of at least 0.2 * Threshold; +x otherwise.
__device__ inline int32_t median_bucket(const int32_t val[kSlots], const int32_t dic[kSlots], const float cnt[kSlots],
                                        uint32_t bflip, int gflip, float thr, uint32_t x) {
#pragma unroll
    for (int i = 0; i < kSlots; ++i)
        if ((uint32_t)val[i] == x && dic[i]) return (int32_t)x;           // hot
#pragma unroll
    for (int i = 0; i < kSlots; ++i)
        if ((uint32_t)val[i] == x && !dic[i]) {
            const float c = (bflip & 1u) != (uint32_t)(gflip & 1) ? decay_once(cnt[i]) : cnt[i];
            return (double)c >= 0.2 * (double)thr ? -(int32_t)x : (int32_t)x;
        }
    return (int32_t)x;
}

// 16-byte vector store with sc1: the bytes are written through, no dirty line stays behind in the XCD's L2.  For the
// bulk outputs of the bandwidth kernels, which this kernel does not read again: the write-back at the kernel's end
// (MI355X_MICROARCH.md "boundary": + dirty bytes / 6 TB/s) has nothing left to do.
__device__ inline void store_through16(void *p, float4 v) {
#ifdef CAFE_PLAIN_STORES
    *reinterpret_cast<float4 *>(p) = v;
#else
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(x) : "memory");
#endif
}

// non-temporal form
__device__ inline void store_nt16(void *p, float4 v) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(x) : "memory");
}

__device__ inline float4 load_nt16(const void *p) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f x = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return make_float4(x.x, x.y, x.z, x.w);
}

}  // namespace cafe

// Diagnostic build (make EXTRA=-DCAFE_PROF): thread 0 of every workgroup of a numbered kernel stamps the 100 MHz wall
// clock when it enters and when it leaves; tools/step_windows.py turns the stamps into the timeline of one step as the
// device ran it, without a profiler attached.  No code in the normal build.
#ifdef CAFE_PROF
constexpr int kWinKernels = 8, kWinBlocks = 4096;
#define CAFE_KWIN_DECL(reader)                                                                                  \
    __device__ unsigned long long g_kwin[kWinKernels][kWinBlocks][2];                                           \
    extern "C" int reader(unsigned long long *out) {                                                            \
        return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kwin), sizeof(g_kwin)) != hipSuccess;                      \
    }
struct KWinScope {
    unsigned long long *p;
    __device__ KWinScope(unsigned long long (*w)[kWinBlocks][2], int id) {
        p = threadIdx.x == 0 && threadIdx.y == 0 ? w[id][(blockIdx.x + blockIdx.y * gridDim.x) & (kWinBlocks - 1)] : nullptr;
        if (p) p[0] = wall_clock64();
    }
    __device__ ~KWinScope() { if (p) p[1] = wall_clock64(); }
};
#define CAFE_KWIN(id) KWinScope kwin_scope_(g_kwin, id)
#else
#define CAFE_KWIN_DECL(reader)
#define CAFE_KWIN(id) do { } while (0)
#endif
