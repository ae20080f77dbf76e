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
    int32_t need_reset, fallback, rounds, converged;
    int32_t pad[4];
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

template <class B>
__device__ inline void sync_flip(B &b, uint32_t gflip) {
    if (b.flip != gflip) {
        b.flip = gflip;
#pragma unroll
        for (int i = 0; i < kSlots; ++i) b.cnt[i] = decay_once(b.cnt[i]);
    }
}

// A move of an entry from the LRU table into the main table (the reference's Insert(), sketch.cpp:232-259),
// requested by lru_insert and carried out on the target main bucket by whoever owns it.
struct MoveReq {
    int32_t kind;       // 0 none, 1 promotion of the inserted id itself, 2 eviction of another id
    uint32_t x;
    float cnt;
    int32_t dic;
};

// sketch.cpp:211-229, the main-table half of insert().  Returns -1 on a miss, else the value of ins[i].
template <class C>
__device__ inline int main_probe(MainBucket &b, uint32_t x, float c, float thr, C &ctx) {
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
        if ((uint32_t)b.val[i] == x) {
            b.cnt[i] += c;
            float now = b.cnt[i];
            int got = 0;
            if (now >= thr && now - c < thr) ctx.crossed();
            if (now >= thr && !b.dic[i] && ctx.can_pop()) { b.dic[i] = ctx.pop(); got = 1; }
            return got;
        }
    }
    return -1;
}

// sketch.cpp:232-259 Insert(): returns 1 if the entry was placed, 0 if it was dropped.
template <class C>
__device__ inline int main_move(MainBucket &b, uint32_t x, float c, int32_t dic, C &ctx) {
    int lo = -1;
    float lo_cnt = 1e9f;
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
        if (b.val[i] == 0) { b.val[i] = (int32_t)x; b.cnt[i] = c; b.dic[i] = dic; return 1; }
        if (b.cnt[i] < lo_cnt) { lo_cnt = b.cnt[i]; lo = i; }
    }
    if (lo < 0) lo = 0;     // all scores >= 1e9: out of bounds in the reference; fenced like the oracle
    // static indexing only (no runtime-indexed register arrays)
    int hot = 0;
#pragma unroll
    for (int i = 0; i < kSlots; ++i) if (i == lo) hot = b.dic[i];
    if (!hot) {
#pragma unroll
        for (int i = 0; i < kSlots; ++i) if (i == lo) { b.cnt[i] += c; b.val[i] = (int32_t)x; b.dic[i] = dic; }
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
__device__ inline int lru_find(const LruBucket &b, uint32_t x) {
    int victim = -1, t_min = 1000000000;
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
        if ((uint32_t)b.val[i] == x) return i;
        if (t_min > b.t[i]) { t_min = b.t[i]; victim = i; }
    }
    if (victim < 0) victim = 0;
    return -(victim + 1);
}

// sketch.cpp:165-188.  Returns the provisional ins[i]; cleared by the caller if a kind-1 move returns 0.
template <class C>
__device__ inline int lru_hit(LruBucket &b, int slot, float c, float thr, C &ctx, MoveReq &mv) {
    int got = 0;
    mv.kind = 0;
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
        if (i == slot) {
            b.t[i] = ctx.tick();
            b.cnt[i] += c;
            float now = b.cnt[i];
            if (now >= thr && now - c < thr) ctx.crossed();
            if (now >= thr && ctx.can_pop() && !b.dic[i]) { b.dic[i] = ctx.pop(); got = 1; }
            if (now >= thr) {
                mv.kind = 1; mv.x = (uint32_t)b.val[i]; mv.cnt = now; mv.dic = b.dic[i];
                b.val[i] = 0; b.t[i] = 0; b.cnt[i] = 0; b.dic[i] = 0;
            }
        }
    }
    return got;
}

// sketch.cpp:193-195: the evicted entry is moved to the main table if it is hot or has score >= 5.
__device__ inline void lru_evict(const LruBucket &b, int victim, MoveReq &mv) {
    mv.kind = 0;
#pragma unroll
    for (int i = 0; i < kSlots; ++i)
        if (i == victim && (b.dic[i] || b.cnt[i] >= 5.0f)) {
            mv.kind = 2; mv.x = (uint32_t)b.val[i]; mv.cnt = b.cnt[i]; mv.dic = b.dic[i];
        }
}

// sketch.cpp:196-209.  Returns ins[i].
template <class C>
__device__ inline int lru_place(LruBucket &b, int victim, uint32_t x, float c, float thr, C &ctx) {
    int got = 0;
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
        if (i == victim) {
            b.val[i] = (int32_t)x;
            b.t[i] = ctx.tick();
            b.cnt[i] = 1.0f;                  // not c: sketch.cpp:198
            b.dic[i] = 0;
            if (1.0f >= thr && 1.0f - c < thr) ctx.crossed();
            if (1.0f >= thr && ctx.can_pop()) { b.dic[i] = ctx.pop(); got = 1; }
        }
    }
    return got;
}

// sketch.cpp:118-128
__device__ inline int32_t query_bucket(const int32_t val[kSlots], const int32_t dic[kSlots], uint32_t x) {
#pragma unroll
    for (int i = 0; i < kSlots; ++i)
        if ((uint32_t)val[i] == x && dic[i]) return -dic[i];
    return (int32_t)x;
}

}  // namespace cafe
