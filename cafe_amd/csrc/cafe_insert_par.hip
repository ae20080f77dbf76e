// cafe_insert_par.hip -- parallel, bit-exact batch insert for the HotSketch (sketch.cpp:281-296).
//
// The reference inserts a batch strictly in order; items interact through (a) their main bucket,
// (b) their LRU bucket, (c) the main bucket an entry leaving the LRU table is moved into, and through
// the global free list, LRU clock and counters.  This file reproduces that order exactly without
// walking the batch serially:
//
//  * Ownership.  Buckets are grouped in contiguous partitions; one workgroup owns a partition, sorts
//    the partition's items by (bucket, batch position) in LDS and lets one lane replay each bucket's
//    items in batch order on a register copy of the bucket (the transition functions of cafe_core.h).
//  * Fix point.  Main-bucket owners need the moves that LRU owners emit (time-stamped with the batch
//    position of the item that caused them); LRU owners need to know which items missed the main table.
//    Round r runs all main owners with the moves of round r-1, then all LRU owners with the misses of
//    round r, always from the untouched pre-batch tables.  An item's outcome depends only on earlier
//    items, so rounds can only correct later and later positions; when a round reproduces the previous
//    round's miss set the result is the in-order result.  Typical batches settle in 2-3 rounds.
//  * Placeholders.  Free-list pops, the LRU clock and pushes are global and ordered by batch position.
//    Owners store placeholders (the popping item's position; clock0+1+position) whose zero/non-zero-ness
//    and relative order are all the transitions ever test; prefix sums over the batch then turn them
//    into the exact slot numbers / timestamps when the staged buckets are committed to the tables.
//  * The free list must not change between "can pop" and "cannot pop" inside the batch.  With a
//    non-empty list that holds iff the batch pops no more than the list held; with an empty list iff
//    nothing is pushed.  Both are checked before anything is committed; a batch that fails the check
//    (or overflows a workspace, or does not settle in max_rounds) is left untouched and handed to the
//    in-order kernel.  Nothing is written to the tables before that decision.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "cafe_internal.h"

namespace cafe {

constexpr int kThreads = 256;
constexpr int kCap = 4096;        // items (probes + moves) a main owner sorts in LDS; items of an LRU owner
constexpr int kCapEv = 512;       // moves one main partition can receive per round
constexpr int kMaxRounds = 16;
constexpr int kScanBlock = 1024;

constexpr uint32_t kSur = 0x80000000u;      // placeholder for "slot popped by item (v & 0x7fffffff)"

// outcome bits per item
constexpr uint8_t oGot = 1, oPop = 2, oCross = 4, oMiss = 8;

struct MoveEv {             // a move into the main table emitted by an LRU owner
    uint32_t time;          // 2 * position + 1: after the item's own probe
    uint32_t x;
    float cnt;
    int32_t dic;
    int32_t kind;
    int32_t pad[3];
};

struct EvRes {              // outcome of a move, written by the main owner that carried it out
    unsigned long long tag; // (call sequence << 8) | round
    int32_t kind, ok, pushed, pad;
};

struct Ctl {                // zeroed at the start of every call
    uint32_t count_m[2], count_l[2];
    uint32_t changed[kMaxRounds + 2];
    uint32_t overflow;
    uint32_t final_round;
    unsigned long long tot_cross;
    uint32_t tot_pop, tot_push, tot_clk, pad;
};

struct ParWorkspace {
    DevBuf ctl, cnt1, off1, rank1, items1, lcnt, linbox, evcnt, ev, evres, miss, outm, outl, stage_m, stage_l;
    DevBuf flags, blocksum, ranks;
    void *evres_seen = nullptr;
    int P1 = 0, P2 = 0;
    unsigned long long call_seq = 0;
};

struct Geo {
    int n, n1, n2, P1, P2, bpp1, bpp2;
    FastMod fm1, fm2;
};

// ------------------------------------------------------------------------------------------------
// small block-wide helpers
// ------------------------------------------------------------------------------------------------
__device__ inline int block_excl_scan(int v, int *sh /* kThreads + 1 */, int &total) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int o = 1; o < kThreads; o <<= 1) {
        int add = t >= o ? sh[t - o] : 0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    total = sh[kThreads - 1];
    int r = sh[t] - v;
    __syncthreads();
    return r;
}

// in-LDS bitonic sort of `len` (power of two) 64-bit keys
__device__ inline void bitonic_sort(unsigned long long *key, int len) {
    for (int k = 2; k <= len; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int c = threadIdx.x; c < (len >> 1); c += kThreads) {
                int i = 2 * c - (c & (j - 1));
                int p = i ^ j;
                unsigned long long a = key[i], b = key[p];
                bool asc = (i & k) == 0;
                if ((a > b) == asc) { key[i] = b; key[p] = a; }
            }
            __syncthreads();
        }
    }
}

__device__ inline int next_pow2(int v) {
    int p = 64;
    while (p < v) p <<= 1;
    return p;
}

// ------------------------------------------------------------------------------------------------
// partition the batch by main-bucket owner
// ------------------------------------------------------------------------------------------------
__global__ void k_assign(Geo g, const int32_t *__restrict__ ids, uint32_t *__restrict__ cnt1, uint32_t *__restrict__ rank1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n) return;
    int p = fast_hash((uint32_t)ids[i], g.fm1) / g.bpp1;
    rank1[i] = atomicAdd(&cnt1[p], 1u);
}

__global__ void __launch_bounds__(1024) k_offsets(Geo g, const uint32_t *__restrict__ cnt1, uint32_t *__restrict__ off1,
                                                  Ctl *ctl, const DevState *st) {
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    uint32_t over = 0;
    for (int base = 0; base < g.P1; base += 1024) {
        int i = base + threadIdx.x;
        uint32_t v = i < g.P1 ? cnt1[i] : 0u;
        if (v > (uint32_t)kCap) over = 1;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            uint32_t add = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < g.P1) off1[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) off1[g.P1] = carry;
    if (over) ctl->overflow = 1;
    // the LRU clock placeholder clock0 + 1 + position must stay a positive int
    if (threadIdx.x == 0 && (long long)st->clk + g.n + 2 >= 0x7fffffffLL) ctl->overflow = 1;
}

__global__ void k_scatter(Geo g, const int32_t *__restrict__ ids, const uint32_t *__restrict__ off1,
                          const uint32_t *__restrict__ rank1, uint32_t *__restrict__ items1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n) return;
    int p = fast_hash((uint32_t)ids[i], g.fm1) / g.bpp1;
    items1[off1[p] + rank1[i]] = (uint32_t)i;
}

// ------------------------------------------------------------------------------------------------
// main-table owners
// ------------------------------------------------------------------------------------------------
struct MainCtx {
    bool full;
    uint32_t pos;
    uint8_t flags;
    int32_t pushed;
    __device__ bool can_pop() const { return full; }
    __device__ int32_t pop() { flags |= oPop; return (int32_t)(kSur | pos); }
    __device__ void push(int32_t d) { pushed = d; }
    __device__ int32_t tick() { return 0; }
    __device__ void crossed() { flags |= oCross; }
};

__device__ inline void load_main_r(MainBucket &b, const MainBucket *src) {
    const int4 *p = reinterpret_cast<const int4 *>(src);
    int4 v = p[0], d = p[1], c = p[2];
    b.val[0] = v.x; b.val[1] = v.y; b.val[2] = v.z; b.val[3] = v.w;
    b.dic[0] = d.x; b.dic[1] = d.y; b.dic[2] = d.z; b.dic[3] = d.w;
    b.cnt[0] = __int_as_float(c.x); b.cnt[1] = __int_as_float(c.y); b.cnt[2] = __int_as_float(c.z); b.cnt[3] = __int_as_float(c.w);
    b.flip = src->flip;
}
__device__ inline void store_main_r(MainBucket *dst, const MainBucket &b, uint32_t tag) {
    int4 *p = reinterpret_cast<int4 *>(dst);
    p[0] = make_int4(b.val[0], b.val[1], b.val[2], b.val[3]);
    p[1] = make_int4(b.dic[0], b.dic[1], b.dic[2], b.dic[3]);
    p[2] = make_int4(__float_as_int(b.cnt[0]), __float_as_int(b.cnt[1]), __float_as_int(b.cnt[2]), __float_as_int(b.cnt[3]));
    p[3] = make_int4((int)b.flip, (int)tag, 0, 0);
}
__device__ inline void load_lru_r(LruBucket &b, const LruBucket *src) {
    const int4 *p = reinterpret_cast<const int4 *>(src);
    int4 v = p[0], d = p[1], c = p[2], t = p[3];
    b.val[0] = v.x; b.val[1] = v.y; b.val[2] = v.z; b.val[3] = v.w;
    b.dic[0] = d.x; b.dic[1] = d.y; b.dic[2] = d.z; b.dic[3] = d.w;
    b.cnt[0] = __int_as_float(c.x); b.cnt[1] = __int_as_float(c.y); b.cnt[2] = __int_as_float(c.z); b.cnt[3] = __int_as_float(c.w);
    b.t[0] = t.x; b.t[1] = t.y; b.t[2] = t.z; b.t[3] = t.w;
    b.flip = src->flip;
}
__device__ inline void store_lru_r(LruBucket *dst, const LruBucket &b, uint32_t tag) {
    int4 *p = reinterpret_cast<int4 *>(dst);
    p[0] = make_int4(b.val[0], b.val[1], b.val[2], b.val[3]);
    p[1] = make_int4(b.dic[0], b.dic[1], b.dic[2], b.dic[3]);
    p[2] = make_int4(__float_as_int(b.cnt[0]), __float_as_int(b.cnt[1]), __float_as_int(b.cnt[2]), __float_as_int(b.cnt[3]));
    p[3] = make_int4(b.t[0], b.t[1], b.t[2], b.t[3]);
    p[4] = make_int4((int)b.flip, (int)tag, 0, 0);
}

// key layout of a main owner: [bucket in partition : 25][2*position + phase : 23][move index | kind<<15 : 16]
__global__ void __launch_bounds__(kThreads) k_main_round(
    Geo g, int round, unsigned long long call_tag, const MainBucket *__restrict__ mt, const DevState *__restrict__ st,
    Ctl *ctl, const int32_t *__restrict__ ids, const float *__restrict__ scores, const uint32_t *__restrict__ off1,
    const uint32_t *__restrict__ items1, uint32_t *lcnt /* [2][P2] */, uint32_t *__restrict__ linbox,
    const uint32_t *__restrict__ evcnt /* [2][P1] */, const MoveEv *__restrict__ ev, EvRes *__restrict__ evres,
    uint8_t *__restrict__ miss, uint8_t *__restrict__ outm, MainBucket *__restrict__ stage_m) {
    __shared__ unsigned long long key[kCap];
    __shared__ uint32_t px[kCap];
    __shared__ float pc[kCap];
    __shared__ int32_t pd[kCap];
    __shared__ int sh[kThreads + 1];
    __shared__ uint32_t stage_base;

    if (ctl->final_round || st->fallback || ctl->overflow) return;
    const int p = blockIdx.x, tid = threadIdx.x;
    const int par = round & 1, nxt = par ^ 1;
    // housekeeping for the next round's counters (nobody reads or writes them during this kernel)
    for (int q = p; q < g.P2; q += gridDim.x) if (tid == 0) lcnt[nxt * g.P2 + q] = 0;
    if (p == 0 && tid == 0) ctl->count_m[nxt] = 0;

    const uint32_t o0 = off1[p];
    const int np = (int)(off1[p + 1] - o0);
    const int ne = round > 1 ? (int)min(evcnt[nxt * g.P1 + p], (uint32_t)kCapEv) : 0;   // moves of round-1
    const int total = np + ne;
    if (total == 0) return;
    if (total > kCap) { if (tid == 0) ctl->overflow = 1; return; }
    const int len = next_pow2(total);
    const int b0 = p * g.bpp1;

    for (int j = tid; j < len; j += kThreads) {
        unsigned long long k = ~0ull;
        if (j < np) {
            uint32_t i = items1[o0 + j];
            uint32_t bl = (uint32_t)(fast_hash((uint32_t)ids[i], g.fm1) - b0);
            k = ((unsigned long long)bl << 39) | ((unsigned long long)(2u * i) << 16);
        } else if (j < total) {
            int e = j - np;
            const MoveEv &m = ev[(size_t)p * kCapEv + e];
            uint32_t bl = (uint32_t)(fast_hash(m.x, g.fm1) - b0);
            k = ((unsigned long long)bl << 39) | ((unsigned long long)m.time << 16) | (unsigned)(e | ((m.kind == 1) << 15));
        }
        key[j] = k;
    }
    __syncthreads();
    bitonic_sort(key, len);
    for (int s = tid; s < total; s += kThreads) {
        unsigned long long k = key[s];
        uint32_t time = (uint32_t)(k >> 16) & 0x7fffffu;
        if (time & 1u) {
            const MoveEv &m = ev[(size_t)p * kCapEv + (int)(k & 0x7fffu)];
            px[s] = m.x; pc[s] = m.cnt; pd[s] = m.dic;
        } else {
            uint32_t i = time >> 1;
            px[s] = (uint32_t)ids[i]; pc[s] = scores ? scores[i] : 1.0f; pd[s] = 0;
        }
    }
    // reserve staging room: one entry per distinct bucket
    int heads = 0;
    for (int s = tid; s < total; s += kThreads)
        heads += (s == 0 || (key[s - 1] >> 39) != (key[s] >> 39)) ? 1 : 0;
    int tot_heads;
    int my = block_excl_scan(heads, sh, tot_heads);
    if (tid == 0) stage_base = atomicAdd(&ctl->count_m[par], (uint32_t)tot_heads);
    __syncthreads();
    uint32_t slot_out = stage_base + (uint32_t)my;

    const float thr = st->thr;
    const uint32_t gflip = (uint32_t)st->flip;
    MainCtx ctx;
    ctx.full = st->fifo_len > 0;
    int changed = 0, over = 0;
    for (int s = tid; s < total; s += kThreads) {
        const uint32_t bl = (uint32_t)(key[s] >> 39);
        if (s != 0 && (uint32_t)(key[s - 1] >> 39) == bl) continue;
        MainBucket b;
        load_main_r(b, &mt[b0 + (int)bl]);
        sync_flip(b, gflip);
        for (int e = s; e < total && (uint32_t)(key[e] >> 39) == bl; ++e) {
            const unsigned long long k = key[e];
            const uint32_t time = (uint32_t)(k >> 16) & 0x7fffffu;
            const uint32_t pos = time >> 1;
            if (!(time & 1u)) {
                ctx.pos = pos; ctx.flags = 0;
                const uint32_t x = px[e];
                int r = main_probe(b, x, pc[e], thr, ctx);
                const bool m = r < 0;
                outm[pos] = (uint8_t)(ctx.flags | (m ? oMiss : 0) | (r > 0 ? oGot : 0));
                if (round == 1 ? m : (m != (miss[pos] != 0))) changed = 1;
                miss[pos] = m ? 1 : 0;
                if (m) {
                    int q = fast_hash(x, g.fm2) / g.bpp2;
                    uint32_t at = atomicAdd(&lcnt[par * g.P2 + q], 1u);
                    if (at < (uint32_t)kCap) linbox[(size_t)q * kCap + at] = pos; else over = 1;
                }
            } else {
                ctx.pushed = 0;
                int ok = main_move(b, px[e], pc[e], pd[e], ctx);
                EvRes r;
                r.tag = call_tag | (unsigned)round; r.kind = (int)((k >> 15) & 1u) ? 1 : 2; r.ok = ok; r.pushed = ctx.pushed; r.pad = 0;
                evres[pos] = r;
            }
        }
        store_main_r(&stage_m[slot_out++], b, (uint32_t)(b0 + (int)bl));
    }
    if (changed) atomicOr(&ctl->changed[round], 1u);
    if (over) ctl->overflow = 1;
}

// ------------------------------------------------------------------------------------------------
// LRU-table owners
// ------------------------------------------------------------------------------------------------
struct LruCtx {
    bool full;
    uint32_t pos;
    int32_t clk0;
    uint8_t flags;
    __device__ bool can_pop() const { return full; }
    __device__ int32_t pop() { flags |= oPop; return (int32_t)(kSur | pos); }
    __device__ void push(int32_t) {}
    __device__ int32_t tick() { return clk0 + 1 + (int32_t)pos; }
    __device__ void crossed() { flags |= oCross; }
};

__global__ void __launch_bounds__(kThreads) k_lru_round(
    Geo g, int round, const LruBucket *__restrict__ lt, DevState *st, Ctl *ctl, const int32_t *__restrict__ ids,
    const float *__restrict__ scores, const uint32_t *__restrict__ lcnt, const uint32_t *__restrict__ linbox,
    uint32_t *evcnt, MoveEv *__restrict__ ev, uint8_t *__restrict__ outl, LruBucket *__restrict__ stage_l) {
    __shared__ unsigned long long key[kCap];
    __shared__ uint32_t px[kCap];
    __shared__ float pc[kCap];
    __shared__ int sh[kThreads + 1];
    __shared__ uint32_t stage_base;

    if (ctl->final_round || st->fallback || ctl->overflow) return;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int par = round & 1, nxt = par ^ 1;
    if (!ctl->changed[round]) {                    // the miss set repeated: the previous round is the answer
        if (q == 0 && tid == 0) { ctl->final_round = (uint32_t)round; }
        return;
    }
    for (int p = q; p < g.P1; p += gridDim.x) if (tid == 0) evcnt[nxt * g.P1 + p] = 0;
    if (q == 0 && tid == 0) ctl->count_l[nxt] = 0;

    const int total = (int)lcnt[par * g.P2 + q];
    if (total == 0) return;
    if (total > kCap) { if (tid == 0) ctl->overflow = 1; return; }
    const int len = next_pow2(total);
    const int b0 = q * g.bpp2;
    for (int j = tid; j < len; j += kThreads) {
        unsigned long long k = ~0ull;
        if (j < total) {
            uint32_t i = linbox[(size_t)q * kCap + j];
            uint32_t bl = (uint32_t)(fast_hash((uint32_t)ids[i], g.fm2) - b0);
            k = ((unsigned long long)bl << 32) | i;
        }
        key[j] = k;
    }
    __syncthreads();
    bitonic_sort(key, len);
    for (int s = tid; s < total; s += kThreads) {
        uint32_t i = (uint32_t)key[s];
        px[s] = (uint32_t)ids[i]; pc[s] = scores ? scores[i] : 1.0f;
    }
    int heads = 0;
    for (int s = tid; s < total; s += kThreads)
        heads += (s == 0 || (key[s - 1] >> 32) != (key[s] >> 32)) ? 1 : 0;
    int tot_heads;
    int my = block_excl_scan(heads, sh, tot_heads);
    if (tid == 0) stage_base = atomicAdd(&ctl->count_l[par], (uint32_t)tot_heads);
    __syncthreads();
    uint32_t slot_out = stage_base + (uint32_t)my;

    const float thr = st->thr;
    const uint32_t gflip = (uint32_t)st->flip;
    LruCtx ctx;
    ctx.full = st->fifo_len > 0;
    ctx.clk0 = st->clk;
    int over = 0;
    for (int s = tid; s < total; s += kThreads) {
        const uint32_t bl = (uint32_t)(key[s] >> 32);
        if (s != 0 && (uint32_t)(key[s - 1] >> 32) == bl) continue;
        LruBucket b;
        load_lru_r(b, &lt[b0 + (int)bl]);
        sync_flip(b, gflip);
        for (int e = s; e < total && (uint32_t)(key[e] >> 32) == bl; ++e) {
            const uint32_t pos = (uint32_t)key[e];
            const uint32_t x = px[e];
            const float c = pc[e];
            ctx.pos = pos; ctx.flags = 0;
            MoveReq mv;
            int f = lru_find(b, x), got;
            if (f >= 0) {
                got = lru_hit(b, f, c, thr, ctx, mv);
            } else {
                lru_evict(b, -f - 1, mv);
                got = lru_place(b, -f - 1, x, c, thr, ctx);
            }
            outl[pos] = (uint8_t)(ctx.flags | (got ? oGot : 0));
            if (mv.kind) {
                int p = fast_hash(mv.x, g.fm1) / g.bpp1;
                uint32_t at = atomicAdd(&evcnt[par * g.P1 + p], 1u);
                if (at < (uint32_t)kCapEv) {
                    MoveEv m;
                    m.time = 2u * pos + 1u; m.x = mv.x; m.cnt = mv.cnt; m.dic = mv.dic; m.kind = mv.kind;
                    m.pad[0] = m.pad[1] = m.pad[2] = 0;
                    ev[(size_t)p * kCapEv + at] = m;
                } else {
                    over = 1;
                }
            }
        }
        store_lru_r(&stage_l[slot_out++], b, (uint32_t)(b0 + (int)bl));
    }
    if (over) ctl->overflow = 1;
}

// ------------------------------------------------------------------------------------------------
// finish: batch-ordered prefix sums, checks, commit
// ------------------------------------------------------------------------------------------------
// flags per item: bit0 pop, bit1 push, bit2 clock tick
__global__ void __launch_bounds__(kScanBlock) k_finish_flags(
    Geo g, unsigned long long call_tag, const DevState *st, Ctl *ctl, const uint8_t *__restrict__ miss,
    const uint8_t *__restrict__ outm, const uint8_t *__restrict__ outl, const EvRes *__restrict__ evres,
    int32_t *__restrict__ ins, uint8_t *__restrict__ flags, int32_t *__restrict__ pushval, uint32_t *__restrict__ blocksum) {
    __shared__ uint32_t s_pop, s_push, s_clk, s_cross;
    if (threadIdx.x == 0) { s_pop = s_push = s_clk = s_cross = 0; }
    __syncthreads();
    const bool live = !(st->fallback || ctl->overflow) && ctl->final_round != 0;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t pop = 0, push = 0, clk = 0, cross = 0;
    if (live && i < g.n) {
        const uint8_t m = miss[i];
        const uint8_t o = m ? outl[i] : outm[i];
        int got = (o & oGot) ? 1 : 0;
        pop = (o & oPop) ? 1 : 0;
        cross = (o & oCross) ? 1 : 0;
        clk = m ? 1 : 0;
        int32_t pv = 0;
        if (m) {
            const EvRes r = evres[i];
            if (r.tag == (call_tag | ctl->final_round)) {
                if (r.kind == 1 && !r.ok) got = 0;           // sketch.cpp:180-182
                if (!r.ok && r.pushed != 0) { push = 1; pv = r.pushed; }
            }
        }
        ins[i] = got;
        pushval[i] = pv;
        flags[i] = (uint8_t)(pop | (push << 1) | (clk << 2));
    }
    // block totals (order inside the block is restored by the scan kernel)
    unsigned long long mp = __ballot(pop), mu = __ballot(push), mc = __ballot(clk), mx = __ballot(cross);
    if ((threadIdx.x & 63) == 0) {
        if (mp) atomicAdd(&s_pop, (uint32_t)__popcll(mp));
        if (mu) atomicAdd(&s_push, (uint32_t)__popcll(mu));
        if (mc) atomicAdd(&s_clk, (uint32_t)__popcll(mc));
        if (mx) atomicAdd(&s_cross, (uint32_t)__popcll(mx));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        blocksum[blockIdx.x * 4 + 0] = s_pop;
        blocksum[blockIdx.x * 4 + 1] = s_push;
        blocksum[blockIdx.x * 4 + 2] = s_clk;
        blocksum[blockIdx.x * 4 + 3] = s_cross;
    }
}

// one workgroup: totals, the free-list check, decision
__global__ void __launch_bounds__(1024) k_finish_decide(Geo g, int nblocks, DevState *st, Ctl *ctl, uint32_t *blocksum) {
    __shared__ unsigned long long tot[4];
    if (threadIdx.x < 4) tot[threadIdx.x] = 0;
    __syncthreads();
    // exclusive scan of the block sums, in place, by one thread per counter (few hundred blocks)
    if (threadIdx.x < 4) {
        unsigned long long run = 0;
        for (int b = 0; b < nblocks; ++b) {
            uint32_t v = blocksum[b * 4 + threadIdx.x];
            blocksum[b * 4 + threadIdx.x] = (uint32_t)run;
            run += v;
        }
        tot[threadIdx.x] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bool ok = !(st->fallback || ctl->overflow) && ctl->final_round != 0;
        if (ok) {
            if (st->fifo_len > 0) ok = tot[0] <= (unsigned long long)st->fifo_len;      // never ran dry
            else ok = tot[1] == 0;                                                       // never refilled
        }
        ctl->tot_pop = (uint32_t)tot[0]; ctl->tot_push = (uint32_t)tot[1]; ctl->tot_clk = (uint32_t)tot[2];
        ctl->tot_cross = tot[3];
        st->rounds = (int32_t)ctl->final_round;
        if (!ok) st->fallback = 1;
    }
}

// per-item exclusive ranks in batch order
__global__ void __launch_bounds__(kScanBlock) k_finish_ranks(Geo g, const DevState *st, const uint8_t *__restrict__ flags,
                                                             const uint32_t *__restrict__ blocksum, uint32_t *__restrict__ poprank,
                                                             uint32_t *__restrict__ pushrank, uint32_t *__restrict__ clkrank) {
    __shared__ uint32_t wsum[3][kScanBlock / 64];
    if (st->fallback) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint8_t f = i < g.n ? flags[i] : 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1;
    unsigned long long m0 = __ballot(f & 1), m1 = __ballot(f & 2), m2 = __ballot(f & 4);
    if (lane == 0) { wsum[0][w] = __popcll(m0); wsum[1][w] = __popcll(m1); wsum[2][w] = __popcll(m2); }
    __syncthreads();
    uint32_t o0 = blocksum[blockIdx.x * 4 + 0], o1 = blocksum[blockIdx.x * 4 + 1], o2 = blocksum[blockIdx.x * 4 + 2];
    for (int k = 0; k < w; ++k) { o0 += wsum[0][k]; o1 += wsum[1][k]; o2 += wsum[2][k]; }
    if (i < g.n) {
        poprank[i] = o0 + (uint32_t)__popcll(m0 & below);
        pushrank[i] = o1 + (uint32_t)__popcll(m1 & below);
        clkrank[i] = o2 + (uint32_t)__popcll(m2 & below);
    }
}

__device__ inline int32_t resolve_dic(int32_t d, const uint32_t *fifo, uint32_t cap, uint32_t head, const uint32_t *poprank) {
    if ((uint32_t)d & kSur) {
        uint32_t pos = (uint32_t)d & ~kSur;
        return (int32_t)fifo[(head + poprank[pos]) % cap];
    }
    return d;
}

// commit: staged buckets -> tables with placeholders resolved; pushes -> free list
__global__ void __launch_bounds__(256) k_commit(Geo g, MainBucket *mt, LruBucket *lt, uint32_t *fifo, uint32_t cap,
                                                const DevState *st, const Ctl *ctl, const MainBucket *__restrict__ stage_m,
                                                const LruBucket *__restrict__ stage_l, const uint8_t *__restrict__ flags,
                                                const int32_t *__restrict__ pushval, const uint32_t *__restrict__ poprank,
                                                const uint32_t *__restrict__ pushrank, const uint32_t *__restrict__ clkrank) {
    if (st->fallback) return;
    const uint32_t head = st->fifo_head, len0 = st->fifo_len;
    const int32_t clk0 = st->clk;
    const uint32_t fr = ctl->final_round;
    const uint32_t nm = ctl->count_m[fr & 1];
    const uint32_t nl = fr > 1 ? ctl->count_l[(fr - 1) & 1] : 0;
    const long work = (long)nm + nl + g.n;
    for (long w = (long)blockIdx.x * blockDim.x + threadIdx.x; w < work; w += (long)gridDim.x * blockDim.x) {
        if (w < nm) {
            const int4 *p = reinterpret_cast<const int4 *>(&stage_m[w]);
            int4 v = p[0], d = p[1], c = p[2], m = p[3];
            d.x = resolve_dic(d.x, fifo, cap, head, poprank); d.y = resolve_dic(d.y, fifo, cap, head, poprank);
            d.z = resolve_dic(d.z, fifo, cap, head, poprank); d.w = resolve_dic(d.w, fifo, cap, head, poprank);
            int4 *o = reinterpret_cast<int4 *>(&mt[m.y]);
            o[0] = v; o[1] = d; o[2] = c;
            mt[m.y].flip = (uint32_t)m.x;
        } else if (w < (long)nm + nl) {
            const int4 *p = reinterpret_cast<const int4 *>(&stage_l[w - nm]);
            int4 v = p[0], d = p[1], c = p[2], t = p[3], m = p[4];
            d.x = resolve_dic(d.x, fifo, cap, head, poprank); d.y = resolve_dic(d.y, fifo, cap, head, poprank);
            d.z = resolve_dic(d.z, fifo, cap, head, poprank); d.w = resolve_dic(d.w, fifo, cap, head, poprank);
            if (t.x > clk0) t.x = clk0 + 1 + (int32_t)clkrank[t.x - clk0 - 1];
            if (t.y > clk0) t.y = clk0 + 1 + (int32_t)clkrank[t.y - clk0 - 1];
            if (t.z > clk0) t.z = clk0 + 1 + (int32_t)clkrank[t.z - clk0 - 1];
            if (t.w > clk0) t.w = clk0 + 1 + (int32_t)clkrank[t.w - clk0 - 1];
            int4 *o = reinterpret_cast<int4 *>(&lt[m.y]);
            o[0] = v; o[1] = d; o[2] = c; o[3] = t;
            lt[m.y].flip = (uint32_t)m.x;
        } else {
            const long i = w - nm - nl;
            if (flags[i] & 2) {
                // pushes land behind everything the list held; pops only ever take what it held at the start
                int32_t d = resolve_dic(pushval[i], fifo, cap, head, poprank);
                fifo[(unsigned long long)((unsigned long long)head + len0 + pushrank[i]) % cap] = (uint32_t)d;
            }
        }
    }
}

__global__ void k_commit_scalars(DevState *st, const Ctl *ctl, uint32_t cap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->fallback) { st->pad[0] += 1; return; }      // cumulative: batches handed to the in-order kernel
    st->pad[1] += (int32_t)ctl->final_round;            // cumulative rounds
    if ((int32_t)ctl->final_round > st->pad[2]) st->pad[2] = (int32_t)ctl->final_round;
    st->fifo_head = (uint32_t)(((unsigned long long)st->fifo_head + ctl->tot_pop) % cap);
    st->fifo_len = st->fifo_len + ctl->tot_push - ctl->tot_pop;
    st->num += (int32_t)ctl->tot_pop;
    st->real_n += (int32_t)ctl->tot_cross;
    st->clk += (int32_t)ctl->tot_clk;
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
static int ensure_ws(cafe_sketch *s, int n, Geo &g) {
    if (!s->par) s->par = new ParWorkspace();
    ParWorkspace &w = *s->par;
    g.n = n; g.n1 = s->n1; g.n2 = s->n2;
    int want = std::max(1, (n + 511) / 512);
    g.P1 = std::min(want, s->n1);
    g.bpp1 = (s->n1 + g.P1 - 1) / g.P1;
    g.P1 = (s->n1 + g.bpp1 - 1) / g.bpp1;
    g.P2 = std::min(want, s->n2);
    g.bpp2 = (s->n2 + g.P2 - 1) / g.P2;
    g.P2 = (s->n2 + g.bpp2 - 1) / g.bpp2;
    g.fm1 = make_fastmod(s->n1);
    g.fm2 = make_fastmod(s->n2);
    size_t N = (size_t)n;
    int nblocks = (n + kScanBlock - 1) / kScanBlock;
    if (w.ctl.reserve(sizeof(Ctl)) || w.cnt1.reserve(4 * (size_t)g.P1 + 4) || w.off1.reserve(4 * (size_t)g.P1 + 8) ||
        w.rank1.reserve(4 * N) || w.items1.reserve(4 * N) || w.lcnt.reserve(8 * (size_t)g.P2) ||
        w.linbox.reserve(4 * (size_t)g.P2 * kCap) || w.evcnt.reserve(8 * (size_t)g.P1) ||
        w.ev.reserve(sizeof(MoveEv) * (size_t)g.P1 * kCapEv) || w.evres.reserve(sizeof(EvRes) * N) ||
        w.miss.reserve(N) || w.outm.reserve(N) || w.outl.reserve(N) ||
        w.stage_m.reserve(sizeof(MainBucket) * (N + (size_t)g.P1 * kCapEv)) || w.stage_l.reserve(sizeof(LruBucket) * N) ||
        w.flags.reserve(5 * N + 64) || w.blocksum.reserve(16 * (size_t)nblocks + 16) || w.ranks.reserve(12 * N))
        return 1;
    if (w.evres.p != w.evres_seen) {      // tags of a fresh allocation must not match by accident
        CAFE_HIP_CHECK(hipMemset(w.evres.p, 0, w.evres.cap));
        w.evres_seen = w.evres.p;
    }
    return 0;
}

int par_insert(cafe_sketch *s, const int32_t *d_ids, const float *d_scores, int n, int32_t *d_ins, hipStream_t st) {
    Geo g;
    if (n > (1 << 21)) {     // the 23-bit time field of the sort key: split very large batches
        int done = 0;
        // a batch is one call in the reference; splitting would change decay/reset points, so use the in-order kernel
        (void)done;
        s->counters[2]++;
        return launch_serial_insert(s, d_ids, d_scores, n, d_ins, 0, st);
    }
    if (ensure_ws(s, n, g)) return 1;
    ParWorkspace &w = *s->par;
    w.call_seq++;
    const unsigned long long tag = w.call_seq << 8;
    Ctl *ctl = w.ctl.as<Ctl>();
    CAFE_HIP_CHECK(hipMemsetAsync(ctl, 0, sizeof(Ctl), st));
    CAFE_HIP_CHECK(hipMemsetAsync(w.cnt1.p, 0, 4 * (size_t)g.P1, st));
    CAFE_HIP_CHECK(hipMemsetAsync(w.lcnt.p, 0, 8 * (size_t)g.P2, st));
    CAFE_HIP_CHECK(hipMemsetAsync(w.evcnt.p, 0, 8 * (size_t)g.P1, st));
    const int gb = (n + 255) / 256;
    hipLaunchKernelGGL(k_assign, dim3(gb), dim3(256), 0, st, g, d_ids, w.cnt1.as<uint32_t>(), w.rank1.as<uint32_t>());
    hipLaunchKernelGGL(k_offsets, dim3(1), dim3(1024), 0, st, g, w.cnt1.as<uint32_t>(), w.off1.as<uint32_t>(), ctl, s->d_state);
    hipLaunchKernelGGL(k_scatter, dim3(gb), dim3(256), 0, st, g, d_ids, w.off1.as<uint32_t>(), w.rank1.as<uint32_t>(),
                       w.items1.as<uint32_t>());
    const int rounds = std::min(std::max(s->opt_max_rounds, 1), kMaxRounds);
    for (int r = 1; r <= rounds + 1; ++r) {
        hipLaunchKernelGGL(k_main_round, dim3(g.P1), dim3(kThreads), 0, st, g, r, tag, s->d_main, s->d_state, ctl, d_ids,
                           d_scores, w.off1.as<uint32_t>(), w.items1.as<uint32_t>(), w.lcnt.as<uint32_t>(),
                           w.linbox.as<uint32_t>(), w.evcnt.as<uint32_t>(), w.ev.as<MoveEv>(), w.evres.as<EvRes>(),
                           w.miss.as<uint8_t>(), w.outm.as<uint8_t>(), w.stage_m.as<MainBucket>());
        hipLaunchKernelGGL(k_lru_round, dim3(g.P2), dim3(kThreads), 0, st, g, r, s->d_lru, s->d_state, ctl, d_ids, d_scores,
                           w.lcnt.as<uint32_t>(), w.linbox.as<uint32_t>(), w.evcnt.as<uint32_t>(), w.ev.as<MoveEv>(),
                           w.outl.as<uint8_t>(), w.stage_l.as<LruBucket>());
    }
    const int nb = (n + kScanBlock - 1) / kScanBlock;
    uint8_t *flags = w.flags.as<uint8_t>();
    int32_t *pushval = (int32_t *)(flags + (((size_t)n + 15) & ~(size_t)15));
    uint32_t *poprank = w.ranks.as<uint32_t>(), *pushrank = poprank + n, *clkrank = pushrank + n;
    hipLaunchKernelGGL(k_finish_flags, dim3(nb), dim3(kScanBlock), 0, st, g, tag, s->d_state, ctl, w.miss.as<uint8_t>(),
                       w.outm.as<uint8_t>(), w.outl.as<uint8_t>(), w.evres.as<EvRes>(), d_ins, flags, pushval,
                       w.blocksum.as<uint32_t>());
    hipLaunchKernelGGL(k_finish_decide, dim3(1), dim3(1024), 0, st, g, nb, s->d_state, ctl, w.blocksum.as<uint32_t>());
    hipLaunchKernelGGL(k_finish_ranks, dim3(nb), dim3(kScanBlock), 0, st, g, s->d_state, flags, w.blocksum.as<uint32_t>(),
                       poprank, pushrank, clkrank);
    const long work = (long)n * 2 + (long)g.P1 * kCapEv;
    const int cb = (int)std::min<long>((work + 255) / 256, 2048);
    hipLaunchKernelGGL(k_commit, dim3(cb), dim3(256), 0, st, g, s->d_main, s->d_lru, s->d_fifo, s->fifo_cap, s->d_state, ctl,
                       w.stage_m.as<MainBucket>(), w.stage_l.as<LruBucket>(), flags, pushval, poprank, pushrank, clkrank);
    hipLaunchKernelGGL(k_commit_scalars, dim3(1), dim3(64), 0, st, s->d_state, ctl, s->fifo_cap);
    CAFE_HIP_CHECK(hipGetLastError());
    // batches the owners could not settle run in order, from the untouched tables
    if (launch_serial_insert(s, d_ids, d_scores, n, d_ins, 1, st)) return 1;
    s->counters[1]++;
    return 0;
}

void par_free(cafe_sketch *s) {
    if (!s->par) return;
    ParWorkspace &w = *s->par;
    DevBuf *all[] = {&w.ctl, &w.cnt1, &w.off1, &w.rank1, &w.items1, &w.lcnt, &w.linbox, &w.evcnt, &w.ev, &w.evres, &w.miss,
                     &w.outm, &w.outl, &w.stage_m, &w.stage_l, &w.flags, &w.blocksum, &w.ranks};
    for (DevBuf *b : all) b->release();
    delete s->par;
    s->par = nullptr;
}

}  // namespace cafe
