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
#include <cstring>
#include <rocprim/block/block_radix_sort.hpp>

#include <algorithm>

#include "cafe_internal.h"

namespace cafe {

constexpr int kThreads = 256;
constexpr int kCapSmall = 1024;   // items an owner of a normal partition sorts in LDS (several workgroups per CU)
constexpr int kCapHot = 2048;    // ... of a hot-bypass partition (they hold the heavy ids)
constexpr int kCapBig = 8192;     // ... of a partition that holds a heavy id (one workgroup per CU)
constexpr int kCapEv = 512;       // moves one main partition can receive per round
constexpr int kCapEvL = 1024;     // moves one LRU partition can emit per round (an item emits at most one)

constexpr int kMaxBins = 8192;    // partitions per table (LDS histogram of k_assign)
constexpr int kAssignChunk = 2048;
constexpr int kMaxRounds = 16;
constexpr int kHeavySeg = 4;      // a bucket with at least this many items in the batch is replayed by a whole wave
constexpr int kMaxHeavy = 64;     // heavy buckets one partition can hand to its waves (more: replayed by single lanes)
constexpr int kScanBlock = 1024;

constexpr uint32_t kSur = 0x80000000u;      // placeholder for "slot popped by item (v & 0x7fffffff)"

// outcome bits per item
constexpr uint8_t oGot = 1, oPop = 2, oCross = 4, oMiss = 8, oEv = 16;   // oEv: the item's LRU step emitted a move

struct MoveEv {             // a move into the main table emitted by an LRU owner
    uint32_t time;          // 2 * position + 1: after the item's own probe
    uint32_t x;
    float cnt;
    int32_t dic;
    int32_t kind;
    int32_t pad[3];
};

struct EvRes {              // outcome of a move, written by the main owner that carried it out
    unsigned long long tag; // call sequence number
    int32_t kind, ok, pushed, pad;
};

struct Ctl {                // zeroed at the start of every call
    uint32_t changed[kMaxRounds + 2];
    uint32_t overflow;      // overflow | final_round are read as one 8-byte word by every owner (8-byte aligned)
    uint32_t final_round;
    uint32_t maxpart;       // largest owner partition (items) of this call
    uint32_t maxhot;        // largest hot-bypass partition
    uint32_t totals[8];     // pops, pushes, clock ticks, threshold crossings, rounds: applied by the closing kernel
    uint32_t segcross[kMaxSeg];   // threshold crossings per segment of g.seg ids (a segment = one per-field reference call)
};

struct ParWorkspace {
    DevBuf zr;              // [Ctl][cnt: P1+P2][evcnt: 2*P1][hot cnt: P1], zeroed by the per-call prologue kernel
    DevBuf off, rank, items, ev, evres, miss, outm, outl, stage_m, stage_l, stage_tot, prevcnt;
    DevBuf hkey, hrun, hfin;
    DevBuf ldirty, myev, mycnt;
    DevBuf flags, blocksum, ranks;
    void *evres_seen = nullptr;
    size_t evres_seen_cap = 0;
    unsigned long long call_seq = 0;
};

struct Geo {
    int n, n1, n2, P1, P2, bpp1, bpp2;
    int end_bit1, end_bit2;     // one past the highest key bit the owners' sorts have to look at
    int hot;                    // 1: ids that sit hot in the main table before the batch bypass the owner rounds
    uint32_t gflip;             // flip bit of this call (known on the host)
    int seg;                    // > 0: ids per reference call when this call stands for several (cafe_backward_update_dev)
    int last_round;             // rounds enqueued; the LRU half of the last one is not (see final_round_of)
    int big, big_hot;           // 1: the big-partition variants of the owners / of the hot bypass are enqueued
    int verify;                 // 1: a batch that cannot be settled, or holds a reset boundary, is REJECTED (nothing is
                                // applied, the host issues it again field by field) instead of going to the in-order kernel
    FastMod fm1, fm2;
};
constexpr size_t kCtlWords = (sizeof(Ctl) + 3) / 4;
static_assert(offsetof(Ctl, overflow) % 8 == 0 && offsetof(Ctl, final_round) == offsetof(Ctl, overflow) + 4, "stop word");

// What an owner needs to know about one of its items, written once per call by k_scatter into the item's
// partition region so that owners read one coalesced stream instead of chasing ids[] / scores[] by position.
struct __attribute__((aligned(16))) Rec {
    uint32_t pos;           // batch position
    uint32_t x;             // id
    float c;                // score
    uint32_t bl;            // bucket index inside the partition
};
constexpr int kRankSortMax = 256;

// Diagnostic build (make EXTRA=-DCAFE_PROF): wave 0 of every owner stamps the 100 MHz wall clock at phase
// boundaries of round `CAFE_PROF_ROUND`; tests/diag_stamps.py reads them back.  No code in the normal build.
#ifdef CAFE_PROF
#ifndef CAFE_PROF_ROUND
#define CAFE_PROF_ROUND 1
#endif
__device__ unsigned long long g_dbg[2][8192 * 8];
#define CAFE_STAMP(t, k) do { if (threadIdx.x == 0 && round == CAFE_PROF_ROUND) g_dbg[t][blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#define CAFE_STAMP_VAL(t, k, v) do { if (threadIdx.x == 0 && round == CAFE_PROF_ROUND) g_dbg[t][blockIdx.x * 8 + (k)] = (v); } while (0)
extern "C" int cafe_debug_read(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(g_dbg)) != hipSuccess; }
#else
#define CAFE_STAMP(t, k) do { } while (0)
#define CAFE_STAMP_VAL(t, k, v) do { } while (0)
#endif   // partitions up to this size are sorted by counting smaller keys (one key per lane)

// ------------------------------------------------------------------------------------------------
// small block-wide helpers
// ------------------------------------------------------------------------------------------------
// Sort the first `total` 64-bit keys of an LDS array on key bits [begin_bit, end_bit): a workgroup radix sort
// (rocprim block primitive, 8 bits a pass), instantiated for 2, 4 and CAP/256 keys per lane so that a
// partition pays for its own size, not for the capacity.  All lanes of the workgroup must call it.
template <int IPT>
__device__ inline void block_sort_ipt(unsigned long long *key, int total, int begin_bit, int end_bit, void *storage) {
    using Sort = rocprim::block_radix_sort<unsigned long long, kThreads, IPT, rocprim::empty_type, 1, 1, 8>;
    unsigned long long k[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int j = threadIdx.x * IPT + i;
        k[i] = j < total ? key[j] : ~0ull;
    }
    __syncthreads();
    Sort().sort(k, *reinterpret_cast<typename Sort::storage_type *>(storage), begin_bit, end_bit);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int j = threadIdx.x * IPT + i;
        if (j < total) key[j] = k[i];
    }
    __syncthreads();
}
template <int IPT>
constexpr size_t sort_bytes() {
    return sizeof(typename rocprim::block_radix_sort<unsigned long long, kThreads, IPT, rocprim::empty_type, 1, 1, 8>::storage_type);
}
constexpr size_t cmax(size_t a, size_t b) { return a > b ? a : b; }
// bytes of the LDS region an owner shares between the sort's scratch and its per-item payload arrays (x, score, flags)
template <int CAP>
constexpr size_t pay_bytes() {
    size_t b = cmax(sort_bytes<2>(), (size_t)9 * CAP);
    if (CAP > 2 * kThreads) b = cmax(b, sort_bytes<4>());
    if (CAP > 4 * kThreads) b = cmax(b, sort_bytes<8>());
    if (CAP > 8 * kThreads) b = cmax(b, sort_bytes<(CAP > 8 * kThreads ? CAP / kThreads : 8)>());
    return (b + 15) & ~(size_t)15;
}
template <int CAP>
__device__ inline void block_sort(unsigned long long *key, int total, int begin_bit, int end_bit, void *storage) {
    if constexpr (CAP <= 2 * kThreads) {
        block_sort_ipt<2>(key, total, begin_bit, end_bit, storage);
    } else {
        if (total <= 2 * kThreads) block_sort_ipt<2>(key, total, begin_bit, end_bit, storage);
        else if (total <= 4 * kThreads) block_sort_ipt<4>(key, total, begin_bit, end_bit, storage);
        else if (CAP <= 8 * kThreads || total <= 8 * kThreads) block_sort_ipt<8>(key, total, begin_bit, end_bit, storage);
        else block_sort_ipt<(CAP > 8 * kThreads ? CAP / kThreads : 8)>(key, total, begin_bit, end_bit, storage);
    }
}

// ------------------------------------------------------------------------------------------------
// partition the batch by main-bucket owner and by LRU-bucket owner (both static for the call).
// Ranks inside a partition come from LDS atomics per 8192-item chunk plus one global atomic per
// (chunk, non-empty partition); the order inside a partition is arbitrary, owners sort by position.
// ------------------------------------------------------------------------------------------------
// An id that sits in the main table with a hot slot before the batch ("hot" here) is a guaranteed main-table hit for
// the whole batch: hot slots are never replaced (sketch.cpp:253-258), it never pops and never reaches the LRU table.
// Its only effects are the ordered sum of its scores and threshold-crossing counts, so such items bypass the owner
// rounds (k_hot_chain).  A bucket whose lazy decay is pending is left to the owners as a whole.
__device__ inline int hot_slot(const MainBucket *__restrict__ mt, int h1, uint32_t x, uint32_t gflip) {
    const int4 *p = reinterpret_cast<const int4 *>(&mt[h1]);
    const int4 v = p[0], d = p[1];
    const uint32_t flip = mt[h1].flip;
    int hit = -1;
    if ((uint32_t)v.w == x && d.w) hit = 3;
    if ((uint32_t)v.z == x && d.z) hit = 2;
    if ((uint32_t)v.y == x && d.y) hit = 1;
    if ((uint32_t)v.x == x && d.x) hit = 0;
    // the reference stops at the first slot holding the id (sketch.cpp:214-215): a hot copy behind a cold one is not hit
    int first = -1;
    if ((uint32_t)v.w == x) first = 3;
    if ((uint32_t)v.z == x) first = 2;
    if ((uint32_t)v.y == x) first = 1;
    if ((uint32_t)v.x == x) first = 0;
    return (flip == gflip && hit >= 0 && hit == first) ? hit : -1;
}
constexpr uint32_t kHotBit = 0x80000000u;       // rank1[i]: bit 31 hot, bits 29-30 slot, low bits rank

__global__ void __launch_bounds__(1024) k_assign(Geo g, const MainBucket *__restrict__ mt, const int32_t *__restrict__ ids,
                                                 uint32_t *__restrict__ cnt /* [P1+P2] */, uint32_t *__restrict__ cnth /* [P1] */,
                                                 uint32_t *__restrict__ rank1, uint32_t *__restrict__ rank2) {
    extern __shared__ uint32_t hist[];          // [P1 + P2 + P1]
    uint32_t *h1 = hist, *h2 = hist + g.P1, *hh = hist + g.P1 + g.P2;
    const int nb = g.P1 + g.P2 + (g.hot ? g.P1 : 0);
    for (int b = threadIdx.x; b < nb; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const int begin = blockIdx.x * kAssignChunk, end = min(begin + kAssignChunk, g.n);
    for (int i = begin + threadIdx.x; i < end; i += blockDim.x) {
        const uint32_t x = (uint32_t)ids[i];
        const int b1 = fast_hash(x, g.fm1);
        const int slot = g.hot ? hot_slot(mt, b1, x, g.gflip) : -1;
        if (slot >= 0) {
            rank1[i] = atomicAdd(&hh[b1 / g.bpp1], 1u) | kHotBit | ((uint32_t)slot << 29);
        } else {
            rank1[i] = atomicAdd(&h1[b1 / g.bpp1], 1u);
            rank2[i] = atomicAdd(&h2[fast_hash(x, g.fm2) / g.bpp2], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        uint32_t c = hist[b];
        if (c) hist[b] = atomicAdd(b < g.P1 + g.P2 ? &cnt[b] : &cnth[b - g.P1 - g.P2], c);
    }
    __syncthreads();
    for (int i = begin + threadIdx.x; i < end; i += blockDim.x) {
        const uint32_t x = (uint32_t)ids[i];
        const int p1 = fast_hash(x, g.fm1) / g.bpp1;
        const uint32_t r1 = rank1[i];
        if (r1 & kHotBit) {
            rank1[i] = r1 + hh[p1];
        } else {
            rank1[i] = r1 + h1[p1];
            rank2[i] += h2[fast_hash(x, g.fm2) / g.bpp2];
        }
    }
}

// block 0: offsets of the main partitions, block 1: of the LRU partitions
__global__ void __launch_bounds__(1024) k_offsets(Geo g, const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ cnth,
                                                  uint32_t *__restrict__ off, Ctl *ctl, const DevState *st) {
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t carry;
    // block 0: main partitions, 1: LRU partitions, 2: hot-bypass partitions (offsets behind the other two)
    const int P = blockIdx.x == 1 ? g.P2 : g.P1;
    const uint32_t *c = blockIdx.x == 0 ? cnt : blockIdx.x == 1 ? cnt + g.P1 : cnth;
    uint32_t *o = blockIdx.x == 0 ? off : blockIdx.x == 1 ? off + g.P1 + 1 : off + g.P1 + g.P2 + 2;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    uint32_t over = 0, vmax = 0;
    for (int base = 0; base < P; base += 1024) {
        int i = base + threadIdx.x;
        uint32_t v = i < P ? c[i] : 0u;
        if (v > (uint32_t)kCapBig) over = 1;
        if (v > vmax) vmax = v;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            uint32_t add = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0u;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < P) o[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) o[P] = carry;
    if (over) ctl->overflow = 1;
    if (vmax) atomicMax(blockIdx.x == 2 ? &ctl->maxhot : &ctl->maxpart, vmax);
    // the LRU clock placeholder clock0 + 1 + position must stay a positive int
    if (threadIdx.x == 0 && (long long)st->clk + g.n + 2 >= 0x7fffffffLL) ctl->overflow = 1;
}

__global__ void k_scatter(Geo g, const int32_t *__restrict__ ids, const float *__restrict__ scores,
                          const uint32_t *__restrict__ off, const uint32_t *__restrict__ rank1,
                          const uint32_t *__restrict__ rank2, Rec *__restrict__ rec1, Rec *__restrict__ rec2,
                          Rec *__restrict__ rech) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n) return;
    const uint32_t x = (uint32_t)ids[i];
    const float c = scores ? scores[i] : 1.0f;
    const int h1 = fast_hash(x, g.fm1);
    const int p1 = h1 / g.bpp1;
    const uint32_t r1 = rank1[i];
    Rec r;
    r.pos = (uint32_t)i; r.x = x; r.c = c;
    r.bl = (uint32_t)(h1 - p1 * g.bpp1);
    if (r1 & kHotBit) {
        r.bl = (r.bl << 2) | ((r1 >> 29) & 3u);            // bucket in partition, slot
        rech[off[g.P1 + g.P2 + 2 + p1] + (r1 & 0x1fffffffu)] = r;
        return;
    }
    rec1[off[p1] + r1] = r;
    const int h2 = fast_hash(x, g.fm2);
    const int p2 = h2 / g.bpp2;
    r.bl = (uint32_t)(h2 - p2 * g.bpp2);
    rec2[off[g.P1 + 1 + p2] + rank2[i]] = r;
}

// Rank sort of up to kThreads keys, one per lane: the rank of a key is the number of smaller keys (keys are
// distinct: they contain the batch position).  `key` holds the unsorted keys on entry.
__device__ inline int rank_of(const unsigned long long *key, int total, unsigned long long mine) {
    int r = 0;
    int j = 0;
    // eight keys per step, all four 16-byte LDS reads issued before the first compare (the reads are broadcasts)
    for (; j + 8 <= total; j += 8) {
        const ulonglong2 k0 = *reinterpret_cast<const ulonglong2 *>(&key[j]);
        const ulonglong2 k1 = *reinterpret_cast<const ulonglong2 *>(&key[j + 2]);
        const ulonglong2 k2 = *reinterpret_cast<const ulonglong2 *>(&key[j + 4]);
        const ulonglong2 k3 = *reinterpret_cast<const ulonglong2 *>(&key[j + 6]);
        r += (k0.x < mine) + (k0.y < mine) + (k1.x < mine) + (k1.y < mine) + (k2.x < mine) + (k2.y < mine) + (k3.x < mine) +
             (k3.y < mine);
    }
    for (; j < total; ++j) r += key[j] < mine;
    return r;
}

// ------------------------------------------------------------------------------------------------
// hot bypass: ordered score sums of the ids that are hot before the batch (see hot_slot above)
// ------------------------------------------------------------------------------------------------
// key layout: [bucket in partition << 2 | slot : 25][2 * position : 23][index of the record : 16]
// Outputs, all indexed like the partition's records (sorted order): hkey (sort keys), hrun (score of the slot after
// the item: what a move into the bucket later in the batch has to see, sketch.cpp:244-247), and at the first item
// of every slot hfin = {bucket * 4 + slot, final score} (-1 elsewhere), applied to the table by k_commit.
template <int CAP, bool BIG>
__global__ void __launch_bounds__(kThreads) k_hot_chain(Geo g, const MainBucket *__restrict__ mt, const DevState *__restrict__ st,
                                                        Ctl *ctl, const uint32_t *__restrict__ off, const Rec *__restrict__ rech,
                                                        unsigned long long *__restrict__ hkey, float *__restrict__ hrun,
                                                        int2 *__restrict__ hfin, uint8_t *__restrict__ miss,
                                                        uint8_t *__restrict__ outm) {
    __shared__ __attribute__((aligned(16))) unsigned long long key[CAP];
    __shared__ __attribute__((aligned(16))) unsigned char pay[pay_bytes<CAP>()];
    float *pc = reinterpret_cast<float *>(pay);              // score, then running score
    uint8_t *po = pay + 4 * CAP;
    __shared__ int heavy[kMaxHeavy];
    __shared__ int n_heavy;
    const int p = blockIdx.x, tid = threadIdx.x;
    const uint32_t o0 = off[g.P1 + g.P2 + 2 + p], o1 = off[g.P1 + g.P2 + 3 + p];
    if (ctl->overflow) return;
    const int total = (int)(o1 - o0);
    if (BIG ? (total <= kCapHot) : (total > kCapHot)) {       // the other size class owns it
        if (!BIG && !g.big_hot && tid == 0) ctl->overflow = 1;
        return;
    }
    if (total > CAP) { if (tid == 0) ctl->overflow = 1; return; }
    if (total == 0) return;
    if (tid == 0) n_heavy = 0;
    const int b0 = p * g.bpp1;
    const Rec *recs = rech + o0;
    if (total > 2 * kRankSortMax) __builtin_amdgcn_s_setprio(3);
    if (total <= kRankSortMax) {
        unsigned long long k = ~0ull;
        float c = 0.f;
        if (tid < total) {
            const Rec r = recs[tid];
            k = ((unsigned long long)r.bl << 39) | ((unsigned long long)(2u * r.pos) << 16);
            c = r.c;
        }
        if (tid < ((total + 1) & ~1)) key[tid] = k;
        __syncthreads();
        const int r = tid < total ? rank_of(key, total, k) : 0;
        __syncthreads();
        if (tid < total) { key[r] = k; pc[r] = c; }
        __syncthreads();
    } else {
        for (int j = tid; j < total; j += kThreads) {
            const Rec r = recs[j];
            key[j] = ((unsigned long long)r.bl << 39) | ((unsigned long long)(2u * r.pos) << 16) | (unsigned)j;
        }
        __syncthreads();
        block_sort<CAP>(key, total, 16, g.end_bit1 + 2, pay);
        for (int s = tid; s < total; s += kThreads) pc[s] = recs[(int)(key[s] & 0xffffu)].c;
        __syncthreads();
    }
    const float thr = st->thr;
    int2 *fin = hfin + o0;
    // short runs: one lane each; long runs: a wave each, 64 items at a time
    for (int s = tid; s < total; s += kThreads) {
        const uint32_t bs = (uint32_t)(key[s] >> 39);
        const bool head = s == 0 || (uint32_t)(key[s - 1] >> 39) != bs;
        if (!head) { fin[s] = make_int2(-1, 0); continue; }
        if (s + kHeavySeg <= total && (uint32_t)(key[s + kHeavySeg - 1] >> 39) == bs) {
            const int at = atomicAdd(&n_heavy, 1);
            if (at < kMaxHeavy) { heavy[at] = s; continue; }
        }
        const int bucket = b0 + (int)(bs >> 2), slot = (int)(bs & 3u);
        float acc = mt[bucket].cnt[slot];
        for (int e = s; e < total && (uint32_t)(key[e] >> 39) == bs; ++e) {
            const float c = pc[e];
            acc += c;
            po[e] = (acc >= thr && acc - c < thr) ? oCross : 0;         // sketch.cpp:218-220
            pc[e] = acc;
        }
        fin[s] = make_int2(bucket * 4 + slot, __float_as_int(acc));
    }
    __syncthreads();
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int nh = min(n_heavy, kMaxHeavy);
        for (int h = wave; h < nh; h += kThreads / 64) {
            const int s = heavy[h];
            const uint32_t bs = (uint32_t)(key[s] >> 39);
            const int bucket = b0 + (int)(bs >> 2), slot = (int)(bs & 3u);
            float acc = mt[bucket].cnt[slot];
            for (int e = s;; e += 64) {
                const int idx = e + lane;
                const bool act = idx < total && (uint32_t)(key[idx] >> 39) == bs;
                const unsigned long long m = __ballot(act);
                if (!m) break;
                const int r = __popcll(m);                    // the run's items are contiguous from lane 0
                const float c = act ? pc[idx] : 0.0f;
                float mine = 0.0f;
                for (int j = 0; j < r; ++j) {
                    acc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), j));
                    mine = lane == j ? acc : mine;
                }
                if (act) {
                    po[idx] = (mine >= thr && mine - c < thr) ? oCross : 0;
                    pc[idx] = mine;
                }
                if (r < 64) break;
            }
            if (lane == 0) fin[s] = make_int2(bucket * 4 + slot, __float_as_int(acc));
        }
    }
    __syncthreads();
    for (int s = tid; s < total; s += kThreads) {
        const unsigned long long k = key[s];
        const uint32_t pos = ((uint32_t)(k >> 16) & 0x7fffffu) >> 1;
        hkey[o0 + s] = k;
        hrun[o0 + s] = pc[s];
        outm[pos] = po[s];
        miss[pos] = 0;
    }
}

// Score of hot slot `slot` of bucket `bl` (partition-local) as a move at `time` sees it: the running score behind the
// slot's last bypassed item before `time`, or `base` (the table's value) if there is none.
__device__ inline float hot_cnt_at(const unsigned long long *__restrict__ hk, const float *__restrict__ hr, int nh,
                                   uint32_t bl, int slot, uint32_t time, float base) {
    const unsigned long long target = ((unsigned long long)((bl << 2) | (uint32_t)slot) << 23) | time;   // in units of key >> 16
    int lo = 0, hi = nh;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((hk[mid] >> 16) < target) lo = mid + 1; else hi = mid;
    }
    if (lo > 0 && (uint32_t)(hk[lo - 1] >> 39) == ((bl << 2) | (uint32_t)slot)) return hr[lo - 1];
    return base;
}
__device__ inline void patch_hot(MainR &b, const unsigned long long *__restrict__ hk, const float *__restrict__ hr, int nh,
                                 uint32_t bl, uint32_t time) {
    if (nh <= 0) return;
    if (b.d0) b.c0 = hot_cnt_at(hk, hr, nh, bl, 0, time, b.c0);
    if (b.d1) b.c1 = hot_cnt_at(hk, hr, nh, bl, 1, time, b.c1);
    if (b.d2) b.c2 = hot_cnt_at(hk, hr, nh, bl, 2, time, b.c2);
    if (b.d3) b.c3 = hot_cnt_at(hk, hr, nh, bl, 3, time, b.c3);
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

// Staged buckets have a static home: the entry of a bucket is the sorted position of its first item inside its
// partition's region, so a partition that is not re-run in a later round keeps its staged result.
// key layout of a main owner: [bucket in partition : 25][2*position + phase : 23][move index | kind<<15 : 16]
template <int CAP, bool BIG>
__global__ void __launch_bounds__(kThreads) k_main_round(
    Geo g, int round, unsigned long long call_tag, const MainBucket *__restrict__ mt, const DevState *__restrict__ st,
    Ctl *ctl, const uint32_t *__restrict__ off, const Rec *__restrict__ rec1, const uint32_t *__restrict__ evcnt /* [2][P1] */,
    const MoveEv *__restrict__ ev, EvRes *__restrict__ evres, uint8_t *__restrict__ miss, uint8_t *__restrict__ outm,
    MainBucket *__restrict__ stage_m, uint32_t *__restrict__ stage_tot, uint32_t *__restrict__ prevcnt,
    uint8_t *__restrict__ ldirty /* [2][P2] */, const unsigned long long *__restrict__ hkey, const float *__restrict__ hrun) {
    __shared__ __attribute__((aligned(16))) unsigned long long key[CAP];
    __shared__ __attribute__((aligned(16))) unsigned char pay[pay_bytes<CAP>()];     // sort scratch, then x | score | flags
    uint32_t *px = reinterpret_cast<uint32_t *>(pay);
    float *pc = reinterpret_cast<float *>(pay + 4 * CAP);
    uint8_t *po = pay + 8 * CAP;
    __shared__ int heavy[kMaxHeavy];
    __shared__ int n_heavy;

    // st->fallback is only ever raised by the finish kernels, behind all rounds: owners need not look at it
    const uint2 stop = *reinterpret_cast<const uint2 *>(&ctl->overflow);
    const uint32_t o0 = off[blockIdx.x], o1 = off[blockIdx.x + 1];       // requested together with the stop word
    if (stop.x | stop.y) return;
    const int p = blockIdx.x, tid = threadIdx.x;
    const int prev = (round & 1) ^ 1;
    if (tid == 0) n_heavy = 0;
    const int np = (int)(o1 - o0);
    const uint32_t ne_raw = round > 1 ? evcnt[prev * g.P1 + p] : 0u;       // moves emitted by LRU round-1
    if (ne_raw > (uint32_t)kCapEv) { if (tid == 0) ctl->overflow = 1; return; }
    const int ne = (int)ne_raw;
    const int total = np + ne;
    if (BIG ? (total <= kCapSmall) : (total > kCapSmall)) {                 // the other size class owns it
        if (!BIG && !g.big && tid == 0) ctl->overflow = 1;
        return;
    }
    if (total > CAP) { if (tid == 0) ctl->overflow = 1; return; }
    // Same moves as the last round this partition ran with => same result: its staged buckets, per-item
    // outcomes and move outcomes stand.  Lists are compared as sets (arrival order is not deterministic).
    const size_t evbase = (size_t)prev * g.P1 * kCapEv + (size_t)p * kCapEv;          // this round's moves
    const size_t evold = (size_t)(prev ^ 1) * g.P1 * kCapEv + (size_t)p * kCapEv;     // the moves of the round before
    if (round > 1) {
        const int pne = (int)prevcnt[p];
        if (ne == 0 && pne == 0) return;             // ran without moves last time, no moves now
        int differ = ne != pne;
        if (!differ) {
            for (int t = tid; t < ne; t += kThreads) {
                const MoveEv a = ev[evbase + t];
                int found = 0;
                for (int j = 0; j < ne; ++j) {
                    const MoveEv b = ev[evold + j];
                    found |= (a.time == b.time && a.x == b.x && __float_as_uint(a.cnt) == __float_as_uint(b.cnt) &&
                              a.dic == b.dic && a.kind == b.kind);
                }
                differ |= !found;
            }
        }
        differ = __syncthreads_or(differ);
        if (tid == 0) prevcnt[p] = (uint32_t)ne;
        if (!differ) return;
    } else if (tid == 0) {
        prevcnt[p] = 0;
    }
    if (tid == 0) stage_tot[p] = (uint32_t)total;
    if (total == 0) return;
    const int b0 = p * g.bpp1;
    const int par = round & 1;
    const Rec *recs = rec1 + o0;
    // this partition's bypassed hot items, looked at only when a move arrives (it compares all four scores)
    const uint32_t oh0 = g.hot && ne ? off[g.P1 + g.P2 + 2 + p] : 0u;
    const int nhot = g.hot && ne ? (int)(off[g.P1 + g.P2 + 3 + p] - oh0) : 0;
    CAFE_STAMP(0, 0);
    CAFE_STAMP_VAL(0, 7, (unsigned long long)total);
    // the kernel ends when its largest partition is done: let that workgroup's waves issue first on their SIMDs
    if (total > 2 * kRankSortMax) __builtin_amdgcn_s_setprio(3);

    // key layout: [bucket in partition : 25][2*position + phase : 23][16 bits: item -> index of its record, move -> index | kind<<15]
    if (total <= kRankSortMax) {
        // one entry per lane: everything about it stays in registers while its rank is counted
        unsigned long long k = ~0ull;
        uint32_t x = 0; float c = 0.f; uint8_t pm = 0;
        if (tid < np) {
            const Rec r = recs[tid];
            k = ((unsigned long long)r.bl << 39) | ((unsigned long long)(2u * r.pos) << 16);
            x = r.x; c = r.c;
            if (round > 1) pm = miss[r.pos];
        } else if (tid < total) {
            const int e = tid - np;
            const MoveEv m = ev[evbase + e];
            const uint32_t bl = (uint32_t)(fast_hash(m.x, g.fm1) - b0);
            k = ((unsigned long long)bl << 39) | ((unsigned long long)m.time << 16) | (unsigned)(e | ((m.kind == 1) << 15));
            x = m.x; c = m.cnt;
        }
        if (tid < ((total + 1) & ~1)) key[tid] = k;
        __syncthreads();
        const int r = tid < total ? rank_of(key, total, k) : 0;
        __syncthreads();
        if (tid < total) { key[r] = k; px[r] = x; pc[r] = c; po[r] = pm; }
        __syncthreads();
    } else {
        for (int j = tid; j < total; j += kThreads) {
            unsigned long long k = ~0ull;
            if (j < np) {
                const Rec r = recs[j];
                k = ((unsigned long long)r.bl << 39) | ((unsigned long long)(2u * r.pos) << 16) | (unsigned)j;
            } else if (j < total) {
                const int e = j - np;
                const MoveEv &m = ev[evbase + e];
                const uint32_t bl = (uint32_t)(fast_hash(m.x, g.fm1) - b0);
                k = ((unsigned long long)bl << 39) | ((unsigned long long)m.time << 16) | (unsigned)(e | ((m.kind == 1) << 15));
            }
            key[j] = k;
        }
        __syncthreads();
        block_sort<CAP>(key, total, 16, g.end_bit1, pay);
        for (int s = tid; s < total; s += kThreads) {
            const unsigned long long k = key[s];
            const uint32_t time = (uint32_t)(k >> 16) & 0x7fffffu;
            if (time & 1u) {
                const MoveEv &m = ev[evbase + (int)(k & 0x7fffu)];
                px[s] = m.x; pc[s] = m.cnt; po[s] = 0;
            } else {
                const Rec r = recs[(int)(k & 0xffffu)];
                px[s] = r.x; pc[s] = r.c;
                po[s] = round > 1 ? miss[r.pos] : 0;
            }
        }
        __syncthreads();
    }

    CAFE_STAMP(0, 1);
    const float thr = st->thr;
    const uint32_t gflip = (uint32_t)st->flip;
    MainBucket *stage = stage_m + o0 + (size_t)p * kCapEv;
    MainCtx ctx;
    ctx.full = st->fifo_len > 0;
    int changed = 0;
    // Phase A: one lane per bucket replays the bucket's items in batch order.  Buckets with many items (a
    // heavy id) are left to phase B, where a whole wave works on one bucket.
    // The table bucket of the next head is requested before the current one is replayed, so that a lane with several
    // heads does not pay a memory round trip (and the completion of its own staging stores) per head.
    auto classify = [&](int s) -> int {        // 0: not a head, 1: replay here, 2: handed to phase B
        if (s >= total) return 0;
        const uint32_t bl = (uint32_t)(key[s] >> 39);
        if (s != 0 && (uint32_t)(key[s - 1] >> 39) == bl) return 0;
        if (s + kHeavySeg <= total && (uint32_t)(key[s + kHeavySeg - 1] >> 39) == bl) {
            const int at = atomicAdd(&n_heavy, 1);
            if (at < kMaxHeavy) { heavy[at] = s; return 2; }
        }
        return 1;
    };
    MainR nb;
    int nmode = classify(tid);
    if (nmode == 1) load_main(nb, &mt[b0 + (int)(uint32_t)(key[tid] >> 39)]);
    for (int s = tid; s < total; s += kThreads) {
        const int mode = nmode;
        MainR b = nb;
        nmode = classify(s + kThreads);
        if (nmode == 1) load_main(nb, &mt[b0 + (int)(uint32_t)(key[s + kThreads] >> 39)]);
        if (mode != 1) continue;
        const uint32_t bl = (uint32_t)(key[s] >> 39);
        sync_flip(b, gflip);
#ifdef CAFE_PROF
        if (s == 0) { CAFE_STAMP(0, 5); }
#endif
        for (int e = s; e < total && (uint32_t)(key[e] >> 39) == bl; ++e) {
            const unsigned long long k = key[e];
            const uint32_t time = (uint32_t)(k >> 16) & 0x7fffffu;
            if (!(time & 1u)) {
                ctx.pos = time >> 1; ctx.flags = 0;
                int r = main_probe(b, px[e], pc[e], thr, ctx);
                const bool m = r < 0;
                if (m != (po[e] != 0)) { changed = 1; if (round > 1) ldirty[par * g.P2 + fast_hash(px[e], g.fm2) / g.bpp2] = 1; }
                po[e] = (uint8_t)(ctx.flags | (m ? oMiss : 0) | (r > 0 ? oGot : 0));
            } else {
                ctx.pushed = 0;
                const int32_t dic = ev[evbase + (int)(k & 0x7fffu)].dic;
                patch_hot(b, hkey + oh0, hrun + oh0, nhot, bl, time);
                int ok = main_move(b, px[e], pc[e], dic, ctx);
                EvRes r;
                r.tag = call_tag; r.kind = ((k >> 15) & 1u) ? 1 : 2; r.ok = ok; r.pushed = ctx.pushed; r.pad = 0;
                evres[time >> 1] = r;
            }
        }
#ifdef CAFE_PROF
        if (s == 0) CAFE_STAMP(0, 6);
#endif
        store_main(&stage[s], b, b0 + (int)bl);
    }
    __syncthreads();
    CAFE_STAMP(0, 2);
    CAFE_STAMP_VAL(0, 7, (unsigned long long)total | ((unsigned long long)n_heavy << 32));
    // Phase B: heavy buckets, one per wave at a time.  All 64 lanes carry the same bucket copy.  A run of
    // consecutive probes of one id that sits in the bucket only needs the ordered float sum of the scores on
    // the serial path (one readlane + add per item); threshold crossing and the pop are then evaluated for 64
    // items at once.  Anything else (misses, moves) takes the one-item step, executed uniformly.
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int nh = min(n_heavy, kMaxHeavy);
        MainR nhb;
        if (wave < nh) load_main(nhb, &mt[b0 + (int)(uint32_t)(key[heavy[wave]] >> 39)]);
        for (int h = wave; h < nh; h += kThreads / 64) {
            const int s = heavy[h];
            const uint32_t bl = (uint32_t)(key[s] >> 39);
            MainR b = nhb;
            if (h + kThreads / 64 < nh) load_main(nhb, &mt[b0 + (int)(uint32_t)(key[heavy[h + kThreads / 64]] >> 39)]);
            sync_flip(b, gflip);
            int e = s;
            while (e < total && (uint32_t)(key[e] >> 39) == bl) {
                const unsigned long long k0 = key[e];
                const uint32_t time0 = (uint32_t)(k0 >> 16) & 0x7fffffu;
                const uint32_t x0 = px[e];
                const int hit = (time0 & 1u) ? -1 : find_slot(b, x0);
                if (hit >= 0) {
                    // A hot entry far enough above the threshold cannot register a crossing or a pop in a run of
                    // probes of its own id (with scores >= 0 the score only grows, and fl(fl(a + c) - c) >=
                    // a - ulp(a + c)): only the ordered sum is needed.  The run may be thousands of items long (a
                    // heavy id): its length, a bound of its sum and the test are taken 64 items at a time, then
                    // every lane walks the same chain of adds with the scores streamed from LDS.
                    if (sel4(b.d0, b.d1, b.d2, b.d3, hit) != 0) {
                        int r = 0;
                        float part = 0.0f;
                        int flag = 0;                      // bit 0: a score is negative or NaN, bit 1: an item missed last round
                        for (int at = e;; at += 64) {
                            const int idx = at + lane;
                            const bool in = idx < total;
                            const unsigned long long kk = in ? key[idx] : ~0ull;
                            const bool same = in && (uint32_t)(kk >> 39) == bl && !((uint32_t)(kk >> 16) & 1u) && px[idx] == x0;
                            const unsigned long long m = __ballot(same);
                            const int rr = (~m) ? __builtin_ctzll(~m) : 64;
                            if (lane < rr) {
                                const float c = pc[idx];
                                part += c;
                                flag |= (c >= 0.0f ? 0 : 1) | (po[idx] != 0 ? 2 : 0);
                            }
                            r += rr;
                            if (rr < 64) break;
                        }
                        for (int o = 32; o > 0; o >>= 1) { part += __shfl_xor(part, o, 64); flag |= __shfl_xor(flag, o, 64); }
                        float acc = sel4f(b.c0, b.c1, b.c2, b.c3, hit);
                        if (!(flag & 1) && acc - (acc + part) * 2.4e-7f >= thr) {
                            const float *cs = pc + e;
                            int j = 0;
                            if (r >= 8) {
                                float n0 = cs[0], n1 = cs[1], n2 = cs[2], n3 = cs[3], n4 = cs[4], n5 = cs[5], n6 = cs[6], n7 = cs[7];
                                for (; j + 16 <= r; j += 8) {
                                    const float c0 = n0, c1 = n1, c2 = n2, c3 = n3, c4 = n4, c5 = n5, c6 = n6, c7 = n7;
                                    n0 = cs[j + 8]; n1 = cs[j + 9]; n2 = cs[j + 10]; n3 = cs[j + 11];
                                    n4 = cs[j + 12]; n5 = cs[j + 13]; n6 = cs[j + 14]; n7 = cs[j + 15];
                                    acc += c0; acc += c1; acc += c2; acc += c3; acc += c4; acc += c5; acc += c6; acc += c7;
                                }
                                acc += n0; acc += n1; acc += n2; acc += n3; acc += n4; acc += n5; acc += n6; acc += n7;
                                j += 8;
                            }
                            for (; j < r; ++j) acc += cs[j];
                            if (flag & 2) {
                                if (lane == 0) { changed = 1; if (round > 1) ldirty[par * g.P2 + fast_hash(x0, g.fm2) / g.bpp2] = 1; }
                            }
                            for (int i = lane; i < r; i += 64) po[e + i] = 0;
                            CAFE_SET4(b.c0, b.c1, b.c2, b.c3, hit, acc);
                            e += r;
                            continue;
                        }
                    }
                    const int idx = e + lane;
                    const bool in = idx < total;
                    const unsigned long long kk = in ? key[idx] : ~0ull;
                    const bool same = in && (uint32_t)(kk >> 39) == bl && !((uint32_t)(kk >> 16) & 1u) && px[idx] == x0;
                    const unsigned long long m = __ballot(same);
                    const int r = (~m) ? __builtin_ctzll(~m) : 64;            // length of the run starting at e (>= 1)
                    const float c = in ? pc[idx] : 0.0f;
                    const bool act = lane < r;
                    float acc = sel4f(b.c0, b.c1, b.c2, b.c3, hit), mine = 0.0f;
                    for (int j = 0; j < r; ++j) {
                        acc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), j));
                        mine = lane == j ? acc : mine;
                    }
                    const bool cross = act && mine >= thr && mine - c < thr;
                    int32_t dic = sel4(b.d0, b.d1, b.d2, b.d3, hit);
                    const unsigned long long pm = (!dic && ctx.full) ? __ballot(act && mine >= thr) : 0ull;
                    const int kpop = pm ? __builtin_ctzll(pm) : -1;
                    if (kpop >= 0) {
                        const uint32_t tpop = (uint32_t)(key[e + kpop] >> 16) & 0x7fffffu;
                        dic = (int32_t)(kSur | (tpop >> 1));
                    }
                    if (act) {
                        if (po[idx] != 0) { changed = 1; if (round > 1) ldirty[par * g.P2 + fast_hash(x0, g.fm2) / g.bpp2] = 1; }   // it missed last round
                        po[idx] = (uint8_t)((cross ? oCross : 0) | (lane == kpop ? (oPop | oGot) : 0));
                    }
                    CAFE_SET4(b.c0, b.c1, b.c2, b.c3, hit, acc);
                    CAFE_SET4(b.d0, b.d1, b.d2, b.d3, hit, dic);
                    e += r;
                } else if (!(time0 & 1u)) {
                    if (lane == 0) {
                        if (po[e] == 0) { changed = 1; if (round > 1) ldirty[par * g.P2 + fast_hash(x0, g.fm2) / g.bpp2] = 1; }
                        po[e] = oMiss;
                    }
                    e += 1;
                } else {
                    ctx.pushed = 0;
                    const int32_t dic = ev[evbase + (int)(k0 & 0x7fffu)].dic;
                    patch_hot(b, hkey + oh0, hrun + oh0, nhot, bl, time0);
                    int ok = main_move(b, x0, pc[e], dic, ctx);
                    if (lane == 0) {
                        EvRes r;
                        r.tag = call_tag; r.kind = ((k0 >> 15) & 1u) ? 1 : 2; r.ok = ok; r.pushed = ctx.pushed; r.pad = 0;
                        evres[time0 >> 1] = r;
                    }
                    e += 1;
                }
            }
            if (lane == 0) store_main(&stage[s], b, b0 + (int)bl);
        }
    }
    __syncthreads();
    CAFE_STAMP(0, 3);
    // parallel write-back of the per-item outcomes; entries that are not a bucket's first item are marked unused
    for (int s = tid; s < total; s += kThreads) {
        const unsigned long long k = key[s];
        const uint32_t time = (uint32_t)(k >> 16) & 0x7fffffu;
        if (!(time & 1u)) {
            const uint32_t pos = time >> 1;
            const uint8_t o = po[s];
            outm[pos] = o;
            miss[pos] = (o & oMiss) ? 1 : 0;
        }
        if (s != 0 && (uint32_t)(key[s - 1] >> 39) == (uint32_t)(k >> 39))
            reinterpret_cast<int32_t *>(&stage[s])[13] = -1;         // tag word of store_main
    }
    if (changed) atomicOr(&ctl->changed[round], 1u);
    CAFE_STAMP(0, 4);
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

// key layout of an LRU owner: [bucket in partition : 24][position : 24][index of the item's record : 16]
__device__ inline uint32_t lkey_bl(unsigned long long k) { return (uint32_t)(k >> 40); }
__device__ inline uint32_t lkey_pos(unsigned long long k) { return (uint32_t)(k >> 16) & 0xffffffu; }

template <int CAP, bool BIG>
__global__ void __launch_bounds__(kThreads) k_lru_round(
    Geo g, int round, const LruBucket *__restrict__ lt, const DevState *__restrict__ st, Ctl *ctl,
    const uint32_t *__restrict__ off, const Rec *__restrict__ rec2, const uint8_t *__restrict__ miss, uint32_t *evcnt,
    MoveEv *__restrict__ ev, uint8_t *__restrict__ outl, LruBucket *__restrict__ stage_l, uint32_t *__restrict__ stage_tot,
    uint8_t *__restrict__ ldirty /* [2][P2] */, MoveEv *__restrict__ myev /* [P2][kCapEvL] */, uint32_t *__restrict__ mycnt) {
    __shared__ __attribute__((aligned(16))) unsigned long long key[CAP];
    __shared__ __attribute__((aligned(16))) unsigned char pay[pay_bytes<CAP>()];     // sort scratch, then x | score | flags
    uint32_t *px = reinterpret_cast<uint32_t *>(pay);
    float *pc = reinterpret_cast<float *>(pay + 4 * CAP);
    uint8_t *po = pay + 8 * CAP;
    __shared__ int n_miss, n_ev;

#ifdef CAFE_PROF
    CAFE_STAMP(1, 4);
    const unsigned long long mt0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const uint2 stop = *reinterpret_cast<const uint2 *>(&ctl->overflow);
    const uint32_t round_changed = ctl->changed[round];
    const uint32_t o0 = off[g.P1 + 1 + blockIdx.x], o1 = off[g.P1 + 2 + blockIdx.x];
    if (stop.x | stop.y) return;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int par = round & 1, nxt = par ^ 1;
#ifdef CAFE_PROF
    CAFE_STAMP(1, 5);
#endif
    if (!round_changed) {                          // the miss set repeated: the previous round is the answer
        if (!BIG && q == 0 && tid == 0) ctl->final_round = (uint32_t)round;
        return;
    }
    if (!BIG) for (int p = q; p < g.P1; p += gridDim.x) if (tid == 0) evcnt[nxt * g.P1 + p] = 0;

    const int cnt = (int)(o1 - o0);
    if (BIG ? (cnt <= kCapSmall) : (cnt > kCapSmall)) {
        if (!BIG && !g.big && tid == 0) ctl->overflow = 1;
        return;
    }
    if (cnt > CAP) { if (tid == 0) ctl->overflow = 1; return; }
    MoveEv *mine_ev = myev + (size_t)q * kCapEvL;
    const bool dirty = round == 1 || ldirty[par * g.P2 + q];
    if (tid == 0) ldirty[nxt * g.P2 + q] = 0;
    if (!dirty) {
        // none of this partition's items changed between hit and miss: same result, so only hand the moves it
        // emitted last time to this round's lists
        const int ne = (int)mycnt[q];
        for (int j = tid; j < ne; j += kThreads) {
            const MoveEv m = mine_ev[j];
            const int p = m.pad[0];
            uint32_t at = atomicAdd(&evcnt[par * g.P1 + p], 1u);
            if (at < (uint32_t)kCapEv) ev[(size_t)par * g.P1 * kCapEv + (size_t)p * kCapEv + at] = m; else ctl->overflow = 1;
        }
        return;
    }
    if (tid == 0) { n_miss = 0; n_ev = 0; }
    __syncthreads();
    const int b0 = q * g.bpp2;
    const Rec *recs = rec2 + o0;
    int total;
    CAFE_STAMP(1, 0);
    if (cnt <= kRankSortMax) {
        unsigned long long k = ~0ull;
        uint32_t x = 0; float c = 0.f;
        if (tid < cnt) {
            const Rec r = recs[tid];
#ifdef CAFE_PROF
            if (tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CAFE_STAMP(1, 6); }
#endif
            if (miss[r.pos]) {
                k = ((unsigned long long)r.bl << 40) | ((unsigned long long)r.pos << 16);
                x = r.x; c = r.c;
            }
        }
#ifdef CAFE_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#endif
        const unsigned long long mm = __ballot(k != ~0ull);
        if ((tid & 63) == 0 && mm) atomicAdd(&n_miss, __popcll(mm));
        if (tid < ((cnt + 1) & ~1)) key[tid] = k;
#ifdef CAFE_PROF
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef CAFE_PROF
        const unsigned long long c2 = __builtin_amdgcn_s_memtime();
#endif
        total = n_miss;
        if (tid == 0) { stage_tot[g.P1 + q] = (uint32_t)total; if (total == 0) mycnt[q] = 0; }
        if (total == 0) return;
        const int r = k != ~0ull ? rank_of(key, cnt, k) : 0;
#ifdef CAFE_PROF
        const unsigned long long c3 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
        if (k != ~0ull) { key[r] = k; px[r] = x; pc[r] = c; }
        __syncthreads();
#ifdef CAFE_PROF
        const unsigned long long c4 = __builtin_amdgcn_s_memtime();
        if (tid == 0 && round == CAFE_PROF_ROUND) {
            unsigned long long *d = &g_dbg[1][blockIdx.x * 8];
            d[4] = ((c1 - c0) << 32) | (c2 - c1);  d[5] = ((c3 - c2) << 32) | (c4 - c3);
        }
#endif
    } else {
        int mine = 0;
        for (int j = tid; j < cnt; j += kThreads) {
            unsigned long long k = ~0ull;
            if (j < cnt) {
                const Rec r = recs[j];
                if (miss[r.pos]) {
                    k = ((unsigned long long)r.bl << 40) | ((unsigned long long)r.pos << 16) | (unsigned)j;
                    ++mine;
                }
            }
            key[j] = k;
        }
        if (mine) atomicAdd(&n_miss, mine);
        __syncthreads();
        total = n_miss;
        if (tid == 0) { stage_tot[g.P1 + q] = (uint32_t)total; if (total == 0) mycnt[q] = 0; }
        if (total == 0) return;
        block_sort<CAP>(key, cnt, 16, g.end_bit2, pay);
        for (int s = tid; s < total; s += kThreads) {
            const Rec r = recs[(int)(key[s] & 0xffffu)];
            px[s] = r.x; pc[s] = r.c;
        }
        __syncthreads();
    }

    CAFE_STAMP(1, 1);
    CAFE_STAMP_VAL(1, 7, (unsigned long long)total);
    const float thr = st->thr;
    const uint32_t gflip = (uint32_t)st->flip;
    LruBucket *stage = stage_l + o0;
    LruCtx ctx;
    ctx.full = st->fifo_len > 0;
    ctx.clk0 = st->clk;
    int over = 0;
    for (int s = tid; s < total; s += kThreads) {
        const uint32_t bl = lkey_bl(key[s]);
        if (s != 0 && lkey_bl(key[s - 1]) == bl) continue;
        LruR b;
        load_lru(b, &lt[b0 + (int)bl]);
        sync_flip(b, gflip);
        for (int e = s; e < total && lkey_bl(key[e]) == bl; ++e) {
            const uint32_t pos = lkey_pos(key[e]);
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
            po[e] = (uint8_t)(ctx.flags | (got ? oGot : 0) | (mv.kind ? oEv : 0));
            if (mv.kind) {
                int p = fast_hash(mv.x, g.fm1) / g.bpp1;
                MoveEv m;
                m.time = 2u * pos + 1u; m.x = mv.x; m.cnt = mv.cnt; m.dic = mv.dic; m.kind = mv.kind;
                m.pad[0] = p; m.pad[1] = m.pad[2] = 0;
                uint32_t at = atomicAdd(&evcnt[par * g.P1 + p], 1u);
                if (at < (uint32_t)kCapEv) ev[(size_t)par * g.P1 * kCapEv + (size_t)p * kCapEv + at] = m; else over = 1;
                int mi = atomicAdd(&n_ev, 1);
                if (mi < kCapEvL) mine_ev[mi] = m; else over = 1;
            }
        }
        store_lru(&stage[s], b, b0 + (int)bl);
    }
    __syncthreads();
    CAFE_STAMP(1, 2);
    if (tid == 0) mycnt[q] = (uint32_t)min(n_ev, kCapEvL);
    for (int s = tid; s < total; s += kThreads) {
        outl[lkey_pos(key[s])] = po[s];
        if (s != 0 && lkey_bl(key[s - 1]) == lkey_bl(key[s]))
            reinterpret_cast<int32_t *>(&stage[s])[17] = -1;         // tag word of store_lru
    }
    if (over) ctl->overflow = 1;
    CAFE_STAMP(1, 3);
#ifdef CAFE_PROF
    if (threadIdx.x == 0 && round == CAFE_PROF_ROUND) {
        const unsigned long long mt1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
        (void)mt1; (void)rt1;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// finish: batch-ordered prefix sums, checks, commit
// ------------------------------------------------------------------------------------------------
// The round whose main half found the miss set unchanged is the answer.  LRU round r says so for rounds < last; for the
// last enqueued round nobody has to run: its LRU half would either only say so or compute moves no main round consumes.
__device__ inline uint32_t final_round_of(const Geo &g, const Ctl *ctl) {
    uint32_t fr = ctl->final_round;
    if (!fr && g.last_round > 0 && ctl->changed[g.last_round] == 0) fr = (uint32_t)g.last_round;
    return fr;
}

// flags per item: bit0 pop, bit1 push, bit2 clock tick
__global__ void __launch_bounds__(kScanBlock) k_finish_flags(
    Geo g, unsigned long long call_tag, const DevState *st, Ctl *ctl, const uint8_t *__restrict__ miss,
    const uint8_t *__restrict__ outm, const uint8_t *__restrict__ outl, const EvRes *__restrict__ evres,
    int32_t *__restrict__ ins, uint8_t *__restrict__ flags, int32_t *__restrict__ pushval, uint32_t *__restrict__ blocksum) {
    __shared__ uint32_t s_pop, s_push, s_clk, s_cross;
    if (threadIdx.x == 0) { s_pop = s_push = s_clk = s_cross = 0; }
    __syncthreads();
    const bool live = !(st->fallback || ctl->overflow) && final_round_of(g, ctl) != 0;
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
            if ((o & oEv) && r.tag == call_tag) {
                if (r.kind == 1 && !r.ok) got = 0;           // sketch.cpp:180-182
                if (!r.ok && r.pushed != 0) { push = 1; pv = r.pushed; }
            }
        }
        ins[i] = got;
        pushval[i] = pv;
        flags[i] = (uint8_t)(pop | (push << 1) | (clk << 2));
        if (cross && g.seg > 0) atomicAdd(&ctl->segcross[i / g.seg], 1u);      // a few hundred per call
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

// per-item exclusive ranks in batch order.  Every workgroup sums the block totals in front of it; workgroup 0 also
// forms the batch totals, checks the free-list assumption and decides whether the result may be committed.
__global__ void __launch_bounds__(kScanBlock) k_finish_ranks(Geo g, int nblocks, double reset_at, DevState *st, Ctl *ctl,
                                                             const uint8_t *__restrict__ flags,
                                                             const uint32_t *__restrict__ blocksum, uint32_t *__restrict__ poprank,
                                                             uint32_t *__restrict__ pushrank, uint32_t *__restrict__ clkrank) {
    __shared__ uint32_t wsum[3][kScanBlock / 64];
    __shared__ unsigned long long red[4][kScanBlock / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // sums of the block totals: [0, blockIdx) for the offsets; workgroup 0 takes all blocks for the totals
    const int upto = blockIdx.x == 0 ? nblocks : (int)blockIdx.x;
    unsigned long long part[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < upto; b += blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4 *>(blocksum)[b];
        part[0] += v.x; part[1] += v.y; part[2] += v.z; part[3] += v.w;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        unsigned long long v = part[c];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[c][w] = v;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint8_t f = i < g.n ? flags[i] : 0;
    const unsigned long long below = (1ull << lane) - 1;
    unsigned long long m0 = __ballot(f & 1), m1 = __ballot(f & 2), m2 = __ballot(f & 4);
    if (lane == 0) { wsum[0][w] = __popcll(m0); wsum[1][w] = __popcll(m1); wsum[2][w] = __popcll(m2); }
    __syncthreads();
    unsigned long long tot[4] = {0, 0, 0, 0};
    for (int k = 0; k < kScanBlock / 64; ++k) { tot[0] += red[0][k]; tot[1] += red[1][k]; tot[2] += red[2][k]; tot[3] += red[3][k]; }
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            ctl->final_round = final_round_of(g, ctl);
            bool ok = !(st->fallback || ctl->overflow) && ctl->final_round != 0;
            if (!ok) st->why[ctl->overflow ? 0 : 1] += 1;
            if (ok) {
                if (st->fifo_len > 0) ok = tot[0] <= (unsigned long long)st->fifo_len;      // never ran dry
                else ok = tot[1] == 0;                                                       // never refilled
                if (!ok) st->why[2] += 1;
            }
            ctl->totals[0] = (uint32_t)tot[0]; ctl->totals[1] = (uint32_t)tot[1]; ctl->totals[2] = (uint32_t)tot[2];
            ctl->totals[3] = (uint32_t)tot[3]; ctl->totals[4] = ctl->final_round;
            st->rounds = (int32_t)ctl->final_round;
            st->maxpart = (int32_t)ctl->maxpart;
            st->pad[3] = (int32_t)ctl->maxhot;
            if (g.verify) {
                // sketch.cpp:291 at the inner boundaries, with the crossings k_finish_flags counted per segment
                bool inner_reset = false;
                long run = st->real_n;
                const int nseg = (g.n + g.seg - 1) / g.seg;
                for (int f = 0; f + 1 < nseg; ++f) {
                    run += ctl->segcross[f];
                    if ((double)run > reset_at) inner_reset = true;
                }
                if (!ok || inner_reset) st->fallback = 2;
            } else if (!ok) {
                st->fallback = 1;
            }
        }
        tot[0] = tot[1] = tot[2] = 0;                 // block 0 starts at rank 0
    }
    uint32_t o0 = (uint32_t)tot[0], o1 = (uint32_t)tot[1], o2 = (uint32_t)tot[2];
    for (int k = 0; k < w; ++k) { o0 += wsum[0][k]; o1 += wsum[1][k]; o2 += wsum[2][k]; }
    if (i < g.n) {
        poprank[i] = o0 + (uint32_t)__popcll(m0 & below);
        pushrank[i] = o1 + (uint32_t)__popcll(m1 & below);
        clkrank[i] = o2 + (uint32_t)__popcll(m2 & below);
    }
}

__device__ inline int32_t resolve_dic(int32_t d, const uint32_t *fifo, uint32_t cap, uint32_t head, const uint32_t *poprank,
                                      uint32_t n = 0xffffffffu) {
    if ((uint32_t)d & kSur) {
        uint32_t pos = (uint32_t)d & ~kSur;
        if (pos >= n) return 0;
        return (int32_t)fifo[(head + poprank[pos]) % cap];
    }
    return d;
}

// commit: staged buckets -> tables with placeholders resolved; pushes -> free list.
// Workgroups walk the partitions' staging regions (main partitions, then LRU partitions), then the batch for pushes.
__global__ void __launch_bounds__(256) k_commit(Geo g, MainBucket *mt, LruBucket *lt, uint32_t *fifo, uint32_t cap,
                                                DevState *st, const Ctl *ctl, const uint32_t *__restrict__ off,
                                                const uint32_t *__restrict__ stage_tot, const MainBucket *__restrict__ stage_m,
                                                const LruBucket *__restrict__ stage_l, const uint8_t *__restrict__ flags,
                                                const int32_t *__restrict__ pushval, const uint32_t *__restrict__ poprank,
                                                const uint32_t *__restrict__ pushrank, const uint32_t *__restrict__ clkrank,
                                                const int2 *__restrict__ hfin) {
    if (st->fallback) return;
    const uint32_t head = st->fifo_head, len0 = st->fifo_len;
    const int32_t clk0 = st->clk;
    const int nparts = g.P1 + (ctl->final_round > 1 ? g.P2 : 0);      // round 1 final = nothing missed, no LRU work
    for (int part = blockIdx.x; part < nparts; part += gridDim.x) {
        int tot = (int)stage_tot[part];
        // Everything below indexes the tables with numbers read from the staging buffers.  They are written by the
        // owners of this very call; the range checks turn a broken invariant into a counted refusal, not a wild store.
        const int room = part < g.P1 ? (int)(off[part + 1] - off[part]) + kCapEv
                                     : (int)(off[g.P1 + 2 + (part - g.P1)] - off[g.P1 + 1 + (part - g.P1)]);
        if (tot < 0 || tot > room) { if (threadIdx.x == 0) atomicAdd(&st->bad_commit, 1); tot = 0; }
        if (part < g.P1) {
            const MainBucket *stage = stage_m + off[part] + (size_t)part * kCapEv;
            for (int s = threadIdx.x; s < tot; s += blockDim.x) {
                const int4 *p = reinterpret_cast<const int4 *>(&stage[s]);
                const int4 m = p[3];
                if (m.y < 0) continue;
                if (m.y >= g.n1) { atomicAdd(&st->bad_commit, 1); continue; }
                int4 v = p[0], d = p[1], c = p[2];
                d.x = resolve_dic(d.x, fifo, cap, head, poprank, (uint32_t)g.n); d.y = resolve_dic(d.y, fifo, cap, head, poprank, (uint32_t)g.n);
                d.z = resolve_dic(d.z, fifo, cap, head, poprank, (uint32_t)g.n); d.w = resolve_dic(d.w, fifo, cap, head, poprank, (uint32_t)g.n);
                int4 *o = reinterpret_cast<int4 *>(&mt[m.y]);
                o[0] = v; o[1] = d; o[2] = c;
                mt[m.y].flip = (uint32_t)m.x;
            }
            if (g.hot) {
                // final scores of the bypassed hot slots of this partition, behind its staged buckets (which carry
                // the pre-batch score in those slots)
                const uint32_t oh0 = off[g.P1 + g.P2 + 2 + part];
                const int nh = (int)(off[g.P1 + g.P2 + 3 + part] - oh0);
                // both writers are waves of this workgroup (one CU, one L2): once the staged stores are acknowledged,
                // later stores to the same words land behind them
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                for (int s = threadIdx.x; s < nh; s += blockDim.x) {
                    const int2 f = hfin[oh0 + s];
                    if (f.x >= 0 && (f.x >> 2) >= g.n1) { atomicAdd(&st->bad_commit, 1); continue; }
                    if (f.x >= 0) mt[f.x >> 2].cnt[f.x & 3] = __int_as_float(f.y);
                }
            }
        } else {
            const LruBucket *stage = stage_l + off[g.P1 + 1 + (part - g.P1)];
            for (int s = threadIdx.x; s < tot; s += blockDim.x) {
                const int4 *p = reinterpret_cast<const int4 *>(&stage[s]);
                const int4 m = p[4];
                if (m.y < 0) continue;
                int4 v = p[0], d = p[1], c = p[2], t = p[3];
                const long tmax = (long)clk0 + g.n;
                if (m.y >= g.n2 || t.x > tmax || t.y > tmax || t.z > tmax || t.w > tmax) { atomicAdd(&st->bad_commit, 1); continue; }
                d.x = resolve_dic(d.x, fifo, cap, head, poprank, (uint32_t)g.n); d.y = resolve_dic(d.y, fifo, cap, head, poprank, (uint32_t)g.n);
                d.z = resolve_dic(d.z, fifo, cap, head, poprank, (uint32_t)g.n); d.w = resolve_dic(d.w, fifo, cap, head, poprank, (uint32_t)g.n);
                if (t.x > clk0) t.x = clk0 + 1 + (int32_t)clkrank[t.x - clk0 - 1];
                if (t.y > clk0) t.y = clk0 + 1 + (int32_t)clkrank[t.y - clk0 - 1];
                if (t.z > clk0) t.z = clk0 + 1 + (int32_t)clkrank[t.z - clk0 - 1];
                if (t.w > clk0) t.w = clk0 + 1 + (int32_t)clkrank[t.w - clk0 - 1];
                int4 *o = reinterpret_cast<int4 *>(&lt[m.y]);
                o[0] = v; o[1] = d; o[2] = c; o[3] = t;
                lt[m.y].flip = (uint32_t)m.x;
            }
        }
    }
    if (ctl->totals[1] == 0) return;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += (long)gridDim.x * blockDim.x) {
        if (flags[i] & 2) {
            // pushes land behind everything the list held; pops only ever take what it held at the start
            int32_t d = resolve_dic(pushval[i], fifo, cap, head, poprank, (uint32_t)g.n);
            fifo[(unsigned long long)((unsigned long long)head + len0 + pushrank[i]) % cap] = (uint32_t)d;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
static void pick_parts(int n, int buckets, int target, int &P, int &bpp) {
    int want = std::max(1, (n + target - 1) / target);
    P = std::min(std::min(want, buckets), kMaxBins);
    bpp = (buckets + P - 1) / P;
    P = (buckets + bpp - 1) / bpp;
}

static int ensure_ws(cafe_sketch *s, int n, Geo &g) {
    if (!s->par) s->par = new ParWorkspace();
    ParWorkspace &w = *s->par;
    g.n = n; g.n1 = s->n1; g.n2 = s->n2;
    pick_parts(n, s->n1, s->opt_target_items, g.P1, g.bpp1);
    pick_parts(n, s->n2, s->opt_target_items, g.P2, g.bpp2);
    g.fm1 = make_fastmod(s->n1);
    g.fm2 = make_fastmod(s->n2);
    auto bits = [](int v) { int b = 1; while ((1 << b) < v) ++b; return b; };
    g.end_bit1 = 39 + bits(g.bpp1);
    g.end_bit2 = 40 + bits(g.bpp2);
    const size_t N = (size_t)n, PP = (size_t)g.P1 + g.P2;
    const int nblocks = (n + kScanBlock - 1) / kScanBlock;
    g.hot = s->opt_hot_bypass ? 1 : 0;
    g.gflip = (uint32_t)s->flip;
    g.seg = s->seg_len > 0 && (n + s->seg_len - 1) / s->seg_len <= kMaxSeg ? s->seg_len : 0;
    g.verify = s->verify_mode && g.seg > 0 ? 1 : 0;
    g.big = g.big_hot = 1;
    g.last_round = 0;
    if (w.zr.reserve(4 * (kCtlWords + PP + 3 * (size_t)g.P1) + 16) || w.off.reserve(4 * (PP + g.P1) + 32) || w.rank.reserve(8 * N) ||
        w.items.reserve(3 * sizeof(Rec) * N) || w.hkey.reserve(8 * N + 16) || w.hrun.reserve(4 * N + 16) || w.hfin.reserve(8 * N + 16) || w.ev.reserve(2 * sizeof(MoveEv) * (size_t)g.P1 * kCapEv) ||
        w.evres.reserve(sizeof(EvRes) * N) || w.miss.reserve(N) || w.outm.reserve(N) || w.outl.reserve(N) ||
        w.stage_m.reserve(sizeof(MainBucket) * (N + (size_t)g.P1 * kCapEv)) || w.stage_l.reserve(sizeof(LruBucket) * N) ||
        w.stage_tot.reserve(4 * PP + 4) || w.prevcnt.reserve(4 * (size_t)g.P1 + 4) || w.ldirty.reserve(2 * (size_t)g.P2 + 4) ||
        w.myev.reserve(sizeof(MoveEv) * (size_t)g.P2 * kCapEvL) || w.mycnt.reserve(4 * (size_t)g.P2 + 4) || w.flags.reserve(5 * N + 64) ||
        w.blocksum.reserve(16 * (size_t)nblocks + 16) || w.ranks.reserve(12 * N))
        return 1;
    if (w.evres.p != w.evres_seen || w.evres.cap != w.evres_seen_cap) {      // tags of a fresh allocation must not match by accident
        CAFE_HIP_CHECK(hipMemset(w.evres.p, 0, w.evres.cap));
        CAFE_HIP_CHECK(hipDeviceSynchronize());
        w.evres_seen = w.evres.p;
        w.evres_seen_cap = w.evres.cap;
    }
    return 0;
}

int par_prepare(cafe_sketch *s, int n, uint32_t **zero, int *zero_words) {
    *zero = nullptr; *zero_words = 0;
    if (n > (1 << 21)) return 0;          // such a batch goes to the in-order kernel (23-bit time field of the sort keys)
    Geo g;
    if (ensure_ws(s, n, g)) return 1;
    *zero = s->par->zr.as<uint32_t>();
    *zero_words = (int)(kCtlWords + (size_t)g.P1 + g.P2 + 3 * (size_t)g.P1);
    return 0;
}

// How many rounds to enqueue and whether to enqueue the big-partition variants are guesses from what recent calls
// needed (a pinned copy of the device state trails every insert).  A wrong guess is never a wrong result: a batch
// that is not settled, or meets a partition without an owner kernel, is handed to the in-order kernel untouched.
static void launch_plan(cafe_sketch *s, int n, int &rounds, bool &big, bool &big_hot) {
    const int max_rounds = std::min(std::max(s->opt_max_rounds, 1), kMaxRounds) + 1;
    const DevState *fb = reinterpret_cast<const DevState *>(s->h_pinned);
    rounds = max_rounds;
    big = true;
    big_hot = true;
    // A batch that left the parallel path (pad[0] counts them) means the guesses were too tight for what the data
    // does now -- a popular id that is new, a shift of the distribution: stay generous for a while.
    if (fb->pad[0] != s->seen_inorder) { s->seen_inorder = fb->pad[0]; s->generous_calls = 256; }
    if (s->last_seen_batch != n) {         // other batch size: forget what was learnt
        for (int &h : s->hint_rounds) h = 0;
        for (int &h : s->hint_part) h = 0;
        for (int &h : s->hint_hot) h = 0;
        s->last_seen_batch = n;
        return;
    }
    if (fb->fallback || fb->rounds <= 0) {
        for (int &h : s->hint_rounds) h = 0;
        return;
    }
    s->round_hist[std::min(std::max(fb->rounds, 0), 17)]++;
    s->hint_rounds[s->hint_pos & 7] = fb->rounds;
    s->hint_part[s->hint_pos & 7] = fb->maxpart;
    s->hint_hot[s->hint_pos & 7] = fb->pad[3];
    s->hint_pos++;
    int known = 0, most = 0, part = 0, hot = 0;
    for (int h : s->hint_rounds) { known += h > 0; most = std::max(most, h); }
    for (int h : s->hint_part) part = std::max(part, h);
    for (int h : s->hint_hot) hot = std::max(hot, h);
    if (s->generous_calls > 0) { s->generous_calls--; return; }
    if (known >= 4) rounds = std::min(max_rounds, most + 2);
    if (known >= 4 && part > 0 && (long)part * 3 / 2 + 256 <= kCapSmall) big = false;
    if (known >= 4 && (long)hot * 3 / 2 <= kCapHot) big_hot = false;
}

int par_insert(cafe_sketch *s, const int32_t *d_ids, const float *d_scores, int n, int32_t *d_ins, hipStream_t st,
               hipStream_t pre) {
    Geo g;
    if (n > (1 << 21)) {     // the 23-bit time field of the owners' sort key
        s->counters[2]++;
        if (pre != st) {
            CAFE_HIP_CHECK(hipEventRecord(s->ev_head_join, pre));
            CAFE_HIP_CHECK(hipStreamWaitEvent(st, s->ev_head_join, 0));
        }
        return launch_serial_insert(s, d_ids, d_scores, n, d_ins, 0, st);
    }
    if (ensure_ws(s, n, g)) return 1;      // sized by par_prepare already; the prologue kernel zeroed w.zr
    ParWorkspace &w = *s->par;
    w.call_seq++;
    const unsigned long long tag = w.call_seq;
    Ctl *ctl = w.zr.as<Ctl>();
    uint32_t *cnt = w.zr.as<uint32_t>() + kCtlWords, *evcnt = cnt + g.P1 + g.P2, *cnth = evcnt + 2 * g.P1;
    uint32_t *off = w.off.as<uint32_t>();
    uint32_t *rank1 = w.rank.as<uint32_t>(), *rank2 = rank1 + n;
    Rec *rec1 = w.items.as<Rec>(), *rec2 = rec1 + n, *rech = rec2 + n;
    unsigned long long *hkey = w.hkey.as<unsigned long long>();
    float *hrun = w.hrun.as<float>();
    int2 *hfin = w.hfin.as<int2>();
    uint32_t *stage_tot = w.stage_tot.as<uint32_t>();
    const int gb = (n + 255) / 256;
    hipLaunchKernelGGL(k_assign, dim3((n + kAssignChunk - 1) / kAssignChunk), dim3(1024),
                       4 * (size_t)(g.P1 + g.P2 + (g.hot ? g.P1 : 0)), pre, g, s->d_main, d_ids, cnt, cnth, rank1, rank2);
    int rounds;
    bool big, big_hot;
    launch_plan(s, n, rounds, big, big_hot);
    // the big-partition owners are only enqueued when recent calls needed them; the normal owners flag a partition that
    // is too large for them and has no other owner, which sends the batch to the in-order kernel
    g.big = big ? 1 : 0;
    g.big_hot = big_hot ? 1 : 0;
    g.last_round = rounds;
    hipLaunchKernelGGL(k_offsets, dim3(g.hot ? 3 : 2), dim3(1024), 0, pre, g, cnt, cnth, off, ctl, s->d_state);
    if (pre != st) {         // the scores come in here
        CAFE_HIP_CHECK(hipEventRecord(s->ev_head_join, pre));
        CAFE_HIP_CHECK(hipStreamWaitEvent(st, s->ev_head_join, 0));
    }
    hipLaunchKernelGGL(k_scatter, dim3(gb), dim3(256), 0, st, g, d_ids, d_scores, off, rank1, rank2, rec1, rec2, rech);
    if (g.hot) {
        // The bypass is independent of round 1 (its results are first read by round 2's move handling and by the
        // finish kernels), so it runs beside round 1 on a stream of its own.
        hipStream_t hs = s->hot_on_side ? s->side_stream : st;
        if (hs != st) {
            CAFE_HIP_CHECK(hipEventRecord(s->ev_fork, st));
            CAFE_HIP_CHECK(hipStreamWaitEvent(hs, s->ev_fork, 0));
        }
        hipLaunchKernelGGL((k_hot_chain<kCapHot, false>), dim3(g.P1), dim3(kThreads), 0, hs, g, s->d_main, s->d_state, ctl, off,
                           rech, hkey, hrun, hfin, w.miss.as<uint8_t>(), w.outm.as<uint8_t>());
        if (big_hot)
            hipLaunchKernelGGL((k_hot_chain<kCapBig, true>), dim3(g.P1), dim3(kThreads), 0, hs, g, s->d_main, s->d_state, ctl, off,
                               rech, hkey, hrun, hfin, w.miss.as<uint8_t>(), w.outm.as<uint8_t>());
        if (hs != st) CAFE_HIP_CHECK(hipEventRecord(s->ev_join, hs));
    }
#define CAFE_MAIN(CAPV, BIGV)                                                                                              \
    hipLaunchKernelGGL((k_main_round<CAPV, BIGV>), dim3(g.P1), dim3(kThreads), 0, st, g, r, tag, s->d_main, s->d_state, ctl, \
                       off, rec1, evcnt, w.ev.as<MoveEv>(), w.evres.as<EvRes>(), w.miss.as<uint8_t>(),    \
                       w.outm.as<uint8_t>(), w.stage_m.as<MainBucket>(), stage_tot, w.prevcnt.as<uint32_t>(), w.ldirty.as<uint8_t>(), \
                       hkey, hrun)
#define CAFE_LRU(CAPV, BIGV)                                                                                               \
    hipLaunchKernelGGL((k_lru_round<CAPV, BIGV>), dim3(g.P2), dim3(kThreads), 0, st, g, r, s->d_lru, s->d_state, ctl, \
                       off, rec2, w.miss.as<uint8_t>(), evcnt, w.ev.as<MoveEv>(), w.outl.as<uint8_t>(),          \
                       w.stage_l.as<LruBucket>(), stage_tot, w.ldirty.as<uint8_t>(), w.myev.as<MoveEv>(), w.mycnt.as<uint32_t>())
    for (int r = 1; r <= rounds; ++r) {
        if (r == 2 && g.hot && s->hot_on_side) CAFE_HIP_CHECK(hipStreamWaitEvent(st, s->ev_join, 0));
        CAFE_MAIN(kCapSmall, false);
        if (big) CAFE_MAIN(kCapBig, true);
        if (r == rounds) break;            // final_round_of
        CAFE_LRU(kCapSmall, false);
        if (big) CAFE_LRU(kCapBig, true);
    }
#undef CAFE_MAIN
#undef CAFE_LRU
    if (rounds < 2 && g.hot && s->hot_on_side) CAFE_HIP_CHECK(hipStreamWaitEvent(st, s->ev_join, 0));
    const int nb = (n + kScanBlock - 1) / kScanBlock;
    uint8_t *flags = w.flags.as<uint8_t>();
    int32_t *pushval = (int32_t *)(flags + (((size_t)n + 15) & ~(size_t)15));
    uint32_t *poprank = w.ranks.as<uint32_t>(), *pushrank = poprank + n, *clkrank = pushrank + n;
    hipLaunchKernelGGL(k_finish_flags, dim3(nb), dim3(kScanBlock), 0, st, g, tag, s->d_state, ctl, w.miss.as<uint8_t>(),
                       w.outm.as<uint8_t>(), w.outl.as<uint8_t>(), w.evres.as<EvRes>(), d_ins, flags, pushval,
                       w.blocksum.as<uint32_t>());
    hipLaunchKernelGGL(k_finish_ranks, dim3(nb), dim3(kScanBlock), 0, st, g, nb, s->adjust ? s->lim * 1.2 : 1e300, s->d_state, ctl, flags,
                       w.blocksum.as<uint32_t>(), poprank, pushrank, clkrank);
    const int cb = std::min(g.P1 + g.P2, 2048);
    hipLaunchKernelGGL(k_commit, dim3(cb), dim3(256), 0, st, g, s->d_main, s->d_lru, s->d_fifo, s->fifo_cap, s->d_state, ctl,
                       off, stage_tot, w.stage_m.as<MainBucket>(), w.stage_l.as<LruBucket>(), flags, pushval, poprank,
                       pushrank, clkrank, hfin);
    CAFE_HIP_CHECK(hipGetLastError());
    // closing kernel: applies the batch totals to the device scalars, or, for a batch the owners could not settle,
    // runs it in order from the untouched tables
    if (launch_serial_insert(s, d_ids, d_scores, n, d_ins, 1, st, ctl->totals, ctl->segcross)) return 1;
    s->counters[1]++;
    return 0;
}

void par_free(cafe_sketch *s) {
    if (!s->par) return;
    ParWorkspace &w = *s->par;
    DevBuf *all[] = {&w.zr, &w.off, &w.rank, &w.items, &w.ev, &w.evres, &w.miss, &w.outm, &w.outl,
                     &w.stage_m, &w.stage_l, &w.stage_tot, &w.prevcnt, &w.ldirty, &w.myev, &w.mycnt, &w.flags, &w.blocksum, &w.ranks,
                     &w.hkey, &w.hrun, &w.hfin};
    for (DevBuf *b : all) b->release();
    delete s->par;
    s->par = nullptr;
}

}  // namespace cafe
