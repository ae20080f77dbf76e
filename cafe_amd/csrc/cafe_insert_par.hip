// cafe_insert_par.hip -- parallel, bit-exact batch insert for the HotSketch (sketch.cpp:281-296).
//
// The reference inserts a batch strictly in order; items interact through (a) their main bucket,
// (b) their LRU bucket, (c) the main bucket an entry leaving the LRU table is moved into, and through
// the global free list, LRU clock and counters.  This file reproduces that order exactly without
// walking the batch serially, as a fixed sequence of seven launches:
//
//   plan (reads ids and the pre-batch main table only; runs beside the score kernel)
//     k_assign   every id is classified with one 32-B probe of its main bucket --
//                  H  sits in the bucket with a hot slot: a guaranteed hit for the whole batch (hot slots are never
//                     replaced, sketch.cpp:253-258); only its ordered score sum matters            -> hot list
//                  C  sits in the bucket without a hot slot (or the bucket's lazy decay is pending): the bucket has
//                     to be replayed in batch order                                                -> main list, front
//                  M  not in the bucket: a miss unless a move brings the id in earlier in the batch -> main list, back
//                and ranked, in batch order, inside its main partition and its LRU partition (a stable
//                multi-split: per-wave LDS histograms, match-any by ballots, no atomics)
//     k_offsets  column sums over the chunks -> partition offsets; per-call scalars (sketch.cpp:282-289)
//     k_scatter  16-B item records written to their partitions, in batch order inside each partition
//   rounds
//     k_round1   one launch, three kinds of workgroups: hot partitions (ordered sums), LRU partitions (all presumed
//                misses, replayed in batch order) and main partitions (class C items only).  Owners sort their
//                partition by bucket with a stable LDS counting sort and replay each bucket on a register copy.
//     k_fix      entries that leave the LRU table (moves, ~tens per batch) change main buckets, which can turn
//                presumed misses into hits and hits into misses.  One small persistent grid alternates main owners
//                (partitions that received moves) and LRU owners (partitions whose miss set changed) until the miss
//                set repeats.  An item's outcome only depends on earlier items, so every round fixes a longer prefix.
//   finish
//     k_finish   batch-ordered ranks of free-list pops, pushes and clock ticks (one pass, decoupled look-back)
//     k_commit   decision (free-list assumption, sketch.cpp:291 at inner boundaries), staged buckets -> tables with
//                placeholders resolved, device scalars, snapshot for the host; the last workgroup closes the call
//                (and, for a batch that could not be settled, runs it in order from the untouched tables).
//
// Nothing is written to the tables before the decision in k_commit.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstring>

#include <algorithm>

#include "cafe_internal.h"

namespace cafe {

CAFE_KWIN_DECL(cafe_debug_windows_insert)

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kChunk = 1024;          // ids per plan workgroup
// Smallest owner instantiation.  Its LDS (17 B per element + 0.5 KB) decides how many round-1 owners a compute unit holds
// (the 160 KB are two halves of 80 KB and an allocation lies in one of them: measured 8 per CU up to 20 KB, 6 up to
// 26.6 KB, 4 up to 40 KB): 1536 elements = 26.6 KB = 6 per CU = 1536 owners at once, the ~1430 of a 213 K-id batch.
constexpr int kCapSmall = 1536;
constexpr int kPlanThreads = 512;     // 8 waves, 256 consecutive ids each
constexpr int kPlanWaves = kPlanThreads / 64;
constexpr int kPlanRows = kChunk / kPlanThreads;     // rows of 64 ids a plan wave walks
constexpr int kMaxParts = 2048;       // partitions per table: 2 P1 + P2 + PH <= 8192 bins of the plan histogram (LDS: 8 waves x 8192 x 2 B)
constexpr int kMaxPlanBins = 8192;
constexpr int kHotKeys = 1024;        // slots (4 per bucket) of a hot partition that fit the per-wave LDS table
constexpr int kCapEv = 512;           // moves one main partition can receive per round
constexpr int kCapEvL = 1024;         // moves one LRU partition can emit per round (an item emits at most one)
constexpr int kMaxRounds = 16;
constexpr int kHeavySeg = 4;          // a bucket with at least this many items in the batch is replayed by a whole wave
constexpr int kLruHeavy = 32;         // an LRU bucket with at least this many items in the batch is replayed by a whole wave
constexpr int kLruQuad = 8;           // ... with at least this many, and no long runs of one id, by four lanes (one per slot)
constexpr int kQuadTag = 0x40000000;  // heavy-list entry: replayed by a lane quad, not by a wave
constexpr int kMaxHeavy = 64;         // heavy buckets one partition can hand to its waves (more: replayed by single lanes)
constexpr int kScanBlock = 1024;
constexpr int kDigitBits = 10, kDigits = 1 << kDigitBits;
constexpr int kFixGrid = 64;          // workgroups of the persistent fix-point kernel (more when many partitions are in play)

constexpr uint32_t kSur = 0x80000000u;      // placeholder for "slot popped by item (v & 0x7fffffff)"
constexpr uint32_t kMissBit = 0x80000000u;  // Rec::bl: the id is not in its main bucket before the batch

// outcome bits per item
constexpr uint8_t oGot = 1, oPop = 2, oCross = 4, oMiss = 8, oEv = 16;   // oEv: the item's LRU step emitted a move

struct MoveEv {             // a move into the main table emitted by an LRU owner
    uint32_t time;          // 2 * position + 1: after the item's own probe
    uint32_t x;
    float cnt;
    int32_t dic;
    int32_t kind;
    int32_t pad[3];         // [0]: main partition of the target bucket
};

struct EvRes {              // outcome of a move, written by the main owner that carried it out
    unsigned long long tag; // call sequence number
    int32_t kind, ok, pushed, pad;
};

struct Ctl {                // zeroed by k_offsets at the start of every call
    uint32_t changed[kMaxRounds + 2];
    uint32_t overflow;      // a workspace limit was hit
    uint32_t final_round;   // round whose main half reproduced the miss set (0: not settled)
    uint32_t maxpart;       // largest owner partition (items) of this call
    uint32_t maxhot;        // largest hot-bypass partition
    uint32_t totals[4];     // pops, pushes, clock ticks, threshold crossings (k_finish)
    uint32_t segcross[kMaxSeg];   // threshold crossings per segment of g.seg ids (a segment = one per-field reference call)
    uint32_t nmoves;        // moves emitted by LRU round 1
    uint32_t ngot;          // entries of the got list
    uint32_t nclassc;       // class C items (diagnostics)
    uint32_t ticket_off, ticket_commit;
    uint32_t ticket_sub[32 * 16];   // first level of the commit ticket, one counter per 64-byte line
    uint32_t bar_count, bar_gen;
    uint32_t pad;
};
constexpr size_t kCtlWords = (sizeof(Ctl) + 3) / 4;

struct ParWorkspace {
    DevBuf ctl;             // [Ctl][evcnt: 2*P1]
    DevBuf cm, colpref, tot, off, plan, items, ev, evres, miss, outm, outl, stage_m, stage_l, stage_tot, prevcnt;
    DevBuf hkey, hrun, hfin;
    DevBuf items_sorted, pre_nmiss;      // k_presort: the partitions' records in owner order, misses per LRU partition
    DevBuf r1_order;                     // k_offsets: the round-1 owners by estimated cost, heaviest first
    DevBuf ldirty, myev, mycnt;
    DevBuf flags, agg, ranks, gotlist;
    void *evres_seen = nullptr;
    size_t evres_seen_cap = 0;
    unsigned long long call_seq = 0;
};

struct Geo {
    int n, n1, n2, P1, P2, bpp1, bpp2;
    int bits1, bits2;           // bits of a bucket index inside a main / LRU partition
    int nchunks, nbins;         // plan geometry: nbins = 2 P1 + P2 + PH
    int hot;                    // 1: ids that sit hot in the main table before the batch bypass the owner rounds
    int PH, bpph, bitsh;        // hot list: partitions, buckets per partition, bits of a bucket index inside one
    int hot_table;              // 1: a hot partition's slots fit the per-wave LDS table (no sort; one wave per partition)
    uint32_t gflip;             // flip bit of this call (known on the host)
    int flip_now;               // sketch.cpp:286-288 happen in this call
    int seg;                    // > 0: ids per reference call when this call stands for several (cafe_backward_update_dev)
    int max_rounds;
    int verify;                 // 1: a batch that cannot be settled, or holds a reset boundary, is REJECTED (nothing is
                                // applied, the host issues it again) instead of going to the in-order loop
    int sseg;                   // > 0: ids per field for the score scaling (Score::psum)
    int nparts;                 // partial sums per field behind Score::psum
    int prof;                   // diagnostic builds: stamp slot of the kernel
    int cap_ev;                 // moves one main partition accepts per round (<= evs; smaller in tests)
    int evs, evls;              // move-list slots per main partition (in) and per LRU partition (out): kCapEv / kCapEvL,
                                // eight times that when a small table is cut into few, large partitions
    FastMod fm1, fm2;
};

// bins of the plan histogram: [0, 2 P1) main partitions, class C at 2p, class M at 2p + 1; [2 P1, 2 P1 + P2) LRU
// partitions; [2 P1 + P2, 2 P1 + P2 + PH) hot partitions
__device__ __host__ inline int bin_lru0(const Geo &g) { return 2 * g.P1; }
__device__ __host__ inline int bin_hot0(const Geo &g) { return 2 * g.P1 + g.P2; }
// offsets: one array, [0 .. 2 P1] main list, then [0 .. P2] LRU list, then [0 .. PH] hot list (each list starts at 0)
__device__ __host__ inline int off_lru0(const Geo &g) { return 2 * g.P1 + 1; }
__device__ __host__ inline int off_hot0(const Geo &g) { return 2 * g.P1 + 1 + g.P2 + 1; }

// What an owner needs to know about one of its items, written once per call by k_scatter into the item's
// partition region, in batch order, so that owners read one coalesced stream.
struct __attribute__((aligned(16))) Rec {
    uint32_t pos;           // batch position
    uint32_t x;             // id
    uint32_t bl;            // bucket index inside the partition (hot list: << 2 | slot); bit 31: presumed miss
    uint32_t pad;
};

// where the ids come from: global ids, or the caller's per-field ids with the fields' offsets (sk_embedding_bag.py:103),
// so that the plan can run beside the forward kernel that writes the global ids
struct IdSrc {
    const int32_t *gids;
    const int64_t *local_ids;
    const int64_t *id_offset;
    int batch;
    __device__ uint32_t at(int i) const {
        return local_ids ? (uint32_t)(int32_t)(local_ids[i] + id_offset[i / batch]) : (uint32_t)gids[i];
    }
};

// where the scores come from: explicit values, or per-lookup norms with per-field partial sums (the scaling of
// sk_embedding_bag.py:189 is then applied where a score is consumed)
struct Score {
    const float *vals;      // [n] scores or norms; nullptr: every score is 1 (batch_insert)
    const float *psum;      // [F][nparts] partial sums of the norms per field, or nullptr
    int vstride;            // floats between two entries of vals (1: contiguous)
};

// Diagnostic build (make EXTRA=-DCAFE_PROF): wave 0 of every round-1 workgroup stamps the 100 MHz wall clock at phase
// boundaries; tools/diag_stamps.py reads them back.  No code in the normal build.
#ifdef CAFE_PROF
constexpr int kProfSlots = 8, kProfBlocks = 4096;
__device__ unsigned long long g_dbg[kProfSlots][kProfBlocks][8];
#define CAFE_STAMP(slot, k) do { if (threadIdx.x == 0) g_dbg[slot][blockIdx.x & (kProfBlocks - 1)][k] = wall_clock64(); } while (0)
#define CAFE_STAMP_VAL(slot, k, v) do { if (threadIdx.x == 0) g_dbg[slot][blockIdx.x & (kProfBlocks - 1)][k] = (v); } while (0)
extern "C" int cafe_debug_read(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(g_dbg)) != hipSuccess; }
#else
#define CAFE_STAMP(slot, k) do { } while (0)
#define CAFE_STAMP_VAL(slot, k, v) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------------
// lanes of the wave that are `valid` and hold the same value v (its low `bits` bits)
__device__ inline unsigned long long match_any(uint32_t v, int bits, bool valid) {
    unsigned long long m = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
        const bool bit = (v >> b) & 1u;
        const unsigned long long bb = __ballot(bit);
        m &= bit ? bb : ~bb;
    }
    return m;
}
__device__ inline unsigned long long lanes_below() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// ------------------------------------------------------------------------------------------------
// plan: classify and rank (stable multi-split), offsets, scatter
// ------------------------------------------------------------------------------------------------
// plan word A: bits 0-10 rank inside (chunk, bin), 11-23 bin, 24-25 slot (hot), 26 presumed miss; word B (LRU list):
// bits 0-10 rank, 11-23 bin
__global__ void __launch_bounds__(kPlanThreads) k_assign(Geo g, const MainBucket *__restrict__ mt, IdSrc ids,
                                                         uint16_t *__restrict__ cm /* [nchunks][nbins] */,
                                                         uint32_t *__restrict__ planA, uint32_t *__restrict__ planB) {
    CAFE_KWIN(0);
    extern __shared__ uint16_t wh[];            // [kPlanWaves][nbins]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = g.nbins;
    CAFE_STAMP(4, 0);
    for (int b = tid; b < (kPlanWaves * nb + 1) / 2; b += kPlanThreads) reinterpret_cast<uint32_t *>(wh)[b] = 0;
    __syncthreads();
    CAFE_STAMP(4, 1);
    uint16_t *mine = wh + (size_t)wave * nb;
    const int bitsb = 32 - __builtin_clz((unsigned)(nb | 1));
    const unsigned long long below = lanes_below();
    uint32_t wa[kPlanRows], wb[kPlanRows];
    const int base = blockIdx.x * kChunk + wave * (kChunk / kPlanWaves);
    // the ids of all rows, then the probes of all rows, are requested before anything is looked at: two memory round
    // trips per wave instead of two per row
    uint32_t xs[kPlanRows], fl[kPlanRows];
    int b1s[kPlanRows];
    int4 vs[kPlanRows], ds[kPlanRows];
#pragma unroll
    for (int r = 0; r < kPlanRows; ++r) {
        const int i = base + r * 64 + lane;
        xs[r] = i < g.n ? ids.at(i) : 0u;
    }
#pragma unroll
    for (int r = 0; r < kPlanRows; ++r) {
        b1s[r] = fast_hash(xs[r], g.fm1);
        const int4 *p = reinterpret_cast<const int4 *>(&mt[b1s[r]]);
        vs[r] = p[0]; ds[r] = p[1]; fl[r] = mt[b1s[r]].flip;
    }
    if (xs[0] == 0xdeadbeefu && vs[0].x == 0x1234567) CAFE_STAMP(4, 7);      // (diagnostic build: forces the loads to land before the stamp)
    CAFE_STAMP(4, 2);
#pragma unroll
    for (int r = 0; r < kPlanRows; ++r) {
        const int i = base + r * 64 + lane;
        const bool valid = i < g.n;
        uint32_t binA = 0, binB = 0, extra = 0;
        bool has_b = false;
        if (valid) {
            const uint32_t x = xs[r];
            const int b1 = b1s[r];
            // the reference stops at the first slot holding the id (sketch.cpp:214-215): a hot copy behind a cold one is not hit
            int slot = -1, dic = 0;
            if ((uint32_t)vs[r].w == x) { slot = 3; dic = ds[r].w; }
            if ((uint32_t)vs[r].z == x) { slot = 2; dic = ds[r].z; }
            if ((uint32_t)vs[r].y == x) { slot = 1; dic = ds[r].y; }
            if ((uint32_t)vs[r].x == x) { slot = 0; dic = ds[r].x; }
            const bool pending = fl[r] != g.gflip;       // the bucket's lazy decay is due
            const bool hot = slot >= 0 && dic != 0;
            const int p1 = b1 / g.bpp1;
            if (g.hot && hot && !pending) {
                binA = (uint32_t)(bin_hot0(g) + b1 / g.bpph);
                extra = (uint32_t)slot << 24;
            } else {
                const bool c = pending || slot >= 0;         // the bucket has to be replayed in round 1
                binA = (uint32_t)(2 * p1 + (c ? 0 : 1));
                if (slot < 0) extra = 1u << 26;               // presumed miss
                binB = (uint32_t)(bin_lru0(g) + fast_hash(x, g.fm2) / g.bpp2);
                has_b = true;
            }
        }
        {
            const unsigned long long m = match_any(binA, bitsb, valid);
            uint32_t old = 0;
            if (valid && !(m & below)) { old = mine[binA]; mine[binA] = (uint16_t)(old + __popcll(m)); }
            old = __shfl(old, m ? __builtin_ctzll(m) : 0, 64);
            wa[r] = (old + (uint32_t)__popcll(m & below)) | (binA << 11) | extra;
        }
        {
            const unsigned long long m = match_any(binB, bitsb, has_b);
            uint32_t old = 0;
            if (has_b && !(m & below)) { old = mine[binB]; mine[binB] = (uint16_t)(old + __popcll(m)); }
            old = __shfl(old, m ? __builtin_ctzll(m) : 0, 64);
            wb[r] = has_b ? ((old + (uint32_t)__popcll(m & below)) | (binB << 11)) : 0xffffffffu;
        }
    }
    CAFE_STAMP(4, 3);
    __syncthreads();
    CAFE_STAMP(4, 4);
    // per bin: exclusive prefix over the waves (batch order), chunk total to the column matrix
    for (int b = tid; b < nb; b += kPlanThreads) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < kPlanWaves; ++w) {
            const uint32_t c = wh[(size_t)w * nb + b];
            wh[(size_t)w * nb + b] = (uint16_t)run;
            run += c;
        }
        cm[(size_t)blockIdx.x * nb + b] = (uint16_t)run;
    }
    __syncthreads();
    CAFE_STAMP(4, 5);
#pragma unroll
    for (int r = 0; r < kPlanRows; ++r) {
        const int i = base + r * 64 + lane;
        if (i >= g.n) continue;
        planA[i] = wa[r] + mine[(wa[r] >> 11) & 0x1fffu];
        if (wb[r] != 0xffffffffu) planB[i] = wb[r] + mine[(wb[r] >> 11) & 0x1fffu];
    }
    CAFE_STAMP(4, 6);
}

// Column prefix sums over the chunks: a workgroup takes 32 bins, eight lanes share a bin's column.  The workgroup that
// finishes last turns the bin totals into partition offsets, zeroes the control block and applies the per-call scalars
// of sketch.cpp:282-289 (the flip decision is the host's: it depends on the call count only).
constexpr int kOffBins = 32, kOffParts = kThreads / kOffBins;
constexpr int kCostClasses = 256;
__global__ void __launch_bounds__(kThreads) k_offsets(Geo g, const uint16_t *__restrict__ cm, uint32_t *__restrict__ colpref,
                                                      uint32_t *__restrict__ tot, uint32_t *__restrict__ off, Ctl *ctl,
                                                      uint32_t *__restrict__ evcnt, DevState *st, uint32_t *__restrict__ order) {
    CAFE_KWIN(1);
    __shared__ uint32_t sh[kThreads];
    __shared__ uint32_t chist[kCostClasses];
    __shared__ uint32_t excl[kMaxPlanBins + 1];
    __shared__ uint32_t last;
    const int tid = threadIdx.x;
    CAFE_STAMP(3, 0);
    {
        const int b = blockIdx.x * kOffBins + (tid & (kOffBins - 1)), part = tid / kOffBins;
        const int cpp = (g.nchunks + kOffParts - 1) / kOffParts;
        const int c0 = part * cpp, c1 = min(c0 + cpp, g.nchunks);
        const bool live = b < g.nbins;
        const int bb = live ? b : 0;
        if (cpp <= 32) {
            // the column piece stays in registers: one memory round trip
            uint32_t v[32], sum = 0;
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = cm[(size_t)min(c0 + u, g.nchunks - 1) * g.nbins + bb];
#pragma unroll
            for (int u = 0; u < 32; ++u) { v[u] = (c0 + u < c1) ? v[u] : 0u; sum += v[u]; }
            sh[tid] = live ? sum : 0u;
            __syncthreads();
            CAFE_STAMP(3, 1);
            uint32_t run = 0, total = 0;
            for (int k = 0; k < kOffParts; ++k) {
                const uint32_t x = sh[k * kOffBins + (tid & (kOffBins - 1))];
                if (k < part) run += x;
                total += x;
            }
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (live && c0 + u < c1) { colpref[(size_t)(c0 + u) * g.nbins + b] = run; run += v[u]; }
            if (live && part == 0) __hip_atomic_store(&tot[b], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            uint32_t sum = 0;
            for (int c = c0; c < c1; c += 8) {          // eight independent loads per step
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = cm[(size_t)min(c + u, g.nchunks - 1) * g.nbins + bb];
#pragma unroll
                for (int u = 0; u < 8; ++u) sum += (c + u < c1) ? v[u] : 0u;
            }
            sh[tid] = live ? sum : 0u;
            __syncthreads();
            CAFE_STAMP(3, 1);
            uint32_t run = 0, total = 0;
            for (int k = 0; k < kOffParts; ++k) {
                const uint32_t x = sh[k * kOffBins + (tid & (kOffBins - 1))];
                if (k < part) run += x;
                total += x;
            }
            for (int c = c0; c < c1; c += 8) {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = cm[(size_t)min(c + u, g.nchunks - 1) * g.nbins + bb];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (live && c + u < c1) { colpref[(size_t)(c + u) * g.nbins + b] = run; run += v[u]; }
            }
            if (live && part == 0) __hip_atomic_store(&tot[b], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t = __hip_atomic_fetch_add(&ctl->ticket_off, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    CAFE_STAMP(3, 2);
    if (!last) return;
    // One exclusive scan over all bins (main bins with C and M of a partition adjacent, LRU bins, hot bins), kept in LDS;
    // each list's offsets are the scan less the value at the list's first bin.  A lane takes `per` consecutive bins.
    {
        const int per = (g.nbins + kThreads - 1) / kThreads;          // <= 32 (kMaxPlanBins / kThreads)
        uint32_t v[32], sum = 0;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int i = tid * per + u;
            v[u] = (u < per && i < g.nbins) ? __hip_atomic_load(&tot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            sum += v[u];
        }
        uint32_t inc = sum;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(inc, d, 64); if ((tid & 63) >= d) inc += x; }
        if ((tid & 63) == 63) sh[tid >> 6] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (int w = 0; w < (tid >> 6); ++w) run += sh[w];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int i = tid * per + u;
            if (u < per && i < g.nbins) excl[i] = run;
            run += v[u];
        }
        if (tid == kThreads - 1) excl[g.nbins] = run;
        __syncthreads();
    }
    uint32_t maxpart = 0, maxhot = 0, nclassc = 0;
    {
        const int l1 = bin_lru0(g), l2 = bin_hot0(g);
        const uint32_t base1 = excl[l1], base2 = excl[l2], total = excl[g.nbins];
        for (int i = tid; i < g.nbins; i += kThreads) {
            const uint32_t e = excl[i], c = excl[i + 1] - e;
            if (i < l1) {
                off[i] = e;
                if (!(i & 1)) { nclassc += c; maxpart = max(maxpart, excl[i + 2] - e); }
            } else if (i < l2) {
                off[off_lru0(g) + (i - l1)] = e - base1;
                maxpart = max(maxpart, c);
            } else {
                off[off_hot0(g) + (i - l2)] = e - base2;
                maxhot = max(maxhot, c);
            }
        }
        if (tid == 0) {
            off[l1] = base1;                                  // list totals close each list
            off[off_lru0(g) + g.P2] = base2 - base1;
            off[off_hot0(g) + g.PH] = total - base2;
        }
    }
    // The round-1 owners by estimated cost, heaviest first (a counting sort over cost classes; the order inside a class is
    // whatever the atomics give: it decides nothing but which workgroup takes which partition).  Used when the owners'
    // LDS lets only some of them be resident at once (k_round1 with the large capacities): a hot partition with a field's
    // most frequent id (thousands of dependent adds) and the LRU partitions (tens of dependent transitions per bucket)
    // then start first and the many light partitions fill in behind them.
    if (order && !(g.hot && g.hot_table)) {
        const int nhot = g.hot ? g.PH : 0, n1 = nhot + g.P2 + g.P1;
        const int l1 = bin_lru0(g), l2 = bin_hot0(g);
        auto cost_class = [&](int k) -> int {
            uint32_t c;
            if (k < nhot) c = 512u + 2u * (excl[l2 + k + 1] - excl[l2 + k]);
            else if (k < nhot + g.P2) c = 512u + 48u * (excl[l1 + (k - nhot) + 1] - excl[l1 + (k - nhot)]) / (uint32_t)max(1, g.bpp2);
            else { const int q = k - nhot - g.P2; c = 512u + 32u * (excl[2 * q + 1] - excl[2 * q]) / (uint32_t)max(1, g.bpp1); }
            return kCostClasses - 1 - (int)min(c >> 6, (uint32_t)(kCostClasses - 1));       // heavier = lower class
        };
        for (int i = tid; i < kCostClasses; i += kThreads) chist[i] = 0;
        __syncthreads();
        for (int k = tid; k < n1; k += kThreads) atomicAdd(&chist[cost_class(k)], 1u);
        __syncthreads();
        if (tid < 64) {
            // exclusive scan over the classes by one wave
            constexpr int per = kCostClasses / 64;
            uint32_t v[per], sum = 0;
#pragma unroll
            for (int u = 0; u < per; ++u) { v[u] = chist[tid * per + u]; sum += v[u]; }
            uint32_t inc = sum;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(inc, d, 64); if (tid >= d) inc += x; }
            uint32_t run = inc - sum;
#pragma unroll
            for (int u = 0; u < per; ++u) { chist[tid * per + u] = run; run += v[u]; }
        }
        __syncthreads();
        for (int k = tid; k < n1; k += kThreads) order[atomicAdd(&chist[cost_class(k)], 1u)] = (uint32_t)k;
    }
    CAFE_STAMP(3, 3);
    // control block and move counters start at zero; the LRU clock placeholder clock0 + 1 + position must stay positive
    uint32_t *cw = reinterpret_cast<uint32_t *>(ctl);
    for (int i = tid; i < (int)kCtlWords; i += kThreads) cw[i] = 0;
    for (int i = tid; i < 2 * g.P1; i += kThreads) evcnt[i] = 0;
    __syncthreads();
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        maxpart = max(maxpart, (uint32_t)__shfl_xor(maxpart, o2, 64));
        maxhot = max(maxhot, (uint32_t)__shfl_xor(maxhot, o2, 64));
        nclassc += __shfl_xor(nclassc, o2, 64);
    }
    if ((tid & 63) == 0) {
        atomicMax(&ctl->maxpart, maxpart);
        atomicMax(&ctl->maxhot, maxhot);
        atomicAdd(&ctl->nclassc, nclassc);
    }
    if (tid == 0) {
        if (g.flip_now) st->thr = (float)((double)st->thr / kDecayV);
        st->flip = (int32_t)g.gflip;
        st->fallback = 0;
        st->rounds = 0;
        if ((long long)st->clk + g.n + 2 >= 0x7fffffffLL) ctl->overflow = 1;
    }
    CAFE_STAMP(3, 4);
}

// a record leaves as one written-through 16-byte store (nothing for the kernel's end to write back)
__device__ inline void put_rec(Rec *p, const Rec &r) {
    store_through16(p, make_float4(__uint_as_float(r.pos), __uint_as_float(r.x), __uint_as_float(r.bl), __uint_as_float(r.pad)));
}

__global__ void __launch_bounds__(256) k_scatter(Geo g, IdSrc ids, const uint32_t *__restrict__ planA,
                                                 const uint32_t *__restrict__ planB, const uint32_t *__restrict__ colpref,
                                                 const uint32_t *__restrict__ off, Rec *__restrict__ rec1, Rec *__restrict__ rec2,
                                                 Rec *__restrict__ rech, uint8_t *__restrict__ miss) {
    CAFE_KWIN(2);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n) return;
    const uint32_t x = ids.at(i);
    const uint32_t a = planA[i];
    const int chunk = i / kChunk;
    const int binA = (int)((a >> 11) & 0x1fffu);
    const uint32_t rankA = (a & 0x7ffu) + colpref[(size_t)chunk * g.nbins + binA];
    const int h1 = fast_hash(x, g.fm1);
    Rec r;
    r.pos = (uint32_t)i; r.x = x; r.pad = 0;
    if (binA >= bin_hot0(g)) {
        const int ph = binA - bin_hot0(g);
        r.bl = ((uint32_t)(h1 - ph * g.bpph) << 2) | ((a >> 24) & 3u);
        put_rec(&rech[off[off_hot0(g) + ph] + rankA], r);
        miss[i] = 0;
        return;
    }
    const uint32_t mbit = (a >> 26) & 1u ? kMissBit : 0u;
    const int p1 = binA >> 1;
    r.bl = (uint32_t)(h1 - p1 * g.bpp1) | mbit;
    put_rec(&rec1[off[binA] + rankA], r);
    const uint32_t b = planB[i];
    const int binB = (int)((b >> 11) & 0x1fffu);
    const int p2 = binB - bin_lru0(g);
    r.bl = (uint32_t)(fast_hash(x, g.fm2) - p2 * g.bpp2) | mbit;
    put_rec(&rec2[off[off_lru0(g) + p2] + (b & 0x7ffu) + colpref[(size_t)chunk * g.nbins + binB]], r);
    miss[i] = mbit ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// owners: LDS layout and the stable sort
// ------------------------------------------------------------------------------------------------
// Sorted view of a partition (what the replay loops walk) and, aliased over it, the scratch of the sort that
// produces it.  CAP elements.
template <int CAP, bool MOVES>
struct Sorted {
    uint32_t sg[CAP];       // bucket index inside the partition (hot list: << 2 | slot)
    uint32_t stm[CAP];      // 2 * position (+ 1 for a move)
    uint32_t px[CAP];       // id (hot list: running score)
    float pc[CAP];          // score
    uint16_t pm[MOVES ? CAP : 2];       // moves: index into the move list | kind == 1 ? 0x8000 : 0
    uint8_t po[CAP];        // per-item flags / outcome
};
template <int CAP, bool MOVES>
struct SortTmp {
    uint32_t grp[CAP];      // sort key per element, in input order
    uint32_t tpos[MOVES ? CAP : 2];     // time per element, in input order (merge of the item lists with the move list)
    uint16_t pa[CAP], pb[CAP], lr[CAP];
    uint16_t hist[kWaves][kDigits];
    uint32_t wtot[kWaves];
};
template <int CAP, bool MOVES>
union OwnerLds {
    Sorted<CAP, MOVES> s;
    SortTmp<CAP, MOVES> t;
};

// Stable sort of n <= CAP elements by t.grp[element] (`bits` wide): LSD counting passes of kDigitBits bits.  Elements
// are taken in the order t.pa[0 .. n) when `init`, else 0 .. n-1.  Each wave walks a contiguous range of the sequence row
// by row; equal digits of a row are found by ballots, so ranks need no atomics and the order inside a digit is the
// sequence order.  On return t.lr[element] is the element's slot.  All lanes of the workgroup call it.
template <int CAP, bool MOVES>
__device__ inline void lsd_sort(SortTmp<CAP, MOVES> &t, int n, int bits, bool init) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long below = lanes_below();
    const int rows = (n + 63) >> 6, rpw = (rows + kWaves - 1) / kWaves;
    const int r0 = wave * rpw, r1 = min(r0 + rpw, rows);
    uint16_t *pin = t.pa, *pout = t.pb;
    bool have = init;
    // the key bits are split evenly over the passes: 12 bits are two passes of 64 digits, not one of 1024 and one of 4
    const int npass = max(1, (bits + kDigitBits - 1) / kDigitBits);
    const int db = max(1, (bits + npass - 1) / npass), nd = 1 << db;
    for (int pass = 0, shift = 0; pass < npass; ++pass, shift += db) {
        for (int i = tid; i < kWaves * nd / 2; i += kThreads) reinterpret_cast<uint32_t *>(&t.hist[0][0])[i] = 0;     // hist as [kWaves][nd]
        __syncthreads();
        uint16_t *hw = &t.hist[0][0] + wave * nd;
        for (int r = r0; r < r1; ++r) {
            const int j = r * 64 + lane;
            const bool valid = j < n;
            const int src = valid ? (have ? pin[j] : j) : 0;
            const uint32_t d = valid ? (t.grp[src] >> shift) & (uint32_t)(nd - 1) : 0u;
            const unsigned long long m = match_any(d, db, valid);
            uint32_t old = 0;
            if (valid && !(m & below)) { old = hw[d]; hw[d] = (uint16_t)(old + __popcll(m)); }
            old = __shfl(old, m ? __builtin_ctzll(m) : 0, 64);
            if (valid) t.lr[j] = (uint16_t)(old + __popcll(m & below));
        }
        __syncthreads();
        {
            // a lane takes dpl consecutive digits: totals over the waves, exclusive scan over all digits, per-wave bases
            const int dpl = max(1, nd / kThreads);
            const int d0 = tid * dpl;
            uint16_t *h0 = &t.hist[0][0];
            uint32_t sum = 0;
            if (d0 < nd)
                for (int k = 0; k < dpl; ++k)
                    for (int w = 0; w < kWaves; ++w) sum += h0[w * nd + d0 + k];
            uint32_t inc = sum;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(inc, o, 64);
                if (lane >= o) inc += v;
            }
            if (lane == 63) t.wtot[wave] = inc;
            __syncthreads();
            uint32_t run = inc - sum;
            for (int w = 0; w < wave; ++w) run += t.wtot[w];
            if (d0 < nd)
                for (int k = 0; k < dpl; ++k)
                    for (int w = 0; w < kWaves; ++w) { const uint32_t c = h0[w * nd + d0 + k]; h0[w * nd + d0 + k] = (uint16_t)run; run += c; }
        }
        __syncthreads();
        for (int r = r0; r < r1; ++r) {
            const int j = r * 64 + lane;
            if (j >= n) continue;
            const int src = have ? pin[j] : j;
            const uint32_t d = (t.grp[src] >> shift) & (uint32_t)(nd - 1);
            pout[hw[d] + t.lr[j]] = (uint16_t)src;
        }
        __syncthreads();
        uint16_t *sw = pin; pin = pout; pout = sw;
        have = true;
    }
    for (int s = tid; s < n; s += kThreads) t.lr[pin[s]] = (uint16_t)s;
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// plan, fourth kernel: the partitions in owner order
// ------------------------------------------------------------------------------------------------
// What a round-1 owner does before it needs a score -- bring its partition into bucket order (a stable sort: batch order
// inside a bucket) -- depends on the ids and the pre-batch tables only, so it belongs to the plan, which runs beside the
// forward or the previous step's last kernels, not to the chain that starts when the gradients arrive.  One workgroup per
// round-1 owner sorts that owner's records exactly as the owner would (same keys, same lsd_sort) and writes them back in
// that order; an LRU partition keeps its presumed misses only, and their number.  The owners of later rounds (k_fix) still
// sort for themselves: their miss sets and move lists change from round to round.
template <int CAP>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(CAP <= kCapSmall ? 6 : 1, 8)))
k_presort(Geo g, Ctl *ctl, const uint32_t *__restrict__ off, const Rec *__restrict__ rec1, const Rec *__restrict__ rec2,
          const Rec *__restrict__ rech, Rec *__restrict__ rec1s, Rec *__restrict__ rec2s, Rec *__restrict__ rechs,
          uint32_t *__restrict__ pre_nmiss) {
    CAFE_KWIN(7);
    __shared__ __attribute__((aligned(16))) SortTmp<CAP, false> T;
    __shared__ int s_miss;
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int nhot = (g.hot && !g.hot_table) ? g.PH : 0;
    const Rec *recs;
    Rec *out;
    int total, bits, kind;           // kind 0 hot, 1 LRU, 2 main (class C)
    if (b < nhot) {
        const uint32_t o0 = off[off_hot0(g) + b];
        total = (int)(off[off_hot0(g) + b + 1] - o0); recs = rech + o0; out = rechs + o0; bits = g.bitsh + 2; kind = 0;
    } else if ((b -= nhot) < g.P2) {
        const uint32_t o0 = off[off_lru0(g) + b];
        total = (int)(off[off_lru0(g) + b + 1] - o0); recs = rec2 + o0; out = rec2s + o0; bits = g.bits2 + 1; kind = 1;
    } else {
        b -= g.P2;
        const uint32_t o0 = off[2 * b];
        total = (int)(off[2 * b + 1] - o0); recs = rec1 + o0; out = rec1s + o0; bits = g.bits1; kind = 2;
    }
    if (kind == 1 && tid == 0) { s_miss = 0; if (total == 0 || total > CAP) pre_nmiss[b] = 0; }
    if (total == 0) return;
    if (total > CAP) { if (tid == 0) ctl->overflow = 1; return; }       // the owner would refuse it too
    constexpr int IPT = CAP / kThreads;
    Rec rr[IPT];
    const uint32_t dead = 1u << g.bits2;         // LRU: items that do not miss sort behind every bucket and are dropped
    int mine = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = tid + k * kThreads;
        if (j < total) {
            rr[k] = recs[j];
            uint32_t key = rr[k].bl;
            if (kind == 1) { const bool ms = (key & kMissBit) != 0; key = ms ? (key & ~kMissBit) : dead; mine += ms; }
            else if (kind == 2) key &= ~kMissBit;
            T.grp[j] = key;
        }
    }
    if (kind == 1) {
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
        if ((tid & 63) == 0 && mine) atomicAdd(&s_miss, mine);
    }
    __syncthreads();
    const int keep = kind == 1 ? s_miss : total;
    if (kind == 1 && tid == 0) pre_nmiss[b] = (uint32_t)keep;
    if (keep == 0) return;
    lsd_sort<CAP, false>(T, total, bits, false);
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = tid + k * kThreads;
        if (j < total) {
            const int s = T.lr[j];
            if (s < keep) out[s] = rr[k];
        }
    }
}

// per-field sums of the norms from the partial sums into LDS (every workgroup gets the same floats: field_sum_tree);
// a wave per field
__device__ inline void field_sums(const Geo &g, const Score &sc, float *fsum) {
    if (!sc.psum) return;
    const int F = (g.n + g.sseg - 1) / g.sseg;
    const int per = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int kMaxIter = kMaxSeg / kWaves;      // workgroups have at least kWaves waves
    float v[kMaxIter];
    // every partial sum this lane contributes to is requested before the first tree runs: one memory round trip
#pragma unroll
    for (int k = 0; k < kMaxIter; ++k) {
        const int f = k * per + wave;
        v[k] = (f < F && lane < g.nparts) ? sc.psum[(size_t)f * g.nparts + lane] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < kMaxIter; ++k) {
        const int f = k * per + wave;
        if (k * per >= F) break;
        float x = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);        // cafe_core.h::field_sum_tree
        if (f < F && lane == 0) fsum[f] = x;
    }
}
__device__ inline ScoreSrc score_src(const Geo &g, const Score &sc, const float *fsum) {
    ScoreSrc s;
    s.vals = sc.vals; s.fsum = sc.psum ? fsum : nullptr; s.seg = g.sseg; s.stride = sc.vstride;
    return s;
}

// ------------------------------------------------------------------------------------------------
// hot bypass: ordered score sums of the ids that are hot before the batch
// ------------------------------------------------------------------------------------------------
// Outputs, all indexed like the partition's records in sorted order: hkey = [bucket in partition << 2 | slot : 25]
// [2 * position : 23][0 : 16], hrun (score of the slot after the item: what a move into the bucket later in the batch
// has to see, sketch.cpp:244-247), and at the first item of every slot hfin = {bucket * 4 + slot, final score}
// (-1 elsewhere), applied to the table by k_commit.
template <int CAP, bool MV>
__device__ inline void hot_owner(OwnerLds<CAP, MV> &L, int *heavy, int *n_heavy, float *fsum, const Geo &g, int p,
                                 const MainBucket *__restrict__ mt, const DevState *__restrict__ st, Ctl *ctl,
                                 const uint32_t *__restrict__ off, const Rec *__restrict__ rech, const Rec *__restrict__ rechs,
                                 const Score &sc,
                                 unsigned long long *__restrict__ hkey, float *__restrict__ hrun, int2 *__restrict__ hfin,
                                 uint8_t *__restrict__ outm) {
    const int tid = threadIdx.x;
    const uint32_t o0 = off[off_hot0(g) + p], o1 = off[off_hot0(g) + p + 1];
    const int total = (int)(o1 - o0);
    if (total == 0) return;
    if (total > CAP) { if (tid == 0) ctl->overflow = 1; return; }
    if (tid == 0) *n_heavy = 0;
    CAFE_STAMP_VAL(g.prof, 7, 1ull | ((unsigned long long)total << 8));
    field_sums(g, sc, fsum);
    const int b0 = p * g.bpph;
    const Rec *recs = rech + o0;
    if (total > 512) __builtin_amdgcn_s_setprio(3);
    constexpr int IPT = CAP / kThreads;
    const ScoreSrc ss = score_src(g, sc, fsum);
    if (rechs) {
        // the plan left the partition in owner order (k_presort): one coalesced read, no sort on this side of the scores
        // (the first records are requested before the field sums are waited for)
        // (the large owners are few per compute unit and have registers to spare: the whole partition in one round of
        // record loads and one round of score loads; the small ones keep to eight records in flight)
        constexpr int PRE = (CAP > 2048 || IPT < 8) ? IPT : 8;
        Rec rr[PRE];
#pragma unroll
        for (int k = 0; k < PRE; ++k) { const int s = tid + k * kThreads; if (s < total) rr[k] = rechs[o0 + s]; }
        __syncthreads();                // field sums in place
        float cc[PRE];
#pragma unroll
        for (int k = 0; k < PRE; ++k) { const int s = tid + k * kThreads; cc[k] = s < total ? ss.at(rr[k].pos) : 0.0f; }
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const int s = tid + k * kThreads;
            if (s < total) { L.s.sg[s] = rr[k].bl; L.s.stm[s] = 2u * rr[k].pos; L.s.pc[s] = cc[k]; }
        }
        for (int s = tid + PRE * kThreads; s < total; s += kThreads) {
            const Rec r = rechs[o0 + s];
            L.s.sg[s] = r.bl; L.s.stm[s] = 2u * r.pos; L.s.pc[s] = ss.at(r.pos);
        }
        __syncthreads();
    } else {
    for (int j = tid; j < total; j += kThreads) L.t.grp[j] = recs[j].bl;
    __syncthreads();
    CAFE_STAMP(g.prof, 1);
    lsd_sort<CAP, MV>(L.t, total, g.bitsh + 2, false);
    CAFE_STAMP(g.prof, 2);
    uint16_t myr[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) { const int j = tid + k * kThreads; myr[k] = j < total ? L.t.lr[j] : 0; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = tid + k * kThreads;
        if (j < total) {
            const Rec r = recs[j];
            const int s = myr[k];
            L.s.sg[s] = r.bl; L.s.stm[s] = 2u * r.pos; L.s.pc[s] = ss.at(r.pos);
        }
    }
    __syncthreads();
    }
    CAFE_STAMP(g.prof, 3);
    const float thr = st->thr;
    int2 *fin = hfin + o0;
    // the table's score of every slot that has items, requested together (a wave with several long runs would
    // otherwise pay one memory round trip per run)
    for (int s = tid; s < total; s += kThreads) {
        const uint32_t bs = L.s.sg[s];
        if (s == 0 || L.s.sg[s - 1] != bs) L.s.px[s] = __float_as_uint(mt[b0 + (int)(bs >> 2)].cnt[bs & 3u]);
    }
    __syncthreads();
    // Ordered sums: the score of a slot after every one of its items, in batch order, left in px[] (the table's score at
    // the head is read before it is overwritten).  Runs below 64 items: one lane each, four items per LDS round trip
    // (po[e] says whether item e + 1 belongs to the same run).  Longer runs: a wave each -- every lane walks the same
    // chain of adds, eight scores requested from LDS ahead of their use, so the chain runs at the latency of a float
    // add per item.  Crossings (sketch.cpp:218-220) are evaluated afterwards for all items at once.
    constexpr int kLongRun = 64;
    // po[s]: bit 0 = item s + 1 belongs to the same run; bit 1 (heads only) = the run is on the heavy list
    for (int s = tid; s < total; s += kThreads) {
        const uint32_t bs = L.s.sg[s];
        const bool head = s == 0 || L.s.sg[s - 1] != bs;
        uint8_t f = (s + 1 < total && L.s.sg[s + 1] == bs) ? 1 : 0;
        if (!head) {
            fin[s] = make_int2(-1, 0);
        } else if (s + kLongRun <= total && L.s.sg[s + kLongRun - 1] == bs) {
            const int at = atomicAdd(n_heavy, 1);
            if (at < kMaxHeavy) { heavy[at] = s; f |= 2; }
        }
        L.s.po[s] = f;
    }
    for (int s = total + tid; s < min(total + 4, CAP); s += kThreads) L.s.po[s] = 0;
    __syncthreads();
    CAFE_STAMP(g.prof, 4);
    // Long runs go to whole waves, the rows of 64 sorted positions (whose heads are walked by single lanes) to the waves
    // that have no long run: a partition with one heavy id keeps one wave on that chain while the other three finish
    // everything else beside it.  (Four or more long runs: runs and rows both round-robin over the four waves.)
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int nh = min(*n_heavy, kMaxHeavy), rows = (total + 63) >> 6;
        const bool split = nh < kWaves;
        const int h_step = split ? kMaxHeavy : kWaves;                      // split: wave w < nh takes run w only
        const int row0 = split ? wave - nh : wave, row_step = split ? kWaves - nh : kWaves;
        for (int w = wave; w < nh; w += h_step) {
            const int s = heavy[w];
            const uint32_t bs = L.s.sg[s];
            int r = 0;                                        // length of the run
            for (int at = s;; at += 64) {
                const int idx = at + lane;
                const unsigned long long m = __ballot(idx < total && L.s.sg[idx] == bs);
                const int rr = (~m) ? __builtin_ctzll(~m) : 64;
                r += rr;
                if (rr < 64) break;
            }
            // Every lane walks the same chain of adds (the reads are broadcasts); lane 0 stores the running scores.  Once
            // the position is a multiple of four the scores move in 16-byte LDS accesses, in blocks of 16 with three
            // register sets in rotation: a block is requested two blocks (32 dependent adds) before its first add, which
            // covers the LDS latency of a compute unit whose other owners are sorting.  (The run length is used by all
            // lanes on purpose: with the chain under `if (lane == 0)` the compiler moves the ballots above into that
            // branch, and the wave never leaves it.)
            {
                const bool wr = lane == 0;
                float acc = __uint_as_float(L.s.px[s]);
                const float *cs = L.s.pc + s;
                float *ws = reinterpret_cast<float *>(L.s.px) + s;
                int j = 0;
                const int peel = min(r, (4 - (s & 3)) & 3);
                for (; j < peel; ++j) { acc += cs[j]; if (wr) ws[j] = acc; }
                const int nblk = (r - j) >> 4;
                const float4 *c4 = reinterpret_cast<const float4 *>(cs + j);
                float4 *w4 = reinterpret_cast<float4 *>(ws + j);
                auto load = [&](float4 (&q)[4], int blk) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) q[k] = c4[4 * blk + k];
                };
                auto proc = [&](const float4 (&q)[4], int blk) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float w0 = acc + q[k].x, w1 = w0 + q[k].y, w2 = w1 + q[k].z, w3 = w2 + q[k].w;
                        acc = w3;
                        if (wr) w4[4 * blk + k] = make_float4(w0, w1, w2, w3);
                    }
                };
                float4 qa[4], qb[4], qc[4];
                if (nblk > 0) load(qa, 0);
                if (nblk > 1) load(qb, 1);
                for (int blk = 0; blk < nblk; blk += 3) {
                    if (blk + 2 < nblk) load(qc, blk + 2);
                    proc(qa, blk);
                    if (blk + 3 < nblk) load(qa, blk + 3);
                    if (blk + 1 < nblk) proc(qb, blk + 1);
                    if (blk + 4 < nblk) load(qb, blk + 4);
                    if (blk + 2 < nblk) proc(qc, blk + 2);
                }
                j += nblk << 4;
                for (; j < r; ++j) { acc += cs[j]; if (wr) ws[j] = acc; }
                if (wr) fin[s] = make_int2((b0 + (int)(bs >> 2)) * 4 + (int)(bs & 3u), __float_as_int(acc));
            }
        }
        for (int row = row0; row >= 0 && row < rows; row += row_step) {
            const int s = row * 64 + lane;
            if (s >= total) continue;
            const uint32_t bs = L.s.sg[s];
            const bool head = s == 0 || L.s.sg[s - 1] != bs;
            if (!head || (L.s.po[s] & 2)) continue;
            const int bucket = b0 + (int)(bs >> 2), slot = (int)(bs & 3u);
            float acc = __uint_as_float(L.s.px[s]);
            int e = s;
            for (;;) {
                // items e .. e + 3 (reads past the end of the run, or of the partition, are not used)
                const float c0 = L.s.pc[e], c1 = L.s.pc[min(e + 1, CAP - 1)], c2 = L.s.pc[min(e + 2, CAP - 1)], c3 = L.s.pc[min(e + 3, CAP - 1)];
                const uint8_t k0 = L.s.po[e], k1 = L.s.po[min(e + 1, CAP - 1)], k2 = L.s.po[min(e + 2, CAP - 1)], k3 = L.s.po[min(e + 3, CAP - 1)];
                acc += c0; L.s.px[e] = __float_as_uint(acc);
                if (!(k0 & 1)) break;
                acc += c1; L.s.px[e + 1] = __float_as_uint(acc);
                if (!(k1 & 1)) break;
                acc += c2; L.s.px[e + 2] = __float_as_uint(acc);
                if (!(k2 & 1)) break;
                acc += c3; L.s.px[e + 3] = __float_as_uint(acc);
                if (!(k3 & 1)) break;
                e += 4;
            }
            fin[s] = make_int2(bucket * 4 + slot, __float_as_int(acc));
        }
    }
    __syncthreads();
    CAFE_STAMP(g.prof, 5);
    for (int s = tid; s < total; s += kThreads) {
        const uint32_t tm = L.s.stm[s];
        const float run = __uint_as_float(L.s.px[s]), c = L.s.pc[s];
        hkey[o0 + s] = ((unsigned long long)L.s.sg[s] << 39) | ((unsigned long long)tm << 16);
        hrun[o0 + s] = run;
        outm[tm >> 1] = (run >= thr && run - c < thr) ? oCross : 0;         // sketch.cpp:218-220
    }
}

// ---- the same without a sort: a wave per hot partition ------------------------------------------------------------
// The slots of a small hot partition (4 per bucket, <= kHotKeys) get one accumulator each in an LDS table of the wave.
// The wave walks the partition's items row by row (64 at a time, batch order); the items of a row that address the same
// slot are found by ballots and added, in lane order, by the first of them.  No sort, no workgroup barrier, no limit
// on the items of a partition.  Outputs as hot_owner's, in batch order (hfin at every item: commit writes the same
// final score several times).
__device__ inline void hot_table_owner(float *acc /* [kHotKeys] */, float *runs /* [64] */, const float *fsum, const Geo &g, int ph,
                                       const MainBucket *__restrict__ mt, const DevState *__restrict__ st,
                                       const uint32_t *__restrict__ off, const Rec *__restrict__ rech, const Score &sc,
                                       unsigned long long *__restrict__ hkey, float *__restrict__ hrun, int2 *__restrict__ hfin,
                                       uint8_t *__restrict__ outm) {
    (void)runs;
    const int lane = threadIdx.x & 63;
    const uint32_t o0 = off[off_hot0(g) + ph], o1 = off[off_hot0(g) + ph + 1];
    const int total = (int)(o1 - o0);
    if (total == 0) return;
    const int b0 = ph * g.bpph;
    const Rec *recs = rech + o0;
    const ScoreSrc ss = score_src(g, sc, fsum);
    const unsigned long long below = lanes_below();
    const float thr = st->thr;
    // a partition with a heavy id is a long chain of dependent adds in ONE wave, and round 1 ends when the longest chain
    // does: that wave issues ahead of the others on its SIMD
    if (total > 96) __builtin_amdgcn_s_setprio(3);
    constexpr int R = 4;             // rows of 64 items whose records and scores are requested together
    // the table's score of every slot that has items (items of one slot write the same value); the first R rows are
    // covered by the main loop's own loads
    for (int j = R * 64 + lane; j < total; j += 64) {
        const uint32_t key = recs[j].bl;
        acc[key] = mt[b0 + (int)(key >> 2)].cnt[key & 3u];
    }
    for (int cbase = 0; cbase < total; cbase += R * 64) {
        Rec rr[R];
        float cc[R], base[R];
#pragma unroll
        for (int k = 0; k < R; ++k) rr[k] = recs[min(cbase + k * 64 + lane, total - 1)];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            cc[k] = ss.at(rr[k].pos);
            base[k] = cbase == 0 ? mt[b0 + (int)(rr[k].bl >> 2)].cnt[rr[k].bl & 3u] : 0.0f;
        }
        if (cbase == 0) {
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (k * 64 + lane < total) acc[rr[k].bl] = base[k];
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int j = cbase + k * 64 + lane;
            if (cbase + k * 64 >= total) break;
            const bool valid = j < total;
            const uint32_t key = rr[k].bl;
            const float c = cc[k];
            const unsigned long long m = match_any(key, g.bitsh + 2, valid);
            const bool leader = valid && !(m & below);
            __builtin_amdgcn_wave_barrier();
            // Slot by slot (the loop is uniform, so a peer's score travels by v_readlane, not through LDS): the slot's
            // items of this row are added in lane order, every item keeps the score behind it.
            float a_all = leader ? acc[key] : 0.0f;
            float run = 0.0f;
            const int mlo = (int)(uint32_t)m, mhi = (int)(uint32_t)(m >> 32);
            unsigned long long leaders = __ballot(leader);
            while (leaders) {
                const int ll = __builtin_ctzll(leaders);
                leaders &= leaders - 1ull;
                unsigned long long rest = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(mhi, ll) << 32) |
                                          (uint32_t)__builtin_amdgcn_readlane(mlo, ll);
                float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_all), ll));
                while (rest) {
                    const int pl = __builtin_ctzll(rest);
                    rest &= rest - 1ull;
                    a += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), pl));
                    run = lane == pl ? a : run;
                }
                a_all = lane == ll ? a : a_all;
            }
            if (leader) acc[key] = a_all;
            if (valid) {
                hkey[o0 + j] = ((unsigned long long)key << 39) | ((unsigned long long)(2u * rr[k].pos) << 16);
                hrun[o0 + j] = run;
                outm[rr[k].pos] = (run >= thr && run - c < thr) ? oCross : 0;         // sketch.cpp:218-220
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (total <= R * 64) {
            // the usual case: the whole partition was in registers
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int j = k * 64 + lane;
                if (j < total) hfin[o0 + j] = make_int2((b0 + (int)(rr[k].bl >> 2)) * 4 + (int)(rr[k].bl & 3u), __float_as_int(acc[rr[k].bl]));
            }
            return;
        }
    }
    for (int j = lane; j < total; j += 64) {
        const uint32_t key = recs[j].bl;
        hfin[o0 + j] = make_int2((b0 + (int)(key >> 2)) * 4 + (int)(key & 3u), __float_as_int(acc[key]));
    }
}

// Score of hot slot `key` (bucket in hot partition << 2 | slot) as a move at `time` sees it: the running score behind
// the slot's last bypassed item before `time`, or `base` (the table's value) if there is none.  The partition's items
// are in batch order, or sorted by (slot, time): either way the last match is the latest.  (A handful of moves per batch
// look here.)
__device__ inline float hot_cnt_at(const unsigned long long *__restrict__ hk, const float *__restrict__ hr, int nh,
                                   uint32_t key, uint32_t time, float base, bool sorted) {
    if (sorted) {
        // keys ascend by (slot, time): the last entry below (slot, time) is the latest one in front of the move, if it
        // is an entry of this slot at all.  (Thousands of moves per batch look here when the id stream churns.)
        const unsigned long long bound = ((unsigned long long)key << 39) | ((unsigned long long)time << 16);
        int lo = 0, hi = nh;                    // first index with hk >= bound
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (hk[mid] < bound) lo = mid + 1; else hi = mid;
        }
        if (lo == 0) return base;
        return (uint32_t)(hk[lo - 1] >> 39) == key ? hr[lo - 1] : base;
    }
    float best = base;
    for (int j = 0; j < nh; ++j) {
        const unsigned long long k = hk[j];
        if ((uint32_t)(k >> 39) == key && ((uint32_t)(k >> 16) & 0x7fffffu) < time) best = hr[j];
    }
    return best;
}
// bucket h1 (global index) of the main table
__device__ inline void patch_hot(MainR &b, const Geo &g, const uint32_t *__restrict__ off, const unsigned long long *__restrict__ hkey,
                                 const float *__restrict__ hrun, int h1, uint32_t time) {
    if (!g.hot || !(b.d0 | b.d1 | b.d2 | b.d3)) return;
    const int ph = h1 / g.bpph;
    const uint32_t oh0 = off[off_hot0(g) + ph];
    const int nh = (int)(off[off_hot0(g) + ph + 1] - oh0);
    if (nh <= 0) return;
    const uint32_t kb = (uint32_t)(h1 - ph * g.bpph) << 2;
    const bool sorted = !g.hot_table;           // the sorting owner leaves (slot, time) order, the table owner batch order
    if (b.d0) b.c0 = hot_cnt_at(hkey + oh0, hrun + oh0, nh, kb | 0u, time, b.c0, sorted);
    if (b.d1) b.c1 = hot_cnt_at(hkey + oh0, hrun + oh0, nh, kb | 1u, time, b.c1, sorted);
    if (b.d2) b.c2 = hot_cnt_at(hkey + oh0, hrun + oh0, nh, kb | 2u, time, b.c2, sorted);
    if (b.d3) b.c3 = hot_cnt_at(hkey + oh0, hrun + oh0, nh, kb | 3u, time, b.c3, sorted);
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

struct MainArgs {
    const Rec *rec1s;           // round 1: the class C items of every partition in owner order (k_presort)
    const MainBucket *mt;
    const DevState *st;
    Ctl *ctl;
    const uint32_t *off;
    const Rec *rec1;
    const uint32_t *evcnt;      // [2][P1]
    const MoveEv *ev;
    EvRes *evres;
    uint8_t *miss, *outm;
    MainBucket *stage_m;
    uint32_t *stage_tot, *prevcnt;
    uint8_t *ldirty;            // [2][P2]
    const unsigned long long *hkey;
    const float *hrun;
    unsigned long long call_tag;
};

// Staged buckets have a static home: the entry of a bucket is the sorted position of its first item inside its
// partition's region, so a partition that is not re-run in a later round keeps its staged result.
// Round 1 (MOVES = false) replays the class C items only; a later round replays every item of the partition together
// with the moves the last LRU round addressed to it, all in time order.
template <int CAP, bool MOVES, bool MV>
__device__ inline void main_owner(OwnerLds<CAP, MV> &L, int *heavy, int *n_heavy, float *fsum, const Geo &g, int p, int round,
                                  const MainArgs &A, const Score &sc) {
    const int tid = threadIdx.x;
    const uint32_t o0 = A.off[2 * p], oM = A.off[2 * p + 1], o1 = A.off[2 * p + 2];
    const int prev = (round & 1) ^ 1, par = round & 1;
    const int np = MOVES ? (int)(o1 - o0) : (int)(oM - o0);
    const int nc = (int)(oM - o0);          // class C items come first, each list in batch order
    int ne = 0;
    if (MOVES) {
        const uint32_t ne_raw = A.evcnt[prev * g.P1 + p];
        if (ne_raw > (uint32_t)g.cap_ev) { if (tid == 0) A.ctl->overflow = 1; return; }
        ne = (int)ne_raw;
    }
    const int total = np + ne;
    if (total > CAP) { if (tid == 0) A.ctl->overflow = 1; return; }
    const size_t evbase = (size_t)prev * g.P1 * g.evs + (size_t)p * g.evs;          // this round's moves
    const size_t evold = (size_t)(prev ^ 1) * g.P1 * g.evs + (size_t)p * g.evs;     // the moves of the round before
    if (MOVES) {
        // Same moves as the last round this partition ran with => same result: its staged buckets, per-item
        // outcomes and move outcomes stand.  Lists are compared as sets (arrival order is not deterministic).
        const int pne = (int)A.prevcnt[p];
        if (ne == 0 && pne == 0) return;             // ran without moves last time, no moves now
        int differ = ne != pne;
        if (!differ) {
            for (int t = tid; t < ne; t += kThreads) {
                const MoveEv a = A.ev[evbase + t];
                int found = 0;
                for (int j = 0; j < ne; ++j) {
                    const MoveEv b = A.ev[evold + j];
                    found |= (a.time == b.time && a.x == b.x && __float_as_uint(a.cnt) == __float_as_uint(b.cnt) &&
                              a.dic == b.dic && a.kind == b.kind);
                }
                differ |= !found;
            }
        }
        differ = __syncthreads_or(differ);
        if (tid == 0) A.prevcnt[p] = (uint32_t)ne;
        if (!differ) return;
    } else if (tid == 0) {
        A.prevcnt[p] = 0;
    }
    if (tid == 0) { A.stage_tot[p] = (uint32_t)total; *n_heavy = 0; }
    if (total == 0) return;
    CAFE_STAMP_VAL(g.prof, 7, 3ull | ((unsigned long long)total << 8));
    field_sums(g, sc, fsum);
    const int b0 = p * g.bpp1;
    const Rec *recs = A.rec1 + o0;
    if (total > 512) __builtin_amdgcn_s_setprio(3);

    constexpr int IPT = CAP / kThreads;
    const ScoreSrc ss = score_src(g, sc, fsum);
    bool presorted = false;
    if constexpr (!MOVES) presorted = A.rec1s != nullptr;
    if (presorted) {
        // round 1: the class C items come in owner order from the plan (k_presort)
        constexpr int PRE = IPT < 8 ? IPT : 8;
        Rec rr[PRE];
#pragma unroll
        for (int k = 0; k < PRE; ++k) { const int s = tid + k * kThreads; if (s < total) rr[k] = A.rec1s[o0 + s]; }
        __syncthreads();                // field sums in place
        auto put = [&](int s, const Rec &r) {
            L.s.sg[s] = r.bl & ~kMissBit; L.s.stm[s] = 2u * r.pos; L.s.px[s] = r.x; L.s.pc[s] = ss.at(r.pos);
            L.s.po[s] = (uint8_t)((r.bl & kMissBit) ? 1 : 0);
        };
#pragma unroll
        for (int k = 0; k < PRE; ++k) { const int s = tid + k * kThreads; if (s < total) put(s, rr[k]); }
        for (int s = tid + PRE * kThreads; s < total; s += kThreads) put(s, A.rec1s[o0 + s]);
        __syncthreads();
    } else {
    // elements: [0, nc) class C items, [nc, np) class M items, [np, total) moves
    for (int j = tid; j < total; j += kThreads) {
        if (j < np) {
            const Rec r = recs[j];
            L.t.grp[j] = r.bl & ~kMissBit;
            if (MOVES) L.t.tpos[j] = 2u * r.pos;
        } else if (MOVES) {
            const MoveEv m = A.ev[evbase + (j - np)];
            L.t.grp[j] = (uint32_t)(fast_hash(m.x, g.fm1) - b0);
            L.t.tpos[j] = m.time;
        }
    }
    __syncthreads();
    if (MOVES) {
        // merge the three lists by time (two are sorted; the moves, few, are ranked among themselves)
        auto lower = [&](int lo, int hi, uint32_t tm) {      // elements of [lo, hi) (sorted) with time < tm
            const int base = lo;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (L.t.tpos[mid] < tm) lo = mid + 1; else hi = mid; }
            return lo - base;
        };
        for (int j = tid; j < total; j += kThreads) {
            const uint32_t tm = L.t.tpos[j];
            int at;
            if (j < nc) {
                at = j + lower(nc, np, tm);
            } else if (j < np) {
                at = (j - nc) + lower(0, nc, tm);
            } else {
                at = lower(0, nc, tm) + lower(nc, np, tm);
            }
            int mv = 0;
            for (int e = np; e < total; ++e) mv += (L.t.tpos[e] < tm) || (L.t.tpos[e] == tm && e < j && j >= np);
            L.t.pa[at + mv] = (uint16_t)j;
        }
        __syncthreads();
    }
    CAFE_STAMP(g.prof, 1);
    lsd_sort<CAP, MV>(L.t, total, g.bits1, MOVES);
    CAFE_STAMP(g.prof, 2);
    uint16_t myr[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) { const int j = tid + k * kThreads; myr[k] = j < total ? L.t.lr[j] : 0; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = tid + k * kThreads;
        if (j < total) {
            const int s = myr[k];
            if (j < np) {
                const Rec r = recs[j];
                L.s.sg[s] = r.bl & ~kMissBit; L.s.stm[s] = 2u * r.pos; L.s.px[s] = r.x; L.s.pc[s] = ss.at(r.pos);
                // what the last round (or the plan) said about the item: 1 = missed
                L.s.po[s] = MOVES ? A.miss[r.pos] : (uint8_t)((r.bl & kMissBit) ? 1 : 0);
                if (MOVES) L.s.pm[s] = 0;
            } else if (MOVES) {
                const int e = j - np;
                const MoveEv m = A.ev[evbase + e];
                L.s.sg[s] = (uint32_t)(fast_hash(m.x, g.fm1) - b0); L.s.stm[s] = m.time; L.s.px[s] = m.x; L.s.pc[s] = m.cnt;
                L.s.po[s] = 0;
                L.s.pm[s] = (uint16_t)(e | (m.kind == 1 ? 0x8000 : 0));
            }
        }
    }
    __syncthreads();
    }

    CAFE_STAMP(g.prof, 3);
    const float thr = A.st->thr;
    const uint32_t gflip = (uint32_t)A.st->flip;
    MainBucket *stage = A.stage_m + o0 + (size_t)p * g.evs;
    MainCtx ctx;
    ctx.full = A.st->fifo_len > 0;
    int changed = 0;
    // Phase A: one lane per bucket replays the bucket's items in batch order.  Buckets with many items (a
    // heavy id) are left to phase B, where a whole wave works on one bucket.
    // The table bucket of the next head is requested before the current one is replayed, so that a lane with several
    // heads does not pay a memory round trip (and the completion of its own staging stores) per head.
    auto classify = [&](int s) -> int {        // 0: not a head, 1: replay here, 2: handed to phase B
        if (s >= total) return 0;
        const uint32_t bl = L.s.sg[s];
        if (s != 0 && L.s.sg[s - 1] == bl) return 0;
        if (s + kHeavySeg <= total && L.s.sg[s + kHeavySeg - 1] == bl) {
            const int at = atomicAdd(n_heavy, 1);
            if (at < kMaxHeavy) { heavy[at] = s; return 2; }
        }
        return 1;
    };
    MainR nb;
    int nmode = classify(tid);
    if (nmode == 1) load_main(nb, &A.mt[b0 + (int)L.s.sg[tid]]);
    for (int s = tid; s < total; s += kThreads) {
        const int mode = nmode;
        MainR b = nb;
        nmode = classify(s + kThreads);
        if (nmode == 1) load_main(nb, &A.mt[b0 + (int)L.s.sg[s + kThreads]]);
        if (mode != 1) continue;
        const uint32_t bl = L.s.sg[s];
        const int32_t keep = hot_keep_mask(b, gflip);
        sync_flip(b, gflip);
        for (int e = s; e < total && L.s.sg[e] == bl; ++e) {
            const uint32_t time = L.s.stm[e];
            if (!(time & 1u)) {
                ctx.pos = time >> 1; ctx.flags = 0;
                int r = main_probe(b, L.s.px[e], L.s.pc[e], thr, ctx);
                const bool m = r < 0;
                if (m != (L.s.po[e] != 0)) { changed = 1; if (MOVES) A.ldirty[par * g.P2 + fast_hash(L.s.px[e], g.fm2) / g.bpp2] = 1; }
                L.s.po[e] = (uint8_t)(ctx.flags | (m ? oMiss : 0) | (r > 0 ? oGot : 0));
            } else if (MOVES) {
                ctx.pushed = 0;
                const uint16_t pm = L.s.pm[e];
                const int32_t dic = A.ev[evbase + (int)(pm & 0x7fffu)].dic;
                patch_hot(b, g, A.off, A.hkey, A.hrun, b0 + (int)bl, time);
                int ok = main_move(b, L.s.px[e], L.s.pc[e], dic, ctx);
                EvRes r;
                r.tag = A.call_tag; r.kind = (pm & 0x8000u) ? 1 : 2; r.ok = ok; r.pushed = ctx.pushed; r.pad = 0;
                A.evres[time >> 1] = r;
            }
        }
        store_main(&stage[s], b, b0 + (int)bl, keep);
    }
    __syncthreads();
    CAFE_STAMP(g.prof, 4);
    // Phase B: heavy buckets, one per wave at a time.  All 64 lanes carry the same bucket copy.  A run of
    // consecutive probes of one id that sits in the bucket only needs the ordered float sum of the scores on
    // the serial path (one readlane + add per item); threshold crossing and the pop are then evaluated for 64
    // items at once.  Anything else (misses, moves) takes the one-item step, executed uniformly.
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int nh = min(*n_heavy, kMaxHeavy);
        MainR nhb;
        if (wave < nh) load_main(nhb, &A.mt[b0 + (int)L.s.sg[heavy[wave]]]);
        for (int h = wave; h < nh; h += kWaves) {
            const int s = heavy[h];
            const uint32_t bl = L.s.sg[s];
            MainR b = nhb;
            if (h + kWaves < nh) load_main(nhb, &A.mt[b0 + (int)L.s.sg[heavy[h + kWaves]]]);
            const int32_t keep = hot_keep_mask(b, gflip);
            sync_flip(b, gflip);
            int e = s;
            while (e < total && L.s.sg[e] == bl) {
                const uint32_t time0 = L.s.stm[e];
                const uint32_t x0 = L.s.px[e];
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
                            const bool same = in && L.s.sg[idx] == bl && !(L.s.stm[idx] & 1u) && L.s.px[idx] == x0;
                            const unsigned long long m = __ballot(same);
                            const int rr = (~m) ? __builtin_ctzll(~m) : 64;
                            if (lane < rr) {
                                const float c = L.s.pc[idx];
                                part += c;
                                flag |= (c >= 0.0f ? 0 : 1) | (L.s.po[idx] != 0 ? 2 : 0);
                            }
                            r += rr;
                            if (rr < 64) break;
                        }
                        for (int o = 32; o > 0; o >>= 1) { part += __shfl_xor(part, o, 64); flag |= __shfl_xor(flag, o, 64); }
                        float acc = sel4f(b.c0, b.c1, b.c2, b.c3, hit);
                        if (!(flag & 1) && acc - (acc + part) * 2.4e-7f >= thr) {
                            const float *cs = L.s.pc + e;
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
                                if (lane == 0) { changed = 1; if (MOVES) A.ldirty[par * g.P2 + fast_hash(x0, g.fm2) / g.bpp2] = 1; }
                            }
                            for (int i = lane; i < r; i += 64) L.s.po[e + i] = 0;
                            CAFE_SET4(b.c0, b.c1, b.c2, b.c3, hit, acc);
                            e += r;
                            continue;
                        }
                    }
                    const int idx = e + lane;
                    const bool in = idx < total;
                    const bool same = in && L.s.sg[idx] == bl && !(L.s.stm[idx] & 1u) && L.s.px[idx] == x0;
                    const unsigned long long m = __ballot(same);
                    const int r = (~m) ? __builtin_ctzll(~m) : 64;            // length of the run starting at e (>= 1)
                    const float c = in ? L.s.pc[idx] : 0.0f;
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
                    if (kpop >= 0) dic = (int32_t)(kSur | (L.s.stm[e + kpop] >> 1));
                    if (act) {
                        if (L.s.po[idx] != 0) { changed = 1; if (MOVES) A.ldirty[par * g.P2 + fast_hash(x0, g.fm2) / g.bpp2] = 1; }   // it missed last round
                        L.s.po[idx] = (uint8_t)((cross ? oCross : 0) | (lane == kpop ? (oPop | oGot) : 0));
                    }
                    CAFE_SET4(b.c0, b.c1, b.c2, b.c3, hit, acc);
                    CAFE_SET4(b.d0, b.d1, b.d2, b.d3, hit, dic);
                    e += r;
                } else if (!(time0 & 1u)) {
                    // a run of probes of an id that is not in the bucket (a frequent id that still lives in the LRU
                    // table): all misses, no change to the bucket, 64 items at a time
                    const int idx = e + lane;
                    const bool same = idx < total && L.s.sg[idx] == bl && !(L.s.stm[idx] & 1u) && L.s.px[idx] == x0;
                    const unsigned long long m = __ballot(same);
                    const int r = (~m) ? __builtin_ctzll(~m) : 64;            // length of the run starting at e (>= 1)
                    if (lane < r) {
                        if (L.s.po[idx] == 0) { changed = 1; if (MOVES) A.ldirty[par * g.P2 + fast_hash(x0, g.fm2) / g.bpp2] = 1; }
                        L.s.po[idx] = oMiss;
                    }
                    e += r;
                } else {
                    if (MOVES) {
                        ctx.pushed = 0;
                        const uint16_t pmv = L.s.pm[e];
                        const int32_t dic = A.ev[evbase + (int)(pmv & 0x7fffu)].dic;
                        patch_hot(b, g, A.off, A.hkey, A.hrun, b0 + (int)bl, time0);
                        int ok = main_move(b, x0, L.s.pc[e], dic, ctx);
                        if (lane == 0) {
                            EvRes r;
                            r.tag = A.call_tag; r.kind = (pmv & 0x8000u) ? 1 : 2; r.ok = ok; r.pushed = ctx.pushed; r.pad = 0;
                            A.evres[time0 >> 1] = r;
                        }
                    }
                    e += 1;
                }
            }
            if (lane == 0) store_main(&stage[s], b, b0 + (int)bl, keep);
        }
    }
    __syncthreads();
    CAFE_STAMP(g.prof, 5);
    // parallel write-back of the per-item outcomes; entries that are not a bucket's first item are marked unused
    for (int s = tid; s < total; s += kThreads) {
        const uint32_t time = L.s.stm[s];
        if (!(time & 1u)) {
            const uint32_t pos = time >> 1;
            const uint8_t o = L.s.po[s];
            A.outm[pos] = o;
            A.miss[pos] = (o & oMiss) ? 1 : 0;
        }
        if (s != 0 && L.s.sg[s - 1] == L.s.sg[s])
            reinterpret_cast<int32_t *>(&stage[s])[13] = -1;         // tag word of store_main
    }
    if (changed) atomicOr(&A.ctl->changed[round], 1u);
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

struct LruArgs {
    const Rec *rec2s;           // round 1: the presumed misses of every partition in owner order (k_presort)
    const uint32_t *pre_nmiss;  // ... and how many there are
    const LruBucket *lt;
    const DevState *st;
    Ctl *ctl;
    const uint32_t *off;
    const Rec *rec2;
    const uint8_t *miss;
    uint32_t *evcnt;
    MoveEv *ev;
    uint8_t *outl;
    LruBucket *stage_l;
    uint32_t *stage_tot;
    uint8_t *ldirty;
    MoveEv *myev;               // [P2][kCapEvL]
    uint32_t *mycnt;
};

// Replays the items of LRU partition q that missed the main table (round 1: the presumed misses of the plan), in
// batch order per bucket, and hands every move out of the table to the main partition of its target bucket.
template <int CAP, bool MV>
__device__ inline void lru_owner(OwnerLds<CAP, MV> &L, int *heavy, int *n_hv, int *n_ev, int *s_total, float *fsum, const Geo &g,
                                 int q, int round, const LruArgs &A, const Score &sc) {
    const int tid = threadIdx.x;
    const int par = round & 1, nxt = par ^ 1;
    const uint32_t o0 = A.off[off_lru0(g) + q], o1 = A.off[off_lru0(g) + q + 1];
    const int cnt = (int)(o1 - o0);
    if (cnt > CAP) { if (tid == 0) A.ctl->overflow = 1; return; }
    MoveEv *mine_ev = A.myev + (size_t)q * g.evls;
    const bool dirty = round == 1 || A.ldirty[par * g.P2 + q];
    if (tid == 0) A.ldirty[nxt * g.P2 + q] = 0;
    if (!dirty) {
        // none of this partition's items changed between hit and miss: same result, so only hand the moves it
        // emitted last time to this round's lists
        const int ne = (int)A.mycnt[q];
        for (int j = tid; j < ne; j += kThreads) {
            const MoveEv m = mine_ev[j];
            const int p = m.pad[0];
            uint32_t at = atomicAdd(&A.evcnt[par * g.P1 + p], 1u);
            if (at < (uint32_t)g.cap_ev) A.ev[(size_t)par * g.P1 * g.evs + (size_t)p * g.evs + at] = m; else A.ctl->overflow = 1;
        }
        return;
    }
    if (tid == 0) { *n_ev = 0; *n_hv = 0; }
    if (cnt == 0) { if (tid == 0) { A.stage_tot[g.P1 + q] = 0; A.mycnt[q] = 0; } return; }
    CAFE_STAMP_VAL(g.prof, 7, 2ull | ((unsigned long long)cnt << 8));
    field_sums(g, sc, fsum);
    const int b0 = q * g.bpp2;
    const Rec *recs = A.rec2 + o0;
    constexpr int IPT = CAP / kThreads;
    const ScoreSrc ss = score_src(g, sc, fsum);
    int nmiss;
    if (round == 1 && A.rec2s) {
        // round 1: the presumed misses come in owner order from the plan (k_presort), with their number
        nmiss = (int)A.pre_nmiss[q];
        if (tid == 0) { A.stage_tot[g.P1 + q] = (uint32_t)nmiss; if (nmiss == 0) A.mycnt[q] = 0; }
        if (nmiss == 0) return;
        constexpr int PRE = IPT < 8 ? IPT : 8;
        Rec rr[PRE];
#pragma unroll
        for (int k = 0; k < PRE; ++k) { const int s = tid + k * kThreads; if (s < nmiss) rr[k] = A.rec2s[o0 + s]; }
        __syncthreads();                // field sums in place
        auto put = [&](int s, const Rec &r) { L.s.sg[s] = r.bl & ~kMissBit; L.s.stm[s] = r.pos; L.s.px[s] = r.x; L.s.pc[s] = ss.at(r.pos); };
#pragma unroll
        for (int k = 0; k < PRE; ++k) { const int s = tid + k * kThreads; if (s < nmiss) put(s, rr[k]); }
        for (int s = tid + PRE * kThreads; s < nmiss; s += kThreads) put(s, A.rec2s[o0 + s]);
        __syncthreads();
    } else {
    // items that did not miss get a key behind every bucket: they sort to the end and are dropped
    const uint32_t dead = 1u << g.bits2;
    int mine = 0;
    for (int j = tid; j < cnt; j += kThreads) {
        const Rec r = recs[j];
        const bool ms = round == 1 ? (r.bl & kMissBit) != 0 : A.miss[r.pos] != 0;
        L.t.grp[j] = ms ? (r.bl & ~kMissBit) : dead;
        mine += ms;
    }
    // number of misses: block-wide sum
    if (tid == 0) *s_total = 0;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((tid & 63) == 0 && mine) atomicAdd(s_total, mine);
    __syncthreads();
    nmiss = *s_total;
    if (tid == 0) { A.stage_tot[g.P1 + q] = (uint32_t)nmiss; if (nmiss == 0) A.mycnt[q] = 0; }
    if (nmiss == 0) return;
    CAFE_STAMP(g.prof, 1);
    lsd_sort<CAP, MV>(L.t, cnt, g.bits2 + 1, false);
    CAFE_STAMP(g.prof, 2);
    uint16_t myr[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) { const int j = tid + k * kThreads; myr[k] = j < cnt ? L.t.lr[j] : 0; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const int j = tid + k * kThreads;
        if (j < cnt && myr[k] < nmiss) {
            const Rec r = recs[j];
            const int s = myr[k];
            L.s.sg[s] = r.bl & ~kMissBit; L.s.stm[s] = r.pos; L.s.px[s] = r.x; L.s.pc[s] = ss.at(r.pos);
        }
    }
    __syncthreads();
    }

    CAFE_STAMP(g.prof, 3);
    const float thr = A.st->thr;
    const uint32_t gflip = (uint32_t)A.st->flip;
    LruBucket *stage = A.stage_l + o0;
    LruCtx ctx;
    ctx.full = A.st->fifo_len > 0;
    ctx.clk0 = A.st->clk;
    int over = 0;
    for (int s = tid; s < nmiss; s += kThreads) {
        const uint32_t bl = L.s.sg[s];
        if (s != 0 && L.s.sg[s - 1] == bl) continue;
        // a bucket with many items (an id that is frequent but not hot yet: it lives here until its score reaches the
        // threshold) is left to phase B, where a whole wave walks it
        // Which replay: a bucket whose items are mostly runs of ONE id (sampled at three places) goes to a wave, which
        // compresses runs; a bucket of many different ids (a small sketch under a large batch: every LRU bucket of cfg2
        // and cfg5 replays tens to hundreds of misses) goes to a lane quad, one lane per slot, whose step per item is a
        // dozen dependent instructions instead of the select chains of the one-lane form.
        if (s + kLruQuad <= nmiss && L.s.sg[s + kLruQuad - 1] == bl) {
            bool runs = false;
            if (s + kLruHeavy <= nmiss && L.s.sg[s + kLruHeavy - 1] == bl) {
                const int same = (L.s.px[s] == L.s.px[s + 1]) + (L.s.px[s + 15] == L.s.px[s + 16]) + (L.s.px[s + 30] == L.s.px[s + 31]);
                runs = same >= 2;
            }
            const int at = atomicAdd(n_hv, 1);
            if (at < kMaxHeavy) { heavy[at] = runs ? s : (s | kQuadTag); continue; }
        }
        LruR b;
        load_lru(b, &A.lt[b0 + (int)bl]);
        sync_flip(b, gflip);
        for (int e = s; e < nmiss && L.s.sg[e] == bl; ++e) {
            const uint32_t pos = L.s.stm[e];
            const uint32_t x = L.s.px[e];
            const float c = L.s.pc[e];
            ctx.pos = pos; ctx.flags = 0;
            MoveReq mv;
            int f = lru_find(b, x), got;
            if (f >= 0) {
                got = lru_hit(b, f, c, thr, ctx, mv);
            } else {
                lru_evict(b, -f - 1, mv);
                got = lru_place(b, -f - 1, x, c, thr, ctx);
            }
            L.s.po[e] = (uint8_t)(ctx.flags | (got ? oGot : 0) | (mv.kind ? oEv : 0));
            if (mv.kind) {
                int p = fast_hash(mv.x, g.fm1) / g.bpp1;
                MoveEv m;
                m.time = 2u * pos + 1u; m.x = mv.x; m.cnt = mv.cnt; m.dic = mv.dic; m.kind = mv.kind;
                m.pad[0] = p; m.pad[1] = m.pad[2] = 0;
                uint32_t at = atomicAdd(&A.evcnt[par * g.P1 + p], 1u);
                if (at < (uint32_t)g.cap_ev) A.ev[(size_t)par * g.P1 * g.evs + (size_t)p * g.evs + at] = m; else over = 1;
                int mi = atomicAdd(n_ev, 1);
                if (mi < g.evls) mine_ev[mi] = m; else over = 1;
            }
        }
        store_lru(&stage[s], b, b0 + (int)bl);
    }
    __syncthreads();
    // Phase Q: buckets of many different ids, one per lane quad (16 per wave, 64 per workgroup at a time).  Lane j of
    // the quad holds slot j of the bucket in registers.  Per item (sketch.cpp:165-209): the hit slot from a ballot of
    // val == id, the victim (first slot with the smallest timestamp) from the four timestamps broadcast inside the quad by
    // DPP, and the lane of the target slot alone applies lru_hit / lru_evict + lru_place to its slot -- no select chains
    // over slots.  The next item's words are requested while the current one is applied.
    {
        const int lane = tid & 63, wave = tid >> 6, sub = lane & 3, qshift = lane & ~3;
        const int nh = min(*n_hv, kMaxHeavy);
        const int hv = lane < nh ? heavy[lane] : 0;
        const unsigned long long qall = __ballot(lane < nh && (hv & kQuadTag));
        for (int base = 0; base < __popcll(qall); base += 16 * kWaves) {
            unsigned long long qm = qall;
            const int gq = base + wave * 16 + (lane >> 2);
            for (int i = 0; i < gq && qm; ++i) qm &= qm - 1;          // the gq-th quad entry of the list
            if (!qm) continue;
            const int s = heavy[__builtin_ctzll(qm)] & ~kQuadTag;
            const uint32_t bl = L.s.sg[s];
            const int32_t *src = reinterpret_cast<const int32_t *>(&A.lt[b0 + (int)bl]);
            int32_t v = src[sub], d = src[4 + sub], t = src[12 + sub];
            float c = __int_as_float(src[8 + sub]);
            if ((uint32_t)src[16] != gflip) c = decay_once(c);           // sync_flip
            int e = s;
            uint32_t x = L.s.px[e], pos = L.s.stm[e];
            float ci = L.s.pc[e];
            bool act = true;
            while (act) {
                const int en = min(e + 1, CAP - 1);
                const uint32_t sgn = L.s.sg[en], xn = L.s.px[en], posn = L.s.stm[en];
                const float cn = L.s.pc[en];
                const uint32_t hn = (uint32_t)(__ballot((uint32_t)v == x) >> qshift) & 0xFu;
                const int32_t t0 = __builtin_amdgcn_update_dpp(0, t, 0x00, 0xF, 0xF, false), t1 = __builtin_amdgcn_update_dpp(0, t, 0x55, 0xF, 0xF, false),
                              t2 = __builtin_amdgcn_update_dpp(0, t, 0xAA, 0xF, 0xF, false), t3 = __builtin_amdgcn_update_dpp(0, t, 0xFF, 0xF, 0xF, false);
                int victim = -1, t_min = 1000000000;                    // lru_find: first slot with the smallest timestamp
                if (t_min > t0) { t_min = t0; victim = 0; }
                if (t_min > t1) { t_min = t1; victim = 1; }
                if (t_min > t2) { t_min = t2; victim = 2; }
                if (t_min > t3) { t_min = t3; victim = 3; }
                if (victim < 0) victim = 0;
                const bool hit = hn != 0u;
                const int target = hit ? __builtin_ctz(hn) : victim;
                // both transitions of the slot, computed by every lane on its own slot without a branch; the target lane
                // keeps the one that applies (lru_hit / lru_evict + lru_place, cafe_core.h)
                const bool mine = sub == target;
                const int32_t tk = ctx.clk0 + 1 + (int32_t)pos;          // LruCtx::tick
                const int32_t pl = (int32_t)(kSur | pos);                // LruCtx::pop
                const float now = c + ci;
                const bool reach = now >= thr;
                const bool h_pop = reach && ctx.full && d == 0;
                const int32_t h_dic = h_pop ? pl : d;
                const bool m_ev = d != 0 || c >= 5.0f;
                const bool m_one = 1.0f >= thr;
                const bool m_pop = m_one && ctx.full;
                const uint8_t fl = hit ? (uint8_t)(((reach && now - ci < thr) ? oCross : 0) | (h_pop ? (oPop | oGot) : 0) | (reach ? oEv : 0))
                                       : (uint8_t)(((m_one && 1.0f - ci < thr) ? oCross : 0) | (m_pop ? (oPop | oGot) : 0) | (m_ev ? oEv : 0));
                const int mk = hit ? (reach ? 1 : 0) : (m_ev ? 2 : 0);
                const uint32_t mx = (uint32_t)v;
                const float mc = hit ? now : c;
                const int32_t md = hit ? h_dic : d;
                const int32_t nv = hit ? (reach ? 0 : v) : (int32_t)x;
                const int32_t nt = hit ? (reach ? 0 : tk) : tk;
                const float nc = hit ? (reach ? 0.0f : now) : 1.0f;
                const int32_t nd = hit ? (reach ? 0 : h_dic) : (m_pop ? pl : 0);
                v = mine ? nv : v; t = mine ? nt : t; c = mine ? nc : c; d = mine ? nd : d;
                if (mine) {
                    L.s.po[e] = fl;
                    if (mk) {
                        const int p = fast_hash(mx, g.fm1) / g.bpp1;
                        MoveEv m;
                        m.time = 2u * pos + 1u; m.x = mx; m.cnt = mc; m.dic = md; m.kind = mk;
                        m.pad[0] = p; m.pad[1] = m.pad[2] = 0;
                        const uint32_t at = atomicAdd(&A.evcnt[par * g.P1 + p], 1u);
                        if (at < (uint32_t)g.cap_ev) A.ev[(size_t)par * g.P1 * g.evs + (size_t)p * g.evs + at] = m; else over = 1;
                        const int mi = atomicAdd(n_ev, 1);
                        if (mi < g.evls) mine_ev[mi] = m; else over = 1;
                    }
                }
                ++e;
                act = e < nmiss && sgn == bl;
                x = xn; pos = posn; ci = cn;
            }
            int32_t *dst = reinterpret_cast<int32_t *>(&stage[s]);       // store_lru, a slot per lane
            dst[sub] = v; dst[4 + sub] = d; dst[8 + sub] = __float_as_int(c); dst[12 + sub] = t;
            if (sub == 0) *reinterpret_cast<int4 *>(dst + 16) = make_int4((int)gflip, b0 + (int)bl, 0, 0);
        }
    }
    // Phase B: buckets of long runs of one id, one per wave at a time; all 64 lanes carry the same bucket copy.  A run of
    // probes of one id that sits in the bucket and stays below the threshold only moves its score (ordered float sum: one
    // readlane + add per item) and its timestamp; the first probe that reaches the threshold, and every miss, takes the
    // one-item step (sketch.cpp:165-209), executed uniformly with lane 0 writing.
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int nh = min(*n_hv, kMaxHeavy);
        for (int h = wave; h < nh; h += kWaves) {
            if (heavy[h] & kQuadTag) continue;
            const int s = heavy[h];
            const uint32_t bl = L.s.sg[s];
            LruR b;
            load_lru(b, &A.lt[b0 + (int)bl]);
            sync_flip(b, gflip);
            // the end of the bucket's items, found 64 at a time (the walk below then needs no key per item)
            int end = s;
            for (;;) {
                const int idx = end + lane;
                const unsigned long long m = __ballot(idx < nmiss && L.s.sg[idx] == bl);
                const int rr = (~m) ? __builtin_ctzll(~m) : 64;
                end += rr;
                if (rr < 64) break;
            }
            int e = s;
            while (e < end) {
                // id, position and score of the item in one LDS round trip
                const uint32_t x = L.s.px[e];
                const uint32_t pos = L.s.stm[e];
                const float c = L.s.pc[e];
                const int f = lru_find(b, x);
                if (f >= 0) {
                    const int idx = e + lane;
                    const bool same = idx < end && L.s.px[idx] == x;
                    const unsigned long long m = __ballot(same);
                    const int r = (~m) ? __builtin_ctzll(~m) : 64;            // length of the run starting at e (>= 1)
                    const float cl = lane < r ? L.s.pc[idx] : 0.0f;
                    float acc = sel4f(b.c0, b.c1, b.c2, b.c3, f), mine = 0.0f;
                    for (int j = 0; j < r; ++j) {
                        acc += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cl), j));
                        mine = lane == j ? acc : mine;
                    }
                    const unsigned long long km = __ballot(lane < r && mine >= thr);
                    const int k = km ? __builtin_ctzll(km) : r;                // probes before the first one that reaches thr
                    if (k > 0) {
                        const float upto = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), k - 1));
                        const int32_t tlast = ctx.clk0 + 1 + (int32_t)L.s.stm[e + k - 1];
                        CAFE_SET4(b.c0, b.c1, b.c2, b.c3, f, upto);
                        CAFE_SET4(b.t0, b.t1, b.t2, b.t3, f, tlast);
                        if (lane < k) L.s.po[e + lane] = 0;
                        e += k;
                        continue;
                    }
                }
                ctx.pos = pos; ctx.flags = 0;
                MoveReq mv;
                int got;
                if (f >= 0) {
                    got = lru_hit(b, f, c, thr, ctx, mv);
                } else {
                    lru_evict(b, -f - 1, mv);
                    got = lru_place(b, -f - 1, x, c, thr, ctx);
                }
                if (lane == 0) {
                    L.s.po[e] = (uint8_t)(ctx.flags | (got ? oGot : 0) | (mv.kind ? oEv : 0));
                    if (mv.kind) {
                        int p = fast_hash(mv.x, g.fm1) / g.bpp1;
                        MoveEv mm;
                        mm.time = 2u * pos + 1u; mm.x = mv.x; mm.cnt = mv.cnt; mm.dic = mv.dic; mm.kind = mv.kind;
                        mm.pad[0] = p; mm.pad[1] = mm.pad[2] = 0;
                        uint32_t at = atomicAdd(&A.evcnt[par * g.P1 + p], 1u);
                        if (at < (uint32_t)g.cap_ev) A.ev[(size_t)par * g.P1 * g.evs + (size_t)p * g.evs + at] = mm; else over = 1;
                        int mi = atomicAdd(n_ev, 1);
                        if (mi < g.evls) mine_ev[mi] = mm; else over = 1;
                    }
                }
                e += 1;
            }
            if (lane == 0) store_lru(&stage[s], b, b0 + (int)bl);
        }
    }
    __syncthreads();
    CAFE_STAMP(g.prof, 4);
    if (tid == 0) {
        A.mycnt[q] = (uint32_t)min(*n_ev, g.evls);
        if (round == 1 && *n_ev) atomicAdd(&A.ctl->nmoves, (uint32_t)*n_ev);
    }
    for (int s = tid; s < nmiss; s += kThreads) {
        A.outl[L.s.stm[s]] = L.s.po[s];
        if (s != 0 && L.s.sg[s - 1] == L.s.sg[s])
            reinterpret_cast<int32_t *>(&stage[s])[17] = -1;         // tag word of store_lru
    }
    if (over) A.ctl->overflow = 1;
}

// ------------------------------------------------------------------------------------------------
// round 1: hot partitions, LRU partitions and main partitions (class C) in one launch
// ------------------------------------------------------------------------------------------------
struct HotArgs {
    const uint32_t *order;      // the owners by estimated cost, heaviest first (k_offsets), or nullptr: in grid order
    const Rec *rechs;           // the hot list's partitions in owner order (k_presort)
    const Rec *rech;
    unsigned long long *hkey;
    float *hrun;
    int2 *hfin;
};

template <int CAP>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(CAP <= kCapSmall ? 6 : 1, 8))) k_round1(Geo g, MainArgs MA, LruArgs LA, HotArgs HA, Score sc) {
    CAFE_KWIN(3);
    __shared__ __attribute__((aligned(16))) OwnerLds<CAP, false> L;
    __shared__ int heavy[kMaxHeavy];
    __shared__ int n_heavy, s_total, n_hv;
    __shared__ float fsum[kMaxSeg];
    // (no look at the control word here: owners only write staging buffers, and k_finish / k_commit refuse a batch whose
    // control word reports a limit -- one memory round trip less in every owner)
    CAFE_STAMP_VAL(0, 7, 0ull);
    CAFE_STAMP(0, 0);
    int b = HA.order ? (int)HA.order[blockIdx.x] : (int)blockIdx.x;
    const int nhot = !g.hot ? 0 : g.hot_table ? (g.PH + kWaves - 1) / kWaves : g.PH;
    if (b < nhot) {
        if (g.hot_table) {
            // a wave per hot partition; the waves share nothing but the field sums
            field_sums(g, sc, fsum);
            __syncthreads();
            float *tab = reinterpret_cast<float *>(&L);
            const int wave = threadIdx.x >> 6, ph = b * kWaves + wave;
            static_assert(sizeof(L) >= kWaves * (kHotKeys + 64) * sizeof(float), "hot table");
            CAFE_STAMP_VAL(0, 7, 1ull);
            if (ph < g.PH)
                hot_table_owner(tab + wave * (kHotKeys + 64), tab + wave * (kHotKeys + 64) + kHotKeys, fsum, g, ph, MA.mt, MA.st, MA.off,
                                HA.rech, sc, HA.hkey, HA.hrun, HA.hfin, MA.outm);
        } else {
            hot_owner<CAP, false>(L, heavy, &n_heavy, fsum, g, b, MA.mt, MA.st, MA.ctl, MA.off, HA.rech, HA.rechs, sc, HA.hkey, HA.hrun, HA.hfin, MA.outm);
        }
        CAFE_STAMP(0, 6);
        return;
    }
    b -= nhot;
    if (b < g.P2) {
        lru_owner<CAP, false>(L, heavy, &n_hv, &n_heavy, &s_total, fsum, g, b, 1, LA, sc);
        CAFE_STAMP(0, 6);
        return;
    }
    b -= g.P2;
    main_owner<CAP, false, false>(L, heavy, &n_heavy, fsum, g, b, 1, MA, sc);
    CAFE_STAMP(0, 6);
}

// ------------------------------------------------------------------------------------------------
// rounds >= 2: one persistent grid of kFixGrid workgroups
// ------------------------------------------------------------------------------------------------
// Grid-wide barrier (monotonic counter, MI355X_MICROARCH.md "barrier-counter"): every workgroup's stores are released
// at agent scope before it arrives and the CU's L1 is invalidated after the wait.  The last arriver also samples the
// control words every workgroup is about to branch on (a workspace limit, "the miss set changed in round r") and
// publishes them WITH the generation: all workgroups act on the same latched value, so none can leave a round that
// another one still enters (read after the barrier, such a flag could be raised meanwhile by a workgroup that is already
// in the next half-round: one workgroup would return, the others would wait for it at the next barrier).  All workgroups
// are resident (the kernel keeps to 128 vector registers, which the gradient pass of a split update leaves free on every
// compute unit); the wait is bounded as a safety net: a barrier that does not complete flags the batch as unsettled.
constexpr uint32_t kLatchOver = 1u << 30, kLatchChanged = 1u << 31, kGenMask = (1u << 30) - 1u;
__device__ inline uint32_t grid_barrier(Ctl *ctl, uint32_t &gen, int sample_round) {
    __shared__ uint32_t latched;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t target = (gen + 1u) * gridDim.x;
        const uint32_t arrived = __hip_atomic_fetch_add(&ctl->bar_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        uint32_t word;
        if (arrived == target) {
            // everything the other workgroups wrote before they arrived is visible behind this fence
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            word = gen + 1u;
            if (__hip_atomic_load(&ctl->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) word |= kLatchOver;
            if (sample_round > 0 && __hip_atomic_load(&ctl->changed[sample_round], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) word |= kLatchChanged;
            __hip_atomic_store(&ctl->bar_gen, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            long spins = 0;
            while (((word = __hip_atomic_load(&ctl->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & kGenMask) < gen + 1u) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1L << 24)) {           // never seen: k_finish / k_commit then refuse the batch
                    __hip_atomic_store(&ctl->overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    word = (gen + 1u) | kLatchOver;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        latched = word;
    }
    __syncthreads();
    gen += 1u;
    return latched;
}

template <int CAP>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(CAP <= 2048 ? 4 : 1, 8))) k_fix(Geo g, MainArgs MA, LruArgs LA, Score sc) {
    CAFE_KWIN(4);
    g.prof = 5;
    __shared__ __attribute__((aligned(16))) OwnerLds<CAP, true> L;
    __shared__ int heavy[kMaxHeavy];
    __shared__ int n_heavy, s_total, n_hv;
    __shared__ float fsum[kMaxSeg];
    Ctl *ctl = MA.ctl;
    const int tid = threadIdx.x;
    // both words were final before this launch: the same value in every workgroup
    if (ctl->overflow) return;
    if (ctl->nmoves == 0) {
        if (blockIdx.x == 0 && tid == 0) ctl->final_round = 1;
        return;
    }
    uint32_t gen = 0;
    for (int round = 2; round <= g.max_rounds; ++round) {
        const int prev = (round & 1) ^ 1;
        const int pslot = round == 2 ? 6 : 7;       // diagnostic builds: rounds 2 and 3 (and later, overwriting) are stamped
        unsigned long long ran = 0;
        (void)pslot; (void)ran;
        CAFE_STAMP(pslot, 0);
        for (int p = blockIdx.x; p < g.P1; p += gridDim.x) {
            // partitions that have moves now, or ran with moves before (they may have to run without)
            const uint32_t need = tid == 0 ? (MA.evcnt[prev * g.P1 + p] | MA.prevcnt[p]) : 0u;
            if (!__syncthreads_or(need != 0)) continue;
            main_owner<CAP, true, true>(L, heavy, &n_heavy, fsum, g, p, round, MA, sc);
            __syncthreads();
            ran += 1;
        }
        CAFE_STAMP(pslot, 1);
        uint32_t latch = grid_barrier(ctl, gen, round);
        CAFE_STAMP(pslot, 2);
        if (latch & kLatchOver) return;
        if (!(latch & kLatchChanged)) {
            if (blockIdx.x == 0 && tid == 0) ctl->final_round = (uint32_t)round;
            CAFE_STAMP_VAL(pslot, 5, ran);
            return;
        }
        if (round == g.max_rounds) return;         // not settled
        const int par = round & 1, nxt = par ^ 1;
        for (int p = blockIdx.x * kThreads + tid; p < g.P1; p += gridDim.x * kThreads) LA.evcnt[nxt * g.P1 + p] = 0;
        for (int q = blockIdx.x; q < g.P2; q += gridDim.x) {
            lru_owner<CAP, true>(L, heavy, &n_hv, &n_heavy, &s_total, fsum, g, q, round, LA, sc);
            __syncthreads();
            ran += 1ull << 16;
        }
        CAFE_STAMP(pslot, 3);
        latch = grid_barrier(ctl, gen, 0);
        CAFE_STAMP(pslot, 4);
        CAFE_STAMP_VAL(pslot, 5, ran);
        if (latch & kLatchOver) return;
    }
}

// ------------------------------------------------------------------------------------------------
// finish: batch-ordered ranks in one pass, commit
// ------------------------------------------------------------------------------------------------
// Per item: final outcome (ins[]), whether it pops / pushes / ticks the LRU clock, and the exclusive ranks of those
// three events in batch order.  Workgroups publish their totals as two self-tagged 8-byte words and sum the words of
// the workgroups in front of them (decoupled look-back: a workgroup only ever waits for lower-numbered ones).
// A workgroup of 256 lanes takes kScanBlock = 1024 consecutive items, four chunks of 256: the whole grid (one workgroup
// per 1024 items, four waves of ~100 registers) is resident at once even beside the gradient pass of a split update,
// and everything a lane needs from memory is requested before any of it is looked at.
constexpr int kFinI = kScanBlock / kThreads;
__global__ void __launch_bounds__(kThreads) k_finish(
    Geo g, unsigned long long call_tag, Ctl *ctl, const uint8_t *__restrict__ miss, const uint8_t *__restrict__ outm,
    const uint8_t *__restrict__ outl, const EvRes *__restrict__ evres, int32_t *__restrict__ ins, uint8_t *__restrict__ flags,
    int32_t *__restrict__ pushval, unsigned long long *agg /* [nblocks][2] */, uint32_t *__restrict__ poprank,
    uint32_t *__restrict__ pushrank, uint32_t *__restrict__ clkrank, uint32_t *__restrict__ gotlist, Score sc,
    float *__restrict__ scores_out) {
    CAFE_KWIN(5);
    __shared__ uint32_t wsum[kFinI][3][kWaves];
    __shared__ uint32_t s_cross;
    __shared__ uint32_t s_base[4];
    __shared__ float fsum[kMaxSeg];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int base = blockIdx.x * kScanBlock;
    CAFE_STAMP(1, 0);
    // the scores as every owner formed them, for the caller (sk_embedding_bag.py:189): written whatever the verdict, but
    // AFTER this workgroup has published its totals -- the workgroups behind it wait for those, not for the scores
    // (what the scores need from memory -- the raw values, the fields' partial sums -- is requested with the kernel's first
    // loads: between a workgroup's totals and its ranks only arithmetic and the stores are left)
    uint8_t mm[kFinI], om[kFinI], ol[kFinI];
    float sv[kFinI];
#pragma unroll
    for (int j = 0; j < kFinI; ++j) {
        const int i = base + j * kThreads + tid;
        const bool in = i < g.n;
        mm[j] = in ? miss[i] : 0; om[j] = in ? outm[i] : 0; ol[j] = in ? outl[i] : 0;
        sv[j] = (scores_out && in && sc.vals) ? sc.vals[(size_t)i * (size_t)sc.vstride] : 1.0f;
    }
    const bool live = !ctl->overflow && ctl->final_round != 0;
    if (scores_out) field_sums(g, sc, fsum);
    auto write_scores = [&]() {
        if (!scores_out) return;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kFinI; ++j) {
            const int i = base + j * kThreads + tid;
            if (i < g.n) scores_out[i] = sc.psum ? sv[j] * (float)g.sseg / fsum[(uint32_t)i / (uint32_t)g.sseg] : sv[j];      // cafe_core.h::ScoreSrc::at
        }
    };
    CAFE_STAMP(1, 1);
    if (!live) { write_scores(); return; }             // uniform; k_commit decides what happens to the batch
    if (tid == 0) s_cross = 0;
    EvRes er[kFinI];
#pragma unroll
    for (int j = 0; j < kFinI; ++j) {
        const int i = base + j * kThreads + tid;
        er[j].tag = 0; er[j].kind = 0; er[j].ok = 1; er[j].pushed = 0;
        if (i < g.n && mm[j] && (ol[j] & oEv)) er[j] = evres[i];
    }
    unsigned long long m0[kFinI], m1[kFinI], m2[kFinI];
    uint32_t ncross = 0;
#pragma unroll
    for (int j = 0; j < kFinI; ++j) {
        const int i = base + j * kThreads + tid;
        uint32_t pop = 0, push = 0, clk = 0, cross = 0;
        if (i < g.n) {
            const uint8_t m = mm[j];
            const uint8_t o = m ? ol[j] : om[j];
            int got = (o & oGot) ? 1 : 0;
            pop = (o & oPop) ? 1 : 0;
            cross = (o & oCross) ? 1 : 0;
            clk = m ? 1 : 0;
            int32_t pv = 0;
            if (m && (o & oEv) && er[j].tag == call_tag) {
                if (er[j].kind == 1 && !er[j].ok) got = 0;           // sketch.cpp:180-182
                if (!er[j].ok && er[j].pushed != 0) { push = 1; pv = er[j].pushed; }
            }
            ins[i] = got;
            pushval[i] = pv;
            flags[i] = (uint8_t)(pop | (push << 1) | (clk << 2));
            if (cross && g.seg > 0) atomicAdd(&ctl->segcross[i / g.seg], 1u);      // a few hundred per call
            if (got) gotlist[atomicAdd(&ctl->ngot, 1u)] = (uint32_t)i;
        }
        m0[j] = __ballot(pop); m1[j] = __ballot(push); m2[j] = __ballot(clk);
        ncross += (uint32_t)__popcll(__ballot(cross));
        if (lane == 0) { wsum[j][0][w] = __popcll(m0[j]); wsum[j][1][w] = __popcll(m1[j]); wsum[j][2][w] = __popcll(m2[j]); }
    }
    CAFE_STAMP(1, 2);
    const unsigned long long below = lanes_below();
    __syncthreads();
    if (lane == 0 && ncross) atomicAdd(&s_cross, ncross);
    // exclusive prefix of this lane's chunk/wave among the workgroup's (chunk, wave) pairs, in batch order
    uint32_t t0 = 0, t1 = 0, t2 = 0, b0[kFinI], b1[kFinI], b2[kFinI];
#pragma unroll
    for (int j = 0; j < kFinI; ++j)
#pragma unroll
        for (int k = 0; k < kWaves; ++k) {
            if (k == w) { b0[j] = t0; b1[j] = t1; b2[j] = t2; }
            t0 += wsum[j][0][k]; t1 += wsum[j][1][k]; t2 += wsum[j][2][k];
        }
    __syncthreads();
    // publish this workgroup's totals, then sum the words in front of it: a lane per workgroup in front, all waves at once
    if (tid == 0) {
        __hip_atomic_store(&agg[2 * blockIdx.x], (1ull << 63) | ((unsigned long long)t0 << 24) | t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&agg[2 * blockIdx.x + 1], (1ull << 63) | ((unsigned long long)s_cross << 24) | t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_base[0] = s_base[1] = s_base[2] = s_base[3] = 0;
    }
    write_scores();
    __syncthreads();
    {
        unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int bq = (int)blockIdx.x - 1 - tid; bq >= 0; bq -= kThreads) {
            unsigned long long v0 = __hip_atomic_load(&agg[2 * bq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long v1 = __hip_atomic_load(&agg[2 * bq + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long spins = 0;
            while (!(v0 >> 63) && ++spins < (1L << 24)) v0 = __hip_atomic_load(&agg[2 * bq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (!(v1 >> 63) && ++spins < (1L << 24)) v1 = __hip_atomic_load(&agg[2 * bq + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!(v0 >> 63) || !(v1 >> 63)) ctl->overflow = 1;         // never seen; k_commit then refuses the batch
            a0 += (v0 >> 24) & 0xffffffull; a1 += v0 & 0xffffffull; a2 += v1 & 0xffffffull; a3 += (v1 >> 24) & 0xffffffull;
        }
        if (__ballot(a0 | a1 | a2 | a3)) {
            for (int o = 32; o > 0; o >>= 1) {
                a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); a3 += __shfl_xor(a3, o, 64);
            }
            if (lane == 0) {
                atomicAdd(&s_base[0], (uint32_t)a0); atomicAdd(&s_base[1], (uint32_t)a1); atomicAdd(&s_base[2], (uint32_t)a2);
                atomicAdd(&s_base[3], (uint32_t)a3);
            }
        }
    }
    __syncthreads();
    CAFE_STAMP(1, 3);
#pragma unroll
    for (int j = 0; j < kFinI; ++j) {
        const int i = base + j * kThreads + tid;
        if (i < g.n) {
            poprank[i] = s_base[0] + b0[j] + (uint32_t)__popcll(m0[j] & below);
            pushrank[i] = s_base[1] + b1[j] + (uint32_t)__popcll(m1[j] & below);
            clkrank[i] = s_base[2] + b2[j] + (uint32_t)__popcll(m2[j] & below);
        }
    }
    // the last workgroup's sums are the batch totals (no contended atomics: 200 workgroups adding to one line cost ~10 us)
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        ctl->totals[0] = s_base[0] + t0; ctl->totals[1] = s_base[1] + t1; ctl->totals[2] = s_base[2] + t2;
        ctl->totals[3] = s_base[3] + s_cross;
    }
    CAFE_STAMP(1, 4);
}

__device__ inline int32_t resolve_dic(int32_t d, const uint32_t *fifo, uint32_t cap, uint32_t head, const uint32_t *poprank,
                                      uint32_t n) {
    if ((uint32_t)d & kSur) {
        uint32_t pos = (uint32_t)d & ~kSur;
        if (pos >= n) return 0;
        return (int32_t)fifo[(head + poprank[pos]) % cap];
    }
    return d;
}

struct CommitArgs {
    MainBucket *mt;
    LruBucket *lt;
    uint32_t *fifo;
    uint32_t cap;
    DevState *st;
    Ctl *ctl;
    const uint32_t *off;
    const uint32_t *stage_tot;
    const MainBucket *stage_m;
    const LruBucket *stage_l;
    const uint8_t *flags;
    const int32_t *pushval;
    const uint32_t *poprank, *pushrank, *clkrank;
    const int2 *hfin;
    unsigned long long *agg;
    int nagg;
    const int32_t *ids;
    int32_t *ins;
    Mirror *mirror;
    uint32_t seq;
    double reset_at;
    uint32_t *gotcount;         // what the migration kernel reads: number of got-list entries, 0xffffffff = scan ins[]
    // migration of the ids that took a slot (sk_embedding_bag.py:199-208), done by the lane that commits their bucket;
    // w_hot == nullptr: none (the caller migrates by itself)
    float *w_hot;
    float *const *w_cold;
    const int32_t *hash_size;
    const int64_t *local_ids;   // per-field ids in field-major order (field = position / batch), or nullptr:
    const int32_t *mig_gids, *mig_fields;       // ... global ids with a field per lookup (local id = gid - id_offset[field])
    const int64_t *id_offset;
    int batch, dim, hotn;
};

// commit: decision; staged buckets -> tables with placeholders resolved; pushes -> free list; the workgroup that
// finishes last applies the batch totals to the device scalars and leaves the snapshot for the host -- or, for a batch
// that was not settled and is not to be handed back to the host, walks it in order from the untouched tables.
__global__ void __launch_bounds__(256) k_commit(Geo g, CommitArgs A, Score sc) {
    CAFE_KWIN(6);
    __shared__ int s_verdict, s_reason, s_last;
    __shared__ float fsum[kMaxSeg];
    const int tid = threadIdx.x;
    DevState *st = A.st;
    const Ctl *ctl = A.ctl;
    CAFE_STAMP(2, 0);
    // what the copy loop below needs first -- the scalars of the sketch and this workgroup's partition -- is requested
    // together with the words of the decision, not behind it (none of them changes before the last workgroup is done)
    const uint32_t head = st->fifo_head, len0 = st->fifo_len;
    const int32_t clk0 = st->clk;
    const int part_mine = blockIdx.x;
    int tot_mine = 0;
    uint32_t offa_mine = 0, offb_mine = 0;
    if (part_mine < g.P1 + g.P2) {
        tot_mine = (int)A.stage_tot[part_mine];
        if (part_mine < g.P1) { offa_mine = A.off[2 * part_mine]; offb_mine = A.off[2 * part_mine + 2]; }
        else { offa_mine = A.off[off_lru0(g) + (part_mine - g.P1)]; offb_mine = A.off[off_lru0(g) + (part_mine - g.P1) + 1]; }
    } else if (g.hot && part_mine < g.P1 + g.P2 + g.PH) {
        offa_mine = A.off[off_hot0(g) + (part_mine - g.P1 - g.P2)]; offb_mine = A.off[off_hot0(g) + (part_mine - g.P1 - g.P2) + 1];
    }
    // every workgroup takes the same decision from the same words; all of them are requested before the first is looked at
    if (tid < 64) {
        const uint32_t len0 = st->fifo_len, w_over = ctl->overflow, w_final = ctl->final_round, w_pop = ctl->totals[0],
                       w_push = ctl->totals[1];
        const int32_t real_n = st->real_n;
        const int nseg = g.seg > 0 ? (g.n + g.seg - 1) / g.seg : 0;
        long run = tid < nseg - 1 ? (long)ctl->segcross[tid] : 0;
        int reason = 0;
        if (w_over) reason = 1;
        else if (w_final == 0) reason = 2;
        else if (len0 > 0 ? w_pop > len0 : w_push != 0) reason = 3;       // never ran dry / never refilled
        // sketch.cpp:291 at the inner boundaries, with the crossings counted per segment
        for (int o = 1; o < 64; o <<= 1) { const long v = __shfl_up(run, o, 64); if (tid >= o) run += v; }
        const bool fires = tid < nseg - 1 && (double)(run + real_n) > A.reset_at;
        if (!reason && __ballot(fires)) reason = 4;
        if (tid == 0) {
            s_reason = reason;
            s_verdict = !reason ? 0 : (g.verify || reason == 4) ? 2 : 1;       // 0 commit, 1 in-order loop, 2 rejected
        }
    }
    __syncthreads();
    const int verdict = s_verdict;
    CAFE_STAMP(2, 1);
    if (verdict == 0) {
        // A slot this batch handed out sits in a staged main bucket as a placeholder: the lane that resolves it also
        // copies the row of the id that took it from its cold table -- the PRE-update row under the LOCAL id
        // (sk_embedding_bag.py:205-208; the row updates run behind this kernel).  An id that popped a slot
        // ends up in the main table with that slot exactly when the reference reports ins != 0 and its re-query reads
        // hot: a slot popped on a main hit stays where it is (hot slots are never replaced, sketch.cpp:253-258); one
        // popped in the LRU table arrives with its entry's move, or is pushed back when that move fails (ins cleared,
        // sketch.cpp:180-182); one that stays in the LRU table is not hot for query() (SURVEY Q3 fence).
        // The copy itself is the wave's: the lanes that found a placeholder are taken one after the other, and all 64
        // lanes move that row (one memory round trip per row instead of one per 16 bytes).
        const int dim4 = A.dim >> 2;
        const int lane = tid & 63;
        auto migrate_wave = [&](int32_t placeholder, int32_t slot) {          // called by all lanes of the wave
            unsigned long long mk = __ballot(A.w_hot != nullptr && ((uint32_t)placeholder & kSur) && slot > 0);
            while (mk) {
                const int l = __builtin_ctzll(mk);
                mk &= mk - 1ull;
                const uint32_t pos = (uint32_t)__shfl(placeholder, l, 64) & ~kSur;
                const uint32_t sl = (uint32_t)__shfl(slot, l, 64);
                const int f = A.local_ids ? (int)(pos / (uint32_t)A.batch) : A.mig_fields[pos];
                const int64_t local = A.local_ids ? A.local_ids[pos] : (int64_t)A.mig_gids[pos] - A.id_offset[f];
                const int64_t src_row = local % (int64_t)A.hash_size[f];
                const float4 *src = reinterpret_cast<const float4 *>(A.w_cold[f] + (size_t)src_row * A.dim);
                float4 *dst = reinterpret_cast<float4 *>(A.w_hot + (size_t)(sl % (uint32_t)A.hotn) * A.dim);
                for (int c = lane; c < dim4; c += 64) dst[c] = src[c];
            }
        };
        const int nparts = g.P1 + g.P2 + (g.hot ? g.PH : 0);
        for (int part = blockIdx.x; part < nparts; part += gridDim.x) {
            const bool mine = part == part_mine;
            if (part >= g.P1 + g.P2) {
                // final scores of the bypassed hot slots of one hot partition
                const int ph = part - g.P1 - g.P2;
                const uint32_t oh0 = mine ? offa_mine : A.off[off_hot0(g) + ph];
                const int nh = (int)((mine ? offb_mine : A.off[off_hot0(g) + ph + 1]) - oh0);
                for (int s = tid; s < nh; s += blockDim.x) {
                    const int2 f = A.hfin[oh0 + s];
                    if (f.x >= 0 && (f.x >> 2) >= g.n1) { atomicAdd(&st->bad_commit, 1); continue; }
                    if (f.x >= 0) A.mt[f.x >> 2].cnt[f.x & 3] = __int_as_float(f.y);
                }
                continue;
            }
            int tot = mine ? tot_mine : (int)A.stage_tot[part];
            // Everything below indexes the tables with numbers read from the staging buffers.  They are written by the
            // owners of this very call; the range checks turn a broken invariant into a counted refusal, not a wild store.
            const uint32_t offa = mine ? offa_mine : part < g.P1 ? A.off[2 * part] : A.off[off_lru0(g) + (part - g.P1)];
            const uint32_t offb = mine ? offb_mine : part < g.P1 ? A.off[2 * part + 2] : A.off[off_lru0(g) + (part - g.P1) + 1];
            const int room = (int)(offb - offa) + (part < g.P1 ? g.evs : 0);
            if (tot < 0 || tot > room) { if (tid == 0) atomicAdd(&st->bad_commit, 1); tot = 0; }
            if (part < g.P1) {
                const MainBucket *stage = A.stage_m + offa + (size_t)part * g.evs;
                for (int sb = 0; sb < tot; sb += blockDim.x) {           // uniform: every lane of a wave reaches the migration
                    const int s = sb + tid;
                    bool on = s < tot;
                    const int4 *p = reinterpret_cast<const int4 *>(&stage[on ? s : 0]);
                    const int4 m = p[3];
                    on = on && m.y >= 0;
                    if (on && m.y >= g.n1) { atomicAdd(&st->bad_commit, 1); on = false; }
                    int4 d0 = make_int4(0, 0, 0, 0), d = d0;
                    if (on) {
                        const int4 v = p[0], c = p[2];
                        d0 = p[1];
                        d = d0;
                        d.x = resolve_dic(d.x, A.fifo, A.cap, head, A.poprank, (uint32_t)g.n); d.y = resolve_dic(d.y, A.fifo, A.cap, head, A.poprank, (uint32_t)g.n);
                        d.z = resolve_dic(d.z, A.fifo, A.cap, head, A.poprank, (uint32_t)g.n); d.w = resolve_dic(d.w, A.fifo, A.cap, head, A.poprank, (uint32_t)g.n);
                        int4 *o = reinterpret_cast<int4 *>(&A.mt[m.y]);
                        // A slot that was hot before the batch, in a bucket without a pending decay, went through the bypass:
                        // its score is written by the hot partition's workgroup (above) -- at any time, so the staged copy's
                        // stale score must not be stored over it, not even as part of a wider store.  The owner left the
                        // mask of such slots in the staged bucket (cafe_core.h::hot_keep_mask): no look at the table here.
                        const int keep = g.hot ? (m.z & 15) : 0;
                        o[0] = v; o[1] = d;
                        if (!keep) {
                            o[2] = c;
                        } else {
                            int32_t *oc = reinterpret_cast<int32_t *>(&o[2]);
                            if (!(keep & 1)) oc[0] = c.x;
                            if (!(keep & 2)) oc[1] = c.y;
                            if (!(keep & 4)) oc[2] = c.z;
                            if (!(keep & 8)) oc[3] = c.w;
                        }
                        A.mt[m.y].flip = (uint32_t)m.x;
                    }
                    migrate_wave(d0.x, d.x); migrate_wave(d0.y, d.y); migrate_wave(d0.z, d.z); migrate_wave(d0.w, d.w);
                }
            } else {
                const LruBucket *stage = A.stage_l + offa;
                for (int s = tid; s < tot; s += blockDim.x) {
                    const int4 *p = reinterpret_cast<const int4 *>(&stage[s]);
                    const int4 m = p[4];
                    if (m.y < 0) continue;
                    int4 v = p[0], d = p[1], c = p[2], t = p[3];
                    const long tmax = (long)clk0 + g.n;
                    if (m.y >= g.n2 || t.x > tmax || t.y > tmax || t.z > tmax || t.w > tmax) { atomicAdd(&st->bad_commit, 1); continue; }
                    d.x = resolve_dic(d.x, A.fifo, A.cap, head, A.poprank, (uint32_t)g.n); d.y = resolve_dic(d.y, A.fifo, A.cap, head, A.poprank, (uint32_t)g.n);
                    d.z = resolve_dic(d.z, A.fifo, A.cap, head, A.poprank, (uint32_t)g.n); d.w = resolve_dic(d.w, A.fifo, A.cap, head, A.poprank, (uint32_t)g.n);
                    if (t.x > clk0) t.x = clk0 + 1 + (int32_t)A.clkrank[t.x - clk0 - 1];
                    if (t.y > clk0) t.y = clk0 + 1 + (int32_t)A.clkrank[t.y - clk0 - 1];
                    if (t.z > clk0) t.z = clk0 + 1 + (int32_t)A.clkrank[t.z - clk0 - 1];
                    if (t.w > clk0) t.w = clk0 + 1 + (int32_t)A.clkrank[t.w - clk0 - 1];
                    int4 *o = reinterpret_cast<int4 *>(&A.lt[m.y]);
                    o[0] = v; o[1] = d; o[2] = c; o[3] = t;
                    A.lt[m.y].flip = (uint32_t)m.x;
                }
            }
        }
        if (ctl->totals[1] != 0) {
            for (long i = (long)blockIdx.x * blockDim.x + tid; i < g.n; i += (long)gridDim.x * blockDim.x) {
                if (A.flags[i] & 2) {
                    // pushes land behind everything the list held; pops only ever take what it held at the start
                    int32_t d = resolve_dic(A.pushval[i], A.fifo, A.cap, head, A.poprank, (uint32_t)g.n);
                    A.fifo[(unsigned long long)((unsigned long long)head + len0 + A.pushrank[i]) % A.cap] = (uint32_t)d;
                }
            }
        }
    }
    CAFE_STAMP(2, 2);
    // the look-back words of k_finish start at zero in the next call
    for (int i = blockIdx.x * blockDim.x + tid; i < 2 * A.nagg; i += gridDim.x * blockDim.x) A.agg[i] = 0;
    // ---- close the call: the workgroup that finishes last (every workgroup has read the scalars by then) ----
    __syncthreads();
    if (tid == 0) {
        // two levels (a thousand workgroups queueing for one word cost ~12 ns each): 32 sub-counters, whoever completes
        // one goes on to the top counter
        const uint32_t sub = blockIdx.x & 31u;
        const uint32_t quota = (gridDim.x - sub + 31u) / 32u;
        int last = 0;
        if (__hip_atomic_fetch_add(&A.ctl->ticket_sub[sub * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == quota - 1) {
            const uint32_t subs = min(32u, gridDim.x);
            last = __hip_atomic_fetch_add(&A.ctl->ticket_commit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == subs - 1;
        }
        s_last = last;
    }
    __syncthreads();
    CAFE_STAMP(2, 3);
    if (!s_last) return;
    const int reason = s_reason;
    Mirror *mr = A.mirror;
    const int32_t real_n0 = st->real_n, num0 = st->num;
    uint32_t cross = 0, pops = 0;
    if (verdict == 1) {
        // in order from the untouched tables (~1.3 us per id); only for calls nobody is waiting for (cafe_insert_dev)
        field_sums(g, sc, fsum);
        __syncthreads();
        if (tid == 0) {
            st->why[reason - 1] += 1;
            st->pad[0] += 1;
            SerialCtx ctx;
            ctx.fifo = A.fifo; ctx.cap = A.cap; ctx.head = st->fifo_head; ctx.len = st->fifo_len;
            ctx.num = st->num; ctx.real_n = st->real_n; ctx.clk = st->clk;
            const ScoreSrc ss = score_src(g, sc, fsum);
            serial_insert_range(A.mt, A.lt, g.n1, g.n2, st->thr, (uint32_t)st->flip, ctx, A.ids, ss, g.n, A.ins, 0, 1e300);
            st->fifo_head = ctx.head; st->fifo_len = ctx.len;
            st->num = ctx.num; st->real_n = ctx.real_n; st->clk = ctx.clk;
            st->fallback = 1;
            cross = (uint32_t)(ctx.real_n - real_n0); pops = (uint32_t)(ctx.num - num0);
            *A.gotcount = 0xffffffffu;
        }
    } else if (tid == 0) {
        if (verdict == 0) {
            // the parallel insert settled: apply its batch totals (pops, pushes, clock ticks, crossings, rounds)
            st->fifo_head = (uint32_t)(((unsigned long long)st->fifo_head + ctl->totals[0]) % A.cap);
            st->fifo_len = st->fifo_len + ctl->totals[1] - ctl->totals[0];
            st->num += (int32_t)ctl->totals[0];
            st->clk += (int32_t)ctl->totals[2];
            st->real_n += (int32_t)ctl->totals[3];
            st->pad[1] += (int32_t)ctl->final_round;
            if ((int32_t)ctl->final_round > st->pad[2]) st->pad[2] = (int32_t)ctl->final_round;
            cross = ctl->totals[3]; pops = ctl->totals[0];
            *A.gotcount = ctl->ngot;
        } else {
            st->why[min(reason, 3) - 1] += reason <= 3 ? 1 : 0;
            st->fallback = 2;
            st->rejected_seq = A.seq;
            *A.gotcount = 0;
        }
        st->rounds = (int32_t)ctl->final_round;
        st->maxpart = (int32_t)ctl->maxpart;
        st->pad[3] = (int32_t)ctl->maxhot;
    }
    if (tid == 0) {
        mr->st = *st;               // pinned host memory: the host reads it without a copy
        mr->n = (uint32_t)g.n; mr->cross = cross; mr->pops = pops;
        mr->rejected = verdict == 2 ? 1u : 0u;
        mr->reason = (uint32_t)reason;
        mr->moves = ctl->nmoves;
        mr->got = ctl->ngot;
        __threadfence_system();
        mr->seq = A.seq;
    }
    CAFE_STAMP(2, 4);
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
static int ilog2_ceil(int v) { int b = 0; while ((1 << b) < v) ++b; return b; }

static void pick_parts(int n, int buckets, int target, int &P, int &bpp) {
    int want = std::max(1, (n + target - 1) / target);
    P = std::min(std::min(want, buckets), kMaxParts);
    bpp = (buckets + P - 1) / P;
    P = (buckets + bpp - 1) / bpp;
}

static int ensure_ws(cafe_sketch *s, int n, Geo &g) {
    if (!s->par) s->par = new ParWorkspace();
    ParWorkspace &w = *s->par;
    memset(&g, 0, sizeof(g));
    g.n = n; g.n1 = s->n1; g.n2 = s->n2;
    pick_parts(n, s->n1, s->opt_target_items, g.P1, g.bpp1);
    pick_parts(n, s->n2, s->opt_target_items, g.P2, g.bpp2);
    // few partitions (a small sketch fed a large batch): each of them sees a large share of the batch's moves
    g.evs = g.P1 <= 128 ? 8 * kCapEv : kCapEv;
    g.evls = g.P2 <= 128 ? 8 * kCapEvL : kCapEvL;
    g.cap_ev = std::min(std::max(s->opt_move_cap, 1), g.evs);
    g.fm1 = make_fastmod(s->n1);
    g.fm2 = make_fastmod(s->n2);
    g.bits1 = ilog2_ceil(g.bpp1);
    g.bits2 = ilog2_ceil(g.bpp2);
    g.nchunks = (n + kChunk - 1) / kChunk;
    // hot list: partitions small enough for the per-wave slot table if the plan's histogram has room for them, else
    // the main table's partitions (sorted by the owner)
    g.hot = s->opt_hot_bypass ? 1 : 0;
    const int ph_table = (s->n1 + kHotKeys / 4 - 1) / (kHotKeys / 4);
    g.hot_table = g.hot && s->opt_hot_table && ph_table + 2 * g.P1 + g.P2 <= kMaxPlanBins;
    g.PH = g.hot_table ? ph_table : g.P1;
    g.bpph = (s->n1 + g.PH - 1) / g.PH;
    g.PH = (s->n1 + g.bpph - 1) / g.bpph;
    g.bitsh = ilog2_ceil(g.bpph);
    g.nbins = 2 * g.P1 + g.P2 + g.PH;
    const size_t N = (size_t)n, PP = (size_t)g.P1 + g.P2;
    const int nblocks = (n + kScanBlock - 1) / kScanBlock;
    g.gflip = (uint32_t)s->flip;
    g.seg = s->seg_len > 0 && (n + s->seg_len - 1) / s->seg_len <= kMaxSeg ? s->seg_len : 0;
    g.verify = s->verify_mode ? 1 : 0;
    g.max_rounds = std::min(std::max(s->opt_max_rounds, 1), kMaxRounds) + 1;
    const size_t mat = (size_t)g.nchunks * g.nbins;
    if (w.ctl.reserve(4 * (kCtlWords + 2 * (size_t)kMaxParts + 16)) || w.cm.reserve(2 * mat + 16) || w.colpref.reserve(4 * mat + 16) ||
        w.tot.reserve(4 * (size_t)g.nbins + 16) || w.off.reserve(4 * ((size_t)g.nbins + 8)) || w.plan.reserve(8 * N + 16) ||
        w.items.reserve(3 * sizeof(Rec) * N + 64) || w.items_sorted.reserve(3 * sizeof(Rec) * N + 64) || w.pre_nmiss.reserve(4 * (size_t)kMaxParts + 16) || w.r1_order.reserve(4 * 3 * (size_t)kMaxParts + 16) || w.hkey.reserve(8 * N + 16) || w.hrun.reserve(4 * N + 16) || w.hfin.reserve(8 * N + 16) ||
        w.ev.reserve(2 * sizeof(MoveEv) * (size_t)g.P1 * g.evs) || w.evres.reserve(sizeof(EvRes) * N + 64) || w.miss.reserve(N + 16) ||
        w.outm.reserve(N + 16) || w.outl.reserve(N + 16) ||
        w.stage_m.reserve(sizeof(MainBucket) * (N + (size_t)g.P1 * g.evs)) || w.stage_l.reserve(sizeof(LruBucket) * N + 128) ||
        w.stage_tot.reserve(4 * PP + 4) || w.prevcnt.reserve(4 * (size_t)g.P1 + 4) || w.ldirty.reserve(2 * (size_t)g.P2 + 4) ||
        w.myev.reserve(sizeof(MoveEv) * (size_t)g.P2 * g.evls) || w.mycnt.reserve(4 * (size_t)g.P2 + 4) || w.flags.reserve(5 * N + 64) ||
        w.agg.reserve(16 * (size_t)nblocks + 16) || w.ranks.reserve(12 * N + 16) || w.gotlist.reserve(4 * N + 16))
        return 1;
    if (w.evres.p != w.evres_seen || w.evres.cap != w.evres_seen_cap) {      // tags of a fresh allocation must not match by accident
        CAFE_HIP_CHECK(hipMemset(w.evres.p, 0, w.evres.cap));
        CAFE_HIP_CHECK(hipDeviceSynchronize());
        w.evres_seen = w.evres.p;
        w.evres_seen_cap = w.evres.cap;
    }
    return 0;
}

int par_prepare(cafe_sketch *s, int n) {
    if (n > kParMaxIds) return 0;          // such a batch goes to the in-order kernel
    Geo g;
    return ensure_ws(s, n, g);
}

const uint32_t *par_gotcount(cafe_sketch *s) { return s->par ? s->par->ctl.as<uint32_t>() + kCtlWords + 2 * (size_t)kMaxParts + 4 : nullptr; }
const uint32_t *par_gotlist(cafe_sketch *s) { return s->par ? s->par->gotlist.as<uint32_t>() : nullptr; }

// Owner capacity: partitions are sized for ~opt_target_items ids; the largest one of the previous call (it travels
// with the snapshot in pinned memory) picks the instantiation.  A wrong guess is never a wrong result: a partition that
// does not fit flags the batch, which is then issued again with the largest owners (verify) or walked in order.
static int pick_cap(cafe_sketch *s, int n) {
    if (s->force_cap_once) return s->force_cap_once;
    if (s->force_cap) return s->force_cap;
    const Mirror *m = reinterpret_cast<const Mirror *>(s->h_pinned);
    // a batch size far from the one the snapshot's partition sizes belong to (or no snapshot yet): the largest owners,
    // which take any partition up to 8192 items; sizes that wander by a few percent (a shard's share of a step) keep
    // using the snapshot, whose margin below covers them
    const long ref_n = s->last_seen_batch;
    if (m->seq == 0 || (long)n * 4 > ref_n * 5 || (long)n * 4 < ref_n * 3) {
        s->last_seen_batch = n;
        for (int i = 0; i < 32; ++i) s->cap_hist[i] = 0;
        s->cap_hist_at = 0;
        s->cap_regime_seq = s->call_seq;         // snapshots of earlier calls belong to the previous regime
        return 8192;
    }
    // the largest partition over the last 32 snapshots of this regime, not just the last one: callers that alternate
    // between differently distributed streams (one call per field) would otherwise size the owners for the previous
    // field, and a batch that does not fit goes to the in-order kernel (~1.3 us per id)
    const long big_now = std::max(m->st.maxpart, s->opt_hot_table ? 0 : m->st.pad[3]);
    if (m->seq != s->cap_seen_seq && (int32_t)(m->seq - s->cap_regime_seq) >= 0) {
        s->cap_seen_seq = m->seq;
        s->cap_hist[s->cap_hist_at++ & 31] = (int)big_now;
    }
    if (s->cap_hist_at < 4) return 8192;        // too few snapshots of this regime yet (they trail the calls)
    long big = 0;
    for (int i = 0; i < 32; ++i) big = std::max(big, (long)s->cap_hist[i]);
    if (m->st.fallback) return 8192;
    if (big + big / 8 + 32 <= kCapSmall) return kCapSmall;
    if (big + big / 8 + 32 <= 2048) return 2048;
    // (two owners per compute unit instead of one: cfg2 -22 % step time; a narrower margin, since a misfit only costs
    // the reissue of that step with the largest owners)
    if (big + big / 8 + 32 <= 4608) return 4608;
    return 8192;
}

// The plan of an insert of n ids: classification, partition offsets, item records; per-call scalars and a zeroed
// control block.  Reads the ids and the pre-batch tables only.
// `done`: an event completed by the last plan kernel itself (its completion signal; no marker packet behind it)
static int pick_cap(cafe_sketch *s, int n);
static int launch_plan(cafe_sketch *s, Geo &g, const IdSrc &ids, hipStream_t pre, hipEvent_t done = nullptr) {
    ParWorkspace &w = *s->par;
    const int n = g.n;
    Ctl *ctl = w.ctl.as<Ctl>();
    uint32_t *evcnt = w.ctl.as<uint32_t>() + kCtlWords;
    uint32_t *off = w.off.as<uint32_t>();
    uint32_t *planA = w.plan.as<uint32_t>(), *planB = planA + n;
    Rec *rec1 = w.items.as<Rec>(), *rec2 = rec1 + n, *rech = rec2 + n;
    const size_t plan_lds = 2 * (size_t)kPlanWaves * g.nbins + 16;
    static size_t plan_lds_set = 48 * 1024;
    if (plan_lds > plan_lds_set) {
        CAFE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_assign), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan_lds));
        plan_lds_set = plan_lds;
    }
    hipLaunchKernelGGL(k_assign, dim3(g.nchunks), dim3(kPlanThreads), plan_lds, pre, g, s->d_main, ids, w.cm.as<uint16_t>(), planA, planB);
    hipLaunchKernelGGL(k_offsets, dim3((g.nbins + kOffBins - 1) / kOffBins), dim3(kThreads), 0, pre, g, w.cm.as<uint16_t>(),
                       w.colpref.as<uint32_t>(), w.tot.as<uint32_t>(), off, ctl, evcnt, s->d_state, w.r1_order.as<uint32_t>());
    // (the plan's completion event rides on its last kernel: no marker packet)
    // The owner-order pass pays where the insert's chain is the step's critical path.  Where the update runs split (its
    // gradient pass sits beside the chain and the row updates wait for IT), a shorter chain buys nothing and one more
    // kernel beside the gather costs: measured on MI355X, cfg2 (D = 16) -5.5 % step time with the pass, cfg3 (D = 128,
    // split update) +2 %.  The caller of the fused step tells which it is (split_active).
    const bool presort = s->opt_presort && !s->split_active;
    s->plan_presort = presort ? 1 : 0;
    hipExtLaunchKernelGGL(k_scatter, dim3((n + 255) / 256), dim3(256), 0, pre, nullptr, presort ? nullptr : done, 0, g, ids,
                          (const uint32_t *)planA, (const uint32_t *)planB, (const uint32_t *)w.colpref.as<uint32_t>(),
                          (const uint32_t *)off, rec1, rec2, rech, w.miss.as<uint8_t>());
    if (!presort) { CAFE_HIP_CHECK(hipGetLastError()); return 0; }     // the owners sort for themselves
    // the partitions in owner order (the owners of this capacity; a partition that does not fit flags the batch, which is
    // then issued again with the largest owners like any other workspace limit)
    Rec *rec1s = w.items_sorted.as<Rec>(), *rec2s = rec1s + n, *rechs = rec2s + n;
    const int grid1 = ((g.hot && !g.hot_table) ? g.PH : 0) + g.P2 + g.P1;
    const int cap = pick_cap(s, n);

#define CAFE_PRESORT(C)                                                                                                         \
    hipExtLaunchKernelGGL((k_presort<C>), dim3(grid1), dim3(kThreads), 0, pre, nullptr, done, 0, g, ctl, (const uint32_t *)off,     \
                          (const Rec *)rec1, (const Rec *)rec2, (const Rec *)rech, rec1s, rec2s, rechs, w.pre_nmiss.as<uint32_t>())
    if (cap == kCapSmall) CAFE_PRESORT(kCapSmall);
    else if (cap == 2048) CAFE_PRESORT(2048);
    else if (cap == 4608) CAFE_PRESORT(4608);
    else CAFE_PRESORT(8192);
#undef CAFE_PRESORT
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

// cafe_forward_dev: the plan of the insert that will follow this forward, on the handle's second stream beside the
// gather kernel.  It stands if the next insert is over the same ids with no call, flip or reset in between.
int par_plan_ahead(cafe_sketch *s, const int64_t *d_local_ids, const int64_t *d_id_offset, int batch, int n, hipStream_t side,
                   const int32_t *d_gids) {
    Geo g;
    if (ensure_ws(s, n, g)) return 1;
    g.flip_now = 0;
    IdSrc ids;
    ids.gids = d_local_ids ? nullptr : d_gids; ids.local_ids = d_local_ids; ids.id_offset = d_id_offset; ids.batch = batch;
    if (launch_plan(s, g, ids, side, s->ev_plan)) return 1;
    s->plan_valid = 1; s->plan_seq = s->call_seq; s->plan_n = n; s->plan_flip = s->flip;
    s->plan_ids = d_local_ids ? (const void *)d_local_ids : (const void *)d_gids;
    s->plan_hot = s->opt_hot_bypass; s->plan_target = s->opt_target_items;
    return 0;
}

int par_insert(cafe_sketch *s, const int32_t *d_ids, const float *d_vals, const float *d_psum, int nparts, int n,
               int32_t *d_ins, float *d_scores_out, int flip_now, hipStream_t st, hipStream_t pre) {
    Geo g;
    if (ensure_ws(s, n, g)) return 1;
    g.flip_now = flip_now;
    g.nparts = nparts;
    g.sseg = d_psum ? s->score_seg : 0;
    if (d_psum && (g.sseg <= 0 || (n + g.sseg - 1) / g.sseg > kMaxSeg)) { set_error("internal: partial sums without segments"); return 1; }
    ParWorkspace &w = *s->par;
    w.call_seq++;
    const unsigned long long tag = w.call_seq;
    Ctl *ctl = w.ctl.as<Ctl>();
    uint32_t *evcnt = w.ctl.as<uint32_t>() + kCtlWords;
    uint32_t *off = w.off.as<uint32_t>();
    Rec *rec1 = w.items.as<Rec>(), *rec2 = rec1 + n, *rech = rec2 + n;
    Score sc;
    sc.vals = d_vals; sc.psum = d_psum; sc.vstride = s->score_stride > 0 ? s->score_stride : 1;

    // ---- plan: the one made beside the forward, if it is still the plan of this call ----
    const bool ahead = s->plan_valid && s->plan_seq + 1 == s->call_seq && s->plan_n == n && s->plan_ids == s->plan_expect && !flip_now &&
                       s->plan_flip == s->flip && s->plan_hot == s->opt_hot_bypass && s->plan_target == s->opt_target_items &&
                       !s->plan_stale;
    s->plan_valid = 0;
    s->plan_stale = 0;
    if (ahead) {
        // (a plan that has already completed -- it was started well ahead -- needs no wait packet on st)
        if (hipEventQuery(s->ev_plan) != hipSuccess) CAFE_HIP_CHECK(hipStreamWaitEvent(st, s->ev_plan, 0));
        s->counters_plan_ahead++;
    } else {
        IdSrc ids;
        ids.gids = d_ids; ids.local_ids = nullptr; ids.id_offset = nullptr; ids.batch = 1;
        if (launch_plan(s, g, ids, pre)) return 1;
        if (pre != st) {         // the scores come in here
            CAFE_HIP_CHECK(hipEventRecord(s->ev_head_join, pre));
            CAFE_HIP_CHECK(hipStreamWaitEvent(st, s->ev_head_join, 0));
        }
    }
    // ---- rounds ----
    MainArgs MA;
    MA.mt = s->d_main; MA.st = s->d_state; MA.ctl = ctl; MA.off = off; MA.rec1 = rec1; MA.evcnt = evcnt;
    MA.ev = w.ev.as<MoveEv>(); MA.evres = w.evres.as<EvRes>(); MA.miss = w.miss.as<uint8_t>(); MA.outm = w.outm.as<uint8_t>();
    MA.stage_m = w.stage_m.as<MainBucket>(); MA.stage_tot = w.stage_tot.as<uint32_t>(); MA.prevcnt = w.prevcnt.as<uint32_t>();
    MA.ldirty = w.ldirty.as<uint8_t>(); MA.hkey = w.hkey.as<unsigned long long>(); MA.hrun = w.hrun.as<float>(); MA.call_tag = tag;
    LruArgs LA;
    LA.lt = s->d_lru; LA.st = s->d_state; LA.ctl = ctl; LA.off = off; LA.rec2 = rec2; LA.miss = w.miss.as<uint8_t>(); LA.evcnt = evcnt;
    LA.ev = w.ev.as<MoveEv>(); LA.outl = w.outl.as<uint8_t>(); LA.stage_l = w.stage_l.as<LruBucket>();
    LA.stage_tot = w.stage_tot.as<uint32_t>(); LA.ldirty = w.ldirty.as<uint8_t>(); LA.myev = w.myev.as<MoveEv>();
    LA.mycnt = w.mycnt.as<uint32_t>();
    HotArgs HA;
    const bool sorted = s->plan_presort != 0;          // as the plan in hand was made (launch_plan)
    HA.rechs = (g.hot_table || !sorted) ? nullptr : (const Rec *)(w.items_sorted.as<Rec>() + 2 * (size_t)n);
    MA.rec1s = sorted ? w.items_sorted.as<Rec>() : nullptr; LA.rec2s = sorted ? w.items_sorted.as<Rec>() + n : nullptr;
    LA.pre_nmiss = w.pre_nmiss.as<uint32_t>();
    HA.rech = rech; HA.hkey = w.hkey.as<unsigned long long>(); HA.hrun = w.hrun.as<float>(); HA.hfin = w.hfin.as<int2>();
    const int cap = pick_cap(s, n);
    // (with the small owners every partition's workgroup is resident at once: grid order, no indirection)
    HA.order = (cap > 2048 && !(g.hot && g.hot_table) && s->opt_r1_order) ? w.r1_order.as<uint32_t>() : nullptr;
    if (s->ev_chain[0]) (void)hipEventRecord(s->ev_chain[0], st);
    const int grid1 = (!g.hot ? 0 : g.hot_table ? (g.PH + kWaves - 1) / kWaves : g.PH) + g.P2 + g.P1;
    // The fix-point kernel is persistent (grid barrier between half-rounds): a quiet stream has a few tens of moves per
    // batch and 64 workgroups suffice; when ids churn (thousands of moves, most partitions in play every round) the
    // partitions are spread over more workgroups.  All of them are resident: at most two per compute unit of the <2048>
    // form (40 KB LDS), one of the <8192> form.
    const uint32_t mv = s->last_moves;
    const int fix_grid = mv <= 256 ? kFixGrid : cap > 2048 ? 256 : 512;
#define CAFE_R1(C) hipLaunchKernelGGL((k_round1<C>), dim3(grid1), dim3(kThreads), 0, st, g, MA, LA, HA, sc)
    if (cap == kCapSmall) CAFE_R1(kCapSmall);
    else if (cap == 2048) CAFE_R1(2048);
    else if (cap == 4608) CAFE_R1(4608);
    else CAFE_R1(8192);
#undef CAFE_R1
    if (cap <= 2048) hipLaunchKernelGGL((k_fix<2048>), dim3(fix_grid), dim3(kThreads), 0, st, g, MA, LA, sc);
    else hipLaunchKernelGGL((k_fix<8192>), dim3(fix_grid), dim3(kThreads), 0, st, g, MA, LA, sc);
    // ---- finish ----
    const int nb = (n + kScanBlock - 1) / kScanBlock;
    uint8_t *flags = w.flags.as<uint8_t>();
    int32_t *pushval = (int32_t *)(flags + (((size_t)n + 15) & ~(size_t)15));
    uint32_t *poprank = w.ranks.as<uint32_t>(), *pushrank = poprank + n, *clkrank = pushrank + n;
    hipLaunchKernelGGL(k_finish, dim3(nb), dim3(kThreads), 0, st, g, tag, ctl, w.miss.as<uint8_t>(), w.outm.as<uint8_t>(),
                       w.outl.as<uint8_t>(), w.evres.as<EvRes>(), d_ins, flags, pushval, w.agg.as<unsigned long long>(), poprank,
                       pushrank, clkrank, w.gotlist.as<uint32_t>(), sc, d_scores_out);
    CommitArgs CA;
    CA.mt = s->d_main; CA.lt = s->d_lru; CA.fifo = s->d_fifo; CA.cap = s->fifo_cap; CA.st = s->d_state; CA.ctl = ctl; CA.off = off;
    CA.stage_tot = w.stage_tot.as<uint32_t>(); CA.stage_m = w.stage_m.as<MainBucket>(); CA.stage_l = w.stage_l.as<LruBucket>();
    CA.flags = flags; CA.pushval = pushval; CA.poprank = poprank; CA.pushrank = pushrank; CA.clkrank = clkrank;
    CA.hfin = w.hfin.as<int2>(); CA.agg = w.agg.as<unsigned long long>(); CA.nagg = nb; CA.ids = d_ids; CA.ins = d_ins;
    CA.mirror = reinterpret_cast<Mirror *>(s->d_pinned); CA.seq = s->call_seq;
    CA.reset_at = s->adjust ? s->lim * 1.2 : 1e300;
    CA.gotcount = w.ctl.as<uint32_t>() + kCtlWords + 2 * (size_t)kMaxParts + 4;
    // migration inside the commit (cafe_backward_update_dev names the tables); only for calls the host verifies -- a
    // batch walked in order by the closing workgroup has no staged placeholders to migrate from
    const bool mig = s->mig.w_hot != nullptr && g.verify;
    CA.w_hot = mig ? s->mig.w_hot : nullptr; CA.w_cold = s->mig.w_cold; CA.hash_size = s->mig.hash_size;
    CA.mig_gids = s->mig.gids; CA.mig_fields = s->mig.fields; CA.id_offset = s->mig.id_offset;
    CA.local_ids = s->mig.local_ids; CA.batch = std::max(1, s->mig.batch); CA.dim = s->mig.dim; CA.hotn = std::max(1, s->mig.hotn);
    s->mig_done = mig ? 1 : 0;
    const int cb = g.P1 + g.P2 + (g.hot ? g.PH : 0);
    // (the kernel's own completion signal is the event a plan made ahead for the NEXT call waits for: no marker packet)
    hipExtLaunchKernelGGL(k_commit, dim3(cb), dim3(256), 0, st, nullptr, s->ev_commit, 0, g, CA, sc);
    if (s->ev_chain[1]) (void)hipEventRecord(s->ev_chain[1], st);
    CAFE_HIP_CHECK(hipGetLastError());
    s->counters[1]++;
    return 0;
}

void par_free(cafe_sketch *s) {
    if (!s->par) return;
    ParWorkspace &w = *s->par;
    DevBuf *all[] = {&w.ctl, &w.cm, &w.colpref, &w.tot, &w.off, &w.plan, &w.items, &w.ev, &w.evres, &w.miss, &w.outm, &w.outl,
                     &w.stage_m, &w.stage_l, &w.stage_tot, &w.prevcnt, &w.ldirty, &w.myev, &w.mycnt, &w.flags, &w.agg, &w.ranks,
                     &w.gotlist, &w.hkey, &w.hrun, &w.hfin, &w.items_sorted, &w.pre_nmiss, &w.r1_order};
    for (DevBuf *b : all) b->release();
    delete s->par;
    s->par = nullptr;
}

}  // namespace cafe
