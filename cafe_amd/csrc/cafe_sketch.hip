// cafe_sketch.hip -- HotSketch on MI355X: handle management, batch query, the in-order insert kernel,
// threshold reset, state import/export and the reference-compatible C ABI (include/cafe_hip.h).
//
// The parallel insert lives in cafe_insert_par.hip; this file holds everything else the sketch needs.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "cafe_internal.h"

using namespace cafe;

// ------------------------------------------------------------------------------------------------
// errors, buffers
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

void cafe::set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

extern "C" const char *cafe_last_error(void) { return g_err.c_str(); }
extern "C" int cafe_version(void) { return 1; }

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = (bytes + (bytes >> 2) + 4095) & ~(size_t)4095;
    CAFE_HIP_CHECK(hipMalloc(&p, want));
    // Workspaces start zeroed, and the zeros are in place before this returns: the kernels that use the buffer may run
    // on any stream (the handles own a second one), a memset left pending on the null stream would race with them.
    CAFE_HIP_CHECK(hipMemset(p, 0, want));
    CAFE_HIP_CHECK(hipDeviceSynchronize());
    cap = want;
    return 0;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

// ------------------------------------------------------------------------------------------------
// kernels: init, query
// ------------------------------------------------------------------------------------------------
__global__ void k_fill_fifo(uint32_t *fifo, int lim) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < lim - 1) fifo[i] = (uint32_t)(i + 1);      // sketch.cpp:55-56: slots 1..lim-1, slot 0 never used
}

__global__ void k_init_state(DevState *st, float thr, int lim) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        DevState z;
        memset(&z, 0, sizeof(z));
        z.thr = thr;
        z.fifo_head = 0;
        z.fifo_len = lim > 1 ? (uint32_t)(lim - 1) : 0u;
        *st = z;
    }
}

// sketch.cpp:260-265 / 118-128.  One lane per id; the probe is one aligned 32-B read (val|dic).
__global__ void k_query(const MainBucket *__restrict__ mt, int n1, const int32_t *__restrict__ ids, int n,
                        int32_t *__restrict__ que) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)ids[i];
    const int4 *p = reinterpret_cast<const int4 *>(&mt[hash_main(x, n1)]);
    int4 v = p[0], d = p[1];
    int32_t val[4] = {v.x, v.y, v.z, v.w}, dic[4] = {d.x, d.y, d.z, d.w};
    que[i] = query_bucket(val, dic, x);
}

// batch_query_median of tricks/sketch_ml_embed.py:82-84 (defined in cafe_core.h::median_bucket)
__global__ void k_query_median(const MainBucket *__restrict__ mt, int n1, const DevState *__restrict__ st, int gflip,
                               const int32_t *__restrict__ ids, int n, int32_t *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)ids[i];
    const MainBucket &b = mt[hash_main(x, n1)];
    out[i] = median_bucket(b.val, b.dic, b.cnt, b.flip, gflip, st->thr, x);
}

// sketch.cpp:321-327 / 142-150: present in the main table at all (hot or not)
__global__ void k_count(const MainBucket *__restrict__ mt, int n1, const int32_t *__restrict__ ids, int n,
                        int mode, unsigned long long *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int hit = 0;
    if (i < n) {
        uint32_t x = (uint32_t)ids[i];
        const MainBucket &b = mt[hash_main(x, n1)];
        if (mode == 0) {
            hit = query_bucket(b.val, b.dic, x) < 0;
        } else {
            for (int s = 0; s < kSlots; ++s)
                if ((uint32_t)b.val[s] == x) { hit = 1; break; }
        }
    }
    unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
}

// ------------------------------------------------------------------------------------------------
// per-call prologue (sketch.cpp:282-289) for the calls that do not go through the parallel insert (which
// applies it in its own k_offsets): the flip decision is made on the host (it depends on the call count
// only); the threshold lives on the device because reset() rewrites it there.
// ------------------------------------------------------------------------------------------------
// The LRU clock (sketch.cpp:17 `t`) only ever grows, one tick per main-table miss, and the reference's victim search
// starts from the constant 1e9 (sketch.cpp:159): past 1e9 ticks it stops finding victims, past 2^31 the clock wraps.
// Only the order of the four timestamps inside a bucket matters, so long before that the host has every bucket's
// timestamps rewritten to their ranks 1..4 (0 = never used stays 0) and the clock set to 4.  A deviation from the
// reference's absolute timestamp values where the reference itself is about to become undefined; every decision
// (victims, ins[], que[]) is unchanged.
constexpr int32_t kClockRenormAt = 1 << 29;
__global__ void k_clock_renorm(LruBucket *lt, int n2, DevState *st) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n2) {
        int32_t t[kSlots], r[kSlots];
        for (int i = 0; i < kSlots; ++i) t[i] = lt[k].t[i];
        for (int i = 0; i < kSlots; ++i) {
            int below = 0;
            for (int j = 0; j < kSlots; ++j) below += (t[j] != 0 && t[j] < t[i]) ? 1 : 0;
            r[i] = t[i] != 0 ? below + 1 : 0;
        }
        for (int i = 0; i < kSlots; ++i) lt[k].t[i] = r[i];
    }
    if (k == 0) st->clk = kSlots;
}

__global__ void k_prologue(DevState *st, int flip_now, int gflip) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (flip_now) st->thr = (float)((double)st->thr / kDecayV);
        st->flip = gflip;
        st->fallback = 0;
        st->rounds = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// in-order insert: one lane walks the batch exactly like the reference's loop (sketch.cpp:292-294;
// cafe_core.h::serial_insert_range).  It is the definition of correctness on the device, the path for
// option "serial_insert" and for batches beyond the parallel insert's size limit, and a test aid.  (The
// parallel insert runs the same loop inside its closing kernel for a batch it could not settle.)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_insert_serial(MainBucket *mt, LruBucket *lt, uint32_t *fifo, uint32_t cap,
                                                      DevState *st, int n1, int n2, const int32_t *__restrict__ ids,
                                                      const float *__restrict__ scores, int n, int32_t *__restrict__ ins,
                                                      Mirror *host_mirror, uint32_t seq, double reset_at) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    // sketch.cpp:291 is evaluated by the host from an upper bound of real_n (insert_dev_on).  Checked here against the
    // real value: a call that should have begun with a reset is counted, never silently accepted.
    const int32_t real_n0 = st->real_n, num0 = st->num;
    if ((double)real_n0 > reset_at) st->cadence_miss += 1;
    st->pad[0] += 1;          // cumulative: batches walked in order
    // (Measured: letting other lanes pull the buckets of the next 64 ids into the cache does not shorten the walk --
    // 1.4 us per id either way: the two dependent loads per id cost an L2 round trip each.)
    SerialCtx ctx;
    ctx.fifo = fifo; ctx.cap = cap; ctx.head = st->fifo_head; ctx.len = st->fifo_len;
    ctx.num = st->num; ctx.real_n = st->real_n; ctx.clk = st->clk;
    ScoreSrc sc;
    sc.vals = scores; sc.fsum = nullptr; sc.seg = 0;
    serial_insert_range(mt, lt, n1, n2, st->thr, (uint32_t)st->flip, ctx, ids, sc, n, ins, 0, reset_at);
    st->fifo_head = ctx.head; st->fifo_len = ctx.len;
    st->num = ctx.num; st->real_n = ctx.real_n; st->clk = ctx.clk;
    st->fallback = 1;
    host_mirror->st = *st;
    host_mirror->n = (uint32_t)n; host_mirror->cross = (uint32_t)(ctx.real_n - real_n0);
    host_mirror->pops = (uint32_t)(ctx.num - num0);
    host_mirror->rejected = 0; host_mirror->reason = 0; host_mirror->moves = 0; host_mirror->got = 0;
    __threadfence_system();
    host_mirror->seq = seq;
}

int cafe::launch_serial_insert(cafe_sketch *s, const int32_t *d_ids, const float *d_scores, int n, int32_t *d_ins,
                               hipStream_t st) {
    const double reset_at = s->adjust ? s->lim * 1.2 : 1e300;
    hipLaunchKernelGGL(k_insert_serial, dim3(1), dim3(64), 0, st, s->d_main, s->d_lru, s->d_fifo, s->fifo_cap,
                       s->d_state, s->n1, s->n2, d_ids, d_scores, n, d_ins, reinterpret_cast<Mirror *>(s->d_pinned),
                       s->call_seq, reset_at);
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// threshold reset (sketch.cpp:76-116).  Rare (real_n > 1.2 lim), so it is built from a collect pass,
// a library radix sort of 64-bit (score, address-order) keys and prefix sums that reproduce the
// reference's in-order pushes and pops on the free list.
// ------------------------------------------------------------------------------------------------
__device__ inline uint32_t orderable(float f) {
    if (f == 0.0f) f = 0.0f;                  // -0 and +0 compare equal in the reference's sort
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float from_orderable(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ void k_reset_collect(MainBucket *mt, LruBucket *lt, int n1, int n2, DevState *st, int lru_first,
                                unsigned long long *keys, unsigned int *count) {
    const float thr = st->thr;
    const uint32_t gflip = (uint32_t)st->flip;
    const long total = (long)n1 + n2;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        if (g < n1) {
            MainR b;
            load_main(b, &mt[g]);
            if (b.flip != gflip) { sync_flip(b, gflip); store_main(&mt[g], b); }
            uint32_t base = (lru_first ? (uint32_t)n2 * kSlots : 0u) + (uint32_t)g * kSlots;
            const float cn[4] = {b.c0, b.c1, b.c2, b.c3};
            const int32_t dc[4] = {b.d0, b.d1, b.d2, b.d3};
#pragma unroll
            for (int i = 0; i < kSlots; ++i)
                if (cn[i] >= thr || dc[i]) {
                    unsigned int at = atomicAdd(count, 1u);
                    keys[at] = ((unsigned long long)orderable(cn[i]) << 32) | (base + i);
                }
        } else {
            long k = g - n1;
            LruR b;
            load_lru(b, &lt[k]);
            if (b.flip != gflip) { sync_flip(b, gflip); store_lru(&lt[k], b); }
            uint32_t base = (lru_first ? 0u : (uint32_t)n1 * kSlots) + (uint32_t)k * kSlots;
            const float cn[4] = {b.c0, b.c1, b.c2, b.c3};
            const int32_t dc[4] = {b.d0, b.d1, b.d2, b.d3};
#pragma unroll
            for (int i = 0; i < kSlots; ++i)
                if (cn[i] >= thr || dc[i]) {
                    unsigned int at = atomicAdd(count, 1u);
                    keys[at] = ((unsigned long long)orderable(cn[i]) << 32) | (base + i);
                }
        }
    }
}

__device__ inline int32_t *slot_dic(MainBucket *mt, LruBucket *lt, int n1, int n2, int lru_first, uint32_t order) {
    uint32_t nl = (uint32_t)n2 * kSlots, nm = (uint32_t)n1 * kSlots;
    bool is_lru = lru_first ? (order < nl) : (order >= nm);
    uint32_t idx = lru_first ? (is_lru ? order : order - nl) : (is_lru ? order - nm : order);
    return is_lru ? &lt[idx / kSlots].dic[idx % kSlots] : &mt[idx / kSlots].dic[idx % kSlots];
}

__global__ void k_reset_flags(MainBucket *mt, LruBucket *lt, int n1, int n2, int lru_first,
                              const unsigned long long *keys, long l, long cut, int *pushf, int *popf) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= l) return;
    int32_t d = *slot_dic(mt, lt, n1, n2, lru_first, (uint32_t)keys[i]);
    pushf[i] = (i <= cut && d != 0) ? 1 : 0;
    popf[i] = (i > cut && d == 0) ? 1 : 0;
}

__global__ void k_reset_push(MainBucket *mt, LruBucket *lt, int n1, int n2, int lru_first,
                             const unsigned long long *keys, long l, const int *pushf, const int *pushr,
                             uint32_t *fifo, uint32_t cap, const DevState *st) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= l || !pushf[i]) return;
    int32_t *d = slot_dic(mt, lt, n1, n2, lru_first, (uint32_t)keys[i]);
    unsigned long long pos = (unsigned long long)st->fifo_head + st->fifo_len + (unsigned)pushr[i];
    fifo[pos % cap] = (uint32_t)*d;
    *d = 0;
}

__global__ void k_reset_pop(MainBucket *mt, LruBucket *lt, int n1, int n2, int lru_first,
                            const unsigned long long *keys, long l, const int *popf, const int *popr,
                            const int *pushf, const int *pushr, const uint32_t *fifo, uint32_t cap, const DevState *st) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= l || !popf[i]) return;
    unsigned long long avail = (unsigned long long)st->fifo_len + (unsigned)(pushr[l - 1] + pushf[l - 1]);
    unsigned long long k = (unsigned)popr[i];
    if (k >= avail) return;                    // free list ran dry (sketch.cpp:107)
    *slot_dic(mt, lt, n1, n2, lru_first, (uint32_t)keys[i]) = (int32_t)fifo[(st->fifo_head + k) % cap];
}

__global__ void k_reset_final(DevState *st, const unsigned long long *keys, long l, long cut, const int *popf,
                              const int *popr, const int *pushf, const int *pushr, uint32_t cap, int lim) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long pushes = 0, pops = 0;
    if (l > 0) {
        pushes = (unsigned)(pushr[l - 1] + pushf[l - 1]);
        unsigned long long want = (unsigned)(popr[l - 1] + popf[l - 1]);
        unsigned long long avail = (unsigned long long)st->fifo_len + pushes;
        pops = want < avail ? want : avail;
    }
    st->fifo_head = (uint32_t)(((unsigned long long)st->fifo_head + pops) % cap);
    st->fifo_len = (uint32_t)((unsigned long long)st->fifo_len + pushes - pops);
    if (cut >= 0 && cut < l) st->thr = from_orderable((uint32_t)(keys[cut] >> 32));
    st->real_n = lim;
    st->resets += 1;
}

static int run_reset(cafe_sketch *s, hipStream_t st) {
    size_t maxc = ((size_t)s->n1 + s->n2) * kSlots;
    if (s->ws_reset_keys.reserve(maxc * 8 + 8)) return 1;
    if (s->ws_reset_keys2.reserve(maxc * 8 + 8)) return 1;
    if (s->ws_reset_flags.reserve(maxc * 16 + 64)) return 1;
    unsigned int *d_count = (unsigned int *)s->ws_misc.as<char>();
    CAFE_HIP_CHECK(hipMemsetAsync(d_count, 0, 4, st));
    unsigned long long *keys = s->ws_reset_keys.as<unsigned long long>();
    unsigned long long *keys2 = s->ws_reset_keys2.as<unsigned long long>();
    long total = (long)s->n1 + s->n2;
    int blocks = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(k_reset_collect, dim3(blocks), dim3(256), 0, st, s->d_main, s->d_lru, s->n1, s->n2, s->d_state,
                       s->lru_first, keys, d_count);
    unsigned int l_u = 0;
    CAFE_HIP_CHECK(hipMemcpyAsync(&l_u, d_count, 4, hipMemcpyDeviceToHost, st));
    CAFE_HIP_CHECK(hipStreamSynchronize(st));
    long l = (long)l_u, cut = l - s->lim;
    int *pushf = s->ws_reset_flags.as<int>(), *popf = pushf + maxc, *pushr = popf + maxc, *popr = pushr + maxc;
    const unsigned long long *sorted = keys;
    if (l > 0) {
        size_t tmp = 0;
        CAFE_HIP_CHECK(rocprim::radix_sort_keys(nullptr, tmp, keys, keys2, (size_t)l, 0, 64, st));
        size_t tmp2 = 0, tmp3 = 0;
        CAFE_HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp2, pushf, pushr, 0, (size_t)l, rocprim::plus<int>(), st));
        tmp3 = std::max(tmp, tmp2);
        if (s->ws_reset_tmp.reserve(tmp3 + 256)) return 1;
        CAFE_HIP_CHECK(rocprim::radix_sort_keys(s->ws_reset_tmp.p, tmp, keys, keys2, (size_t)l, 0, 64, st));
        sorted = keys2;
        int gb = (int)((l + 255) / 256);
        hipLaunchKernelGGL(k_reset_flags, dim3(gb), dim3(256), 0, st, s->d_main, s->d_lru, s->n1, s->n2, s->lru_first,
                           sorted, l, cut, pushf, popf);
        CAFE_HIP_CHECK(rocprim::exclusive_scan(s->ws_reset_tmp.p, tmp2, pushf, pushr, 0, (size_t)l, rocprim::plus<int>(), st));
        CAFE_HIP_CHECK(rocprim::exclusive_scan(s->ws_reset_tmp.p, tmp2, popf, popr, 0, (size_t)l, rocprim::plus<int>(), st));
        hipLaunchKernelGGL(k_reset_push, dim3(gb), dim3(256), 0, st, s->d_main, s->d_lru, s->n1, s->n2, s->lru_first,
                           sorted, l, pushf, pushr, s->d_fifo, s->fifo_cap, s->d_state);
        hipLaunchKernelGGL(k_reset_pop, dim3(gb), dim3(256), 0, st, s->d_main, s->d_lru, s->n1, s->n2, s->lru_first,
                           sorted, l, popf, popr, pushf, pushr, s->d_fifo, s->fifo_cap, s->d_state);
    }
    hipLaunchKernelGGL(k_reset_final, dim3(1), dim3(64), 0, st, s->d_state, sorted, l, cut, popf, popr, pushf, pushr,
                       s->fifo_cap, s->lim);
    CAFE_HIP_CHECK(hipGetLastError());
    s->counters[4]++;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
// The handle's own streams, created by the first step that is large enough to use them (>= 32768 lookups): a sketch
// that only ever sees small calls holds no hardware queues (a process with dozens of handles would otherwise keep far
// more queues than the hardware schedules at once).
int cafe::ensure_streams(cafe_sketch *s) {
    if (s->side_stream) return 0;
    // the second stream carries the short plan kernels beside the bandwidth kernels: highest priority, so that their
    // workgroups are placed as soon as compute-unit slots free up
    int prio_least = 0, prio_most = 0;
    CAFE_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_most));
    CAFE_HIP_CHECK(hipStreamCreateWithPriority(&s->side_stream, hipStreamNonBlocking, prio_most));
    // The third one carries a bandwidth kernel beside the insert's short kernels.  Default priority, not the lowest: the
    // caller's stream waits for this one (the row updates wait for the segment sums), and in a process that holds more
    // queues than the hardware schedules at once a lowest-priority queue was never scheduled while the waiting queue
    // counted as busy (a full test session hung at the first step of its 29th test).
    (void)prio_least;
    CAFE_HIP_CHECK(hipStreamCreateWithPriority(&s->low_stream, hipStreamNonBlocking, 0));
    return 0;
}

extern "C" int cafe_sketch_create(cafe_sketch_t **out, int lim, int threshold, int adjust_thres, double alpha,
                                  int lru_first) {
    if (!out || lim < 10) { set_error("cafe_sketch_create: lim must be >= 10 (got %d)", lim); return 1; }
    cafe_sketch *s = new cafe_sketch();
    CAFE_HIP_CHECK(hipGetDevice(&s->device));
    s->lim = lim; s->adjust = adjust_thres; s->alpha = alpha; s->lru_first = lru_first ? 1 : 0;
    s->host_thr = (float)threshold;
    s->fifo_hint = lim - 1;
    if (const char *e = getenv("CAFE_TARGET_ITEMS")) s->opt_target_items = std::max(16, atoi(e));   // tuning aids
    if (const char *e = getenv("CAFE_HOT_BYPASS")) s->opt_hot_bypass = atoi(e) ? 1 : 0;
    if (const char *e = getenv("CAFE_PRESORT")) s->opt_presort = atoi(e) ? 1 : 0;
    if (const char *e = getenv("CAFE_OWNER_CAP")) s->force_cap = atoi(e);
    if (const char *e = getenv("CAFE_R1_ORDER")) s->opt_r1_order = atoi(e) ? 1 : 0;
    s->n1 = (int)(lim * 0.9);           // sketch.cpp:51-52
    s->n2 = (int)(lim * 0.1);
    s->fifo_cap = (uint32_t)lim;
    CAFE_HIP_CHECK(hipMalloc(&s->d_main, sizeof(MainBucket) * (size_t)s->n1));
    CAFE_HIP_CHECK(hipMalloc(&s->d_lru, sizeof(LruBucket) * (size_t)s->n2));
    CAFE_HIP_CHECK(hipMalloc(&s->d_fifo, sizeof(uint32_t) * (size_t)s->fifo_cap));
    CAFE_HIP_CHECK(hipMalloc(&s->d_state, sizeof(DevState)));
    CAFE_HIP_CHECK(hipHostMalloc((void **)&s->h_pinned, 256, hipHostMallocMapped));
    memset(s->h_pinned, 0, 256);
    CAFE_HIP_CHECK(hipHostGetDevicePointer((void **)&s->d_pinned, s->h_pinned, 0));
    if (const char *e = getenv("CAFE_OVERLAP")) s->opt_overlap = atoi(e);
    if (const char *e = getenv("CAFE_SPLIT_UPDATE")) s->opt_split_update = atoi(e);
    CAFE_HIP_CHECK(hipEventCreateWithFlags(&s->ev_head_fork, kDeviceEvent));
    CAFE_HIP_CHECK(hipEventCreateWithFlags(&s->ev_head_join, kDeviceEvent));
    CAFE_HIP_CHECK(hipEventCreateWithFlags(&s->ev_plan, kDeviceEvent));
    CAFE_HIP_CHECK(hipEventCreateWithFlags(&s->ev_fwd, kDeviceEvent));
    CAFE_HIP_CHECK(hipEventCreateWithFlags(&s->ev_gn, kDeviceEvent));
    CAFE_HIP_CHECK(hipEventCreateWithFlags(&s->ev_commit, kDeviceEvent));
    if (s->ws_misc.reserve(4096)) return 1;
    CAFE_HIP_CHECK(hipMemset(s->d_main, 0, sizeof(MainBucket) * (size_t)s->n1));
    CAFE_HIP_CHECK(hipMemset(s->d_lru, 0, sizeof(LruBucket) * (size_t)s->n2));
    hipLaunchKernelGGL(k_fill_fifo, dim3((lim + 255) / 256), dim3(256), 0, 0, s->d_fifo, lim);
    hipLaunchKernelGGL(k_init_state, dim3(1), dim3(64), 0, 0, s->d_state, (float)threshold, lim);
    CAFE_HIP_CHECK(hipDeviceSynchronize());
    CAFE_HIP_CHECK(hipDeviceSynchronize());      // tables and scalars are in place whatever stream touches them first
    *out = s;
    return 0;
}

extern "C" int cafe_sketch_destroy(cafe_sketch_t *s) {
    if (!s) return 0;
    (void)hipDeviceSynchronize();
    par_free(s);
    (void)hipFree(s->d_main); (void)hipFree(s->d_lru); (void)hipFree(s->d_fifo); (void)hipFree(s->d_state);
    if (s->h_pinned) (void)hipHostFree(s->h_pinned);
    if (s->side_stream) (void)hipStreamDestroy(s->side_stream);
    if (s->low_stream) (void)hipStreamDestroy(s->low_stream);
    if (s->ev_gn) (void)hipEventDestroy(s->ev_gn);
    if (s->ev_head_fork) (void)hipEventDestroy(s->ev_head_fork);
    if (s->ev_head_join) (void)hipEventDestroy(s->ev_head_join);
    if (s->ev_plan) (void)hipEventDestroy(s->ev_plan);
    if (s->ev_fwd) (void)hipEventDestroy(s->ev_fwd);
    if (s->ev_commit) (void)hipEventDestroy(s->ev_commit);
    s->ws_reset_keys.release(); s->ws_reset_keys2.release(); s->ws_reset_tmp.release();
    s->ws_reset_flags.release(); s->ws_misc.release();
    delete s;
    return 0;
}

extern "C" int cafe_sketch_set_option(cafe_sketch_t *s, const char *name, int value) {
    if (!strcmp(name, "serial_insert")) s->opt_serial = value;
    else if (!strcmp(name, "max_rounds")) s->opt_max_rounds = value < 1 ? 1 : value;
    else if (!strcmp(name, "target_items")) s->opt_target_items = value < 16 ? 16 : value;
    else if (!strcmp(name, "hot_bypass")) s->opt_hot_bypass = value ? 1 : 0;
    else if (!strcmp(name, "exact_reset")) (void)value;          // kept for callers of older builds: the decision is always exact
    else if (!strcmp(name, "hot_table")) s->opt_hot_table = value ? 1 : 0;
    else if (!strcmp(name, "plan_ahead")) s->opt_plan_ahead = value ? 1 : 0;
    else if (!strcmp(name, "move_cap")) s->opt_move_cap = value;
    else if (!strcmp(name, "presort")) s->opt_presort = value ? 1 : 0;
    else if (!strcmp(name, "ids_ready")) s->opt_ids_ready = value ? 1 : 0;
    else if (!strcmp(name, "split_update")) s->opt_split_update = value;
    else if (!strcmp(name, "owner_cap")) s->force_cap = value == 1024 || value == 1536 ? 1536 : value == 2048 || value == 4608 || value == 8192 ? value : 0;     // 1536 = kCapSmall (cafe_insert_par.hip)
    else if (!strcmp(name, "overlap")) s->opt_overlap = value;
    else if (!strcmp(name, "fuse_fields")) s->opt_fuse_fields = value ? 1 : 0;
    else if (!strcmp(name, "verify")) s->opt_verify = value ? 1 : 0;
    else { set_error("unknown option %s", name); return 1; }
    return 0;
}

static int read_state(cafe_sketch *s, DevState *h);

extern "C" int cafe_sketch_fallback_reasons(cafe_sketch_t *s, int64_t *out4) {
    DevState h;
    if (read_state(s, &h)) return 1;
    for (int i = 0; i < 3; ++i) out4[i] = h.why[i];
    out4[3] = s->inner_reset_rejects;
    return 0;
}

extern "C" int cafe_sketch_counters(cafe_sketch_t *s, int64_t *out8) {
    DevState h;
    if (read_state(s, &h)) return 1;
    memcpy(out8, s->counters, sizeof(s->counters));
    out8[2] += h.pad[0];            // parallel calls that were handed to the in-order kernel
    out8[1] -= h.pad[0];
    out8[3] = h.pad[1];
    out8[5] = h.pad[2];
    out8[6] = h.cadence_miss;
    if (h.bad_commit) { set_error("internal: the commit kernel refused %d staged entries", h.bad_commit); return 1; }
    return 0;
}

static int read_state(cafe_sketch *s, DevState *h) {
    CAFE_HIP_CHECK(hipDeviceSynchronize());
    CAFE_HIP_CHECK(hipMemcpy(h, s->d_state, sizeof(DevState), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int cafe_sketch_clock_renorms(cafe_sketch_t *s, int64_t *out) {
    *out = s->clock_renorms;
    return 0;
}

extern "C" int cafe_sketch_dims(cafe_sketch_t *s, int32_t *out) {
    DevState h;
    if (read_state(s, &h)) return 1;
    out[0] = s->n1; out[1] = s->n2; out[2] = s->lim; out[3] = (int32_t)h.fifo_len;
    return 0;
}

extern "C" int cafe_sketch_scalars(cafe_sketch_t *s, float *thr, int32_t *ints6, double *decay) {
    DevState h;
    if (read_state(s, &h)) return 1;
    *thr = h.thr;
    ints6[0] = h.num; ints6[1] = h.real_n; ints6[2] = h.clk; ints6[3] = s->flip; ints6[4] = s->batch_num;
    ints6[5] = h.resets;
    *decay = s->decay;
    return 0;
}

extern "C" int cafe_sketch_export(cafe_sketch_t *s, int32_t *mval, float *mcnt, int32_t *mdic, uint8_t *mflip,
                                  int32_t *lval, float *lcnt, int32_t *ldic, int32_t *lt, uint8_t *lflip,
                                  uint32_t *fifo) {
    DevState h;
    if (read_state(s, &h)) return 1;
    std::vector<MainBucket> hm((size_t)s->n1);
    std::vector<LruBucket> hl((size_t)s->n2);
    std::vector<uint32_t> hf((size_t)s->fifo_cap);
    CAFE_HIP_CHECK(hipMemcpy(hm.data(), s->d_main, sizeof(MainBucket) * hm.size(), hipMemcpyDeviceToHost));
    CAFE_HIP_CHECK(hipMemcpy(hl.data(), s->d_lru, sizeof(LruBucket) * hl.size(), hipMemcpyDeviceToHost));
    CAFE_HIP_CHECK(hipMemcpy(hf.data(), s->d_fifo, 4 * hf.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < s->n1; ++k) {
        for (int i = 0; i < kSlots; ++i) {
            mval[k * kSlots + i] = hm[k].val[i]; mcnt[k * kSlots + i] = hm[k].cnt[i]; mdic[k * kSlots + i] = hm[k].dic[i];
        }
        mflip[k] = (uint8_t)hm[k].flip;
    }
    for (int k = 0; k < s->n2; ++k) {
        for (int i = 0; i < kSlots; ++i) {
            lval[k * kSlots + i] = hl[k].val[i]; lcnt[k * kSlots + i] = hl[k].cnt[i]; ldic[k * kSlots + i] = hl[k].dic[i];
            lt[k * kSlots + i] = hl[k].t[i];
        }
        lflip[k] = (uint8_t)hl[k].flip;
    }
    for (uint32_t i = 0; i < h.fifo_len; ++i) fifo[i] = hf[(h.fifo_head + i) % s->fifo_cap];
    return 0;
}

extern "C" int cafe_sketch_import(cafe_sketch_t *s, const int32_t *mval, const float *mcnt, const int32_t *mdic,
                                  const uint8_t *mflip, const int32_t *lval, const float *lcnt, const int32_t *ldic,
                                  const int32_t *lt, const uint8_t *lflip, const uint32_t *fifo, int fifo_len,
                                  float thr, const int32_t *ints6, double decay) {
    if (fifo_len < 0 || (uint32_t)fifo_len > s->fifo_cap) { set_error("import: bad free-list length %d", fifo_len); return 1; }
    // hot-table slots index rows of the caller's hot table (W_h[slot]); a value outside [0, lim) would address memory
    // the reference merely aliases (|dic| % hot_nums, sk_embedding_bag.py:134)
    for (size_t i = 0; i < (size_t)s->n1 * kSlots; ++i)
        if (mdic[i] < 0 || mdic[i] >= s->lim) { set_error("import: main-table slot %d outside [0, %d)", mdic[i], s->lim); return 1; }
    for (size_t i = 0; i < (size_t)s->n2 * kSlots; ++i) {
        if (ldic[i] < 0 || ldic[i] >= s->lim) { set_error("import: LRU-table slot %d outside [0, %d)", ldic[i], s->lim); return 1; }
        if (lt[i] < 0 || lt[i] > ints6[2]) { set_error("import: LRU timestamp %d outside [0, clock %d]", lt[i], ints6[2]); return 1; }
    }
    for (int i = 0; i < fifo_len; ++i)
        if (fifo[i] == 0 || fifo[i] >= (uint32_t)s->lim) { set_error("import: free slot %u outside [1, %d)", fifo[i], s->lim); return 1; }
    if (ints6[2] < 0) { set_error("import: negative LRU clock %d", ints6[2]); return 1; }
    std::vector<MainBucket> hm((size_t)s->n1);
    std::vector<LruBucket> hl((size_t)s->n2);
    std::vector<uint32_t> hf((size_t)s->fifo_cap, 0u);
    memset(hm.data(), 0, sizeof(MainBucket) * hm.size());
    memset(hl.data(), 0, sizeof(LruBucket) * hl.size());
    for (int k = 0; k < s->n1; ++k) {
        for (int i = 0; i < kSlots; ++i) {
            hm[k].val[i] = mval[k * kSlots + i]; hm[k].cnt[i] = mcnt[k * kSlots + i]; hm[k].dic[i] = mdic[k * kSlots + i];
        }
        hm[k].flip = mflip[k];
    }
    for (int k = 0; k < s->n2; ++k) {
        for (int i = 0; i < kSlots; ++i) {
            hl[k].val[i] = lval[k * kSlots + i]; hl[k].cnt[i] = lcnt[k * kSlots + i]; hl[k].dic[i] = ldic[k * kSlots + i];
            hl[k].t[i] = lt[k * kSlots + i];
        }
        hl[k].flip = lflip[k];
    }
    for (int i = 0; i < fifo_len; ++i) hf[i] = fifo[i];
    DevState h;
    memset(&h, 0, sizeof(h));
    h.thr = thr; h.num = ints6[0]; h.real_n = ints6[1]; h.clk = ints6[2]; h.flip = ints6[3]; h.resets = ints6[5];
    h.fifo_head = 0; h.fifo_len = (uint32_t)fifo_len;
    CAFE_HIP_CHECK(hipDeviceSynchronize());
    CAFE_HIP_CHECK(hipMemcpy(s->d_main, hm.data(), sizeof(MainBucket) * hm.size(), hipMemcpyHostToDevice));
    CAFE_HIP_CHECK(hipMemcpy(s->d_lru, hl.data(), sizeof(LruBucket) * hl.size(), hipMemcpyHostToDevice));
    CAFE_HIP_CHECK(hipMemcpy(s->d_fifo, hf.data(), 4 * hf.size(), hipMemcpyHostToDevice));
    CAFE_HIP_CHECK(hipMemcpy(s->d_state, &h, sizeof(h), hipMemcpyHostToDevice));
    s->flip = ints6[3]; s->batch_num = ints6[4]; s->decay = decay;
    s->plan_valid = 0;
    s->base_real_n = h.real_n;
    s->base_clk = h.clk;
    s->base_seq = s->call_seq;
    s->host_thr = thr;
    s->thr_dirty_seq = 0;
    s->fifo_hint = fifo_len;
    return 0;
}

// file format (little endian): "CAFESK01", int32 lim n1 n2 adjust lru_first fifo_len, float thr,
// int32 ints6[6], double decay, double alpha, then the export arrays in the order of cafe_sketch_export.
extern "C" int cafe_sketch_save(cafe_sketch_t *s, const char *path) {
    int32_t dims[4];
    if (cafe_sketch_dims(s, dims)) return 1;
    size_t a = (size_t)s->n1 * kSlots, b = (size_t)s->n2 * kSlots;
    std::vector<int32_t> mval(a), mdic(a), lval(b), ldic(b), lt(b);
    std::vector<float> mcnt(a), lcnt(b);
    std::vector<uint8_t> mflip(s->n1 + 1), lflip(s->n2 + 1);
    std::vector<uint32_t> fifo((size_t)dims[3] + 1);
    float thr; int32_t ints[6]; double decay;
    if (cafe_sketch_scalars(s, &thr, ints, &decay)) return 1;
    if (cafe_sketch_export(s, mval.data(), mcnt.data(), mdic.data(), mflip.data(), lval.data(), lcnt.data(), ldic.data(),
                           lt.data(), lflip.data(), fifo.data())) return 1;
    FILE *f = fopen(path, "wb");
    if (!f) { set_error("cannot open %s for writing", path); return 1; }
    int32_t hdr[6] = {s->lim, s->n1, s->n2, s->adjust, s->lru_first, dims[3]};
    bool ok = fwrite("CAFESK01", 1, 8, f) == 8 && fwrite(hdr, 4, 6, f) == 6 && fwrite(&thr, 4, 1, f) == 1 &&
              fwrite(ints, 4, 6, f) == 6 && fwrite(&decay, 8, 1, f) == 1 && fwrite(&s->alpha, 8, 1, f) == 1;
    ok = ok && fwrite(mval.data(), 4, a, f) == a && fwrite(mcnt.data(), 4, a, f) == a && fwrite(mdic.data(), 4, a, f) == a &&
         fwrite(mflip.data(), 1, s->n1, f) == (size_t)s->n1 && fwrite(lval.data(), 4, b, f) == b &&
         fwrite(lcnt.data(), 4, b, f) == b && fwrite(ldic.data(), 4, b, f) == b && fwrite(lt.data(), 4, b, f) == b &&
         fwrite(lflip.data(), 1, s->n2, f) == (size_t)s->n2 && fwrite(fifo.data(), 4, dims[3], f) == (size_t)dims[3];
    fclose(f);
    if (!ok) { set_error("short write to %s", path); return 1; }
    return 0;
}

extern "C" int cafe_sketch_load(cafe_sketch_t *s, const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { set_error("cannot open %s", path); return 1; }
    char magic[8]; int32_t hdr[6]; float thr; int32_t ints[6]; double decay, alpha;
    bool ok = fread(magic, 1, 8, f) == 8 && !memcmp(magic, "CAFESK01", 8) && fread(hdr, 4, 6, f) == 6 &&
              fread(&thr, 4, 1, f) == 1 && fread(ints, 4, 6, f) == 6 && fread(&decay, 8, 1, f) == 1 &&
              fread(&alpha, 8, 1, f) == 1;
    if (!ok || hdr[0] != s->lim || hdr[1] != s->n1 || hdr[2] != s->n2 || hdr[5] < 0 || (uint32_t)hdr[5] > s->fifo_cap) {
        fclose(f);
        set_error("%s: not a CAFE sketch state for lim=%d", path, s->lim);
        return 1;
    }
    size_t a = (size_t)s->n1 * kSlots, b = (size_t)s->n2 * kSlots, fl = (size_t)hdr[5];
    std::vector<int32_t> mval(a), mdic(a), lval(b), ldic(b), lt(b);
    std::vector<float> mcnt(a), lcnt(b);
    std::vector<uint8_t> mflip(s->n1 + 1), lflip(s->n2 + 1);
    std::vector<uint32_t> fifo(fl + 1);
    ok = fread(mval.data(), 4, a, f) == a && fread(mcnt.data(), 4, a, f) == a && fread(mdic.data(), 4, a, f) == a &&
         fread(mflip.data(), 1, s->n1, f) == (size_t)s->n1 && fread(lval.data(), 4, b, f) == b &&
         fread(lcnt.data(), 4, b, f) == b && fread(ldic.data(), 4, b, f) == b && fread(lt.data(), 4, b, f) == b &&
         fread(lflip.data(), 1, s->n2, f) == (size_t)s->n2 && fread(fifo.data(), 4, fl, f) == fl;
    fclose(f);
    if (!ok) { set_error("%s: truncated", path); return 1; }
    s->adjust = hdr[3]; s->lru_first = hdr[4]; s->alpha = alpha;
    return cafe_sketch_import(s, mval.data(), mcnt.data(), mdic.data(), mflip.data(), lval.data(), lcnt.data(), ldic.data(),
                              lt.data(), lflip.data(), fifo.data(), (int)fl, thr, ints, decay);
}

// rebuild the free list: ascending slot numbers that no entry of either table holds
__global__ void k_mark_slots(const MainBucket *mt, const LruBucket *lt, int n1, int n2, uint32_t *mark, int lim) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n1) {
        for (int i = 0; i < kSlots; ++i) { int d = mt[g].dic[i]; if (d > 0 && d < lim) mark[d] = 1; }
    } else if (g < (long)n1 + n2) {
        for (int i = 0; i < kSlots; ++i) { int d = lt[g - n1].dic[i]; if (d > 0 && d < lim) mark[d] = 1; }
    }
}
__global__ void k_rebuild_fifo(const uint32_t *mark, uint32_t *fifo, DevState *st, int lim) {
    // one workgroup, ordered compaction in chunks of blockDim.x
    __shared__ int base, wsum[16];
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int c = 1; c < lim; c += blockDim.x) {
        int i = c + threadIdx.x;
        int f = (i < lim && !mark[i]) ? 1 : 0;
        unsigned long long m = __ballot(f);
        int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) wsum[w] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int k = 0; k < w; ++k) off += wsum[k];
        if (f) fifo[off + __popcll(m & ((1ull << lane) - 1))] = (uint32_t)i;
        __syncthreads();
        if (threadIdx.x == 0) { int tot = 0; for (int k = 0; k < (int)(blockDim.x >> 6); ++k) tot += wsum[k]; base += tot; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { st->fifo_head = 0; st->fifo_len = (uint32_t)base; }
}

extern "C" int cafe_sketch_rebuild_free_list(cafe_sketch_t *s, cafe_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    if (s->ws_reset_flags.reserve((size_t)s->lim * 4 + 64)) return 1;
    uint32_t *mark = s->ws_reset_flags.as<uint32_t>();
    CAFE_HIP_CHECK(hipMemsetAsync(mark, 0, (size_t)s->lim * 4, st));
    long total = (long)s->n1 + s->n2;
    hipLaunchKernelGGL(k_mark_slots, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, s->d_main, s->d_lru, s->n1,
                       s->n2, mark, s->lim);
    hipLaunchKernelGGL(k_rebuild_fifo, dim3(1), dim3(1024), 0, st, mark, s->d_fifo, s->d_state, s->lim);
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// device-pointer entry points
// ------------------------------------------------------------------------------------------------
extern "C" int cafe_query_dev(cafe_sketch_t *s, const int32_t *d_ids, int n, int32_t *d_que, cafe_stream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_query, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->d_main, s->n1, d_ids, n, d_que);
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cafe_query_median_dev(cafe_sketch_t *s, const int32_t *d_ids, int n, int32_t *d_out, cafe_stream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_query_median, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->d_main, s->n1, s->d_state,
                       s->flip, d_ids, n, d_out);
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cafe_insert_dev(cafe_sketch_t *s, const int32_t *d_ids, const float *d_scores, int n, int32_t *d_ins,
                               cafe_stream_t stream) {
    return insert_dev_on(s, d_ids, d_scores, nullptr, 0, n, d_ins, nullptr, (hipStream_t)stream, (hipStream_t)stream);
}

// ---- real_n as the host knows it ------------------------------------------------------------------
// sketch.cpp:291 needs real_n at the start of every call.  The value lives on the device; the kernel that closes an
// insert leaves a tagged snapshot in pinned memory.  Upper bound = newest snapshot + the ids of the calls issued since
// (every inserted id raises real_n by at most one); only when that bound passes 1.2 lim does the host wait for the
// exact value, so the decision is never an estimate.  cafe_backward_update_dev waits for every step's verdict anyway,
// which keeps the snapshot current.
static void read_mirror(cafe_sketch *s, Mirror *out) {
    // the device writes the payload, a system-scope fence, then seq: a snapshot is consistent when seq reads the same
    // before and after the payload
    const volatile uint32_t *src = reinterpret_cast<const volatile uint32_t *>(s->h_pinned);
    const volatile uint32_t *seqp = &reinterpret_cast<const volatile Mirror *>(s->h_pinned)->seq;
    uint32_t a[sizeof(Mirror) / 4];
    for (int tries = 0; tries < 64; ++tries) {
        const uint32_t s0 = *seqp;
        __sync_synchronize();
        for (size_t i = 0; i < sizeof(Mirror) / 4; ++i) a[i] = src[i];
        __sync_synchronize();
        if (*seqp == s0 && reinterpret_cast<Mirror *>(a)->seq == s0) break;
    }
    memcpy(out, a, sizeof(Mirror));
}

static void take_snapshot(cafe_sketch *s, const Mirror &m) {
    s->base_real_n = m.st.real_n;
    s->base_clk = m.st.clk;
    s->last_moves = m.moves;
    s->base_seq = m.seq;
    s->fifo_hint = m.st.fifo_len;
    if (s->thr_dirty_seq && (int32_t)(m.seq - s->thr_dirty_seq) >= 0) {
        s->host_thr = m.st.thr;         // the snapshot is behind the reset (and every flip) that made it unknown
        s->thr_dirty_seq = 0;
    }
}

long long cafe::real_n_bound(cafe_sketch *s) {
    Mirror m;
    read_mirror(s, &m);
    if (m.seq != 0 && (int32_t)(m.seq - s->base_seq) > 0 && (int32_t)(s->call_seq - m.seq) >= 0) take_snapshot(s, m);
    if (s->call_seq - s->base_seq > 64u) return (long long)1 << 60;        // issued calls no longer in the ring
    long long ub = s->base_real_n;
    for (uint32_t q = s->base_seq + 1; (int32_t)(s->call_seq - q) >= 0; ++q) ub += s->issued[q & 63].n;
    return ub;
}

// the same for the LRU clock (at most one tick per id)
static long long clk_bound(cafe_sketch *s) {
    (void)cafe::real_n_bound(s);        // consumes the newest snapshot
    if (s->call_seq - s->base_seq > 64u) return (long long)1 << 60;
    long long ub = s->base_clk;
    for (uint32_t q = s->base_seq + 1; (int32_t)(s->call_seq - q) >= 0; ++q) ub += s->issued[q & 63].n;
    return ub;
}

// with Threshold <= 1 an id can turn hot inside the LRU table, and whether an eviction of a later field moves
// it into the main table decides a field-by-field migration; such steps are issued field by field.
bool cafe::threshold_known_above_one(cafe_sketch *s) {
    (void)real_n_bound(s);              // consumes the newest snapshot
    return s->thr_dirty_seq == 0 && s->host_thr > 1.0f;
}

// waits for the insert issued last; afterwards the snapshot in pinned memory is that call's.  The closing kernel writes
// the snapshot, a system-scope fence, then its call number: the host polls that word (no event in the stream, no
// interrupt), and falls back to a stream synchronisation when the call takes long (the in-order paths).
static int wait_last(cafe_sketch *s, Mirror *m) {
    const volatile uint32_t *seqp = &reinterpret_cast<const volatile Mirror *>(s->h_pinned)->seq;
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (long spin = 0; !seen; ++spin) {
        if (*seqp == s->call_seq) { seen = true; break; }
        __builtin_ia32_pause();
        if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    if (!seen) {
        // a step that does not come back within ten seconds is reported once, with what each stream of the handle is
        // doing, before the wait goes on
        for (;;) {
            const hipError_t q = hipStreamQuery(s->last_stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) CAFE_HIP_CHECK(q);
            if (*seqp == s->call_seq) break;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) {
                auto name = [](hipError_t e) { return e == hipSuccess ? "done" : e == hipErrorNotReady ? "busy" : "error"; };
                fprintf(stderr, "[cafe] call %u: no verdict after 10 s (snapshot %u): caller's stream %s, second stream %s, third stream %s; "
                        "events: plan %s, forward %s, norms %s\n", s->call_seq, *seqp, name(q),
                        s->side_stream ? name(hipStreamQuery(s->side_stream)) : "-", s->low_stream ? name(hipStreamQuery(s->low_stream)) : "-",
                        name(hipEventQuery(s->ev_plan)), name(hipEventQuery(s->ev_fwd)), s->ev_gn ? name(hipEventQuery(s->ev_gn)) : "-");
                // a library call must not block for ever: the call fails (CAFE_WAIT_FOREVER=1: keep waiting, for debugging)
                if (const char *e = getenv("CAFE_WAIT_FOREVER"); e && atoi(e)) {
                    CAFE_HIP_CHECK(hipStreamSynchronize(s->last_stream));
                    break;
                }
                set_error("call %u: no verdict from the device after 10 s (snapshot %u); the streams' states are on stderr", s->call_seq, *seqp);
                return 1;
            }
            __builtin_ia32_pause();
        }
    }
    read_mirror(s, m);
    if (m->seq != s->call_seq) { set_error("internal: state snapshot %u does not belong to call %u", m->seq, s->call_seq); return 1; }
    take_snapshot(s, *m);
    s->counters[7]++;
    return 0;
}

// the exact value: waits for the last insert's snapshot
int cafe::real_n_exact(cafe_sketch *s, long long *out) {
    if (s->call_seq != s->base_seq) {
        Mirror m;
        if (wait_last(s, &m)) return 1;
    }
    *out = s->base_real_n;
    return 0;
}

// waits for the insert issued last and tells whether the device rejected it (verify mode), and why
int cafe::last_insert_verdict(cafe_sketch *s, bool *rejected, int *reason) {
    Mirror m;
    if (wait_last(s, &m)) return 1;
    *rejected = m.rejected != 0;
    *reason = (int)m.reason;
    if (*rejected) s->counters[1]--;        // it was counted as a call finished by the parallel path
    return 0;
}

int cafe::insert_dev_on(cafe_sketch *s, const int32_t *d_ids, const float *d_vals, const float *d_psum, int nparts, int n,
                        int32_t *d_ins, float *d_scores_out, hipStream_t st, hipStream_t pre) {
    if (n < 0) { set_error("cafe_insert_dev: negative length"); return 1; }
    // sketch.cpp:282-289
    s->batch_num++;
    s->decay *= s->alpha;
    int flip_now = 0;
    if (s->decay > kDecayV) { s->decay /= kDecayV; s->flip ^= 1; flip_now = 1; }
    if (flip_now) {                              // thr = (float)((double)thr / V)
        if (s->thr_dirty_seq) s->thr_dirty_seq = s->call_seq + 1;
        else s->host_thr = (float)((double)s->host_thr / kDecayV);
    }
    // a call that stands for several reference calls (one per seg_len ids): the caller made sure that no flip falls on
    // an inner boundary (the device checks the reset boundaries), so the remaining calls only advance the counters
    if (s->seg_len > 0 && n > 0)
        for (int k = 1; k < (n + s->seg_len - 1) / s->seg_len; ++k) { s->batch_num++; s->decay *= s->alpha; }
    const bool parallel = !s->opt_serial && n > 0 && n <= kParMaxIds;
    if (parallel && par_prepare(s, n)) return 1;
    if (pre != st && !parallel) {           // nothing to run beside the scores
        CAFE_HIP_CHECK(hipEventRecord(s->ev_head_join, pre));
        CAFE_HIP_CHECK(hipStreamWaitEvent(st, s->ev_head_join, 0));
        pre = st;
    }
    // LRU clock lifetime (k_clock_renorm): every id ticks the clock at most once
    if (clk_bound(s) + n + 2 >= (long long)kClockRenormAt) {
        Mirror m;
        if (s->call_seq != s->base_seq && wait_last(s, &m)) return 1;       // the exact clock
        if ((long long)s->base_clk + n + 2 >= (long long)kClockRenormAt) {
            hipLaunchKernelGGL(k_clock_renorm, dim3((s->n2 + 255) / 256), dim3(256), 0, pre, s->d_lru, s->n2, s->d_state);
            s->base_clk = kSlots;
            s->clock_renorms++;
        }
    }
    // sketch.cpp:291
    if (s->adjust && (double)real_n_bound(s) > s->lim * 1.2) {
        long long exact;
        if (real_n_exact(s, &exact)) return 1;          // only when a reset may be due
        if ((double)exact > s->lim * 1.2) {
            // the reference flips first (sketch.cpp:284-289), then resets (:291): the scalars go in ahead of the reset
            hipLaunchKernelGGL(k_prologue, dim3(1), dim3(64), 0, pre, s->d_state, flip_now, s->flip);
            flip_now = 0;
            s->plan_stale = 1;                           // the tables change under a plan made beside the forward
            if (run_reset(s, pre)) return 1;
            s->base_real_n = s->lim;                     // sketch.cpp:113
            s->base_seq = s->call_seq;
            s->thr_dirty_seq = s->call_seq + 1;          // sketch.cpp:112 rewrites Threshold on the device
            s->fifo_hint = -1;                           // ... and hands slots back and out
        }
    }
    s->counters[0]++;
    if (n == 0) {
        hipLaunchKernelGGL(k_prologue, dim3(1), dim3(64), 0, pre, s->d_state, flip_now, s->flip);
        if (pre != st) {
            CAFE_HIP_CHECK(hipEventRecord(s->ev_head_join, pre));
            CAFE_HIP_CHECK(hipStreamWaitEvent(st, s->ev_head_join, 0));
        }
        return 0;
    }
    s->call_seq++;
    s->issued[s->call_seq & 63] = {s->call_seq, n};
    int rc;
    if (!parallel) {
        hipLaunchKernelGGL(k_prologue, dim3(1), dim3(64), 0, st, s->d_state, flip_now, s->flip);
        rc = launch_serial_insert(s, d_ids, d_vals, n, d_ins, st);
        s->counters[2]++;
    } else {
        rc = par_insert(s, d_ids, d_vals, d_psum, nparts, n, d_ins, d_scores_out, flip_now, st, pre);
    }
    if (rc == 0) {
        s->last_stream = st;       // the closing kernel mirrors the device state into pinned host memory
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------
// (A) reference-compatible ABI on host pointers: ids are staged to the device, the same kernels run,
// results come back into library-owned storage (the reference returns static arrays, sketch.cpp:10).
// ------------------------------------------------------------------------------------------------
static cafe_sketch *g_sk = nullptr;
static std::vector<int> g_ins, g_que;
static DevBuf g_d_ids, g_d_sc, g_d_out;

static int stage(const uint32_t *data, const float *v, int len) {
    if (g_d_ids.reserve((size_t)len * 4 + 4) || g_d_out.reserve((size_t)len * 4 + 4)) return 1;
    CAFE_HIP_CHECK(hipMemcpy(g_d_ids.p, data, (size_t)len * 4, hipMemcpyHostToDevice));
    if (v) {
        if (g_d_sc.reserve((size_t)len * 4 + 4)) return 1;
        CAFE_HIP_CHECK(hipMemcpy(g_d_sc.p, v, (size_t)len * 4, hipMemcpyHostToDevice));
    }
    return 0;
}

static void die_if(int rc, const char *what) {
    if (rc) {
        // the reference ABI has no error channel (sketch.cpp:300-327); a GPU failure must not pass silently
        fprintf(stderr, "libcafe_hip: %s failed: %s\n", what, cafe_last_error());
        abort();
    }
}

extern "C" cafe_sketch_t *cafe_global_sketch(void) { return g_sk; }

extern "C" void init(int n, int Threshold, int adjust_thres, double alp) {
    if (g_sk) { cafe_sketch_destroy(g_sk); g_sk = nullptr; }
    die_if(cafe_sketch_create(&g_sk, n, Threshold, adjust_thres, alp, 1), "init");
}

extern "C" int *batch_query(uint32_t *data, int len) {
    die_if(g_sk == nullptr, "batch_query before init");
    g_que.resize((size_t)std::max(len, 1));
    if (len <= 0) return g_que.data();
    die_if(stage(data, nullptr, len), "batch_query");
    die_if(cafe_query_dev(g_sk, g_d_ids.as<int32_t>(), len, g_d_out.as<int32_t>(), nullptr), "batch_query");
    die_if(hipMemcpy(g_que.data(), g_d_out.p, (size_t)len * 4, hipMemcpyDeviceToHost) != hipSuccess, "batch_query copy");
    return g_que.data();
}

extern "C" int *batch_query_median(uint32_t *data, int len) {
    static std::vector<int> med;
    die_if(g_sk == nullptr, "batch_query_median before init");
    med.resize((size_t)std::max(len, 1));
    if (len <= 0) return med.data();
    die_if(stage(data, nullptr, len), "batch_query_median");
    die_if(cafe_query_median_dev(g_sk, g_d_ids.as<int32_t>(), len, g_d_out.as<int32_t>(), nullptr), "batch_query_median");
    die_if(hipMemcpy(med.data(), g_d_out.p, (size_t)len * 4, hipMemcpyDeviceToHost) != hipSuccess, "batch_query_median copy");
    return med.data();
}

static int *insert_common(uint32_t *data, float *v, int len) {
    die_if(g_sk == nullptr, "batch_insert before init");
    g_ins.resize((size_t)std::max(len, 1));
    if (len > 0) die_if(stage(data, v, len), "batch_insert");
    die_if(cafe_insert_dev(g_sk, g_d_ids.as<int32_t>(), v ? g_d_sc.as<float>() : nullptr, len < 0 ? 0 : len,
                           g_d_out.as<int32_t>(), nullptr), "batch_insert");
    if (len > 0)
        die_if(hipMemcpy(g_ins.data(), g_d_out.p, (size_t)len * 4, hipMemcpyDeviceToHost) != hipSuccess, "batch_insert copy");
    return g_ins.data();
}

extern "C" int *batch_insert(uint32_t *data, int len) { return insert_common(data, nullptr, len); }
extern "C" int *batch_insert_val(uint32_t *data, float *v, int len) { return insert_common(data, v, len); }

static double count_common(uint32_t *data, int len, int mode) {
    die_if(g_sk == nullptr, "analyse before init");
    if (len <= 0) return 0.0 / 0.0;           // the reference divides by len too
    die_if(stage(data, nullptr, len), "analyse");
    unsigned long long *d_cnt = (unsigned long long *)g_sk->ws_misc.as<char>() + 8;
    die_if(hipMemset(d_cnt, 0, 8) != hipSuccess, "analyse memset");
    hipLaunchKernelGGL(k_count, dim3((len + 255) / 256), dim3(256), 0, 0, g_sk->d_main, g_sk->n1, g_d_ids.as<int32_t>(), len,
                       mode, d_cnt);
    unsigned long long h = 0;
    die_if(hipMemcpy(&h, d_cnt, 8, hipMemcpyDeviceToHost) != hipSuccess, "analyse copy");
    return 1.0 * (double)h / len;
}
extern "C" double analyse(uint32_t *data, int len) { return count_common(data, len, 0); }
extern "C" double analyse1(uint32_t *data, int len) { return count_common(data, len, 1); }

extern "C" void update(void) {
    die_if(g_sk == nullptr, "update before init");
    die_if(cafe_sketch_rebuild_free_list(g_sk, nullptr), "update");
    die_if(hipDeviceSynchronize() != hipSuccess, "update sync");
}
extern "C" int save_state(const char *path) { return g_sk ? cafe_sketch_save(g_sk, path) : 1; }
extern "C" int load_state(const char *path) { return g_sk ? cafe_sketch_load(g_sk, path) : 1; }
