// cafe_ae.hip -- the earlier HotSketch generation of the reference's artifact (ArtifactEvaluation/embeddings/
// sketch.cpp:9-151, class SS; C entry points :207-228) on the GPU, bit-exact: `lim` buckets of four slots kept
// sorted by score, bucket = id % lim, a FIFO free list of slot numbers, a multiplicative decay() when the scores
// inserted since the last one exceed lim * Threshold * 10.  One handle per field (the reference keeps ss[26]).
//
// The reference inserts a batch strictly in order.  Items interact through their bucket and through the free list
// only, so a batch is replayed by bucket owners:
//     k_ae_pre      decay due? (tot > lim * Threshold * 10, :138, :146)
//     k_ae_decay_*  three passes that do nothing unless it is: slots handed back in (bucket, slot) order (:76-88)
//     k_ae_keys + stable radix sort (rocPRIM) by bucket: every bucket's items contiguous, in batch order
//     k_ae_replay   one lane per bucket walks its items on a register copy of the bucket (:89-124).  "Takes a slot
//                   from the free list" is recorded as a placeholder (the item's batch position): which slot it gets
//                   depends on how many items in front of it, in ALL buckets, took one.  Results go to a staging area.
//     exclusive scan (rocPRIM) of the take flags in batch order
//     k_ae_commit   if the free list covers every take of the batch: staged buckets -> table with the placeholders
//                   replaced by fifo[head + rank]
//     k_ae_close    free-list head / length, num, and tot += v in batch order (a double accumulated sequentially: the
//                   only order that gives the reference's bits).  If the free list would run dry inside the batch
//                   (takes fail from then on, which changes what later items of those buckets do), nothing was
//                   committed and one lane runs the batch in order from the untouched table instead.
//
// Sizes here are the artifact's: one field's lookups per call (the reference's result buffers hold 16384, :7).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <string.h>

#include <algorithm>
#include <vector>

#include "cafe_internal.h"

using namespace cafe;

namespace {

constexpr int kW = 4;
constexpr uint32_t kTake = 0x80000000u;        // dic placeholder: "slot taken by the item at batch position (v & 0x7fffffff)"

struct __attribute__((aligned(16))) AeBucket {
    uint32_t val[kW];
    float cnt[kW];
    int32_t dic[kW];
};

struct AeState {
    double tot;
    int32_t num;
    uint32_t head, len;
    int32_t do_decay;
    int32_t decays;
    int32_t serial_calls;      // batches that ran in order because the free list ran dry inside them
    uint32_t decay_total;
};

}  // namespace

struct cafe_ae_sketch {
    int device = 0;
    int s = 0, k = 0, lim = 0;
    double decay_rate = 1.0;
    AeBucket *d_b = nullptr;
    uint32_t *d_fifo = nullptr;
    uint32_t cap = 1;
    AeState *d_state = nullptr;
    DevBuf keys, pos, att, rank, stage, tmp, blockcnt, mark;
    size_t tmp_bytes = 0;
    // host staging of the reference-style entry points
    std::vector<int> h_out;
    DevBuf st_ids, st_vals, st_out;
};

namespace {

__global__ void k_ae_fill(uint32_t *fifo, int lim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < lim) fifo[i] = (uint32_t)(i + 1);
}

__global__ void k_ae_pre(AeState *st, int s, int k) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        st->do_decay = st->tot > (double)(1ll * s * k * 10) ? 1 : 0;      // :138, :146
        st->decay_total = 0;
    }
}

// decay (:76-88), slot j = bucket * 4 + i in ascending order.  A block takes 1024 consecutive slots.
__device__ inline bool ae_gives_back(const AeBucket *b, int j, double dr, int k) {
    const AeBucket &bk = b[j >> 2];
    return bk.dic[j & 3] != 0 && (double)bk.cnt[j & 3] * dr < (double)k;
}

__global__ void __launch_bounds__(256) k_ae_decay_count(const AeBucket *b, const AeState *st, int s, double dr, int k, uint32_t *blockcnt) {
    if (!st->do_decay) return;
    __shared__ uint32_t ws[4];
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
    uint32_t c = 0;
    for (int u = 0; u < 4; ++u) c += (base + u < s * kW && ae_gives_back(b, base + u, dr, k)) ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ void __launch_bounds__(256) k_ae_decay_apply(AeBucket *b, AeState *st, int s, double dr, int k, const uint32_t *blockcnt,
                                                        uint32_t *fifo, uint32_t cap) {
    if (!st->do_decay) return;
    __shared__ uint32_t ws[4];
    __shared__ uint32_t s_base;
    // slots handed back by the blocks in front of this one
    uint32_t before = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) before += blockcnt[i];
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = before;
    __syncthreads();
    if (threadIdx.x == 0) s_base = ws[0] + ws[1] + ws[2] + ws[3];
    __syncthreads();
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
    bool gb[4];
    uint32_t c = 0;
    for (int u = 0; u < 4; ++u) { gb[u] = base + u < s * kW && ae_gives_back(b, base + u, dr, k); c += gb[u]; }
    // exclusive prefix of c over the block, in thread order
    uint32_t inc = c;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(inc, d, 64); if ((threadIdx.x & 63) >= d) inc += x; }
    __syncthreads();
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t at = s_base + inc - c;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) at += ws[w];
    const uint32_t tail = st->head + st->len;          // both unchanged until k_ae_decay_fin
    for (int u = 0; u < 4; ++u) {
        const int j = base + u;
        if (j >= s * kW) break;
        AeBucket &bk = b[j >> 2];
        if (gb[u]) {
            fifo[(tail + at) % cap] = (uint32_t)bk.dic[j & 3];
            bk.dic[j & 3] = 0;
            ++at;
        }
        bk.cnt[j & 3] = (float)((double)bk.cnt[j & 3] * dr);
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) st->decay_total = at;       // the last lane's end = the total
}

__global__ void k_ae_decay_fin(AeState *st) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && st->do_decay) {
        st->len += st->decay_total;
        st->tot = 0;
        st->decays += 1;
    }
}

__global__ void k_ae_keys(const int32_t *ids, int n, int s, uint32_t *keys, uint32_t *pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = (uint32_t)ids[i] % (uint32_t)s; pos[i] = (uint32_t)i; }
}

// :89-124 on a register copy.  `take`: the free list is known not to be empty (a placeholder is recorded) -- or, in the
// in-order path, ctx pops for real.
template <class Pop>
__device__ inline int ae_insert(AeBucket &b, uint32_t v, float c, int k, bool can_take, Pop pop) {
    int id = 0;
    for (int i = 0; i < kW; ++i)
        if (b.cnt[i] != 0.0f && b.val[i] == v) {
            b.cnt[i] += c;
            if (b.cnt[i] >= (float)k && can_take && !b.dic[i]) { b.dic[i] = pop(); id = 1; }
            while (i && b.cnt[i] > b.cnt[i - 1]) {
                const float tc = b.cnt[i]; b.cnt[i] = b.cnt[i - 1]; b.cnt[i - 1] = tc;
                const uint32_t tv = b.val[i]; b.val[i] = b.val[i - 1]; b.val[i - 1] = tv;
                const int32_t td = b.dic[i]; b.dic[i] = b.dic[i - 1]; b.dic[i - 1] = td;
                --i;
            }
            return id;
        }
    for (int i = 0; i < kW; ++i)
        if (b.cnt[i] == 0.0f) { b.cnt[i] = c; b.val[i] = v; return 0; }
    if (!b.dic[3]) { b.cnt[3] += c; b.val[3] = v; }
    return 0;
}

__global__ void __launch_bounds__(256) k_ae_replay(const AeBucket *b, const AeState *st, const int32_t *ids, const float *vals, int n, int k,
                                                   const uint32_t *keys, const uint32_t *pos, uint32_t *att, int32_t *ins,
                                                   AeBucket *stage) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t key = keys[i];
    if (i != 0 && keys[i - 1] == key) return;           // not the first item of its bucket
    AeBucket bk = b[key];
    const bool can_take = st->len > 0;                  // stays true for the whole batch if the commit goes through
    for (int j = i; j < n && keys[j] == key; ++j) {
        const uint32_t p = pos[j];
        const int got = ae_insert(bk, (uint32_t)ids[p], vals ? vals[p] : 1.0f, k, can_take, [&]() { return (int32_t)(kTake | p); });
        att[p] = (uint32_t)got;
        ins[p] = got;
    }
    stage[i] = bk;
}

__global__ void __launch_bounds__(256) k_ae_commit(AeBucket *b, const AeState *st, int n, const uint32_t *keys, const uint32_t *att,
                                                   const uint32_t *rank, const AeBucket *stage, const uint32_t *fifo, uint32_t cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t takes = rank[n - 1] + att[n - 1];
    if (takes > st->len) return;                        // the free list runs dry inside the batch: k_ae_close runs it in order
    const uint32_t key = keys[i];
    if (i != 0 && keys[i - 1] == key) return;
    AeBucket bk = stage[i];
    for (int u = 0; u < kW; ++u)
        if ((uint32_t)bk.dic[u] & kTake) bk.dic[u] = (int32_t)fifo[(st->head + rank[(uint32_t)bk.dic[u] & ~kTake]) % cap];
    b[key] = bk;
}

// one workgroup: scalars, the ordered double sum, and the in-order path
__global__ void __launch_bounds__(256) k_ae_close(AeBucket *b, AeState *st, const int32_t *ids, const float *vals, int n, int s, int k,
                                                  const uint32_t *att, const uint32_t *rank, uint32_t *fifo, uint32_t cap, int32_t *ins) {
    __shared__ double chunk[1024];
    const uint32_t takes = rank[n - 1] + att[n - 1];
    const bool dry = takes > st->len;
    if (!dry) {
        double tot = st->tot;
        for (int base = 0; base < n; base += 1024) {
            for (int j = threadIdx.x; j < 1024; j += 256) chunk[j] = base + j < n ? (double)(vals ? vals[base + j] : 1.0f) : 0.0;
            __syncthreads();
            if (threadIdx.x == 0) {
                const int m = min(1024, n - base);
                int j = 0;
                for (; j + 8 <= m; j += 8) {
                    const double c0 = chunk[j], c1 = chunk[j + 1], c2 = chunk[j + 2], c3 = chunk[j + 3], c4 = chunk[j + 4],
                                 c5 = chunk[j + 5], c6 = chunk[j + 6], c7 = chunk[j + 7];
                    tot += c0; tot += c1; tot += c2; tot += c3; tot += c4; tot += c5; tot += c6; tot += c7;
                }
                for (; j < m; ++j) tot += chunk[j];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            st->tot = tot;
            st->head = (st->head + takes) % cap;
            st->len -= takes;
            st->num += (int32_t)takes;
        }
        return;
    }
    if (threadIdx.x == 0) {
        // :142-150 as written, on the table itself (nothing of this batch was committed)
        uint32_t head = st->head, len = st->len;
        int32_t num = st->num;
        double tot = st->tot;
        for (int i = 0; i < n; ++i) {
            const uint32_t v = (uint32_t)ids[i];
            const float c = vals ? vals[i] : 1.0f;
            tot += c;
            AeBucket bk = b[v % (uint32_t)s];
            ins[i] = ae_insert(bk, v, c, k, len > 0, [&]() {
                const uint32_t d = fifo[head];
                head = head + 1 == cap ? 0 : head + 1;
                --len; ++num;
                return (int32_t)d;
            });
            b[v % (uint32_t)s] = bk;
        }
        st->head = head; st->len = len; st->num = num; st->tot = tot;
        st->serial_calls += 1;
    }
}

__global__ void k_ae_query(const AeBucket *b, int s, const int32_t *ids, int n, int32_t *que) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = (uint32_t)ids[i];
    const AeBucket bk = b[v % (uint32_t)s];
    int32_t q = (int32_t)v;
    for (int u = kW - 1; u >= 0; --u)                   // :46-50 returns at the first slot that matches and is hot
        if (bk.cnt[u] != 0.0f && bk.val[u] == v && bk.dic[u]) q = -bk.dic[u];
    que[i] = q;
}

__global__ void k_ae_mark(const AeBucket *b, int s, int lim, uint32_t *mark) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < s * kW) {
        const int32_t d = b[j >> 2].dic[j & 3];
        if (d > 0 && d < lim) mark[d] = 1;
    }
}

// one workgroup, ordered compaction of the unused slot numbers 1 .. lim-1 (:61-75, intended semantics: cafe_hip.h)
__global__ void __launch_bounds__(1024) k_ae_rebuild(const uint32_t *mark, uint32_t *fifo, AeState *st, int lim) {
    __shared__ int base, wsum[16];
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int c = 1; c < lim; c += 1024) {
        const int i = c + threadIdx.x;
        const int f = (i < lim && !mark[i]) ? 1 : 0;
        const unsigned long long m = __ballot(f);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) wsum[w] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int q = 0; q < w; ++q) off += wsum[q];
        if (f) fifo[off + __popcll(m & ((1ull << lane) - 1))] = (uint32_t)i;
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int q = 0; q < 16; ++q) t += wsum[q]; base += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { st->head = 0; st->len = (uint32_t)base; }
}

int ae_bits(int s) { int b = 1; while ((1ll << b) < s) ++b; return b; }

int ae_reserve(cafe_ae_sketch *s, int n) {
    const size_t N = (size_t)std::max(n, 1);
    if (s->keys.reserve(8 * N + 16) || s->pos.reserve(8 * N + 16) || s->att.reserve(4 * N + 16) || s->rank.reserve(4 * N + 16) ||
        s->stage.reserve(sizeof(AeBucket) * N + 16)) return 1;
    size_t a = 0, b = 0;
    uint32_t *k0 = s->keys.as<uint32_t>(), *p0 = s->pos.as<uint32_t>();
    CAFE_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, a, k0, k0 + N, p0, p0 + N, N, 0, ae_bits(s->s), (hipStream_t)0));
    CAFE_HIP_CHECK(rocprim::exclusive_scan(nullptr, b, s->att.as<uint32_t>(), s->rank.as<uint32_t>(), 0u, N, rocprim::plus<uint32_t>(),
                                           (hipStream_t)0));
    s->tmp_bytes = std::max(a, b);
    return s->tmp.reserve(s->tmp_bytes + 16);
}

}  // namespace

extern "C" int cafe_ae_create(cafe_ae_sketch_t **out, int n, int threshold, double decay_rate) {
    if (!out || n < 2 || threshold < 1) { set_error("cafe_ae_create: need n >= 2 slots and Threshold >= 1 (got %d, %d)", n, threshold); return 1; }
    cafe_ae_sketch *s = new cafe_ae_sketch();
    CAFE_HIP_CHECK(hipGetDevice(&s->device));
    s->s = s->lim = n; s->k = threshold; s->decay_rate = decay_rate; s->cap = (uint32_t)n;
    CAFE_HIP_CHECK(hipMalloc(&s->d_b, sizeof(AeBucket) * (size_t)n));
    CAFE_HIP_CHECK(hipMalloc(&s->d_fifo, 4 * (size_t)n));
    CAFE_HIP_CHECK(hipMalloc(&s->d_state, sizeof(AeState)));
    CAFE_HIP_CHECK(hipMemset(s->d_b, 0, sizeof(AeBucket) * (size_t)n));
    CAFE_HIP_CHECK(hipMemset(s->d_fifo, 0, 4 * (size_t)n));
    AeState h;
    memset(&h, 0, sizeof(h));
    h.len = (uint32_t)(n - 1);                          // :36-37: slots 1 .. lim-1
    CAFE_HIP_CHECK(hipMemcpy(s->d_state, &h, sizeof(h), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_ae_fill, dim3((n + 255) / 256), dim3(256), 0, 0, s->d_fifo, n);
    CAFE_HIP_CHECK(hipDeviceSynchronize());
    if (s->blockcnt.reserve(4 * ((size_t)n * kW / 1024 + 2))) return 1;
    *out = s;
    return 0;
}

extern "C" int cafe_ae_destroy(cafe_ae_sketch_t *s) {
    if (!s) return 0;
    (void)hipDeviceSynchronize();
    (void)hipFree(s->d_b); (void)hipFree(s->d_fifo); (void)hipFree(s->d_state);
    DevBuf *all[] = {&s->keys, &s->pos, &s->att, &s->rank, &s->stage, &s->tmp, &s->blockcnt, &s->mark, &s->st_ids, &s->st_vals, &s->st_out};
    for (DevBuf *b : all) b->release();
    delete s;
    return 0;
}

extern "C" int cafe_ae_insert_dev(cafe_ae_sketch_t *s, const int32_t *d_ids, const float *d_vals, int n, int32_t *d_ins,
                                  cafe_stream_t stream) {
    if (n < 0) { set_error("cafe_ae_insert_dev: negative length"); return 1; }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ae_pre, dim3(1), dim3(64), 0, st, s->d_state, s->s, s->k);
    const int dblocks = (s->s * kW + 1023) / 1024;
    uint32_t *bc = s->blockcnt.as<uint32_t>();
    hipLaunchKernelGGL(k_ae_decay_count, dim3(dblocks), dim3(256), 0, st, s->d_b, s->d_state, s->s, s->decay_rate, s->k, bc);
    hipLaunchKernelGGL(k_ae_decay_apply, dim3(dblocks), dim3(256), 0, st, s->d_b, s->d_state, s->s, s->decay_rate, s->k, bc, s->d_fifo, s->cap);
    hipLaunchKernelGGL(k_ae_decay_fin, dim3(1), dim3(64), 0, st, s->d_state);
    if (n == 0) { CAFE_HIP_CHECK(hipGetLastError()); return 0; }
    if (ae_reserve(s, n)) return 1;
    const size_t N = (size_t)n;
    uint32_t *k0 = s->keys.as<uint32_t>(), *k1 = k0 + N, *p0 = s->pos.as<uint32_t>(), *p1 = p0 + N;
    uint32_t *att = s->att.as<uint32_t>(), *rank = s->rank.as<uint32_t>();
    AeBucket *stage = s->stage.as<AeBucket>();
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_ae_keys, dim3(blocks), dim3(256), 0, st, d_ids, n, s->s, k0, p0);
    size_t tb = s->tmp_bytes;
    CAFE_HIP_CHECK(rocprim::radix_sort_pairs(s->tmp.p, tb, k0, k1, p0, p1, N, 0, ae_bits(s->s), st));
    hipLaunchKernelGGL(k_ae_replay, dim3(blocks), dim3(256), 0, st, s->d_b, s->d_state, d_ids, d_vals, n, s->k, k1, p1, att, d_ins, stage);
    tb = s->tmp_bytes;
    CAFE_HIP_CHECK(rocprim::exclusive_scan(s->tmp.p, tb, att, rank, 0u, N, rocprim::plus<uint32_t>(), st));
    hipLaunchKernelGGL(k_ae_commit, dim3(blocks), dim3(256), 0, st, s->d_b, s->d_state, n, k1, att, rank, stage, s->d_fifo, s->cap);
    hipLaunchKernelGGL(k_ae_close, dim3(1), dim3(256), 0, st, s->d_b, s->d_state, d_ids, d_vals, n, s->s, s->k, att, rank, s->d_fifo, s->cap,
                       d_ins);
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cafe_ae_query_dev(cafe_ae_sketch_t *s, const int32_t *d_ids, int n, int32_t *d_que, cafe_stream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_ae_query, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->d_b, s->s, d_ids, n, d_que);
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cafe_ae_update(cafe_ae_sketch_t *s, cafe_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    if (s->mark.reserve(4 * (size_t)s->lim + 16)) return 1;
    uint32_t *mark = s->mark.as<uint32_t>();
    CAFE_HIP_CHECK(hipMemsetAsync(mark, 0, 4 * (size_t)s->lim, st));
    hipLaunchKernelGGL(k_ae_mark, dim3((s->s * kW + 255) / 256), dim3(256), 0, st, s->d_b, s->s, s->lim, mark);
    hipLaunchKernelGGL(k_ae_rebuild, dim3(1), dim3(1024), 0, st, mark, s->d_fifo, s->d_state, s->lim);
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cafe_ae_dims(cafe_ae_sketch_t *s, int64_t *out5) {
    AeState h;
    CAFE_HIP_CHECK(hipDeviceSynchronize());
    CAFE_HIP_CHECK(hipMemcpy(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost));
    out5[0] = s->s; out5[1] = h.len; out5[2] = h.num; out5[3] = h.decays; out5[4] = h.serial_calls;
    return 0;
}

extern "C" int cafe_ae_export(cafe_ae_sketch_t *s, uint32_t *val, float *cnt, int32_t *dic, uint32_t *fifo, double *tot) {
    AeState h;
    CAFE_HIP_CHECK(hipDeviceSynchronize());
    CAFE_HIP_CHECK(hipMemcpy(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost));
    std::vector<AeBucket> hb((size_t)s->s);
    std::vector<uint32_t> hf((size_t)s->cap);
    CAFE_HIP_CHECK(hipMemcpy(hb.data(), s->d_b, sizeof(AeBucket) * hb.size(), hipMemcpyDeviceToHost));
    CAFE_HIP_CHECK(hipMemcpy(hf.data(), s->d_fifo, 4 * hf.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < s->s; ++b)
        for (int i = 0; i < kW; ++i) {
            // the id of an empty slot is undefined in the reference (never initialised, only read behind cnt != 0): 0
            val[b * kW + i] = hb[b].cnt[i] != 0.0f ? hb[b].val[i] : 0u;
            cnt[b * kW + i] = hb[b].cnt[i];
            dic[b * kW + i] = hb[b].dic[i];
        }
    for (uint32_t i = 0; i < h.len; ++i) fifo[i] = hf[(h.head + i) % s->cap];
    *tot = h.tot;
    return 0;
}

extern "C" int cafe_ae_import(cafe_ae_sketch_t *s, const uint32_t *val, const float *cnt, const int32_t *dic, const uint32_t *fifo,
                              int fifo_len, double tot, int num) {
    if (fifo_len < 0 || (uint32_t)fifo_len > s->cap) { set_error("cafe_ae_import: bad free-list length %d", fifo_len); return 1; }
    std::vector<AeBucket> hb((size_t)s->s);
    for (int b = 0; b < s->s; ++b)
        for (int i = 0; i < kW; ++i) {
            if (dic[b * kW + i] < 0 || dic[b * kW + i] >= s->lim) { set_error("cafe_ae_import: slot %d outside [0, %d)", dic[b * kW + i], s->lim); return 1; }
            hb[b].val[i] = val[b * kW + i]; hb[b].cnt[i] = cnt[b * kW + i]; hb[b].dic[i] = dic[b * kW + i];
        }
    std::vector<uint32_t> hf((size_t)s->cap, 0u);
    for (int i = 0; i < fifo_len; ++i) {
        if (fifo[i] == 0 || fifo[i] >= (uint32_t)s->lim) { set_error("cafe_ae_import: free slot %u outside [1, %d)", fifo[i], s->lim); return 1; }
        hf[i] = fifo[i];
    }
    AeState h;
    CAFE_HIP_CHECK(hipDeviceSynchronize());
    CAFE_HIP_CHECK(hipMemcpy(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost));
    h.head = 0; h.len = (uint32_t)fifo_len; h.tot = tot; h.num = num;
    CAFE_HIP_CHECK(hipMemcpy(s->d_b, hb.data(), sizeof(AeBucket) * hb.size(), hipMemcpyHostToDevice));
    CAFE_HIP_CHECK(hipMemcpy(s->d_fifo, hf.data(), 4 * hf.size(), hipMemcpyHostToDevice));
    CAFE_HIP_CHECK(hipMemcpy(s->d_state, &h, sizeof(h), hipMemcpyHostToDevice));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The artifact's C entry points on host pointers (ArtifactEvaluation/embeddings/sketch.cpp:207-228), one instance per
// field index; the names carry a prefix because tricks/sketch.cpp's entry points of the same names (other signatures)
// live in this library too.  Results come back in library-owned storage, as the reference returns static arrays (:7).
// ------------------------------------------------------------------------------------------------
static cafe_ae_sketch *g_ae[64];

static int ae_stage(cafe_ae_sketch *s, const uint32_t *data, const float *v, int len) {
    if (s->st_ids.reserve(4 * (size_t)len + 16) || s->st_out.reserve(4 * (size_t)len + 16) || (v && s->st_vals.reserve(4 * (size_t)len + 16))) return 1;
    CAFE_HIP_CHECK(hipMemcpy(s->st_ids.p, data, 4 * (size_t)len, hipMemcpyHostToDevice));
    if (v) CAFE_HIP_CHECK(hipMemcpy(s->st_vals.p, v, 4 * (size_t)len, hipMemcpyHostToDevice));
    s->h_out.resize((size_t)std::max(len, 1));
    return 0;
}

static void ae_die_if(int rc, const char *what) {
    if (rc) { fprintf(stderr, "libcafe_hip: %s failed: %s\n", what, cafe_last_error()); abort(); }
}

extern "C" void cafe_ae_ref_init(int n, int Threshold, double dr, int idx) {
    ae_die_if(idx < 0 || idx >= 64, "cafe_ae_ref_init: field index outside 0..63");
    if (g_ae[idx]) { cafe_ae_destroy(g_ae[idx]); g_ae[idx] = nullptr; }
    ae_die_if(cafe_ae_create(&g_ae[idx], n, Threshold, dr), "cafe_ae_ref_init");
}

static int *ae_ref_call(int idx, uint32_t *data, float *v, int len, int what) {
    ae_die_if(idx < 0 || idx >= 64 || !g_ae[idx], "cafe_ae_ref: no sketch for this field index (call cafe_ae_ref_init first)");
    cafe_ae_sketch *s = g_ae[idx];
    ae_die_if(ae_stage(s, data, v, len), "cafe_ae_ref: staging");
    const int32_t *ids = s->st_ids.as<int32_t>();
    int32_t *out = s->st_out.as<int32_t>();
    if (what == 0) ae_die_if(cafe_ae_query_dev(s, ids, len, out, nullptr), "cafe_ae_ref_batch_query");
    else ae_die_if(cafe_ae_insert_dev(s, ids, v ? s->st_vals.as<float>() : nullptr, len, out, nullptr), "cafe_ae_ref_batch_insert");
    if (len > 0) ae_die_if(hipMemcpy(s->h_out.data(), out, 4 * (size_t)len, hipMemcpyDeviceToHost) != hipSuccess, "cafe_ae_ref: copy back");
    return s->h_out.data();
}

extern "C" int *cafe_ae_ref_batch_query(uint32_t *data, int len, int idx) { return ae_ref_call(idx, data, nullptr, len, 0); }
extern "C" int *cafe_ae_ref_batch_insert(uint32_t *data, int len, int idx) { return ae_ref_call(idx, data, nullptr, len, 1); }
extern "C" int *cafe_ae_ref_batch_insert_val(uint32_t *data, float *v, int len, int idx) { return ae_ref_call(idx, data, v, len, 1); }
extern "C" void cafe_ae_ref_update(int idx) {
    ae_die_if(idx < 0 || idx >= 64 || !g_ae[idx], "cafe_ae_ref_update: no sketch for this field index");
    ae_die_if(cafe_ae_update(g_ae[idx], nullptr), "cafe_ae_ref_update");
}
