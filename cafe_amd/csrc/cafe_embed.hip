// cafe_embed.hip -- the tensor half of CAFE's embedding layer on MI355X: fused sketch query + hot/cold
// routed row gather (forward), importance scores from the output gradient, cold->hot row migration, and
// the sparse backward + SGD update.  Behaviour follows tricks/sk_embedding_bag.py (cited per kernel).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstring>

#include <algorithm>
#include <chrono>
#include <vector>

#include "cafe_internal.h"

CAFE_KWIN_DECL(cafe_debug_windows_embed)

using namespace cafe;

static inline double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------------
// Scratch of cafe_scores_noalloc_dev: the per-lookup gradient norms that cafe_scores_dev keeps in a workspace it
// allocates.  One copy per GPU, held until the process ends (4 MiB).  `g_noalloc_scratch` is a host-side stand-in for
// the symbol, never a device address: host code uses the per-device pointer hipGetSymbolAddress returned.
// ------------------------------------------------------------------------------------------------
constexpr int kNoAllocMaxLookups = 1 << 20;
constexpr int kNoAllocMaxFields = 128;        // x 64 partial sums per field behind the norms
constexpr int kNoAllocMaxDevices = 64;
static __device__ __align__(256) float g_noalloc_scratch[kNoAllocMaxLookups + kNoAllocMaxFields * 64];
static float *g_noalloc_ptr[kNoAllocMaxDevices];      // [device] -> address of that device's copy, nullptr = not looked up

// once per device, with that device current (called from cafe_tables_create)
static int noalloc_setup(int device) {
    if (device < 0 || device >= kNoAllocMaxDevices) return 0;      // the entry point then reports "no scratch"
    if (g_noalloc_ptr[device]) return 0;
    float *p = nullptr;
    CAFE_HIP_CHECK(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(g_noalloc_scratch)));
    g_noalloc_ptr[device] = p;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// tables descriptor
// ------------------------------------------------------------------------------------------------
extern "C" int cafe_tables_create(cafe_tables_t **out, int num_fields, int dim, int hotn, float *d_w_hot,
                                  float *const *d_w_cold, const int64_t *id_offset, const int32_t *hash_size) {
    if (!out || num_fields <= 0 || dim <= 0 || (dim & 3) || dim > 1024 || hotn <= 0) {
        set_error("cafe_tables_create: need num_fields > 0, hotn > 0 and dim a multiple of 4 up to 1024 (got %d, %d, %d)",
                  num_fields, hotn, dim);
        return 1;
    }
    cafe_tables *t = new cafe_tables();
    t->num_fields = num_fields; t->dim = dim; t->hotn = hotn; t->d_w_hot = d_w_hot;
    CAFE_HIP_CHECK(hipGetDevice(&t->device));
    if (noalloc_setup(t->device)) { delete t; return 1; }
    std::vector<int64_t> base(num_fields + 1, 0);
    for (int f = 0; f < num_fields; ++f) {
        if (hash_size[f] <= 0) { set_error("cafe_tables_create: hash_size[%d] = %d", f, hash_size[f]); delete t; return 1; }
        // global ids are the sketch's 32-bit keys and the sign of a routing word tells hot from cold (sk_embedding_bag.py:
        // 121-123): a field whose ids start at 2^31 or beyond cannot be represented
        if (id_offset[f] < 0 || id_offset[f] >= ((int64_t)1 << 31)) {
            set_error("cafe_tables_create: id_offset[%d] = %lld outside [0, 2^31)", f, (long long)id_offset[f]); delete t; return 1;
        }
        base[f + 1] = base[f] + hash_size[f];
    }
    t->total_cold_rows = base[num_fields];
    CAFE_HIP_CHECK(hipMalloc(&t->d_w_cold, sizeof(float *) * num_fields));
    CAFE_HIP_CHECK(hipMalloc(&t->d_id_offset, 8 * num_fields));
    CAFE_HIP_CHECK(hipMalloc(&t->d_hash_size, 4 * num_fields));
    CAFE_HIP_CHECK(hipMalloc(&t->d_cold_base, 8 * (num_fields + 1)));
    CAFE_HIP_CHECK(hipMemcpy(t->d_w_cold, d_w_cold, sizeof(float *) * num_fields, hipMemcpyHostToDevice));
    CAFE_HIP_CHECK(hipMemcpy(t->d_id_offset, id_offset, 8 * num_fields, hipMemcpyHostToDevice));
    CAFE_HIP_CHECK(hipMemcpy(t->d_hash_size, hash_size, 4 * num_fields, hipMemcpyHostToDevice));
    CAFE_HIP_CHECK(hipMemcpy(t->d_cold_base, base.data(), 8 * (num_fields + 1), hipMemcpyHostToDevice));
    *out = t;
    return 0;
}

extern "C" int cafe_tables_set_timing(cafe_tables_t *t, int on) {
    if (on && !t->ev_t[0][0])
        for (int k = 0; k < 5; ++k)
            for (int j = 0; j < 2; ++j) CAFE_HIP_CHECK(hipEventCreate(&t->ev_t[k][j]));
    t->timing = on ? 1 : 0;
    return 0;
}

extern "C" int cafe_tables_kernel_ms(cafe_tables_t *t, int which, float *ms) {
    if (which < 0 || which > 4 || !t->ev_t[which][0]) { set_error("cafe_tables_kernel_ms: timing was not enabled"); return 1; }
    // a pair that the LAST step did not record (the second kernel of the update when it ran as one kernel; the other form
    // after the rule switched) would return an older step's time
    if (t->ev_epoch[which] != (which == 0 ? t->fwd_epoch : t->bwd_epoch)) {
        set_error("cafe_tables_kernel_ms: timer %d was not recorded by the last step", which);
        return 1;
    }
    CAFE_HIP_CHECK(hipEventSynchronize(t->ev_t[which][1]));
    const hipError_t e = hipEventElapsedTime(ms, t->ev_t[which][0], t->ev_t[which][1]);
    if (e != hipSuccess) {
        // a pair that no step has recorded (the second kernel of the update when updates run as one kernel): not an error
        // of the device, and it must not stay behind as HIP's last error
        (void)hipGetLastError();
        set_error("cafe_tables_kernel_ms: timer %d was not recorded", which);
        return 1;
    }
    return 0;
}

// diagnostics: host time spent enqueueing (forward, update) and waiting for the verdict, microseconds per fused step
extern "C" int cafe_debug_host_times(cafe_sketch_t *s, double *out4) {
    for (int i = 0; i < 4; ++i) { out4[i] = s->host_us[i]; s->host_us[i] = 0; }
    return 0;
}

extern "C" int cafe_tables_destroy(cafe_tables_t *t) {
    if (!t) return 0;
    (void)hipDeviceSynchronize();
    for (int k = 0; k < 5; ++k)
        for (int j = 0; j < 2; ++j)
            if (t->ev_t[k][j]) (void)hipEventDestroy(t->ev_t[k][j]);
    if (t->ev_plan) (void)hipEventDestroy(t->ev_plan);
    if (t->ev_sums) (void)hipEventDestroy(t->ev_sums);
    if (t->ev_rows) (void)hipEventDestroy(t->ev_rows);
    if (t->h_rows) (void)hipHostFree(t->h_rows);
    t->ws_bwd_e.release(); t->ws_bwd_f.release();
    (void)hipFree(t->d_w_cold); (void)hipFree(t->d_id_offset); (void)hipFree(t->d_hash_size); (void)hipFree(t->d_cold_base);
    t->ws_norm.release(); t->ws_psum.release(); t->ws_bwd_a.release(); t->ws_bwd_b.release(); t->ws_bwd_c.release(); t->ws_bwd_d.release();
    t->ws_misc.release();
    delete t;
    return 0;
}

// row of the hot table for a routing word q < 0: sk_embedding_bag.py:134 `abs(dic) % hot_nums`
__device__ inline uint32_t hot_row(int32_t q, int hotn) { return (0u - (uint32_t)q) % (uint32_t)hotn; }

static inline int lanes_per_row(int dim) {
    int v = dim / 4, l = 1;
    while (l < v) l <<= 1;
    return l > 64 ? 64 : l;
}

// ------------------------------------------------------------------------------------------------
// forward: sk_embedding_bag.py:101-113 (query) + 119-152 (two embedding_bag + where), fused.
// LPG lanes cooperate on one lookup: every lane probes the same 32-B val|dic half bucket (one request
// per group), then the group copies the selected row as float4 per lane (coalesced 16 B/lane).
// ------------------------------------------------------------------------------------------------
// ML = the three-tier variant of tricks/sketch_ml_embed.py:154-199: a cold lookup of medium importance adds the row
// of a second hash table (W_median[gid % median_n]) to its cold row.
template <int LPG, bool ML = false, int WL = 64, bool NT = false>
__global__ void __launch_bounds__(256) k_forward(const MainBucket *__restrict__ mt, FastMod fm1,
                                                 const float *__restrict__ w_hot, float *const *__restrict__ w_cold,
                                                 const int64_t *__restrict__ id_offset, const int32_t *__restrict__ hash_size,
                                                 const int64_t *__restrict__ local_ids,
                                                 const int32_t *__restrict__ in_gids, const int32_t *__restrict__ fields,
                                                 int batch, int total, int dim, int hotn,
                                                 int32_t *__restrict__ gids, int32_t *__restrict__ que,
                                                 float *__restrict__ out, const float *__restrict__ w_med = nullptr,
                                                 int median_n = 1, const DevState *__restrict__ st = nullptr, int gflip = 0,
                                                 int32_t *__restrict__ qmed = nullptr) {
    CAFE_KWIN(0);
    constexpr int wl = WL;
    constexpr bool nt = NT;
    // A wave takes 64 consecutive lookups.  First every lane resolves one lookup on its own (id, one 32-B probe of
    // val|dic, routing, source row address): three dependent memory steps for 64 lookups instead of per lookup.  Then
    // the wave copies the 64 rows, 64/LPG rows per instruction (LPG lanes x float4 = one row), eight instructions in
    // flight per lane; the source row address travels by cross-lane read.
    constexpr int RPI = 64 / LPG;                       // rows per wave instruction
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
    const int dim4 = dim >> 2;
    const int lig = lane % LPG, sub = lane / LPG;
    for (int base = wave * wl; base < total; base += waves * wl) {
        const int i = base + lane;
        const float *row = nullptr;
        const float *row2 = nullptr;
        if (i < total && lane < wl) {
            const int f = fields ? fields[i] : i / batch;
            const int32_t g = local_ids ? (int32_t)(local_ids[i] + id_offset[f]) : in_gids[i];      // sk_embedding_bag.py:103
            const uint32_t x = (uint32_t)g;
            const int4 *p = reinterpret_cast<const int4 *>(&mt[fast_hash(x, fm1)]);
            const int4 v = p[0], d = p[1];
            const int32_t val[4] = {v.x, v.y, v.z, v.w}, dic[4] = {d.x, d.y, d.z, d.w};
            const int32_t q = query_bucket(val, dic, x);
            // hot: W_h[|q| % hotn] = W_h[slot]; cold: W_hash[|q| % hash_size] (sk_embedding_bag.py:134-150)
            // (the modulo on the hot side is the reference's `abs(dic) % hot_nums`: a slot is < lim <= hotn, so it only
            // matters for a negative global id, which the reference aliases onto a hot row as well)
            row = q < 0 ? w_hot + (size_t)(hot_row(q, hotn)) * dim : w_cold[f] + (size_t)((uint32_t)q % (uint32_t)hash_size[f]) * dim;
            if (gids) gids[i] = g;
            que[i] = q;
            if constexpr (ML) {
                const int4 cw = p[2];
                const float cnt[4] = {__int_as_float(cw.x), __int_as_float(cw.y), __int_as_float(cw.z), __int_as_float(cw.w)};
                const int32_t m = median_bucket(val, dic, cnt, (uint32_t)p[3].x, gflip, st->thr, x);
                qmed[i] = m;
                if (m < 0) row2 = w_med + (size_t)(x % (uint32_t)median_n) * dim;
            }
        }
        const unsigned long long ra = (unsigned long long)row;
        const int lo = (int)(uint32_t)ra, hi = (int)(uint32_t)(ra >> 32);
        const unsigned long long rb = (unsigned long long)row2;
        const int lo2 = (int)(uint32_t)rb, hi2 = (int)(uint32_t)(rb >> 32);
        const int nrow = min(wl, total - base);
        float4 *obase = reinterpret_cast<float4 *>(out + (size_t)base * dim);
        for (int c0 = 0; c0 < dim4; c0 += LPG) {
            const int c = c0 + lig;
            for (int j = 0; j < nrow; j += 8 * RPI) {
                float4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = j + u * RPI + sub;
                    const unsigned long long a = ((unsigned long long)(uint32_t)__shfl(hi, r & 63, 64) << 32) | (uint32_t)__shfl(lo, r & 63, 64);
                    t[u] = (r < nrow && c < dim4) ? reinterpret_cast<const float4 *>(a)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (ML) {
                        const unsigned long long a2 = ((unsigned long long)(uint32_t)__shfl(hi2, r & 63, 64) << 32) | (uint32_t)__shfl(lo2, r & 63, 64);
                        if (r < nrow && c < dim4 && a2) {
                            const float4 m = reinterpret_cast<const float4 *>(a2)[c];
                            t[u].x += m.x; t[u].y += m.y; t[u].z += m.z; t[u].w += m.w;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = j + u * RPI + sub;
                    if (r < nrow && c < dim4) {
                        if (nt) cafe::store_nt16(&obase[(size_t)r * dim4 + c], t[u]);
                        else cafe::store_through16(&obase[(size_t)r * dim4 + c], t[u]);
                    }
                }
            }
        }
    }
}

// a slot handed out by the sketch (< lim) must be a row of the hot table: the reference aliases a larger slot onto
// another row (`abs(dic) % hot_nums`, sk_embedding_bag.py:134); here that pairing is refused
static int pair_ok(const cafe_sketch_t *s, const cafe_tables_t *t, const char *what) {
    if (s->lim > t->hotn) { set_error("%s: the sketch hands out slots up to %d but the hot table has %d rows", what, s->lim - 1, t->hotn); return 1; }
    return 0;
}

// The update in two kernels (segment sums beside the sketch insert, row updates behind the migration) pays when the
// gradient pass is long enough to be worth hiding and the insert is quiet enough to share the device: measured on
// MI355X, +5 % (cfg3, 109 MB of gradient per step) and +7 % (218 MB); -4 % at 872 MB (the kernels are long, the overlap
// buys little and costs the insert its latency), -3 % when the id stream churns (cfg4: the persistent fix-point kernel
// waits at its grid barrier for workgroups the gradient pass keeps off the compute units), -15 % at D = 16 (14 MB: two
// extra launches, nothing to hide), -4 % when most lookups hit distinct rows (cfg4: every row's sum is written and read
// once more, and there are nearly as many rows as lookups; the split form reports rows / lookups, and above 0.4 it is
// paused for 256 steps).  Evaluated alike by the gather (which queues the partition passes) and the update.
static bool want_split_update(const cafe_sketch_t *s, const cafe_tables_t *t, int total) {
    if (!s->opt_split_update || !s->low_stream) return false;
    if (s->opt_split_update == 2) return true;                   // forced (tests)
    const double grad_mb = (double)total * t->dim * 4.0 / 1e6;
    return grad_mb >= 64.0 && grad_mb <= 512.0 && s->last_moves <= 256 && t->split_pause == 0;
}

static int backward_common(cafe_tables_t *t, const int32_t *d_fields, const int32_t *d_que, const float *d_grad_out,
                           int batch, int total, float lr, hipStream_t st, hipStream_t pst, int part = 3,
                           const uint32_t *skip_rej = nullptr, uint32_t skip_seq = 0, int gstride = 0);

// spec: queued by cafe_backward_update_dev for the forward the caller has named (cafe_forward_hint_dev), behind the update
// kernels of a step whose verdict the host has not seen: the plan then waits for that step's commit kernel on the device
static int forward_common(cafe_sketch_t *s, cafe_tables_t *t, const int64_t *d_local_ids, const int32_t *d_in_gids,
                          const int32_t *d_fields, int batch, long total_l, int32_t *d_gids, int32_t *d_que, float *d_out,
                          hipStream_t st, bool spec = false) {
    if (total_l == 0) return 0;
    if (batch <= 0 || total_l < 0 || total_l > 0x7fffffffL / 4) { set_error("cafe forward: bad size %ld", total_l); return 1; }
    if (pair_ok(s, t, "cafe_forward_dev")) return 1;
    const double t_begin = now_us();
    const int total = (int)total_l;
    const int lpg = lanes_per_row(t->dim);
    // (32 lookups per wave instead of 64: twice the waves, half the copy rounds per wave; cfg3 -1 %)
    constexpr int kFwdWl = 32;
    // The gathered rows leave as NON-TEMPORAL stores: 109 MB per step at D = 128 that nothing in the library reads again;
    // written through the caches (sc1) they push out what the following kernels want there -- the gradient, the hot
    // rows.  cfg3: 0.1824 -> 0.1744 ms per step, the norm kernel behind the gather 29 -> 24.5 us (tools/micro/wr.hip: a
    // gather-copy of 213 k random 512-byte rows takes 26.7 us with nt stores, 34.7 us with sc1).  CAFE_FWD_NT=0: sc1.
    // (the gid form -- a shard's gather, whose rows the all-to-all reads at once -- keeps sc1: the sharded step is 8 % slower
    // with nt there)
    static const int fwd_nt_env = getenv("CAFE_FWD_NT") ? atoi(getenv("CAFE_FWD_NT")) : 1;
    const int fwd_nt = (fwd_nt_env == 1 && !d_local_ids) ? 0 : (fwd_nt_env ? 1 : 0);
    const int fwd_wl = kFwdWl;
    const int blocks = (int)std::min<long>(((long)total + 4 * fwd_wl - 1) / (4 * fwd_wl), 256 * 64);      // a wave per fwd_wl lookups
    FastMod fm1 = make_fastmod(s->n1);
    // Training steps: the plan of the insert that will follow (it reads the ids and the same pre-batch tables the gather
    // reads) runs beside the gather on the sketch's second stream, behind everything queued on `st` so far.  Stopped
    // when forwards come without updates (inference), resumed by the next cafe_backward_update_dev.
    bool plan = false, plan_launched = false;
    if (total >= 32768 && s->opt_overlap && cafe::ensure_streams(s)) return 1;
    if ((d_local_ids || d_in_gids) && s->opt_plan_ahead && s->opt_overlap && !s->opt_serial && s->opt_fuse_fields && s->side_stream &&
        s->side_stream != st && total >= 32768 && total <= cafe::kParMaxIds && t->num_fields <= kMaxSeg) {
        if (s->plan_valid) s->unused_plans++;
        s->plan_valid = 0;
        // (queued ahead of the verdict, real_n is only known as a bound: a reset at the next insert marks the plan stale)
        const bool reset_due = !spec && s->adjust && (double)cafe::real_n_bound(s) > s->lim * 1.2;
        if (s->unused_plans < 2 && !reset_due) {
            if (cafe::par_prepare(s, total)) return 1;
            // What the plan reads must be final: the sketch (the last insert's commit) and the ids.  Normally that means
            // "behind everything queued on st so far".  With option ids_ready the caller promises that the ids were
            // written before this call was made (a data loader's batch, not the output of a kernel still queued on st);
            // if the host has also seen the last insert's snapshot (cafe_backward_update_dev waits for it), nothing is
            // left to wait for: the plan starts at once, beside the previous step's update kernels instead of beside
            // this gather.
            if (spec) {
                // behind the commit kernel of the insert just queued (its completion signal), beside the migration, the
                // row updates and this gather
                CAFE_HIP_CHECK(hipStreamWaitEvent(s->side_stream, s->ev_commit, 0));
                if (cafe::par_plan_ahead(s, d_local_ids, t->d_id_offset, batch, total, s->side_stream, d_in_gids)) return 1;
                plan_launched = true;
            } else if (s->opt_ids_ready && s->call_seq == s->base_seq) {
                if (cafe::par_plan_ahead(s, d_local_ids, t->d_id_offset, batch, total, s->side_stream, d_in_gids)) return 1;
                plan_launched = true;
            } else {
                CAFE_HIP_CHECK(hipEventRecord(s->ev_head_fork, st));
                CAFE_HIP_CHECK(hipStreamWaitEvent(s->side_stream, s->ev_head_fork, 0));
            }
            plan = true;
        }
    }
    // the kernel's own completion signal is the event the update's partition passes wait for on the second stream (no
    // marker packet on `st` between the gather and the norm kernel)
    hipEvent_t fwd_done = s->ev_fwd;
    s->fwd_que = d_que;
    s->fwd_count++;
#define CAFE_FWD(L)                                                                                              \
    if (fwd_nt) CAFE_FWD2(L, true); else CAFE_FWD2(L, false)
#define CAFE_FWD2(L, N)                                                                                          \
    hipExtLaunchKernelGGL((k_forward<L, false, kFwdWl, N>), dim3(blocks), dim3(256), 0, st, nullptr, fwd_done, 0,           \
                          (const MainBucket *)s->d_main, fm1, (const float *)t->d_w_hot, (float *const *)t->d_w_cold, \
                          (const int64_t *)t->d_id_offset, (const int32_t *)t->d_hash_size, d_local_ids, d_in_gids, d_fields, batch, total, \
                          t->dim, t->hotn, d_gids, d_que, d_out, (const float *)nullptr, 1, (const DevState *)nullptr, 0, (int32_t *)nullptr)
    t->fwd_epoch++;
    if (t->timing) { (void)hipEventRecord(t->ev_t[0][0], st); t->ev_epoch[0] = t->fwd_epoch; }
    switch (lpg) {
        case 1: CAFE_FWD(1); break;
        case 2: CAFE_FWD(2); break;
        case 4: CAFE_FWD(4); break;
        case 8: CAFE_FWD(8); break;
        case 16: CAFE_FWD(16); break;
        case 32: CAFE_FWD(32); break;
        default: CAFE_FWD(64); break;
    }
#undef CAFE_FWD
#undef CAFE_FWD2
    if (t->timing) (void)hipEventRecord(t->ev_t[0][1], st);
    CAFE_HIP_CHECK(hipGetLastError());
    if (plan && !plan_launched) {
        if (cafe::par_plan_ahead(s, d_local_ids, t->d_id_offset, batch, total, s->side_stream, d_in_gids)) return 1;
    }
    if (plan && want_split_update(s, t, total)) {
        // The partition passes of the update that will follow read nothing but this gather's routing: they run right
        // behind it on the third stream (in a training step: under the dense model's forward and backward), and the
        // gradient pass of the split update follows them there without a cross-stream hand-over.
        CAFE_HIP_CHECK(hipStreamWaitEvent(s->low_stream, s->ev_fwd, 0));
        if (backward_common(t, d_fields, d_que, nullptr, batch, total, 0.f, st, s->low_stream, 17)) return 1;
        t->bwd_plan_que = d_que;
        t->bwd_plan_fwd = s->fwd_count;
    }
    s->host_us[0] += now_us() - t_begin;
    return 0;
}

extern "C" int cafe_forward_ml_dev(cafe_sketch_t *s, cafe_tables_t *t, const int64_t *d_local_ids, int batch,
                                   const float *d_w_median, int median_n, int32_t *d_gids, int32_t *d_que, int32_t *d_qmed,
                                   float *d_out, cafe_stream_t stream) {
    const long total_l = (long)t->num_fields * batch;
    if (total_l == 0) return 0;
    if (batch <= 0 || total_l > 0x7fffffffL / 4 || median_n <= 0 || !d_w_median || !d_qmed) {
        set_error("cafe_forward_ml_dev: bad arguments");
        return 1;
    }
    const int total = (int)total_l;
    const int blocks = (int)std::min<long>(((long)total + 255) / 256, 256 * 16);
    if (pair_ok(s, t, "cafe_forward_ml_dev")) return 1;
    FastMod fm1 = make_fastmod(s->n1);
    hipStream_t st = (hipStream_t)stream;
#define CAFE_FWD(L)                                                                                                        \
    hipLaunchKernelGGL((k_forward<L, true>), dim3(blocks), dim3(256), 0, st, s->d_main, fm1, t->d_w_hot, t->d_w_cold,      \
                       t->d_id_offset, t->d_hash_size, d_local_ids, nullptr, nullptr, batch, total, t->dim, t->hotn, d_gids, d_que, \
                       d_out, d_w_median, median_n, s->d_state, s->flip, d_qmed)
    switch (lanes_per_row(t->dim)) {
        case 1: CAFE_FWD(1); break;
        case 2: CAFE_FWD(2); break;
        case 4: CAFE_FWD(4); break;
        case 8: CAFE_FWD(8); break;
        case 16: CAFE_FWD(16); break;
        case 32: CAFE_FWD(32); break;
        default: CAFE_FWD(64); break;
    }
#undef CAFE_FWD
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cafe_forward_dev(cafe_sketch_t *s, cafe_tables_t *t, const int64_t *d_local_ids, int batch,
                                int32_t *d_gids, int32_t *d_que, float *d_out, cafe_stream_t stream) {
    cafe_sketch::Hint &h = s->hint;
    if (h.valid) {
        const bool same = h.t == t && h.ids == d_local_ids && h.batch == batch && h.gids == d_gids && h.que == d_que &&
                          h.out == d_out && h.st == (hipStream_t)stream;
        const bool queued = same && h.enqueued;
        h.valid = 0; h.enqueued = 0;
        if (queued) { s->hint_used++; return 0; }       // cafe_backward_update_dev has queued exactly this call
    }
    return forward_common(s, t, d_local_ids, nullptr, nullptr, batch, (long)t->num_fields * batch, d_gids, d_que, d_out,
                          (hipStream_t)stream);
}

extern "C" int cafe_forward_hint_dev(cafe_sketch_t *s, cafe_tables_t *t, const int64_t *d_local_ids, int batch,
                                     int32_t *d_gids, int32_t *d_que, float *d_out, cafe_stream_t stream) {
    cafe_sketch::Hint &h = s->hint;
    h.valid = d_local_ids != nullptr && batch > 0;
    h.enqueued = 0;
    h.t = t; h.ids = d_local_ids; h.batch = batch; h.gids = d_gids; h.que = d_que; h.out = d_out; h.st = (hipStream_t)stream;
    return 0;
}

extern "C" int cafe_forward_hint_stats(cafe_sketch_t *s, int64_t *out2) {
    out2[0] = s->hint_used; out2[1] = s->hint_cancelled;
    return 0;
}

extern "C" int cafe_forward_gid_dev(cafe_sketch_t *s, cafe_tables_t *t, const int32_t *d_gids, const int32_t *d_fields,
                                    int n, int32_t *d_que, float *d_out, cafe_stream_t stream) {
    return forward_common(s, t, nullptr, d_gids, d_fields, 1, (long)n, nullptr, d_que, d_out, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// scores: sk_embedding_bag.py:178-189, intended semantics: the L2 norm of each lookup's own
// output-gradient row, scaled per field so that scores average 1.
// One pass over the gradient: a workgroup takes a contiguous run of one field's rows, writes their norms and the
// run's partial sum; a field's sum is the tree sum of its <= 64 partials (cafe_core.h::field_sum_tree), formed where it is needed (the
// insert's owners, k_scale) -- always the same floats, no second pass over the norms and no extra launch.
// ------------------------------------------------------------------------------------------------
static void norm_geometry(int batch, int *nparts, int *rows) {
    int np = std::min(std::max((batch + 127) / 128, 1), kMaxNormParts);
    *rows = (batch + np - 1) / np;
    *nparts = (batch + *rows - 1) / *rows;
}

template <int LPG>
__global__ void __launch_bounds__(256) k_grad_norm(const float *__restrict__ g, int batch, int rows, int dim,
                                                   float *__restrict__ gn, float *__restrict__ psum) {
    CAFE_KWIN(1);
    constexpr int G = 256 / LPG;                 // row groups of the workgroup
    __shared__ float gs[G];
    const int part = blockIdx.x, f = blockIdx.y, nparts = gridDim.x;
    const int lig = threadIdx.x % LPG, grp = threadIdx.x / LPG;
    const int dim4 = dim >> 2;
    const int r0 = part * rows, r1 = min(r0 + rows, batch);
    const size_t fbase = (size_t)f * batch;
    float run = 0.f;
    // eight rows per step and group; rows past the end are loaded again from the last row and not counted, so that
    // the loads of a step carry no condition and are all in flight together
    constexpr int U = 8;
    for (int r = r0 + grp; r < r1; r += U * G) {
        float acc[U];
        float4 v[U];
        if (dim4 <= LPG) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = min(r + u * G, r1 - 1);
                v[u] = lig < dim4 ? reinterpret_cast<const float4 *>(g + (fbase + rr) * dim)[lig] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc[u] = 0.f;
                acc[u] += v[u].x * v[u].x; acc[u] += v[u].y * v[u].y; acc[u] += v[u].z * v[u].z; acc[u] += v[u].w * v[u].w;
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = min(r + u * G, r1 - 1);
                const float4 *row = reinterpret_cast<const float4 *>(g + (fbase + rr) * dim);
                acc[u] = 0.f;
                for (int c = lig; c < dim4; c += LPG) {
                    const float4 w = row[c];
                    acc[u] += w.x * w.x; acc[u] += w.y * w.y; acc[u] += w.z * w.z; acc[u] += w.w * w.w;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int o = LPG >> 1; o > 0; o >>= 1) acc[u] += __shfl_xor(acc[u], o, LPG);
            const int rr = r + u * G;
            if (rr < r1) {
                const float nv = sqrtf(acc[u]);
                if (lig == 0) gn[fbase + rr] = nv;
                run += nv;
            }
        }
    }
    if (lig == 0) gs[grp] = run;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = 0.f;
        for (int k = threadIdx.x; k < G; k += 64) v += gs[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (threadIdx.x == 0) psum[(size_t)f * nparts + part] = v;
    }
}

// s_i = gn_i * batch / sum (the expression of cafe_core.h::ScoreSrc, the sum formed by field_sum_tree); sums[f] on
// request.  A workgroup covers 256 consecutive lookups, i.e. one or two fields at the usual batch sizes.
__global__ void __launch_bounds__(256) k_scale(const float *__restrict__ gn, const float *__restrict__ psum, int nparts,
                                               int batch, int total, float *__restrict__ scores, float *__restrict__ sums,
                                               int num_fields) {
    __shared__ float fs[4];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * 256, i = i0 + tid;
    const int f_lo = min(i0, max(total - 1, 0)) / batch, f_hi = min(i0 + 255, max(total - 1, 0)) / batch;
    float mine = 0.f;
    const int f_me = i < total ? i / batch : -1;
    for (int base = f_lo; base <= f_hi; base += 4) {
        const int f = base + (tid >> 6);
        const float v = field_sum_tree(psum, nparts, f, f <= f_hi);
        __syncthreads();
        if (f <= f_hi && (tid & 63) == 0) fs[tid >> 6] = v;
        __syncthreads();
        if (f_me >= base && f_me < base + 4) mine = fs[f_me - base];
    }
    if (sums) {
        // field sums for the caller: workgroup 0 walks all fields
        if (blockIdx.x == 0)
            for (int base = 0; base < num_fields; base += 4) {
                const int f = base + (tid >> 6);
                const float v = field_sum_tree(psum, nparts, f, f < num_fields);
                if (f < num_fields && (tid & 63) == 0) sums[f] = v;
            }
    }
    if (scores && i < total) scores[i] = gn[i] * (float)batch / mine;
}

// norms and partial sums of one step: gn [F * batch], psum [F * nparts]
static int norms_common(cafe_tables_t *t, const float *d_grad_out, int batch, float *gn, float *psum, int *nparts_out,
                        hipStream_t st, hipEvent_t done = nullptr) {
    int nparts, rows;
    norm_geometry(batch, &nparts, &rows);
    *nparts_out = nparts;
    if (t->num_fields * batch <= 0) return 0;
    const int lpg = lanes_per_row(t->dim);
    // `done`: the kernel's own completion event (no marker packet on st)
#define CAFE_GN(L) hipExtLaunchKernelGGL(k_grad_norm<L>, dim3(nparts, t->num_fields), dim3(256), 0, st, nullptr, done, 0, d_grad_out, batch, rows, t->dim, gn, psum)
    switch (lpg) {
        case 1: CAFE_GN(1); break;
        case 2: CAFE_GN(2); break;
        case 4: CAFE_GN(4); break;
        case 8: CAFE_GN(8); break;
        case 16: CAFE_GN(16); break;
        case 32: CAFE_GN(32); break;
        default: CAFE_GN(64); break;
    }
#undef CAFE_GN
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

static int scores_common(cafe_tables_t *t, const float *d_grad_out, int batch, float *gn, float *psum, float *d_scores,
                         float *d_sums, hipStream_t st) {
    const int total = t->num_fields * batch;
    if (total <= 0) return 0;
    int nparts;
    if (norms_common(t, d_grad_out, batch, gn, psum, &nparts, st)) return 1;
    hipLaunchKernelGGL(k_scale, dim3((total + 255) / 256), dim3(256), 0, st, gn, psum, nparts, batch,
                       total, d_scores, d_sums, t->num_fields);
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cafe_scores_dev(cafe_tables_t *t, const float *d_grad_out, int batch, float *d_scores, cafe_stream_t stream) {
    if (t->ws_norm.reserve(((size_t)t->num_fields * batch + (size_t)t->num_fields * kMaxNormParts) * 4 + 16)) return 1;
    float *gn = t->ws_norm.as<float>();
    return scores_common(t, d_grad_out, batch, gn, gn + (size_t)t->num_fields * batch, d_scores, nullptr, (hipStream_t)stream);
}

// cafe_scores_dev for callers that cannot allocate (HIP graph capture, tight serving loops): same kernels, the norms
// and partial sums go to this GPU's copy of g_noalloc_scratch.  No hipMalloc/hipFree here or below.
extern "C" int cafe_scores_noalloc_dev(cafe_tables_t *t, const float *d_grad_out, int batch, float *d_scores,
                                       cafe_stream_t stream) {
    const long total = (long)t->num_fields * batch;
    if (batch < 0 || total > kNoAllocMaxLookups || t->num_fields > kNoAllocMaxFields) {
        set_error("cafe_scores_noalloc_dev: %ld lookups in %d fields exceed the fixed scratch of %d lookups, %d fields "
                  "(use cafe_scores_dev)", total, t->num_fields, kNoAllocMaxLookups, kNoAllocMaxFields);
        return 1;
    }
    int dev = -1;
    CAFE_HIP_CHECK(hipGetDevice(&dev));
    if (dev != t->device) { set_error("cafe_scores_noalloc_dev: tables belong to device %d, current is %d", t->device, dev); return 1; }
    float *scratch = dev >= 0 && dev < kNoAllocMaxDevices ? g_noalloc_ptr[dev] : nullptr;
    if (!scratch) { set_error("cafe_scores_noalloc_dev: no scratch on device %d", dev); return 1; }
    return scores_common(t, d_grad_out, batch, scratch, scratch + kNoAllocMaxLookups, d_scores, nullptr, (hipStream_t)stream);
}

extern "C" int cafe_grad_norm_dev(cafe_tables_t *t, const float *d_grad_out, int batch, float *d_norms, float *d_field_sums,
                                  cafe_stream_t stream) {
    if (t->ws_psum.reserve((size_t)t->num_fields * kMaxNormParts * 4 + 16)) return 1;
    return scores_common(t, d_grad_out, batch, d_norms, t->ws_psum.as<float>(), nullptr, d_field_sums, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// migration: sk_embedding_bag.py:199-208.  Newly hot ids are few, so one lane checks each lookup and
// copies the row itself.  Only ids that the sketch now reports hot are copied (SURVEY Q3 fence).
// ------------------------------------------------------------------------------------------------
__global__ void k_migrate(const MainBucket *__restrict__ mt, FastMod fm1, float *__restrict__ w_hot,
                          float *const *__restrict__ w_cold, const int32_t *__restrict__ hash_size,
                          const int64_t *__restrict__ id_offset,
                          const int64_t *__restrict__ local_ids, const int32_t *__restrict__ gids,
                          const int32_t *__restrict__ fields,
                          const int32_t *__restrict__ ins, int batch, int begin, int total, int dim, int hotn) {
    int i = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total || !ins[i]) return;
    const uint32_t x = (uint32_t)gids[i];
    const MainBucket &b = mt[fast_hash(x, fm1)];
    const int32_t q = query_bucket(b.val, b.dic, x);        // the re-query of sk_embedding_bag.py:201-204
    if (q >= 0) return;
    const int f = fields ? fields[i] : i / batch;
    const int64_t local = local_ids ? local_ids[i] : (int64_t)gids[i] - id_offset[f];
    const int64_t src_row = local % (int64_t)hash_size[f];     // LOCAL id: sk_embedding_bag.py:207
    const float4 *src = reinterpret_cast<const float4 *>(w_cold[f] + (size_t)src_row * dim);
    float4 *dst = reinterpret_cast<float4 *>(w_hot + (size_t)hot_row(q, hotn) * dim);
    for (int c = 0; c < (dim >> 2); ++c) dst[c] = src[c];
}

extern "C" int cafe_migrate_dev(cafe_sketch_t *s, cafe_tables_t *t, const int64_t *d_local_ids, const int32_t *d_gids,
                                const int32_t *d_ins, int batch, int first_field, int n_fields, cafe_stream_t stream) {
    if (first_field < 0 || n_fields < 0 || first_field + n_fields > t->num_fields) {
        set_error("cafe_migrate_dev: field range [%d, %d) outside 0..%d", first_field, first_field + n_fields, t->num_fields);
        return 1;
    }
    const int begin = first_field * batch, total = (first_field + n_fields) * batch;
    if (total == begin) return 0;
    if (pair_ok(s, t, "cafe_migrate_dev")) return 1;
    hipLaunchKernelGGL(k_migrate, dim3((total - begin + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->d_main,
                       make_fastmod(s->n1), t->d_w_hot, t->d_w_cold, t->d_hash_size, t->d_id_offset, d_local_ids, d_gids,
                       nullptr, d_ins, batch, begin, total, t->dim, t->hotn);
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cafe_migrate_gid_dev(cafe_sketch_t *s, cafe_tables_t *t, const int32_t *d_gids, const int32_t *d_fields,
                                    const int32_t *d_ins, int n, cafe_stream_t stream) {
    if (n <= 0) return 0;
    if (pair_ok(s, t, "cafe_migrate_gid_dev")) return 1;
    hipLaunchKernelGGL(k_migrate, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->d_main,
                       make_fastmod(s->n1), t->d_w_hot, t->d_w_cold, t->d_hash_size, t->d_id_offset, nullptr, d_gids,
                       d_fields, d_ins, 1, 0, n, t->dim, t->hotn);
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The same two steps for a routing that another sketch generation produced (the artifact's earlier HotSketch,
// cafe_ae.hip: ArtifactEvaluation/embeddings/sk_embedding_bag.py:96-118 gathers from `dic`, :178-182 migrates from the
// re-query): rows chosen by que[] alone.
// ------------------------------------------------------------------------------------------------
template <int LPG>
__global__ void __launch_bounds__(256) k_gather_que(const float *__restrict__ w_hot, float *const *__restrict__ w_cold,
                                                    const int32_t *__restrict__ hash_size, const int32_t *__restrict__ que,
                                                    const int32_t *__restrict__ fields, int batch, int total, int dim, int hotn,
                                                    float *__restrict__ out) {
    // a lane group per lookup: hot W_h[|q| % hotn], cold W_hash[f][|q| % hash_size[f]]
    const int groups = (gridDim.x * blockDim.x) / LPG, gi = (blockIdx.x * blockDim.x + threadIdx.x) / LPG, lig = threadIdx.x % LPG;
    const int dim4 = dim >> 2;
    for (int i = gi; i < total; i += groups) {
        const int32_t q = que[i];
        const int f = fields ? fields[i] : i / batch;
        const float *row = q < 0 ? w_hot + (size_t)hot_row(q, hotn) * dim
                                 : w_cold[f] + (size_t)((uint32_t)q % (uint32_t)hash_size[f]) * dim;
        for (int c = lig; c < dim4; c += LPG)
            reinterpret_cast<float4 *>(out + (size_t)i * dim)[c] = reinterpret_cast<const float4 *>(row)[c];
    }
}

extern "C" int cafe_gather_que_dev(cafe_tables_t *t, const int32_t *d_que, const int32_t *d_fields, int batch, int n,
                                   float *d_out, cafe_stream_t stream) {
    if (n <= 0) return 0;
    if (batch <= 0) { set_error("cafe_gather_que_dev: bad batch %d", batch); return 1; }
    if (!d_fields && n > t->num_fields * batch) { set_error("cafe_gather_que_dev: %d lookups for %d fields of %d", n, t->num_fields, batch); return 1; }
    const int lpg = lanes_per_row(t->dim);
    const int blocks = std::min((n + 256 / lpg - 1) / (256 / lpg), 256 * 32);
#define CAFE_GQ(L) hipLaunchKernelGGL(k_gather_que<L>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)t->d_w_hot, \
                                      (float *const *)t->d_w_cold, (const int32_t *)t->d_hash_size, d_que, d_fields, batch, n, t->dim, t->hotn, d_out)
    switch (lpg) {
        case 1: CAFE_GQ(1); break;
        case 2: CAFE_GQ(2); break;
        case 4: CAFE_GQ(4); break;
        case 8: CAFE_GQ(8); break;
        case 16: CAFE_GQ(16); break;
        case 32: CAFE_GQ(32); break;
        default: CAFE_GQ(64); break;
    }
#undef CAFE_GQ
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

// W_h[|que[i]|] = W_hash[f][local_id[i] % hash_size[f]] for the lookups with ins[i] != 0 whose re-query says hot
__global__ void k_migrate_que(float *__restrict__ w_hot, float *const *__restrict__ w_cold, const int32_t *__restrict__ hash_size,
                              const int64_t *__restrict__ local_ids, const int32_t *__restrict__ que,
                              const int32_t *__restrict__ fields, const int32_t *__restrict__ ins, int batch, int total, int dim,
                              int hotn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total || !ins[i]) return;
    const int32_t q = que[i];
    if (q >= 0) return;                                     // SURVEY Q3 fence, as in k_migrate
    const int f = fields ? fields[i] : i / batch;
    const int64_t src_row = local_ids[i] % (int64_t)hash_size[f];
    const float4 *src = reinterpret_cast<const float4 *>(w_cold[f] + (size_t)src_row * dim);
    float4 *dst = reinterpret_cast<float4 *>(w_hot + (size_t)hot_row(q, hotn) * dim);
    for (int c = 0; c < (dim >> 2); ++c) dst[c] = src[c];
}

extern "C" int cafe_migrate_que_dev(cafe_tables_t *t, const int64_t *d_local_ids, const int32_t *d_que, const int32_t *d_fields,
                                    const int32_t *d_ins, int batch, int n, cafe_stream_t stream) {
    if (n <= 0) return 0;
    if (batch <= 0 || (!d_fields && n > t->num_fields * batch)) { set_error("cafe_migrate_que_dev: bad sizes (batch %d, n %d)", batch, n); return 1; }
    hipLaunchKernelGGL(k_migrate_que, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t->d_w_hot, t->d_w_cold, t->d_hash_size,
                       d_local_ids, d_que, d_fields, d_ins, batch, n, t->dim, t->hotn);
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

// The same for the lookups the parallel insert listed as newly hot (cafe_insert_par.hip: got list), LPG lanes per row.
// A count of 0xffffffff means the batch was walked in order (no list): then ins[] is scanned.
template <int LPG>
__global__ void __launch_bounds__(256) k_migrate_list(const MainBucket *__restrict__ mt, FastMod fm1, float *__restrict__ w_hot,
                                                      float *const *__restrict__ w_cold, const int32_t *__restrict__ hash_size,
                                                      const int64_t *__restrict__ local_ids, const int32_t *__restrict__ gids,
                                                      const int32_t *__restrict__ ins, int batch, int total, int dim, int hotn,
                                                      const uint32_t *__restrict__ gotcount, const uint32_t *__restrict__ gotlist) {
    CAFE_KWIN(2);
    const uint32_t count = *gotcount;
    const int groups = (gridDim.x * blockDim.x) / LPG, gi = (blockIdx.x * blockDim.x + threadIdx.x) / LPG, lig = threadIdx.x % LPG;
    const bool scan = count == 0xffffffffu;
    const int n = scan ? total : (int)min(count, (uint32_t)total);
    for (int e = gi; e < n; e += groups) {
        const int i = scan ? e : (int)gotlist[e];
        if (i >= total || !ins[i]) continue;
        const uint32_t x = (uint32_t)gids[i];
        const int4 *p = reinterpret_cast<const int4 *>(&mt[fast_hash(x, fm1)]);
        const int4 v = p[0], d = p[1];
        const int32_t val[4] = {v.x, v.y, v.z, v.w}, dic[4] = {d.x, d.y, d.z, d.w};
        const int32_t q = query_bucket(val, dic, x);            // the re-query of sk_embedding_bag.py:201-204
        if (q >= 0) continue;
        const int f = i / batch;
        const int64_t src_row = local_ids[i] % (int64_t)hash_size[f];     // LOCAL id: sk_embedding_bag.py:207
        const float4 *src = reinterpret_cast<const float4 *>(w_cold[f] + (size_t)src_row * dim);
        float4 *dst = reinterpret_cast<float4 *>(w_hot + (size_t)hot_row(q, hotn) * dim);
        for (int c = lig; c < (dim >> 2); c += LPG) dst[c] = src[c];
    }
}

static int migrate_list(cafe_sketch_t *s, cafe_tables_t *t, const int64_t *d_local_ids, const int32_t *d_gids, const int32_t *d_ins,
                        int batch, hipStream_t st) {
    const int total = t->num_fields * batch;
    const uint32_t *cnt = cafe::par_gotcount(s), *lst = cafe::par_gotlist(s);
    if (!cnt || total <= 0) return 0;
    FastMod fm1 = make_fastmod(s->n1);
#define CAFE_MIG(L)                                                                                                        \
    hipLaunchKernelGGL(k_migrate_list<L>, dim3(64), dim3(256), 0, st, s->d_main, fm1, t->d_w_hot, t->d_w_cold, t->d_hash_size, \
                       d_local_ids, d_gids, d_ins, batch, total, t->dim, t->hotn, cnt, lst)
    switch (lanes_per_row(t->dim)) {
        case 1: CAFE_MIG(1); break;
        case 2: CAFE_MIG(2); break;
        case 4: CAFE_MIG(4); break;
        case 8: CAFE_MIG(8); break;
        case 16: CAFE_MIG(16); break;
        case 32: CAFE_MIG(32); break;
        default: CAFE_MIG(64); break;
    }
#undef CAFE_MIG
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// backward + SGD: the autograd backward of the forward routes each output-gradient row to the table row
// it was read from; torch.optim.SGD then does W -= lr * (sum of the rows' gradients)
// (dlrm_s_pytorch.py:1658).  Here: unified row keys -> sort -> segment sums -> one read-modify-write per
// touched row.  Segments that cross a chunk border finish with float atomics.
// ------------------------------------------------------------------------------------------------
constexpr int kBwdChunk = 2048;     // items per workgroup of the partitioning pass
constexpr int kBwdCap = 256;        // items a workgroup sorts in LDS, one per lane (larger partitions are cut into tiles)
constexpr int kBwdMaxBins = 8192;
constexpr int kBwdTarget = 160;     // average items per partition: a partition without a heavy row fits one tile
constexpr int kBwdCH = 8;           // items a lane group takes at a time
constexpr int kListRegions = 64;    // split update: the row list is kept in this many regions, a counter each (tile t claims in region t mod 64)

__device__ inline uint32_t row_key(const int32_t *__restrict__ fields, const int32_t *__restrict__ que,
                                   const int32_t *__restrict__ hash_size, const int64_t *__restrict__ cold_base, int hotn,
                                   int batch, int i) {
    const int32_t q = que[i];
    if (q < 0) return hot_row(q, hotn);
    const int f = fields ? fields[i] : i / batch;
    return (uint32_t)hotn + (uint32_t)cold_base[f] + (uint32_t)q % (uint32_t)hash_size[f];
}

// partitions [0, Ph) take hot-table rows, [Ph, Ph + Pc) cold-table rows: the hot half of the update does not depend on
// the sketch insert or the migration (a slot handed out during a batch was free when the forward ran, so no gradient of
// this step addresses it), and can run beside them
__device__ inline uint32_t bwd_part(uint32_t k, uint32_t hotn, uint32_t Ph, uint32_t Pc) {
    return k < hotn ? k % Ph : Ph + (k - hotn) % Pc;
}

// Partition the lookups by table row: every row's gradients meet in one workgroup, so closed segments are
// updated with a plain read-modify-write and nothing needs a device-wide sort.
__global__ void __launch_bounds__(1024) k_bwd_assign(int Ph, int Pc, const int32_t *__restrict__ fields, const int32_t *__restrict__ que,
                                                     const int32_t *__restrict__ hash_size, const int64_t *__restrict__ cold_base,
                                                     int hotn, int batch, int total, uint32_t *__restrict__ cnt,
                                                     uint32_t *__restrict__ rank, uint32_t *__restrict__ keys) {
    CAFE_KWIN(3);
    extern __shared__ uint32_t hist[];
    const int P = Ph + Pc;
    for (int b = threadIdx.x; b < P; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const int begin = blockIdx.x * kBwdChunk, end = min(begin + kBwdChunk, total);
    for (int i = begin + threadIdx.x; i < end; i += blockDim.x) {
        const uint32_t k = row_key(fields, que, hash_size, cold_base, hotn, batch, i);
        keys[i] = k;
        rank[i] = atomicAdd(&hist[bwd_part(k, (uint32_t)hotn, (uint32_t)Ph, (uint32_t)Pc)], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < P; b += blockDim.x) {
        const uint32_t c = hist[b];
        if (c) hist[b] = atomicAdd(&cnt[b], c);
    }
    __syncthreads();
    for (int i = begin + threadIdx.x; i < end; i += blockDim.x) rank[i] += hist[bwd_part(keys[i], (uint32_t)hotn, (uint32_t)Ph, (uint32_t)Pc)];
}

// offsets of the partitions, and the list of (partition, tile) work items: one workgroup per <= kBwdCap pairs
__global__ void __launch_bounds__(1024) k_bwd_offsets(int Ph, int P, const uint32_t *__restrict__ cnt, uint32_t *__restrict__ off,
                                                      uint32_t *__restrict__ tiles /* [0] = hot tiles, [1] = all tiles, then (p << 8 | t)... */,
                                                      uint32_t *__restrict__ ecount) {
    CAFE_KWIN(4);
    if (threadIdx.x < kListRegions && ecount) ecount[threadIdx.x] = 0;        // rows of the split update, per list region
    __shared__ uint32_t sh[1024], sh2[1024];
    __shared__ uint32_t carry, carry2;
    if (threadIdx.x == 0) { carry = 0; carry2 = 0; }
    __syncthreads();
    for (int base = 0; base < P; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < P ? cnt[i] : 0u;
        const uint32_t nt = min((v + kBwdCap - 1) / kBwdCap, 256u);   // tile 255 walks whatever is left
        sh[threadIdx.x] = v; sh2[threadIdx.x] = nt;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const uint32_t add = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0u;
            const uint32_t add2 = (int)threadIdx.x >= d ? sh2[threadIdx.x - d] : 0u;
            __syncthreads();
            sh[threadIdx.x] += add; sh2[threadIdx.x] += add2;
            __syncthreads();
        }
        if (i < P) {
            off[i] = carry + sh[threadIdx.x] - v;
            const uint32_t t0 = carry2 + sh2[threadIdx.x] - nt;
            for (uint32_t t = 0; t < nt; ++t) tiles[2 + t0 + t] = ((uint32_t)i << 8) | t;
            if (i == Ph - 1) tiles[0] = t0 + nt;
        }
        __syncthreads();
        if (threadIdx.x == 1023) { carry += sh[1023]; carry2 += sh2[1023]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { off[P] = carry; tiles[1] = carry2; }
}

__global__ void k_bwd_scatter(int Ph, int Pc, int hotn, int total, const uint32_t *__restrict__ keys, const uint32_t *__restrict__ rank,
                              const uint32_t *__restrict__ off, uint32_t *__restrict__ cnt, unsigned long long *__restrict__ items) {
    CAFE_KWIN(5);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Ph + Pc) cnt[i] = 0;                 // ready for the next call (k_bwd_offsets has consumed it)
    if (i >= total) return;
    const uint32_t k = keys[i];
    items[off[bwd_part(k, (uint32_t)hotn, (uint32_t)Ph, (uint32_t)Pc)] + rank[i]] = ((unsigned long long)k << 32) | (uint32_t)i;
}

__device__ inline float *row_ptr(uint32_t key, uint32_t item, const int32_t *__restrict__ fields, float *__restrict__ w_hot,
                                 float *const *__restrict__ w_cold, const int64_t *__restrict__ cold_base, int hotn, int batch,
                                 int dim) {
    if (key < (uint32_t)hotn) return w_hot + (size_t)key * dim;
    const int f = fields ? fields[item] : (int)(item / (uint32_t)batch);
    return w_cold[f] + (size_t)((int64_t)(key - (uint32_t)hotn) - cold_base[f]) * dim;
}

// One workgroup per (partition, tile): sort the tile's (row, lookup) pairs in LDS, then lane groups of LPG lanes walk
// contiguous ranges of the sorted tile kBwdCH pairs at a time.  Everything a chunk needs from HBM -- its gradient rows
// and the table rows of the segments that end inside it -- is requested before anything is used, so a chunk costs one
// memory round trip, not one per row.  Sums run in lookup order (reproducible); each row gets W -= lr * sum.  A row
// whose pairs all lie inside one group's range is updated with a plain read-modify-write; a row spread over several
// groups or tiles finishes with float atomics.
__device__ inline int bwd_rank_of(const unsigned long long *key, int n, unsigned long long mine) {
    int r = 0, j = 0;
    for (; j + 8 <= n; j += 8) {       // four 16-byte broadcast reads in flight per step
        const ulonglong2 k0 = *reinterpret_cast<const ulonglong2 *>(&key[j]);
        const ulonglong2 k1 = *reinterpret_cast<const ulonglong2 *>(&key[j + 2]);
        const ulonglong2 k2 = *reinterpret_cast<const ulonglong2 *>(&key[j + 4]);
        const ulonglong2 k3 = *reinterpret_cast<const ulonglong2 *>(&key[j + 6]);
        r += (k0.x < mine) + (k0.y < mine) + (k1.x < mine) + (k1.y < mine) + (k2.x < mine) + (k2.y < mine) + (k3.x < mine) +
             (k3.y < mine);
    }
    for (; j < n; ++j) r += key[j] < mine;
    return r;
}

#ifdef CAFE_PROF
__device__ unsigned long long g_dbg_bwd[8192 * 8];
#define BWD_STAMP(k) do { if (threadIdx.x == 0) g_dbg_bwd[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#define BWD_VAL(k, v) do { if (threadIdx.x == 0) g_dbg_bwd[blockIdx.x * 8 + (k)] = (v); } while (0)
extern "C" int cafe_debug_read_bwd(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_bwd), sizeof(g_dbg_bwd)) != hipSuccess; }
#else
#define BWD_STAMP(k) do { } while (0)
#define BWD_VAL(k, v) do { } while (0)
#endif

// EMIT: the same walk, but nothing touches the tables: every row's sum goes to sums[entry][dim] and an entry
// {row key, a lookup of the row, 1 plain / 2 atomic, 0} to entries[entry]; k_bwd_apply does the read-modify-writes later.
// The sums do not depend on the sketch insert or on the migration, the row updates do (a migrated row is copied before it
// is updated): split this way the gradient pass runs beside the insert.  Entries live in the partitioned order of the
// lookups: a tile of n lookups with r rows fills the first r of its n entries (a row's entry is its rank among the
// tile's rows) and marks the rest empty.  The grid is persistent (workgroup w walks tiles w, w + gridDim.x, ...): it
// never holds more compute-unit slots than it was given, whatever the number of tiles.
template <int LPG, int VPL, bool EMIT>
__global__ void __launch_bounds__(256) k_bwd_segments(const uint32_t *__restrict__ off, const uint32_t *__restrict__ tiles,
                                                      const unsigned long long *__restrict__ items,
                                                      const int32_t *__restrict__ fields, const float *__restrict__ grad,
                                                      float *__restrict__ w_hot, float *const *__restrict__ w_cold,
                                                      const int64_t *__restrict__ cold_base, int hotn, int batch, int dim, float lr,
                                                      const uint32_t *__restrict__ skip_rej, uint32_t skip_seq, int gstride,
                                                      uint4 *__restrict__ entries, float4 *__restrict__ sums, int ntld,
                                                      uint32_t *__restrict__ ecount, uint4 *__restrict__ list, uint32_t rcap) {
    CAFE_KWIN(6);
    (void)entries;
    __shared__ __attribute__((aligned(16))) unsigned long long key[kBwdCap];
    __shared__ uint4 lent[EMIT ? kBwdCap : 1];        // EMIT: the tile's list entries, flushed to the claimed place at its end
    __shared__ uint32_t s_all, s_lbase;
    __shared__ float4 bpart[2][VPL][256];      // border partial sums: [0] head of a group's range, [1] tail
    __shared__ int bflag[256];
    __shared__ uint16_t srank[EMIT ? kBwdCap : 2];     // EMIT: rows that start before sorted position j
    __shared__ uint32_t wrows[4], s_ebase;
    const int tid = threadIdx.x;
    // queued behind an insert whose verdict the host has not seen yet: if the device rejected it (DevState::rejected_seq
    // names that call) the host issues the step again, and this update with it
    if (skip_rej && *skip_rej == skip_seq) return;
    const uint32_t ntiles = tiles[1];
    uint32_t claim = 0, region = 0;
    for (uint32_t tile_at = blockIdx.x; tile_at < ntiles; tile_at += gridDim.x) {
    // the tile index is kept in 8 bits; a partition with more tiles lets its last workgroup walk the rest
    const uint32_t tl = tiles[2 + tile_at];
    const int p = (int)(tl >> 8), tile = (int)(tl & 255u);
    const uint32_t o0 = off[p];
    const int cnt = (int)(off[p + 1] - o0);
    const int lig = tid % LPG, grp = tid / LPG;
    constexpr int G = 256 / LPG;
    const int dim4 = dim >> 2;
    // A partition larger than the LDS tile holds a row that very many lookups share.  It is walked tile by tile;
    // a row can then show up in several tiles, so every update of such a partition is an atomic add.
    const bool multi = cnt > kBwdCap;
    const int t_end = tile < 255 ? min(cnt, (tile + 1) * kBwdCap) : cnt;
    BWD_STAMP(0);
    BWD_VAL(7, (unsigned long long)cnt);
    for (int t0 = tile * kBwdCap; t0 < t_end; t0 += kBwdCap) {
        const int n = min(kBwdCap, cnt - t0);
        __syncthreads();
        {
            // one pair per lane: its rank is the number of smaller pairs (pairs are distinct)
            const unsigned long long k = tid < n ? items[o0 + t0 + tid] : ~0ull;
            if (tid < ((n + 1) & ~1)) key[tid] = k;
            __syncthreads();
            const int r = tid < n ? bwd_rank_of(key, n, k) : 0;
            __syncthreads();
            if (tid < n) key[r] = k;
            __syncthreads();
        }
        if constexpr (EMIT) {
            const bool start = tid < n && (tid == 0 || (uint32_t)(key[tid - 1] >> 32) != (uint32_t)(key[tid] >> 32));
            const unsigned long long m = __ballot(start);
            const int lane = tid & 63, wave = tid >> 6;
            if (lane == 0) wrows[wave] = (uint32_t)__popcll(m);
            __syncthreads();
            uint32_t before = 0, all = 0;
            for (int w = 0; w < 4; ++w) { const uint32_t c = wrows[w]; all += c; if (w < wave) before += c; }
            srank[tid] = (uint16_t)(before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)));
            // A row's sum lives at the tile's place in the partitioned order + its rank among the tile's rows (no counter to
            // contend for).  Its LIST entry -- row address, kind, where the sum is: what k_bwd_apply walks -- goes to a place
            // claimed with ONE atomic per tile, issued here and not needed before the tile's end; the entries wait in LDS
            // meanwhile.  The list is kept in kListRegions regions with a counter each: all workgroups claim at the same
            // moment, and a thousand atomics on ONE address took 12 us to drain, during which the wave that issued one could
            // not receive its gradient rows either (returns are in order).  (Until round 3 a kernel of its own compacted
            // the entries between the gradient pass and the row updates: 7.5 us on the step's critical path.)
            if (tid == 0) {
                s_ebase = o0 + (uint32_t)t0; s_all = all;
                region = (tile_at + (uint32_t)(t0 / kBwdCap - tile)) & (kListRegions - 1);
                claim = all ? atomicAdd(&ecount[region], all) : 0u;
            }
            __syncthreads();
        }
        BWD_STAMP(1);
        // Each lane group owns a contiguous range of the sorted tile and carries its sum from chunk to chunk.  A row
        // that lies inside one group's range gets its update there.  A row that crosses group borders is summed through
        // LDS: every group leaves the partial sum of its border segment(s) in LDS, and the group in whose range the row
        // starts adds the following groups' partials in order and issues the one update (atomic only if the partition
        // has several tiles, i.e. the row may also be touched by another workgroup).
        const int span = ((n + G - 1) / G + kBwdCH - 1) / kBwdCH * kBwdCH;
        const int rbegin = grp * span, rend = min(rbegin + span, n);
        const bool head_same = rbegin > 0 && rbegin < n && (uint32_t)(key[rbegin - 1] >> 32) == (uint32_t)(key[rbegin] >> 32);
        const bool tail_same = rbegin < rend && rend < n && (uint32_t)(key[rend] >> 32) == (uint32_t)(key[rend - 1] >> 32);
        if (lig == 0) { bflag[grp] = 0; }
        float4 acc[VPL];
#pragma unroll
        for (int v = 0; v < VPL; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        int seg_start = rbegin;
        float *lead_row = nullptr;
        uint32_t lead_slot = 0xffffffffu;
        const uint32_t ebase = EMIT ? s_ebase : 0u;
        for (int begin = rbegin; begin < rend; begin += kBwdCH) {
            const int end = min(begin + kBwdCH, rend);
            float4 gr[kBwdCH][VPL], wr[EMIT ? 1 : kBwdCH][VPL];
            float *rowp[kBwdCH];
            uint32_t slot_of[kBwdCH];
            unsigned mode = 0;       // per item 3 bits: 0 inside a segment, 1 plain update, 2 atomic update, 3 head partial, 4 tail partial
            int ss = seg_start;
#pragma unroll
            for (int e = 0; e < kBwdCH; ++e) {
                const bool live = begin + e < end;
                const unsigned long long ke = key[min(begin + e, n - 1)];
                const uint32_t i = (uint32_t)ke, k = (uint32_t)(ke >> 32);
                const float4 *g = reinterpret_cast<const float4 *>(grad + (size_t)i * gstride);
                bool last = false;
                unsigned m = 0;
                if (live) {
                    last = begin + e + 1 == rend || (uint32_t)(key[begin + e + 1] >> 32) != k;
                    if (last) {
                        const bool head = ss == rbegin && head_same, tail = begin + e + 1 == rend && tail_same;
                        m = head ? 3u : tail ? 4u : multi ? 2u : 1u;
                        if (head && tail) m = 5u;                    // the whole range is the middle of one row's run
                        if constexpr (EMIT) {
                            // the row starts at ss (in this group's range unless it is a head partial)
                            const uint32_t lrank = srank[ss];
                            slot_of[e] = ebase + lrank;
                            if ((m == 1u || m == 2u || m == 4u) && lig == 0) {
                                lent[lrank] = make_uint4(k, i, (m == 2u || multi) ? 2u : 1u, slot_of[e]);       // row key, a lookup of the row
                            }
                        }
                        ss = begin + e + 1;
                    }
                }
                if constexpr (!EMIT) rowp[e] = last ? row_ptr(k, i, fields, w_hot, w_cold, cold_base, hotn, batch, dim) : nullptr;
                mode |= m << (3 * e);
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    const int c = lig + v * LPG;
                    gr[e][v] = (live && c < dim4) ? (ntld ? cafe::load_nt16(&g[c]) : g[c]) : make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (!EMIT)
                        wr[e][v] = (m == 1u && c < dim4) ? reinterpret_cast<const float4 *>(rowp[e])[c]
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            seg_start = ss;
#pragma unroll
            for (int e = 0; e < kBwdCH; ++e) {
                const unsigned m = (mode >> (3 * e)) & 7u;
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    acc[v].x += gr[e][v].x; acc[v].y += gr[e][v].y; acc[v].z += gr[e][v].z; acc[v].w += gr[e][v].w;
                }
                if (m) {
                    if (m >= 3u) {
                        // border segment: leave the partial sum in LDS.  Flags: 1 head partial, 2 tail partial (this
                        // group leads the run), 4 the head partial covers the whole range (the run goes on)
                        const int slot = m == 4u ? 1 : 0;
#pragma unroll
                        for (int v = 0; v < VPL; ++v) bpart[slot][v][tid] = acc[v];
                        if constexpr (EMIT) { if (m == 4u) lead_slot = slot_of[e]; }
                        else if (m == 4u) lead_row = rowp[e];
                        if (lig == 0) bflag[grp] |= m == 3u ? 1 : m == 4u ? 2 : 5;
                    } else {
#pragma unroll
                        for (int v = 0; v < VPL; ++v) {
                            const int c = lig + v * LPG;
                            if (c < dim4) {
                                if constexpr (EMIT) {
                                    sums[(size_t)slot_of[e] * dim4 + c] = acc[v];
                                } else if (m == 2u) {
#ifndef CAFE_NO_ATOMIC
                                    float *r = rowp[e] + c * 4;
                                    atomicAdd(r + 0, -lr * acc[v].x); atomicAdd(r + 1, -lr * acc[v].y);
                                    atomicAdd(r + 2, -lr * acc[v].z); atomicAdd(r + 3, -lr * acc[v].w);
#endif
                                } else {
                                    float4 w = wr[e][v];
                                    w.x -= lr * acc[v].x; w.y -= lr * acc[v].y; w.z -= lr * acc[v].z; w.w -= lr * acc[v].w;
                                    cafe::store_through16(&reinterpret_cast<float4 *>(rowp[e])[c], w);
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int v = 0; v < VPL; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        __syncthreads();
        if (EMIT ? lead_slot != 0xffffffffu : lead_row != nullptr) {          // uniform over the lane group
            float4 sum[VPL];
#pragma unroll
            for (int v = 0; v < VPL; ++v) sum[v] = bpart[1][v][tid];
            for (int g2 = grp + 1; g2 < G; ++g2) {
                const int f = bflag[g2];
                if (!(f & 1)) break;
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    const float4 q = bpart[0][v][g2 * LPG + lig];
                    sum[v].x += q.x; sum[v].y += q.y; sum[v].z += q.z; sum[v].w += q.w;
                }
                if (!(f & 4)) break;
            }
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                const int c = lig + v * LPG;
                if (c < dim4) {
                    if constexpr (EMIT) {
                        sums[(size_t)lead_slot * dim4 + c] = sum[v];
                    } else if (multi) {
#ifndef CAFE_NO_ATOMIC
                        float *r = lead_row + c * 4;
                        atomicAdd(r + 0, -lr * sum[v].x); atomicAdd(r + 1, -lr * sum[v].y);
                        atomicAdd(r + 2, -lr * sum[v].z); atomicAdd(r + 3, -lr * sum[v].w);
#endif
                    } else {
                        float4 w = reinterpret_cast<const float4 *>(lead_row)[c];
                        w.x -= lr * sum[v].x; w.y -= lr * sum[v].y; w.z -= lr * sum[v].z; w.w -= lr * sum[v].w;
                        cafe::store_through16(&reinterpret_cast<float4 *>(lead_row)[c], w);
                    }
                }
            }
        }
        if constexpr (EMIT) {
            if (tid == 0) s_lbase = region * rcap + claim;
            __syncthreads();
            if (tid < (int)s_all) {
                // the row's address is worked out here, one entry per lane at the tile's end (inside the walk above it cost
                // three dependent loads per row), so the row updates start from it
                const uint4 en = lent[tid];
                const unsigned long long a = (unsigned long long)row_ptr(en.x, en.y, fields, w_hot, w_cold, cold_base, hotn, batch, dim);
                list[s_lbase + tid] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), en.z, en.w);
            }
        }
    }
    __syncthreads();
    }
    BWD_STAMP(2);
}

// Second half of the split update: list entry = {address of the table row (two words), 1 plain / 2 atomic, where the row's sum is};
// W[row] -= lr * sums[entry] (plain read-modify-write;
// float atomics for the rows of a partition that was walked in several tiles, as in the one-kernel form).  A lane group
// per row, grid-stride over the *ecount rows of the list.
template <int LPG, int VPL>
__global__ void __launch_bounds__(256) k_bwd_apply(const uint32_t *__restrict__ ecount, const uint4 *__restrict__ list,
                                                   const float4 *__restrict__ sums, int dim, float lr,
                                                   const uint32_t *__restrict__ skip_rej, uint32_t skip_seq, int nt, uint32_t cap) {
    CAFE_KWIN(7);
    constexpr int G = 256 / LPG;
    const int lig = threadIdx.x % LPG, grp = threadIdx.x / LPG;
    const int dim4 = dim >> 2;
    // the list's regions (k_bwd_segments) as one sequence: prefix of the region counters, then row i lies in the region
    // whose range holds i (`cap` = entries a region has room for)
    __shared__ uint32_t pre[kListRegions + 1];
    if (threadIdx.x < 64) {
        const uint32_t c = ecount[threadIdx.x];
        uint32_t inc = c;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(inc, d, 64); if ((int)threadIdx.x >= d) inc += x; }
        pre[threadIdx.x + 1] = inc;
        if (threadIdx.x == 0) pre[0] = 0;
    }
    const bool skip = skip_rej && *skip_rej == skip_seq;       // see k_bwd_segments
    __syncthreads();
    if (skip) return;
    const uint32_t n = pre[kListRegions];
    for (uint32_t i = blockIdx.x * G + grp; i < n; i += gridDim.x * G) {
        int r = 0;
#pragma unroll
        for (int b = kListRegions / 2; b > 0; b >>= 1) r += (pre[r + b] <= i) ? b : 0;
        const uint4 en = list[(size_t)r * cap + (i - pre[r])];
        float *row = reinterpret_cast<float *>(((unsigned long long)en.y << 32) | en.x);
        float4 sm[VPL], w[VPL];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int c = lig + v * LPG;
            sm[v] = c < dim4 ? ((nt & 2) ? cafe::load_nt16(&sums[(size_t)en.w * dim4 + c]) : sums[(size_t)en.w * dim4 + c]) : make_float4(0.f, 0.f, 0.f, 0.f);
            w[v] = (c < dim4 && en.z == 1u) ? reinterpret_cast<const float4 *>(row)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int c = lig + v * LPG;
            if (c >= dim4) continue;
            if (en.z == 1u) {
                w[v].x -= lr * sm[v].x; w[v].y -= lr * sm[v].y; w[v].z -= lr * sm[v].z; w[v].w -= lr * sm[v].w;
                if (nt & 1) cafe::store_nt16(&reinterpret_cast<float4 *>(row)[c], w[v]);
                else cafe::store_through16(&reinterpret_cast<float4 *>(row)[c], w[v]);
            } else {
#ifndef CAFE_NO_ATOMIC
                float *r = row + c * 4;
                atomicAdd(r + 0, -lr * sm[v].x); atomicAdd(r + 1, -lr * sm[v].y);
                atomicAdd(r + 2, -lr * sm[v].z); atomicAdd(r + 3, -lr * sm[v].w);
#endif
            }
        }
    }
}

// The backward in parts (bits of `part`): 1 the partition passes (they read the routing only) on `pst`; 2 the segment
// kernel (gradient rows in, table rows updated) on `st`, ordered behind t->ev_plan (recorded here on pst) when the
// streams differ.  The split form of 2, for a step whose sketch insert runs meanwhile: 4 the segment sums (gradient rows
// in, sums and entries out; tables untouched) on `st`, behind t->ev_plan; 8 the row updates from those sums on `st`,
// behind t->ev_sums (recorded by part 4 on its stream); 16 with 1: the split form follows (its workspaces are sized and
// the offsets pass zeroes the entry counter).
static int backward_common(cafe_tables_t *t, const int32_t *d_fields, const int32_t *d_que, const float *d_grad_out,
                           int batch, int total, float lr, hipStream_t st, hipStream_t pst, int part,
                           const uint32_t *skip_rej, uint32_t skip_seq, int gstride) {
    if (total <= 0) return 0;
    const int64_t rows = (int64_t)t->hotn + t->total_cold_rows;
    if (rows >= (int64_t)0xFFFFFFFFLL) { set_error("cafe_backward_sgd_dev: more than 2^32 table rows"); return 1; }
    const int lpg = lanes_per_row(t->dim);
    const int vpl = (t->dim / 4 + lpg - 1) / lpg;
    if (vpl > 4) { set_error("cafe_backward_sgd_dev: dim %d too large", t->dim); return 1; }
    // each half is sized for 60 % of the lookups
    const int Ph = std::max(1, std::min(((total * 3 / 5) + kBwdTarget - 1) / kBwdTarget, kBwdMaxBins / 2)), Pc = Ph, P = Ph + Pc;
    const size_t cnt_bytes = 4 * (size_t)kBwdMaxBins + 16;
    // partition counters: zero from the allocation (DevBuf::reserve), kept zero by the scatter pass
    if (t->ws_bwd_a.reserve(cnt_bytes)) return 1;
    const int max_tiles = P + total / kBwdCap + 1;
    if (t->ws_bwd_b.reserve(4 * (size_t)kBwdMaxBins + 16) || t->ws_misc.reserve(4 * (size_t)max_tiles + 32) ||
        t->ws_bwd_c.reserve(8 * (size_t)total) ||
        t->ws_bwd_d.reserve(8 * (size_t)total)) return 1;
    // split form: kListRegions row counters, then the row list in as many regions of rcap entries (a region holds the rows
    // of every 64th tile; tiles beyond a partition's 255th -- a row with more than 65 k lookups -- are spread likewise)
    const uint32_t rcap = (uint32_t)(max_tiles / kListRegions + total / 65536 + 4) * kBwdCap;
    if ((part & 28) && (t->ws_bwd_e.reserve(256 + 16 * (size_t)kListRegions * rcap + 64) || t->ws_bwd_f.reserve(4 * (size_t)total * t->dim + 64))) return 1;
    uint32_t *cnt = t->ws_bwd_a.as<uint32_t>(), *off = t->ws_bwd_b.as<uint32_t>();
    uint32_t *rank = t->ws_bwd_c.as<uint32_t>(), *keys = rank + total;
    unsigned long long *items = t->ws_bwd_d.as<unsigned long long>();
    uint32_t *tiles = t->ws_misc.as<uint32_t>();
    // (the counters are zeroed by the offsets pass of every step; the list starts at byte 256)
    uint32_t *ecount = t->ws_bwd_e.p ? t->ws_bwd_e.as<uint32_t>() : nullptr;
    uint4 *entries = nullptr, *elist = ecount ? reinterpret_cast<uint4 *>(ecount + 64) : nullptr;
    float4 *sums = t->ws_bwd_f.p ? t->ws_bwd_f.as<float4>() : nullptr;
    if (part & 1) {
        // The partition workspaces are one set per tables handle.  Passes queued earlier on another stream (a forward's, on
        // the third stream) may still be running: these wait for them; and whatever partition an earlier forward left is
        // gone (cafe_forward_dev names its own again after this call).
        if (t->ev_plan && t->plan_stream && t->plan_stream != pst) CAFE_HIP_CHECK(hipStreamWaitEvent(pst, t->ev_plan, 0));
        t->bwd_plan_que = nullptr;
        hipLaunchKernelGGL(k_bwd_assign, dim3((total + kBwdChunk - 1) / kBwdChunk), dim3(1024), 4 * (size_t)P, pst, Ph, Pc, d_fields,
                           d_que, t->d_hash_size, t->d_cold_base, t->hotn, batch, total, cnt, rank, keys);
        hipLaunchKernelGGL(k_bwd_offsets, dim3(1), dim3(1024), 0, pst, Ph, P, cnt, off, tiles, ecount);
        hipLaunchKernelGGL(k_bwd_scatter, dim3((std::max(total, P) + 255) / 256), dim3(256), 0, pst, Ph, Pc, t->hotn, total, keys, rank, off,
                           cnt, items);
        if (pst != st || (part & 4)) {
            if (!t->ev_plan) CAFE_HIP_CHECK(hipEventCreateWithFlags(&t->ev_plan, kDeviceEvent));
            CAFE_HIP_CHECK(hipEventRecord(t->ev_plan, pst));
            t->plan_stream = pst;
        }
    }
    if (!(part & 14)) { CAFE_HIP_CHECK(hipGetLastError()); return 0; }
    const int grid = max_tiles;
    const int gs = gstride ? gstride : t->dim;
    // the gradient pass of the split form shares the device with the insert's kernels: a persistent grid of four
    // workgroups per compute unit (1024 .. 2560 measured within 1 % of each other on cfg3).  Four, at 96 vector registers
    // a lane, leave every compute unit room for one workgroup of the insert's persistent tail kernel (128 registers),
    // whose grid barriers need all of its workgroups resident.
    static const int emit_grid = getenv("CAFE_EMIT_GRID") ? atoi(getenv("CAFE_EMIT_GRID")) : 1024;
    static const int seg_ntld = getenv("CAFE_SEG_NTLD") ? atoi(getenv("CAFE_SEG_NTLD")) : 0;       // the gradient's last read: non-temporal
#define CAFE_SEG(L, V, E)                                                                                                \
    hipExtLaunchKernelGGL((k_bwd_segments<L, V, E>), dim3(E ? std::min(grid, emit_grid) : grid), dim3(256), 0, st, nullptr, E ? t->ev_sums : nullptr, 0, \
                          (const uint32_t *)off, (const uint32_t *)tiles, (const unsigned long long *)items, d_fields, d_grad_out, t->d_w_hot, \
                          (float *const *)t->d_w_cold, (const int64_t *)t->d_cold_base, t->hotn, batch, t->dim, lr, skip_rej, skip_seq, gs, entries, sums, \
                          seg_ntld, ecount, elist, rcap)
#define CAFE_SEG_ALL(E)                                                                                                  \
    if (vpl == 1) {                                                                                                      \
        switch (lpg) {                                                                                                   \
            case 1: CAFE_SEG(1, 1, E); break;                                                                            \
            case 2: CAFE_SEG(2, 1, E); break;                                                                            \
            case 4: CAFE_SEG(4, 1, E); break;                                                                            \
            case 8: CAFE_SEG(8, 1, E); break;                                                                            \
            case 16: CAFE_SEG(16, 1, E); break;                                                                          \
            case 32: CAFE_SEG(32, 1, E); break;                                                                          \
            default: CAFE_SEG(64, 1, E); break;                                                                          \
        }                                                                                                                \
    } else if (vpl == 2) {                                                                                               \
        CAFE_SEG(64, 2, E);                                                                                              \
    } else {                                                                                                             \
        CAFE_SEG(64, 4, E);                                                                                              \
    }
    if (part & 2) {
        if (pst != st) CAFE_HIP_CHECK(hipStreamWaitEvent(st, t->ev_plan, 0));
        if (t->timing) { (void)hipEventRecord(t->ev_t[1][0], st); t->ev_epoch[1] = t->bwd_epoch; }
        CAFE_SEG_ALL(false)
        if (t->timing) (void)hipEventRecord(t->ev_t[1][1], st);
    }
    if (part & 4) {
        if (pst != st) CAFE_HIP_CHECK(hipStreamWaitEvent(st, t->ev_plan, 0));
        if (t->timing) { (void)hipEventRecord(t->ev_t[1][0], st); t->ev_epoch[1] = t->bwd_epoch; }
        // (the completion event of the sums and the list rides on the kernel itself: no marker packet between it and the
        // waiting stream)
        if (!t->ev_sums) CAFE_HIP_CHECK(hipEventCreateWithFlags(&t->ev_sums, kDeviceEvent));
        CAFE_SEG_ALL(true)
        if (t->timing) (void)hipEventRecord(t->ev_t[1][1], st);          // the gradient pass alone (as a profiler reports it)
        // the row count for the host's rule: copied to pinned memory behind the list kernel, read by a later step once the
        // event behind the copy reports it landed (never waited for; no new copy while one is in flight)
        if (!t->h_rows) { CAFE_HIP_CHECK(hipHostMalloc((void **)&t->h_rows, 4 * kListRegions, hipHostMallocDefault)); memset(t->h_rows, 0, 4 * kListRegions); }
        if (!t->ev_rows) CAFE_HIP_CHECK(hipEventCreateWithFlags(&t->ev_rows, hipEventDisableTiming));
        if (!t->rows_pending) {
            CAFE_HIP_CHECK(hipMemcpyAsync(t->h_rows, ecount, 4 * kListRegions, hipMemcpyDeviceToHost, st));
            CAFE_HIP_CHECK(hipEventRecord(t->ev_rows, st));
            t->rows_pending = 1;
        }
    }
#undef CAFE_SEG_ALL
#undef CAFE_SEG
    if (part & 8) {
        if (pst != st) CAFE_HIP_CHECK(hipStreamWaitEvent(st, t->ev_sums, 0));
        // a lane group per row; the rows of a step are at most its lookups, usually a fraction of them
        const int G = 256 / lpg;
        const int agrid = std::max(1, std::min((total + G - 1) / G, 4096));
        // bit 0: row stores non-temporal (slower: the next gather reads hot rows), bit 1: the sums are read once (cfg3 -1 %)
        static const int apply_nt = getenv("CAFE_APPLY_NT") ? atoi(getenv("CAFE_APPLY_NT")) : 2;
        if (t->timing) { (void)hipEventRecord(t->ev_t[4][0], st); t->ev_epoch[4] = t->bwd_epoch; }
#define CAFE_APPLY(L, V)                                                                                                 \
    hipLaunchKernelGGL((k_bwd_apply<L, V>), dim3(agrid), dim3(256), 0, st, ecount, elist, sums, t->dim, lr, skip_rej, skip_seq, apply_nt, rcap)
        if (vpl == 1) {
            switch (lpg) {
                case 1: CAFE_APPLY(1, 1); break;
                case 2: CAFE_APPLY(2, 1); break;
                case 4: CAFE_APPLY(4, 1); break;
                case 8: CAFE_APPLY(8, 1); break;
                case 16: CAFE_APPLY(16, 1); break;
                case 32: CAFE_APPLY(32, 1); break;
                default: CAFE_APPLY(64, 1); break;
            }
        } else if (vpl == 2) {
            CAFE_APPLY(64, 2);
        } else {
            CAFE_APPLY(64, 4);
        }
#undef CAFE_APPLY
        if (t->timing) (void)hipEventRecord(t->ev_t[4][1], st);
    }
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cafe_backward_sgd_dev(cafe_tables_t *t, const int32_t *d_gids, const int32_t *d_que,
                                     const float *d_grad_out, int batch, float lr, cafe_stream_t stream) {
    (void)d_gids;
    return backward_common(t, nullptr, d_que, d_grad_out, batch, t->num_fields * batch, lr, (hipStream_t)stream,
                           (hipStream_t)stream);
}

extern "C" int cafe_backward_sgd_gid_dev(cafe_tables_t *t, const int32_t *d_fields, const int32_t *d_que,
                                         const float *d_grad_out, int n, float lr, cafe_stream_t stream) {
    return backward_common(t, d_fields, d_que, d_grad_out, 1, n, lr, (hipStream_t)stream, (hipStream_t)stream);
}

// the same with gradient rows `grad_row_stride` floats apart (a multiple of 4, >= dim): the rows of a packed exchange
// buffer [n][dim + 4] that carries a lookup's score behind its gradient row (one all-to-all instead of two)
extern "C" int cafe_backward_sgd_gid_strided_dev(cafe_tables_t *t, const int32_t *d_fields, const int32_t *d_que,
                                                 const float *d_grad_out, int grad_row_stride, int n, float lr,
                                                 cafe_stream_t stream) {
    if (grad_row_stride < t->dim || (grad_row_stride & 3)) {
        set_error("cafe_backward_sgd_gid_strided_dev: row stride %d must be a multiple of 4 and at least dim = %d", grad_row_stride, t->dim);
        return 1;
    }
    return backward_common(t, d_fields, d_que, d_grad_out, 1, n, lr, (hipStream_t)stream, (hipStream_t)stream, 3, nullptr, 0,
                           grad_row_stride);
}

// scores -> batch_insert_val -> migration -> backward + SGD, in the reference's order (dlrm_s_pytorch.py:1642-1658),
// as one host call.
// per_field != 0: the sketch is fed as the reference feeds it, one batch_insert_val per field (dlrm_s_pytorch.py:
// 734-738): decay and batch counter advance once per field and sketch.cpp:291 is evaluated at every field boundary.
// Without a flip or a reset on an inner boundary, F calls over consecutive parts of the id stream equal ONE in-order
// pass over the whole stream, so the step runs as one parallel insert: the host proves "no flip in these F calls" from
// its own counters; the reset test at the first boundary is made on the exact real_n (the host waits for every step's
// verdict, so its snapshot is current); the inner boundaries are checked by the commit kernel from per-field crossing
// counts BEFORE anything is applied, and a step in which one fires is rejected and issued again field by field.  A
// step that runs while Threshold <= 1 (migration then depends on the state after each field) is issued field
// by field from the start.  per_field == 0: one batch_insert_val for the whole step (decay and reset once per step).
//
// The migration and the SGD update are queued behind the insert before the host waits for the verdict, so the device
// does not idle; both do nothing if the device rejected the insert (the host then repeats them behind the repair).
// One step of the owner-side work, in either layout: lookups in field-major order with per-field ids (the single-GPU
// collection: field = position / batch, scores = norms of the gradient rows scaled per field), or lookups in arbitrary
// order with a field each and their scores given (a shard of the multi-GPU path: gradient rows `gstride` floats apart,
// the score of a lookup at scores_in[i * scores_stride]).
struct StepArgs {
    const int64_t *local_ids = nullptr;
    const int32_t *gids = nullptr, *fields = nullptr, *que = nullptr;
    const float *grad = nullptr;
    int gstride = 0;                    // 0: rows are dim floats apart
    int batch = 1, total = 0;
    float lr = 0.f;
    const float *scores_in = nullptr;
    int scores_stride = 1;
    float *scores_out = nullptr;
    int32_t *ins = nullptr;
    int per_field = 0;
    const char *who = "cafe_backward_update_dev";
};

__global__ void k_strided_copy(const float *__restrict__ src, int stride, int n, float *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[(size_t)i * stride];
}

static int step_common(cafe_sketch_t *s, cafe_tables_t *t, const StepArgs &a, cafe_stream_t stream) {
    const int64_t *d_local_ids = a.local_ids;
    const int32_t *d_gids = a.gids, *d_fields = a.fields, *d_que = a.que;
    const float *d_grad_out = a.grad, *scores_in = a.scores_in;
    const int batch = a.batch, gstride = a.gstride, per_field = a.per_field;
    const float lr = a.lr;
    float *d_scores = a.scores_out;
    int32_t *d_ins = a.ins;
    const int F = t->num_fields;
    const int total = a.total;
    hipStream_t st = (hipStream_t)stream;
    if (total <= 0) return 0;
    if (pair_ok(s, t, a.who)) return 1;
    const bool cadence = per_field && F > 1 && !d_fields;
    // a place for the scores when the caller does not ask for them (the call-by-call paths read them back)
    const size_t norm_floats = (size_t)total + (size_t)F * kMaxNormParts;
    if (t->ws_norm.reserve(norm_floats * 4 + 16)) return 1;
    if (scores_in && !d_scores) d_scores = t->ws_norm.as<float>();
    bool fuse = (scores_in || F <= kMaxSeg) && s->opt_fuse_fields && !s->opt_serial && total <= cafe::kParMaxIds;
    if (fuse) {
        double d = s->decay;
        for (int f = 0; f < (cadence ? F : 1); ++f) {                    // sketch.cpp:284-289 for the calls ahead
            d *= s->alpha;
            if (d > kDecayV) { d /= kDecayV; fuse = false; }             // a flip: call by call
        }
        if (cadence && !cafe::threshold_known_above_one(s)) fuse = false;
    }
    const double t_begin = now_us();
    t->bwd_epoch++;
    if (fuse) {
        // What does not depend on this step's gradients runs on the sketch's second stream, beside the norm kernel: the
        // plan of the insert (ids and pre-batch tables only) and the partition passes of the backward (routing of the
        // forward only).  Everything that was queued on `stream` before this call (the previous step's commit and SGD,
        // this step's forward) is ordered before them.  (Small steps are launch-bound: the cross-stream dependencies
        // cost more than the overlap returns.)
        hipStream_t side = st;
        if (s->opt_overlap && total >= 32768 && cafe::ensure_streams(s)) return 1;
        if (s->opt_overlap && s->side_stream && s->side_stream != st && total >= 32768) side = s->side_stream;
        if (cafe::par_prepare(s, total)) return 1;
        float *gn = t->ws_norm.as<float>(), *psum = gn + total;
        const double decay0 = s->decay;
        const int batch_num0 = s->batch_num;
        bool plan_done = false, sums_done = false;
        // `simple`: a rejected step whose repair must not lean on anything queued earlier -- the forward the caller named
        // (cafe_forward_hint_dev) was queued behind this step and has reused the update's workspaces -- runs on `st` alone
        bool simple = false;
        // the update in two kernels when there is a stream to run the first of them beside the insert
        const bool split = side != st && want_split_update(s, t, total);
        s->split_active = split ? 1 : 0;
        uint32_t rows_seen = 0;
        if (t->rows_pending && hipEventQuery(t->ev_rows) == hipSuccess) {
            for (int k = 0; k < kListRegions; ++k) rows_seen += t->h_rows[k];
            t->rows_pending = 0;
        }
        if (split && s->opt_split_update != 2 && (double)rows_seen > 0.4 * total) {
            t->split_pause = 256;            // (this step still runs split: the gather has queued its partition passes)
        } else if (!split && t->split_pause > 0) {
            t->split_pause--;
        }
        for (int attempt = 0; attempt < 2; ++attempt) {
            if (side != st && !simple) {
                // the second stream continues behind the forward that produced this routing (its own completion event),
                // or, for a routing of unknown origin, behind everything queued on st
                if (attempt == 0 && s->fwd_que == d_que && s->fwd_que) {
                    CAFE_HIP_CHECK(hipStreamWaitEvent(side, s->ev_fwd, 0));
                } else {
                    CAFE_HIP_CHECK(hipEventRecord(s->ev_head_fork, st));
                    CAFE_HIP_CHECK(hipStreamWaitEvent(side, s->ev_head_fork, 0));
                }
            }
            int nparts = 0;
            // the partition passes were made behind the forward that wrote this routing (cafe_forward_dev)
            const bool early = split && attempt == 0 && s->fwd_que == d_que && t->bwd_plan_que == d_que && t->bwd_plan_fwd == s->fwd_count;
            if (early) plan_done = true;
            if (attempt == 0) {
                if (!scores_in) {
                    if (t->timing) { (void)hipEventRecord(t->ev_t[3][0], st); t->ev_epoch[3] = t->bwd_epoch; }
                    if (norms_common(t, d_grad_out, batch, gn, psum, &nparts, st, split ? s->ev_gn : nullptr)) return 1;
                    if (t->timing) (void)hipEventRecord(t->ev_t[3][1], st);
                } else if (split) {
                    CAFE_HIP_CHECK(hipEventRecord(s->ev_gn, st));        // the gradient is complete at this point of `st`
                }
                if (early) {
                    // the gradient pass behind them, once the norm kernel (which read the same gradient) is complete
                    CAFE_HIP_CHECK(hipStreamWaitEvent(s->low_stream, s->ev_gn, 0));
                    if (backward_common(t, d_fields, d_que, d_grad_out, batch, total, lr, s->low_stream, s->low_stream, 4, nullptr, 0, gstride)) return 1;
                    sums_done = true;
                }
            } else if (!scores_in) {
                int rows;
                norm_geometry(batch, &nparts, &rows);
            }
            s->seg_len = cadence ? batch : 0;
            s->score_seg = batch;
            s->verify_mode = 1;
            s->plan_expect = d_local_ids ? (const void *)d_local_ids : (const void *)d_gids;
            s->score_stride = scores_in ? a.scores_stride : 0;
            s->unused_plans = 0;
            if (t->timing) { s->ev_chain[0] = t->ev_t[2][0], s->ev_chain[1] = t->ev_t[2][1]; t->ev_epoch[2] = t->bwd_epoch; }
            // the insert's commit copies the rows of the ids that take a slot itself (sk_embedding_bag.py:199-208)
            s->mig.w_hot = t->d_w_hot; s->mig.w_cold = t->d_w_cold; s->mig.hash_size = t->d_hash_size; s->mig.local_ids = d_local_ids;
            s->mig.batch = batch; s->mig.dim = t->dim; s->mig.hotn = t->hotn;
            s->mig.gids = d_gids; s->mig.fields = d_fields; s->mig.id_offset = t->d_id_offset;
            s->mig_done = 0;
            const int rc = cafe::insert_dev_on(s, d_gids, scores_in ? scores_in : gn, scores_in ? nullptr : psum, nparts, total, d_ins,
                                               d_scores, st, (s->opt_overlap & 1) && !simple ? side : st);
            s->score_stride = 0;
            const uint32_t my_seq = s->call_seq;
            s->mig.w_hot = nullptr;
            s->seg_len = 0; s->score_seg = 0; s->verify_mode = 0; s->plan_expect = nullptr;
            s->ev_chain[0] = s->ev_chain[1] = nullptr;
            if (rc) return 1;
            if (side != st && !plan_done && !simple) {
                // the partition passes of the update (they read the routing only) beside the insert.  (The row updates
                // stay behind the migration: a migrated row is copied before it is updated, and hot rows are not disjoint
                // from migration targets after a reset.  Run beside the insert as a whole, the update slowed the insert's
                // latency-bound kernels by more than it saved; its gradient pass alone -- the split form -- pays.)
                if (backward_common(t, d_fields, d_que, d_grad_out, batch, total, lr, st, side, split ? 17 : 1, nullptr, 0, gstride)) return 1;
                plan_done = true;
                if (split) {
                    // The gradient pass (segment sums; tables untouched) on the low-priority stream: behind the norm
                    // kernel (which read the same gradient: it is complete) and the partition passes.  It runs while the
                    // insert's latency-bound kernels leave the memory system idle; the row updates follow the migration.
                    CAFE_HIP_CHECK(hipStreamWaitEvent(s->low_stream, s->ev_gn, 0));
                    if (backward_common(t, d_fields, d_que, d_grad_out, batch, total, lr, s->low_stream, side, 4, nullptr, 0, gstride)) return 1;
                    sums_done = true;
                }
            }
            if (!s->mig_done) {
                if (d_local_ids ? migrate_list(s, t, d_local_ids, d_gids, d_ins, batch, st)
                                : cafe_migrate_gid_dev(s, t, d_gids, d_fields, d_ins, total, stream)) return 1;
            }
            if (backward_common(t, d_fields, d_que, d_grad_out, batch, total, lr, st,
                                simple ? st : sums_done ? s->low_stream : plan_done ? side : st,
                                simple ? 3 : sums_done ? 8 : plan_done ? 2 : 3, &s->d_state->rejected_seq, my_seq, gstride)) return 1;
            // The forward the caller has named as its next call goes in here, before the host waits: the gather behind the
            // row updates on `st`, its plan behind the commit kernel on the second stream, so the device never waits for
            // the host between two steps.  Same batch size only (no workspace may grow under the kernels in flight).
            bool spec = false;
            cafe_sketch::Hint &h = s->hint;
            if (attempt == 0 && d_local_ids && h.valid && !h.enqueued && h.t == t && h.st == st && h.batch == batch && h.que != d_que &&
                h.gids != d_gids) {
                if (forward_common(s, t, h.ids, nullptr, nullptr, batch, (long)total, h.gids, h.que, h.out, st, true)) return 1;
                h.enqueued = 1;
                spec = true;
            }
            bool rejected = false;
            int reason = 0;
            const double t_wait = now_us();
            if (cafe::last_insert_verdict(s, &rejected, &reason)) return 1;
            s->host_us[1] += t_wait - t_begin; s->host_us[2] += now_us() - t_wait; s->host_us[3] += 1;
            if (!rejected) {
                s->force_cap_once = 0;
                if (const char *report = getenv("CAFE_STUCK_REPORT")) {
                    // diagnostic: wait here for everything the step queued (10 s at most); a step that does not finish
                    // is reported to the named file with the state of every stream and event it uses, and the call fails
                    const double t_dbg = now_us();
                    for (;;) {
                        const hipError_t qa = hipStreamQuery(st), b = s->low_stream ? hipStreamQuery(s->low_stream) : hipSuccess;
                        if (qa == hipSuccess && b == hipSuccess) break;
                        if (now_us() - t_dbg > 10e6) {
                            auto name = [](hipError_t e) { return e == hipSuccess ? "done" : e == hipErrorNotReady ? "busy" : "error"; };
                            if (FILE *dbg = fopen(report, "w")) {
                                fprintf(dbg, "step %u does not finish: caller's stream %s, second stream %s, third stream %s | events: norms %s, "
                                        "gather %s, partition passes %s, segment sums %s | split %d sums queued %d partition passes queued %d\n",
                                        s->call_seq, name(qa), s->side_stream ? name(hipStreamQuery(s->side_stream)) : "-", name(b), name(hipEventQuery(s->ev_gn)),
                                        name(hipEventQuery(s->ev_fwd)), t->ev_plan ? name(hipEventQuery(t->ev_plan)) : "-",
                                        t->ev_sums ? name(hipEventQuery(t->ev_sums)) : "-", (int)split, (int)sums_done, (int)plan_done);
                                fclose(dbg);
                            }
                            set_error("%s: step %u does not finish (report in %s)", a.who, s->call_seq, report);
                            return 1;
                        }
                    }
                }
                return 0;
            }
            if (reason == 4) s->inner_reset_rejects++;
            if (spec) {
                // The forward queued ahead read the tables of a step that was not applied, and its plan and partition
                // passes own the workspaces now: it is dropped (the caller's cafe_forward_dev will run as usual), and the
                // repair starts behind everything it queued.
                h.enqueued = 0;
                s->hint_cancelled++;
                s->plan_valid = 0;
                s->fwd_que = nullptr;
                t->bwd_plan_que = nullptr;
                if (s->side_stream) {
                    CAFE_HIP_CHECK(hipEventRecord(s->ev_head_join, s->side_stream));
                    CAFE_HIP_CHECK(hipStreamWaitEvent(st, s->ev_head_join, 0));
                }
                if (s->low_stream) {
                    CAFE_HIP_CHECK(hipEventRecord(s->ev_head_fork, s->low_stream));
                    CAFE_HIP_CHECK(hipStreamWaitEvent(st, s->ev_head_fork, 0));
                }
                simple = true;
            }
            // nothing was applied on the device: take the call counters back as well
            s->decay = decay0;
            s->batch_num = batch_num0;
            s->counters[0]--;
            plan_done = true;                     // the partition passes of the backward stand (they read the routing only)
            // a workspace limit or too few rounds: once more with the largest owners; anything else: call by call
            if (attempt == 0 && (reason == 1 || reason == 2) && !s->force_cap) { s->force_cap_once = 8192; continue; }
            break;
        }
        s->force_cap_once = 0;
        // call by call; the scores are in d_scores (the finish kernel writes them whatever the verdict)
        for (int f = 0; f < (cadence ? F : 1); ++f) {
            const size_t o = cadence ? (size_t)f * batch : 0;
            const int nn = cadence ? batch : total;
            if (cafe_insert_dev(s, d_gids + o, d_scores + o, nn, d_ins + o, stream)) return 1;
            if (d_local_ids ? cafe_migrate_dev(s, t, d_local_ids, d_gids, d_ins, batch, cadence ? f : 0, cadence ? 1 : F, stream)
                            : cafe_migrate_gid_dev(s, t, d_gids, d_fields, d_ins, total, stream)) return 1;
        }
        if (simple) return backward_common(t, d_fields, d_que, d_grad_out, batch, total, lr, st, st, 3, nullptr, 0, gstride);
        return backward_common(t, d_fields, d_que, d_grad_out, batch, total, lr, st, sums_done ? s->low_stream : side, sums_done ? 8 : 2,
                               nullptr, 0, gstride);
    }
    if (scores_in) {
        hipLaunchKernelGGL(k_strided_copy, dim3((total + 255) / 256), dim3(256), 0, st, scores_in, a.scores_stride, total, d_scores);
        CAFE_HIP_CHECK(hipGetLastError());
    } else if (cafe_scores_dev(t, d_grad_out, batch, d_scores, stream)) {
        return 1;
    }
    for (int f = 0; f < (cadence ? F : 1); ++f) {
        const size_t o = cadence ? (size_t)f * batch : 0;
        const int nn = cadence ? batch : total;
        if (cafe_insert_dev(s, d_gids + o, d_scores + o, nn, d_ins + o, stream)) return 1;
        if (d_local_ids ? cafe_migrate_dev(s, t, d_local_ids, d_gids, d_ins, batch, cadence ? f : 0, cadence ? 1 : F, stream)
                        : cafe_migrate_gid_dev(s, t, d_gids, d_fields, d_ins, total, stream)) return 1;
    }
    return backward_common(t, d_fields, d_que, d_grad_out, batch, total, lr, st, st, 3, nullptr, 0, gstride);
}

extern "C" int cafe_backward_update_dev(cafe_sketch_t *s, cafe_tables_t *t, const int64_t *d_local_ids, const int32_t *d_gids,
                                        const int32_t *d_que, const float *d_grad_out, int batch, float lr, float *d_scores,
                                        int32_t *d_ins, int per_field, cafe_stream_t stream) {
    StepArgs a;
    a.local_ids = d_local_ids; a.gids = d_gids; a.que = d_que; a.grad = d_grad_out; a.batch = batch;
    a.total = t->num_fields * batch; a.lr = lr; a.scores_out = d_scores; a.ins = d_ins; a.per_field = per_field;
    return step_common(s, t, a, stream);
}

// The same step for a shard of the multi-GPU path (no reference counterpart): n lookups in arbitrary order,
// each with its global id, its field and its routing word, their gradient rows `row_stride` floats apart in d_rows with
// the lookup's score at column `dim` of its row (read in place).  ONE batch_insert_val over the n ids in the order
// given, migration, SGD -- as one parallel insert with the device's verdict, the plan made beside the forward
// (cafe_forward_gid_dev), the update split by the library's rule, exactly like cafe_backward_update_dev.
extern "C" int cafe_backward_update_gid_dev(cafe_sketch_t *s, cafe_tables_t *t, const int32_t *d_gids, const int32_t *d_fields,
                                            const int32_t *d_que, const float *d_rows, int row_stride, int n, float lr,
                                            int32_t *d_ins, cafe_stream_t stream) {
    if (row_stride < t->dim + 1 || (row_stride & 3)) {
        set_error("cafe_backward_update_gid_dev: row stride %d must be a multiple of 4 and larger than dim = %d", row_stride, t->dim);
        return 1;
    }
    if (n < 0 || !d_fields) { set_error("cafe_backward_update_gid_dev: bad arguments"); return 1; }
    StepArgs a;
    a.gids = d_gids; a.fields = d_fields; a.que = d_que; a.grad = d_rows; a.gstride = row_stride; a.batch = 1; a.total = n;
    a.lr = lr; a.scores_in = d_rows + t->dim; a.scores_stride = row_stride; a.ins = d_ins; a.per_field = 0;
    a.who = "cafe_backward_update_gid_dev";
    return step_common(s, t, a, stream);
}


// ------------------------------------------------------------------------------------------------
// The two permutations of the multi-GPU exchange (cafe_amd/sharded.py; no reference counterpart), one launch each
// instead of a zero-fill and indexed copies: the send buffer of the backward all-to-all, rows [gradient row | score | 0 0 0]
// in destination order, with the score formed here (sk_embedding_bag.py:189 over the global batch: norm * scale / field
// sum); and the rows that came back from the owners, put where their lookups are.
// ------------------------------------------------------------------------------------------------
template <int LPG>
__global__ void __launch_bounds__(256) k_pack_rows(const float *__restrict__ grad, const float *__restrict__ norms,
                                                   const float *__restrict__ fsums, float scale, int batch,
                                                   const int64_t *__restrict__ order, int n, int dim, float *__restrict__ out,
                                                   float *__restrict__ scores) {
    const int groups = (gridDim.x * blockDim.x) / LPG, gi = (blockIdx.x * blockDim.x + threadIdx.x) / LPG, lig = threadIdx.x % LPG;
    const int dim4 = dim >> 2, row4 = dim4 + 1;
    for (int j = gi; j < n; j += groups) {
        const int64_t i = order[j];
        const float4 *src = reinterpret_cast<const float4 *>(grad + (size_t)i * dim);
        float4 *dst = reinterpret_cast<float4 *>(out + (size_t)j * (dim + 4));
        for (int c = lig; c < dim4; c += LPG) cafe::store_through16(&dst[c], src[c]);
        if (lig == 0) {
            const float sc = norms[i] * scale / fsums[i / batch];
            cafe::store_through16(&dst[row4 - 1], make_float4(sc, 0.f, 0.f, 0.f));
            if (scores) scores[i] = sc;
        }
    }
}

template <int LPG>
__global__ void __launch_bounds__(256) k_unpack_rows(const float *__restrict__ back, const int64_t *__restrict__ order, int n,
                                                     int dim, float *__restrict__ out) {
    const int groups = (gridDim.x * blockDim.x) / LPG, gi = (blockIdx.x * blockDim.x + threadIdx.x) / LPG, lig = threadIdx.x % LPG;
    const int dim4 = dim >> 2;
    for (int j = gi; j < n; j += groups) {
        const float4 *src = reinterpret_cast<const float4 *>(back + (size_t)j * dim);
        float4 *dst = reinterpret_cast<float4 *>(out + (size_t)order[j] * dim);
        for (int c = lig; c < dim4; c += LPG) cafe::store_through16(&dst[c], src[c]);
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU route (cafe_amd/sharded.py): owner rank of every lookup, the lookups in stable order by owner, the number
// per owner.  A counting sort over `world` keys in three small launches -- per-block counts, offsets, scatter -- in
// place of a device-wide merge sort and a dozen elementwise launches.  owner = ((gid & 0xffffffff) * 2654435761 >> 15)
// mod world in 64-bit two's-complement arithmetic, the expression of sharded.py::owner_of (every rank and the CPU tests
// must agree on it bit for bit).
// ------------------------------------------------------------------------------------------------
constexpr int kRouteBlock = 1024, kRouteWaves = kRouteBlock / 64, kRouteMaxWorld = 64;

__device__ inline int route_owner(uint32_t gid, int world) {
    const long long p = (long long)((unsigned long long)gid * 2654435761ull);
    long long r = (p >> 15) % (long long)world;
    if (r < 0) r += world;
    return (int)r;
}

// mode 0: counts per (block, owner) -> ws[block * world + owner].  mode 1: the scatter, from the offsets k_route_offsets
// left in the same array (+ the owners' bases behind it)
template <int MODE>
__global__ void __launch_bounds__(kRouteBlock) k_route(const int64_t *__restrict__ local_ids, const int64_t *__restrict__ id_offset,
                                                       int batch, int n, int world, uint32_t *__restrict__ ws, int nblocks,
                                                       int64_t *__restrict__ order, int64_t *__restrict__ packed) {
    __shared__ uint32_t wcnt[kRouteWaves][kRouteMaxWorld];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * kRouteBlock + tid;
    const bool live = i < n;
    const int f = live ? i / batch : 0;
    const uint32_t gid = live ? (uint32_t)(local_ids[i] + id_offset[f]) : 0u;        // sk_embedding_bag.py:103, as int32 bits
    const int own = live ? route_owner(gid, world) : -1;
    unsigned long long mine = 0;
    for (int w = 0; w < world; ++w) {
        const unsigned long long m = __ballot(own == w);
        if (own == w) mine = m;
        if (lane == 0) wcnt[wave][w] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (MODE == 0) {
        if (tid < world) {
            uint32_t c = 0;
            for (int k = 0; k < kRouteWaves; ++k) c += wcnt[k][tid];
            ws[(size_t)blockIdx.x * world + tid] = c;
        }
    } else if (live) {
        uint32_t at = ws[(size_t)nblocks * world + own] + ws[(size_t)blockIdx.x * world + own];
        for (int k = 0; k < wave; ++k) at += wcnt[k][own];
        at += (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
        order[at] = i;
        packed[at] = (int64_t)(((unsigned long long)(uint32_t)f << 32) | (unsigned long long)gid);
    }
}

// one workgroup: per owner the exclusive prefix of the block counts (in place), the owners' totals and bases
__global__ void __launch_bounds__(kRouteBlock) k_route_offsets(uint32_t *__restrict__ ws, int nblocks, int world,
                                                               int64_t *__restrict__ send_counts) {
    __shared__ uint32_t tot[kRouteMaxWorld];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int o = wave; o < world; o += kRouteWaves) {
        uint32_t carry = 0;
        for (int b0 = 0; b0 < nblocks; b0 += 64) {
            const int b = b0 + lane;
            const uint32_t v = b < nblocks ? ws[(size_t)b * world + o] : 0u;
            uint32_t inc = v;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(inc, d, 64); if (lane >= d) inc += x; }
            if (b < nblocks) ws[(size_t)b * world + o] = carry + inc - v;
            carry += __shfl(inc, 63, 64);
        }
        if (lane == 0) tot[o] = carry;
    }
    __syncthreads();
    if (tid < world) {
        uint32_t base = 0;
        for (int k = 0; k < tid; ++k) base += tot[k];
        ws[(size_t)nblocks * world + tid] = base;
        send_counts[tid] = (int64_t)tot[tid];
    }
}

extern "C" int cafe_route_ws_words(int n, int world) {
    const int nblocks = (std::max(n, 1) + kRouteBlock - 1) / kRouteBlock;
    return (nblocks + 1) * std::max(world, 1);
}

extern "C" int cafe_route_dev(const int64_t *d_local_ids, const int64_t *d_id_offset, int num_fields, int batch, int world,
                              uint32_t *d_ws, int64_t *d_order, int64_t *d_packed, int64_t *d_send_counts, cafe_stream_t stream) {
    const long long n = (long long)num_fields * batch;
    if (world < 1 || world > kRouteMaxWorld) { set_error("cafe_route_dev: world size %d outside [1, %d]", world, kRouteMaxWorld); return 1; }
    if (num_fields <= 0 || batch <= 0 || n > 0x7fffffffLL) { set_error("cafe_route_dev: bad sizes (%d fields, batch %d)", num_fields, batch); return 1; }
    const int nblocks = (int)((n + kRouteBlock - 1) / kRouteBlock);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_route<0>, dim3(nblocks), dim3(kRouteBlock), 0, st, d_local_ids, d_id_offset, batch, (int)n, world, d_ws, nblocks,
                       d_order, d_packed);
    hipLaunchKernelGGL(k_route_offsets, dim3(1), dim3(kRouteBlock), 0, st, d_ws, nblocks, world, d_send_counts);
    hipLaunchKernelGGL(k_route<1>, dim3(nblocks), dim3(kRouteBlock), 0, st, d_local_ids, d_id_offset, batch, (int)n, world, d_ws, nblocks,
                       d_order, d_packed);
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void __launch_bounds__(256) k_split_ids(const int64_t *__restrict__ packed, int n, int32_t *__restrict__ gids,
                                                   int32_t *__restrict__ fields) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long v = (unsigned long long)packed[i];
    gids[i] = (int32_t)(uint32_t)v;
    fields[i] = (int32_t)(uint32_t)(v >> 32);
}

extern "C" int cafe_split_ids_dev(const int64_t *d_packed, int n, int32_t *d_gids, int32_t *d_fields, cafe_stream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_split_ids, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_packed, n, d_gids, d_fields);
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cafe_pack_rows_dev(const float *d_grad, const float *d_norms, const float *d_field_sums, float scale, int batch,
                                  const int64_t *d_order, int n, int dim, float *d_out, float *d_scores, cafe_stream_t stream) {
    if (n <= 0) return 0;
    if (dim <= 0 || (dim & 3) || batch <= 0) { set_error("cafe_pack_rows_dev: bad sizes (dim %d, batch %d)", dim, batch); return 1; }
    const int lpg = lanes_per_row(dim);
    const int blocks = std::min((n + 256 / lpg - 1) / (256 / lpg), 256 * 16);
#define CAFE_PK(L) hipLaunchKernelGGL(k_pack_rows<L>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_grad, d_norms, d_field_sums, scale, \
                                      batch, d_order, n, dim, d_out, d_scores)
    switch (lpg) {
        case 1: CAFE_PK(1); break;
        case 2: CAFE_PK(2); break;
        case 4: CAFE_PK(4); break;
        case 8: CAFE_PK(8); break;
        case 16: CAFE_PK(16); break;
        case 32: CAFE_PK(32); break;
        default: CAFE_PK(64); break;
    }
#undef CAFE_PK
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cafe_unpack_rows_dev(const float *d_back, const int64_t *d_order, int n, int dim, float *d_out, cafe_stream_t stream) {
    if (n <= 0) return 0;
    if (dim <= 0 || (dim & 3)) { set_error("cafe_unpack_rows_dev: bad dim %d", dim); return 1; }
    const int lpg = lanes_per_row(dim);
    const int blocks = std::min((n + 256 / lpg - 1) / (256 / lpg), 256 * 16);
#define CAFE_UP(L) hipLaunchKernelGGL(k_unpack_rows<L>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_back, d_order, n, dim, d_out)
    switch (lpg) {
        case 1: CAFE_UP(1); break;
        case 2: CAFE_UP(2); break;
        case 4: CAFE_UP(4); break;
        case 8: CAFE_UP(8); break;
        case 16: CAFE_UP(16); break;
        case 32: CAFE_UP(32); break;
        default: CAFE_UP(64); break;
    }
#undef CAFE_UP
    CAFE_HIP_CHECK(hipGetLastError());
    return 0;
}
