// cafe_internal.h -- host-side handle definitions shared by the translation units of libcafe_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/cafe_hip.h"
#include "cafe_core.h"

namespace cafe {

void set_error(const char *fmt, ...);

#define CAFE_HIP_CHECK(expr)                                                                         \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            cafe::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

// Growable device buffer owned by a handle; grows outside of any kernel's lifetime only
// (hipMalloc/hipFree synchronise, so callers size workspaces before launching).
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T> T *as() { return (T *)p; }
};

struct ParWorkspace;   // parallel insert workspace, cafe_insert_par.hip

}  // namespace cafe

// events that order work of one device only: no timing, device-scope release (no system-scope cache write-back)
constexpr unsigned kDeviceEvent = hipEventDisableTiming | hipEventReleaseToDevice;

struct cafe_sketch {
    int device = 0;
    int lim = 0, n1 = 0, n2 = 0, adjust = 1, lru_first = 1;
    // host-tracked scalars: they depend on the number of calls only (sketch.cpp:282-289)
    double alpha = 1.0, decay = 1.0;
    int batch_num = 0, flip = 0;
    // device state
    cafe::MainBucket *d_main = nullptr;
    cafe::LruBucket *d_lru = nullptr;
    uint32_t *d_fifo = nullptr;
    uint32_t fifo_cap = 1;
    cafe::DevState *d_state = nullptr;
    // options / counters
    int opt_serial = 0, opt_max_rounds = 6, opt_target_items = 448, opt_hot_bypass = 1, opt_hot_table = 0, opt_move_cap = 1 << 20, opt_ids_ready = 0;
    int64_t inner_reset_rejects = 0;     // fused steps the device rejected because sketch.cpp:291 fired at an inner field boundary
    int opt_r1_order = 1;               // round-1 owners start heaviest first when they cannot all be resident (tuning aid: CAFE_R1_ORDER)
    int force_cap = 0;                  // option "owner_cap": items an owner workgroup can hold (0 = chosen per call)
    int force_cap_once = 0;             // ... for the call being issued (a step issued again after a workspace limit)
    int score_seg = 0;                  // > 0: vals are norms, scaled per segment of score_seg ids (ScoreSrc)
    int score_stride = 0;               // > 1: floats between two scores of the call being issued (a column of packed rows)
    hipEvent_t ev_chain[2] = {nullptr, nullptr};    // timing aid: recorded around round 1 .. commit of the next insert
    int64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // workspaces
    cafe::DevBuf ws_reset_keys, ws_reset_keys2, ws_reset_tmp, ws_reset_flags, ws_misc;
    cafe::ParWorkspace *par = nullptr;
    int32_t *h_pinned = nullptr;    // pinned host mirror of DevState, written by the kernel that closes every insert
    int32_t *d_pinned = nullptr;    // the same memory as the device sees it
    hipStream_t last_stream = nullptr;
    // Second stream of the handle.  The part of an insert that does not need the scores (the plan: classification,
    // partition offsets, item records) and the partition passes of the backward run there, beside the score kernel on
    // the caller's stream (cafe_backward_update_dev).
    hipStream_t side_stream = nullptr;
    // Third stream, lowest priority: the gradient pass of the split update (segment sums) beside the insert's kernels
    hipStream_t low_stream = nullptr;
    hipEvent_t ev_gn = nullptr;         // completion of the norm kernel (it read the gradient: the gradient is ready)
    int opt_split_update = 1;
    uint64_t fwd_count = 0;             // gathers issued (cafe_tables::bwd_plan_fwd names the one its partition passes belong to)
    int opt_overlap = 3;
    hipEvent_t ev_head_fork = nullptr, ev_head_join = nullptr;
    hipEvent_t ev_plan = nullptr;       // plan made beside the forward (cafe_forward_dev) is complete
    // the plan cafe_forward_dev made for the insert that follows it: valid for call plan_seq + 1 over the ids at plan_ids
    int plan_valid = 0, plan_n = 0, plan_flip = 0, plan_hot = 0, plan_target = 0, plan_stale = 0;
    uint32_t plan_seq = 0;
    const void *plan_ids = nullptr, *plan_expect = nullptr;   // plan_expect: the ids of the insert being issued
    int opt_plan_ahead = 1, unused_plans = 0;
    int opt_presort = 1, plan_presort = 0;     // the plan leaves the partitions in owner order (k_presort); as the last plan was made
    int split_active = 0;                      // the fused step's update runs split (cafe_backward_update_dev): see launch_plan
    double host_us[4] = {0, 0, 0, 0};   // diagnostics (cafe_debug_host_times)
    long long counters_plan_ahead = 0;
    long long last_seen_batch = -1;
    // real_n as the host knows it without a sync: the last snapshot of the pinned mirror plus the ids of the calls
    // issued since (see real_n_bound in cafe_sketch.hip)
    uint32_t call_seq = 0;              // sequence number of the insert being issued (0 = none yet)
    uint32_t base_seq = 0;              // snapshots older than this are stale (a reset or a state import came later)
    long long base_real_n = 0;          // real_n after call base_seq
    long long base_clk = 0;             // LRU clock after call base_seq
    int cap_hist[32] = {0};             // largest partition of the last snapshots seen (owner capacity choice)
    uint32_t cap_hist_at = 0, cap_seen_seq = 0, cap_regime_seq = 0;
    hipEvent_t ev_commit = nullptr;     // completion of the last parallel insert's closing kernel (k_commit)
    // cafe_forward_hint_dev: the caller has named its next cafe_forward_dev call; cafe_backward_update_dev queues that
    // forward behind its own kernels BEFORE it waits for the insert's verdict, so the device goes on while the host waits
    struct Hint {
        int valid = 0;                  // a next call has been named
        int enqueued = 0;               // ... and it has been queued (cafe_forward_dev with these arguments returns at once)
        cafe_tables *t = nullptr;
        const int64_t *ids = nullptr;
        int batch = 0;
        int32_t *gids = nullptr, *que = nullptr;
        float *out = nullptr;
        hipStream_t st = nullptr;
    } hint;
    int64_t hint_used = 0, hint_cancelled = 0;
    // tables of the step being issued (cafe_backward_update_dev): the insert's commit migrates the rows of the ids that
    // take a slot itself (w_hot == nullptr: a bare sketch call, nothing to migrate); mig_done: the last insert did
    struct Mig {
        float *w_hot = nullptr;
        float *const *w_cold = nullptr;
        const int32_t *hash_size = nullptr;
        const int64_t *local_ids = nullptr;
        const int32_t *gids = nullptr, *fields = nullptr;      // local_ids == nullptr: a field per lookup, local id = gid - offset
        const int64_t *id_offset = nullptr;
        int batch = 0, dim = 0, hotn = 0;
    } mig;
    int mig_done = 0;
    hipEvent_t ev_fwd = nullptr;        // completion of the last cafe_forward_dev gather kernel
    const void *fwd_que = nullptr;      // ... and the routing array it wrote
    uint32_t last_moves = 0;            // moves (LRU -> main) of the newest call the host has a snapshot of
    int64_t clock_renorms = 0;          // times the LRU timestamps were rewritten to ranks (k_clock_renorm)
    int opt_fuse_fields = 1, opt_verify = 1;
    struct Issued { uint32_t seq; int n; };
    Issued issued[64];                  // ring of the most recent calls
    int verify_mode = 0;                // with seg_len: see Geo::verify
    int seg_len = 0;                    // > 0: the call being issued stands for one reference call per seg_len ids
    // Threshold as the host knows it: it changes at flips (host-driven) and resets (computed on the device: unknown
    // until a snapshot of call thr_dirty_seq or later has been read)
    long long fifo_hint = -1;           // free-list length of the newest snapshot (-1: unknown, a reset came later)
    float host_thr = 0.f;
    uint32_t thr_dirty_seq = 0;         // 0 = host_thr is current
};

struct cafe_tables {
    int device = 0;
    int num_fields = 0, dim = 0, hotn = 0;
    float *d_w_hot = nullptr;
    float **d_w_cold = nullptr;      // device array [F] of device pointers
    int64_t *d_id_offset = nullptr;  // device array [F]
    int32_t *d_hash_size = nullptr;  // device array [F]
    int64_t *d_cold_base = nullptr;  // device array [F+1]: first row of field f in the unified cold row space
    int64_t total_cold_rows = 0;
    cafe::DevBuf ws_norm, ws_psum, ws_bwd_a, ws_bwd_b, ws_bwd_c, ws_bwd_d, ws_bwd_e, ws_bwd_f, ws_misc;
    // optional HIP-event timing (bench.py's roofline figures): [0] k_forward, [1] k_bwd_segments, [2] the sketch insert
    // (round 1 .. commit), [3] k_grad_norm
    int timing = 0;
    // [4] the hot-table half of k_bwd_segments when it runs on the second stream
    hipEvent_t ev_t[5][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    hipEvent_t ev_plan = nullptr;    // partition passes of the backward done (recorded on the sketch's side stream)
    hipStream_t plan_stream = nullptr;  // ... the stream it was last recorded on
    hipEvent_t ev_rows = nullptr;    // the copy of the split update's row count to pinned memory has landed
    int rows_pending = 0;
    uint64_t fwd_epoch = 0, bwd_epoch = 0, ev_epoch[5] = {0, 0, 0, 0, 0};     // which step recorded each timer pair
    const void *bwd_plan_que = nullptr; // routing array the partition passes queued by the last gather read
    uint64_t bwd_plan_fwd = 0;
    // rows / lookups of the last split update that has reported (k_bwd_compact's count, copied to pinned memory): a step
    // whose lookups mostly hit distinct rows gains nothing from the split (the sums travel through memory once more)
    uint32_t *h_rows = nullptr;
    int split_pause = 0;                // steps left before the split form is tried again
    hipEvent_t ev_sums = nullptr;    // segment sums of the split update done (recorded on the sketch's low-priority stream)
};

namespace cafe {
constexpr int kParMaxIds = 1 << 21;     // largest batch of the parallel insert (22-bit time field of its items)
// cafe_insert_par.hip
// vals: scores -- or, with psum ([fields][nparts] partial sums per field, seg_len set), per-lookup norms that are scaled
// where they are consumed (d_scores_out then receives the scores).  `pre`: stream for the kernels that do not read
// vals (== st: everything in order on st).
int par_insert(cafe_sketch *s, const int32_t *d_ids, const float *d_vals, const float *d_psum, int nparts, int n,
               int32_t *d_ins, float *d_scores_out, int flip_now, hipStream_t st, hipStream_t pre);
int par_prepare(cafe_sketch *s, int n);          // sizes the workspaces (may allocate: before any kernel of the call)
// (d_local_ids == nullptr: the ids are the global ids at d_gids, complete in memory)
int par_plan_ahead(cafe_sketch *s, const int64_t *d_local_ids, const int64_t *d_id_offset, int batch, int n, hipStream_t side,
                   const int32_t *d_gids = nullptr);
void par_free(cafe_sketch *s);
// what the last parallel insert leaves for the migration: number of newly hot lookups (0xffffffff: scan ins[]) and
// their batch positions
const uint32_t *par_gotcount(cafe_sketch *s);
const uint32_t *par_gotlist(cafe_sketch *s);
// cafe_sketch.hip
int ensure_streams(cafe_sketch *s);          // creates the handle's second and third stream on first use
int launch_serial_insert(cafe_sketch *s, const int32_t *d_ids, const float *d_scores, int n, int32_t *d_ins, hipStream_t st);
long long real_n_bound(cafe_sketch *s);
bool threshold_known_above_one(cafe_sketch *s);
int real_n_exact(cafe_sketch *s, long long *out);
int last_insert_verdict(cafe_sketch *s, bool *rejected, int *reason);
// cafe_insert_dev with the score-independent head on `pre`; the caller has made `pre` wait for whatever of `st` the head
// depends on (the previous insert, the ids) and may have queued the score kernel on `st` behind that point
int insert_dev_on(cafe_sketch *s, const int32_t *d_ids, const float *d_vals, const float *d_psum, int nparts, int n,
                  int32_t *d_ins, float *d_scores_out, hipStream_t st, hipStream_t pre);
}  // namespace cafe
