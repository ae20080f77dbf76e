/*
 * cafe_hip.h -- C ABI of libcafe_hip.so, the MI355X (gfx950) implementation of CAFE's sketch-driven
 * compact embedding layer.  Plain pointers and sizes only; no torch types.
 *
 * Two groups of entry points:
 *
 *  (A) the reference's own sketch ABI (tricks/sketch.cpp:300-327), same names, same argument meaning,
 *      host pointers in / pointer to library-owned result storage out.  A process that does
 *      ctypes.CDLL("libcafe_hip.so") instead of the g++-built sklib.so (dlrm_s_pytorch.py:1402-1403)
 *      gets the GPU sketch behind the calls it already makes.
 *
 *  (B) handle-based entry points on DEVICE pointers and a HIP stream, which is what the Python
 *      module (cafe_amd/sk_embedding_bag.py, mirroring tricks/sk_embedding_bag.py) uses so that ids,
 *      scores and embedding rows never leave HBM.
 *
 * All functions of group (B) return 0 on success, non-zero on error (text via cafe_last_error()).
 * Nothing here falls back to the CPU: without a GPU every compute call fails.
 */
#ifndef CAFE_HIP_H
#define CAFE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cafe_sketch cafe_sketch_t;   /* one HotSketch (the reference's global `ss`, sketch.cpp:297) */
typedef struct cafe_tables cafe_tables_t;   /* hot table + per-field cold tables of one model replica/shard */
typedef void *cafe_stream_t;                /* hipStream_t */

const char *cafe_last_error(void);
int cafe_version(void);

/* ------------------------------------------------------------------------------------------------
 * (A) reference-compatible ABI -- one process-global sketch on the current HIP device
 * ---------------------------------------------------------------------------------------------- */

/* sketch.cpp:309-313 `init`: (re)creates the global sketch.  Unlike the reference it also resets the
 * decay counters, which the reference leaves as process globals (sketch.cpp:11-13). */
void init(int n, int Threshold, int adjust_thres, double alp);
/* sketch.cpp:300-302 / 260-265.  Returns library-owned storage, valid until the next batch_query. */
int *batch_query(uint32_t *data, int len);
/* sketch.cpp:303-305 / 266-280. */
int *batch_insert(uint32_t *data, int len);
/* sketch.cpp:306-308 / 281-296. */
int *batch_insert_val(uint32_t *data, float *v, int len);
/* Bound by tricks/sketch_ml_embed.py:82-84 (MLEmbeddingBag's median tier) but exported by no sketch.cpp of the
 * reference -- defined here (parity unpinned): out[i] = -data[i] when the id is not hot, sits in the main table
 * without a hot slot and its score (pending lazy decay applied, read-only) is >= 0.2 * Threshold
 * (threshold2 = 0.2 * threshold1, sketch_ml_embed.py:66-67); +data[i] otherwise.  The module reads "< 0" (:151). */
int *batch_query_median(uint32_t *data, int len);
/* sketch.cpp:314-320 and 321-327. */
double analyse(uint32_t *data, int len);
double analyse1(uint32_t *data, int len);
/* Called by the reference driver on checkpoint load (dlrm_s_pytorch.py:1510-1511) but absent from
 * tricks/sketch.cpp; semantics of ArtifactEvaluation/embeddings/sketch.cpp:61-75,219-222: rebuild the
 * free list from the hot slots found in the tables. */
void update(void);
/* TOIS_revision/PinSAGE/sketch.cpp:333-402 counterparts (own file format). */
int save_state(const char *path);
int load_state(const char *path);
/* The handle behind group (A), for callers that want to mix both groups. */
cafe_sketch_t *cafe_global_sketch(void);

/* ------------------------------------------------------------------------------------------------
 * (B) handle-based, device pointers
 * ---------------------------------------------------------------------------------------------- */

/* Replaces lib.init (sketch.cpp:309-313, constructor 44-70) with an explicit handle on the current
 * device.  lru_first: how reset() orders equal scores across the two tables; the reference orders by
 * heap address (sketch.cpp:79,96), which at production sizes puts the LRU table first (1). */
int cafe_sketch_create(cafe_sketch_t **out, int lim, int threshold, int adjust_thres, double alpha, int lru_first);
int cafe_sketch_destroy(cafe_sketch_t *s);

/* out[0..3] = n1, n2, lim, free-list length (synchronises the device). */
int cafe_sketch_dims(cafe_sketch_t *s, int32_t *out);
/* thr, {num, real_n, clock, flip, batch_num, resets}, decay (synchronises). */
int cafe_sketch_scalars(cafe_sketch_t *s, float *thr, int32_t *ints6, double *decay);
/* Full state in the oracle's structure-of-arrays layout ([n1*4], [n1], [n2*4], [n2], free list in pop
 * order); host pointers; synchronises.  Used by the parity tests and by save_state. */
int cafe_sketch_export(cafe_sketch_t *s, int32_t *mval, float *mcnt, int32_t *mdic, uint8_t *mflip,
                       int32_t *lval, float *lcnt, int32_t *ldic, int32_t *lt, uint8_t *lflip, uint32_t *fifo);
int cafe_sketch_import(cafe_sketch_t *s, const int32_t *mval, const float *mcnt, const int32_t *mdic,
                       const uint8_t *mflip, const int32_t *lval, const float *lcnt, const int32_t *ldic,
                       const int32_t *lt, const uint8_t *lflip, const uint32_t *fifo, int fifo_len,
                       float thr, const int32_t *ints6, double decay);
int cafe_sketch_save(cafe_sketch_t *s, const char *path);
int cafe_sketch_load(cafe_sketch_t *s, const char *path);
/* Rebuild the free list from the tables (see `update` above). */
int cafe_sketch_rebuild_free_list(cafe_sketch_t *s, cafe_stream_t stream);

/* Options (tuning and test aids; results never depend on them): "serial_insert" (1 = always use the one-lane in-order
 * kernel), "max_rounds" (rounds the fix-point kernel of the parallel insert may take before the batch counts as not
 * settled; default 6), "target_items" (ids per bucket-owner workgroup, default 448), "hot_bypass" (0 = ids that are
 * already hot go through the owner rounds too), "hot_table" (1 = sort-free hot owner with a per-wave LDS slot table),
 * "owner_cap" (1536 / 2048 / 8192 = fixed owner capacity; 0 = chosen from the largest partitions of the last batches),
 * "plan_ahead" (0 = cafe_forward_dev does not plan the insert that follows), "ids_ready" (1 = the id batches handed to
 * cafe_forward_dev are complete in memory before the call is made, so the plan may start before the gather),
 * "split_update" (0 / 1 / 2 = cafe_backward_update_dev updates the tables with one kernel / with two by the library's
 * rule / always with two: segment sums on a third stream beside the sketch insert, row updates behind the migration), "overlap" (bit 0: the plan of an insert,
 * bit 1: the update's partition passes run on the handle's second stream; 0 = one stream), "fuse_fields" (0 =
 * cafe_backward_update_dev with per_field always issues field by field), "verify", "presort" (0 = the plan of an insert does not leave the partitions in owner order: the round-1 owners sort for
 * themselves), "move_cap" (moves one main
 * partition accepts per round, at most 512: a test aid), "exact_reset" (accepted and ignored: the reset decision is
 * always made on the exact real_n, on the device). */
int cafe_sketch_set_option(cafe_sketch_t *s, const char *name, int value);
/* Counters since creation: [0] insert calls, [1] calls finished by the parallel path, [2] calls given
 * to the in-order kernel, [3] fix-point rounds used in total, [4] resets, [5] largest round count, [6] call
 * boundaries at which the reference would have reset and this library did not: 0 by construction (the test of
 * sketch.cpp:291 is made on the device's exact real_n at every boundary, before anything is applied), kept as a
 * check the tests assert, [7] times the host waited for a snapshot of the device state (once per fused step). */
int cafe_sketch_counters(cafe_sketch_t *s, int64_t *out8);
/* Why batches left the parallel path (given to the in-order kernel, or -- cafe_backward_update_dev -- rejected before
 * anything was applied and issued again), since creation: [0] a workspace limit (partition larger than an owner can
 * sort, move list full, clock range), [1] the fix-point rounds that were enqueued did not settle the batch, [2] the
 * free list ran dry or was refilled inside the batch, [3] fused steps rejected because the reset test of
 * sketch.cpp:291 fired at an inner field boundary (such a step is then issued field by field; this is how the
 * per-field reset cadence stays exact, not a failure). */
int cafe_sketch_fallback_reasons(cafe_sketch_t *s, int64_t *out4);
/* Lifetime of the LRU clock (sketch.cpp:17 `t`, one tick per main-table miss): the reference's victim search starts
 * from the constant 1e9 (sketch.cpp:159) and its int clock wraps at 2^31, so it is undefined from 1e9 ticks on (about
 * 10^4 Criteo-Terabyte steps).  Only the order of the four timestamps inside an LRU bucket matters; when the clock
 * would pass 2^29 within the next call, this library rewrites every bucket's timestamps to their ranks 1..4 (0 stays
 * 0) and sets the clock to 4 before the call.  ins[], que[] and every other state word stay exactly the reference's;
 * exported LRU timestamps differ from the reference's absolute values from that point on (same order).  *out = number
 * of such rewrites since creation.  cafe_sketch_import rejects states whose slots, free-list entries or timestamps
 * are out of range (slots index rows of the hot table). */
int cafe_sketch_clock_renorms(cafe_sketch_t *s, int64_t *out);

/* sketch.cpp:260-265 batch_query: d_que[i] = -slot if d_ids[i] holds a hot slot, else d_ids[i]. */
int cafe_query_dev(cafe_sketch_t *s, const int32_t *d_ids, int n, int32_t *d_que, cafe_stream_t stream);
/* batch_query_median (group A) on device pointers. */
int cafe_query_median_dev(cafe_sketch_t *s, const int32_t *d_ids, int n, int32_t *d_out, cafe_stream_t stream);
/* sketch.cpp:281-296 batch_insert_val (d_scores != NULL) / 266-280 batch_insert (NULL): exactly the
 * state and d_ins[] that the reference's in-order loop produces for this batch. */
int cafe_insert_dev(cafe_sketch_t *s, const int32_t *d_ids, const float *d_scores, int n, int32_t *d_ins,
                    cafe_stream_t stream);

/* ---- tables: shared hot table (dlrm_s_pytorch.py:439-442) + per-field cold tables
 *      (sk_embedding_bag.py:75-80).  Arrays are HOST arrays of length num_fields; they are copied. ---- */
int cafe_tables_create(cafe_tables_t **out, int num_fields, int dim, int hotn, float *d_w_hot,
                       float *const *d_w_cold, const int64_t *id_offset, const int32_t *hash_size);
int cafe_tables_destroy(cafe_tables_t *t);
/* Measurement aid (no reference counterpart): when on, HIP events are recorded on the caller's stream around the two
 * bandwidth kernels of the path; cafe_tables_kernel_ms waits for and returns the last launch's duration.
 * which: 0 = fused query+gather (cafe_forward*_dev), 1 = segment-sum + SGD (cafe_backward_sgd*_dev). */
int cafe_tables_set_timing(cafe_tables_t *t, int on);
int cafe_tables_kernel_ms(cafe_tables_t *t, int which, float *ms);

/* sk_embedding_bag.py:101-113 (query) + 119-152 (forward), all fields in one launch.
 * d_local_ids: [num_fields, batch] int64 per-field ids (the reference's lS_i, one index per bag);
 * d_gids out:  [num_fields*batch] int32 = (int32)(id + id_offset[f]) (sk_embedding_bag.py:103);
 * d_que out:   [num_fields*batch] int32, batch_query of d_gids;
 * d_out:       [num_fields, batch, dim] float32 = hot ? W_hot[slot] : W_cold[f][gid % hash_size[f]]. */
int cafe_forward_dev(cafe_sketch_t *s, cafe_tables_t *t, const int64_t *d_local_ids, int batch,
                     int32_t *d_gids, int32_t *d_que, float *d_out, cafe_stream_t stream);

/* No counterpart in the reference (its loop is synchronous on the host).  Names the caller's NEXT cafe_forward_dev call
 * -- the same eight arguments -- typically the batch a data loader has already delivered.  The following
 * cafe_backward_update_dev on the same stream queues that forward (gather, plan of the next insert, partition passes)
 * behind its own kernels BEFORE it waits for the device's verdict, so the device never idles while the host waits and
 * returns; the announced cafe_forward_dev call then finds its work queued and returns at once.  Results are exactly
 * those of the plain call order: if the step is rejected and issued again, the queued forward is dropped and the
 * announced call runs as usual.  Requirements: d_gids / d_que of the announced call must not be the arrays the pending
 * update reads (alternate two sets), the batch size must equal the current step's, and the ids must be complete in
 * memory when the hint is given.  A hint that is not followed by the announced call is forgotten. */
int cafe_forward_hint_dev(cafe_sketch_t *s, cafe_tables_t *t, const int64_t *d_local_ids, int batch,
                          int32_t *d_gids, int32_t *d_que, float *d_out, cafe_stream_t stream);
/* out2[0] = announced forwards that were found queued, out2[1] = queued forwards dropped because their step was rejected. */
int cafe_forward_hint_stats(cafe_sketch_t *s, int64_t *out2);

/* sketch_ml_embed.py:154-199, the three-tier forward, in one launch: as cafe_forward_dev, plus
 * d_qmed out [num_fields*batch] = batch_query_median of d_gids, and a cold lookup of medium importance adds
 * d_w_median[gid % median_n] ([median_n, dim] float32, shared by all fields like the hot table) to its cold row. */
int cafe_forward_ml_dev(cafe_sketch_t *s, cafe_tables_t *t, const int64_t *d_local_ids, int batch,
                        const float *d_w_median, int median_n, int32_t *d_gids, int32_t *d_que, int32_t *d_qmed,
                        float *d_out, cafe_stream_t stream);

/* sk_embedding_bag.py:178-189 with the intended semantics: d_scores[f*batch+i] =
 * ||g_i|| * batch / sum_j ||g_j|| over field f.  d_grad_out: [num_fields, batch, dim]. */
int cafe_scores_dev(cafe_tables_t *t, const float *d_grad_out, int batch, float *d_scores, cafe_stream_t stream);

/* No counterpart in the reference.  cafe_scores_dev for callers that cannot allocate -- code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc: same kernels, same results, but the per-lookup norms (the
 * workspace cafe_scores_dev allocates and grows) live in a fixed 4 MiB device array inside the library, one copy per
 * GPU.  Allocates nothing; fails with a clear error when num_fields * batch > 1 048 576 lookups.  Every call on a GPU
 * shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race. */
int cafe_scores_noalloc_dev(cafe_tables_t *t, const float *d_grad_out, int batch, float *d_scores, cafe_stream_t stream);

/* sk_embedding_bag.py:199-208: for every lookup with d_ins != 0 whose id now reads hot,
 * W_hot[slot] = W_cold[f][local_id % hash_size[f]] (pre-update cold row; local id).
 * Arrays are the full [num_fields*batch] ones; only fields [first_field, first_field+n_fields) are visited,
 * so a caller that inserts field by field (dlrm_s_pytorch.py:734-738) can migrate field by field too. */
int cafe_migrate_dev(cafe_sketch_t *s, cafe_tables_t *t, const int64_t *d_local_ids, const int32_t *d_gids,
                     const int32_t *d_ins, int batch, int first_field, int n_fields, cafe_stream_t stream);

/* backward of the forward + torch.optim.SGD on the two sparse gradients (dlrm_s_pytorch.py:1658):
 * duplicates summed per row, then W -= lr * g.  d_que is the forward's routing. */
int cafe_backward_sgd_dev(cafe_tables_t *t, const int32_t *d_gids, const int32_t *d_que, const float *d_grad_out,
                          int batch, float lr, cafe_stream_t stream);

/* The four calls above in the reference's order (scores, batch_insert_val, migration, backward + SGD) behind one
 * host call; d_scores / d_ins [num_fields*batch] are outputs.
 * per_field != 0: the sketch sees what the reference's loop over fields gives it (dlrm_s_pytorch.py:734-738): one
 *   batch_insert_val per field, i.e. decay and batch counter advance once per field and the reset test of
 *   sketch.cpp:291 applies at every field boundary.  The fields still run as ONE parallel insert (identical state and
 *   ins[], since the calls walk consecutive parts of the same id stream) unless the host's own counters show a decay
 *   flip inside the step or Threshold <= 1 (migration then depends on the state after each field); such
 *   steps are issued field by field.  Whether the reset test fires at an inner boundary is decided on the device, from
 *   the exact real_n and per-field crossing counts, BEFORE anything is applied: such a step is rejected and issued again
 *   field by field inside this call, as is a step the parallel insert could not settle (first with the largest
 *   owners).  The call therefore WAITS for the device's verdict on the insert (a snapshot in pinned host memory; the
 *   migration and update kernels are queued behind the insert beforehand and do nothing for a rejected step); when it
 *   returns, the remaining kernels are enqueued on `stream` as usual.
 * per_field == 0: ONE batch_insert_val over all fields' ids in field-major order.
 * cafe_forward_dev starts the plan of this insert beside its gather kernel when it is called with d_local_ids; the
 * plan is used if this call follows with the same d_local_ids and no sketch call in between.
 * Streams: a large step (>= 32768 lookups) also uses the handle's own streams -- the plan and the update's partition
 *   passes on a second one, and, by the library's rule (option "split_update"), the update's pass over the gradient on
 *   a third one beside the insert, with the row updates behind the migration on `stream`.  All of it is ordered before
 *   the call's last kernel on `stream`: d_grad_out, d_que and the id arrays must stay unchanged until the work queued on
 *   `stream` by this call is complete, nothing else. */
int cafe_backward_update_dev(cafe_sketch_t *s, cafe_tables_t *t, const int64_t *d_local_ids, const int32_t *d_gids,
                             const int32_t *d_que, const float *d_grad_out, int batch, float lr, float *d_scores,
                             int32_t *d_ins, int per_field, cafe_stream_t stream);

/* ---- the same four steps for lookups in arbitrary order (each with its own field index), which is what
 *      a shard owner receives from its peers in the multi-GPU path (no reference
 *      counterpart: the reference is single-device).  local id = gid - id_offset[field]. ---- */
int cafe_forward_gid_dev(cafe_sketch_t *s, cafe_tables_t *t, const int32_t *d_gids, const int32_t *d_fields, int n,
                         int32_t *d_que, float *d_out, cafe_stream_t stream);
int cafe_migrate_gid_dev(cafe_sketch_t *s, cafe_tables_t *t, const int32_t *d_gids, const int32_t *d_fields,
                         const int32_t *d_ins, int n, cafe_stream_t stream);
int cafe_backward_sgd_gid_dev(cafe_tables_t *t, const int32_t *d_fields, const int32_t *d_que, const float *d_grad_out,
                              int n, float lr, cafe_stream_t stream);
/* The same with gradient rows grad_row_stride floats apart (a multiple of 4, >= dim): the multi-GPU path sends a
 * lookup's gradient row and its score in ONE all-to-all, as rows of dim + 4 floats with the score at column dim
 * (cafe_amd/sharded.py), and the owner updates straight from that buffer. */
int cafe_backward_sgd_gid_strided_dev(cafe_tables_t *t, const int32_t *d_fields, const int32_t *d_que,
                                      const float *d_grad_out, int grad_row_stride, int n, float lr, cafe_stream_t stream);
/* cafe_backward_update_dev for a shard: n lookups in arbitrary order (global id, field and routing word each), their rows
 * row_stride floats apart in d_rows (a multiple of 4, > dim) with the lookup's gradient in columns [0, dim) and its SCORE at
 * column dim, read in place -- the buffer the backward all-to-all delivers.  One batch_insert_val over the n ids in the
 * order given, migration (local id = gid - id_offset[field]), SGD: run as one parallel insert with the device's verdict,
 * the plan made beside cafe_forward_gid_dev, the update split by the library's rule, rejected steps issued again, like
 * cafe_backward_update_dev with per_field == 0.  d_ins [n] out. */
int cafe_backward_update_gid_dev(cafe_sketch_t *s, cafe_tables_t *t, const int32_t *d_gids, const int32_t *d_fields,
                                 const int32_t *d_que, const float *d_rows, int row_stride, int n, float lr, int32_t *d_ins,
                                 cafe_stream_t stream);
/* The route of a batch (no reference counterpart: the reference is single-device; cafe_amd/sharded.py): global id of
 * every lookup (local id + id_offset[field], as int32 bits: sk_embedding_bag.py:103), its owner rank
 * ((gid & 0xffffffff) * 2654435761 >> 15) mod world, and the lookups in stable order by owner: d_order [n] = positions in
 * the batch, d_packed [n] = field << 32 | gid in that order (what is sent to the owners), d_send_counts [world] = lookups
 * per owner.  d_ws: cafe_route_ws_words(n, world) 32-bit words of scratch.  world <= 64.  cafe_split_ids_dev takes
 * received d_packed words apart into the gid and field arrays the *_gid_dev entry points read. */
int cafe_route_ws_words(int n, int world);
int cafe_route_dev(const int64_t *d_local_ids, const int64_t *d_id_offset, int num_fields, int batch, int world,
                   uint32_t *d_ws, int64_t *d_order, int64_t *d_packed, int64_t *d_send_counts, cafe_stream_t stream);
int cafe_split_ids_dev(const int64_t *d_packed, int n, int32_t *d_gids, int32_t *d_fields, cafe_stream_t stream);
/* The two permutations around the exchanges, one launch each.  pack: d_out[j] = [d_grad[order[j]] (dim floats) | score | 0 0 0]
 * with score = d_norms[i] * scale / d_field_sums[i / batch] for i = order[j] (sk_embedding_bag.py:189 over the global
 * batch: scale = global batch size), also written to d_scores[i] when d_scores != NULL.  unpack: d_out[order[j]] =
 * d_back[j] (rows of dim floats).  d_order: int64 [n], a permutation. */
int cafe_pack_rows_dev(const float *d_grad, const float *d_norms, const float *d_field_sums, float scale, int batch,
                       const int64_t *d_order, int n, int dim, float *d_out, float *d_scores, cafe_stream_t stream);
int cafe_unpack_rows_dev(const float *d_back, const int64_t *d_order, int n, int dim, float *d_out, cafe_stream_t stream);
/* First half of cafe_scores_dev: d_norms[f*batch+i] = ||g_i||, d_field_sums[f] = sum_i ||g_i|| (fixed order), so that
 * ranks can all-reduce the sums before scaling (sk_embedding_bag.py:178-189 over the global batch). */
int cafe_grad_norm_dev(cafe_tables_t *t, const float *d_grad_out, int batch, float *d_norms, float *d_field_sums,
                       cafe_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * (E) The earlier HotSketch generation of the reference's artifact (ArtifactEvaluation/embeddings/sketch.cpp:9-151,
 *     class SS): `n` buckets of four slots kept sorted by score, bucket = id % n, a multiplicative decay() instead of
 *     the lazy flip bit, no LRU table; one instance per field (the reference keeps ss[26], :152).  Same contract as
 *     (B): device pointers, a stream, results bit-exact with the reference's in-order loop (ins[], que[], every slot,
 *     the free list in pop order, num, and tot -- the double the reference accumulates item by item).
 * ------------------------------------------------------------------------------------------------ */
typedef struct cafe_ae_sketch cafe_ae_sketch_t;
/* The gather and the migration for a routing produced by another sketch generation (section E): rows chosen by que[]
 * alone.  ArtifactEvaluation/embeddings/sk_embedding_bag.py:96-118 (`dic % hot_nums` / `dic % hash_size` behind
 * `torch.where(dic_mask, ...)`) and :175-182 (`weight_h[dic[x]] = weight_hash[input[x] % hash_size]` for the lookups the
 * insert reported, with the hot-only fence of SURVEY Q3).  d_fields == NULL: lookup i belongs to field i / batch. */
int cafe_gather_que_dev(cafe_tables_t *t, const int32_t *d_que, const int32_t *d_fields, int batch, int n, float *d_out,
                        cafe_stream_t stream);
int cafe_migrate_que_dev(cafe_tables_t *t, const int64_t *d_local_ids, const int32_t *d_que, const int32_t *d_fields,
                         const int32_t *d_ins, int batch, int n, cafe_stream_t stream);

/* :207-209 init(n, Threshold, dr, idx) -> new SS(Threshold, n, dr) (:26-43). */
int cafe_ae_create(cafe_ae_sketch_t **out, int n, int threshold, double decay_rate);
int cafe_ae_destroy(cafe_ae_sketch_t *s);
/* :135-151 batch_insert / batch_insert_val (d_vals == NULL: every score is 1): decay() first if the scores inserted
 * since the last decay exceed n * Threshold * 10, then Insert (:89-124) for every id in order.  d_ins[i] = 1 if the id
 * took a hot slot at this item.  Any n (the reference's own result buffers hold 16384, :7). */
int cafe_ae_insert_dev(cafe_ae_sketch_t *s, const int32_t *d_ids, const float *d_vals, int n, int32_t *d_ins,
                       cafe_stream_t stream);
/* :125-134 batch_query: -slot if the id holds a hot slot, else the id. */
int cafe_ae_query_dev(cafe_ae_sketch_t *s, const int32_t *d_ids, int n, int32_t *d_que, cafe_stream_t stream);
/* :61-75 update(): the free list becomes the ascending slot numbers 1..n-1 that no entry holds.  (The reference clears
 * only the first 8 bytes of its marker array -- memset(v, 0, sizeof(v)) on a pointer -- so its own result depends on
 * what operator new returned; this is the intended semantics.) */
int cafe_ae_update(cafe_ae_sketch_t *s, cafe_stream_t stream);
/* State for checkpoints and parity tests.  out5: buckets, free-list length, num, decay() calls so far, batches that
 * ran in order because the free list ran dry inside them.  export/import: val, cnt, dic [n * 4] (the id of an empty
 * slot, never initialised by the reference, is reported as 0), the free list in pop order, tot. */
int cafe_ae_dims(cafe_ae_sketch_t *s, int64_t *out5);
int cafe_ae_export(cafe_ae_sketch_t *s, uint32_t *val, float *cnt, int32_t *dic, uint32_t *fifo, double *tot);
int cafe_ae_import(cafe_ae_sketch_t *s, const uint32_t *val, const float *cnt, const int32_t *dic, const uint32_t *fifo,
                   int fifo_len, double tot, int num);
/* The artifact's own C entry points on HOST pointers (:207-228: init, batch_query, batch_insert, batch_insert_val,
 * update, each with the field index as last argument), under a prefix because tricks/sketch.cpp's entry points of the
 * same names and other signatures (section A) live in this library too; INTEGRATION.md shows the three-line ctypes
 * shim that binds them under the reference's names.  Results are returned in library-owned storage like the
 * reference's static arrays. */
void cafe_ae_ref_init(int n, int Threshold, double dr, int idx);
int *cafe_ae_ref_batch_query(uint32_t *data, int len, int idx);
int *cafe_ae_ref_batch_insert(uint32_t *data, int len, int idx);
int *cafe_ae_ref_batch_insert_val(uint32_t *data, float *v, int len, int idx);
void cafe_ae_ref_update(int idx);

#ifdef __cplusplus
}
#endif
#endif /* CAFE_HIP_H */
