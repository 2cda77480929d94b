// kernels.hpp — launch wrappers for the hand-written gfx950 kernels (definitions in kernels_*.hip).
#pragma once
#include "common.hpp"

namespace comet {

// Row layout of every fp32 matrix the distance kernels touch: row stride `ld` = dim rounded up to
// ROW_PAD floats, padding zero-filled. Zero padding is exact for every metric: the reference's
// sequential sums only ever gain "+ 0.0f" terms at the tail (x + 0 == x bitwise for x >= +0, and
// dot + 0*0 keeps dot), so padded and unpadded sums are bit-identical.
constexpr int ROW_PAD = 32;
inline int padded_dim(int d) { return (int)round_up(d, ROW_PAD); }

// ---- kernels_dist.hip -------------------------------------------------------------------------
// src: n x d (dense) -> dst: n x ld (padded), preprocessing each row for `metric`
// (cosine: x *= 1.0f/norm, distance.go:244-290). zero_flag[i] = 1 if row i has zero norm under cosine.
void launch_ingest_rows(Ctx* c, int metric, const float* src, int64_t n, int d, float* dst, int ld, int32_t* zero_flag);
// dst (n x d dense) <- src (n x ld padded): strip padding (for read-back)
void launch_unpad_rows(Ctx* c, const float* src, int64_t n, int ld, float* dst, int d);
// Norm / Normalize / Scale (distance.go:312-428) over dense rows; vec_scale with norms != nullptr normalises (zero rows unchanged)
void launch_vec_norm(Ctx* c, const float* x, int64_t n, int d, float* norms);
void launch_vec_scale(Ctx* c, const float* x, int64_t n, int d, const float* norms, float scalar, float* out);
// Exact-arithmetic distances: D[q][row] = Calculate(Q[q], X[row]) with the reference's float32
// evaluation order (sequential over the dimension, no FMA). X: n x ld, Q: B x ld, D: B x ldD.
// elig (nullable): per-row eligibility bytes; ineligible rows get the EXCLUDED sentinel in D.
// coarse quantiser: probe_list[q][0..np) = the np nearest centroids by (exact distance, index); false = not applicable here
// list_len != nullptr: the pick kernel also writes the probe bookkeeping — seg_off[q][0..np] (prefix of the probed lists' lengths), cnts[q]
// (nullable) and uoff[q][0..np] (nullable: prefix of ceil(len / unit_rows)) — instead of a launch of its own behind it
bool launch_coarse_probe_fast(Ctx* c, int metric, const float* C, int nlist, int ld, int dim, const float* Qp, int B, int np, uint32_t* probe_list,
                              const int32_t* list_len = nullptr, int32_t* seg_off = nullptr, int32_t* cnts = nullptr, int32_t* uoff = nullptr, int unit_rows = 64);
void launch_dist_exact(Ctx* c, int metric, const float* X, int64_t n, int ld, const float* Q, int B, float* D, int64_t ldD,
                       const uint8_t* elig);
// ineligible / excluded candidates carry this bit pattern (a NaN the arithmetic cannot produce) in distance matrices
constexpr uint32_t EXCLUDED_BITS = 0xFFFFFFFFu;
// one pair / a few pairs, single thread (comet.Distance singletons)
void launch_dist_pairs(Ctx* c, int metric, const float* A, const float* Bv, int npairs, int d, int a_stride, int b_stride, float* out);
// SplitMix64 synthetic fill
void launch_synth_fill(Ctx* c, uint64_t seed, uint64_t offset, uint64_t n, float* out);
// clustered synthetic rows [row_base, row_base + n_rows) x dim (n_centers <= 0: the plain stream)
void launch_synth_mixture(Ctx* c, uint64_t seed, int n_centers, float sigma, int n_sub, float sigma_noise, uint64_t row_base, uint64_t n_rows, int dim, float* out);
// elig[row] = id not in deleted_sorted && (n_filter == 0 || id in filter_sorted)
void launch_build_elig(Ctx* c, const uint32_t* ids, int64_t n, const uint32_t* deleted_sorted, int n_deleted,
                       const uint32_t* filter_sorted, int n_filter, uint8_t* elig);
// out[i] = table[idx[i]] (ids gather), idx == 0xFFFFFFFF -> 0
void launch_gather_u32(Ctx* c, const uint32_t* table, const uint32_t* idx, int64_t n, uint32_t* out);

void launch_gather_indirect(Ctx* c, const uint32_t* table, int64_t ldt, const uint32_t* idx, int B, int k, uint32_t* out);

// ---- kernels_select.hip -----------------------------------------------------------------------
// Exact top-K by (score asc, position asc) over per-query candidate rows D[q][0..C_q).
//   cnts: per-query candidate counts (device, int32) or nullptr -> every query has C candidates.
//   entries equal to EXCLUDED_BITS are skipped;
//   thr : candidates with thr > 0 && score > thr are dropped (flat_index_search.go:269).
//   K   : <= 0 -> all. Results: out_pos/out_scores [B x k_cap] sorted; out_counts[q] = min(K_q, #eligible)
//         (number of results; rows hold min(count, k_cap)). Requires min(K, C) <= select_max_k().
int select_max_k();
// merge R per-shard sorted top-k lists per query (ties: lower shard, then lower position). R*k_cap <= select_max_k().
void launch_merge_topk(Ctx* c, const uint32_t* ids, const float* scores, const int32_t* counts, int R, int B, int k_cap, int k,
                       uint32_t* out_ids, float* out_scores, int32_t* out_counts, int64_t rank_stride = 0, int64_t rank_stride_counts = 0, void* workspace = nullptr);
size_t merge_topk_workspace_bytes(int R, int B, int k_cap);      // global-memory workspace of the merge (0: it runs in LDS); see launch_merge_topk
// segment merge (storage_merge.go:13-54): S per-segment top-k lists per query -> distinct ids with their highest score, score descending, cut to k
void launch_merge_segments(Ctx* c, const uint32_t* ids, const float* scores, const int32_t* counts, int S, int B, int k_cap, int k,
                           uint32_t* out_ids, float* out_scores, int32_t* out_counts, int out_ld);
// top-K of rows of (key << 32 | position) composites already filtered by a producer kernel (cursor[q] of them per row)
void launch_select_composites(Ctx* c, const unsigned long long* comp, int64_t ld, const int32_t* cursor, int B, int K, uint32_t* out_pos, float* out_scores,
                              int32_t* out_counts, int k_cap);
void launch_select_topk(Ctx* c, const float* D, int64_t ldD, int B, int64_t C, const int32_t* cnts, float thr, int K,
                        uint32_t* out_pos, float* out_scores, int32_t* out_counts, int k_cap);


// ---- kernels_quant.hip ------------------------------------------------------------------------
void launch_gather_rows(Ctx* c, const float* src, int ld, const int32_t* idx, int64_t k, float* dst);
void launch_argmin_update(Ctx* c, const float* D, int64_t ldD, int kb, int c0, int64_t n, float* best, int32_t* best_idx, bool first);
void launch_apply_assign(Ctx* c, const int32_t* new_idx, int32_t* assign, int64_t n, int32_t* changed);
void launch_fill_i32(Ctx* c, int32_t* p, int64_t n, int32_t v);
// centroid update step of k-means (clustering.go:213-239), member sums in ascending vector order
void launch_kmeans_update(Ctx* c, const float* V, int64_t n, int ld, const int32_t* assign, int k, float* centroids);
void launch_extract_sub(Ctx* c, const float* src, int ld_src, int64_t n, int col0, int dsub, float* dst, int ld_dst);
void launch_residual_rows(Ctx* c, const float* V, int ld, int64_t n, const float* C, const int32_t* assign, float* R);
void launch_pq_encode(Ctx* c, const float* R, int ld, int64_t n, const float* codebooks, int M, int Ksub, int dsub,
                      uint8_t* codes, int code_stride);
void launch_pq_bound_tab3(Ctx* c, const float* codebooks, int M, int dsub, float* btab3 /*64 * (dsub + 1) * M * 4 floats*/, float* cmax2 /*M*/);   // Ksub == 256
void launch_pq_list_rmax(Ctx* c, const float* codebooks, int M, int Ksub, int dsub, const uint32_t* codes_arr, int M4, const uint32_t* row_of_slot,
                         const int64_t* list_base, const int32_t* list_len, int nlist, float* rmax);
void launch_interleave_codes(Ctx* c, const uint32_t* src_words, int M4, const uint32_t* row_of_slot, int64_t nslots, uint32_t* dst);
void launch_probe_complete(Ctx* c, uint32_t* probe_list, int np, const int32_t* pcnt, int B);   // queries the exact coarse ranking gave fewer than np lists: lists 0 .. np-1
void launch_probe_segments(Ctx* c, const uint32_t* probe_list, int ldp, const int32_t* probe_cnt, const int32_t* list_len, int B, int np,
                           int32_t* seg_off, int32_t* cnts);
bool launch_order_pairs(Ctx* c, const uint32_t* probe_list, int ldp, int np, const int32_t* seg_off, int n_pairs, int nlist, uint32_t* order,
                        uint32_t* olist = nullptr);
void launch_dist_list(Ctx* c, int metric, const float* X, int ld, const float* Q, const uint32_t* order, const uint32_t* olist, int n_pairs, int nlist, int np,
                      const uint32_t* probe_list, int ldp, const int32_t* seg_off, const int64_t* list_base, const int32_t* list_len,
                      const uint32_t* row_of_slot, const uint8_t* elig, int max_list_len, float* D, int64_t ldD);
void launch_finalize_probe(Ctx* c, const uint32_t* pos, int B, int k_cap, const uint32_t* probe_list, int ldp, const int32_t* seg_off, int np,
                           const int64_t* list_base, const uint32_t* ids_of_slot, const int32_t* zflag, uint32_t* out_ids, int32_t* counts);
size_t adc_lds_bytes(int M, int Ksub, int dim);
int64_t adc_codes_pad();    // code slots the ADC scan may read (never use) past the last list
// fused top-K filter of the ADC scan (K in [1, ADC_FILTER_MAX_K]): survivors as composites in cand[q * ldD ..], cursor[q] of them
constexpr int ADC_FILTER_MAX_K = 64;
struct AdcFilter { unsigned long long* cand; int32_t* cursor; uint32_t* tq; int K; float thr;
                   int one_stage;    // 1: scan every probed candidate in one pass (search mode 1), no lower-bound pruning
                   // multi-GPU list shards: called between stage 1 and the lower-bound test with the sub-batch's bounds (float bits of sums, +inf where
                   // this rank scanned nothing for the query) — an all-reduce(min) over the ranks makes every rank prune with the tightest bound. A rank's
                   // bound is the K-th smallest sum of SOME candidates, i.e. an upper bound on the global K-th smallest, and so is the minimum.
                   void (*exchange)(void* user, uint32_t* tq, int n); void* exchange_user;
                   const float* bound_tab3; const float* bound_cmax2;   // nullable: the codebook as [k / 4][dimension | norm][subspace][k % 4] and the largest squared codeword norm per subspace (launch_pq_bound_tab3): pq_bound3_kernel
                   const float* list_rmax;   // nullable: per list an upper bound on the norm of its members' decoded residuals (launch_pq_list_rmax): the table-free lower bound
                   int32_t* stats;   // nullable, 8 ints: [0] += pairs behind the nearest lists the lower bound left alive, [1] += pairs behind the nearest lists (two-stage search),
                                     // [2] += 64-code blocks the scan's items cover, [3] += items (each streams one duo table), [4] += (query, block) pairs, [5] += searches
};
void launch_adc_scan(Ctx* c, const float* Qp, int ld, int dim, const float* centroids, const float* codebooks, int M, int Ksub, int dsub,
                     const uint32_t* codes, int M4, const int64_t* list_base, const int32_t* list_len, const uint32_t* probe_list, int ldp,
                     int np, const int32_t* seg_off, const uint8_t* elig, int B, int nlist, int max_list_len, float* D, int64_t ldD, const AdcFilter* flt = nullptr,
                     int64_t n_codes = 0 /*codes in the index: picks the scan kernel (adc_scan2_kernel on lists of a few thousand codes)*/);
// the exchanges launch_adc_scan would issue for B queries, with +inf bounds: for a rank of a sharded search whose shard has no candidate (flt->tq: B words)
void adc_exchange_idle(Ctx* c, const AdcFilter* flt, int M, int Ksub, int np, int B, int nlist);
// the bound exchanges a two-stage search with an exchange callback issues for a batch of B queries: their number, *per = queries per exchange (the last one
// carries the rest). Depends only on (M, Ksub, np, B, nlist): the same on every rank of a sharded search. 0: the search is single-stage for this shape.
int adc_exchange_plan(int M, int Ksub, int np, int B, int nlist, int* per);

// ---- kernels_fast.hip (MFMA fast path of the Flat scan) ------------------------------------------
int flat_fast_tile_rows();
int flat_fast_unit_rows(int nq, int64_t n, int64_t k, int ldh);    // rows per key unit for a slice of nq queries over n rows (64-row units on the narrow tile and where 128-row units would be expanded often)
int flat_fast_batch();
// fp32 padded rows -> fp16 shadow in the TILED layout [256-row tile][64-half K step][row][64] (each (tile, K step)
// slab is 32 KiB contiguous). X / rn point at the first new row whose global row index is row_base; the shadow
// buffer must hold whole tiles (zero-initialised padding rows). rn[row] = sum x^2 (nullable);
// stats[0] = max |x| bits, stats[1] = max row norm^2 bits (atomicMax; caller zero-initialises)
void launch_to_half_rows(Ctx* c, const float* X, int64_t n, int ld, void* Xh, int ldh, int64_t row_base, float* rn, uint32_t* stats);
// mode 0 cosine / 1 L2 family. Qh: 256 x ldh fp16. S0: 256 x ldS (2 packed keys per 256-row tile), bound: 256 x ldB.
void launch_flat_scan_f16(Ctx* c, int mode, const void* Xh, int64_t n, int ldh, const void* Qh, int nq_used, const float* rn, const float* qn,
                          const uint8_t* elig, float* S0, int64_t ldS, float* bound, int64_t ldB, int unit_rows);
void launch_flat_post(Ctx* c, int metric, const float* S0, int64_t ldS, const float* bound, int64_t ldB, int64_t n_tiles, int unit_rows, int64_t n, const uint8_t* elig,
                      const float* err_abs, int K, int kappa_rank, float thr, const float* X, int ld, const float* Qp, int B,
                      const uint32_t* ids_table, const int32_t* zflag, uint32_t* out_ids, float* out_scores, int32_t* out_counts, int k_cap,
                      int32_t* overflow, int32_t* stats);
// int8 shadow of the wide tile (round 3): rows [256-row tile][64-byte slab][row][64 codes] + one scale per TILE. Quantises every row of
// tiles tile0 .. ceil(n / 256) - 1 (X: row 0 of the index, n: its rows); stats[2] = max ||x - s c||^2 bits (atomicMax)
void launch_to_i8_tiles(Ctx* c, const float* X, int64_t n, int ld, void* X8, int ld8, int64_t tile0, float* st, uint32_t* stats);
// kernels_scanq.hip: the register-stationary tiles (int8 shadow rows of 256 / 512 / 768 bytes); false: not handled, use the older tiles
int flat_scan_qr_steps(int ld8);
bool launch_flat_scan_qr(Ctx* c, int mode, const void* X8, int64_t n, int ld8, const void* Q8F, int nq_used, const float* rn, const float* qn,
                         const float* sx, const float* sq, const uint8_t* elig, float* S0, int64_t ldS, float* bound, int64_t ldB, int unit_rows);
void launch_flat_scan_i8(Ctx* c, int mode, const void* X8, int64_t n, int ld8, const void* Q8F, const void* Q8R, int nq_used, const float* rn, const float* qn,
                         const float* sx, const float* sq, const uint8_t* elig, float* S0, int64_t ldS, float* bound, int64_t ldB, int unit_rows);
bool prep_queries_i8_ok(int dim);
// src != nullptr: raw queries (B x dim), preprocessed into Qp; src == nullptr: Qp holds preprocessed padded rows. Q8F: 256 x ld8 codes in fragment order
// (wide tile), Q8R: the same codes row-major (narrow tile; nullable).
void launch_prep_queries_i8(Ctx* c, int metric, const float* src, int B, int dim, float* Qp, int ld, int32_t* zero_flag, void* Q8F, void* Q8R, int ld8, float* sq, float* qn,
                            float* err_abs, int mode, float xmax_norm2, float dx_max, int32_t* stats4);
bool prep_queries_fused_ok(int dim);
void launch_prep_queries_fused(Ctx* c, int metric, const float* src, int B, int dim, float* Qp, int ld, int32_t* zero_flag, void* Qh, int ldh, float* qn,
                               float* err_abs, int mode, float xmax_norm2, int32_t* stats4);
void launch_prep_queries_fast(Ctx* c, const float* Qp, int B, int ld, int dim, void* Qh, int ldh, float* qn, float* err_abs, int mode, float xmax_norm2,
                              int32_t* stats4);

// the same post stage over the key rows of the IVF fast path
void launch_ivf_post(Ctx* c, int metric, const float* D, int64_t ldD, const float* umin, int64_t ldu, const int32_t* uoff, int np, const uint32_t* probe_list, int ldp,
                     const int64_t* list_base, const int32_t* list_len, const uint32_t* row_of_slot, const uint32_t* ids_slot, const uint8_t* elig,
                     const float* err_abs, int K, float thr, const float* X, int ld, const float* Qp, int B, const int32_t* zflag,
                     uint32_t* out_ids, float* out_scores, int32_t* out_counts, int k_cap, int32_t* overflow, int32_t* stats);

// ---- kernels_ivf.hip (MFMA fast path of the IVF list scan) ------------------------------------------
int ivf_fast_unit_rows();      // rows per key unit = the list alignment the slot layout must have (64)
int ivf_fast_max_lists();      // list counts the item builder takes
size_t ivf_group_bytes();
size_t ivf_item_bytes();
// slot-ordered fp16 shadow [unit of 64 slots][K step of 64 halves][row][64 halves] of the rows V[row_of_slot[slot]] (padding slots: zeros),
// rn[slot] = sum x^2, stats[0] = max |x| bits, stats[1] = max row norm^2 bits (atomicMax; caller zero-initialises)
void launch_ivf_shadow(Ctx* c, const float* V, int ld, const uint32_t* row_of_slot, int64_t nslots, void* Vh, int ldh, float* rn, uint32_t* stats);
// seg_off[q][0..np] = prefix of the probed lists' lengths, uoff[q][0..np] = prefix of their 64-row unit counts
void launch_ivf_probe_units(Ctx* c, const uint32_t* probe_list, int ldp, const int32_t* list_len, int B, int np, int32_t* seg_off, int32_t* uoff);
// (query, list) pairs -> groups of <= 64 queries per list -> (group, 256-row tile) items; counts[0] = items, counts[1] = groups
void launch_ivf_items(Ctx* c, const uint32_t* probe_list, int ldp, int np, const int32_t* uoff, int n_pairs, int nlist, const int32_t* list_len,
                      const int64_t* list_base, void* groups, void* items, int32_t* counts);
// int8 shadow of the IVF scan: one scale per 64-slot unit (su), codes in the fp16 shadow's layout with 128 dimensions per 128-byte K step; ld8 % 128 == 0
void launch_ivf_shadow_i8(Ctx* c, const float* V, int ld, const uint32_t* row_of_slot, int64_t nslots, void* V8, int ld8, float* su, uint32_t* stats);
void launch_ivf_scan_i8(Ctx* c, int mode, const void* V8, int ld8, const void* Q8R, const float* rn, const float* qn, const float* su, const float* sq, const uint8_t* elig,
                        const void* groups, const void* items, const int32_t* counts, float* D, int64_t ldD, float* umin, int64_t ldu);
void launch_ivf_scan_f16(Ctx* c, int mode, const void* Vh, int ldh, const void* Qh, const float* rn, const float* qn, const uint8_t* elig,
                         const void* groups, const void* items, const int32_t* counts, float* D, int64_t ldD, float* umin, int64_t ldu);

}  // namespace comet
