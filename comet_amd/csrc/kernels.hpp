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
// Exact-arithmetic distances: D[q][row] = Calculate(Q[q], X[row]) with the reference's float32
// evaluation order (sequential over the dimension, no FMA). X: n x ld, Q: B x ld, D: B x ldD.
// elig (nullable): per-row eligibility bytes; ineligible rows get the EXCLUDED sentinel in D.
void launch_dist_exact(Ctx* c, int metric, const float* X, int64_t n, int ld, const float* Q, int B, float* D, int64_t ldD,
                       const uint8_t* elig);
// Bit pattern written into a distance matrix for candidates that must not be returned (soft-deleted,
// filtered out). A negative quiet NaN with all payload bits set — never produced by the arithmetic here.
constexpr uint32_t EXCLUDED_BITS = 0xFFFFFFFFu;
// one pair / a few pairs, single thread (comet.Distance singletons)
void launch_dist_pairs(Ctx* c, int metric, const float* A, const float* Bv, int npairs, int d, int a_stride, int b_stride, float* out);
// SplitMix64 synthetic fill
void launch_synth_fill(Ctx* c, uint64_t seed, uint64_t offset, uint64_t n, float* out);
// elig[row] = id not in deleted_sorted && (n_filter == 0 || id in filter_sorted)
void launch_build_elig(Ctx* c, const uint32_t* ids, int64_t n, const uint32_t* deleted_sorted, int n_deleted,
                       const uint32_t* filter_sorted, int n_filter, uint8_t* elig);
// out[i] = table[idx[i]] (ids gather), idx == 0xFFFFFFFF -> 0
void launch_gather_u32(Ctx* c, const uint32_t* table, const uint32_t* idx, int64_t n, uint32_t* out);

// ---- kernels_select.hip -----------------------------------------------------------------------
// Exact top-K by (score asc, position asc) over per-query candidate rows D[q][0..C_q).
//   cnts: per-query candidate counts (device, int32) or nullptr -> every query has C candidates.
//   entries equal to EXCLUDED_BITS are skipped;
//   thr : candidates with thr > 0 && score > thr are dropped (flat_index_search.go:269).
//   K   : <= 0 -> all. Results: out_pos/out_scores [B x k_cap] sorted; out_counts[q] = min(K_q, #eligible)
//         (number of results; rows hold min(count, k_cap)). Requires min(K, C) <= select_max_k().
int select_max_k();
void launch_select_topk(Ctx* c, const float* D, int64_t ldD, int B, int64_t C, const int32_t* cnts, float thr, int K,
                        uint32_t* out_pos, float* out_scores, int32_t* out_counts, int k_cap);

}  // namespace comet
