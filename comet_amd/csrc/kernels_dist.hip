// kernels_dist.hip — exact-arithmetic distance kernels for gfx950 (CDNA4, wave64).
//
// "Exact" = the value each kernel writes is bit-identical to the reference's scalar Go loop
// (distance.go:114-121 L2, :158-165 L2², :201-216 cosine): float32 accumulation strictly in
// dimension order, product rounded before the add (the file is compiled with -ffp-contract=off, so
// no v_fma/v_fmac/v_mad is emitted for these chains), sqrt correctly rounded.
//
// Hardware mapping (MI355X): the sum over the dimension for ONE (query,row) pair is a serial chain,
// so parallelism comes from pairs: one lane owns one corpus row and QT query accumulators. Corpus
// tiles (256 rows x 32 floats) are fetched with coalesced 16-byte loads (8 lanes cover one row's
// 128-byte line), transposed through LDS (row stride 36 floats: conflict-free ds_write_b128 /
// ds_read_b128 per the 64-bank rule), next tile prefetched into registers while the current one is
// consumed. Query values are wave-uniform and come in through scalar loads (SGPR operands of the
// VALU ops). Workgroups that share a row tile (different query groups) are mapped to the same XCD so
// the tile is fetched from HBM once and re-served from that XCD's L2.
#include "kernels.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace comet {

// ------------------------------------------------------------------------------------------------
// per-element steps — written as separate statements so each product / difference is rounded to
// float32 before the add, exactly like the Go source.
// ------------------------------------------------------------------------------------------------
// float32(math.Sqrt(float64(x))) (distance.go:120,258): evaluated literally in binary64 — the f64 sqrt is
// correctly rounded and 53 >= 2*24+2 bits make the second rounding innocuous, so this is the correctly
// rounded float32 square root. (The f32 v_sqrt_f32 path hipcc emits for sqrtf is only 1-ulp accurate.)
__device__ __forceinline__ float go_sqrt32(float x) { return (float)__builtin_sqrt((double)x); }

template <int METRIC> __device__ __forceinline__ float acc_step(float acc, float q, float x) {
    if constexpr (METRIC == COMET_COSINE) {
        float p = q * x;          // dot += a[i] * b[i]  (distance.go:204-206)
        return acc + p;
    } else {
        float diff = q - x;       // diff := a[i] - b[i] (distance.go:116-119)
        float sq = diff * diff;
        return acc + sq;
    }
}
template <int METRIC> __device__ __forceinline__ float acc_finish(float acc) {
    if constexpr (METRIC == COMET_COSINE) {
        if (acc > 1.0f) acc = 1.0f; else if (acc < -1.0f) acc = -1.0f;   // distance.go:209-213
        return 1.0f - acc;
    } else if constexpr (METRIC == COMET_L2) {
        return go_sqrt32(acc);
    } else {
        return acc;
    }
}

// ------------------------------------------------------------------------------------------------
// ingest: dense rows -> padded, preprocessed rows. One lane per row, sequential over the dimension.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ingest_rows_kernel(int metric, const float* __restrict__ src, long n, int d,
                                                          float* __restrict__ dst, int ld, int* __restrict__ zero_flag) {
    long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* s = src + r * (long)d;
    float* o = dst + r * (long)ld;
    float scale = 1.0f;
    int zf = 0;
    if (metric == COMET_COSINE) {
        float sum = 0.0f;
        for (int i = 0; i < d; i++) { float p = s[i] * s[i]; sum = sum + p; }   // distance.go:247-250
        float norm = go_sqrt32(sum);
        if (norm == 0.0f) zf = 1;
        else scale = 1.0f / norm;                                              // distance.go:258 (float32 divide)
    }
    if (metric == COMET_COSINE && !zf) { for (int i = 0; i < d; i++) o[i] = s[i] * scale; }
    else { for (int i = 0; i < d; i++) o[i] = s[i]; }
    for (int i = d; i < ld; i++) o[i] = 0.0f;
    if (zero_flag) zero_flag[r] = zf;
}
// wave-per-row variant: coalesced row load, squares staged in LDS, lane 0 runs the serial float32 sum
// (same order as the Go loop), all lanes write the scaled row. Used when the row fits the LDS budget.
constexpr int INGEST_MAX_D = 2048;
__global__ __launch_bounds__(256) void ingest_rows_wave_kernel(int metric, const float* __restrict__ src, long n, int d, float* __restrict__ dst,
                                                               int ld, int* __restrict__ zero_flag) {
    extern __shared__ __attribute__((aligned(16))) float sq[];   // [4 waves][dpad]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long r = (long)blockIdx.x * 4 + w;
    if (r >= n) return;
    const int dpad = (d + 3) & ~3;
    float* my = sq + (long)w * dpad;
    const float* s = src + r * (long)d;
    float* o = dst + r * (long)ld;
    float scale = 1.0f; int zf = 0;
    if (metric == COMET_COSINE) {
        for (int i = lane; i < dpad; i += 64) { float v = i < d ? s[i] : 0.0f; my[i] = v * v; }
        __builtin_amdgcn_s_waitcnt(0);   // this wave's LDS writes are complete before lane 0 reads them back
        __builtin_amdgcn_wave_barrier();
        float sum = 0.0f;
        if (lane == 0) {
            // the serial float32 sum of the reference (distance.go Norm); eight 16-byte reads in flight so that the chain
            // of adds, not the LDS latency, paces the loop
#pragma unroll 8
            for (int i = 0; i < dpad; i += 4) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(&my[i]);
                sum = sum + p[0]; sum = sum + p[1]; sum = sum + p[2]; sum = sum + p[3];   // trailing pad terms are +0
            }
        }
        sum = __shfl(sum, 0, 64);
        const float norm = go_sqrt32(sum);
        if (norm == 0.0f) zf = 1; else scale = 1.0f / norm;
    }
    if (metric == COMET_COSINE && !zf) { for (int i = lane; i < d; i += 64) o[i] = s[i] * scale; }
    else { for (int i = lane; i < d; i += 64) o[i] = s[i]; }
    for (int i = d + lane; i < ld; i += 64) o[i] = 0.0f;
    if (zero_flag && lane == 0) zero_flag[r] = zf;
}
void launch_ingest_rows(Ctx* c, int metric, const float* src, int64_t n, int d, float* dst, int ld, int32_t* zero_flag) {
    if (n <= 0) return;
    ProfScope ps(c, "ingest_rows");
    if (d <= INGEST_MAX_D) {
        const size_t lds = (size_t)4 * ((d + 3) & ~3) * sizeof(float);
        ingest_rows_wave_kernel<<<dim3((unsigned)ceil_div(n, 4)), dim3(256), lds, c->stream>>>(metric, src, n, d, dst, ld, zero_flag);
        LAUNCH_CHECK();
        return;
    }
    ingest_rows_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, c->stream>>>(metric, src, n, d, dst, ld, zero_flag);
    LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void unpad_rows_kernel(const float* __restrict__ src, long n, int ld, float* __restrict__ dst, int d) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = n * (long)d;
    if (i >= total) return;
    long r = i / d; int col = (int)(i - r * d);
    dst[i] = src[r * (long)ld + col];
}
void launch_unpad_rows(Ctx* c, const float* src, int64_t n, int ld, float* dst, int d) {
    if (n <= 0) return;
    unpad_rows_kernel<<<dim3((unsigned)ceil_div(n * d, 256)), dim3(256), 0, c->stream>>>(src, n, ld, dst, d);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// exact distance matrix
// ------------------------------------------------------------------------------------------------
constexpr int DC = 32;          // floats per row per staged chunk (one 128-byte line)
constexpr int TILE_ROWS = 256;  // rows per workgroup == threads per workgroup
constexpr int XS_LD = DC + 4;   // LDS row stride in floats: 144 B keeps 16-B alignment, conflict-free

template <int METRIC, int QT>
__global__ __launch_bounds__(256) void dist_exact_kernel(const float* __restrict__ X, long n, int ld,
                                                         const float* __restrict__ Qt, int B, float* __restrict__ D,
                                                         long ldD, int n_qg, long n_tiles,
                                                         const unsigned char* __restrict__ elig) {
    __shared__ __attribute__((aligned(16))) float xs[TILE_ROWS * XS_LD];
    // XCD-aware decomposition: block b runs on XCD b%8; give each XCD whole row tiles (all query groups).
    const long L = blockIdx.x;
    const int xcd = (int)(L & 7);
    const long slot = L >> 3;
    const long tile = (slot / n_qg) * 8 + xcd;
    const int qg = (int)(slot % n_qg);
    if (tile >= n_tiles) return;

    const int t = threadIdx.x;
    const long row0 = tile * TILE_ROWS;
    const int q0 = qg * QT;

    float acc[QT];
#pragma unroll
    for (int q = 0; q < QT; q++) acc[q] = 0.0f;

    // loader: thread t fetches float4 #(j*256+t) of the 256x32 tile -> row j*32 + t/8, column 4*(t%8)
    const int lrow = t >> 3, lc4 = (t & 7) * 4;
    f32x4 pre[8];
    const float* xrow[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        long r = row0 + j * 32 + lrow;
        if (r > n - 1) r = n - 1;              // tail rows re-read the last row; their results are not stored
        xrow[j] = X + r * (long)ld + lc4;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) pre[j] = *reinterpret_cast<const f32x4*>(xrow[j]);

    const int nchunks = ld / DC;
    for (int c = 0; c < nchunks; c++) {
        __syncthreads();  // everyone finished reading the previous chunk
#pragma unroll
        for (int j = 0; j < 8; j++) *reinterpret_cast<f32x4*>(&xs[(j * 32 + lrow) * XS_LD + lc4]) = pre[j];
        __syncthreads();
        {   // prefetch the next chunk into registers while this one is consumed (the last iteration
            // harmlessly re-reads its own chunk: keeps the loads unconditional and `pre` in VGPRs)
            const int cn = (c + 1 < nchunks) ? c + 1 : c;
#pragma unroll
            for (int j = 0; j < 8; j++) pre[j] = *reinterpret_cast<const f32x4*>(xrow[j] + cn * DC);
        }
        // query values: Qt is the per-query-group transposed copy [qg][col][QT], so the QT values that
        // multiply element `col` are contiguous -> one scalar load per element, operands stay in SGPRs.
        const float* __restrict__ qt = Qt + ((long)qg * ld + (long)c * DC) * QT;
#pragma unroll 1
        for (int i4 = 0; i4 < DC / 4; i4++) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(&xs[t * XS_LD + i4 * 4]);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float xe = x[e];
                const float* __restrict__ qrow = qt + (i4 * 4 + e) * QT;
#pragma unroll
                for (int q = 0; q < QT; q++) acc[q] = acc_step<METRIC>(acc[q], qrow[q], xe);
            }
        }
    }
    const long row = row0 + t;
    if (row < n) {
        const bool ok = elig ? (elig[row] != 0) : true;
#pragma unroll
        for (int q = 0; q < QT; q++) {
            if (q0 + q < B) {
                float v = ok ? acc_finish<METRIC>(acc[q]) : __uint_as_float(EXCLUDED_BITS);
                D[(long)(q0 + q) * ldD + row] = v;
            }
        }
    }
}

// Qt[(qg*ld + col)*QT + qq] = Q[min(qg*QT+qq, B-1)][col]
__global__ __launch_bounds__(256) void transpose_queries_kernel(const float* __restrict__ Q, int B, int ld, int QT, int n_qg, float* __restrict__ Qt) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)n_qg * ld * QT;
    if (i >= total) return;
    int qq = (int)(i % QT); long r = i / QT; int col = (int)(r % ld); int qg = (int)(r / ld);
    int q = qg * QT + qq; if (q > B - 1) q = B - 1;
    Qt[i] = Q[(long)q * ld + col];
}

template <int METRIC, int QT>
static void launch_dist_exact_t(Ctx* c, const float* X, int64_t n, int ld, const float* Q, int B, float* D, int64_t ldD,
                                const uint8_t* elig) {
    const int n_qg = (int)ceil_div(B, QT);
    const long n_tiles = ceil_div(n, TILE_ROWS);
    const long tiles8 = round_up(n_tiles, 8);
    const long grid = tiles8 * n_qg;
    float* Qt = c->salloc<float>((size_t)n_qg * ld * QT);
    {
        long total = (long)n_qg * ld * QT;
        transpose_queries_kernel<<<dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, c->stream>>>(Q, B, ld, QT, n_qg, Qt);
        LAUNCH_CHECK();
    }
    dist_exact_kernel<METRIC, QT><<<dim3((unsigned)grid), dim3(256), 0, c->stream>>>(X, n, ld, Qt, B, D, ldD, n_qg, n_tiles, elig);
    LAUNCH_CHECK();
}
template <int METRIC>
static void launch_dist_exact_m(Ctx* c, const float* X, int64_t n, int ld, const float* Q, int B, float* D, int64_t ldD,
                                const uint8_t* elig) {
    // queries per workgroup: 16 amortises the LDS transpose best, but a short matrix (coarse centroids, small indexes)
    // then gives fewer workgroups than the chip has CUs — halve the query group until the grid covers ~2 per CU
    int qt = 16;
    const int64_t tiles = ceil_div(n, TILE_ROWS);
    while (qt > 4 && tiles * ceil_div(B, qt) < 2 * (int64_t)c->prop.multiProcessorCount) qt >>= 1;
    if (B <= 1) launch_dist_exact_t<METRIC, 1>(c, X, n, ld, Q, B, D, ldD, elig);
    else if (B <= 2) launch_dist_exact_t<METRIC, 2>(c, X, n, ld, Q, B, D, ldD, elig);
    else if (B <= 4 || qt == 4) launch_dist_exact_t<METRIC, 4>(c, X, n, ld, Q, B, D, ldD, elig);
    else if (B <= 8 || qt == 8) launch_dist_exact_t<METRIC, 8>(c, X, n, ld, Q, B, D, ldD, elig);
    else launch_dist_exact_t<METRIC, 16>(c, X, n, ld, Q, B, D, ldD, elig);
}
void launch_dist_exact(Ctx* c, int metric, const float* X, int64_t n, int ld, const float* Q, int B, float* D, int64_t ldD,
                       const uint8_t* elig) {
    if (n <= 0 || B <= 0) return;
    ProfScope ps(c, "dist_exact");
    switch (metric) {
        case COMET_L2: launch_dist_exact_m<COMET_L2>(c, X, n, ld, Q, B, D, ldD, elig); break;
        case COMET_L2SQ: launch_dist_exact_m<COMET_L2SQ>(c, X, n, ld, Q, B, D, ldD, elig); break;
        default: launch_dist_exact_m<COMET_COSINE>(c, X, n, ld, Q, B, D, ldD, elig); break;
    }
}

// ------------------------------------------------------------------------------------------------
// Norm / Normalize / Scale (distance.go:312-428) over dense rows. The norm is the reference's serial float32 sum (one thread per row: a
// chain by definition), float32(sqrt(float64)); Normalize multiplies by 1 / norm (a zero row stays as it is); Scale is elementwise.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vec_norm_kernel(const float* __restrict__ x, long n, int d, float* __restrict__ norms) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const float* v = x + r * d;
    float sum = 0.0f;
    for (int i = 0; i < d; i++) { const float p = v[i] * v[i]; sum = sum + p; }
    norms[r] = (float)__builtin_sqrt((double)sum);
}
// out[r][i] = x[r][i] * (norms ? (norms[r] == 0 ? 1 (unchanged) : 1 / norms[r]) : scalar)
__global__ __launch_bounds__(256) void vec_scale_kernel(const float* __restrict__ x, long n, int d, const float* __restrict__ norms, float scalar, float* __restrict__ out) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * d) return;
    if (norms) {
        const float nv = norms[e / d];
        if (nv == 0.0f) { out[e] = x[e]; return; }
        const float sc = 1.0f / nv;
        out[e] = x[e] * sc;
    } else out[e] = x[e] * scalar;
}
void launch_vec_norm(Ctx* c, const float* x, int64_t n, int d, float* norms) {
    if (n <= 0) return;
    vec_norm_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, c->stream>>>(x, n, d, norms);
    LAUNCH_CHECK();
}
void launch_vec_scale(Ctx* c, const float* x, int64_t n, int d, const float* norms, float scalar, float* out) {
    if (n <= 0 || d <= 0) return;
    vec_scale_kernel<<<dim3((unsigned)ceil_div(n * d, 256)), dim3(256), 0, c->stream>>>(x, n, d, norms, scalar, out);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// Coarse quantiser, fast form (ivf_index_search.go:246-261: distances to ALL centroids, full sort, first nprobes).
// Ranking nlist centroids exactly costs B * nlist serial 768-term chains for the sake of the nprobes nearest. Here:
//   coarse_dot_kernel   S = Q . C^T in plain float32 FMAs (any order), plus ||q||^2 and ||c||^2
//   coarse_pick_kernel  one workgroup per query: approximate distances A from S (L2 family: ||q||^2 + ||c||^2 - 2S; cosine:
//                       1 - clamp(S)), kappa = the nprobes-th smallest A (32-step bit search on the keys in LDS), every centroid
//                       with A <= kappa + 2E is re-scored EXACTLY (the reference's serial float32 chain: acc_step / acc_finish),
//                       the (exact distance, index) order of those gives the probe list.
// Containment as for the Flat fast path: |A - exact| <= E for every centroid (E covers the FMA sums on this side and the
// rounding of the reference's own chain against the real value), so the nprobes centroids with the smallest A have exact
// distance <= kappa + E, hence the exact nprobes-th distance <= kappa + E and every exact top-nprobes centroid has A <= kappa + 2E.
// Euclidean ranks in the squared domain (+ a relative slack for sums that round to the same root). The probe list is
// bit-identical to the exact path's; the chains actually run are ~nprobes + a few per query instead of nlist.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned cq_f2key(unsigned u) { return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ unsigned cq_key2f(unsigned k) { return (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; }
constexpr int CQ_T = 32, CQ_LD = 36, CQ_MAX_LISTS = 8192;
__global__ __launch_bounds__(256) void coarse_dot_kernel(const float* __restrict__ Q, int B, const float* __restrict__ C, int nlist, int ld,
                                                         float* __restrict__ S, long ldS, float* __restrict__ qn, float* __restrict__ cn) {
    __shared__ __attribute__((aligned(16))) float Qs[CQ_T * CQ_LD], Cs[CQ_T * CQ_LD];
    const int t = threadIdx.x, q0 = blockIdx.y * CQ_T, l0 = blockIdx.x * CQ_T;
    const int tq = t >> 3, tl = (t & 7) * 4;             // this thread: query tq, centroids tl..tl+3 of the tile
    const int lrow = t >> 3, lc4 = (t & 7) * 4;          // loader: one float4 of each tile per K chunk
    const float* qsrc = Q + (long)min(q0 + lrow, B - 1) * ld + lc4;
    const float* csrc = C + (long)min(l0 + lrow, nlist - 1) * ld + lc4;
    // two K chunks of both tiles are kept in flight (registers): with one workgroup per CU a single chunk of look-ahead leaves
    // most of the load latency exposed
    const int nch = ld / CQ_T;                           // ld is a multiple of 32
    f32x4 pq[2], pc[2];
#pragma unroll
    for (int d = 0; d < 2; d++) { const int kk = min(d, nch - 1) * CQ_T; pq[d] = *reinterpret_cast<const f32x4*>(qsrc + kk); pc[d] = *reinterpret_cast<const f32x4*>(csrc + kk); }
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, qq = 0.f, cc[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_q = blockIdx.x == 0 && tl == 0, do_c = blockIdx.y == 0 && tq == 0;
    auto chunk = [&](int c, f32x4& rq, f32x4& rc) {
        __syncthreads();
        *reinterpret_cast<f32x4*>(&Qs[lrow * CQ_LD + lc4]) = rq;
        *reinterpret_cast<f32x4*>(&Cs[lrow * CQ_LD + lc4]) = rc;
        __syncthreads();
        const int kn = min(c + 2, nch - 1) * CQ_T;       // refill this slot two chunks ahead (past the end: a harmless re-read)
        rq = *reinterpret_cast<const f32x4*>(qsrc + kn); rc = *reinterpret_cast<const f32x4*>(csrc + kn);
#pragma unroll
        for (int k = 0; k < CQ_T; k += 4) {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(&Qs[tq * CQ_LD + k]);
            if (do_q) qq = __builtin_fmaf(qv[0], qv[0], __builtin_fmaf(qv[1], qv[1], __builtin_fmaf(qv[2], qv[2], __builtin_fmaf(qv[3], qv[3], qq))));
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x4 cv = *reinterpret_cast<const f32x4*>(&Cs[(tl + j) * CQ_LD + k]);
                acc[j] = __builtin_fmaf(qv[0], cv[0], __builtin_fmaf(qv[1], cv[1], __builtin_fmaf(qv[2], cv[2], __builtin_fmaf(qv[3], cv[3], acc[j]))));
                if (do_c) cc[j] = __builtin_fmaf(cv[0], cv[0], __builtin_fmaf(cv[1], cv[1], __builtin_fmaf(cv[2], cv[2], __builtin_fmaf(cv[3], cv[3], cc[j]))));
            }
        }
    };
    for (int c = 0; c < nch; c += 2) {
        chunk(c, pq[0], pc[0]);
        if (c + 1 < nch) chunk(c + 1, pq[1], pc[1]);
    }
    if (q0 + tq < B) {
#pragma unroll
        for (int j = 0; j < 4; j++) if (l0 + tl + j < nlist) S[(long)(q0 + tq) * ldS + l0 + tl + j] = acc[j];
        if (do_q) qn[q0 + tq] = qq;
    }
    if (do_c) {
#pragma unroll
        for (int j = 0; j < 4; j++) if (l0 + tl + j < nlist) cn[l0 + tl + j] = cc[j];
    }
}

// The same product on the matrix cores: v_mfma_f32_32x32x2_f32 (float32 operands and accumulation — E's "float32 sums in any order"
// covers it; 256 flop per clock and CU instead of the 128 of the FMA loop, and no per-thread LDS re-reads). 64 queries x 64
// centroids per workgroup, one 32 x 32 block per wave, K in chunks of 32 staged through LDS with two chunks of look-ahead in
// registers. A lane reads two consecutive K values per operand (ds_read_b64) and spends them on two MFMAs: lane (row, h) brings
// k = k0 + 2h + j to MFMA j — any pairing of K indices is fine as long as both operands use the same one. nlist 4096: 0.094 ->
// 0.0xx ms per 256 queries.
typedef float f32x16c __attribute__((ext_vector_type(16)));
typedef float f32x2c __attribute__((ext_vector_type(2)));
constexpr int CM_T = 64, CM_K = 32, CM_LD = 36;
__global__ __launch_bounds__(256) void coarse_dot_mfma_kernel(const float* __restrict__ Q, int B, const float* __restrict__ C, int nlist, int ld,
                                                              float* __restrict__ S, long ldS, float* __restrict__ qn, float* __restrict__ cn, int nsplit) {
    // K is split over blockIdx.z: a workgroup's time is its chain of K chunks (a barrier pair and a load round trip each), so for few
    // tiles (64 at nlist 1024) four times as many workgroups with a quarter of the chain each finish sooner; split z writes its partial
    // products and norms to plane z (S + z * B * ldS, qn + z * B, cn + z * nlist), coarse_pick_kernel adds the planes in a fixed order.
    __shared__ __attribute__((aligned(16))) float Qs[CM_T * CM_LD], Cs[CM_T * CM_LD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, q0 = blockIdx.y * CM_T, l0 = blockIdx.x * CM_T;
    const int wq = (w >> 1) * 32, wc = (w & 1) * 32;                 // this wave's 32 x 32 block inside the tile
    // loader: thread t moves two float4 of each tile per chunk: rows t >> 3 and 32 + (t >> 3), columns (t & 7) * 4 ..
    const int lrow = t >> 3, lc4 = (t & 7) * 4;
    const float* qsrc0 = Q + (long)min(q0 + lrow, B - 1) * ld + lc4;
    const float* qsrc1 = Q + (long)min(q0 + 32 + lrow, B - 1) * ld + lc4;
    const float* csrc0 = C + (long)min(l0 + lrow, nlist - 1) * ld + lc4;
    const float* csrc1 = C + (long)min(l0 + 32 + lrow, nlist - 1) * ld + lc4;   // (advanced to this split's first chunk below)
    const int nch_all = ld / CM_K;                                   // ld is a multiple of 32
    const int per = (nch_all + nsplit - 1) / nsplit, c_lo = blockIdx.z * per, nch = max(0, min(nch_all, c_lo + per) - c_lo);
    qsrc0 += c_lo * CM_K; qsrc1 += c_lo * CM_K; csrc0 += c_lo * CM_K; csrc1 += c_lo * CM_K;
    S += (long)blockIdx.z * B * ldS; qn += (long)blockIdx.z * B; cn += (long)blockIdx.z * nlist;
    f32x4 pq[2][2], pc[2][2];
#pragma unroll
    for (int d = 0; d < 2; d++) {
        const int kk = max(0, min(d, nch - 1)) * CM_K;
        pq[d][0] = *reinterpret_cast<const f32x4*>(qsrc0 + kk); pq[d][1] = *reinterpret_cast<const f32x4*>(qsrc1 + kk);
        pc[d][0] = *reinterpret_cast<const f32x4*>(csrc0 + kk); pc[d][1] = *reinterpret_cast<const f32x4*>(csrc1 + kk);
    }
    f32x16c acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    const bool do_q = blockIdx.x == 0, do_c = blockIdx.y == 0;       // the norms: thread t < 64 owns row t of the tile
    float nq = 0.0f, nc = 0.0f;
    const int orow = lane & 31, oh = lane >> 5;
    auto chunk = [&](int c, f32x4 (&rq)[2], f32x4 (&rc)[2]) {
        __syncthreads();
        *reinterpret_cast<f32x4*>(&Qs[lrow * CM_LD + lc4]) = rq[0]; *reinterpret_cast<f32x4*>(&Qs[(32 + lrow) * CM_LD + lc4]) = rq[1];
        *reinterpret_cast<f32x4*>(&Cs[lrow * CM_LD + lc4]) = rc[0]; *reinterpret_cast<f32x4*>(&Cs[(32 + lrow) * CM_LD + lc4]) = rc[1];
        __syncthreads();
        const int kn = min(c + 2, nch - 1) * CM_K;                   // refill this slot two chunks ahead (past the end: a harmless re-read)
        rq[0] = *reinterpret_cast<const f32x4*>(qsrc0 + kn); rq[1] = *reinterpret_cast<const f32x4*>(qsrc1 + kn);
        rc[0] = *reinterpret_cast<const f32x4*>(csrc0 + kn); rc[1] = *reinterpret_cast<const f32x4*>(csrc1 + kn);
#pragma unroll
        for (int k = 0; k < CM_K; k += 4) {
            const f32x2c a = *reinterpret_cast<const f32x2c*>(&Qs[(wq + orow) * CM_LD + k + 2 * oh]);
            const f32x2c b = *reinterpret_cast<const f32x2c*>(&Cs[(wc + orow) * CM_LD + k + 2 * oh]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
        }
        if (t < CM_T) {
            if (do_q) { for (int k = 0; k < CM_K; k++) { const float v = Qs[t * CM_LD + k]; nq = __builtin_fmaf(v, v, nq); } }
            if (do_c) { for (int k = 0; k < CM_K; k++) { const float v = Cs[t * CM_LD + k]; nc = __builtin_fmaf(v, v, nc); } }
        }
    };
    for (int c = 0; c < nch; c += 2) {
        chunk(c, pq[0], pc[0]);
        if (c + 1 < nch) chunk(c + 1, pq[1], pc[1]);
    }
    // C layout of the 32 x 32 MFMA: column = lane & 31 (operand B: centroid), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (operand A: query)
    const int lcol = l0 + wc + orow;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int qi = q0 + wq + (r & 3) + 8 * (r >> 2) + 4 * oh;
        if (qi < B && lcol < nlist) S[(long)qi * ldS + lcol] = acc[r];
    }
    if (t < CM_T) {
        if (do_q && q0 + t < B) qn[q0 + t] = nq;
        if (do_c && l0 + t < nlist) cn[l0 + t] = nc;
    }
}

template <int KR>
__device__ __forceinline__ unsigned cq_kappa_wave(const unsigned* keys, int nlist, int np, int lane) {
    unsigned kr[KR];
#pragma unroll
    for (int j = 0; j < KR; j++) kr[j] = (j * 64 + lane < nlist) ? keys[j * 64 + lane] : 0xFFFFFFFFu;
    unsigned kth = 0u;
    for (int bit = 31; bit >= 0; bit--) {
        const unsigned tv = kth | ((1u << bit) - 1u);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < KR; j++) cnt += (int)__builtin_popcountll(__ballot(j * 64 + lane < nlist && kr[j] <= tv));
        if (cnt < np) kth |= 1u << bit;
    }
    return kth;
}
template <int METRIC>
__global__ __launch_bounds__(256) void coarse_pick_kernel(const float* __restrict__ S, long ldS, const float* __restrict__ qn, const float* __restrict__ cn,
                                                          const float* __restrict__ C, int nlist, int ld, int dim, const float* __restrict__ Qp, int np,
                                                          int n2, unsigned* __restrict__ probe_list, int nsplit, int B,
                                                          const int* __restrict__ list_len /*nullable: also write the probe bookkeeping below*/, int* __restrict__ seg_off,
                                                          int* __restrict__ cnts /*nullable*/, int* __restrict__ uoff /*nullable*/, int unit_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cqs[];
    unsigned long long* comp = reinterpret_cast<unsigned long long*>(cqs);                 // [n2] composites of the re-scored centroids
    unsigned* keys = reinterpret_cast<unsigned*>(comp + n2);                                  // [nlist] keys of the approximate distances
    unsigned* list = keys + nlist;                                                            // [nlist] candidate centroids
    float* qs = reinterpret_cast<float*>(list + nlist);                                       // [ld] the query
    __shared__ int wcnt[2][4];
    __shared__ float wmax[4];
    __shared__ int s_n;
    const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    for (int i = t; i < ld; i += 256) qs[i] = Qp[(long)q * ld + i];
    float qnv = qn[q];
    for (int z = 1; z < nsplit; z++) qnv += qn[(long)z * B + q];      // the K splits of coarse_dot_mfma_kernel, in a fixed order
    float cmax = 0.0f;
    for (int l = t; l < nlist; l += 256) {
        float cv = cn[l], sv = S[(long)q * ldS + l];
        for (int z = 1; z < nsplit; z++) { cv += cn[(long)z * nlist + l]; sv += S[((long)z * B + q) * ldS + l]; }
        cmax = fmaxf(cmax, cv);
        float a;
        if constexpr (METRIC == COMET_COSINE) { float d = sv; if (d > 1.0f) d = 1.0f; else if (d < -1.0f) d = -1.0f; a = 1.0f - d; }
        else a = qnv + cv - 2.0f * sv;
        keys[l] = cq_f2key(__float_as_uint(a));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
    if (lane == 0) wmax[w] = cmax;
    if (t == 0) s_n = 0;
    __syncthreads();
    cmax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const float nq = sqrtf(qnv) * 1.0001f, nc = sqrtf(cmax) * 1.0001f, u = 6.0e-8f;
    float E;
    if constexpr (METRIC == COMET_COSINE) E = 4.0f * ((float)dim + 2.0f) * u * nq * nc + 2.0e-7f;
    else E = 4.0f * ((float)dim + 3.0f) * u * (nq + nc) * (nq + nc);
    // kappa: the np-th smallest key, bit by bit from the top. Up to 1024 lists ONE wave does it alone with the keys in registers
    // (16 per lane): a step is a ballot and a scalar popcount per register — no LDS traffic, no barrier (an s_memtime trace put
    // the four-wave form below, one barrier and two LDS round trips per bit, at 39 k of the kernel's 87 k clocks). With 64 registers
    // per lane (4096 lists) the one wave is slower than the four (0.082 vs 0.053 ms): 2048 dependent ballot / popcount pairs.
    unsigned kth = 0u;
    __shared__ unsigned s_kth;
    if (nlist <= 1024) {
        if (w == 0) kth = cq_kappa_wave<16>(keys, nlist, np, lane);
        if (t == 0) s_kth = kth;
        __syncthreads();
        kth = s_kth;
    } else
    // (one barrier per bit: the wave counts alternate between two slots; a nibble per step with 16 ballot counters was measured and
    // is slower: 0.041 vs 0.037 ms)
    for (int bit = 31; bit >= 0; bit--) {
        const unsigned tv = kth | ((1u << bit) - 1u);
        int cnt = 0;
        for (int l = t; l < nlist; l += 256) cnt += (keys[l] <= tv) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) wcnt[bit & 1][w] = cnt;
        __syncthreads();
        const int tot = wcnt[bit & 1][0] + wcnt[bit & 1][1] + wcnt[bit & 1][2] + wcnt[bit & 1][3];
        if (tot < np) kth |= 1u << bit;
    }
    const float kappa = __uint_as_float(cq_key2f(kth));
    const float tau = kappa + 2.0f * E + 2.0e-6f * fabsf(kappa);
    // a query with non-finite components has non-finite approximate distances: kappa / tau are NaN or +inf, `a <= tau` would admit nothing (or not the np
    // the list needs) and the probe list would stay unwritten — every centroid is re-scored then (keys of NaN sort last, ties by index: deterministic)
    const bool all_lists = !(fabsf(tau) < INFINITY);
    for (int l = t; l < nlist; l += 256) {
        const float a = __uint_as_float(cq_key2f(keys[l]));
        if (all_lists || a <= tau) list[atomicAdd(&s_n, 1)] = (unsigned)l;
    }
    __syncthreads();
    const int ncand = s_n;
    for (int ci = t; ci < ncand; ci += 256) {
        const unsigned l = list[ci];
        const float* __restrict__ cr = C + (long)l * ld;
        float acc = 0.0f;
        // the chain is serial, its operands are not: 32 elements of the centroid row are requested two blocks ahead (ld is a multiple of 32),
        // so the thread — one of ~40 at work in the workgroup — waits for the first block only
        f32x4 cur[8], nxt[8];
#pragma unroll
        for (int j = 0; j < 8; j++) cur[j] = *reinterpret_cast<const f32x4*>(cr + j * 4);
        for (int i0 = 0; i0 < ld; i0 += 32) {
            const int in = min(i0 + 32, ld - 32);
#pragma unroll
            for (int j = 0; j < 8; j++) nxt[j] = *reinterpret_cast<const f32x4*>(cr + in + j * 4);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const f32x4 qv = *reinterpret_cast<const f32x4*>(qs + i0 + j * 4);
                acc = acc_step<METRIC>(acc, qv[0], cur[j][0]); acc = acc_step<METRIC>(acc, qv[1], cur[j][1]);
                acc = acc_step<METRIC>(acc, qv[2], cur[j][2]); acc = acc_step<METRIC>(acc, qv[3], cur[j][3]);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) cur[j] = nxt[j];
        }
        const float d = acc_finish<METRIC>(acc);
        comp[ci] = ((unsigned long long)cq_f2key(__float_as_uint(d)) << 32) | l;
    }
    __syncthreads();
    if (ncand <= 1024) {
        // rank by counting (composites are unique): the first np ranks are the probe list
        for (int ci = t; ci < ncand; ci += 256) {
            const unsigned long long me = comp[ci];
            int rank = 0;
            for (int j = 0; j < ncand; j++) rank += (comp[j] < me) ? 1 : 0;
            if (rank < np) { probe_list[(long)q * np + rank] = (unsigned)(me & 0xFFFFFFFFull); keys[rank] = (unsigned)(me & 0xFFFFFFFFull); }
        }
    } else {
        int m2 = 64; while (m2 < ncand) m2 <<= 1;
        for (int i = ncand + t; i < m2; i += 256) comp[i] = ~0ull;
        __syncthreads();
        for (int k = 2; k <= m2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = t; i < m2; i += 256) {
                    const int ixj = i ^ j;
                    if (ixj > i) { const unsigned long long a = comp[i], b = comp[ixj]; if ((a > b) == ((i & k) == 0)) { comp[i] = b; comp[ixj] = a; } }
                }
                __syncthreads();
            }
        for (int i = t; i < np; i += 256) { probe_list[(long)q * np + i] = (unsigned)(comp[i] & 0xFFFFFFFFull); keys[i] = (unsigned)(comp[i] & 0xFFFFFFFFull); }
    }
    // probe bookkeeping of the list scans, fused (was a launch of its own behind this one: probe_segments_kernel / ivf_probe_units_kernel):
    // seg_off[q][0..np] = prefix of the probed lists' lengths (cnts[q] = their total), uoff[q][0..np] = prefix of their unit counts.
    // The probe list of this query is in `keys` (the approximate keys are not needed any more; np <= nlist / 4 entries).
    if (list_len) {
        __syncthreads();
        if (w == 0) {
            int run = 0, urun = 0;
            for (int p0 = 0; p0 < np; p0 += 64) {
                const int p = p0 + lane;
                const int len = p < np ? list_len[keys[p]] : 0;
                const int un = uoff ? (len + unit_rows - 1) / unit_rows : 0;
                int inc = len, uinc = un;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64), u2 = __shfl_up(uinc, o, 64); if (lane >= o) { inc += v; uinc += u2; } }
                if (p < np) { seg_off[(long)q * (np + 1) + p] = run + inc - len; if (uoff) uoff[(long)q * (np + 1) + p] = urun + uinc - un; }
                run += __shfl(inc, 63, 64); urun += __shfl(uinc, 63, 64);
            }
            if (lane == 0) { seg_off[(long)q * (np + 1) + np] = run; if (cnts) cnts[q] = run; if (uoff) uoff[(long)q * (np + 1) + np] = urun; }
        }
    }
}
// false: not applicable (too many lists for the LDS of the pick kernel, or most lists are probed anyway) — use the exact ranking
bool launch_coarse_probe_fast(Ctx* c, int metric, const float* C, int nlist, int ld, int dim, const float* Qp, int B, int np, uint32_t* probe_list,
                              const int32_t* list_len, int32_t* seg_off, int32_t* cnts, int32_t* uoff, int unit_rows) {
    static const bool off = getenv("COMET_COARSE_EXACT") != nullptr;
    if (off || nlist > CQ_MAX_LISTS || nlist < 64 || (int64_t)np * 4 > nlist || B <= 0) return false;
    ScratchMark mark(c);
    const int64_t ldS = round_up(nlist, 16);
    static const bool fma_dot = getenv("COMET_COARSE_FMA") != nullptr;      // the float32 FMA tile instead of the MFMA one
    const int nchunks = ld / CM_K, per4 = (nchunks + 3) / 4;
    const int nsplit = (!fma_dot && nchunks >= 8) ? (nchunks + per4 - 1) / per4 : 1;      // every split owns at least one K chunk
    float* S = c->salloc<float>((size_t)nsplit * B * ldS);
    float* qn = c->salloc<float>((size_t)nsplit * B);
    float* cn = c->salloc<float>((size_t)nsplit * nlist);
    { ProfScope ps(c, "coarse_dot");
      if (fma_dot) coarse_dot_kernel<<<dim3((unsigned)ceil_div(nlist, CQ_T), (unsigned)ceil_div(B, CQ_T)), dim3(256), 0, c->stream>>>(Qp, B, C, nlist, ld, S, ldS, qn, cn);
      else coarse_dot_mfma_kernel<<<dim3((unsigned)ceil_div(nlist, CM_T), (unsigned)ceil_div(B, CM_T), (unsigned)nsplit), dim3(256), 0, c->stream>>>(Qp, B, C, nlist, ld, S, ldS, qn, cn, nsplit);
      LAUNCH_CHECK(); }
    int n2 = 64; while (n2 < nlist) n2 <<= 1;
    const size_t lds = (size_t)n2 * 8 + (size_t)nlist * 8 + (size_t)ld * 4;
    { ProfScope ps(c, "coarse_pick");
#define CP(M) do { HIP_CHECK(hipFuncSetAttribute((const void*)coarse_pick_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                   coarse_pick_kernel<M><<<dim3(B), dim3(256), lds, c->stream>>>(S, ldS, qn, cn, C, nlist, ld, dim, Qp, np, n2, probe_list, nsplit, B, list_len, seg_off, cnts, uoff, unit_rows); } while (0)
      switch (metric) { case COMET_L2: CP(COMET_L2); break; case COMET_L2SQ: CP(COMET_L2SQ); break; default: CP(COMET_COSINE); break; }
#undef CP
      LAUNCH_CHECK(); }
    return true;
}

// ------------------------------------------------------------------------------------------------
// IVF list scan, list-ordered (ivf_index_search.go:277-301). One lane per candidate, the same LDS-transposed 128-byte-chunk
// pipeline as dist_exact_kernel (row indirection is free because each row chunk is fetched independently anyway); the work
// is laid out by (query, probed list) PAIR in list order: blockIdx.y walks the pairs sorted by list, so the workgroups
// that scan one inverted list for its different queries run back to back and the list's rows are served by L2 for all but
// the first (the per-query layout re-read every probed row from HBM once per query: 46 GB per 256-query batch at
// nprobe = 32 on 1M x 768). Rows are addressed straight from the slot layout (no candidate-row matrix).
// ------------------------------------------------------------------------------------------------
constexpr int LG_WINDOW = 16;     // sorted positions per grouping window = the largest query group
template <int METRIC, int QT>
__device__ __forceinline__ void list_group_scan(const float* __restrict__ X, int ld, const float* __restrict__ Q, const int* mq, const int* mso, int cnt,
                                                long base, int len, const unsigned* __restrict__ row_of_slot, const unsigned char* __restrict__ elig,
                                                float* __restrict__ D, long ldD, float* xs, float* qs, unsigned* rows) {
    const int t = threadIdx.x;
    const int nchunks = ld / DC;
    // loader roles: thread t fetches float4 #(j*256+t) of the 256 x 32 row tile, and floats t, t+256 of the QT x 32 query tile
    const int lrow = t >> 3, lc4 = (t & 7) * 4;
    const float* qsrc[2]; int qdst[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int e = t + h * 256, j = e >> 5, col = e & 31;           // member j (clamped: pad with the last one), column col
        qsrc[h] = Q + (long)mq[j < cnt ? j : cnt - 1] * ld + col;
        qdst[h] = col * QT + (j < QT ? j : 0);
    }
    constexpr bool QH1 = QT * 32 > 256;                                 // the second half-load exists only for QT = 16
    for (int pos0 = blockIdx.x * TILE_ROWS; pos0 < len; pos0 += gridDim.x * TILE_ROWS) {
        const int pos = pos0 + t;
        unsigned myrow = 0xFFFFFFFFu;
        if (pos < len && (!elig || elig[base + pos])) myrow = row_of_slot[base + pos];
        __syncthreads();                                   // previous tile's rows[] / xs[] / qs[] no longer in use
        rows[t] = myrow;
        __syncthreads();
        f32x4 pre[8];
        const float* xrow[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            unsigned r = rows[j * 32 + lrow];
            if (r == 0xFFFFFFFFu) r = 0;                   // excluded / past the end: read row 0, result discarded
            xrow[j] = X + (long)r * ld + lc4;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) pre[j] = *reinterpret_cast<const f32x4*>(xrow[j]);
        float qpre0 = (QT > 1 && t < QT * 32) ? qsrc[0][0] : 0.0f, qpre1 = 0.0f;
        if constexpr (QH1) qpre1 = qsrc[1][0];
        float acc[QT];
#pragma unroll
        for (int j = 0; j < QT; j++) acc[j] = 0.0f;
        for (int c = 0; c < nchunks; c++) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; j++) *reinterpret_cast<f32x4*>(&xs[(j * 32 + lrow) * XS_LD + lc4]) = pre[j];
            if (QT > 1 && t < QT * 32) qs[qdst[0]] = qpre0;
            if constexpr (QH1) qs[qdst[1]] = qpre1;
            __syncthreads();
            {
                const int cn = (c + 1 < nchunks) ? c + 1 : c;
#pragma unroll
                for (int j = 0; j < 8; j++) pre[j] = *reinterpret_cast<const f32x4*>(xrow[j] + cn * DC);
                if (QT > 1 && t < QT * 32) qpre0 = qsrc[0][cn * DC];
                if constexpr (QH1) qpre1 = qsrc[1][cn * DC];
            }
            if constexpr (QT == 1) {
                // a lone query: its values are wave-uniform, read them with scalar loads straight from the query row
                const float* __restrict__ qv = Q + (long)mq[0] * ld + c * DC;
#pragma unroll
                for (int i4 = 0; i4 < DC / 4; i4++) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(&xs[t * XS_LD + i4 * 4]);
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[0] = acc_step<METRIC>(acc[0], qv[i4 * 4 + e], x[e]);
                }
            } else {
#pragma unroll 2
                for (int i4 = 0; i4 < DC / 4; i4++) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(&xs[t * XS_LD + i4 * 4]);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float xe = x[e];
                        const float* qrow = qs + (i4 * 4 + e) * QT;          // QT consecutive floats: broadcast LDS reads
#pragma unroll
                        for (int j = 0; j < QT; j++) acc[j] = acc_step<METRIC>(acc[j], qrow[j], xe);
                    }
                }
            }
        }
        if (pos < len) {
#pragma unroll
            for (int j = 0; j < QT; j++)
                if (j < cnt) D[(long)mq[j] * ldD + mso[j] + pos] = (myrow == 0xFFFFFFFFu) ? __uint_as_float(EXCLUDED_BITS) : acc_finish<METRIC>(acc[j]);
        }
    }
}

template <int METRIC, bool GROUPED>
__global__ __launch_bounds__(256) void dist_list_kernel(const float* __restrict__ X, int ld, const float* __restrict__ Q,
                                                        const unsigned* __restrict__ order, const unsigned* __restrict__ olist /*nullable*/, int n_pairs, int nlist,
                                                        int np, const unsigned* __restrict__ probe_list, int ldp,
                                                        const int* __restrict__ seg_off, const long* __restrict__ list_base, const int* __restrict__ list_len,
                                                        const unsigned* __restrict__ row_of_slot, const unsigned char* __restrict__ elig,
                                                        float* __restrict__ D, long ldD) {
    __shared__ __attribute__((aligned(16))) float xs[TILE_ROWS * XS_LD];
    __shared__ __attribute__((aligned(16))) float qs[32 * LG_WINDOW];
    __shared__ unsigned rows[TILE_ROWS];
    __shared__ int mq[LG_WINDOW], mso[LG_WINDOW];
    const int y = blockIdx.y;
    // groups: inside every window of 16 sorted positions, a maximal run of pairs that probe the same list is ONE workgroup's
    // job (its leader = the run's first position); the row tile is fetched and transposed once for the whole group
    int cnt = 1;
    if (GROUPED) {
        const unsigned L0 = olist[y];
        if (L0 >= (unsigned)nlist) return;                              // pairs with nothing to scan sort last
        if ((y % LG_WINDOW) != 0 && olist[y - 1] == L0) return;         // not a leader
        while (y + cnt < n_pairs && ((y + cnt) % LG_WINDOW) != 0 && olist[y + cnt] == L0) cnt++;
    }
    const int t = threadIdx.x;
    if (t < cnt) {
        const int pair = (int)order[y + t];
        const int q = pair / np, pi = pair - q * np;
        mq[t] = q; mso[t] = seg_off[(long)q * (np + 1) + pi];
    }
    const int pair0 = (int)order[y];
    const int q0 = pair0 / np, pi0 = pair0 - q0 * np;
    if (seg_off[(long)q0 * (np + 1) + pi0 + 1] == seg_off[(long)q0 * (np + 1) + pi0]) return;   // (identity order) empty list / unused probe slot
    const unsigned L = probe_list[(long)q0 * ldp + pi0];
    const int len = list_len[L];
    const long base = list_base[L];
    __syncthreads();
#define LG(QTV) list_group_scan<METRIC, QTV>(X, ld, Q, mq, mso, cnt, base, len, row_of_slot, elig, D, ldD, xs, qs, rows)
    if constexpr (!GROUPED) LG(1);      // a kernel of its own: the 16-accumulator path would set the register count (and occupancy) for everyone
    else { if (cnt == 1) LG(1); else if (cnt == 2) LG(2); else if (cnt <= 4) LG(4); else if (cnt <= 8) LG(8); else LG(16); }
#undef LG
}
void launch_dist_list(Ctx* c, int metric, const float* X, int ld, const float* Q, const uint32_t* order, const uint32_t* olist, int n_pairs, int nlist, int np,
                      const uint32_t* probe_list, int ldp, const int32_t* seg_off, const int64_t* list_base, const int32_t* list_len,
                      const uint32_t* row_of_slot, const uint8_t* elig, int max_list_len, float* D, int64_t ldD) {
    if (n_pairs <= 0 || max_list_len <= 0) return;
    ProfScope ps(c, "dist_list");
    dim3 grid((unsigned)std::min<int64_t>(8, ceil_div(max_list_len, TILE_ROWS)), (unsigned)n_pairs), blk(256);
#define DL(M) do { if (olist) dist_list_kernel<M, true><<<grid, blk, 0, c->stream>>>(X, ld, Q, order, olist, n_pairs, nlist, np, probe_list, ldp, seg_off, (const long*)list_base, list_len, row_of_slot, elig, D, ldD); \
                    else dist_list_kernel<M, false><<<grid, blk, 0, c->stream>>>(X, ld, Q, order, olist, n_pairs, nlist, np, probe_list, ldp, seg_off, (const long*)list_base, list_len, row_of_slot, elig, D, ldD); } while (0)
    switch (metric) { case COMET_L2: DL(COMET_L2); break; case COMET_L2SQ: DL(COMET_L2SQ); break; default: DL(COMET_COSINE); break; }
#undef DL
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// a handful of pairs, one thread each (comet.Distance singletons; not a throughput path)
// ------------------------------------------------------------------------------------------------
__global__ void dist_pairs_kernel(int metric, const float* __restrict__ A, const float* __restrict__ Bv, int npairs, int d,
                                  int a_stride, int b_stride, float* __restrict__ out) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const float* a = A + (long)p * a_stride;
    const float* b = Bv + (long)p * b_stride;
    float acc = 0.0f;
    if (metric == COMET_COSINE) { for (int i = 0; i < d; i++) acc = acc_step<COMET_COSINE>(acc, a[i], b[i]); out[p] = acc_finish<COMET_COSINE>(acc); }
    else {
        for (int i = 0; i < d; i++) acc = acc_step<COMET_L2SQ>(acc, a[i], b[i]);
        out[p] = metric == COMET_L2 ? acc_finish<COMET_L2>(acc) : acc;
    }
}
void launch_dist_pairs(Ctx* c, int metric, const float* A, const float* Bv, int npairs, int d, int a_stride, int b_stride, float* out) {
    if (npairs <= 0) return;
    dist_pairs_kernel<<<dim3((unsigned)ceil_div(npairs, 64)), dim3(64), 0, c->stream>>>(metric, A, Bv, npairs, d, a_stride, b_stride, out);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// SplitMix64 synthetic data; element i uses counter offset+i
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void synth_fill_kernel(unsigned long long seed, unsigned long long offset, unsigned long long n, float* __restrict__ out) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long z = seed + (offset + i + 1ull) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        float u = (float)(z >> 40) * (1.0f / 16777216.0f);
        out[i] = 2.0f * u - 1.0f;
    }
}
void launch_synth_fill(Ctx* c, uint64_t seed, uint64_t offset, uint64_t n, float* out) {
    if (!n) return;
    unsigned grid = (unsigned)std::min<uint64_t>(ceil_div((int64_t)n, 256), 256 * 32);
    synth_fill_kernel<<<dim3(grid), dim3(256), 0, c->stream>>>(seed, offset, n, out);
    LAUNCH_CHECK();
}

// Clustered synthetic rows (SURVEY.md §8d "mixture-of-Gaussians variant" for meaningful ANN recall), two levels:
//   row r, column j = C1[b1(r)][j] + sigma * C2[b2(r)][j] + sigma_noise * u(seed, r*dim + j)          (n_sub > 0)
//                   = C1[b1(r)][j] + sigma * u(seed, r*dim + j)                                          (n_sub <= 0)
//   C1[c][j] = u(seed ^ 0x5EED, c*dim + j), C2[c][j] = u(seed ^ 0x5EED2, c*dim + j), u = the SplitMix64 stream of synth_fill,
//   b1(r) = ((r * 2654435761) >> 7) % n_centers, b2(r) = ((r * 0x9E3779B1) >> 5) % n_sub (64-bit arithmetic);
//   every product is rounded before its add (no FMA), left to right.
__device__ __forceinline__ float synth_u(unsigned long long seed, unsigned long long counter) {
    unsigned long long z = seed + (counter + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return 2.0f * ((float)(z >> 40) * (1.0f / 16777216.0f)) - 1.0f;
}
__global__ __launch_bounds__(256) void synth_mixture_kernel(unsigned long long seed, int n_centers, float sigma, int n_sub, float sigma_noise,
                                                            unsigned long long row_base, unsigned long long n_rows, int dim, float* __restrict__ out) {
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x, total = n_rows * (unsigned long long)dim;
    for (; i < total; i += stride) {
        const unsigned long long r = row_base + i / (unsigned long long)dim; const unsigned j = (unsigned)(i % (unsigned long long)dim);
        const unsigned long long b1 = ((r * 2654435761ull) >> 7) % (unsigned long long)n_centers;
        float v = synth_u(seed ^ 0x5EEDull, b1 * (unsigned long long)dim + j);
        const float nz = synth_u(seed, r * (unsigned long long)dim + j);
        if (n_sub > 0) {
            const unsigned long long b2 = ((r * 0x9E3779B1ull) >> 5) % (unsigned long long)n_sub;
            const float s2 = synth_u(seed ^ 0x5EED2ull, b2 * (unsigned long long)dim + j) * sigma;
            v = v + s2;
            const float t = nz * sigma_noise;
            v = v + t;
        } else {
            const float t = nz * sigma;
            v = v + t;
        }
        out[i] = v;
    }
}
void launch_synth_mixture(Ctx* c, uint64_t seed, int n_centers, float sigma, int n_sub, float sigma_noise, uint64_t row_base, uint64_t n_rows, int dim, float* out) {
    if (!n_rows || dim <= 0) return;
    if (n_centers <= 0) { launch_synth_fill(c, seed, row_base * (uint64_t)dim, n_rows * (uint64_t)dim, out); return; }
    unsigned grid = (unsigned)std::min<uint64_t>(ceil_div((int64_t)(n_rows * (uint64_t)dim), 256), 256 * 32);
    synth_mixture_kernel<<<dim3(grid), dim3(256), 0, c->stream>>>(seed, n_centers, sigma, n_sub, sigma_noise, row_base, n_rows, dim, out);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// eligibility bytes and id gather
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool sorted_contains(const unsigned* __restrict__ a, int n, unsigned v) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo < n && a[lo] == v;
}
__global__ __launch_bounds__(256) void build_elig_kernel(const unsigned* __restrict__ ids, long n, const unsigned* __restrict__ del, int nd,
                                                         const unsigned* __restrict__ flt, int nf, unsigned char* __restrict__ elig) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned id = ids[i];
    bool ok = true;
    if (nd > 0 && sorted_contains(del, nd, id)) ok = false;          // deletedNodes.Contains(id) flat_index_search.go:256
    if (ok && nf > 0 && !sorted_contains(flt, nf, id)) ok = false;   // docFilter.ShouldSkip(id)   flat_index_search.go:261
    elig[i] = ok ? 1 : 0;
}
void launch_build_elig(Ctx* c, const uint32_t* ids, int64_t n, const uint32_t* deleted_sorted, int n_deleted,
                       const uint32_t* filter_sorted, int n_filter, uint8_t* elig) {
    if (n <= 0) return;
    ProfScope ps(c, "build_elig");
    build_elig_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, c->stream>>>(ids, n, deleted_sorted, n_deleted, filter_sorted, n_filter, elig);
    LAUNCH_CHECK();
}
__global__ __launch_bounds__(256) void gather_u32_kernel(const unsigned* __restrict__ table, const unsigned* __restrict__ idx, long n, unsigned* __restrict__ out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned p = idx[i];
    out[i] = p == 0xFFFFFFFFu ? 0u : table[p];
}
void launch_gather_u32(Ctx* c, const uint32_t* table, const uint32_t* idx, int64_t n, uint32_t* out) {
    if (n <= 0) return;
    gather_u32_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, c->stream>>>(table, idx, n, out);
    LAUNCH_CHECK();
}

// out[q][i] = table[q][idx[q][i]] (per-query table of width ldt); idx == 0xFFFFFFFF stays 0xFFFFFFFF
__global__ __launch_bounds__(256) void gather_indirect_kernel(const unsigned* __restrict__ table, long ldt, const unsigned* __restrict__ idx, int B, int k, unsigned* __restrict__ out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * k) return;
    const long q = i / k;
    const unsigned p = idx[i];
    out[i] = p == 0xFFFFFFFFu ? p : table[q * ldt + p];
}
void launch_gather_indirect(Ctx* c, const uint32_t* table, int64_t ldt, const uint32_t* idx, int B, int k, uint32_t* out) {
    if (B <= 0 || k <= 0) return;
    gather_indirect_kernel<<<dim3((unsigned)ceil_div((long)B * k, 256)), dim3(256), 0, c->stream>>>(table, ldt, idx, B, k, out);
    LAUNCH_CHECK();
}

}  // namespace comet
