// kernels_fast.hip — the MFMA fast path of the Flat scan for gfx950.
//
// Idea: the exact-arithmetic kernel is VALU-bound at batch 256 (3 float ops per
// element-pair, no FMA allowed). The fast path therefore does the O(B*N*d) work on the matrix cores
// with a half-precision SHADOW of the corpus, but only to PROPOSE candidates; every returned score
// is still produced by the exact kernel, and the candidate set provably contains the exact top-K:
//
//   1. scan   : S = Xh (N x d, fp16) . Qh^T (d x 256, fp16) on v_mfma_f32_32x32x16_f16, one 256-row x
//               256-query tile per workgroup, K streamed through LDS with global_load_lds (16 B/lane,
//               XOR-swizzled source so ds_read_b128 is conflict-free). The epilogue never writes S: it
//               turns each accumulator into a non-negative approximate distance, packs the row-in-tile
//               into the low 8 mantissa bits, and keeps per (query, 128-row key unit) the two smallest keys
//               plus the third smallest (a lower bound for every row of the unit that was NOT emitted) with a
//               branch-free min/med3 network.
//   2. post   : one workgroup per query (flat_post_kernel): kappa = exact K-th smallest emitted key;
//               tau = kappa + 2E, E a rigorous bound on |approx - exact| (fp16 rounding of both operands, fp32
//               accumulation, key packing). Candidates = emitted keys <= tau, plus ALL rows of any unit whose
//               bound <= tau. Every row with approx <= tau is in that set, and the exact top-K all have
//               approx <= a_K + 2E <= tau, so the set contains the exact answer. The candidates are then
//               re-scored in the reference's float32 order and selected exactly: ids and scores are
//               bit-identical to the strict path; a query whose candidate list overflows is flagged and
//               re-run on the strict path.
#include <type_traits>
#include "kernels.hpp"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

namespace comet {

// ------------------------------------------------------------------------------------------------
// fp32 padded rows -> fp16 shadow rows (+ squared norms, max |x|, max norm^2)
// ------------------------------------------------------------------------------------------------
// Shadow layout ("tiled"): [tile of 256 rows][K slab of 32 halves][row in tile][32 halves] — the 16 KiB a
// workgroup needs for one 32-wide K slab of its tile are CONTIGUOUS in HBM (whole DRAM pages stream in), instead of
// 256 separate 64-byte pieces at a 1536-byte stride as in a row-major shadow.
__device__ __forceinline__ long tiled_off(long row, int k, int ldh) {
    const long tile = row >> 8; const int r = (int)(row & 255);
    return ((tile * (ldh >> 5) + (k >> 5)) * 256 + r) * 32 + (k & 31);
}
__global__ __launch_bounds__(256) void to_half_rows_kernel(const float* __restrict__ X, long n, int ld, _Float16* __restrict__ Xh, int ldh, long row_base,
                                                           float* __restrict__ rn, unsigned* __restrict__ stats /*[0]=max|x| bits, [1]=max norm2 bits*/) {
    // one wave per row; X / rn point at the first NEW row, row_base is its global index in the shadow
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* x = X + row * (long)ld;
    float s = 0.0f, mx = 0.0f;
    for (int i = lane; i < ldh; i += 64) {
        float v = i < ld ? x[i] : 0.0f;
        Xh[tiled_off(row_base + row, i, ldh)] = (_Float16)v;
        s += v * v;
        mx = fmaxf(mx, fabsf(v));
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); mx = fmaxf(mx, __shfl_xor(mx, off, 64)); }
    if (lane == 0) {
        if (rn) rn[row] = s;
        if (stats) { atomicMax(&stats[0], __float_as_uint(mx)); atomicMax(&stats[1], __float_as_uint(s)); }
    }
}
void launch_to_half_rows(Ctx* c, const float* X, int64_t n, int ld, void* Xh, int ldh, int64_t row_base, float* rn, uint32_t* stats) {
    if (n <= 0) return;
    ProfScope ps(c, "to_half_rows");
    to_half_rows_kernel<<<dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, c->stream>>>(X, n, ld, (_Float16*)Xh, ldh, row_base, rn, stats);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// fp32 padded rows -> int8 shadow (round 3): x_i = s_T * c_i + delta_i, c_i in [-127, 127], ONE SCALE PER 256-ROW TILE
// ------------------------------------------------------------------------------------------------
// The wide scan tile is matrix-pipe / power bound on fp16 (0.45-0.48 ms for 1M x 768 x 256 queries); v_mfma_i32_32x32x32_i8 does the
// same tile in half the instructions from half the bytes. The int8 dot product s_T s_q sum c_i e_i is the EXACT dot product of the two
// dequantised vectors; what it misses of x . q is  delta . q_hat + x . eps  (delta, eps: the quantisation residuals of row and query),
// bounded by Cauchy-Schwarz with the residual NORMS MEASURED here (not a worst-case model): |x.q - x_hat.q_hat| <= ||delta|| ||q_hat|| +
// ||x|| ||eps||. The index keeps max ||delta|| over its rows (stats[2]); prep_queries_i8_kernel turns it into the per-query E.
// Scale per tile, not per row: the rows of a tile compete inside key units, and with a common scale the selection network runs on the
// raw integer sums (one conversion per accumulator more than the fp16 epilogue). A per-row scale needs a second per-row LDS stream
// in the L2 epilogue, which drove hipcc into spilling 260-600 registers, in-flight query fragments among them. The price: a row
// is quantised as coarsely as the largest component of its 255 neighbours demands — measured, so still rigorous.
// Layout: the fp16 shadow's, byte for byte — [256-row tile][64-byte slab][row][64 bytes], a K step (two slabs, 128 bytes per row) now
// holds 128 dimensions; ld8 = dimensions padded to 256 (the query-stationary tile runs K steps in pairs).
__device__ __forceinline__ long tiled_off8(long row, int k, int ld8) {
    const long tile = row >> 8; const int r = (int)(row & 255);
    return ((tile * (ld8 >> 6) + (k >> 6)) * 256 + r) * 64 + (k & 63);
}
__device__ __forceinline__ unsigned pack4_i8(float a, float b, float c, float d) {
    return ((unsigned)(int)a & 0xFFu) | (((unsigned)(int)b & 0xFFu) << 8) | (((unsigned)(int)c & 0xFFu) << 16) | (((unsigned)(int)d & 0xFFu) << 24);
}
// One workgroup per tile, (re)quantising ALL rows the tile holds: rows [tile * 256, min(n, tile * 256 + 256)) of X (X = row 0 of the index).
__global__ __launch_bounds__(256) void to_i8_tiles_kernel(const float* __restrict__ X, long n, int ld, signed char* __restrict__ X8, int ld8, long tile0,
                                                          float* __restrict__ st /*scale per tile*/, unsigned* __restrict__ stats /*[2] = max ||delta||^2 bits*/) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long tile = tile0 + blockIdx.x, r0 = tile * 256, r1 = r0 + 256 < n ? r0 + 256 : n;
    float amax = 0.0f;
    for (long row = r0 + w; row < r1; row += 4) {
        const float* x = X + row * (long)ld;                   // ld is a multiple of 32, rows are 16-byte aligned
        for (int i0 = lane * 4; i0 < ld; i0 += 256) {
            const f32x4v v = *reinterpret_cast<const f32x4v*>(x + i0);
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    if (lane == 0) red[w] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float s = amax > 0.0f ? amax / 127.0f : 1.0f;
    float e2max = 0.0f;
    for (long row = r0 + w; row < r1; row += 4) {
        const float* x = X + row * (long)ld;
        float e2 = 0.0f;
        for (int i0 = lane * 16; i0 < ld8; i0 += 1024) {
            u32x4 out;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f32x4v v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (i0 + 4 * j < ld) v = *reinterpret_cast<const f32x4v*>(x + i0 + 4 * j);
                float cq[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    cq[e] = fminf(fmaxf(rintf(v[e] / s), -127.0f), 127.0f);
                    const float dl = v[e] - s * cq[e];
                    e2 += dl * dl;
                }
                out[j] = pack4_i8(cq[0], cq[1], cq[2], cq[3]);
            }
            *reinterpret_cast<u32x4*>(X8 + tiled_off8(row, i0, ld8)) = out;
        }
        for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off, 64);
        e2max = fmaxf(e2max, e2);
    }
    if (threadIdx.x == 0) st[tile] = s;
    if (lane == 0 && stats) atomicMax(&stats[2], __float_as_uint(e2max));
}
void launch_to_i8_tiles(Ctx* c, const float* X, int64_t n, int ld, void* X8, int ld8, int64_t tile0, float* st, uint32_t* stats) {
    const int64_t tiles = ceil_div(n, 256) - tile0;
    if (tiles <= 0) return;
    ProfScope ps(c, "to_i8_tiles");
    to_i8_tiles_kernel<<<dim3((unsigned)tiles), dim3(256), 0, c->stream>>>(X, n, ld, (signed char*)X8, ld8, tile0, st, stats);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// the scan kernel
// ------------------------------------------------------------------------------------------------
constexpr int FB_M = 256;     // corpus rows per workgroup tile
#ifndef FAST_ROW_AUX
#define FAST_ROW_AUX 2    // cache policy of the streamed row pieces (LDS-DMA aux immediate): 2 = nt (non-temporal) — the shadow is read once per scan and should not displace the query slab and the keys in L2; narrow tile 0.287 -> 0.270 ms (71 % of 8 TB/s), wide tile unchanged (0 / 1 = sc0 / 3: 0.287 / 0.292 / 0.27, tools/scan_check)
#endif
constexpr int FB_UNIT = 128;  // rows per key unit (a wave row group): 2 emitted keys + 1 bound per (query, unit)
constexpr int FB_N = 256;     // queries per tile (the whole batch)
constexpr int FB_K = 64;      // halves per K step (128 bytes per row)
constexpr int FB_THREADS = 512;

// byte offset of 16-byte slot `kslot` (0..7) of row `row` inside a [rows][64 halves] LDS tile with the
// XOR swizzle that makes the 16-lane ds_read_b128 groups hit 16 distinct 16-byte bank slots.
__device__ __forceinline__ int swz_off(int row, int kslot) { return row * 128 + ((kslot ^ ((row >> 1) & 7)) << 4); }

// keep the three smallest of (t0 <= t1 <= t2) U {a}, branch-free
__device__ __forceinline__ void ins3(float& t0, float& t1, float& t2, float a) {
    const float n2 = __builtin_amdgcn_fmed3f(t1, t2, a);
    const float n1 = __builtin_amdgcn_fmed3f(t0, t1, a);
    t0 = fminf(t0, a); t1 = n1; t2 = n2;
}

// ---- epilogue shared by the scan kernels: waves as WM (rows) x NWN (queries), wave tile MB*32 rows x NB*32 queries, acc[mb][nb] ----
template <int MODE, int MB = 4, int NB = 2, int NWN = 4>
__device__ __forceinline__ void scan_epilogue(f32x16 (&acc)[MB][NB], unsigned char* smem, long tile, long row0, long n,
                                              const float* __restrict__ rn, const float* __restrict__ qn, const unsigned char* __restrict__ elig,
                                              float* __restrict__ S0, long ldS, float* __restrict__ bound, long ldB) {
    constexpr int WM = FB_M / (MB * 32);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / NWN, wn = wid % NWN, khalf = lane >> 5;
    // per (query, key unit = row group of the tile): the two smallest packed keys + the third smallest (bound).
    // C layout of the 32x32 MFMA: column = lane & 31 (query), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    const float INF = __builtin_inff();
    const int lane_rowbits = 4 * khalf + MB * 32 * wm;   // the row-in-tile bits that depend on the lane / wave (disjoint from rconst's)
    const long nvalid = n - row0;
    const bool check = (nvalid < FB_M) || (elig != nullptr);   // workgroup-uniform
    // per-lane mask of usable rows (bit mb*16+e), only built on the slow path
    unsigned long long okmask = ~0ull;
    if (check) {
        okmask = 0ull;
        for (int mb = 0; mb < MB; mb++)
            for (int e = 0; e < 16; e++) {
                const long r = mb * 32 + (e & 3) + 8 * (e >> 2) + lane_rowbits;
                bool ok = r < nvalid;
                if (ok && elig) ok = elig[row0 + r] != 0;
                if (ok) okmask |= 1ull << (mb * 16 + e);
            }
    }
    float rnv[MODE == 1 ? MB * 16 : 1];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int mb = 0; mb < MB; mb++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                long r = row0 + mb * 32 + (e & 3) + 8 * (e >> 2) + lane_rowbits;
                rnv[mb * 16 + e] = rn[r < n ? r : n - 1];
            }
    }
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        const int q = wn * (NB * 32) + nb * 32 + (lane & 31);
        float qnv = 0.0f;
        if constexpr (MODE == 1) qnv = qn[q];
        float t0 = INF, t1 = INF, t2 = INF;
#pragma unroll
        for (int mb = 0; mb < MB; mb++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int rconst = mb * 32 + (e & 3) + 8 * (e >> 2);      // compile-time part of the row
                float a;
                if constexpr (MODE == 0) a = 1.0f - acc[mb][nb][e];
                else a = (qnv + rnv[mb * 16 + e]) - 2.0f * acc[mb][nb][e];
                a = fmaxf(a, 0.0f);
                float key = __uint_as_float((__float_as_uint(a) & 0xFFFFFF00u) | (unsigned)rconst);
                if (check) key = ((okmask >> (mb * 16 + e)) & 1ull) ? key : INF;
                ins3(t0, t1, t2, key);
            }
        }
        // add the lane-dependent row bits to the survivors; inf stays inf
        auto addbits = [&](float v) { return v == INF ? v : __uint_as_float(__float_as_uint(v) | (unsigned)lane_rowbits); };
        t0 = addbits(t0); t1 = addbits(t1); t2 = addbits(t2);
        // merge with the other half-wave (rows +4): exchange triples across lane ^ 32
        const float o0 = __shfl_xor(t0, 32, 64), o1 = __shfl_xor(t1, 32, 64), o2 = __shfl_xor(t2, 32, 64);
        ins3(t0, t1, t2, o0); ins3(t0, t1, t2, o1); ins3(t0, t1, t2, o2);
        // one key unit = the MB*32 rows of this wave's row group: two smallest keys + the third as the unit's bound
        if (lane < 32) {
            const long u = tile * WM + wm;
            S0[(long)q * ldS + 2 * u] = t0;
            S0[(long)q * ldS + 2 * u + 1] = t1;
            bound[(long)q * ldB + u] = t2;
        }
    }
}

// MODE 0: cosine   key = max(0, 1 - s)
// MODE 1: L2 family key = max(0, qn[q] + rn[row] - 2 s)
template <int MODE>
__global__ __launch_bounds__(FB_THREADS) void flat_scan_f16_kernel(const _Float16* __restrict__ Xh, long n, int ldh,
                                                                   const _Float16* __restrict__ Qh /*256 x ldh*/,
                                                                   const float* __restrict__ rn, const float* __restrict__ qn,
                                                                   const unsigned char* __restrict__ elig,
                                                                   float* __restrict__ S0 /*[256][ldS]: 2 keys per tile*/, long ldS,
                                                                   float* __restrict__ bound /*[256][ldB]*/, long ldB, long n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [buf][X 32 KiB | Q 32 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    // XCD-aware tile order: contiguous chunks of tiles per XCD keep an XCD's L2 working on neighbouring rows
    long tile;
    {
        const long L = blockIdx.x, nx = 8;
        const long q = n_tiles / nx, r = n_tiles % nx, xcd = L % nx, idx = L / nx;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        if (idx >= (xcd < r ? q + 1 : q)) return;
    }
    const long row0 = tile * FB_M;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    // ---- staging: each wave moves 4 X pieces + 4 Q pieces (8 rows x 128 B each) per K step ----
    const int prow = lane >> 3, pslot = lane & 7;
    const char* xsrc[4]; const char* qsrc[4]; int ldsoff[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = (wid * 4 + i) * 8 + prow;                 // row inside the tile (0..255)
        const int ks = pslot ^ ((r >> 1) & 7);                  // logical slot stored at this physical slot
        // tiled shadow: a 64-wide K step = two consecutive 16 KiB slabs (256 rows x 64 B); rows past n are zero-filled padding
        xsrc[i] = reinterpret_cast<const char*>(Xh) + ((tile * (long)(ldh >> 5) + (ks >> 2)) * 256 + r) * 64 + (ks & 3) * 16;
        qsrc[i] = reinterpret_cast<const char*>(Qh + (long)r * ldh) + ks * 16;
        ldsoff[i] = (wid * 4 + i) * 8 * 128;                    // wave-uniform LDS base of the piece
    }
    auto stage = [&](int buf, int kt) {
        unsigned char* xb = smem + buf * 65536;
        unsigned char* qb = xb + 32768;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i] + (long)kt * (256 * 128)),
                                             (__attribute__((address_space(3))) void*)(xb + ldsoff[i]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qsrc[i] + (long)kt * 128),
                                             (__attribute__((address_space(3))) void*)(qb + ldsoff[i]), 16, 0, 0);
        }
    };

    const int nk = ldh / FB_K;
    stage(0, 0);
    __syncthreads();   // compiler drains vmcnt before the barrier (LDS-DMA counts on vmcnt)
    const int arow = wm * 128 + (lane & 31), brow = wn * 64 + (lane & 31), khalf = lane >> 5;
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
        const unsigned char* xb = smem + buf * 65536;
        const unsigned char* qb = xb + 32768;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            half8 a[4], b[2];
#pragma unroll
            for (int mb = 0; mb < 4; mb++) a[mb] = *reinterpret_cast<const half8*>(xb + swz_off(arow + mb * 32, ks * 2 + khalf));
#pragma unroll
            for (int nb = 0; nb < 2; nb++) b[nb] = *reinterpret_cast<const half8*>(qb + swz_off(brow + nb * 32, ks * 2 + khalf));
#pragma unroll
            for (int mb = 0; mb < 4; mb++)
#pragma unroll
                for (int nb = 0; nb < 2; nb++) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb], b[nb], acc[mb][nb], 0, 0, 0);
        }
        __syncthreads();
    }
    scan_epilogue<MODE>(acc, smem, tile, row0, n, rn, qn, elig, S0, ldS, bound, ldB);
}
// ------------------------------------------------------------------------------------------------
// narrow variant for batches of at most 64 queries (the reference's API runs ONE query per Execute()): 256 rows x 64 queries per
// workgroup, 8 waves as 4 (rows) x 2 (queries), wave tile 64 x 32 (32 accumulators), 2 x 40 KiB of LDS — two workgroups share a
// CU. A quarter of the MFMA work and of the query staging of the 256-query tile; key units are the 64-row wave groups.
// ------------------------------------------------------------------------------------------------
constexpr int FN_N = 64, FN_UNIT = 64, FN_STAGE = 32768 + 8192;
// I8: the int8 shadow (per-tile scale) — same bytes per K step, 128 dimensions instead of 64; `ldh` = row bytes / 2, Qh = row-major int8
// queries (64 x ld8), sx / sq = tile / query scales: the integer sums are scaled to scores before the shared epilogue.
template <int MODE, bool I8 = false>
__global__ __launch_bounds__(FB_THREADS) void flat_scan_f16_n64_kernel(const _Float16* __restrict__ Xh, long n, int ldh,
                                                                       const _Float16* __restrict__ Qh /*>= 64 x ldh*/,
                                                                       const float* __restrict__ rn, const float* __restrict__ qn,
                                                                       const unsigned char* __restrict__ elig,
                                                                       float* __restrict__ S0, long ldS, float* __restrict__ bound, long ldB, long n_tiles,
                                                                       const float* __restrict__ sx = nullptr, const float* __restrict__ sq = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [buf][X 32 KiB | Q 8 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    long tile;
    {
        const long L = blockIdx.x, nx = 8;
        const long q = n_tiles / nx, r = n_tiles % nx, xcd = L % nx, idx = L / nx;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        if (idx >= (xcd < r ? q + 1 : q)) return;
    }
    const long row0 = tile * FB_M;
    std::conditional_t<I8, i32x16, f32x16> acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][0][e] = 0;
    // staging: 32 row pieces (4 per wave) + 8 query pieces (1 per wave), 8 rows x 128 B each
    const int prow = lane >> 3, pslot = lane & 7;
    const char* xsrc[4]; int xdst[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = (wid * 4 + i) * 8 + prow;
        const int ks = pslot ^ ((r >> 1) & 7);
        xsrc[i] = reinterpret_cast<const char*>(Xh) + ((tile * (long)(ldh >> 5) + (ks >> 2)) * 256 + r) * 64 + (ks & 3) * 16;
        xdst[i] = (wid * 4 + i) * 8 * 128;
    }
    const int qr = wid * 8 + prow;
    const char* qsrc = reinterpret_cast<const char*>(Qh + (long)qr * ldh) + (pslot ^ ((qr >> 1) & 7)) * 16;
    const int qdst = 32768 + wid * 8 * 128;
    auto stage = [&](int buf, int kt) {
        unsigned char* sb = smem + buf * FN_STAGE;
#pragma unroll
        for (int i = 0; i < 4; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i] + (long)kt * (256 * 128)),
                                             (__attribute__((address_space(3))) void*)(sb + xdst[i]), 16, 0, FAST_ROW_AUX);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qsrc + (long)kt * 128),
                                         (__attribute__((address_space(3))) void*)(sb + qdst), 16, 0, 0);
    };
    const int nk = ldh / FB_K;
    stage(0, 0);
    __syncthreads();
    const int arow = wm * 64 + (lane & 31), brow = wn * 32 + (lane & 31), khalf = lane >> 5;
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
        const unsigned char* xb = smem + buf * FN_STAGE;
        const unsigned char* qb = xb + 32768;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            half8 a[2], b;
#pragma unroll
            for (int mb = 0; mb < 2; mb++) a[mb] = *reinterpret_cast<const half8*>(xb + swz_off(arow + mb * 32, ks * 2 + khalf));
            b = *reinterpret_cast<const half8*>(qb + swz_off(brow, ks * 2 + khalf));
#pragma unroll
            for (int mb = 0; mb < 2; mb++) {
                if constexpr (I8) acc[mb][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, a[mb]), __builtin_bit_cast(i32x4v, b), acc[mb][0], 0, 0, 0);
                else acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb], b, acc[mb][0], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if constexpr (I8) {
        const float scale = sx[tile] * sq[brow];          // C layout: the lane's column = query brow
        f32x16 accf[2][1];
#pragma unroll
        for (int mb = 0; mb < 2; mb++)
#pragma unroll
            for (int e = 0; e < 16; e++) accf[mb][0][e] = (float)acc[mb][0][e] * scale;
        scan_epilogue<MODE, 2, 1, 2>(accf, smem, tile, row0, n, rn, qn, elig, S0, ldS, bound, ldB);
    } else scan_epilogue<MODE, 2, 1, 2>(acc, smem, tile, row0, n, rn, qn, elig, S0, ldS, bound, ldB);
}
// ------------------------------------------------------------------------------------------------
// Query-stationary scan tile (round 2): the 256-query batch never goes through LDS.
//
// What limited the 2 x 4 tile above was the CU's LDS path: per 64-wide K step it had to take 64 KiB of LDS-DMA (32 KiB of
// corpus rows from HBM + 32 KiB of the SAME query slab again for every row tile) next to 192 KiB of fragment reads, and the
// second-dispatched half of the waves spent 2000+ cycles per step just issuing its DMA pieces (profiles/r01_flat_scan_investigation.txt).
// Here the eight waves split the QUERIES instead (wave w: queries 32w..32w+31 against all 256 rows, 8 accumulator tiles):
//   * a wave's query fragments are private, 4 KiB per K step, and come straight from L2 into registers
//     (global_load_dwordx4 from a fragment-ordered copy: one contiguous 1 KiB per instruction), one K step ahead;
//   * only the corpus rows use LDS: a 4-stage ring of 32 KiB slabs filled by LDS-DMA two to three K steps ahead, so the HBM
//     latency is covered by the ring and not by registers; half the DMA pieces per step (32 instead of 64);
//   * one barrier per K step, waits are counted (vmcnt(4): the newest slab stays in flight across the barrier);
//   * a wave owns whole (query, 128-row) key units, so the epilogue needs no cross-wave merge.
// Per K step a wave issues 32 MFMAs, 32 ds_read_b128, 4 DMA pieces and 4 query loads.
// ------------------------------------------------------------------------------------------------
// barrier that waits for this wave's LDS traffic only (not for the LDS-DMA of later slabs, which __syncthreads() would drain)
__device__ __forceinline__ void __syncthreads_lds_only() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int FQ_STAGES = 4, FQ_STAGE_BYTES = 32768;
#ifndef FQ_PRIO_FROM
#define FQ_PRIO_FROM 4
#endif

// slow path of the epilogues (last tile / soft deletes / filters): bit mb*16+e set if row mb*32 + (e&3) + 8*(e>>2) + rowbits of the
// tile is a usable candidate. Kept out of line so that its address arithmetic is not hoisted into the persistent K loop.
__device__ __attribute__((noinline)) unsigned long long unit_okmask(int rowbits, long nvalid, const unsigned char* __restrict__ elig, long row0) {
    unsigned long long okmask = 0ull;
    for (int mb = 0; mb < 4; mb++)
        for (int e = 0; e < 16; e++) {
            const long r = mb * 32 + (e & 3) + 8 * (e >> 2) + rowbits;
            bool ok = r < nvalid;
            if (ok && elig) ok = elig[row0 + r] != 0;
            if (ok) okmask |= 1ull << (mb * 16 + e);
        }
    return okmask;
}

// keep the three LARGEST of (t0 >= t1 >= t2) U {v}, branch-free
__device__ __forceinline__ void ins3max(float& t0, float& t1, float& t2, float v) {
    const float n2 = __builtin_amdgcn_fmed3f(t1, t2, v);
    const float n1 = __builtin_amdgcn_fmed3f(t0, t1, v);
    t0 = fmaxf(t0, v); t1 = n1; t2 = n2;
}

// Epilogue of the query-stationary tile: wave `wid` owns queries 32*wid .. 32*wid+31 for all 256 rows, i.e. two whole
// (query, 128-row) key units per lane pair (lane, lane ^ 32). Per accumulator 4 VALU (cosine) / 5 (L2 family):
//   cosine : the key 1 - s is monotone in the score s, so the selection network runs on the raw scores (three LARGEST, with
//            the row-in-unit packed into the low 8 mantissa bits) and only the three survivors are turned into keys;
//   L2     : t = rn[row] - 2 s (one fma; the shadow norms of the tile's rows sit in LDS), three SMALLEST of the packed t,
//            survivors get + qn.
// Containment is unchanged: two emitted keys are real rows' approximate distances, the third is a lower bound of every
// other row of the unit up to the packing perturbation (2^-15 relative to |s| resp. |t|, twice), which flat_post_kernel's
// tau carries as an absolute slack (FAST_PACK_SLACK).
// UR = rows per key unit: 128, or 64 where 128-row units would often hold three candidates (small indexes / large k: every such unit
// is rescored whole by the post stage).
// I8 (int8 shadow): the accumulators are the integer dot products of the codes; the score is s_T[tile] * s_q[query] * acc, both
// scales positive and common to the rows a lane compares: the selection runs on float(acc) (cosine; the survivors are scaled) resp.
// on 2 s_T s_q float(acc) - rn (L2): one conversion per accumulator more than the fp16 epilogue.
template <int MODE, bool CHECK, int UR, bool I8 = false>
__device__ __forceinline__ void scan_epilogue_q(std::conditional_t<I8, i32x16, f32x16> (&acc)[8], int wid, long tile, long row0, long n, const float* __restrict__ rn_lds /*256 floats, MODE 1*/,
                                                const float* __restrict__ qn, const unsigned char* __restrict__ elig,
                                                float* __restrict__ S0, long ldS, float* __restrict__ bound, long ldB, unsigned long long* etr = nullptr,
                                                const float* __restrict__ sq = nullptr, float st = 1.0f /*I8: tile scale*/) {
    // C layout of the 32x32 MFMA: column = lane & 31 (query), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); tile mb covers rows mb*32..
    const int lane = threadIdx.x & 63, khalf = lane >> 5;
    int q = wid * 32 + (lane & 31);
    // opaque to the optimiser: otherwise the per-lane output addresses (S0 + q ldS, bound + q ldB) are hoisted out of the persistent K loop,
    // live across it in 64-bit register pairs, spilled, and their reloads put a conservative s_waitcnt vmcnt(0) into the K steps
    asm volatile("" : "+v"(q));
    const float INF = __builtin_inff();
    const long nvalid = n - row0;
    float qnv = 0.0f, sqv = 1.0f;
    if constexpr (MODE == 1) qnv = qn[q];
    if constexpr (I8) sqv = sq[q] * st;
    const float sq2 = 2.0f * sqv;
    constexpr int NU = 256 / UR, MBU = UR / 32;
#pragma unroll
    for (int u = 0; u < NU; u++) {                              // key unit = UR rows (tiles mb = MBU*u .. MBU*u + MBU-1)
        if (etr && u < 2) etr[u * 2] = __builtin_amdgcn_s_memtime();
        const int lane_rowbits = 4 * khalf + UR * u;
        unsigned long long okmask = ~0ull;                      // CHECK: last tile / soft deletes / filters (a separate instantiation: the
        if constexpr (CHECK) okmask = unit_okmask(lane_rowbits, nvalid, elig, row0);   // per-element selects cost more than the network itself)
        // selection on v: cosine v = score (largest wins), L2 v = -(rn - 2 s) (largest wins as well: one network for both)
        float t0 = -INF, t1 = -INF, t2 = -INF;
#pragma unroll
        for (int mb = 0; mb < MBU; mb++) {
            // I8: keep the scheduler from pulling every block's scale loads and conversions to the top (the L2 form spilled 260 registers)
#pragma unroll
            for (int e4 = 0; e4 < 4; e4++) {
                f32x4v rnv;
                if constexpr (MODE == 1) rnv = *reinterpret_cast<const f32x4v*>(rn_lds + u * UR + mb * 32 + 8 * e4 + 4 * khalf);   // rows e&3 = 0..3 are consecutive
#pragma unroll
                for (int e1 = 0; e1 < 4; e1++) {
                    const int e = e4 * 4 + e1;
                    const int rconst = mb * 32 + e1 + 8 * e4;              // compile-time part of the row-in-unit (bit 2 = lane >> 5 is added to the survivors)
                    float v;
                    if constexpr (I8) {
                        if constexpr (MODE == 0) v = (float)acc[MBU * u + mb][e];
                        else v = __builtin_fmaf(sq2, (float)acc[MBU * u + mb][e], -rnv[e1]);
                    } else if constexpr (MODE == 0) v = acc[MBU * u + mb][e];
                    else v = __builtin_fmaf(2.0f, acc[MBU * u + mb][e], -rnv[e1]);
                    v = __uint_as_float((__float_as_uint(v) & 0xFFFFFF00u) | (unsigned)rconst);
                    if constexpr (CHECK) v = ((okmask >> (mb * 16 + e)) & 1ull) ? v : -INF;
                    ins3max(t0, t1, t2, v);
                }
            }
        }
        if (etr && u < 2) etr[u * 2 + 1] = __builtin_amdgcn_s_memtime();
        const float o0 = __shfl_xor(t0, 32, 64), o1 = __shfl_xor(t1, 32, 64), o2 = __shfl_xor(t2, 32, 64);
        const unsigned mybit = (unsigned)(4 * khalf), otherbit = (unsigned)(4 * (khalf ^ 1));
        auto tag = [&](float v, unsigned bit) { return v == -INF ? v : __uint_as_float(__float_as_uint(v) | bit); };
        t0 = tag(t0, mybit); t1 = tag(t1, mybit); t2 = tag(t2, mybit);
        ins3max(t0, t1, t2, tag(o0, otherbit)); ins3max(t0, t1, t2, tag(o1, otherbit)); ins3max(t0, t1, t2, tag(o2, otherbit));
        if (lane < 32) {
            auto to_key = [&](float v) {
                if (v == -INF) return INF;
                const unsigned row = __float_as_uint(v) & 0xFFu;
                const float clean = __uint_as_float(__float_as_uint(v) & 0xFFFFFF00u);
                float a;
                if constexpr (MODE == 0) a = 1.0f - (I8 ? sqv * clean : clean); else a = qnv - clean;      // clean = -(rn - 2 s)
                a = fmaxf(a, 0.0f);
                return __uint_as_float((__float_as_uint(a) & 0xFFFFFF00u) | row);
            };
            const long un = tile * NU + u;
            S0[(long)q * ldS + 2 * un] = to_key(t0);
            S0[(long)q * ldS + 2 * un + 1] = to_key(t1);
            bound[(long)q * ldB + un] = to_key(t2);
        }
    }
}

// Persistent form: 256-ish workgroups (one per CU), each streams its share of the row tiles as ONE pipeline over (tile, K step)
// pairs — the slab ring and the counted waits run straight across tile boundaries, so the first slabs of tile T+1 are in flight
// while the waves run tile T's last K steps and its epilogue (a one-shot tile exposes ~4000 cycles of HBM latency before its
// first MFMA: s_memtime trace in profiles/r02_flat_scan_investigation.txt).
// I8: the int8 shadow (to_i8_rows_kernel) on v_mfma_i32_32x32x32_i8 — the data path is byte-identical (a K step of 128 bytes per row
// holds 128 dimensions instead of 64; `ldh` is then the row's BYTES / 2), half the K steps per tile.
template <int MODE, int UR, bool I8 = false>
__global__ __launch_bounds__(FB_THREADS) void flat_scan_q8_kernel(const _Float16* __restrict__ Xh, long n, int ldh,
                                                                  const _Float16* __restrict__ QF /*fragment-ordered queries*/,
                                                                  const float* __restrict__ rn, const float* __restrict__ qn,
                                                                  const unsigned char* __restrict__ elig,
                                                                  float* __restrict__ S0, long ldS, float* __restrict__ bound, long ldB, long n_tiles,
                                                                  unsigned long long* __restrict__ trace /*nullable: s_memtime stamps of workgroup 8 (tools/scan_check)*/,
                                                                  const float* __restrict__ sx = nullptr /*I8: tile scales*/, const float* __restrict__ sq = nullptr /*I8: query scales*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // FQ_STAGES x 32 KiB row slabs (+ 1 KiB of row norms, MODE 1)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware static split: workgroup b runs on XCD b % 8; every XCD owns a contiguous range of tiles, its workgroups take them round-robin
    const long nx = 8, xcd = blockIdx.x % nx, wgx = blockIdx.x / nx, wgs_per_xcd = gridDim.x / nx;
    const long tq = n_tiles / nx, trem = n_tiles % nx;
    const long xbase = xcd < trem ? xcd * (tq + 1) : trem * (tq + 1) + (xcd - trem) * tq, xcount = xcd < trem ? tq + 1 : tq;
    const long my_tiles = wgx < xcount ? (xcount - wgx + wgs_per_xcd - 1) / wgs_per_xcd : 0;
    if (my_tiles == 0) return;
    const int nk = ldh / FB_K;
    const long total = my_tiles * nk;                          // global steps of this workgroup
    float* rn_lds = reinterpret_cast<float*>(smem + FQ_STAGES * FQ_STAGE_BYTES);

    std::conditional_t<I8, i32x16, f32x16> acc[8];
    // ---- row slabs: each wave moves 4 pieces (8 rows x 128 B) per K step, XOR swizzle on the source side ----
    const int prow = lane >> 3, pslot = lane & 7;
    unsigned poff[4]; int ldsoff[4];                          // byte offset of the piece inside a (tile, K step) slab pair / inside the ring stage
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = (wid * 4 + i) * 8 + prow;
        const int ks = pslot ^ ((r >> 1) & 7);
        poff[i] = (unsigned)(((ks >> 2) * 256 + r) * 64 + (ks & 3) * 16);
        ldsoff[i] = (wid * 4 + i) * 8 * 128;
    }
    const long tile_bytes = (long)(ldh >> 5) * 256 * 64;       // fp16 shadow bytes of one 256-row tile
    auto tile_of = [&](long j) { return xbase + wgx + j * wgs_per_xcd; };
    // slab of a global step -> ring stage (step & 3). The source address of the slab THREE steps ahead is carried incrementally
    // (no division in the K loop): `pre_ptr` walks K steps inside a tile and jumps to this workgroup's next tile at the wrap.
    const char* pre_ptr = reinterpret_cast<const char*>(Xh) + tile_of(0) * tile_bytes;
    int pre_kt = 0;
    const long tile_jump = wgs_per_xcd * tile_bytes - (long)nk * (256 * 128);        // from the end of a tile's slabs to the next tile of this workgroup
    auto pre_advance = [&]() { pre_ptr += 256 * 128; if (++pre_kt == nk) { pre_kt = 0; pre_ptr += tile_jump; } };
    auto stage_piece = [&](long g, int i) {
        unsigned char* xb = smem + (g & (FQ_STAGES - 1)) * FQ_STAGE_BYTES;
        unsigned po = poff[i];
        asm volatile("" : "+v"(po));          // the 64-bit lane address is formed here, per piece: hoisted, the four zero-extended pairs cost 8 registers
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pre_ptr + po), (__attribute__((address_space(3))) void*)(xb + ldsoff[i]), 16, 0, FAST_ROW_AUX);
    };
    // ---- query fragments: [wave][K step][ks][lane] 16-byte pieces, 4 KiB contiguous per (wave, K step); the same for every tile ----
    const char* qbase = reinterpret_cast<const char*>(QF) + ((long)wid * nk) * 4096 + lane * 16;
#define FQ_LOADQ(DST, P, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=v"(DST) : "v"(P) : "memory")
    // wait-only statements: the fragment registers are INPUTS, so the compiler has nothing to merge or copy in front of the wait;
    // the sched_barrier behind it keeps the register-only MFMAs below the wait
#define FQ_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" :: "v"(qcur[0]), "v"(qcur[1]), "v"(qcur[2]), "v"(qcur[3]) : "memory")
    const int arow = lane & 31, khalf = lane >> 5;
    u32x4 qa[4], qb[4];
    long cur_tile = tile_of(0);
    int rn_par = 0;
    const bool tr = trace != nullptr && blockIdx.x == 8 && lane == 0;
    auto stamp = [&](long g, int slot) { if (tr && g < 2 * nk + 2) trace[(wid * 32 + g) * 4 + slot] = __builtin_amdgcn_s_memtime(); };

    // prologue: X(0), Q(0), X(1), X(2); afterwards pre_ptr points at slab 3
#pragma unroll
    for (int i = 0; i < 4; i++) stage_piece(0, i);
    pre_advance();
    { const char* qp = qbase; FQ_LOADQ(qa[0], qp, 0); FQ_LOADQ(qa[1], qp, 1024); FQ_LOADQ(qa[2], qp, 2048); FQ_LOADQ(qa[3], qp, 3072); }
    if (1 < total) {
#pragma unroll
        for (int i = 0; i < 4; i++) stage_piece(1, i);
    }
    pre_advance();
    if (2 < total) {
#pragma unroll
        for (int i = 0; i < 4; i++) stage_piece(2, i);
    }
    pre_advance();

    // One global step. Every step issues exactly 4 query loads (the last ones re-read: an inline-asm load on one side of a branch
    // makes the compiler merge "loaded" and "not loaded" registers with copies placed before the data has landed) and then, if
    // there is one, the 4 DMA pieces of the slab three steps ahead — the counted waits rely on this order.
    auto step = [&](long g, int kt, u32x4 (&qcur)[4], u32x4 (&qnext)[4]) {
        stamp(g, 0);
        // X(g) and Q(g) have landed once at most the slabs issued AFTER Q(g) are outstanding: X(1), X(2) for g = 0, X(g+2) otherwise
        const int newer = g == 0 ? ((1 < total) + (2 < total)) : (g + 2 < total ? 1 : 0);
        if (newer == 2) FQ_WAIT(8); else if (newer == 1) FQ_WAIT(4); else FQ_WAIT(0);
        __builtin_amdgcn_sched_barrier(0);
        stamp(g, 1);
        if (kt == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][e] = 0;
            if constexpr (MODE == 1) {
                // the tile's 256 row norms go straight into LDS (one 4-byte LDS-DMA piece per lane of waves 0..3) while its K steps run; older
                // than this step's query loads, so the next step's counted wait covers it. Two buffers: the previous tile's epilogue may
                // still be reading the other one in a slower wave.
                if (wid < 4) {
                    const long r = cur_tile * FB_M + tid;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rn + (r < n ? r : n - 1)),
                                                     (__attribute__((address_space(3))) void*)(rn_lds + rn_par * 256 + wid * 64), 4, 0, 0);
                }
            }
        }
        const unsigned char* xb = smem + (g & (FQ_STAGES - 1)) * FQ_STAGE_BYTES;
        const int ktn = kt + 1 < nk ? kt + 1 : 0;                      // next step's K index (wraps into the next tile)
        const char* qp = qbase + (long)(g + 1 < total ? ktn : kt) * 4096;
        const bool more = g + 3 < total;                              // workgroup-uniform
        // Row fragments double-buffered in registers in groups of four (group gi = rows 128*(gi&1).., sub-step gi>>1): the reads of
        // group gi+1 are in flight under the 4 MFMAs of group gi. One vector-memory instruction follows every group instead of a
        // burst at the top of the step, where all eight waves collide on the address path (~150 cycles per instruction for the
        // younger half of the waves).
        half8 a[2][8];                                                 // sub-step ks+1's eight fragments load under sub-step ks's eight MFMAs
#pragma unroll
        for (int mb = 0; mb < 8; mb++) a[0][mb] = *reinterpret_cast<const half8*>(xb + swz_off(arow + mb * 32, khalf));
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const half8 b = __builtin_bit_cast(half8, qcur[ks]);
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int m2 = 0; m2 < 4; m2++) {
                    const int mb = h * 4 + m2;
                    if (ks < 3) a[(ks + 1) & 1][mb] = *reinterpret_cast<const half8*>(xb + swz_off(arow + mb * 32, (ks + 1) * 2 + khalf));
                    if constexpr (I8) acc[mb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, a[ks & 1][mb]), __builtin_bit_cast(i32x4v, b), acc[mb], 0, 0, 0);
                    else acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks & 1][mb], b, acc[mb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int gi = ks * 2 + h;
                if (gi == 0) FQ_LOADQ(qnext[0], qp, 0);
                else if (gi == 1) FQ_LOADQ(qnext[1], qp, 1024);
                else if (gi == 2) FQ_LOADQ(qnext[2], qp, 2048);
                else if (gi == 3) FQ_LOADQ(qnext[3], qp, 3072);
                else if (more) stage_piece(g + 3, gi - 4);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        pre_advance();
        stamp(g, 2);
        if (kt == nk - 1) {                                            // tile finished: its epilogue runs while the next tile's slabs arrive
            const long tile = cur_tile, row0 = tile * FB_M;
            cur_tile += wgs_per_xcd;
            const float* rn_tile = rn_lds + rn_par * 256;             // MODE 1: landed and visible since the tile's second K step
            rn_par ^= 1;
            float stv = 1.0f;
            if constexpr (I8) stv = sx[tile];
            unsigned long long* etr = (tr && g < 2 * nk + 2) ? trace + 8 * 32 * 4 + wid * 8 : nullptr;
            if ((n - row0 < FB_M) || elig != nullptr) scan_epilogue_q<MODE, true, UR, I8>(acc, wid, tile, row0, n, rn_tile, qn, elig, S0, ldS, bound, ldB, etr, sq, stv);   // workgroup-uniform
            else scan_epilogue_q<MODE, false, UR, I8>(acc, wid, tile, row0, n, rn_tile, qn, elig, S0, ldS, bound, ldB, etr, sq, stv);
            stamp(g, 3);
        }
    };
    // nk is even here (the launcher sends odd K-step counts to the 2 x 4 tile), so steps come in pairs with static fragment buffers
    // the second-dispatched half of the waves loses every issue arbitration to the older half (priority, then age): one static
    // s_setprio for it evens out the two waves of each SIMD (MI355X_MICROARCH.md "static priority for the younger half")
    static_assert(FB_THREADS == 512, "wave 4..7 = the younger half");
    if (wid >= FQ_PRIO_FROM) __builtin_amdgcn_s_setprio(1);
    int kt = 0;
    for (long g = 0; g < total; g += 2) {
        step(g, kt, qa, qb);
        step(g + 1, kt + 1, qb, qa);
        kt += 2; if (kt == nk) kt = 0;
    }
    // the redundant query loads of the last step must be home before the wave may end
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(qa[0]), "v"(qa[1]), "v"(qa[2]), "v"(qa[3]), "v"(qb[0]), "v"(qb[1]), "v"(qb[2]), "v"(qb[3]) : "memory");
#undef FQ_WAIT
#undef FQ_LOADQ
}

// What limits this kernel (profiles/r01_flat_scan_investigation.txt): of a K step's ~3900 cycles the 2 x 32 MFMAs of a SIMD
// need 2048. The CU accepts the step's 64 LDS-DMA pieces only over ~2400 cycles while fragment reads are in flight; the four
// waves that win arbitration finish their pieces after ~800 cycles and run their MFMAs while the other four are still issuing,
// which then compute while the first four wait at the barrier: the MFMA phases of the two waves of a SIMD never overlap.
// Deeper rings, ping-pong phases, interleaved issue, dedicated loader waves (1 KiB pieces cost 22-75 cycles CU-wide
// depending on how many waves issue, tools/dma_issue_probe.hip, and 3-5x that next to ds_read traffic), 16-wave workgroups,
// persistent tiles, wave priorities and classic register staging (global_load -> ds_write_b128, no LDS-DMA: 0.59 ms) were all
// built and measured in round 1: 0.51-0.63 ms against 0.51-0.55 for this one.
unsigned long long* g_scan_trace = nullptr;   // set by tools/scan_check.hip only
static bool scan_uses_q8(int ldh) {           // the query-stationary tile takes even K-step counts; the few odd ones (and COMET_SCAN_VARIANT=1) keep the 2 x 4 tile
    static const int variant0 = [] { const char* e = getenv("COMET_SCAN_VARIANT"); return e ? atoi(e) : 0; }();
    const char* rt = getenv("COMET_SCAN_VARIANT_RT");           // tools/scan_check.hip switches variants inside one process
    const int variant = rt ? atoi(rt) : variant0;
    return variant == 0 && ((ldh / FB_K) & 1) == 0;
}
void launch_flat_scan_f16(Ctx* c, int mode, const void* Xh, int64_t n, int ldh, const void* Qh, int nq_used, const float* rn, const float* qn,
                          const uint8_t* elig, float* S0, int64_t ldS, float* bound, int64_t ldB, int unit_rows) {
    const long n_tiles = ceil_div(n, FB_M);
    // (timed through Ctx::launch_timed: when a bench times this scope the events ride on the kernel's own dispatch — no event-record
    // packets between the step's kernels)
    const char* scope = nq_used <= FN_N ? "flat_scan_f16_n64" : "flat_scan_f16";
    if (nq_used <= FN_N) {      // S0 / bound are laid out for 64-row units in this case (flat_fast_unit_rows(nq))
        const size_t ldsn = 2 * FN_STAGE;
        const long gridn = round_up(n_tiles, 8);
        auto gon = [&](auto kernel) {
            HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn));
            c->launch_timed(scope, kernel, dim3((unsigned)gridn), dim3(FB_THREADS), ldsn, (const _Float16*)Xh, (long)n, ldh, (const _Float16*)Qh, rn, qn, (const unsigned char*)elig, S0, (long)ldS, bound, (long)ldB, n_tiles,
                            (const float*)nullptr, (const float*)nullptr);
        };
        if (mode == 0) gon(flat_scan_f16_n64_kernel<0, false>); else gon(flat_scan_f16_n64_kernel<1, false>);
        LAUNCH_CHECK();
        return;
    }
    if (scan_uses_q8(ldh)) {          // query-stationary tile: rows through a 4-stage LDS ring, query fragments straight from L2
        const size_t ldsq = (size_t)FQ_STAGES * FQ_STAGE_BYTES + 2048;
        const long gridq = std::min<long>(round_up(n_tiles, 8), (long)round_up(c->prop.multiProcessorCount, 8));   // persistent: one workgroup per CU
        const _Float16* QF = (const _Float16*)Qh + (size_t)FB_N * ldh;
        auto go = [&](auto kernel) {
            HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq));
            c->launch_timed(scope, kernel, dim3((unsigned)gridq), dim3(FB_THREADS), ldsq, (const _Float16*)Xh, (long)n, ldh, QF, rn, qn, (const unsigned char*)elig, S0, (long)ldS, bound, (long)ldB, n_tiles, g_scan_trace,
                            (const float*)nullptr, (const float*)nullptr);
        };
        if (unit_rows == 64) { if (mode == 0) go(flat_scan_q8_kernel<0, 64>); else go(flat_scan_q8_kernel<1, 64>); }
        else { if (mode == 0) go(flat_scan_q8_kernel<0, 128>); else go(flat_scan_q8_kernel<1, 128>); }
        LAUNCH_CHECK();
        return;
    }
    if (unit_rows != FB_UNIT) COMET_FAIL(COMET_ERR_INVALID_ARG, "the 2 x 4 scan tile emits 128-row key units");
    const size_t lds = 2 * 65536;
    const long grid = round_up(n_tiles, 8);
    auto go2 = [&](auto kernel) {
        HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->launch_timed(scope, kernel, dim3((unsigned)grid), dim3(FB_THREADS), lds, (const _Float16*)Xh, (long)n, ldh, (const _Float16*)Qh, rn, qn, (const unsigned char*)elig, S0, (long)ldS, bound, (long)ldB, n_tiles);
    };
    if (mode == 0) go2(flat_scan_f16_kernel<0>); else go2(flat_scan_f16_kernel<1>);
    LAUNCH_CHECK();
}
// The wide tile on the int8 shadow (more than 64 queries; ld8 a multiple of 256). Q8F: fragment-ordered int8 queries
// (prep_queries_i8_kernel), sx / sq: tile / query scales. Keys and bounds come out exactly as from the fp16 tile.
void launch_flat_scan_i8(Ctx* c, int mode, const void* X8, int64_t n, int ld8, const void* Q8F, const void* Q8R, int nq_used, const float* rn, const float* qn,
                         const float* sx, const float* sq, const uint8_t* elig, float* S0, int64_t ldS, float* bound, int64_t ldB, int unit_rows) {
    if ((ld8 & 255) != 0) COMET_FAIL(COMET_ERR_INVALID_ARG, "int8 shadow rows are padded to 256 bytes");
    const long n_tiles = ceil_div(n, FB_M);
    if (nq_used <= FN_N) {      // the narrow tile (64-row units)
        if (launch_flat_scan_qr(c, mode, X8, n, ld8, Q8F, nq_used, rn, qn, sx, sq, elig, S0, ldS, bound, ldB, unit_rows)) return;     // register-stationary narrow tile (kernels_scanq.hip)
        const size_t ldsn = 2 * FN_STAGE;
        const long gridn = round_up(n_tiles, 8);
        auto gon = [&](auto kernel) {
            HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsn));
            c->launch_timed("flat_scan_i8_n64", kernel, dim3((unsigned)gridn), dim3(FB_THREADS), ldsn, (const _Float16*)X8, (long)n, ld8 / 2, (const _Float16*)Q8R, rn, qn, (const unsigned char*)elig, S0, (long)ldS,
                            bound, (long)ldB, n_tiles, sx, sq);
        };
        if (mode == 0) gon(flat_scan_f16_n64_kernel<0, true>); else gon(flat_scan_f16_n64_kernel<1, true>);
        LAUNCH_CHECK();
        return;
    }
    if (launch_flat_scan_qr(c, mode, X8, n, ld8, Q8F, nq_used, rn, qn, sx, sq, elig, S0, ldS, bound, ldB, unit_rows)) return;         // register-stationary wide tile (kernels_scanq.hip)
    const size_t ldsq = (size_t)FQ_STAGES * FQ_STAGE_BYTES + 2048;
    const long gridq = std::min<long>(round_up(n_tiles, 8), (long)round_up(c->prop.multiProcessorCount, 8));
    auto go = [&](auto kernel) {
        HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq));
        c->launch_timed("flat_scan_i8", kernel, dim3((unsigned)gridq), dim3(FB_THREADS), ldsq, (const _Float16*)X8, (long)n, ld8 / 2, (const _Float16*)Q8F, rn, qn, (const unsigned char*)elig, S0, (long)ldS, bound,
                        (long)ldB, n_tiles, g_scan_trace, sx, sq);
    };
    if (unit_rows == 64) { if (mode == 0) go(flat_scan_q8_kernel<0, 64, true>); else go(flat_scan_q8_kernel<1, 64, true>); }
    else { if (mode == 0) go(flat_scan_q8_kernel<0, 128, true>); else go(flat_scan_q8_kernel<1, 128, true>); }
    LAUNCH_CHECK();
}
int flat_fast_tile_rows() { return FB_M; }
// Rows per key unit of a search: the narrow tile (<= 64 queries) always emits 64-row units; the wide tile emits 128-row units unless
// those would often hold three candidates — a unit whose third-smallest key passes tau is rescored whole by the post stage. With
// c ~ 1.5 k + 16 candidates per query spread over n rows a unit of 128 rows holds Poisson(lambda = 128 c / n) of them, so a query
// expands about (n / 128) * lambda^3 / 6 units: 0.01 at 1M rows and k = 100, 0.2 at 250k, 3 at 62.5k (measured 0.004, 0.12 and 1.3) —
// beyond 0.1 the scan switches to 64-row units (an eighth of the expansion work: a quarter of the expansions, half the rows each;
// twice the keys for the post stage to select from). Measured step, B = 256, K = 100, 128- vs 64-row units: 62.5k rows 0.172 vs
// 0.114 ms, 250k 0.211 vs 0.203, 500k 0.312 vs 0.319, 1M 0.543 vs 0.579.
int flat_fast_unit_rows(int nq, int64_t n, int64_t k, int ldh) {
    if (nq <= FN_N) return FN_UNIT;
    static const int forced = [] { const char* e = getenv("COMET_FLAT_UNIT"); return e ? atoi(e) : 0; }();
    if (!scan_uses_q8(ldh)) return FB_UNIT;                     // the 2 x 4 tile knows 128-row units only
    if (forced == 64 || forced == 128) return forced;
    const double keff = (k <= 0 || k > n) ? (double)n : (double)k, cand = 1.5 * keff + 16.0, lambda = 128.0 * cand / (double)std::max<int64_t>(n, 1);
    return ((double)n / 128.0) * lambda * lambda * lambda / 6.0 > 0.1 ? 64 : FB_UNIT;
}
int flat_fast_batch() { return FB_N; }

// ------------------------------------------------------------------------------------------------
// query side: fp16 copy (zero-padded to 256 rows), squared norms, rigorous error bound per query
// ------------------------------------------------------------------------------------------------
// mode 0 cosine: |approx - exact| <= E; mode 1 L2 family (squared space).
__global__ __launch_bounds__(256) void prep_queries_fast_kernel(const float* __restrict__ Qp, int B, int ld, int dim, _Float16* __restrict__ Qh, int ldh,
                                                                float* __restrict__ qn, float* __restrict__ err_abs, int mode, float xmax_norm2, int* __restrict__ stats4) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (stats4 && blockIdx.x == 0 && threadIdx.x < 4) stats4[threadIdx.x] = 0;   // the post stage accumulates into these
    if (q >= FB_N) return;
    _Float16* o = Qh + (long)q * ldh;
    // second copy in MFMA fragment order for the query-stationary scan: [wave = q / 32][K step][ks][lane = khalf * 32 + q % 32][8 halves]
    _Float16* qf = Qh + (long)FB_N * ldh;
    const int nk = ldh >> 6;
    float s = 0.0f;
    for (int i = lane; i < ldh; i += 64) {
        float v = (q < B && i < ld) ? Qp[(long)q * ld + i] : 0.0f;
        o[i] = (_Float16)v;
        qf[(((((long)(q >> 5) * nk + (i >> 6)) * 4 + ((i >> 4) & 3)) * 64) + ((i >> 3) & 1) * 32 + (q & 31)) * 8 + (i & 7)] = (_Float16)v;
        s += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) {
        qn[q] = s;
        const float nq = sqrtf(s) * 1.0001f, nx = sqrtf(xmax_norm2) * 1.0001f, d = (float)dim;
        // fp16 operand rounding (2^-11 relative each, both sides: 2^-10 with slack), subnormal floor 2^-24 per element,
        // fp32 accumulation on either side (d * 2^-23)
        float edot = (1.0f / 1024.0f + 2.0f * d * 1.2e-7f) * nq * nx + 6.0e-8f * sqrtf(d) * (nq + nx);
        float e = mode == 0 ? edot : 2.0f * edot + (d + 8.0f) * 1.2e-7f * (nq + nx) * (nq + nx);
        // key packing of the query-stationary scan: the row index replaces the low 8 mantissa bits of the score s (cosine) resp. of
        // rn - 2 s (L2 family) BEFORE the final subtraction: 2^-15 relative to those magnitudes, once for the emitted key and once
        // for the bound (FAST_PACK_SLACK; the older tiles pack the final distance, covered by the relative term in flat_post_kernel)
        e += 6.2e-5f * (mode == 0 ? nq * nx : nx * nx + 2.0f * nq * nx);
        err_abs[q] = 1.25f * e;
    }
}
void launch_prep_queries_fast(Ctx* c, const float* Qp, int B, int ld, int dim, void* Qh, int ldh, float* qn, float* err_abs, int mode, float xmax_norm2,
                              int32_t* stats4) {
    prep_queries_fast_kernel<<<dim3(FB_N / 4), dim3(256), 0, c->stream>>>(Qp, B, ld, dim, (_Float16*)Qh, ldh, qn, err_abs, mode, xmax_norm2, stats4);
    LAUNCH_CHECK();
}

// Distance.Preprocess(query) (the cosine normalisation of ingest_rows_wave_kernel: serial float32 norm, same order as the Go loop)
// and the fp16 / fragment-ordered / error-bound side of prep_queries_fast_kernel in ONE launch, a wave per query: a search step
// is a chain of short dependent kernels, and every link costs a ~6 us launch gap on top of its own time.
constexpr int PREPF_MAX_D = 2048;
__global__ __launch_bounds__(256) void prep_queries_fused_kernel(int metric, const float* __restrict__ src, int B, int d, float* __restrict__ Qp, int ld,
                                                                 int* __restrict__ zero_flag, _Float16* __restrict__ Qh, int ldh, float* __restrict__ qn,
                                                                 float* __restrict__ err_abs, int mode, float xmax_norm2, int* __restrict__ stats4) {
    extern __shared__ __attribute__((aligned(16))) float sq[];   // [4 waves][dpad]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (stats4 && blockIdx.x == 0 && threadIdx.x < 4) stats4[threadIdx.x] = 0;   // the post stage accumulates into these
    if (q >= FB_N) return;
    const int dpad = (d + 3) & ~3;
    float* my = sq + (long)w * dpad;
    const float* sp = src + (long)q * d;
    float scale = 1.0f; int zf = 0;
    const bool live = q < B;
    // a lane owns 8 consecutive dimensions: two 16-byte loads, and 16-byte stores of the fp16 row and of the fragment-ordered
    // copy (8 consecutive dimensions are contiguous there); rows are only 4-byte aligned when d % 4 != 0 -> scalar loads then
    const bool vec = (d & 3) == 0 && ((unsigned long long)src & 15ull) == 0ull;
    auto load8 = [&](int i0, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = 0.0f;
        if (!live) return;
        if (vec && i0 + 8 <= d) {
            const f32x4v a = *reinterpret_cast<const f32x4v*>(sp + i0), b = *reinterpret_cast<const f32x4v*>(sp + i0 + 4);
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) if (i0 + e < d) v[e] = sp[i0 + e];
        }
    };
    if (live && metric == COMET_COSINE) {
        for (int i0 = lane * 8; i0 < dpad; i0 += 512) {
            float v[8]; load8(i0, v);
#pragma unroll
            for (int e = 0; e < 8; e++) if (i0 + e < dpad) my[i0 + e] = v[e] * v[e];
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        float sum = 0.0f;
        if (lane == 0) {
#pragma unroll 8
            for (int i = 0; i < dpad; i += 4) {
                const f32x4v p = *reinterpret_cast<const f32x4v*>(&my[i]);
                sum = sum + p[0]; sum = sum + p[1]; sum = sum + p[2]; sum = sum + p[3];
            }
        }
        sum = __shfl(sum, 0, 64);
        const float norm = (float)__builtin_sqrt((double)sum);      // float32(math.Sqrt(float64(sum)))
        if (norm == 0.0f) zf = 1; else scale = 1.0f / norm;
    }
    _Float16* o = Qh + (long)q * ldh;
    _Float16* qf = Qh + (long)FB_N * ldh;
    const int nk = ldh >> 6;
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    float s = 0.0f;
    for (int i0 = lane * 8; i0 < ldh; i0 += 512) {               // ldh is a multiple of 64
        float v[8]; load8(i0, v);
        h8 hv;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if (metric == COMET_COSINE && !zf) v[e] = v[e] * scale;
            hv[e] = (_Float16)v[e];
            s += v[e] * v[e];
        }
        if (live && i0 < ld) {                                    // ld is a multiple of 32
            *reinterpret_cast<f32x4v*>(Qp + (long)q * ld + i0) = f32x4v{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4v*>(Qp + (long)q * ld + i0 + 4) = f32x4v{v[4], v[5], v[6], v[7]};
        }
        *reinterpret_cast<h8*>(o + i0) = hv;
        *reinterpret_cast<h8*>(qf + (((((long)(q >> 5) * nk + (i0 >> 6)) * 4 + ((i0 >> 4) & 3)) * 64) + ((i0 >> 3) & 1) * 32 + (q & 31)) * 8) = hv;
    }
    if (live) for (int i = ldh + lane; i < ld; i += 64) Qp[(long)q * ld + i] = 0.0f;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) {
        if (live && zero_flag) zero_flag[q] = zf;
        qn[q] = s;
        const float nq = sqrtf(s) * 1.0001f, nx = sqrtf(xmax_norm2) * 1.0001f, dd = (float)d;
        float edot = (1.0f / 1024.0f + 2.0f * dd * 1.2e-7f) * nq * nx + 6.0e-8f * sqrtf(dd) * (nq + nx);
        float e = mode == 0 ? edot : 2.0f * edot + (dd + 8.0f) * 1.2e-7f * (nq + nx) * (nq + nx);
        e += 6.2e-5f * (mode == 0 ? nq * nx : nx * nx + 2.0f * nq * nx);
        err_abs[q] = 1.25f * e;
    }
}
bool prep_queries_fused_ok(int dim) { return dim <= PREPF_MAX_D; }

// The int8 side of a query slice, a wave per query: Distance.Preprocess (raw queries: the cosine normalisation exactly as in
// prep_queries_fused_kernel; src == nullptr: Qp is preprocessed already), then q_i = s_q e_i + eps_i with one scale per query,
// the codes in MFMA fragment order [wave = q / 32][K step of 128][ks][lane = khalf * 32 + q % 32][16 codes], and the rigorous bound
//   |approx - exact| <= E = dx_max ||q_hat|| + ||x||_max ||eps||  (+ float32 evaluation on both sides, + key packing)
// from the MEASURED residual norms (dx_max: the index's largest row residual norm, to_i8_tiles_kernel).
__global__ __launch_bounds__(256) void prep_queries_i8_kernel(int metric, const float* __restrict__ src, int B, int d, float* __restrict__ Qp, int ld,
                                                              int* __restrict__ zero_flag, signed char* __restrict__ Q8F, signed char* __restrict__ Q8R /*nullable: row-major copy (narrow tile)*/, int ld8, float* __restrict__ sq,
                                                              float* __restrict__ qn, float* __restrict__ err_abs, int mode, float xmax_norm2, float dx_max,
                                                              int* __restrict__ stats4) {
    extern __shared__ __attribute__((aligned(16))) float sqm[];   // [4 waves][ld8]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (stats4 && blockIdx.x == 0 && threadIdx.x < 4) stats4[threadIdx.x] = 0;   // the post stage accumulates into these
    if (q >= FB_N) return;
    float* my = sqm + (long)w * ld8;
    const bool live = q < B;
    const float* sp = src ? src + (long)q * d : Qp + (long)q * ld;
    const int dsrc = src ? d : ld;                               // readable elements of the source row
    const bool vec = src ? ((d & 3) == 0 && ((unsigned long long)src & 15ull) == 0ull) : true;
    auto load4 = [&](int i0, float (&v)[4]) {
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = 0.0f;
        if (!live) return;
        if (vec && i0 + 4 <= dsrc) { const f32x4v a = *reinterpret_cast<const f32x4v*>(sp + i0); v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; }
        else {
#pragma unroll
            for (int e = 0; e < 4; e++) if (i0 + e < dsrc) v[e] = sp[i0 + e];
        }
    };
    float scale = 1.0f; int zf = 0;
    if (live && src && metric == COMET_COSINE) {                  // float32(math.Sqrt(float64(sum))) over the serial float32 sum of squares
        const int dpad = (d + 3) & ~3;
        for (int i0 = lane * 4; i0 < dpad; i0 += 256) {
            float v[4]; load4(i0, v);
            *reinterpret_cast<f32x4v*>(&my[i0]) = f32x4v{v[0] * v[0], v[1] * v[1], v[2] * v[2], v[3] * v[3]};
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        float sum = 0.0f;
        if (lane == 0) {
#pragma unroll 8
            for (int i = 0; i < dpad; i += 4) {
                const f32x4v p = *reinterpret_cast<const f32x4v*>(&my[i]);
                sum = sum + p[0]; sum = sum + p[1]; sum = sum + p[2]; sum = sum + p[3];
            }
        }
        sum = __shfl(sum, 0, 64);
        const float norm = (float)__builtin_sqrt((double)sum);
        if (norm == 0.0f) zf = 1; else scale = 1.0f / norm;
    }
    float s = 0.0f, amax = 0.0f;
    for (int i0 = lane * 4; i0 < ld8; i0 += 256) {
        float v[4]; load4(i0, v);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (src && metric == COMET_COSINE && !zf) v[e] = v[e] * scale;
            s += v[e] * v[e];
            amax = fmaxf(amax, fabsf(v[e]));
        }
        if (src && live && i0 < ld) *reinterpret_cast<f32x4v*>(Qp + (long)q * ld + i0) = f32x4v{v[0], v[1], v[2], v[3]};   // ld is a multiple of 32; beyond d: zeros
        *reinterpret_cast<f32x4v*>(&my[i0]) = f32x4v{v[0], v[1], v[2], v[3]};
    }
    if (src && live) for (int i = ld8 + lane; i < ld; i += 64) Qp[(long)q * ld + i] = 0.0f;
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); amax = fmaxf(amax, __shfl_xor(amax, off, 64)); }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const float sc = (amax > 0.0f && amax < __builtin_inff()) ? amax / 127.0f : 1.0f;
    const int nk = ld8 >> 7;
    float e2 = 0.0f, h2 = 0.0f;
    for (int i0 = lane * 16; i0 < ld8; i0 += 1024) {
        u32x4 out;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f32x4v v = *reinterpret_cast<const f32x4v*>(&my[i0 + 4 * j]);
            float cq[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                cq[e] = fminf(fmaxf(rintf(v[e] / sc), -127.0f), 127.0f);
                const float dq = sc * cq[e], dl = v[e] - dq;
                e2 += dl * dl; h2 += dq * dq;
            }
            out[j] = pack4_i8(cq[0], cq[1], cq[2], cq[3]);
        }
        *reinterpret_cast<u32x4*>(Q8F + ((((long)(q >> 5) * nk + (i0 >> 7)) * 4 + ((i0 >> 5) & 3)) * 64 + ((i0 >> 4) & 1) * 32 + (q & 31)) * 16) = out;
        if (Q8R) *reinterpret_cast<u32x4*>(Q8R + (long)q * ld8 + i0) = out;
    }
    for (int off = 32; off > 0; off >>= 1) { e2 += __shfl_xor(e2, off, 64); h2 += __shfl_xor(h2, off, 64); }
    if (lane == 0) {
        if (live && zero_flag) zero_flag[q] = zf;
        qn[q] = s; sq[q] = sc;
        const float nq = sqrtf(s) * 1.0001f, nx = sqrtf(xmax_norm2) * 1.0001f, dd = (float)d;
        // measured parts: residual norms are float sums of squares of float differences (relative error ~ d 2^-24 + 2^-24 ||x|| / ||delta||
        // per component, carried by the 1.002 and the 1e-7 terms); model parts: the exact side's float32 accumulation (d 2^-23), the
        // scan's int -> float conversion and two multiplications, the key packing (6.2e-5, see prep_queries_fast_kernel)
        const float nqh = sqrtf(h2) * 1.002f, neps = sqrtf(e2) * 1.002f + 1.0e-7f * nq, ndx = dx_max * 1.002f + 1.0e-7f * nx;
        float edot = ndx * nqh + nx * neps + 1.25f * (2.0f * dd + 8.0f) * 1.2e-7f * nq * nx;
        float e = mode == 0 ? edot : 2.0f * edot + 1.25f * (dd + 8.0f) * 1.2e-7f * (nq + nx) * (nq + nx);
        e += 1.25f * 6.2e-5f * (mode == 0 ? nq * nx : nx * nx + 2.0f * nq * nx);
        err_abs[q] = e;
    }
}
void launch_prep_queries_i8(Ctx* c, int metric, const float* src, int B, int dim, float* Qp, int ld, int32_t* zero_flag, void* Q8F, void* Q8R, int ld8, float* sq, float* qn,
                            float* err_abs, int mode, float xmax_norm2, float dx_max, int32_t* stats4) {
    ProfScope ps(c, "prep_queries");
    const size_t lds = (size_t)4 * ld8 * sizeof(float);
    if (lds > 65536) COMET_FAIL(COMET_ERR_INVALID_ARG, "int8 query preparation holds a query row in LDS (dimension <= 4096)");
    prep_queries_i8_kernel<<<dim3(FB_N / 4), dim3(256), lds, c->stream>>>(metric, src, B, dim, Qp, ld, zero_flag, (signed char*)Q8F, (signed char*)Q8R, ld8, sq, qn, err_abs, mode, xmax_norm2, dx_max, stats4);
    LAUNCH_CHECK();
}
bool prep_queries_i8_ok(int dim) { return round_up(dim, 256) <= 4096; }
void launch_prep_queries_fused(Ctx* c, int metric, const float* src, int B, int dim, float* Qp, int ld, int32_t* zero_flag, void* Qh, int ldh, float* qn,
                               float* err_abs, int mode, float xmax_norm2, int32_t* stats4) {
    ProfScope ps(c, "prep_queries");
    const size_t lds = (size_t)4 * ((dim + 3) & ~3) * sizeof(float);
    prep_queries_fused_kernel<<<dim3(FB_N / 4), dim3(256), lds, c->stream>>>(metric, src, B, dim, Qp, ld, zero_flag, (_Float16*)Qh, ldh, qn, err_abs, mode, xmax_norm2, stats4);
    LAUNCH_CHECK();
}


// ------------------------------------------------------------------------------------------------
// fused post-scan stage: ONE workgroup per query does what used to be five launches (K-th tile key, candidate
// collection, exact rescoring, final selection, row gather). Each of those is a short, latency-bound, one-workgroup-
// per-query kernel; chained through HBM they cost ~0.2 ms per 256-query batch, a quarter of the whole search step.
//   1. kappa = exact K-th smallest unit key of the query (keys streamed from L2; linear binning over [min, max], then the
//      few keys of the K-th's bin are ranked directly)
//   2. tau = kappa + 2E (+ key-packing slack); candidates = emitted rows with key <= tau, plus every row of a tile whose
//      third-smallest key (bound) is <= tau; more than POST_CAP -> overflow flag (the host re-runs that query strictly)
//   3. candidates sorted by row (canonical tie order), exact distances in the reference's float32 order: a wave takes
//      32 candidates, stages 32-float slices of their products in LDS, lanes 0..31 run the serial sums
//   4. (score, row) sort, threshold + sanitizeK, ids / scores / count written (count = -ErrZeroVector for a zero cosine query)
// ------------------------------------------------------------------------------------------------
constexpr int POST_THREADS = 1024, POST_WAVES = 16, POST_MAXKEYS = 16384, POST_CAP = 4096, POST_CPW = 32, POST_CHUNK = 32;
constexpr int POST_LPC = POST_CHUNK / 4, POST_CPI = 64 / POST_LPC;   // lanes per candidate slice, candidates per load instruction
constexpr size_t POST_LDS = (size_t)POST_MAXKEYS * 4 + 4096 * 4 + (size_t)POST_CAP * 4 + (size_t)POST_CAP * 4;   // keys|hist , list , scores

__device__ __forceinline__ unsigned pf2key(unsigned u) { return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ unsigned pkey2f(unsigned k) { return (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; }

// block-wide (1024 threads): bin of a 4096-bin LDS histogram holding the rank-th element (1-based) and the count below it
__device__ __forceinline__ void post_find_bin(const unsigned* hist, int rank, unsigned* wsum, int* bin_out, int* before_out) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const unsigned h0 = hist[4 * t], h1 = hist[4 * t + 1], h2 = hist[4 * t + 2], h3 = hist[4 * t + 3];
    const unsigned mine = h0 + h1 + h2 + h3;
    unsigned incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { unsigned y = __shfl_up(incl, off, 64); if (lane >= off) incl += y; }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    unsigned before = incl - mine;
    for (int i = 0; i < w; i++) before += wsum[i];
    const unsigned r = (unsigned)rank;
    if (before < r && before + mine >= r) {
        unsigned run = before; int b = 4 * t;
        if (run + h0 >= r) { b = 4 * t; }
        else { run += h0; if (run + h1 >= r) { b = 4 * t + 1; } else { run += h1; if (run + h2 >= r) { b = 4 * t + 2; } else { run += h2; b = 4 * t + 3; } } }
        *bin_out = b; *before_out = (int)run;
    }
    __syncthreads();
}
template <typename T>
__device__ __forceinline__ void post_bitonic(T* sm, int n2) {
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += POST_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) { const T a = sm[i], b = sm[ixj]; if ((a > b) == ((i & k) == 0)) { sm[i] = b; sm[ixj] = a; } }
            }
            __syncthreads();
        }
}

// sort n distinct values (n <= n2, tail padded with all-ones) ascending: counting rank for short lists (one pass of
// broadcast LDS reads, no barrier ladder), bitonic network otherwise. `tmp` must hold n values.
template <typename T>
__device__ __forceinline__ void post_sort(T* sm, int n, int n2, T* tmp) {
    if (n <= 1024) {
        // every value's rank = the number of smaller values (all distinct). The n * n comparisons are spread over ALL 1024 threads: P = 1024 / n2s
        // threads per value (n2s = n rounded up to a power of two, >= 64), each over 1 / P of the list (broadcast LDS reads, 16 values requested
        // at a time); the partial ranks meet in an LDS counter per value (tmp[1024 ..]: `tmp` holds >= 2048 values at every call site).
        const int t = threadIdx.x;
        int n2s = 64; while (n2s < n) n2s <<= 1;
        const int P = POST_THREADS / n2s, e = t & (n2s - 1), part = t / n2s;
        int* rk = reinterpret_cast<int*>(tmp + 1024);
        if (t < n) rk[t] = 0;
        __syncthreads();
        T me = 0;
        if (e < n) {
            me = sm[e];
            const int j0 = (int)((long)n * part / P), j1 = (int)((long)n * (part + 1) / P);
            int rank = 0;
#pragma unroll 16
            for (int j = j0; j < j1; j++) rank += (sm[j] < me) ? 1 : 0;
            if (P == 1) rk[e] = rank; else if (rank) atomicAdd(&rk[e], rank);
        }
        __syncthreads();
        if (t < n) tmp[rk[t]] = me;          // (part 0: e == t)
        __syncthreads();
        if (t < n) sm[t] = tmp[t];
        __syncthreads();
    } else {
        post_bitonic(sm, n2);
    }
}
// append `v` to list[] for the lanes with `want`, one LDS atomic per wave
__device__ __forceinline__ void post_append(bool want, unsigned v, unsigned* list, int* counter, int cap) {
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63, leader = __builtin_ctzll(m);
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, (int)__builtin_popcountll(m));
    base = __shfl(base, leader, 64);
    if (want) { const int s = base + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull)); if (s < cap) list[s] = v; }
}

// Geometry of the key rows the post stage reads: which rows a key unit stands for, where a candidate's vector and id live.
// A candidate is named by its POSITION = unit * unit_rows + row-in-unit inside the query's key row; ascending position is the
// canonical tie order of the strict path for both geometries.
//   FlatGeom: unit u = rows u * unit_rows .. of the index, the same for every query; position = row.
//   IvfGeom : the key row of query q holds the 64-row units of its probed lists in probe order (uoff[q][j] = first unit of probe j);
//             position -> probe j (binary search over uoff[q]) -> slot = list_base[list] + offset -> arrival row / id of the slot.
struct FlatGeom {
    long n_units; int unit_rows; long n; const unsigned char* elig; const unsigned* ids_table;
    static constexpr bool kSlots = false, kDense = false, kHier = false;
    struct Unit { long row0; long nvalid; };
    __device__ __forceinline__ void bind(int) {}
    __device__ __forceinline__ long units() const { return n_units; }
    __device__ __forceinline__ int urows() const { return unit_rows; }
    __device__ __forceinline__ Unit unit(long u) const { return Unit{u * unit_rows, n - u * unit_rows}; }
    __device__ __forceinline__ bool ok(const Unit& U, int r) const { return r < U.nvalid && (!elig || elig[U.row0 + r]); }
    __device__ __forceinline__ long slot_of(unsigned pos) const { return (long)pos; }
    __device__ __forceinline__ long xrow(long slot) const { return slot; }
    __device__ __forceinline__ unsigned id_of(long slot) const { return ids_table[slot]; }
};
struct IvfGeom {
    const int* uoff; int np; const unsigned* probe_list; int ldp; const long* list_base; const int* list_len;
    const unsigned* row_of_slot; const unsigned* ids_slot; const unsigned char* elig;      // elig per slot
    const float* umin; long ldu;                                                            // per (query, unit): the smallest approximate distance of the unit (written by the scan)
    const int* uo = nullptr; const unsigned* pl = nullptr;                                  // this query's rows (bind)
    static constexpr bool kSlots = true, kDense = true, kHier = true;                                      // dense: the key row holds one approximate distance per position (+inf: no candidate)
    struct Unit { long row0; long nvalid; };                                                // row0 = first slot of the unit
    __device__ __forceinline__ void bind(int q) { uo = uoff + (long)q * (np + 1); pl = probe_list + (long)q * ldp; }
    __device__ __forceinline__ long units() const { return uo[np]; }
    __device__ __forceinline__ int urows() const { return 64; }
    __device__ __forceinline__ int probe_of(long u) const {      // largest j with uo[j] <= u (uo non-decreasing: skips lists without units)
        int lo = 0, hi = np;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (uo[mid] <= u) lo = mid; else hi = mid; }
        return lo;
    }
    __device__ __forceinline__ Unit unit(long u) const {
        const int j = probe_of(u); const unsigned l = pl[j]; const long off = (u - uo[j]) * 64;
        return Unit{list_base[l] + off, (long)list_len[l] - off};
    }
    __device__ __forceinline__ bool ok(const Unit& U, int r) const { return r < U.nvalid && (!elig || elig[U.row0 + r]); }
    __device__ __forceinline__ long slot_of(unsigned pos) const { const Unit U = unit((long)(pos >> 6)); return U.row0 + (pos & 63u); }
    __device__ __forceinline__ long xrow(long slot) const { return (long)row_of_slot[slot]; }
    __device__ __forceinline__ unsigned id_of(long slot) const { return ids_slot[slot]; }
};

template <int METRIC, class GEOM>
__global__ __launch_bounds__(POST_THREADS) void fast_post_kernel(GEOM geom, const float* __restrict__ S0, long ldS, const float* __restrict__ bound, long ldB,
                                                                 const float* __restrict__ err_abs, int K /*requested, sanitised against n*/, int kappa_rank /*0: tau = inf*/,
                                                                 float thr, const float* __restrict__ X, int ld, const float* __restrict__ Qp,
                                                                 const int* __restrict__ zflag,
                                                                 unsigned* __restrict__ out_ids, float* __restrict__ out_scores, int* __restrict__ out_counts,
                                                                 int k_cap, int* __restrict__ overflow, int* __restrict__ stats, unsigned long long* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    geom.bind(blockIdx.x);
    const long n_tiles = geom.units();
    const int unit_rows = geom.urows();
    unsigned long long tr_prev = trace ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long tf_prev = tr_prev;
    auto TF = [&](int ph) { if (trace && threadIdx.x == 0) { const unsigned long long now = __builtin_amdgcn_s_memtime(); atomicAdd(&trace[8 + ph], now - tf_prev); tf_prev = now; } };   // fine stamps (COMET_POST_TRACE)
    auto TR = [&](int ph) { if (trace && threadIdx.x == 0) { const unsigned long long now = __builtin_amdgcn_s_memtime(); atomicAdd(&trace[ph], now - tr_prev); tr_prev = now; } };
    unsigned* hist = reinterpret_cast<unsigned*>(psm) + POST_MAXKEYS;       // [4096]; the 64 KiB in front: rescoring slices, then the sort buffer
    unsigned* lst = hist + 4096;                                             // [POST_CAP] candidate rows
    float* sc = reinterpret_cast<float*>(lst + POST_CAP);                    // [POST_CAP] exact scores
    unsigned* slotv = reinterpret_cast<unsigned*>(sc + POST_CAP);            // [POST_CAP] slot of every candidate (GEOM::kSlots only: the launcher adds the 16 KiB)
    __shared__ unsigned wsum[16];
    __shared__ int s_bin, s_before, s_cnt, s_exp;
    const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, wid = t >> 6;
    constexpr bool DENSE = GEOM::kDense;
    const int nkeys = DENSE ? (int)(n_tiles * unit_rows) : (int)(2 * n_tiles);
    const float* s0 = S0 + (long)q * ldS;
    const float* bd = DENSE ? nullptr : bound + (long)q * ldB;
    const float INF = __builtin_inff();
    if (t == 0) { s_cnt = 0; s_exp = 0; }
    float tau = INF;
    float kappa_v = INF; bool kappa_ok = false;          // the K-th smallest emitted key (general path), for the anchored threshold below
    bool hier_done = false;
    if constexpr (GEOM::kHier) {
        // Two levels for dense rows (IVF): the scan also wrote every unit's MINIMUM. The K-th smallest unit minimum bounds the K-th smallest
        // value of the row from above (K distinct positions), and every value at or below a bound lives in a unit whose minimum is: kappa and
        // then the candidates are found from the few units that can hold them (~K .. 2K of several hundred) instead of three passes over all
        // of the row's values. Anything unusual (more than 8192 units, more than 4096 units or 1024 values under the bound: mass ties, K beyond
        // the finite values) leaves through the general dense path below.
        __shared__ float h_lo[16], h_hi[16]; __shared__ int h_ct[16]; __shared__ int s_nu; __shared__ unsigned s_kap;
        const float* __restrict__ um = geom.umin + (long)q * geom.ldu;
        const int nu = (int)n_tiles;
        constexpr int HU = 8;
        if (kappa_rank > 0 && nu > 0 && nu <= HU * POST_THREADS) {
            float ur[HU];
#pragma unroll
            for (int j = 0; j < HU; j++) { const int i = j * POST_THREADS + t; ur[j] = i < nu ? um[i] : INF; }
            int cnt = 0; float lo = INF, hi = 0.0f;
#pragma unroll
            for (int j = 0; j < HU; j++) if (ur[j] != INF) { cnt++; lo = fminf(lo, ur[j]); hi = fmaxf(hi, ur[j]); }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off, 64)); hi = fmaxf(hi, __shfl_xor(hi, off, 64)); cnt += __shfl_xor(cnt, off, 64); }
            if (lane == 0) { h_lo[wid] = lo; h_hi[wid] = hi; h_ct[wid] = cnt; }
            __syncthreads();
            int cnt_all = 0;
            for (int w = 0; w < POST_WAVES; w++) { lo = fminf(lo, h_lo[w]); hi = fmaxf(hi, h_hi[w]); cnt_all += h_ct[w]; }
            __syncthreads();
            float B0 = 3.0e38f;                                          // fewer than K finite unit minima: every unit is looked at
            if (cnt_all >= kappa_rank) {
                const float scale = hi > lo ? 4095.0f / (hi - lo) : 0.0f;
                auto ubin = [&](float v) { const int bq = (int)((v - lo) * scale); return bq < 0 ? 0 : (bq > 4095 ? 4095 : bq); };
                for (int i = t; i < 4096; i += POST_THREADS) hist[i] = 0;
                __syncthreads();
#pragma unroll
                for (int j = 0; j < HU; j++) if (ur[j] != INF) atomicAdd(&hist[ubin(ur[j])], 1u);
                __syncthreads();
                post_find_bin(hist, kappa_rank, wsum, &s_bin, &s_before);
                const int bbin = s_bin;
                float bmax = 0.0f;
#pragma unroll
                for (int j = 0; j < HU; j++) if (ur[j] != INF && ubin(ur[j]) <= bbin) bmax = fmaxf(bmax, ur[j]);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, off, 64));
                if (lane == 0) h_hi[wid] = bmax;
                __syncthreads();
                for (int w = 0; w < POST_WAVES; w++) bmax = fmaxf(bmax, h_hi[w]);
                B0 = bmax;
            }
            unsigned* ulist = hist;                                      // units that can hold a value at or under the bound
            if (t == 0) { s_nu = 0; s_cnt = 0; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < HU; j++) post_append(ur[j] <= B0, (unsigned)(j * POST_THREADS + t), ulist, &s_nu, 4096);
            __syncthreads();
            const int nu1 = s_nu;
            if (nu1 <= 4096) {
                for (int e = wid; e < nu1; e += POST_WAVES) {
                    const float v = s0[(long)ulist[e] * 64 + lane];
                    post_append(v <= B0, __float_as_uint(v), lst, &s_cnt, POST_CAP);
                }
                __syncthreads();
                const int m = s_cnt;
                if (m >= kappa_rank && m <= 1024) {
                    if (t < m) {
                        const unsigned me = lst[t]; int less = 0;
                        for (int j = 0; j < m; j++) { const unsigned o = lst[j]; less += (o < me || (o == me && j < t)) ? 1 : 0; }
                        if (less == kappa_rank - 1) s_kap = me;          // exactly one thread
                    }
                    __syncthreads();
                    const float kappa = __uint_as_float(s_kap);
                    tau = kappa + 2.0f * err_abs[q] + 1.0e-4f * fabsf(kappa) + 1e-30f;
                    if (t == 0) { s_nu = 0; s_cnt = 0; }
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < HU; j++) post_append(ur[j] <= tau, (unsigned)(j * POST_THREADS + t), ulist, &s_nu, 4096);
                    __syncthreads();
                    const int nu2 = s_nu;
                    if (nu2 <= 4096) {
                        for (int e = wid; e < nu2; e += POST_WAVES) {
                            const unsigned u = ulist[e];
                            const float v = s0[(long)u * 64 + lane];
                            post_append(v <= tau && v != INF, u * 64u + (unsigned)lane, lst, &s_cnt, POST_CAP);
                        }
                        hier_done = true;
                    }
                }
            }
            __syncthreads();
            if (!hier_done) { tau = INF; if (t == 0) s_cnt = 0; }
            __syncthreads();
        }
    }
    // The query's unit keys (two per unit) and unit bounds are read ONCE, into registers, when they fit (<= POST_RU units per
    // thread: 1M rows at 128-row units); every pass below then runs out of registers. Longer rows are streamed from L2 per pass.
    // Dense rows (one approximate distance per position): POST_RD values per thread, value j of thread t = position j * 1024 + t.
    // (16-byte loads: value e of vector j of thread t = position (j * 1024 + t) * 4 + e; a dense row is a multiple of 64 values long)
    constexpr int POST_RU = DENSE ? 1 : 8, POST_RD = DENSE ? 10 : 1;
    const bool reg = DENSE ? nkeys <= POST_RD * POST_THREADS * 4 : n_tiles <= (long)POST_RU * POST_THREADS;
    f32x2v kreg[POST_RU]; float breg[POST_RU]; f32x4v dreg[POST_RD];
    const f32x4v INF4 = {INF, INF, INF, INF};
    auto load_keys = [&]() {
        if constexpr (DENSE) {
#pragma unroll
            for (int j = 0; j < POST_RD; j++) { const int i = (j * POST_THREADS + t) * 4; dreg[j] = i < nkeys ? *reinterpret_cast<const f32x4v*>(s0 + i) : INF4; }
        } else {
#pragma unroll
            for (int j = 0; j < POST_RU; j++) {
                const long tl = (long)j * POST_THREADS + t;
                const bool lv = tl < n_tiles;
                kreg[j] = lv ? *reinterpret_cast<const f32x2v*>(s0 + 2 * tl) : f32x2v{INF, INF};
            }
        }
    };
    // the unit bounds are wanted by the LAST collection only (expansions): loaded right before it — held from the start they cost eight registers across
    // the kappa stage, which the allocator paid for with spills AND a full wait after every bound load (eight serial memory round trips at entry)
    auto load_bounds = [&]() {
        if constexpr (!DENSE) {
#pragma unroll
            for (int j = 0; j < POST_RU; j++) { const long tl = (long)j * POST_THREADS + t; breg[j] = tl < n_tiles ? bd[tl] : INF; }
        }
    };
    if (reg && !hier_done) load_keys();
    TF(0);
    // dense rows beyond the registers: streamed per pass, 4 x 16 bytes per thread in flight (a one-load-per-iteration loop pays the
    // L2 / HBM latency once per value: ~300 iterations for a query that probes a 40 k-row list); every thread makes the same
    // number of calls (absent values arrive as +inf), so f may use wave-wide ballots
    auto for_keys_idx = [&](auto&& f) {             // dense rows: f(value, position)
        if (reg) {
#pragma unroll
            for (int j = 0; j < POST_RD; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) f(dreg[j][e], (j * POST_THREADS + t) * 4 + e);
        } else {
            for (int i0 = 0; i0 < nkeys; i0 += 16 * POST_THREADS) {
                f32x4v b[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { const int i = i0 + (k * POST_THREADS + t) * 4; b[k] = i < nkeys ? *reinterpret_cast<const f32x4v*>(s0 + i) : INF4; }
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int e = 0; e < 4; e++) f(b[k][e], i0 + (k * POST_THREADS + t) * 4 + e);
            }
        }
    };
    auto for_keys = [&](auto&& f) {                 // f(key) over every key of the query, in no particular order
        if constexpr (DENSE) {
            for_keys_idx([&](float v, int) { f(v); });
        } else if (reg) {
#pragma unroll
            for (int j = 0; j < POST_RU; j++) { f(kreg[j][0]); f(kreg[j][1]); }
        } else {
            for (int i = t; i < nkeys; i += POST_THREADS) f(s0[i]);
        }
    };
    // ---- 1. kappa ----
    if (kappa_rank > 0 && !hier_done) {
        // keys >= 0 so bit order = value order. The keys of a query crowd into a few exponent bins, so radix histograms
        // serialise on LDS atomics: bin them LINEARLY over [min, max] instead (monotone: float subtract, multiply and floor
        // are), locate the bin of the K-th smallest, and rank the handful of keys inside it directly.
        int mine = 0; float lo = INF, hi = 0.0f;
        for_keys([&](float v) { if (v != INF) { mine++; lo = fminf(lo, v); hi = fmaxf(hi, v); } });
        const float mn_thread = lo;                  // this thread's smallest value (dense rows: the sample of the bound below)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off, 64)); hi = fmaxf(hi, __shfl_xor(hi, off, 64)); mine += __shfl_xor(mine, off, 64); }
        float* wlo = reinterpret_cast<float*>(hist); float* whi = wlo + 16; int* wct = reinterpret_cast<int*>(whi + 16);
        if (lane == 0) { wlo[wid] = lo; whi[wid] = hi; wct[wid] = mine; }
        __syncthreads();
        int valid = 0;
        for (int w = 0; w < POST_WAVES; w++) { lo = fminf(lo, wlo[w]); hi = fmaxf(hi, whi[w]); valid += wct[w]; }
        __syncthreads();
        TF(1);
        unsigned kap = 0; bool have_kap = false;
        if constexpr (DENSE) {
            // Dense rows hold tens of thousands of values for the sake of the K smallest, and most of them share a few histogram bins
            // (the far lists' distances): binning all of them costs ~30 k contended LDS atomics per query. Instead: every thread's
            // MINIMUM over its (strided, i.e. spread over the whole row) values — 1024 distinct positions, K <= 1024 — is binned; the
            // bin of the K-th smallest minimum bounds the K-th smallest of the row from above; the few values at or below that bin
            // are collected and ranked exactly.
            if (valid >= kappa_rank) {
                const float mn = mn_thread;
                float mlo = mn, mhi = mn == INF ? 0.0f : mn;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { mlo = fminf(mlo, __shfl_xor(mlo, off, 64)); mhi = fmaxf(mhi, __shfl_xor(mhi, off, 64)); }
                if (lane == 0) { wlo[wid] = mlo; whi[wid] = mhi; }
                __syncthreads();
                for (int w = 0; w < POST_WAVES; w++) { mlo = fminf(mlo, wlo[w]); mhi = fmaxf(mhi, whi[w]); }
                __syncthreads();
                const float mscale = mhi > mlo ? 4095.0f / (mhi - mlo) : 0.0f;
                auto mbin = [&](float v) { const int bq = (int)((v - mlo) * mscale); return bq < 0 ? 0 : (bq > 4095 ? 4095 : bq); };   // monotone; values beyond the largest minimum land in the last bin
                for (int i = t; i < 4096; i += POST_THREADS) hist[i] = 0;
                if (t == 0) { s_bin = 4095; s_cnt = 0; }
                __syncthreads();
                if (mn != INF) atomicAdd(&hist[mbin(mn)], 1u);
                const int nfin = __syncthreads_count(mn != INF);           // finite minima
                if (nfin >= kappa_rank) post_find_bin(hist, kappa_rank, wsum, &s_bin, &s_before);
                const int bbin = s_bin;
                // the largest minimum at or below that bin: at least K positions hold a value <= it
                float bmax = (mn != INF && mbin(mn) <= bbin) ? mn : 0.0f;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, off, 64));
                if (lane == 0) whi[wid] = bmax;
                __syncthreads();
                for (int w = 0; w < POST_WAVES; w++) bmax = fmaxf(bmax, whi[w]);
                if (nfin < kappa_rank) bmax = 3.0e38f;                     // fewer than K finite minima: every finite value survives
                for_keys([&](float v) { post_append(v <= bmax, __float_as_uint(v), lst, &s_cnt, POST_CAP); });
                __syncthreads();
                const int m = s_cnt;
                if (m <= 1024) {                                          // (m >= K: the K smallest minima are among them)
                    if (t < m) {
                        const unsigned me = lst[t]; int less = 0;
                        for (int j = 0; j < m; j++) { const unsigned o = lst[j]; less += (o < me || (o == me && j < t)) ? 1 : 0; }
                        if (less == kappa_rank - 1) hist[0] = me;         // exactly one thread
                    }
                    __syncthreads();
                    kap = hist[0]; have_kap = true;
                }
                __syncthreads();
                if (t == 0) s_cnt = 0;
                __syncthreads();
            }
        }
        if (valid >= kappa_rank && !have_kap) {
            const float scale = hi > lo ? 4095.0f / (hi - lo) : 0.0f;
            auto bin_of = [&](unsigned k) { const int bq = (int)((__uint_as_float(k) - lo) * scale); return bq < 0 ? 0 : (bq > 4095 ? 4095 : bq); };
            for (int i = t; i < 4096; i += POST_THREADS) hist[i] = 0;
            __syncthreads();
            for_keys([&](float v) { const unsigned k = __float_as_uint(v); if (k != 0x7F800000u) atomicAdd(&hist[bin_of(k)], 1u); });
            __syncthreads();
            post_find_bin(hist, kappa_rank, wsum, &s_bin, &s_before);
            const int kbin = s_bin, rank_in = kappa_rank - s_before, members = (int)hist[kbin];
            __syncthreads();
            TF(2);
            if (members <= 1024) {
                unsigned* mem = lst;                      // the candidate list is not in use yet
                if (t == 0) s_cnt = 0;
                __syncthreads();
                for_keys([&](float v) { const unsigned k = __float_as_uint(v); if (k != 0x7F800000u && bin_of(k) == kbin) mem[atomicAdd(&s_cnt, 1)] = k; });
                __syncthreads();
                if (t < members) {
                    const unsigned me = mem[t]; int less = 0;
                    for (int j = 0; j < members; j++) { const unsigned o = mem[j]; less += (o < me || (o == me && j < t)) ? 1 : 0; }
                    if (less == rank_in - 1) hist[0] = me;    // exactly one thread
                }
                __syncthreads();
                kap = hist[0];
                __syncthreads();
                if (t == 0) s_cnt = 0;
            } else {
                // mass ties inside one bin: exact radix selection restricted to the bin's members
                unsigned prefix = 0, mask = 0; int rank = rank_in;
                const int shifts[3] = {20, 8, 0}, nbits[3] = {12, 12, 8};
                for (int p = 0; p < 3; p++) {
                    for (int i = t; i < 4096; i += POST_THREADS) hist[i] = 0;
                    __syncthreads();
                    const unsigned bm = (1u << nbits[p]) - 1u;
                    for_keys([&](float v) {
                        const unsigned k = __float_as_uint(v);
                        if (k != 0x7F800000u && bin_of(k) == kbin && (k & mask) == prefix) atomicAdd(&hist[(k >> shifts[p]) & bm], 1u);
                    });
                    __syncthreads();
                    post_find_bin(hist, rank, wsum, &s_bin, &s_before);
                    prefix |= ((unsigned)s_bin) << shifts[p]; mask |= bm << shifts[p]; rank -= s_before;
                    __syncthreads();
                }
                kap = prefix;
            }
            have_kap = true;
        }
        if (have_kap) {
            const float kappa = __uint_as_float(kap);
            tau = kappa + 2.0f * err_abs[q] + 1.0e-4f * fabsf(kappa) + 1e-30f;   // 1e-4 ~ 3 * 2^-15: key packing slack, both sides
            kappa_v = kappa; kappa_ok = true;
        }
    }
    __syncthreads();
    // ---- 1b. anchored threshold (unit keys; round 3) ----
    // tau = kappa + 2E pays E twice: once because the exact K-th distance may exceed kappa by E, once because a true neighbour's key may
    // exceed its exact distance by E. The first E can be MEASURED instead: the >= K rows whose keys are <= kappa are rescored exactly
    // (phase 0), U = their largest exact distance bounds the exact K-th distance from above, and every true neighbour has
    // key <= U + E. Worth its extra pass when tau admits many candidates (the int8 shadow's E is ~6x the fp16 one: 707 -> ~300 rows
    // rescored per query at 1M x 768, K = 100); phase 1 is the unchanged collection + rescoring with the lower tau.
    int anchor = 0;
    if constexpr (!DENSE) {
        if (kappa_ok) {                                   // workgroup-uniform
            int c = 0;
            for_keys([&](float v) { c += (v <= tau) ? 1 : 0; });
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
            int* wct = reinterpret_cast<int*>(hist);
            if (lane == 0) wct[wid] = c;
            __syncthreads();
            int tot = 0;
            for (int w = 0; w < POST_WAVES; w++) tot += wct[w];
            __syncthreads();
            anchor = tot > 2 * kappa_rank + 64 ? 1 : 0;
        }
    }
    TF(3); TR(0);
    int cnt = 0, cnt_anchor = 0;
    // ---- 2. candidates ----
    // rows with key <= tau_c, except the emitted rows with key <= key_lo (phase 0 holds those already, -1: none); at most cap of them
    auto collect = [&](const float tau_c, const bool expand_ok, const float key_lo, const int cap) {
    auto candidates_of = [&](long t0, float bnd, float k0, float k1, float x0, float x1) {      // (x0, x1: the unit's emitted keys, for the expansion; k0, k1: the keys offered)
        const long tl = t0 + t;
        const bool live = tl < n_tiles;
        const bool expand = live && expand_ok && bnd <= tau_c;
        // rare: some non-emitted row of a tile may qualify -> take the whole tile; the wave does it together, one tile at a time
        unsigned long long em = __ballot(expand);
        while (em) {
            const int src = __builtin_ctzll(em); em &= em - 1ull;
            const long te = t0 + (t & ~63) + src;      // tile of lane `src`
            if (lane == src) atomicAdd(&s_exp, 1);
            const typename GEOM::Unit U = geom.unit(te);
            const float e0 = __shfl(x0, src, 64), e1 = __shfl(x1, src, 64);     // the unit's emitted keys: phase 0 took those <= key_lo
            const int r0 = e0 <= key_lo ? (int)(__float_as_uint(e0) & (unsigned)(unit_rows - 1)) : -1;
            const int r1 = e1 <= key_lo ? (int)(__float_as_uint(e1) & (unsigned)(unit_rows - 1)) : -1;
            for (int j = lane; j < unit_rows; j += 64)
                post_append(geom.ok(U, j) && j != r0 && j != r1, (unsigned)(te * unit_rows + j), lst, &s_cnt, cap);
        }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float key = (live && !expand) ? (e ? k1 : k0) : INF;
            post_append(key <= tau_c && key > key_lo && key != INF, (unsigned)(tl * unit_rows + (__float_as_uint(key) & (unsigned)(unit_rows - 1))), lst, &s_cnt, cap);
        }
    };
    if (hier_done) {                                     // the two-level path collected them already
    } else if constexpr (DENSE) {                        // every position whose approximate distance is within tau
        for_keys_idx([&](float v, int i) { post_append(v <= tau_c && v != INF, (unsigned)i, lst, &s_cnt, cap); });
    } else if (reg) {
        // keys in registers: the wave counts its takers first (16 ballots, wave-uniform), reserves their slots with ONE LDS atomic and then writes
        // them — an atomic round trip per key register (three of four have a taker at 1M rows) was most of this stage. Expansions (rare) go through
        // candidates_of's path, one unit at a time.
        unsigned long long bm[POST_RU][2];
        int total = 0;
#pragma unroll
        for (int j = 0; j < POST_RU; j++) {
            const long tl = (long)j * POST_THREADS + t;
            const bool live = tl < n_tiles;
            const bool expand = live && expand_ok && breg[j] <= tau_c;
            if (__ballot(expand)) candidates_of((long)j * POST_THREADS, breg[j], INF, INF, kreg[j][0], kreg[j][1]);   // the expansions only (keys +inf: no takers)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float key = (live && !expand) ? kreg[j][e] : INF;
                bm[j][e] = __ballot(key <= tau_c && key > key_lo && key != INF);
                total += (int)__builtin_popcountll(bm[j][e]);
            }
        }
        int base = 0;
        if (total) {                                   // wave-uniform
            if (lane == 0) base = atomicAdd(&s_cnt, total);
            base = __shfl(base, 0, 64);
            const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
            for (int j = 0; j < POST_RU; j++)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const unsigned long long m = bm[j][e];
                    if ((m >> lane) & 1ull) {
                        const int sl = base + (int)__builtin_popcountll(m & below);
                        const long tl = (long)j * POST_THREADS + t;
                        if (sl < cap) lst[sl] = (unsigned)(tl * unit_rows + (__float_as_uint(kreg[j][e]) & (unsigned)(unit_rows - 1)));
                    }
                    base += (int)__builtin_popcountll(m);
                }
        }
    } else {
        for (long t0 = 0; t0 < n_tiles; t0 += POST_THREADS) {
            const long tl = t0 + t;
            const bool lv = tl < n_tiles;
            const float k0 = lv ? s0[2 * tl] : INF, k1 = lv ? s0[2 * tl + 1] : INF;
            candidates_of(t0, lv ? bd[tl] : INF, k0, k1, k0, k1);
        }
    }
    __syncthreads();
    };
    // ---- 3. exact distances (of lst[0 .. cnt) into sc[]) ----
    float* terms = reinterpret_cast<float*>(psm) + wid * (POST_CPW * POST_CHUNK);   // 4 KiB per wave over the key area
    // The query is staged in LDS (the 16 KiB histogram area is idle here): read from global memory inside the slice loop it
    // would be the youngest load of every iteration, and waiting for it (loads return in order) would also wait for the
    // row slices just requested — the whole prefetch pipeline would drain once per slice.
    const float* __restrict__ qv = Qp + (long)q * ld;
    const bool q_lds = ld <= 4096;
    float* qs = reinterpret_cast<float*>(hist);      // stays an LDS pointer for the compiler (a generic one would make the reads
    auto stage_query = [&]() {                       // flat loads, which count in vmcnt AND lgkmcnt and drain both every slice)
        if (q_lds) {
            for (int i = t * 4; i < ld; i += POST_THREADS * 4) *reinterpret_cast<f32x4v*>(qs + i) = *reinterpret_cast<const f32x4v*>(qv + i);
            __syncthreads();
        }
    };
    const int sub = lane / POST_LPC, lp = lane % POST_LPC;   // a load instruction covers POST_CPI candidates x POST_CHUNK floats
    // Row slices are random 128-byte reads and a candidate's sum is one serial chain, so the stage is latency-bound: spread the
    // candidates over all 16 waves (8, 16 or 32 per wave) and keep more slices in flight the fewer candidates a wave has.
    auto rescore = [&](auto njc, auto depthc, auto qldsc) {
    constexpr int NJ = decltype(njc)::value, POST_DEPTH = decltype(depthc)::value, CPW = NJ * POST_CPI;
    constexpr bool QLDS = decltype(qldsc)::value;
    for (int c0 = wid * CPW; c0 < cnt; c0 += POST_WAVES * CPW) {
        const float* xr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int ci = c0 + j * POST_CPI + sub, cj = ci < cnt ? ci : c0;
            long row;
            if constexpr (GEOM::kSlots) row = geom.xrow((long)slotv[cj]); else row = (long)lst[cj];
            xr[j] = X + row * ld + lp * 4;
        }
        float acc = 0.0f;
        // A block of POST_DEPTH slices per candidate is requested at once and consumed in order (counted waits: only the block's
        // first slice pays the random-read latency). A register ring refilled slot by slot compiles to vmcnt(0) in every
        // iteration — the rotation copies make the compiler wait for everything — and pays that latency per slice.
        const int nsl = ld / POST_CHUNK;                 // ld is a multiple of 32 = POST_CHUNK
        for (int sb = 0; sb < nsl; sb += POST_DEPTH) {
            f32x4v xb[POST_DEPTH][NJ];
#pragma unroll
            for (int d = 0; d < POST_DEPTH; d++)
#pragma unroll
                for (int j = 0; j < NJ; j++) xb[d][j] = *reinterpret_cast<const f32x4v*>(xr[j] + min(sb + d, nsl - 1) * POST_CHUNK);
#pragma unroll
            for (int d = 0; d < POST_DEPTH; d++) {
                const int sl = sb + d;
                if (sl < nsl) {                          // wave-uniform
                    f32x4v qq;
                    if constexpr (QLDS) qq = *reinterpret_cast<const f32x4v*>(qs + sl * POST_CHUNK + lp * 4);
                    else qq = *reinterpret_cast<const f32x4v*>(qv + sl * POST_CHUNK + lp * 4);
#pragma unroll
                    for (int j = 0; j < NJ; j++) {
                        const f32x4v xv = xb[d][j];
                        f32x4v tt;
                        if constexpr (METRIC == COMET_COSINE) { tt[0] = qq[0] * xv[0]; tt[1] = qq[1] * xv[1]; tt[2] = qq[2] * xv[2]; tt[3] = qq[3] * xv[3]; }
                        else {
                            const float d0 = qq[0] - xv[0], d1 = qq[1] - xv[1], d2 = qq[2] - xv[2], d3 = qq[3] - xv[3];
                            tt[0] = d0 * d0; tt[1] = d1 * d1; tt[2] = d2 * d2; tt[3] = d3 * d3;
                        }
                        const int cand = j * POST_CPI + sub;     // 16-byte pieces XOR-swizzled by candidate: the summing lanes (stride 128 B) would
                        *reinterpret_cast<f32x4v*>(&terms[cand * POST_CHUNK + ((lp ^ ((cand >> 1) & 7)) * 4)]) = tt;   // otherwise hit two bank groups, 8 deep
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (lane < CPW) {
                        const float* tp = terms + lane * POST_CHUNK;
                        const int sw = (lane >> 1) & 7;
#pragma unroll
                        for (int i = 0; i < POST_CHUNK / 4; i++) {
                            const f32x4v p = *reinterpret_cast<const f32x4v*>(tp + ((i ^ sw) * 4));
                            acc = acc + p[0]; acc = acc + p[1]; acc = acc + p[2]; acc = acc + p[3];
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (lane < CPW && c0 + lane < cnt) {
            float v;
            if constexpr (METRIC == COMET_COSINE) { float a = acc; if (a > 1.0f) a = 1.0f; else if (a < -1.0f) a = -1.0f; v = 1.0f - a; }   // distance.go:209-213
            else if constexpr (METRIC == COMET_L2) v = (float)__builtin_sqrt((double)acc);
            else v = acc;
            sc[c0 + lane] = v;
        }
    }
    };
    auto rescore_all = [&]() {
        const int per = (cnt + POST_WAVES - 1) / POST_WAVES;
        using T = std::true_type; using F = std::false_type;
        if (q_lds) {
            if (per <= 8) rescore(std::integral_constant<int, 1>{}, std::integral_constant<int, 8>{}, T{});
            else if (per <= 16) rescore(std::integral_constant<int, 2>{}, std::integral_constant<int, 8>{}, T{});
            else rescore(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{}, T{});
        } else rescore(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{}, F{});
        __syncthreads();
    };
    // phase 0's rows and scores wait behind the score array (the 16 KiB the slot geometries use for their slots): they are not collected or
    // rescored a second time, the final order merges them
    constexpr int POST_ACAP = 2048;
    unsigned* lstA = slotv; float* scA = reinterpret_cast<float*>(slotv + POST_ACAP);
    float key_lo = -1.0f;                                // keys are >= 0
    float u_score = INF;                                 // phase 0's largest exact score (the final order's cut)
    if (anchor) {                                        // phase 0: the rows under kappa, exactly (no order needed, no expansions)
        collect(kappa_v, false, -1.0f, POST_CAP);
        cnt = s_cnt;
        TF(4);
        if (cnt <= POST_ACAP && cnt >= kappa_rank) {     // (more: mass ties at kappa — the plain threshold decides)
        if constexpr (GEOM::kSlots) {
            for (int i = t; i < cnt; i += POST_THREADS) slotv[i] = (unsigned)geom.slot_of(lst[i]);
            __syncthreads();
        }
        stage_query();
        TF(5);
        rescore_all();
        TF(6);
        // U = the largest exact distance among the (at least K) rows under kappa, in the keys' space (squared for the L2 family:
        // sqrt is monotone, and U^2 (1 + 2^-21) is no smaller than the sum the largest score was rooted from)
        if (reg) { load_keys(); load_bounds(); }          // requested now: the 48 MB of key rows (all queries) arrive under the reduction and the copies below
        float u = 0.0f;
        for (int i = t; i < cnt; i += POST_THREADS) u = fmaxf(u, sc[i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) u = fmaxf(u, __shfl_xor(u, off, 64));
        float* wmx = reinterpret_cast<float*>(hist);
        if (lane == 0) wmx[wid] = u;
        __syncthreads();
        for (int w = 0; w < POST_WAVES; w++) u = fmaxf(u, wmx[w]);
        u_score = u;
        if constexpr (METRIC == COMET_L2) u = u * u * 1.0000005f;
        const float tau_a = u + err_abs[q] + 1.0e-4f * fabsf(u) + 1e-30f;
        if (tau_a < tau) tau = tau_a;
        cnt_anchor = cnt; key_lo = kappa_v;
        for (int i = t; i < cnt; i += POST_THREADS) { lstA[i] = lst[i]; scA[i] = sc[i]; }
        }
        __syncthreads();
        if (t == 0) s_cnt = 0;
        // the key registers are loaded again rather than kept alive across the rescoring above (24 registers the 128-register budget of a
        // 1024-thread workgroup does not have): inside the branch above when phase 0 ran, here when it was abandoned (mass ties at kappa)
        if (reg && cnt_anchor == 0) load_keys();
        __syncthreads();
    }
    if (reg && cnt_anchor == 0) load_bounds();
    TF(7);
    collect(tau, true, key_lo, POST_CAP - cnt_anchor);
    TF(8); TR(1);
    cnt = s_cnt;
    if (cnt > POST_CAP - cnt_anchor) {                 // empty row; the host re-runs this query on the strict path
        for (int i = t; i < k_cap; i += POST_THREADS) { out_ids[(long)q * k_cap + i] = 0u; out_scores[(long)q * k_cap + i] = 0.0f; }
        if (t == 0) { out_counts[q] = 0; overflow[q] = 1; if (stats) atomicAdd(&stats[1], 1); }
        return;
    }
    int n2 = 64; while (n2 < cnt) n2 <<= 1;
    for (int i = cnt + t; i < n2; i += POST_THREADS) lst[i] = 0xFFFFFFFFu;
    __syncthreads();
    // slot geometries: ascending position = canonical tie order of the strict path, carried by the INDEX into the sorted list (hist: 16 KiB scratch).
    // Unit keys: the final composites carry the row itself, so the order of lst[] is immaterial and is left as collected.
    if constexpr (GEOM::kSlots) post_sort(lst, cnt, n2, hist);
    if constexpr (GEOM::kSlots) {                      // position -> slot, once per candidate
        for (int i = t; i < cnt; i += POST_THREADS) slotv[i] = (unsigned)geom.slot_of(lst[i]);
        __syncthreads();
    }
    TF(9); TR(2);
    stage_query();
    TF(10);
    rescore_all();
    TF(11); TR(3);
    // ---- 4. final order: (score, row position) ----
    // unit-key geometries: the composite's low word is the ROW (= position; ascending row is what the index in the row-sorted list stood
    // for), so that phase 0's rows merge in; slot geometries keep the index (their slot sits in slotv[index])
    unsigned long long* comp = reinterpret_cast<unsigned long long*>(psm);   // POST_CAP composites (32 KiB) over the slices
    const int tot = cnt + cnt_anchor;
    // Only the K smallest leave, and phase 0 measured a bound on them: at least cnt_anchor >= K candidates have an exact score <= u_score, so a
    // candidate above it is not among the K best — it does not enter the final order (the order's n^2 comparisons are LDS-return-bound: ~270
    // composites cost ~9 k clocks, the ~120 at or under u_score a fifth). A threshold below u_score leaves only scores the cut keeps anyway, one at
    // or above it leaves phase 0's rows all valid: min(K, valid) is the same number with and without the cut.
    const bool cut = cnt_anchor > 0 && K > 0 && cnt_anchor >= K;
    if (t == 0) s_cnt = 0;
    __syncthreads();
    for (int i0 = 0; i0 < tot; i0 += POST_THREADS) {                          // (workgroup-uniform trip count: the appends ballot)
        const int i = i0 + t;
        bool keep = false; unsigned long long v = ~0ull;
        if (i < tot) {
            const float sv = i < cnt ? sc[i] : scA[i - cnt];
            const unsigned low = GEOM::kSlots ? (unsigned)i : (i < cnt ? lst[i] : lstA[i - cnt]);
            keep = !(thr > 0.0f && sv > thr) && !(cut && sv > u_score);     // `s.threshold > 0 && dist > s.threshold`
            v = ((unsigned long long)pf2key(__float_as_uint(sv)) << 32) | low;
        }
        const unsigned long long m = __ballot(keep);
        if (m) {
            const int leader = __builtin_ctzll(m);
            int base = 0;
            if (lane == leader) base = atomicAdd(&s_cnt, (int)__builtin_popcountll(m));
            base = __shfl(base, leader, 64);
            if (keep) comp[base + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = v;
        }
    }
    __syncthreads();
    const int valid = s_cnt;                                                  // every composite kept is a valid result
    int n3 = 64; while (n3 < valid) n3 <<= 1;
    for (int i = valid + t; i < n3; i += POST_THREADS) comp[i] = ~0ull;
    __syncthreads();
    TF(12);
    post_sort(comp, valid, n3, comp + POST_CAP);         // second half of the 64 KiB key area as scratch
    TF(13);
    const int kq = (K <= 0 || K > valid) ? valid : K;                          // sanitizeK limiter.go:12-17
    const int nw = kq < k_cap ? kq : k_cap;
    for (int i = t; i < k_cap; i += POST_THREADS) {
        if (i < nw) {
            const unsigned long long cc = comp[i];
            const unsigned ci = (unsigned)(cc & 0xFFFFFFFFull);
            long slot;
            if constexpr (GEOM::kSlots) slot = (long)slotv[ci]; else slot = (long)ci;
            out_ids[(long)q * k_cap + i] = geom.id_of(slot);     // VectorResult.Node.ID()
            out_scores[(long)q * k_cap + i] = __uint_as_float(pkey2f((unsigned)(cc >> 32)));
        } else {
            out_ids[(long)q * k_cap + i] = 0u;
            out_scores[(long)q * k_cap + i] = 0.0f;
        }
    }
    if (t == 0) {
        out_counts[q] = (zflag && zflag[q]) ? -(int)COMET_ERR_ZERO_VECTOR : kq;      // a zero cosine query fails as a whole (ErrZeroVector)
        overflow[q] = 0;
        if (stats) { atomicAdd(&stats[0], cnt + cnt_anchor); atomicAdd(&stats[2], s_exp); }      // rows rescored exactly (both phases)
    }
    TF(14); TR(4);
    if (trace && threadIdx.x == 0) { atomicAdd(&trace[5], (unsigned long long)(cnt + cnt_anchor)); atomicAdd(&trace[6], 1ull); }
}
void launch_flat_post(Ctx* c, int metric, const float* S0, int64_t ldS, const float* bound, int64_t ldB, int64_t n_tiles, int unit_rows, int64_t n, const uint8_t* elig,
                      const float* err_abs, int K, int kappa_rank, float thr, const float* X, int ld, const float* Qp, int B,
                      const uint32_t* ids_table, const int32_t* zflag, uint32_t* out_ids, float* out_scores, int32_t* out_counts, int k_cap,
                      int32_t* overflow, int32_t* stats) {
    if (B <= 0) return;
    static unsigned long long* trace = [] { unsigned long long* p = nullptr; if (getenv("COMET_POST_TRACE")) { HIP_CHECK(hipMalloc(&p, 256)); HIP_CHECK(hipMemset(p, 0, 256)); } return p; }();
    ProfScope ps(c, "flat_post");
    const FlatGeom geom{(long)n_tiles, unit_rows, (long)n, (const unsigned char*)elig, ids_table};
    const size_t lds_flat = POST_LDS + (size_t)POST_CAP * 4;     // + phase 0's rows and scores
#define POST(M) do { HIP_CHECK(hipFuncSetAttribute((const void*)fast_post_kernel<M, FlatGeom>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_flat)); \
        fast_post_kernel<M, FlatGeom><<<dim3(B), dim3(POST_THREADS), lds_flat, c->stream>>>(geom, S0, ldS, bound, ldB, err_abs, K, kappa_rank, thr, X, ld, Qp, \
                                                                                 zflag, out_ids, out_scores, out_counts, k_cap, overflow, stats, trace); } while (0)
    switch (metric) { case COMET_L2: POST(COMET_L2); break; case COMET_L2SQ: POST(COMET_L2SQ); break; default: POST(COMET_COSINE); break; }
#undef POST
    LAUNCH_CHECK();
    if (trace) {          // COMET_POST_TRACE: cumulative s_memtime ticks (100 MHz) per phase, printed every 64 launches
        static int calls = 0;
        if ((++calls & 63) == 0) {
            unsigned long long h2[32]; HIP_CHECK(hipStreamSynchronize(c->stream)); HIP_CHECK(hipMemcpy(h2, trace, 256, hipMemcpyDeviceToHost));
            const unsigned long long* h = h2;
            const double wg = (double)h[6];
            fprintf(stderr, "[post trace fine] clocks per query: keys %.0f minmax %.0f hist+bin %.0f kappa-rank+anchor? %.0f | A.collect %.0f A.stage %.0f A.rescore %.0f A.U+reload %.0f | collect %.0f sort %.0f stage %.0f rescore %.0f | comp %.0f sort %.0f out %.0f\n",
                    h2[8] / wg, h2[9] / wg, h2[10] / wg, h2[11] / wg, h2[12] / wg, h2[13] / wg, h2[14] / wg, h2[15] / wg, h2[16] / wg, h2[17] / wg, h2[18] / wg, h2[19] / wg, h2[20] / wg, h2[21] / wg, h2[22] / wg);
            fprintf(stderr, "[post trace] per query, us: kappa %.1f  candidates %.1f  sort %.1f  rescoring %.1f  final %.1f   (%.0f candidates)\n", h[0] / wg / 100.0, h[1] / wg / 100.0,
                    h[2] / wg / 100.0, h[3] / wg / 100.0, h[4] / wg / 100.0, h[5] / wg);
        }
    }
}

// the same post stage over the key rows of the IVF fast path (kernels_ivf.hip): units of the probed lists in probe order
void launch_ivf_post(Ctx* c, int metric, const float* D, int64_t ldD, const float* umin, int64_t ldu, const int32_t* uoff, int np, const uint32_t* probe_list, int ldp,
                     const int64_t* list_base, const int32_t* list_len, const uint32_t* row_of_slot, const uint32_t* ids_slot, const uint8_t* elig,
                     const float* err_abs, int K, float thr, const float* X, int ld, const float* Qp, int B, const int32_t* zflag,
                     uint32_t* out_ids, float* out_scores, int32_t* out_counts, int k_cap, int32_t* overflow, int32_t* stats) {
    if (B <= 0) return;
    static unsigned long long* trace = [] { unsigned long long* p = nullptr; if (getenv("COMET_POST_TRACE")) { HIP_CHECK(hipMalloc(&p, 256)); HIP_CHECK(hipMemset(p, 0, 256)); } return p; }();
    ProfScope ps(c, "ivf_post");
    const IvfGeom geom{uoff, np, probe_list, ldp, (const long*)list_base, list_len, row_of_slot, ids_slot, (const unsigned char*)elig, umin, (long)ldu};
    const size_t lds = POST_LDS + (size_t)POST_CAP * 4;
    const int kr = K >= 1 ? K : 0;       // fewer valid keys than K (or K <= 0 = all): tau = inf inside the kernel, every unit is expanded
#define POST(M) do { HIP_CHECK(hipFuncSetAttribute((const void*)fast_post_kernel<M, IvfGeom>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        fast_post_kernel<M, IvfGeom><<<dim3(B), dim3(POST_THREADS), lds, c->stream>>>(geom, D, ldD, nullptr, 0, err_abs, K, kr, thr, X, ld, Qp, \
                                                                                 zflag, out_ids, out_scores, out_counts, k_cap, overflow, stats, trace); } while (0)
    switch (metric) { case COMET_L2: POST(COMET_L2); break; case COMET_L2SQ: POST(COMET_L2SQ); break; default: POST(COMET_COSINE); break; }
#undef POST
    LAUNCH_CHECK();
    if (trace) {          // COMET_POST_TRACE: cumulative s_memtime ticks (100 MHz) per phase, printed every 16 launches
        static int calls = 0;
        if ((++calls & 15) == 0) {
            unsigned long long h[8]; HIP_CHECK(hipStreamSynchronize(c->stream)); HIP_CHECK(hipMemcpy(h, trace, 64, hipMemcpyDeviceToHost));
            const double wg = (double)h[6];
            fprintf(stderr, "[ivf post trace] per query, us: kappa %.1f  candidates %.1f  sort+slots %.1f  rescoring %.1f  final %.1f   (%.0f candidates)\n", h[0] / wg / 100.0, h[1] / wg / 100.0,
                    h[2] / wg / 100.0, h[3] / wg / 100.0, h[4] / wg / 100.0, h[5] / wg);
            HIP_CHECK(hipMemset(trace, 0, 64));
        }
    }
}

}  // namespace comet
