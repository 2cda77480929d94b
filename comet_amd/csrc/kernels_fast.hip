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
//               into the low 8 mantissa bits, and keeps per (query, tile) the two smallest keys plus the
//               third smallest (a lower bound for every row of the tile that was NOT emitted) with a
//               branch-free min/med3 network.
//   2. collect: kappa = exact K-th smallest emitted key; tau = kappa + 2E, E a rigorous bound on
//               |approx - exact| (fp16 rounding of both operands, fp32 accumulation, key packing).
//               Candidates = emitted keys <= tau, plus ALL rows of any tile whose bound <= tau.
//               Every row with approx <= tau is in that set, and the exact top-K all have
//               approx <= a_K + 2E <= tau, so the set contains the exact answer.
//   3. rescore: the exact gather kernel (kernels_dist.hip) + the exact selection give ids and scores
//               bit-identical to the strict path; a query whose candidate list overflows falls back to
//               the strict path.
#include "kernels.hpp"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace comet {

// ------------------------------------------------------------------------------------------------
// fp32 padded rows -> fp16 shadow rows (+ squared norms, max |x|, max norm^2)
// ------------------------------------------------------------------------------------------------
// Shadow layout ("tiled"): [tile of 256 rows][K step of 64 halves][row in tile][64 halves] — the 32 KiB a
// workgroup needs for one K step of its tile are CONTIGUOUS in HBM (whole DRAM pages stream in), instead of
// 256 separate 128-byte pieces at a 1536-byte stride as in a row-major shadow.
__device__ __forceinline__ long tiled_off(long row, int k, int ldh) {
    const long tile = row >> 8; const int r = (int)(row & 255);
    return ((tile * (ldh >> 6) + (k >> 6)) * 256 + r) * 64 + (k & 63);
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
// the scan kernel
// ------------------------------------------------------------------------------------------------
constexpr int FB_M = 256;     // corpus rows per workgroup tile
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
        // tiled shadow: (tile, kt) slab = 256 rows x 128 B contiguous; rows past n are zero-filled padding of the last tile
        xsrc[i] = reinterpret_cast<const char*>(Xh) + (tile * (long)(ldh >> 6) * 256 + r) * 128 + ks * 16;
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

    // ---- epilogue: per (query, tile) two smallest packed keys + third smallest (bound) ----
    // C layout of the 32x32 MFMA: column = lane & 31 (query), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    const float INF = __builtin_inff();
    float* trip = reinterpret_cast<float*>(smem);     // [wm][query 256][3]  (6 KiB), LDS is free now
    const int lane_rowbits = 4 * khalf + 128 * wm;    // bits 2 and 7 of the row-in-tile
    const long nvalid = n - row0;
    const bool check = (nvalid < FB_M) || (elig != nullptr);   // workgroup-uniform
    // per-lane 64-bit mask of usable rows (bit mb*16+e), only built on the slow path
    unsigned long long okmask = ~0ull;
    if (check) {
        okmask = 0ull;
        for (int mb = 0; mb < 4; mb++)
            for (int e = 0; e < 16; e++) {
                const long r = mb * 32 + (e & 3) + 8 * (e >> 2) + lane_rowbits;
                bool ok = r < nvalid;
                if (ok && elig) ok = elig[row0 + r] != 0;
                if (ok) okmask |= 1ull << (mb * 16 + e);
            }
    }
    float rnv[MODE == 1 ? 64 : 1];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int mb = 0; mb < 4; mb++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                long r = row0 + mb * 32 + (e & 3) + 8 * (e >> 2) + lane_rowbits;
                rnv[mb * 16 + e] = rn[r < n ? r : n - 1];
            }
    }
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
        const int q = wn * 64 + nb * 32 + (lane & 31);
        float qnv = 0.0f;
        if constexpr (MODE == 1) qnv = qn[q];
        float t0 = INF, t1 = INF, t2 = INF;
#pragma unroll
        for (int mb = 0; mb < 4; mb++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int rconst = mb * 32 + (e & 3) + 8 * (e >> 2);      // compile-time part of the row
                float a;
                if constexpr (MODE == 0) a = 1.0f - acc[mb][nb][e];
                else a = (qnv + rnv[mb * 16 + e]) - 2.0f * acc[mb][nb][e];
                a = fmaxf(a, 0.0f);
                float key = __uint_as_float((__float_as_uint(a) & 0xFFFFFF00u) | (unsigned)rconst);   // row bits {0,1,3,4,5,6}
                if (check) key = ((okmask >> (mb * 16 + e)) & 1ull) ? key : INF;
                ins3(t0, t1, t2, key);
            }
        }
        // add the lane-dependent row bits (2 and 7) to the survivors; inf stays inf
        auto addbits = [&](float v) { return v == INF ? v : __uint_as_float(__float_as_uint(v) | (unsigned)lane_rowbits); };
        t0 = addbits(t0); t1 = addbits(t1); t2 = addbits(t2);
        // merge with the other half-wave (rows +4): exchange triples across lane ^ 32
        const float o0 = __shfl_xor(t0, 32, 64), o1 = __shfl_xor(t1, 32, 64), o2 = __shfl_xor(t2, 32, 64);
        ins3(t0, t1, t2, o0); ins3(t0, t1, t2, o1); ins3(t0, t1, t2, o2);
        if (lane < 32) { float* p = trip + ((wm * 256 + q) * 3); p[0] = t0; p[1] = t1; p[2] = t2; }
    }
    __syncthreads();
    if (tid < 256) {
        const float* pa = trip + tid * 3;
        const float* pb = trip + (256 + tid) * 3;
        float t0 = pa[0], t1 = pa[1], t2 = pa[2];
        ins3(t0, t1, t2, pb[0]); ins3(t0, t1, t2, pb[1]); ins3(t0, t1, t2, pb[2]);
        S0[(long)tid * ldS + 2 * tile] = t0;
        S0[(long)tid * ldS + 2 * tile + 1] = t1;
        bound[(long)tid * ldB + tile] = t2;
    }
}
void launch_flat_scan_f16(Ctx* c, int mode, const void* Xh, int64_t n, int ldh, const void* Qh, int /*nq_used*/, const float* rn, const float* qn,
                          const uint8_t* elig, float* S0, int64_t ldS, float* bound, int64_t ldB) {
    const long n_tiles = ceil_div(n, FB_M);
    ProfScope ps(c, "flat_scan_f16");
    const size_t lds = 2 * 65536;
    const long grid = round_up(n_tiles, 8);
    if (mode == 0) {
        HIP_CHECK(hipFuncSetAttribute((const void*)flat_scan_f16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        flat_scan_f16_kernel<0><<<dim3((unsigned)grid), dim3(FB_THREADS), lds, c->stream>>>((const _Float16*)Xh, n, ldh, (const _Float16*)Qh, rn, qn, elig, S0, ldS, bound, ldB, n_tiles);
    } else {
        HIP_CHECK(hipFuncSetAttribute((const void*)flat_scan_f16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        flat_scan_f16_kernel<1><<<dim3((unsigned)grid), dim3(FB_THREADS), lds, c->stream>>>((const _Float16*)Xh, n, ldh, (const _Float16*)Qh, rn, qn, elig, S0, ldS, bound, ldB, n_tiles);
    }
    LAUNCH_CHECK();
}
int flat_fast_tile_rows() { return FB_M; }
int flat_fast_batch() { return FB_N; }

// ------------------------------------------------------------------------------------------------
// collect: candidates of one query (one workgroup per query)
// ------------------------------------------------------------------------------------------------
constexpr int COLLECT_THREADS = 256;

// kth_keys: [B][kcap] sorted emitted keys (select_topk output), kth_cnt[B]; K = requested k (after sanitising
// against the eligible count is not known here: if fewer than K keys were found tau = +inf).
// err_abs[q]: E (absolute) for this query; candidates: cand[q][cap] ascending row indices, cand_cnt[q];
// overflow[q] = 1 if more than cap candidates (the host re-runs those queries on the strict path).
__global__ __launch_bounds__(COLLECT_THREADS) void flat_collect_kernel(const float* __restrict__ S0, long ldS, const float* __restrict__ bound, long ldB,
                                                                       long n_tiles, long n, const unsigned char* __restrict__ elig,
                                                                       const float* __restrict__ kth_keys, int kcap, const int* __restrict__ kth_cnt, int K,
                                                                       const float* __restrict__ err_abs, unsigned* __restrict__ cand, int cap,
                                                                       int* __restrict__ cand_cnt, int* __restrict__ overflow, int* __restrict__ stats) {
    extern __shared__ unsigned lst[];   // cap2 entries (power of two >= cap)
    __shared__ int s_cnt; __shared__ int s_exp;
    const int q = blockIdx.x;
    int cap2 = 1; while (cap2 < cap) cap2 <<= 1;
    if (threadIdx.x == 0) { s_cnt = 0; s_exp = 0; }
    for (int i = threadIdx.x; i < cap2; i += COLLECT_THREADS) lst[i] = 0xFFFFFFFFu;
    __syncthreads();
    const int found = kth_cnt[q];
    float tau = __builtin_inff();
    if (K > 0 && found >= K) {
        const float kappa = kth_keys[(long)q * kcap + (K - 1)];
        // tau = kappa + 2E, plus the relative slack of the 8 truncated mantissa bits of both sides
        tau = kappa + 2.0f * err_abs[q] + 1.0e-4f * fabsf(kappa) + 1e-30f;   // 1e-4 ~ 3 * 2^-15: key packing slack, both sides
    }
    const float* s0 = S0 + (long)q * ldS;
    const float* bd = bound + (long)q * ldB;
    for (long t = threadIdx.x; t < n_tiles; t += COLLECT_THREADS) {
        if (bd[t] <= tau) {
            // some non-emitted row of this tile may qualify: take the whole tile
            atomicAdd(&s_exp, 1);
            const long r0 = t * FB_M;
            for (int j = 0; j < FB_M; j++) {
                const long r = r0 + j;
                if (r < n && (!elig || elig[r])) { int s = atomicAdd(&s_cnt, 1); if (s < cap) lst[s] = (unsigned)r; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float key = s0[2 * t + e];
                if (key <= tau) {   // inf keys (masked rows) only pass when tau is inf; they carry no row -> skip
                    if (key == __builtin_inff()) continue;
                    const long r = t * FB_M + (__float_as_uint(key) & 0xFFu);
                    int s = atomicAdd(&s_cnt, 1); if (s < cap) lst[s] = (unsigned)r;
                }
            }
        }
    }
    __syncthreads();
    const int cnt = s_cnt;
    if (cnt > cap) {
        if (threadIdx.x == 0) { overflow[q] = 1; cand_cnt[q] = 0; if (stats) { atomicAdd(&stats[1], 1); } }
        return;
    }
    // ascending row order = the canonical tie order of the strict path (sort only the used prefix)
    int n2 = 64; while (n2 < cnt) n2 <<= 1;
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += COLLECT_THREADS) {
                int ixj = i ^ j;
                if (ixj > i) {
                    unsigned a = lst[i], b = lst[ixj];
                    bool up = ((i & k) == 0);
                    if ((a > b) == up) { lst[i] = b; lst[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < cap; i += COLLECT_THREADS) cand[(long)q * cap + i] = lst[i];
    if (threadIdx.x == 0) {
        cand_cnt[q] = cnt; overflow[q] = 0;
        if (stats) { atomicAdd(&stats[0], cnt); atomicAdd(&stats[2], s_exp); }
    }
}
void launch_flat_collect(Ctx* c, const float* S0, int64_t ldS, const float* bound, int64_t ldB, int64_t n_tiles, int64_t n, const uint8_t* elig,
                         const float* kth_keys, int kcap, const int32_t* kth_cnt, int K, const float* err_abs, int B, uint32_t* cand, int cap,
                         int32_t* cand_cnt, int32_t* overflow, int32_t* stats) {
    int cap2 = 1; while (cap2 < cap) cap2 <<= 1;
    ProfScope ps(c, "flat_collect");
    flat_collect_kernel<<<dim3(B), dim3(COLLECT_THREADS), sizeof(unsigned) * cap2, c->stream>>>(S0, ldS, bound, ldB, n_tiles, n, elig, kth_keys, kcap, kth_cnt, K,
                                                                                            err_abs, cand, cap, cand_cnt, overflow, stats);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// query side: fp16 copy (zero-padded to 256 rows), squared norms, rigorous error bound per query
// ------------------------------------------------------------------------------------------------
// mode 0 cosine: |approx - exact| <= E; mode 1 L2 family (squared space).
__global__ __launch_bounds__(256) void prep_queries_fast_kernel(const float* __restrict__ Qp, int B, int ld, int dim, _Float16* __restrict__ Qh, int ldh,
                                                                float* __restrict__ qn, float* __restrict__ err_abs, int mode, float xmax_norm2) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= FB_N) return;
    _Float16* o = Qh + (long)q * ldh;
    float s = 0.0f;
    for (int i = lane; i < ldh; i += 64) {
        float v = (q < B && i < ld) ? Qp[(long)q * ld + i] : 0.0f;
        o[i] = (_Float16)v;
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
        err_abs[q] = 1.25f * e;
    }
}
void launch_prep_queries_fast(Ctx* c, const float* Qp, int B, int ld, int dim, void* Qh, int ldh, float* qn, float* err_abs, int mode, float xmax_norm2) {
    prep_queries_fast_kernel<<<dim3(FB_N / 4), dim3(256), 0, c->stream>>>(Qp, B, ld, dim, (_Float16*)Qh, ldh, qn, err_abs, mode, xmax_norm2);
    LAUNCH_CHECK();
}

}  // namespace comet
