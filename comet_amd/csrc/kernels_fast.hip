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
#include <cstdlib>

#include "kernels.hpp"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace comet {

// ------------------------------------------------------------------------------------------------
// fp32 padded rows -> fp16 shadow rows (+ squared norms, max |x|, max norm^2)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void to_half_rows_kernel(const float* __restrict__ X, long n, int ld, _Float16* __restrict__ Xh, int ldh,
                                                           float* __restrict__ rn, unsigned* __restrict__ stats /*[0]=max|x| bits, [1]=max norm2 bits*/) {
    // one wave per row
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* x = X + row * (long)ld;
    _Float16* o = Xh + row * (long)ldh;
    float s = 0.0f, mx = 0.0f;
    for (int i = lane; i < ldh; i += 64) {
        float v = i < ld ? x[i] : 0.0f;
        o[i] = (_Float16)v;
        s += v * v;
        mx = fmaxf(mx, fabsf(v));
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); mx = fmaxf(mx, __shfl_xor(mx, off, 64)); }
    if (lane == 0) {
        if (rn) rn[row] = s;
        if (stats) { atomicMax(&stats[0], __float_as_uint(mx)); atomicMax(&stats[1], __float_as_uint(s)); }
    }
}
void launch_to_half_rows(Ctx* c, const float* X, int64_t n, int ld, void* Xh, int ldh, float* rn, uint32_t* stats) {
    if (n <= 0) return;
    ProfScope ps(c, "to_half_rows");
    to_half_rows_kernel<<<dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, c->stream>>>(X, n, ld, (_Float16*)Xh, ldh, rn, stats);
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
        long gr = row0 + r; if (gr > n - 1) gr = n - 1;
        xsrc[i] = reinterpret_cast<const char*>(Xh + gr * (long)ldh) + ks * 16;
        qsrc[i] = reinterpret_cast<const char*>(Qh + (long)r * ldh) + ks * 16;
        ldsoff[i] = (wid * 4 + i) * 8 * 128;                    // wave-uniform LDS base of the piece
    }
    auto stage = [&](int buf, int kt) {
        unsigned char* xb = smem + buf * 65536;
        unsigned char* qb = xb + 32768;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i] + (long)kt * 128),
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
// ------------------------------------------------------------------------------------------------
// v2 of the scan kernel: 256 rows x 128 queries per 256-thread workgroup (4 waves as 2x2, the same
// 128-row x 64-query, 128-accumulator wave tile), K streamed in 32-half steps through a THREE-stage
// LDS ring of 24 KiB stages filled by global_load_lds, raw s_barrier + COUNTED vmcnt so that two stages
// stay in flight across every barrier (the 2-barrier vmcnt(0) loop of v1 drained the queue at each K step),
// 72 KiB of LDS per workgroup -> two workgroups per CU whose epilogue / main-loop phases interleave on
// each SIMD. The two query halves of a row tile are adjacent on the same XCD, so the second read of the
// tile's rows is an L2 hit.
// ------------------------------------------------------------------------------------------------
constexpr int F2_M = 256, F2_N = 128, F2_K = 32, F2_THREADS = 256, F2_STAGES = 3;
constexpr int F2_XBYTES = F2_M * F2_K * 2, F2_QBYTES = F2_N * F2_K * 2, F2_STAGE = F2_XBYTES + F2_QBYTES;   // 16 KiB + 8 KiB

// [rows][32 halves] tile: 4 sixteen-byte slots per 64-byte row; slot ^= (row >> 2) & 3 makes every 16-lane
// ds_read_b128 group (rows covering all residues mod 16) hit 16 distinct bank slots.
__device__ __forceinline__ int swz32_off(int row, int kslot) { return row * 64 + ((kslot ^ ((row >> 2) & 3)) << 4); }

template <int MODE>
__global__ __launch_bounds__(F2_THREADS, 2) void flat_scan_f16_v2_kernel(const _Float16* __restrict__ Xh, long n, int ldh,
                                                                      const _Float16* __restrict__ Qh /*256 x ldh*/,
                                                                      const float* __restrict__ rn, const float* __restrict__ qn,
                                                                      const unsigned char* __restrict__ elig,
                                                                      float* __restrict__ S0, long ldS, float* __restrict__ bound, long ldB,
                                                                      long n_tiles, int n_halves) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    long tile; int half;
    {
        const long L = blockIdx.x, nx = 8;
        const long q = n_tiles / nx, r = n_tiles % nx, xcd = L % nx, idx = L / nx;
        const long t_in = idx / n_halves; half = (int)(idx % n_halves);
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + t_in;
        if (t_in >= (xcd < r ? q + 1 : q)) return;
    }
    const long row0 = tile * F2_M;
    const int q0 = half * F2_N;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    // staging: one wave instruction moves 16 rows x 64 B. X: 16 pieces, Q: 8 pieces per stage -> 4 + 2 per wave.
    const int prow = lane >> 2, pslot = lane & 3;
    const char* xsrc[4]; const char* qsrc[2]; int xoff[4], qoff[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = (wid * 4 + i) * 16 + prow;
        const int ks = pslot ^ ((r >> 2) & 3);
        long gr = row0 + r; if (gr > n - 1) gr = n - 1;
        xsrc[i] = reinterpret_cast<const char*>(Xh + gr * (long)ldh) + ks * 16;
        xoff[i] = (wid * 4 + i) * 16 * 64;
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int r = (wid * 2 + i) * 16 + prow;
        const int ks = pslot ^ ((r >> 2) & 3);
        qsrc[i] = reinterpret_cast<const char*>(Qh + (long)(q0 + r) * ldh) + ks * 16;
        qoff[i] = F2_XBYTES + (wid * 2 + i) * 16 * 64;
    }
    auto stage = [&](int buf, int kt) {
        unsigned char* sb = smem + buf * F2_STAGE;
#pragma unroll
        for (int i = 0; i < 4; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i] + (long)kt * 64),
                                             (__attribute__((address_space(3))) void*)(sb + xoff[i]), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qsrc[i] + (long)kt * 64),
                                             (__attribute__((address_space(3))) void*)(sb + qoff[i]), 16, 0, 0);
    };

    const int nk = ldh / F2_K;
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    const int arow = wm * 128 + (lane & 31), brow = wn * 64 + (lane & 31), khalf = lane >> 5;
    for (int kt = 0; kt < nk; kt++) {
        // stage kt must have landed (this wave's 6 pieces): all but the newest stage's 6 loads may stay in flight
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // every wave's pieces of stage kt landed; stage kt-1's buffer is free
        if (kt + 2 < nk) stage((kt + 2) % F2_STAGES, kt + 2);
        const unsigned char* xb = smem + (kt % F2_STAGES) * F2_STAGE;
        const unsigned char* qb = xb + F2_XBYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            half8 a[4], b[2];
#pragma unroll
            for (int mb = 0; mb < 4; mb++) a[mb] = *reinterpret_cast<const half8*>(xb + swz32_off(arow + mb * 32, ks * 2 + khalf));
#pragma unroll
            for (int nb = 0; nb < 2; nb++) b[nb] = *reinterpret_cast<const half8*>(qb + swz32_off(brow + nb * 32, ks * 2 + khalf));
#pragma unroll
            for (int mb = 0; mb < 4; mb++)
#pragma unroll
                for (int nb = 0; nb < 2; nb++) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb], b[nb], acc[mb][nb], 0, 0, 0);
        }
    }
    __syncthreads();   // all LDS reads done before the epilogue reuses the ring

    // ---- epilogue (same network as v1; 2 M-waves merge through LDS) ----
    const float INF = __builtin_inff();
    float* trip = reinterpret_cast<float*>(smem);     // [wm][query 128][3]
    const int lane_rowbits = 4 * khalf + 128 * wm;
    const long nvalid = n - row0;
    const bool check = (nvalid < F2_M) || (elig != nullptr);
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
        const int ql = wn * 64 + nb * 32 + (lane & 31);        // query within this workgroup's half
        float qnv = 0.0f;
        if constexpr (MODE == 1) qnv = qn[q0 + ql];
        float t0 = INF, t1 = INF, t2 = INF;
#pragma unroll
        for (int mb = 0; mb < 4; mb++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int rconst = mb * 32 + (e & 3) + 8 * (e >> 2);
                float a;
                if constexpr (MODE == 0) a = 1.0f - acc[mb][nb][e];
                else a = (qnv + rnv[mb * 16 + e]) - 2.0f * acc[mb][nb][e];
                a = fmaxf(a, 0.0f);
                float key = __uint_as_float((__float_as_uint(a) & 0xFFFFFF00u) | (unsigned)rconst);
                if (check) key = ((okmask >> (mb * 16 + e)) & 1ull) ? key : INF;
                ins3(t0, t1, t2, key);
            }
        }
        auto addbits = [&](float v) { return v == INF ? v : __uint_as_float(__float_as_uint(v) | (unsigned)lane_rowbits); };
        t0 = addbits(t0); t1 = addbits(t1); t2 = addbits(t2);
        const float o0 = __shfl_xor(t0, 32, 64), o1 = __shfl_xor(t1, 32, 64), o2 = __shfl_xor(t2, 32, 64);
        ins3(t0, t1, t2, o0); ins3(t0, t1, t2, o1); ins3(t0, t1, t2, o2);
        if (lane < 32) { float* p = trip + ((wm * F2_N + ql) * 3); p[0] = t0; p[1] = t1; p[2] = t2; }
    }
    __syncthreads();
    if (tid < F2_N) {
        const float* pa = trip + tid * 3;
        const float* pb = trip + (F2_N + tid) * 3;
        float t0 = pa[0], t1 = pa[1], t2 = pa[2];
        ins3(t0, t1, t2, pb[0]); ins3(t0, t1, t2, pb[1]); ins3(t0, t1, t2, pb[2]);
        const long q = q0 + tid;
        S0[q * ldS + 2 * tile] = t0;
        S0[q * ldS + 2 * tile + 1] = t1;
        bound[q * ldB + tile] = t2;
    }
}

// ------------------------------------------------------------------------------------------------
// v3 of the scan kernel. Measurements of v1/v2 (0.52 / 0.70 ms per launch against a 0.25 ms HBM floor and a
// 0.16 ms MFMA floor) showed the limiter is bytes in flight, not matrix throughput: an LDS ring can keep only
// 2 stages (<= 64 KiB per CU) outstanding. v3 therefore streams the corpus rows STRAIGHT INTO REGISTERS:
//   * 256 rows x 128 queries per 256-thread workgroup, 4 waves along M (64 rows each, all 128 queries,
//     acc[2][4] of 32x32 tiles = 128 accumulators) — a wave's rows are private, so X never needs LDS;
//   * each lane loads its own A fragments (16 B = 8 halves of its row) with plain global_load_dwordx4 into a
//     4-deep register ring: 16 KiB of HBM requests in flight per wave, 128 KiB per CU, no barrier in the way.
//     Two lanes cover 32 contiguous bytes of a row per instruction; the row's 128-byte line is consumed by
//     consecutive K steps (L1/TA rate needed: ~10 B/clk/CU). Since the MFMA sums over k, A and B only have to
//     agree on which k a (lane, element) holds — they do: both use chunk (ks*2 + lane/32) of the 64-byte step;
//   * the 128-query tile of Q (8 KiB per K step, L2-resident) goes global -> registers -> LDS (swizzled,
//     double buffered, one plain barrier per K step) and is read back as B fragments with ds_read_b128.
// All loads are ordinary loads, so hipcc's own counted vmcnt waits keep the ring in flight.
// ------------------------------------------------------------------------------------------------
constexpr int F3_M = 256, F3_N = 128, F3_K = 32, F3_THREADS = 256, F3_D = 4;

template <int MODE>
__global__ __launch_bounds__(F3_THREADS, 2) void flat_scan_f16_v3_kernel(const _Float16* __restrict__ Xh, long n, int ldh,
                                                                      const _Float16* __restrict__ Qh /*256 x ldh*/,
                                                                      const float* __restrict__ rn, const float* __restrict__ qn,
                                                                      const unsigned char* __restrict__ elig,
                                                                      float* __restrict__ S0, long ldS, float* __restrict__ bound, long ldB,
                                                                      long n_tiles, int n_halves) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * F3_N * F3_K * 2];   // Q double buffer, 16 KiB (reused by the epilogue)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    long tile; int half;
    {
        const long L = blockIdx.x, nx = 8;
        const long q = n_tiles / nx, r = n_tiles % nx, xcd = L % nx, idx = L / nx;
        const long t_in = idx / n_halves; half = (int)(idx % n_halves);
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + t_in;
        if (t_in >= (xcd < r ? q + 1 : q)) return;
    }
    const long row0 = tile * F3_M;
    const int q0 = half * F3_N;
    const int khalf = lane >> 5;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    // A-fragment sources: this lane's two rows (mb = 0, 1); chunk ks*2 + khalf of each 64-byte K step
    const char* xrow[2];
#pragma unroll
    for (int mb = 0; mb < 2; mb++) {
        long gr = row0 + wid * 64 + mb * 32 + (lane & 31); if (gr > n - 1) gr = n - 1;
        xrow[mb] = reinterpret_cast<const char*>(Xh + gr * (long)ldh) + khalf * 16;
    }
    // Q staging: 512 sixteen-byte chunks per step, two per thread
    const char* qsrc[2]; int qdst[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int id = tid + 256 * i, r = id >> 2, slot = id & 3;
        qsrc[i] = reinterpret_cast<const char*>(Qh + (long)(q0 + r) * ldh) + slot * 16;
        qdst[i] = swz32_off(r, slot);
    }
    const int nk = ldh / F3_K;
    half8 xa[F3_D][4];     // [ring slot][ks * 2 + mb]
    half8 qreg[2];
    auto load_x = [&](half8 (&dst)[4], int kt) {
        const int k = kt < nk ? kt : nk - 1;       // clamped re-read past the end: harmless, never consumed
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int mb = 0; mb < 2; mb++) dst[ks * 2 + mb] = *reinterpret_cast<const half8*>(xrow[mb] + (long)k * 64 + ks * 32);
    };
    auto load_q = [&](int kt) {
        const int k = kt < nk ? kt : nk - 1;
#pragma unroll
        for (int i = 0; i < 2; i++) qreg[i] = *reinterpret_cast<const half8*>(qsrc[i] + (long)k * 64);
    };
    auto store_q = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; i++) *reinterpret_cast<half8*>(smem + buf * (F3_N * F3_K * 2) + qdst[i]) = qreg[i];
    };
    auto compute = [&](const half8 (&a)[4], int buf) {
        const unsigned char* qb = smem + buf * (F3_N * F3_K * 2);
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            half8 b[4];
#pragma unroll
            for (int nb = 0; nb < 4; nb++) b[nb] = *reinterpret_cast<const half8*>(qb + swz32_off(nb * 32 + (lane & 31), ks * 2 + khalf));
#pragma unroll
            for (int mb = 0; mb < 2; mb++)
#pragma unroll
                for (int nb = 0; nb < 4; nb++) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks * 2 + mb], b[nb], acc[mb][nb], 0, 0, 0);
        }
    };
    // prologue: ring slots 0..D-2 in flight, Q(0) staged
    load_x(xa[0], 0); load_x(xa[1], 1); load_x(xa[2], 2);
    load_q(0); store_q(0);
    __syncthreads();
#define F3_STEP(S)                                                         \
    if (kt0 + (S) < nk) {                                                  \
        const int kt = kt0 + (S);                                          \
        load_x(xa[((S) + F3_D - 1) % F3_D], kt + F3_D - 1);                \
        load_q(kt + 1);                                                    \
        compute(xa[(S)], kt & 1);                                          \
        store_q((kt + 1) & 1);                                             \
        __syncthreads();                                                   \
    }
    for (int kt0 = 0; kt0 < nk; kt0 += F3_D) { F3_STEP(0) F3_STEP(1) F3_STEP(2) F3_STEP(3) }
#undef F3_STEP

    // ---- epilogue: this wave owns rows wid*64 .. +63 of the tile for all 128 queries ----
    // C layout: column = lane & 31 (query), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    const float INF = __builtin_inff();
    float* trip = reinterpret_cast<float*>(smem);     // [wave 4][query 128][3] = 6 KiB
    const int lane_rowbits = 4 * khalf + 64 * wid;    // bits 2, 6, 7 of the row-in-tile
    const long nvalid = n - row0;
    const bool check = (nvalid < F3_M) || (elig != nullptr);
    unsigned okmask = ~0u;                            // bit mb*16+e
    if (check) {
        okmask = 0u;
        for (int mb = 0; mb < 2; mb++)
            for (int e = 0; e < 16; e++) {
                const long r = mb * 32 + (e & 3) + 8 * (e >> 2) + lane_rowbits;
                bool ok = r < nvalid;
                if (ok && elig) ok = elig[row0 + r] != 0;
                if (ok) okmask |= 1u << (mb * 16 + e);
            }
    }
    float rnv[MODE == 1 ? 32 : 1];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int mb = 0; mb < 2; mb++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                long r = row0 + mb * 32 + (e & 3) + 8 * (e >> 2) + lane_rowbits;
                rnv[mb * 16 + e] = rn[r < n ? r : n - 1];
            }
    }
#pragma unroll
    for (int nb = 0; nb < 4; nb++) {
        const int ql = nb * 32 + (lane & 31);
        float qnv = 0.0f;
        if constexpr (MODE == 1) qnv = qn[q0 + ql];
        float t0 = INF, t1 = INF, t2 = INF;
#pragma unroll
        for (int mb = 0; mb < 2; mb++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int rconst = mb * 32 + (e & 3) + 8 * (e >> 2);      // row bits {0,1,3,4,5}
                float a;
                if constexpr (MODE == 0) a = 1.0f - acc[mb][nb][e];
                else a = (qnv + rnv[mb * 16 + e]) - 2.0f * acc[mb][nb][e];
                a = fmaxf(a, 0.0f);
                float key = __uint_as_float((__float_as_uint(a) & 0xFFFFFF00u) | (unsigned)rconst);
                if (check) key = ((okmask >> (mb * 16 + e)) & 1u) ? key : INF;
                ins3(t0, t1, t2, key);
            }
        }
        auto addbits = [&](float v) { return v == INF ? v : __uint_as_float(__float_as_uint(v) | (unsigned)lane_rowbits); };
        t0 = addbits(t0); t1 = addbits(t1); t2 = addbits(t2);
        const float o0 = __shfl_xor(t0, 32, 64), o1 = __shfl_xor(t1, 32, 64), o2 = __shfl_xor(t2, 32, 64);
        ins3(t0, t1, t2, o0); ins3(t0, t1, t2, o1); ins3(t0, t1, t2, o2);
        if (lane < 32) { float* p = trip + ((wid * F3_N + ql) * 3); p[0] = t0; p[1] = t1; p[2] = t2; }
    }
    __syncthreads();
    if (tid < F3_N) {
        float t0 = INF, t1 = INF, t2 = INF;
#pragma unroll
        for (int w = 0; w < 4; w++) { const float* pw = trip + (w * F3_N + tid) * 3; ins3(t0, t1, t2, pw[0]); ins3(t0, t1, t2, pw[1]); ins3(t0, t1, t2, pw[2]); }
        const long q = q0 + tid;
        S0[q * ldS + 2 * tile] = t0;
        S0[q * ldS + 2 * tile + 1] = t1;
        bound[q * ldB + tile] = t2;
    }
}

// ------------------------------------------------------------------------------------------------
// v4: v3 with the k-assignment changed so that a lane's loads are CONTIGUOUS. PMC on v3 showed why it lost:
// with (lane/32) choosing between adjacent 16-byte chunks, one 128-byte line of a row was requested by four
// instructions spread over two K steps — far beyond the reach of the 32 KiB L1 with 128 KiB of loads in flight
// per CU, so lines were re-fetched. Here a K step is 64 halves (one 128-byte line per row) and lane half h owns
// bytes [64h, 64h+64) of the line: its four 16-byte loads (the four k-substeps) are consecutive addresses issued
// back to back, so a line is fetched once. B fragments use the same (lane, element) -> k map.
// ------------------------------------------------------------------------------------------------
constexpr int F4_M = 256, F4_N = 128, F4_K = 64, F4_THREADS = 256, F4_D = 2;

// Q tile [128 queries][64 halves]: 128-byte rows, 8 sixteen-byte slots; slot ^= (row >> 1) & 7 (as v1)
template <int MODE>
__global__ __launch_bounds__(F4_THREADS, 2) void flat_scan_f16_v4_kernel(const _Float16* __restrict__ Xh, long n, int ldh,
                                                                         const _Float16* __restrict__ Qh /*256 x ldh*/,
                                                                         const float* __restrict__ rn, const float* __restrict__ qn,
                                                                         const unsigned char* __restrict__ elig,
                                                                         float* __restrict__ S0, long ldS, float* __restrict__ bound, long ldB,
                                                                         long n_tiles, int n_halves) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * F4_N * F4_K * 2];   // Q double buffer, 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    long tile; int half;
    {
        const long L = blockIdx.x, nx = 8;
        const long q = n_tiles / nx, r = n_tiles % nx, xcd = L % nx, idx = L / nx;
        const long t_in = idx / n_halves; half = (int)(idx % n_halves);
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + t_in;
        if (t_in >= (xcd < r ? q + 1 : q)) return;
    }
    const long row0 = tile * F4_M;
    const int q0 = half * F4_N;
    const int khalf = lane >> 5;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    const char* xrow[2];
#pragma unroll
    for (int mb = 0; mb < 2; mb++) {
        long gr = row0 + wid * 64 + mb * 32 + (lane & 31); if (gr > n - 1) gr = n - 1;
        xrow[mb] = reinterpret_cast<const char*>(Xh + gr * (long)ldh) + khalf * 64;     // this lane's 64-byte half of each line
    }
    // Q staging: 128 rows x 8 slots = 1024 chunks per step, four per thread
    const char* qsrc[4]; int qdst[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int id = tid + 256 * i, r = id >> 3, slot = id & 7;
        qsrc[i] = reinterpret_cast<const char*>(Qh + (long)(q0 + r) * ldh) + slot * 16;
        qdst[i] = swz_off(r, slot);
    }
    const int nk = ldh / F4_K;
    half8 xa[F4_D][8];     // [ring slot][mb * 4 + ks]
    half8 qreg[4];
    auto load_x = [&](half8 (&dst)[8], int kt) {
        const int k = kt < nk ? kt : nk - 1;
#pragma unroll
        for (int mb = 0; mb < 2; mb++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) dst[mb * 4 + ks] = *reinterpret_cast<const half8*>(xrow[mb] + (long)k * 128 + ks * 16);
    };
    auto load_q = [&](int kt) {
        const int k = kt < nk ? kt : nk - 1;
#pragma unroll
        for (int i = 0; i < 4; i++) qreg[i] = *reinterpret_cast<const half8*>(qsrc[i] + (long)k * 128);
    };
    auto store_q = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) *reinterpret_cast<half8*>(smem + buf * (F4_N * F4_K * 2) + qdst[i]) = qreg[i];
    };
    auto compute = [&](const half8 (&a)[8], int buf) {
        const unsigned char* qb = smem + buf * (F4_N * F4_K * 2);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            half8 b[4];
#pragma unroll
            for (int nb = 0; nb < 4; nb++) b[nb] = *reinterpret_cast<const half8*>(qb + swz_off(nb * 32 + (lane & 31), khalf * 4 + ks));
#pragma unroll
            for (int mb = 0; mb < 2; mb++)
#pragma unroll
                for (int nb = 0; nb < 4; nb++) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb * 4 + ks], b[nb], acc[mb][nb], 0, 0, 0);
        }
    };
    load_x(xa[0], 0);
    load_q(0); store_q(0);
    __syncthreads();
#define F4_STEP(S)                                                         \
    if (kt0 + (S) < nk) {                                                  \
        const int kt = kt0 + (S);                                          \
        load_q(kt + 1);                                                    \
        load_x(xa[((S) + 1) % F4_D], kt + 1);                              \
        __builtin_amdgcn_sched_barrier(0);  /* keep the prefetch above the MFMA block */ \
        compute(xa[(S)], kt & 1);                                          \
        __builtin_amdgcn_sched_barrier(0);                                 \
        store_q((kt + 1) & 1);                                             \
        __syncthreads();                                                   \
    }
    for (int kt0 = 0; kt0 < nk; kt0 += F4_D) { F4_STEP(0) F4_STEP(1) }
#undef F4_STEP

    // ---- epilogue (identical to v3) ----
    const float INF = __builtin_inff();
    float* trip = reinterpret_cast<float*>(smem);     // [wave 4][query 128][3] = 6 KiB
    const int lane_rowbits = 4 * khalf + 64 * wid;
    const long nvalid = n - row0;
    const bool check = (nvalid < F4_M) || (elig != nullptr);
    unsigned okmask = ~0u;
    if (check) {
        okmask = 0u;
        for (int mb = 0; mb < 2; mb++)
            for (int e = 0; e < 16; e++) {
                const long r = mb * 32 + (e & 3) + 8 * (e >> 2) + lane_rowbits;
                bool ok = r < nvalid;
                if (ok && elig) ok = elig[row0 + r] != 0;
                if (ok) okmask |= 1u << (mb * 16 + e);
            }
    }
    float rnv[MODE == 1 ? 32 : 1];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int mb = 0; mb < 2; mb++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                long r = row0 + mb * 32 + (e & 3) + 8 * (e >> 2) + lane_rowbits;
                rnv[mb * 16 + e] = rn[r < n ? r : n - 1];
            }
    }
#pragma unroll
    for (int nb = 0; nb < 4; nb++) {
        const int ql = nb * 32 + (lane & 31);
        float qnv = 0.0f;
        if constexpr (MODE == 1) qnv = qn[q0 + ql];
        float t0 = INF, t1 = INF, t2 = INF;
#pragma unroll
        for (int mb = 0; mb < 2; mb++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int rconst = mb * 32 + (e & 3) + 8 * (e >> 2);
                float a;
                if constexpr (MODE == 0) a = 1.0f - acc[mb][nb][e];
                else a = (qnv + rnv[mb * 16 + e]) - 2.0f * acc[mb][nb][e];
                a = fmaxf(a, 0.0f);
                float key = __uint_as_float((__float_as_uint(a) & 0xFFFFFF00u) | (unsigned)rconst);
                if (check) key = ((okmask >> (mb * 16 + e)) & 1u) ? key : INF;
                ins3(t0, t1, t2, key);
            }
        }
        auto addbits = [&](float v) { return v == INF ? v : __uint_as_float(__float_as_uint(v) | (unsigned)lane_rowbits); };
        t0 = addbits(t0); t1 = addbits(t1); t2 = addbits(t2);
        const float o0 = __shfl_xor(t0, 32, 64), o1 = __shfl_xor(t1, 32, 64), o2 = __shfl_xor(t2, 32, 64);
        ins3(t0, t1, t2, o0); ins3(t0, t1, t2, o1); ins3(t0, t1, t2, o2);
        if (lane < 32) { float* p = trip + ((wid * F4_N + ql) * 3); p[0] = t0; p[1] = t1; p[2] = t2; }
    }
    __syncthreads();
    if (tid < F4_N) {
        float t0 = INF, t1 = INF, t2 = INF;
#pragma unroll
        for (int w = 0; w < 4; w++) { const float* pw = trip + (w * F4_N + tid) * 3; ins3(t0, t1, t2, pw[0]); ins3(t0, t1, t2, pw[1]); ins3(t0, t1, t2, pw[2]); }
        const long q = q0 + tid;
        S0[q * ldS + 2 * tile] = t0;
        S0[q * ldS + 2 * tile + 1] = t1;
        bound[q * ldB + tile] = t2;
    }
}

static int flat_scan_version() {
    static int v = [] { const char* e = getenv("COMET_FLAT_SCAN_VERSION"); return e ? atoi(e) : 1; }();
    return v;
}

void launch_flat_scan_f16(Ctx* c, int mode, const void* Xh, int64_t n, int ldh, const void* Qh, int nq_used, const float* rn, const float* qn,
                          const uint8_t* elig, float* S0, int64_t ldS, float* bound, int64_t ldB) {
    const long n_tiles = ceil_div(n, FB_M);
    ProfScope ps(c, "flat_scan_f16");
    if (flat_scan_version() == 4) {
        const int n_halves = nq_used > F4_N ? 2 : 1;
        const long grid4 = round_up(n_tiles, 8) * n_halves;
        if (mode == 0) flat_scan_f16_v4_kernel<0><<<dim3((unsigned)grid4), dim3(F4_THREADS), 0, c->stream>>>((const _Float16*)Xh, n, ldh, (const _Float16*)Qh, rn, qn, elig, S0, ldS, bound, ldB, n_tiles, n_halves);
        else flat_scan_f16_v4_kernel<1><<<dim3((unsigned)grid4), dim3(F4_THREADS), 0, c->stream>>>((const _Float16*)Xh, n, ldh, (const _Float16*)Qh, rn, qn, elig, S0, ldS, bound, ldB, n_tiles, n_halves);
        LAUNCH_CHECK();
        return;
    }
    if (flat_scan_version() == 3) {
        const int n_halves = nq_used > F3_N ? 2 : 1;
        const long grid3 = round_up(n_tiles, 8) * n_halves;
        if (mode == 0) flat_scan_f16_v3_kernel<0><<<dim3((unsigned)grid3), dim3(F3_THREADS), 0, c->stream>>>((const _Float16*)Xh, n, ldh, (const _Float16*)Qh, rn, qn, elig, S0, ldS, bound, ldB, n_tiles, n_halves);
        else flat_scan_f16_v3_kernel<1><<<dim3((unsigned)grid3), dim3(F3_THREADS), 0, c->stream>>>((const _Float16*)Xh, n, ldh, (const _Float16*)Qh, rn, qn, elig, S0, ldS, bound, ldB, n_tiles, n_halves);
        LAUNCH_CHECK();
        return;
    }
    if (flat_scan_version() == 2) {
        const int n_halves = nq_used > F2_N ? 2 : 1;
        const size_t lds2 = (size_t)F2_STAGES * F2_STAGE;
        const long grid2 = round_up(n_tiles, 8) * n_halves;
        if (mode == 0) {
            HIP_CHECK(hipFuncSetAttribute((const void*)flat_scan_f16_v2_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            flat_scan_f16_v2_kernel<0><<<dim3((unsigned)grid2), dim3(F2_THREADS), lds2, c->stream>>>((const _Float16*)Xh, n, ldh, (const _Float16*)Qh, rn, qn, elig, S0, ldS, bound, ldB, n_tiles, n_halves);
        } else {
            HIP_CHECK(hipFuncSetAttribute((const void*)flat_scan_f16_v2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            flat_scan_f16_v2_kernel<1><<<dim3((unsigned)grid2), dim3(F2_THREADS), lds2, c->stream>>>((const _Float16*)Xh, n, ldh, (const _Float16*)Qh, rn, qn, elig, S0, ldS, bound, ldB, n_tiles, n_halves);
        }
        LAUNCH_CHECK();
        return;
    }
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
