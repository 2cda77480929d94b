#pragma once
// kernels_scanq.inc.hpp — (included by kernels_scanq.hip: cosine, and kernels_scanq_l2.hip: the L2 family; one translation unit per MODE so that the
// instantiations compile in parallel)
// kernels_scanq — register-stationary scan tiles of the Flat fast path's int8 shadow for gfx950 (round 4): the queries of a batch live in
// registers for the whole launch. Wide tile (65..256 queries): flat_scan_qr_kernel; narrow tile (<= 64 queries): flat_scan_qn_kernel. Keys and bounds come
// out exactly as from kernels_fast.hip's flat_scan_q8_kernel / flat_scan_f16_n64_kernel (same unit numbering, same key format); the post stage does not
// know which tile ran. A translation unit of its own: the 24 instantiations compile beside kernels_fast.hip instead of after it.
#include <type_traits>
#include "kernels.hpp"

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
#ifndef FAST_ROW_AUX
#define FAST_ROW_AUX 2
#endif

namespace comet {
extern unsigned long long* g_scan_trace;   // kernels_fast.hip (set by tools/scan_check.hip only)

// ------------------------------------------------------------------------------------------------
// Register-stationary scan tile for the int8 shadow (round 4): the QUERIES live in registers for the whole launch.
//
// What held the query-stationary tile above at a third of either roofline (0.245-0.277 ms for 1M x 768 x 256: HBM floor 0.12 ms,
// int8 MFMA floor 0.09 ms): per K step its eight waves pull the SAME 32 KiB of query fragments from L2 again for every row tile —
// as many bytes through the CU's vector-memory path as the corpus rows themselves — every A fragment read from LDS feeds ONE
// MFMA, eight waves meet at a barrier every 32 MFMAs, and the selection epilogue (a quarter of an int8 tile) runs with the matrix
// pipe idle because both waves of a SIMD reach it together.
// Here: four waves per workgroup, ONE per SIMD, 512 registers each (launch_bounds(256, 1)):
//   * wave w keeps the int8 fragments of queries 64 w .. 64 w + 63 for the whole K range in registers (ld8 / 4 of them: 192 at d = 768),
//     loaded once per launch: no query traffic in the loop at all;
//   * the rows stream HBM -> LDS by LDS-DMA in PASSES of 64 rows x ld8 bytes (48 KiB at 768) through a ring of whole passes; a wave's
//     tile is 64 rows x 64 queries (2 x 2 MFMA blocks), so an A fragment read from LDS feeds two MFMAs (half the LDS traffic per
//     MFMA) and there is ONE barrier per pass (96 MFMAs per wave at 768), placed one fragment-prefetch ahead of the pass boundary;
//   * the selection epilogue of pass p-1 (accumulators double-buffered: 2 x 64 registers) is issued between the MFMAs of pass p,
//     a few VALU instructions per MFMA: the matrix pipe never waits for it;
//   * work is dealt in 128-row halves of the 256-row shadow tiles (XCD-contiguous, round robin inside an XCD).
// Keys and bounds come out exactly as from flat_scan_q8_kernel<MODE, UR, true> (same packed values, same unit numbering).
// ------------------------------------------------------------------------------------------------
constexpr int QR_THREADS = 256, QR_PASS_ROWS = 64;
__device__ __forceinline__ int qr_med3_i32(int a, int b, int c) { int d; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// keep the three LARGEST of (t0 >= t1 >= t2) U {v}. INT: signed integer keys; otherwise float keys carried as bits (all three updates through
// v_med3_f32: fmaxf() would put a canonicalising v_max_f32 v, v, v in front of every comparison of a value that came out of bit operations)
template <bool INT> __device__ __forceinline__ void qr_ins3(int& t0, int& t1, int& t2, int v) {
    if constexpr (INT) {
        const int n2 = qr_med3_i32(t1, t2, v), n1 = qr_med3_i32(t0, t1, v);
        t0 = t0 > v ? t0 : v; t1 = n1; t2 = n2;
    } else {
        const float f0 = __int_as_float(t0), f1 = __int_as_float(t1), f2 = __int_as_float(t2), fv = __int_as_float(v);
        const float n2 = __builtin_amdgcn_fmed3f(f1, f2, fv), n1 = __builtin_amdgcn_fmed3f(f0, f1, fv), n0 = __builtin_amdgcn_fmed3f(f0, fv, __builtin_inff());
        t0 = __float_as_int(n0); t1 = __float_as_int(n1); t2 = __float_as_int(n2);
    }
}
// LDS-DMA pieces issued from inline asm: 16 bytes (resp. 4) per lane from `sbase + voff` to LDS byte address `lds_addr` + lane * 16 (4), non-temporal.
// Not the builtin: hipcc's wait-count pass treats a pending LDS-DMA as a FLAT access that may complete on either counter and then turns
// EVERY lgkmcnt wait into lgkmcnt(0) for as long as it has not seen a vmcnt wait for the DMA (the kernel's counted vmcnt waits are asm and invisible
// to it): the fragment prefetch (reads for step kk + 1 in flight under step kk's MFMAs) would be drained at every second step. M0 is
// written and NOT restored (two scalar instructions per piece less): nothing else in flat_scan_qr_kernel uses it — no LDS-DMA builtin, no
// relative indexing, no message — and the statement that reads M0 is the one that writes it.
// Operands go through readfirstlane (free for values hipcc already keeps in SGPRs; where it has moved a uniform chain to the vector ALU the
// v_readfirstlane result needs 5 wait states before a VMEM instruction may read it as base: s_mov + s_nop 3).
__device__ __forceinline__ void qr_dma16(const char* sbase /*wave-uniform*/, unsigned voff, unsigned lds_addr /*wave-uniform*/) {
    const unsigned long long b = (unsigned long long)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32)), la = __builtin_amdgcn_readfirstlane(lds_addr);
    const unsigned long long bu = ((unsigned long long)hi << 32) | lo;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(voff), "s"(bu), "s"(la) : "memory");
}
__device__ __forceinline__ void qr_dma4(const char* sbase /*wave-uniform*/, unsigned voff, unsigned lds_addr /*wave-uniform*/) {
    const unsigned long long b = (unsigned long long)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32)), la = __builtin_amdgcn_readfirstlane(lds_addr);
    const unsigned long long bu = ((unsigned long long)hi << 32) | lo;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dword %0, %1" :: "v"(voff), "s"(bu), "s"(la) : "memory");
}
template <int I, int N, class F> __device__ __forceinline__ void qr_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); qr_static_for<I + 1, N>(f); }
}
template <int NKS> struct QrGeom {
    static constexpr int NKK = NKS * 4;                        // 32-dimension MFMA steps per row
    static constexpr int STAGE = NKS * 8192;                   // bytes of one pass in LDS: [K step][64 rows][128 B]
    static constexpr int RING = (144 * 1024) / STAGE > 8 ? 8 : (144 * 1024) / STAGE;   // passes in the ring (3 at 768, 4 at 512, 8 at 256)
    static constexpr int PPW = 2 * NKS;                        // DMA pieces (1 KiB) per wave per pass
    static constexpr int LDS = RING * STAGE + (RING + 1) * (256 + 64); // + the row norms (MODE 1) and the eligibility bytes of the ring's passes and of the pass whose selection is still running
};
// s_waitcnt vmcnt(min(k, KMAX) * VMW), k a run-time count (the immediate must be a constant: a compare chain over the few possible values)
template <int VMW, int K> __device__ __forceinline__ void qr_wait_chain(int k) {
    if constexpr (K <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else { if (k >= K) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K * VMW) : "memory"); else qr_wait_chain<VMW, K - 1>(k); }
}
template <int VMW, int KMAX> __device__ __forceinline__ void qr_wait_passes(int k) {
    static_assert(KMAX * VMW <= 63, "vmcnt is a 6-bit counter");
    qr_wait_chain<VMW, KMAX>(k);
}

template <int MODE, int UR, int NKS, bool ELIG>
__global__ __launch_bounds__(QR_THREADS, 1) void flat_scan_qr_kernel(const signed char* __restrict__ X8, long n, const signed char* __restrict__ Q8F,
                                                                     const float* __restrict__ rn, const float* __restrict__ qn,
                                                                     const unsigned char* __restrict__ elig,
                                                                     float* __restrict__ S0, long ldS, float* __restrict__ bound, long ldB, long n_tiles,
                                                                     const float* __restrict__ sx, const float* __restrict__ sq,
                                                                     unsigned long long* __restrict__ trace /*nullable: s_memtime stamps of workgroup 8 (tools/scan_check)*/) {
    using G = QrGeom<NKS>;
    constexpr int NKK = G::NKK, STAGE = G::STAGE, RING = G::RING, PPW = G::PPW;
    constexpr int VMW = PPW + (MODE == 1 ? 1 : 0) + (ELIG ? 1 : 0);   // vector-memory operations a wave issues per pass (pieces + its share of the row norms + of the eligibility bytes)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* rn_ring = reinterpret_cast<float*>(smem + RING * STAGE);
    const unsigned* el_ring = reinterpret_cast<const unsigned*>(smem + RING * STAGE + (RING + 1) * 256);     // 16 dwords (64 eligibility bytes) per pass
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, khalf = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- work split: 128-row halves; every XCD owns a contiguous range, its workgroups take them round robin ----
    const long nx = 8, xcd = blockIdx.x % nx, wgx = blockIdx.x / nx, W = gridDim.x / nx;
    const long H = n_tiles * 2, hq = H / nx, hrem = H % nx;
    const long xbase = xcd < hrem ? xcd * (hq + 1) : hrem * (hq + 1) + (xcd - hrem) * hq, xcount = xcd < hrem ? hq + 1 : hq;
    const long my_halves = wgx < xcount ? (xcount - wgx + W - 1) / W : 0;
    if (my_halves == 0) return;
    const int P = (int)(my_halves * 2);                          // passes of this workgroup (even)
    auto row0_of = [&](int p) __attribute__((always_inline)) { return ((xbase + wgx + (long)(p >> 1) * W) * 2 + (p & 1)) * (long)QR_PASS_ROWS; };
    // ---- per-lane constants ----
    // fragment reads: row l31 (+ 32 mb) of the pass, 16-byte slot (kk & 3) * 2 + khalf of the K step, XOR swizzle of swz_off()
    unsigned sw[4];
#pragma unroll
    for (int j = 0; j < 4; j++) sw[j] = (unsigned)(l31 * 128 + (((j * 2 + khalf) ^ ((l31 >> 1) & 7)) << 4));
    // DMA pieces: piece i of wave w = K step i >> 1, row group 4 (i & 1) + w (8 rows x 128 B); the lane's source offset inside the K step's slab pair
    const int prow = lane >> 3, pslot = lane & 7;
    const int ksl = pslot ^ (((wid & 1) * 4 + (prow >> 1)) & 7);
    const unsigned po0 = (unsigned)((ksl >> 2) * 16384 + (wid * 8 + prow) * 64 + (ksl & 3) * 16);
    const long tile_bytes = (long)NKS * 2 * 16384;              // int8 shadow bytes of one 256-row tile
    auto pass_src = [&](int p) __attribute__((always_inline)) {                                 // uniform: first byte of the pass's rows in slab 0 of its tile
        const long r0 = row0_of(p);
        return reinterpret_cast<const char*>(X8) + (r0 >> 8) * tile_bytes + (r0 & 255) * 64;
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;      // LDS byte address of the ring
    auto dma_piece = [&](const char* src, int slot, int i) __attribute__((always_inline)) {
        qr_dma16(src + (long)(i >> 1) * 32768, po0 + (unsigned)((i & 1) * 2048), lds0 + (unsigned)(slot * STAGE + (i >> 1) * 8192 + ((i & 1) * 4 + wid) * 1024));
    };
    auto dma_rn = [&](int p, int slot) __attribute__((always_inline)) {                         // MODE 1: 16 row norms per wave (lanes 16 w .. 16 w + 15 of a 4-byte piece)
        if constexpr (MODE == 1) {
            const long r0 = row0_of(p);
            const long last = n - 1 - r0;                          // rows past n read the last row's norm (their keys are masked)
            const unsigned voff = (unsigned)((long)lane < last ? (long)lane : (last > 0 ? last : 0)) * 4u;
            if ((lane >> 4) == wid) qr_dma4(reinterpret_cast<const char*>(rn + r0), voff, lds0 + (unsigned)(RING * STAGE + slot * 256));
        }
    };
    // The pass's 64 eligibility bytes (soft deletes / WithDocumentIDs) ride the same DMA stream into LDS, 16 bytes per wave (lanes 4 w .. 4 w + 3 of a
    // 4-byte piece): the masked selection never issues a vector-memory load of its own (a compiler-visible load in the loop would make hipcc wait
    // vmcnt(0) and drain the ring once per pass). ELIG is a template parameter — the instantiation without a filter issues nothing (the extra piece
    // and its address arithmetic cost the unfiltered scan 8 %) — so that every wave's per-pass operation count, what the counted waits rely on, is a
    // compile-time constant either way.
    auto dma_el = [&](int p, int slot) __attribute__((always_inline)) {
        if constexpr (ELIG) {
            const long r0 = row0_of(p);
            const long lastdw = ((n - 1 - r0) > 0 ? (n - 1 - r0) : 0) & ~3l;           // rows past n are masked by their index: read the last valid dword instead
            const unsigned voff = (unsigned)((long)lane * 4 < lastdw ? (long)lane * 4 : lastdw);
            if ((lane >> 2) == wid) qr_dma4(reinterpret_cast<const char*>(elig) + r0, voff, lds0 + (unsigned)(RING * STAGE + (RING + 1) * 256 + slot * 64));
        }
    };
    // ---- prologue: passes 0 .. RING-2 in flight ----
#pragma unroll
    for (int pp = 0; pp < RING - 1; pp++) {
        if (pp < P) {
            const char* src = pass_src(pp);
#pragma unroll
            for (int i = 0; i < PPW; i++) dma_piece(src, pp, i);
            dma_rn(pp, pp);
            dma_el(pp, pp);
        }
    }
    // ---- queries: fragments [group of 32 queries][kk][lane] 16 bytes (prep_queries_i8_kernel), groups 2 w and 2 w + 1 ----
    // Loaded AFTER the prologue's DMA pieces and pinned as landed before the loop: hipcc does not see the asm DMAs, so a counted vmcnt wait it
    // placed for a query load inside the loop would count the ring's pieces as its own younger loads and drain the ring on every pass.
    i32x4v Q[2][NKK];
#pragma unroll
    for (int nb = 0; nb < 2; nb++)
#pragma unroll
        for (int kk = 0; kk < NKK; kk++)
            Q[nb][kk] = *reinterpret_cast<const i32x4v*>(Q8F + ((long)(wid * 2 + nb) * NKK + kk) * 1024 + lane * 16);
    float sqv[2], qnv[2];
#pragma unroll
    for (int nb = 0; nb < 2; nb++) { sqv[nb] = sq[wid * 64 + nb * 32 + l31]; qnv[nb] = MODE == 1 ? qn[wid * 64 + nb * 32 + l31] : 0.0f; }
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
#pragma unroll
        for (int kk = 0; kk < NKK; kk++) asm volatile("" : "+a"(Q[nb][kk]));     // "a": the fragments live in the accumulator half of the register file (MFMA reads B from there)
        asm volatile("" : "+v"(sqv[nb]), "+v"(qnv[nb]));
    }
    const float INF = __builtin_inff();
    i32x16 acc[2][2][2];                                         // [parity][mb][nb]
    // Selection keys. MODE 0 (cosine): the score s_T s_q acc is monotone in the integer sum, so the network runs on EXACT integer keys
    // (acc << 7) | row-in-unit (|acc| <= 127^2 * 768 < 2^24: no overflow; bit 2 of the row = the half-wave, added at the merge) — no
    // conversion, nothing lost to the packing. MODE 1 (L2 family): float keys 2 s - rn with the row in the low 8 mantissa bits, as in
    // scan_epilogue_q. NONE = "no row yet": below every key, and still below every key with the half-wave bit or-ed in.
    constexpr bool IK = MODE == 0;
    const int NONE = IK ? (int)0x80000000 : __float_as_int(-3.0e38f);
    auto is_none = [&](int v) __attribute__((always_inline)) { if constexpr (IK) return v < -2000000000; else return __int_as_float(v) < -1.0e38f; };
    int t0[2], t1[2], t2[2];                                     // running three largest of the current key unit, per query block
#pragma unroll
    for (int nb = 0; nb < 2; nb++) { t0[nb] = NONE; t1[nb] = NONE; t2[nb] = NONE; }
    const float sq_lane = khalf ? sqv[1] : sqv[0], qn_lane = khalf ? qnv[1] : qnv[0];     // of the query this lane STORES keys for (block khalf, query l31)

    // Unit end in eight small stages (issued one per MFMA gap behind the selection): the half-waves trade blocks — after ONE
    // v_permlane32_swap per value the lower half holds both halves' triples of query block 0 (its own rows in x, the rows + 4 in y) and the
    // upper half those of block 1 (x: the lower half's rows, y: its own rows + 4) — merge, turn the three survivors into keys, store.
    int ux0 = 0, ux1 = 0, ux2 = 0, uy0 = 0, uy1 = 0, uy2 = 0; float uk0 = 0.0f, uk1 = 0.0f, uk2 = 0.0f;
    auto to_key = [&](int v, float stv) __attribute__((always_inline)) {
        float a; unsigned row;
        if constexpr (IK) { row = (unsigned)v & 0x7Fu; a = 1.0f - (sq_lane * stv) * (float)(v >> 7); }
        else { row = (unsigned)v & 0xFFu; a = qn_lane - __uint_as_float((unsigned)v & 0xFFFFFF00u); }    // the key is -(rn - 2 s)
        a = fmaxf(a, 0.0f);
        const float k = __uint_as_float((__float_as_uint(a) & 0xFFFFFF00u) | row);
        return is_none(v) ? INF : k;
    };
    auto unit_stage = [&](auto S_c, long un, float stv, bool store) __attribute__((always_inline)) {
        constexpr int S = decltype(S_c)::value;
        if constexpr (S == 0) {
            const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)t0[0], (unsigned)t0[1], false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)t1[0], (unsigned)t1[1], false, false);
            const auto r2 = __builtin_amdgcn_permlane32_swap((unsigned)t2[0], (unsigned)t2[1], false, false);
            ux0 = (int)r0[0]; ux1 = (int)r1[0]; ux2 = (int)r2[0];
            uy0 = (int)r0[1] | 4; uy1 = (int)r1[1] | 4; uy2 = (int)r2[1] | 4;       // y: the rows + 4 of the block (NONE stays below every key)
#pragma unroll
            for (int nb = 0; nb < 2; nb++) { t0[nb] = NONE; t1[nb] = NONE; t2[nb] = NONE; }
        } else if constexpr (S == 1) qr_ins3<IK>(ux0, ux1, ux2, uy0);
        else if constexpr (S == 2) qr_ins3<IK>(ux0, ux1, ux2, uy1);
        else if constexpr (S == 3) qr_ins3<IK>(ux0, ux1, ux2, uy2);
        else if constexpr (S == 4) uk0 = to_key(ux0, stv);
        else if constexpr (S == 5) uk1 = to_key(ux1, stv);
        else if constexpr (S == 6) uk2 = to_key(ux2, stv);
        else if (store) {
            int q = wid * 64 + lane;                              // lane = khalf * 32 + l31: query l31 of block khalf
            asm volatile("" : "+v"(q));                          // (keeps the output addresses out of the loop-carried state)
            S0[(long)q * ldS + 2 * un] = uk0;
            S0[(long)q * ldS + 2 * un + 1] = uk1;
            bound[(long)q * ldB + un] = uk2;
        }
    };
    constexpr int UNIT_STAGES = 8;

    // one element of the selection: block (mb, nb), accumulator register e of the pass with unit half `ppar`
    auto select_one = [&](const i32x16 (&prev)[2][2], int ppar, int prs, float s2_0, float s2_1, int x, bool ok) __attribute__((always_inline)) {
        // element order: nb fastest (two independent chains), then e, then mb
        const int xnb = x & 1, xe = (x >> 1) & 15, xmb = x >> 5;
        const int rconst = (UR == 128 ? ppar * 64 : 0) + xmb * 32 + (xe & 3) + 8 * (xe >> 2);
        int v_;
        if constexpr (IK) v_ = (prev[xmb][xnb][xe] << 7) | rconst;
        else {
            const float rnv = rn_ring[prs * 64 + xmb * 32 + (xe & 3) + 8 * (xe >> 2) + 4 * khalf];
            const float f = __builtin_fmaf(xnb ? s2_1 : s2_0, (float)prev[xmb][xnb][xe], -rnv);
            v_ = (int)((__float_as_uint(f) & 0xFFFFFF00u) | (unsigned)rconst);
        }
        if (!ok) v_ = NONE;
        qr_ins3<IK>(t0[xnb], t1[xnb], t2[xnb], v_);
    };
    // standalone (not interleaved) selection of a finished pass: the workgroup's last pass, and every pass that needs row masks
    // (rows past n, soft deletes, filters)
    auto epilogue_plain = [&](const i32x16 (&prev)[2][2], int ppar, long prow0, int prs, float stv, bool check) __attribute__((always_inline)) {
        unsigned okm = 0xFFFFFFFFu;
        if (check) {
            okm = 0u;
#pragma unroll
            for (int mb = 0; mb < 2; mb++)
#pragma unroll
                for (int e4 = 0; e4 < 4; e4++) {
                    unsigned dw = 0x01010101u;                                                        // the 4 consecutive rows e & 3 = 0 .. 3
                    if constexpr (ELIG) dw = el_ring[prs * 16 + mb * 8 + e4 * 2 + khalf];
#pragma unroll
                    for (int e1 = 0; e1 < 4; e1++) {
                        const long r = prow0 + mb * 32 + e1 + 8 * e4 + 4 * khalf;
                        if (r < n && ((dw >> (8 * e1)) & 0xFFu)) okm |= 1u << (mb * 16 + e4 * 4 + e1);
                    }
                }
        }
        const float s2_0 = 2.0f * (sqv[0] * stv), s2_1 = 2.0f * (sqv[1] * stv);
        qr_static_for<0, 64>([&](auto X) __attribute__((always_inline)) {
            constexpr int x = decltype(X)::value;
            select_one(prev, ppar, prs, s2_0, s2_1, x, (okm >> ((x >> 5) * 16 + ((x >> 1) & 15))) & 1u);
        });
        if (UR == 64 || ppar == 1) qr_static_for<0, UNIT_STAGES>([&](auto S) __attribute__((always_inline)) { unit_stage(S, prow0 / UR, stv, true); });
    };

    // ---- first pass ready: own pieces of pass 0 landed (the younger passes of the prologue may still be in flight), then everybody's ----
    qr_wait_passes<VMW, RING - 2>(RING - 2 < P - 1 ? RING - 2 : P - 1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // A fragments, double-buffered: [buffer][mb]
    i32x4v A[2][2];
    auto lds_frag = [&](unsigned base, int kk, int mb) __attribute__((always_inline)) {
        return *reinterpret_cast<const i32x4v*>(smem + (kk >> 2) * 8192 + mb * 4096 + base);
    };
    A[0][0] = lds_frag(sw[0], 0, 0); A[0][1] = lds_frag(sw[0], 0, 1);

    // ---- uniform state of the pipeline ----
    int slot = 0;                                                // ring slot of the current pass
    int rslot = 0;                                               // its slot in the row-norm ring (RING + 1 entries: the norms of pass p - 1 are read by its selection all through pass p)
    long row0 = row0_of(0);                                      // first row of the current pass
    float st_cur = sx[row0 >> 8];                                // its tile's scale
    const char* dsrc = pass_src(RING - 1 < P ? RING - 1 : P - 1); // source of the pass the current pass issues the DMA pieces of (pass p + RING - 1)
    long prev_row0 = 0; int prev_rslot = 0; float st_prev = 1.0f; bool prev_check = false;
    const i32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // s_memtime stamps of workgroup 8 — compiled in only for tools/scan_check.hip (-DQR_TRACE): even the untaken branches cost issue slots in the pass
#ifdef QR_TRACE
    const bool tr = trace != nullptr && blockIdx.x == 8 && lane == 0;
    auto stamp = [&](int p, int slot_) __attribute__((always_inline)) { if (tr && p < 24) trace[(wid * 24 + p) * 4 + slot_] = __builtin_amdgcn_s_memtime(); };
#else
    auto stamp = [&](int, int) __attribute__((always_inline)) {};
    (void)trace;
#endif

    // The body of pass p with accumulator parity PAR (= p & 1 = the pass's 64-row half of its 128-row unit). INTER: the previous pass's
    // selection is issued between this pass's MFMAs (spread over the first 3/4 of them; the unit's merge + stores follow in eight stages).
    // DMA: the pieces of pass p + RING - 1 are issued (one per MFMA group, no branch). Every index below is a compile-time constant
    // (static_for): accumulators, fragments and queries stay in registers. The scalar bookkeeping of pass p + 1 (first row, tile scale, the
    // next DMA source) is computed in the middle of the pass, so that nothing but register moves sits between two passes' MFMAs.
    auto pass_body = [&](auto PAR_c, auto INTER_c, auto DMA_c, int p) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_c)::value, PPAR = PAR ^ 1;
        constexpr bool INTER = decltype(INTER_c)::value, DMA = decltype(DMA_c)::value;
        stamp(p, 0);
        const int nslot = slot + 1 == RING ? 0 : slot + 1;
        const int fslot = slot == 0 ? RING - 1 : slot - 1;   // the slot pass p - 1 has left = where pass p + RING - 1 lands
        const unsigned fr0 = sw[0] + slot * STAGE, fr1 = sw[1] + slot * STAGE, fr2 = sw[2] + slot * STAGE, fr3 = sw[3] + slot * STAGE;
        const unsigned nfr0 = sw[0] + nslot * STAGE;
        const float s2_0 = 2.0f * (sqv[0] * st_prev), s2_1 = 2.0f * (sqv[1] * st_prev);
        const long un_prev = prev_row0 / UR;
        const bool store_prev = p > 0;
        long nrow0 = row0; float nst = st_cur; const char* ndsrc = dsrc;
        qr_static_for<0, NKK>([&](auto KK) __attribute__((always_inline)) {
            constexpr int kk = decltype(KK)::value, ab = kk & 1;
            if constexpr (kk == NKK - 1) {
                // pass p + 1 readable by all: this wave's pieces of it have landed once only the passes issued after it are outstanding;
                // this wave's fragment reads of pass p are done (lgkmcnt), so after the barrier pass p's slot may be overwritten
                if (p + 1 < P) {
                    stamp(p, 1);
                    qr_wait_passes<VMW, RING - 2>(RING - 2 < P - 2 - p ? RING - 2 : P - 2 - p);
                    stamp(p, 2);
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    stamp(p, 3);
                    A[ab ^ 1][0] = lds_frag(nfr0, 0, 0); A[ab ^ 1][1] = lds_frag(nfr0, 0, 1);
                }
            } else {
                constexpr int j = (kk + 1) & 3;
                const unsigned fb = j == 0 ? fr0 : (j == 1 ? fr1 : (j == 2 ? fr2 : fr3));
                A[ab ^ 1][0] = lds_frag(fb, kk + 1, 0); A[ab ^ 1][1] = lds_frag(fb, kk + 1, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            qr_static_for<0, 4>([&](auto J) __attribute__((always_inline)) {
                constexpr int mb = decltype(J)::value >> 1, nb = decltype(J)::value & 1, m = kk * 4 + decltype(J)::value;
                if constexpr (kk == 0) acc[PAR][mb][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[ab][mb], Q[nb][kk], zero16, 0, 0, 0);
                else acc[PAR][mb][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[ab][mb], Q[nb][kk], acc[PAR][mb][nb], 0, 0, 0);
                if constexpr (INTER) {
                    constexpr int NM = 4 * NKK, NME = NM - UNIT_STAGES - NM / 8;
                    constexpr int e_lo = m * 64 / NME < 64 ? m * 64 / NME : 64, e_hi = (m + 1) * 64 / NME < 64 ? (m + 1) * 64 / NME : 64;
                    qr_static_for<e_lo, e_hi>([&](auto X) __attribute__((always_inline)) { select_one(acc[PPAR], PPAR, prev_rslot, s2_0, s2_1, decltype(X)::value, true); });
                    if constexpr (m >= NME && m < NME + UNIT_STAGES && (UR == 64 || PPAR == 1)) unit_stage(std::integral_constant<int, m - NME>{}, un_prev, st_prev, store_prev);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // one DMA piece of pass p + RING - 1 per MFMA group, into the slot pass p - 1 has left (free since the barrier of pass p - 1)
            if constexpr (DMA && kk < PPW) dma_piece(dsrc, fslot, kk);
            else if constexpr (DMA && kk == PPW) dma_rn(p + RING - 1, rslot + RING - 1 > RING ? rslot - 2 : rslot + RING - 1);
            else if constexpr (DMA && ELIG && kk == PPW + 1) dma_el(p + RING - 1, rslot + RING - 1 > RING ? rslot - 2 : rslot + RING - 1);
            else if constexpr (kk == PPW + 2) {
                // the next pass's scalars, a dozen MFMA groups before they are needed
                if (p + 1 < P) { nrow0 = row0_of(p + 1); nst = sx[nrow0 >> 8]; }
                ndsrc = pass_src(p + RING < P ? p + RING : P - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        prev_row0 = row0; prev_rslot = rslot; st_prev = st_cur;
        prev_check = (n - row0 < QR_PASS_ROWS) || ELIG;
        row0 = nrow0; st_cur = nst; dsrc = ndsrc;
        slot = nslot;
        rslot = rslot == RING ? 0 : rslot + 1;
    };
    // p = 0 has no predecessor: it runs the interleaved body over zeroed "previous" accumulators (their keys are dropped at the unit end: store_prev)
#pragma unroll
    for (int mb = 0; mb < 2; mb++)
#pragma unroll
        for (int nb = 0; nb < 2; nb++) acc[1][mb][nb] = zero16;
    auto one_pass = [&](auto PAR_c, int p) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_c)::value;
        using T = std::integral_constant<bool, true>; using F = std::integral_constant<bool, false>;
        const bool dma = p + RING - 1 < P;
        if (prev_check) {                                        // rare: the previous pass needs row masks — its selection runs on its own, then a plain body
            if (p > 0) epilogue_plain(acc[PAR ^ 1], PAR ^ 1, prev_row0, prev_rslot, st_prev, true);
            if (dma) pass_body(PAR_c, F{}, T{}, p); else pass_body(PAR_c, F{}, F{}, p);
        } else if (dma) pass_body(PAR_c, T{}, T{}, p);
        else pass_body(PAR_c, T{}, F{}, p);
    };
    for (int p = 0; p < P; p += 2) {
        one_pass(std::integral_constant<int, 0>{}, p);
        one_pass(std::integral_constant<int, 1>{}, p + 1);
    }
    // the last pass (parity 1) has no successor to hide its selection under
    epilogue_plain(acc[1], 1, prev_row0, prev_rslot, st_prev, prev_check);
}
// ------------------------------------------------------------------------------------------------
// Register-stationary NARROW tile (at most 64 queries: the reference's API runs ONE query per Execute()): the scan is HBM-bound, so the
// kernel is built around the row stream. All four waves keep the SAME 64 queries in registers (ld8 / 4 AGPRs) and split the ROWS: a wave
// owns whole 64-row key units, streams their rows through a PRIVATE LDS ring by LDS-DMA (slabs of 32 rows x 128 bytes, re-issued as soon as
// their fragments are in registers) and waits on its own vmcnt only — no barrier anywhere, no cross-wave traffic, a quarter of the wide
// tile's MFMA work per row. The selection of a 32-row pass runs under the next pass's MFMAs exactly as in flat_scan_qr_kernel.
// ------------------------------------------------------------------------------------------------
template <int NKS> struct QnGeom {
    static constexpr int NKK = NKS * 4;
    static constexpr int C = NKS == 6 ? 1 : 2;                  // passes in a wave's ring
    static constexpr int RS = C * NKS;                          // slabs (32 rows x 128 B = 4 KiB) in a wave's ring
    static constexpr int WRB = RS * 4096;                       // ring bytes per wave
    static constexpr int LDS = 4 * WRB + 4 * (C + 2) * (128 + 32);   // + per wave the row norms (MODE 1) and the eligibility bytes of C + 2 passes
};
template <int MODE, int NKS>
__global__ __launch_bounds__(QR_THREADS, 1) void flat_scan_qn_kernel(const signed char* __restrict__ X8, long n, const signed char* __restrict__ Q8F,
                                                                     const float* __restrict__ rn, const float* __restrict__ qn,
                                                                     const unsigned char* __restrict__ elig,
                                                                     float* __restrict__ S0, long ldS, float* __restrict__ bound, long ldB, long n_tiles,
                                                                     const float* __restrict__ sx, const float* __restrict__ sq) {
    using G = QnGeom<NKS>;
    constexpr int NKK = G::NKK, C = G::C, RS = G::RS, WRB = G::WRB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, khalf = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* rn_ring = reinterpret_cast<float*>(smem + 4 * WRB) + wid * (C + 2) * 32;
    const unsigned* el_ring = reinterpret_cast<const unsigned*>(smem + 4 * WRB + 4 * (C + 2) * 128) + wid * (C + 2) * 8;     // 8 dwords (32 eligibility bytes) per pass
    // ---- work split: 64-row key units; every XCD owns a contiguous range, its waves take them round robin ----
    const long nx = 8, xcd = blockIdx.x % nx, WX = (gridDim.x / nx) * 4, wx = (blockIdx.x / nx) * 4 + wid;
    const long U = n_tiles * 4, uq = U / nx, urem = U % nx;
    const long xbase = xcd < urem ? xcd * (uq + 1) : urem * (uq + 1) + (xcd - urem) * uq, xcount = xcd < urem ? uq + 1 : uq;
    const long my_units = wx < xcount ? (xcount - wx + WX - 1) / WX : 0;
    if (my_units == 0) return;
    const int P = (int)(my_units * 2);                           // 32-row passes of this wave (even)
    const int T = P * NKS;                                       // slabs of this wave
    auto row0_of = [&](int p) __attribute__((always_inline)) { return ((xbase + wx + (long)(p >> 1) * WX) * 2 + (p & 1)) * 32L; };
    unsigned sw[4];
#pragma unroll
    for (int j = 0; j < 4; j++) sw[j] = (unsigned)(wid * WRB + l31 * 128 + (((j * 2 + khalf) ^ ((l31 >> 1) & 7)) << 4));
    // DMA pieces of a slab: piece i = rows 8 i .. 8 i + 7 of the pass (8 rows x 128 B); the lane's source offset inside the K step's slab pair
    const int prow = lane >> 3, pslot = lane & 7;
    const int ks_e = pslot ^ ((prow >> 1) & 7), ks_o = pslot ^ ((4 + (prow >> 1)) & 7);
    const unsigned po_e = (unsigned)((ks_e >> 2) * 16384 + prow * 64 + (ks_e & 3) * 16), po_o = (unsigned)((ks_o >> 2) * 16384 + (8 + prow) * 64 + (ks_o & 3) * 16);
    const long tile_bytes = (long)NKS * 2 * 16384;
    auto pass_src = [&](int p) __attribute__((always_inline)) {
        const long r0 = row0_of(p);
        return reinterpret_cast<const char*>(X8) + (r0 >> 8) * tile_bytes + (r0 & 255) * 64;
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto dma_piece = [&](const char* src /*of the pass*/, int ks, int slot, int i) __attribute__((always_inline)) {
        qr_dma16(src + (long)ks * 32768, ((i & 1) ? po_o : po_e) + (unsigned)((i >> 1) * 1024), lds0 + (unsigned)(wid * WRB + slot * 4096 + i * 1024));
    };
    auto dma_rn = [&](int p) __attribute__((always_inline)) {   // MODE 1: the pass's 32 row norms (lanes 0 .. 31 of a 4-byte piece)
        if constexpr (MODE == 1) {
            const long r0 = row0_of(p);
            const long last = n - 1 - r0;
            const unsigned voff = (unsigned)((long)lane < last ? (long)lane : (last > 0 ? last : 0)) * 4u;
            if (lane < 32) qr_dma4(reinterpret_cast<const char*>(rn + r0), voff, lds0 + (unsigned)(4 * WRB + (wid * (C + 2) + p % (C + 2)) * 128));
        }
    };
    auto dma_el = [&](int p) __attribute__((always_inline)) {   // the pass's 32 eligibility bytes (lanes 0 .. 7 of a 4-byte piece), only when there is a filter
        if (elig) {
            const long r0 = row0_of(p);
            const long lastdw = ((n - 1 - r0) > 0 ? (n - 1 - r0) : 0) & ~3l;
            const unsigned voff = (unsigned)((long)lane * 4 < lastdw ? (long)lane * 4 : lastdw);
            if (lane < 8) qr_dma4(reinterpret_cast<const char*>(elig) + r0, voff, lds0 + (unsigned)(4 * WRB + 4 * (C + 2) * 128 + (wid * (C + 2) + p % (C + 2)) * 32));
        }
    };
    // ---- prologue: the first RS slabs in flight (passes 0 .. C-1) ----
#pragma unroll
    for (int g = 0; g < RS; g++) {
        if (g < T) {
            if (g % NKS == 0) { dma_rn(g / NKS); dma_el(g / NKS); }
            const char* src = pass_src(g / NKS);
#pragma unroll
            for (int i = 0; i < 4; i++) dma_piece(src, g % NKS, g, i);
        }
    }
    // ---- queries 0 .. 63: fragments [group of 32 queries][kk][lane] 16 bytes, groups 0 and 1; pinned as landed before the loop (see flat_scan_qr_kernel) ----
    i32x4v Q[2][NKK];
#pragma unroll
    for (int nb = 0; nb < 2; nb++)
#pragma unroll
        for (int kk = 0; kk < NKK; kk++)
            Q[nb][kk] = *reinterpret_cast<const i32x4v*>(Q8F + ((long)nb * NKK + kk) * 1024 + lane * 16);
    float sq_lane = sq[lane], qn_lane = MODE == 1 ? qn[lane] : 0.0f;   // of the query this lane STORES keys for (block khalf, query l31 = query `lane`)
    float sqv0 = sq[l31], sqv1 = sq[32 + l31];
#pragma unroll
    for (int nb = 0; nb < 2; nb++)
#pragma unroll
        for (int kk = 0; kk < NKK; kk++) asm volatile("" : "+a"(Q[nb][kk]));
    asm volatile("" : "+v"(sq_lane), "+v"(qn_lane), "+v"(sqv0), "+v"(sqv1));

    const float INF = __builtin_inff();
    i32x16 acc[2][2];                                            // [parity][nb]
    constexpr bool IK = MODE == 0;
    const int NONE = IK ? (int)0x80000000 : __float_as_int(-3.0e38f);
    auto is_none = [&](int v) __attribute__((always_inline)) { if constexpr (IK) return v < -2000000000; else return __int_as_float(v) < -1.0e38f; };
    int t0[2], t1[2], t2[2];
#pragma unroll
    for (int nb = 0; nb < 2; nb++) { t0[nb] = NONE; t1[nb] = NONE; t2[nb] = NONE; }
    int ux0 = 0, ux1 = 0, ux2 = 0, uy0 = 0, uy1 = 0, uy2 = 0; float uk0 = 0.0f, uk1 = 0.0f, uk2 = 0.0f;
    auto to_key = [&](int v, float stv) __attribute__((always_inline)) {
        float a; unsigned row;
        if constexpr (IK) { row = (unsigned)v & 0x7Fu; a = 1.0f - (sq_lane * stv) * (float)(v >> 7); }
        else { row = (unsigned)v & 0xFFu; a = qn_lane - __uint_as_float((unsigned)v & 0xFFFFFF00u); }
        a = fmaxf(a, 0.0f);
        const float k = __uint_as_float((__float_as_uint(a) & 0xFFFFFF00u) | row);
        return is_none(v) ? INF : k;
    };
    auto unit_stage = [&](auto S_c, long un, float stv, bool store) __attribute__((always_inline)) {
        constexpr int S = decltype(S_c)::value;
        if constexpr (S == 0) {
            const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)t0[0], (unsigned)t0[1], false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)t1[0], (unsigned)t1[1], false, false);
            const auto r2 = __builtin_amdgcn_permlane32_swap((unsigned)t2[0], (unsigned)t2[1], false, false);
            ux0 = (int)r0[0]; ux1 = (int)r1[0]; ux2 = (int)r2[0];
            uy0 = (int)r0[1] | 4; uy1 = (int)r1[1] | 4; uy2 = (int)r2[1] | 4;
#pragma unroll
            for (int nb = 0; nb < 2; nb++) { t0[nb] = NONE; t1[nb] = NONE; t2[nb] = NONE; }
        } else if constexpr (S == 1) qr_ins3<IK>(ux0, ux1, ux2, uy0);
        else if constexpr (S == 2) qr_ins3<IK>(ux0, ux1, ux2, uy1);
        else if constexpr (S == 3) qr_ins3<IK>(ux0, ux1, ux2, uy2);
        else if constexpr (S == 4) uk0 = to_key(ux0, stv);
        else if constexpr (S == 5) uk1 = to_key(ux1, stv);
        else if constexpr (S == 6) uk2 = to_key(ux2, stv);
        else if (store) {
            int q = lane;                                         // lane = khalf * 32 + l31: query l31 of block khalf
            asm volatile("" : "+v"(q));
            S0[(long)q * ldS + 2 * un] = uk0;
            S0[(long)q * ldS + 2 * un + 1] = uk1;
            bound[(long)q * ldB + un] = uk2;
        }
    };
    constexpr int UNIT_STAGES = 8;
    auto select_one = [&](const i32x16 (&prev)[2], int ppar, int prs, float s2_0, float s2_1, int x, bool ok) __attribute__((always_inline)) {
        const int xnb = x & 1, xe = x >> 1;
        const int rconst = ppar * 32 + (xe & 3) + 8 * (xe >> 2);
        int v_;
        if constexpr (IK) v_ = (prev[xnb][xe] << 7) | rconst;
        else {
            const float rnv = rn_ring[prs * 32 + (xe & 3) + 8 * (xe >> 2) + 4 * khalf];
            const float f = __builtin_fmaf(xnb ? s2_1 : s2_0, (float)prev[xnb][xe], -rnv);
            v_ = (int)((__float_as_uint(f) & 0xFFFFFF00u) | (unsigned)rconst);
        }
        if (!ok) v_ = NONE;
        qr_ins3<IK>(t0[xnb], t1[xnb], t2[xnb], v_);
    };
    auto epilogue_plain = [&](const i32x16 (&prev)[2], int ppar, long prow0, int prs, float stv, bool check) __attribute__((always_inline)) {
        unsigned okm = 0xFFFFu;
        if (check) {
            okm = 0u;
#pragma unroll
            for (int e4 = 0; e4 < 4; e4++) {
                const unsigned dw = elig ? el_ring[prs * 8 + e4 * 2 + khalf] : 0x01010101u;
#pragma unroll
                for (int e1 = 0; e1 < 4; e1++) {
                    const long r = prow0 + e1 + 8 * e4 + 4 * khalf;
                    if (r < n && ((dw >> (8 * e1)) & 0xFFu)) okm |= 1u << (e4 * 4 + e1);
                }
            }
        }
        const float s2_0 = 2.0f * (sqv0 * stv), s2_1 = 2.0f * (sqv1 * stv);
        qr_static_for<0, 32>([&](auto X) __attribute__((always_inline)) {
            constexpr int x = decltype(X)::value;
            select_one(prev, ppar, prs, s2_0, s2_1, x, (okm >> (x >> 1)) & 1u);
        });
        if (ppar == 1) qr_static_for<0, UNIT_STAGES>([&](auto S) __attribute__((always_inline)) { unit_stage(S, prow0 / 64, stv, true); });
    };

    // ---- slab 0 landed (only the RS - 1 younger slabs of the prologue may still be in flight) ----
    qr_wait_passes<4, RS - 1>(RS - 1 < T - 1 ? RS - 1 : T - 1);
    i32x4v F[2][4];                                              // fragments of a slab (kk & 3 = 0..3), double-buffered by slab
    auto lds_frag = [&](int slot, int j) __attribute__((always_inline)) { return *reinterpret_cast<const i32x4v*>(smem + slot * 4096 + sw[j]); };
#pragma unroll
    for (int j = 0; j < 4; j++) F[0][j] = lds_frag(0, j);

    long row0 = row0_of(0); float st_cur = sx[row0 >> 8];
    const char* dsrc = pass_src(C < P ? C : P - 1);              // source of the pass whose slabs the current pass issues (pass p + C)
    long prev_row0 = 0; float st_prev = 1.0f; bool prev_check = false;
    const i32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int nb = 0; nb < 2; nb++) acc[1][nb] = zero16;

    auto pass_body = [&](auto PAR_c, auto INTER_c, int p) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_c)::value, PPAR = PAR ^ 1;
        constexpr bool INTER = decltype(INTER_c)::value;
        const float s2_0 = 2.0f * (sqv0 * st_prev), s2_1 = 2.0f * (sqv1 * st_prev);
        const long un_prev = prev_row0 / 64;
        const bool store_prev = p > 0;
        const int prs = (p + C + 1) % (C + 2);                   // (p - 1) mod (C + 2)
        const int sbase = C == 1 ? 0 : (p & 1) * NKS;            // ring slot of the pass's first slab
        long nrow0 = row0; float nst = st_cur; const char* ndsrc = dsrc;
        const bool dma = p + C < P;
        qr_static_for<0, NKS>([&](auto KS) __attribute__((always_inline)) {
            constexpr int ks = decltype(KS)::value, fb = ks & 1;   // NKS is even: the fragment buffer of a slab is its K step's parity
            const int g = p * NKS + ks;
            // slab g + 1 (this pass's next K step, or the next pass's first) has landed once only the slabs issued after it are outstanding
            if (g + 1 < T) {
                qr_wait_passes<4, RS - 2>(RS - 2 < T - 2 - g ? RS - 2 : T - 2 - g);
                const int nslot = ks + 1 < NKS ? sbase + ks + 1 : (C == 1 ? 0 : (sbase ? 0 : NKS));
#pragma unroll
                for (int j = 0; j < 4; j++) F[fb ^ 1][j] = lds_frag(nslot, j);
            }
            __builtin_amdgcn_sched_barrier(0);
            qr_static_for<0, 8>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value >> 1, nb = decltype(J)::value & 1, kk = ks * 4 + j, m = ks * 8 + decltype(J)::value;
                if constexpr (kk == 0) acc[PAR][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[fb][j], Q[nb][kk], zero16, 0, 0, 0);
                else acc[PAR][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F[fb][j], Q[nb][kk], acc[PAR][nb], 0, 0, 0);
                if constexpr (INTER) {
                    constexpr int NM = 2 * NKK, NME = NM - UNIT_STAGES - NM / 8;
                    constexpr int e_lo = m * 32 / NME < 32 ? m * 32 / NME : 32, e_hi = (m + 1) * 32 / NME < 32 ? (m + 1) * 32 / NME : 32;
                    qr_static_for<e_lo, e_hi>([&](auto X) __attribute__((always_inline)) { select_one(acc[PPAR], PPAR, prs, s2_0, s2_1, decltype(X)::value, true); });
                    if constexpr (m >= NME && m < NME + UNIT_STAGES && PPAR == 1) unit_stage(std::integral_constant<int, m - NME>{}, un_prev, st_prev, store_prev);
                }
                // the slab's fragments are in registers (the MFMAs above waited for them): its slot takes slab g + RS = K step ks of pass p + C
                if constexpr ((decltype(J)::value & 1) == 1) { if (dma) dma_piece(dsrc, ks, sbase + ks, decltype(J)::value >> 1); }
                if constexpr (decltype(J)::value == 0 && ks == 0) { if (dma) { dma_rn(p + C); dma_el(p + C); } }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (ks == NKS / 2) {
                if (p + 1 < P) { nrow0 = row0_of(p + 1); nst = sx[nrow0 >> 8]; }
                ndsrc = pass_src(p + 1 + C < P ? p + 1 + C : P - 1);
            }
        });
        prev_row0 = row0; st_prev = st_cur;
        prev_check = (n - row0 < 32) || elig != nullptr;
        row0 = nrow0; st_cur = nst; dsrc = ndsrc;
    };
    auto one_pass = [&](auto PAR_c, int p) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_c)::value;
        using TT = std::integral_constant<bool, true>; using FF = std::integral_constant<bool, false>;
        if (prev_check) {
            if (p > 0) epilogue_plain(acc[PAR ^ 1], PAR ^ 1, prev_row0, (p + C + 1) % (C + 2), st_prev, true);
            pass_body(PAR_c, FF{}, p);
        } else pass_body(PAR_c, TT{}, p);
    };
    for (int p = 0; p < P; p += 2) {
        one_pass(std::integral_constant<int, 0>{}, p);
        one_pass(std::integral_constant<int, 1>{}, p + 1);
    }
    epilogue_plain(acc[1], 1, prev_row0, (P - 1) % (C + 2), st_prev, prev_check);
}



// launch the tile of one MODE (0 cosine / 1 L2 family); explicit instantiations: kernels_scanq.hip (0), kernels_scanq_l2.hip (1)
template <int MODE>
void launch_flat_scan_qr_mode(Ctx* c, int nks, const void* X8, int64_t n, const void* Q8F, int nq_used, const float* rn, const float* qn,
                              const float* sx, const float* sq, const uint8_t* elig, float* S0, int64_t ldS, float* bound, int64_t ldB, int unit_rows) {
    const long n_tiles = ceil_div(n, 256);
    if (nq_used <= 64) {        // narrow: the waves split the rows, private LDS rings, no barrier (64-row key units)
        const long gridw = std::max<long>(8, std::min<long>(round_up(ceil_div(n_tiles * 4, 4), 8), (long)round_up(c->prop.multiProcessorCount, 8)));
        auto gow = [&](auto kernel, size_t lds) {
            HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            c->launch_timed("flat_scan_i8_n64", kernel, dim3((unsigned)gridw), dim3(QR_THREADS), lds, (const signed char*)X8, (long)n, (const signed char*)Q8F, rn, qn, (const unsigned char*)elig, S0, (long)ldS,
                            bound, (long)ldB, n_tiles, sx, sq);
        };
        if (nks == 2) gow(flat_scan_qn_kernel<MODE, 2>, QnGeom<2>::LDS); else if (nks == 4) gow(flat_scan_qn_kernel<MODE, 4>, QnGeom<4>::LDS); else gow(flat_scan_qn_kernel<MODE, 6>, QnGeom<6>::LDS);
        LAUNCH_CHECK();
        return;
    }
    const long gridr = std::min<long>(round_up(2 * n_tiles, 8), (long)round_up(c->prop.multiProcessorCount, 8));
    auto gor = [&](auto kernel, size_t lds) {
        HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->launch_timed("flat_scan_i8", kernel, dim3((unsigned)gridr), dim3(QR_THREADS), lds, (const signed char*)X8, (long)n, (const signed char*)Q8F, rn, qn, (const unsigned char*)elig, S0, (long)ldS,
                        bound, (long)ldB, n_tiles, sx, sq, g_scan_trace);
    };
#define QR_GO4(UR, NKS) do { if (elig) gor(flat_scan_qr_kernel<MODE, UR, NKS, true>, QrGeom<NKS>::LDS); else gor(flat_scan_qr_kernel<MODE, UR, NKS, false>, QrGeom<NKS>::LDS); } while (0)
#define QR_GO(NKS) do { if (unit_rows == 64) QR_GO4(64, NKS); else QR_GO4(128, NKS); } while (0)
    if (nks == 2) QR_GO(2); else if (nks == 4) QR_GO(4); else QR_GO(6);
#undef QR_GO
#undef QR_GO4
    LAUNCH_CHECK();
}

}  // namespace comet
