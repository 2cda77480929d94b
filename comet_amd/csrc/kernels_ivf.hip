// kernels_ivf.hip — the MFMA fast path of the IVF list scan (ivf_index_search.go:277-301) for gfx950.
//
// The reference scans every member of the nprobes nearest lists exactly. The exact kernel (dist_list_kernel) is VALU-bound: three
// non-fusable float32 operations per element pair on serial chains. This path does what the Flat fast path does (kernels_fast.hip),
// restricted to the probed lists:
//
//   1. a half-precision SHADOW of the vectors in SLOT order (list-major, every list padded to 64-row units), tiled
//      [unit of 64 slots][K step of 64 halves][row][64 halves] so that the 8 KiB a wave pair needs for one K step of one unit are
//      contiguous in HBM and a list's units follow each other;
//   2. ivf_items_kernel (one workgroup): counting sort of the batch's (query, probed list) pairs by list, the queries of a list cut
//      into GROUPS of <= 64, every (group, 256-row tile of the list) an ITEM; tile-major inside a list, so the groups that share a
//      tile run next to each other on one XCD and share its rows through that XCD's L2;
//   3. ivf_scan_f16_kernel: persistent 512-thread workgroups (two per CU), one item at a time: S = Xh . Qh^T on
//      v_mfma_f32_32x32x16_f16, waves as 4 (units of the tile) x 2 (query blocks of 32), rows and the gathered query rows staged
//      with global_load_lds (XOR swizzle on the source side). A wave owns whole (query, 64-row unit) pairs and writes the pair's
//      64 approximate distances into the query's SCORE ROW at the pair's unit offset, i.e. in probe order. Unlike the Flat scan
//      (B x N scores = 1 GB per batch) an IVF batch has only B x nprobe x len scores — 32 MB at 1M rows, nlist 1024, nprobe 32,
//      B 256 against 1.5 GB of rows read — so nothing is gained by reducing them in the epilogue, and the post stage gets the
//      exact K-th smallest approximation instead of a bound derived from two keys per unit;
//   4. the post stage of the Flat fast path with the IVF geometry (kernels_fast.hip: fast_post_kernel<METRIC, IvfGeom>):
//      kappa = K-th smallest approximate distance, tau = kappa + 2E, candidates = every row with approximation <= tau, exact
//      float32 rescoring in the reference's order, (score, scan position) order — bit-identical to the strict path.
//
// Measured and not kept (round 3): fetching the NEXT item's descriptor during the current item and requesting its first K step before the
// current item's epilogue (scan 0.2785 ms against 0.278: the second workgroup of the CU already covers an item's prologue and epilogue).
//
// The scan is HBM-bound by construction: every probed list is streamed once per group of 64 of its queries (12 GFLOP of MFMA work
// for 1.5 GB of rows at 1M x 768, nprobe 32, B 256). Algorithmic bytes per launch = sum over items of the tile's rows x ldh x 2.
#include <type_traits>
#include "kernels.hpp"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

namespace comet {

constexpr int IV_UNIT = 64;        // rows per key unit = list alignment in the slot layout
constexpr int IV_TILE = 256;       // rows per item (4 units)
constexpr int IV_NQ = 64;          // queries per group
constexpr int IV_K = 64;           // halves per K step
constexpr int IV_THREADS = 512;
constexpr int IV_STAGE = 32768 + 8192;
constexpr int IV_MAX_LISTS = 8192; // bins of the item builder (3 int arrays in LDS)

int ivf_fast_unit_rows() { return IV_UNIT; }
int ivf_fast_max_lists() { return IV_MAX_LISTS; }

// ------------------------------------------------------------------------------------------------
// shadow: slot-ordered fp16 rows (+ squared norms per slot, magnitude statistics)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ long ivf_tiled_off(long slot, int k, int ldh) {
    const long unit = slot >> 6; const int r = (int)(slot & 63);
    return ((unit * (ldh >> 6) + (k >> 6)) * 64 + r) * 64 + (k & 63);
}
__global__ __launch_bounds__(256) void ivf_shadow_kernel(const float* __restrict__ V, int ld, const unsigned* __restrict__ row_of_slot, long nslots,
                                                         _Float16* __restrict__ Vh, int ldh, float* __restrict__ rn, unsigned* __restrict__ stats) {
    // one wave per slot; padding slots (no row) become zero rows
    const int lane = threadIdx.x & 63;
    const long slot = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= nslots) return;
    const unsigned row = row_of_slot[slot];
    const float* x = V + (long)row * ld;
    float s = 0.0f, mx = 0.0f;
    for (int i = lane; i < ldh; i += 64) {
        const float v = (row != 0xFFFFFFFFu && i < ld) ? x[i] : 0.0f;
        Vh[ivf_tiled_off(slot, i, ldh)] = (_Float16)v;
        s += v * v;
        mx = fmaxf(mx, fabsf(v));
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); mx = fmaxf(mx, __shfl_xor(mx, off, 64)); }
    if (lane == 0) {
        rn[slot] = s;
        atomicMax(&stats[0], __float_as_uint(mx)); atomicMax(&stats[1], __float_as_uint(s));
    }
}
void launch_ivf_shadow(Ctx* c, const float* V, int ld, const uint32_t* row_of_slot, int64_t nslots, void* Vh, int ldh, float* rn, uint32_t* stats) {
    if (nslots <= 0) return;
    ProfScope ps(c, "ivf_shadow");
    ivf_shadow_kernel<<<dim3((unsigned)ceil_div(nslots, 4)), dim3(256), 0, c->stream>>>(V, ld, row_of_slot, nslots, (_Float16*)Vh, ldh, rn, stats);
    LAUNCH_CHECK();
}

// int8 shadow (round 3; the Flat one's rationale and bound: kernels_fast.hip to_i8_tiles_kernel): x_i = s_U c_i + delta_i with ONE SCALE PER
// 64-SLOT UNIT (a unit belongs to one list and is what a wave scans), codes in the fp16 shadow's layout byte for byte —
// [unit][K step of 128 bytes][row][128 bytes], a K step now holds 128 dimensions; ld8 = dimensions padded to 128. The scan is HBM-bound:
// half the bytes. stats[2] = max ||delta||^2 over all rows (atomicMax), su[unit] = the unit's scale.
__global__ __launch_bounds__(256) void ivf_shadow_i8_kernel(const float* __restrict__ V, int ld, const unsigned* __restrict__ row_of_slot, long nslots,
                                                            signed char* __restrict__ V8, int ld8, float* __restrict__ su, unsigned* __restrict__ stats) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long unit = blockIdx.x, s0 = unit * 64;
    float amax = 0.0f;
    for (int r = w; r < 64; r += 4) {
        const long slot = s0 + r;
        const unsigned row = slot < nslots ? row_of_slot[slot] : 0xFFFFFFFFu;
        if (row == 0xFFFFFFFFu) continue;                      // wave-uniform
        const float* x = V + (long)row * ld;
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
    const int nk = ld8 >> 7;
    float e2max = 0.0f;
    for (int r = w; r < 64; r += 4) {
        const long slot = s0 + r;
        const unsigned row = slot < nslots ? row_of_slot[slot] : 0xFFFFFFFFu;
        const float* x = V + (long)(row == 0xFFFFFFFFu ? 0 : row) * ld;
        float e2 = 0.0f;
        for (int i0 = lane * 16; i0 < ld8; i0 += 1024) {
            u32x4 out;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f32x4v v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (row != 0xFFFFFFFFu && i0 + 4 * j < ld) v = *reinterpret_cast<const f32x4v*>(x + i0 + 4 * j);
                float cq[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    cq[e] = fminf(fmaxf(rintf(v[e] / s), -127.0f), 127.0f);
                    const float dl = v[e] - s * cq[e];
                    e2 += dl * dl;
                }
                out[j] = ((unsigned)(int)cq[0] & 0xFFu) | (((unsigned)(int)cq[1] & 0xFFu) << 8) | (((unsigned)(int)cq[2] & 0xFFu) << 16) | (((unsigned)(int)cq[3] & 0xFFu) << 24);
            }
            *reinterpret_cast<u32x4*>(V8 + ((unit * nk + (i0 >> 7)) * 64 + r) * 128 + (i0 & 127)) = out;
        }
        for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off, 64);
        e2max = fmaxf(e2max, e2);
    }
    if (threadIdx.x == 0) su[unit] = s;
    if (lane == 0) atomicMax(&stats[2], __float_as_uint(e2max));
}
void launch_ivf_shadow_i8(Ctx* c, const float* V, int ld, const uint32_t* row_of_slot, int64_t nslots, void* V8, int ld8, float* su, uint32_t* stats) {
    if (nslots <= 0) return;
    ProfScope ps(c, "ivf_shadow_i8");
    ivf_shadow_i8_kernel<<<dim3((unsigned)ceil_div(nslots, 64)), dim3(256), 0, c->stream>>>(V, ld, row_of_slot, nslots, (signed char*)V8, ld8, su, stats);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// per-query unit offsets: uoff[q][0..np] = prefix of ceil(len / 64) over the probed lists (the key row of a query holds its
// pairs' units in probe order), seg_off as probe_segments_kernel
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ivf_probe_units_kernel(const unsigned* __restrict__ probe_list, int ldp, const int* __restrict__ list_len, int np,
                                                             int* __restrict__ seg_off, int* __restrict__ uoff) {
    const int q = blockIdx.x, lane = threadIdx.x;
    int run = 0, urun = 0;
    for (int p0 = 0; p0 < np; p0 += 64) {
        const int p = p0 + lane;
        int len = 0;
        if (p < np) len = list_len[probe_list[(long)q * ldp + p]];
        const int un = (len + IV_UNIT - 1) / IV_UNIT;
        int inc = len, uinc = un;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64), w = __shfl_up(uinc, o, 64); if (lane >= o) { inc += v; uinc += w; } }
        if (p < np) { seg_off[(long)q * (np + 1) + p] = run + inc - len; uoff[(long)q * (np + 1) + p] = urun + uinc - un; }
        run += __shfl(inc, 63, 64); urun += __shfl(uinc, 63, 64);
    }
    if (lane == 0) { seg_off[(long)q * (np + 1) + np] = run; uoff[(long)q * (np + 1) + np] = urun; }
}
void launch_ivf_probe_units(Ctx* c, const uint32_t* probe_list, int ldp, const int32_t* list_len, int B, int np, int32_t* seg_off, int32_t* uoff) {
    if (B <= 0) return;
    ProfScope ps(c, "ivf_probe_units");
    ivf_probe_units_kernel<<<dim3((unsigned)B), dim3(64), 0, c->stream>>>(probe_list, ldp, list_len, np, seg_off, uoff);
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// items
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) IvfGroup { int list, nq, unit0, len; int q[IV_NQ]; int kb[IV_NQ]; };    // 528 bytes
struct IvfItem { int group, tile; };

// One workgroup. counts[0] = number of items, counts[1] = number of groups, counts[2] = 64-row units streamed by the scan.
__global__ __launch_bounds__(1024) void ivf_items_kernel(const unsigned* __restrict__ probe_list, int ldp, int np, const int* __restrict__ uoff, int n_pairs, int nlist,
                                                         const int* __restrict__ list_len, const long* __restrict__ list_base,
                                                         IvfGroup* __restrict__ groups, IvfItem* __restrict__ items, int* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) int ism[];   // cnt / cursor [nlist] | group base [nlist] | item base [nlist]
    __shared__ int part_g[1024], part_i[1024];
    int* cnt = ism; int* gbase = ism + nlist; int* ibase = ism + 2 * nlist;
    const int t = threadIdx.x;
    for (int i = t; i < nlist; i += 1024) cnt[i] = 0;
    __syncthreads();
    auto list_of = [&](int i, int& q, int& pi) { q = i / np; pi = i - q * np; return (int)min(probe_list[(long)q * ldp + pi], (unsigned)(nlist - 1)); };
    for (int i = t; i < n_pairs; i += 1024) { int q, pi; const int l = list_of(i, q, pi); if (list_len[l] > 0) atomicAdd(&cnt[l], 1); }
    __syncthreads();
    // prefix sums of groups and items over the lists
    const int per = (nlist + 1023) / 1024, lo = min(nlist, t * per), hi = min(nlist, lo + per);
    int sg = 0, si = 0;
    for (int l = lo; l < hi; l++) {
        const int ng = (cnt[l] + IV_NQ - 1) / IV_NQ, nt = (list_len[l] + IV_TILE - 1) / IV_TILE;
        sg += ng; si += ng * nt;
    }
    part_g[t] = sg; part_i[t] = si;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int vg = (t >= off) ? part_g[t - off] : 0, vi = (t >= off) ? part_i[t - off] : 0;
        __syncthreads();
        part_g[t] += vg; part_i[t] += vi;
        __syncthreads();
    }
    if (t == 1023) { counts[0] = part_i[1023]; counts[1] = part_g[1023]; }
    {   // counts[2] = 64-row units the scan will stream (every group reads its list once): the algorithmic bytes of the launch / (64 x ldh x 2)
        int su = 0;
        for (int l = lo; l < hi; l++) su += ((cnt[l] + IV_NQ - 1) / IV_NQ) * ((list_len[l] + IV_UNIT - 1) / IV_UNIT);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) su += __shfl_xor(su, off, 64);
        if (t == 0) counts[2] = 0;
        __syncthreads();
        if ((t & 63) == 0 && su) atomicAdd(&counts[2], su);
    }
    int rg = part_g[t] - sg, ri = part_i[t] - si;
    for (int l = lo; l < hi; l++) {
        const int ng = (cnt[l] + IV_NQ - 1) / IV_NQ, nt = (list_len[l] + IV_TILE - 1) / IV_TILE;
        gbase[l] = rg; ibase[l] = ri;
        rg += ng; ri += ng * nt;
    }
    __syncthreads();
    // group headers and items, one per thread and round (a thread per LIST would leave the one that owns a 40 k-row list writing
    // hundreds of items alone): the list of group / item i is the last l with base[l] <= i — lists without groups share their
    // successor's base, so the last of a run of equal bases is the one that owns the index
    const int n_groups = part_g[1023], n_items_all = part_i[1023];
    auto owner = [&](const int* base, int i) { int a = 0, b = nlist; while (b - a > 1) { const int m = (a + b) >> 1; if (base[m] <= i) a = m; else b = m; } return a; };
    for (int gi = t; gi < n_groups; gi += 1024) {
        const int l = owner(gbase, gi), g = gi - gbase[l];
        IvfGroup* G = groups + gi;
        G->list = l; G->nq = min(IV_NQ, cnt[l] - g * IV_NQ); G->unit0 = (int)(list_base[l] >> 6); G->len = list_len[l];
    }
    for (int ii = t; ii < n_items_all; ii += 1024) {      // tile-major inside a list, the groups of one tile adjacent
        const int l = owner(ibase, ii), loc = ii - ibase[l];
        const int ng = (cnt[l] + IV_NQ - 1) / IV_NQ;
        items[ii] = IvfItem{gbase[l] + loc % ng, loc / ng};
    }
    __syncthreads();
    for (int i = t; i < nlist; i += 1024) cnt[i] = 0;     // now the scatter cursors
    __syncthreads();
    for (int i = t; i < n_pairs; i += 1024) {
        int q, pi; const int l = list_of(i, q, pi);
        if (list_len[l] <= 0) continue;
        const int pos = atomicAdd(&cnt[l], 1);
        IvfGroup* G = groups + gbase[l] + pos / IV_NQ;
        G->q[pos % IV_NQ] = q; G->kb[pos % IV_NQ] = uoff[(long)q * (np + 1) + pi];
    }
}
void launch_ivf_items(Ctx* c, const uint32_t* probe_list, int ldp, int np, const int32_t* uoff, int n_pairs, int nlist, const int32_t* list_len,
                      const int64_t* list_base, void* groups, void* items, int32_t* counts) {
    static bool attr_done = false;
    if (!attr_done) { HIP_CHECK(hipFuncSetAttribute((const void*)ivf_items_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, IV_MAX_LISTS * 12)); attr_done = true; }
    ProfScope ps(c, "ivf_items");
    ivf_items_kernel<<<dim3(1), dim3(1024), (size_t)nlist * 12, c->stream>>>(probe_list, ldp, np, uoff, n_pairs, nlist, list_len, (const long*)list_base,
                                                                              (IvfGroup*)groups, (IvfItem*)items, counts);
    LAUNCH_CHECK();
}
size_t ivf_group_bytes() { return sizeof(IvfGroup); }
size_t ivf_item_bytes() { return sizeof(IvfItem); }

// ------------------------------------------------------------------------------------------------
// the scan
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int iv_swz_off(int row, int kslot) { return row * 128 + ((kslot ^ ((row >> 1) & 7)) << 4); }
// MODE 0: cosine   key = max(0, 1 - s)
// MODE 1: L2 family key = max(0, qn[q] + rn[slot] - 2 s)
// I8: the int8 shadow — `ldh` = row bytes / 2, Qh = row-major int8 queries, su / sq = unit / query scales (the integer sums become
// scores in the epilogue; everything else is byte-identical).
template <int MODE, bool I8 = false>
__global__ __launch_bounds__(IV_THREADS) void ivf_scan_f16_kernel(const _Float16* __restrict__ Vh, int ldh, const _Float16* __restrict__ Qh /*row-major fp16 queries, ldh*/,
                                                                  const float* __restrict__ rn /*per slot*/, const float* __restrict__ qn,
                                                                  const unsigned char* __restrict__ elig /*per slot, nullable*/,
                                                                  const IvfGroup* __restrict__ groups, const IvfItem* __restrict__ items, const int* __restrict__ counts,
                                                                  float* __restrict__ D /*[query][unit position x 64]: approximate distances, +inf for rows that are no candidates*/, long ldD,
                                                                  float* __restrict__ Umin /*[query][unit position]: the unit's smallest approximate distance*/, long ldU,
                                                                  const float* __restrict__ su = nullptr, const float* __restrict__ sq = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [buf][X 32 KiB | Q 8 KiB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int n_items = counts[0];
    // XCD-aware static split: workgroup b runs on XCD b % 8; every XCD owns a contiguous range of items, its workgroups take them round-robin
    const int nx = 8, xcd = blockIdx.x % nx, wgx = blockIdx.x / nx, wgs_per_xcd = gridDim.x / nx;
    const int iq = n_items / nx, irem = n_items % nx;
    const int xbase = xcd < irem ? xcd * (iq + 1) : irem * (iq + 1) + (xcd - irem) * iq, xcount = xcd < irem ? iq + 1 : iq;
    const int nk = ldh / IV_K;
    const long unit_bytes = (long)nk * (64 * 128);                // fp16 shadow bytes of one 64-row unit
    const int prow = lane >> 3, pslot = lane & 7, khalf = lane >> 5;
    const float INF = __builtin_inff();

    for (int li = wgx; li < xcount; li += wgs_per_xcd) {
        const IvfItem im = items[xbase + li];
        const IvfGroup* __restrict__ G = groups + im.group;
        const int nq = G->nq, len = G->len, tl = im.tile;
        const int nun = min(4, (len + IV_UNIT - 1) / IV_UNIT - 4 * tl);       // units of this tile that exist
        const long unit_g0 = (long)G->unit0 + 4 * tl;
        const bool has_rows = wm < nun;                                       // wave-uniform: this wave's unit exists (its DMA pieces and its MFMAs)
        const bool has_q = wn * 32 < nq;                                      // wave-uniform: this wave's query block holds a query
        // ---- staging: 4 row pieces per wave (rows of its own unit) + 1 query piece per wave, 8 rows x 128 B each ----
        const char* xsrc[4]; int xdst[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = (wid * 4 + i) * 8 + prow;                           // row in the tile; its unit is wid >> 1 = wm
            const int ks = pslot ^ ((r >> 1) & 7);                            // logical 16-byte slot stored at this physical slot
            xsrc[i] = reinterpret_cast<const char*>(Vh) + (unit_g0 + wm) * unit_bytes + (long)(r & 63) * 128 + ks * 16;
            xdst[i] = (wid * 4 + i) * 8 * 128;
        }
        const int qr = wid * 8 + prow;                                        // query slot staged by this lane
        const bool q_piece = wid * 8 < nq;                                    // wave-uniform: rows at or beyond nq are never read into a kept column
        const int qsl = q_piece ? G->q[min(qr, nq - 1)] : 0;
        const char* qsrc = reinterpret_cast<const char*>(Qh + (long)qsl * ldh) + (pslot ^ ((qr >> 1) & 7)) * 16;
        const int qdst = 32768 + wid * 8 * 128;
        auto stage = [&](int buf, int kt) {
            unsigned char* sb = smem + buf * IV_STAGE;
            if (has_rows) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i] + (long)kt * (64 * 128)),
                                                     (__attribute__((address_space(3))) void*)(sb + xdst[i]), 16, 0, 2);
            }
            if (q_piece)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qsrc + (long)kt * 128),
                                                 (__attribute__((address_space(3))) void*)(sb + qdst), 16, 0, 0);
        };
        std::conditional_t<I8, i32x16, f32x16> acc[2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][e] = 0;
        stage(0, 0);
        __syncthreads();
        const int arow = wm * 64 + (lane & 31), brow = wn * 32 + (lane & 31);
        for (int kt = 0; kt < nk; kt++) {
            const int buf = kt & 1;
            if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
            if (has_rows && has_q) {
                const unsigned char* xb = smem + buf * IV_STAGE;
                const unsigned char* qb = xb + 32768;
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    half8 a[2], b;
#pragma unroll
                    for (int mb = 0; mb < 2; mb++) a[mb] = *reinterpret_cast<const half8*>(xb + iv_swz_off(arow + mb * 32, ks * 2 + khalf));
                    b = *reinterpret_cast<const half8*>(qb + iv_swz_off(brow, ks * 2 + khalf));
#pragma unroll
                    for (int mb = 0; mb < 2; mb++) {
                        if constexpr (I8) acc[mb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, a[mb]), __builtin_bit_cast(i32x4v, b), acc[mb], 0, 0, 0);
                        else acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb], b, acc[mb], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
        // ---- epilogue: the wave's (query, unit) pairs -> 64 approximate distances each, into the query's score row ----
        if (has_rows && has_q) {
            const int s = wn * 32 + (lane & 31);                              // query slot of this lane's column
            const bool live = s < nq;
            const int q = G->q[live ? s : 0];
            const long ukey = (long)G->kb[live ? s : 0] + 4 * tl + wm;        // unit index inside the query's score row
            const long slot0 = (unit_g0 + wm) * IV_UNIT;
            const int nvalid = len - (4 * tl + wm) * IV_UNIT;                 // rows of the unit that are list members (>= 1 here)
            float qnv = 0.0f, scl = 1.0f;
            if constexpr (MODE == 1) qnv = qn[q];
            if constexpr (I8) scl = su[unit_g0 + wm] * sq[q];
            const bool check = nvalid < IV_UNIT || elig != nullptr;           // wave-uniform
            float* __restrict__ drow = D + (long)q * ldD + ukey * IV_UNIT;
            float umn = INF;
#pragma unroll
            for (int mb = 0; mb < 2; mb++) {
#pragma unroll
                for (int e4 = 0; e4 < 4; e4++) {
                    const int r0 = mb * 32 + 8 * e4 + 4 * khalf;              // C layout: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): registers 4 e4 .. 4 e4 + 3 are rows r0 .. r0 + 3
                    f32x4v v;
                    f32x4v rnv = {0.0f, 0.0f, 0.0f, 0.0f};
                    if constexpr (MODE == 1) rnv = *reinterpret_cast<const f32x4v*>(rn + slot0 + r0);
#pragma unroll
                    for (int e1 = 0; e1 < 4; e1++) {
                        float a, sdot;
                        if constexpr (I8) sdot = (float)acc[mb][e4 * 4 + e1] * scl; else sdot = acc[mb][e4 * 4 + e1];
                        if constexpr (MODE == 0) a = 1.0f - sdot;
                        else a = (qnv + rnv[e1]) - 2.0f * sdot;
                        a = fmaxf(a, 0.0f);
                        if (check) { bool ok = r0 + e1 < nvalid; if (ok && elig) ok = elig[slot0 + r0 + e1] != 0; a = ok ? a : INF; }
                        v[e1] = a;
                        umn = fminf(umn, a);
                    }
                    if (live) *reinterpret_cast<f32x4v*>(drow + r0) = v;
                }
            }
            umn = fminf(umn, __shfl_xor(umn, 32, 64));                        // the other half-wave holds the unit's other 32 rows of this query
            if (live && lane < 32) Umin[(long)q * ldU + ukey] = umn;
        }
    }
}
void launch_ivf_scan_f16(Ctx* c, int mode, const void* Vh, int ldh, const void* Qh, const float* rn, const float* qn, const uint8_t* elig,
                         const void* groups, const void* items, const int32_t* counts, float* D, int64_t ldD, float* umin, int64_t ldu) {
    const size_t lds = 2 * IV_STAGE;
    const long grid = (long)round_up(c->prop.multiProcessorCount, 8) * 2;     // persistent: two workgroups per CU
    auto go = [&](auto kernel) {
        HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->launch_timed("ivf_scan_f16", kernel, dim3((unsigned)grid), dim3(IV_THREADS), lds, (const _Float16*)Vh, ldh, (const _Float16*)Qh, rn, qn, (const unsigned char*)elig,
                        (const IvfGroup*)groups, (const IvfItem*)items, (const int*)counts, D, (long)ldD, umin, (long)ldu, (const float*)nullptr, (const float*)nullptr);
    };
    if (mode == 0) go(ivf_scan_f16_kernel<0, false>); else go(ivf_scan_f16_kernel<1, false>);
    LAUNCH_CHECK();
}
void launch_ivf_scan_i8(Ctx* c, int mode, const void* V8, int ld8, const void* Q8R, const float* rn, const float* qn, const float* su, const float* sq, const uint8_t* elig,
                        const void* groups, const void* items, const int32_t* counts, float* D, int64_t ldD, float* umin, int64_t ldu) {
    const size_t lds = 2 * IV_STAGE;
    const long grid = (long)round_up(c->prop.multiProcessorCount, 8) * 2;
    auto go = [&](auto kernel) {
        HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->launch_timed("ivf_scan_i8", kernel, dim3((unsigned)grid), dim3(IV_THREADS), lds, (const _Float16*)V8, ld8 / 2, (const _Float16*)Q8R, rn, qn, (const unsigned char*)elig,
                        (const IvfGroup*)groups, (const IvfItem*)items, (const int*)counts, D, (long)ldD, umin, (long)ldu, su, sq);
    };
    if (mode == 0) go(ivf_scan_f16_kernel<0, true>); else go(ivf_scan_f16_kernel<1, true>);
    LAUNCH_CHECK();
}

}  // namespace comet
