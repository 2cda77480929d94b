// kernels_select.hip — exact, deterministic top-K selection on gfx950.
//
// Replaces the reference's "sort everything, take k" (flat_index_search.go:277-291 and the identical
// tails of the IVF / PQ / IVFPQ / HNSW searches). The reference sorts with an unstable sort keyed on
// distance only, so the order among equal distances is undefined there; here it is canonical:
// (score ascending, scan position ascending) — what a stable sort of the reference's candidate list
// would give, and what the oracle implements.
//
// Algorithm (per query, all queries of a batch in the same launches):
//   three radix-histogram passes over the candidate row (12 + 12 + 8 bits of the order-preserving
//   uint32 image of the float score) find the K-th smallest key key* and r = how many candidates
//   equal to key* belong to the answer; a count pass + an ordered gather pass then emit every
//   candidate with key < key* plus the first r (by position) with key == key*; one workgroup per
//   query finally sorts the K composites (key<<32 | position) in LDS (bitonic) and writes the row.
// All passes are streaming reads of the distance matrix with 16-byte loads, histogram in LDS.
#include "kernels.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace comet {

constexpr int SEL_THREADS = 256;
constexpr int SEL_EPT = 16;                         // elements per thread per chunk (4 x float4)
constexpr int SEL_CHUNK = SEL_THREADS * SEL_EPT;    // 4096 candidates per workgroup
constexpr int SEL_BINS = 4096;
constexpr int SORT_MAX = 4096;                      // max K sorted in LDS

int select_max_k() { return 0x7FFFFFFF; }   // no limit: selections of more than SORT_MAX results sort in global memory

struct SelState {
    unsigned prefix;     // bits of key* decided so far
    unsigned mask;       // which bits are decided
    int remaining;       // rank of key* among candidates matching the prefix (1-based count still to take)
    int kq;              // number of results for this query = min(K, #eligible)
    int total;           // #eligible candidates
    int less_cursor;     // atomic cursor for "key < key*" outputs
    int pad0, pad1;
};

// order-preserving map float -> uint32 (ascending floats => ascending keys; -0 < +0)
__device__ __forceinline__ unsigned f2key(unsigned u) { return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ unsigned key2f(unsigned k) { return (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; }

// candidate participates iff not the EXCLUDED sentinel and not beyond the threshold
__device__ __forceinline__ bool cand_ok(unsigned bits, float thr) {
    if (bits == EXCLUDED_BITS) return false;
    if (thr > 0.0f && __uint_as_float(bits) > thr) return false;   // `s.threshold > 0 && dist > s.threshold`
    return true;
}

// load SEL_EPT consecutive candidates of one thread: positions base + t*16 .. +15 (clipped to cnt)
__device__ __forceinline__ void load16(const float* __restrict__ row, long base, long cnt, unsigned (&v)[SEL_EPT], bool (&in)[SEL_EPT]) {
    const long p0 = base + (long)threadIdx.x * SEL_EPT;
    if (p0 + SEL_EPT <= cnt && ((reinterpret_cast<uintptr_t>(row + p0) & 15) == 0)) {
#pragma unroll
        for (int j = 0; j < SEL_EPT / 4; j++) {
            f32x4 x = *reinterpret_cast<const f32x4*>(row + p0 + j * 4);
            v[j * 4 + 0] = __float_as_uint(x[0]); v[j * 4 + 1] = __float_as_uint(x[1]);
            v[j * 4 + 2] = __float_as_uint(x[2]); v[j * 4 + 3] = __float_as_uint(x[3]);
        }
#pragma unroll
        for (int j = 0; j < SEL_EPT; j++) in[j] = true;
    } else {
#pragma unroll
        for (int j = 0; j < SEL_EPT; j++) {
            in[j] = (p0 + j) < cnt;
            v[j] = in[j] ? __float_as_uint(row[p0 + j]) : EXCLUDED_BITS;
        }
    }
}

// ---- pass: histogram of `bits` bits at `shift` among candidates matching the decided prefix --------
__global__ __launch_bounds__(SEL_THREADS) void sel_hist_kernel(const float* __restrict__ D, long ldD, long C,
                                                               const int* __restrict__ cnts, float thr,
                                                               const SelState* __restrict__ st, int shift, int bits,
                                                               unsigned* __restrict__ hist /*[B][SEL_BINS]*/, const int* __restrict__ rowlist) {
    __shared__ unsigned lh[SEL_BINS];
    if (rowlist && (int)blockIdx.y >= rowlist[0]) return;          // rowlist = {n, row_0, row_1, ...}: only the rows that need the generic path
    const int q = rowlist ? rowlist[1 + blockIdx.y] : (int)blockIdx.y;
    const long cnt = cnts ? (long)cnts[q] : C;
    const long base = (long)blockIdx.x * SEL_CHUNK;
    if (base >= cnt) return;
    const int nb = 1 << bits;
    for (int i = threadIdx.x; i < nb; i += SEL_THREADS) lh[i] = 0;
    __syncthreads();
    const unsigned prefix = st[q].prefix, mask = st[q].mask;
    unsigned v[SEL_EPT]; bool in[SEL_EPT];
    load16(D + (long)q * ldD, base, cnt, v, in);
#pragma unroll
    for (int j = 0; j < SEL_EPT; j++) {
        if (in[j] && cand_ok(v[j], thr)) {
            unsigned k = f2key(v[j]);
            if ((k & mask) == prefix) atomicAdd(&lh[(k >> shift) & (nb - 1)], 1u);
        }
    }
    __syncthreads();
    unsigned* gh = hist + (long)q * SEL_BINS;
    for (int i = threadIdx.x; i < nb; i += SEL_THREADS) { unsigned c = lh[i]; if (c) atomicAdd(&gh[i], c); }
}

// ---- scan: one workgroup per query; pick the bin holding the `remaining`-th candidate ---------------
__global__ __launch_bounds__(SEL_THREADS) void sel_scan_kernel(unsigned* __restrict__ hist, SelState* __restrict__ st,
                                                               int shift, int bits, int K, int first, const int* __restrict__ rowlist) {
    __shared__ unsigned part[SEL_THREADS];
    __shared__ int s_bin; __shared__ unsigned s_before; __shared__ int s_remaining;
    if (rowlist && (int)blockIdx.x >= rowlist[0]) return;
    const int q = rowlist ? rowlist[1 + blockIdx.x] : (int)blockIdx.x;
    const int nb = 1 << bits;
    unsigned* gh = hist + (long)q * SEL_BINS;
    const int per = (nb + SEL_THREADS - 1) / SEL_THREADS;   // bins per thread (contiguous)
    const int b0 = threadIdx.x * per;
    unsigned mine = 0;
    for (int i = 0; i < per; i++) if (b0 + i < nb) mine += gh[b0 + i];
    part[threadIdx.x] = mine;
    __syncthreads();
    // serial exclusive scan by one thread (256 partials) — negligible
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int i = 0; i < SEL_THREADS; i++) { unsigned x = part[i]; part[i] = run; run += x; }
        SelState s = st[q];
        if (first) {
            s.total = (int)run;
            s.kq = (K <= 0 || (unsigned)K > run) ? (int)run : K;   // sanitizeK limiter.go:12-17
            s.remaining = s.kq; s.prefix = 0; s.mask = 0; s.less_cursor = 0;
        }
        st[q] = s;
        s_bin = -1; s_before = 0; s_remaining = s.remaining;
    }
    __syncthreads();
    const int remaining = s_remaining;
    if (remaining > 0) {
        unsigned run = part[threadIdx.x];
        for (int i = 0; i < per; i++) {
            if (b0 + i < nb) {
                unsigned c = gh[b0 + i];
                if (run < (unsigned)remaining && run + c >= (unsigned)remaining) { s_bin = b0 + i; s_before = run; }
                run += c;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && remaining > 0) {
        SelState s = st[q];
        s.prefix |= ((unsigned)s_bin) << shift;
        s.mask |= ((unsigned)(nb - 1)) << shift;
        s.remaining = remaining - (int)s_before;
        st[q] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += SEL_THREADS) gh[i] = 0;   // ready for the next pass
}

// ---- count of candidates equal to key* per chunk ---------------------------------------------------
__global__ __launch_bounds__(SEL_THREADS) void sel_count_eq_kernel(const float* __restrict__ D, long ldD, long C,
                                                                   const int* __restrict__ cnts, float thr,
                                                                   const SelState* __restrict__ st,
                                                                   int* __restrict__ eqcnt, int nchunks, const int* __restrict__ rowlist) {
    __shared__ int s_cnt;
    if (rowlist && (int)blockIdx.y >= rowlist[0]) return;
    const int q = rowlist ? rowlist[1 + blockIdx.y] : (int)blockIdx.y;
    const long cnt = cnts ? (long)cnts[q] : C;
    const long base = (long)blockIdx.x * SEL_CHUNK;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    if (base < cnt && st[q].kq > 0) {
        const unsigned keystar = st[q].prefix;
        unsigned v[SEL_EPT]; bool in[SEL_EPT];
        load16(D + (long)q * ldD, base, cnt, v, in);
        int mine = 0;
#pragma unroll
        for (int j = 0; j < SEL_EPT; j++) if (in[j] && cand_ok(v[j], thr) && f2key(v[j]) == keystar) mine++;
        if (mine) atomicAdd(&s_cnt, mine);
    }
    __syncthreads();
    if (threadIdx.x == 0) eqcnt[(long)q * nchunks + blockIdx.x] = s_cnt;
}

// exclusive prefix over chunks, one workgroup per query (in place)
__global__ __launch_bounds__(SEL_THREADS) void sel_scan_eq_kernel(int* __restrict__ eqcnt, int nchunks, const int* __restrict__ rowlist) {
    __shared__ int part[SEL_THREADS];
    if (rowlist && (int)blockIdx.x >= rowlist[0]) return;
    int* e = eqcnt + (long)(rowlist ? rowlist[1 + blockIdx.x] : (int)blockIdx.x) * nchunks;
    const int per = (nchunks + SEL_THREADS - 1) / SEL_THREADS;
    const int b0 = threadIdx.x * per;
    int mine = 0;
    for (int i = 0; i < per; i++) if (b0 + i < nchunks) mine += e[b0 + i];
    part[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < SEL_THREADS; i++) { int x = part[i]; part[i] = run; run += x; } }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = 0; i < per; i++) if (b0 + i < nchunks) { int x = e[b0 + i]; e[b0 + i] = run; run += x; }
}

// ---- gather: emit composites (key<<32 | pos) of the selected candidates ---------------------------
__global__ __launch_bounds__(SEL_THREADS) void sel_gather_kernel(const float* __restrict__ D, long ldD, long C,
                                                                 const int* __restrict__ cnts, float thr,
                                                                 SelState* __restrict__ st, const int* __restrict__ eqpre,
                                                                 int nchunks, unsigned long long* __restrict__ comp, int comp_ld,
                                                                 const int* __restrict__ rowlist) {
    __shared__ int wsum[SEL_THREADS / 64];
    if (rowlist && (int)blockIdx.y >= rowlist[0]) return;
    const int q = rowlist ? rowlist[1 + blockIdx.y] : (int)blockIdx.y;
    const long cnt = cnts ? (long)cnts[q] : C;
    const long base = (long)blockIdx.x * SEL_CHUNK;
    if (base >= cnt) return;
    const SelState s = st[q];
    if (s.kq <= 0) return;
    const unsigned keystar = s.prefix;
    const int r = s.remaining;            // how many key*-equal candidates to take (in position order)
    const int n_less = s.kq - r;          // all candidates with key < key*
    unsigned v[SEL_EPT]; bool in[SEL_EPT];
    load16(D + (long)q * ldD, base, cnt, v, in);
    unsigned long long* out = comp + (long)q * comp_ld;
    const long p0 = base + (long)threadIdx.x * SEL_EPT;
    // pass 1: "less" candidates -> any free slot in [0, n_less) (order fixed later by the sort)
    int my_eq = 0;
#pragma unroll
    for (int j = 0; j < SEL_EPT; j++) {
        if (in[j] && cand_ok(v[j], thr)) {
            unsigned k = f2key(v[j]);
            if (k < keystar) {
                int slot = atomicAdd(&st[q].less_cursor, 1);
                out[slot] = ((unsigned long long)k << 32) | (unsigned)(p0 + j);
            } else if (k == keystar) my_eq++;
        }
    }
    // pass 2: ordered rank among equals: chunk prefix (eqpre) + threads before me + my own earlier ones
    // wave-level inclusive scan via DPP-free shuffles, then cross-wave via LDS
    int incl = my_eq;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { int y = __shfl_up(incl, off, 64); if (lane >= off) incl += y; }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int before = eqpre[(long)q * nchunks + blockIdx.x];
    for (int w = 0; w < wid; w++) before += wsum[w];
    int rank = before + incl - my_eq;
    if (my_eq && rank < r) {
#pragma unroll
        for (int j = 0; j < SEL_EPT; j++) {
            if (in[j] && cand_ok(v[j], thr) && f2key(v[j]) == keystar) {
                if (rank < r) out[n_less + rank] = ((unsigned long long)keystar << 32) | (unsigned)(p0 + j);
                rank++;
            }
        }
    }
}

// ---- final: sort K composites per query in LDS, write positions / scores / counts ------------------
__global__ __launch_bounds__(1024) void sel_sort_kernel(const unsigned long long* __restrict__ comp, int comp_ld,
                                                        const SelState* __restrict__ st, unsigned* __restrict__ out_pos,
                                                        float* __restrict__ out_scores, int* __restrict__ out_counts, int k_cap,
                                                        const int* __restrict__ rowlist) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sm[];
    if (rowlist && (int)blockIdx.x >= rowlist[0]) return;
    const int q = rowlist ? rowlist[1 + blockIdx.x] : (int)blockIdx.x;
    const int kq = st[q].kq;
    int n2 = 1; while (n2 < kq) n2 <<= 1;
    const unsigned long long* in = comp + (long)q * comp_ld;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) sm[i] = i < kq ? in[i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                int ixj = i ^ j;
                if (ixj > i) {
                    unsigned long long a = sm[i], b = sm[ixj];
                    bool up = ((i & k) == 0);
                    if ((a > b) == up) { sm[i] = b; sm[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    const int nw = kq < k_cap ? kq : k_cap;
    for (int i = threadIdx.x; i < k_cap; i += blockDim.x) {
        if (i < nw) {
            unsigned long long c = sm[i];
            out_pos[(long)q * k_cap + i] = (unsigned)(c & 0xFFFFFFFFull);
            out_scores[(long)q * k_cap + i] = __uint_as_float(key2f((unsigned)(c >> 32)));
        } else {
            out_pos[(long)q * k_cap + i] = 0xFFFFFFFFu;
            out_scores[(long)q * k_cap + i] = 0.0f;
        }
    }
    if (threadIdx.x == 0) out_counts[q] = kq;
}

// ---- rows of 64-bit composites sorted in global memory: selections larger than the LDS sort --------------
// Bitonic network over B rows of ld (a power of two) composites: 4096-element tiles take every step with j < 4096 in LDS,
// the wider steps are one launch each. Nothing here is on a hot path (k > 4096 on rows longer than 8192 candidates).
constexpr int GS_TILE = 4096;
template <bool INIT>
__global__ __launch_bounds__(1024) void bitonic_tile_kernel(unsigned long long* __restrict__ comp, long ld, int k_fixed) {
    __shared__ unsigned long long sm[GS_TILE];
    unsigned long long* row = comp + (long)blockIdx.y * ld;
    const long base = (long)blockIdx.x * GS_TILE;
    const int n = (int)(ld < GS_TILE ? ld : GS_TILE);
    for (int i = threadIdx.x; i < n; i += 1024) sm[i] = row[base + i];
    __syncthreads();
    auto step = [&](long k, int j) {
        for (int i = threadIdx.x; i < n; i += 1024) {
            const int ixj = i ^ j;
            if (ixj > i) {
                const unsigned long long a = sm[i], b = sm[ixj];
                const bool up = (((base + i) & k) == 0);
                if ((a > b) == up) { sm[i] = b; sm[ixj] = a; }
            }
        }
        __syncthreads();
    };
    if (INIT) { for (int k = 2; k <= n; k <<= 1) for (int j = k >> 1; j > 0; j >>= 1) step(k, j); }
    else { for (int j = GS_TILE >> 1; j > 0; j >>= 1) step((long)k_fixed, j); }
    for (int i = threadIdx.x; i < n; i += 1024) row[base + i] = sm[i];
}
__global__ __launch_bounds__(256) void bitonic_global_kernel(unsigned long long* __restrict__ comp, long ld, long k, long j) {
    unsigned long long* row = comp + (long)blockIdx.y * ld;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;           // one compare-exchange per thread
    if (t >= ld / 2) return;
    const long i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), ixj = i | j;
    const unsigned long long a = row[i], b = row[ixj];
    const bool up = ((i & k) == 0);
    if ((a > b) == up) { row[i] = b; row[ixj] = a; }
}
void sort_rows_u64(Ctx* c, unsigned long long* comp, int64_t ld, int B) {
    ProfScope ps(c, "select_sort_global");
    const unsigned tiles = (unsigned)std::max<int64_t>(1, ld / GS_TILE);
    bitonic_tile_kernel<true><<<dim3(tiles, B), dim3(1024), 0, c->stream>>>(comp, ld, 0);
    for (int64_t k = 2 * GS_TILE; k <= ld; k <<= 1) {
        for (int64_t j = k >> 1; j >= GS_TILE; j >>= 1)
            bitonic_global_kernel<<<dim3((unsigned)ceil_div(ld / 2, 256), B), dim3(256), 0, c->stream>>>(comp, ld, k, j);
        bitonic_tile_kernel<false><<<dim3(tiles, B), dim3(1024), 0, c->stream>>>(comp, ld, (int)k);
    }
    LAUNCH_CHECK();
}
// comp rows hold kq composites each: everything behind them sorts last
__global__ __launch_bounds__(256) void sel_pad_kernel(unsigned long long* __restrict__ comp, long ld, const SelState* __restrict__ st) {
    const int q = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < ld && i >= st[q].kq) comp[(long)q * ld + i] = ~0ull;
}
__global__ __launch_bounds__(256) void sel_emit_kernel(const unsigned long long* __restrict__ comp, long ld, const SelState* __restrict__ st,
                                                       unsigned* __restrict__ out_pos, float* __restrict__ out_scores, int* __restrict__ out_counts, int k_cap) {
    const int q = blockIdx.y, kq = st[q].kq;
    const int nw = kq < k_cap ? kq : k_cap;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < k_cap) {
        if (i < nw) {
            const unsigned long long cc = comp[(long)q * ld + i];
            out_pos[(long)q * k_cap + i] = (unsigned)(cc & 0xFFFFFFFFull);
            out_scores[(long)q * k_cap + i] = __uint_as_float(key2f((unsigned)(cc >> 32)));
        } else { out_pos[(long)q * k_cap + i] = 0xFFFFFFFFu; out_scores[(long)q * k_cap + i] = 0.0f; }
    }
    if (i == 0) out_counts[q] = kq;
}

// ---- small candidate rows (C <= 8192): one workgroup per query, everything in LDS ---------------------
// Same result as the multi-pass path (composite = order-preserving key << 32 | position, ascending), one
// launch instead of nine: used for the coarse top-nprobe, short IVF / IVFPQ candidate lists and HNSW result lists.
//   * up to 1024 candidates: bitonic sort of the composites;
//   * more: two 12-bit radix histogram passes in LDS locate the 24-bit prefix of the K-th key, the
//     candidates at or below that prefix are compacted (typically K + a handful) and only those are sorted;
//     if the compaction would exceed its buffer (heavy ties) the whole row is sorted instead.
constexpr int SMALL_MAX = 8192;
constexpr int SMALL_LIST = 2048;

__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* sm, int n2) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                int ixj = i ^ j;
                if (ixj > i) {
                    unsigned long long a = sm[i], b = sm[ixj];
                    bool up = ((i & k) == 0);
                    if ((a > b) == up) { sm[i] = b; sm[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}
// block-wide: find the bin of a 4096-bin LDS histogram holding the `rank`-th element (1-based);
// returns bin in *bin_out and the number of elements in lower bins in *before_out. blockDim.x == 1024.
__device__ __forceinline__ void find_bin_4096(const unsigned* hist, int rank, unsigned* wsum /*16*/, int* bin_out, int* before_out) {
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

__global__ __launch_bounds__(1024) void select_small_kernel(const float* __restrict__ D, long ldD, long C, const int* __restrict__ cnts, float thr,
                                                            int K, unsigned* __restrict__ out_pos, float* __restrict__ out_scores,
                                                            int* __restrict__ out_counts, int k_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sm[];   // 64 KiB when C > 1024, else 8 * n2 bytes
    __shared__ int s_valid, s_bin, s_before, s_m;
    __shared__ unsigned wsum[16];
    const int q = blockIdx.x;
    long cnt = cnts ? (long)cnts[q] : C;
    if (cnt > C) cnt = C;
    if (threadIdx.x == 0) { s_valid = 0; s_m = 0; }
    __syncthreads();
    const float* row = D + (long)q * ldD;
    unsigned long long* sorted = sm;
    int kq = 0;
    if (cnt <= 1024 || blockDim.x < 1024) {
        // ---- direct sort ----
        int n2 = 64; while (n2 < cnt) n2 <<= 1;
        int mine = 0;
        for (int i = threadIdx.x; i < n2; i += blockDim.x) {
            unsigned long long v = ~0ull;
            if (i < cnt) { const unsigned bits = __float_as_uint(row[i]); if (cand_ok(bits, thr)) { v = ((unsigned long long)f2key(bits) << 32) | (unsigned)i; mine++; } }
            sm[i] = v;
        }
        if (mine) atomicAdd(&s_valid, mine);
        __syncthreads();
        bitonic_sort_lds(sm, n2);
        const int valid = s_valid;
        kq = (K <= 0 || K > valid) ? valid : K;    // sanitizeK limiter.go:12-17
    } else {
        // ---- radix-compaction ----
        unsigned* keys = reinterpret_cast<unsigned*>(sm);            // 8192 keys   (32 KiB)
        unsigned* hist = keys + SMALL_MAX;                            // 4096 bins   (16 KiB)
        unsigned long long* list = sm + (SMALL_MAX + 4096) / 2;       // 2048 composites (16 KiB)
        int mine = 0;
        for (int i = threadIdx.x; i < SMALL_MAX; i += 1024) {
            unsigned k = 0xFFFFFFFFu;
            if (i < cnt) { const unsigned bits = __float_as_uint(row[i]); if (cand_ok(bits, thr)) { k = f2key(bits); mine++; } }
            keys[i] = k;   // note: a real key of 0xFFFFFFFF (a negative NaN) is indistinguishable from "invalid"; dropped
        }
        for (int i = threadIdx.x; i < 4096; i += 1024) hist[i] = 0;
        if (mine) atomicAdd(&s_valid, mine);
        __syncthreads();
        const int valid = s_valid;
        kq = (K <= 0 || K > valid) ? valid : K;
        bool full_sort = kq > SMALL_LIST;
        unsigned prefix24 = 0;
        if (!full_sort && kq > 0) {
            for (int i = threadIdx.x; i < cnt; i += 1024) { const unsigned k = keys[i]; if (k != 0xFFFFFFFFu) atomicAdd(&hist[k >> 20], 1u); }
            __syncthreads();
            find_bin_4096(hist, kq, wsum, &s_bin, &s_before);
            const int bin1 = s_bin, before1 = s_before;
            __syncthreads();
            for (int i = threadIdx.x; i < 4096; i += 1024) hist[i] = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < cnt; i += 1024) { const unsigned k = keys[i]; if (k != 0xFFFFFFFFu && (int)(k >> 20) == bin1) atomicAdd(&hist[(k >> 8) & 4095u], 1u); }
            __syncthreads();
            find_bin_4096(hist, kq - before1, wsum, &s_bin, &s_before);
            prefix24 = ((unsigned)bin1 << 12) | (unsigned)s_bin;
            __syncthreads();
            // compact every candidate whose 24-bit prefix is <= the K-th key's prefix
            for (int i = threadIdx.x; i < cnt; i += 1024) {
                const unsigned k = keys[i];
                if (k != 0xFFFFFFFFu && (k >> 8) <= prefix24) { int s = atomicAdd(&s_m, 1); if (s < SMALL_LIST) list[s] = ((unsigned long long)k << 32) | (unsigned)i; }
            }
            __syncthreads();
            if (s_m > SMALL_LIST) full_sort = true;
        }
        if (full_sort) {
            __syncthreads();
            int n2 = 64; while (n2 < cnt) n2 <<= 1;
            for (int i = threadIdx.x; i < n2; i += 1024) {
                unsigned long long v = ~0ull;
                if (i < cnt) { const unsigned bits = __float_as_uint(row[i]); if (cand_ok(bits, thr)) v = ((unsigned long long)f2key(bits) << 32) | (unsigned)i; }
                sm[i] = v;
            }
            __syncthreads();
            bitonic_sort_lds(sm, n2);
        } else if (kq > 0) {
            const int m = s_m;
            int n2 = 64; while (n2 < m) n2 <<= 1;
            for (int i = m + threadIdx.x; i < n2; i += 1024) list[i] = ~0ull;
            __syncthreads();
            bitonic_sort_lds(list, n2);
            sorted = list;
        }
    }
    const int nw = kq < k_cap ? kq : k_cap;
    for (int i = threadIdx.x; i < k_cap; i += blockDim.x) {
        if (i < nw) {
            unsigned long long c = sorted[i];
            out_pos[(long)q * k_cap + i] = (unsigned)(c & 0xFFFFFFFFull);
            out_scores[(long)q * k_cap + i] = __uint_as_float(key2f((unsigned)(c >> 32)));
        } else {
            out_pos[(long)q * k_cap + i] = 0xFFFFFFFFu;
            out_scores[(long)q * k_cap + i] = 0.0f;
        }
    }
    if (threadIdx.x == 0) out_counts[q] = kq;
}

// ---- long rows, small K: sample bound + one filtering pass --------------------------------------------
// The three histogram passes above re-read the whole distance matrix and serialise on LDS atomics (the top key bits of
// a row of distances are nearly all equal). For K much smaller than the row, a bound does the same job in ONE pass:
//   sel_bound   per row: the 24-bit key prefix of the K-th smallest valid candidate among the first S candidates of
//               the row (any subset gives an upper bound on the row's K-th smallest; for IVF-type rows the prefix is
//               the nearest lists, i.e. a tight one),
//   sel_filter  streams the matrix once and keeps the composites (key<<32 | position) of candidates at or below the bound,
//   sel_final   sorts the kept composites (bitonic, LDS) and writes the row; a row that kept more than BOUND_CAP
//               (adversarial order, heavy ties) is appended to the row list of the generic path, which then runs for
//               those rows only (its workgroups exit at once when the list is empty).
// Exact for the same reason the merge is: the kept set contains every candidate <= the K-th smallest.
constexpr int BOUND_CAP = 4096;
constexpr int BOUND_SAMPLE_MAX = 65536;
__global__ __launch_bounds__(1024) void sel_bound_kernel(const float* __restrict__ D, long ldD, long C, const int* __restrict__ cnts, float thr,
                                                         int K, int S, unsigned* __restrict__ bound, int* __restrict__ cursor) {
    __shared__ unsigned hist[4096];
    __shared__ unsigned wsum[16];
    __shared__ int s_valid, s_bin, s_before;
    const int q = blockIdx.x;
    long cnt = cnts ? (long)cnts[q] : C;
    if (cnt > C) cnt = C;
    const int n = (int)(cnt < S ? cnt : S);
    const float* row = D + (long)q * ldD;
    for (int i = threadIdx.x; i < 4096; i += 1024) hist[i] = 0;
    if (threadIdx.x == 0) { s_valid = 0; cursor[q] = 0; }
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const unsigned bits = __float_as_uint(row[i]);
        if (cand_ok(bits, thr)) { atomicAdd(&hist[f2key(bits) >> 20], 1u); mine++; }
    }
    if (mine) atomicAdd(&s_valid, mine);
    __syncthreads();
    if (s_valid < K) { if (threadIdx.x == 0) bound[q] = 0xFFFFFFFFu; return; }   // sample too thin: keep everything valid
    find_bin_4096(hist, K, wsum, &s_bin, &s_before);
    const int bin1 = s_bin, before1 = s_before;
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const unsigned bits = __float_as_uint(row[i]);
        if (cand_ok(bits, thr)) { const unsigned k = f2key(bits); if ((int)(k >> 20) == bin1) atomicAdd(&hist[(k >> 8) & 4095u], 1u); }
    }
    __syncthreads();
    find_bin_4096(hist, K - before1, wsum, &s_bin, &s_before);
    if (threadIdx.x == 0) bound[q] = ((((unsigned)bin1 << 12) | (unsigned)s_bin) << 8) | 0xFFu;
}
__global__ __launch_bounds__(SEL_THREADS) void sel_filter_kernel(const float* __restrict__ D, long ldD, long C, const int* __restrict__ cnts, float thr,
                                                                 const unsigned* __restrict__ bound, int* __restrict__ cursor,
                                                                 unsigned long long* __restrict__ comp) {
    const int q = blockIdx.y;
    long cnt = cnts ? (long)cnts[q] : C;
    if (cnt > C) cnt = C;
    const long base = (long)blockIdx.x * SEL_CHUNK;
    if (base >= cnt) return;
    const unsigned bnd = bound[q];
    unsigned v[SEL_EPT]; bool in[SEL_EPT];
    load16(D + (long)q * ldD, base, cnt, v, in);
    const long p0 = base + (long)threadIdx.x * SEL_EPT;
    unsigned keep = 0;                               // bit j: candidate j of this thread is kept
#pragma unroll
    for (int j = 0; j < SEL_EPT; j++)
        if (in[j] && cand_ok(v[j], thr) && f2key(v[j]) <= bnd) keep |= 1u << j;
    // one global atomic per workgroup (same-address atomics of a row serialise in L2): workgroup-wide exclusive scan
    __shared__ int wsum[SEL_THREADS / 64];
    __shared__ int s_base;
    const int mine = __builtin_popcount(keep);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { int y = __shfl_up(incl, off, 64); if (lane >= off) incl += y; }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int before = incl - mine, total = 0;
#pragma unroll
    for (int w = 0; w < SEL_THREADS / 64; w++) { if (w < wid) before += wsum[w]; total += wsum[w]; }
    if (total == 0) return;                          // workgroup-uniform
    if (threadIdx.x == 0) s_base = atomicAdd(&cursor[q], total);
    __syncthreads();
    int slot = s_base + before;
#pragma unroll
    for (int j = 0; j < SEL_EPT; j++) {
        if (keep & (1u << j)) {
            if (slot < BOUND_CAP) comp[(long)q * BOUND_CAP + slot] = ((unsigned long long)f2key(v[j]) << 32) | (unsigned)(p0 + j);
            slot++;
        }
    }
}
__global__ __launch_bounds__(1024) void sel_final_kernel(const unsigned long long* __restrict__ comp, const int* __restrict__ cursor, int K,
                                                         unsigned* __restrict__ out_pos, float* __restrict__ out_scores, int* __restrict__ out_counts,
                                                         int k_cap, int* __restrict__ rowlist) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sm[];
    const int q = blockIdx.x;
    const int m = cursor[q];
    if (m > BOUND_CAP) {                             // generic path takes this row
        if (threadIdx.x == 0) { const int s = atomicAdd(&rowlist[0], 1); rowlist[1 + s] = q; }
        return;
    }
    int n2 = 64; while (n2 < m) n2 <<= 1;
    for (int i = threadIdx.x; i < n2; i += 1024) sm[i] = i < m ? comp[(long)q * BOUND_CAP + i] : ~0ull;
    __syncthreads();
    bitonic_sort_lds(sm, n2);
    const int kq = K > m ? m : K;                    // m < K only when the whole row holds fewer than K valid candidates
    const int nw = kq < k_cap ? kq : k_cap;
    for (int i = threadIdx.x; i < k_cap; i += 1024) {
        if (i < nw) {
            const unsigned long long c = sm[i];
            out_pos[(long)q * k_cap + i] = (unsigned)(c & 0xFFFFFFFFull);
            out_scores[(long)q * k_cap + i] = __uint_as_float(key2f((unsigned)(c >> 32)));
        } else {
            out_pos[(long)q * k_cap + i] = 0xFFFFFFFFu;
            out_scores[(long)q * k_cap + i] = 0.0f;
        }
    }
    if (threadIdx.x == 0) out_counts[q] = kq;
}

// Rows the bound could not handle: ONE workgroup per such row does the whole radix selection (three histogram passes,
// ordered gather, sort) so that the common case — no such row — costs a single near-empty launch.
__global__ __launch_bounds__(1024) void sel_row_kernel(const float* __restrict__ D, long ldD, long C, const int* __restrict__ cnts, float thr, int K,
                                                       const int* __restrict__ rowlist, unsigned* __restrict__ out_pos, float* __restrict__ out_scores,
                                                       int* __restrict__ out_counts, int k_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sm[];   // SORT_MAX composites
    __shared__ unsigned hist[4096];
    __shared__ unsigned wsum[16];
    __shared__ int s_bin, s_before, s_less, s_eqbase;
    if ((int)blockIdx.x >= rowlist[0]) return;
    const int q = rowlist[1 + blockIdx.x];
    long cnt = cnts ? (long)cnts[q] : C;
    if (cnt > C) cnt = C;
    const float* row = D + (long)q * ldD;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    unsigned prefix = 0, mask = 0;
    int remaining = 0, kq = 0;
    const int shifts[3] = {20, 8, 0}, bitsv[3] = {12, 12, 8};
    for (int p = 0; p < 3; p++) {
        const int shift = shifts[p], nb = 1 << bitsv[p];
        for (int i = t; i < 4096; i += 1024) hist[i] = 0;
        __syncthreads();
        for (long i0 = 0; i0 < cnt; i0 += 1024) {
            const long i = i0 + t;
            bool ok = false; unsigned bin = 0;
            if (i < cnt) { const unsigned bits = __float_as_uint(row[i]); if (cand_ok(bits, thr)) { const unsigned k = f2key(bits); if ((k & mask) == prefix) { ok = true; bin = (k >> shift) & (nb - 1); } } }
            // the top bits of a row of distances are nearly all equal: one atomic for the lanes that agree with the first one
            const unsigned long long act = __ballot(ok);
            if (act) {
                const int first = __builtin_ctzll(act);
                const unsigned bin0 = __shfl(bin, first, 64);
                const unsigned long long same = __ballot(ok && bin == bin0);
                if (lane == first) atomicAdd(&hist[bin0], (unsigned)__builtin_popcountll(same));
                if (ok && bin != bin0) atomicAdd(&hist[bin], 1u);
            }
        }
        __syncthreads();
        if (p == 0) {
            unsigned mine = 0;
            for (int i = t; i < 4096; i += 1024) mine += hist[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
            if (lane == 0) wsum[wid] = mine;
            __syncthreads();
            unsigned total = 0;
            for (int w = 0; w < 16; w++) total += wsum[w];
            kq = (K <= 0 || (unsigned)K > total) ? (int)total : K;     // sanitizeK limiter.go:12-17
            remaining = kq;
            __syncthreads();
        }
        if (kq <= 0) break;
        find_bin_4096(hist, remaining, wsum, &s_bin, &s_before);
        prefix |= ((unsigned)s_bin) << shift;
        mask |= ((unsigned)(nb - 1)) << shift;
        remaining -= s_before;
        __syncthreads();
    }
    if (kq > 0) {
        const unsigned keystar = prefix;
        const int r = remaining, n_less = kq - r;    // every candidate below key*, then the first r (by position) equal to it
        if (t == 0) { s_less = 0; s_eqbase = 0; }
        __syncthreads();
        for (long i0 = 0; i0 < cnt; i0 += 1024) {
            const long i = i0 + t;
            bool eq = false;
            if (i < cnt) {
                const unsigned bits = __float_as_uint(row[i]);
                if (cand_ok(bits, thr)) {
                    const unsigned k = f2key(bits);
                    if (k < keystar) { const int s = atomicAdd(&s_less, 1); sm[s] = ((unsigned long long)k << 32) | (unsigned)i; }
                    else eq = (k == keystar);
                }
            }
            const int eqbase = s_eqbase;
            if (eqbase < r) {                        // workgroup-uniform: still taking equals, in position order
                const unsigned long long m = __ballot(eq);
                if (lane == 0) wsum[wid] = (unsigned)__builtin_popcountll(m);
                __syncthreads();
                int before = eqbase;
                for (int w = 0; w < wid; w++) before += (int)wsum[w];
                const int rank = before + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                if (eq && rank < r) sm[n_less + rank] = ((unsigned long long)keystar << 32) | (unsigned)i;
                __syncthreads();
                if (t == 0) { int tot = 0; for (int w = 0; w < 16; w++) tot += (int)wsum[w]; s_eqbase = eqbase + tot; }
                __syncthreads();
            }
        }
        __syncthreads();
        int n2 = 64; while (n2 < kq) n2 <<= 1;
        for (int i = kq + t; i < n2; i += 1024) sm[i] = ~0ull;
        __syncthreads();
        bitonic_sort_lds(sm, n2);
    }
    const int nw = kq < k_cap ? kq : k_cap;
    for (int i = t; i < k_cap; i += 1024) {
        if (i < nw) {
            const unsigned long long c = sm[i];
            out_pos[(long)q * k_cap + i] = (unsigned)(c & 0xFFFFFFFFull);
            out_scores[(long)q * k_cap + i] = __uint_as_float(key2f((unsigned)(c >> 32)));
        } else {
            out_pos[(long)q * k_cap + i] = 0xFFFFFFFFu;
            out_scores[(long)q * k_cap + i] = 0.0f;
        }
    }
    if (t == 0) out_counts[q] = kq;
}

// ---- top-K of rows of COMPOSITES (key << 32 | position) that a producer kernel already filtered ---------------------
// (the ADC scan's fused filter: survivors of a per-query running bound, m = cursor[q] of them in comp[q*ld ..]). One workgroup
// per row: up to SORT_MAX survivors are sorted in LDS; more than that (a bound that stayed loose: descending data, mass ties)
// first finds the K-th smallest composite by an 8-pass byte radix select over the row in HBM, then sorts the K it keeps.
__global__ __launch_bounds__(1024) void sel_composites_kernel(const unsigned long long* __restrict__ comp, long ld, const int* __restrict__ cursor, int K,
                                                              unsigned* __restrict__ out_pos, float* __restrict__ out_scores, int* __restrict__ out_counts, int k_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sm[];   // SORT_MAX composites
    __shared__ unsigned hist[256];
    __shared__ int s_bin, s_before, s_n;
    const int q = blockIdx.x, t = threadIdx.x;
    const unsigned long long* row = comp + (long)q * ld;
    const int m = cursor[q];
    const int kq = (K <= 0 || K > m) ? m : K;        // fewer survivors than K only when the whole row holds fewer than K valid candidates
    int n = m;                                       // composites to sort
    if (m > SORT_MAX) {
        unsigned long long prefix = 0, mask = 0; int remaining = kq;
        for (int shift = 56; shift >= 0; shift -= 8) {
            if (t < 256) hist[t] = 0;
            __syncthreads();
            for (int i = t; i < m; i += 1024) { const unsigned long long k = row[i]; if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255ull], 1u); }
            __syncthreads();
            if (t == 0) {
                int run = 0, b = 0;
                for (b = 0; b < 256; b++) { if (run + (int)hist[b] >= remaining) break; run += (int)hist[b]; }
                s_bin = b; s_before = run;
            }
            __syncthreads();
            prefix |= ((unsigned long long)s_bin) << shift; mask |= 255ull << shift; remaining -= s_before;
            __syncthreads();
        }
        if (t == 0) s_n = 0;
        __syncthreads();
        for (int i = t; i < m; i += 1024) { const unsigned long long k = row[i]; if (k <= prefix) { const int sl = atomicAdd(&s_n, 1); if (sl < SORT_MAX) sm[sl] = k; } }   // composites are unique: exactly kq
        __syncthreads();
        n = min(s_n, SORT_MAX);
    } else if (m > 1024 && kq <= 16) {
        // a few thousand survivors, a handful wanted: every wave sorts 64-composite chunks in registers (bitonic network over the
        // lanes, no barriers) and keeps each chunk's kq smallest — the row's kq smallest are among them — at most 64 * 16 = 1024
        // composites, which the counting rank below finishes. (A 2048-wide LDS bitonic sort is 66 barrier steps: the slowest
        // row of a batch set the kernel's time.)
        if (t == 0) s_n = 0;
        __syncthreads();
        const int lane = t & 63, w = t >> 6, nchunks = (m + 63) >> 6;
        for (int ch = w; ch < nchunks; ch += 16) {
            const int i = ch * 64 + lane;
            unsigned long long v = i < m ? row[i] : ~0ull;
#pragma unroll
            for (int k2 = 2; k2 <= 64; k2 <<= 1)
#pragma unroll
                for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(v & 0xFFFFFFFFull), j2, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), j2, 64);
                    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
                    const bool up = ((lane & k2) == 0), low = ((lane & j2) == 0);
                    v = (up == low) ? (v < o ? v : o) : (v > o ? v : o);
                }
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_n, kq);
            base = __shfl(base, 0, 64);
            if (lane < kq) sm[base + lane] = v;               // lanes 0..kq-1 hold the chunk's kq smallest (padding sorts last)
        }
        __syncthreads();
        n = s_n;
    } else {
        for (int i = t; i < m; i += 1024) sm[i] = row[i];
    }
    if (n <= 1024) {
        // short lists (the usual case: a few hundred survivors): rank by counting — one pass of broadcast LDS reads instead of a
        // ladder of ~50 barriers. Composites are unique, so ranks are a permutation.
        __syncthreads();
        unsigned long long me = ~0ull; int rank = 0;
        if (t < n) { me = sm[t]; for (int j = 0; j < n; j++) rank += (sm[j] < me) ? 1 : 0; }
        __syncthreads();
        if (t < n) sm[rank] = me;
        __syncthreads();
    } else {
        int n2 = 64; while (n2 < n) n2 <<= 1;
        for (int i = n + t; i < n2; i += 1024) sm[i] = ~0ull;
        __syncthreads();
        bitonic_sort_lds(sm, n2);
    }
    const int nw = kq < k_cap ? kq : k_cap;
    for (int i = t; i < k_cap; i += 1024) {
        if (i < nw && i < n && sm[i] != ~0ull) {
            const unsigned long long cc = sm[i];
            out_pos[(long)q * k_cap + i] = (unsigned)(cc & 0xFFFFFFFFull);
            out_scores[(long)q * k_cap + i] = __uint_as_float(key2f((unsigned)(cc >> 32)));
        } else { out_pos[(long)q * k_cap + i] = 0xFFFFFFFFu; out_scores[(long)q * k_cap + i] = 0.0f; }
    }
    if (t == 0) out_counts[q] = kq;
}
int select_composites_max_k() { return SORT_MAX; }
void launch_select_composites(Ctx* c, const unsigned long long* comp, int64_t ld, const int32_t* cursor, int B, int K, uint32_t* out_pos, float* out_scores,
                              int32_t* out_counts, int k_cap) {
    if (B <= 0) return;
    ProfScope ps(c, "select_composites");
    sel_composites_kernel<<<dim3(B), dim3(1024), sizeof(unsigned long long) * SORT_MAX, c->stream>>>(comp, ld, cursor, K, out_pos, out_scores, out_counts, k_cap);
    LAUNCH_CHECK();
}

void launch_select_topk(Ctx* c, const float* D, int64_t ldD, int B, int64_t C, const int32_t* cnts, float thr, int K,
                        uint32_t* out_pos, float* out_scores, int32_t* out_counts, int k_cap) {
    if (B <= 0) return;
    if (C <= 0) {  // nothing to select from: zero counts, cleared rows
        c->zero(out_counts, sizeof(int32_t) * B);
        HIP_CHECK(hipMemsetAsync(out_pos, 0xFF, sizeof(uint32_t) * (size_t)B * k_cap, c->stream));
        c->zero(out_scores, sizeof(float) * (size_t)B * k_cap);
        return;
    }
    if (C <= SMALL_MAX) {
        int n2 = 64; while (n2 < C) n2 <<= 1;
        ProfScope ps(c, "select_small");
        const int threads = C > 1024 ? 1024 : (n2 >= 512 ? 256 : 64);
        const size_t lds = C > 1024 ? (size_t)64 * 1024 : sizeof(unsigned long long) * n2;
        if (lds > 48 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)select_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        select_small_kernel<<<dim3(B), dim3(threads), lds, c->stream>>>(D, ldD, C, cnts, thr, K, out_pos, out_scores, out_counts, k_cap);
        LAUNCH_CHECK();
        return;
    }
    const int64_t kmax = (K <= 0 || K > C) ? C : K;
    const bool big = kmax > SORT_MAX;                 // the final sort does not fit in LDS: radix select as usual, then a sort in global memory
    const int nchunks = (int)ceil_div(C, SEL_CHUNK);
    dim3 grid(nchunks, B), blk(SEL_THREADS);
    // sample-bound path: expected kept candidates K*C/S held near 512 (BOUND_CAP = 8x that); needs K >= 1
    int* rowlist = nullptr;
    static const bool bound_off = getenv("COMET_SELECT_NO_BOUND") != nullptr;
    if (K >= 1 && !bound_off && !big) {
        int64_t S = std::max<int64_t>(SMALL_MAX, round_up((int64_t)K * C / 512, 1024));
        if (S <= BOUND_SAMPLE_MAX && S * 2 <= C) {
            unsigned* bound = c->salloc<unsigned>(B);
            int* cursor = c->salloc<int>(B);
            unsigned long long* kept = c->salloc<unsigned long long>((size_t)B * BOUND_CAP);
            rowlist = c->salloc<int>(B + 1);
            c->zero(rowlist, sizeof(int));
            { ProfScope ps(c, "select_bound");
              sel_bound_kernel<<<dim3(B), dim3(1024), 0, c->stream>>>(D, ldD, C, cnts, thr, K, (int)S, bound, cursor); LAUNCH_CHECK(); }
            { ProfScope ps(c, "select_filter");
              sel_filter_kernel<<<grid, blk, 0, c->stream>>>(D, ldD, C, cnts, thr, bound, cursor, kept); LAUNCH_CHECK(); }
            { ProfScope ps(c, "select_final");
              sel_final_kernel<<<dim3(B), dim3(1024), sizeof(unsigned long long) * BOUND_CAP, c->stream>>>(kept, cursor, K, out_pos, out_scores, out_counts, k_cap, rowlist);
              LAUNCH_CHECK(); }
            { ProfScope ps(c, "select_rows");
              int n2 = 64; while (n2 < kmax) n2 <<= 1;
              sel_row_kernel<<<dim3(B), dim3(1024), sizeof(unsigned long long) * n2, c->stream>>>(D, ldD, C, cnts, thr, K, rowlist, out_pos, out_scores, out_counts, k_cap);
              LAUNCH_CHECK(); }
            return;
        }
    }
    SelState* st = c->salloc<SelState>(B);
    unsigned* hist = c->salloc<unsigned>((size_t)B * SEL_BINS);
    int* eqcnt = c->salloc<int>((size_t)B * nchunks);
    int64_t comp_ld = 1; while (comp_ld < kmax) comp_ld <<= 1;
    unsigned long long* comp = c->salloc<unsigned long long>((size_t)B * comp_ld);
    c->zero(st, sizeof(SelState) * B);
    c->zero(hist, sizeof(unsigned) * (size_t)B * SEL_BINS);
    const int shifts[3] = {20, 8, 0}, bitsv[3] = {12, 12, 8};
    for (int p = 0; p < 3; p++) {
        { ProfScope ps(c, "select_hist");
          sel_hist_kernel<<<grid, blk, 0, c->stream>>>(D, ldD, C, cnts, thr, st, shifts[p], bitsv[p], hist, rowlist); LAUNCH_CHECK(); }
        { ProfScope ps(c, "select_scan");
          sel_scan_kernel<<<dim3(B), blk, 0, c->stream>>>(hist, st, shifts[p], bitsv[p], K, p == 0, rowlist); LAUNCH_CHECK(); }
    }
    { ProfScope ps(c, "select_count_eq");
      sel_count_eq_kernel<<<grid, blk, 0, c->stream>>>(D, ldD, C, cnts, thr, st, eqcnt, nchunks, rowlist); LAUNCH_CHECK(); }
    { ProfScope ps(c, "select_scan");
      sel_scan_eq_kernel<<<dim3(B), blk, 0, c->stream>>>(eqcnt, nchunks, rowlist); LAUNCH_CHECK(); }
    { ProfScope ps(c, "select_gather");
      sel_gather_kernel<<<grid, blk, 0, c->stream>>>(D, ldD, C, cnts, thr, st, eqcnt, nchunks, comp, comp_ld, rowlist); LAUNCH_CHECK(); }
    if (big) {
        sel_pad_kernel<<<dim3((unsigned)ceil_div(comp_ld, 256), B), dim3(256), 0, c->stream>>>(comp, comp_ld, st);
        sort_rows_u64(c, comp, comp_ld, B);
        sel_emit_kernel<<<dim3((unsigned)ceil_div(k_cap, 256), B), dim3(256), 0, c->stream>>>(comp, comp_ld, st, out_pos, out_scores, out_counts, k_cap);
        LAUNCH_CHECK();
        return;
    }
    { ProfScope ps(c, "select_sort");
      int threads = comp_ld >= 2048 ? 1024 : (comp_ld >= 512 ? 256 : 64);
      sel_sort_kernel<<<dim3(B), dim3(threads), sizeof(unsigned long long) * comp_ld, c->stream>>>(comp, comp_ld, st, out_pos, out_scores, out_counts, k_cap, rowlist);
      LAUNCH_CHECK(); }
}


// ---- multi-GPU merge: R sorted per-shard lists per query -> global top-k ----------------------------
// (the reference's analogue is mergeResults, storage_merge.go:13-46). Exact because the top-k of a
// union is the top-k of the per-part top-ks. Composite key = (score key, shard, position).
__global__ __launch_bounds__(1024) void merge_topk_kernel(const unsigned* __restrict__ ids, const float* __restrict__ scores,
                                                          const int* __restrict__ counts, int R, int B, int k_cap, int k,
                                                          unsigned* __restrict__ out_ids, float* __restrict__ out_scores,
                                                          int* __restrict__ out_counts, int n2, long rs /*elements between two shards' blocks*/, long rsc) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sm[];
    const int q = blockIdx.x;
    int total = 0;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        unsigned long long v = ~0ull;
        if (i < R * k_cap) {
            const int r = i / k_cap, j = i - r * k_cap;
            int cnt = counts[r * rsc + q]; if (cnt > k_cap) cnt = k_cap;
            if (j < cnt) v = ((unsigned long long)f2key(__float_as_uint(scores[r * rs + (long)q * k_cap + j])) << 32) | (unsigned)i;
        }
        sm[i] = v;
    }
    for (int r = 0; r < R; r++) { int cnt = counts[r * rsc + q]; if (cnt < 0) { total = cnt; break; } total += cnt < k_cap ? cnt : k_cap; }
    __syncthreads();
    for (int kk = 2; kk <= n2; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                int ixj = i ^ j;
                if (ixj > i) {
                    unsigned long long a = sm[i], b = sm[ixj];
                    bool up = ((i & kk) == 0);
                    if ((a > b) == up) { sm[i] = b; sm[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    int kq = total < 0 ? 0 : ((k <= 0 || k > total) ? total : k);
    if (kq > k_cap) kq = k_cap;
    for (int i = threadIdx.x; i < k_cap; i += blockDim.x) {
        unsigned id = 0; float sc = 0.0f;
        if (i < kq) {
            unsigned long long cmp = sm[i];
            unsigned src = (unsigned)(cmp & 0xFFFFFFFFull);
            const int r = src / k_cap, j = src - r * k_cap;
            id = ids[r * rs + (long)q * k_cap + j];
            sc = __uint_as_float(key2f((unsigned)(cmp >> 32)));
        }
        out_ids[(long)q * k_cap + i] = id; out_scores[(long)q * k_cap + i] = sc;
    }
    if (threadIdx.x == 0) out_counts[q] = total < 0 ? total : kq;
}
// the same merge for R x k_cap beyond the LDS: build composites, sort_rows_u64, emit
__global__ __launch_bounds__(256) void merge_build_kernel(const float* __restrict__ scores, const int* __restrict__ counts, int R, int k_cap,
                                                          unsigned long long* __restrict__ comp, long n2, long rs, long rsc) {
    const int q = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n2) return;
    unsigned long long v = ~0ull;
    if (i < (long)R * k_cap) {
        const int r = (int)(i / k_cap), j = (int)(i - (long)r * k_cap);
        int cnt = counts[r * rsc + q]; if (cnt > k_cap) cnt = k_cap;
        if (j < cnt) v = ((unsigned long long)f2key(__float_as_uint(scores[r * rs + (long)q * k_cap + j])) << 32) | (unsigned)i;
    }
    comp[(long)q * n2 + i] = v;
}
__global__ __launch_bounds__(256) void merge_emit_kernel(const unsigned* __restrict__ ids, const int* __restrict__ counts, const unsigned long long* __restrict__ comp,
                                                         int R, int k_cap, int k, unsigned* __restrict__ out_ids, float* __restrict__ out_scores,
                                                         int* __restrict__ out_counts, long n2, long rs, long rsc) {
    const int q = blockIdx.y;
    int total = 0;
    for (int r = 0; r < R; r++) { int cnt = counts[r * rsc + q]; if (cnt < 0) { total = cnt; break; } total += cnt < k_cap ? cnt : k_cap; }
    int kq = total < 0 ? 0 : ((k <= 0 || k > total) ? total : k);
    if (kq > k_cap) kq = k_cap;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < k_cap) {
        unsigned id = 0; float sc = 0.0f;
        if (i < kq) {
            const unsigned long long cmp = comp[(long)q * n2 + i];
            const unsigned src = (unsigned)(cmp & 0xFFFFFFFFull);
            const int r = src / k_cap, j = src - r * k_cap;
            id = ids[r * rs + (long)q * k_cap + j];
            sc = __uint_as_float(key2f((unsigned)(cmp >> 32)));
        }
        out_ids[(long)q * k_cap + i] = id; out_scores[(long)q * k_cap + i] = sc;
    }
    if (i == 0) out_counts[q] = total < 0 ? total : kq;
}
// bytes of workspace the merge needs in global memory (0: it runs in LDS)
size_t merge_topk_workspace_bytes(int R, int B, int k_cap) {
    int64_t n2 = 1; while (n2 < (int64_t)R * k_cap) n2 <<= 1;
    return n2 > 2 * SORT_MAX ? (size_t)B * n2 * sizeof(unsigned long long) : 0;
}
void launch_merge_topk(Ctx* c, const uint32_t* ids, const float* scores, const int32_t* counts, int R, int B, int k_cap, int k,
                       uint32_t* out_ids, float* out_scores, int32_t* out_counts, int64_t rank_stride, int64_t rank_stride_counts, void* workspace) {
    if (B <= 0) return;
    const long rs = rank_stride > 0 ? rank_stride : (long)B * k_cap, rsc = rank_stride_counts > 0 ? rank_stride_counts : B;
    int64_t n2 = 1; while (n2 < (int64_t)R * k_cap) n2 <<= 1;
    if (n2 > 2 * SORT_MAX) {            // does not fit in LDS: composites to global memory, global sort, emit
        // `workspace` (merge_topk_workspace_bytes): a caller that runs the merge on ANOTHER stream than the context's (the exchange stream of
        // comm.hip) must own the buffer — the context's scratch arena is recycled by whatever the context's own stream runs next
        ScratchMark mark(c);
        unsigned long long* comp = workspace ? static_cast<unsigned long long*>(workspace) : c->salloc<unsigned long long>((size_t)B * n2);
        { ProfScope ps(c, "merge_topk");
          merge_build_kernel<<<dim3((unsigned)ceil_div(n2, 256), B), dim3(256), 0, c->stream>>>(scores, counts, R, k_cap, comp, n2, rs, rsc); LAUNCH_CHECK(); }
        sort_rows_u64(c, comp, n2, B);
        { ProfScope ps(c, "merge_topk");
          merge_emit_kernel<<<dim3((unsigned)ceil_div(k_cap, 256), B), dim3(256), 0, c->stream>>>(ids, counts, comp, R, k_cap, k, out_ids, out_scores, out_counts, n2, rs, rsc); LAUNCH_CHECK(); }
        return;
    }
    ProfScope ps(c, "merge_topk");
    int threads = n2 >= 2048 ? 1024 : (n2 >= 512 ? 256 : 64);
    merge_topk_kernel<<<dim3(B), dim3(threads), sizeof(unsigned long long) * n2, c->stream>>>(ids, scores, counts, R, B, k_cap, k, out_ids, out_scores, out_counts, (int)n2, rs, rsc);
    LAUNCH_CHECK();
}

// ---- segment merge (mergeResults storage_merge.go:13-46 + sortResultsByScore :50-54 + the cut to k, storage.go:618-623) -----------
// in: S segments x B queries x k_cap (ids, scores), counts S x B — every segment's own top-k for the same query batch.
// out: per query the distinct ids with their HIGHEST score, sorted by score DESCENDING (the reference sorts hybrid scores — for a vector-only
// query those are distances — descending; restated as it is), cut to k. Equal scores: ascending id (the reference's order is Go map
// iteration order there, i.e. unspecified). A negative count (a search-time error code, e.g. a zero query under cosine) is passed through.
__device__ __forceinline__ unsigned long long seg_comp1(const unsigned* ids, const float* scores, const int* counts, int S, int B, int k_cap, int q, long i) {
    const long s = i / k_cap; const int j = (int)(i - s * k_cap);
    if (s >= S) return ~0ull;
    int cnt = counts[s * B + q]; if (cnt > k_cap) cnt = k_cap;
    if (j >= cnt) return ~0ull;
    const long at = (s * B + q) * (long)k_cap + j;
    unsigned b = __float_as_uint(scores[at]); if (b == 0x80000000u) b = 0u;       // -0 == +0
    return ((unsigned long long)ids[at] << 32) | (unsigned)~f2key(b);              // by id, the id's highest score first
}
__device__ __forceinline__ int seg_error(const int* counts, int S, int B, int q) { for (int s = 0; s < S; s++) { const int c = counts[s * B + q]; if (c < 0) return c; } return 0; }

__global__ __launch_bounds__(1024) void seg_merge_kernel(const unsigned* __restrict__ ids, const float* __restrict__ scores, const int* __restrict__ counts, int S, int B, int k_cap, int k,
                                                         int n2, unsigned* __restrict__ out_ids, float* __restrict__ out_scores, int* __restrict__ out_counts, int out_ld) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sm[];
    __shared__ int s_heads;
    const int q = blockIdx.x;
    if (threadIdx.x == 0) s_heads = 0;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) sm[i] = seg_comp1(ids, scores, counts, S, B, k_cap, q, i);
    __syncthreads();
    bitonic_sort_lds(sm, n2);
    // heads of the id groups -> (score descending, id ascending) composites; everything else sorts last
    constexpr int EPT = 8;                                           // n2 <= 8192 = 1024 threads x 8
    unsigned long long mine[EPT]; int nh = 0;
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int i = threadIdx.x + e * blockDim.x;
        mine[e] = ~0ull;
        if (i < n2) {
            const unsigned long long c = sm[i];
            if (c != ~0ull && (i == 0 || (sm[i - 1] >> 32) != (c >> 32))) { mine[e] = ((c & 0xFFFFFFFFull) << 32) | (c >> 32); nh++; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; e++) { const int i = threadIdx.x + e * blockDim.x; if (i < n2) sm[i] = mine[e]; }
    if (nh) atomicAdd(&s_heads, nh);
    __syncthreads();
    bitonic_sort_lds(sm, n2);
    const int err = seg_error(counts, S, B, q);
    const int heads = s_heads, kq = err ? 0 : (heads < k ? heads : k), nw = kq < out_ld ? kq : out_ld;
    for (int i = threadIdx.x; i < out_ld; i += blockDim.x) {
        unsigned id = 0xFFFFFFFFu; float sc = 0.0f;
        if (i < nw) { const unsigned long long c = sm[i]; id = (unsigned)(c & 0xFFFFFFFFull); sc = __uint_as_float(key2f(~(unsigned)(c >> 32))); }
        out_ids[(long)q * out_ld + i] = id; out_scores[(long)q * out_ld + i] = sc;
    }
    if (threadIdx.x == 0) out_counts[q] = err ? err : kq;
}
// the same beyond 8192 entries per query: composites in global memory, two global sorts
__global__ __launch_bounds__(256) void seg_build_kernel(const unsigned* __restrict__ ids, const float* __restrict__ scores, const int* __restrict__ counts, int S, int B, int k_cap,
                                                        unsigned long long* __restrict__ comp, long n2, int* __restrict__ heads) {
    const int q = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) heads[q] = 0;
    if (i < n2) comp[(long)q * n2 + i] = seg_comp1(ids, scores, counts, S, B, k_cap, q, i);
}
__global__ __launch_bounds__(256) void seg_heads_kernel(const unsigned long long* __restrict__ comp, unsigned long long* __restrict__ comp2, long n2, int* __restrict__ heads) {
    const int q = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    bool head = false;
    if (i < n2) {
        const unsigned long long c = comp[(long)q * n2 + i];
        head = c != ~0ull && (i == 0 || (comp[(long)q * n2 + i - 1] >> 32) != (c >> 32));
        comp2[(long)q * n2 + i] = head ? ((c & 0xFFFFFFFFull) << 32) | (c >> 32) : ~0ull;
    }
    const unsigned long long m = __ballot(head);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&heads[q], __popcll(m));
}
__global__ __launch_bounds__(256) void seg_emit_kernel(const unsigned long long* __restrict__ comp2, long n2, const int* __restrict__ heads, const int* __restrict__ counts, int S, int B, int k,
                                                       unsigned* __restrict__ out_ids, float* __restrict__ out_scores, int* __restrict__ out_counts, int out_ld) {
    const int q = blockIdx.y;
    const int err = seg_error(counts, S, B, q);
    const int h = heads[q], kq = err ? 0 : (h < k ? h : k), nw = kq < out_ld ? kq : out_ld;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < out_ld) {
        unsigned id = 0xFFFFFFFFu; float sc = 0.0f;
        if (i < nw) { const unsigned long long c = comp2[(long)q * n2 + i]; id = (unsigned)(c & 0xFFFFFFFFull); sc = __uint_as_float(key2f(~(unsigned)(c >> 32))); }
        out_ids[(long)q * out_ld + i] = id; out_scores[(long)q * out_ld + i] = sc;
    }
    if (i == 0) out_counts[q] = err ? err : kq;
}
void launch_merge_segments(Ctx* c, const uint32_t* ids, const float* scores, const int32_t* counts, int S, int B, int k_cap, int k,
                           uint32_t* out_ids, float* out_scores, int32_t* out_counts, int out_ld) {
    if (B <= 0) return;
    int64_t n2 = 64; while (n2 < (int64_t)S * k_cap) n2 <<= 1;
    if (n2 <= 8192) {
        ProfScope ps(c, "segments_merge");
        seg_merge_kernel<<<dim3(B), dim3(n2 >= 2048 ? 1024 : 256), sizeof(unsigned long long) * n2, c->stream>>>(ids, scores, counts, S, B, k_cap, k, (int)n2, out_ids, out_scores, out_counts, out_ld);
        LAUNCH_CHECK();
        return;
    }
    ScratchMark mark(c);
    unsigned long long* comp = c->salloc<unsigned long long>((size_t)B * n2);
    unsigned long long* comp2 = c->salloc<unsigned long long>((size_t)B * n2);
    int* heads = c->salloc<int>(B);
    const dim3 grid((unsigned)ceil_div(n2, 256), B);
    { ProfScope ps(c, "segments_merge"); seg_build_kernel<<<grid, dim3(256), 0, c->stream>>>(ids, scores, counts, S, B, k_cap, comp, n2, heads); LAUNCH_CHECK(); }
    sort_rows_u64(c, comp, n2, B);
    { ProfScope ps(c, "segments_merge"); seg_heads_kernel<<<grid, dim3(256), 0, c->stream>>>(comp, comp2, n2, heads); LAUNCH_CHECK(); }
    sort_rows_u64(c, comp2, n2, B);
    { ProfScope ps(c, "segments_merge"); seg_emit_kernel<<<dim3((unsigned)ceil_div(out_ld, 256), B), dim3(256), 0, c->stream>>>(comp2, n2, heads, counts, S, B, k, out_ids, out_scores, out_counts, out_ld); LAUNCH_CHECK(); }
}

}  // namespace comet
