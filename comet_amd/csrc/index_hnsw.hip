// index_hnsw.hip — HNSW search on the GPU (reference: hnsw_index_search.go:248-354 and
// HNSWIndex.searchLayer hnsw_index.go:565-629).
//
// One wave per query. Graph traversal is a serial pointer chase whose order is fixed by the two Go
// container/heap structures (candidates min-heap, result max-heap); to return the reference's results —
// including its behaviour on equal distances — lane 0 executes exactly those heap operations (same
// up/down sift code as Go's stdlib) on LDS arrays. What is data-parallel is the neighbour batch: the
// up-to-64 neighbours of the node being expanded are filtered (soft-deleted, visited bitmap) and their
// exact float32 distances computed one lane per neighbour from an LDS-transposed tile (coalesced
// 256-byte row pieces, conflict-free ds_read_b128), after which lane 0 replays the reference's
// sequential push/pop decisions over the batch in edge order.
//
// Graph construction (insertNode, hnsw_index.go:493-552) is not on the GPU yet: a graph built by the
// reference (or the oracle) is loaded with comet_hnsw_load_graph.
#include <type_traits>
#include "index.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace comet {

__device__ __forceinline__ float hn_sqrt32(float x) { return (float)__builtin_sqrt((double)x); }
template <int METRIC> __device__ __forceinline__ float hn_step(float acc, float q, float x) {
    if constexpr (METRIC == COMET_COSINE) { float p = q * x; return acc + p; }
    else { float diff = q - x; float sq = diff * diff; return acc + sq; }
}
template <int METRIC> __device__ __forceinline__ float hn_finish(float acc) {
    if constexpr (METRIC == COMET_COSINE) { if (acc > 1.0f) acc = 1.0f; else if (acc < -1.0f) acc = -1.0f; return 1.0f - acc; }
    else if constexpr (METRIC == COMET_L2) return hn_sqrt32(acc);
    else return acc;
}

constexpr int HN_TD = 64;            // floats of each row staged per pass (256 B)
constexpr int HN_LD = HN_TD + 4;     // LDS row stride
constexpr int HN_CAND_CAP = 4096;    // candidates min-heap capacity (LDS)
constexpr int HN_EF_MAX = 1024;
constexpr unsigned HN_NONE = 0xFFFFFFFFu;
#ifdef HN_TRACE
__device__ unsigned long long* g_hn_trace = nullptr;      // phase sums (s_memtime ticks) over all queries: pop | record+edges | distances | replay | - | loop | row fetch | slices
#endif

struct HC { unsigned id; float d; };

// Go container/heap on an LDS array (heap.go: up / down / Push / Pop)
template <bool MAXHEAP> __device__ __forceinline__ bool hless(const HC& a, const HC& b) { return MAXHEAP ? (a.d > b.d) : (a.d < b.d); }
template <bool MAXHEAP> __device__ void heap_push(HC* h, int& n, HC x) {
    h[n] = x; int j = n; n++;
    for (;;) { int i = (j - 1) / 2; if (i == j || !hless<MAXHEAP>(h[j], h[i])) break; HC t = h[i]; h[i] = h[j]; h[j] = t; j = i; }
}
template <bool MAXHEAP> __device__ HC heap_pop(HC* h, int& n) {
    const int m = n - 1;
    { HC t = h[0]; h[0] = h[m]; h[m] = t; }
    int i = 0;
    for (;;) {
        int j1 = 2 * i + 1; if (j1 >= m || j1 < 0) break;
        int j = j1; const int j2 = j1 + 1;
        if (j2 < m && hless<MAXHEAP>(h[j2], h[j1])) j = j2;
        if (!hless<MAXHEAP>(h[j], h[i])) break;
        HC t = h[i]; h[i] = h[j]; h[j] = t; i = j;
    }
    n = m;
    return h[m];
}

// The same two operations with the whole wave cooperating (every lane calls them with the same arguments and gets the same
// result). Heap traffic — not the memory round trips of the traversal — is what a search spends most of its time on (about a
// thousand cycles per push or pop when one lane chases dependent LDS reads level by level), so:
//   push: the ancestors of the new slot are known up front, a_k = ((n + 1) >> k) - 1: lane k reads ancestor k, a ballot finds the
//         first one the new element does not beat, the ancestors below it move down one level each and the element lands — one
//         read and one write round instead of one per level. The final array is exactly what heap.go's up() leaves.
//   pop:  down() must follow the better child level by level, but the moving element stays in registers and both children
//         arrive in one round trip per level (every lane reads the same two slots: broadcasts).
typedef unsigned long long hc_bits;
__device__ __forceinline__ HC hc_load(const HC* p) { const hc_bits v = *reinterpret_cast<const hc_bits*>(p); HC r; r.id = (unsigned)(v & 0xFFFFFFFFull); r.d = __uint_as_float((unsigned)(v >> 32)); return r; }
__device__ __forceinline__ void hc_store(HC* p, HC x) { *reinterpret_cast<hc_bits*>(p) = ((hc_bits)__float_as_uint(x.d) << 32) | x.id; }
template <bool MAXHEAP> __device__ __forceinline__ void heap_push_wave(HC* h, int& n, HC x) {
    const int lane = threadIdx.x & 63;
    const int p1 = n + 1;                                   // 1-based position of the new slot
    const bool valid = lane >= 1 && lane < 31 && (p1 >> lane) >= 1;      // lane k: k-th ancestor exists
    const int ak = valid ? (p1 >> lane) - 1 : 0;
    HC v = hc_load(h + ak);
    const unsigned long long stopm = __ballot(valid && !hless<MAXHEAP>(x, v));
    const unsigned long long validm = __ballot(valid);
    int depth = validm ? 64 - __builtin_clzll(validm) : 1;  // ancestors are lanes 1 .. depth-1
    const int t = stopm ? __builtin_ctzll(stopm) : depth;   // first ancestor x does not beat (depth: x becomes the root)
    // ancestors 1 .. t-1 move down to the slot of their child on the path (a_{k-1}); x takes a_{t-1}
    if (valid && lane < t) hc_store(h + ((p1 >> (lane - 1)) - 1), v);
    if (lane == 0) hc_store(h + ((p1 >> (t - 1)) - 1), x);
    n = n + 1;
    __builtin_amdgcn_wave_barrier();
}
// Pop of a heap of at most 129 entries (m = n - 1 <= 128 live slots after the swap: at most 64 internal nodes, ONE PER LANE; the result heap at
// efSearch <= 128). down() follows, from the root, the child that wins among each node's two children — which child that is does not depend on the
// element being sifted, and a swap only ever moves y and the followed child: the children of the nodes further down are the array's values from
// before the pop. So every lane decides "left or right" for its own node in one round of LDS reads, a ballot carries the 64 decisions to every lane,
// lane l walks l steps of the path in registers (the path has at most 7 levels), reads the path's node of its level, and one more ballot finds
// where y stops; the path's nodes above that move up one level each and y lands. The same array heap.go's down() leaves, in ~60 instructions
// instead of ~20 per level (a wave of this kernel pays ~8 clocks per instruction of any kind).
template <bool MAXHEAP> __device__ __forceinline__ HC heap_pop_wave_small(HC* h, int& n) {
    const int lane = threadIdx.x & 63;
    const int m = n - 1;
    const HC root = hc_load(h), y = hc_load(h + m);         // h.Swap(0, m)
    const int j1 = 2 * lane + 1;
    const HC c1 = hc_load(h + (j1 < m ? j1 : 0)), c2 = hc_load(h + (j1 + 1 < m ? j1 + 1 : 0));
    const bool right = (j1 + 1 < m) && hless<MAXHEAP>(c2, c1);          // `j2 < n && h.Less(j2, j1)`
    const unsigned long long rmask = __ballot(right);
    int node = 0, par = 0;                                              // the path's node at level `lane`, and the one above it
#pragma unroll
    for (int k = 0; k < 7; k++) if (k < lane) { par = node; node = 2 * node + 1 + (int)((rmask >> (node & 63)) & 1ull); }
    const bool valid = lane >= 1 && lane <= 7 && node < m;              // `j1 >= n -> break`
    const HC c = hc_load(h + (valid ? node : 0));
    const bool cont = valid && hless<MAXHEAP>(c, y);                    // `!h.Less(j, i) -> break`
    const int t = __builtin_ctzll(__ballot(!cont && lane >= 1));        // first level down() does not reach: y lands on the path's node of level t - 1
    if (cont && lane < t) hc_store(h + par, c);
    if (lane == t - 1) hc_store(h + node, y);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) hc_store(h + m, root);
    __builtin_amdgcn_wave_barrier();
    n = m;
    return root;
}
// The same for a heap of up to 4 095 entries (the candidate heap: <= 12 levels) in TWO rounds: the decisions of the 63 nodes of levels 0-5 give the
// path's node of level 6; the 31 nodes of the five levels under THAT node (a sub-tree with the same index arithmetic relative to its root) give the
// rest. Four dependent LDS round trips instead of one per level.
template <bool MAXHEAP> __device__ __forceinline__ HC heap_pop_wave_mid(HC* h, int& n) {
    const int lane = threadIdx.x & 63;
    const int m = n - 1;
    const HC root = hc_load(h), y = hc_load(h + m);         // h.Swap(0, m)
    auto decide = [&](int node) __attribute__((always_inline)) {      // would down() go right at `node`?
        const int j1 = 2 * node + 1;
        const HC c1 = hc_load(h + (j1 < m ? j1 : 0)), c2 = hc_load(h + (j1 + 1 < m ? j1 + 1 : 0));
        return (j1 + 1 < m) && hless<MAXHEAP>(c2, c1);
    };
    // a sub-tree's node t (0 = its root, children 2t+1 / 2t+2) as an index of the heap whose node `r` is that root
    auto glob = [&](int r, int t) __attribute__((always_inline)) { const int d = 31 - __builtin_clz(t + 1); return ((r + 1) << d) - 1 + (t + 1 - (1 << d)); };
    const unsigned long long mA = __ballot(lane < 63 && decide(lane));
    int p6 = 0;                                                         // the path's node of level 6 (uniform; may lie beyond m)
#pragma unroll
    for (int k = 0; k < 6; k++) p6 = 2 * p6 + 1 + (int)((mA >> p6) & 1ull);
    const unsigned long long mB = __ballot(lane < 31 && decide(glob(p6, lane)));
    // lane l: the path's node of level l and of level l - 1
    int node = 0, par = 0;
    {
        int t = 0, tp = 0;                                               // walk inside tree A (levels <= 6) or tree B (levels 7 ..)
        const bool inB = lane > 6;
        const unsigned long long mk = inB ? mB : mA;
        const int steps = inB ? lane - 6 : lane;
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < steps) { tp = t; t = 2 * t + 1 + (int)((mk >> (t & 63)) & 1ull); }
        node = inB ? glob(p6, t) : t;
        par = inB ? glob(p6, tp) : tp;                                    // (lane 7: tp = 0 -> p6)
    }
    const bool valid = lane >= 1 && lane <= 11 && node < m && (lane <= 6 || p6 < m);
    const HC c = hc_load(h + (valid ? node : 0));
    const bool cont = valid && hless<MAXHEAP>(c, y);
    const int t_stop = __builtin_ctzll(__ballot(!cont && lane >= 1));
    if (cont && lane < t_stop) hc_store(h + par, c);
    if (lane == t_stop - 1) hc_store(h + node, y);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) hc_store(h + m, root);
    __builtin_amdgcn_wave_barrier();
    n = m;
    return root;
}
template <bool MAXHEAP> __device__ __forceinline__ HC heap_pop_wave(HC* h, int& n) {
    if (n <= 129) return heap_pop_wave_small<MAXHEAP>(h, n);
    if (n <= 4095) return heap_pop_wave_mid<MAXHEAP>(h, n);
    const int m = n - 1;
    const HC root = hc_load(h), y = hc_load(h + m);         // h.Swap(0, m): y sifts down among the first m slots, the old root leaves
    int i = 0;
    for (;;) {
        const int j1 = 2 * i + 1;
        if (j1 >= m || j1 < 0) break;
        const HC c1 = hc_load(h + j1), c2 = hc_load(h + (j1 + 1 < m ? j1 + 1 : j1));
        int j = j1; HC c = c1;
        if (j1 + 1 < m && hless<MAXHEAP>(c2, c1)) { j = j1 + 1; c = c2; }
        if (!hless<MAXHEAP>(c, y)) break;
        if ((threadIdx.x & 63) == 0) hc_store(h + i, c);
        __builtin_amdgcn_wave_barrier();
        i = j;
    }
    if ((threadIdx.x & 63) == 0) { hc_store(h + i, y); hc_store(h + m, root); }
    __builtin_amdgcn_wave_barrier();
    n = m;
    return root;
}

// exact distances from the query to up to 64 rows (rows[j] == HN_NONE -> skipped); lane j gets row j's distance.
// The traversal is a chain of dependent random reads, so every exposed load latency is paid once per expansion — and a row is ld / 64
// slices of 256 bytes. Round 5: the slices are requested in BLOCKS of DEPTH (all 256-byte pieces of DEPTH slices of every row in flight at
// once, 192 registers: a wave of this kernel has the SIMD to itself) and consumed in order — at d = 384 and <= 32 neighbours the whole
// row set is ONE block, one latency per expansion instead of one per slice (the one-slice-ahead prefetch of rounds 2-4 exposed most of
// a ~1 us random-read latency six times per expansion). NI = load instructions per slice (4 rows each): 8 for <= 32 rows, else 16.
// a - b on two floats at once. hipcc turns `a - b` (and `a + (-b)`) on float2 into two v_sub_f32; the packed add with both halves of the second operand
// negated is the same IEEE subtraction, one instruction
__device__ __forceinline__ f32x2 hn_pk_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int METRIC, int NI, int DEPTH>
__device__ __forceinline__ float wave_dists_n(const float* __restrict__ V, int ld, const float* __restrict__ qv, const unsigned* rows, int cnt, float* tile) {
    const int lane = threadIdx.x;
    float acc = 0.0f;
    const int lrow = lane >> 4, lc = (lane & 15) * 4;      // loader: 4 rows x 16 float4 per instruction
    const float* src[NI];
#pragma unroll
    for (int j = 0; j < NI; j++) {
        const int r = j * 4 + lrow;
        unsigned row = r < cnt ? rows[r] : HN_NONE; if (row == HN_NONE) row = 0;
        src[j] = V + (long)row * ld + lc;
    }
    const int nsl = (ld + HN_TD - 1) / HN_TD;
    for (int sb = 0; sb < nsl; sb += DEPTH) {
        f32x4 xb[DEPTH][NI], qb[DEPTH];                    // qb: the query's four floats of this lane's column group, per slice
#pragma unroll
        for (int dd = 0; dd < DEPTH; dd++) {
            if (sb + dd >= nsl) break;                      // wave-uniform
            const int k0 = (sb + dd) * HN_TD;
            const int kn = min(HN_TD, ld - k0);             // multiple of 32
            qb[dd] = (lc < kn) ? *reinterpret_cast<const f32x4*>(qv + k0 + lc) : f32x4{0, 0, 0, 0};
            // every one of the NI instructions is issued (rows past cnt re-read row 0: cache hits) — a wave of this kernel pays ~8 clocks per instruction of
            // ANY kind, and a uniform branch + exec juggling per piece was as much of a slice as its sums
            if (kn == HN_TD) {
#pragma unroll
                for (int j = 0; j < NI; j++) xb[dd][j] = *reinterpret_cast<const f32x4*>(src[j] + k0);
            } else {
#pragma unroll
                for (int j = 0; j < NI; j++) xb[dd][j] = (lc < kn) ? *reinterpret_cast<const f32x4*>(src[j] + k0) : f32x4{0, 0, 0, 0};
            }
        }
#ifdef HN_TRACE
        const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tw1 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int dd = 0; dd < DEPTH; dd++) {
            if (sb + dd < nsl) {                            // wave-uniform
                const int k0 = (sb + dd) * HN_TD;
                const int kn = min(HN_TD, ld - k0);
                // The PRODUCTS are formed in the loader's layout — all 64 lanes busy, the lane's query floats in registers, differences and squares two at a
                // time (v_pk_add_f32 with a negated operand / v_pk_mul_f32: the same IEEE operations as the scalar ones) — and the tile carries them to the
                // lane that owns the row, which only adds, in the reference's order. (Rounds 2-4 formed them in the summing lane: half the lanes idle,
                // the query piece in SGPRs — which the packed instructions do not take — three instructions per element instead of 1.5.)
                const f32x2 qlo = {qb[dd][0], qb[dd][1]}, qhi = {qb[dd][2], qb[dd][3]};
#pragma unroll
                for (int j = 0; j < NI; j++) {
                    const f32x2 xlo = {xb[dd][j][0], xb[dd][j][1]}, xhi = {xb[dd][j][2], xb[dd][j][3]};
                    f32x2 tlo, thi;
                    if constexpr (METRIC == COMET_COSINE) { tlo = qlo * xlo; thi = qhi * xhi; }
                    else { const f32x2 dlo = hn_pk_sub(qlo, xlo), dhi = hn_pk_sub(qhi, xhi); tlo = dlo * dlo; thi = dhi * dhi; }
                    *reinterpret_cast<f32x4*>(&tile[(j * 4 + lrow) * HN_LD + lc]) = f32x4{tlo[0], tlo[1], thi[0], thi[1]};      // (a short last slice: zeros past the row's end)
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < cnt) {
                    const float* tp = &tile[lane * HN_LD];
                    auto sum = [&](auto NQ_c) __attribute__((always_inline)) {
                        constexpr int NQ = decltype(NQ_c)::value;
                        f32x4 tv[NQ];
#pragma unroll
                        for (int i = 0; i < NQ; i++) tv[i] = *reinterpret_cast<const f32x4*>(tp + 4 * i);
#pragma unroll
                        for (int i = 0; i < NQ; i++) { acc = acc + tv[i][0]; acc = acc + tv[i][1]; acc = acc + tv[i][2]; acc = acc + tv[i][3]; }
                    };
                    if (kn == HN_TD) sum(std::integral_constant<int, HN_TD / 4>{}); else sum(std::integral_constant<int, HN_TD / 8>{});
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
#ifdef HN_TRACE
        if (lane == 0 && g_hn_trace && cnt > 1) { atomicAdd(&g_hn_trace[6], tw1 - tw0); atomicAdd(&g_hn_trace[7], __builtin_amdgcn_s_memtime() - tw1); }
#endif
    }
    return hn_finish<METRIC>(acc);
}
template <int METRIC>
__device__ __forceinline__ float wave_dists(const float* __restrict__ V, int ld, const float* __restrict__ qv, const unsigned* rows, int cnt, float* tile) {
    if (cnt <= 32) return wave_dists_n<METRIC, 8, 6>(V, ld, qv, rows, cnt, tile);
    return wave_dists_n<METRIC, 16, 3>(V, ld, qv, rows, cnt, tile);
}

// Graph in HBM: per (node, layer <= level) one edge SLOT. edge_off[s] is the slot's first entry in `edges` (dense node indices),
// deg[s] its current length; a slot has a fixed CAPACITY edge_off[s+1] - edge_off[s] (M on the upper layers, 2M on layer 0 — what
// pruneConnections keeps, hnsw_index.go:667-694 — or the list's length for a loaded graph that is longer), so that insertNode can
// append and prune in place.
struct HnswGraph {
    const float* V; int ld; long n;
    const int* level; const long* slot_base; const long* edge_off; const int* deg; unsigned* edges;
    unsigned entry; int max_level;
    const unsigned* deleted;   // bitmap over dense node indices (nullable)
    const long* rec0 = nullptr;   // per node {edge_off[slot_base[node]], slot_base[node]}: the layer-0 slot in ONE dependent load (hnsw_rec0_kernel)
    long n_edges = 0;             // entries of `edges` in use (the speculative first edge batch is clamped to it)
};
__global__ __launch_bounds__(256) void hnsw_rec0_kernel(const long* __restrict__ slot_base, const long* __restrict__ edge_off, long first, long count, long* __restrict__ rec0) {
    const long i = first + (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= first + count) return;
    const long s = slot_base[i];
    rec0[2 * i] = edge_off[s]; rec0[2 * i + 1] = s;
}
__device__ __forceinline__ bool bit_get(const unsigned* bm, unsigned i) { return bm && ((bm[i >> 5] >> (i & 31)) & 1u); }

// LDS work area of one wave (search and insert kernels)
struct HnswLds { HC* cand; HC* res; float* tile; unsigned* rows; float* dd; unsigned* s_cur; int* s_flag; float* s_dist; int cand_cap; };

// greedy descent through layers (from_layer .. to_layer+1): hnsw_index_search.go:274-300 / insertNode hnsw_index.go:497-518
template <int METRIC>
__device__ __forceinline__ void hn_descend(const HnswGraph& g, const float* __restrict__ qv, int from_layer, int to_layer, unsigned& curr, float& curr_dist,
                                           const HnswLds& L, unsigned long long& n_eval) {
    const int lane = threadIdx.x;
    for (int lc = from_layer; lc > to_layer; lc--) {
        bool changed = true;
        while (changed) {
            changed = false;
            const unsigned node = curr;                 // `node := s.index.nodes[curr]` — its edge list is scanned to the end
            if (lc <= g.level[node]) {                  // `if lc < len(node.Edges)`
                const long s = g.slot_base[node] + lc;
                const long off = g.edge_off[s]; const int deg = g.deg[s];
                for (int b0 = 0; b0 < deg; b0 += 64) {
                    const int cnt = min(64, deg - b0);
                    unsigned nb = HN_NONE;
                    if (lane < cnt) { nb = g.edges[off + b0 + lane]; if (bit_get(g.deleted, nb)) nb = HN_NONE; }   // deleted neighbours are skipped
                    L.rows[lane] = nb;
                    __builtin_amdgcn_wave_barrier();
                    const float d = wave_dists<METRIC>(g.V, g.ld, qv, L.rows, cnt, L.tile);
                    L.dd[lane] = d;
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) {
                        unsigned c = curr; float cd = curr_dist; int ch = 0;
                        for (int j = 0; j < cnt; j++) if (L.rows[j] != HN_NONE && L.dd[j] < cd) { cd = L.dd[j]; c = L.rows[j]; ch = 1; }
                        *L.s_cur = c; *L.s_flag = ch; *L.s_dist = cd;
                    }
                    __builtin_amdgcn_wave_barrier();
                    curr = *L.s_cur; curr_dist = *L.s_dist; if (*L.s_flag) changed = true;
                    __builtin_amdgcn_wave_barrier();
                    n_eval += cnt;
                }
            }
        }
    }
}

// HNSWIndex.searchLayer (hnsw_index.go:565-629) on `layer` with Go container/heap semantics; leaves the result max-heap in L.res
// (nres entries). `vis` must be all-clear on entry. Returns the overflow flag of the candidate heap.
template <int METRIC>
__device__ __forceinline__ int hn_search_layer(const HnswGraph& g, const float* __restrict__ qv, unsigned entry, int ef, int layer, unsigned* __restrict__ vis,
                                               const HnswLds& L, int& nres_out, unsigned long long& n_eval, unsigned long long& n_exp) {
    const int lane = threadIdx.x;
    HC* cand = L.cand; HC* res = L.res;
    int ncand = 0, nres = 0, overflow = 0;                  // wave-uniform: every lane runs the control flow, the heap ops are cooperative
    {
        const bool dead = bit_get(g.deleted, entry);
        if (lane == 0) L.rows[0] = entry;
        __builtin_amdgcn_wave_barrier();
        const float d = __shfl(wave_dists<METRIC>(g.V, g.ld, qv, L.rows, 1, L.tile), 0, 64);
        if (!dead) { const HC x{entry, d}; heap_push_wave<false>(cand, ncand, x); heap_push_wave<true>(res, nres, x); }
        if (lane == 0) atomicOr(&vis[entry >> 5], 1u << (entry & 31));
        n_eval += 1;
    }
#ifdef HN_TRACE
    unsigned long long tr_acc[6] = {0, 0, 0, 0, 0, 0}, tr_last = __builtin_amdgcn_s_memtime();
#define HN_T(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tr_acc[i] += now_ - tr_last; tr_last = now_; } while (0)
#else
#define HN_T(i) do { } while (0)
#endif
    for (;;) {
        if (ncand == 0) break;
        HN_T(5);
        const HC cur = heap_pop_wave<false>(cand, ncand);
        HN_T(0);
        if (nres >= ef && cur.d > hc_load(res).d) break;    // early termination hnsw_index.go:592-594
        const unsigned cid = cur.id;
        n_exp += 1;
        // layer 0 (every node has it): the slot comes from ONE 16-byte record per node, and the first 64 entries of the list are requested beside
        // the degree (entries past the degree are dropped, past the array clamped) — three dependent round trips per expansion instead of four
        long s, off; unsigned e0 = HN_NONE; bool have_e0 = false;
        if (layer == 0 && g.rec0) {
            off = g.rec0[2 * (long)cid]; s = g.rec0[2 * (long)cid + 1];
            const long e_at = off + lane < g.n_edges ? off + lane : g.n_edges - 1;
            e0 = g.edges[e_at]; have_e0 = true;
        } else {
            if (layer > g.level[cid]) continue;                                        // `if layer < len(node.Edges)`
            s = g.slot_base[cid] + layer; off = g.edge_off[s];
        }
        const int deg = g.deg[s];
        for (int b0 = 0; b0 < deg; b0 += 64) {
            const int cnt = min(64, deg - b0);
            // The traversal is a chain of dependent memory round trips, so the visited test-and-set and the soft-delete lookup are
            // only REQUESTED here; the neighbours' rows are fetched and their distances evaluated meanwhile, and a neighbour that
            // turns out visited or deleted is dropped afterwards (a lane per neighbour: the wasted evaluations cost no time).
            unsigned nb = HN_NONE, vold = 0u; bool dead = false;
            if (lane < cnt) {
                nb = (have_e0 && b0 == 0) ? e0 : g.edges[off + b0 + lane];
                dead = bit_get(g.deleted, nb);                                             // SOFT DELETE CHECK
                if (!dead) vold = atomicOr(&vis[nb >> 5], 1u << (nb & 31));                // !visited.Contains -> visited.Add
            }
            L.rows[lane] = nb;
            __builtin_amdgcn_wave_barrier();
#ifdef HN_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            HN_T(1);
            const float d = wave_dists<METRIC>(g.V, g.ld, qv, L.rows, cnt, L.tile);
            HN_T(2);
            if (lane < cnt && (dead || ((vold >> (nb & 31)) & 1u))) nb = HN_NONE;
            // the batch in edge-list order (hnsw_index.go:600-619), every lane following along: lane j's pair is broadcast
            // the top of a FULL result heap only ever moves down while the batch is replayed, so a neighbour at or above the top as it is now cannot be
            // admitted later in the batch either: such lanes leave the replay up front (a NaN distance stays in: `dj < top` is false for it in the
            // reference too, but the comparison below must be the one that says so)
            const bool full0 = nres >= ef; const float top0 = full0 ? hc_load(res).d : 0.0f;
            unsigned long long todo = __ballot(lane < cnt && nb != HN_NONE && !(full0 && d >= top0));
            while (todo) {
                const int j = __builtin_ctzll(todo); todo &= todo - 1ull;
                const float dj = __shfl(d, j, 64); const unsigned idj = (unsigned)__shfl((int)nb, j, 64);
                if (nres < ef || dj < hc_load(res).d) {
                    if (ncand >= L.cand_cap) { overflow = 1; continue; }
                    const HC x{idj, dj};
                    heap_push_wave<false>(cand, ncand, x);
                    heap_push_wave<true>(res, nres, x);
                    if (nres > ef) (void)heap_pop_wave<true>(res, nres);
                }
            }
            n_eval += cnt;
            HN_T(3);
        }
    }
#ifdef HN_TRACE
    if (lane == 0 && g_hn_trace) for (int i = 0; i < 6; i++) atomicAdd(&g_hn_trace[i], tr_acc[i]);
#endif
    nres_out = nres;
    return overflow;
}

// LDS of one wave: candidate heap [cand_cap] | result heap [res_cap] | staging tile | rows | distances. The search kernel sizes the heaps
// for its efSearch (the 59 KiB of the full-size layout admit two waves per CU: a search at efSearch 128 needs 1 KiB of results and
// rarely more than a few hundred live candidates); the insert kernel keeps the full-size layout.
__device__ __forceinline__ HnswLds hn_carve(unsigned char* smem, unsigned* s_cur, int* s_flag, float* s_dist, int cand_cap = HN_CAND_CAP, int res_cap = HN_EF_MAX + 1, int tile_rows = 64) {
    HnswLds L;
    L.cand = reinterpret_cast<HC*>(smem);                                    // cand_cap
    L.res = L.cand + cand_cap;                                               // ef + 1 <= res_cap
    L.tile = reinterpret_cast<float*>(L.res + res_cap);                      // tile_rows x HN_LD (32 rows when no edge list of the graph is longer)
    L.rows = reinterpret_cast<unsigned*>(L.tile + tile_rows * HN_LD);        // 64
    L.dd = reinterpret_cast<float*>(L.rows + 64);                            // 64
    L.s_cur = s_cur; L.s_flag = s_flag; L.s_dist = s_dist; L.cand_cap = cand_cap;
    return L;
}
static size_t hn_lds_bytes(int cand_cap, int res_cap, int tile_rows = 64) { return sizeof(HC) * ((size_t)cand_cap + res_cap) + sizeof(float) * tile_rows * HN_LD + 64 * 4 + 64 * 4 + 64; }
// The result max-heap drained into ascending order (hnsw_index.go:623-626: pop everything, fill from the back). Popping is a
// chain of dependent LDS round trips on one lane (~800 cycles per pop); when all distances are distinct the popped order is
// simply the sorted order, which the whole wave finds by counting ranks. Equal distances (duplicated vectors) keep the serial
// pops: their order depends on the heap's shape. emit(i, id, d) receives position i of the ascending list.
template <class Emit>
__device__ __forceinline__ void hn_drain(HC* res, int nres, const HnswLds& L, Emit emit) {
    const int lane = threadIdx.x;
    bool tie = false;
    for (int i = lane; i < nres && !tie; i += 64) {
        const float di = res[i].d;
        if (!(di == di)) { tie = true; break; }     // a NaN distance (non-finite query): ranks by counting would all be 0 — the serial pops are what the reference does
        for (int j = 0; j < nres; j++) if (j != i && res[j].d == di) { tie = true; break; }
    }
    if (__ballot(tie) == 0ull) {
        for (int i = lane; i < nres; i += 64) {
            const HC me = res[i]; int rank = 0;
            for (int j = 0; j < nres; j++) rank += (res[j].d < me.d) ? 1 : 0;
            emit(rank, me.id, me.d);
        }
    } else if (lane == 0) {
        int m = nres;
        for (int i = nres - 1; i >= 0; i--) { const HC x = heap_pop<true>(res, m); emit(i, x.id, x.d); }
    }
    __builtin_amdgcn_wave_barrier();
}


constexpr size_t HN_LDS_BYTES = sizeof(HC) * (HN_CAND_CAP + HN_EF_MAX + 1) + sizeof(float) * 64 * HN_LD + 64 * 4 + 64 * 4 + 64;

template <int METRIC>
__global__ __launch_bounds__(64) void hnsw_search_kernel(HnswGraph g, const float* __restrict__ Qp, int ef, unsigned* __restrict__ visited /*[B][vwords]*/,
                                                         long vwords, unsigned* __restrict__ res_idx, float* __restrict__ res_dist,
                                                         int* __restrict__ res_cnt, int* __restrict__ status, unsigned long long* __restrict__ stats,
                                                         HC* __restrict__ spill /*nullable: heaps in HBM*/, long spill_cand, long spill_res, int ef_ld,
                                                         int cand_cap, int res_cap, int tile_rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned s_cur; __shared__ int s_flag; __shared__ float s_dist;
    HnswLds L = hn_carve(smem, &s_cur, &s_flag, &s_dist, cand_cap, res_cap, tile_rows);
    const int q = blockIdx.x, lane = threadIdx.x;
    if (spill) {        // ef beyond the LDS heaps (or a candidate heap that outgrew them): same code, heaps in HBM, one slab per query
        L.cand = spill + (long)q * (spill_cand + spill_res); L.res = L.cand + spill_cand;
        L.cand_cap = (int)min(spill_cand, 0x7FFFFFFFl);
    }
    const float* __restrict__ qv = Qp + (long)q * g.ld;
    unsigned* vis = visited + (long)q * vwords;
    unsigned long long n_eval = 0, n_exp = 0;

    // ---- phase 1: greedy descent through the upper layers (hnsw_index_search.go:271-296) ----
    unsigned curr = g.entry;
    if (lane == 0) L.rows[0] = curr;
    __builtin_amdgcn_wave_barrier();
    float d0 = wave_dists<METRIC>(g.V, g.ld, qv, L.rows, 1, L.tile);
    float curr_dist = __shfl(d0, 0, 64);
    n_eval += 1;
    hn_descend<METRIC>(g, qv, g.max_level, 0, curr, curr_dist, L, n_eval);

    // ---- phase 2: searchLayer(query, curr, ef, 0) (hnsw_index.go:565-629) ----
    int nres = 0;
    const int overflow = hn_search_layer<METRIC>(g, qv, curr, ef, 0, vis, L, nres, n_eval, n_exp);
    // drain the result heap into an ascending array (hnsw_index.go:623-626)
    hn_drain(L.res, nres, L, [&](int i, unsigned id, float d) { res_idx[(long)q * ef_ld + i] = id; res_dist[(long)q * ef_ld + i] = d; });
    if (lane == 0) {
        const int n = nres;
        res_cnt[q] = n;
        if (overflow) *status = 1;
        if (stats) { atomicAdd(&stats[0], n_eval); atomicAdd(&stats[1], n_exp); }
    }
}

// ------------------------------------------------------------------------------------------------
// insertNode on the GPU (hnsw_index.go:493-552 with selectNeighbors :637-656 and pruneConnections :667-694), one wave, the
// nodes of a batch strictly one after the other — the reference's semantics are sequential (every insertion searches the
// graph the previous ones left behind), so the parallelism is the same as in the search: the neighbour batch.
// The nodes [first, first + count) already have their vectors, levels and (empty) edge slots in place.
// state[0] = maxLevel, state[1] = entry point (dense index), state[2] = 1 once the graph has an entry point.
// Reference quirks kept: maxLevel is raised BEFORE the descent (hnsw_index.go:258-260) and the entry point never moves;
// pruneConnections runs while the new node is not yet in idx.nodes (:281-282), so a full neighbour list drops the fresh
// back-edge and is merely re-sorted by distance.
// ------------------------------------------------------------------------------------------------
template <int METRIC>
__global__ __launch_bounds__(64) void hnsw_insert_kernel(HnswGraph g, int* __restrict__ deg_rw, unsigned char* __restrict__ sorted, long first, long count, int M, int efc, unsigned* __restrict__ vis,
                                                         long vwords, int* __restrict__ state, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned s_cur; __shared__ int s_flag; __shared__ float s_dist;
    const HnswLds L = hn_carve(smem, &s_cur, &s_flag, &s_dist);
    unsigned* sel = reinterpret_cast<unsigned*>(smem + HN_LDS_BYTES);          // neighbours selected on the current layer (<= max(2M, efc))
    float* pd = reinterpret_cast<float*>(sel + HN_EF_MAX + 1);                 // prune: distances of a neighbour's list (<= capacity)
    unsigned* pe = reinterpret_cast<unsigned*>(pd + 256);                      // prune: the list itself
    const int lane = threadIdx.x;
    unsigned long long n_eval = 0, n_exp = 0;
    int ml = state[0], entry = state[1], has_entry = state[2];                 // wave-uniform; written back at the end
    for (long t = 0; t < count; t++) {
        const unsigned ix = (unsigned)(first + t);
        const int lx = g.level[ix];
        const float* __restrict__ xv = g.V + (long)ix * g.ld;
        if (lx > ml) ml = lx;                                                  // `if level > idx.maxLevel { idx.maxLevel = level }` before insertNode
        if (!has_entry) { entry = (int)ix; has_entry = 1; continue; }          // first node of an empty graph: becomes the entry point (:266-271)
        g.n = ix;                                                              // nodes visible to this insertion: everything before it
        g.entry = (unsigned)entry; g.max_level = ml;
        unsigned curr = g.entry;
        if (lane == 0) L.rows[0] = curr;
        __builtin_amdgcn_wave_barrier();
        float d0 = wave_dists<METRIC>(g.V, g.ld, xv, L.rows, 1, L.tile);
        float curr_dist = __shfl(d0, 0, 64);
        hn_descend<METRIC>(g, xv, ml, lx, curr, curr_dist, L, n_eval);
        for (int lc = lx; lc >= 0; lc--) {
            for (long w = lane; w < vwords; w += 64) vis[w] = 0u;              // a fresh visited set per searchLayer call (:566)
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            int nres = 0;
            if (hn_search_layer<METRIC>(g, xv, curr, efc, lc, vis, L, nres, n_eval, n_exp) && lane == 0) *status = 1;
            // candidates ascending = the drained result heap (:623-626); selectNeighbors keeps the first Mmax (stable order)
            const int Mmax = lc == 0 ? 2 * M : M;
            hn_drain(L.res, nres, L, [&](int i, unsigned id, float) { if (i < HN_EF_MAX + 1) sel[i] = id; });
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            const int nsel = min(nres, Mmax);
            const long sx = g.slot_base[ix] + lc; const long offx = g.edge_off[sx];
            for (int i = 0; i < nsel; i++) {
                const unsigned nb = sel[i];
                if (lane == 0) { g.edges[offx + i] = nb; }                      // node.Edges[lc] = append(node.Edges[lc], neighborID)
                if (lc <= g.level[nb]) {
                    const long sn = g.slot_base[nb] + lc; const long offn = g.edge_off[sn];
                    const int dn = deg_rw[sn];
                    if (dn < Mmax) {                                           // room: plain append of the back-edge
                        if (lane == 0) { g.edges[offn + dn] = ix; deg_rw[sn] = dn + 1; }
                    } else if (dn == Mmax && sorted[sn]) {
                        // pruneConnections on a list an earlier prune already sorted: dropping the fresh back-edge and re-sorting the
                        // same Mmax entries (stable) changes nothing — a full list only ever sees this case again
                    } else {
                        // pruneConnections(nb, lc, Mmax) with len = Mmax + 1: the fresh back-edge is not in idx.nodes yet and is dropped,
                        // the Mmax old entries are re-sorted by their distance to nb (stable: sort.Slice on distance, canonical order)
                        const float* __restrict__ nv = g.V + (long)nb * g.ld;
                        for (int b0 = 0; b0 < dn; b0 += 64) {
                            const int cnt = min(64, dn - b0);
                            unsigned e = HN_NONE;
                            if (lane < cnt) e = g.edges[offn + b0 + lane];
                            L.rows[lane] = e;
                            __builtin_amdgcn_wave_barrier();
                            const float d = wave_dists<METRIC>(g.V, g.ld, nv, L.rows, cnt, L.tile);
                            if (lane < cnt) { pd[b0 + lane] = d; pe[b0 + lane] = e; }
                            __builtin_amdgcn_wave_barrier();
                        }
                        for (int j = lane; j < dn; j += 64) {                  // rank by counting: stable ascending
                            const float dj = pd[j]; int rank = 0;
                            for (int i2 = 0; i2 < dn; i2++) { const float di = pd[i2]; rank += (di < dj || (di == dj && i2 < j)) ? 1 : 0; }
                            g.edges[offn + rank] = pe[j];
                        }
                        if (lane == 0 && dn == Mmax) sorted[sn] = 1;
                        __threadfence_block();
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                __threadfence_block();
                __builtin_amdgcn_wave_barrier();
            }
            if (lane == 0) deg_rw[sx] = nsel;
            if (nres > 0) curr = sel[0];                                       // `if len(candidates) > 0 { curr = candidates[0].id }`
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane == 0) { state[0] = ml; state[1] = entry; state[2] = has_entry; }
}

// D[q][i] = res_dist[q][i] unless the node is filtered out (hnsw_index_search.go:321-325)
__global__ __launch_bounds__(256) void hnsw_mask_kernel(const unsigned* __restrict__ res_idx, const float* __restrict__ res_dist, const int* __restrict__ res_cnt,
                                                        int ef, int B, const unsigned char* __restrict__ elig, float* __restrict__ D) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * ef) return;
    const int q = (int)(i / ef), j = (int)(i - (long)q * ef);
    float v = __uint_as_float(EXCLUDED_BITS);
    if (j < res_cnt[q]) { const unsigned n = res_idx[i]; if (!elig || elig[n]) v = res_dist[i]; }
    D[i] = v;
}

struct HNSWIndex : comet_index {
    int M = 16, efC = 200, efS = 200;
    int64_t n = 0; int max_level = -1; uint32_t entry = 0;   // entry: dense node index
    DevBuf V, ids_dev, level, slot_base, edge_off, deg_dev, edges, del_bm, state_dev, ins_vis;
    DevBuf rec0;                // per node {edge offset of the layer-0 slot, its slot index} (hnsw_rec0_kernel): the first thing a layer-0 expansion reads
    DevBuf sorted_dev;          // per edge slot: 1 once a full list has been re-sorted by pruneConnections (a further prune is then a no-op)
    int64_t n_slots = 0, edge_cap = 0;       // slots (sum of level+1) and total edge capacity in use
    bool mirror_dirty = false;               // the device graph changed (GPU insert): host mirror is rebuilt before Flush / WriteTo
    uint64_t level_rng = 0x9E3779B97F4A7C15ull;
    std::vector<uint32_t> ids; std::unordered_map<uint32_t, uint32_t> id2idx;
    // host mirror of the graph as loaded (hnswNode{Level, Edges} hnsw_index.go:50-61): Flush and WriteTo work on it
    std::vector<int32_t> h_levels; std::vector<int64_t> h_eoff; std::vector<uint32_t> h_edges;   // h_edges: neighbour NODE IDS, per (node, layer)
    uint32_t entry_id = 0;
    bool del_dirty = true;
    uint64_t st_evals = 0, st_exp = 0;
    int64_t max_list = 0;       // longest edge list / slot capacity of the graph (<= 32: the search stages 32-row tiles)
    int cand_hint = 0;          // candidate-heap capacity the last overflowing search needed (LDS sizing of the next searches)
    // Asynchronous searches (round 3): search_begin enqueues the search with the LDS heaps sized for its efSearch and copies the overflow
    // flag and the counters to pinned memory behind it; search_finish waits for the search's event and, if a query's live candidates
    // outgrew the heap (rare), runs the whole search again synchronously with the escalation of search_dev. Nothing persistent is
    // written on the device, so every other asynchronous search may use the context's second lane: at B = 256 a search occupies
    // 256 waves — one per CU — and two of them run side by side.
    struct Pending {
        bool active = false; uint64_t ticket = 0; hipEvent_t ev = nullptr;
        int B = 0, k_cap = 0; const float* queries = nullptr; comet_search_params p{}; std::vector<uint32_t> flt;
        uint32_t* out_ids = nullptr; float* out_scores = nullptr; int32_t* out_counts = nullptr;
        unsigned long long* host = nullptr;      // pinned: [0] evals, [1] expansions, [2] overflow flag
    };
    static constexpr int kRing = 8;
    Pending ring[kRing]; uint64_t next_ticket = 1;
    ~HNSWIndex() override { for (auto& r : ring) { if (r.ev) (void)hipEventDestroy(r.ev); if (r.host) (void)hipHostFree(r.host); } }
    int max_lanes() const override { return 8; }      // 8 x 256 queries = two query-waves per SIMD (18 KiB of LDS per wave: eight per CU); kRing pending searches

    int64_t size() const override { return n; }
    bool contains_id(uint32_t id) const override { return id2idx.count(id) != 0; }
    int64_t row_of_id(uint32_t id) override { auto it = id2idx.find(id); return it == id2idx.end() ? -1 : (int64_t)it->second; }
    const float* rows_dev() const override { return V.as<float>(); }
    // randomLevel hnsw_index.go:474-484: geometric(p = 1/M), capped at 16. The reference draws from the unseeded global
    // math/rand/v2, so no run of it is reproducible; this index draws from its own SplitMix64 stream (u = (next >> 11) * 2^-53).
    int random_level() {
        const double prob = 1.0 / (double)M; int lv = 0;
        while (lv < 16) {
            uint64_t z = (level_rng += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
            if (!((double)(z >> 11) * (1.0 / 9007199254740992.0) < prob)) break;
            lv++;
        }
        return lv;
    }
    const int32_t* pending_levels = nullptr;    // comet_hnsw_add_with_levels: explicit levels for the next add_dev call

    // HNSWIndex.Add hnsw_index.go:228-288 for a batch, inserted strictly in order by one wave (hnsw_insert_kernel)
    int64_t add_dev(const uint32_t*, const uint32_t* ids_h, const float* vecs_dev, int64_t m, int64_t* zero_at, float* normalized_dev) override {
        *zero_at = -1;
        if (m <= 0) return 0;
        if (2 * M > 256) COMET_FAIL(COMET_ERR_UNSUPPORTED, "GPU insertion supports M <= 128");
        if (efC > HN_EF_MAX) COMET_FAIL(COMET_ERR_UNSUPPORTED, "efConstruction %d exceeds the on-device limit %d", efC, HN_EF_MAX);
        V.reserve((size_t)(n + m) * ld * sizeof(float), c->stream, (size_t)n * ld * sizeof(float));
        float* dst = V.as<float>() + (size_t)n * ld;
        int32_t* zf = c->salloc<int32_t>(m);
        launch_ingest_rows(c, metric, vecs_dev, m, dim, dst, ld, zf);
        int64_t added = m;
        if (metric == COMET_COSINE) {
            std::vector<int32_t> h(m);
            c->d2h(h.data(), zf, m * sizeof(int32_t));
            HIP_CHECK(hipStreamSynchronize(c->stream));
            for (int64_t i = 0; i < m; i++) if (h[i]) { *zero_at = i; added = i; break; }
        }
        if (added <= 0) return 0;
        for (int64_t i = 0; i < added; i++) {
            if (ids_h[i] == 0) COMET_FAIL(COMET_ERR_UNSUPPORTED, "HNSW on the GPU needs explicit non-zero node ids (the reference auto-assigns ids for 0, hnsw_index.go:248-256)");
            if (id2idx.count(ids_h[i])) COMET_FAIL(COMET_ERR_UNSUPPORTED, "node id %u is already in the graph (re-adding an id is not supported on the GPU)", ids_h[i]);
        }
        if (normalized_dev) launch_unpad_rows(c, dst, added, ld, normalized_dev, dim);
        // levels + empty edge slots of the new nodes
        std::vector<int32_t> lv(added); std::vector<int64_t> sb(added + 1), eo; std::vector<uint32_t> nid(ids_h, ids_h + added);
        int64_t slots = n_slots, ecap = edge_cap;
        for (int64_t i = 0; i < added; i++) {
            lv[i] = pending_levels ? pending_levels[i] : random_level();
            if (lv[i] < 0 || lv[i] > 64) COMET_FAIL(COMET_ERR_INVALID_ARG, "node level %d out of range", lv[i]);
            sb[i] = slots;
            for (int l = 0; l <= lv[i]; l++) { eo.push_back(ecap); ecap += (l == 0 ? 2 * M : M); }
            slots += lv[i] + 1;
        }
        sb[added] = slots; eo.push_back(ecap);
        max_list = std::max<int64_t>(max_list, 2 * (int64_t)M);
        level.reserve((size_t)(n + added) * 4, c->stream, (size_t)n * 4);
        ids_dev.reserve((size_t)(n + added) * 4, c->stream, (size_t)n * 4);
        slot_base.reserve((size_t)(n + added + 1) * 8, c->stream, (size_t)(n + 1) * 8);
        edge_off.reserve((size_t)(slots + 1) * 8, c->stream, (size_t)(n_slots + 1) * 8);
        deg_dev.reserve((size_t)std::max<int64_t>(slots, 1) * 4, c->stream, (size_t)n_slots * 4);
        edges.reserve((size_t)std::max<int64_t>(ecap, 1) * 4, c->stream, (size_t)edge_cap * 4);
        c->h2d(level.as<int32_t>() + n, lv.data(), added * 4);
        c->h2d(ids_dev.as<uint32_t>() + n, nid.data(), added * 4);
        c->h2d(slot_base.as<int64_t>() + n, sb.data(), (added + 1) * 8);
        c->h2d(edge_off.as<int64_t>() + n_slots, eo.data(), eo.size() * 8);
        rec0.reserve((size_t)(n + added) * 16, c->stream, (size_t)n * 16);
        hnsw_rec0_kernel<<<dim3((unsigned)ceil_div(added, 256)), dim3(256), 0, c->stream>>>((const long*)slot_base.p, (const long*)edge_off.p, (long)n, (long)added, (long*)rec0.p);
        LAUNCH_CHECK();
        c->zero(deg_dev.as<int32_t>() + n_slots, (size_t)(slots - n_slots) * 4);
        sorted_dev.reserve((size_t)std::max<int64_t>(slots, 1), c->stream, (size_t)n_slots);
        c->zero(sorted_dev.as<uint8_t>() + n_slots, (size_t)(slots - n_slots));
        state_dev.reserve(16, c->stream, 0);
        const int32_t st[3] = {max_level, (int32_t)entry, (n > 0 && max_level >= 0) ? 1 : 0};
        c->h2d(state_dev.p, st, sizeof(st));
        const int64_t vwords = (n + added + 31) / 32;
        ins_vis.reserve((size_t)vwords * 4, c->stream, 0);
        int32_t* status = c->salloc<int32_t>(1);
        c->zero(status, 4);
        HIP_CHECK(hipStreamSynchronize(c->stream));          // the host vectors above are temporaries
        HnswGraph g{V.as<float>(), ld, n, level.as<int>(), (const long*)slot_base.p, (const long*)edge_off.p, deg_dev.as<int>(), edges.as<uint32_t>(), entry, max_level, deleted_bitmap()};
        const size_t lds = HN_LDS_BYTES + (HN_EF_MAX + 1 + 256 + 256) * 4;
        {
            ProfScope ps(c, "hnsw_insert");
#define HI(MT) do { HIP_CHECK(hipFuncSetAttribute((const void*)hnsw_insert_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                    hnsw_insert_kernel<MT><<<dim3(1), dim3(64), lds, c->stream>>>(g, deg_dev.as<int>(), sorted_dev.as<uint8_t>(), n, added, M, efC, ins_vis.as<uint32_t>(), vwords, state_dev.as<int>(), status); } while (0)
            switch (metric) { case COMET_L2: HI(COMET_L2); break; case COMET_L2SQ: HI(COMET_L2SQ); break; default: HI(COMET_COSINE); break; }
#undef HI
            LAUNCH_CHECK();
        }
        int32_t hst[3], hs = 0;
        c->d2h(hst, state_dev.p, sizeof(hst)); c->d2h(&hs, status, 4);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        if (hs) COMET_FAIL(COMET_ERR_UNSUPPORTED, "HNSW candidate heap overflow during insertion (more than %d live candidates): lower efConstruction", HN_CAND_CAP);
        for (int64_t i = 0; i < added; i++) { id2idx[nid[i]] = (uint32_t)(n + i); ids.push_back(nid[i]); }
        h_levels.insert(h_levels.end(), lv.begin(), lv.end());
        max_level = hst[0]; entry = (uint32_t)hst[1]; entry_id = ids[entry];
        n += added; n_slots = slots; edge_cap = ecap;
        trained = true; del_dirty = true; mirror_dirty = true;
        return added;
    }

    // rebuild the host mirror (edge lists as node ids) from the device graph after GPU insertions
    void sync_mirror() {
        if (!mirror_dirty) return;
        std::vector<int64_t> eo(n_slots + 1); std::vector<int32_t> dg(std::max<int64_t>(n_slots, 1)); std::vector<uint32_t> ed(std::max<int64_t>(edge_cap, 1));
        c->d2h(eo.data(), edge_off.p, (size_t)(n_slots + 1) * 8); c->d2h(dg.data(), deg_dev.p, (size_t)n_slots * 4); c->d2h(ed.data(), edges.p, (size_t)edge_cap * 4);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        h_eoff.assign(n_slots + 1, 0); h_edges.clear();
        for (int64_t sl = 0; sl < n_slots; sl++) {
            for (int j = 0; j < dg[sl]; j++) h_edges.push_back(ids[ed[eo[sl] + j]]);
            h_eoff[sl + 1] = (int64_t)h_edges.size();
        }
        mirror_dirty = false;
    }

    // stored (preprocessed) vectors, dense n x dim, host
    std::vector<float> download_vectors() {
        std::vector<float> out((size_t)std::max<int64_t>(n, 1) * dim);
        if (n == 0) return out;
        const int64_t chunk = 65536;
        ScratchMark sm(c);
        float* tmp = c->salloc<float>((size_t)std::min(chunk, n) * dim);
        for (int64_t r0 = 0; r0 < n; r0 += chunk) {
            const int64_t m = std::min(chunk, n - r0);
            launch_unpad_rows(c, V.as<float>() + (size_t)r0 * ld, m, ld, tmp, dim);
            c->d2h(&out[(size_t)r0 * dim], tmp, (size_t)m * dim * 4);
            HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        return out;
    }

    // HNSWIndex.Flush hnsw_index.go:348-431: drop edges to deleted nodes, re-seat the entry point, delete the nodes.
    // The reference picks the replacement entry point by iterating a Go map (undefined order); here nodes are
    // visited in ascending id order (first node at maxLevel, else the first node of the highest remaining level).
    void flush() override {
        if (deleted.empty()) return;
        sync_mirror();
        std::vector<float> vecs = download_vectors();
        std::vector<int64_t> order(n);
        for (int64_t i = 0; i < n; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return ids[a] < ids[b]; });
        std::vector<int64_t> sb(n + 1, 0);
        for (int64_t i = 0; i < n; i++) sb[i + 1] = sb[i] + h_levels[i] + 1;
        uint32_t new_entry = entry_id; int new_max = max_level;
        if (deleted.count(entry_id)) {                                        // PHASE 2 (:383-412)
            bool found = false;
            for (int64_t oi : order) if (!deleted.count(ids[oi]) && h_levels[oi] == max_level) { new_entry = ids[oi]; found = true; break; }
            if (!found) {
                int best = -1;
                for (int64_t oi : order) if (!deleted.count(ids[oi]) && h_levels[oi] > best) { best = h_levels[oi]; new_entry = ids[oi]; }
                if (best >= 0) new_max = best; else { new_entry = 0; new_max = -1; }
            }
        }
        std::vector<uint32_t> nids; std::vector<int32_t> nlev; std::vector<float> nvec; std::vector<int64_t> neoff; std::vector<uint32_t> nedges;
        neoff.push_back(0);
        for (int64_t i = 0; i < n; i++) {
            if (deleted.count(ids[i])) continue;                              // PHASE 3
            nids.push_back(ids[i]); nlev.push_back(h_levels[i]);
            nvec.insert(nvec.end(), vecs.begin() + (size_t)i * dim, vecs.begin() + (size_t)(i + 1) * dim);
            for (int l = 0; l <= h_levels[i]; l++) {                          // PHASE 1: keep an edge only if its target is not deleted
                for (int64_t e = h_eoff[sb[i] + l]; e < h_eoff[sb[i] + l + 1]; e++) if (!deleted.count(h_edges[e])) nedges.push_back(h_edges[e]);
                neoff.push_back((int64_t)nedges.size());
            }
        }
        if (nedges.empty()) nedges.push_back(0);
        load((int64_t)nids.size(), nids.data(), nlev.data(), nvec.data(), neoff.data(), nedges.data(), new_entry, new_max);
        deleted.clear(); deleted_dirty = true; del_dirty = true;               // PHASE 4
    }

    void load(int64_t nn, const uint32_t* ids_h, const int32_t* levels, const float* vecs, const int64_t* eoff, const uint32_t* edge_ids,
              uint32_t entry_id_in, int maxl) {
        std::vector<uint32_t> nid(ids_h, ids_h + nn); std::unordered_map<uint32_t, uint32_t> nmap;
        for (int64_t i = 0; i < nn; i++) nmap[nid[i]] = (uint32_t)i;
        std::vector<int64_t> sb(nn + 1); int64_t slots = 0;
        for (int64_t i = 0; i < nn; i++) { if (levels[i] < 0) COMET_FAIL(COMET_ERR_INVALID_ARG, "node %u has negative level", nid[i]); sb[i] = slots; slots += levels[i] + 1; }
        sb[nn] = slots;
        const int64_t ne = eoff[slots];
        // device copy: dense indices, duplicate neighbours inside one edge list removed (first occurrence kept). A duplicate
        // never changes the reference's result — the second occurrence is already in `visited` (hnsw_index.go:604), and in the
        // greedy descent it recomputes an equal distance — but it would make the kernel's parallel claim of `visited` unordered.
        // every slot gets the capacity insertNode needs (2M on layer 0, M above; more if the loaded list is longer)
        std::vector<uint32_t> eidx; eidx.reserve(std::max<int64_t>(ne, 1));
        std::vector<int64_t> deoff(slots + 1, 0); std::vector<int32_t> ddeg(std::max<int64_t>(slots, 1), 0);
        uint32_t new_entry = 0; int64_t new_max_list = 0;
        {
            int64_t sl = 0;
            for (int64_t i = 0; i < nn; i++)
                for (int l = 0; l <= levels[i]; l++, sl++) {
                    const size_t s0 = eidx.size();
                    for (int64_t e = eoff[sl]; e < eoff[sl + 1]; e++) {
                        auto it = nmap.find(edge_ids[e]);
                        if (it == nmap.end()) COMET_FAIL(COMET_ERR_INVALID_ARG, "edge references unknown node id %u", edge_ids[e]);
                        bool dup = false;
                        for (size_t j = s0; j < eidx.size(); j++) if (eidx[j] == it->second) { dup = true; break; }
                        if (!dup) eidx.push_back(it->second);
                    }
                    ddeg[sl] = (int32_t)(eidx.size() - s0);
                    const size_t cap = std::max<size_t>(eidx.size() - s0, (size_t)(l == 0 ? 2 * M : M));
                    eidx.resize(s0 + cap, 0u);
                    new_max_list = std::max<int64_t>(new_max_list, (int64_t)cap);
                    deoff[sl + 1] = (int64_t)eidx.size();
                }
        }
        if (eidx.empty()) eidx.push_back(0);
        if (nn > 0 && maxl >= 0) { auto it = nmap.find(entry_id_in); if (it == nmap.end()) COMET_FAIL(COMET_ERR_INVALID_ARG, "entry point %u is not a node", entry_id_in); new_entry = it->second; }
        n = nn; max_level = nn > 0 ? maxl : -1; entry = new_entry; entry_id = entry_id_in; max_list = new_max_list; cand_hint = 0;
        ids.swap(nid); id2idx.swap(nmap);
        h_levels.assign(levels, levels + nn); h_eoff.assign(eoff, eoff + slots + 1); h_edges.assign(edge_ids, edge_ids + ne);
        V.reserve(std::max<size_t>(4, (size_t)nn * ld * 4), c->stream, 0);
        {
            ScratchMark sm(c);
            float* raw = c->salloc<float>(std::max<size_t>(1, (size_t)nn * dim));
            c->h2d(raw, vecs, (size_t)nn * dim * 4);
            launch_ingest_rows(c, COMET_L2SQ, raw, nn, dim, V.as<float>(), ld, nullptr);   // stored vectors are already preprocessed
            HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        ids_dev.reserve(std::max<size_t>(4, nn * 4), c->stream, 0); level.reserve(std::max<size_t>(4, nn * 4), c->stream, 0);
        slot_base.reserve((nn + 1) * 8, c->stream, 0); edge_off.reserve((slots + 1) * 8, c->stream, 0); edges.reserve(eidx.size() * 4, c->stream, 0);
        deg_dev.reserve(ddeg.size() * 4, c->stream, 0); c->h2d(deg_dev.p, ddeg.data(), ddeg.size() * 4);
        sorted_dev.reserve(std::max<size_t>(ddeg.size(), 1), c->stream, 0); c->zero(sorted_dev.p, std::max<size_t>(ddeg.size(), 1));   // loaded lists: order unknown
        n_slots = slots; edge_cap = deoff[slots]; mirror_dirty = false;
        c->h2d(ids_dev.p, ids.data(), nn * 4); c->h2d(level.p, levels, nn * 4);
        c->h2d(slot_base.p, sb.data(), (nn + 1) * 8); c->h2d(edge_off.p, deoff.data(), (slots + 1) * 8); c->h2d(edges.p, eidx.data(), eidx.size() * 4);
        rec0.reserve(std::max<size_t>(16, nn * 16), c->stream, 0);
        if (nn > 0) { hnsw_rec0_kernel<<<dim3((unsigned)ceil_div((int64_t)nn, 256)), dim3(256), 0, c->stream>>>((const long*)slot_base.p, (const long*)edge_off.p, 0l, (long)nn, (long*)rec0.p); LAUNCH_CHECK(); }
        HIP_CHECK(hipStreamSynchronize(c->stream));
        trained = true; del_dirty = true;
    }

    // HNSWIndex.WriteTo hnsw_index.go:734-880: Flush; "HNSW", version, dim, kind, M, efConstruction, efSearch, levelMult (f64),
    // maxLevel (i32), entryPoint, node count, per node {id, level, dim, floats, layer count, per layer {count, edge ids}},
    // empty bitmap. Nodes in ascending id order (the reference ranges over a Go map).
    void write_to(Sink& s) override {
        flush();
        sync_mirror();
        write_header(s, "HNSW", dim, metric);
        s.u32((uint32_t)M); s.u32((uint32_t)efC); s.u32((uint32_t)efS);
        s.f64(1.0 / go_log((double)M));                     // levelMult = 1/ln(M) hnsw_index.go:206
        s.i32(max_level); s.u32(n > 0 ? entry_id : 0u); s.u32((uint32_t)n);
        std::vector<float> vecs = download_vectors();
        std::vector<int64_t> order(n);
        for (int64_t i = 0; i < n; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return ids[a] < ids[b]; });
        std::vector<int64_t> sb(n + 1, 0);
        for (int64_t i = 0; i < n; i++) sb[i + 1] = sb[i] + h_levels[i] + 1;
        for (int64_t i : order) {
            s.u32(ids[i]); s.i32(h_levels[i]); s.u32((uint32_t)dim); s.put(&vecs[(size_t)i * dim], (size_t)dim * 4);
            s.u32((uint32_t)(h_levels[i] + 1));
            for (int l = 0; l <= h_levels[i]; l++) {
                const int64_t e0 = h_eoff[sb[i] + l], e1 = h_eoff[sb[i] + l + 1];
                s.u32((uint32_t)(e1 - e0));
                if (e1 > e0) s.put(&h_edges[e0], (size_t)(e1 - e0) * 4);
            }
        }
        write_empty_bitmap(s);
        s.flush();
    }
    // HNSWIndex.ReadFrom hnsw_index.go:898-1096
    void read_from(Source& s) override {
        read_header(s, "HNSW", dim, metric);
        const uint32_t fM = s.u32("M"), fC = s.u32("efConstruction"), fS = s.u32("efSearch");
        check_param("M", M, fM); check_param("efConstruction", efC, fC); check_param("efSearch", efS, fS);   // :975-986
        (void)s.f64("levelMult");
        const int32_t maxl = s.i32("maxLevel");
        const uint32_t ep = s.u32("entryPoint");
        const uint32_t count = s.u32("node count");
        std::vector<uint32_t> nids(count); std::vector<int32_t> nlev(count); std::vector<float> nvec((size_t)std::max<uint32_t>(count, 1) * dim);
        std::vector<int64_t> neoff; std::vector<uint32_t> nedges;
        neoff.push_back(0);
        for (uint32_t i = 0; i < count; i++) {
            nids[i] = s.u32("node ID"); nlev[i] = s.i32("node level");
            const uint32_t vd = s.u32("vector dimension");
            if ((int)vd != dim) COMET_FAIL(COMET_ERR_FORMAT, "node %u has dimension %u, expected %d", nids[i], vd, dim);
            s.get(&nvec[(size_t)i * dim], (size_t)dim * 4, "vector data");
            const uint32_t layers = s.u32("edge layer count");
            if (nlev[i] < 0 || nlev[i] > 64 || (int64_t)layers != (int64_t)nlev[i] + 1) COMET_FAIL(COMET_ERR_FORMAT, "node %u: %u edge layers for level %d", nids[i], layers, nlev[i]);
            for (uint32_t l = 0; l < layers; l++) {
                const uint32_t ec = s.u32("edge count");
                if (ec > (1u << 24)) COMET_FAIL(COMET_ERR_FORMAT, "node %u layer %u: edge count %u is not plausible", nids[i], l, ec);
                const size_t o = nedges.size(); nedges.resize(o + ec);
                s.get(nedges.data() + o, (size_t)ec * 4, "edge ids");
                neoff.push_back((int64_t)nedges.size());
            }
        }
        const std::vector<uint32_t> del = read_bitmap(s);
        if (nedges.empty()) nedges.push_back(0);
        load(count, nids.data(), nlev.data(), nvec.data(), neoff.data(), nedges.data(), ep, maxl);
        deleted.clear(); deleted.insert(del.begin(), del.end()); deleted_dirty = true; del_dirty = true;
    }

    const uint32_t* deleted_bitmap() {
        if (deleted.empty()) return nullptr;
        if (del_dirty || deleted_dirty) {
            c->quiesce_all();      // a search in flight on either lane may still be reading the bitmap that is about to be replaced
            std::vector<uint32_t> bm((n + 31) / 32, 0);
            for (uint32_t id : deleted) { auto it = id2idx.find(id); if (it != id2idx.end()) bm[it->second >> 5] |= 1u << (it->second & 31); }
            del_bm.reserve(bm.size() * 4, c->stream, 0);
            c->h2d(del_bm.p, bm.data(), bm.size() * 4);
            HIP_CHECK(hipStreamSynchronize(c->stream));
            del_dirty = false; deleted_dirty = false;       // HNSW consumes the soft-delete set only through this bitmap
        }
        return del_bm.as<uint32_t>();
    }
    // hnswIndexSearch.searchSingleQuery hnsw_index_search.go:248-354
    void search_dev(const float* queries_dev, int B, const comet_search_params& p, uint32_t* out_ids, float* out_scores,
                    int32_t* out_counts, int k_cap) override { search_impl(queries_dev, B, p, out_ids, out_scores, out_counts, k_cap, nullptr); }
    uint64_t search_begin(const float* queries_dev, int B, const comet_search_params& p, uint32_t* out_ids, float* out_scores,
                          int32_t* out_counts, int k_cap) override {
        Pending* slot = nullptr;
        for (auto& r : ring) if (!r.active) { slot = &r; break; }
        if (!slot) { slot = &ring[0]; for (auto& r : ring) if (r.ticket < slot->ticket) slot = &r; search_finish(slot->ticket); }
        if (!slot->ev) HIP_CHECK(hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming));
        if (!slot->host) HIP_CHECK(hipHostMalloc((void**)&slot->host, 32, hipHostMallocDefault));
        slot->ticket = next_ticket++; slot->B = B; slot->k_cap = k_cap; slot->queries = queries_dev; slot->p = p; slot->flt.clear();
        if (p.filter_ids && p.n_filter > 0) { slot->flt.assign(p.filter_ids, p.filter_ids + p.n_filter); slot->p.filter_ids = slot->flt.data(); }
        slot->out_ids = out_ids; slot->out_scores = out_scores; slot->out_counts = out_counts;
        slot->host[0] = slot->host[1] = slot->host[2] = 0;
        search_impl(queries_dev, B, p, out_ids, out_scores, out_counts, k_cap, slot->host);
        HIP_CHECK(hipEventRecord(slot->ev, c->stream));
        slot->active = true;
        return slot->ticket;
    }
    bool search_finish(uint64_t ticket) override {
        Pending* slot = nullptr;
        for (auto& r : ring) if (r.active && r.ticket == ticket) { slot = &r; break; }
        if (!slot) return false;
        HIP_CHECK(hipEventSynchronize(slot->ev));
        slot->active = false;
        st_evals = slot->host[0]; st_exp = slot->host[1];
        if (!slot->host[2]) return false;
        ScratchMark sm(c);       // a candidate heap overflowed: the search again, synchronously, with the larger heaps
        search_impl(slot->queries, slot->B, slot->p, slot->out_ids, slot->out_scores, slot->out_counts, slot->k_cap, nullptr);
        return true;
    }
    // deferred != nullptr: no host round trip — the overflow flag and the counters are copied to `deferred` (pinned) behind the search
    void search_impl(const float* queries_dev, int B, const comet_search_params& p, uint32_t* out_ids, float* out_scores,
                     int32_t* out_counts, int k_cap, unsigned long long* deferred) {
        float* Qp; int32_t* zflag;
        prepare_queries(c, metric, queries_dev, B, dim, ld, &Qp, &zflag, true);
        uint32_t* pos = c->salloc<uint32_t>((size_t)B * k_cap);
        if (n == 0 || max_level == -1) {      // empty graph -> [] (:258)
            launch_select_topk(c, nullptr, 0, B, 0, nullptr, 0.0f, p.k, pos, out_scores, out_counts, k_cap);
            launch_finalize(c, nullptr, pos, B, k_cap, zflag, out_ids, out_counts);
            return;
        }
        const int ef = p.ef_search > 0 ? p.ef_search : efS;    // :302-305
        // the result heap never holds more than min(ef, live nodes) entries, the candidate heap never more than n (a node is pushed once)
        const int ef_ld = (int)std::min<int64_t>(ef, n);
        const int64_t vwords = (n + 31) / 32;
        uint32_t* vis = c->salloc<uint32_t>((size_t)B * vwords);
        uint32_t* res_idx = c->salloc<uint32_t>((size_t)B * ef_ld);
        float* res_dist = c->salloc<float>((size_t)B * ef_ld);
        int32_t* res_cnt = c->salloc<int32_t>(B);
        int32_t* status = c->salloc<int32_t>(1);
        unsigned long long* st = c->salloc<unsigned long long>(2);
        c->zero(st, 16);
        HnswGraph g{V.as<float>(), ld, n, level.as<int>(), (const long*)slot_base.p, (const long*)edge_off.p, deg_dev.as<int>(), edges.as<uint32_t>(), entry, max_level, deleted_bitmap()};
        g.rec0 = (const long*)rec0.p; g.n_edges = (long)std::max<int64_t>(edge_cap, 1);
        // heaps in LDS (ef <= 1024, <= 4096 live candidates) or, beyond that, in HBM: one slab per query, queries in sub-batches of <= 2 GiB.
        // The LDS heaps are sized for this search: result heap ef + 1, candidate heap 16 ef (>= 1024) first — 35 KiB per wave at ef 128,
        // four waves per CU instead of two — and the full 4096 only if a query's live candidates outgrow that (the batch is re-run).
        const int res_cap = (int)std::min<int64_t>(HN_EF_MAX, ef_ld) + 1;
        // LDS per wave decides how many searches a CU holds (a lone wave issues one instruction per ~8 clocks: two waves per SIMD nearly double a large
        // batch's rate). Candidate heap: 8 ef (>= 1024) entries to start with, doubled — and remembered by the index — when a search outgrows it (a graph
        // whose searches run ~efSearch expansions keeps ~12 ef live candidates); staging tile: 32 rows when no edge list of the graph is longer.
        // 18 KiB per wave at efSearch 128 on a graph built here (M 16): 8 waves per CU; 26 KiB after one doubling: 6.
        int cand_cap = (int)std::min<int64_t>(HN_CAND_CAP, std::max<int64_t>(std::max<int64_t>(1024, 8 * (int64_t)ef), cand_hint));
        const int tile_rows = max_list <= 32 ? 32 : 64;
        auto run = [&](bool spill) {
            const size_t lds = spill ? hn_lds_bytes(64, 64, tile_rows) : hn_lds_bytes(cand_cap, res_cap, tile_rows);
            c->zero(vis, (size_t)B * vwords * 4); c->zero(status, 4);
            ScratchMark mark(c);
            const int64_t s_cand = n, s_res = (int64_t)ef_ld + 1;
            const int bs = spill ? (int)std::max<int64_t>(1, std::min<int64_t>(B, ((int64_t)2 << 30) / ((s_cand + s_res) * (int64_t)sizeof(HC)))) : B;
            HC* slab = spill ? c->salloc<HC>((size_t)bs * (s_cand + s_res)) : nullptr;
            ProfScope ps(c, spill ? "hnsw_search_spill" : "hnsw_search");
            for (int b0 = 0; b0 < B; b0 += bs) {
                const int bn = std::min(bs, B - b0);
                // the attribute is per function and process-wide: always the maximum any launch may ask for (two contexts on two threads may interleave
                // "set" and "launch"); only the launch parameter varies
#define HS(MT) do { HIP_CHECK(hipFuncSetAttribute((const void*)hnsw_search_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HN_LDS_BYTES)); \
                    hnsw_search_kernel<MT><<<dim3(bn), dim3(64), lds, c->stream>>>(g, Qp + (size_t)b0 * ld, ef, vis + (size_t)b0 * vwords, vwords, res_idx + (size_t)b0 * ef_ld, \
                        res_dist + (size_t)b0 * ef_ld, res_cnt + b0, status, st, slab, s_cand, s_res, ef_ld, spill ? 64 : cand_cap, spill ? 64 : res_cap, tile_rows); } while (0)
                switch (metric) { case COMET_L2: HS(COMET_L2); break; case COMET_L2SQ: HS(COMET_L2SQ); break; default: HS(COMET_COSINE); break; }
#undef HS
                LAUNCH_CHECK();
            }
        };
        bool spill = ef > HN_EF_MAX;
        run(spill);
        while (!spill && !deferred) {      // more live candidates than the LDS heap holds: the same search again with the full-size heap, then with the heaps in HBM
            int32_t hs = 0;
            c->d2h(&hs, status, 4);
            HIP_CHECK(hipStreamSynchronize(c->stream));
            if (!hs) break;
            c->zero(st, 16);
            if (cand_cap < HN_CAND_CAP) { cand_cap = std::min(HN_CAND_CAP, 2 * cand_cap); cand_hint = cand_cap; } else spill = true;
            run(spill);
        }
        // phase 3: document filter + threshold applied AFTER the search (can return < k), sort, top-k (:321-351)
        const uint8_t* elig = nullptr;
        int nf = 0;
        const uint32_t* flt = filter_sorted_scratch(p, &nf);
        if (nf > 0) { uint8_t* e = c->salloc<uint8_t>(n); launch_build_elig(c, ids_dev.as<uint32_t>(), n, nullptr, 0, flt, nf, e); elig = e; }
        float* D = c->salloc<float>((size_t)B * ef_ld);
        hnsw_mask_kernel<<<dim3((unsigned)ceil_div((int64_t)B * ef_ld, 256)), dim3(256), 0, c->stream>>>(res_idx, res_dist, res_cnt, ef_ld, B, elig, D);
        LAUNCH_CHECK();
        uint32_t* pos2 = c->salloc<uint32_t>((size_t)B * k_cap);
        launch_select_topk(c, D, ef_ld, B, ef_ld, res_cnt, p.threshold, p.k, pos2, out_scores, out_counts, k_cap);
        launch_gather_indirect(c, res_idx, ef_ld, pos2, B, k_cap, pos);
        launch_finalize(c, ids_dev.as<uint32_t>(), pos, B, k_cap, zflag, out_ids, out_counts);
        if (deferred) {          // counters and overflow flag follow the search to pinned memory; search_finish reads them
            HIP_CHECK(hipMemcpyAsync(deferred, st, 16, hipMemcpyDeviceToHost, c->stream));
            if (!spill) HIP_CHECK(hipMemcpyAsync(deferred + 2, status, 4, hipMemcpyDeviceToHost, c->stream));
            return;
        }
        unsigned long long hst[2];
        c->d2h(hst, st, 16);
        HIP_CHECK(hipStreamSynchronize(c->stream));
#ifdef HN_TRACE
        {
            static unsigned long long* tb = nullptr;
            if (!tb) { HIP_CHECK(hipMalloc(&tb, 64)); HIP_CHECK(hipMemset(tb, 0, 64)); HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_hn_trace), &tb, sizeof(tb))); }
            else {
                unsigned long long h[8]; HIP_CHECK(hipMemcpy(h, tb, 64, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemset(tb, 0, 64));
                const double ex = (double)hst[1];
                fprintf(stderr, "[hnsw trace] ticks per expansion: pop %.0f  record+edges+visited-issue %.0f  distances %.0f  replay %.0f  loop %.0f | inside distances: row fetch wait %.0f  slices %.0f   (%.0f expansions, %.0f evals)\n",
                        h[0] / ex, h[1] / ex, h[2] / ex, h[3] / ex, h[5] / ex, h[6] / ex, h[7] / ex, ex, (double)hst[0]);
            }
        }
#endif
        st_evals = hst[0]; st_exp = hst[1];
    }
    bool get_stat(const char* name, double* out) const override {
        std::string k(name);
        if (k == "hnsw_distance_evals") *out = (double)st_evals;
        else if (k == "hnsw_expansions") *out = (double)st_exp;
        else return false;
        return true;
    }
};

}  // namespace comet

using namespace comet;

extern "C" {

// NewHNSWIndex hnsw_index.go:160-202 (defaults: M 16, efConstruction 200, efSearch = efConstruction)
int comet_hnsw_create(comet_ctx* c, int dim, int metric, int m, int ef_construction, int ef_search, comet_index** out) {
    return guarded([&] {
        *out = nullptr;
        if (dim <= 0) COMET_FAIL(COMET_ERR_INVALID_ARG, "dimension must be positive");
        if (metric < COMET_L2 || metric > COMET_COSINE) COMET_FAIL(COMET_ERR_UNKNOWN_METRIC, "unknown distance kind");
        if (m <= 0) m = 16;
        if (ef_construction <= 0) ef_construction = 200;
        if (ef_search <= 0) ef_search = ef_construction;
        c->bind();
        auto* h = new HNSWIndex();
        h->c = c; h->kind = COMET_KIND_HNSW; h->dim = dim; h->ld = padded_dim(dim); h->metric = metric; h->M = m; h->efC = ef_construction; h->efS = ef_search;
        h->trained = true;
        *out = h;
        return (int)COMET_OK;
    });
}

// Add with explicit node levels (hnswNode.Level): what Add() does when randomLevel() returned levels[i] for vector i.
int comet_hnsw_add_with_levels(comet_index* idx, const uint32_t* ids, const float* vecs, const int32_t* levels, int64_t n, int64_t* out_added) {
    return guarded([&] {
        if (out_added) *out_added = 0;
        if (idx->kind != COMET_KIND_HNSW) COMET_FAIL(COMET_ERR_INVALID_ARG, "not an HNSW index");
        if (n <= 0) return (int)COMET_OK;
        auto* h = static_cast<HNSWIndex*>(idx);
        h->pending_levels = levels;
        const int rc = comet_index_add(idx, ids, vecs, n, out_added, nullptr);
        h->pending_levels = nullptr;
        return rc;
    });
}
// seed of the index's own level stream (the reference's is the unseeded global math/rand/v2)
int comet_hnsw_set_level_seed(comet_index* idx, uint64_t seed) {
    return guarded([&] {
        if (idx->kind != COMET_KIND_HNSW) COMET_FAIL(COMET_ERR_INVALID_ARG, "not an HNSW index");
        static_cast<HNSWIndex*>(idx)->level_rng = seed;
        return (int)COMET_OK;
    });
}
// Export the graph as comet_hnsw_load_graph takes it: two-call protocol (NULL arrays -> sizes only).
int comet_hnsw_export_graph(comet_index* idx, int64_t* out_n, int64_t* out_slots, int64_t* out_edges, uint32_t* ids, int32_t* levels, float* vecs,
                            int64_t* edge_offsets, uint32_t* edges, uint32_t* out_entry_id, int32_t* out_max_level) {
    return guarded([&] {
        if (idx->kind != COMET_KIND_HNSW) COMET_FAIL(COMET_ERR_INVALID_ARG, "not an HNSW index");
        auto* h = static_cast<HNSWIndex*>(idx);
        Ctx* c = idx->c; std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind(); c->scratch_reset();
        h->sync_mirror();
        if (out_n) *out_n = h->n;
        if (out_slots) *out_slots = (int64_t)h->h_eoff.size() - 1;
        if (out_edges) *out_edges = (int64_t)h->h_edges.size();
        if (out_entry_id) *out_entry_id = h->n > 0 ? h->entry_id : 0;
        if (out_max_level) *out_max_level = h->max_level;
        if (ids) std::copy(h->ids.begin(), h->ids.end(), ids);
        if (levels) std::copy(h->h_levels.begin(), h->h_levels.end(), levels);
        if (edge_offsets) std::copy(h->h_eoff.begin(), h->h_eoff.end(), edge_offsets);
        if (edges) std::copy(h->h_edges.begin(), h->h_edges.end(), edges);
        if (vecs && h->n > 0) { std::vector<float> v = h->download_vectors(); std::copy(v.begin(), v.begin() + (size_t)h->n * h->dim, vecs); }
        return (int)COMET_OK;
    });
}

int comet_hnsw_load_graph(comet_index* idx, int64_t n, const uint32_t* ids, const int32_t* levels, const float* vecs,
                          const int64_t* edge_offsets, const uint32_t* edges, uint32_t entry_id, int32_t max_level) {
    return guarded([&] {
        if (idx->kind != COMET_KIND_HNSW) COMET_FAIL(COMET_ERR_INVALID_ARG, "not an HNSW index");
        Ctx* c = idx->c;
        std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind(); c->scratch_reset();
        static_cast<HNSWIndex*>(idx)->load(n, ids, levels, vecs, edge_offsets, edges, entry_id, max_level);
        return (int)COMET_OK;
    });
}

}  // extern "C"
