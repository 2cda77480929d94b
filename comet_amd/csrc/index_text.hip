// index_text.hip — BM25 on the GPU (reference: bm25_index.go, bm25_index_search.go:278-397).
//
// Tokenisation / NFKC / lower-casing stay in Go (third-party uax29, x/text): documents and queries
// arrive as token ids. The host keeps the reference's structures (postings in ascending doc-id order,
// term frequencies, document lengths, running average); a CSR snapshot lives in HBM. Scoring is
// term-at-a-time in float64 exactly like the reference: one launch per query-token POSITION, so the
// accumulation order `scores[doc] += score` (bm25_index_search.go:325) is the query-token order, and
// within one launch every posting of a term touches a different document (no atomics, no reordering).
// Selection keeps the reference's "top-k by float64 score, descending" with the canonical tie order
// (ascending doc id) in one workgroup per query: 8 radix passes on the order-preserving 64-bit key.
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <algorithm>
#include <cmath>

#include "index.hpp"

namespace comet {

// ---- scoring: one launch per query-token position ---------------------------------------------------
// The accumulator is a dense float64 row per query, zeroed before the first launch: a posting that finds 0.0 is the
// document's first touch (scores are > 0: idf = ln(1 + ...) > 0, tf > 0) and appends it to the query's TOUCHED LIST; selection
// walks that list only, never the row.
constexpr int BM_TCOUNT_STRIDE = 32;      // ints between two queries' touched counters: one 128-byte line each
struct TermRef { int off; int df; double idf; };   // per (query, position); df == 0 -> no such term / padding

__global__ __launch_bounds__(256) void bm25_score_kernel(const TermRef* __restrict__ refs /*[B]*/, const int* __restrict__ post_doc,
                                                         const int* __restrict__ post_tf, const double* __restrict__ kin,
                                                         const unsigned char* __restrict__ elig,
                                                         double* __restrict__ acc, long nd, int* __restrict__ touched, long tcap, int* __restrict__ tcount) {
    const int q = blockIdx.y;
    const TermRef r = refs[q];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < r.df;
    if ((int)(blockIdx.x * blockDim.x) >= r.df) return;       // the whole workgroup is past the posting list (uniform: barriers below)
    const int doc = live ? post_doc[r.off + i] : 0;
    const bool ok = live && !(elig && !elig[doc]);        // deleted / filtered documents are skipped (:312-319)
    const double tfv = ok ? (double)post_tf[r.off + i] : 1.0;
    // score := idf * (tfVal * (K1 + 1)) / (tfVal + K1*(1-B+B*(docLen/avgDocLen)))  (:321-324), K1=1.2, B=0.75;
    // the untyped constants K1+1 and 1-B fold exactly to 2.2 and 0.25. Compiled with -ffp-contract=off. The document's part of the denominator,
    // kin[doc] = 1.2 * (0.25 + 0.75 * (docLen / avgDocLen)), is formed once per document (compile()) with these very operations.
    const double den = tfv + kin[doc];
    const double num = r.idf * (tfv * 2.2);
    const double score = num / den;
    const double old = ok ? acc[(long)q * nd + doc] : 1.0;     // one thread per (query, document) in a launch: no race
    // First touches take their slots with ONE atomic per workgroup, on a counter that has a 128-byte line to itself: a frequent
    // token touches tens of thousands of documents of a query, and returning atomics on counters that share cache lines are
    // serialised by the L2 (measured: one atomic per wave on packed counters made the launch 14x slower, 0.08 -> 1.1 ms).
    __shared__ int s_wcnt[4], s_base;
    const bool first = ok && old == 0.0;
    const unsigned long long m = __ballot(first);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_wcnt[w] = (int)__builtin_popcountll(m);
    __syncthreads();
    if (threadIdx.x == 0) { const int tot = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3]; s_base = tot ? atomicAdd(&tcount[q * BM_TCOUNT_STRIDE], tot) : 0; }
    __syncthreads();
    if (first) {
        int s = s_base + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        for (int j = 0; j < w; j++) s += s_wcnt[j];
        if (s < tcap) touched[(long)q * tcap + s] = doc;
    }
    if (ok) acc[(long)q * nd + doc] = old + score;
}

// ---- document-major scoring for queries whose terms are all HOT (round 5) -------------------------------
// The term-at-a-time launches above are a read-modify-write stream over the accumulator: a frequent term touches most documents of every query that holds
// it (Zipf-like vocabularies: a three-term query touches ~50 % of 100 k documents, 38 M float64 updates per batch of 256 = 0.38 ms). For the terms with
// df >= n_docs / 64 the index keeps a DENSE tf column (uint16 per document; compile()), and a query all of whose known terms are hot is scored by one thread
// per document: its terms in token order (the reference's order of additions: scores[docID] += score, bm25_index_search.go:325, term by term), each
// `idf * (tf * 2.2) / (tf + kin[doc])` exactly as above (kin[doc] = 1.2 * (0.25 + 0.75 * (docLen / avgDocLen)), formed once per document with the same
// operations), one write per touched document, no read, no zeroed row. Unknown terms contribute nothing, as in the reference (:301-304).
struct DenseRef { int h; int pad; double idf; };     // h: hot-column index, -1: the term has no postings
__global__ __launch_bounds__(256) void bm25_dense_kernel(const int* __restrict__ qrow /*[n]: local query (row) index*/, const DenseRef* __restrict__ refs /*[n][maxlen]*/, int maxlen,
                                                         const unsigned short* __restrict__ tfcol, const double* __restrict__ kin, const unsigned char* __restrict__ elig,
                                                         double* __restrict__ acc, long nd, int* __restrict__ touched, long tcap, int* __restrict__ tcount) {
    const int q = qrow[blockIdx.y];
    const long doc = (long)blockIdx.x * 256 + threadIdx.x;
    const bool ok = doc < nd && !(elig && !elig[doc]);
    const DenseRef* r = refs + (long)blockIdx.y * maxlen;
    double score = 0.0;
    if (ok) {
        const double kd = kin[doc];
        for (int j = 0; j < maxlen; j++) {
            const int h = r[j].h;
            if (h < 0) continue;
            const unsigned tf = tfcol[(long)h * nd + doc];
            if (!tf) continue;
            const double tfv = (double)tf;
            const double den = tfv + kd;
            const double num = r[j].idf * (tfv * 2.2);
            score = score + num / den;
        }
    }
    __shared__ int s_wcnt[4], s_base;
    const bool first = ok && score != 0.0;               // (scores are > 0: idf = ln(1 + ...) > 0, tf > 0)
    const unsigned long long m = __ballot(first);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_wcnt[w] = (int)__builtin_popcountll(m);
    __syncthreads();
    if (threadIdx.x == 0) { const int tot = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3]; s_base = tot ? atomicAdd(&tcount[q * BM_TCOUNT_STRIDE], tot) : 0; }
    __syncthreads();
    if (first) {
        int sl = s_base + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        for (int j = 0; j < w; j++) sl += s_wcnt[j];
        if (sl < tcap) touched[(long)q * tcap + sl] = (int)doc;
    }
    if (doc < nd) acc[(long)q * nd + doc] = score;     // every entry of the row, zeros included: the top-K walks the ROW, not the touched list, when most documents were touched
}
// the accumulator rows of the queries that take the term-at-a-time path (the document-major kernel writes every entry of its rows itself)
__global__ __launch_bounds__(256) void bm25_zero_rows_kernel(const int* __restrict__ qrow, double* __restrict__ acc, long nd) {
    double* row = acc + (long)qrow[blockIdx.y] * nd;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nd; i += (long)gridDim.x * 256) row[i] = 0.0;
}

// ---- top-k by (score desc, doc index asc) on float64 ------------------------------------------------
// key = ~ordered(score): ascending key == descending score
__device__ __forceinline__ unsigned long long d2key_desc(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    u = (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
    return ~u;
}
__device__ __forceinline__ double key2d_desc(unsigned long long k) {
    unsigned long long u = ~k;
    u = (u & 0x8000000000000000ull) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
    return __longlong_as_double((long long)u);
}
constexpr int BM_THREADS = 1024;
constexpr int BM_KMAX = 2048;        // selections up to this size sort in LDS, larger ones in a slab in HBM
constexpr int BM_SMALL = 1024;       // members of the K-th key's radix bin that are ranked directly (<= BM_KMAX: they borrow the LDS sort buffer)
struct KP { unsigned long long key; unsigned pos; unsigned pad; };

// One workgroup per query. The touched list is unordered, so the reference's canonical order (score descending, then
// document index ascending) is a radix select on the 96-bit composite (key, doc index): 8 byte-passes over the keys, 4 over the
// doc indices of the documents that tie with the k-th key. Then exactly kq composites are gathered and sorted.
__global__ __launch_bounds__(BM_THREADS) void bm25_topk_kernel(const double* __restrict__ acc, long nd, const int* __restrict__ touched, long tcap,
                                                               const int* __restrict__ tcount, unsigned long long* __restrict__ tkeys, int K,
                                                               const unsigned* __restrict__ doc_ids, KP* __restrict__ slab, long slab_ld,
                                                               unsigned* __restrict__ out_ids, float* __restrict__ out_scores,
                                                               double* __restrict__ out_scores64, int* __restrict__ out_counts, int k_cap) {
    __shared__ unsigned hist[256];
    __shared__ int s_bin, s_before, s_n;
    __shared__ unsigned long long s_key; __shared__ unsigned s_pos;
    __shared__ KP sel_lds[BM_KMAX];
    const int q = blockIdx.x, t = threadIdx.x;
    const double* row = acc + (long)q * nd;
    const int* tl = touched + (long)q * tcap;
    unsigned long long* tk = tkeys + (long)q * tcap;
    const int total = (int)min((long)tcount[q * BM_TCOUNT_STRIDE], tcap);
    const int kq_all = (K <= 0 || K >= total) ? total : K;          // `k <= 0 || k >= len(scores)` -> all (:330)
    const int kq = kq_all < k_cap ? kq_all : k_cap;                 // what the caller's rows can hold
    KP* sel = (kq > BM_KMAX) ? slab + (long)q * slab_ld : sel_lds;
    // Dense form, for a query that touched a good part of the collection and wants few results (a frequent token touches most
    // documents: 99 of 100 k in the bench): the touched list, its gather of 8-byte accumulators and the 64-bit keys written to and
    // re-read from HBM by every radix pass below (256 queries x 100 k x 12 bytes do not fit in any cache: ~0.08 ms per pass, five
    // passes) are replaced by two coalesced walks over the accumulator row itself — untouched documents hold exactly 0.0, touched
    // ones a score > 0. Walk 1: every thread's smallest composite (key, document index); the kq-th smallest of the 1024 minima —
    // distinct documents — bounds the kq-th smallest composite of the row. Walk 2: the composites at or under the bound (a few dozen)
    // are collected, sorted, and the first kq are the result. More than BM_SMALL of them (mass ties): the radix form below.
    bool done = false;
    if (kq > 0 && kq <= 64 && (long)total * 16 >= nd && nd >= BM_THREADS) {
        KP mine; mine.key = ~0ull; mine.pos = 0xFFFFFFFFu; mine.pad = 0;
        // both walks read the row as 16-byte pairs, four pairs per thread in flight (one 8-byte load per trip of a branchy loop left a workgroup with a
        // single load per thread outstanding: 2.9 TB/s over the two walks); of equal keys the lowest document index stays
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const long npair = nd >> 1;
        auto walk = [&](auto&& f) {
            for (long p0 = 0; p0 < npair; p0 += 4 * BM_THREADS) {
                f64x2 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const long pr = p0 + (long)u * BM_THREADS + t; v[u] = pr < npair ? *reinterpret_cast<const f64x2*>(row + 2 * pr) : f64x2{0.0, 0.0}; }
#pragma unroll
                for (int u = 0; u < 4; u++) { const long pr = p0 + (long)u * BM_THREADS + t; f(v[u][0], 2 * pr); f(v[u][1], 2 * pr + 1); }
            }
            if ((nd & 1) && t == 0) f(row[nd - 1], nd - 1);
        };
        walk([&](double v, long i) {
            if (v != 0.0) { const unsigned long long k = d2key_desc(v); if (k < mine.key || (k == mine.key && (unsigned)i < mine.pos)) { mine.key = k; mine.pos = (unsigned)i; } }
        });
        sel_lds[t] = mine;
        if (t == 0) s_n = 0;
        __syncthreads();
        auto sort_lds = [&](KP* a, int n2) {                        // bitonic, ascending by (key, pos)
            for (int k2 = 2; k2 <= n2; k2 <<= 1)
                for (int j = k2 >> 1; j > 0; j >>= 1) {
                    for (int i = t; i < n2; i += BM_THREADS) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            KP x = a[i], y = a[ixj];
                            const bool gt = (x.key > y.key) || (x.key == y.key && x.pos > y.pos);
                            if (gt == ((i & k2) == 0)) { a[i] = y; a[ixj] = x; }
                        }
                    }
                    __syncthreads();
                }
        };
        sort_lds(sel_lds, BM_THREADS);
        const KP bnd = sel_lds[kq - 1];                              // (an unused thread's ~0 minimum here: everything passes, the count below decides)
        __syncthreads();
        KP* col = sel_lds + BM_THREADS;                              // BM_KMAX - BM_THREADS = BM_SMALL slots
        walk([&](double v, long i) {
            if (v != 0.0) {
                const unsigned long long k = d2key_desc(v);
                if (k < bnd.key || (k == bnd.key && (unsigned)i <= bnd.pos)) { const int sl = atomicAdd(&s_n, 1); if (sl < BM_SMALL) { col[sl].key = k; col[sl].pos = (unsigned)i; col[sl].pad = 0; } }
            }
        });
        __syncthreads();
        const int members = s_n;
        if (members <= BM_SMALL) {                                   // (>= kq: the kq smallest minima are among them)
            int n2 = 1; while (n2 < members) n2 <<= 1;
            for (int i = members + t; i < n2; i += BM_THREADS) { col[i].key = ~0ull; col[i].pos = 0xFFFFFFFFu; }
            __syncthreads();
            sort_lds(col, n2);
            if (t < kq) { const KP x = col[t]; sel_lds[t] = x; }     // kq <= 64 < BM_THREADS: no overlap between source and destination
            __syncthreads();
            done = true;
        }
        __syncthreads();
    }
    if (kq > 0 && !done) {
    // pass 0: keys of the touched documents
    for (int i = t; i < total; i += BM_THREADS) tk[i] = d2key_desc(row[tl[i]]);
    __syncthreads();
        // Byte-wise radix select on the keys; as soon as the bin that holds the kq-th composite has <= BM_SMALL members they are
        // ranked directly in LDS as (key, doc index) composites — usually after two or three passes instead of 8 + 4.
        unsigned long long prefix = 0, mask = 0; int remaining = kq;
        unsigned long long keystar = 0; unsigned posstar = 0; bool found = false;
        for (int shift = 56; shift >= 0 && !found; shift -= 8) {
            if (t < 256) hist[t] = 0;
            __syncthreads();
            for (int i = t; i < total; i += BM_THREADS) { const unsigned long long k = tk[i]; if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255ull], 1u); }
            __syncthreads();
            if (t == 0) {
                int run = 0, b = 0;
                for (b = 0; b < 256; b++) { if (run + (int)hist[b] >= remaining) break; run += (int)hist[b]; }
                s_bin = b; s_before = run; s_n = 0;
            }
            __syncthreads();
            const int members = (int)hist[s_bin];
            prefix |= ((unsigned long long)s_bin) << shift; mask |= 255ull << shift; remaining -= s_before;
            __syncthreads();
            if (members <= BM_SMALL) {
                for (int i = t; i < total; i += BM_THREADS) { const unsigned long long k = tk[i]; if ((k & mask) == prefix) { const int sl = atomicAdd(&s_n, 1); sel_lds[sl].key = k; sel_lds[sl].pos = (unsigned)tl[i]; } }
                __syncthreads();
                if (t < members) {
                    const KP me = sel_lds[t]; int less = 0;
                    for (int j = 0; j < members; j++) { const KP o = sel_lds[j]; less += (o.key < me.key || (o.key == me.key && o.pos < me.pos)) ? 1 : 0; }
                    if (less == remaining - 1) { s_key = me.key; s_pos = me.pos; }          // exactly one thread
                }
                __syncthreads();
                keystar = s_key; posstar = s_pos; found = true;
                __syncthreads();
            }
        }
        if (!found) {
            // every key byte decided and the tie at key* is still large: `remaining` documents with key == key* are taken, lowest
            // document index first — select on the index
            keystar = prefix;
            unsigned pprefix = 0, pmask = 0;
            for (int shift = 24; shift >= 0; shift -= 8) {
                if (t < 256) hist[t] = 0;
                __syncthreads();
                for (int i = t; i < total; i += BM_THREADS) { if (tk[i] == keystar) { const unsigned p = (unsigned)tl[i]; if ((p & pmask) == pprefix) atomicAdd(&hist[(p >> shift) & 255u], 1u); } }
                __syncthreads();
                if (t == 0) {
                    int run = 0, b = 0;
                    for (b = 0; b < 256; b++) { if (run + (int)hist[b] >= remaining) break; run += (int)hist[b]; }
                    s_bin = b; s_before = run;
                }
                __syncthreads();
                pprefix |= ((unsigned)s_bin) << shift; pmask |= 255u << shift; remaining -= s_before;
                __syncthreads();
            }
            posstar = pprefix;
        }
        if (t == 0) s_n = 0;
        __syncthreads();
        for (int i = t; i < total; i += BM_THREADS) {
            const unsigned long long k = tk[i]; const unsigned p = (unsigned)tl[i];
            if (k < keystar || (k == keystar && p <= posstar)) { const int s = atomicAdd(&s_n, 1); sel[s].key = k; sel[s].pos = p; }
        }
        __syncthreads();
        // sort the kq selected (key, pos) pairs
        int n2 = 1; while (n2 < kq) n2 <<= 1;
        for (int i = kq + t; i < n2; i += BM_THREADS) { sel[i].key = ~0ull; sel[i].pos = 0xFFFFFFFFu; }
        __syncthreads();
        for (int k2 = 2; k2 <= n2; k2 <<= 1) {
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int i = t; i < n2; i += BM_THREADS) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        KP a = sel[i], b = sel[ixj];
                        const bool gt = (a.key > b.key) || (a.key == b.key && a.pos > b.pos);
                        const bool up = ((i & k2) == 0);
                        if (gt == up) { sel[i] = b; sel[ixj] = a; }
                    }
                }
                __syncthreads();
            }
        }
    }
    for (int i = t; i < k_cap; i += BM_THREADS) {
        if (i < kq) {
            const double sc = key2d_desc(sel[i].key);
            out_ids[(long)q * k_cap + i] = doc_ids[sel[i].pos];
            out_scores[(long)q * k_cap + i] = (float)sc;                // Score: float32(r.Score) (:392)
            if (out_scores64) out_scores64[(long)q * k_cap + i] = sc;
        } else {
            out_ids[(long)q * k_cap + i] = 0; out_scores[(long)q * k_cap + i] = 0.0f;
            if (out_scores64) out_scores64[(long)q * k_cap + i] = 0.0;
        }
    }
    if (t == 0) out_counts[q] = kq_all;
}

}  // namespace comet

using namespace comet;

struct comet_text_index {
    Ctx* c = nullptr;
    // host structures (bm25_index.go:101-114)
    std::map<uint32_t, std::vector<uint32_t>> postings;                       // term -> ascending doc ids
    std::unordered_map<uint32_t, std::unordered_map<uint32_t, int>> tf;       // term -> doc -> tf
    std::unordered_map<uint32_t, int> doc_len;
    std::unordered_map<uint32_t, std::vector<uint32_t>> doc_tokens;
    std::unordered_set<uint32_t> deleted;
    uint32_t num_docs = 0; long total_tokens = 0; double avg_doc_len = 0;
    // device snapshot
    bool dirty = true;
    std::vector<uint32_t> doc_ids_h; std::unordered_map<uint32_t, int> term_index; std::vector<int> term_off_h;
    DevBuf doc_ids, doc_len_dev, post_doc, post_tf, deleted_dev;
    DevBuf kin_dev;                                   // per document: 1.2 * (0.25 + 0.75 * (docLen / avgDocLen)) — the document's part of the score's denominator
    DevBuf tfcol; std::vector<int> hot_of_term;       // dense tf columns (uint16 per document) of the hot terms; hot_of_term[term index] = column or -1
    DevBuf acc; int64_t acc_rows = 0, acc_nd = -1;   // dense float64 accumulator rows (<= 224 MB: see comet_bm25_search)
    int64_t nd = 0;

    void update_avg() { avg_doc_len = num_docs == 0 ? 0 : (double)total_tokens / (double)num_docs; }   // bm25_index.go updateAvgDocLen
    void remove_internal(uint32_t id) {                                                                 // bm25_index.go removeInternal
        auto it = doc_tokens.find(id); if (it == doc_tokens.end()) return;
        const int dl = doc_len[id];
        for (uint32_t t : it->second) {
            auto pit = postings.find(t);
            if (pit != postings.end()) { auto& v = pit->second; auto f = std::lower_bound(v.begin(), v.end(), id); if (f != v.end() && *f == id) v.erase(f); if (v.empty()) postings.erase(pit); }
            auto tit = tf.find(t);
            if (tit != tf.end()) { tit->second.erase(id); if (tit->second.empty()) tf.erase(tit); }
        }
        doc_tokens.erase(it); doc_len.erase(id); num_docs--; total_tokens -= dl;
        if (num_docs > 0) update_avg(); else { avg_doc_len = 0; total_tokens = 0; }
        dirty = true;
    }
    void compile() {
        if (!dirty) return;
        doc_ids_h.clear();
        for (auto& kv : doc_len) doc_ids_h.push_back(kv.first);
        std::sort(doc_ids_h.begin(), doc_ids_h.end());
        nd = (int64_t)doc_ids_h.size();
        std::unordered_map<uint32_t, int> didx; didx.reserve(nd * 2);
        std::vector<int> dl(nd);
        for (int64_t i = 0; i < nd; i++) { didx[doc_ids_h[i]] = (int)i; dl[i] = doc_len[doc_ids_h[i]]; }
        term_index.clear(); term_off_h.assign(1, 0);
        std::vector<int> pd, pt;
        for (auto& kv : postings) {
            term_index[kv.first] = (int)term_off_h.size() - 1;
            auto& tfm = tf[kv.first];
            for (uint32_t d : kv.second) { pd.push_back(didx[d]); pt.push_back(tfm[d]); }
            term_off_h.push_back((int)pd.size());
        }
        doc_ids.reserve(std::max<size_t>(4, nd * 4), c->stream, 0); doc_len_dev.reserve(std::max<size_t>(4, nd * 4), c->stream, 0);
        post_doc.reserve(std::max<size_t>(4, pd.size() * 4), c->stream, 0); post_tf.reserve(std::max<size_t>(4, pt.size() * 4), c->stream, 0);
        // the document's part of the denominator, once per document (the scoring kernels' own operations, in their order)
        std::vector<double> kin(nd);
        for (int64_t i = 0; i < nd; i++) { const double ratio = (double)dl[i] / avg_doc_len; const double inner = 0.25 + 0.75 * ratio; kin[i] = 1.2 * inner; }
        kin_dev.reserve(std::max<size_t>(8, nd * 8), c->stream, 0);
        c->h2d(kin_dev.p, kin.data(), nd * 8);
        // dense tf columns for the hot terms (df >= n_docs / 64, at most 256 of them, every tf <= 65535): bm25_dense_kernel
        hot_of_term.assign(term_off_h.size() - 1, -1);
        std::vector<unsigned short> cols;
        if (nd >= 4096) {
            std::vector<std::pair<int, int>> hot;             // (df, term index)
            for (size_t t = 0; t + 1 < term_off_h.size(); t++) { const int df = term_off_h[t + 1] - term_off_h[t]; if ((int64_t)df * 64 >= nd) hot.push_back({df, (int)t}); }
            std::sort(hot.begin(), hot.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
            // at most 256 columns, inside a byte budget (COMET_BM25_DENSE_MB, default 1024: 2 bytes per (column, document), on the host and on the device — 256 columns
            // of a 10 M-document collection would be 5 GB each; the lowest-df columns go first, their terms keep the term-at-a-time path). `cols` is sized ONCE.
            static const size_t budget = [] { const char* e = getenv("COMET_BM25_DENSE_MB"); long mb = e ? atol(e) : 1024; return (size_t)std::max(0l, mb) << 20; }();
            const size_t fit = std::min<size_t>(256, budget / ((size_t)nd * 2));
            std::vector<int> keep;
            for (auto& hp : hot) {
                if (keep.size() >= fit) break;
                const int t = hp.second; bool fits = true;
                for (int i = term_off_h[t]; i < term_off_h[t + 1]; i++) if (pt[i] > 65535) { fits = false; break; }
                if (fits) keep.push_back(t);
            }
            cols.assign(keep.size() * (size_t)nd, 0);
            int nh = 0;
            for (int t : keep) {
                for (int i = term_off_h[t]; i < term_off_h[t + 1]; i++) cols[(size_t)nh * nd + pd[i]] = (unsigned short)pt[i];
                hot_of_term[t] = nh++;
            }
        }
        tfcol.reserve(std::max<size_t>(4, cols.size() * 2), c->stream, 0);
        c->h2d(tfcol.p, cols.data(), cols.size() * 2);
        c->h2d(doc_ids.p, doc_ids_h.data(), nd * 4); c->h2d(doc_len_dev.p, dl.data(), nd * 4);
        c->h2d(post_doc.p, pd.data(), pd.size() * 4); c->h2d(post_tf.p, pt.data(), pt.size() * 4);
        HIP_CHECK(hipStreamSynchronize(c->stream));
        dirty = false;
    }
};

extern "C" {

int comet_bm25_create(comet_ctx* c, comet_text_index** out) {
    return guarded([&] { c->bind(); auto* t = new comet_text_index(); t->c = c; *out = t; return (int)COMET_OK; });
}
int comet_bm25_destroy(comet_text_index* idx) {
    return guarded([&] { if (!idx) return (int)COMET_OK; Ctx* c = idx->c; std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind(); (void)hipStreamSynchronize(c->stream); delete idx; return (int)COMET_OK; });
}
// BM25SearchIndex.Add bm25_index.go:168-201 with the text already tokenised into ids
int comet_bm25_add(comet_text_index* ix, uint32_t id, const uint32_t* tokens, int32_t n) {
    return guarded([&] {
        std::lock_guard<std::recursive_mutex> lk(ix->c->mu);
        if (ix->doc_tokens.count(id)) ix->remove_internal(id);
        ix->doc_tokens[id].assign(tokens, tokens + n);
        ix->doc_len[id] = n; ix->num_docs++; ix->total_tokens += n;
        for (int i = 0; i < n; i++) {
            auto& v = ix->postings[tokens[i]];
            auto f = std::lower_bound(v.begin(), v.end(), id);
            if (f == v.end() || *f != id) v.insert(f, id);
            ix->tf[tokens[i]][id]++;
        }
        ix->update_avg(); ix->dirty = true;
        return (int)COMET_OK;
    });
}
// Remove: soft delete bm25_index.go:203-222
int comet_bm25_remove(comet_text_index* ix, uint32_t id) {
    return guarded([&] { std::lock_guard<std::recursive_mutex> lk(ix->c->mu); if (ix->doc_tokens.count(id)) ix->deleted.insert(id); return (int)COMET_OK; });
}
// Flush bm25_index.go:374-400
int comet_bm25_flush(comet_text_index* ix) {
    return guarded([&] {
        std::lock_guard<std::recursive_mutex> lk(ix->c->mu);
        std::vector<uint32_t> d(ix->deleted.begin(), ix->deleted.end()); std::sort(d.begin(), d.end());
        for (uint32_t id : d) ix->remove_internal(id);
        ix->deleted.clear();
        return (int)COMET_OK;
    });
}
int64_t comet_bm25_num_docs(const comet_text_index* ix) { return ix->num_docs; }
double comet_bm25_avg_doc_len(const comet_text_index* ix) { return ix->avg_doc_len; }

}  // extern "C"

// bm25TextSearch.searchSingleQuery bm25_index_search.go:278-397 for B tokenised queries: everything ENQUEUED on the context's current stream, results in
// device memory (B x k_cap ids / float32 scores / float64 scores, B counts). The caller holds the context mutex and has reset the scratch arena.
// Returns false when the index is empty (`if N == 0 { return nil, nil }`, :290): nothing was enqueued, every count is zero.
static bool bm25_search_enqueue(comet_text_index* ix, const uint32_t* q_tokens, const int32_t* q_offsets, int32_t B, int32_t k, const uint32_t* filter_ids, int32_t n_filter,
                                uint32_t* d_ids, float* d_sc, double* d_sc64, int32_t* d_cn, int32_t k_cap) {
        Ctx* c = ix->c;
        ix->compile();
        const int64_t nd = ix->nd;
        const double N = (double)ix->num_docs;
        if (nd == 0 || N == 0) return false;
        // eligibility: soft deletes + document filter (by id)
        const uint8_t* elig = nullptr;
        std::vector<uint32_t> del(ix->deleted.begin(), ix->deleted.end()); std::sort(del.begin(), del.end());
        std::vector<uint32_t> flt;
        if (filter_ids && n_filter > 0) { flt.assign(filter_ids, filter_ids + n_filter); std::sort(flt.begin(), flt.end()); flt.erase(std::unique(flt.begin(), flt.end()), flt.end()); }
        if (!del.empty() || !flt.empty()) {
            uint32_t* dd = c->salloc<uint32_t>(std::max<size_t>(1, del.size())); uint32_t* df = c->salloc<uint32_t>(std::max<size_t>(1, flt.size()));
            c->h2d(dd, del.data(), del.size() * 4); c->h2d(df, flt.data(), flt.size() * 4);
            uint8_t* e = c->salloc<uint8_t>(nd);
            launch_build_elig(c, ix->doc_ids.as<uint32_t>(), nd, dd, (int)del.size(), df, (int)flt.size(), e);
            elig = e;
        }
        int maxlen = 0;
        for (int b = 0; b < B; b++) maxlen = std::max(maxlen, q_offsets[b + 1] - q_offsets[b]);
        std::vector<TermRef> refs((size_t)std::max(1, maxlen) * B);
        // queries all of whose known terms have a dense tf column are scored document-major (bm25_dense_kernel); the others term-at-a-time
        static const bool dense_off = getenv("COMET_BM25_NO_DENSE") != nullptr;
        std::vector<DenseRef> dense_refs; std::vector<int> dense_q, sparse_q;
        std::vector<DenseRef> one((size_t)std::max(1, maxlen));
        int maxdf = 0; int64_t tcap = 1;
        for (int b = 0; b < B; b++) {
            int64_t touched_max = 0;
            const int len = q_offsets[b + 1] - q_offsets[b];
            bool dense = !dense_off && ix->tfcol.p != nullptr && len > 0;
            int q_maxdf = 0;
            for (int j = 0; j < maxlen; j++) {
                TermRef r{0, 0, 0.0};
                one[j] = DenseRef{-1, 0, 0.0};
                if (j < len) {
                    auto it = ix->term_index.find(q_tokens[q_offsets[b] + j]);
                    if (it != ix->term_index.end()) {
                        r.off = ix->term_off_h[it->second];
                        r.df = ix->term_off_h[it->second + 1] - r.off;
                        const double df = (double)r.df;
                        r.idf = go_log((N - df + 0.5) / (df + 0.5) + 1.0);     // :306
                        q_maxdf = std::max(q_maxdf, r.df);
                        touched_max += r.df;
                        const int h = ix->hot_of_term[it->second];
                        if (h < 0) dense = false; else one[j] = DenseRef{h, 0, r.idf};
                    }
                }
                refs[(size_t)j * B + b] = r;
            }
            // (document-major costs one thread per DOCUMENT whatever the terms' frequencies: it pays when the query's postings cover a good part of the collection)
            if (dense && touched_max * 8 >= nd) {
                for (int j = 0; j < maxlen; j++) refs[(size_t)j * B + b] = TermRef{0, 0, 0.0};       // nothing for the term-at-a-time launches
                dense_q.push_back(b); dense_refs.insert(dense_refs.end(), one.begin(), one.end());
            } else { sparse_q.push_back(b); maxdf = std::max(maxdf, q_maxdf); }
            tcap = std::max(tcap, std::min<int64_t>(touched_max, nd));
        }
        // The dense accumulator lives in the index and is zeroed per sub-batch of queries — deliberately: sized to stay inside the
        // 256 MB Infinity Cache, the memset leaves its lines there and the random read-modify-writes of the scoring launches hit
        // cache instead of HBM (measured: a cold accumulator costs 1.1 ms per launch instead of 0.2 at 100 k docs x 256 queries).
        // More documents mean fewer queries per sub-batch, not a bigger accumulator; selection never scans it (touched lists).
        const int64_t rows = std::max<int64_t>(1, std::min<int64_t>(B, ((int64_t)224 << 20) / (nd * 8)));
        if (ix->acc_nd != nd || ix->acc_rows < rows) {
            ix->acc.reserve((size_t)rows * nd * 8, c->stream, 0);
            ix->acc_rows = rows; ix->acc_nd = nd;
        }
        double* acc = ix->acc.as<double>();
        // the search's host-made tables go up in ONE copy (term references | dense references | the two query lists): a copy from pageable memory costs
        // ~10 us of host time whatever its size, and there were four of them in a 0.5 ms call
        std::vector<int> loc_d(dense_q.size()), loc_s(sparse_q.size());
        const int64_t rows_pre = std::max<int64_t>(1, std::min<int64_t>(B, ((int64_t)224 << 20) / (nd * 8)));
        for (size_t i = 0; i < dense_q.size(); i++) loc_d[i] = dense_q[i] % (int)rows_pre;          // local row inside the query's sub-batch
        for (size_t i = 0; i < sparse_q.size(); i++) loc_s[i] = sparse_q[i] % (int)rows_pre;
        const size_t o_refs = 0, o_dense = round_up(o_refs + refs.size() * sizeof(TermRef), 16), o_dq = round_up(o_dense + dense_refs.size() * sizeof(DenseRef), 16),
                     o_sq = round_up(o_dq + ((size_t)B + 1) * 4, 16), up_bytes = round_up(o_sq + ((size_t)B + 1) * 4, 16);
        std::vector<unsigned char> up(up_bytes, 0);
        memcpy(up.data() + o_refs, refs.data(), refs.size() * sizeof(TermRef));
        if (!dense_refs.empty()) memcpy(up.data() + o_dense, dense_refs.data(), dense_refs.size() * sizeof(DenseRef));
        if (!loc_d.empty()) memcpy(up.data() + o_dq, loc_d.data(), loc_d.size() * 4);
        if (!loc_s.empty()) memcpy(up.data() + o_sq, loc_s.data(), loc_s.size() * 4);
        unsigned char* d_up = c->salloc<unsigned char>(up_bytes);
        c->h2d(d_up, up.data(), up_bytes);
        TermRef* drefs = reinterpret_cast<TermRef*>(d_up + o_refs);
        int32_t* touched = c->salloc<int32_t>((size_t)rows * tcap);
        unsigned long long* tkeys = c->salloc<unsigned long long>((size_t)rows * tcap);
        int32_t* tcount = c->salloc<int32_t>((size_t)rows * BM_TCOUNT_STRIDE);
        int64_t slab_ld = 1; while (slab_ld < std::min<int64_t>(k_cap, tcap)) slab_ld <<= 1;
        KP* slab = (std::min<int64_t>(k_cap, tcap) > BM_KMAX) ? c->salloc<KP>((size_t)rows * slab_ld) : nullptr;
        DenseRef* d_dense = dense_refs.empty() ? nullptr : reinterpret_cast<DenseRef*>(d_up + o_dense);
        int* d_dq = reinterpret_cast<int*>(d_up + o_dq); int* d_sq = reinterpret_cast<int*>(d_up + o_sq);
        size_t di = 0, si = 0;                        // dense_q / sparse_q are ascending: a sub-batch owns a contiguous run of each
        for (int b0 = 0; b0 < B; b0 += (int)rows) {
            const int bn = std::min<int>((int)rows, B - b0);
            c->zero(tcount, sizeof(int32_t) * bn * BM_TCOUNT_STRIDE);
            size_t d1 = di, s1 = si;
            while (d1 < dense_q.size() && dense_q[d1] < b0 + bn) d1++;
            while (s1 < sparse_q.size() && sparse_q[s1] < b0 + bn) s1++;
            const int n_dense = (int)(d1 - di), n_sparse = (int)(s1 - si);
            if (n_sparse > 0 && maxdf > 0) {
                { ProfScope ps(c, "bm25_zero");
                  bm25_zero_rows_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div(nd, 256 * 8), 64), n_sparse), dim3(256), 0, c->stream>>>(d_sq + si, acc, nd); LAUNCH_CHECK(); }
                for (int j = 0; j < maxlen; j++) {
                    ProfScope ps(c, "bm25_score");
                    bm25_score_kernel<<<dim3((unsigned)ceil_div(maxdf, 256), bn), dim3(256), 0, c->stream>>>(drefs + (size_t)j * B + b0, ix->post_doc.as<int>(), ix->post_tf.as<int>(),
                                                                                                        ix->kin_dev.as<double>(), elig, acc, nd, touched, tcap, tcount);
                    LAUNCH_CHECK();
                }
            }
            if (n_dense > 0) {
                ProfScope ps(c, "bm25_dense");
                bm25_dense_kernel<<<dim3((unsigned)ceil_div(nd, 256), n_dense), dim3(256), 0, c->stream>>>(d_dq + di, d_dense + di * (size_t)std::max(1, maxlen), std::max(1, maxlen),
                                                                                                          ix->tfcol.as<unsigned short>(), ix->kin_dev.as<double>(), elig, acc, nd, touched, tcap, tcount);
                LAUNCH_CHECK();
            }
            di = d1; si = s1;
            {
                ProfScope ps(c, "bm25_topk");
                bm25_topk_kernel<<<dim3(bn), dim3(BM_THREADS), 0, c->stream>>>(acc, nd, touched, tcap, tcount, tkeys, k, ix->doc_ids.as<uint32_t>(), slab, slab_ld,
                                                                              d_ids + (size_t)b0 * k_cap, d_sc + (size_t)b0 * k_cap, d_sc64 + (size_t)b0 * k_cap, d_cn + b0, k_cap);
                LAUNCH_CHECK();
            }
        }
        return true;
}

extern "C" {

int comet_bm25_search(comet_text_index* ix, const uint32_t* q_tokens, const int32_t* q_offsets, int32_t B, int32_t k,
                      const uint32_t* filter_ids, int32_t n_filter, uint32_t* out_ids, float* out_scores, double* out_scores64,
                      int32_t* out_counts, int32_t k_cap) {
    return guarded([&] {
        if (B <= 0) return (int)COMET_OK;
        if (k_cap <= 0) COMET_FAIL(COMET_ERR_INVALID_ARG, "k_cap must be positive");
        Ctx* c = ix->c;
        std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind(); c->switch_lane(0); c->quiesce_alt(); c->scratch_reset();
        // one result block (float64 scores | ids | float32 scores | counts), one copy back into pinned memory
        const size_t nk = (size_t)B * k_cap, o_sc64 = 0, o_ids = nk * 8, o_sc = o_ids + nk * 4, o_cn = o_sc + nk * 4, blk = o_cn + (size_t)B * 4;
        unsigned char* d_blk = c->salloc<unsigned char>(blk);
        uint32_t* d_ids = reinterpret_cast<uint32_t*>(d_blk + o_ids);
        float* d_sc = reinterpret_cast<float*>(d_blk + o_sc);
        double* d_sc64 = reinterpret_cast<double*>(d_blk + o_sc64);
        int32_t* d_cn = reinterpret_cast<int32_t*>(d_blk + o_cn);
        if (!bm25_search_enqueue(ix, q_tokens, q_offsets, B, k, filter_ids, n_filter, d_ids, d_sc, d_sc64, d_cn, k_cap)) {
            std::fill(out_counts, out_counts + B, 0);
            return (int)COMET_OK;
        }
        unsigned char* hb = static_cast<unsigned char*>(c->pinned_buf(blk));
        c->d2h(hb, d_blk, blk);
        c->sync();
        memcpy(out_ids, hb + o_ids, nk * 4);
        memcpy(out_scores, hb + o_sc, nk * 4);
        if (out_scores64) memcpy(out_scores64, hb + o_sc64, nk * 8);
        memcpy(out_counts, hb + o_cn, (size_t)B * 4);
        return (int)COMET_OK;
    });
}

// ---- hybrid search on the device: vector leg + text leg + Reciprocal Rank Fusion (hybridSearch.Execute hybrid_search_index.go:477-615 with
// WithFusionKind(ReciprocalRankFusion); reciprocalRankFusion.Combine + scoreMapToRanks fusion.go:174-243) ----------------------------------------------
// The reference runs the two sub-searches one after the other, cuts both to k (:518, :555), turns each result list into 0-based ranks, sums 1 / (K + rank)
// per document in float64 (vector term first), sorts descending and cuts to k. Here the vector leg is enqueued on another execution lane of the context
// and runs BESIDE the text leg (whose host part — term lookup, idf — happens while the vector leg's kernels are already queued), the fusion is one wave
// per query on the device, and the host receives one block of results: no per-leg download, no host-side rank arithmetic.
}  // extern "C"

namespace comet {
// one wave per query. Candidates in insertion order: the vector hits (position = rank), then the text hits that are not vector hits; a vector hit that is
// also a text hit adds the text term (existing + rrfScore). Order: score descending, ties in insertion order (the stable sort of the host mirror).
__global__ __launch_bounds__(64) void rrf_fuse_kernel(const unsigned* __restrict__ v_ids, const float* __restrict__ v_sc, const int* __restrict__ v_cnt, int kv,
                                                     const unsigned* __restrict__ t_ids, const float* __restrict__ t_sc, const int* __restrict__ t_cnt, int kt, double rrf_k, int k_out,
                                                     unsigned* __restrict__ out_ids, double* __restrict__ out_sc, int* __restrict__ out_cnt, int out_ld) {
    __shared__ unsigned s_id[128]; __shared__ double s_sc[128]; __shared__ unsigned char s_ok[128];
    const int q = blockIdx.x, lane = threadIdx.x;
    const int vc = min(max(v_cnt[q], 0), kv), tc = min(max(t_cnt[q], 0), kt);
    const int verr = v_cnt[q] < 0 ? v_cnt[q] : 0;                              // a search-time error of the vector leg (a zero query under cosine) is passed through
    const unsigned vid = lane < vc ? v_ids[(long)q * kv + lane] : 0u, tid = lane < tc ? t_ids[(long)q * kt + lane] : 0u;
    s_id[lane] = vid; s_id[64 + lane] = tid;
    __syncthreads();
    int hit = -1;                                                               // position of this lane's vector hit in the text list
    if (lane < vc) for (int j = 0; j < tc; j++) if (s_id[64 + j] == vid) { hit = j; break; }
    bool t_only = lane < tc;
    if (t_only) for (int i = 0; i < vc; i++) if (s_id[i] == tid) { t_only = false; break; }
    // `if len(vectorResults) > 0 && len(textResults) > 0` (hybrid_search_index.go:577): both legs have hits -> fused ranks; otherwise the leg that has hits
    // keeps its OWN scores (vector: distances; text: float32 BM25 scores) and only the descending sort + the cut apply (:603-611)
    const bool fuse = vc > 0 && tc > 0;
    const double r_own = 1.0 / (rrf_k + (double)lane);                         // 1 / (K + rank): a hit's rank is its position in its list
    double sv, st;
    if (fuse) { sv = hit >= 0 ? r_own + 1.0 / (rrf_k + (double)hit) : r_own; st = r_own; }      // existing + rrfScore: the vector term first (fusion.go:196-199)
    else { sv = lane < vc ? (double)v_sc[(long)q * kv + lane] : 0.0; st = lane < tc ? (double)t_sc[(long)q * kt + lane] : 0.0; }
    // concatenated candidates in insertion order: slot lane = vector hit, slot 64 + lane = text hit that is not a vector hit
    s_sc[lane] = sv; s_ok[lane] = lane < vc ? 1 : 0;
    s_sc[64 + lane] = st; s_ok[64 + lane] = (lane < tc && t_only) ? 1 : 0;
    __syncthreads();
    const int total = vc + (int)__builtin_popcountll(__ballot(lane < tc && t_only));
    const int kq = verr ? 0 : ((k_out <= 0 || k_out > total) ? total : k_out);
    for (int h = 0; h < 2; h++) {
        const int me = h * 64 + lane;
        if (!s_ok[me]) continue;
        const double sc = s_sc[me];
        // rank by counting on a TOTAL order (round-5 advisor: with `o > sc || (o == sc && j < me)` every NaN — a leg's own distances of a non-finite query —
        // got the same rank, slots below kq stayed unwritten and came back as ids): the order-preserving image of the score, NaN last, ties by slot
        auto key = [](double x) -> unsigned long long {
            if (x != x) return 0ull;
            const unsigned long long u = (unsigned long long)__double_as_longlong(x + 0.0);      // (-0 + 0 = +0: the two zeros tie, as under ==)
            return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
        };
        const unsigned long long ks = key(sc);
        int rank = 0;
        for (int j = 0; j < 128; j++) { const unsigned long long ko = key(s_sc[j]); rank += (s_ok[j] && (ko > ks || (ko == ks && j < me))) ? 1 : 0; }
        if (rank < kq && rank < out_ld) { out_ids[(long)q * out_ld + rank] = s_id[me]; out_sc[(long)q * out_ld + rank] = sc; }
    }
    for (int i = kq + lane; i < out_ld; i += 64) { out_ids[(long)q * out_ld + i] = 0u; out_sc[(long)q * out_ld + i] = 0.0; }
    if (lane == 0) out_cnt[q] = verr ? verr : min(kq, out_ld);
}
}  // namespace comet

extern "C" {

int comet_hybrid_rrf_search(comet_index* vec, comet_text_index* txt, const float* queries, const uint32_t* q_tokens, const int32_t* q_offsets, int32_t B,
                            int32_t k, int32_t nprobes, int32_t ef_search, double rrf_k, uint32_t* out_ids, double* out_scores, int32_t* out_counts) {
    return guarded([&] {
        if (B <= 0) return (int)COMET_OK;
        if (k <= 0 || k > 64) COMET_FAIL(COMET_ERR_UNSUPPORTED, "the device fusion ranks up to 64 hits per leg (k = %d): fuse larger lists on the host", k);
        if (vec->c != txt->c) COMET_FAIL(COMET_ERR_INVALID_ARG, "the vector index and the text index live on different contexts");
        Ctx* c = vec->c;
        std::lock_guard<std::recursive_mutex> lk(c->mu); c->bind(); c->switch_lane(0); c->quiesce_alt(); c->scratch_reset();
        const size_t nk = (size_t)B * k;
        // results of the legs and of the fusion: lane 0's scratch (alive until the next call resets it; this call ends with a sync)
        uint32_t* v_ids = c->salloc<uint32_t>(nk); float* v_sc = c->salloc<float>(nk); int32_t* v_cn = c->salloc<int32_t>(B);
        uint32_t* t_ids = c->salloc<uint32_t>(nk); float* t_sc = c->salloc<float>(nk); double* t_sc64 = c->salloc<double>(nk); int32_t* t_cn = c->salloc<int32_t>(B);
        uint32_t* f_ids = c->salloc<uint32_t>(nk); double* f_sc = c->salloc<double>(nk); int32_t* f_cn = c->salloc<int32_t>(B);
        float* qd = c->salloc<float>((size_t)B * vec->dim);
        c->h2d(qd, queries, (size_t)B * vec->dim * sizeof(float));
        c->fence_lane0();
        // vector leg: on lane 1 (its own stream and scratch arena) when the context has more than one lane, behind the upload
        comet_search_params p{}; p.k = k; p.nprobes = nprobes; p.ef_search = ef_search; p.mode = 0;
        const int vlane = (c->lanes > 1 && vec->max_lanes() > 1) ? 1 : 0;
        uint64_t ticket = 0;
        {
            struct LaneBack { Ctx* c; ~LaneBack() { if (c->cur_lane != 0) { c->mark_dirty(); c->switch_lane(0); } } } lane_back{c};
            c->switch_lane(vlane);
            if (vlane) { c->scratch_reset(); c->follow_lane0(); }
            ticket = vec->search_begin(qd, B, p, v_ids, v_sc, v_cn, k);
        }
        // text leg on lane 0 (host part first: the vector leg's kernels are already queued)
        if (!bm25_search_enqueue(txt, q_tokens, q_offsets, B, k, nullptr, 0, t_ids, t_sc, t_sc64, t_cn, k)) HIP_CHECK(hipMemsetAsync(t_cn, 0, (size_t)B * 4, c->stream));
        // the vector leg's results are final (the Flat / IVF fast path may re-run a query exactly), then the fusion behind both legs
        {
            struct LaneBack { Ctx* c; ~LaneBack() { if (c->cur_lane != 0) { c->mark_dirty(); c->switch_lane(0); } } } lane_back{c};
            c->switch_lane(vlane);
            vec->search_finish(ticket);
            if (vlane) HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        { ProfScope ps(c, "rrf_fuse");
          rrf_fuse_kernel<<<dim3((unsigned)B), dim3(64), 0, c->stream>>>(v_ids, v_sc, v_cn, k, t_ids, t_sc, t_cn, k, rrf_k, k, f_ids, f_sc, f_cn, k); LAUNCH_CHECK(); }
        c->d2h(out_ids, f_ids, nk * 4); c->d2h(out_scores, f_sc, nk * 8); c->d2h(out_counts, f_cn, (size_t)B * 4);
        c->sync();
        return (int)COMET_OK;
    });
}

}  // extern "C"
