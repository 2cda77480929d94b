/* comet_gpu.h — C ABI of the MI355X (gfx950) backend for wizenheimer/comet's vector-search hot path.
 *
 * This is the drop-in boundary: the entry points a cgo shim binds to implement comet's own Go
 * interfaces (`VectorIndex` index.go:32-63, `VectorSearch` index_search.go:141-279, `TextIndex`
 * index.go:65-81, `Distance` distance.go:50-81) on top of hand-written HIP kernels. Plain pointers
 * and sizes only; no torch / C++ types. The Go-side binding is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns a comet_status (0 = OK); comet_last_error() gives a thread-local
 *    message whose text mirrors the reference's fmt.Errorf strings where one exists.
 *  - host-pointer entry points copy their inputs before returning (cgo may not retain Go pointers).
 *  - `*_dev` entry points take DEVICE pointers (memory from comet_dev_alloc or any HIP allocation on
 *    the context's device) and run asynchronously on the context's stream (lane 0; asynchronous searches on
 *    one of the context's execution lanes, see comet_ctx_stream / comet_ctx_set_lanes); comet_ctx_sync() waits for all of them.
 *  - one search call = B independent queries -> B result rows (the Go shim maps the reference's
 *    "multi-query Execute() aggregates into one list" semantics onto it, flat_index_search.go:143-153).
 *  - result rows are sorted ascending by score; exact score ties are ordered by scan order (the
 *    reference's own order among ties is undefined: unstable sort.Slice, flat_index_search.go:277).
 *  - threading: any number of concurrent searches per index; mutators (train/add/remove/flush)
 *    must be externally serialised against searches (the Go shim's RWMutex, flat_index.go:93).
 */
#ifndef COMET_GPU_H
#define COMET_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COMET_API __attribute__((visibility("default")))

typedef struct comet_ctx comet_ctx;     /* one per GPU (HIP device + stream + scratch arena) */
typedef struct comet_index comet_index; /* Flat / IVF / PQ / IVFPQ / HNSW / BM25 */

/* DistanceKind, distance.go:19-39 ("l2", "l2_squared", "cosine") */
typedef enum { COMET_L2 = 0, COMET_L2SQ = 1, COMET_COSINE = 2 } comet_metric;

/* VectorIndexKind, index.go */
typedef enum { COMET_KIND_FLAT = 0, COMET_KIND_IVF = 1, COMET_KIND_PQ = 2, COMET_KIND_IVFPQ = 3,
               COMET_KIND_HNSW = 4, COMET_KIND_BM25 = 5 } comet_kind;

typedef enum {
    COMET_OK = 0,
    COMET_ERR_INVALID_ARG = 1,     /* constructor / argument validation (e.g. "dimension must be positive") */
    COMET_ERR_DIM_MISMATCH = 2,    /* "query dimension mismatch: expected %d, got %d" flat_index_search.go:227 */
    COMET_ERR_ZERO_VECTOR = 3,     /* ErrZeroVector distance.go:12 */
    COMET_ERR_NOT_TRAINED = 4,     /* "index must be trained before searching" ivf_index_search.go:223 */
    COMET_ERR_NOT_FOUND = 5,       /* "vector with ID %d not found" flat_index.go:233 */
    COMET_ERR_ALREADY_DELETED = 6, /* "vector with ID %d already deleted" flat_index.go:236 */
    COMET_ERR_TRAIN_DATA = 7,      /* "need at least %d vectors for training" ivfpq_index.go:186 */
    COMET_ERR_HIP = 8,             /* HIP runtime failure */
    COMET_ERR_NO_DEVICE = 9,       /* no usable gfx950 device */
    COMET_ERR_UNSUPPORTED = 10,
    COMET_ERR_UNKNOWN_METRIC = 11, /* ErrUnknownDistanceKind distance.go:9 */
    COMET_ERR_FORMAT = 12,         /* ReadFrom: "invalid magic number: expected 'FLAT', got '%s'" flat_index.go:516, "unsupported version", "dimension mismatch", ... */
    COMET_ERR_IO = 13              /* ReadFrom / WriteTo: the caller's reader ran dry ("failed to read ...") or its writer failed */
} comet_status;

COMET_API const char* comet_last_error(void);
COMET_API const char* comet_version(void);
COMET_API int comet_device_count(int* out_count);

/* ---- context ------------------------------------------------------------------------------ */
COMET_API int comet_ctx_create(int device_id, comet_ctx** out);
COMET_API int comet_ctx_destroy(comet_ctx* ctx);
COMET_API int comet_ctx_sync(comet_ctx* ctx);
/* raw HIP stream handle (hipStream_t) of execution lane 0: the stream every call that is not an asynchronous search enqueues on
 * (uploads, comet_synth_*, adds, the synchronous searches). Asynchronous searches rotate through the context's lanes (comet_ctx_set_lanes);
 * a search that lands on another lane starts behind the work the LIBRARY last queued on lane 0 (so `comet_synth_fill_dev(queries)` followed by
 * `comet_index_search_dev_async(queries)` is ordered), but the library cannot see kernels the caller enqueues on this stream itself: call
 * comet_ctx_fence() after them (or comet_ctx_sync()) before an asynchronous search that reads what they write. */
COMET_API void* comet_ctx_stream(comet_ctx* ctx);
/* Marks "everything queued on lane 0's stream up to here" as a dependency of every asynchronous search enqueued afterwards, whichever lane it runs
 * on. No host wait. Needed only after the caller's OWN work on comet_ctx_stream(); the library's own calls place the fence themselves. */
COMET_API int comet_ctx_fence(comet_ctx* ctx);
COMET_API int comet_dev_alloc(comet_ctx* ctx, size_t bytes, void** out_dev);
COMET_API int comet_dev_free(comet_ctx* ctx, void* dev);
COMET_API int comet_memcpy_h2d(comet_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
COMET_API int comet_memcpy_d2h(comet_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* SplitMix64 synthetic data, value = 2*((next>>40)*2^-24)-1 in [-1,1); element i of the stream uses
 * counter (offset+i) — bit-identical to the oracle's orc_synth_fill. */
COMET_API int comet_synth_fill_dev(comet_ctx* ctx, uint64_t seed, uint64_t offset, uint64_t n, float* out_dev);
/* Clustered variant (for meaningful ANN recall), two levels: rows [row_base, row_base+n_rows) x dim of
 *   C1[b1(r)][j] + sigma * C2[b2(r)][j] + sigma_noise * u(seed, r*dim+j)    (n_sub > 0; n_sub <= 0: C1[b1(r)][j] + sigma * u(...))
 * C1[c][j] = u(seed ^ 0x5EED, c*dim+j), C2[c][j] = u(seed ^ 0x5EED2, c*dim+j), b1(r) = ((r*2654435761)>>7) % n_centers,
 * b2(r) = ((r*0x9E3779B1)>>5) % n_sub; bit-identical to the oracle's orc_synth_mixture. n_centers <= 0: the plain stream. */
COMET_API int comet_synth_mixture_dev(comet_ctx* ctx, uint64_t seed, int32_t n_centers, float sigma, int32_t n_sub, float sigma_noise,
                                      uint64_t row_base, uint64_t n_rows, int32_t dim, float* out_dev);

/* Execution lanes of a context (1 .. 8; default 8, or COMET_LANES): the asynchronous searches of an index (comet_index_search_dev_async,
 * comet_index_search_sharded_async) rotate through that many streams, each with a scratch arena of its own — as many as the index kind gains
 * from: two for Flat and IVF (one kernel of their step fills the GPU), four for PQ / IVFPQ, eight for HNSW (one wave per query: 8 x 256 queries are
 * two waves per SIMD) — so that the short latency-bound kernels
 * of the batches in flight run beside each other. Results and the meaning of comet_index_search_wait are unchanged; every other call first
 * waits for the other lanes. Replaces nothing in the reference (a Go caller overlaps searches with goroutines); the call itself waits for all lanes. */
COMET_API int comet_ctx_set_lanes(comet_ctx* ctx, int32_t lanes);

/* per-kernel timing (HIP events recorded on the context's stream around each launch) */
COMET_API int comet_profile_enable(comet_ctx* ctx, int on);
COMET_API int comet_profile_reset(comet_ctx* ctx);
/* time only the scopes named in the '|'-separated list `name` (NULL or "": all). Every timed scope costs two event records — barrier packets in the
 * stream — which is ~10 us in a chain of short dependent kernels; a benchmark times just the kernel its roofline is about. */
COMET_API int comet_profile_only(comet_ctx* ctx, const char* name);
/* total milliseconds and launch count of kernels whose name starts with `prefix` since the last reset */
COMET_API int comet_profile_get(comet_ctx* ctx, const char* prefix, double* out_total_ms, int64_t* out_launches);
/* writes a '\n'-separated "name total_ms launches" listing into buf */
COMET_API int comet_profile_dump(comet_ctx* ctx, char* buf, size_t cap);

/* ---- Distance singletons (comet.Distance, distance.go:50-81) -------------------------------- */
/* Calculate(a, b): distance.go:114/158/201 */
COMET_API int comet_distance(comet_ctx* ctx, int metric, const float* a, const float* b, int d, float* out);
/* CalculateBatch(queries, target): distance.go:123/167/218 */
COMET_API int comet_distance_batch(comet_ctx* ctx, int metric, const float* queries, int nq, const float* target,
                                   int d, float* out);
/* Preprocess / PreprocessInPlace (out may alias x): distance.go:138-147,182-191,244-290 */
COMET_API int comet_preprocess(comet_ctx* ctx, int metric, const float* x, int d, float* out);
/* Norm distance.go:312-318 for n vectors of d floats (dense rows): out_norms[i] = float32(sqrt(float64(sum))), the sum formed serially in
 * float32 as the Go loop does. */
COMET_API int comet_norm_batch(comet_ctx* ctx, const float* x, int64_t n, int32_t d, float* out_norms);
/* Normalize distance.go:374-399 / NormalizeInPlace :403-428 (out may alias x): every row times 1 / Norm(row); a zero row is returned
 * unchanged (no error, unlike the cosine Preprocess). */
COMET_API int comet_normalize_batch(comet_ctx* ctx, const float* x, int64_t n, int32_t d, float* out);
/* Scale distance.go:341-347: out[i] = x[i] * scalar (out may alias x). */
COMET_API int comet_scale_batch(comet_ctx* ctx, const float* x, int64_t n, int32_t d, float scalar, float* out);

/* ---- k-means (clustering.go:60,112,259) ---------------------------------------------------- */
/* KMeans / KMeansSubspace. out_centroids: min(k,n) x d; out_assign: n int32; *out_k = effective k */
COMET_API int comet_kmeans(comet_ctx* ctx, const float* vecs, int64_t n, int d, int k, int metric, int max_iter,
                           float* out_centroids, int32_t* out_assign, int* out_k);
/* FindNearestCentroidIndex for a batch of vectors */
COMET_API int comet_nearest_centroid(comet_ctx* ctx, const float* vecs, int64_t n, int d, const float* centroids,
                                     int k, int metric, int32_t* out_index);

/* ---- index lifecycle ------------------------------------------------------------------------ */
COMET_API int comet_flat_create(comet_ctx* ctx, int dim, int metric, comet_index** out);  /* NewFlatIndex flat_index.go:127 */
COMET_API int comet_ivf_create(comet_ctx* ctx, int dim, int metric, int nlist, comet_index** out); /* NewIVFIndex ivf_index.go:140 */
COMET_API int comet_pq_create(comet_ctx* ctx, int dim, int metric, int M, int nbits, comet_index** out); /* NewPQIndex pq_index.go:135 */
COMET_API int comet_ivfpq_create(comet_ctx* ctx, int dim, int metric, int nlist, int M, int nbits,
                                 comet_index** out); /* NewIVFPQIndex ivfpq_index.go:113 */
/* NewHNSWIndex hnsw_index.go:160 (m / ef_construction / ef_search <= 0 -> 16 / 200 / ef_construction) */
COMET_API int comet_hnsw_create(comet_ctx* ctx, int dim, int metric, int m, int ef_construction, int ef_search, comet_index** out);
/* Load a graph built by the reference (or the oracle): n nodes in any order with their ids, levels
 * (hnswNode.Level), stored (already preprocessed) vectors n x dim, and for every (node, layer <= level) in
 * node-major order an edge list: edge_offsets has sum(level+1)+1 entries into `edges` (neighbour NODE IDS,
 * hnswNode.Edges hnsw_index.go:50-61). */
COMET_API int comet_hnsw_load_graph(comet_index* idx, int64_t n, const uint32_t* ids, const int32_t* levels, const float* vecs,
                                    const int64_t* edge_offsets, const uint32_t* edges, uint32_t entry_id, int32_t max_level);
/* HNSW construction on the GPU: comet_index_add on an HNSW index runs insertNode (hnsw_index.go:493-552, selectNeighbors :637-656,
 * pruneConnections :667-694) for the batch strictly in order — one wave walks the graph exactly as the reference does, its neighbour
 * batches evaluated 64 lanes wide — and keeps the reference's quirks (maxLevel raised before the descent, entry point never promoted,
 * a full neighbour list drops the fresh back-edge). Node levels come from the index's own SplitMix64 stream (geometric p = 1/M, cap 16,
 * randomLevel :474-484; the reference draws from the unseeded global RNG, so its graphs are not reproducible either) or, with
 * comet_hnsw_add_with_levels, from the caller — which makes the graph bit-identical to an insertion sequence with those levels.
 * Limits: explicit non-zero ids, no re-adding of an id, M <= 128, efConstruction <= 1024. */
COMET_API int comet_hnsw_add_with_levels(comet_index* idx, const uint32_t* ids, const float* vecs, const int32_t* levels, int64_t n, int64_t* out_added);
COMET_API int comet_hnsw_set_level_seed(comet_index* idx, uint64_t seed);
/* the graph as comet_hnsw_load_graph takes it (sizes with NULL arrays first) */
COMET_API int comet_hnsw_export_graph(comet_index* idx, int64_t* out_n, int64_t* out_slots, int64_t* out_edges, uint32_t* ids, int32_t* levels,
                                      float* vecs, int64_t* edge_offsets, uint32_t* edges, uint32_t* out_entry_id, int32_t* out_max_level);
COMET_API int comet_index_destroy(comet_index* idx);

COMET_API int comet_index_kind(const comet_index* idx);
COMET_API int comet_index_dim(const comet_index* idx);
COMET_API int comet_index_metric(const comet_index* idx);
COMET_API int comet_index_trained(const comet_index* idx);
COMET_API int64_t comet_index_size(const comet_index* idx); /* stored vectors incl. soft-deleted */
/* default nprobes = floor(sqrt(nlist)) ivf_index.go:406-413; 0 for non-IVF kinds */
COMET_API int comet_index_default_nprobes(const comet_index* idx);

/* Train (VectorIndex.Train): vecs n x dim, host memory. No-op for Flat. */
COMET_API int comet_index_train(comet_index* idx, const float* vecs, int64_t n);
COMET_API int comet_index_train_dev(comet_index* idx, const float* vecs_dev, int64_t n);
/* Add (VectorIndex.Add) for a batch; processed in order, stops at the first failing vector
 * (*out_added = vectors added before the failure, like n sequential reference Add calls).
 * If normalized_out != NULL (host API only) the preprocessed vectors are written back — the reference
 * normalises the caller's slice in place for cosine (flat_index.go:182). */
COMET_API int comet_index_add(comet_index* idx, const uint32_t* ids, const float* vecs, int64_t n,
                              int64_t* out_added, float* normalized_out);
COMET_API int comet_index_add_dev(comet_index* idx, const uint32_t* ids_dev, const float* vecs_dev, int64_t n,
                                  int64_t* out_added);
/* Remove (soft delete) one id — flat_index.go:216-249 */
COMET_API int comet_index_remove(comet_index* idx, uint32_t id);
/* Flush (hard-delete soft-deleted rows) — flat_index.go:268-296 */
COMET_API int comet_index_flush(comet_index* idx);

/* ---- persistence: VectorIndex embeds io.WriterTo / io.ReaderFrom (index.go:58-60) --------------------
 * comet_index_write_to emits, and comet_index_read_from parses, the reference's own on-disk layouts byte for byte:
 *   Flat  "FLAT" flat_index.go:348-360 (WriteTo :366, ReadFrom :488)     IVF   "IVFX" ivf_index.go:441-462 (:468, :620)
 *   PQ    "PQIX" pq_index.go:480-505 (:509, :672)                        IVFPQ "IVPQ" ivfpq_index.go:507-535 (:544, :745)
 *   HNSW  "HNSW" hnsw_index.go:701-727 (:734, :898)
 * so an index written by the Go reference loads into the GPU backend and vice versa. Like the reference, write_to
 * calls Flush() first (soft-deleted vectors are never serialised; the roaring tail is the empty bitmap), and
 * read_from validates magic / version / every constructor parameter against the receiving index and replaces its
 * contents only when the whole stream parsed. HNSW nodes are written in ascending id order (the reference iterates
 * a Go map: any order is valid). The callbacks are what a cgo shim forwards to io.Writer.Write / io.ReadFull;
 * they return 0 on success. *out_bytes = bytes written / consumed (the int64 the Go methods return). */
typedef int (*comet_write_cb)(void* user, const void* data, size_t len);
typedef int (*comet_read_cb)(void* user, void* dst, size_t len);   /* must fill exactly len bytes */
COMET_API int comet_index_write_to(comet_index* idx, comet_write_cb cb, void* user, int64_t* out_bytes);
COMET_API int comet_index_read_from(comet_index* idx, comet_read_cb cb, void* user, int64_t* out_bytes);
/* Buffer forms: serialize with buf == NULL only reports the size (after the flush); deserialize consumes a prefix of buf. */
COMET_API int comet_index_serialize(comet_index* idx, uint8_t* buf, size_t cap, size_t* out_len);
COMET_API int comet_index_deserialize(comet_index* idx, const uint8_t* buf, size_t len, size_t* out_consumed);

/* ---- search --------------------------------------------------------------------------------- */
typedef struct {
    int32_t k;                  /* WithK; <=0 or > candidates => all (limiter.go:12-17) */
    float threshold;            /* WithThreshold; active only if > 0 (flat_index_search.go:269) */
    int32_t nprobes;            /* WithNProbes; <=0 or > nlist => nlist (ivf_index_search.go:233-236) */
    int32_t ef_search;          /* WithEfSearch; <=0 => index default (hnsw_index_search.go:302-305) */
    const uint32_t* filter_ids; /* WithDocumentIDs; NULL/0 => no filter (document_filter.go:27-30). HOST pointer. */
    int32_t n_filter;
    int32_t mode;               /* 0 = auto, 1 = strict: the reference's literal work (Flat: exact-arithmetic kernels only; IVFPQ: every
                                 * candidate of every probed list is scored, no lower-bound pruning), 2 = force the fast path. The
                                 * results are identical in every mode. */
} comet_search_params;

/* B queries (B x dim, host) -> out_ids / out_scores (B x k_cap, host), out_counts[B] = number of
 * results the reference would return for that query (rows hold min(count, k_cap) entries). */
COMET_API int comet_index_search(comet_index* idx, const float* queries, int32_t B, const comet_search_params* p,
                                 uint32_t* out_ids, float* out_scores, int32_t* out_counts, int32_t k_cap);
/* Same with device-resident queries and outputs; asynchronous on the context's stream.
 * The query buffer must stay valid and unchanged until the search has completed (comet_index_search_wait / a context sync): it may
 * be read in place. Per-query failures: the device entry points never synchronise, so a query the reference would fail as a whole (a zero
 * vector on a cosine index: ErrZeroVector, distance.go:12) is reported ONLY as out_counts[q] = -COMET_ERR_ZERO_VECTOR (a
 * negative count; its row is cleared) while the call and the other queries of the batch succeed. Read counts as signed.
 * The host-pointer comet_index_search above additionally returns COMET_ERR_ZERO_VECTOR when any query of the batch failed. */
COMET_API int comet_index_search_dev(comet_index* idx, const float* queries_dev, int32_t B,
                                     const comet_search_params* p, uint32_t* out_ids_dev, float* out_scores_dev,
                                     int32_t* out_counts_dev, int32_t k_cap);

/* Pipelined form: `_async` only ENQUEUES the search — on one of the context's execution lanes (streams), rotating per index — and
 * returns a ticket; comet_index_search_wait(ticket) blocks until THAT search has finished on the device and makes its results final
 * (the Flat fast path re-runs the rare queries whose candidate list overflowed on the exact kernels there).
 * Queries and output buffers must stay untouched until the wait returns. Several searches may be in flight; with more than one lane
 * they run beside each other and may finish in any order (wait for the ticket whose results you read). A search starts behind the
 * non-search work the library last queued on lane 0 (see comet_ctx_stream / comet_ctx_fence). comet_index_search_dev == async + wait. */
COMET_API int comet_index_search_dev_async(comet_index* idx, const float* queries_dev, int32_t B, const comet_search_params* p,
                                           uint32_t* out_ids_dev, float* out_scores_dev, int32_t* out_counts_dev, int32_t k_cap,
                                           uint64_t* out_ticket);
COMET_API int comet_index_search_wait(comet_index* idx, uint64_t ticket);

/* ---- multi-GPU merge (the step after the RCCL all-gather of per-shard top-K; the reference's
 * analogue is mergeResults storage_merge.go:13-46) -------------------------------------------- */
/* in: R shards x B queries x k_cap (ids, scores ascending), counts R x B. Ties: lower shard first. */
COMET_API int comet_merge_topk_dev(comet_ctx* ctx, const uint32_t* ids_dev, const float* scores_dev,
                                   const int32_t* counts_dev, int32_t R, int32_t B, int32_t k_cap, int32_t k,
                                   uint32_t* out_ids_dev, float* out_scores_dev, int32_t* out_counts_dev);
/* Same merge over R packed per-shard blocks, `block_words` 32-bit words apart, each laid out as
 * [B*k_cap ids | B*k_cap scores | B counts] — what ONE all-gather of per-rank result blocks produces (a search writes its
 * three outputs straight into such a block). */
COMET_API int comet_merge_topk_packed_dev(comet_ctx* ctx, const uint32_t* packed_dev, int64_t block_words, int32_t R, int32_t B,
                                          int32_t k_cap, int32_t k, uint32_t* out_ids_dev, float* out_scores_dev,
                                          int32_t* out_counts_dev);

/* ---- segment layer: persistentHybridSearch.Execute storage.go:489-626, vector leg -----------
 * The persistent store fans one query over a hybrid index per memtable / disk segment and merges on the host
 * (mergeResults storage_merge.go:13-46: highest score per document id; sortResultsByScore :50-54: DESCENDING;
 * merged[:k] storage.go:621-623). Here the segments' vector indexes stay resident in HBM and ONE call searches all of
 * them for a batch of queries (the per-segment searches are enqueued back to back, each with the caller's parameters
 * exactly as :503-533 / :568-597 pass them) and merges on the device. `segments[i]`: any vector index kinds of one
 * dimension on one context, oldest first. Outputs as comet_index_search_dev: row q holds min(k, distinct ids) entries,
 * counts[q] of them (or a negative status code, see comet_index_search_dev). Scores are the per-segment search scores
 * (distances) — the reference sorts them descending at this layer, and so does this call. Equal scores: ascending id
 * (Go map order in the reference, i.e. unspecified). k == 0 returns empty rows (merged[:0]); k < 0 is an error (a slice
 * bounds panic in the reference). */
COMET_API int comet_segments_search_dev(comet_index* const* segments, int32_t n_segments, const float* queries_dev, int32_t B,
                                        const comet_search_params* p, uint32_t* out_ids_dev, float* out_scores_dev,
                                        int32_t* out_counts_dev, int32_t k_cap);
/* host buffers in and out (one H2D, one D2H per call) */
COMET_API int comet_segments_search(comet_index* const* segments, int32_t n_segments, const float* queries, int32_t B,
                                    const comet_search_params* p, uint32_t* out_ids, float* out_scores, int32_t* out_counts,
                                    int32_t k_cap);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI inside the library --------------------------------
 * Every rank holds one shard of an index (Flat: a contiguous row block; IVF / PQ / IVFPQ: see comet_index_set_shard)
 * and searches it for the SAME query batch; the per-shard top-K blocks are exchanged with ONE ncclAllGather per batch on
 * the communicator's own HIP stream and merged on every rank (ties: lower rank, then lower position — the unsharded
 * canonical order when lower ranks hold earlier rows / lists). The reference's analogue is the per-segment fan-out +
 * mergeResults of storage.go:546-626 / storage_merge.go:13-46. The host only has to carry the 128-byte RCCL id from rank 0
 * to the other processes (any side channel: a socket, a file, MPI). */
typedef struct comet_comm comet_comm;
#define COMET_COMM_ID_BYTES 128
COMET_API int comet_comm_unique_id(uint8_t* out_id128);                                   /* rank 0: ncclGetUniqueId */
COMET_API int comet_comm_create(comet_ctx* ctx, const uint8_t* id128, int32_t rank, int32_t world, comet_comm** out); /* collective */
COMET_API int comet_comm_destroy(comet_comm* comm);
COMET_API int comet_comm_rank(const comet_comm* comm);
COMET_API int comet_comm_world(const comet_comm* comm);
/* host-value all-reduce (op 0 = max, 1 = sum); returns when every rank's contribution is in: a barrier */
COMET_API int comet_comm_allreduce_f64(comet_comm* comm, double* inout, int32_t op);
COMET_API int comet_comm_barrier(comet_comm* comm);
COMET_API int comet_comm_sync(comet_comm* comm);          /* wait for the context's stream and the exchange stream */
/* Sharded search. `_async` enqueues this rank's shard search and returns a ticket (up to 4 in flight); `_wait` makes the
 * local results final, enqueues all-gather + merge on the exchange stream and — if block != 0 — returns when the MERGED
 * rows (same layout and meaning as comet_index_search_dev's outputs) are in the output buffers; with block == 0 they are
 * final after the next blocking wait / comet_comm_sync. Every rank must issue the same sequence of calls.
 * A rank whose own shard search fails after the batch was entered (out of memory, a shape its kernels refuse) still takes
 * part in every collective of the batch: `_async` returns COMET_OK there, the rank's block carries out_counts = -code for
 * every query — which the merge hands to EVERY rank — and its `_wait` returns the error. No rank is left waiting. */
/* List sharding for IVF / IVFPQ (call before the first Add, after or before Train; every rank trains on the same vectors —
 * the GPU k-means is deterministic, so centroids and codebooks are replicated bit for bit — and is handed EVERY vector):
 * this rank keeps only the members of the lists it owns. Lists are dealt to the ranks BY LENGTH: longest first to the rank
 * with the fewest estimated rows (LPT over the members per list seen in training — identical on every rank because training
 * is; before training, and on a rank that loaded its quantisers with ReadFrom instead of training, list l belongs to rank
 * l % world). All ranks rank all centroids and pick the same nprobe lists; lists a rank does not own are empty there, so
 * its table build, scan and selection cover 1/world of the work. Remove / filters act on the local members (a Remove of an
 * id stored on another rank reports NOT_FOUND here).
 * The placement is host-side state, NOT part of the reference's on-disk layouts: a host that checkpoints a sharded index
 * saves comet_index_get_list_owners() next to the shard files and calls comet_index_set_list_owners() on every rank that
 * loads one. The first sharded search of an index on a communicator compares a fingerprint of the placement across the
 * ranks and fails on every rank (COMET_ERR_INVALID_ARG) if they disagree. */
COMET_API int comet_index_set_shard(comet_index* idx, int32_t rank, int32_t world);
COMET_API int comet_index_get_list_owners(const comet_index* idx, int32_t* out_owners, int32_t n_lists);   /* out_owners[l] = rank owning list l */
COMET_API int comet_index_set_list_owners(comet_index* idx, const int32_t* owners, int32_t n_lists);       /* after set_shard, before the first Add */
COMET_API int comet_index_search_sharded_async(comet_index* idx, comet_comm* comm, const float* queries_dev, int32_t B,
                                               const comet_search_params* p, uint32_t* out_ids_dev, float* out_scores_dev,
                                               int32_t* out_counts_dev, int32_t k_cap, uint64_t* out_ticket);
COMET_API int comet_index_search_sharded_wait(comet_index* idx, comet_comm* comm, uint64_t ticket, int32_t block);

/* ---- introspection --------------------------------------------------------------------------- */
/* Read back trained state (host copies). centroids: nlist x dim, codebooks: M x Ksub x dsub. */
COMET_API int comet_index_get_centroids(const comet_index* idx, float* out);
COMET_API int comet_index_get_codebooks(const comet_index* idx, float* out);
COMET_API int comet_index_list_size(const comet_index* idx, int32_t list, int64_t* out);
/* ids (and, for PQ/IVFPQ, M-byte codes; for Flat/IVF, dim-float vectors) of one inverted list in Add
 * order; list = 0 for Flat/PQ. Any output pointer may be NULL. */
COMET_API int comet_index_list_read(const comet_index* idx, int32_t list, uint32_t* out_ids, uint8_t* out_codes,
                                    float* out_vecs);

/* ---- BM25 text index (TextIndex index.go:65-81, bm25_index.go, bm25_index_search.go) ----------------
 * Tokenisation / NFKC / lower-casing stay in Go (third-party uax29, x/text): documents and queries are
 * passed as token ids. Scoring is float64, term-at-a-time in query-token order like the reference. */
typedef struct comet_text_index comet_text_index;
COMET_API int comet_bm25_create(comet_ctx* ctx, comet_text_index** out);                 /* NewBM25SearchIndex */
COMET_API int comet_bm25_destroy(comet_text_index* idx);
COMET_API int comet_bm25_add(comet_text_index* idx, uint32_t doc_id, const uint32_t* tokens, int32_t n); /* Add bm25_index.go:168 */
COMET_API int comet_bm25_remove(comet_text_index* idx, uint32_t doc_id);                 /* soft delete bm25_index.go:203 */
COMET_API int comet_bm25_flush(comet_text_index* idx);                                   /* Flush bm25_index.go:374 */
COMET_API int64_t comet_bm25_num_docs(const comet_text_index* idx);
COMET_API double comet_bm25_avg_doc_len(const comet_text_index* idx);
/* B queries: tokens of query b are q_tokens[q_offsets[b] .. q_offsets[b+1]). k <= 0 or >= hits => all.
 * out_scores: float32(score) like TextResult.Score; out_scores64 (nullable): the float64 accumulators.
 * Rows are sorted by score descending (ties: ascending doc id). k_cap <= 2048. All pointers are HOST memory. */
COMET_API int comet_bm25_search(comet_text_index* idx, const uint32_t* q_tokens, const int32_t* q_offsets, int32_t B, int32_t k,
                                const uint32_t* filter_ids, int32_t n_filter, uint32_t* out_ids, float* out_scores,
                                double* out_scores64, int32_t* out_counts, int32_t k_cap);

/* Hybrid search with Reciprocal Rank Fusion on the device — hybridSearch.Execute (hybrid_search_index.go:477-615) with
 * WithFusionKind(ReciprocalRankFusion) for B (vector, text) query pairs: both sub-searches cut to k (:518, :555), a hit's
 * 0-based rank is its position in its list (scoreMapToRanks fusion.go:205-243 on lists that arrive sorted), a document's
 * score is the float64 sum of 1 / (rrf_k + rank) over the lists it appears in (vector term first: fusion.go:174-203), the
 * fused list is sorted by score descending (ties: vector hits in their order, then text-only hits in theirs) and cut to k.
 * The vector leg runs on a second execution lane beside the text leg, the fusion is one wave per query, the host receives
 * one block: out_ids / out_scores (float64) are B x k, out_counts B (a vector-leg search error, e.g. a zero query under
 * cosine, arrives as a negative count). queries: B x dim HOST floats; tokens as for comet_bm25_search. 1 <= k <= 64;
 * nprobes / ef_search as in comet_search_params (0 = the index default). Metadata filters (WithDocumentIDs) stay with the
 * two-call form (comet_index_search + comet_bm25_search + a host-side fusion). */
COMET_API int comet_hybrid_rrf_search(comet_index* vec, comet_text_index* txt, const float* queries, const uint32_t* q_tokens,
                                      const int32_t* q_offsets, int32_t B, int32_t k, int32_t nprobes, int32_t ef_search, double rrf_k,
                                      uint32_t* out_ids, double* out_scores, int32_t* out_counts);

/* WithNode(nodeIDs...): the stored (preprocessed) vectors of the given node ids, n x dim, in order — lookupNodeVectors
 * flat_index_search.go:171-196, ivf_index_search.go:176-206, hnsw_index_search.go:212-226 (first match in the reference's
 * scan order; "node ID %d not found in index" / "... (deleted)"). PQ / IVFPQ keep codes only: UNSUPPORTED. */
COMET_API int comet_index_fetch_vectors(comet_index* idx, const uint32_t* ids, int32_t n, float* out_vecs);

/* Bulk export in arrival (Add) order: ids[n], list index per element lists[n] (0 for Flat / PQ), and M-byte
 * PQ codes codes[n*M] (PQ / IVFPQ only). Any pointer may be NULL. Used to hand a GPU-built index to the CPU
 * reference path (the CPU baseline searches the index the GPU built). */
COMET_API int comet_index_export(const comet_index* idx, uint32_t* out_ids, int32_t* out_lists, uint8_t* out_codes);

/* named counters of the last search / of the index (bench + tests): "fast_candidates", "fast_overflows",
 * "fast_expansions", "fast_queries", "strict_queries", "max_abs", "max_norm2". Unknown name -> INVALID_ARG. */
COMET_API int comet_index_get_stat(const comet_index* idx, const char* name, double* out);

#ifdef __cplusplus
}
#endif
#endif /* COMET_GPU_H */
