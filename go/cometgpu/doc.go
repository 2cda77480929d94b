// Package cometgpu is the MI355X (gfx950) backend for comet's vector-search hot path.
//
// It implements comet's own interfaces — comet.VectorIndex (index.go:32-63), comet.VectorSearch
// (index_search.go:141-279), comet.TextIndex / comet.TextSearch (index.go:65-81, index_search.go:358-430) — on top of
// libcomet_hip.so (hand-written HIP kernels behind the C ABI of include/comet_gpu.h), so a GPU index drops into
// comet.NewHybridSearchIndex(vec, txt, meta) and the persistence layer unchanged:
//
//	ctx, _ := cometgpu.NewContext(0)
//	idx, _ := cometgpu.NewFlatIndex(ctx, 768, comet.Cosine)          // comet.NewFlatIndex(768, comet.Cosine)
//	idx.Add(*comet.NewVectorNodeWithID(1, vec))
//	res, _ := idx.NewSearch().WithQuery(q).WithK(10).Execute()
//
// What stays in Go is exactly what stays on the host in the reference: multi-query aggregation
// (comet.NewVectorAggregation), comet.LimitResults / comet.AutocutResults, rerankers, argument validation, the
// RWMutex discipline (searches share the read lock; Add / Train / Flush / ReadFrom take the write lock), BM25
// tokenisation (uax29 words + NFKC + lower-casing, bm25_index.go:154-166). All distance arithmetic, list scans,
// graph traversal, BM25 accumulation and top-k selection run on the GPU and return the reference's results bit for bit
// (see tests/ in the backend repository: every path is compared with a CPU restatement of the Go loops).
//
// Place this directory at github.com/wizenheimer/comet/cometgpu (it imports the parent package) and point cgo at the
// backend checkout: CGO_CFLAGS=-I<backend>/include CGO_LDFLAGS="-L<backend>/comet_amd -lcomet_hip".
//
// This package could not be compiled in the backend's build image (no Go toolchain there); the same call sequences
// are exercised against the C ABI by tests/abi_harness.c and by the ctypes-based test-suite.
package cometgpu
