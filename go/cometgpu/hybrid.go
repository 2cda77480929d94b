package cometgpu

/*
#include "comet_gpu.h"
*/
import "C"

import (
	"fmt"

	comet "github.com/wizenheimer/comet"
)

// HybridRRFSearch runs a BATCH of hybrid searches with Reciprocal Rank Fusion in one call on the GPU (comet_hybrid_rrf_search):
// what len(queries) calls of
//
//	comet.NewHybridSearchIndex(vec, txt, nil).NewSearch().WithVector(q).WithText(text).WithK(k).WithNProbes(nProbes).
//		WithFusionKind(comet.ReciprocalRankFusion).Execute()
//
// return (hybrid_search_index.go:477-615): both legs cut to k, ranks = positions, score = the float64 sum of 1/(60+rank), sorted
// descending, cut to k — the vector leg on a second execution lane beside the text leg, the fusion one wave per query, one block
// of results back. vec and txt must live on one Context. Metadata filters keep the per-query form (the reference's
// hybridSearch.Execute over the cometgpu indexes, unchanged).
func HybridRRFSearch(vec comet.VectorIndex, txt *BM25SearchIndex, queries [][]float32, texts []string, k, nProbes, efSearch int) ([][]comet.HybridSearchResult, error) {
	ix, ok := vec.(*vectorIndex)
	if !ok {
		return nil, fmt.Errorf("the vector index is not a cometgpu index")
	}
	B := len(queries)
	if B == 0 {
		return nil, nil
	}
	if len(texts) != B {
		return nil, fmt.Errorf("one text query per vector query: %d vs %d", len(texts), B)
	}
	if k < 1 || k > 64 {
		return nil, fmt.Errorf("the device fusion ranks 1..64 hits per leg, got k = %d", k)
	}
	flat := make([]float32, 0, B*ix.dim)
	for i, q := range queries {
		if len(q) != ix.dim {
			return nil, fmt.Errorf("query %d: dimension mismatch: expected %d, got %d", i, ix.dim, len(q))
		}
		flat = append(flat, q...)
	}
	txt.mu.RLock()
	defer txt.mu.RUnlock()
	ix.mu.RLock()
	defer ix.mu.RUnlock()
	offs := make([]int32, B+1)
	var qtok []uint32
	for b, t := range texts {
		qtok = append(qtok, txt.tokenIDs(tokenize(normalize(t)), false)...)
		offs[b+1] = int32(len(qtok))
	}
	var qp *C.uint32_t
	if len(qtok) > 0 {
		qp = (*C.uint32_t)(&qtok[0])
	}
	ids := make([]uint32, B*k)
	scores := make([]float64, B*k)
	counts := make([]int32, B)
	rc := C.comet_hybrid_rrf_search(ix.h, txt.h, (*C.float)(&flat[0]), qp, (*C.int32_t)(&offs[0]), C.int32_t(B), C.int32_t(k), C.int32_t(nProbes),
		C.int32_t(efSearch), C.double(60.0), (*C.uint32_t)(&ids[0]), (*C.double)(&scores[0]), (*C.int32_t)(&counts[0]))
	if rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	out := make([][]comet.HybridSearchResult, B)
	for b := 0; b < B; b++ {
		if counts[b] < 0 { // a search-time error of the vector leg (ErrZeroVector under cosine)
			if -counts[b] == int32(C.COMET_ERR_ZERO_VECTOR) {
				return nil, comet.ErrZeroVector
			}
			return nil, fmt.Errorf("hybrid search: query %d failed with status %d", b, -counts[b])
		}
		res := make([]comet.HybridSearchResult, 0, counts[b])
		for i := 0; i < int(counts[b]); i++ {
			res = append(res, comet.HybridSearchResult{ID: ids[b*k+i], Score: scores[b*k+i]})
		}
		out[b] = res
	}
	return out, nil
}
