package cometgpu

/*
#include "comet_gpu.h"
*/
import "C"

import (
	"fmt"
	"runtime"

	comet "github.com/wizenheimer/comet"
)

// vectorSearch implements comet.VectorSearch (index_search.go:141-279): a single-use fluent builder.
type vectorSearch struct {
	index       *vectorIndex
	queries     [][]float32
	nodeIDs     []uint32
	k           int
	nProbes     int
	efSearch    int
	threshold   float32
	aggregation comet.ScoreAggregationKind
	cutoff      int
	documentIDs []uint32
	reranker    comet.Reranker
}

var _ comet.VectorSearch = (*vectorSearch)(nil)

func (s *vectorSearch) WithQuery(queries ...[]float32) comet.VectorSearch { s.queries = queries; return s }
func (s *vectorSearch) WithNode(nodeIDs ...uint32) comet.VectorSearch     { s.nodeIDs = nodeIDs; return s }
func (s *vectorSearch) WithK(k int) comet.VectorSearch                    { s.k = k; return s }
func (s *vectorSearch) WithNProbes(nProbes int) comet.VectorSearch        { s.nProbes = nProbes; return s }
func (s *vectorSearch) WithEfSearch(efSearch int) comet.VectorSearch      { s.efSearch = efSearch; return s }
func (s *vectorSearch) WithThreshold(threshold float32) comet.VectorSearch {
	s.threshold = threshold
	return s
}
func (s *vectorSearch) WithScoreAggregation(kind comet.ScoreAggregationKind) comet.VectorSearch {
	s.aggregation = kind
	return s
}
func (s *vectorSearch) WithCutoff(cutoff int) comet.VectorSearch { s.cutoff = cutoff; return s }
func (s *vectorSearch) WithDocumentIDs(docIDs ...uint32) comet.VectorSearch {
	s.documentIDs = docIDs
	return s
}
func (s *vectorSearch) WithReranker(reranker comet.Reranker) comet.VectorSearch {
	s.reranker = reranker
	return s
}

// Execute mirrors flatIndexSearch.Execute (flat_index_search.go:109-165) and its four siblings: every query (and every
// node-id query) is searched independently — here as ONE batched device call — the per-query rows are concatenated,
// aggregated by node id, limited, auto-cut and re-ranked exactly as the reference does it on the host.
func (s *vectorSearch) Execute() ([]comet.VectorResult, error) {
	if len(s.queries) == 0 && len(s.nodeIDs) == 0 {
		return nil, fmt.Errorf("must specify either queries or node IDs")
	}
	ix := s.index
	ix.mu.RLock()
	defer ix.mu.RUnlock()
	if !ix.Trained() {
		if ix.kind == comet.PQIndexKind {
			return nil, fmt.Errorf("index not trained") // pq_index_search.go:224
		}
		return nil, fmt.Errorf("index must be trained before searching") // ivf_index_search.go:223
	}
	all := make([][]float32, 0, len(s.queries)+len(s.nodeIDs))
	all = append(all, s.queries...)
	if len(s.nodeIDs) > 0 {
		nv, err := ix.nodeVectors(s.nodeIDs)
		if err != nil {
			return nil, err
		}
		all = append(all, nv...)
	}
	B := len(all)
	flat := make([]float32, 0, B*ix.dim)
	for _, q := range all {
		if len(q) != ix.dim {
			return nil, fmt.Errorf("query dimension mismatch: expected %d, got %d", ix.dim, len(q)) // flat_index_search.go:227
		}
		flat = append(flat, q...)
	}
	// rows hold min(count, kCap) results; kCap = what sanitizeK (limiter.go:12-17) can return for this index
	n := ix.Len()
	kCap := s.k
	if kCap <= 0 || kCap > n {
		kCap = n
	}
	if kCap < 1 {
		kCap = 1
	}
	ids := make([]uint32, B*kCap)
	scores := make([]float32, B*kCap)
	counts := make([]int32, B)
	p := C.comet_search_params{k: C.int32_t(s.k), threshold: C.float(s.threshold), nprobes: C.int32_t(s.nProbes), ef_search: C.int32_t(s.efSearch)}
	// p lives in Go memory and p.filter_ids would be a Go pointer stored inside it: cgo's pointer-passing rules (and the
	// default cgocheck) forbid handing C a Go pointer to memory that holds an UNPINNED Go pointer, so the id slice is pinned
	// for the duration of the call (the library copies the ids before it returns and never retains the pointer).
	var pin runtime.Pinner
	defer pin.Unpin()
	if len(s.documentIDs) > 0 {
		pin.Pin(&s.documentIDs[0])
		p.filter_ids = (*C.uint32_t)(&s.documentIDs[0])
		p.n_filter = C.int32_t(len(s.documentIDs))
	}
	rc := C.comet_index_search(ix.h, (*C.float)(&flat[0]), C.int32_t(B), &p, (*C.uint32_t)(&ids[0]), (*C.float)(&scores[0]), (*C.int32_t)(&counts[0]), C.int32_t(kCap))
	if rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	var results []comet.VectorResult
	for b := 0; b < B; b++ {
		c := int(counts[b])
		if c > kCap {
			c = kCap
		}
		for i := 0; i < c; i++ {
			// the node carries its id; vectors stay on the device (fetch them with WithNode-style lookups when a reranker needs them)
			results = append(results, comet.VectorResult{Node: *comet.NewVectorNodeWithID(ids[b*kCap+i], nil), Score: scores[b*kCap+i]})
		}
	}
	aggKind := s.aggregation
	if aggKind == "" {
		aggKind = comet.SumAggregation
	}
	agg, err := comet.NewVectorAggregation(aggKind) // aggregation.go:72
	if err != nil {
		return nil, err
	}
	results = agg.Aggregate(results)
	results = comet.LimitResults(results, s.k) // limiter.go:28
	if s.cutoff != -1 {
		results = comet.AutocutResults(results, s.cutoff) // limiter.go:52
	}
	if s.reranker != nil { // flat_index_search.go:160-163
		if s.rerankerNeedsVectors() {
			if err := s.attachVectors(results); err != nil {
				return nil, err
			}
		}
		results = s.reranker.Rerank(results)
	}
	return results, nil
}

// rerankerNeedsVectors: rerankers see VectorResult.Node; index kinds that keep vectors (Flat / IVF / HNSW) can supply them.
func (s *vectorSearch) rerankerNeedsVectors() bool {
	k := s.index.kind
	return k == comet.FlatIndexKind || k == comet.IVFIndexKind || k == comet.HNSWIndexKind
}

func (s *vectorSearch) attachVectors(results []comet.VectorResult) error {
	if len(results) == 0 {
		return nil
	}
	ids := make([]uint32, len(results))
	for i, r := range results {
		ids[i] = r.Node.ID()
	}
	vecs, err := s.index.nodeVectors(ids)
	if err != nil {
		return err
	}
	for i := range results {
		results[i].Node = *comet.NewVectorNodeWithID(ids[i], vecs[i])
	}
	return nil
}
