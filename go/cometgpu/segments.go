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

// SegmentSet is the vector leg of persistentHybridSearch.Execute (storage.go:489-626) with the fan-out and the merge on the
// GPU: the vector indexes of the store's memtables and disk segments (oldest first) stay resident in HBM, and one call searches
// every one of them for the query — per-segment top-k, then mergeResults (storage_merge.go:13-46: highest score per document
// id), sortResultsByScore (:50-54: descending) and the cut to k (storage.go:621-623) — instead of len(segments) goroutines,
// len(segments) result slices and a map on the host. A store keeps one SegmentSet and rebuilds it when a memtable is flushed or
// segments are compacted (segmentManager.list(), storage.go:547).
type SegmentSet struct {
	segs    []*vectorIndex
	handles []*C.comet_index
	dim     int
}

// NewSegmentSet takes cometgpu indexes of one dimension that live on one Context.
func NewSegmentSet(indexes ...comet.VectorIndex) (*SegmentSet, error) {
	if len(indexes) == 0 {
		return nil, fmt.Errorf("no segments")
	}
	s := &SegmentSet{}
	for i, v := range indexes {
		ix, ok := v.(*vectorIndex)
		if !ok {
			return nil, fmt.Errorf("segment %d is not a cometgpu index", i)
		}
		if i > 0 && ix.dim != s.dim {
			return nil, fmt.Errorf("segment %d: dimension %d, expected %d", i, ix.dim, s.dim)
		}
		s.dim = ix.dim
		s.segs = append(s.segs, ix)
		s.handles = append(s.handles, ix.h)
	}
	return s, nil
}

// Search returns what persistentHybridSearch.Execute returns for a vector-only query (scores are the per-segment distances,
// sorted DESCENDING as the reference does at this layer). nProbes / efSearch / threshold <= 0: not set (storage.go:522-533).
func (s *SegmentSet) Search(query []float32, k, nProbes, efSearch int, threshold float32, documentIDs ...uint32) ([]comet.HybridSearchResult, error) {
	if len(query) != s.dim {
		return nil, fmt.Errorf("query dimension mismatch: expected %d, got %d", s.dim, len(query))
	}
	if k < 0 {
		return nil, fmt.Errorf("k must not be negative")
	}
	if k == 0 {
		return []comet.HybridSearchResult{}, nil
	}
	for _, ix := range s.segs {
		ix.mu.RLock()
		defer ix.mu.RUnlock()
	}
	p := C.comet_search_params{k: C.int32_t(k), nprobes: C.int32_t(nProbes), ef_search: C.int32_t(efSearch)}
	if threshold > 0 {
		p.threshold = C.float(threshold)
	}
	var pin runtime.Pinner // see vectorSearch.Execute: a Go pointer stored in the parameter block must be pinned
	defer pin.Unpin()
	if len(documentIDs) > 0 {
		pin.Pin(&documentIDs[0])
		p.filter_ids = (*C.uint32_t)(&documentIDs[0])
		p.n_filter = C.int32_t(len(documentIDs))
	}
	ids := make([]uint32, k)
	scores := make([]float32, k)
	var count C.int32_t
	if rc := C.comet_segments_search(&s.handles[0], C.int32_t(len(s.handles)), (*C.float)(&query[0]), 1, &p,
		(*C.uint32_t)(&ids[0]), (*C.float)(&scores[0]), &count, C.int32_t(k)); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	out := make([]comet.HybridSearchResult, 0, int(count))
	for i := 0; i < int(count) && i < k; i++ {
		out = append(out, comet.HybridSearchResult{ID: ids[i], Score: float64(scores[i])})
	}
	return out, nil
}
