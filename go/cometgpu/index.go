package cometgpu

/*
#include "comet_gpu.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"sync"
	"unsafe"

	comet "github.com/wizenheimer/comet"
)

// vectorIndex implements comet.VectorIndex for every GPU index kind; the kind-specific state lives behind comet_index.
type vectorIndex struct {
	mu      sync.RWMutex // searches share RLock; Train / Add / Flush / ReadFrom take Lock (flat_index.go:93)
	ctx     *Context
	h       *C.comet_index
	dim     int
	kind    comet.VectorIndexKind
	dist    comet.DistanceKind
	nlist   int // IVF / IVFPQ
	efS     int // HNSW default efSearch
	keepVec bool
}

var _ comet.VectorIndex = (*vectorIndex)(nil)

func (ix *vectorIndex) finish() *vectorIndex {
	runtime.SetFinalizer(ix, func(ix *vectorIndex) { ix.Close() })
	return ix
}

// Close frees the device memory of the index.
func (ix *vectorIndex) Close() {
	ix.mu.Lock()
	defer ix.mu.Unlock()
	if ix.h != nil {
		C.comet_index_destroy(ix.h)
		ix.h = nil
	}
}

// NewFlatIndex mirrors comet.NewFlatIndex(dim, distanceKind) (flat_index.go:118).
func NewFlatIndex(ctx *Context, dim int, distanceKind comet.DistanceKind) (comet.VectorIndex, error) {
	if dim <= 0 {
		return nil, fmt.Errorf("dimension must be positive")
	}
	m, err := metricCode(distanceKind)
	if err != nil {
		return nil, err
	}
	ix := &vectorIndex{ctx: ctx, dim: dim, kind: comet.FlatIndexKind, dist: distanceKind}
	if rc := C.comet_flat_create(ctx.h, C.int(dim), m, &ix.h); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	return ix.finish(), nil
}

// NewIVFIndex mirrors comet.NewIVFIndex(dim, nlist, distanceKind) (ivf_index.go:147).
func NewIVFIndex(ctx *Context, dim int, nlist int, distanceKind comet.DistanceKind) (comet.VectorIndex, error) {
	m, err := metricCode(distanceKind)
	if err != nil {
		return nil, err
	}
	ix := &vectorIndex{ctx: ctx, dim: dim, kind: comet.IVFIndexKind, dist: distanceKind, nlist: nlist}
	if rc := C.comet_ivf_create(ctx.h, C.int(dim), m, C.int(nlist), &ix.h); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	return ix.finish(), nil
}

// NewPQIndex mirrors comet.NewPQIndex(dim, distanceKind, M, Nbits) (pq_index.go:135).
func NewPQIndex(ctx *Context, dim int, distanceKind comet.DistanceKind, M int, Nbits int) (comet.VectorIndex, error) {
	m, err := metricCode(distanceKind)
	if err != nil {
		return nil, err
	}
	ix := &vectorIndex{ctx: ctx, dim: dim, kind: comet.PQIndexKind, dist: distanceKind}
	if rc := C.comet_pq_create(ctx.h, C.int(dim), m, C.int(M), C.int(Nbits), &ix.h); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	return ix.finish(), nil
}

// NewIVFPQIndex mirrors comet.NewIVFPQIndex(dim, distanceKind, nlist, m, nbits) (ivfpq_index.go:114).
func NewIVFPQIndex(ctx *Context, dim int, distanceKind comet.DistanceKind, nlist int, m int, nbits int) (comet.VectorIndex, error) {
	mc, err := metricCode(distanceKind)
	if err != nil {
		return nil, err
	}
	ix := &vectorIndex{ctx: ctx, dim: dim, kind: comet.IVFPQIndexKind, dist: distanceKind, nlist: nlist}
	if rc := C.comet_ivfpq_create(ctx.h, C.int(dim), mc, C.int(nlist), C.int(m), C.int(nbits), &ix.h); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	return ix.finish(), nil
}

// NewHNSWIndex mirrors comet.NewHNSWIndex(dim, distanceKind, m, efConstruction, efSearch) (hnsw_index.go:172); zero
// values select the reference's defaults (16 / 200 / efConstruction).
func NewHNSWIndex(ctx *Context, dim int, distanceKind comet.DistanceKind, m, efConstruction, efSearch int) (comet.VectorIndex, error) {
	if dim <= 0 {
		return nil, fmt.Errorf("dimension must be positive")
	}
	mc, err := metricCode(distanceKind)
	if err != nil {
		return nil, err
	}
	ix := &vectorIndex{ctx: ctx, dim: dim, kind: comet.HNSWIndexKind, dist: distanceKind}
	if rc := C.comet_hnsw_create(ctx.h, C.int(dim), mc, C.int(m), C.int(efConstruction), C.int(efSearch), &ix.h); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	return ix.finish(), nil
}

// ---- comet.VectorIndex ---------------------------------------------------------------------------------------------

// Train: FlatIndex / HNSW no-op; IVF k-means; PQ codebooks; IVFPQ both (ivf_index.go:206, pq_index.go:193, ivfpq_index.go:180).
func (ix *vectorIndex) Train(vectors []comet.VectorNode) error {
	if ix.kind == comet.FlatIndexKind || ix.kind == comet.HNSWIndexKind {
		return nil
	}
	flat := make([]float32, 0, len(vectors)*ix.dim)
	for _, v := range vectors {
		if len(v.Vector()) != ix.dim {
			return fmt.Errorf("vector dimension mismatch: expected %d, got %d", ix.dim, len(v.Vector()))
		}
		flat = append(flat, v.Vector()...)
	}
	ix.mu.Lock()
	defer ix.mu.Unlock()
	var p *C.float
	if len(flat) > 0 {
		p = (*C.float)(&flat[0])
	}
	return lastError(C.comet_index_train(ix.h, p, C.int64_t(len(vectors))))
}

// Add inserts one vector. Like the reference, a cosine index normalises the caller's slice in place (flat_index.go:182).
func (ix *vectorIndex) Add(vector comet.VectorNode) error {
	v := vector.Vector()
	if len(v) != ix.dim {
		return fmt.Errorf("vector dimension mismatch: expected %d, got %d", ix.dim, len(v))
	}
	id := C.uint32_t(vector.ID())
	ix.mu.Lock()
	defer ix.mu.Unlock()
	var added C.int64_t
	// normalized_out aliases the input: the library copies the input before it writes the preprocessed row back
	rc := C.comet_index_add(ix.h, &id, (*C.float)(&v[0]), 1, &added, (*C.float)(&v[0]))
	return lastError(rc)
}

// AddBatch is the GPU-friendly form of n sequential Add calls (same order, stops at the first failing vector).
func (ix *vectorIndex) AddBatch(vectors []comet.VectorNode) (added int, err error) {
	ids := make([]uint32, len(vectors))
	flat := make([]float32, 0, len(vectors)*ix.dim)
	for i, v := range vectors {
		if len(v.Vector()) != ix.dim {
			return 0, fmt.Errorf("vector dimension mismatch: expected %d, got %d", ix.dim, len(v.Vector()))
		}
		ids[i] = v.ID()
		flat = append(flat, v.Vector()...)
	}
	if len(vectors) == 0 {
		return 0, nil
	}
	ix.mu.Lock()
	defer ix.mu.Unlock()
	var n C.int64_t
	rc := C.comet_index_add(ix.h, (*C.uint32_t)(&ids[0]), (*C.float)(&flat[0]), C.int64_t(len(vectors)), &n, (*C.float)(&flat[0]))
	if ix.dist == comet.Cosine { // write the normalised rows back into the callers' slices, as Add does
		for i := 0; i < int(n); i++ {
			copy(vectors[i].Vector(), flat[i*ix.dim:(i+1)*ix.dim])
		}
	}
	return int(n), lastError(rc)
}

// Remove soft-deletes by id (flat_index.go:216-249): read-check, then a short write.
func (ix *vectorIndex) Remove(vector comet.VectorNode) error {
	ix.mu.Lock()
	defer ix.mu.Unlock()
	return lastError(C.comet_index_remove(ix.h, C.uint32_t(vector.ID())))
}

// Flush hard-deletes soft-deleted vectors (flat_index.go:268-296; HNSW: graph repair hnsw_index.go:348-431).
func (ix *vectorIndex) Flush() error {
	ix.mu.Lock()
	defer ix.mu.Unlock()
	return lastError(C.comet_index_flush(ix.h))
}

func (ix *vectorIndex) NewSearch() comet.VectorSearch {
	s := &vectorSearch{index: ix, k: 10, cutoff: -1}
	switch ix.kind { // per-kind defaults: ivf_index.go:406-413, ivfpq_index.go:442-449
	case comet.IVFIndexKind, comet.IVFPQIndexKind:
		s.nProbes = int(C.comet_index_default_nprobes(ix.h))
	}
	return s
}

func (ix *vectorIndex) Dimensions() int                   { return ix.dim }
func (ix *vectorIndex) DistanceKind() comet.DistanceKind   { return ix.dist }
func (ix *vectorIndex) Kind() comet.VectorIndexKind        { return ix.kind }
func (ix *vectorIndex) Trained() bool                      { return C.comet_index_trained(ix.h) != 0 }
func (ix *vectorIndex) Len() int                           { return int(C.comet_index_size(ix.h)) }

// SetShard enables multi-GPU list sharding on an IVF / IVFPQ index (comet_index_set_shard) — call before the first Add.
func (ix *vectorIndex) SetShard(rank, world int) error {
	return lastError(C.comet_index_set_shard(ix.h, C.int32_t(rank), C.int32_t(world)))
}

// ListOwners returns the rank owning each inverted list (comet_index_get_list_owners): host-side state that is not part of the
// reference's on-disk layouts — a sharded checkpoint saves it beside the shard files.
func (ix *vectorIndex) ListOwners(nlist int) ([]int32, error) {
	out := make([]int32, nlist)
	if nlist == 0 {
		return out, nil
	}
	if err := lastError(C.comet_index_get_list_owners(ix.h, (*C.int32_t)(unsafe.Pointer(&out[0])), C.int32_t(nlist))); err != nil {
		return nil, err
	}
	return out, nil
}

// SetListOwners hands a rank that LOADED its quantisers (ReadFrom) the placement the ranks that trained derived (comet_index_set_list_owners).
func (ix *vectorIndex) SetListOwners(owners []int32) error {
	if len(owners) == 0 {
		return nil
	}
	return lastError(C.comet_index_set_list_owners(ix.h, (*C.int32_t)(unsafe.Pointer(&owners[0])), C.int32_t(len(owners))))
}

// nodeVectors = lookupNodeVectors (flat_index_search.go:171-196): the stored vectors behind WithNode(ids...).
func (ix *vectorIndex) nodeVectors(ids []uint32) ([][]float32, error) {
	if len(ids) == 0 {
		return nil, nil
	}
	buf := make([]float32, len(ids)*ix.dim)
	if rc := C.comet_index_fetch_vectors(ix.h, (*C.uint32_t)(&ids[0]), C.int32_t(len(ids)), (*C.float)(&buf[0])); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	out := make([][]float32, len(ids))
	for i := range ids {
		out[i] = buf[i*ix.dim : (i+1)*ix.dim]
	}
	return out, nil
}

var _ = unsafe.Pointer(nil)

// ---- HNSW construction extras (no counterpart in the reference, whose randomLevel draws from the unseeded global RNG) ------

// HNSWBuilder is implemented by the HNSW index returned from NewHNSWIndex: Add (VectorIndex) inserts on the GPU with levels
// from the index's own seeded stream; these two make a build reproducible.
type HNSWBuilder interface {
	// AddWithLevels inserts the nodes in order with the given hnswNode levels (insertNode hnsw_index.go:493-552).
	AddWithLevels(nodes []comet.VectorNode, levels []int32) error
	// SetLevelSeed seeds the index's level stream (geometric, p = 1/M, capped at 16: randomLevel hnsw_index.go:474-484).
	SetLevelSeed(seed uint64) error
}

func (ix *vectorIndex) AddWithLevels(nodes []comet.VectorNode, levels []int32) error {
	if ix.kind != comet.HNSWIndexKind {
		return fmt.Errorf("AddWithLevels: not an HNSW index")
	}
	if len(nodes) != len(levels) {
		return fmt.Errorf("AddWithLevels: %d nodes, %d levels", len(nodes), len(levels))
	}
	if len(nodes) == 0 {
		return nil
	}
	ids := make([]uint32, len(nodes))
	flat := make([]float32, 0, len(nodes)*ix.dim)
	for i, v := range nodes {
		if len(v.Vector()) != ix.dim {
			return fmt.Errorf("vector dimension mismatch: expected %d, got %d", ix.dim, len(v.Vector()))
		}
		ids[i] = v.ID()
		flat = append(flat, v.Vector()...)
	}
	ix.mu.Lock()
	defer ix.mu.Unlock()
	var added C.int64_t
	return lastError(C.comet_hnsw_add_with_levels(ix.h, (*C.uint32_t)(&ids[0]), (*C.float)(&flat[0]), (*C.int32_t)(&levels[0]), C.int64_t(len(nodes)), &added))
}

func (ix *vectorIndex) SetLevelSeed(seed uint64) error {
	if ix.kind != comet.HNSWIndexKind {
		return fmt.Errorf("SetLevelSeed: not an HNSW index")
	}
	ix.mu.Lock()
	defer ix.mu.Unlock()
	return lastError(C.comet_hnsw_set_level_seed(ix.h, C.uint64_t(seed)))
}
