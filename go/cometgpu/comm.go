package cometgpu

/*
#include "comet_gpu.h"
*/
import "C"

import (
	"fmt"

	comet "github.com/wizenheimer/comet"
)

// Comm is an RCCL communicator owned by the library (comet_comm): one process per GPU, RCCL over xGMI. The host only
// carries the 128-byte id from rank 0 to the other processes (UniqueID / NewComm), by any side channel it likes.
type Comm struct {
	h   *C.comet_comm
	ctx *Context
}

// UniqueID is called on rank 0 (ncclGetUniqueId); ship the bytes to every other rank.
func UniqueID() ([]byte, error) {
	id := make([]byte, C.COMET_COMM_ID_BYTES)
	if rc := C.comet_comm_unique_id((*C.uint8_t)(&id[0])); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	return id, nil
}

// NewComm is collective: every rank calls it with the same id.
func NewComm(ctx *Context, id []byte, rank, world int) (*Comm, error) {
	if len(id) != C.COMET_COMM_ID_BYTES {
		return nil, fmt.Errorf("communicator id must be %d bytes", C.COMET_COMM_ID_BYTES)
	}
	c := &Comm{ctx: ctx}
	if rc := C.comet_comm_create(ctx.h, (*C.uint8_t)(&id[0]), C.int32_t(rank), C.int32_t(world), &c.h); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	return c, nil
}

func (c *Comm) Close()         { C.comet_comm_destroy(c.h); c.h = nil }
func (c *Comm) Barrier() error { return lastError(C.comet_comm_barrier(c.h)) }

// ShardedSearch searches this rank's shard of a sharded index for the batch, exchanges the per-shard top-K with one RCCL
// all-gather and returns the merged rows (the same on every rank). Flat / PQ indexes are sharded by the caller (each rank
// adds its own contiguous block of rows); IVF / IVFPQ with SetShard(rank, world). Every rank must call it with the same
// queries. The reference's analogue is the per-segment fan-out + mergeResults of storage.go:546-626.
func (c *Comm) ShardedSearch(index comet.VectorIndex, queries [][]float32, k int, nProbes int, threshold float32) ([][]comet.VectorResult, error) {
	ix, ok := index.(*vectorIndex)
	if !ok {
		return nil, fmt.Errorf("not a cometgpu index")
	}
	B := len(queries)
	if B == 0 {
		return nil, nil
	}
	if k <= 0 {
		return nil, fmt.Errorf("sharded search needs an explicit k")
	}
	flat := make([]float32, 0, B*ix.dim)
	for _, q := range queries {
		if len(q) != ix.dim {
			return nil, fmt.Errorf("query dimension mismatch: expected %d, got %d", ix.dim, len(q))
		}
		flat = append(flat, q...)
	}
	ix.mu.RLock()
	defer ix.mu.RUnlock()
	var dq, dids, dsc, dcn C.uintptr_t
	bytes := []C.size_t{C.size_t(B * ix.dim * 4), C.size_t(B * k * 4), C.size_t(B * k * 4), C.size_t(B * 4)}
	ptrs := []*C.uintptr_t{&dq, &dids, &dsc, &dcn}
	for i := range ptrs {
		var p unsafe_ptr
		if rc := C.comet_dev_alloc(c.ctx.h, bytes[i], (*unsafe_ptr)(&p)); rc != C.COMET_OK {
			return nil, lastError(rc)
		}
		*ptrs[i] = C.uintptr_t(uintptr(p))
		defer C.comet_dev_free(c.ctx.h, p)
	}
	if rc := C.comet_memcpy_h2d(c.ctx.h, unsafe_ptr(uintptr(dq)), unsafe_ptr(&flat[0]), bytes[0]); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	p := C.comet_search_params{k: C.int32_t(k), threshold: C.float(threshold), nprobes: C.int32_t(nProbes)}
	var ticket C.uint64_t
	if rc := C.comet_index_search_sharded_async(ix.h, c.h, (*C.float)(unsafe_ptr(uintptr(dq))), C.int32_t(B), &p,
		(*C.uint32_t)(unsafe_ptr(uintptr(dids))), (*C.float)(unsafe_ptr(uintptr(dsc))), (*C.int32_t)(unsafe_ptr(uintptr(dcn))), C.int32_t(k), &ticket); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	if rc := C.comet_index_search_sharded_wait(ix.h, c.h, ticket, 1); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	ids := make([]uint32, B*k)
	scores := make([]float32, B*k)
	counts := make([]int32, B)
	C.comet_memcpy_d2h(c.ctx.h, unsafe_ptr(&ids[0]), unsafe_ptr(uintptr(dids)), bytes[1])
	C.comet_memcpy_d2h(c.ctx.h, unsafe_ptr(&scores[0]), unsafe_ptr(uintptr(dsc)), bytes[2])
	C.comet_memcpy_d2h(c.ctx.h, unsafe_ptr(&counts[0]), unsafe_ptr(uintptr(dcn)), bytes[3])
	out := make([][]comet.VectorResult, B)
	for b := 0; b < B; b++ {
		if counts[b] < 0 {
			return nil, comet.ErrZeroVector
		}
		for i := 0; i < int(counts[b]) && i < k; i++ {
			out[b] = append(out[b], comet.VectorResult{Node: *comet.NewVectorNodeWithID(ids[b*k+i], nil), Score: scores[b*k+i]})
		}
	}
	return out, nil
}
