package cometgpu

/*
#include <stdint.h>
#include "comet_gpu.h"

// trampolines: cgo cannot take the address of a Go function directly, so the exported Go callbacks are wrapped here
extern int cometgpuGoWrite(uintptr_t handle, void* data, size_t len);
extern int cometgpuGoRead(uintptr_t handle, void* dst, size_t len);
static int cometgpu_write_cb(void* user, const void* data, size_t len) { return cometgpuGoWrite((uintptr_t)user, (void*)data, len); }
static int cometgpu_read_cb(void* user, void* dst, size_t len) { return cometgpuGoRead((uintptr_t)user, dst, len); }
static int cometgpu_write_to(comet_index* idx, uintptr_t h, int64_t* n) { return comet_index_write_to(idx, cometgpu_write_cb, (void*)h, n); }
static int cometgpu_read_from(comet_index* idx, uintptr_t h, int64_t* n) { return comet_index_read_from(idx, cometgpu_read_cb, (void*)h, n); }
*/
import "C"

import (
	"io"
	"runtime/cgo"
	"unsafe"
)

type ioState struct {
	w   io.Writer
	r   io.Reader
	err error
}

//export cometgpuGoWrite
func cometgpuGoWrite(handle C.uintptr_t, data unsafe.Pointer, n C.size_t) C.int {
	st := cgo.Handle(handle).Value().(*ioState)
	if _, err := st.w.Write(unsafe.Slice((*byte)(data), int(n))); err != nil {
		st.err = err
		return 1
	}
	return 0
}

//export cometgpuGoRead
func cometgpuGoRead(handle C.uintptr_t, dst unsafe.Pointer, n C.size_t) C.int {
	st := cgo.Handle(handle).Value().(*ioState)
	if _, err := io.ReadFull(st.r, unsafe.Slice((*byte)(dst), int(n))); err != nil {
		st.err = err
		return 1
	}
	return 0
}

// WriteTo implements io.WriterTo with the reference's own on-disk layout for this index kind ("FLAT" flat_index.go:366,
// "IVFX" ivf_index.go:468, "PQIX" pq_index.go:509, "IVPQ" ivfpq_index.go:544, "HNSW" hnsw_index.go:734): like the
// reference it flushes first, so it takes the write lock.
func (ix *vectorIndex) WriteTo(w io.Writer) (int64, error) {
	ix.mu.Lock()
	defer ix.mu.Unlock()
	st := &ioState{w: w}
	h := cgo.NewHandle(st)
	defer h.Delete()
	var n C.int64_t
	rc := C.cometgpu_write_to(ix.h, C.uintptr_t(h), &n)
	if st.err != nil {
		return int64(n), st.err
	}
	return int64(n), lastError(rc)
}

// ReadFrom implements io.ReaderFrom: parses the reference's layout, validates it against this index's constructor
// parameters and replaces the contents (flat_index.go:488, ivf_index.go:620, pq_index.go:672, ivfpq_index.go:745,
// hnsw_index.go:898). A file written by the pure-Go index loads into the GPU index and vice versa.
func (ix *vectorIndex) ReadFrom(r io.Reader) (int64, error) {
	ix.mu.Lock()
	defer ix.mu.Unlock()
	st := &ioState{r: r}
	h := cgo.NewHandle(st)
	defer h.Delete()
	var n C.int64_t
	rc := C.cometgpu_read_from(ix.h, C.uintptr_t(h), &n)
	if rc != C.COMET_OK && st.err != nil && st.err != io.ErrUnexpectedEOF && st.err != io.EOF {
		return int64(n), st.err
	}
	return int64(n), lastError(rc)
}
