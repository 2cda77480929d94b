package cometgpu

/*
#cgo LDFLAGS: -lcomet_hip
#include <stdlib.h>
#include "comet_gpu.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"

	comet "github.com/wizenheimer/comet"
)

// Context is one GPU: HIP device + stream + scratch arena (comet_ctx). Share one Context between all indexes that
// live on the same device.
type Context struct{ h *C.comet_ctx }

// NewContext binds device `device` (must be a gfx950 part; there is no CPU fallback).
func NewContext(device int) (*Context, error) {
	c := &Context{}
	if rc := C.comet_ctx_create(C.int(device), &c.h); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	runtime.SetFinalizer(c, func(c *Context) { c.Close() })
	return c, nil
}

// Close releases the device resources. Indexes created on the context must be closed first.
func (c *Context) Close() {
	if c.h != nil {
		C.comet_ctx_destroy(c.h)
		c.h = nil
	}
}

// SetLanes sets the number of execution lanes (1 .. 8, default 8): the streams the asynchronous searches of an index rotate through.
// The synchronous Execute() path of this package always runs on lane 0; the setting matters to callers of the asynchronous C entry points.
func (c *Context) SetLanes(lanes int) error {
	if rc := C.comet_ctx_set_lanes(c.h, C.int32_t(lanes)); rc != C.COMET_OK {
		return lastError(rc)
	}
	return nil
}

// Fence marks everything queued so far on the context's lane-0 stream (comet_ctx_stream) as a dependency of every asynchronous search
// enqueued afterwards. Only callers that run their own HIP work on that stream need it; the library's own calls place the fence themselves.
func (c *Context) Fence() error {
	if rc := C.comet_ctx_fence(c.h); rc != C.COMET_OK {
		return lastError(rc)
	}
	return nil
}

// lastError maps a comet_status to the error values the reference returns.
func lastError(rc C.int) error {
	msg := C.GoString(C.comet_last_error())
	switch rc {
	case C.COMET_OK:
		return nil
	case C.COMET_ERR_ZERO_VECTOR:
		return comet.ErrZeroVector // sentinel compared with == in the reference's tests (distance_test.go:450)
	case C.COMET_ERR_UNKNOWN_METRIC:
		return comet.ErrUnknownDistanceKind
	default:
		return errors.New(msg)
	}
}

func metricCode(kind comet.DistanceKind) (C.int, error) {
	switch kind {
	case comet.Euclidean:
		return C.COMET_L2, nil
	case comet.L2Squared:
		return C.COMET_L2SQ, nil
	case comet.Cosine:
		return C.COMET_COSINE, nil
	}
	return 0, comet.ErrUnknownDistanceKind
}

// Distance evaluates comet.Distance.Calculate on the device kernels (distance.go:114/158/201) — for parity checks.
func (c *Context) Distance(kind comet.DistanceKind, a, b []float32) (float32, error) {
	m, err := metricCode(kind)
	if err != nil {
		return 0, err
	}
	if len(a) != len(b) {
		return 0, fmt.Errorf("dimension mismatch: %d vs %d", len(a), len(b))
	}
	if len(a) == 0 {
		return 0, nil
	}
	var out C.float
	if rc := C.comet_distance(c.h, m, (*C.float)(&a[0]), (*C.float)(&b[0]), C.int(len(a)), &out); rc != C.COMET_OK {
		return 0, lastError(rc)
	}
	return float32(out), nil
}
