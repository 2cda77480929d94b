package cometgpu

/*
#include "comet_gpu.h"
*/
import "C"

import "fmt"

// Batched forms of comet.Norm / comet.Normalize / comet.Scale (distance.go:312-428) over n dense rows of d floats. The arithmetic is the
// reference's (serial float32 sum, float32(sqrt(float64)), multiplication by 1 / norm; Normalize returns a zero row unchanged), so
// every element is bit-identical to calling the Go function row by row; worth it only for batches (one upload, one download).
func NormBatch(ctx *Context, rows []float32, d int) ([]float32, error) {
	n, err := rowCount(rows, d)
	if err != nil || n == 0 {
		return nil, err
	}
	out := make([]float32, n)
	if rc := C.comet_norm_batch(ctx.h, (*C.float)(&rows[0]), C.int64_t(n), C.int32_t(d), (*C.float)(&out[0])); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	return out, nil
}

func NormalizeBatch(ctx *Context, rows []float32, d int) ([]float32, error) {
	n, err := rowCount(rows, d)
	if err != nil || n == 0 {
		return nil, err
	}
	out := make([]float32, len(rows))
	if rc := C.comet_normalize_batch(ctx.h, (*C.float)(&rows[0]), C.int64_t(n), C.int32_t(d), (*C.float)(&out[0])); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	return out, nil
}

func ScaleBatch(ctx *Context, rows []float32, d int, scalar float32) ([]float32, error) {
	n, err := rowCount(rows, d)
	if err != nil || n == 0 {
		return nil, err
	}
	out := make([]float32, len(rows))
	if rc := C.comet_scale_batch(ctx.h, (*C.float)(&rows[0]), C.int64_t(n), C.int32_t(d), C.float(scalar), (*C.float)(&out[0])); rc != C.COMET_OK {
		return nil, lastError(rc)
	}
	return out, nil
}

func rowCount(rows []float32, d int) (int, error) {
	if d <= 0 || len(rows)%d != 0 {
		return 0, fmt.Errorf("rows must hold a whole number of %d-float vectors", d)
	}
	return len(rows) / d, nil
}
