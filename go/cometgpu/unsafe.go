package cometgpu

import "unsafe"

// unsafe_ptr keeps the cgo call sites of comm.go readable (device addresses travel as uintptr between calls).
type unsafe_ptr = unsafe.Pointer
