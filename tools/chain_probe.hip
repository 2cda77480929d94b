// chain_probe.hip — what a serial float32 sum costs one wave on gfx950: the HNSW distance slice (64 products from LDS, 64 dependent adds) in isolation.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/chain_probe.hip -o tools/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* ticks, int iters, int active) {
    __shared__ __attribute__((aligned(16))) float tile[64 * 68];
    __shared__ __attribute__((aligned(16))) float qs[64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 68; i += 64) tile[i] = (float)(i % 17) * 0.01f;
    qs[lane] = 0.5f + lane * 0.001f;
    __syncthreads();
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (lane < active) {
            const float* tp = &tile[lane * 68];
            f32x4 xv[16], qq[16];
#pragma unroll
            for (int i = 0; i < 16; i++) xv[i] = *reinterpret_cast<const f32x4*>(tp + 4 * i);
#pragma unroll
            for (int i = 0; i < 16; i++) qq[i] = *reinterpret_cast<const f32x4*>(qs + 4 * i);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (MODE == 0) {        // serial chain
                    const f32x2 qa = {qq[i][0], qq[i][1]}, qb = {qq[i][2], qq[i][3]}, xa = {xv[i][0], xv[i][1]}, xc = {xv[i][2], xv[i][3]};
                    const f32x2 da = qa - xa, db = qb - xc; const f32x2 ta = da * da, tb = db * db;
                    acc = acc + ta[0]; acc = acc + ta[1]; acc = acc + tb[0]; acc = acc + tb[1];
                } else {                // no chain: independent partial sums (NOT the reference's order; the floor of the loads + products)
                    const f32x4 d = qq[i] - xv[i]; const f32x4 t = d * d; acc += (t[0] + t[1]) + (t[2] + t[3]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) tile[(it & 63) * 68] = acc * 1e-9f;      // keeps the loop's loads from being hoisted
        __builtin_amdgcn_wave_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* tk; hipMalloc(&out, 256 * 64 * 4); hipMalloc(&tk, 256 * 8);
    const int iters = 20000;
    for (int mode = 0; mode < 2; mode++) for (int active : {32, 64}) for (int blocks : {1, 256, 1024}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        if (mode == 0) k<0><<<blocks, 64>>>(out, tk, 100, active); else k<1><<<blocks, 64>>>(out, tk, 100, active);
        hipEventRecord(a);
        if (mode == 0) k<0><<<blocks, 64>>>(out, tk, iters, active); else k<1><<<blocks, 64>>>(out, tk, iters, active);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long h; hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
        printf("mode %d (%s) active %d blocks %4d: %.1f ns per slice, %.0f s_memtime ticks per slice (%.2f ticks/ns)\n", mode, mode ? "independent sums" : "serial chain", active, blocks,
               ms * 1e6 / iters, (double)h / iters, (double)h / (ms * 1e6));
    }
    return 0;
}
