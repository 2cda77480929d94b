// mfma_i8_probe.hip — issue rate of the two int8 MFMA shapes of gfx950, one wave per SIMD (4 waves per workgroup, one workgroup per CU), four independent accumulators.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_i8_probe.hip -o tools/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE> __global__ __launch_bounds__(256, 1) void k(int* out, int iters, int seed) {
    i32x4 a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 3, seed * 5, seed * 7, seed * 11};
    if (SHAPE == 0) {
        i32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        i32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {};
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c7, 0, 0, 0);
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
    }
}
int main() {
    int* out; hipMalloc(&out, 256 * 256 * 4);
    const int iters = 20000;
    for (int shape = 0; shape < 2; shape++) for (int seed : {0, 12345}) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        if (shape == 0) k<0><<<256, 256>>>(out, 100, seed); else k<1><<<256, 256>>>(out, 100, seed);
        hipEventRecord(a);
        if (shape == 0) k<0><<<256, 256>>>(out, iters, seed); else k<1><<<256, 256>>>(out, iters, seed);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double n = (double)iters * (shape == 0 ? 4 : 8) * 1024, ops = n * (shape == 0 ? 65536.0 : 32768.0);
        printf("%s, operands %s: %.1f TOPS, %.1f ns per MFMA per SIMD (= %.1f clocks at 2.4 GHz)\n", shape == 0 ? "v_mfma_i32_32x32x32_i8" : "v_mfma_i32_16x16x64_i8", seed ? "non-zero" : "zero",
               ops / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)iters * (shape == 0 ? 4 : 8)), ms * 1e6 / ((double)iters * (shape == 0 ? 4 : 8)) * 2.4);
    }
    return 0;
}
