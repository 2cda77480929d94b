// lds_gather_probe — measured ceiling of the PQ asymmetric-distance scan's inner operation on this GPU: random 8-byte (and 4-byte)
// LDS gathers indexed by code bytes, with the scan's instruction mix around them (byte extract, address add, two float adds per
// gather), from 1024-thread workgroups, one per CU, tables resident in LDS. What it prints is the rate bench.py prices the
// adc_scan kernel against ("lds_gather_ceiling"): a MEASUREMENT on the box the bench runs on, not a model of bank conflicts.
//   table: [32 subspace rows][256 entries] of float2 (64 KiB, one phase of the scan's ring) / float (32 KiB)
//   a lane holds 8 random code words (32 code bytes) and walks the 32 rows with them, `iters` times
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_gather_probe tools/lds_gather_probe.hip   (done by __graft_entry__.build())
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int WIDE>
__global__ __launch_bounds__(1024) void gather_kernel(const unsigned* __restrict__ codes, int iters, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x;
    constexpr int ESZ = WIDE ? 8 : 4;
    for (int i = t; i < 32 * 256 * ESZ / 4; i += 1024) reinterpret_cast<float*>(lds)[i] = (float)(i & 1023) * 1e-3f;
    unsigned w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = codes[((long)blockIdx.x * 1024 + t) * 8 + j];
    __syncthreads();
    float a0 = 0.0f, a1 = 0.0f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int m = j * 4 + b;
                const unsigned code = (w[j] >> (8 * b)) & 0xFFu;
                if constexpr (WIDE) {
                    const f32x2 v = *reinterpret_cast<const f32x2*>(lds + m * 256 * 8 + code * 8);
                    a0 += v[0]; a1 += v[1];
                } else {
                    a0 += *reinterpret_cast<const float*>(lds + m * 256 * 4 + code * 4);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = w[j] * 1664525u + 1013904223u;      // new codes for the next round (8 VALU per 32 gathers)
    }
    if (a0 + a1 == 12345.678f) out[0] = a0;       // keep the sums alive
}

template <int WIDE>
static int run(int cus, const unsigned* dcodes, float* dout, int iters, double* gathers_per_s, double* clocks_per_wave_gather, int clock_khz) {
    const size_t lds = 32 * 256 * (WIDE ? 8 : 4);
    CK(hipFuncSetAttribute((const void*)gather_kernel<WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    gather_kernel<WIDE><<<dim3(cus), dim3(1024), lds>>>(dcodes, 8, dout);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(a));
        gather_kernel<WIDE><<<dim3(cus), dim3(1024), lds>>>(dcodes, iters, dout);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double gathers = (double)cus * 1024.0 * 32.0 * iters;
    *gathers_per_s = gathers / (best * 1e-3);
    *clocks_per_wave_gather = (best * 1e-3) * clock_khz * 1e3 / (16.0 * 32.0 * iters / 4.0) / 4.0;   // per SIMD: 4 waves x 32 x iters wave-gathers
    return 0;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, iters = 2000;
    std::vector<unsigned> h((size_t)cus * 1024 * 8);
    unsigned long long x = 0x9E3779B97F4A7C15ull;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (unsigned)(x >> 16); }
    unsigned* d; float* o;
    CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o, 64));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    double g8 = 0, c8 = 0, g4 = 0, c4 = 0;
    if (run<1>(cus, d, o, iters, &g8, &c8, p.clockRate)) return 1;
    if (run<0>(cus, d, o, iters, &g4, &c4, p.clockRate)) return 1;
    printf("{\"what\": \"random code-byte-indexed LDS gathers with the ADC scan's instruction mix, 1024-thread workgroups, one per CU, %d CUs, table resident (tools/lds_gather_probe.hip)\", "
           "\"gathers_per_s_b64\": %.4g, \"lookups_per_s_two_queries_per_gather\": %.4g, \"gathers_per_s_b32\": %.4g, \"lookups_per_s_one_query_per_gather\": %.4g, "
           "\"simd_clocks_per_wave_gather_b64_at_nominal_clock\": %.2f, \"nominal_clock_mhz\": %d}\n",
           cus, g8, 2.0 * g8, g4, g4, c8, p.clockRate / 1000);
    return 0;
}
