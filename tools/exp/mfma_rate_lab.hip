// Lab: issue rate of v_mfma_f32_32x32x16_bf16 as a function of (a) how many INDEPENDENT accumulator chains a wavefront
// interleaves, (b) how many wavefronts share a SIMD, (c) operand data (zeros vs random bits: power), on the whole chip.
// Answers why a kernel whose MFMAs alternate between two accumulators runs at ~52 cycles per MFMA instead of 32.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/mfma_rate_lab.hip -o gpurun_in/lab/mfma_rate_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// every wavefront: iters x 48 MFMAs, round-robin over CH accumulators (dependency distance CH)
template <int CH>
__global__ __launch_bounds__(256) void rate_kernel(const uint4* __restrict__ in, float* __restrict__ out, int iters, uint64_t* __restrict__ cycles) {
    const bf16x8 a = __builtin_bit_cast(bf16x8, in[threadIdx.x]);
    const bf16x8 b = __builtin_bit_cast(bf16x8, in[256 + threadIdx.x]);
    f32x16 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 48; ++j) acc[j % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j % CH], 0, 0, 0);
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float r = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) r += acc[c][0] + acc[c][7];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int CH>
void run(const uint4* in, float* out, uint64_t* cyc, int wg, const char* what) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<CH>, dim3(wg), dim3(256), 0, 0, in, out, 10, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<CH>, dim3(wg), dim3(256), 0, 0, in, out, iters, cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    uint64_t c;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double n = (double)iters * 48;
    const double tf = (double)wg * 4 * n * 32768.0 / (ms * 1e-3) * 1e-12;
    printf("  %-12s chains %d, %4d workgroups (%d wavefront(s) per SIMD): %6.1f s_memtime ticks per MFMA per wavefront, %7.1f TF (%.3f of 2500), %.2f ms\n", what, CH, wg,
           wg / 256, (double)c / n, tf, tf / 2500.0, ms);
}

int main() {
    uint4* in;
    float* out;
    uint64_t* cyc;
    CK(hipMalloc(&in, 512 * 16));
    CK(hipMalloc(&out, 1024 * 256 * 4));
    CK(hipMalloc(&cyc, 8));
    for (int pass = 0; pass < 2; ++pass) {
        uint32_t h[2048];
        uint32_t s = 99u;
        for (auto& v : h) {
            s = s * 1664525u + 1013904223u;
            // random bf16 pairs in (-2, 2): sign + exponent 0x3f.. + random mantissa
            v = pass == 0 ? 0u : ((s & 0x807f807fu) | 0x3f803f80u);
        }
        CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
        const char* what = pass == 0 ? "zero data" : "random data";
        for (int wg : {256, 512}) {
            run<1>(in, out, cyc, wg, what);
            run<2>(in, out, cyc, wg, what);
            run<3>(in, out, cyc, wg, what);
            run<4>(in, out, cyc, wg, what);
            run<6>(in, out, cyc, wg, what);
        }
    }
    return 0;
}
