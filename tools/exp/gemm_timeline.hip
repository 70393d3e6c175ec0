// Lab: where does a 256 x 128 workgroup of the second-generation GEMM spend its time on the one-column-tile layer (415 -> 128, M = 64 K)?
// The main loop of mhgemm2::gemm2_tile restated with s_memtime stamps after the prologue, after every k-tile and after the stores
// (lane 0 of wave 0 of a few workgroups).   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/gemm_timeline.hip -Imodels_amd/csrc -o gemm_timeline
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../models_amd/csrc/mh_gemm2.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace mhgemm2;

template <int BM, int BN, int WM, int WN, int STAGES, int BKT>
__global__ __launch_bounds__(WM* WN * 64) void timed_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                            int64_t M, int N, int K, float* __restrict__ C, int64_t ldc, Epilogue ep,
                                                            unsigned long long* __restrict__ stamps) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    unsigned long long w0 = wall_clock64();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    f32x16 acc[TM][TN];
    gemm2_tile<BM, BN, WM, WN, false, STAGES, true, BKT, 0>(A, lda, B, ldb, M, N, K, row0, 0, smem, acc);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    mhgemm::store_tile<TM, TN>(acc, C, ldc, row0 + (wave / WN) * TM * 32, (wave % WN) * TN * 32, M, N, (int)(threadIdx.x & 63), ep);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 4 + 0] = t0;
        stamps[blockIdx.x * 4 + 1] = t1;
        stamps[blockIdx.x * 4 + 2] = t2;
        stamps[1024 + blockIdx.x * 2 + 0] = w0;
        stamps[1024 + blockIdx.x * 2 + 1] = wall_clock64();
    }
}

__global__ void spin_kernel(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t = t0;
    while (t - t0 < ticks) t = __builtin_amdgcn_s_memtime();
    out[0] = t - t0;
}

int main() {
    {   // calibrate s_memtime: spin for 10 M ticks, time with events
        unsigned long long* o; CK(hipMalloc(&o, 8));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, 1000000ull, o); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a)); hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, 10000000ull, o); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("s_memtime: 10 M ticks = %.3f ms -> %.1f MHz\n", ms, 10000.0 / ms / 1.0);
    }
    const int64_t M = 65536; const int K = 416, N = 128;
    std::vector<float> h((size_t)M * K, 0.5f);
    float *A, *B, *C; unsigned long long* st;
    CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)K * N * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data(), (size_t)K * N * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&st, (1024 + 512) * 8));
    auto run = [&](auto kern, size_t lds, int threads, const char* name) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        Epilogue ep{}; ep.act = MH_ACT_RELU;
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), lds, 0, A, (int64_t)K, B, (int64_t)N, M, N, K, C, (int64_t)N, ep, st);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> s(1024 + 512);
        CK(hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long first = ~0ull, last = 0;
        double loop = 0, store = 0, start_spread = 0;
        for (int b = 0; b < 256; ++b) { if (s[b * 4] < first) first = s[b * 4]; if (s[b * 4 + 2] > last) last = s[b * 4 + 2]; }
        for (int b = 0; b < 256; ++b) { loop += (double)(s[b * 4 + 1] - s[b * 4]); store += (double)(s[b * 4 + 2] - s[b * 4 + 1]); start_spread += (double)(s[b * 4] - first); }
        unsigned long long w_first = ~0ull, w_last_start = 0, w_first_end = ~0ull, w_last_end = 0;
        for (int b = 0; b < 256; ++b) {
            const unsigned long long a_ = s[1024 + b * 2], e_ = s[1024 + b * 2 + 1];
            if (a_ < w_first) w_first = a_;
            if (a_ > w_last_start) w_last_start = a_;
            if (e_ < w_first_end) w_first_end = e_;
            if (e_ > w_last_end) w_last_end = e_;
        }
        printf("   wall clock (100 MHz): last workgroup starts %.2f us after the first; first ends at %.2f us, last at %.2f us\n",
               (w_last_start - w_first) / 100.0, (w_first_end - w_first) / 100.0, (w_last_end - w_first) / 100.0);
        printf("%-26s mean per workgroup: loop (incl. prologue) %8.0f ticks, stores %6.0f ticks\n", name, loop / 256, store / 256);
    };
    run(timed_kernel<256, 128, 4, 2, 3, 16>, (size_t)3 * (256 + 128) * 16 * 4, 512, "256x128 8w 3-stage BK16");
    run(timed_kernel<256, 128, 4, 2, 3, 32>, (size_t)3 * (256 + 128) * 32 * 4, 512, "256x128 8w 3-stage BK32");
    run(timed_kernel<256, 128, 4, 2, 2, 32>, (size_t)2 * (256 + 128) * 32 * 4, 512, "256x128 8w 2-stage BK32");
    return 0;
}
