// Lab for the fp32 MFMA GEMM core (gfx950): direct-to-LDS DMA tiles, STAGES-deep LDS ring, 64x64 per wavefront.
// Compares candidate kernels with the shipped ones (through libmerlin_hip.so) on the shapes of the hot path: bit-exact
// check + TF/s.  Build / run (GPU box):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/gemm_lab.hip -Imodels_amd/csrc -Lmodels_amd/csrc -lmerlin_hip \
//         -Wl,-rpath,$PWD/models_amd/csrc -o tools/exp/gemm_lab && tools/exp/gemm_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/merlin_hip.h"
#include "../../models_amd/csrc/mh_gemm2.h"

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

static float* dev_rand(size_t n, uint32_t seed, float scale = 1.f) {
    std::vector<float> h(n);
    uint32_t s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = ((int32_t)(s >> 8) % 2001 - 1000) * 0.001f * scale;
    }
    float* d;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

template <typename F>
static float time_ms(F&& f, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

static size_t mismatches(const float* d0, const float* d1, int64_t M, int N, int64_t ld) {
    std::vector<float> a((size_t)M * ld), b((size_t)M * ld);
    CK(hipMemcpy(a.data(), d0, a.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d1, b.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int64_t r = 0; r < M; ++r)
        for (int c = 0; c < N; ++c)
            if (memcmp(&a[r * ld + c], &b[r * ld + c], 4) != 0) ++bad;
    return bad;
}

// Staggered co-resident workgroups (UNMEASURED lead of round 2): with one column tile (N = 128) every CU runs ONE generation of
// workgroups in lock step -- their barrier gaps, LDS-read latencies, prologues and store phases coincide.  Odd workgroups sleep
// `units` x 1024 cycles (~0.43 us each at 2.4 GHz) before their first tile load, so that two workgroups sharing a CU (128-row
// tiles, 4 waves each: one wave per SIMD per workgroup) run out of phase and fill each other's gaps.
template <int BM, int BN, int WM, int WN, bool NT, int STAGES>
__global__ __launch_bounds__(WM* WN * 64) void staggered_kernel(const float* __restrict__ A, int64_t lda,
                                                              const float* __restrict__ B, int64_t ldb, int64_t M, int N, int K,
                                                              float* __restrict__ C, int64_t ldc, const mhgemm2::Epilogue ep,
                                                              int ncol_tiles, int units) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    if (blockIdx.x & 1)
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(16);
    const int64_t row0 = (int64_t)(blockIdx.x / ncol_tiles) * BM;
    const int n0 = (int)(blockIdx.x % ncol_tiles) * BN;
    f32x16 acc[TM][TN];
    mhgemm2::gemm2_tile<BM, BN, WM, WN, NT, STAGES, true, 16, 0>(A, lda, B, ldb, M, N, K, row0, n0, smem, acc);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    mhgemm::store_tile<TM, TN>(acc, C, ldc, row0 + (wave / WN) * TM * 32, n0 + (wave % WN) * TN * 32, M, N,
                               (int)(threadIdx.x & 63), ep);
}

template <int BM, int BN, int WM, int WN, bool NT, int STAGES>
static void run_staggered(const char* name, int units, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int N,
                          int K, float* C, int64_t ldc, const float* ref, int iters) {
    mhgemm2::Epilogue ep{};
    auto kern = staggered_kernel<BM, BN, WM, WN, NT, STAGES>;
    const size_t lds = (size_t)STAGES * (BM + BN) * 16 * sizeof(float);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int ncol = (N + BN - 1) / BN;
    const int64_t nrow = (M + BM - 1) / BM;
    auto launch = [&]() {
        hipLaunchKernelGGL(kern, dim3((unsigned)(nrow * ncol)), dim3(WM * WN * 64), lds, 0, A, lda, B, ldb, M, N, K, C, ldc, ep, ncol,
                           units);
    };
    CK(hipMemset(C, 0xff, (size_t)M * ldc * 4));
    launch();
    CK(hipDeviceSynchronize());
    const size_t bad = ref ? mismatches(C, ref, M, N, ldc) : 0;
    const float ms = time_ms(launch, iters);
    char label[96];
    snprintf(label, sizeof label, "%s stagger %d", name, units);
    printf("  %-34s %8.1f us  %6.1f TF  mismatches %zu\n", label, ms * 1e3, 2.0 * M * N * K / ms * 1e-9, bad);
    fflush(stdout);
}

template <int BM, int BN, int WM, int WN, bool NT, int STAGES, bool PIPE = true, int BKT = 16, int ABLATE = 0>
static void run_variant(const char* name, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int N, int K,
                        float* C, int64_t ldc, const float* ref, int iters) {
    mhgemm2::Epilogue ep{};
    auto launch = [&]() { mhgemm2::launch<BM, BN, WM, WN, NT, STAGES, PIPE, BKT, ABLATE>(A, lda, B, ldb, M, N, K, C, ldc, ep, 0); };
    CK(hipMemset(C, 0xff, (size_t)M * ldc * 4));
    launch();
    CK(hipDeviceSynchronize());
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        printf("  %-34s launch error %s\n", name, hipGetErrorString(e));
        return;
    }
    const size_t bad = ref ? mismatches(C, ref, M, N, ldc) : 0;
    const float ms = time_ms(launch, iters);
    printf("  %-34s %8.1f us  %6.1f TF  mismatches %zu\n", name, ms * 1e3, 2.0 * M * N * K / ms * 1e-9, bad);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int64_t M = getenv("LAB_M") ? atoll(getenv("LAB_M")) : 65536;
    struct Shape {
        const char* name;
        int K, N;
        bool nt;
        int iters;
    };
    const Shape shapes[] = {{"top  415x128 fwd (NN, K=416)", 416, 128, false, 30},
                            {"dX   128->416   (NT, K=128)", 128, 416, true, 30},
                            {"tower 512x256   (NN)", 512, 256, false, 30},
                            {"deep 3344x512   (NN)", 3344, 512, false, 5},
                            {"cross 3344x3344 (NN)", 3344, 3344, false, 2},
                            {"dXcross 3344x3344 (NT)", 3344, 3344, true, 2}};
    const char* only = argc > 1 ? argv[1] : nullptr;
    for (const Shape& sh : shapes) {
        if (only && !strstr(sh.name, only)) continue;
        const int K = sh.K, N = sh.N;
        printf("%s  M=%lld\n", sh.name, (long long)M);
        float* A = dev_rand((size_t)M * K, 1);
        float* B = dev_rand((size_t)K * N, 2, 0.05f);  // NN: W[K, N]; NT: W[N, K]
        float *C0, *C1;
        const int64_t ldc = (N + 3) / 4 * 4;
        CK(hipMalloc(&C0, (size_t)M * ldc * 4));
        CK(hipMalloc(&C1, (size_t)M * ldc * 4));
        CK(hipMemset(C0, 0, (size_t)M * ldc * 4));
        auto base = [&]() {
            if (!sh.nt)
                mh_linear_bias_act_fwd(A, K, B, nullptr, M, K, N, MH_ACT_NONE, C0, ldc, nullptr);
            else  // dx[M, N] = dz[M, K] W[N, K]^T through the backward entry point (dx only)
                mh_linear_bias_act_bwd(C1, N, B, nullptr, 0, A, K, M, N, K, MH_ACT_NONE, MH_ACT_NONE, C0, ldc, nullptr,
                                       nullptr, nullptr, 0, nullptr);
        };
        base();
        CK(hipDeviceSynchronize());
        if (mh_last_error() && mh_last_error()[0]) printf("  baseline: %s\n", mh_last_error());
        const float ms = time_ms(base, sh.iters);
        printf("  %-34s %8.1f us  %6.1f TF\n", "shipped kernel", ms * 1e3, 2.0 * M * N * K / ms * 1e-9);
        const int64_t lda = K, ldb = sh.nt ? K : N;
        if (sh.nt) {
            run_variant<256, 128, 4, 2, true, 3>("256x128 8w 3-stage", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<256, 128, 4, 2, true, 3, true, 32>("256x128 8w 3-stage BK32", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<256, 128, 4, 2, true, 2, true, 32>("256x128 8w 2-stage BK32", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<128, 128, 2, 2, true, 3, true, 32>("128x128 4w 3-stage BK32", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<128, 128, 4, 2, true, 3, true, 32>("128x128 8w 3-stage BK32", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<256, 128, 4, 2, true, 2>("256x128 8w 2-stage", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<128, 128, 2, 2, true, 3>("128x128 4w 3-stage", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<128, 128, 2, 2, true, 4>("128x128 4w 4-stage", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<128, 64, 2, 1, true, 4>("128x64  2w 4-stage", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            for (int u : {0, 1, 2, 4, 8}) run_staggered<128, 128, 2, 2, true, 3>("128x128 4w", u, A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            for (int u : {0, 2, 4, 8}) run_staggered<256, 128, 4, 2, true, 3>("256x128 8w", u, A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
        } else {
            run_variant<256, 128, 4, 2, false, 3>("256x128 8w 3-stage", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<256, 128, 4, 2, false, 3, true, 16, 1>("256x128 8w 3-stage NO LOADS", A, lda, B, ldb, M, N, K, C1, ldc, nullptr, sh.iters);
            run_variant<256, 128, 4, 2, false, 3, true, 16, 2>("256x128 8w 3-stage NO MFMA", A, lda, B, ldb, M, N, K, C1, ldc, nullptr, sh.iters);
            run_variant<128, 128, 2, 2, false, 3, true, 16, 1>("128x128 4w 3-stage NO LOADS", A, lda, B, ldb, M, N, K, C1, ldc, nullptr, sh.iters);
            run_variant<128, 128, 2, 2, false, 3, true, 16, 2>("128x128 4w 3-stage NO MFMA", A, lda, B, ldb, M, N, K, C1, ldc, nullptr, sh.iters);
            run_variant<256, 128, 4, 2, false, 3, true, 32>("256x128 8w 3-stage BK32", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<256, 128, 4, 2, false, 2, true, 32>("256x128 8w 2-stage BK32", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<128, 128, 2, 2, false, 3, true, 32>("128x128 4w 3-stage BK32", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<128, 128, 4, 2, false, 3, true, 32>("128x128 8w 3-stage BK32", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<256, 128, 4, 2, false, 2>("256x128 8w 2-stage", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<128, 128, 2, 2, false, 3>("128x128 4w 3-stage", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<128, 128, 2, 2, false, 4>("128x128 4w 4-stage", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            run_variant<128, 64, 2, 1, false, 4>("128x64  2w 4-stage", A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            for (int u : {0, 1, 2, 4, 8}) run_staggered<128, 128, 2, 2, false, 3>("128x128 4w", u, A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
            for (int u : {0, 2, 4, 8}) run_staggered<256, 128, 4, 2, false, 3>("256x128 8w", u, A, lda, B, ldb, M, N, K, C1, ldc, C0, sh.iters);
        }
        CK(hipFree(A));
        CK(hipFree(B));
        CK(hipFree(C0));
        CK(hipFree(C1));
    }
    return 0;
}
