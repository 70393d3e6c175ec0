// Lab: the one-column-tile Dense layers (415 -> 128 at M = 64 K, the two-tower layers) as a WEIGHT-STATIONARY kernel.
//   * a workgroup keeps a 64-column slice of W[K, N] in LDS for its whole life (416 x 64 floats = 104 KB), laid out so that the B
//     fragment of an MFMA step is ONE linear ds_read_b32 per 32-column block;
//   * every WAVEFRONT streams its own 32-row blocks of A through a private LDS ring by DMA (global_load_lds) and never shares
//     them: the main loop has NO workgroup barrier; the ring stays primed across row blocks, the stores of one block drain while
//     the next block's MFMAs run (they are issued behind the prefetches, so the in-order vmcnt wait for a tile never waits for them).
// What the tiled kernels lose on these shapes -- prologue, one barrier per k-tile, a lock-step store phase (r2 / r3 notes) -- has no
// counterpart here.  Numerics: one k-ascending fmaf chain per output, bit-identical to the shipped kernels.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/skinny_lab.hip -Imodels_amd/csrc -Lmodels_amd/csrc -lmerlin_hip \
//         -Wl,-rpath,$PWD/models_amd/csrc -o tools/exp/skinny_lab && tools/exp/skinny_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/merlin_hip.h"
#include "../../models_amd/csrc/mh_gemm2.h"

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

using mhgemm2::dma16;
using mhgemm2::g_zero_chunk;
using mhgemm2::kmajor_src_chunk;
using mhgemm2::kmajor_swz;
using mhgemm2::mfma32;

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int NSTORE = 32;  // epilogue stores of a 32 x 64 block per lane (16 per accumulator)

// C[M, N] = act(A[M, K] W[K, N] + bias).  grid = nslices * nwg; slice = blockIdx.x % nslices owns columns [64 slice, 64 slice + 64).
template <int WAVES, int STAGES, int ABLATE = 0>
__global__ __launch_bounds__(WAVES * 64) void skinny_fwd_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                                int64_t ldw, const float* __restrict__ bias, int64_t M, int N, int K,
                                                                float* __restrict__ C, int64_t ldc, int act, int nslices) {
    constexpr int BK = 16, CH = 4, NI = 2;  // a wave's A tile: 32 rows x 16 k = 2 KB = two 1 KB DMA instructions
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int l31 = lane & 31, h = lane >> 5;
    const int nk = (K + BK - 1) / BK;
    float* Bs = smem;                                   // [nk * 8 k-pairs][128]: (pair, cb, h, c32) -> pair * 128 + cb * 64 + h * 32 + c32
    float* As = smem + nk * 1024 + wave * (STAGES * 512);  // this wave's ring: STAGES x [32 rows][16 k], chunk-swizzled
    const int slice = blockIdx.x % nslices, n0 = slice * 64;
    const int wg = blockIdx.x / nslices, nwg = gridDim.x / nslices;

    // ---- the W slice, once: instruction ci covers Bs[ci * 256 .. + 256) = two k-pairs ------------------------------------------
    for (int ci = wave; ci < nk * 4; ci += WAVES) {
        const int d = ci * 256 + lane * 4;
        const int pr = d >> 7, rem = d & 127, cb = rem >> 6, hh = (rem >> 5) & 1, c32 = rem & 31;
        const int k = 2 * pr + hh, col = n0 + cb * 32 + c32;
        const float* src = (k < K && col < N) ? W + (int64_t)k * ldw + col : g_zero_chunk;
        dma16(src, Bs + ci * 256);
    }
    wait_vm<0>();
    __syncthreads();

    // ---- this wave's row blocks -------------------------------------------------------------------------------------------------
    const int64_t nrb = (M + 31) / 32;
    const int64_t gw = (int64_t)wg * WAVES + wave, tw = (int64_t)nwg * WAVES;
    const int64_t nb = gw < nrb ? (nrb - gw + tw - 1) / tw : 0;
    const int64_t T = nb * nk;  // tiles of this wave, block-major
    // DMA source of instruction i of a tile: chunk position p = i * 64 + lane -> row p / 4, source chunk c
    int rowi[NI], kci[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = i * 64 + lane;
        rowi[i] = p / CH;
        kci[i] = 4 * kmajor_src_chunk<CH>(p);
    }
    int64_t ib = 0;  // block / k-tile of the next tile to ISSUE
    int ikt = 0;
    auto issue = [&](int slot) {
        const int64_t r0 = (gw + ib * tw) * 32;
        const int k0 = ikt * BK;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int64_t row = r0 + rowi[i];
            if (row > M - 1) row = M - 1;
            const float* g = A + row * lda + k0 + kci[i];
            if (k0 + kci[i] >= K) g = g_zero_chunk;
            dma16(g, As + slot * 512 + i * 256);
        }
        if (++ikt == nk) {
            ikt = 0;
            ++ib;
        }
    };
    const int fa = l31 * BK + (kmajor_swz<CH>(l31) << 2);  // this lane's row in a tile; chunk c: fa ^ (c << 2)
    const int ktail = K & 3;                               // K % 4 != 0: the last chunk holds pad elements that must not enter sums

    for (int s = 0; s < STAGES - 1; ++s)
        if (s < T) issue(s);
    f32x16 acc0, acc1;
    int64_t t = 0;
    for (int64_t j = 0; j < nb; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
        for (int kt = 0; kt < nk; ++kt, ++t) {
            // tile t has landed when at most the operations issued after it are outstanding: the DMAs of STAGES - 2 later tiles
            // and, for the first STAGES - 1 tiles of a block, the stores of the previous block (issued behind those prefetches)
            if (t + STAGES - 2 < T) {
                if (j > 0 && kt < STAGES - 1) wait_vm<(STAGES - 2) * NI + NSTORE>(); else wait_vm<(STAGES - 2) * NI>();
            } else {
                wait_vm<0>();
            }
            if (ABLATE != 1 && t + STAGES - 1 < T) issue((int)((t + STAGES - 1) % STAGES));
            const float* st = As + (int)(t % STAGES) * 512;
            const float* bp = Bs + kt * 1024 + lane;
            const bool last = ktail && kt == nk - 1;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                f32x4 v = *reinterpret_cast<const f32x4*>(st + (fa ^ (c << 2)));
                if (last) {  // uniform: zero the elements at or past K of the chunk that straddles it
                    const int kb = kt * BK + 4 * c;
                    if (kb + 1 >= K) v.y = 0.f;
                    if (kb + 2 >= K) v.z = 0.f;
                    if (kb + 3 >= K) v.w = 0.f;
                    if (kb >= K) v.x = 0.f;
                }
                const float a0 = h ? v.y : v.x, a1 = h ? v.w : v.z;  // k = 4c + h, 4c + 2 + h
                const float b00 = bp[(2 * c) * 128], b01 = bp[(2 * c) * 128 + 64];
                const float b10 = bp[(2 * c + 1) * 128], b11 = bp[(2 * c + 1) * 128 + 64];
                if (ABLATE != 2) {
                    acc0 = mfma32(a0, b00, acc0);
                    acc1 = mfma32(a0, b01, acc1);
                    acc0 = mfma32(a1, b10, acc0);
                    acc1 = mfma32(a1, b11, acc1);
                } else {
                    acc0[c] += a0 + b00 + b01 + a1 + b10 + b11;
                }
            }
        }
        // ---- epilogue of block j: bias + activation, 32 x 64 outputs ----
        const int64_t r0 = (gw + j * tw) * 32;
        const int c0 = n0 + l31, c1 = n0 + 32 + l31;
        const float bv0 = bias ? bias[c0] : 0.f, bv1 = bias ? bias[c1] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v0 = acc0[r] + bv0, v1 = acc1[r] + bv1;
            if (act == MH_ACT_RELU) {
                v0 = v0 > 0.f ? v0 : 0.f;
                v1 = v1 > 0.f ? v1 : 0.f;
            }
            // N % 64 == 0 (checked by the launcher): both column blocks are stored.  Rows at or past M exist only in the LAST block of
            // the matrix, i.e. the last block of its wave: no later wait depends on the number of stores issued here
            if (row < M) {
                C[row * ldc + c0] = v0;
                C[row * ldc + c1] = v1;
            }
        }
    }
}

static float* dev_rand(size_t n, uint32_t seed, float scale = 1.f) {
    std::vector<float> hbuf(n);
    uint32_t s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        hbuf[i] = ((int32_t)(s >> 8) % 2001 - 1000) * 0.001f * scale;
    }
    float* d;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemcpy(d, hbuf.data(), n * sizeof(float), hipMemcpyHostToDevice));
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

template <int WAVES, int STAGES, int ABLATE = 0>
static void run(const char* name, const float* A, int64_t lda, const float* W, const float* bias, int64_t M, int N, int K, float* C,
                const float* ref, int wg_per_slice) {
    const int nslices = (N + 63) / 64;
    const int nk = (K + 15) / 16;
    const size_t lds = ((size_t)nk * 1024 + (size_t)WAVES * STAGES * 512) * 4;
    auto kern = skinny_fwd_kernel<WAVES, STAGES, ABLATE>;
    if (lds > 160 * 1024) {
        printf("  %-40s needs %zu KB of LDS: skipped\n", name, lds >> 10);
        return;
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto launch = [&]() {
        hipLaunchKernelGGL(kern, dim3(nslices * wg_per_slice), dim3(WAVES * 64), lds, 0, A, lda, W, (int64_t)N, bias, M, N, K, C, (int64_t)N,
                           MH_ACT_RELU, nslices);
    };
    CK(hipMemset(C, 0xff, (size_t)M * N * 4));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    std::vector<float> a((size_t)M * N), b((size_t)M * N);
    CK(hipMemcpy(a.data(), C, a.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), ref, b.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < a.size(); ++i) bad += memcmp(&a[i], &b[i], 4) != 0;
    const float ms = time_ms(launch, 30);
    printf("  %-40s %8.1f us  %6.1f TF  (%.3f of 157.3)  mismatches %zu\n", name, ms * 1e3, 2.0 * M * N * K / ms * 1e-9,
           2.0 * M * N * K / ms * 1e-9 / 157.3, bad);
    fflush(stdout);
}

int main() {
    struct Shape {
        const char* name;
        int64_t M;
        int K, Kld, N;
    };
    const Shape shapes[] = {{"top 415 -> 128, M = 65536 (ld 416)", 65536, 415, 416, 128},
                            {"tower 256 -> 128, M = 32768", 32768, 256, 256, 128},
                            {"tower 256 -> 256, M = 32768", 32768, 256, 256, 256},
                            {"tower 512 -> 256, M = 32768", 32768, 512, 512, 256}};
    int ncu = 256;
    for (const Shape& sh : shapes) {
        printf("%s\n", sh.name);
        float* A = dev_rand((size_t)sh.M * sh.Kld, 1);
        if (sh.Kld != sh.K) {  // the pad column holds garbage in the product's buffers: make sure it does not matter
            std::vector<float> hbuf((size_t)sh.M * sh.Kld);
            CK(hipMemcpy(hbuf.data(), A, hbuf.size() * 4, hipMemcpyDeviceToHost));
            for (int64_t r = 0; r < sh.M; ++r) hbuf[r * sh.Kld + sh.K] = 1e30f;
            CK(hipMemcpy(A, hbuf.data(), hbuf.size() * 4, hipMemcpyHostToDevice));
        }
        float* W = dev_rand((size_t)sh.K * sh.N, 2, 0.05f);
        float* bias = dev_rand(sh.N, 3, 0.1f);
        float *C0, *C1;
        CK(hipMalloc(&C0, (size_t)sh.M * sh.N * 4));
        CK(hipMalloc(&C1, (size_t)sh.M * sh.N * 4));
        auto base = [&]() { mh_linear_bias_act_fwd(A, sh.Kld, W, bias, sh.M, sh.K, sh.N, MH_ACT_RELU, C0, sh.N, nullptr); };
        base();
        CK(hipDeviceSynchronize());
        const float ms = time_ms(base, 30);
        printf("  %-40s %8.1f us  %6.1f TF  (%.3f of 157.3)\n", "shipped kernel", ms * 1e3, 2.0 * sh.M * sh.N * sh.K / ms * 1e-9,
               2.0 * sh.M * sh.N * sh.K / ms * 1e-9 / 157.3);
        const int ns = (sh.N + 63) / 64;
        for (int per : {ncu / ns, 2 * ncu / ns}) {
            char nm[96];
            snprintf(nm, sizeof nm, "W-stationary 8 waves 3 stages, %d wg", per * ns);
            run<8, 3>(nm, A, sh.Kld, W, bias, sh.M, sh.N, sh.K, C1, C0, per);
            snprintf(nm, sizeof nm, "  8 waves 3 stages NO LOADS, %d wg", per * ns);
            run<8, 3, 1>(nm, A, sh.Kld, W, bias, sh.M, sh.N, sh.K, C1, C0, per);
            snprintf(nm, sizeof nm, "  8 waves 3 stages NO MFMA, %d wg", per * ns);
            run<8, 3, 2>(nm, A, sh.Kld, W, bias, sh.M, sh.N, sh.K, C1, C0, per);
            snprintf(nm, sizeof nm, "W-stationary 8 waves 4 stages, %d wg", per * ns);
            run<8, 4>(nm, A, sh.Kld, W, bias, sh.M, sh.N, sh.K, C1, C0, per);
            snprintf(nm, sizeof nm, "W-stationary 4 waves 6 stages, %d wg", per * ns);
            run<4, 6>(nm, A, sh.Kld, W, bias, sh.M, sh.N, sh.K, C1, C0, per);
            snprintf(nm, sizeof nm, "W-stationary 4 waves 4 stages, %d wg", per * ns);
            run<4, 4>(nm, A, sh.Kld, W, bias, sh.M, sh.N, sh.K, C1, C0, per);
        }
        CK(hipFree(A));
        CK(hipFree(W));
        CK(hipFree(bias));
        CK(hipFree(C0));
        CK(hipFree(C1));
    }
    return 0;
}
