// Experiment: how fast can gfx950 do independent random 256-byte read-modify-writes (Adagrad row update)?
// Bounds the segment-reduce kernel of the embedding backward: U unique rows, each reads w + acc rows and
// writes both (4 x 256 B), plus G gradient rows read once.  hipcc --offload-arch=gfx950 -O3 rmw_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void rmw_kernel(float* __restrict__ W, float* __restrict__ S,
                                                  const int* __restrict__ rows, const float* __restrict__ grad,
                                                  int U, int gper, int mode) {
    const int gi = threadIdx.x >> 4, c4 = threadIdx.x & 15;
    const int u = blockIdx.x * 16 + gi;
    if (u >= U) return;
    const int64_t r = rows[u];
    f32x4 g = {0, 0, 0, 0};
    for (int j = 0; j < gper; ++j) g += *reinterpret_cast<const f32x4*>(grad + ((int64_t)u * gper + j) * 64 + c4 * 4);
    if (mode == 0) {  // adagrad
        f32x4 w = *reinterpret_cast<f32x4*>(W + r * 64 + c4 * 4);
        f32x4 s = *reinterpret_cast<f32x4*>(S + r * 64 + c4 * 4);
        s += g * g;
        w.x -= 0.01f * g.x / (sqrtf(s.x) + 1e-7f);
        w.y -= 0.01f * g.y / (sqrtf(s.y) + 1e-7f);
        w.z -= 0.01f * g.z / (sqrtf(s.z) + 1e-7f);
        w.w -= 0.01f * g.w / (sqrtf(s.w) + 1e-7f);
        *reinterpret_cast<f32x4*>(S + r * 64 + c4 * 4) = s;
        *reinterpret_cast<f32x4*>(W + r * 64 + c4 * 4) = w;
    } else if (mode == 1) {  // sgd
        f32x4 w = *reinterpret_cast<f32x4*>(W + r * 64 + c4 * 4);
        w -= g * 0.01f;
        *reinterpret_cast<f32x4*>(W + r * 64 + c4 * 4) = w;
    } else {  // write only
        *reinterpret_cast<f32x4*>(W + r * 64 + c4 * 4) = g;
    }
}

int main() {
    const int64_t V = 16 << 20;  // 16M rows x 256 B = 4 GB per array
    const int U = 590000;
    float *W, *S, *grad;
    int* rows;
    hipMalloc(&W, V * 256);
    hipMalloc(&S, V * 256);
    hipMemset(W, 0, V * 256);
    hipMemset(S, 0, V * 256);
    const int GP = 3;
    hipMalloc(&grad, (int64_t)U * GP * 256);
    hipMemset(grad, 0, (int64_t)U * GP * 256);
    std::vector<int> h(U);
    std::mt19937_64 rng(1);
    for (int sorted = 0; sorted < 2; ++sorted) {
        for (auto& x : h) x = (int)(rng() % V);
        if (sorted) std::sort(h.begin(), h.end());
        hipMalloc(&rows, U * 4);
        hipMemcpy(rows, h.data(), U * 4, hipMemcpyHostToDevice);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        for (int mode = 0; mode < 3; ++mode)
            for (int gper : {0, 1, 3}) {
                for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(rmw_kernel, dim3((U + 15) / 16), dim3(256), 0, 0, W, S, rows, grad, U, gper, mode);
                hipEventRecord(a);
                const int R = 10;
                for (int it = 0; it < R; ++it) hipLaunchKernelGGL(rmw_kernel, dim3((U + 15) / 16), dim3(256), 0, 0, W, S, rows, grad, U, gper, mode);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                const double us = ms * 1e3 / R;
                const double bytes = (double)U * 256 * ((mode == 0 ? 4 : mode == 1 ? 2 : 1) + gper);
                printf("sorted=%d mode=%d gper=%d  %8.1f us  %7.1f GB/s\n", sorted, mode, gper, us, bytes / us / 1e3);
            }
    }
    return 0;
}
