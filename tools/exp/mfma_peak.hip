// Development experiment: what limits the fp32 MFMA loop?  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDK 36

// variant 0: registers only; 1: operands re-read from LDS each k-tile (no barriers); 2: + barrier per k-tile
// 3: + ds_write of a dummy tile each k-tile (LDS store traffic like the staging)
template <int V, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, int iters, const float* __restrict__ gA, const float* __restrict__ gB, int ld) {
    __shared__ __attribute__((aligned(16))) float As[2 * 128 * LDK + 2 * 128 * LDK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < 4 * 128 * LDK; i += WAVES * 64) As[i] = (float)((i * 37) % 1001) * 0.002f - 1.0f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int wm = (wave / 2) % 2, wn = wave % 2;
    const float* Ab = As + (wm * 64 + l31) * LDK + h * 16;
    const float* Bb = As + 2 * 128 * LDK + (wn * 64 + l31) * LDK + h * 16;
    f32x4 a0 = {1, 2, 3, 4}, a1 = {2, 3, 4, 5}, b0 = {1, 1, 2, 2}, b1 = {3, 3, 1, 1};
    f32x4 st = {1.f, 2.f, 3.f, 4.f};
    constexpr int NVL = (WAVES == 4 ? 4 : 2);
    f32x4 ra[NVL], rb[NVL];
    const long rowA = (long)blockIdx.x * 128, rowB = (long)(blockIdx.x % 64) * 128;
    for (int it = 0; it < iters; ++it) {
        const int buf = (it & 1) * 128 * LDK;
        if (V >= 4) {
            const int k0 = (it * 32) % ld;
#pragma unroll
            for (int i = 0; i < NVL; ++i) {
                const int idx = threadIdx.x + i * WAVES * 64;
                const int r = idx >> 3, c4 = idx & 7;
                ra[i] = *(const f32x4*)(gA + (rowA + r) * ld + k0 + c4 * 4);
                rb[i] = *(const f32x4*)(gB + (rowB + r) * ld + k0 + c4 * 4);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (V >= 1) {
                a0 = *(const f32x4*)(Ab + buf + g * 4);
                a1 = *(const f32x4*)(Ab + buf + 32 * LDK + g * 4);
                b0 = *(const f32x4*)(Bb + buf + g * 4);
                b1 = *(const f32x4*)(Bb + buf + 32 * LDK + g * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
            }
        }
        if (V >= 4) {
            float* d = As + (1 - (it & 1)) * 128 * LDK;
#pragma unroll
            for (int i = 0; i < NVL; ++i) {
                const int idx = threadIdx.x + i * WAVES * 64;
                const int r = idx >> 3, c4 = idx & 7;
                *(float2*)(d + r * LDK + c4 * 2) = make_float2(ra[i].x, ra[i].z);
                *(float2*)(d + r * LDK + c4 * 2 + 16) = make_float2(ra[i].y, ra[i].w);
                *(float2*)(d + 2 * 128 * LDK + r * LDK + c4 * 2) = make_float2(rb[i].x, rb[i].z);
                *(float2*)(d + 2 * 128 * LDK + r * LDK + c4 * 2 + 16) = make_float2(rb[i].y, rb[i].w);
            }
        } else if (V >= 3) {
            // 8 float4 per thread of staging stores, as 16 b64 writes (4-wave block) into the other buffer
            float* d = As + (1 - (it & 1)) * 128 * LDK;
#pragma unroll
            for (int i = 0; i < (WAVES == 4 ? 8 : 4); ++i) {
                const int idx = threadIdx.x + i * WAVES * 64;
                const int r = (idx >> 3) & 127, c4 = idx & 7;
                *(float2*)(d + r * LDK + c4 * 2 + (i >= (WAVES == 4 ? 4 : 2) ? 2 * 128 * LDK : 0)) = make_float2(st.x, st.z);
                *(float2*)(d + r * LDK + c4 * 2 + 16 + (i >= (WAVES == 4 ? 4 : 2) ? 2 * 128 * LDK : 0)) = make_float2(st.y, st.w);
            }
        }
        if (V >= 2) __syncthreads();
    }
    float s = 0;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void fill(float* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { unsigned x = (unsigned)(i * 2654435761u) ^ 0x9e3779b9u; x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15; p[i] = ((x & 0xffffff) / 8388608.0f) - 1.0f; } }
static float *gA, *gB;
static const int LD = 4096;
template <int V, int WAVES>
void run(const char* name, int blocks) {
    float* out;
    hipMalloc(&out, blocks * WAVES * 64 * 4);
    const int iters = 4000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<V, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, 10, gA, gB, LD);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<V, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, iters, gA, gB, LD);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double flops = (double)blocks * WAVES * iters * 64 * 4096.0;
    printf("%-44s blocks=%4d  %8.3f ms  %7.1f TF/s\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    hipMalloc(&gA, (size_t)1024 * 128 * LD * 4); hipMalloc(&gB, (size_t)64 * 128 * LD * 4);
    { size_t n = (size_t)1024 * 128 * LD; fill<<<(n + 255) / 256, 256>>>(gA, n); n = (size_t)64 * 128 * LD; fill<<<(n + 255) / 256, 256>>>(gB, n); hipDeviceSynchronize(); }
    run<0, 4>("regs only, 4 waves/block", 512);
    run<0, 4>("regs only, 4 waves/block", 256);
    run<1, 4>("LDS reads, 4 waves/block", 512);
    run<2, 4>("LDS reads + barrier, 4 waves/block", 512);
    run<3, 4>("LDS reads + writes + barrier, 4 w/b", 512);
    run<3, 8>("LDS reads + writes + barrier, 8 w/b", 512);
    run<3, 4>("LDS reads + writes + barrier, 4 w/b", 1024);
    run<4, 4>("global loads + LDS staging, 4 w/b", 512);
    run<4, 8>("global loads + LDS staging, 8 w/b", 512);
    run<4, 4>("global loads + LDS staging, 4 w/b", 1024);
    return 0;
}
