// Lab: streaming-copy variants (which one becomes mh_stream_copy).  hipcc -O3 --offload-arch=gfx950 tools/exp/copy_lab.hip -o copy_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// A: grid-stride, U loads in flight, stride = whole grid
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_a(const f32x4* __restrict__ s, f32x4* __restrict__ d, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u]; }
    }
    for (; i < n4; i += stride) d[i] = s[i];
}
// B: a workgroup owns contiguous tiles of 256 * U float4 (U KiB * 4 per wave contiguous)
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_b(const f32x4* __restrict__ s, f32x4* __restrict__ d, int64_t n4) {
    const int64_t tile = 256 * U;
    for (int64_t t = blockIdx.x; t * tile < n4; t += gridDim.x) {
        const int64_t base = t * tile + threadIdx.x;
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t i = base + u * 256; if (i < n4) v[u] = NT ? __builtin_nontemporal_load(s + i) : s[i]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t i = base + u * 256; if (i < n4) { if (NT) __builtin_nontemporal_store(v[u], d + i); else d[i] = v[u]; } }
    }
}
template <typename K>
static void run(const char* name, K kern, int grid, const f32x4* s, f32x4* d, int64_t n4) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, s, d, n4);
    CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, s, d, n4);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    printf("%-28s grid %6d: %7.1f us  %7.1f GB/s\n", name, grid, ms * 1e3, 2.0 * n4 * 16 / ms * 1e-6);
}
int main() {
    const int64_t bytes = 1ll << 30, n4 = bytes / 16;
    f32x4 *s, *d; CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMemset(s, 1, bytes));
    for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
        run("A U=4", copy_a<4, false>, grid, s, d, n4);
        run("A U=4 nt", copy_a<4, true>, grid, s, d, n4);
        run("A U=8 nt", copy_a<8, true>, grid, s, d, n4);
        run("B U=4", copy_b<4, false>, grid, s, d, n4);
        run("B U=4 nt", copy_b<4, true>, grid, s, d, n4);
        run("B U=8 nt", copy_b<8, true>, grid, s, d, n4);
        run("B U=16 nt", copy_b<16, true>, grid, s, d, n4);
    }
    run("A U=1 one pass", copy_a<1, false>, (int)(n4 / 256), s, d, n4);
    run("A U=1 nt one pass", copy_a<1, true>, (int)(n4 / 256), s, d, n4);
    run("B U=4 nt one pass", copy_b<4, true>, (int)(n4 / 1024), s, d, n4);
    CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("hipMemcpyAsync D2D: %7.1f GB/s\n", 2.0 * bytes / (ms / 10) * 1e-6);
    return 0;
}
