// Lab: what do global atomics cost on gfx950 when they are spread over many addresses?  (Decides whether the sparse update can
// dedup its (table, id) pairs with per-row linked lists -- one atomicExch per lookup -- instead of a two-pass radix sort.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/atomic_lab.hip -o tools/exp/atomic_lab && tools/exp/atomic_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x)                                                                                    \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);        \
            exit(1);                                                                             \
        }                                                                                        \
    } while (0)

template <typename F>
static float time_us(F&& f, int iters = 10) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters * 1e3f;
}

// one atomicExch per entry: head[row[e]] <- e, next[e] <- previous head  (16 entries per thread in flight)
__global__ __launch_bounds__(256) void link_kernel(const uint32_t* __restrict__ row, int* __restrict__ head, int* __restrict__ next, int n) {
    const int base = blockIdx.x * 4096 + threadIdx.x;
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = base + i * 256;
        r[i] = row[e < n ? e : n - 1];
    }
    int prev[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = base + i * 256;
        prev[i] = (e < n) ? atomicExch(&head[r[i]], e) : -1;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = base + i * 256;
        if (e < n) next[e] = prev[i];
    }
}

// plain scattered store of the same shape (what the radix scatter does): the non-atomic floor
__global__ __launch_bounds__(256) void scatter_kernel(const uint32_t* __restrict__ row, int* __restrict__ head, int n) {
    const int base = blockIdx.x * 4096 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = base + i * 256;
        if (e < n) head[row[e]] = e;
    }
}

// is-head test + reset: what the apply kernel adds per entry
__global__ __launch_bounds__(256) void probe_kernel(const uint32_t* __restrict__ row, int* __restrict__ head, int* __restrict__ out, int n) {
    const int base = blockIdx.x * 4096 + threadIdx.x;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = base + i * 256;
        if (e < n && head[row[e]] == e) {
            ++c;
            head[row[e]] = -1;
        }
    }
    if (c) atomicAdd(out, c);
}

// dense float accumulation for mid-size tables: acc[row][0..63] += g  (one 16-lane group per entry, float4 per lane)
__global__ __launch_bounds__(256) void dense_add_kernel(const uint32_t* __restrict__ row, const float* __restrict__ grad, float* __restrict__ acc, int n) {
    const int gi = threadIdx.x >> 4, c4 = threadIdx.x & 15;
    for (int e = blockIdx.x * 16 + gi; e < n; e += gridDim.x * 16) {
        const float4 g = *reinterpret_cast<const float4*>(grad + (size_t)e * 64 + c4 * 4);
        float* a = acc + (size_t)row[e] * 64 + c4 * 4;
        atomicAdd(a + 0, g.x);
        atomicAdd(a + 1, g.y);
        atomicAdd(a + 2, g.z);
        atomicAdd(a + 3, g.w);
    }
}

// tiny tables: LDS accumulation per workgroup, then one partial slab per workgroup
__global__ __launch_bounds__(256) void lds_add_kernel(const uint32_t* __restrict__ row, const float* __restrict__ grad, float* __restrict__ part, int n, int rows) {
    extern __shared__ float acc[];  // rows x 64
    for (int i = threadIdx.x; i < rows * 64; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int gi = threadIdx.x >> 4, c4 = threadIdx.x & 15;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int beg = blockIdx.x * per, end = beg + per < n ? beg + per : n;
    for (int e = beg + gi; e < end; e += 16) {
        const float4 g = *reinterpret_cast<const float4*>(grad + (size_t)e * 64 + c4 * 4);
        float* a = acc + row[e] * 64 + c4 * 4;
        atomicAdd(a + 0, g.x);
        atomicAdd(a + 1, g.y);
        atomicAdd(a + 2, g.z);
        atomicAdd(a + 3, g.w);
    }
    __syncthreads();
    float* out = part + (size_t)blockIdx.x * rows * 64;
    for (int i = threadIdx.x; i < rows * 64; i += 256) out[i] = acc[i];
}

int main() {
    const int n = 26 * 65536;
    std::vector<uint32_t> h(n);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 4; };
    int *head, *next, *out;
    uint32_t* row;
    CK(hipMalloc(&row, n * 4));
    CK(hipMalloc(&next, n * 4));
    CK(hipMalloc(&out, 4));
    for (int64_t space : {6240000ll, 26000000ll, 100000000ll, 4000ll, 64ll}) {
        for (int i = 0; i < n; ++i) h[i] = rnd() % (uint32_t)space;
        CK(hipMemcpy(row, h.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&head, space * 4));
        CK(hipMemset(head, 0xff, space * 4));
        const int nb = (n + 4095) / 4096;
        float t_fill = time_us([&] { CK(hipMemsetAsync(head, 0xff, space * 4)); });
        float t_sc = time_us([&] { hipLaunchKernelGGL(scatter_kernel, dim3(nb), dim3(256), 0, 0, row, head, n); });
        float t_link = time_us([&] { hipLaunchKernelGGL(link_kernel, dim3(nb), dim3(256), 0, 0, row, head, next, n); });
        float t_probe = time_us([&] { hipLaunchKernelGGL(probe_kernel, dim3(nb), dim3(256), 0, 0, row, head, out, n); });
        printf("space %9lld rows: memset %7.1f us | scattered store %7.1f us | atomicExch link %7.1f us | head probe %7.1f us\n", (long long)space,
               t_fill, t_sc, t_link, t_probe);
        CK(hipFree(head));
    }
    // dense float atomics: 2 mid tables' worth (131072 entries) over 1311 / 2210 rows, and 8 tiny tables' worth (524288 entries) over 4..96 rows
    float *grad, *acc, *part;
    CK(hipMalloc(&grad, (size_t)524288 * 64 * 4));
    CK(hipMemset(grad, 0, (size_t)524288 * 64 * 4));
    CK(hipMalloc(&acc, (size_t)20000 * 64 * 4));
    CK(hipMalloc(&part, (size_t)1024 * 128 * 64 * 4));
    for (int rows : {1311, 2210, 10000, 96, 4}) {
        const int m = 65536;
        for (int i = 0; i < m; ++i) h[i] = rnd() % (uint32_t)rows;
        CK(hipMemcpy(row, h.data(), m * 4, hipMemcpyHostToDevice));
        float t = time_us([&] { hipLaunchKernelGGL(dense_add_kernel, dim3(1024), dim3(256), 0, 0, row, grad, acc, m); });
        printf("global float atomics, 65536 entries x 64 floats over %5d rows: %7.1f us\n", rows, t);
        if (rows <= 128) {
            for (int nbk : {256, 512, 1024}) {
                float t2 = time_us([&] { hipLaunchKernelGGL(lds_add_kernel, dim3(nbk), dim3(256), rows * 64 * 4, 0, row, grad, part, m, rows); });
                printf("   LDS accumulation, %4d workgroups (+ %d partial slabs of %d rows): %7.1f us\n", nbk, nbk, rows, t2);
            }
        }
    }
    return 0;
}
