// Lab: the READ-ONLY ceiling of a random 256-byte row gather on MI355X.  mh_embedding_gather's cold figure (5.3-6.3 TB/s) counts the
// rows read AND the copy written; the fused gather -> interaction kernel writes only a quarter of what it reads, so its ceiling is the
// rate at which random rows can be READ.  Every 16-lane group fetches rows (one float4 per lane) and sums them in registers; U rows per
// group are in flight at a time; one float4 per lane is written at the end.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/gather_read_lab.hip -o gpurun_in/lab/gather_read_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ids[n]: group g handles ids g, g + G, g + 2G, ... (G = number of 16-lane groups); U independent loads per lane in flight
template <int U, bool NT>
__global__ __launch_bounds__(256) void gather_sum_kernel(const float* __restrict__ table, const int* __restrict__ ids, int64_t n,
                                                       f32x4* __restrict__ out) {
    const int64_t G = (int64_t)gridDim.x * 16;
    const int64_t g = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int c4 = threadIdx.x & 15;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t e = g; e < n; e += G * U) {
        int id[U];
#pragma unroll
        for (int u = 0; u < U; ++u) id[u] = (e + u * G < n) ? ids[e + u * G] : -1;
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4* p = reinterpret_cast<const f32x4*>(table + (int64_t)(id[u] < 0 ? 0 : id[u]) * 64) + c4;
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (id[u] >= 0) acc += v[u];
    }
    out[(int64_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename F>
float time_us(F&& f, int iters) {
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

template <int U, bool NT>
void run(const float* table, const int* ids, int64_t n, f32x4* out, int wg, const char* what) {
    const float t = time_us([&] { hipLaunchKernelGGL((gather_sum_kernel<U, NT>), dim3(wg), dim3(256), 0, 0, table, ids, n, out); }, 10);
    printf("  %-22s U=%2d %s %5d workgroups: %7.1f us  %6.2f TB/s read  (%5.1f G rows/s)\n", what, U, NT ? "nt" : "  ", wg, t, n * 256.0 / t * 1e-6, n / t * 1e-3);
}

int main() {
    const int64_t n = 26 * 65536;  // the row fetches of one DLRM batch
    std::vector<int> h(n);
    f32x4* out;
    int* ids;
    CK(hipMalloc(&out, (size_t)8192 * 256 * 16));
    CK(hipMalloc(&ids, n * 4));
    for (int64_t rows : {50000000ll, 6240000ll, 1000000ll, 100000ll}) {
        float* table;
        CK(hipMalloc(&table, (size_t)rows * 256));
        CK(hipMemset(table, 0, (size_t)rows * 256));
        uint64_t s = 88172645463325252ull;
        for (auto& v : h) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            v = (int)(s % (uint64_t)rows);
        }
        CK(hipMemcpy(ids, h.data(), n * 4, hipMemcpyHostToDevice));
        char what[64];
        snprintf(what, sizeof what, "%lld rows (%.1f GB)", (long long)rows, rows * 256.0 / 1e9);
        printf("%s, %lld random row reads of 256 B\n", what, (long long)n);
        run<1, false>(table, ids, n, out, 4096, "uniform ids");
        run<2, false>(table, ids, n, out, 4096, "uniform ids");
        run<4, false>(table, ids, n, out, 2048, "uniform ids");
        run<4, false>(table, ids, n, out, 4096, "uniform ids");
        run<8, false>(table, ids, n, out, 1024, "uniform ids");
        run<8, false>(table, ids, n, out, 2048, "uniform ids");
        run<8, true>(table, ids, n, out, 2048, "uniform ids");
        run<16, false>(table, ids, n, out, 1024, "uniform ids");
        run<16, true>(table, ids, n, out, 1024, "uniform ids");
        CK(hipFree(table));
    }
    return 0;
}
