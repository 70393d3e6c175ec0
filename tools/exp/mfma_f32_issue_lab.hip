// Lab: issue rate of v_mfma_f32_32x32x2_f32 for ONE wavefront per SIMD, two alternating accumulators, as a function of what sits
// between the MFMAs: (0) nothing, (1) a v_cndmask that writes the B operand of the NEXT MFMA into the SAME register every time (what
// the compiler generated in gemm_nt_astat_kernel), (2) the selects of 8 MFMAs done first into 8 registers, then 8 MFMAs back to back.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/mfma_f32_issue_lab.hip -o gpurun_in/lab/mfma_f32_issue_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* cyc) {
    const int h = threadIdx.x >> 5 & 1;
    f32x4 b[8];
    for (int i = 0; i < 8; ++i) b[i] = in[threadIdx.x + 256 * i];
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = in[threadIdx.x][i & 3] + i;
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * i], b[i].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * i + 1], b[i].y, acc1, 0, 0, 0);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float s = h ? b[i].y : b[i].x;
                asm volatile("" : "+v"(s));
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * i], s, acc0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                float u = h ? b[i].w : b[i].z;
                asm volatile("" : "+v"(u));
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * i + 1], u, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            float s[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s[2 * i] = h ? b[i].y : b[i].x;
                s[2 * i + 1] = h ? b[i].w : b[i].z;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * i], s[2 * i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * i + 1], s[2 * i + 1], acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const f32x4* in, float* out, unsigned long long* cyc, int wg, const char* what) {
    const int iters = 1000;
    hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(256), 0, 0, in, out, 10, cyc);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(256), 0, 0, in, out, iters, cyc);
    CK(hipDeviceSynchronize());
    unsigned long long c;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("  %-58s %d wavefront(s) per SIMD: %6.1f cycle-counter ticks per MFMA per wavefront\n", what, wg / 256, (double)c / (iters * 16.0));
}


// the shape of gemm_nt_astat_kernel's inner loop: 8 wavefronts per workgroup (TWO per SIMD, one workgroup per CU), groups of 8 MFMAs on two
// alternating accumulators; MODE 0: operands ready in registers; 1: 8 selects in front of every group; 2: + 4 ds_read_b128 per group
// (next group's fragments, issued behind the first MFMA pair); 3: as 2, but the selects are gone (each lane uses its whole float4)
template <int MODE>
__global__ __launch_bounds__(512) void k8(const f32x4* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float tile[64 * 128];
    for (int i = threadIdx.x; i < 64 * 128; i += 512) tile[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    float a[64];
    for (int i = 0; i < 64; ++i) a[i] = in[threadIdx.x & 255][i & 3] + i;
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    const int fb = l31 * 128 + ((l31 & 31) << 2);
    f32x4 b0[2][2], b1[2][2];
    for (int i = 0; i < 2; ++i) { b0[0][i] = in[lane + 64 * i]; b1[0][i] = in[lane + 64 * (i + 2)]; b0[1][i] = b0[0][i]; b1[1][i] = b1[0][i]; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            float s0[4], s1[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 x = b0[g & 1][i], y = b1[g & 1][i];
                if (MODE == 0 || MODE == 3) { s0[2 * i] = x.x; s0[2 * i + 1] = x.z; s1[2 * i] = y.y; s1[2 * i + 1] = y.w; }
                else { s0[2 * i] = h ? x.y : x.x; s0[2 * i + 1] = h ? x.w : x.z; s1[2 * i] = h ? y.y : y.x; s1[2 * i + 1] = h ? y.w : y.z; }
            }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g], s0[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g], s1[0], acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    b0[(g + 1) & 1][i] = *reinterpret_cast<const f32x4*>(tile + (fb ^ (((g * 2 + i) & 31) << 2)));
                    b1[(g + 1) & 1][i] = *reinterpret_cast<const f32x4*>(tile + 32 * 128 + (fb ^ (((g * 2 + i) & 31) << 2)));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 1; i < 4; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + i], s0[i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * g + i], s1[i], acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc1[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run8(const f32x4* in, float* out, unsigned long long* cyc, const char* what) {
    const int iters = 200;
    hipLaunchKernelGGL(k8<MODE>, dim3(256), dim3(512), 0, 0, in, out, 5, cyc);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k8<MODE>, dim3(256), dim3(512), 0, 0, in, out, iters, cyc);
    CK(hipDeviceSynchronize());
    unsigned long long c;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("  8 wavefronts per workgroup, one workgroup per CU, %-66s %6.1f ticks per MFMA of a wavefront (pipe-bound: 128)\n", what, (double)c / (iters * 128.0));
}

int main() {
    f32x4* in; float* out; unsigned long long* cyc;
    CK(hipMalloc(&in, 256 * 8 * 16)); CK(hipMemset(in, 0, 256 * 8 * 16)); CK(hipMalloc(&out, 1024 * 512 * 4)); CK(hipMalloc(&cyc, 8));
    for (int wg : {256, 512}) {
        run<0>(in, out, cyc, wg, "nothing between the MFMAs");
        run<1>(in, out, cyc, wg, "a select into the next MFMA's B operand between MFMAs");
        run<2>(in, out, cyc, wg, "16 selects first, then 16 MFMAs back to back");
    }
    run8<0>(in, out, cyc, "operands in registers:");
    run8<1>(in, out, cyc, "+ 8 selects in front of every group of 8 MFMAs:");
    run8<2>(in, out, cyc, "+ 4 ds_read_b128 per group behind the first MFMA pair:");
    run8<3>(in, out, cyc, "the 4 ds_read_b128 without the selects:");
    return 0;
}
