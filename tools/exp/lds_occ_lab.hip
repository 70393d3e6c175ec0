// Lab: how many 256-thread workgroups with `lds` bytes of dynamic LDS does the runtime place on one gfx950 CU?  (The allocation
// granule decides whether three 53 952-byte workgroups -- dlrm_fused_bwd at F = 28 -- share the 160 KB of a CU.)
//   hipcc -O3 --offload-arch=gfx950 tools/exp/lds_occ_lab.hip -o gpurun_in/lds_occ_lab && gpurun_in/lds_occ_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(float* out, int n) {
    extern __shared__ float s[];
    s[threadIdx.x] = (float)n;
    __syncthreads();
    if (out) out[threadIdx.x] = s[(threadIdx.x + 1) & 255];
}
int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int lds = 50 * 1024; lds <= 56 * 1024; lds += 256) {
        int occ = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, lds);
        printf("lds %6d B (%5.2f KB): %d workgroups per CU\n", lds, lds / 1024.0, occ);
    }
    for (int lds : {52096, 53952, 53504, 53248, 54272, 54613}) {
        int occ = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, lds);
        printf("lds %6d B: %d\n", lds, occ);
    }
    return 0;
}
