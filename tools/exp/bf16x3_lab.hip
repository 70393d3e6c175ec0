// Lab (NOT the product, NOT an f32 line): the in-batch scorer forward with a 3-term split-bf16 product on the bf16 MFMA
// (v_mfma_f32_32x32x16_bf16: 16 x the fp32 MFMA rate on gfx950).
//
//   x = hi + lo + r,  hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-17 |x|
//   q . n  ~=  sum_k  hi_q hi_n + hi_q lo_n + lo_q hi_n          (the dropped lo lo and r terms are <= ~2^-16 |q_k n_k| each)
//   lse_i  =   log sum_j exp((q_i . n_j) / T)                     (fp32 accumulators, fp32 online softmax)
//
// What it answers: the fp32 scorer is MFMA-bound (0.70 of the 157 TF fp32 peak = 2.46 ms at 32768 x 32768 x 128).  Is the
// split product a way under that floor, and what does it cost in accuracy?  It prints the time of the split pre-pass and
// of the scorer, the "fp32-equivalent" TF/s (2 B N E flops over the time), and the error of the raw dot products and of
// the LSE against fp64 at a size the host can check, next to the error of a plain fp32 fmaf chain on the same inputs.
//
// Layout: a workgroup (4 wavefronts) owns 256 queries and one split of the candidates.  The query fragments of a wavefront
// (64 queries x 128 k x {hi, lo} = 128 VGPRs) stay in REGISTERS for the whole kernel; candidate tiles (32 rows x 128 k x
// {hi, lo} = 16 KB) stream through a 4-deep LDS ring by DMA (global_load_lds_dwordx4, chunk-swizzled source addresses so
// that ds_read_b128 of 16 consecutive rows hits 16 distinct 4-bank groups).  The product is TRANSPOSED (rows = candidates,
// columns = queries): a lane of the 32 x 32 C layout holds 16 candidates of ONE query, so the online (max, sum) is a
// register loop; the two half-wavefronts and the candidate splits are merged once at the end.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/exp/bf16x3_lab.hip -o gpurun_in/bf16x3_lab && gpurun_in/bf16x3_lab
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int E = 128;              // embedding width = the whole K
constexpr int KS = E / 16;          // MFMA k-steps
constexpr int NWV = 4;              // wavefronts per workgroup
constexpr int QW = 64;              // queries per wavefront (two 32-wide column blocks)
constexpr int QB = QW * NWV;        // queries per workgroup
constexpr int CT = 32;              // candidates per tile
constexpr int STAGES = 4;
constexpr int ARR_BYTES = CT * E * 2;      // one of {hi, lo} of a tile: 8 KB
constexpr int TILE_BYTES = 2 * ARR_BYTES;  // 16 KB
constexpr int DMA_PER_THREAD = TILE_BYTES / (NWV * 64 * 16);  // 4

__device__ __forceinline__ uint16_t bf16_rne(float x) {
    uint32_t u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);  // finite inputs only (lab)
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// x[n] -> hi[n], lo[n]  (4 values per thread)
__global__ __launch_bounds__(256) void split_kernel(const float4* __restrict__ x, uint2* __restrict__ hi, uint2* __restrict__ lo, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = x[i];
    const float a[4] = {v.x, v.y, v.z, v.w};
    uint16_t h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = bf16_rne(a[j]);
        l[j] = bf16_rne(a[j] - bf16_f32(h[j]));
    }
    hi[i] = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    lo[i] = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
}

__device__ __forceinline__ void dma16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm_and_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

typedef float f32x2 __attribute__((ext_vector_type(2)));

// (m, s) of two queries (tn = 0, 1) <- merged with 16 more logits each; log2 domain, raw v_exp_f32 (arguments <= 0: a result
// in the denormal range flushes to zero, which is what a sum of exponentials wants), packed fma / add
__device__ __forceinline__ void online_softmax(const f32x16 (&acc)[2], float (&m)[2], float (&s)[2], float scale2) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        float mx = acc[tn][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = __builtin_fmaxf(mx, acc[tn][i]);
        const float mn = __builtin_fmaxf(m[tn], mx * scale2);
        // packed fma / add.  (Unpacked v_fma / v_add -- the SLP vectoriser re-packs plain C, so the adds were inline asm -- timed the
        // same here, and inline asm right behind v_exp_f32 skips the compiler's trans -> VALU hazard nop: wrong sums.)
        const f32x2 sc = {scale2, scale2}, nm = {-mn, -mn};
        f32x2 sum2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const f32x2 a = {acc[tn][i], acc[tn][i + 1]};
            const f32x2 e = __builtin_elementwise_fma(a, sc, nm);
            const f32x2 p = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
            sum2 += p;
        }
        const float sum = sum2[0] + sum2[1];
        s[tn] = fmaf(s[tn], __builtin_amdgcn_exp2f(m[tn] - mn), sum);
        m[tn] = mn;
    }
}

// part[split][query] = (m, s) in the log2 domain: sum_j 2^(z_j - m), z = dot * scale2  (scale2 = log2(e) / T)
// TERMS = 3: the split product; TERMS = 1: hi hi only (plain bf16 inputs, for the error table)
// ABL (ablations, timing only -- results are garbage): 1 = no candidate DMA inside the loop, 2 = no softmax, 3 = no MFMA
template <int TERMS, bool DEBUG_Z, bool PIPE, int ABL = 0>
__global__ __launch_bounds__(NWV * 64, 2) void scorer_bf16x3_kernel(const uint16_t* __restrict__ nhi, const uint16_t* __restrict__ nlo,
                                                                    const uint16_t* __restrict__ qhi, const uint16_t* __restrict__ qlo,
                                                                    int Nn, int B, float scale2, float2* __restrict__ part, int nsplit,
                                                                    float* __restrict__ zdbg) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int qb = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
    const int per = Nn / nsplit;         // candidates of this split (a multiple of CT)
    const int64_t c0 = (int64_t)sp * per;
    const int T = per / CT;

    // query fragments: B operand of the MFMA, lane = column l31, k = 16 ks + 8 h .. + 7
    bf16x8 qh[2][KS], ql[2][KS];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int64_t qi = (int64_t)qb * QB + wave * QW + tn * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qh[tn][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qhi + qi * E + ks * 16 + h * 8));
            if (TERMS == 3) ql[tn][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qlo + qi * E + ks * 16 + h * 8));
        }
    }
    // the DMA sources of this thread inside a tile: chunk position L = (j NWV + wave) 64 + lane of the 1024-chunk tile image
    const uint16_t* src[DMA_PER_THREAD];
#pragma unroll
    for (int j = 0; j < DMA_PER_THREAD; ++j) {
        const int L = (j * NWV + wave) * 64 + lane;
        const int arr = L >> 9, Lp = L & 511, r = Lp >> 4, p = Lp & 15, c = p ^ (r & 15);
        src[j] = (arr ? nlo : nhi) + (c0 + r) * E + c * 8;
    }
    auto issue = [&](int t) {
        unsigned char* st = smem + (t % STAGES) * TILE_BYTES;
#pragma unroll
        for (int j = 0; j < DMA_PER_THREAD; ++j) {
            if (TERMS == 1 && j >= DMA_PER_THREAD / 2) break;
            dma16(src[j] + (int64_t)t * CT * E, st + (j * NWV + wave) * 1024);
        }
    };
    constexpr int PER_TILE = TERMS == 3 ? DMA_PER_THREAD : DMA_PER_THREAD / 2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the query fragments: keeps the ring's vmcnt arithmetic exact
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < T) issue(t);

    float m[2] = {-INFINITY, -INFINITY}, s[2] = {0.f, 0.f};
    // before the first tile: 16 'logits' far below any real one -- their sum is wiped by the first real rescale (2^(m - m') = 0).
    // Not -1e30: fma(a, scale, -round(a scale)) returns the ROUNDING ERROR of the product, ~1e24 there, and 2^1e24 = inf.
    f32x16 prev[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) prev[0][i] = prev[1][i] = -65536.f;
    const int rd = l31 * 256;  // byte offset of this lane's row inside a tile array
    for (int t = 0; t < T; ++t) {
        // tile t is complete when at most the loads of tiles t+1 .. t+STAGES-2 are outstanding
        if (ABL != 1 && t + STAGES - 2 < T) wait_vm_and_barrier<(STAGES - 2) * PER_TILE>();
        else wait_vm_and_barrier<0>();
        if (ABL != 1 && t + STAGES - 1 < T) issue(t + STAGES - 1);
        const unsigned char* st = smem + (t % STAGES) * TILE_BYTES;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int pos = ((2 * ks + h) ^ (l31 & 15)) * 16;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + rd + pos));
            if (ABL == 3) {
                const uint4 u = __builtin_bit_cast(uint4, ah);
                acc[0][ks] = __uint_as_float(u.x & 0x3fffffffu);
                acc[1][ks] = __uint_as_float(u.y & 0x3fffffffu);
                if (TERMS == 3) {
                    const uint4 v = *reinterpret_cast<const uint4*>(st + ARR_BYTES + rd + pos);
                    acc[0][8 + ks] = __uint_as_float(v.x & 0x3fffffffu);
                    acc[1][8 + ks] = __uint_as_float(v.y & 0x3fffffffu);
                }
                continue;
            }
            if (TERMS == 3) {
                const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + ARR_BYTES + rd + pos));
                acc[0] = mfma_bf16(al, qh[0][ks], acc[0]);
                acc[1] = mfma_bf16(al, qh[1][ks], acc[1]);
                acc[0] = mfma_bf16(ah, ql[0][ks], acc[0]);
                acc[1] = mfma_bf16(ah, ql[1][ks], acc[1]);
            }
            acc[0] = mfma_bf16(ah, qh[0][ks], acc[0]);
            acc[1] = mfma_bf16(ah, qh[1][ks], acc[1]);
        }
        if (DEBUG_Z) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int64_t cand = c0 + (int64_t)t * CT + (i / 4) * 8 + h * 4 + (i % 4);
                    const int64_t qi = (int64_t)qb * QB + wave * QW + tn * 32 + l31;
                    zdbg[cand * B + qi] = acc[tn][i];
                }
        }
        if (ABL == 2) {
            s[0] += acc[0][0] + acc[0][15];
            s[1] += acc[1][0] + acc[1][15];
        } else if (PIPE) {
            // the softmax of the PREVIOUS tile's logits is independent of this tile's MFMAs: one basic block, and the
            // scheduling groups below ask for 1 MFMA : 2 VALU so that the matrix pipe and the vector ALU run side by side
            online_softmax(prev, m, s, scale2);
            prev[0] = acc[0];
            prev[1] = acc[1];
#pragma unroll
            for (int i = 0; i < (TERMS == 3 ? 48 : 16); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, TERMS == 3 ? 2 : 6, 0);
            }
        } else {
            online_softmax(acc, m, s, scale2);
        }
    }
    if (PIPE && ABL != 2) online_softmax(prev, m, s, scale2);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {  // join the two half-wavefronts (same query, other candidates)
        const float m2 = __shfl_xor(m[tn], 32), s2 = __shfl_xor(s[tn], 32);
        const float mm = fmaxf(m[tn], m2);
        const float ss = s[tn] * exp2f(m[tn] - mm) + s2 * exp2f(m2 - mm);
        if (h == 0) part[(int64_t)sp * B + (int64_t)qb * QB + wave * QW + tn * 32 + l31] = make_float2(mm, ss);
    }
}

__global__ void finalize_kernel(const float2* __restrict__ part, int nsplit, int B, float* __restrict__ lse) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    float m = -INFINITY, s = 0.f;
    for (int k = 0; k < nsplit; ++k) {
        const float2 p = part[(int64_t)k * B + i];
        const float mm = fmaxf(m, p.x);
        s = s * exp2f(m - mm) + p.y * exp2f(p.x - mm);
        m = mm;
    }
    lse[i] = (m + log2f(s)) * 0.69314718055994530942f;
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

struct Split {
    uint16_t *hi, *lo;
};

template <int TERMS, bool DBG, bool PIPE, int ABL = 0>
void launch(const Split& n, const Split& q, int Nn, int B, float inv_T, float2* part, int nsplit, float* z, float* lse) {
    auto kern = scorer_bf16x3_kernel<TERMS, DBG, PIPE, ABL>;
    const size_t lds = (size_t)STAGES * TILE_BYTES;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((B / QB) * nsplit), dim3(NWV * 64), lds, 0, n.hi, n.lo, q.hi, q.lo, Nn, B, inv_T * 1.44269504088896340736f, part,
                       nsplit, z);
    hipLaunchKernelGGL(finalize_kernel, dim3((B + 255) / 256), dim3(256), 0, 0, part, nsplit, B, lse);
}

}  // namespace

int main() {
    const float inv_T = 1.f / 0.05f;  // the retrieval configs' temperature region: errors of the logits are x 20 those of the dot products
    // ---------------- accuracy at a host-checkable size ----------------
    {
        const int Nn = 2048, B = 1024, nsplit = 4;
        std::vector<float> hq((size_t)B * E), hn((size_t)Nn * E);
        uint32_t sd = 12345u;
        auto rnd = [&]() { sd = sd * 1664525u + 1013904223u; return ((sd >> 8) & 0xffff) / 65536.f * 2.f - 1.f; };
        // L2-normalised rows (what the towers feed the scorer): |dot| <= 1
        auto fill = [&](std::vector<float>& v, int rows) {
            for (int r = 0; r < rows; ++r) {
                double ss = 0;
                for (int k = 0; k < E; ++k) { v[(size_t)r * E + k] = rnd(); ss += (double)v[(size_t)r * E + k] * v[(size_t)r * E + k]; }
                const float inv = (float)(1.0 / sqrt(ss));
                for (int k = 0; k < E; ++k) v[(size_t)r * E + k] *= inv;
            }
        };
        fill(hq, B);
        fill(hn, Nn);
        float *dq, *dn, *z, *lse;
        Split sq, sn;
        float2* part;
        CK(hipMalloc(&dq, hq.size() * 4));
        CK(hipMalloc(&dn, hn.size() * 4));
        CK(hipMalloc(&sq.hi, hq.size() * 2));
        CK(hipMalloc(&sq.lo, hq.size() * 2));
        CK(hipMalloc(&sn.hi, hn.size() * 2));
        CK(hipMalloc(&sn.lo, hn.size() * 2));
        CK(hipMalloc(&z, (size_t)Nn * B * 4));
        CK(hipMalloc(&lse, B * 4));
        CK(hipMalloc(&part, (size_t)nsplit * B * 8));
        CK(hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dn, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(split_kernel, dim3((unsigned)((hq.size() / 4 + 255) / 256)), dim3(256), 0, 0, (const float4*)dq, (uint2*)sq.hi, (uint2*)sq.lo, (int64_t)hq.size() / 4);
        hipLaunchKernelGGL(split_kernel, dim3((unsigned)((hn.size() / 4 + 255) / 256)), dim3(256), 0, 0, (const float4*)dn, (uint2*)sn.hi, (uint2*)sn.lo, (int64_t)hn.size() / 4);
        // fp64 reference + the plain fp32 fmaf chain on the host
        std::vector<double> z64((size_t)Nn * B), lse64(B);
        std::vector<float> z32((size_t)Nn * B);
        for (int j = 0; j < Nn; ++j)
            for (int i = 0; i < B; ++i) {
                double a = 0;
                float f = 0.f;
                for (int k = 0; k < E; ++k) {
                    a += (double)hq[(size_t)i * E + k] * hn[(size_t)j * E + k];
                    f = fmaf(hq[(size_t)i * E + k], hn[(size_t)j * E + k], f);
                }
                z64[(size_t)j * B + i] = a;
                z32[(size_t)j * B + i] = f;
            }
        for (int i = 0; i < B; ++i) {
            double mx = -1e300, sum = 0;
            for (int j = 0; j < Nn; ++j) mx = fmax(mx, z64[(size_t)j * B + i] * inv_T);
            for (int j = 0; j < Nn; ++j) sum += exp(z64[(size_t)j * B + i] * inv_T - mx);
            lse64[i] = mx + log(sum);
        }
        double e32 = 0;
        for (size_t i = 0; i < z64.size(); ++i) e32 = fmax(e32, fabs((double)z32[i] - z64[i]));
        printf("accuracy, %d candidates x %d queries x %d, L2-normalised rows, 1/T = %.0f (fp64 reference on the host)\n", Nn, B, E, inv_T);
        printf("  plain fp32 fmaf chain      : max |dot - dot64| = %.3e\n", e32);
        std::vector<float> hz((size_t)Nn * B), hl(B);
        for (int terms : {3, 4, 1}) {  // 4: the 3-term kernel with the software-pipelined softmax
            if (terms == 3) launch<3, true, false>(sn, sq, Nn, B, inv_T, part, nsplit, z, lse);
            else if (terms == 4) launch<3, true, true>(sn, sq, Nn, B, inv_T, part, nsplit, z, lse);
            else launch<1, true, false>(sn, sq, Nn, B, inv_T, part, nsplit, z, lse);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hz.data(), z, hz.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hl.data(), lse, B * 4, hipMemcpyDeviceToHost));
            double ez = 0, el = 0;
            for (size_t i = 0; i < z64.size(); ++i) ez = fmax(ez, fabs((double)hz[i] - z64[i]));
            int bad = 0;
            for (int i = 0; i < B; ++i) {
                if (!(fabs((double)hl[i]) < 1e30)) ++bad;
                else el = fmax(el, fabs((double)hl[i] - lse64[i]));
            }
            if (bad) printf("  !! %d non-finite lse values (lse[0] = %g, fp64 %g)\n", bad, hl[0], lse64[0]);
            printf("  bf16 MFMA, %d term%s: max |dot - dot64| = %.3e  -> logits (x 1/T) %.3e;  max |lse - lse64| = %.3e\n", terms == 1 ? 1 : 3, terms == 3 ? "s (split)           " : terms == 4 ? "s (split, pipelined)" : "  (plain)            ",
                   ez, ez * inv_T, el);
        }
        // rank agreement: how many queries keep their fp64 top-1 / their fp64 top-10 SET under the split product
        int top1 = 0, top10 = 0;
        launch<3, true, true>(sn, sq, Nn, B, inv_T, part, nsplit, z, lse);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hz.data(), z, hz.size() * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < B; ++i) {
            std::vector<int> a(Nn), b(Nn);
            for (int j = 0; j < Nn; ++j) a[j] = b[j] = j;
            auto by64 = [&](int x, int y) { return z64[(size_t)x * B + i] > z64[(size_t)y * B + i] || (z64[(size_t)x * B + i] == z64[(size_t)y * B + i] && x < y); };
            auto byz = [&](int x, int y) { return hz[(size_t)x * B + i] > hz[(size_t)y * B + i] || (hz[(size_t)x * B + i] == hz[(size_t)y * B + i] && x < y); };
            std::partial_sort(a.begin(), a.begin() + 10, a.end(), by64);
            std::partial_sort(b.begin(), b.begin() + 10, b.end(), byz);
            top1 += a[0] == b[0];
            bool same = true;
            for (int k = 0; k < 10; ++k) same = same && a[k] == b[k];
            top10 += same;
        }
        printf("  split product vs fp64 ordering: top-1 identical for %d / %d queries, ordered top-10 identical for %d / %d\n", top1, B, top10, B);
        printf("  (a top-k built on it needs the exact fp32 re-scoring of a margin of survivors to stay bit-exact: see DESIGN.md)\n");
        CK(hipFree(dq)); CK(hipFree(dn)); CK(hipFree(sq.hi)); CK(hipFree(sq.lo)); CK(hipFree(sn.hi)); CK(hipFree(sn.lo)); CK(hipFree(z)); CK(hipFree(lse)); CK(hipFree(part));
    }
    // ---------------- speed at the C3 size ----------------
    {
        const int Nn = 32768, B = 32768;
        float *dq, *dn, *lse;
        Split sq, sn;
        float2* part;
        const size_t nq = (size_t)B * E, nn = (size_t)Nn * E;
        CK(hipMalloc(&dq, nq * 4));
        CK(hipMalloc(&dn, nn * 4));
        CK(hipMalloc(&sq.hi, nq * 2));
        CK(hipMalloc(&sq.lo, nq * 2));
        CK(hipMalloc(&sn.hi, nn * 2));
        CK(hipMalloc(&sn.lo, nn * 2));
        CK(hipMalloc(&lse, B * 4));
        CK(hipMalloc(&part, (size_t)16 * B * 8));
        std::vector<float> hx(nq);
        uint32_t sd = 777u;
        for (auto& v : hx) { sd = sd * 1664525u + 1013904223u; v = (((sd >> 8) & 0xffff) / 65536.f * 2.f - 1.f) * 0.088f; }
        CK(hipMemcpy(dq, hx.data(), nq * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dn, hx.data(), nn * 4, hipMemcpyHostToDevice));
        const float t_split = time_us([&] {
            hipLaunchKernelGGL(split_kernel, dim3((unsigned)((nq / 4 + 255) / 256)), dim3(256), 0, 0, (const float4*)dq, (uint2*)sq.hi, (uint2*)sq.lo, (int64_t)nq / 4);
            hipLaunchKernelGGL(split_kernel, dim3((unsigned)((nn / 4 + 255) / 256)), dim3(256), 0, 0, (const float4*)dn, (uint2*)sn.hi, (uint2*)sn.lo, (int64_t)nn / 4);
        }, 20);
        printf("speed, %d candidates x %d queries x %d\n", Nn, B, E);
        printf("  split pre-pass (queries + candidates -> hi, lo): %7.1f us\n", t_split);
        const double flops = 2.0 * Nn * (double)B * E;
        for (int nsplit : {2, 4, 8, 16}) {
            const float t3 = time_us([&] { launch<3, false, false>(sn, sq, Nn, B, inv_T, part, nsplit, nullptr, lse); }, 10);
            const float t3p = time_us([&] { launch<3, false, true>(sn, sq, Nn, B, inv_T, part, nsplit, nullptr, lse); }, 10);
            const float t1 = time_us([&] { launch<1, false, false>(sn, sq, Nn, B, inv_T, part, nsplit, nullptr, lse); }, 10);
            const float t1p = time_us([&] { launch<1, false, true>(sn, sq, Nn, B, inv_T, part, nsplit, nullptr, lse); }, 10);
            printf("  %2d candidate splits (%4d workgroups): 3-term %7.1f us, pipelined %7.1f us = %6.1f fp32-equivalent TF (%6.1f bf16 TF, %.3f of 2500);  1-term %7.1f / %7.1f us\n",
                   nsplit, (B / QB) * nsplit, t3, t3p, flops / t3p * 1e-6, 3 * flops / t3p * 1e-6, 3 * flops / t3p * 1e-6 / 2500.0, t1, t1p);
        }
        for (int nsplit : {2, 4}) {  // where the time goes: the 3-term kernel with one part removed (results are garbage, timing only)
            const float a1 = time_us([&] { launch<3, false, false, 1>(sn, sq, Nn, B, inv_T, part, nsplit, nullptr, lse); }, 10);
            const float a2 = time_us([&] { launch<3, false, false, 2>(sn, sq, Nn, B, inv_T, part, nsplit, nullptr, lse); }, 10);
            const float a3 = time_us([&] { launch<3, false, false, 3>(sn, sq, Nn, B, inv_T, part, nsplit, nullptr, lse); }, 10);
            const float a2p = time_us([&] { launch<3, false, true, 1>(sn, sq, Nn, B, inv_T, part, nsplit, nullptr, lse); }, 10);
            printf("  ablations, %d splits: no candidate DMA %7.1f us (pipelined %7.1f) | no softmax %7.1f us | no MFMA %7.1f us\n", nsplit, a1, a2p, a2, a3);
        }
        {  // the two schedules of the 3-term kernel agree (same arithmetic, other order of the running-max updates)
            std::vector<float> a(B), b(B);
            launch<3, false, false>(sn, sq, Nn, B, inv_T, part, 4, nullptr, lse);
            CK(hipMemcpy(a.data(), lse, B * 4, hipMemcpyDeviceToHost));
            launch<3, false, true>(sn, sq, Nn, B, inv_T, part, 4, nullptr, lse);
            CK(hipMemcpy(b.data(), lse, B * 4, hipMemcpyDeviceToHost));
            double d = 0;
            int bad = 0;
            for (int i = 0; i < B; ++i) {
                if (!(fabs((double)a[i]) < 1e30) || !(fabs((double)b[i]) < 1e30)) ++bad;
                else d = fmax(d, fabs((double)a[i] - b[i]));
            }
            printf("  plain vs pipelined schedule at full size: max |lse difference| = %.3e, non-finite %d (lse[0] = %g / %g)\n", d, bad, a[0], b[0]);
        }
        printf("  (fp32 MFMA scorer forward on the same problem: tools/exp/scorer_lab = 2.46-2.50 ms, product stream kernel 2.52-2.62 ms)\n");
    }
    return 0;
}
