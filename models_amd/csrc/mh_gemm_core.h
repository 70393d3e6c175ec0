// fp32 MFMA GEMM building blocks for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF peak).
//
// Numerics contract: every output element is ONE k-ascending fp32 fmaf chain
//     acc = fmaf(a[k], b[k], acc),  k = 0 .. K-1,  acc0 = 0
// (no split-K, no multi-accumulator), which oracle/oracle_c.c restates bit for bit.  To keep
// the natural k order while still reading LDS with 16-byte ds_read_b128, k-major tiles are
// stored DE-INTERLEAVED: for a 32-wide k-tile, element k = 2s + h lives at column h*16 + s.
// The MFMA's two k-slots (lane>>5 = h) then read 4 consecutive steps s..s+3 with one b128 read,
// and step s consumes k = 2s (slot 0) then k = 2s+1 (slot 1) -- ascending.
//
// LDS bank check (ds_read_b128, 64 banks, 16-lane groups): row stride 36 floats -> row r starts
// at bank 36r mod 64 = 4*(9r mod 16): the 16 rows of any lane group hit 16 disjoint 4-bank
// slots -> conflict-free.
#pragma once
#include "mh_common.h"

namespace mhgemm {

constexpr int BK = 32;        // k-tile
constexpr int LDK = BK + 4;   // row stride (floats) of a k-major LDS tile
constexpr int NT = 256;       // threads per workgroup (4 wavefronts)

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Global -> registers -> LDS staging of a [ROWS x BK] tile of a row-major matrix whose
// contraction dimension is contiguous (A[M,K], or B[N,K] of an "NT" product).
template <int ROWS>
struct KMajorTile {
    static constexpr int NV = (ROWS * (BK / 4) + NT - 1) / NT;
    f32x4 regs[NV];

    __device__ __forceinline__ void load(const float* __restrict__ src, int64_t ld, int64_t row0,
                                         int64_t nrows, int k0, int K, bool vec_ok) {
        // interior tile: unconditional 16-byte loads (workgroup-uniform branch)
        if (vec_ok && row0 + ROWS <= nrows && k0 + BK <= K && (ROWS * (BK / 4)) % NT == 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int idx = threadIdx.x + i * NT;
                const int r = idx >> 3, c4 = idx & 7;
                regs[i] = *reinterpret_cast<const f32x4*>(src + (row0 + r) * ld + k0 + c4 * 4);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int r = idx >> 3, c4 = idx & 7;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int64_t row = row0 + r;
            const int k = k0 + c4 * 4;
            if (r < ROWS && row < nrows && k < K) {
                const float* p = src + row * ld + k;
                if (vec_ok && k + 3 < K) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else {
                    v.x = p[0];
                    if (k + 1 < K) v.y = p[1];
                    if (k + 2 < K) v.z = p[2];
                    if (k + 3 < K) v.w = p[3];
                }
            }
            regs[i] = v;
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ lds) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int r = idx >> 3, c4 = idx & 7;
            if (r < ROWS) {
                float* d = lds + r * LDK + c4 * 2;
                *reinterpret_cast<float2*>(d) = make_float2(regs[i].x, regs[i].z);       // even k -> slot 0
                *reinterpret_cast<float2*>(d + 16) = make_float2(regs[i].y, regs[i].w);  // odd k  -> slot 1
            }
        }
    }
};

// Staging of a [BK x COLS] tile of a row-major W[K, N] (the "NN" B operand, n contiguous).
template <int COLS>
struct NMajorTile {
    static constexpr int VPR = COLS / 4;
    static constexpr int NV = (BK * VPR + NT - 1) / NT;
    f32x4 regs[NV];

    __device__ __forceinline__ void load(const float* __restrict__ W, int64_t ldw, int k0, int K,
                                         int n0, int N, bool vec_ok) {
        if (vec_ok && k0 + BK <= K && n0 + COLS <= N && (BK * VPR) % NT == 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int idx = threadIdx.x + i * NT;
                const int r = idx / VPR, c4 = idx - r * VPR;
                regs[i] = *reinterpret_cast<const f32x4*>(W + (int64_t)(k0 + r) * ldw + n0 + c4 * 4);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int r = idx / VPR, c4 = idx - r * VPR;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int k = k0 + r, n = n0 + c4 * 4;
            if (r < BK && k < K && n < N) {
                const float* p = W + (int64_t)k * ldw + n;
                if (vec_ok && n + 3 < N) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else {
                    v.x = p[0];
                    if (n + 1 < N) v.y = p[1];
                    if (n + 2 < N) v.z = p[2];
                    if (n + 3 < N) v.w = p[3];
                }
            }
            regs[i] = v;
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ lds) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * NT;
            const int r = idx / VPR, c4 = idx - r * VPR;
            if (r < BK) *reinterpret_cast<f32x4*>(lds + r * COLS + c4 * 4) = regs[i];
        }
    }
};

// One k-tile (32 k values = 16 MFMA steps) of a wave's TM x TN grid of 32x32 accumulators.
// As: k-major tile, rows a_row0 + tm*32 + (lane&31).
// B_NT: Bs k-major tile, rows b_row0 + tn*32 + (lane&31); else Bs is n-major [BK][ldb].
template <int TM, int TN, bool B_NT>
struct Frag {
    f32x4 a[TM];
    f32x4 b[TN];  // 4 consecutive k-steps of one k-slot
    __device__ __forceinline__ void load(const float* __restrict__ As, int a_row0, const float* __restrict__ Bs,
                                         int b_row0, int ldb, int g, int l31, int h) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
            a[tm] = *reinterpret_cast<const f32x4*>(As + (a_row0 + tm * 32 + l31) * LDK + h * 16 + g * 4);
        if (B_NT) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                b[tn] = *reinterpret_cast<const f32x4*>(Bs + (b_row0 + tn * 32 + l31) * LDK + h * 16 + g * 4);
        } else {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int j = 0; j < 4; ++j) b[tn][j] = Bs[(2 * (4 * g + j) + h) * ldb + b_row0 + tn * 32 + l31];
        }
    }
};

// LDS -> register fragments are double-buffered across the four 4-step groups of a k-tile: the
// reads of group g+1 are issued before the 4*TM*TN MFMAs of group g, so the matrix pipe never waits
// on an LDS round trip (hipcc otherwise places each read right in front of its first use).
template <int TM, int TN, bool B_NT>
__device__ __forceinline__ void mma_ktile(const float* __restrict__ As, int a_row0,
                                          const float* __restrict__ Bs, int b_row0, int ldb,
                                          f32x16 (&acc)[TM][TN]) {
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, h = lane >> 5;
    Frag<TM, TN, B_NT> f0, f1;
    f0.load(As, a_row0, Bs, b_row0, ldb, 0, l31, h);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        Frag<TM, TN, B_NT>& cur = (g & 1) ? f1 : f0;
        Frag<TM, TN, B_NT>& nxt = (g & 1) ? f0 : f1;
        if (g < 3) nxt.load(As, a_row0, Bs, b_row0, ldb, g + 1, l31, h);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE this group's MFMAs
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(cur.a[tm][j], cur.b[tn][j], acc[tm][tn]);
    }
}

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
}

// C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int acc_col(int lane) { return lane & 31; }

}  // namespace mhgemm
