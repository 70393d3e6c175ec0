// DLRM pairwise dot interaction for gfx950.
// Reference: DotProductInteraction.call (merlin/models/tf/blocks/interaction.py:86-116):
//   Z = X X^T per sample, keep the strict upper triangle in row-major (i<j) order, then the
//   "concat" aggregation with the shortcut branch (core/combinators.py:669-693,
//   core/aggregation.py:54-66).  The reference's DLRM order is [bottom_mlp_out | interactions]: the shortcut branch is
//   Filter("bottom_block"), whose dict output keeps the key "bottom_block" (core/tabular.py:552-576), ParallelBlock.call
//   merges dict-valued branches by `update` (core/combinators.py:564-569) and "bottom_block" < "sequential_block_<n>" in
//   ConcatFeatures' sorted-key order; torch twin: cat((continuous, interactions)) (torch/blocks/dlrm.py:102-104).
//   Every kernel here takes the column offsets of both parts of an output row -- `pofs` (pairs) and `tofs` (the T shortcut
//   columns): tail_first (the reference's order) = (T, 0), appended = (0, P).
//
// One wavefront owns one sample at a time.  X[F<=32, D] is staged in LDS ([32][D+4] fp32, rows
// >= F zero) and Z is formed on the matrix pipe with v_mfma_f32_16x16x4_f32: only the three
// 16x16 tiles that intersect the upper triangle are computed -- (0,0), (0,1), (1,1) -- and A == B
// for X X^T, so one LDS fragment feeds both operands.  The four k-slots of the instruction take
// the four contiguous quarter-rows [q*D/4, (q+1)*D/4) (the contraction order is
// s, D/4+s, D/2+s, 3D/4+s for s = 0..D/4-1; documented so the oracle can restate it).
// HBM-bound: algorithmic bytes/sample = F*D*4 + (F(F-1)/2 + T)*4 (+ T*4 tail read).
#include "mh_common.h"

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int IMAXF = 32;

__device__ __forceinline__ void store_tile(const f32x4& acc, int ti, int tj, int lane, int F,
                                           float* __restrict__ orow) {
    const int j = 16 * tj + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 16 * ti + (lane >> 4) * 4 + r;
        if (i < j && j < F) orow[i * (2 * F - i - 1) / 2 + (j - i - 1)] = acc[r];
    }
}

// block = 256 threads = 4 wavefronts, each with a private [32][LD] LDS slab.
__global__ __launch_bounds__(256) void dot_interaction_fwd_kernel(const float* __restrict__ x, int64_t B,
                                                                 int F, int D,
                                                                 const float* __restrict__ tail,
                                                                 int64_t ld_tail, int T, int tail_first,
                                                                 float* __restrict__ out, int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LD = D + 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* Xs = smem + wave * IMAXF * LD;
    const int i16 = lane & 15, q = lane >> 4;
    const int P = F * (F - 1) / 2;
    const int pofs = tail_first ? T : 0, tofs = tail_first ? 0 : P;
    const int vpr = D / 4;            // float4 per row
    const int nvec = F * vpr;         // float4 per sample
    const int qoff = q * (D / 4);
    const int steps = D / 4;          // MFMA steps (each consumes 4 k)

    // zero the padding rows once (rows F..31 are never rewritten)
    for (int idx = lane; idx < (IMAXF - F) * LD; idx += 64) Xs[F * LD + idx] = 0.f;

    const int64_t nblk_samples = (B + 3) / 4;
    for (int64_t it = blockIdx.x; it < nblk_samples; it += gridDim.x) {
        const int64_t b = it * 4 + wave;
        const bool live = b < B;
        __syncthreads();  // previous iteration's fragment reads are done
        if (live) {
            const f32x4* src = reinterpret_cast<const f32x4*>(x + b * (int64_t)F * D);
            for (int idx = lane; idx < nvec; idx += 64) {
                const int r = idx / vpr, c4 = idx - r * vpr;
                *reinterpret_cast<f32x4*>(Xs + r * LD + c4 * 4) = src[idx];
            }
        }
        __syncthreads();
        if (!live) continue;
        f32x4 acc00 = {0.f, 0.f, 0.f, 0.f}, acc01 = acc00, acc11 = acc00;
        const float* r0 = Xs + i16 * LD + qoff;
        const float* r1 = Xs + (16 + i16) * LD + qoff;
        const bool two = F > 16;
        int s = 0;
        for (; s + 3 < steps; s += 4) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(r0 + s);
            if (two) {
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(r1 + s);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc00 = mfma16(a0[j], a0[j], acc00);
                    acc01 = mfma16(a0[j], a1[j], acc01);
                    acc11 = mfma16(a1[j], a1[j], acc11);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc00 = mfma16(a0[j], a0[j], acc00);
            }
        }
        for (; s < steps; ++s) {
            const float a0 = r0[s];
            acc00 = mfma16(a0, a0, acc00);
            if (two) {
                const float a1 = r1[s];
                acc01 = mfma16(a0, a1, acc01);
                acc11 = mfma16(a1, a1, acc11);
            }
        }
        float* orow = out + b * ldo;
        store_tile(acc00, 0, 0, lane, F, orow + pofs);
        if (two) {
            store_tile(acc01, 0, 1, lane, F, orow + pofs);
            store_tile(acc11, 1, 1, lane, F, orow + pofs);
        }
        if (tail) {
            const float* trow = tail + b * ld_tail;
            for (int t = lane; t < T; t += 64) orow[tofs + t] = trow[t];
        }
    }
}


// Pipelined variant (F*D <= 2048 floats, D % 16 == 0): wavefronts are fully independent (private
// LDS slab, no workgroup barrier); the NEXT sample's rows are prefetched into registers while the
// current one is on the matrix pipe, so HBM latency is hidden by MFMA work instead of by occupancy.
template <int NV, bool STAGED>
__global__ __launch_bounds__(256) void dot_interaction_fwd_pipe_kernel(const float* __restrict__ x, int64_t B,
                                                                      int F, int D,
                                                                      const float* __restrict__ tail,
                                                                      int64_t ld_tail, int T, int tail_first,
                                                                      float* __restrict__ out, int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LD = D + 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* Xs = smem + wave * IMAXF * LD;
    const int i16 = lane & 15, q = lane >> 4;
    const int P = F * (F - 1) / 2;
    const int pofs = tail_first ? T : 0, tofs = tail_first ? 0 : P;
    const int vpr = D / 4, nvec = F * vpr;
    const int qoff = q * (D / 4);
    const int steps = D / 4;
    for (int idx = lane; idx < (IMAXF - F) * LD; idx += 64) Xs[F * LD + idx] = 0.f;

    int lds_off[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = lane + 64 * i;
        const int r = idx / vpr, c4 = idx - r * vpr;
        lds_off[i] = (idx < nvec) ? r * LD + c4 * 4 : -1;
    }
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t b = (int64_t)blockIdx.x * 4 + wave;
    f32x4 xr[NV];
    if (b < B) {
        const f32x4* src = reinterpret_cast<const f32x4*>(x + b * (int64_t)F * D);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lds_off[i] >= 0) xr[i] = src[lane + 64 * i];
    }
    const bool two = F > 16;
    while (b < B) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lds_off[i] >= 0) *reinterpret_cast<f32x4*>(Xs + lds_off[i]) = xr[i];
        __builtin_amdgcn_wave_barrier();
        const int64_t bn = b + stride;
        if (bn < B) {
            const f32x4* src = reinterpret_cast<const f32x4*>(x + bn * (int64_t)F * D);
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (lds_off[i] >= 0) xr[i] = src[lane + 64 * i];
        }
        float tv = 0.f;  // tail element of this lane (T <= 64 fast path; longer tails loop below)
        if (tail && lane < T) tv = tail[b * ld_tail + lane];
        f32x4 acc00 = {0.f, 0.f, 0.f, 0.f}, acc01 = acc00, acc11 = acc00;
        const float* r0 = Xs + i16 * LD + qoff;
        const float* r1 = Xs + (16 + i16) * LD + qoff;
        for (int s = 0; s < steps; s += 4) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(r0 + s);
            if (two) {
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(r1 + s);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc00 = mfma16(a0[j], a0[j], acc00);
                    acc01 = mfma16(a0[j], a1[j], acc01);
                    acc11 = mfma16(a1[j], a1[j], acc11);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc00 = mfma16(a0[j], a0[j], acc00);
            }
        }
        float* orow = out + b * ldo;
        if (STAGED) {
            // coalesced 16-byte stores staged through the first rows of the slab, as in dlrm_fused_fwd_kernel (the launcher
            // checks: 16-byte aligned rows, P + T <= 512, T <= 64)
            __builtin_amdgcn_wave_barrier();
            store_tile(acc00, 0, 0, lane, F, Xs + pofs);
            if (two) {
                store_tile(acc01, 0, 1, lane, F, Xs + pofs);
                store_tile(acc11, 1, 1, lane, F, Xs + pofs);
            }
            if (tail && lane < T) Xs[tofs + lane] = tv;
            __builtin_amdgcn_wave_barrier();
            const int nout = P + T, n4 = nout >> 2, rem = nout & 3;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(Xs + lane * 4);
            f32x4 v1 = {0.f, 0.f, 0.f, 0.f};
            if (nout > 256) v1 = *reinterpret_cast<const f32x4*>(Xs + 256 + lane * 4);
            if (lane < n4) *reinterpret_cast<f32x4*>(orow + lane * 4) = v0;
            if (64 + lane < n4) *reinterpret_cast<f32x4*>(orow + 256 + lane * 4) = v1;
            if (rem) {
                const bool lo = n4 < 64;
                if (lane == (lo ? n4 : n4 - 64)) {
                    const f32x4 v = lo ? v0 : v1;
                    for (int r = 0; r < rem; ++r) orow[n4 * 4 + r] = v[r];
                }
            }
        } else {
            store_tile(acc00, 0, 0, lane, F, orow + pofs);
            if (two) {
                store_tile(acc01, 0, 1, lane, F, orow + pofs);
                store_tile(acc11, 1, 1, lane, F, orow + pofs);
            }
            if (tail) {
                if (lane < T) orow[tofs + lane] = tv;
                for (int t = lane + 64; t < T; t += 64) orow[tofs + t] = tail[b * ld_tail + t];
            }
        }
        __builtin_amdgcn_wave_barrier();
        b = bn;
    }
}

// ---- backward -----------------------------------------------------------------------------------
// dX = (G + G^T) X with G the strict-upper-triangular matrix scattered from dout[:, :P].
// One wavefront per sample: S = G + G^T ([32][34] fp32 in LDS, zero diagonal / padding) is the
// MFMA A operand, X ([32][D+16]) the B operand; 2 x D/16 output tiles of 16x16, contraction over
// the 32 (padded) features in 8 steps of 4.  If tail_slot >= 0 the gradient of the shortcut
// copy, dout[:, tofs:tofs+T], is added to dX[:, tail_slot, :T] in the same pass (pair gradients at dout[:, pofs:pofs+P]).
constexpr int LDS_S = 34;

__global__ __launch_bounds__(256) void dot_interaction_bwd_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ dout, int64_t ldo,
                                                                 int64_t B, int F, int D, float* __restrict__ dx,
                                                                 int tail_slot, int T, int tail_first) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LD = D + 16;
    const int P = F * (F - 1) / 2;
    const int pofs = tail_first ? T : 0, tofs = tail_first ? 0 : P;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned char* pair_i = reinterpret_cast<unsigned char*>(smem);        // [512]
    unsigned char* pair_j = pair_i + 512;                                  // [512]
    float* base = smem + 256;                                              // 1024 B of pair tables
    float* Xs = base + wave * (IMAXF * LD + IMAXF * LDS_S);
    float* Ss = Xs + IMAXF * LD;
    const int i16 = lane & 15, q = lane >> 4;
    const int vpr = D / 4, nvec = F * vpr;

    for (int p = threadIdx.x; p < P; p += 256) {
        // invert p = i(2F-i-1)/2 + (j-i-1) by scanning rows (P <= 496)
        int i = 0, start = 0;
        while (start + (F - 1 - i) <= p) {
            start += F - 1 - i;
            ++i;
        }
        pair_i[p] = (unsigned char)i;
        pair_j[p] = (unsigned char)(i + 1 + (p - start));
    }
    for (int idx = lane; idx < IMAXF * LDS_S; idx += 64) Ss[idx] = 0.f;
    for (int idx = lane; idx < (IMAXF - F) * LD; idx += 64) Xs[F * LD + idx] = 0.f;
    __syncthreads();

    const int64_t nblk = (B + 3) / 4;
    for (int64_t it = blockIdx.x; it < nblk; it += gridDim.x) {
        const int64_t b = it * 4 + wave;
        const bool live = b < B;
        __syncthreads();
        if (live) {
            const f32x4* src = reinterpret_cast<const f32x4*>(x + b * (int64_t)F * D);
            for (int idx = lane; idx < nvec; idx += 64) {
                const int r = idx / vpr, c4 = idx - r * vpr;
                *reinterpret_cast<f32x4*>(Xs + r * LD + c4 * 4) = src[idx];
            }
            const float* g = dout + b * ldo + pofs;
            for (int p = lane; p < P; p += 64) {
                const float v = g[p];
                const int i = pair_i[p], j = pair_j[p];
                Ss[i * LDS_S + j] = v;
                Ss[j * LDS_S + i] = v;
            }
        }
        __syncthreads();
        if (!live) continue;
        float* drow = dx + b * (int64_t)F * D;
        const float* g = dout + b * ldo + tofs;  // shortcut-copy gradient
        const int nti = F > 16 ? 2 : 1;
        for (int tn = 0; tn < D / 16; ++tn) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int kk = 4 * st + q;
                const float bv = Xs[kk * LD + 16 * tn + i16];
                const float a0 = Ss[i16 * LDS_S + kk];
                acc0 = mfma16(a0, bv, acc0);
                if (nti == 2) {
                    const float a1 = Ss[(16 + i16) * LDS_S + kk];
                    acc1 = mfma16(a1, bv, acc1);
                }
            }
            const int d = 16 * tn + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f0 = q * 4 + r;
                if (f0 < F) {
                    float v = acc0[r];
                    if (f0 == tail_slot && d < T) v += g[d];
                    drow[f0 * D + d] = v;
                }
                const int f1 = 16 + f0;
                if (nti == 2 && f1 < F) {
                    float v = acc1[r];
                    if (f1 == tail_slot && d < T) v += g[d];
                    drow[f1 * D + d] = v;
                }
            }
        }
        // tail columns beyond the last multiple of 16 of D (D % 16 != 0)
        for (int d = (D / 16) * 16 + i16; d < D; d += 16) {
            for (int f = q; f < F; f += 4) {
                float v = 0.f;
                for (int kk = 0; kk < F; ++kk) v = fmaf(Ss[f * LDS_S + kk], Xs[kk * LD + d], v);
                if (f == tail_slot && d < T) v += g[d];
                drow[f * D + d] = v;
            }
        }
    }
}

// Pipelined backward (D in {16,32,64,128}, F*D <= 2048): independent wavefronts, next-sample prefetch
// into registers, and the dX tile is transposed through the (dead) X slab so that global stores are
// full 16-byte-per-lane rows instead of 64-byte fragments.
template <int DT, int NV>
__global__ __launch_bounds__(256) void dot_interaction_bwd_pipe_kernel(const float* __restrict__ x,
                                                                      const float* __restrict__ dout, int64_t ldo,
                                                                      int64_t B, int F, float* __restrict__ dx,
                                                                      int tail_slot, int T, int tail_first) {
    constexpr int D = DT * 16;
    constexpr int LD = D + 16;
    constexpr int NP = 8;  // ceil(496 / 64) gradient values per lane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = F * (F - 1) / 2;
    const int pofs = tail_first ? T : 0, tofs = tail_first ? 0 : P;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned char* pair_i = reinterpret_cast<unsigned char*>(smem);
    unsigned char* pair_j = pair_i + 512;
    float* base = smem + 256;
    float* Xs = base + wave * (IMAXF * LD + IMAXF * LDS_S);
    float* Ss = Xs + IMAXF * LD;
    const int i16 = lane & 15, q = lane >> 4;
    constexpr int vpr = D / 4;
    const int nvec = F * vpr;

    for (int p = threadIdx.x; p < P; p += 256) {
        int i = 0, start = 0;
        while (start + (F - 1 - i) <= p) {
            start += F - 1 - i;
            ++i;
        }
        pair_i[p] = (unsigned char)i;
        pair_j[p] = (unsigned char)(i + 1 + (p - start));
    }
    for (int idx = lane; idx < IMAXF * LDS_S; idx += 64) Ss[idx] = 0.f;
    for (int idx = lane; idx < (IMAXF - F) * LD; idx += 64) Xs[F * LD + idx] = 0.f;
    __syncthreads();  // pair table (the only cross-wave dependency)

    int lds_off[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = lane + 64 * i;
        const int r = idx / vpr, c4 = idx - r * vpr;
        lds_off[i] = (idx < nvec) ? r * LD + c4 * 4 : -1;
    }
    int s_off0[NP], s_off1[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = lane + 64 * k;
        if (p < P) {
            s_off0[k] = pair_i[p] * LDS_S + pair_j[p];
            s_off1[k] = pair_j[p] * LDS_S + pair_i[p];
        } else {
            s_off0[k] = s_off1[k] = -1;
        }
    }
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t b = (int64_t)blockIdx.x * 4 + wave;
    f32x4 xr[NV];
    float gr[NP];
    auto prefetch = [&](int64_t bb) {
        const f32x4* src = reinterpret_cast<const f32x4*>(x + bb * (int64_t)F * D);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lds_off[i] >= 0) xr[i] = src[lane + 64 * i];
        const float* g = dout + bb * ldo + pofs;
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if (s_off0[k] >= 0) gr[k] = g[lane + 64 * k];
        // tail gradient for column d = (lane & 15) + 16*tn is fetched lazily below
    };
    if (b < B) prefetch(b);
    const int nti = F > 16 ? 2 : 1;
    while (b < B) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lds_off[i] >= 0) *reinterpret_cast<f32x4*>(Xs + lds_off[i]) = xr[i];
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if (s_off0[k] >= 0) {
                Ss[s_off0[k]] = gr[k];
                Ss[s_off1[k]] = gr[k];
            }
        __builtin_amdgcn_wave_barrier();
        const int64_t bn = b + stride;
        const float* gcur = dout + b * ldo;
        float tgv[DT];
#pragma unroll
        for (int tn = 0; tn < DT; ++tn) {
            const int d = 16 * tn + i16;
            tgv[tn] = (tail_slot >= 0 && d < T) ? gcur[tofs + d] : 0.f;
        }
        if (bn < B) prefetch(bn);
        float a0[8], a1[8];
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            a0[st] = Ss[i16 * LDS_S + 4 * st + q];
            a1[st] = (nti == 2) ? Ss[(16 + i16) * LDS_S + 4 * st + q] : 0.f;
        }
        f32x4 acc0[DT], acc1[DT];
#pragma unroll
        for (int tn = 0; tn < DT; ++tn) {
            acc0[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc1[tn] = acc0[tn];
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const float bv = Xs[(4 * st + q) * LD + 16 * tn + i16];
                acc0[tn] = mfma16(a0[st], bv, acc0[tn]);
                if (nti == 2) acc1[tn] = mfma16(a1[st], bv, acc1[tn]);
            }
        }
        __builtin_amdgcn_wave_barrier();  // all reads of X done: reuse the slab for dX
#pragma unroll
        for (int tn = 0; tn < DT; ++tn) {
            const int d = 16 * tn + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f0 = q * 4 + r, f1 = 16 + f0;
                Xs[f0 * LD + d] = acc0[tn][r] + (f0 == tail_slot ? tgv[tn] : 0.f);
                if (nti == 2) Xs[f1 * LD + d] = acc1[tn][r] + (f1 == tail_slot ? tgv[tn] : 0.f);
            }
        }
        __builtin_amdgcn_wave_barrier();
        f32x4* dst = reinterpret_cast<f32x4*>(dx + b * (int64_t)F * D);
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (lds_off[i] >= 0) dst[lane + 64 * i] = *reinterpret_cast<const f32x4*>(Xs + lds_off[i]);
        __builtin_amdgcn_wave_barrier();
        b = bn;
    }
}

// =================================================================================================
// Fused gather -> interaction (DLRM forward / backward without the stacked [B, F, D] round trip)
// =================================================================================================
// The stacked tensor X[b] = [emb rows in sorted-name order, bottom-MLP row] exists only in LDS: each
// wavefront gathers the F-1 table rows of its sample straight from the tables (16-byte lanes, the same
// access pattern as gather_fwd_kernel) plus the dense row, and feeds the matrix pipe.  HBM bytes per
// sample drop from (gather 13 416 + interaction 8 572) to (F-1)*D*4 + ids + D*4 + out*4 = 8 680 at C2.
// Ids are prefetched two samples ahead and rows one sample ahead, so both dependent HBM round trips
// overlap the MFMA work of the current sample.
struct FusedArgs {
    const float* table[IMAXF];  // by STACK SLOT (sorted feature order); nullptr = the dense (bottom) slot
    const void* ids[IMAXF];
    int64_t rows[IMAXF];
};

// Lane f (< F) of every wavefront owns stack slot f: its table, id column and row count live in 6 VGPRs.  Per sample it
// turns its id into the ADDRESS of the row (0 = out of range / no slot: the row reads as zeros) and the lanes that load
// the row fetch that address with two ds_bpermute -- no per-load pointer / bound arrays (the first version of this
// kernel kept those per lane: 215 VGPRs, 2 waves per SIMD, slower than the unfused pair).
template <typename IdT>
struct SlotLane {
    const float* tab;
    const IdT* ids;
    int64_t rows;
    bool mine, dense;

    __device__ __forceinline__ void init(const FusedArgs& a, int lane, int F) {
        mine = lane < F;
        const int f = mine ? lane : 0;
        tab = a.table[f];
        ids = static_cast<const IdT*>(a.ids[f]);
        rows = a.rows[f];
        dense = mine && tab == nullptr;
    }
    // The id is kept RAW (as loaded) until address() of the next iteration uses it: any arithmetic on it right after the
    // load -- the sign extension of `(int64_t)ids[b]` was enough -- makes the compiler wait for the load at once, and with
    // the in-order vmcnt counter that wait also covered the row loads issued just before it: the "prefetch" of the first
    // version was synchronous (s_waitcnt vmcnt(0) in front of the MFMA section, seen in the ISA).
    __device__ __forceinline__ IdT load_id(int64_t b) const {
        IdT v = 0;
        if (mine && !dense) v = ids[b];
        return v;
    }
    __device__ __forceinline__ uint64_t address(int64_t b, IdT raw_id, const float* __restrict__ dense_src, int64_t ld_dense,
                                                int D) const {
        if (!mine) return 0;
        if (dense) return dense_src ? reinterpret_cast<uint64_t>(dense_src + b * ld_dense) : 0;
        const int64_t id = (int64_t)raw_id;
        return (id >= 0 && id < rows) ? reinterpret_cast<uint64_t>(tab + id * D) : 0;
    }
};

// rows i*RP + sub (sub = lane / vpr) of the sample whose slot addresses are in `addr` (lane f = slot f)
template <int D, int NV>
__device__ __forceinline__ void fetch_rows(uint64_t addr, int F, int lane, f32x4 (&xr)[NV]) {
    constexpr int vpr = D / 4, RP = 64 / vpr;
    const int sub = lane / vpr, c4 = lane - sub * vpr;
    const int alo = (int)(uint32_t)addr, ahi = (int)(uint32_t)(addr >> 32);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (i * RP >= F) break;  // uniform
        const int src = i * RP + sub;  // < 64: NV * RP <= 32 + RP
        const uint32_t lo = (uint32_t)__shfl(alo, src), hi = (uint32_t)__shfl(ahi, src);
        const uint64_t a = ((uint64_t)hi << 32) | lo;
        // the address was rebuilt from two shuffled halves: without the explicit GLOBAL address space the access is a
        // flat_load, which also counts in lgkmcnt -- every LDS wait of the MFMA section then waited for the row loads too
        xr[i] = a ? *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(a + (uint64_t)c4 * 16)
                  : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// contiguous samples per wavefront (the 4-byte id loads of consecutive samples share cache lines)
__device__ __forceinline__ void wave_chunk(int64_t B, int64_t* b0, int64_t* b1) {
    const int64_t nw = (int64_t)gridDim.x * 4;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t per = (B + nw - 1) / nw;
    *b0 = w * per < B ? w * per : B;
    *b1 = *b0 + per < B ? *b0 + per : B;
}

template <typename IdT, int DT, int NV, bool STAGED>
__global__ __launch_bounds__(256) void dlrm_fused_fwd_kernel(const FusedArgs a, const float* __restrict__ dense,
                                                            int64_t ld_dense, int dense_slot, int64_t B, int F,
                                                            int append_dense, int tail_first, float* __restrict__ out,
                                                            int64_t ldo) {
    constexpr int D = DT * 16;
    constexpr int LD = D + 4;
    constexpr int vpr = D / 4, RP = 64 / vpr;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* Xs = smem + wave * IMAXF * LD;
    const int i16 = lane & 15, q = lane >> 4;
    const int P = F * (F - 1) / 2;
    const int qoff = q * (D / 4);
    constexpr int steps = D / 4;
    for (int idx = lane; idx < IMAXF * LD; idx += 64) Xs[idx] = 0.f;  // rows >= F stay zero (never written again)
    const int sub = lane / vpr, c4 = lane - sub * vpr;
    float* xdst = Xs + sub * LD + c4 * 4;  // row i*RP + sub lands at xdst + i*RP*LD
    SlotLane<IdT> sl;
    sl.init(a, lane, F);
    int64_t b, b1;
    wave_chunk(B, &b, &b1);
    f32x4 xr[NV];
    IdT idn = 0;
    if (b < b1) {
        fetch_rows<D, NV>(sl.address(b, sl.load_id(b), dense, ld_dense, D), F, lane, xr);
        if (b + 1 < b1) idn = sl.load_id(b + 1);
    }
    const bool two = F > 16;
    // The outputs of sample b are stored one iteration LATE, right behind the prefetch loads of iteration b + 1: gfx9-family
    // stores count in vmcnt like loads, so the s_waitcnt vmcnt(0) in front of the LDS writes at the top of an iteration also
    // waited for the stores the previous iteration had issued at its very end; issued together with the loads they have a
    // whole MFMA phase to complete.
    // They go out as TWO coalesced 16-byte stores per lane (the P + T <= 416 floats of a sample are contiguous): straight from
    // the MFMA C layout a sample took 12 scattered dword stores of four <= 64-byte runs each -- ~80 partial-line write requests
    // at the L2 per sample, 60 % of all L2 requests of the kernel, with the wavefronts stalled on instruction issue 71 % of
    // their cycles (SQ_WAIT_INST_ANY) and the L1 stalled on pending requests 72 % of its busy cycles.  The values pass through
    // the first rows of the wavefront's own X slab, which is dead between the last fragment read of a sample and the row
    // writes of the next one (LDS operations of one wavefront execute in order).
    // STAGED is the launcher's choice: 16-byte aligned output rows of at most 512 floats.  (The staging area [0, nout) stays inside
    // rows < F, which the next sample's row writes restore: F (F - 1) / 2 + D <= F (D + 4) for every F <= 32, D >= 16.)
    const int Tc = (append_dense && dense_slot >= 0) ? D : 0;  // shortcut columns of one output row
    const int nout = P + Tc;                                  // floats of one output row
    const int pofs = tail_first ? Tc : 0, tofs = tail_first ? 0 : P;
    constexpr bool staged = STAGED;
    f32x4 ov0 = {0.f, 0.f, 0.f, 0.f}, ov1 = ov0;
    f32x4 o00 = ov0, o01 = ov0, o11 = ov0;
    float odense = 0.f;
    float* oprev = nullptr;
    auto flush = [&]() {
        if (!oprev) return;  // wave-uniform
        if (staged) {
            // whole float4s inside the row: the tail of the last one (< 4 floats past P + T) lands in the row's ld padding or is
            // cut to scalar stores when the row has none
            const int n4 = nout >> 2, rem = nout & 3;
            if (lane < n4) *reinterpret_cast<f32x4*>(oprev + lane * 4) = ov0;
            if (64 + lane < n4) *reinterpret_cast<f32x4*>(oprev + 256 + lane * 4) = ov1;
            if (rem) {
                const bool lo = n4 < 64;
                if (lane == (lo ? n4 : n4 - 64)) {
                    const f32x4 v = lo ? ov0 : ov1;
                    for (int r = 0; r < rem; ++r) oprev[n4 * 4 + r] = v[r];
                }
            }
            return;
        }
        store_tile(o00, 0, 0, lane, F, oprev + pofs);
        if (two) {
            store_tile(o01, 0, 1, lane, F, oprev + pofs);
            store_tile(o11, 1, 1, lane, F, oprev + pofs);
        }
        if (append_dense && dense_slot >= 0 && lane < D) oprev[tofs + lane] = odense;
    };
    while (b < b1) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (i * RP < F && i * RP + sub < IMAXF) *reinterpret_cast<f32x4*>(xdst + i * RP * LD) = xr[i];
        __builtin_amdgcn_wave_barrier();
        if (b + 1 < b1) {
            // order matters for the waits the compiler inserts: the id of b + 1 (loaded one iteration ago) is consumed, the id
            // load of b + 2 is issued into the same register, THEN the row loads -- nothing touches a loaded register between
            // the issue of this sample's loads and the MFMA section
            const uint64_t addr = sl.address(b + 1, idn, dense, ld_dense, D);
            if (b + 2 < b1) idn = sl.load_id(b + 2);
            fetch_rows<D, NV>(addr, F, lane, xr);
        }
        // the stores stay HERE: without the two fences the optimizer moves them (nothing aliases `out`) to the top of the
        // loop and, after rotation, back to the end of the previous iteration -- exactly the placement this avoids
        asm volatile("" ::: "memory");
        flush();  // sample b - 1
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc00 = {0.f, 0.f, 0.f, 0.f}, acc01 = acc00, acc11 = acc00;
        const float* r0 = Xs + i16 * LD + qoff;
        const float* r1 = Xs + (16 + i16) * LD + qoff;
#pragma unroll
        for (int s = 0; s < steps; s += 4) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(r0 + s);
            if (two) {
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(r1 + s);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc00 = mfma16(a0[j], a0[j], acc00);
                    acc01 = mfma16(a0[j], a1[j], acc01);
                    acc11 = mfma16(a1[j], a1[j], acc11);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc00 = mfma16(a0[j], a0[j], acc00);
            }
        }
        oprev = out + b * ldo;
        if (staged) {
            float dlo = 0.f, dhi = 0.f;
            if (append_dense && dense_slot >= 0) {
                if (lane < D) dlo = Xs[dense_slot * LD + lane];
                if (D > 64) dhi = Xs[dense_slot * LD + 64 + lane];
            }
            __builtin_amdgcn_wave_barrier();  // the dense row is in registers before the staging area is overwritten
            store_tile(acc00, 0, 0, lane, F, Xs + pofs);
            if (two) {
                store_tile(acc01, 0, 1, lane, F, Xs + pofs);
                store_tile(acc11, 1, 1, lane, F, Xs + pofs);
            }
            if (append_dense && dense_slot >= 0) {
                if (lane < D) Xs[tofs + lane] = dlo;
                if (D > 64) Xs[tofs + 64 + lane] = dhi;
            }
            __builtin_amdgcn_wave_barrier();
            ov0 = *reinterpret_cast<const f32x4*>(Xs + lane * 4);
            if (nout > 256) ov1 = *reinterpret_cast<const f32x4*>(Xs + 256 + lane * 4);  // (reads past nout: stale slab floats, never stored)
        } else {
            o00 = acc00;
            o01 = acc01;
            o11 = acc11;
            if (append_dense && dense_slot >= 0) {
                static_assert(D <= 128, "dense copy: two values per lane at most");
                if (D <= 64) {
                    if (lane < D) odense = Xs[dense_slot * LD + lane];
                } else {  // D = 128: the second half goes out at once (rare configuration)
                    odense = Xs[dense_slot * LD + lane];
                    oprev[tofs + 64 + lane] = Xs[dense_slot * LD + 64 + lane];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        ++b;
    }
    flush();
}

template <typename IdT, int DT, int NV>
__global__ __launch_bounds__(256) void dlrm_fused_bwd_kernel(const FusedArgs a, const float* __restrict__ dense,
                                                            int64_t ld_dense, const float* __restrict__ dout,
                                                            int64_t ldo, int64_t B, int F, float* __restrict__ dx,
                                                            int tail_slot, int T, int tail_first) {
    constexpr int D = DT * 16;
    constexpr int LD = D + 16;
    constexpr int NP = 8;
    constexpr int vpr = D / 4, RP = 64 / vpr;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = F * (F - 1) / 2;
    const int pofs = tail_first ? T : 0, tofs = tail_first ? 0 : P;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned char* pair_i = reinterpret_cast<unsigned char*>(smem);
    unsigned char* pair_j = pair_i + 512;
    float* base = smem + 256;
    // The slab of a wavefront holds F + 1 rows of X and of the gradient matrix S, not IMAXF: rows F .. 31 of the 32-row MFMA
    // operands are all zero, so every read of a row >= F is redirected to ONE zero row (row F, never written).  The LDS saved
    // decides how many workgroups a CU holds: 27 features, D = 64: 52 instead of 59 KB -> 3 instead of 2 (130 + 20 registers
    // allow 3 wavefronts per SIMD).
    const int R = F + 1;
    const int slab = R * LD + ((R * LDS_S + 3) & ~3);  // multiple of 4 floats: the next wavefront's rows stay 16-byte aligned
    float* Xs = base + wave * slab;
    float* Ss = Xs + R * LD;
    const int i16 = lane & 15, q = lane >> 4;
    for (int p = threadIdx.x; p < P; p += 256) {
        int i = 0, start = 0;
        while (start + (F - 1 - i) <= p) {
            start += F - 1 - i;
            ++i;
        }
        pair_i[p] = (unsigned char)i;
        pair_j[p] = (unsigned char)(i + 1 + (p - start));
    }
    for (int idx = lane; idx < R * LDS_S; idx += 64) Ss[idx] = 0.f;
    for (int idx = lane; idx < R * LD; idx += 64) Xs[idx] = 0.f;
    __syncthreads();
    const int sub = lane / vpr, c4 = lane - sub * vpr;
    float* xdst = Xs + sub * LD + c4 * 4;
    SlotLane<IdT> sl;
    sl.init(a, lane, F);
    int s_off[NP];  // both scatter offsets of pair p = lane + 64 k, 16 bits each (-1: no pair)
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = lane + 64 * k;
        s_off[k] = (p < P) ? ((pair_i[p] * LDS_S + pair_j[p]) | ((pair_j[p] * LDS_S + pair_i[p]) << 16)) : -1;
    }
    int64_t b, b1;
    wave_chunk(B, &b, &b1);
    f32x4 xr[NV];
    float gr[NP];
    IdT idn = 0;
    float tgn[DT];  // tail gradient (the appended dense copy) of the NEXT sample: fetched with the rest of its prefetch
#pragma unroll
    for (int tn = 0; tn < DT; ++tn) tgn[tn] = 0.f;
    auto load_g = [&](int64_t bb) {
        const float* gp = dout + bb * ldo;
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if (s_off[k] >= 0) gr[k] = gp[pofs + lane + 64 * k];
#pragma unroll
        for (int tn = 0; tn < DT; ++tn) {
            const int d = 16 * tn + i16;
            if (tail_slot >= 0 && d < T) tgn[tn] = gp[tofs + d];
        }
    };
    if (b < b1) {
        fetch_rows<D, NV>(sl.address(b, sl.load_id(b), dense, ld_dense, D), F, lane, xr);
        load_g(b);
        if (b + 1 < b1) idn = sl.load_id(b + 1);
    }
    const int nti = F > 16 ? 2 : 1;
    const int srow0 = (i16 < F ? i16 : F) * LDS_S, srow1 = (16 + i16 < F ? 16 + i16 : F) * LDS_S;
    int xrow[8];  // operand row 4 st + q of X (the zero row for rows >= F)
#pragma unroll
    for (int st = 0; st < 8; ++st) xrow[st] = (4 * st + q < F ? 4 * st + q : F) * LD;
    while (b < b1) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (i * RP + sub < F) *reinterpret_cast<f32x4*>(xdst + i * RP * LD) = xr[i];
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if (s_off[k] >= 0) {
                Ss[s_off[k] & 0xffff] = gr[k];
                Ss[s_off[k] >> 16] = gr[k];
            }
        __builtin_amdgcn_wave_barrier();
        float tgv[DT];
#pragma unroll
        for (int tn = 0; tn < DT; ++tn) tgv[tn] = tgn[tn];
        if (b + 1 < b1) {
            const uint64_t addr = sl.address(b + 1, idn, dense, ld_dense, D);  // same order as in the forward kernel
            if (b + 2 < b1) idn = sl.load_id(b + 2);
            fetch_rows<D, NV>(addr, F, lane, xr);
            load_g(b + 1);
        }
        float a0[8], a1[8];
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            a0[st] = Ss[srow0 + 4 * st + q];
            a1[st] = (nti == 2) ? Ss[srow1 + 4 * st + q] : 0.f;
        }
        f32x4 acc0[DT], acc1[DT];
#pragma unroll
        for (int tn = 0; tn < DT; ++tn) {
            acc0[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc1[tn] = acc0[tn];
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const float bv = Xs[xrow[st] + 16 * tn + i16];
                acc0[tn] = mfma16(a0[st], bv, acc0[tn]);
                if (nti == 2) acc1[tn] = mfma16(a1[st], bv, acc1[tn]);
            }
        }
        __builtin_amdgcn_wave_barrier();  // all reads of X done: the slab is reused for dX
#pragma unroll
        for (int tn = 0; tn < DT; ++tn) {
            const int d = 16 * tn + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f0 = q * 4 + r, f1 = 16 + f0;
                if (f0 < F) Xs[f0 * LD + d] = acc0[tn][r] + (f0 == tail_slot ? tgv[tn] : 0.f);
                if (nti == 2 && f1 < F) Xs[f1 * LD + d] = acc1[tn][r] + (f1 == tail_slot ? tgv[tn] : 0.f);
            }
        }
        __builtin_amdgcn_wave_barrier();
        f32x4* dst = reinterpret_cast<f32x4*>(dx + b * (int64_t)F * D) + c4;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (i * RP + sub < F) dst[(i * RP + sub) * vpr] = *reinterpret_cast<const f32x4*>(xdst + i * RP * LD);
        __builtin_amdgcn_wave_barrier();
        ++b;
    }
}

}  // namespace

extern "C" {

int32_t mh_dot_interaction_fwd(const float* x, int64_t B, int32_t F, int32_t D, const float* tail,
                               int64_t ld_tail, int32_t T, int32_t tail_first, float* out, int64_t ldo,
                               mh_stream_t stream) {
    MH_REQUIRE(x && out, "mh_dot_interaction_fwd: null argument");
    MH_REQUIRE(F >= 2 && F <= IMAXF, "mh_dot_interaction_fwd: F=%d outside [2,%d]", F, IMAXF);
    MH_REQUIRE(D >= 4 && D % 4 == 0 && D <= 256, "mh_dot_interaction_fwd: D=%d must be a multiple of 4 in [4,256]", D);
    MH_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "mh_dot_interaction_fwd: x must be 16-byte aligned");
    const int P = F * (F - 1) / 2;
    if (!tail) T = 0;
    tail_first = tail_first ? 1 : 0;
    MH_REQUIRE(T >= 0 && ldo >= P + T, "mh_dot_interaction_fwd: ldo=%lld < %d", (long long)ldo, P + T);
    MH_REQUIRE(!tail || ld_tail >= T, "mh_dot_interaction_fwd: ld_tail < T");
    if (B <= 0) return MH_OK;
    const size_t lds = (size_t)4 * IMAXF * (D + 4) * sizeof(float);
    const int64_t want = mh_ceil_div(B, 4);
    const int64_t cap = (int64_t)mh_num_cus() * 8;
    dim3 grid((unsigned)(want < cap ? want : cap));
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dot_interaction_fwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            mh_set_error("mh_dot_interaction_fwd: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
            return MH_ERR_LAUNCH;
        }
    }
    if (D % 16 == 0 && F * (D / 4) <= 64 * 8) {
        // (staging area [0, P + T) inside rows < F of the [32][D + 4] slab: F (F - 1) / 2 + T <= F (D + 4) holds for T <= D + 4)
        const bool staged = ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && P + T <= 512 && T <= 64 &&
                            P + T <= F * (D + 4);
        auto kern = staged ? dot_interaction_fwd_pipe_kernel<8, true> : dot_interaction_fwd_pipe_kernel<8, false>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        MH_LAUNCH(kern, grid, dim3(256), lds, mh_stream(stream), x, B, F, D, tail, ld_tail, T, tail_first, out, ldo);
    } else {
        MH_LAUNCH(dot_interaction_fwd_kernel, grid, dim3(256), lds, mh_stream(stream), x, B, F, D,
                           tail, ld_tail, T, tail_first, out, ldo);
    }
    MH_CHECK_LAUNCH("mh_dot_interaction_fwd");
    return MH_OK;
}

int32_t mh_dot_interaction_bwd(const float* x, const float* dout, int64_t ldo, int64_t B, int32_t F,
                               int32_t D, float* dx, int32_t tail_slot, int32_t T, int32_t tail_first,
                               mh_stream_t stream) {
    MH_REQUIRE(x && dout && dx, "mh_dot_interaction_bwd: null argument");
    MH_REQUIRE(F >= 2 && F <= IMAXF, "mh_dot_interaction_bwd: F=%d outside [2,%d]", F, IMAXF);
    MH_REQUIRE(D >= 4 && D % 4 == 0 && D <= 256, "mh_dot_interaction_bwd: D=%d must be a multiple of 4 in [4,256]", D);
    MH_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "mh_dot_interaction_bwd: x must be 16-byte aligned");
    const int P = F * (F - 1) / 2;
    if (tail_slot < 0) T = 0;
    tail_first = tail_first ? 1 : 0;
    MH_REQUIRE(tail_slot < F && T >= 0 && T <= D && ldo >= P + T, "mh_dot_interaction_bwd: bad tail_slot/T/ldo");
    if (B <= 0) return MH_OK;
    const size_t lds = 1024 + (size_t)4 * (IMAXF * (D + 16) + IMAXF * LDS_S) * sizeof(float);
    const int64_t want = mh_ceil_div(B, 4);
    const int64_t cap = (int64_t)mh_num_cus() * 8;
    dim3 grid((unsigned)(want < cap ? want : cap));
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dot_interaction_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            mh_set_error("mh_dot_interaction_bwd: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
            return MH_ERR_LAUNCH;
        }
    }
    const bool pipe = (D == 16 || D == 32 || D == 64 || D == 128) && F * (D / 4) <= 64 * 8;
    if (pipe) {
        hipStream_t s_ = mh_stream(stream);
#define MH_LAUNCH_BWD_PIPE(DT)                                                                                   \
    {                                                                                                            \
        auto kern = dot_interaction_bwd_pipe_kernel<DT, 8>;                                                      \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        MH_LAUNCH(kern, grid, dim3(256), lds, s_, x, dout, ldo, B, F, dx, tail_slot, T, tail_first);    \
    }
        if (D == 16) MH_LAUNCH_BWD_PIPE(1)
        else if (D == 32) MH_LAUNCH_BWD_PIPE(2)
        else if (D == 64) MH_LAUNCH_BWD_PIPE(4)
        else MH_LAUNCH_BWD_PIPE(8)
#undef MH_LAUNCH_BWD_PIPE
    } else {
        MH_LAUNCH(dot_interaction_bwd_kernel, grid, dim3(256), lds, mh_stream(stream), x, dout, ldo, B, F, D,
                           dx, tail_slot, T, tail_first);
    }
    MH_CHECK_LAUNCH("mh_dot_interaction_bwd");
    return MH_OK;
}

static int32_t fill_fused_args(FusedArgs* a, const float* const* slot_tables, const int64_t* slot_rows,
                               const void* const* slot_ids, int32_t F, int32_t* dense_slot, const char* who) {
    *dense_slot = -1;
    for (int f = 0; f < F; ++f) {
        a->table[f] = slot_tables[f];
        a->ids[f] = slot_ids ? slot_ids[f] : nullptr;
        a->rows[f] = slot_rows ? slot_rows[f] : 0;
        if (!slot_tables[f]) {
            if (*dense_slot >= 0) {
                mh_set_error("%s: at most one dense (NULL-table) slot", who);
                return MH_ERR_INVALID_ARGUMENT;
            }
            *dense_slot = f;
        } else if (!a->ids[f]) {
            mh_set_error("%s: slot %d has a table but no ids", who, f);
            return MH_ERR_INVALID_ARGUMENT;
        }
    }
    for (int f = F; f < IMAXF; ++f) {
        a->table[f] = nullptr;
        a->ids[f] = nullptr;
        a->rows[f] = 0;
    }
    return MH_OK;
}

// one resident set of workgroups (every wavefront walks one contiguous run of samples): CUs x what LDS and a <= 128 VGPR
// budget allow
static dim3 fused_grid(int64_t B, size_t lds, int max_occ = 4) {
    // LDS is handed out in granules of 1280 bytes (160 KB / 128; measured: a request of 53 952 bytes -- 27 tables + the dense slot at
    // D = 64 -- is resident TWICE per CU, not the three times 163 840 / 53 952 promises and the occupancy query reports; a grid sized for
    // three then ran 1.5 rounds: the 278 us of `dlrm_fused_bwd` at F = 28 against 219 us with this rounding, profiles/r5_notes.md).
    // MERLIN_HIP_FUSED_LDS_GRANULE overrides (1 = the old arithmetic).
    static int lds_gran = -1;
    if (lds_gran < 0) {
        const char* e = MH_LAB_ENV("MERLIN_HIP_FUSED_LDS_GRANULE");
        lds_gran = e ? atoi(e) : 1280;
        if (lds_gran < 1) lds_gran = 1;
    }
    lds = (lds + lds_gran - 1) / lds_gran * lds_gran;
    int occ = (int)((160 * 1024) / lds);
    if (occ > max_occ) occ = max_occ;
    if (occ < 1) occ = 1;
    const int64_t want = mh_ceil_div(B, 4);
    const int64_t cap = (int64_t)mh_num_cus() * occ;
    return dim3((unsigned)(want < cap ? want : cap));
}

int32_t mh_dlrm_interaction_fused_fwd(const float* const* slot_tables, const int64_t* slot_rows,
                                      const void* const* slot_ids, int32_t ids_dtype, const float* dense,
                                      int64_t ld_dense, int64_t B, int32_t F, int32_t D, int32_t append_dense,
                                      int32_t tail_first, float* out, int64_t ldo, mh_stream_t stream) {
    MH_REQUIRE(slot_tables && slot_rows && slot_ids && out, "mh_dlrm_interaction_fused_fwd: null argument");
    MH_REQUIRE(F >= 2 && F <= IMAXF, "mh_dlrm_interaction_fused_fwd: F=%d outside [2,%d]", F, IMAXF);
    MH_REQUIRE((D == 16 || D == 32 || D == 64 || D == 128) && F * (D / 4) <= 64 * 8,
               "mh_dlrm_interaction_fused_fwd: needs D in {16,32,64,128} and F*D <= 2048 (got F=%d D=%d)", F, D);
    MH_REQUIRE(ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_dlrm_interaction_fused_fwd: bad ids_dtype");
    FusedArgs a;
    int32_t dense_slot;
    int32_t st = fill_fused_args(&a, slot_tables, slot_rows, slot_ids, F, &dense_slot, "mh_dlrm_interaction_fused_fwd");
    if (st != MH_OK) return st;
    MH_REQUIRE(dense_slot < 0 || (dense && ld_dense >= D && ld_dense % 4 == 0 && (reinterpret_cast<uintptr_t>(dense) & 15) == 0),
               "mh_dlrm_interaction_fused_fwd: dense slot needs a 16-byte aligned [B, D] source");
    const int P = F * (F - 1) / 2;
    const int T = (append_dense && dense_slot >= 0) ? D : 0;
    MH_REQUIRE(ldo >= P + T, "mh_dlrm_interaction_fused_fwd: ldo too small");
    if (B <= 0) return MH_OK;
    const size_t lds = (size_t)4 * IMAXF * (D + 4) * sizeof(float);
    const dim3 grid = fused_grid(B, lds);
    hipStream_t s_ = mh_stream(stream);
    // coalesced 16-byte output stores through the LDS slab (see the kernel) when the rows allow them
    const bool staged = ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && P + T <= 512;
#define MH_LAUNCH_FUSED_FWD(IDT, DT)                                                                             \
    {                                                                                                            \
        auto kern = staged ? dlrm_fused_fwd_kernel<IDT, DT, 8, true> : dlrm_fused_fwd_kernel<IDT, DT, 8, false>; \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        MH_LAUNCH(kern, grid, dim3(256), lds, s_, a, dense, ld_dense, dense_slot, B, F, append_dense, tail_first ? 1 : 0, out, ldo); \
    }
    if (ids_dtype == MH_I32) {
        if (D == 16) MH_LAUNCH_FUSED_FWD(int32_t, 1)
        else if (D == 32) MH_LAUNCH_FUSED_FWD(int32_t, 2)
        else if (D == 64) MH_LAUNCH_FUSED_FWD(int32_t, 4)
        else MH_LAUNCH_FUSED_FWD(int32_t, 8)
    } else {
        if (D == 16) MH_LAUNCH_FUSED_FWD(int64_t, 1)
        else if (D == 32) MH_LAUNCH_FUSED_FWD(int64_t, 2)
        else if (D == 64) MH_LAUNCH_FUSED_FWD(int64_t, 4)
        else MH_LAUNCH_FUSED_FWD(int64_t, 8)
    }
#undef MH_LAUNCH_FUSED_FWD
    MH_CHECK_LAUNCH("mh_dlrm_interaction_fused_fwd");
    return MH_OK;
}

int32_t mh_dlrm_interaction_fused_bwd(const float* const* slot_tables, const int64_t* slot_rows,
                                      const void* const* slot_ids, int32_t ids_dtype, const float* dense,
                                      int64_t ld_dense, const float* dout, int64_t ldo, int64_t B, int32_t F,
                                      int32_t D, int32_t tail_to_dense, int32_t tail_first, float* dx,
                                      mh_stream_t stream) {
    MH_REQUIRE(slot_tables && slot_rows && slot_ids && dout && dx, "mh_dlrm_interaction_fused_bwd: null argument");
    MH_REQUIRE(F >= 2 && F <= IMAXF, "mh_dlrm_interaction_fused_bwd: F=%d outside [2,%d]", F, IMAXF);
    MH_REQUIRE((D == 16 || D == 32 || D == 64 || D == 128) && F * (D / 4) <= 64 * 8,
               "mh_dlrm_interaction_fused_bwd: needs D in {16,32,64,128} and F*D <= 2048 (got F=%d D=%d)", F, D);
    MH_REQUIRE(ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_dlrm_interaction_fused_bwd: bad ids_dtype");
    MH_REQUIRE((reinterpret_cast<uintptr_t>(dx) & 15) == 0, "mh_dlrm_interaction_fused_bwd: dx must be 16-byte aligned");
    FusedArgs a;
    int32_t dense_slot;
    int32_t st = fill_fused_args(&a, slot_tables, slot_rows, slot_ids, F, &dense_slot, "mh_dlrm_interaction_fused_bwd");
    if (st != MH_OK) return st;
    MH_REQUIRE(dense_slot < 0 || (dense && ld_dense >= D && ld_dense % 4 == 0 && (reinterpret_cast<uintptr_t>(dense) & 15) == 0),
               "mh_dlrm_interaction_fused_bwd: dense slot needs a 16-byte aligned [B, D] source");
    const int P = F * (F - 1) / 2;
    const int tail_slot = (tail_to_dense && dense_slot >= 0) ? dense_slot : -1;
    const int T = tail_slot >= 0 ? D : 0;
    MH_REQUIRE(ldo >= P + T, "mh_dlrm_interaction_fused_bwd: ldo too small");
    if (B <= 0) return MH_OK;
    const size_t lds = 1024 + (size_t)4 * ((F + 1) * (D + 16) + (((F + 1) * LDS_S + 3) & ~3)) * sizeof(float);
    const dim3 grid = fused_grid(B, lds, D == 16 ? 4 : (D == 128 ? 2 : 3));  // what the register count of each instantiation allows
    hipStream_t s_ = mh_stream(stream);
#define MH_LAUNCH_FUSED_BWD(IDT, DT)                                                                             \
    {                                                                                                            \
        auto kern = dlrm_fused_bwd_kernel<IDT, DT, 8>;                                                           \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        MH_LAUNCH(kern, grid, dim3(256), lds, s_, a, dense, ld_dense, dout, ldo, B, F, dx, tail_slot, T, tail_first ? 1 : 0); \
    }
    if (ids_dtype == MH_I32) {
        if (D == 16) MH_LAUNCH_FUSED_BWD(int32_t, 1)
        else if (D == 32) MH_LAUNCH_FUSED_BWD(int32_t, 2)
        else if (D == 64) MH_LAUNCH_FUSED_BWD(int32_t, 4)
        else MH_LAUNCH_FUSED_BWD(int32_t, 8)
    } else {
        if (D == 16) MH_LAUNCH_FUSED_BWD(int64_t, 1)
        else if (D == 32) MH_LAUNCH_FUSED_BWD(int64_t, 2)
        else if (D == 64) MH_LAUNCH_FUSED_BWD(int64_t, 4)
        else MH_LAUNCH_FUSED_BWD(int64_t, 8)
    }
#undef MH_LAUNCH_FUSED_BWD
    MH_CHECK_LAUNCH("mh_dlrm_interaction_fused_bwd");
    return MH_OK;
}

}  // extern "C"
