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
//
// The hot loop is branch-free: per-thread row pointers are computed ONCE (rows past the end are
// CLAMPED to the last valid row -- they only feed output rows/columns that are never stored), full
// k-tiles use unconditional 16-byte loads, and only the final partial k-tile (K % 32 != 0) or an
// unaligned operand takes the guarded path (zeros beyond K are required there: they enter sums).
template <int ROWS, int NTH = NT>
struct KMajorTile {
    static constexpr int NV = (ROWS * (BK / 4) + NTH - 1) / NTH;
    f32x4 regs[NV];
    const float* ptr[NV];

    __device__ __forceinline__ void init(const float* __restrict__ src, int64_t ld, int64_t row0, int64_t nrows) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * NTH;
            const int r = idx >> 3, c4 = idx & 7;
            int64_t row = row0 + r;
            if (row > nrows - 1) row = nrows - 1;
            if (row < 0) row = 0;
            ptr[i] = src + row * ld + c4 * 4;
        }
    }
    // full tile, 16-byte aligned rows
    __device__ __forceinline__ void load_fast(int k0) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if ((ROWS * (BK / 4)) % NTH == 0 || (int)(threadIdx.x + i * NTH) < ROWS * (BK / 4))
                regs[i] = *reinterpret_cast<const f32x4*>(ptr[i] + k0);
    }
    // partial tile and/or unaligned rows: element-wise, zero beyond K.  Branch-free: every lane loads from a
    // valid address (its row start when the element is out of range) and the select happens afterwards, so the
    // four scalar loads of a group are in flight together instead of one dependent round trip each.
    __device__ __forceinline__ void load_guarded(int k0, int K) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * NTH;
            const int c4 = idx & 7;
            const int k = k0 + c4 * 4;
            const bool in = idx < ROWS * (BK / 4);
            const float* row = ptr[i] - c4 * 4;  // element 0 of the (clamped) row: always readable
            const bool o0 = in && k < K, o1 = in && k + 1 < K, o2 = in && k + 2 < K, o3 = in && k + 3 < K;
            const float x0 = *(o0 ? row + k : row);
            const float x1 = *(o1 ? row + k + 1 : row);
            const float x2 = *(o2 ? row + k + 2 : row);
            const float x3 = *(o3 ? row + k + 3 : row);
            regs[i] = f32x4{o0 ? x0 : 0.f, o1 ? x1 : 0.f, o2 ? x2 : 0.f, o3 ? x3 : 0.f};
        }
    }
    __device__ __forceinline__ void load(int k0, int K, bool vec_ok) {
        if (vec_ok && k0 + BK <= K) load_fast(k0); else load_guarded(k0, K);
    }
    __device__ __forceinline__ void store(float* __restrict__ lds) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * NTH;
            const int r = idx >> 3, c4 = idx & 7;
            if (r < ROWS) {
                float* d = lds + r * LDK + c4 * 2;
                *reinterpret_cast<float2*>(d) = make_float2(regs[i].x, regs[i].z);       // even k -> slot 0
                *reinterpret_cast<float2*>(d + 16) = make_float2(regs[i].y, regs[i].w);  // odd k  -> slot 1
            }
        }
    }
};

// Staging of a [BK x COLS] tile of a row-major W[K, N] (n contiguous): the "NN" B operand, and both
// operands of the split-M "TN" product.  Column groups past N are clamped (never stored), rows past K
// (the contraction) must read as zero and take the guarded path.
template <int COLS, int NTH = NT>
struct NMajorTile {
    static constexpr int VPR = COLS / 4;
    static constexpr int NV = (BK * VPR + NTH - 1) / NTH;
    f32x4 regs[NV];
    const float* ptr[NV];  // W + r*ldw + clamped column
    int col[NV];

    // Column groups are clamped to the group that holds column N - 1 (rounded up to the 16-byte group, within the leading
    // dimension): in vector mode (16-byte aligned rows, ldw % 4 == 0) the group straddling N reads the row's padding (it
    // only feeds outputs that are never stored), so an ld-padded operand such as x[M, 415] with ld 416 takes the 16-byte
    // path for every group.  Every row -- the last included -- must therefore own ceil(N / 4) * 4 floats.
    __device__ __forceinline__ void init(const float* __restrict__ W, int64_t ldw, int n0, int N) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * NTH;
            const int r = idx / VPR, c4 = idx - r * VPR;
            int n = n0 + c4 * 4;
            col[i] = n;
            int last = (ldw >= 4) ? (int)((ldw - 4) / 4) * 4 : 0;
            const int need = (N + 3) / 4 * 4 - 4;  // the group that holds column N - 1: nothing beyond it is ever needed
            if (need >= 0 && need < last) last = need;  // (a column-offset view must not be read past its rows)
            if (n > last) n = last;
            ptr[i] = W + (int64_t)r * ldw + n;
        }
    }
    __device__ __forceinline__ void load_fast(int k0, int64_t ldw) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if ((BK * VPR) % NTH == 0 || (int)(threadIdx.x + i * NTH) < BK * VPR)
                regs[i] = *reinterpret_cast<const f32x4*>(ptr[i] + (int64_t)k0 * ldw);
    }
    // rows past K (the contraction) and columns past N read as zero; branch-free like KMajorTile::load_guarded
    __device__ __forceinline__ void load_guarded(const float* __restrict__ W, int64_t ldw, int k0, int K, int N) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * NTH;
            const int r = idx / VPR;
            const int k = k0 + r, n = col[i];
            const bool in = idx < BK * VPR && k < K;
            const float* p = W + (int64_t)k * ldw + n;
            const bool o0 = in && n < N, o1 = in && n + 1 < N, o2 = in && n + 2 < N, o3 = in && n + 3 < N;
            const float x0 = *(o0 ? p : W);
            const float x1 = *(o1 ? p + 1 : W);
            const float x2 = *(o2 ? p + 2 : W);
            const float x3 = *(o3 ? p + 3 : W);
            regs[i] = f32x4{o0 ? x0 : 0.f, o1 ? x1 : 0.f, o2 ? x2 : 0.f, o3 ? x3 : 0.f};
        }
    }
    // vec_ok: 16-byte aligned rows AND ldw % 4 == 0
    __device__ __forceinline__ void load(const float* __restrict__ W, int64_t ldw, int k0, int K, int N, bool vec_ok) {
        if (vec_ok && k0 + BK <= K) load_fast(k0, ldw); else load_guarded(W, ldw, k0, K, N);
    }
    __device__ __forceinline__ void store(float* __restrict__ lds) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * NTH;
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

// One k-tile = four groups of 4 MFMA steps; each group's fragments are fetched with ds_read_b128
// right before use (hipcc emits counted lgkmcnt waits; an explicit register double-buffer of the
// fragments was measured to be neutral and costs 24 VGPRs -- profiles/r1_notes.md).
template <int TM, int TN, bool B_NT>
__device__ __forceinline__ void mma_ktile(const float* __restrict__ As, int a_row0,
                                          const float* __restrict__ Bs, int b_row0, int ldb,
                                          f32x16 (&acc)[TM][TN]) {
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        Frag<TM, TN, B_NT> f;
        f.load(As, a_row0, Bs, b_row0, ldb, g, l31, h);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(f.a[tm][j], f.b[tn][j], acc[tm][tn]);
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

// ---- epilogue shared by every GEMM kernel (both cores) ----------------------------------------------------------------
// out = x_act'(maskx) * act( [x0 * (acc + bias) + xres]  or  (acc + bias) ) [+ addend],  optionally p_out = acc + bias.
struct EpiArgs {
    const float* bias;  // [N] or NULL
    int act;            // MH_ACT_*
    const float* x0;    // DCN-v2 cross epilogue (blocks/cross.py:188-202): out = x0 * (.) + xres, both [M, ld_x0]
    const float* xres;
    int64_t ld_x0;
    float* p_out;  // cross layer under a gradient tape: also store p = x W + b ([M, ldp]) -- the backward needs it (dx0 = dout * p)
    int64_t ldp;
    const float* maskx;  // dX: the producer's activation derivative folded in (x_act of mh_linear_bias_act_bwd)
    int64_t ldm;
    int x_act;
    const float* addend;  // out += addend[row, col] at the very end (dX of a cross layer: dx = g W^T + dout), [M, ld_add]
    int64_t ld_add;
};

template <int A>
struct EpiTag {
    static constexpr int value = A;
};
template <bool B>
struct EpiFlag {
    static constexpr bool value = B;
};

__device__ __forceinline__ float epi_act(float v, int act) {
    if (act == MH_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MH_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

// A wavefront's TM x TN grid of 32x32 accumulators -> C.  The common cases (Dense forward: bias + activation; dX: optional
// relu mask) on a tile without ragged edges run a loop body specialised on the activation with row pointers advanced by
// additions and the column tiles as immediate offsets: the generic body costs ~25 scalar / vector instructions per element
// (runtime activation tests, 64-bit row * ld products, bounds tests) -- with one workgroup generation per CU and all of
// them in lockstep that was ~10 us of an 80 us skinny-layer GEMM during which no MFMA issues.
template <int TM, int TN>
__device__ __forceinline__ void store_tile(f32x16 (&acc)[TM][TN], float* __restrict__ C, int64_t ldc, int64_t row_base,
                                           int col_base, int64_t M, int N, int lane, const EpiArgs& ep) {
    const int l31 = lane & 31, h = lane >> 5;
    const bool full = row_base + TM * 32 <= M && col_base + TN * 32 <= N;  // uniform per wavefront
    const bool plain = !ep.x0 && !ep.p_out && !ep.addend &&
                       (ep.x_act == MH_ACT_NONE || (ep.x_act == MH_ACT_RELU && !ep.bias && ep.act == MH_ACT_NONE));
    if (full && ep.addend && !ep.x0 && !ep.p_out && !ep.bias && ep.act == MH_ACT_NONE && ep.x_act == MH_ACT_NONE) {
        // dX of a cross layer: out = acc + addend.  The addend values of four rows x TN column blocks are fetched together and
        // then used (same reason as in the cross branch below: loads inside per-element tests compile to a round trip each)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t r0 = row_base + tm * 32 + 8 * g + 4 * h;
                const float* pa = ep.addend + r0 * ep.ld_add + col_base + l31;
                float ad[4][TN];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) ad[q][tn] = pa[q * ep.ld_add + 32 * tn];
                float* pc = C + r0 * ldc + col_base + l31;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) pc[q * ldc + 32 * tn] = acc[tm][tn][4 * g + q] + ad[q][tn];
            }
        return;
    }
    if (full && plain) {
        float bv[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bv[tn] = ep.bias ? ep.bias[col_base + tn * 32 + l31] : 0.f;
        float* p0 = C + (row_base + 4 * h) * ldc + col_base + l31;
        const int64_t ld8 = 8 * ldc;
        auto body = [&](auto act_tag, auto mask_tag) {
            constexpr int ACT = decltype(act_tag)::value;
            constexpr bool MASK = decltype(mask_tag)::value;
            const float* m0 = MASK ? ep.maskx + (row_base + 4 * h) * ep.ldm + col_base + l31 : nullptr;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                float* pg = p0 + (int64_t)tm * 32 * ldc;
                const float* mg = MASK ? m0 + (int64_t)tm * 32 * ep.ldm : nullptr;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float* pr = pg;
                    const float* mr = mg;
                    // the mask values of the group's 4 rows are fetched BEFORE its first store: nothing tells the compiler that
                    // C and the mask do not overlap, so a mask load behind a store stayed behind it -- load, wait, store per
                    // element (seen in the ISA)
                    float mk[4][TN];
                    if (MASK) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn) mk[q][tn] = mr[q * ep.ldm + 32 * tn];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) {
                            float v = acc[tm][tn][4 * g + q] + bv[tn];
                            if (ACT == MH_ACT_RELU) v = v > 0.f ? v : 0.f;
                            if (ACT == MH_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                            if (MASK) v = (mk[q][tn] > 0.f) ? v : 0.f;
                            pr[32 * tn] = v;
                        }
                        pr += ldc;
                    }
                    pg += ld8;
                    if (MASK) mg += 8 * ep.ldm;
                }
            }
        };
        if (ep.x_act == MH_ACT_RELU) body(EpiTag<MH_ACT_NONE>{}, EpiFlag<true>{});
        else if (ep.act == MH_ACT_RELU) body(EpiTag<MH_ACT_RELU>{}, EpiFlag<false>{});
        else if (ep.act == MH_ACT_SIGMOID) body(EpiTag<MH_ACT_SIGMOID>{}, EpiFlag<false>{});
        else body(EpiTag<MH_ACT_NONE>{}, EpiFlag<false>{});
        return;
    }
    if (full && ep.x0 && ep.act == MH_ACT_NONE && ep.x_act == MH_ACT_NONE) {
        // Cross layer (optionally storing the pre-activation p): out = x0 * (acc + b) + x.  The x0 / x values of four rows x TN
        // column blocks are fetched together, then used: in the element-wise fallback below every element's two loads sit
        // inside `if (row < M)` with their use right behind them, and each compiled to a load + s_waitcnt vmcnt(0) -- 128
        // sequential round trips per wavefront at the end of every 256 x 128 tile (seen in the ISA).
        float bv[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bv[tn] = ep.bias ? ep.bias[col_base + tn * 32 + l31] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t r0 = row_base + tm * 32 + 8 * g + 4 * h;
                const float* px0 = ep.x0 + r0 * ep.ld_x0 + col_base + l31;
                const float* pxr = ep.xres + r0 * ep.ld_x0 + col_base + l31;
                float a0[4][TN], ar[4][TN];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        a0[q][tn] = px0[q * ep.ld_x0 + 32 * tn];
                        ar[q][tn] = pxr[q * ep.ld_x0 + 32 * tn];
                    }
                float* pc = C + r0 * ldc + col_base + l31;
                float* pp = ep.p_out ? ep.p_out + r0 * ep.ldp + col_base + l31 : nullptr;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        float v = acc[tm][tn][4 * g + q] + bv[tn];
                        if (pp) pp[q * ep.ldp + 32 * tn] = v;
                        v = a0[q][tn] * v + ar[q][tn];
                        pc[q * ldc + 32 * tn] = v;
                    }
            }
        return;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = col_base + tn * 32 + l31;
        if (col >= N) continue;
        const float bv = ep.bias ? ep.bias[col] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row_base + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M) {
                    float v = acc[tm][tn][r] + bv;
                    if (ep.p_out) ep.p_out[row * ep.ldp + col] = v;
                    if (ep.x0) v = ep.x0[row * ep.ld_x0 + col] * v + ep.xres[row * ep.ld_x0 + col];
                    v = epi_act(v, ep.act);
                    if (ep.x_act == MH_ACT_RELU) {
                        v = (ep.maskx[row * ep.ldm + col] > 0.f) ? v : 0.f;
                    } else if (ep.x_act == MH_ACT_SIGMOID) {
                        const float xx = ep.maskx[row * ep.ldm + col];
                        v *= xx * (1.f - xx);
                    }
                    if (ep.addend) v += ep.addend[row * ep.ld_add + col];
                    C[row * ldc + col] = v;
                }
            }
        }
    }
}

// C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int acc_col(int lane) { return lane & 31; }

}  // namespace mhgemm
