// Dense layers with N = 128 outputs and a short contraction (K <= 1024) at large batch -- the tower layers north_star's MFMA target is about
// (tf/blocks/mlp.py:275-280: the DLRM's 415 -> 128, the two-tower's 256 -> 128) -- on the bf16 matrix pipe in a SIX-term split, "bf16x6":
// every fp32 operand is x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (three 8-bit pieces of the 24-bit significand:
// |x - h - m - l| <= 2^-27 |x|), every product is   h h + (h m + m h) + (h l + l h + m m)   on v_mfma_f32_32x32x16_bf16 with fp32 accumulators.
// bf16 x bf16 products are exact in fp32; the dropped terms (m l, l m, l l) are <= 2^-25 |x y| -- HALF an fp32 rounding of the product itself.
// The arithmetic is therefore as accurate as the fp32 fmaf chain of mh_linear*.hip (it differs from it by the order of the additions, like
// any other fp32 GEMM does), at 16 / 6 of the fp32 MFMA rate: six bf16 MFMAs (2.5 PFLOP/s) per fp32-equivalent one (157 TFLOP/s).  Unlike the
// three-term "bf16x3" of mh_gemm_split.hip (2^-17 per operand: opt-in, own dtype label) this is a DEFAULT path for the shapes it covers;
// MERLIN_HIP_GEMM_ARITH=f32 keeps the exact-chain kernels.
//
// Why these layers: K <= 1024 and N = 128 make a product of 65 536 x 415 x 128 a 143 MB stream with 7 GFLOP -- 24 us of HBM time, 44 us of
// fp32 MFMA time at 100 % of that pipe (the round-5 kernels: 77 us forward, 104 us dX, 76 us dW alone and 241 us beside the interaction
// backward: the largest consumers of GPU time of the DLRM step), 17 us of bf16 MFMA time in six terms.
//
// Forward (tower_fwd_kernel): a workgroup of 4 wavefronts owns 128 rows x all 128 columns.  The A operand never touches LDS: a lane's MFMA
// fragment is 8 consecutive k of ONE row -- 32 contiguous bytes of x -- loaded straight from global memory (two 16-byte loads, two k-tiles
// ahead) and split into (h, m, l) in registers.  The B operand (W^T as three [128][Kp] bf16 images, made once per call by tower_prep_kernel:
// 320 KB, L2-resident) streams through a 3-deep LDS ring of 24 KB k-tiles by direct-to-LDS DMA, chunk-swizzled like mh_gemm_split.hip.
#include "mh_common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int TW_BK = 32;                    // k-tile
constexpr int TW_N = 128;                    // output columns (all of them in one workgroup)
constexpr int TW_WV = 4;                     // wavefronts per workgroup, 32 rows each
constexpr int TW_BM = TW_WV * 32;            // rows per workgroup
constexpr int TW_ST = 3;                     // LDS ring depth
constexpr int TW_IMG = TW_N * TW_BK * 2;     // bytes of one image of a k-tile (8 KB)
constexpr int TW_STAGE = 3 * TW_IMG;         // h, m, l
constexpr int TW_DMA = TW_STAGE / (TW_WV * 64 * 16);  // 16-byte chunks per thread and k-tile (6)

__device__ __forceinline__ void tw_dma16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
__device__ __forceinline__ f32x16 tw_mfma(bf16x8_t a, bf16x8_t b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// (v0, v1) -> packed bf16 pairs h, m, l with v = h + m + l (+ <= 2^-27 |v|); round-to-nearest-even conversions (v_cvt_pk_bf16_f32)
__device__ __forceinline__ void tw_split3(float v0, float v1, uint32_t& h, uint32_t& m, uint32_t& l) {
    const mh_f32x2_t v = {v0, v1};
    h = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mh_bf16x2_t));
    const mh_f32x2_t hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    const mh_f32x2_t r1 = v - hf;  // exact
    m = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, mh_bf16x2_t));
    const mh_f32x2_t mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    const mh_f32x2_t r2 = r1 - mf;  // exact
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, mh_bf16x2_t));
}

struct Frag3 {
    bf16x8_t h, m, l;
};

__device__ __forceinline__ Frag3 tw_split8(const f32x4 a, const f32x4 b) {
    uint32_t h[4], m[4], l[4];
    tw_split3(a.x, a.y, h[0], m[0], l[0]);
    tw_split3(a.z, a.w, h[1], m[1], l[1]);
    tw_split3(b.x, b.y, h[2], m[2], l[2]);
    tw_split3(b.z, b.w, h[3], m[3], l[3]);
    Frag3 f;
    f.h = __builtin_bit_cast(bf16x8_t, make_uint4(h[0], h[1], h[2], h[3]));
    f.m = __builtin_bit_cast(bf16x8_t, make_uint4(m[0], m[1], m[2], m[3]));
    f.l = __builtin_bit_cast(bf16x8_t, make_uint4(l[0], l[1], l[2], l[3]));
    return f;
}

// src [R, C] fp32 (leading dimension ld) -> three images img[i][Rp][Cp] bf16 (i = h, m, l), zero outside [R, C].  TRANSPOSE: the images hold
// src^T ([C rows -> Rp_rows][R -> k]): image row = a column of src.  One thread per 8 consecutive k of one image row.
__global__ __launch_bounds__(256) void tower_prep_kernel(const float* __restrict__ src, int R, int C, int64_t ld, int rows_p, int kp,
                                                        int transpose, uint16_t* __restrict__ img) {
    const int g8 = kp / 8;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)rows_p * g8) return;
    const int r = (int)(i / g8), k0 = (int)(i - (int64_t)r * g8) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        // not transposed: image row r = src row r, k = src column; transposed: image row r = src column r, k = src row
        const int sr = transpose ? k : r, sc = transpose ? r : k;
        v[j] = (sr < R && sc < C) ? src[(int64_t)sr * ld + sc] : 0.f;
    }
    const Frag3 f = tw_split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]});
    const int64_t one = (int64_t)rows_p * kp;
    const int64_t off = (int64_t)r * kp + k0;
    *reinterpret_cast<uint4*>(img + off) = __builtin_bit_cast(uint4, f.h);
    *reinterpret_cast<uint4*>(img + one + off) = __builtin_bit_cast(uint4, f.m);
    *reinterpret_cast<uint4*>(img + 2 * one + off) = __builtin_bit_cast(uint4, f.l);
}

struct TwArgs {
    const float* x;       // A [M, K] fp32, leading dimension ldx (16-byte aligned rows)
    int64_t ldx, M;
    int K, Kp;            // Kp = K rounded up to the k-tile
    const uint16_t* bimg; // B^T images [3][128][Kp]
    const float* bias;    // [128] or null
    int act;
    float* y;
    int64_t ldy;
    int lab;  // -DMH_LAB builds only (MERLIN_HIP_TOWER_ABLATE; results are wrong): 1 no A loads, 2 no MFMAs, 4 no B tile loads
};

__device__ __forceinline__ float tw_act(float v, int act) {
    if (act == MH_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MH_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

__global__ __launch_bounds__(TW_WV * 64, 2) void tower_fwd_kernel(const TwArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char tw_smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * TW_BM + wave * 32;
    int64_t row = row0 + l31;
    if (row > a.M - 1) row = a.M - 1;  // clamped: the products of such rows are never stored
    const int T = a.Kp / TW_BK;
    const int64_t one = (int64_t)TW_N * a.Kp;  // elements of one B image

    // DMA sources of this thread inside a k-tile: chunk L = j * 256 + tid of the 1536-chunk stage image; position p of row r holds source
    // chunk p ^ ((r >> 2) & 3): ds_read_b128 of 32 consecutive rows at one chunk index then spreads over all banks
    const uint16_t* src[TW_DMA];
#pragma unroll
    for (int j = 0; j < TW_DMA; ++j) {
        const int L = j * (TW_WV * 64) + threadIdx.x;
        const int im = L / (TW_N * 4), Lp = L % (TW_N * 4), r = Lp >> 2, p = Lp & 3;
        const int c = p ^ ((r >> 2) & 3);
        src[j] = a.bimg + im * one + (int64_t)r * a.Kp + c * 8;
    }
    auto issue_b = [&](int t) {
        unsigned char* st = tw_smem + (t % TW_ST) * TW_STAGE;
#pragma unroll
        for (int j = 0; j < TW_DMA; ++j) tw_dma16(src[j] + ((a.lab & 4) ? 0 : t * TW_BK), st + (j * (TW_WV * 64) + wave * 64) * 16);
    };
    // A: the lane's 2 x 8 floats of k-tile t (k-step ks: floats t * 32 + ks * 16 + h * 8 .. + 7 of its row).  ALWAYS four loads (the explicit
    // vmcnt arithmetic below counts them): a piece at or past ldx -- or a tile past the last -- re-reads the start of the row and is zeroed.
    const float* xr = a.x + row * a.ldx;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_a = [&](int t, f32x4 (&v)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = t * TW_BK + (q >> 1) * 16 + h * 8 + (q & 1) * 4;
            const bool ok = (k + 4 <= a.ldx) && t < T;  // columns at or past ldx do not exist (their B rows are zero anyway)
#ifdef MH_LAB
            if (a.lab & 1) { v[q] = f32x4{(float)k, 1.f, 2.f, (float)lane}; continue; }
#endif
            const f32x4 w = *reinterpret_cast<const f32x4*>(xr + (ok ? k : 0));
            v[q] = ok ? w : zero4;
        }
    };
    f32x16 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    f32x4 a0[4], a1[4], a2[4];
    // prologue: A(0), B(0), A(1), B(1) in flight, in this order
    load_a(0, a0);
    issue_b(0);
    load_a(1, a1);
    if (1 < T) issue_b(1);
    int b_off[4], b_sw[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int r = nb * 32 + l31;
        b_off[nb] = r * (TW_BK * 2);
        b_sw[nb] = (r >> 2) & 3;
    }
    // One k-tile: `cur` holds the A registers of tile t; `fill` (the registers of tile t - 1, dead) takes tile t + 2.  The loop below is
    // unrolled three times with the three register sets rotating by NAME: a register-to-register rotation would have to wait for the
    // loads it moves.
    auto read_b = [&](const unsigned char* st, int ks, Frag3 (&fb)[4]) {
        const int c = 2 * ks + h;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const unsigned char* p = st + b_off[nb] + ((c ^ b_sw[nb]) << 4);
            fb[nb].h = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p));
            fb[nb].m = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p + TW_IMG));
            fb[nb].l = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p + 2 * TW_IMG));
        }
    };
    auto mfma24 = [&](const Frag3& fa, const Frag3 (&fb)[4]) {
        // six terms, small first; consecutive MFMAs go to DIFFERENT accumulators (four independent chains)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = tw_mfma(fa.l, fb[nb].h, acc[nb]);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = tw_mfma(fa.h, fb[nb].l, acc[nb]);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = tw_mfma(fa.m, fb[nb].m, acc[nb]);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = tw_mfma(fa.m, fb[nb].h, acc[nb]);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = tw_mfma(fa.h, fb[nb].m, acc[nb]);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = tw_mfma(fa.h, fb[nb].h, acc[nb]);
    };
    auto tile = [&](int t, f32x4 (&cur)[4], f32x4 (&fill)[4]) {
        // tile t (its A registers and its B stage) is complete when at most the loads of tile t + 1 are outstanding (4 + TW_DMA of them)
        if (t + 1 < T) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 + TW_DMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");  // only the four dummy A loads of tile T may be outstanding
        // every wavefront has finished tile t - 1 (barrier): its stage takes tile t + 2
        load_a(t + 2, fill);
        if (t + 2 < T) issue_b(t + 2);
        const unsigned char* st = tw_smem + (t % TW_ST) * TW_STAGE;
        // (a hand-interleaved form of the tile -- the fragments and the split of k-step 1 inside the MFMAs of k-step 0, pinned by scheduling
        // barriers -- needed all 256 registers and measured 67 us against 62 for this plain order at 65 536 x 415: not kept)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag3 fb[4];
            read_b(st, ks, fb);
            const Frag3 fa = tw_split8(cur[2 * ks], cur[2 * ks + 1]);
#ifdef MH_LAB
            if (a.lab & 2) {
                acc[0][0] += __builtin_bit_cast(f32x4, fa.h).x + __builtin_bit_cast(f32x4, fa.l).y + __builtin_bit_cast(f32x4, fb[0].h).x + __builtin_bit_cast(f32x4, fb[3].l).w;
                continue;
            }
#endif
            mfma24(fa, fb);
        }
    };
    for (int t = 0; t < T; t += 3) {
        tile(t, a0, a2);
        if (t + 1 < T) tile(t + 1, a1, a0);
        if (t + 2 < T) tile(t + 2, a2, a1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy A loads of the tiles past the last
    // epilogue: C layout of the 32 x 32 block: entry i of lane (l31, h) = row (i >> 2) * 8 + h * 4 + (i & 3), column l31
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int col = nb * 32 + l31;
        const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t r = row0 + (i >> 2) * 8 + h * 4 + (i & 3);
            if (r < a.M) a.y[r * a.ldy + col] = tw_act(acc[nb][i] + bv, a.act);
        }
    }
}

// dX of a tower layer: dx[M, Kin] = dz[M, 128] W^T (W [Kin, 128] row-major IS the B^T operand: row n' = an input column, k' = the 128 outputs).
// The contraction is SHORT (8 k-steps): a wavefront loads and splits its 32 rows of dz ONCE (96 registers: (h, m, l) x 8 k-steps) and keeps them
// for the whole kernel; the 32-column blocks of the output (13 at Kin = 415) are walked one per iteration, their B images (32 rows x 128 k x 3 =
// 24 KB) streaming through the same 3-deep LDS ring as the forward's k-tiles.  Per block 48 MFMAs on two accumulators (even / odd k-steps:
// two independent chains), summed in the epilogue, which stores the finished 32 x 32 block while the next one is computed.
struct TwDxArgs {
    const float* dz;      // [M, 128] fp32, leading dimension lddz
    int64_t lddz, M;
    int Kin, Np;          // output columns, rounded up to 32
    const uint16_t* bimg; // W images [3][Np][128]
    float* dx;
    int64_t lddx;
};

__global__ __launch_bounds__(TW_WV * 64, 2) void tower_dx_kernel(const TwDxArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char tw_smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * TW_BM + wave * 32;
    int64_t row = row0 + l31;
    if (row > a.M - 1) row = a.M - 1;
    const int NBK = a.Np / 32;
    const int64_t one = (int64_t)a.Np * TW_N;  // elements of one image
    // DMA: chunk L = j * 256 + tid of a block's stage image [3][32 rows][16 chunks]; position p of row r holds source chunk p ^ (r & 15)
    const uint16_t* src[TW_DMA];
#pragma unroll
    for (int j = 0; j < TW_DMA; ++j) {
        const int L = j * (TW_WV * 64) + threadIdx.x;
        const int im = L >> 9, Lp = L & 511, r = Lp >> 4, p = Lp & 15;
        src[j] = a.bimg + im * one + (int64_t)r * TW_N + (p ^ (r & 15)) * 8;
    }
    auto issue_b = [&](int nb) {
        unsigned char* st = tw_smem + (nb % TW_ST) * TW_STAGE;
#pragma unroll
        for (int j = 0; j < TW_DMA; ++j) tw_dma16(src[j] + (int64_t)nb * 32 * TW_N, st + (j * (TW_WV * 64) + wave * 64) * 16);
    };
    issue_b(0);
    if (1 < NBK) issue_b(1);
    // the stationary operand: 8 k-steps x (h, m, l)
    Frag3 fa[8];
    {
        const float* zr = a.dz + row * a.lddz + h * 8;
        f32x4 raw[16];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            raw[2 * ks] = *reinterpret_cast<const f32x4*>(zr + ks * 16);
            raw[2 * ks + 1] = *reinterpret_cast<const f32x4*>(zr + ks * 16 + 4);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) fa[ks] = tw_split8(raw[2 * ks], raw[2 * ks + 1]);
    }
    const int rd = l31 * (TW_N * 2);  // byte offset of this lane's B row inside an image of the stage
    for (int nb = 0; nb < NBK; ++nb) {
        // block nb has arrived when at most the DMA of block nb + 1 is outstanding
        if (nb + 1 < NBK) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(TW_DMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (nb + 2 < NBK) issue_b(nb + 2);  // into the stage block nb - 1 has left (every wavefront is past it: barrier)
        const unsigned char* st = tw_smem + (nb % TW_ST) * TW_STAGE;
        f32x16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ks += 2) {
            Frag3 fb0, fb1;
            {
                const unsigned char* p = st + rd + (((2 * ks + h) ^ (l31 & 15)) << 4);
                fb0.h = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p));
                fb0.m = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p + TW_IMG));
                fb0.l = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p + 2 * TW_IMG));
                const unsigned char* q = st + rd + (((2 * ks + 2 + h) ^ (l31 & 15)) << 4);
                fb1.h = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(q));
                fb1.m = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(q + TW_IMG));
                fb1.l = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(q + 2 * TW_IMG));
            }
            // two independent chains (even / odd k-steps), small terms first
            acc0 = tw_mfma(fa[ks].l, fb0.h, acc0);
            acc1 = tw_mfma(fa[ks + 1].l, fb1.h, acc1);
            acc0 = tw_mfma(fa[ks].h, fb0.l, acc0);
            acc1 = tw_mfma(fa[ks + 1].h, fb1.l, acc1);
            acc0 = tw_mfma(fa[ks].m, fb0.m, acc0);
            acc1 = tw_mfma(fa[ks + 1].m, fb1.m, acc1);
            acc0 = tw_mfma(fa[ks].m, fb0.h, acc0);
            acc1 = tw_mfma(fa[ks + 1].m, fb1.h, acc1);
            acc0 = tw_mfma(fa[ks].h, fb0.m, acc0);
            acc1 = tw_mfma(fa[ks + 1].h, fb1.m, acc1);
            acc0 = tw_mfma(fa[ks].h, fb0.h, acc0);
            acc1 = tw_mfma(fa[ks + 1].h, fb1.h, acc1);
        }
        const int col = nb * 32 + l31;
        if (col < a.Kin) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t r = row0 + (i >> 2) * 8 + h * 4 + (i & 3);
                if (r < a.M) a.dx[r * a.lddx + col] = acc0[i] + acc1[i];
            }
        }
    }
}

// (dW = x^T dz of these layers was built in the same arithmetic too -- both operands transposed through LDS as bf16 images [3][column][32 batch
// rows], a workgroup per batch slab accumulating its whole [Kin, 128] partial in registers, partials summed by a second kernel -- and measured
// 72 + 16 us at 65 536 x 415 against 83 us for the exact-chain split-M kernel alone, and a SLOWER train step (0.99 against 0.95 ms: 104 KB of
// LDS per workgroup beside the interaction backward).  Not kept: dW / db of a tower layer stay on the exact-chain kernels; profiles/r6_notes.md.)
inline int64_t tw_al256(int64_t v) { return (v + 255) / 256 * 256; }
inline int tw_kp(int K) { return (K + TW_BK - 1) / TW_BK * TW_BK; }

}  // namespace

extern "C" {

int32_t mh_tower_supported(int64_t M, int32_t K, int32_t N) {
    return (N == TW_N && K >= 32 && K <= 1024 && M >= 4096) ? 1 : 0;
}

int64_t mh_tower_workspace_bytes(int64_t M, int32_t K, int32_t N) {
    if (!mh_tower_supported(M, K, N)) return 0;
    // forward: W^T images [3][128][Kp]; backward: W images [3][Kp][128] as well (dX), the same bytes
    return 2 * tw_al256((int64_t)3 * TW_N * tw_kp(K) * 2) + 256;
}

// y[M, 128] = act(x[M, K] W[K, 128] + b) in the bf16x6 arithmetic (see the file comment): fp32 in, fp32 out, fp32-grade accuracy
int32_t mh_tower_linear_fwd(const float* x, int64_t ldx, const float* W, const float* b, int64_t M, int32_t K, int32_t N, int32_t act,
                            float* y, int64_t ldy, void* workspace, int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(x && W && y, "mh_tower_linear_fwd: null argument");
    MH_REQUIRE(mh_tower_supported(M, K, N), "mh_tower_linear_fwd: shape M=%lld K=%d N=%d is outside this kernel family (N = 128, 32 <= K <= 1024, M >= 4096)",
               (long long)M, K, N);
    MH_REQUIRE(act >= MH_ACT_NONE && act <= MH_ACT_SIGMOID, "mh_tower_linear_fwd: bad activation %d", act);
    MH_REQUIRE(ldx >= K && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "mh_tower_linear_fwd: x rows must be 16-byte aligned (ldx %% 4 == 0)");
    MH_REQUIRE(ldy >= N, "mh_tower_linear_fwd: ldy < N");
    const int64_t need = mh_tower_workspace_bytes(M, K, N);
    MH_REQUIRE(workspace && workspace_bytes >= need, "mh_tower_linear_fwd: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)need);
    hipStream_t s = mh_stream(stream);
    const int Kp = tw_kp(K);
    uint16_t* wt = reinterpret_cast<uint16_t*>(workspace);
    // W [K, 128] -> W^T images [128][Kp]
    MH_LAUNCH(tower_prep_kernel, dim3((unsigned)mh_ceil_div((int64_t)TW_N * (Kp / 8), 256)), dim3(256), 0, s, W, K, N, (int64_t)N, TW_N, Kp, 1, wt);
    static bool attr_done = false;
    const size_t lds = (size_t)TW_ST * TW_STAGE;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(tower_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            mh_set_error("mh_tower_linear_fwd: cannot raise the dynamic LDS limit");
            return MH_ERR_LAUNCH;
        }
        attr_done = true;
    }
    TwArgs a;
    a.x = x; a.ldx = ldx; a.M = M; a.K = K; a.Kp = Kp; a.bimg = wt; a.bias = b; a.act = act; a.y = y; a.ldy = ldy;
    a.lab = 0;
    if (const char* e = MH_LAB_ENV("MERLIN_HIP_TOWER_ABLATE")) a.lab = atoi(e);
    MH_LAUNCH(tower_fwd_kernel, dim3((unsigned)mh_ceil_div(M, TW_BM)), dim3(TW_WV * 64), lds, s, a);
    MH_CHECK_LAUNCH("mh_tower_linear_fwd");
    return MH_OK;
}

// dx[M, K] = dz[M, 128] W^T (W [K, 128]) in the bf16x6 arithmetic; dz is the layer's pre-activation gradient (the caller has applied act')
int32_t mh_tower_linear_dx(const float* dz, int64_t lddz, const float* W, int64_t M, int32_t K, int32_t N, float* dx, int64_t lddx,
                           void* workspace, int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(dz && W && dx, "mh_tower_linear_dx: null argument");
    MH_REQUIRE(mh_tower_supported(M, K, N), "mh_tower_linear_dx: shape M=%lld K=%d N=%d is outside this kernel family", (long long)M, K, N);
    MH_REQUIRE(lddz >= N && lddz % 4 == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0, "mh_tower_linear_dx: dz rows must be 16-byte aligned");
    MH_REQUIRE(lddx >= K, "mh_tower_linear_dx: lddx < K");
    const int64_t need = mh_tower_workspace_bytes(M, K, N);
    MH_REQUIRE(workspace && workspace_bytes >= need, "mh_tower_linear_dx: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)need);
    hipStream_t s = mh_stream(stream);
    const int Np = tw_kp(K);
    // the second half of the workspace: W images [3][Np][128] (the first half holds the forward's W^T images)
    uint16_t* wi = reinterpret_cast<uint16_t*>(static_cast<char*>(workspace) + tw_al256((int64_t)3 * TW_N * Np * 2));
    MH_LAUNCH(tower_prep_kernel, dim3((unsigned)mh_ceil_div((int64_t)Np * (TW_N / 8), 256)), dim3(256), 0, s, W, K, N, (int64_t)N, Np, TW_N, 0, wi);
    static bool attr_done = false;
    const size_t lds = (size_t)TW_ST * TW_STAGE;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(tower_dx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            mh_set_error("mh_tower_linear_dx: cannot raise the dynamic LDS limit");
            return MH_ERR_LAUNCH;
        }
        attr_done = true;
    }
    TwDxArgs a;
    a.dz = dz; a.lddz = lddz; a.M = M; a.Kin = K; a.Np = Np; a.bimg = wi; a.dx = dx; a.lddx = lddx;
    MH_LAUNCH(tower_dx_kernel, dim3((unsigned)mh_ceil_div(M, TW_BM)), dim3(TW_WV * 64), lds, s, a);
    MH_CHECK_LAUNCH("mh_tower_linear_dx");
    return MH_OK;
}

}  // extern "C"
