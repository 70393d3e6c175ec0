// The three GEMMs of a full-rank DCN-v2 cross layer, and wide Dense layers, on the bf16 matrix pipe through a split of every fp32 operand
// into bf16 pieces (mh_set_gemm_arith): mode 2 "bf16x6" -- six terms, fp32-grade, what the host side selects by default -- or mode 1
// "bf16x3" -- three terms, opt-in:
//   forward   out = x0 * (x W + b) + x            (Cross.call, tf/blocks/cross.py:188-202; W [d, d], d = 3341 padded to 3344 at C5)
//   backward  dx = g W^T + dout,  dW = x^T g,  db = column sums of g     (g = dout * x0; mh_cross_layer_bwd's phases)
// The exact-fp32 kernels (mh_gemm2.h) run these at 0.76-0.82 of the 157 TF fp32 MFMA peak: the DCN step is the sum of its GEMMs
// (115 ms, 84 ms at 100 % of that peak).
//   bf16x3: x = hi + lo + r with |r| <= 2^-18 |x|, every product hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_bf16 with fp32
//           accumulators: three bf16 MFMAs per fp32-equivalent one, 16 / 3 of the fp32 rate.  Dropped terms <= 3 * 2^-18 |a b| per
//           element: NOT fp32-grade, never the default, reported under its own dtype label.
//   bf16x6: x = h + m + l (three bf16 pieces hold the 24-bit significand exactly), every product h h + h m + m h + h l + l h + m m:
//           dropped terms <= 2^-25 |a b| -- as close to the real product as the fp32 chain (argument and float64 test:
//           mh_tower_split.hip, tests/test_gpu_gemm_split.py).  Three images per operand (Geo<.., NIMG = 3>), 16 / 6 of the fp32 rate.
//
// ONE kernel form: C[M, N] = A[M, K] B^T with BOTH operands K-contiguous ("NT").  The prepare kernels make that true for the three
// products -- forward: A = x, B^T = W^T; dX: A = g, B^T = W (row-major [d_in, d_out] IS [n][k] for it); dW: A = x^T, B^T = g^T
// (contraction over the batch: both operands are transposed once, 0.35 ms each at 65536 x 3344).  K is padded to a multiple of 32
// with zeros by the same kernels, so the loop has no k-tail.
//
// Kernel: 256 x 128 output tile per workgroup, 8 wavefronts (4 x 2) of 64 x 64 (2 x 2 accumulators of 32 x 32), 32-wide k-tiles of
// A (hi, lo) and B (hi, lo) through a 3-deep LDS ring by direct-to-LDS DMA, one barrier per k-tile; the LDS image is chunk-swizzled
// by permuting the per-lane SOURCE address (chunk c of row r at position c ^ ((r >> 2) & 3): ds_read_b128 of 16 consecutive rows
// hits 16 distinct 4-bank groups).  Column tiles are the fastest grid dimension: the workgroups resident at one time sweep the B
// panel (L2-resident) against a few A panels read from HBM once.
#include "mh_common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int GBK = 32;
// Two geometries of the same kernel (template parameter BM):
//   BM = 256: 8 wavefronts, 3 LDS stages of 48 KB (144 KB): ONE workgroup per CU, its eight wavefronts in lock step at the barriers;
//   BM = 128: 4 wavefronts, 2 stages of 32 KB (64 KB): TWO independent workgroups per CU -- the barrier stall of one is covered
//             by the MFMAs of the other; the A panel is re-read twice as often from L2.
//   BM = 256, BN = 256: 8 wavefronts of 64 x 128 (2 x 4 accumulators), 2 stages of 64 KB: half the barriers and 3 / 4 of the LDS reads per
//             MFMA of the 256 x 128 tile; 7 % of the column tiles of d = 3344 are padding (3.3 % at BN = 128).
//   BM = BN = 256, BK = 16, 5 stages of 32 KB (160 KB): the same tile with FOUR k-tiles in flight instead of one -- the ring of the
//             BK = 32 form tolerates ~1.3 us of load latency (one 64 KB tile ahead of 1.3 us of MFMAs), this one ~2.5 us.
template <int BM, int BN, int BK_ = 32, int ST_ = 0, int NIMG_ = 2>
struct Geo {
    static constexpr int NIMG = NIMG_;                   // bf16 images per operand: 2 (hi, lo: bf16x3) or 3 (h, m, l: bf16x6)
    static constexpr int BK = BK_;
    static constexpr int CPR = BK / 8;                   // 16-byte chunks per row of a k-tile
    static constexpr int RS = BK * 2;                    // bytes per row of a k-tile in LDS
    static constexpr int KS = BK / 16;                   // 16-wide k-steps per tile
    static constexpr int NT = BM * 2;                    // threads: 64 rows per wavefront, two column wavefronts
    static constexpr int NB = BN / 64;                   // 32-column blocks per wavefront
    static constexpr int ST = ST_ ? ST_ : ((BM == 256 && BN == 128 && NIMG_ == 2) ? 3 : 2);  // ring depth
    static constexpr int A_ARR = BM * BK * 2;
    static constexpr int B_ARR = BN * BK * 2;
    static constexpr int STAGE = NIMG * (A_ARR + B_ARR);
    static constexpr int LDS = ST * STAGE;
    static constexpr int DMA_A = NIMG * BM * CPR / NT;   // A chunks of 16 bytes per thread and k-tile
    static constexpr int DMA_B = NIMG * BN * CPR / NT;
    static constexpr int DMA = DMA_A + DMA_B;
};

__device__ __forceinline__ uint16_t g_bf16(float x) {
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float g_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// x [R, C] (leading dimension ld) -> hi, lo [R, Cp] bf16, columns C .. Cp - 1 zero (Cp % 8 == 0).  One 8-column group per thread.
// mid == nullptr: two images (hi, lo) = the three-term arithmetic; mid != nullptr: three images (h, m, l in hi, mid, lo) = the six-term one
__global__ __launch_bounds__(256) void gs_split_rows_kernel(const float* __restrict__ x, int64_t R, int C, int64_t ld, int Cp,
                                                           uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int pack,
                                                           uint16_t* __restrict__ mid = nullptr) {
    const int g8 = Cp / 8;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * g8) return;
    // row-major: thread i -> (row, group of 8 columns), groups fastest.  Packed (k-tile major, see GsArgs): i -> (k-tile, row, quarter
    // of the tile's 32 columns), quarters fastest: four lanes read 128 contiguous bytes of a row and a wavefront writes 1 KB contiguous.
    int64_t r;
    int c0;
    int64_t dst;
    if (pack) {
        const int64_t kt = i / (R * 4), rem = i - kt * (R * 4);
        r = rem >> 2;
        c0 = (int)kt * 32 + (int)(rem & 3) * 8;
        dst = (kt * R + r) * 32 + (rem & 3) * 8;
    } else {
        r = i / g8;
        c0 = (int)(i - r * g8) * 8;
        dst = r * Cp + c0;
    }
    // two 16-byte loads where the row allows them (C % 4 == 0 and 16-byte aligned rows: what the callers pass), scalars otherwise
    float v8[8];
    const bool vec = (C % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = c0 + 4 * q;
        if (vec && c + 4 <= C) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(x + r * ld + c);
            v8[4 * q] = t.x; v8[4 * q + 1] = t.y; v8[4 * q + 2] = t.z; v8[4 * q + 3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v8[4 * q + j] = (c + j < C) ? x[r * ld + c + j] : 0.f;
        }
    }
    uint32_t wh[4], wl[4], wm[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (mid) mh_split3_pair(v8[2 * k], v8[2 * k + 1], wh[k], wm[k], wl[k]);
        else mh_split_pair(v8[2 * k], v8[2 * k + 1], wh[k], wl[k]);
    }
    *reinterpret_cast<uint4*>(hi + dst) = make_uint4(wh[0], wh[1], wh[2], wh[3]);
    *reinterpret_cast<uint4*>(lo + dst) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
    if (mid) *reinterpret_cast<uint4*>(mid + dst) = make_uint4(wm[0], wm[1], wm[2], wm[3]);
}

// x [R, C] (ld) -> hiT, loT [C, Rp] bf16 (the transpose), columns R .. Rp - 1 zero (Rp % 64 == 0).  64 x 64 tiles through LDS.
// The output packs two consecutive ROWS of x into one 32-bit word, so a thread loads the same four columns of rows 2 rp and 2 rp + 1
// (two float4), splits the four (row, row + 1) pairs -- one mh_split*_pair each, the packed word of every image falls out -- and stores
// them as one ds_write_b128 per image; the transposing side reads eight words per image.  (First version: one element per thread and
// pass, 16-bit LDS cells: 48 ds_write_b16 + 48 ds_read_u16 per thread, half of every pair split wasted: 2.8 TB/s.)
__global__ __launch_bounds__(256) void gs_split_transpose_kernel(const float* __restrict__ x, int64_t R, int C, int64_t ld, int64_t Rp,
                                                                uint16_t* __restrict__ hiT, uint16_t* __restrict__ loT, int pack,
                                                                uint16_t* __restrict__ midT = nullptr) {
    __shared__ __attribute__((aligned(16))) uint32_t sw[3][32][68];  // [image][row pair][column]; 272-byte rows keep the b128 stores aligned
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const bool vec_ok = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const int c4 = threadIdx.x & 15, cc = c0 + 4 * c4;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int rp = (threadIdx.x >> 4) + 16 * it;
        const int64_t ra = r0 + 2 * rp, rb = ra + 1;
        float va[4] = {0.f, 0.f, 0.f, 0.f}, vb[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec_ok && cc + 3 < C) {
            if (ra < R) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(x + ra * ld + cc);
                va[0] = t[0]; va[1] = t[1]; va[2] = t[2]; va[3] = t[3];
            }
            if (rb < R) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(x + rb * ld + cc);
                vb[0] = t[0]; vb[1] = t[1]; vb[2] = t[2]; vb[3] = t[3];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ra < R && cc + j < C) va[j] = x[ra * ld + cc + j];
                if (rb < R && cc + j < C) vb[j] = x[rb * ld + cc + j];
            }
        }
        uint32_t wh[4], wm[4] = {0u, 0u, 0u, 0u}, wl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (midT) mh_split3_pair(va[j], vb[j], wh[j], wm[j], wl[j]);
            else mh_split_pair(va[j], vb[j], wh[j], wl[j]);
        }
        *reinterpret_cast<uint4*>(&sw[0][rp][4 * c4]) = make_uint4(wh[0], wh[1], wh[2], wh[3]);
        *reinterpret_cast<uint4*>(&sw[1][rp][4 * c4]) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
        if (midT) *reinterpret_cast<uint4*>(&sw[2][rp][4 * c4]) = make_uint4(wm[0], wm[1], wm[2], wm[3]);
    }
    __syncthreads();
    // thread (c, seg): rows seg * 16 .. + 15 of column c -> 32 contiguous bytes of row c0 + c of the transposed arrays
    const int c = threadIdx.x & 63, seg = threadIdx.x >> 6;
    if (c0 + c >= C) return;
    uint32_t wh[8], wl[8], wm[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        wh[k] = sw[0][seg * 8 + k][c];
        wl[k] = sw[1][seg * 8 + k][c];
        wm[k] = midT ? sw[2][seg * 8 + k][c] : 0u;
    }
    // packed: element (row c0 + c, k = r0 + seg 16 ...) of the [C, Rp] result lives at ((k / 32) C + row) 32 + k % 32
    const int64_t off = pack ? (((r0 >> 5) + (seg >> 1)) * (int64_t)C + c0 + c) * 32 + (seg & 1) * 16 : (int64_t)(c0 + c) * Rp + r0 + seg * 16;
    uint16_t* ph = hiT + off;
    uint16_t* pl = loT + off;
    *reinterpret_cast<uint4*>(ph) = make_uint4(wh[0], wh[1], wh[2], wh[3]);
    *reinterpret_cast<uint4*>(ph + 8) = make_uint4(wh[4], wh[5], wh[6], wh[7]);
    *reinterpret_cast<uint4*>(pl) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
    *reinterpret_cast<uint4*>(pl + 8) = make_uint4(wl[4], wl[5], wl[6], wl[7]);
    if (midT) {
        uint16_t* pm = midT + off;
        *reinterpret_cast<uint4*>(pm) = make_uint4(wm[0], wm[1], wm[2], wm[3]);
        *reinterpret_cast<uint4*>(pm + 8) = make_uint4(wm[4], wm[5], wm[6], wm[7]);
    }
}

struct GsArgs {
    const uint16_t *ai[3], *bi[3];      // images of A [M, Kp] and B^T [N, Kp] (bf16 bit patterns, zero-padded to Kp): [0] = hi, [1] = lo
                                        // (two images, bf16x3) or [0] = h, [1] = m, [2] = l (three, bf16x6).  Row-major (element (r, k) at
                                        // r lda + k) or PACKED k-tile major (at ((k / 32) rows + r) 32 + k % 32: the 64 bytes a row
                                        // contributes to a k-tile lie next to those of its neighbours, so a tile-load instruction of a
                                        // wavefront fetches 1 KB contiguous = 8 whole cache lines instead of 16 half lines)
    int64_t M, lda, ldb;                // row-major leading dimensions (unused when packed)
    int pack;
    int N, Kp;
    float* C;
    int64_t ldc;
    int ntn;             // column tiles
    int xcd_map;         // see the kernel
    int kt_per_split;    // k-tiles per blockIdx.y (split-K: slab z of C at C + z * slab)
    int64_t slab;
    // epilogues
    const float* addend;  // EPI 0: C = acc (+ addend[m, n])
    int64_t ld_add;
    const float *bias, *x0, *xres;  // EPI 1: p = acc + bias[n]; C = x0 * p + xres (all [M, ld_e]); p_out optional
    float* p_out;
    int64_t ld_e;
    int act;  // EPI 2: C = act(acc + bias[n])  (Dense forward)
    int ablate;  // -DMH_LAB builds only: timing experiments (MERLIN_HIP_GEMM_SPLIT_ABLATE; results are wrong): 1 = no tile loads inside the k-loop,
                 // 2 = the loads are issued but a tile is used without waiting for it (up to two tiles in flight)
};

__device__ __forceinline__ void g_dma16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void g_wait_vm_and_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
__device__ __forceinline__ f32x16 g_mfma(bf16x8_t a, bf16x8_t b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

template <int EPI, int BM, int BN, bool PIPE, int BK = 32, int ST = 0, int NIMG = 2>
__global__ __launch_bounds__(BM * 2, BM == 256 ? 1 : 2) void gemm_split_nt_kernel(const GsArgs a) {
    using G = Geo<BM, BN, BK, ST, NIMG>;
    constexpr int TERMS = NIMG == 3 ? 6 : 3;  // bf16 MFMAs per fp32-equivalent one
    constexpr int CPR = G::CPR, RS = G::RS, KS = G::KS;
    constexpr int GBM = BM, GBN = BN, GNT = G::NT, GST = G::ST, G_A_ARR = G::A_ARR, G_B_ARR = G::B_ARR, G_STAGE = G::STAGE, G_DMA = G::DMA;
    constexpr int NB = G::NB;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroup -> output tile.  Plain: column tiles fastest.  XCD-aware (a.xcd_map): workgroup b runs on XCD b % 8 (round-robin
    // dispatch); the j-th workgroup of an XCD takes column tile j % ntn of row tile xcd + 8 (j / ntn) -- all column tiles of one A row
    // panel run on ONE XCD, so the panel (3.5 MB of hi + lo at K = 3392) enters one L2 instead of eight.
    int64_t rt = blockIdx.x / a.ntn;
    int ct = (int)(blockIdx.x % a.ntn);
    if (a.xcd_map) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        rt = xcd + 8 * (int64_t)(j / a.ntn);
        ct = j % a.ntn;
        if (rt * GBM >= a.M) return;
    }
    const int64_t row0 = rt * GBM;
    const int n0 = ct * GBN;
    const int nkt_all = a.Kp / BK;
    const int kt_beg = blockIdx.y * a.kt_per_split;
    const int kt_end = (kt_beg + a.kt_per_split < nkt_all) ? kt_beg + a.kt_per_split : nkt_all;
    const int T = kt_end - kt_beg;

    // DMA sources of this thread inside a k-tile (rows clamped to the last valid row: their products are never stored)
    const uint16_t* src[G_DMA];
#pragma unroll
    for (int j = 0; j < G::DMA_A; ++j) {  // A: NIMG arrays x BM rows x 4 chunks
        const int L = j * GNT + threadIdx.x;
        const int arr = L / (BM * CPR), Lp = L % (BM * CPR), r = Lp / CPR, p = Lp % CPR;
        const int c = p ^ (CPR == 4 ? ((r >> 2) & 3) : ((r >> 3) & 1));
        int64_t row = row0 + r;
        if (row > a.M - 1) row = a.M - 1;
        src[j] = a.ai[arr] + row * (a.pack ? (int64_t)32 : a.lda) + c * 8;
    }
#pragma unroll
    for (int j = 0; j < G::DMA_B; ++j) {  // B: NIMG arrays x BN rows x 4 chunks
        const int L = j * GNT + threadIdx.x;
        const int arr = L / (BN * CPR), Lp = L % (BN * CPR), r = Lp / CPR, p = Lp % CPR;
        const int c = p ^ (CPR == 4 ? ((r >> 2) & 3) : ((r >> 3) & 1));
        int col = n0 + r;
        if (col > a.N - 1) col = a.N - 1;
        src[G::DMA_A + j] = a.bi[arr] + (int64_t)col * (a.pack ? (int64_t)32 : a.ldb) + c * 8;
    }
    const int64_t ks_a = a.pack ? a.M * 32 : (int64_t)BK, ks_b = a.pack ? (int64_t)a.N * 32 : (int64_t)BK;  // elements per k-tile step
    auto issue_at = [&](int tile, int stage) {  // k-tile kt_beg + tile -> LDS stage
        unsigned char* st = smem + stage * G_STAGE;
        const int64_t ka = (int64_t)(kt_beg + tile) * ks_a, kb = (int64_t)(kt_beg + tile) * ks_b;
#pragma unroll
        for (int j = 0; j < G::DMA_A; ++j) g_dma16(src[j] + ka, st + (j * GNT + wave * 64) * 16);
#pragma unroll
        for (int j = 0; j < G::DMA_B; ++j) g_dma16(src[G::DMA_A + j] + kb, st + NIMG * G_A_ARR + (j * GNT + wave * 64) * 16);
    };
    auto issue = [&](int t) { issue_at(t, t % GST); };
    f32x16 acc[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
#pragma unroll
    for (int t = 0; t < GST - 1; ++t)
        if (t < T) issue(t);
    // LDS byte offsets of this lane's fragments inside a stage (k-step s adds the swizzled chunk)
    int a_off[2], a_sw[2], b_off[NB], b_sw[NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int r = wm * 64 + mb * 32 + l31;
        a_off[mb] = r * RS;
        a_sw[mb] = CPR == 4 ? ((r >> 2) & 3) : ((r >> 3) & 1);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int r = wn * (BN / 2) + nb * 32 + l31;
        b_off[nb] = NIMG * G_A_ARR + r * RS;
        b_sw[nb] = CPR == 4 ? ((r >> 2) & 3) : ((r >> 3) & 1);
    }
    // fragments of one 16-wide k-step: fa[image][row block], fb[image][column block]; image 0 = hi / h, 1 = lo / m, 2 = l
    auto read_frags = [&](const unsigned char* st, int ks, bf16x8_t (&fa)[NIMG][2], bf16x8_t (&fb)[NIMG][NB]) {
        const int c = 2 * ks + h;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const unsigned char* p = st + a_off[mb] + ((c ^ a_sw[mb]) << 4);
#pragma unroll
            for (int im = 0; im < NIMG; ++im) fa[im][mb] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p + im * G_A_ARR));
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const unsigned char* p = st + b_off[nb] + ((c ^ b_sw[nb]) << 4);
#pragma unroll
            for (int im = 0; im < NIMG; ++im) fb[im][nb] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p + im * G_B_ARR));
        }
    };
    // terms in the OUTER loop: consecutive MFMAs go to different accumulators (2 x NB independent chains); every accumulator still takes its
    // terms small first: lo hi, hi lo, hi hi (bf16x3) / l h, h l, m m, m h, h m, h h (bf16x6)
    auto mfma_step = [&](const bf16x8_t (&fa)[NIMG][2], const bf16x8_t (&fb)[NIMG][NB]) {
        constexpr int TA[2][6] = {{1, 0, 0, 0, 0, 0}, {2, 0, 1, 1, 0, 0}};  // image of A per term
        constexpr int TB[2][6] = {{0, 1, 0, 0, 0, 0}, {0, 2, 1, 0, 1, 0}};  // image of B per term
#pragma unroll
        for (int tm = 0; tm < TERMS; ++tm)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = g_mfma(fa[TA[NIMG - 2][tm]][mb], fb[TB[NIMG - 2][tm]][nb], acc[mb][nb]);
    };
    if (PIPE && KS == 2 && GST == 2) {
        // The barrier of a k-tile sits in the MIDDLE of the previous tile's MFMAs, and every batch of fragment reads is issued right
        // behind the FIRST MFMA of the other batch: [MFMA, 12 LDS reads, 23 MFMAs] twice per tile.  The fragments of k-step 0 of tile
        // t + 1 fly while k-step 1 of tile t runs, those of k-step 1 while k-step 0 runs; the wait the compiler puts in front of the
        // first MFMA of a batch (lgkmcnt(0): it cannot count across the back edge) then only covers reads issued ~750 cycles
        // earlier.  (The plain loop reads the whole tile right behind its barrier: eight wavefronts want 96 KB from LDS at once and
        // the matrix pipe idles for the first of it -- halving the k-tile, i.e. doubling the number of such phases, cost 24 %.)
        // No branch inside the body (the scheduler works on one block): the last iterations re-load tile t into the stage it just
        // left and read fragments nobody uses.
        bf16x8_t fa0[NIMG][2], fb0[NIMG][NB], fa1[NIMG][2], fb1[NIMG][NB];
        g_wait_vm_and_barrier<0>();  // tile 0
        if (1 < T) issue(1);
        read_frags(smem, 0, fa0, fb0);
        for (int t = 0; t < T; ++t) {
            const unsigned char* st = smem + (t % GST) * G_STAGE;
            const unsigned char* sn = smem + ((t + 1) % GST) * G_STAGE;
            read_frags(st, 1, fa1, fb1);
            mfma_step(fa0, fb0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NIMG * (2 + NB), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TERMS * NB - 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            // tile t + 1 has arrived and every wavefront has read all of tile t (its k-step-1 fragments are in registers: lgkmcnt 0)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            issue_at((t + 2 < T) ? t + 2 : t, t % GST);  // tile t + 2 (or, at the end, tile t again) into the stage tile t has just left
            read_frags(sn, 0, fa0, fb0);
            mfma_step(fa1, fb1);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x100, NIMG * (2 + NB), 1);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TERMS * NB - 1, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy re-loads of the last iterations
    } else {
    for (int t = 0; t < T; ++t) {
        // tile t is complete when at most the loads of tiles t + 1 .. t + GST - 2 are outstanding
#ifdef MH_LAB
        if (a.ablate & 2) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");  // experiment: does not wait for the tile
        else
#endif
        if (t + GST - 2 < T) g_wait_vm_and_barrier<(GST - 2) * G_DMA>();
        else g_wait_vm_and_barrier<0>();
#ifdef MH_LAB
        if (t + GST - 1 < T && !(a.ablate & 1)) issue(t + GST - 1);
#else
        if (t + GST - 1 < T) issue(t + GST - 1);
#endif
        const unsigned char* st = smem + (t % GST) * G_STAGE;
        // The fragments of ALL 16-wide k-steps of the tile are read from LDS up front (LDS returns in order: the MFMAs of step 0 wait
        // for the first half only), so the reads of step 1 can run behind the MFMAs of step 0.
        bf16x8_t fa[KS][NIMG][2], fb[KS][NIMG][NB];
#pragma unroll
        for (int s = 0; s < KS; ++s) read_frags(st, s, fa[s], fb[s]);
#pragma unroll
        for (int s = 0; s < KS; ++s) mfma_step(fa[s], fb[s]);
    }
    }
    // ---- epilogue: acc[mb][nb][i] = C[row0 + wm 64 + mb 32 + (i & 3) + 8 (i >> 2) + 4 h][n0 + wn 64 + nb 32 + l31] --------------------
    // The operands of an epilogue (x0 and x of the cross form, the addend of the dX form) are fetched for a whole 32 x 32 block before
    // the first of them is used: 16-32 loads in flight.  (First version: load, use, store per element -- 64 dependent round trips per
    // lane at the end of every workgroup, ~40 % of its life at K = 3392.)  Rows past M read row M - 1 and are not stored.
    float* C = a.C + (int64_t)blockIdx.y * a.slab;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + wn * (BN / 2) + nb * 32 + l31;
        const bool n_ok = n < a.N;
        const int nc = n_ok ? n : a.N - 1;
        float bias = 0.f;
        if ((EPI == 1 || EPI == 2) && a.bias) bias = a.bias[nc];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int64_t mbase = row0 + wm * 64 + mb * 32 + 4 * h;
            // EB elements per batch: the cross form holds two operands per element beside the 128 accumulators -- 16 at a time spilled 61
            // registers per lane (244 bytes of scratch in the resource report); 8 at a time fit
            constexpr int EB = (EPI == 1) ? 8 : 16;
#pragma unroll
            for (int i0 = 0; i0 < 16; i0 += EB) {
                float e0[EB], e1[EB];
                if (EPI == 1 || a.addend) {
#pragma unroll
                    for (int u = 0; u < EB; ++u) {
                        const int i = i0 + u;
                        int64_t m = mbase + (i & 3) + 8 * (i >> 2);
                        if (m > a.M - 1) m = a.M - 1;
                        if (EPI == 1) {
                            e0[u] = a.x0[m * a.ld_e + nc];
                            e1[u] = a.xres[m * a.ld_e + nc];
                        } else {
                            e0[u] = a.addend[m * a.ld_add + nc];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < EB; ++u) {
                    const int i = i0 + u;
                    const int64_t m = mbase + (i & 3) + 8 * (i >> 2);
                    float v = acc[mb][nb][i];
                    if (EPI == 1) {
                        v += bias;
                        if (a.p_out && n_ok && m < a.M) a.p_out[m * a.ld_e + n] = v;
                        v = fmaf(e0[u], v, e1[u]);
                    } else if (EPI == 2) {
                        v += bias;
                        if (a.act == MH_ACT_RELU) v = v > 0.f ? v : 0.f;
                        else if (a.act == MH_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                    } else if (a.addend) {
                        v += e0[u];
                    }
                    if (n_ok && m < a.M) C[m * a.ldc + n] = v;
                }
                __builtin_amdgcn_sched_barrier(0);  // the next batch's loads stay behind this batch's stores
            }
        }
    }
}

// out[i] = sum_s part[s][i]  (S slabs of len4 float4, ascending order: deterministic)
__global__ __launch_bounds__(256) void gs_reduce_slabs_kernel(const f32x4* __restrict__ part, int S, int64_t len4, f32x4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= len4) return;
    f32x4 t = part[i];
    for (int s = 1; s < S; ++s) t += part[(int64_t)s * len4 + i];
    out[i] = t;
}

__global__ __launch_bounds__(256) void gs_reduce_slabs_scalar_kernel(const float* __restrict__ part, int S, int64_t len, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    float t = part[i];
    for (int s = 1; s < S; ++s) t += part[(int64_t)s * len + i];
    out[i] = t;
}

// column sums of g [M, d]: stage 1 -- workgroup (column block of 64, row slab) sums its rows in a fixed order.  Vector form (d and ld
// multiples of 4, 16-byte aligned base): a thread owns four columns (one float4 per row) and every 16th row of the slab, four independent
// accumulators; the first version's one column and one dependent add per 4-byte load ran at 2.1 TB/s.
__global__ __launch_bounds__(256) void gs_colsum_kernel(const float* __restrict__ g, int64_t M, int d, int64_t ld, int rows_per_slab,
                                                       float* __restrict__ part) {
    __shared__ float red[16][64];
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
    int64_t r1 = r0 + rows_per_slab;
    if (r1 > M) r1 = M;
    const bool vec_ok = (d % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
    if (vec_ok) {
        const int c4 = threadIdx.x & 15, q = threadIdx.x >> 4, c = blockIdx.x * 64 + 4 * c4;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        if (c < d) {
            int64_t r = r0 + q;
            for (; r + 48 < r1; r += 64) {
                a0 += *reinterpret_cast<const f32x4*>(g + r * ld + c);
                a1 += *reinterpret_cast<const f32x4*>(g + (r + 16) * ld + c);
                a2 += *reinterpret_cast<const f32x4*>(g + (r + 32) * ld + c);
                a3 += *reinterpret_cast<const f32x4*>(g + (r + 48) * ld + c);
            }
            for (; r < r1; r += 16) a0 += *reinterpret_cast<const f32x4*>(g + r * ld + c);
        }
        const f32x4 t = (a0 + a1) + (a2 + a3);
#pragma unroll
        for (int j = 0; j < 4; ++j) red[q][4 * c4 + j] = t[j];
        __syncthreads();
        if (threadIdx.x < 64 && blockIdx.x * 64 + (int)threadIdx.x < d) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += red[k][threadIdx.x];
            part[(int64_t)blockIdx.y * d + blockIdx.x * 64 + threadIdx.x] = sum;
        }
        return;
    }
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float s = 0.f;
    if (c < d)
        for (int64_t r = r0 + q; r < r1; r += 4) s += g[r * ld + c];
    red[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < d) part[(int64_t)blockIdx.y * d + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void gs_colsum_finish_kernel(const float* __restrict__ part, int S, int d, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += part[(int64_t)k * d + c];
    out[c] = s;
}

int g_gemm_arith = 0;  // what the *_split entry points compute in: 1 = bf16x3 (three terms), 2 = bf16x6 (six terms, fp32-grade); 0 behaves as 1 (mh_set_gemm_arith)

inline int64_t al256(int64_t v) { return (v + 255) / 256 * 256; }
inline int64_t pad_to(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// number of bf16 images per operand the entry points of this file use: 2 = the three-term bf16x3 (mh_set_gemm_arith(1)), 3 = the six-term,
// fp32-grade bf16x6 (mh_set_gemm_arith(2)); workspaces are sized for three
int split_images() { return g_gemm_arith == 2 ? 3 : 2; }

struct SplitBuf {
    uint16_t* img[3];  // [0] = hi / h, [1] = lo / m, [2] = l (null with two images)
    int64_t ld;
};

// carve the images of a [rows, ld] operand out of the workspace cursor
SplitBuf take_pair(char*& p, int64_t rows, int64_t ld) {
    SplitBuf b;
    b.ld = ld;
    b.img[2] = nullptr;
    for (int i = 0; i < split_images(); ++i) {
        b.img[i] = reinterpret_cast<uint16_t*>(p);
        p += al256(rows * ld * 2);
    }
    return b;
}
inline int64_t pair_bytes(int64_t rows, int64_t ld) { return 3 * al256(rows * ld * 2); }
inline void set_operands(GsArgs& a, const SplitBuf& A, const SplitBuf& B) {
    for (int i = 0; i < 3; ++i) {
        a.ai[i] = A.img[i];
        a.bi[i] = B.img[i];
    }
}

int gemm_geo();
int gemm_pack() {  // MERLIN_HIP_GEMM_SPLIT_PACK = 1 (default) | 0: k-tile major operand images (GsArgs); the 16-wide k-tile forms are row-major
    static int pack = -1;
    if (pack < 0) {
        const char* e = MH_LAB_ENV("MERLIN_HIP_GEMM_SPLIT_PACK");
        pack = (e ? atoi(e) : 1) && gemm_geo() < 3;
    }
    return pack;
}

void split_rows(const float* x, int64_t R, int C, int64_t ld, const SplitBuf& b, hipStream_t s) {
    const int64_t n = R * (b.ld / 8);
    const bool three = split_images() == 3;
    MH_LAUNCH(gs_split_rows_kernel, dim3((unsigned)mh_ceil_div(n, 256)), dim3(256), 0, s, x, R, C, ld, (int)b.ld, b.img[0], three ? b.img[2] : b.img[1],
              gemm_pack(), three ? b.img[1] : (uint16_t*)nullptr);
}
void split_transpose(const float* x, int64_t R, int C, int64_t ld, const SplitBuf& b, hipStream_t s) {
    const bool three = split_images() == 3;
    MH_LAUNCH(gs_split_transpose_kernel, dim3((unsigned)(b.ld / 64), (unsigned)mh_ceil_div(C, 64)), dim3(256), 0, s, x, R, C, ld, b.ld,
              b.img[0], three ? b.img[2] : b.img[1], gemm_pack(), three ? b.img[1] : (uint16_t*)nullptr);
}

template <int EPI, int BM, int BN, bool PIPE, int BK = 32, int ST = 0, int NIMG = 2>
int32_t launch_gemm_geo(GsArgs a, int splits, hipStream_t s) {
    using G = Geo<BM, BN, BK, ST, NIMG>;
    auto kern = gemm_split_nt_kernel<EPI, BM, BN, PIPE, BK, ST, NIMG>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) {
            mh_set_error("gemm (split bf16): cannot raise the dynamic LDS limit");
            return MH_ERR_LAUNCH;
        }
        attr_done = true;
    }
    a.ntn = (int)mh_ceil_div(a.N, BN);
    const int nkt = a.Kp / BK;
    a.kt_per_split = (int)mh_ceil_div(nkt, splits);
    static int xmap = -1;  // MERLIN_HIP_GEMM_SPLIT_XCD = 0 | 1
    if (xmap < 0) {
        const char* e = MH_LAB_ENV("MERLIN_HIP_GEMM_SPLIT_XCD");
        xmap = e ? atoi(e) : 1;
    }
    a.xcd_map = xmap;
    a.pack = gemm_pack();
    static int ablate = -1;
    if (ablate < 0) {
        const char* e = MH_LAB_ENV("MERLIN_HIP_GEMM_SPLIT_ABLATE");
        ablate = e ? atoi(e) : 0;
    }
    a.ablate = ablate;
    int64_t nrt = mh_ceil_div(a.M, BM);
    if (xmap) nrt = mh_ceil_div(nrt, 8) * 8;  // whole groups of 8 row tiles (the surplus workgroups exit at once)
    const int64_t tiles = nrt * a.ntn;
    MH_REQUIRE(tiles < (1ll << 31), "gemm (split bf16): grid too large");
    MH_LAUNCH(kern, dim3((unsigned)tiles, (unsigned)splits), dim3(G::NT), (size_t)G::LDS, s, a);
    return MH_OK;
}

int gemm_geo() {  // MERLIN_HIP_GEMM_SPLIT_GEO = 256x256 (default) | 256x128 | 128x128
    static int geo = -1;
    if (geo < 0) {
        const char* e = MH_LAB_ENV("MERLIN_HIP_GEMM_SPLIT_GEO");
        // measured at 65536 x 3344 x 3344 (DCN-v2 step, same box): 256x256 56.9 ms, 256x128 59.5 ms, 128x128 79 ms
        geo = !e ? 2 : (!strcmp(e, "128x128") ? 1 : (!strcmp(e, "256x128") ? 0 : (!strcmp(e, "256x256k16") ? 3 : (!strcmp(e, "256x256k16s4") ? 4 : 2))));
    }
    return geo;
}

template <int EPI>
int32_t launch_gemm(GsArgs a, int splits, hipStream_t s) {
    // six-term arithmetic (three images per operand): 256 x 128 tiles, two 72 KB stages, the k-loop with the mid-tile barrier
#ifdef MH_LAB  // lab: 256 x 256 tiles with 16-wide k-tiles in a three-stage ring (3 x 48 KB) for the six-term arithmetic
    if (split_images() == 3 && gemm_geo() == 3) return launch_gemm_geo<EPI, 256, 256, false, 16, 3, 3>(a, splits, s);
    if (split_images() == 3 && gemm_geo() == 4) return launch_gemm_geo<EPI, 256, 256, true, 16, 3, 3>(a, splits, s);
    if (split_images() == 3 && gemm_geo() == 0) return launch_gemm_geo<EPI, 256, 128, false, 32, 2, 3>(a, splits, s);
#endif
    if (split_images() == 3) return launch_gemm_geo<EPI, 256, 128, true, 32, 2, 3>(a, splits, s);
    const int geo = gemm_geo();
    static int pipe = -1;  // MERLIN_HIP_GEMM_SPLIT_PIPE = 1 | 0: the k-loop with the barrier in the middle of a tile's MFMAs (see the kernel)
    if (pipe < 0) {
        const char* e = MH_LAB_ENV("MERLIN_HIP_GEMM_SPLIT_PIPE");
        pipe = e ? atoi(e) : 1;  // measured (DCN-v2 step, same box, row-major operands): 54.2-54.5 ms plain, 53.2-53.5 with the mid-tile barrier
    }
    if (geo == 1) return launch_gemm_geo<EPI, 128, 128, false>(a, splits, s);
    // 16-wide k-tiles, 5 / 4 stages (three / two more tiles in flight): +24 % time, also with the mid-tile barrier -- twice the tile-load
    // requests per byte (32-byte row pieces); kept selectable for that measurement only
    if (geo == 3) return launch_gemm_geo<EPI, 256, 256, false, 16, 5>(a, splits, s);
    if (geo == 4) return launch_gemm_geo<EPI, 256, 256, false, 16, 4>(a, splits, s);
    if (geo == 2) return pipe ? launch_gemm_geo<EPI, 256, 256, true>(a, splits, s) : launch_gemm_geo<EPI, 256, 256, false>(a, splits, s);
    return pipe ? launch_gemm_geo<EPI, 256, 128, true>(a, splits, s) : launch_gemm_geo<EPI, 256, 128, false>(a, splits, s);
}

}  // namespace

extern "C" {

int32_t mh_set_gemm_arith(int32_t mode) {
    MH_REQUIRE(mode >= 0 && mode <= 2, "mh_set_gemm_arith: mode must be 0 (f32), 1 (bf16x3) or 2 (bf16x6)");
    g_gemm_arith = mode;
    return MH_OK;
}

int64_t mh_cross_layer_split_workspace_bytes(int64_t M, int32_t d) {
    if (M <= 0 || d <= 0) return 0;
    const int64_t Kp = pad_to(d, 64), Mp = pad_to(M, 64);
    // forward: x, W^T;  backward: g, W, x^T, g^T, two dW slabs, db partials
    const int64_t fwd = pair_bytes(M, Kp) + pair_bytes(d, Kp);
    const int64_t bwd = pair_bytes(M, Kp) + pair_bytes(d, Kp) + 2 * pair_bytes(d, Mp) + al256(8 * (int64_t)d * d * 4) + al256(64 * (int64_t)d * 4);
    return (fwd > bwd ? fwd : bwd) + 1024;
}

int32_t mh_cross_layer_fwd_split(const float* x0, const float* x, const float* W, const float* b, int64_t M, int32_t d, float* out,
                                 float* p_out, void* workspace, int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(x0 && x && W && out, "mh_cross_layer_fwd_split: null argument");
    MH_REQUIRE(M >= 0 && d >= 8 && d % 4 == 0, "mh_cross_layer_fwd_split: bad shape M=%lld d=%d (d must be a multiple of 4)", (long long)M, d);
    if (M == 0) return MH_OK;
    MH_REQUIRE(workspace && workspace_bytes >= mh_cross_layer_split_workspace_bytes(M, d), "mh_cross_layer_fwd_split: workspace too small");
    hipStream_t s = mh_stream(stream);
    char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t Kp = pad_to(d, 64);
    const SplitBuf sx = take_pair(p, M, Kp), sw = take_pair(p, d, Kp);
    split_rows(x, M, d, d, sx, s);
    split_transpose(W, d, d, d, sw, s);  // W^T [n = d_out][k = d_in], pad columns zero (Kp is a multiple of 64: whole tiles)
    GsArgs a{};
    set_operands(a, sx, sw);
    a.M = M; a.N = d; a.Kp = (int)Kp; a.lda = Kp; a.ldb = Kp;
    a.C = out; a.ldc = d; a.slab = 0;
    a.bias = b; a.x0 = x0; a.xres = x; a.p_out = p_out; a.ld_e = d;
    const int32_t st = launch_gemm<1>(a, 1, s);
    if (st != MH_OK) return st;
    MH_CHECK_LAUNCH("mh_cross_layer_fwd_split");
    return MH_OK;
}

// same phases as mh_cross_layer_bwd (selected by the non-NULL outputs); the element-wise phase is shared with it
int32_t mh_cross_layer_bwd_split(const float* x0, const float* x, const float* p, const float* dout, const float* W, int64_t M,
                                 int32_t d, float* g, float* dx0_acc, int32_t accumulate_dx0, float* dx, float* dW, float* db,
                                 void* workspace, int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(M >= 1 && d >= 8 && d % 4 == 0, "mh_cross_layer_bwd_split: d=%d must be a positive multiple of 4", d);
    MH_REQUIRE(g && dout, "mh_cross_layer_bwd_split: g and dout are required");
    hipStream_t s = mh_stream(stream);
    if (dx0_acc) {  // g = dout * x0, dx0_acc (+)= dout * p: the fp32 library's element-wise pass
        const int32_t st = mh_cross_layer_bwd(x0, nullptr, p, dout, nullptr, M, d, d, g, dx0_acc, accumulate_dx0, nullptr, nullptr, nullptr,
                                              nullptr, 0, stream);
        if (st != MH_OK) return st;
    }
    if (!dx && !dW) return MH_OK;
    MH_REQUIRE(workspace && workspace_bytes >= mh_cross_layer_split_workspace_bytes(M, d), "mh_cross_layer_bwd_split: workspace too small");
    char* wp = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t Kp = pad_to(d, 64), Mp = pad_to(M, 64);
    const SplitBuf sg = take_pair(wp, M, Kp), sw = take_pair(wp, d, Kp), sxt = take_pair(wp, d, Mp), sgt = take_pair(wp, d, Mp);
    float* slabs = reinterpret_cast<float*>(wp);
    wp += al256(8 * (int64_t)d * d * 4);
    float* dbp = reinterpret_cast<float*>(wp);
    if (dx) {
        MH_REQUIRE(W, "mh_cross_layer_bwd_split: W is required for dx");
        split_rows(g, M, d, d, sg, s);
        split_rows(W, d, d, d, sw, s);  // W [d_in, d_out] row-major = B^T [n = d_in][k = d_out]
        GsArgs a{};
        set_operands(a, sg, sw);
        a.M = M; a.N = d; a.Kp = (int)Kp; a.lda = Kp; a.ldb = Kp;
        a.C = dx; a.ldc = d; a.addend = dout; a.ld_add = d;
        const int32_t st = launch_gemm<0>(a, 1, s);
        if (st != MH_OK) return st;
    }
    if (dW) {
        MH_REQUIRE(x, "mh_cross_layer_bwd_split: x is required for dW");
        split_transpose(x, M, d, d, sxt, s);  // x^T [d_in, Mp]
        split_transpose(g, M, d, d, sgt, s);  // g^T [d_out, Mp]
        GsArgs a{};
        set_operands(a, sxt, sgt);
        a.M = d; a.N = d; a.Kp = (int)Mp; a.lda = Mp; a.ldb = Mp;
        // few output tiles (378 of 256 x 128 at d = 3344, 196 of 256 x 256), a long contraction: split it so that the grid fills the 256 CUs
        // about three times over
        const int64_t otiles = mh_ceil_div(d, 256) * mh_ceil_div(d, (gemm_geo() >= 2 && split_images() == 2) ? 256 : 128);
        int splits = (int)mh_ceil_div(3 * (int64_t)mh_num_cus(), otiles);
        if (splits > 8) splits = 8;
        if (splits > Mp / GBK / 32) splits = (int)(Mp / GBK / 32);
        if (splits < 1) splits = 1;
        a.C = splits > 1 ? slabs : dW; a.ldc = d; a.slab = (int64_t)d * d;
        const int32_t st = launch_gemm<0>(a, splits, s);
        if (st != MH_OK) return st;
        if (splits > 1) {
            const int64_t len4 = (int64_t)d * d / 4;
            MH_LAUNCH(gs_reduce_slabs_kernel, dim3((unsigned)mh_ceil_div(len4, 256)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(slabs), splits,
                      len4, reinterpret_cast<f32x4*>(dW));
        }
        if (db) {
            const int S = 64;
            const int rps = (int)mh_ceil_div(M, S);
            MH_LAUNCH(gs_colsum_kernel, dim3((unsigned)mh_ceil_div(d, 64), (unsigned)S), dim3(256), 0, s, (const float*)g, M, (int)d, (int64_t)d, rps, dbp);
            MH_LAUNCH(gs_colsum_finish_kernel, dim3((unsigned)mh_ceil_div(d, 256)), dim3(256), 0, s, (const float*)dbp, S, (int)d, db);
        }
    }
    MH_CHECK_LAUNCH("mh_cross_layer_bwd_split");
    return MH_OK;
}


// ---- Dense layer  y = act(x W + b)  (blocks/mlp.py:275-280) in the same arithmetic ---------------------------------------------
// dW slabs of the split-M form: at most 8 when the output has >= 32 tiles of 256 x 128 (the DCN-v2 3341 -> 512 layer: 56 tiles x 8
// splits), up to 64 for small outputs over a long batch (512 x 256 at M = 65 536: 4 tiles -- 8 splits were 32 workgroups on 256 CUs)
static int dw_split_cap(int K, int N) { return mh_ceil_div(K, 256) * mh_ceil_div(N, 128) >= 32 ? 8 : 64; }

int64_t mh_linear_split_workspace_bytes(int64_t M, int32_t K, int32_t N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    const int64_t Kp = pad_to(K, 64), Np = pad_to(N, 64), Mp = pad_to(M, 64);
    // forward: x, W^T;  backward: dz, W (dX), x^T, dz^T, eight dW slabs, db partials (dW)
    const int64_t fwd = pair_bytes(M, Kp) + pair_bytes(N, Kp);
    const int64_t bwd = pair_bytes(M, Np) + pair_bytes(K, Np) + pair_bytes(K, Mp) + pair_bytes(N, Mp) +
                        al256(dw_split_cap(K, N) * (int64_t)K * N * 4) + al256(64 * (int64_t)N * 4);
    return (fwd > bwd ? fwd : bwd) + 1024;
}

int32_t mh_linear_bias_act_fwd_split(const float* x, int64_t ldx, const float* W, const float* b, int64_t M, int32_t K, int32_t N,
                                     int32_t act, float* y, int64_t ldy, void* workspace, int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(x && W && y, "mh_linear_bias_act_fwd_split: null argument");
    MH_REQUIRE(M >= 0 && K >= 1 && N >= 1, "mh_linear_bias_act_fwd_split: bad shape M=%lld K=%d N=%d", (long long)M, K, N);
    MH_REQUIRE(ldx >= K && ldy >= N, "mh_linear_bias_act_fwd_split: leading dimension smaller than row");
    MH_REQUIRE(act >= MH_ACT_NONE && act <= MH_ACT_SIGMOID, "mh_linear_bias_act_fwd_split: bad activation %d", act);
    if (M == 0) return MH_OK;
    MH_REQUIRE(workspace && workspace_bytes >= mh_linear_split_workspace_bytes(M, K, N), "mh_linear_bias_act_fwd_split: workspace too small");
    hipStream_t s = mh_stream(stream);
    char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t Kp = pad_to(K, 64);
    const SplitBuf sx = take_pair(p, M, Kp), sw = take_pair(p, N, Kp);
    split_rows(x, M, K, ldx, sx, s);
    split_transpose(W, K, N, N, sw, s);  // W [K, N] -> W^T [n][k], pad columns zero
    GsArgs a{};
    set_operands(a, sx, sw);
    a.M = M; a.N = N; a.Kp = (int)Kp; a.lda = Kp; a.ldb = Kp;
    a.C = y; a.ldc = ldy; a.slab = 0;
    a.bias = b; a.act = act;
    const int32_t st = launch_gemm<2>(a, 1, s);
    if (st != MH_OK) return st;
    MH_CHECK_LAUNCH("mh_linear_bias_act_fwd_split");
    return MH_OK;
}

// same contract as mh_linear_bias_act_bwd (dy becomes dz = dy * act'(y) in place; dx masked by x_act; dW / db optional)
int32_t mh_linear_bias_act_bwd_split(const float* x, int64_t ldx, const float* W, const float* y, int64_t ldy, float* dy, int64_t lddy,
                                     int64_t M, int32_t K, int32_t N, int32_t act, int32_t x_act, float* dx, int64_t lddx, float* dW,
                                     float* db, void* workspace, int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(x && dy, "mh_linear_bias_act_bwd_split: null argument");
    MH_REQUIRE(!dx || W, "mh_linear_bias_act_bwd_split: W is required for dx");
    MH_REQUIRE(M >= 1 && K >= 1 && N >= 1, "mh_linear_bias_act_bwd_split: bad shape");
    MH_REQUIRE(dW || !db, "mh_linear_bias_act_bwd_split: db rides on the dW pass (pass dW too)");
    MH_REQUIRE(ldx >= K && lddy >= N && (!dx || lddx >= K), "mh_linear_bias_act_bwd_split: bad leading dimension");
    hipStream_t s = mh_stream(stream);
    if (act != MH_ACT_NONE) {  // dz = dy * act'(y) in place: the fp32 library's element-wise pass
        const int32_t st = mh_linear_bias_act_bwd(x, ldx, nullptr, y, ldy, dy, lddy, M, K, N, act, MH_ACT_NONE, nullptr, 0, nullptr, nullptr,
                                                  nullptr, 0, stream);
        if (st != MH_OK) return st;
    }
    if (!dx && !dW) return MH_OK;
    MH_REQUIRE(workspace && workspace_bytes >= mh_linear_split_workspace_bytes(M, K, N), "mh_linear_bias_act_bwd_split: workspace too small");
    char* wp = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t Np = pad_to(N, 64), Mp = pad_to(M, 64);
    const SplitBuf sz = take_pair(wp, M, Np), sw = take_pair(wp, K, Np), sxt = take_pair(wp, K, Mp), szt = take_pair(wp, N, Mp);
    float* slabs = reinterpret_cast<float*>(wp);
    wp += al256(dw_split_cap(K, N) * (int64_t)K * N * 4);
    float* dbp = reinterpret_cast<float*>(wp);
    if (dx) {
        split_rows(dy, M, N, lddy, sz, s);
        split_rows(W, K, N, N, sw, s);  // W [K, N] row-major = B^T [n = k_in][k = n_out]
        GsArgs a{};
        set_operands(a, sz, sw);
        a.M = M; a.N = K; a.Kp = (int)Np; a.lda = Np; a.ldb = Np;
        a.C = dx; a.ldc = lddx;
        const int32_t st = launch_gemm<0>(a, 1, s);
        if (st != MH_OK) return st;
        if (x_act != MH_ACT_NONE) {  // dx *= x_act'(x): x is the previous layer's activated output
            const int32_t st2 = mh_linear_bias_act_bwd(x, ldx, nullptr, x, ldx, dx, lddx, M, K, K, x_act, MH_ACT_NONE, nullptr, 0, nullptr,
                                                       nullptr, nullptr, 0, stream);
            if (st2 != MH_OK) return st2;
        }
    }
    if (dW) {
        split_transpose(x, M, K, ldx, sxt, s);    // x^T  [K, Mp]
        split_transpose(dy, M, N, lddy, szt, s);  // dz^T [N, Mp]
        GsArgs a{};
        set_operands(a, sxt, szt);
        a.M = K; a.N = N; a.Kp = (int)Mp; a.lda = Mp; a.ldb = Mp;
        const int64_t otiles = mh_ceil_div(K, 256) * mh_ceil_div(N, (gemm_geo() >= 2 && split_images() == 2) ? 256 : 128);
        int splits = (int)mh_ceil_div(3 * (int64_t)mh_num_cus(), otiles);
        if (splits > dw_split_cap(K, N)) splits = dw_split_cap(K, N);
        if (splits > Mp / GBK / 32) splits = (int)(Mp / GBK / 32);
        if (splits < 1) splits = 1;
        a.C = splits > 1 ? slabs : dW; a.ldc = N; a.slab = (int64_t)K * N;
        const int32_t st = launch_gemm<0>(a, splits, s);
        if (st != MH_OK) return st;
        if (splits > 1) {
            const int64_t len = (int64_t)K * N;
            if (len % 4 == 0) {
                MH_LAUNCH(gs_reduce_slabs_kernel, dim3((unsigned)mh_ceil_div(len / 4, 256)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(slabs),
                          splits, len / 4, reinterpret_cast<f32x4*>(dW));
            } else {
                MH_LAUNCH(gs_reduce_slabs_scalar_kernel, dim3((unsigned)mh_ceil_div(len, 256)), dim3(256), 0, s, (const float*)slabs, splits, len, dW);
            }
        }
        if (db) {
            const int S = 64;
            const int rps = (int)mh_ceil_div(M, S);
            MH_LAUNCH(gs_colsum_kernel, dim3((unsigned)mh_ceil_div(N, 64), (unsigned)S), dim3(256), 0, s, (const float*)dy, M, (int)N, lddy, rps, dbp);
            MH_LAUNCH(gs_colsum_finish_kernel, dim3((unsigned)mh_ceil_div(N, 256)), dim3(256), 0, s, (const float*)dbp, S, (int)N, db);
        }
    }
    MH_CHECK_LAUNCH("mh_linear_bias_act_bwd_split");
    return MH_OK;
}

}  // extern "C"
