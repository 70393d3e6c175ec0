// In-batch sampled-softmax scorer on the bf16 matrix pipe: the OPT-IN "bf16x3" arithmetic of mh_inbatch_softmax_fwd_dq / _bwd.
// Reference: the same functions as mh_scorer_stream.hip -- ContrastiveOutput.outputs (tf/outputs/contrastive.py:276-344),
// ItemRetrievalScorer.call_outputs (tf/blocks/retrieval/base.py:283-429), rescore_false_negatives (tf/utils/tf_utils.py:126-154),
// CategoricalCrossEntropy(from_logits=True) (tf/losses/listwise.py:38-52) and their gradients (tf/models/base.py:1121-1174).
//
// Why.  The fp32 kernels are MFMA-bound at 0.75-0.79 of the fp32 peak (157 TF); v_mfma_f32_32x32x16_bf16 runs at 16x that rate.
// Every fp32 operand is split once, x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-18 |x|, and every product
// of the two GEMMs of a pass is formed as  hi hi + hi lo + lo hi  in fp32 accumulators: three bf16 MFMAs per fp32-equivalent one.
// Dropped terms are <= 3 * 2^-18 |a b| per element; measured on L2-normalised rows: max |dot - dot64| = 2.3e-6 (the plain fp32
// fmaf chain: 2.5e-7), i.e. 4.6e-5 on a logit at 1/T = 20 -- inside north_star's 1e-4.  This is NOT bit-identical to the fp32
// path and therefore never the default: `mh_set_scorer_arith(1)` / MERLIN_HIP_SCORER_ARITH=bf16x3 turns it on, bench.py reports
// it under its own dtype label, and the parity suite runs under both settings.
//
// Structure (the row-stationary streaming core of mh_scorer_stream.hip, re-tiled for the 32x32x16 bf16 MFMA):
//   * a workgroup (4 wavefronts, ONE per SIMD: 512 registers each) owns 256 rows of the stationary matrix X; a wavefront keeps
//     its 64 rows (two 32-row blocks) as B-operand fragments, hi and lo, in 128 registers for the whole kernel;
//   * the streamed matrix Y arrives in 64-row tiles by direct-to-LDS DMA, double-buffered, one barrier per tile, in TWO images:
//     row-major [row][e] (hi, lo) for GEMM 1 and TRANSPOSED [e][row] (hi, lo) for GEMM 2 -- the transposed copy of the whole
//     matrix is made once per call by split_prepare_kernel (a bf16 MFMA operand is 8 consecutive k per lane: GEMM 2 contracts
//     over the streamed rows, so its A operand wants 8 rows of ONE column);
//   * GEMM 1 per 32-row unit: S^T[j, x] = sum_e Y[j, e] X[x, e], 24 MFMAs per x-block.  In the C layout a lane holds ONE
//     stationary row and 16 streamed rows: masks, temperature, the (lazy) online max and exp2 are per-lane loops;
//   * GEMM 2: O^T[e, x] += sum_j Y^T[e, j] P^T[j, x].  The 16 probabilities a lane holds ARE its B operand (split into hi / lo in
//     registers): MFMA k-slot (step s, half h, slot i) is defined to mean streamed row 16 s + 8 (i >> 2) + 4 h + (i & 3), and the
//     A operand is read from the transposed image in that order (two 8-byte LDS reads per fragment): no shuffle, no transpose.
//   * partial results per candidate split in the layout of the fp32 kernels (part_m / part_s / opart): the combine kernels of
//     mh_scorer_stream.hip finish the pass unchanged.
#include "mh_common.h"

#include <math.h>
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int PE = 128;            // embedding width (the whole K of GEMM 1)
constexpr int PKS = PE / 16;       // k-steps of GEMM 1
constexpr int PXB = 256;           // stationary rows per workgroup
constexpr int PBN = 64;            // streamed rows per tile (two 32-row units)
constexpr int P_ARR = PBN * PE * 2;                // one of {hi, lo} of either image: 16 KB
constexpr int P_STAGE = 4 * P_ARR;                 // Y hi, Y lo, Y^T hi, Y^T lo: 64 KB
constexpr int P_AUX = 2 * P_STAGE;                 // ids (2 x 512 B), lse (2 x 256 B) behind the two stages
constexpr int P_LDS = P_AUX + 2 * 512 + 2 * 256;
constexpr float P_LOG2E = 1.4426950408889634f;
constexpr float P_NEG_BIG = -1.0e30f;
constexpr float P_LAZY = 16.f;

enum { PM_FWD = 0, PM_GRAD = 1, PM_FWD_GRAD = 2 };  // PM_FWD: loss / lse only (no second GEMM, no transposed image)

__device__ __forceinline__ uint16_t p_bf16(float x) {  // round to nearest even (finite inputs; inf / nan keep their class)
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float p_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// x[N, 128] fp32 -> hi, lo [N, 128] (bf16 bit patterns) and hiT, loT [128, ldT] (ldT = N rounded up to 64; the columns past N
// are written as zeros).  One workgroup per 64 rows; the transpose goes through LDS.
__global__ __launch_bounds__(256) void split_prepare_kernel(const float* __restrict__ x, int64_t N, uint16_t* __restrict__ hi,
                                                           uint16_t* __restrict__ lo, uint16_t* __restrict__ hiT,
                                                           uint16_t* __restrict__ loT, int64_t ldT) {
    __shared__ uint16_t sh[64][PE + 2], sl[64][PE + 2];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * (PE / 4); i += 256) {
        const int r = i / (PE / 4), c4 = i % (PE / 4);
        const int64_t row = r0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < N) v = *reinterpret_cast<const f32x4*>(x + row * PE + c4 * 4);
        uint16_t h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
            uint32_t wh2, wl2;
            mh_split_pair(v[k], v[k + 1], wh2, wl2);
            h[k] = (uint16_t)wh2; h[k + 1] = (uint16_t)(wh2 >> 16);
            l[k] = (uint16_t)wl2; l[k + 1] = (uint16_t)(wl2 >> 16);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sh[r][c4 * 4 + k] = h[k];
            sl[r][c4 * 4 + k] = l[k];
        }
        if (row < N) {
            *reinterpret_cast<uint2*>(hi + row * PE + c4 * 4) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
            *reinterpret_cast<uint2*>(lo + row * PE + c4 * 4) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
        }
    }
    __syncthreads();
    if (!hiT) return;
    // thread (e, half): 32 consecutive rows of column e -> 64 contiguous bytes of the transposed arrays
    const int e = threadIdx.x & (PE - 1), half = threadIdx.x >> 7;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint32_t wh[4], wl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = half * 32 + g * 8 + 2 * k;
            wh[k] = (uint32_t)sh[r][e] | ((uint32_t)sh[r + 1][e] << 16);
            wl[k] = (uint32_t)sl[r][e] | ((uint32_t)sl[r + 1][e] << 16);
        }
        const int64_t col = r0 + half * 32 + g * 8;
        *reinterpret_cast<uint4*>(hiT + (int64_t)e * ldT + col) = make_uint4(wh[0], wh[1], wh[2], wh[3]);
        *reinterpret_cast<uint4*>(loT + (int64_t)e * ldT + col) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
    }
}

struct SplitArgs {
    const uint16_t *xhi, *xlo;               // stationary [Nx, 128]
    const uint16_t *yhi, *ylo;               // streamed   [Ny, 128]
    const uint16_t *ythi, *ytlo;             // streamed, transposed [128, ldT]
    int64_t Nx, Ny, ldT;
    const void *x_ids, *y_ids;
    const float* lse;   // GRAD: natural-log lse of the softmax rows (stationary side, or streamed side if LSE_STREAM)
    const float* pos;   // FWD_GRAD: positive scores [Nx]
    float invT, fns, gscale;
    float *part_m, *part_s, *opart;
    int tiles_per_split;
};

__device__ __forceinline__ void p_dma16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
__device__ __forceinline__ void p_dma4(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 4, 0, 0);
}
__device__ __forceinline__ f32x16 p_mfma(bf16x8_t a, bf16x8_t b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// XT = 32-row blocks of X per wavefront: 2 -> 4 wavefronts per workgroup, one per SIMD with 512 registers; 1 -> 8 wavefronts, two
// per SIMD with 256 registers each (the epilogue of one hides behind the MFMAs of the other; twice the LDS reads per MFMA).
// Measured and NOT kept: the two units of a tile as one software-pipelined block (G1(0) | G1(1) + E(0) | G2(0) + E(1) | G2(1)) in
// the gradient mode: 5.49-5.51 ms against 5.48-5.54 for this loop -- with two wavefronts per SIMD the overlap is already there.
// What did move the kernel: the probabilities are split into bf16 hi / lo by mh_split_pair (v_cvt_pk_bf16_f32: 5 instructions per
// pair instead of ~30 of bit arithmetic): forward + dq 6.8 -> 5.6 ms, gradient pass 6.7 -> 5.5 ms at 65 536 x 65 536 x 128.
template <int MODE, typename IdT, bool HAS_IDS, bool LSE_STREAM, int XT>
__global__ __launch_bounds__(512 / XT, 1) void stream_split_kernel(const SplitArgs a) {
    constexpr int IDW = sizeof(IdT) / 4;
    constexpr int NW = 8 / XT;  // wavefronts per workgroup
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int64_t x0 = (int64_t)blockIdx.x * PXB + wave * 32 * XT;
    const int split = blockIdx.y;
    const int nt_all = (int)((a.Ny + PBN - 1) / PBN);
    const int t_beg = split * a.tiles_per_split;
    const int t_end = (t_beg + a.tiles_per_split < nt_all) ? t_beg + a.tiles_per_split : nt_all;

    // one tile: 4 arrays x 1024 chunks of 16 bytes; chunk position L of an array <- a swizzled source chunk
    auto issue = [&](int t, int stage) {
        unsigned char* st = smem + stage * P_STAGE;
        const int64_t row0 = (int64_t)t * PBN;
#pragma unroll
        for (int j = 0; j < 32 / NW; ++j) {  // row-major image: position (r, p) holds chunk p ^ (r & 15) of row r (rows clamped)
            const int L = (j * NW + wave) * 64 + lane;
            const int arr = L >> 10, Lp = L & 1023, r = Lp >> 4, p = Lp & 15, c = p ^ (r & 15);
            int64_t row = row0 + r;
            if (row > a.Ny - 1) row = a.Ny - 1;
            p_dma16((arr ? a.ylo : a.yhi) + row * PE + c * 8, st + (j * NW + wave) * 1024);
        }
        if (MODE != PM_FWD)
#pragma unroll
        for (int j = 0; j < 32 / NW; ++j) {  // transposed image: position (e, p) holds chunk p ^ ((e >> 1) & 7) of row e (8 rows of Y each)
            const int L = (j * NW + wave) * 64 + lane;
            const int arr = L >> 10, Lp = L & 1023, e = Lp >> 3, p = Lp & 7, c = p ^ ((e >> 1) & 7);
            p_dma16((arr ? a.ytlo : a.ythi) + (int64_t)e * a.ldT + row0 + c * 8, st + 2 * P_ARR + (j * NW + wave) * 1024);
        }
        if (HAS_IDS && wave < IDW) {  // 64 ids = IDW wave-instructions of 64 words
            int64_t w = row0 * IDW + wave * 64 + lane;
            const int64_t last = a.Ny * IDW - 1;
            if (w > last) w = last;
            p_dma4(static_cast<const uint32_t*>(a.y_ids) + w, smem + P_AUX + stage * 512 + wave * 256);
        }
        if (MODE == PM_GRAD && LSE_STREAM && wave == NW - 1) {
            int64_t w = row0 + lane;
            if (w > a.Ny - 1) w = a.Ny - 1;
            p_dma4(a.lse + w, smem + P_AUX + 1024 + stage * 256);
        }
    };
    if (t_beg < t_end) issue(t_beg, 0);

    // stationary fragments (B operand of GEMM 1: lane = column l31 of its 32-row block, k = 16 ks + 8 h .. + 7)
    bf16x8_t xh[XT][PKS], xl[XT][PKS];
    bool xvalid[XT];
    IdT x_id[XT];
    float lse2_x[XT], m_run[XT], s_run[XT];
#pragma unroll
    for (int tn = 0; tn < XT; ++tn) {
        int64_t xrow = x0 + tn * 32 + l31;
        xvalid[tn] = xrow < a.Nx;
        if (!xvalid[tn]) xrow = a.Nx - 1;
#pragma unroll
        for (int ks = 0; ks < PKS; ++ks) {
            xh[tn][ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a.xhi + xrow * PE + ks * 16 + h * 8));
            xl[tn][ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a.xlo + xrow * PE + ks * 16 + h * 8));
        }
        x_id[tn] = 0;
        if (HAS_IDS) x_id[tn] = static_cast<const IdT*>(a.x_ids)[xrow];
        lse2_x[tn] = 0.f;
        if (MODE == PM_GRAD && !LSE_STREAM) lse2_x[tn] = a.lse[xrow] * P_LOG2E;
        m_run[tn] = (MODE != PM_GRAD) ? a.pos[xrow] * a.invT * P_LOG2E : P_NEG_BIG;  // the reference max starts at the positive logit
        s_run[tn] = 0.f;
    }
    const float scale2 = a.invT * P_LOG2E;
    constexpr int OB = (MODE == PM_FWD) ? 1 : 4;  // forward-only: no output accumulators (one dummy block keeps the code uniform)
    f32x16 o[OB][XT];  // O^T: block eb of 32 columns e x block tn of 32 stationary rows; lane: row x = l31, e = (i & 3) + 8 (i >> 2) + 4 h
#pragma unroll
    for (int eb = 0; eb < OB; ++eb)
#pragma unroll
        for (int tn = 0; tn < XT; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[eb][tn][i] = 0.f;

    __syncthreads();  // vmcnt(0) + barrier: tile t_beg has landed for every wavefront

    for (int t = t_beg; t < t_end; ++t) {
        const int odd = (t - t_beg) & 1;
        if (t + 1 < t_end) issue(t + 1, odd ^ 1);
        const unsigned char* st = smem + odd * P_STAGE;
        const int64_t j_tile = (int64_t)t * PBN;
        const int nvalid = (j_tile + PBN <= a.Ny) ? PBN : (int)(a.Ny - j_tile);
        const IdT* ids = reinterpret_cast<const IdT*>(smem + P_AUX + odd * 512);
        const float* lsej = reinterpret_cast<const float*>(smem + P_AUX + 1024 + odd * 256);
#pragma unroll 1
        for (int u = 0; u < 2; ++u) {
            if (u * 32 >= nvalid) break;
            // ---- GEMM 1 on the unit's 32 streamed rows, software-pipelined over the 8 k-steps ---------------------------------
            f32x16 acc[XT];
#pragma unroll
            for (int tn = 0; tn < XT; ++tn)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[tn][i] = 0.f;
            const int rd = (u * 32 + l31) * 256;
            auto frag = [&](int ks, int arr) {
                const int pos = ((2 * ks + h) ^ (l31 & 15)) * 16;
                return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(st + arr * P_ARR + rd + pos));
            };
            bf16x8_t ah = frag(0, 0), al = frag(0, 1);
#pragma unroll
            for (int ks = 0; ks < PKS; ++ks) {
                bf16x8_t nh = ah, nl = al;
                if (ks + 1 < PKS) {
                    nh = frag(ks + 1, 0);
                    nl = frag(ks + 1, 1);
                }
#pragma unroll
                for (int tn = 0; tn < XT; ++tn) acc[tn] = p_mfma(al, xh[tn][ks], acc[tn]);  // small terms first
#pragma unroll
                for (int tn = 0; tn < XT; ++tn) acc[tn] = p_mfma(ah, xl[tn][ks], acc[tn]);
#pragma unroll
                for (int tn = 0; tn < XT; ++tn) acc[tn] = p_mfma(ah, xh[tn][ks], acc[tn]);
                ah = nh;
                al = nl;
            }
            // ---- epilogue: lane = stationary row tn * 32 + l31, streamed rows jl(i) = u * 32 + (i >> 2) * 8 + 4 h + (i & 3) ----------
            bf16x8_t ph[XT][2], pl[XT][2];  // [tn][k-step]: the probabilities as the B operand of GEMM 2, hi and lo
            const int jl0 = u * 32 + 4 * h;
#pragma unroll
            for (int tn = 0; tn < XT; ++tn) {
                // the 16 scores of this lane are turned into base-2 logits and then into probabilities IN PLACE (acc[tn]); the mask
                // of rescored false negatives is one bit per score
                unsigned mbits = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int jl = jl0 + (i >> 2) * 8 + (i & 3);
                    bool masked = false;
                    if (HAS_IDS) masked = (ids[jl] == x_id[tn]);
                    mbits |= (masked ? 1u : 0u) << i;
                    float v2 = (masked ? a.fns : acc[tn][i]) * scale2;
                    if (nvalid < PBN && jl >= nvalid) v2 = -INFINITY;  // only the last tile of Y can be partial
                    acc[tn][i] = v2;
                }
                if (MODE == PM_GRAD) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float l2 = LSE_STREAM ? lsej[jl0 + (i >> 2) * 8 + (i & 3)] * P_LOG2E : lse2_x[tn];
                        const float e = __builtin_amdgcn_exp2f(acc[tn][i] - l2) * a.gscale;  // -inf on invalid rows -> 0
                        acc[tn][i] = ((mbits >> i) & 1u) ? 0.f : e;
                    }
                } else {  // FWD / FWD_GRAD: lazy reference max shared by the two lanes of a row
                    float tmax = acc[tn][0];
#pragma unroll
                    for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, acc[tn][i]);
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                    if (__any(tmax > m_run[tn] + P_LAZY)) {
                        const float m_new = (tmax > m_run[tn] + P_LAZY) ? tmax : m_run[tn];
                        const float f = __builtin_amdgcn_exp2f(m_run[tn] - m_new);
                        s_run[tn] *= f;
                        m_run[tn] = m_new;
                        if (MODE != PM_FWD)
#pragma unroll
                        for (int eb = 0; eb < OB; ++eb)
#pragma unroll
                            for (int i = 0; i < 16; ++i) o[eb][tn][i] *= f;  // every accumulator element of this lane belongs to its row
                    }
                    float s_add = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float e = __builtin_amdgcn_exp2f(acc[tn][i] - m_run[tn]);
                        s_add += e;  // rescored false negatives stay in the denominator
                        acc[tn][i] = ((mbits >> i) & 1u) ? 0.f : e;
                    }
                    s_run[tn] += s_add;
                }
                if (MODE != PM_FWD)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    uint32_t wh[4], wl[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        mh_split_pair(acc[tn][8 * s + 2 * k], acc[tn][8 * s + 2 * k + 1], wh[k], wl[k]);
                    }
                    ph[tn][s] = __builtin_bit_cast(bf16x8_t, make_uint4(wh[0], wh[1], wh[2], wh[3]));
                    pl[tn][s] = __builtin_bit_cast(bf16x8_t, make_uint4(wl[0], wl[1], wl[2], wl[3]));
                }
            }
            // ---- GEMM 2: O^T[e, x] += sum_j Y^T[e, j] P^T[j, x]; A = two 8-byte pieces of row e of the transposed image ----------
            const unsigned char* yt = st + 2 * P_ARR;
            if (MODE != PM_FWD)
#pragma unroll
            for (int eb = 0; eb < OB; ++eb) {
                const int e = eb * 32 + l31, sw = (e >> 1) & 7;
                const unsigned char* row_h = yt + e * 128 + 8 * h;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int c0 = 4 * u + 2 * s;
                    const uint2 h0 = *reinterpret_cast<const uint2*>(row_h + ((c0 ^ sw) << 4));
                    const uint2 h1 = *reinterpret_cast<const uint2*>(row_h + (((c0 + 1) ^ sw) << 4));
                    const uint2 l0 = *reinterpret_cast<const uint2*>(row_h + P_ARR + ((c0 ^ sw) << 4));
                    const uint2 l1 = *reinterpret_cast<const uint2*>(row_h + P_ARR + (((c0 + 1) ^ sw) << 4));
                    const bf16x8_t ath = __builtin_bit_cast(bf16x8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
                    const bf16x8_t atl = __builtin_bit_cast(bf16x8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
#pragma unroll
                    for (int tn = 0; tn < XT; ++tn) o[eb][tn] = p_mfma(atl, ph[tn][s], o[eb][tn]);
#pragma unroll
                    for (int tn = 0; tn < XT; ++tn) o[eb][tn] = p_mfma(ath, pl[tn][s], o[eb][tn]);
#pragma unroll
                    for (int tn = 0; tn < XT; ++tn) o[eb][tn] = p_mfma(ath, ph[tn][s], o[eb][tn]);
                }
            }
        }
        __syncthreads();  // every wavefront is done with this tile; the next one (vmcnt(0)) has landed
    }

    // ---- results: the partial layouts of mh_scorer_stream.hip ---------------------------------------------------------------------
    if (MODE != PM_GRAD) {
#pragma unroll
        for (int tn = 0; tn < XT; ++tn) {
            const float ss = s_run[tn] + __shfl_xor(s_run[tn], 32);  // the two lanes of a row share m_run
            if (h == 0 && xvalid[tn]) {
                a.part_m[(int64_t)split * a.Nx + x0 + tn * 32 + l31] = m_run[tn];
                a.part_s[(int64_t)split * a.Nx + x0 + tn * 32 + l31] = ss;
            }
        }
    }
    if (MODE == PM_FWD) return;
    float* op = a.opart + (int64_t)split * a.Nx * PE;
#pragma unroll
    for (int tn = 0; tn < XT; ++tn) {
        if (!xvalid[tn]) continue;
        float* orow = op + (x0 + tn * 32 + l31) * PE;
#pragma unroll
        for (int eb = 0; eb < OB; ++eb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {o[eb][tn][4 * g], o[eb][tn][4 * g + 1], o[eb][tn][4 * g + 2], o[eb][tn][4 * g + 3]};
                *reinterpret_cast<f32x4*>(orow + eb * 32 + 8 * g + 4 * h) = v;
            }
    }
}

template <int MODE, bool LSE_STREAM, int XT>
int32_t launch_split_mode(const SplitArgs& a, int ids_dtype, dim3 grid, hipStream_t s) {
#define MH_LAUNCH_SPLIT(IdT, HAS)                                                                                          \
    do {                                                                                                                   \
        auto kern = stream_split_kernel<MODE, IdT, HAS, LSE_STREAM, XT>;                                                   \
        static bool attr_done = false;                                                                                     \
        if (!attr_done) {                                                                                                  \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                    P_LDS) != hipSuccess) {                                                                \
                mh_set_error("scorer (bf16x3): cannot raise the dynamic LDS limit");                                       \
                return MH_ERR_LAUNCH;                                                                                      \
            }                                                                                                              \
            attr_done = true;                                                                                              \
        }                                                                                                                  \
        MH_LAUNCH(kern, grid, dim3(512 / XT), (size_t)P_LDS, s, a);                                                        \
    } while (0)
    if (!a.x_ids) MH_LAUNCH_SPLIT(int32_t, false);
    else if (ids_dtype == MH_I32) MH_LAUNCH_SPLIT(int32_t, true);
    else MH_LAUNCH_SPLIT(int64_t, true);
#undef MH_LAUNCH_SPLIT
    return MH_OK;
}

}  // namespace

// ---- internal interface used by mh_scorer.hip ------------------------------------------------------------------------------------
// Bytes of the split of ONE [N, 128] matrix: hi, lo [N, 128] and hiT, loT [128, ldT] (all bf16), 256-byte aligned parts.
int64_t mh_split_matrix_bytes(int64_t N) {
    const int64_t ldT = (N + 63) / 64 * 64;
    auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
    return 2 * al(N * PE * 2) + 2 * al(PE * ldT * 2);
}

struct MhSplitMatrix {
    uint16_t *hi, *lo, *hiT, *loT;
    int64_t ldT;
};

MhSplitMatrix mh_split_prepare(const float* x, int64_t N, void* buf, hipStream_t s) {
    MhSplitMatrix m;
    auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
    char* p = static_cast<char*>(buf);
    m.ldT = (N + 63) / 64 * 64;
    m.hi = reinterpret_cast<uint16_t*>(p);
    p += al(N * PE * 2);
    m.lo = reinterpret_cast<uint16_t*>(p);
    p += al(N * PE * 2);
    m.hiT = reinterpret_cast<uint16_t*>(p);
    p += al(PE * m.ldT * 2);
    m.loT = reinterpret_cast<uint16_t*>(p);
    MH_LAUNCH(split_prepare_kernel, dim3((unsigned)mh_ceil_div(N, 64)), dim3(256), 0, s, x, N, m.hi, m.lo, m.hiT, m.loT, m.ldT);
    return m;
}

// number of candidate splits (<= the fp32 plan's for the same shapes: the partial buffers are sized by that one)
int mh_split_plan(int64_t Nx, int64_t Ny, int* tiles_per_split) {
    const int row_tiles = (int)mh_ceil_div(Nx, PXB);
    const int nt = (int)mh_ceil_div(Ny, PBN);
    int want = mh_num_cus() / row_tiles;
    if (want < 1) want = 1;
    if (want > nt) want = nt;
    const int tps = (int)mh_ceil_div(nt, want);
    *tiles_per_split = tps;
    return (int)mh_ceil_div(nt, tps);
}

// mode: 0 = FWD (part_m / part_s only), 1 = GRAD (p = exp(z - lse) g), 2 = FWD_GRAD (online max, part_m / part_s, opart); same outputs
// as mh_stream_launch
int32_t mh_stream_split_launch(int mode, int lse_stream, const MhSplitMatrix& X, int64_t Nx, const MhSplitMatrix& Y, int64_t Ny,
                               const void* x_ids, const void* y_ids, int ids_dtype, const float* lse, const float* pos, float invT,
                               float fns, float gscale, float* part_m, float* part_s, float* opart, hipStream_t s) {
    SplitArgs a;
    a.xhi = X.hi; a.xlo = X.lo; a.yhi = Y.hi; a.ylo = Y.lo; a.ythi = Y.hiT; a.ytlo = Y.loT;
    a.Nx = Nx; a.Ny = Ny; a.ldT = Y.ldT; a.x_ids = x_ids; a.y_ids = y_ids; a.lse = lse; a.pos = pos;
    a.invT = invT; a.fns = fns; a.gscale = gscale; a.part_m = part_m; a.part_s = part_s; a.opart = opart;
    int tps = 1;
    const int nsplit = mh_split_plan(Nx, Ny, &tps);
    a.tiles_per_split = tps;

    dim3 grid((unsigned)mh_ceil_div(Nx, PXB), (unsigned)nsplit);
    // MERLIN_HIP_SCORER_XT = 1 | 2 (experiments): 32-row blocks of X per wavefront (see the kernel)
    static int xt = -1;
    if (xt < 0) {
        const char* e = MH_LAB_ENV("MERLIN_HIP_SCORER_XT");
        xt = (e && atoi(e) == 2) ? 2 : 1;
    }
    if (mode == PM_FWD) return launch_split_mode<PM_FWD, false, 1>(a, ids_dtype, grid, s);
    if (xt == 2) {
        if (mode == PM_FWD_GRAD) return launch_split_mode<PM_FWD_GRAD, false, 2>(a, ids_dtype, grid, s);
        if (lse_stream) return launch_split_mode<PM_GRAD, true, 2>(a, ids_dtype, grid, s);
        return launch_split_mode<PM_GRAD, false, 2>(a, ids_dtype, grid, s);
    }
    if (mode == PM_FWD_GRAD) return launch_split_mode<PM_FWD_GRAD, false, 1>(a, ids_dtype, grid, s);
    if (lse_stream) return launch_split_mode<PM_GRAD, true, 1>(a, ids_dtype, grid, s);
    return launch_split_mode<PM_GRAD, false, 1>(a, ids_dtype, grid, s);
}
