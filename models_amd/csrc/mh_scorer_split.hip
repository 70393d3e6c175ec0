// In-batch sampled-softmax scorer on the bf16 matrix pipe: the six-term "bf16x6" (fp32-grade, the host side's default) and the OPT-IN
// three-term "bf16x3" arithmetic of mh_inbatch_softmax_fwd_dq / _bwd.
// Reference: the same functions as mh_scorer_stream.hip -- ContrastiveOutput.outputs (tf/outputs/contrastive.py:276-344),
// ItemRetrievalScorer.call_outputs (tf/blocks/retrieval/base.py:283-429), rescore_false_negatives (tf/utils/tf_utils.py:126-154),
// CategoricalCrossEntropy(from_logits=True) (tf/losses/listwise.py:38-52) and their gradients (tf/models/base.py:1121-1174).
//
// Why.  The fp32 kernels are MFMA-bound at 0.75-0.79 of the fp32 peak (157 TF); v_mfma_f32_32x32x16_bf16 runs at 16x that rate.
// Every fp32 operand is split once, x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-18 |x|, and every product
// of the two GEMMs of a pass is formed as  hi hi + hi lo + lo hi  in fp32 accumulators: three bf16 MFMAs per fp32-equivalent one.
// Dropped terms are <= 3 * 2^-18 |a b| per element; measured on L2-normalised rows: max |dot - dot64| = 2.3e-6 (the plain fp32
// fmaf chain: 2.5e-7), i.e. 4.6e-5 on a logit at 1/T = 20 -- inside north_star's 1e-4.  This three-term form is NOT fp32-grade
// and therefore never the default: `mh_set_scorer_arith(1)` / MERLIN_HIP_SCORER_ARITH=bf16x3 turns it on, bench.py reports
// it under its own dtype label, and the parity suite runs under every setting.
//
// Structure (the row-stationary streaming core of mh_scorer_stream.hip, re-tiled for the 32x32x16 bf16 MFMA):
//   * a workgroup owns 256 rows of the stationary matrix X: 8 wavefronts (two per SIMD, 256 registers each) with one 32-row block
//     each (XT = 1, the shipped form; XT = 2 -- 4 wavefronts with two blocks each -- is a lab build's experiment); a wavefront keeps its
//     rows as B-operand fragments, all images, in registers for the whole kernel;
//   * the streamed matrix Y arrives in tiles by direct-to-LDS DMA through a ring, one barrier per tile, in TWO orientations:
//     row-major [row][e] for GEMM 1 and TRANSPOSED [e][row] for GEMM 2 -- the transposed copy of the whole matrix is made once per
//     call by split_prepare_kernel (a bf16 MFMA operand is 8 consecutive k per lane: GEMM 2 contracts over the streamed rows, so its
//     A operand wants 8 rows of ONE column);
//   * GEMM 1 per 32-row unit: S^T[j, x] = sum_e Y[j, e] X[x, e].  In the C layout a lane holds ONE stationary row and 16 streamed
//     rows: masks, logQ corrections, temperature, the (lazy) online max and exp2 are per-lane loops; rescored false negatives and the
//     rows past the end of Y are handled under wave-uniform branches (rare);
//   * GEMM 2: O^T[e, x] += sum_j Y^T[e, j] P^T[j, x].  The 16 probabilities a lane holds ARE its B operand (split into bf16 pieces in
//     registers): MFMA k-slot (step s, half h, slot i) is defined to mean streamed row 16 s + 8 (i >> 2) + 4 h + (i & 3), and the
//     transposed images store every 16-row group in that order (split_prepare_kernel), so the A operand of a k-step is ONE 16-byte
//     LDS read: no shuffle, no transpose, no register moves;
//   * partial results per candidate split in the layout of the fp32 kernels (part_m / part_s / opart): the combine kernels of
//     mh_scorer_stream.hip finish the pass unchanged.
//
// bf16x3 (NIMG = 2): 64-row tiles (two units), two stages of 64 KB, E = 128 without logQ corrections.
//
// bf16x6 (NIMG = 3).  x = h + m + l, three bf16 pieces that hold the 24-bit significand exactly, and every product of BOTH GEMMs as
// h h + h m + m h + h l + l h + m m: the dropped terms are <= 2^-25 of the product -- half an fp32 rounding, so the result is as close to
// the real dot product as the fp32 fmaf chain is (the argument and the float64 test are mh_tower_split.hip's).  Six MFMAs per
// fp32-equivalent one = 16 / 6 of the fp32 MFMA rate.  Differences from the three-term kernel: three images of every matrix (and of
// the probabilities, split in registers by mh_split3_pair); 32-row tiles of the streamed matrix in a THREE-stage ring (3 x 48 KB at
// E = 128) with the second wavefront of every SIMD one phase behind (rotated barrier, see the kernel); the six terms of GEMM 1 go to
// TWO accumulators alternately and GEMM 2 walks two 32-column blocks of O^T at a time (independent MFMA chains); the embedding width
// is a template parameter (128 or 64); the logQ sampling corrections are applied in the epilogue (HAS_CORR instantiations).
#include "mh_common.h"

#include <math.h>
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// The embedding width EW (the whole K of GEMM 1) is a template parameter: 128 (both arithmetics) or 64 (six-term kernel only)
constexpr int PXB = 256;           // stationary rows per workgroup
// Geometry by the number of bf16 images per matrix: 2 (bf16x3) -> 64 streamed rows per tile, 3 (bf16x6) -> 32
template <int NIMG, int EW = 128>
struct PG {
    static constexpr int BN = (NIMG == 3) ? 32 : 64;  // streamed rows per tile (units of 32 rows)
    static constexpr int ARR = BN * EW * 2;            // one image of either orientation: 16 KB / 8 KB (EW = 128)
    static constexpr int STAGE = 2 * NIMG * ARR;       // row-major images, then the transposed ones: 64 KB / 48 KB
    static constexpr int NST = (NIMG == 3) ? 3 : 2;    // stages of the ring (three: the late wavefronts work one tile behind, see the kernel)
    static constexpr int AUX = NST * STAGE;            // ids (NST x 512 B), lse (NST x 256 B), logQ corrections (NST x 256 B) behind the stages
    static constexpr int LDS = AUX + NST * 512 + NST * 256 + NST * 256;
};
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
// are written as zeros).  One workgroup per 64 rows; the transpose goes through LDS.  THREE: also the middle piece (mid, midT) and
// lo = the third piece of mh_split3_pair (bf16x6).
template <bool THREE, int EW>
__global__ __launch_bounds__(256) void split_prepare_kernel(const float* __restrict__ x, int64_t N, uint16_t* __restrict__ hi,
                                                           uint16_t* __restrict__ mid, uint16_t* __restrict__ lo,
                                                           uint16_t* __restrict__ hiT, uint16_t* __restrict__ midT,
                                                           uint16_t* __restrict__ loT, int64_t ldT) {
    constexpr int NI = THREE ? 3 : 2;
    __shared__ uint16_t sp[NI][64][EW + 2];
    uint16_t* const img[3] = {hi, THREE ? mid : lo, lo};
    uint16_t* const imgT[3] = {hiT, THREE ? midT : loT, loT};
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * (EW / 4); i += 256) {
        const int r = i / (EW / 4), c4 = i % (EW / 4);
        const int64_t row = r0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < N) v = *reinterpret_cast<const f32x4*>(x + row * EW + c4 * 4);
        uint32_t w[NI][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (THREE) mh_split3_pair(v[2 * k], v[2 * k + 1], w[0][k], w[1][k], w[NI - 1][k]);
            else mh_split_pair(v[2 * k], v[2 * k + 1], w[0][k], w[1][k]);
        }
#pragma unroll
        for (int a = 0; a < NI; ++a) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                sp[a][r][c4 * 4 + 2 * k] = (uint16_t)w[a][k];
                sp[a][r][c4 * 4 + 2 * k + 1] = (uint16_t)(w[a][k] >> 16);
            }
            if (row < N) *reinterpret_cast<uint2*>(img[a] + row * EW + c4 * 4) = make_uint2(w[a][0], w[a][1]);
        }
    }
    __syncthreads();
    if (!hiT) return;
    // thread (e, part): RP consecutive rows of column e -> 2 RP contiguous bytes of the transposed arrays.  Inside every group of 16
    // rows the order is 0-3, 8-11, 4-7, 12-15 (bits 2 and 3 of the row swapped): the 8 rows a lane of GEMM 2 needs for one k-step --
    // streamed rows 8 (i >> 2) + 4 h + (i & 3) of the group, the order in which it holds its probabilities -- are then 16
    // CONTIGUOUS bytes (one ds_read_b128 = the MFMA operand, no register moves)
    constexpr int NPART = 256 / EW, RP = 64 / NPART;  // EW = 128: two parts of 32 rows; 64: four parts of 16
    const int e = threadIdx.x & (EW - 1), part = threadIdx.x / EW;
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int g = 0; g < RP / 8; ++g) {
            uint32_t wv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = 2 * k;  // stored position g * 8 + j of the part <- row (g >> 1) * 16 + 8 (j >> 2) + 4 (g & 1) + (j & 3)
                const int r = part * RP + (g >> 1) * 16 + 8 * (j >> 2) + 4 * (g & 1) + (j & 3);
                wv[k] = (uint32_t)sp[a][r][e] | ((uint32_t)sp[a][r + 1][e] << 16);
            }
            const int64_t col = r0 + part * RP + g * 8;
            *reinterpret_cast<uint4*>(imgT[a] + (int64_t)e * ldT + col) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        }
}

struct SplitArgs {
    const uint16_t* x[3];                    // stationary [Nx, 128]: hi, (mid,) lo
    const uint16_t* y[3];                    // streamed   [Ny, 128]
    const uint16_t* yt[3];                   // streamed, transposed [128, ldT]
    int64_t Nx, Ny, ldT;
    const void *x_ids, *y_ids;
    const float* lse;   // GRAD: natural-log lse of the softmax rows (stationary side, or streamed side if LSE_STREAM)
    const float* pos;   // FWD_GRAD: positive scores [Nx]
    // logQ sampling correction (outputs/contrastive.py:309-319, transforms/bias.py:238-254): score -= x_corr[x] + y_corr[j] (either may be
    // NULL); corr_after_mask = 1 applies it AFTER the false-negative rescoring.  Six-term kernel only (HAS_CORR instantiations)
    const float *x_corr, *y_corr;
    int corr_after_mask;
    float invT, fns, gscale;
    float *part_m, *part_s, *opart;
    int tiles_per_split;
    int lab;  // ablation bits of lab builds (MH_LAB); 0 in the shipped library
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
template <int MODE, typename IdT, bool HAS_IDS, bool LSE_STREAM, int XT, int NIMG, bool HAS_CORR, int EW>
__global__ __launch_bounds__(512 / XT, 1) void stream_split_kernel(const SplitArgs a) {
    using G = PG<NIMG, EW>;
    constexpr int PKS = EW / 16;   // k-steps of GEMM 1
    constexpr int CR = EW / 8;     // 16-byte chunks per row of the row-major images
    // chunk swizzle of row r of the row-major images (the b128 fragment reads of GEMM 1 are conflict-free): 256-byte rows r mod 16,
    // 128-byte rows (r / 2) mod 8
    auto swzr = [](int r) { return EW == 128 ? (r & 15) : ((r >> 1) & 7); };
    constexpr int PBN = G::BN, P_ARR = G::ARR, P_STAGE = G::STAGE, P_AUX = G::AUX, NST = G::NST;
    constexpr bool ROT = (NIMG == 3);  // rotated schedule of the second wavefront of every SIMD (below)
    static_assert(!ROT || PBN == 32, "the rotated schedule is written for one 32-row unit per tile");
    constexpr int IDW = sizeof(IdT) / 4;
    constexpr int NW = 8 / XT;  // wavefronts per workgroup
    constexpr int ID_WI = (PBN * IDW + 63) / 64;  // wave instructions of 64 words that bring a tile's ids
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int64_t x0 = (int64_t)blockIdx.x * PXB + wave * 32 * XT;
    const int split = blockIdx.y;
    const int nt_all = (int)((a.Ny + PBN - 1) / PBN);
    const int t_beg = split * a.tiles_per_split;
    const int t_end = (t_beg + a.tiles_per_split < nt_all) ? t_beg + a.tiles_per_split : nt_all;

    // one tile: NIMG arrays x PBN * CR chunks of 16 bytes per orientation; chunk position L of an array <- a swizzled source chunk
    constexpr int CPA = PBN * CR;            // chunks per array (both orientations)
    constexpr int WI = NIMG * CPA / 64;      // wave instructions per orientation
    constexpr int WJ = (WI + NW - 1) / NW;   // ... per wavefront (the last round may be partial: wave-uniform guard)
    auto issue = [&](int t, int stage) {
        unsigned char* st = smem + stage * P_STAGE;
        const int64_t row0 = (int64_t)t * PBN;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {  // row-major image: position (r, p) holds chunk p ^ swzr(r) of row r (rows clamped)
            const int wi = j * NW + wave;
            if (WI % NW != 0 && wi >= WI) break;
            const int arr = wi / (CPA / 64), Lp = (wi % (CPA / 64)) * 64 + lane, r = Lp / CR, p = Lp % CR, c = p ^ swzr(r);
            int64_t row = row0 + r;
            if (row > a.Ny - 1) row = a.Ny - 1;
            p_dma16(a.y[arr] + row * EW + c * 8, st + wi * 1024);
        }
        if (MODE != PM_FWD)
#pragma unroll
        for (int j = 0; j < WJ; ++j) {  // transposed image: position (e, p) holds chunk p ^ swz(e) of row e (8 rows of Y each)
            constexpr int CPE = PBN / 8;  // chunks per row e; swz(e) = (e / (16 / CPE)) mod CPE: the b128 reads of GEMM 2 are conflict-free
            const int wi = j * NW + wave;
            if (WI % NW != 0 && wi >= WI) break;
            const int arr = wi / (CPA / 64), Lp = (wi % (CPA / 64)) * 64 + lane, e = Lp / CPE, p = Lp % CPE,
                      c = p ^ ((e / (16 / CPE)) & (CPE - 1));
            p_dma16(a.yt[arr] + (int64_t)e * a.ldT + row0 + c * 8, st + NIMG * P_ARR + wi * 1024);
        }
        if (HAS_IDS && wave < ID_WI) {  // PBN ids = PBN * IDW words (a whole wave instruction is loaded; the words past the tile are ignored)
            int64_t w = row0 * IDW + wave * 64 + lane;
            const int64_t last = a.Ny * IDW - 1;
            if (w > last) w = last;
            p_dma4(static_cast<const uint32_t*>(a.y_ids) + w, smem + P_AUX + stage * 512 + wave * 256);
        }
        if (MODE == PM_GRAD && LSE_STREAM && wave == NW - 1) {
            int64_t w = row0 + lane;
            if (w > a.Ny - 1) w = a.Ny - 1;
            p_dma4(a.lse + w, smem + P_AUX + NST * 512 + stage * 256);
        }
        if (HAS_CORR && a.y_corr && wave == NW - 2) {
            int64_t w = row0 + lane;
            if (w > a.Ny - 1) w = a.Ny - 1;
            p_dma4(a.y_corr + w, smem + P_AUX + NST * 768 + stage * 256);
        }
    };
    if (t_beg < t_end) issue(t_beg, 0);

    // stationary fragments (B operand of GEMM 1: lane = column l31 of its 32-row block, k = 16 ks + 8 h .. + 7), image by image
    bf16x8_t xs[NIMG][XT][PKS];
    bool xvalid[XT];
    IdT x_id[XT];
    float lse2_x[XT], m_run[XT], s_run[XT], xc[XT];
#pragma unroll
    for (int tn = 0; tn < XT; ++tn) {
        int64_t xrow = x0 + tn * 32 + l31;
        xvalid[tn] = xrow < a.Nx;
        if (!xvalid[tn]) xrow = a.Nx - 1;
#pragma unroll
        for (int g = 0; g < NIMG; ++g)
#pragma unroll
            for (int ks = 0; ks < PKS; ++ks)
                xs[g][tn][ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a.x[g] + xrow * EW + ks * 16 + h * 8));
        x_id[tn] = 0;
        if (HAS_IDS) x_id[tn] = static_cast<const IdT*>(a.x_ids)[xrow];
        xc[tn] = (HAS_CORR && a.x_corr) ? a.x_corr[xrow] : 0.f;
        lse2_x[tn] = 0.f;
        if (MODE == PM_GRAD && !LSE_STREAM) lse2_x[tn] = a.lse[xrow] * P_LOG2E;
        m_run[tn] = (MODE != PM_GRAD) ? a.pos[xrow] * a.invT * P_LOG2E : P_NEG_BIG;  // the reference max starts at the positive logit
        s_run[tn] = 0.f;
    }
    const float scale2 = a.invT * P_LOG2E;
    constexpr int OB = (MODE == PM_FWD) ? 1 : EW / 32;  // forward-only: no output accumulators (one dummy block keeps the code uniform)
    f32x16 o[OB][XT];  // O^T: block eb of 32 columns e x block tn of 32 stationary rows; lane: row x = l31, e = (i & 3) + 8 (i >> 2) + 4 h
#pragma unroll
    for (int eb = 0; eb < OB; ++eb)
#pragma unroll
        for (int tn = 0; tn < XT; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[eb][tn][i] = 0.f;

    // The unit of work: 32 streamed rows against the wavefront's 32 * XT stationary rows, in three phases -- GEMM 1 (matrix pipe),
    // epilogue (vector ALU), GEMM 2 (matrix pipe + the split of the probabilities).  `acc` carries the scores, then the probabilities.
    f32x16 acc[XT];
    auto gemm1 = [&](const unsigned char* st, int u) {
        // ---- GEMM 1 on the unit's 32 streamed rows, software-pipelined over the 8 k-steps ---------------------------------
        f32x16 acc2[XT];  // the second chain of the six-term form (unused otherwise)
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // C of a chain's first MFMA
        const int rd = (u * 32 + l31) * (EW * 2);
        auto frag = [&](int ks, int arr) {
            const int pos = ((2 * ks + h) ^ swzr(l31)) * 16;
            return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(st + arr * P_ARR + rd + pos));
        };
        bf16x8_t af[NIMG];
#pragma unroll
        for (int g = 0; g < NIMG; ++g) af[g] = frag(0, g);
#pragma unroll
        for (int ks = 0; ks < PKS; ++ks) {
            bf16x8_t nf[NIMG];
#pragma unroll
            for (int g = 0; g < NIMG; ++g) nf[g] = (ks + 1 < PKS) ? frag(ks + 1, g) : af[g];
            if (NIMG == 2) {
#pragma unroll
                for (int tn = 0; tn < XT; ++tn) acc[tn] = p_mfma(af[1], xs[0][tn][ks], ks == 0 ? zero : acc[tn]);  // small terms first
#pragma unroll
                for (int tn = 0; tn < XT; ++tn) acc[tn] = p_mfma(af[0], xs[1][tn][ks], acc[tn]);
#pragma unroll
                for (int tn = 0; tn < XT; ++tn) acc[tn] = p_mfma(af[0], xs[0][tn][ks], acc[tn]);
            } else {  // six terms, two chains: (l h, h l, m m) into acc2 and (m h, h m, h h) into acc, alternately
                constexpr int TA[6] = {2, 1, 0, 0, 1, 0}, TB[6] = {0, 0, 2, 1, 1, 0};  // image of the streamed / of the stationary operand
#pragma unroll
                for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                    for (int tn = 0; tn < XT; ++tn) {
                        if (tm & 1) acc[tn] = p_mfma(af[TA[tm] % NIMG], xs[TB[tm] % NIMG][tn][ks], (ks == 0 && tm == 1) ? zero : acc[tn]);
                        else acc2[tn] = p_mfma(af[TA[tm] % NIMG], xs[TB[tm] % NIMG][tn][ks], (ks == 0 && tm == 0) ? zero : acc2[tn]);
                    }
            }
#pragma unroll
            for (int g = 0; g < NIMG; ++g) af[g] = nf[g];
        }
        if (NIMG == 3)
#pragma unroll
            for (int tn = 0; tn < XT; ++tn)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[tn][i] += acc2[tn][i];
    };
    auto epilogue = [&](int u, int nvalid, const IdT* ids, const float* lsej, const float* ycorr) {
        // ---- epilogue: lane = stationary row tn * 32 + l31, streamed rows jl(i) = u * 32 + (i >> 2) * 8 + 4 h + (i & 3) ----------
        const int jl0 = u * 32 + 4 * h;
#pragma unroll
        for (int tn = 0; tn < XT; ++tn) {
            // the 16 scores of this lane are turned into probabilities IN PLACE (acc[tn]).  Rescored false negatives (one bit per
            // score) and the rows past the end of Y are rare: both are handled under wave-uniform branches, the common tile pays
            // 16 compares for them
            bool hit = false;
            if (HAS_IDS)
#pragma unroll
                for (int i = 0; i < 16; ++i) hit |= (ids[jl0 + (i >> 2) * 8 + (i & 3)] == x_id[tn]);
            const bool any_masked = HAS_IDS && __any(hit);
            if (HAS_CORR && !a.corr_after_mask)  // logQ correction of the scores that are NOT rescored
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[tn][i] -= xc[tn] + (a.y_corr ? ycorr[jl0 + (i >> 2) * 8 + (i & 3)] : 0.f);
            unsigned mbits = 0;
            if (any_masked)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool masked = ids[jl0 + (i >> 2) * 8 + (i & 3)] == x_id[tn];
                    mbits |= (masked ? 1u : 0u) << i;
                    acc[tn][i] = masked ? a.fns : acc[tn][i];
                }
            if (HAS_CORR && a.corr_after_mask)  // the `post` form: the rescored entries are corrected too
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[tn][i] -= xc[tn] + (a.y_corr ? ycorr[jl0 + (i >> 2) * 8 + (i & 3)] : 0.f);
            if (nvalid < PBN)  // only the last tile of Y can be partial
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[tn][i] = (jl0 + (i >> 2) * 8 + (i & 3) >= nvalid) ? -INFINITY : acc[tn][i];
            if (MODE == PM_GRAD) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float l2 = LSE_STREAM ? lsej[jl0 + (i >> 2) * 8 + (i & 3)] * P_LOG2E : lse2_x[tn];
                    acc[tn][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[tn][i], scale2, -l2)) * a.gscale;  // -inf on invalid rows -> 0
                }
            } else {  // FWD / FWD_GRAD: lazy reference max shared by the two lanes of a row (scale2 > 0: the max commutes with it)
                float tmax = acc[tn][0];
#pragma unroll
                for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, acc[tn][i]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32)) * scale2;
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
                    acc[tn][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[tn][i], scale2, -m_run[tn]));
                    s_add += acc[tn][i];  // rescored false negatives stay in the denominator
                }
                s_run[tn] += s_add;
            }
            if (any_masked && MODE != PM_FWD)  // ... and out of the gradient
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[tn][i] = ((mbits >> i) & 1u) ? 0.f : acc[tn][i];
        }
    };
    auto gemm2 = [&](const unsigned char* st, int u) {
        // ---- GEMM 2: O^T[e, x] += sum_j Y^T[e, j] P^T[j, x]; A = one 16-byte piece of row e of the transposed image; the 16
        // probabilities of a lane, split into bf16 pieces, are the B operand of its two k-steps --------------------------------
        const unsigned char* yt = st + NIMG * P_ARR;
        if (MODE != PM_FWD)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t pp[NIMG][XT];
#pragma unroll
            for (int tn = 0; tn < XT; ++tn) {
                uint32_t w[NIMG][4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (NIMG == 3) mh_split3_pair(acc[tn][8 * s + 2 * k], acc[tn][8 * s + 2 * k + 1], w[0][k], w[1][k], w[NIMG - 1][k]);
                    else mh_split_pair(acc[tn][8 * s + 2 * k], acc[tn][8 * s + 2 * k + 1], w[0][k], w[1][k]);
                }
#pragma unroll
                for (int g = 0; g < NIMG; ++g) pp[g][tn] = __builtin_bit_cast(bf16x8_t, make_uint4(w[g][0], w[g][1], w[g][2], w[g][3]));
            }
            constexpr int EBG = (NIMG == 3 && OB > 1) ? 2 : 1;  // column blocks of O^T walked together (independent MFMA chains)
            const int c0 = 4 * u + 2 * s;
#pragma unroll
            for (int eb0 = 0; eb0 < OB; eb0 += EBG) {
                bf16x8_t at[NIMG][EBG];
#pragma unroll
                for (int q = 0; q < EBG; ++q) {
                    // row e of the transposed image, chunk 2 (2 u + s) + h: the lane's 8 rows of this k-step (see split_prepare_kernel)
                    constexpr int CPE = PBN / 8;
                    const int e = (eb0 + q) * 32 + l31, sw = (e / (16 / CPE)) & (CPE - 1);
                    const unsigned char* frag_at = yt + e * (PBN * 2) + (((c0 + h) ^ sw) << 4);
#pragma unroll
                    for (int g = 0; g < NIMG; ++g)
                        at[g][q] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(frag_at + g * P_ARR));
                }
                if (NIMG == 2) {
#pragma unroll
                    for (int tn = 0; tn < XT; ++tn) o[eb0][tn] = p_mfma(at[1][0], pp[0][tn], o[eb0][tn]);
#pragma unroll
                    for (int tn = 0; tn < XT; ++tn) o[eb0][tn] = p_mfma(at[0][0], pp[1][tn], o[eb0][tn]);
#pragma unroll
                    for (int tn = 0; tn < XT; ++tn) o[eb0][tn] = p_mfma(at[0][0], pp[0][tn], o[eb0][tn]);
                } else {
                    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};  // l h, h l, m m, m h, h m, h h
#pragma unroll
                    for (int tm = 0; tm < 6; ++tm)
#pragma unroll
                        for (int tn = 0; tn < XT; ++tn)
#pragma unroll
                            for (int q = 0; q < EBG; ++q)
                                o[eb0 + q][tn] = p_mfma(at[TA[tm] % NIMG][q], pp[TB[tm] % NIMG][tn], o[eb0 + q][tn]);
                }
            }
        }
    };
    auto tile_rows = [&](int t) {
        const int64_t j_tile = (int64_t)t * PBN;
        return (j_tile + PBN <= a.Ny) ? PBN : (int)(a.Ny - j_tile);
    };
    auto tile_ids = [&](int stage) { return reinterpret_cast<const IdT*>(smem + P_AUX + stage * 512); };
    auto tile_lse = [&](int stage) { return reinterpret_cast<const float*>(smem + P_AUX + NST * 512 + stage * 256); };
    auto tile_corr = [&](int stage) { return reinterpret_cast<const float*>(smem + P_AUX + NST * 768 + stage * 256); };

    __syncthreads();  // vmcnt(0) + barrier: tile t_beg has landed for every wavefront

    if constexpr (!ROT) {
        for (int t = t_beg; t < t_end; ++t) {
            const int odd = (t - t_beg) & 1;
            if (t + 1 < t_end) issue(t + 1, odd ^ 1);
            const unsigned char* st = smem + odd * P_STAGE;
            const int nvalid = tile_rows(t);
#pragma unroll 1
            for (int u = 0; u < PBN / 32; ++u) {
                if (u * 32 >= nvalid) break;
                gemm1(st, u);
                epilogue(u, nvalid, tile_ids(odd), tile_lse(odd), tile_corr(odd));
                gemm2(st, u);
            }
            __syncthreads();  // every wavefront is done with this tile; the next one (vmcnt(0)) has landed
        }
    } else {
        // Rotated schedule.  The workgroup barrier of every tile puts the two wavefronts of a SIMD in LOCKSTEP: both in GEMM 1 (the
        // matrix pipe shared, the vector ALU idle), then both in the epilogue (the matrix pipe idle) -- the phases ADD (measured: the
        // three-term kernel's 3674 cycles per unit = 1536 of MFMA + ~1900 of vector work).  Here the second wavefront of a SIMD
        // (waves NW/2 ...) executes its barrier of tile t BETWEEN its GEMM 1 and its epilogue, the first one at the end of the tile
        // (s_barrier counts wavefronts, not code addresses): between two barriers the early wavefront runs G1(t) E(t) G2(t) and the
        // late one E(t-1) G2(t-1) G1(t), so that each epilogue runs beside the other wavefront's MFMAs.  Tile t + 2 is requested by
        // a wavefront when it has passed barrier t (nobody reads tile t - 1 any more: three stages) and has landed before it arrives
        // at barrier t + 1.
#ifdef MH_LAB  // lab builds: 1 = odd wavefronts late, 2 = nobody late, 4 / 8 / 16 / 32 / 64 = skip GEMM 1 / epilogue / GEMM 2 / the tile requests / the barrier (timing only)
        const bool late = (a.lab & 2) ? false : ((a.lab & 1) ? ((wave & 1) != 0) : (wave >= NW / 2));
#define MH_SKIP(bit) (a.lab & (bit))
#else
        const bool late = wave >= NW / 2;
#define MH_SKIP(bit) false
#endif
        if (t_beg + 1 < t_end) issue(t_beg + 1, 1);
        int s_cur = 0;  // stage of tile t
        for (int t = t_beg; t < t_end; ++t) {
            const int s_prev = (s_cur == 0) ? NST - 1 : s_cur - 1;  // = the stage of tile t + 2
            const unsigned char* st = smem + s_cur * P_STAGE;
            if (!MH_SKIP(4)) gemm1(st, 0);
            if (late) {
                if (!MH_SKIP(64)) __syncthreads();
                if (t + 2 < t_end && !MH_SKIP(32)) issue(t + 2, s_prev);
            }
            if (!MH_SKIP(8)) epilogue(0, tile_rows(t), tile_ids(s_cur), tile_lse(s_cur), tile_corr(s_cur));
            if (!MH_SKIP(16)) gemm2(st, 0);
            if (!late) {
                if (!MH_SKIP(64)) __syncthreads();
                if (t + 2 < t_end && !MH_SKIP(32)) issue(t + 2, s_prev);
            }
            s_cur = (s_cur == NST - 1) ? 0 : s_cur + 1;
        }
#undef MH_SKIP
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
    float* op = a.opart + (int64_t)split * a.Nx * EW;
#pragma unroll
    for (int tn = 0; tn < XT; ++tn) {
        if (!xvalid[tn]) continue;
        float* orow = op + (x0 + tn * 32 + l31) * EW;
#pragma unroll
        for (int eb = 0; eb < OB; ++eb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {o[eb][tn][4 * g], o[eb][tn][4 * g + 1], o[eb][tn][4 * g + 2], o[eb][tn][4 * g + 3]};
                *reinterpret_cast<f32x4*>(orow + eb * 32 + 8 * g + 4 * h) = v;
            }
    }
}

template <int MODE, bool LSE_STREAM, int XT, int NIMG, bool HAS_CORR = false, int EW = 128>
int32_t launch_split_mode(const SplitArgs& a, int ids_dtype, dim3 grid, hipStream_t s) {
    constexpr int LDS_BYTES = PG<NIMG, EW>::LDS;
#define MH_LAUNCH_SPLIT(IdT, HAS)                                                                                          \
    do {                                                                                                                   \
        auto kern = stream_split_kernel<MODE, IdT, HAS, LSE_STREAM, XT, NIMG, HAS_CORR, EW>;                               \
        static bool attr_done = false;                                                                                     \
        if (!attr_done) {                                                                                                  \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                    LDS_BYTES) != hipSuccess) {                                                            \
                mh_set_error("scorer (split bf16): cannot raise the dynamic LDS limit");                                   \
                return MH_ERR_LAUNCH;                                                                                      \
            }                                                                                                              \
            attr_done = true;                                                                                              \
        }                                                                                                                  \
        MH_LAUNCH(kern, grid, dim3(512 / XT), (size_t)LDS_BYTES, s, a);                                                    \
    } while (0)
    if (!a.x_ids) MH_LAUNCH_SPLIT(int32_t, false);
    else if (ids_dtype == MH_I32) MH_LAUNCH_SPLIT(int32_t, true);
    else MH_LAUNCH_SPLIT(int64_t, true);
#undef MH_LAUNCH_SPLIT
    return MH_OK;
}

// the six-term kernel by mode, with or without the logQ corrections, at embedding width EW
template <int EW>
int32_t launch_six_term(int mode, int lse_stream, bool corr, const SplitArgs& a, int ids_dtype, dim3 grid, hipStream_t s) {
    if (corr) {
        if (mode == PM_FWD) return launch_split_mode<PM_FWD, false, 1, 3, true, EW>(a, ids_dtype, grid, s);
        if (mode == PM_FWD_GRAD) return launch_split_mode<PM_FWD_GRAD, false, 1, 3, true, EW>(a, ids_dtype, grid, s);
        if (lse_stream) return launch_split_mode<PM_GRAD, true, 1, 3, true, EW>(a, ids_dtype, grid, s);
        return launch_split_mode<PM_GRAD, false, 1, 3, true, EW>(a, ids_dtype, grid, s);
    }
    if (mode == PM_FWD) return launch_split_mode<PM_FWD, false, 1, 3, false, EW>(a, ids_dtype, grid, s);
    if (mode == PM_FWD_GRAD) return launch_split_mode<PM_FWD_GRAD, false, 1, 3, false, EW>(a, ids_dtype, grid, s);
    if (lse_stream) return launch_split_mode<PM_GRAD, true, 1, 3, false, EW>(a, ids_dtype, grid, s);
    return launch_split_mode<PM_GRAD, false, 1, 3, false, EW>(a, ids_dtype, grid, s);
}

}  // namespace

// ---- internal interface used by mh_scorer.hip ------------------------------------------------------------------------------------
// Bytes of the split of ONE [N, E] matrix (E = 64 or 128): three images [N, E] and three transposed ones [E, ldT] (all bf16; the
// three-term arithmetic uses two of each), 256-byte aligned parts.
int64_t mh_split_matrix_bytes(int64_t N, int E) {
    const int64_t ldT = (N + 63) / 64 * 64;
    auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
    return 3 * al(N * E * 2) + 3 * al(E * ldT * 2);
}

struct MhSplitMatrix {
    uint16_t *img[3], *imgT[3];  // hi, (mid,) lo -- img[nimg - 1] is the last piece
    int64_t ldT;
    int nimg, E;
};

MhSplitMatrix mh_split_prepare(const float* x, int64_t N, int E, void* buf, int nimg, hipStream_t s) {
    MhSplitMatrix m;
    auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
    char* p = static_cast<char*>(buf);
    m.ldT = (N + 63) / 64 * 64;
    m.nimg = nimg;
    m.E = E;
    for (int g = 0; g < 3; ++g) {
        m.img[g] = reinterpret_cast<uint16_t*>(p);
        p += al(N * E * 2);
    }
    for (int g = 0; g < 3; ++g) {
        m.imgT[g] = reinterpret_cast<uint16_t*>(p);
        p += al(E * m.ldT * 2);
    }
    const dim3 grid((unsigned)mh_ceil_div(N, 64));
    if (nimg == 3 && E == 64)
        MH_LAUNCH((split_prepare_kernel<true, 64>), grid, dim3(256), 0, s, x, N, m.img[0], m.img[1], m.img[2], m.imgT[0], m.imgT[1], m.imgT[2], m.ldT);
    else if (nimg == 3)
        MH_LAUNCH((split_prepare_kernel<true, 128>), grid, dim3(256), 0, s, x, N, m.img[0], m.img[1], m.img[2], m.imgT[0], m.imgT[1], m.imgT[2], m.ldT);
    else
        MH_LAUNCH((split_prepare_kernel<false, 128>), grid, dim3(256), 0, s, x, N, m.img[0], (uint16_t*)nullptr, m.img[1], m.imgT[0],
                  (uint16_t*)nullptr, m.imgT[1], m.ldT);
    return m;
}

// number of candidate splits (<= the fp32 plan's for the same shapes: the partial buffers are sized by the largest of the plans)
int mh_split_plan(int64_t Nx, int64_t Ny, int nimg, int* tiles_per_split) {
    const int row_tiles = (int)mh_ceil_div(Nx, PXB);
    const int nt = (int)mh_ceil_div(Ny, nimg == 3 ? PG<3>::BN : PG<2>::BN);
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
                               float fns, float gscale, float* part_m, float* part_s, float* opart, const float* x_corr,
                               const float* y_corr, int corr_after_mask, hipStream_t s) {
    SplitArgs a;
    a.x_corr = x_corr; a.y_corr = y_corr; a.corr_after_mask = corr_after_mask;
    const int nimg = X.nimg;
    for (int g = 0; g < 3; ++g) {
        a.x[g] = X.img[g < nimg ? g : nimg - 1];
        a.y[g] = Y.img[g < nimg ? g : nimg - 1];
        a.yt[g] = Y.imgT[g < nimg ? g : nimg - 1];
    }
    a.Nx = Nx; a.Ny = Ny; a.ldT = Y.ldT; a.x_ids = x_ids; a.y_ids = y_ids; a.lse = lse; a.pos = pos;
    a.invT = invT; a.fns = fns; a.gscale = gscale; a.part_m = part_m; a.part_s = part_s; a.opart = opart;
    int tps = 1;
    const int nsplit = mh_split_plan(Nx, Ny, nimg, &tps);
    a.tiles_per_split = tps;
    static int lab = -1;
    if (lab < 0) {
        const char* e = MH_LAB_ENV("MERLIN_HIP_SCORER_LAB");
        lab = e ? atoi(e) : 0;
    }
    a.lab = lab;

    dim3 grid((unsigned)mh_ceil_div(Nx, PXB), (unsigned)nsplit);
    const bool corr = x_corr || y_corr;
    if ((corr || X.E != 128) && nimg != 3) {
        mh_set_error("scorer (bf16x3): the logQ correction and E = 64 exist in the six-term kernel only");
        return MH_ERR_UNSUPPORTED;
    }
    static int xt = -1;  // MERLIN_HIP_SCORER_XT = 1 | 2 (lab builds): 32-row blocks of X per wavefront (see the kernel)
    if (xt < 0) {
        const char* e = MH_LAB_ENV("MERLIN_HIP_SCORER_XT");
        xt = (e && atoi(e) == 2) ? 2 : 1;
    }
#ifdef MH_LAB  // measured: 20.5 ms per pass against 9.7 (560-720 bytes of scratch)
    if (nimg == 3 && xt == 2 && !corr && X.E == 128) {
        if (mode == PM_FWD) return launch_split_mode<PM_FWD, false, 2, 3>(a, ids_dtype, grid, s);
        if (mode == PM_FWD_GRAD) return launch_split_mode<PM_FWD_GRAD, false, 2, 3>(a, ids_dtype, grid, s);
        if (lse_stream) return launch_split_mode<PM_GRAD, true, 2, 3>(a, ids_dtype, grid, s);
        return launch_split_mode<PM_GRAD, false, 2, 3>(a, ids_dtype, grid, s);
    }
#endif
    if (nimg == 3) {
        if (X.E == 64) return launch_six_term<64>(mode, lse_stream, corr, a, ids_dtype, grid, s);
        return launch_six_term<128>(mode, lse_stream, corr, a, ids_dtype, grid, s);
    }
    if (mode == PM_FWD) return launch_split_mode<PM_FWD, false, 1, 2>(a, ids_dtype, grid, s);
    if (xt == 2) {
        if (mode == PM_FWD_GRAD) return launch_split_mode<PM_FWD_GRAD, false, 2, 2>(a, ids_dtype, grid, s);
        if (lse_stream) return launch_split_mode<PM_GRAD, true, 2, 2>(a, ids_dtype, grid, s);
        return launch_split_mode<PM_GRAD, false, 2, 2>(a, ids_dtype, grid, s);
    }
    if (mode == PM_FWD_GRAD) return launch_split_mode<PM_FWD_GRAD, false, 1, 2>(a, ids_dtype, grid, s);
    if (lse_stream) return launch_split_mode<PM_GRAD, true, 1, 2>(a, ids_dtype, grid, s);
    return launch_split_mode<PM_GRAD, false, 1, 2>(a, ids_dtype, grid, s);
}
