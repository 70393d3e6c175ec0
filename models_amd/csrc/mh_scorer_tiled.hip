// In-batch sampled-softmax scorer, FORWARD-ONLY (eval / testing / predict: loss and log-sum-exp, no gradients, no materialised
// logits), as a TILED kernel on the second-generation GEMM core (mh_gemm2.h).
// Reference: ContrastiveOutput.outputs (tf/outputs/contrastive.py:276-344) / ItemRetrievalScorer.call_outputs
// (tf/blocks/retrieval/base.py:283-429) + rescore_false_negatives (tf/utils/tf_utils.py:126-154) + the temperature
// (tf/transforms/bias.py:65-68) + CategoricalCrossEntropy(from_logits=True) (tf/losses/listwise.py:38-52).
//
//   z[i, j] = ((q_i . n_j) - logq_j) / T     (false negatives: pos_id[i] == neg_id[j] -> fns; logQ before or after the mask)
//   lse_i   = log(exp(z_pos_i) + sum_j exp(z[i, j]))
//
// The product is computed TRANSPOSED: the 256-row operand of a workgroup tile is the candidate block n[256, E], the 128-column
// (NT) operand the query block q[128, E].  In the MFMA C layout a lane then holds 32 CANDIDATES of each of its two queries per
// wavefront tile: the online (max, sum 2^x) of a query over the candidates of its wavefront is a register loop (the
// untransposed layout needs 5 shuffle steps per row, twice, for 32 rows per lane), one shuffle joins the two half-waves and
// LDS joins the four wavefronts that share the query columns.  Partials go to part[candidate tile][query] in the format of the
// stream kernel (base-2 maximum, sum of 2^(x - max)), so mh_stream_fwd_finalize merges them with the positive logit.
// Measured against the row-stationary stream kernel (forward-only mode: 0.675-0.69 of the fp32 MFMA peak) in
// tools/exp/scorer_lab.hip: 0.70-0.71 (profiles/r3_labs_call1.txt); in the product: profiles/r4_notes.md.
#include "mh_gemm2.h"

namespace {

constexpr int BM = 256, BN = 128, WM = 4, WN = 2, STAGES = 3;
constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// (m, s) <- merge of two online partials in base 2
__device__ __forceinline__ void merge2(float& m, float& s, float m2, float s2) {
    const float mm = fmaxf(m, m2);
    if (mm == -INFINITY) {  // both empty: 2^(-inf - -inf) would be NaN
        s = 0.f;
        return;
    }
    s = s * fast_exp2(m - mm) + s2 * fast_exp2(m2 - mm);
    m = mm;
}

template <typename IdT, bool HAS_IDS, bool HAS_CORR>
__global__ __launch_bounds__(WM* WN * 64) void scorer_tiled_fwd_kernel(const float* __restrict__ neg, const float* __restrict__ q,
                                                                     int64_t Nn, int B, int E, float invT, float fns,
                                                                     const IdT* __restrict__ pos_ids,
                                                                     const IdT* __restrict__ neg_ids,
                                                                     const float* __restrict__ neg_corr, int corr_after_mask,
                                                                     float* __restrict__ part_m, float* __restrict__ part_s,
                                                                     int ncol_tiles) {
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    __shared__ float2 wpart[WM][BN];
    __shared__ IdT ids_s[HAS_IDS ? BM : 1];
    __shared__ float corr_s[HAS_CORR ? BM : 1];
    const int64_t row0 = (int64_t)(blockIdx.x / ncol_tiles) * BM;  // candidates
    const int n0 = (int)(blockIdx.x % ncol_tiles) * BN;            // queries
    if ((HAS_IDS || HAS_CORR) && threadIdx.x < BM) {
        const int64_t r = row0 + threadIdx.x;
        if (HAS_IDS) ids_s[threadIdx.x] = (r < Nn) ? neg_ids[r] : (IdT)0;
        if (HAS_CORR) corr_s[threadIdx.x] = (r < Nn) ? neg_corr[r] : 0.f;
    }
    f32x16 acc[TM][TN];
    mhgemm2::gemm2_tile<BM, BN, WM, WN, true, STAGES, true, 16, 0>(neg, E, q, E, Nn, B, E, row0, n0, smem, acc);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    if (HAS_IDS || HAS_CORR) __syncthreads();  // ids_s / corr_s (the K loop's barriers ordered them already; explicit for every E)
    const float scale2 = invT * LOG2E;
    // (structure of tools/exp/scorer_lab.hip, which was measured at 0.70-0.71: one query column at a time, its 32 logits in a
    // register array beside the accumulators; the variant that rewrote the accumulators in place and shared the LDS reads of a
    // candidate between the two columns measured 9 % SLOWER on the same box -- profiles/r4_notes.md)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = wn * TN * 32 + tn * 32 + l31;  // query inside the tile
        const int qi = n0 + col;
        IdT pid = 0;
        const bool has_pid = HAS_IDS && qi < B;
        if (has_pid) pid = pos_ids[qi];
        float m = -INFINITY, s = 0.f;
        float z[TM][16];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;  // candidate inside the tile
                float v = acc[tm][tn][r];
                if (HAS_CORR && !corr_after_mask) v -= corr_s[rl];
                if (HAS_IDS && has_pid && ids_s[rl] == pid) v = fns;
                if (HAS_CORR && corr_after_mask) v -= corr_s[rl];
                v *= scale2;
                if (row0 + rl >= Nn) v = -INFINITY;
                z[tm][r] = v;
                m = fmaxf(m, v);
            }
        const float mref = (m == -INFINITY) ? 0.f : m;  // every candidate of this lane is padding: s stays 0
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += fast_exp2(z[tm][r] - mref);
        merge2(m, s, __shfl_xor(m, 32), __shfl_xor(s, 32));
        if (h == 0) wpart[wm][col] = make_float2(m, s);
    }
    __syncthreads();
    if (threadIdx.x < BN && n0 + (int)threadIdx.x < B) {
        float2 p = wpart[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < WM; ++w) merge2(p.x, p.y, wpart[w][threadIdx.x].x, wpart[w][threadIdx.x].y);
        const int64_t o = (int64_t)(blockIdx.x / ncol_tiles) * B + n0 + threadIdx.x;
        part_m[o] = p.x;
        part_s[o] = p.y;
    }
}

}  // namespace

// number of partials per query the tiled forward writes (the workspace holds nsplit x B maxima and sums)
int mh_scorer_tiled_nsplit(int64_t Nn) { return (int)((Nn + BM - 1) / BM); }

bool mh_scorer_tiled_supported(const float* q, const float* neg, int64_t B, int64_t Nn, int E) {
    return E % 4 == 0 && E >= 16 && B < (1ll << 31) && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(neg)) & 15) == 0 &&
           mh_ceil_div(Nn, BM) * mh_ceil_div(B, BN) < (1ll << 31);
}

int32_t mh_scorer_tiled_fwd(const float* q, const float* neg, const void* pos_ids, const void* neg_ids, int ids_dtype, int64_t B,
                            int64_t Nn, int E, float invT, float fns, const float* neg_corr, int corr_after_mask, float* part_m,
                            float* part_s, hipStream_t s) {
    const size_t lds = (size_t)STAGES * (BM + BN) * 16 * sizeof(float);
    const int ncol = (int)mh_ceil_div(B, BN);
    const int64_t nrow = mh_ceil_div(Nn, BM);
    const dim3 grid((unsigned)(nrow * ncol));
#define MH_LAUNCH_TILED(IDT, HAS)                                                                                              \
    if (neg_corr) MH_LAUNCH_TILED2(IDT, HAS, true) else MH_LAUNCH_TILED2(IDT, HAS, false)
#define MH_LAUNCH_TILED2(IDT, HAS, CORR)                                                                                       \
    {                                                                                                                          \
        auto kern = scorer_tiled_fwd_kernel<IDT, HAS, CORR>;                                                                    \
        static bool attr_done = false;                                                                                         \
        if (!attr_done) {                                                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_done = true;                                                                                                  \
        }                                                                                                                      \
        MH_LAUNCH(kern, grid, dim3(WM * WN * 64), lds, s, neg, q, Nn, (int)B, E, invT, fns, (const IDT*)pos_ids,       \
                           (const IDT*)neg_ids, neg_corr, corr_after_mask, part_m, part_s, ncol);                              \
    }
    if (!pos_ids) {
        MH_LAUNCH_TILED(int32_t, false)
    } else if (ids_dtype == MH_I32) {
        MH_LAUNCH_TILED(int32_t, true)
    } else {
        MH_LAUNCH_TILED(int64_t, true)
    }
#undef MH_LAUNCH_TILED
#undef MH_LAUNCH_TILED2
    MH_CHECK_LAUNCH("mh_inbatch_softmax_fwd(tiled)");
    return MH_OK;
}
