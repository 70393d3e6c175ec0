// In-batch sampled-softmax scorer for gfx950: C-ABI entry points.  E <= 128 (every BASELINE configuration) runs on the
// row-stationary streaming core of mh_scorer_stream.hip (forward, forward + dq, flash-style backward: nothing of size
// B x Nn is written).  The tiled kernels below remain for E > 128 only.
// Reference: ItemRetrievalScorer.call_outputs (merlin/models/tf/blocks/retrieval/base.py:283-429),
// ContrastiveOutput.outputs (tf/outputs/contrastive.py:276-344), rescore_false_negatives
// (tf/utils/tf_utils.py:126-154), LogitsTemperatureScaler (tf/transforms/bias.py:65-68),
// CategoricalCrossEntropy(from_logits=True) (tf/losses/listwise.py:38-52).
//
// scores = q neg^T is a [B, Nn] GEMM with K = E (275 GFLOP at B = Nn = 32 K, E = 128): the only
// MFMA-bound piece of the retrieval path.  Materialised, the logits are 4.3 GB per pass and the
// reference makes 5 passes over them (matmul, where, concat, /T, softmax-CE).  Here one kernel
// owns a 128-row tile of q, streams 128-column tiles of neg through LDS and keeps a per-lane
// online log-sum-exp of every row it owns in registers; the id-equality mask, the temperature
// and the (optional) logits store are the epilogue of each 128x128 tile.  In fused mode nothing
// of size B*Nn touches HBM: algorithmic bytes = 4(2BE + NnE) + ids + 8B.
//
// Numerics: every score is one k-ascending fp32 fmaf chain (mh_gemm_core.h).  The masked value
// is false_neg_score (then divided by T, exactly as the reference scales AFTER rescoring).
#include "mh_gemm_core.h"

#include <stdlib.h>
#include <string.h>

#include <math.h>

using namespace mhgemm;

namespace {

constexpr int SBM = 128, SBN = 128;
constexpr int SWM = 4, SWN = 2;  // 8 wavefronts per workgroup: 4 per SIMD keep the matrix pipe fed during epilogues
constexpr float NEG_INF = -INFINITY;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // raw v_exp_f32
constexpr float M_INIT = -1.0e30f;  // finite "minus infinity" of the running max (base-2 domain)

// MODE 0: forward (optional logits store + online LSE partials)
// MODE 1: backward helper: ds[row, col] = masked ? 0 : exp(z - lse[row]) * gscale   (z = s / T)
template <int MODE, bool HAS_IDS, typename IdT, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 2) void scorer_kernel(const float* __restrict__ q, const float* __restrict__ neg,
                                                    const IdT* __restrict__ pos_ids,
                                                    const IdT* __restrict__ neg_ids, int64_t B, int64_t Nn, int E,
                                                    float invT, float fns, float* __restrict__ logits,
                                                    int64_t ld_logits, float* __restrict__ part_m,
                                                    float* __restrict__ part_s, int tiles_per_split,
                                                    const float* __restrict__ lse, float gscale,
                                                    float* __restrict__ ds, int vec_q, int vec_n,
                                                    const float* __restrict__ neg_corr, int corr_after_mask) {
    constexpr int TM = SBM / WM / 32, TN = SBN / WN / 32, NTH = WM * WN * 64;
    __shared__ __attribute__((aligned(16))) float smem[2 * SBM * LDK + 2 * SBN * LDK + 2 * SBM + 2 * SBM + SBM];
    float* As0 = smem;
    float* As1 = smem + SBM * LDK;
    float* Bs0 = smem + 2 * SBM * LDK;
    float* Bs1 = Bs0 + SBN * LDK;
    float* comb = Bs1 + SBN * LDK;  // [2][SBM] cross-wave combine scratch (MODE 0)
    IdT* pid_s = reinterpret_cast<IdT*>(comb + 2 * SBM);  // [SBM] positive ids of this row tile
    float* lse_s = comb + 4 * SBM;                         // [SBM] row lse (MODE 1)

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave / WN, wn = wave % WN;
    constexpr int WROWS = TM * 32, WCOLS = TN * 32;
    const int64_t row0 = (int64_t)blockIdx.x * SBM;
    const int split = blockIdx.y;
    const int nct_all = (int)((Nn + SBN - 1) / SBN);
    const int ct_beg = split * tiles_per_split;
    const int ct_end = (ct_beg + tiles_per_split < nct_all) ? ct_beg + tiles_per_split : nct_all;
    const int nk = (E + BK - 1) / BK;
    const int total = (ct_end - ct_beg) * nk;

    // row metadata of this tile -> LDS (keeps 96 VGPRs free)
    if (threadIdx.x < SBM) {
        const int64_t row = row0 + threadIdx.x;
        pid_s[threadIdx.x] = (HAS_IDS && row < B) ? pos_ids[row] : (IdT)0;
        lse_s[threadIdx.x] = (MODE == 1 && row < B) ? lse[row] * LOG2E : 0.f;
    }
    float m[TM][16], ssum[TM][16];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            m[tm][r] = M_INIT;
            ssum[tm][r] = 0.f;
        }

    KMajorTile<SBM, NTH> ta;
    KMajorTile<SBN, NTH> tb;
    f32x16 acc[TM][TN];
    zero_acc<TM, TN>(acc);

    ta.init(q, E, row0, B);
    tb.init(neg, E, (int64_t)ct_beg * SBN, Nn);
    if (total > 0) {
        ta.load(0, E, vec_q);
        tb.load(0, E, vec_n);
        ta.store(As0);
        tb.store(Bs0);
    }
    __syncthreads();
    for (int it = 0; it < total; ++it) {
        const int ct = ct_beg + it / nk, kt = it - (it / nk) * nk;
        const bool more = it + 1 < total;
        float* Ac = (it & 1) ? As1 : As0;
        float* Bc = (it & 1) ? Bs1 : Bs0;
        float* An = (it & 1) ? As0 : As1;
        float* Bn = (it & 1) ? Bs0 : Bs1;
        if (more) {
            const int it2 = it + 1;
            const int ct2 = ct_beg + it2 / nk, kt2 = it2 - (it2 / nk) * nk;
            if (kt2 == 0) tb.init(neg, E, (int64_t)ct2 * SBN, Nn);  // next column tile of items
            ta.load(kt2 * BK, E, vec_q);
            tb.load(kt2 * BK, E, vec_n);
        }
        mma_ktile<TM, TN, true>(Ac, wm * WROWS, Bc, wn * WCOLS, 0, acc);
        if (kt == nk - 1) {
            // ---- tile epilogue (branch-free, base-2 domain: z2 = z * log2(e)) ---------------------
            const int64_t c0 = (int64_t)ct * SBN + wn * WCOLS;
            IdT nid[TN];
            bool cvalid[TN];
            float ncorr[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int64_t col = c0 + tn * 32 + acc_col(lane);
                cvalid[tn] = col < Nn;
                nid[tn] = (HAS_IDS && cvalid[tn]) ? neg_ids[col] : (IdT)0;
                ncorr[tn] = (neg_corr && cvalid[tn]) ? neg_corr[col] : 0.f;
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * WROWS + tm * 32 + acc_row(r, lane);
                    const IdT my_pid = HAS_IDS ? pid_s[rl] : (IdT)0;
                    float z[TN];
                    bool masked[TN];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        masked[tn] = HAS_IDS && (my_pid == nid[tn]);
                        const float sc = corr_after_mask ? acc[tm][tn][r] : acc[tm][tn][r] - ncorr[tn];
                        z[tn] = ((masked[tn] ? fns : sc) - (corr_after_mask ? ncorr[tn] : 0.f)) * invT;
                    }
                    if (MODE == 0) {
                        if (logits) {
                            const int64_t row = row0 + rl;
                            if (row < B) {
#pragma unroll
                                for (int tn = 0; tn < TN; ++tn)
                                    if (cvalid[tn]) logits[row * ld_logits + 1 + c0 + tn * 32 + acc_col(lane)] = z[tn];
                            }
                        }
                        float t2[TN];
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) t2[tn] = cvalid[tn] ? z[tn] * LOG2E : NEG_INF;
                        float tmax = t2[0];
#pragma unroll
                        for (int tn = 1; tn < TN; ++tn) tmax = fmaxf(tmax, t2[tn]);
                        const float m_new = fmaxf(m[tm][r], tmax);  // m starts at -1e30 (finite): no inf - inf
                        float acc_s = ssum[tm][r] * fast_exp2(m[tm][r] - m_new);
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) acc_s += fast_exp2(t2[tn] - m_new);
                        ssum[tm][r] = acc_s;
                        m[tm][r] = m_new;
                    } else {
                        const int64_t row = row0 + rl;
                        if (row < B) {
                            const float l2 = lse_s[rl];  // lse * log2(e), staged once per row tile
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn)
                                if (cvalid[tn])
                                    ds[row * Nn + c0 + tn * 32 + acc_col(lane)] =
                                        masked[tn] ? 0.f : fast_exp2(z[tn] * LOG2E - l2) * gscale;
                        }
                    }
                }
            zero_acc<TM, TN>(acc);
        }
        if (more) {
            ta.store(An);
            tb.store(Bn);
        }
        __syncthreads();
    }

    if (MODE == 0) {
        // combine the 32 lanes that share a row, then the two column-half waves
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float mm = m[tm][r], ss = ssum[tm][r];
#pragma unroll
                for (int off = 1; off < 32; off <<= 1) {
                    const float mo = __shfl_xor(mm, off), so = __shfl_xor(ss, off);
                    const float M = fmaxf(mm, mo);
                    ss = ss * fast_exp2(mm - M) + so * fast_exp2(mo - M);
                    mm = M;
                }
                if ((lane & 31) == 0) {
                    const int rl = wm * WROWS + tm * 32 + acc_row(r, lane);
                    As0[wn * SBM + rl] = mm;  // the k-tile buffers are dead after the loop: [WN][SBM] scratch
                    As1[wn * SBM + rl] = ss;
                }
            }
        __syncthreads();
        if (threadIdx.x < SBM) {
            const int rl = threadIdx.x;
            const int64_t row = row0 + rl;
            if (row < B) {
                float M = As0[rl];
#pragma unroll
                for (int w = 1; w < WN; ++w) M = fmaxf(M, As0[w * SBM + rl]);
                float S = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) S += As1[w * SBM + rl] * fast_exp2(As0[w * SBM + rl] - M);
                part_m[(int64_t)split * B + row] = M;  // base-2 running max and sum of 2^(z2 - M)
                part_s[(int64_t)split * B + row] = S;
            }
        }
    }
}

// lse / loss from the per-split partials and the positive logit z0 = pos * invT
__global__ __launch_bounds__(256) void scorer_finalize_kernel(const float* __restrict__ pos, int64_t B, int nsplit,
                                                             const float* __restrict__ part_m,
                                                             const float* __restrict__ part_s, float invT,
                                                             float* __restrict__ logits, int64_t ld_logits,
                                                             float* __restrict__ loss, float* __restrict__ lse) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= B) return;
    const float z0 = pos[row] * invT;
    const float z0_2 = z0 * LOG2E;
    float M = z0_2;
    for (int k = 0; k < nsplit; ++k) M = fmaxf(M, part_m[(int64_t)k * B + row]);
    float S = fast_exp2(z0_2 - M);
    for (int k = 0; k < nsplit; ++k) S += part_s[(int64_t)k * B + row] * fast_exp2(part_m[(int64_t)k * B + row] - M);
    const float l = (M + log2f(S)) * LN2;
    if (logits) logits[row * ld_logits] = z0;
    if (lse) lse[row] = l;
    if (loss) loss[row] = l - z0;
}

__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    int64_t M, int N, float* __restrict__ out,
                                                    const float* __restrict__ corr = nullptr) {
    const int sub = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    float s = 0.f;
    if (row < M)
        for (int k = sub; k < N; k += 16) s = fmaf(a[row * N + k], b[row * N + k], s);
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (row < M && sub == 0) out[row] = corr ? s - corr[row] : s;  // corr: logQ of the positives
}

// positive-column gradient: ds0 = (exp(z0 - lse) - 1) * gscale ; dq += ds0 * item ; ditem = ds0 * q
__global__ __launch_bounds__(256) void scorer_pos_grad_kernel(const float* __restrict__ q, const float* __restrict__ item,
                                                             const float* __restrict__ pos, const float* __restrict__ lse,
                                                             int64_t B, int E, float invT, float gscale,
                                                             float* __restrict__ dq, float* __restrict__ ditem) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * E) return;
    const int64_t row = idx / E;
    const float z0 = pos[row] * invT;
    const float ds0 = (expf(z0 - lse[row]) - 1.f) * gscale;
    dq[idx] += ds0 * item[idx];
    if (ditem) ditem[idx] = ds0 * q[idx];
}

struct Plan {
    int row_tiles, nct, nsplit, tps;
};

Plan make_plan(int64_t B, int64_t Nn) {
    Plan p;
    p.row_tiles = (int)mh_ceil_div(B, SBM);
    p.nct = (int)mh_ceil_div(Nn, SBN);
    if (p.nct < 1) p.nct = 1;
    int want = (int)mh_ceil_div(2 * mh_num_cus(), p.row_tiles);
    if (want < 1) want = 1;
    if (want > p.nct) want = p.nct;
    p.tps = (int)mh_ceil_div(p.nct, want);
    p.nsplit = (int)mh_ceil_div(p.nct, p.tps);
    return p;
}

template <int MODE>
void launch_scorer(const Plan& p, const float* q, const float* neg, const void* pos_ids, const void* neg_ids,
                   int ids_dtype, int64_t B, int64_t Nn, int E, float invT, float fns, float* logits,
                   int64_t ld_logits, float* part_m, float* part_s, const float* lse, float gscale, float* ds,
                   const float* neg_corr, int corr_after_mask, hipStream_t s) {
    const int vec_q = ((reinterpret_cast<uintptr_t>(q) & 15) == 0) && (E % 4 == 0);
    const int vec_n = ((reinterpret_cast<uintptr_t>(neg) & 15) == 0) && (E % 4 == 0);
    dim3 grid((unsigned)p.row_tiles, (unsigned)p.nsplit);
    if (!pos_ids) {
        MH_LAUNCH((scorer_kernel<MODE, false, int32_t, SWM, SWN>), grid, dim3(SWM * SWN * 64), 0, s, q, neg, (const int32_t*)nullptr,
                           (const int32_t*)nullptr, B, Nn, E, invT, fns, logits, ld_logits, part_m, part_s, p.tps, lse,
                           gscale, ds, vec_q, vec_n, neg_corr, corr_after_mask);
    } else if (ids_dtype == MH_I32) {
        MH_LAUNCH((scorer_kernel<MODE, true, int32_t, SWM, SWN>), grid, dim3(SWM * SWN * 64), 0, s, q, neg, (const int32_t*)pos_ids,
                           (const int32_t*)neg_ids, B, Nn, E, invT, fns, logits, ld_logits, part_m, part_s, p.tps, lse,
                           gscale, ds, vec_q, vec_n, neg_corr, corr_after_mask);
    } else {
        MH_LAUNCH((scorer_kernel<MODE, true, int64_t, SWM, SWN>), grid, dim3(SWM * SWN * 64), 0, s, q, neg, (const int64_t*)pos_ids,
                           (const int64_t*)neg_ids, B, Nn, E, invT, fns, logits, ld_logits, part_m, part_s, p.tps, lse,
                           gscale, ds, vec_q, vec_n, neg_corr, corr_after_mask);
    }
}

}  // namespace

// ---- streaming core (mh_scorer_stream.hip) -----------------------------------------------------------------------------
struct MhStreamPlan {
    int bn, row_tiles, nt, nsplit, tps;
    size_t lds;
};
enum { SM_FWD = 0, SM_GRAD = 1, SM_FWD_GRAD = 2 };
MhStreamPlan mh_stream_plan(int mode, int64_t Nx, int64_t Ny, int E, int ids_bytes);
int32_t mh_stream_launch(int mode, int lse_stream, const MhStreamPlan& p, const float* X, int64_t Nx, const float* Y,
                         int64_t Ny, int E, const void* x_ids, const void* y_ids, int ids_dtype, const float* lse,
                         const float* pos, float invT, float fns, float gscale, float* logits, int64_t ld_logits,
                         float* part_m, float* part_s, float* opart, const float* x_corr, const float* y_corr,
                         int corr_after_mask, hipStream_t s);
void mh_stream_fwd_finalize(const float* pos, int64_t B, int nsplit, const float* part_m, const float* part_s, float invT,
                            float* logits, int64_t ld_logits, float* loss, float* lse, hipStream_t s);
void mh_stream_fwd_grad_combine(const float* q, const float* item, const float* pos, int64_t B, int E, int nsplit,
                                const float* part_m, const float* part_s, const float* opart, float invT, float g,
                                float* loss, float* lse, float* dq, float* ditem, hipStream_t s);
void mh_stream_grad_combine(const float* opart, int64_t N, int E, int nsplit, const float* pos, const float* lse,
                            const float* other, const float* self, float invT, float g, float* out, float* out_pos,
                            hipStream_t s);
void mh_stream_pad_rows(const float* src, int64_t N, int E, int Ep, float* dst, hipStream_t s);
// tiled forward-only scorer on the second-generation GEMM core (mh_scorer_tiled.hip)
int mh_scorer_tiled_nsplit(int64_t Nn);
bool mh_scorer_tiled_supported(const float* q, const float* neg, int64_t B, int64_t Nn, int E);
int32_t mh_scorer_tiled_fwd(const float* q, const float* neg, const void* pos_ids, const void* neg_ids, int ids_dtype, int64_t B,
                            int64_t Nn, int E, float invT, float fns, const float* neg_corr, int corr_after_mask, float* part_m,
                            float* part_s, hipStream_t s);
void mh_stream_unpad_rows(const float* src, int64_t N, int E, int Ep, float* dst, hipStream_t s);
// split-bf16 arithmetics of the gradient passes (mh_scorer_split.hip): two (bf16x3, three-term products) or three (bf16x6, six-term,
// fp32-grade) bf16 images of both matrices
struct MhSplitMatrix {
    uint16_t *img[3], *imgT[3];
    int64_t ldT;
    int nimg, E;
};
int64_t mh_split_matrix_bytes(int64_t N, int E);
MhSplitMatrix mh_split_prepare(const float* x, int64_t N, int E, void* buf, int nimg, hipStream_t s);
int mh_split_plan(int64_t Nx, int64_t Ny, int nimg, int* tiles_per_split);
int32_t mh_stream_split_launch(int mode, int lse_stream, const MhSplitMatrix& X, int64_t Nx, const MhSplitMatrix& Y, int64_t Ny,
                               const void* x_ids, const void* y_ids, int ids_dtype, const float* lse, const float* pos, float invT,
                               float fns, float gscale, float* part_m, float* part_s, float* opart, const float* x_corr,
                               const float* y_corr, int corr_after_mask, hipStream_t s);

namespace {

int g_scorer_arith = 0;  // 0 = f32, 1 = bf16x3 (opt-in), 2 = bf16x6 (fp32-grade; what the host side selects by default): mh_set_scorer_arith
inline int split_images() { return g_scorer_arith == 2 ? 3 : 2; }
// the largest split count of the two split-bf16 plans (the partial buffers do not depend on the arithmetic selected later)
inline int split_plan_max(int64_t Nx, int64_t Ny) {
    int tps = 0;
    const int a = mh_split_plan(Nx, Ny, 2, &tps), b = mh_split_plan(Nx, Ny, 3, &tps);
    return a > b ? a : b;
}

inline int padded_E(int E) { return E <= 32 ? 32 : (E <= 64 ? 64 : 128); }
inline int64_t align64(int64_t n) { return (n + 63) / 64 * 64; }  // floats: keeps every sub-buffer 256-byte aligned

// Workspace layout of the streaming path (floats), identical for the query and the calls:
//   pos[B] | qp[B,Ep] itemp[B,Ep] negp[Nn,Ep] (only when E != Ep; negp omitted for in-batch negatives is NOT assumed)
//   | pass-specific partials
struct StreamWs {
    int Ep;
    bool pad;
    int64_t pos, qp, itemp, negp, part_m, part_s, opart_row, opart_col, outp_row, outp_col, outp_item, split_q, split_n, total;
    MhStreamPlan row, col;
};

StreamWs stream_ws(int pass, int64_t B, int64_t Nn, int E, int ids_bytes) {
    StreamWs w;
    w.Ep = padded_E(E);
    w.pad = (w.Ep != E);
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += align64(n); return at; };
    w.pos = take(B);
    w.qp = w.itemp = w.negp = w.outp_row = w.outp_col = w.outp_item = 0;
    if (w.pad) {
        w.qp = take(B * w.Ep);
        w.itemp = take(B * w.Ep);
        w.negp = take(Nn * w.Ep);
    }
    const int row_mode = (pass == 0) ? SM_FWD : (pass == 2 ? SM_FWD_GRAD : SM_GRAD);
    w.row = mh_stream_plan(row_mode, B, Nn, w.Ep, ids_bytes);
    w.col = mh_stream_plan(SM_GRAD, Nn, B, w.Ep, ids_bytes);
    w.part_m = w.part_s = w.opart_row = w.opart_col = 0;
    if (pass == 0 || pass == 2) {
        // pass 0 may run the tiled forward, which writes one partial per 256-candidate tile
        int64_t ns = w.row.nsplit;
        if (pass == 0 && mh_scorer_tiled_nsplit(Nn) > ns) ns = mh_scorer_tiled_nsplit(Nn);
        if ((E == 128 || E == 64) && split_plan_max(B, Nn) > ns) ns = split_plan_max(B, Nn);
        w.part_m = take(ns * B);
        w.part_s = take(ns * B);
    }
    // partials: the largest of the fp32 plan's and the split-bf16 plans' split counts (64- / 32-row tiles: more splits on small shapes)
    int ns_row = w.row.nsplit, ns_col = w.col.nsplit;
    if ((E == 128 || E == 64) && pass != 0) {
        if (split_plan_max(B, Nn) > ns_row) ns_row = split_plan_max(B, Nn);
        if (split_plan_max(Nn, B) > ns_col) ns_col = split_plan_max(Nn, B);
    }
    if (pass == 1 || pass == 2) w.opart_row = take((int64_t)ns_row * B * w.Ep);
    if (pass == 1) w.opart_col = take((int64_t)ns_col * Nn * w.Ep);
    if (w.pad && pass != 0) {
        w.outp_row = take(B * w.Ep);
        w.outp_item = take(B * w.Ep);
        if (pass == 1) w.outp_col = take(Nn * w.Ep);
    }
    w.split_q = w.split_n = 0;
    if (E == 128 || E == 64) {  // the bf16 images of q and of the negatives, both orientations (sized for three images: bf16x6)
        w.split_q = take(mh_split_matrix_bytes(B, E) / 4);
        w.split_n = take(mh_split_matrix_bytes(Nn, E) / 4);
    }
    w.total = o;
    return w;
}

// the split-bf16 kernels cover the in-batch case at E = 128 (six-term: also E = 64) with 16-byte aligned rows (logQ: six-term only; the partial
// buffers are sized for the larger of the two plans' split counts: stream_ws)
bool split_ok(int64_t Nx, int64_t Ny, int E, const float* x_corr, const float* y_corr, const float* a, const float* b, int fp32_nsplit) {
    if (g_scorer_arith == 0 || (E != 128 && E != 64) || Ny < 64) return false;
    if ((x_corr || y_corr || E == 64) && g_scorer_arith != 2) return false;  // the logQ correction and E = 64: six-term kernel only
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) return false;
    (void)Nx;
    (void)fp32_nsplit;
    return true;
}

}  // namespace

extern "C" {

int32_t mh_set_scorer_arith(int32_t mode) {
    MH_REQUIRE(mode >= 0 && mode <= 2, "mh_set_scorer_arith: mode must be 0 (f32), 1 (bf16x3) or 2 (bf16x6)");
    g_scorer_arith = mode;
    return MH_OK;
}

int64_t mh_inbatch_softmax_workspace_bytes(int64_t B, int64_t Nn, int32_t E, int32_t pass) {
    if (B <= 0 || E <= 0) return 0;
    if (Nn < 1) Nn = 1;
    if (E > 128) {  // tiled kernels
        const Plan p = make_plan(B, Nn);
        int64_t floats = B + 2 * (int64_t)p.nsplit * B;
        if (pass != 0) floats = B + B * Nn;
        return floats * (int64_t)sizeof(float);
    }
    // ids of either width: size for the larger plan (the plan only depends on ids_bytes through the LDS size)
    return stream_ws(pass, B, Nn, E, 8).total * (int64_t)sizeof(float);
}

int32_t mh_inbatch_softmax_fwd(const float* q, const float* item, const float* neg_item, const void* pos_ids,
                               const void* neg_ids, int32_t ids_dtype, int64_t B, int64_t Nn, int32_t E,
                               float temperature, float false_neg_score, const float* pos_logq,
                               const float* neg_logq, int32_t logq_after_mask, float* logits, int64_t ld_logits,
                               float* loss, float* lse, void* workspace, int64_t workspace_bytes,
                               mh_stream_t stream) {
    MH_REQUIRE(q && item && neg_item, "mh_inbatch_softmax_fwd: null argument");
    MH_REQUIRE(B >= 1 && Nn >= 1 && E >= 4 && E % 4 == 0 && E <= 1024, "mh_inbatch_softmax_fwd: bad shape B=%lld Nn=%lld E=%d",
               (long long)B, (long long)Nn, E);
    MH_REQUIRE((pos_ids == nullptr) == (neg_ids == nullptr), "mh_inbatch_softmax_fwd: pass both id arrays or neither");
    MH_REQUIRE(!pos_ids || ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_inbatch_softmax_fwd: bad ids_dtype");
    MH_REQUIRE(temperature > 0.f, "mh_inbatch_softmax_fwd: temperature must be > 0");
    MH_REQUIRE(!logits || ld_logits >= Nn + 1, "mh_inbatch_softmax_fwd: ld_logits < 1 + Nn");
    const int64_t need = mh_inbatch_softmax_workspace_bytes(B, Nn, E, 0);
    if (!workspace || workspace_bytes < need) {
        mh_set_error("mh_inbatch_softmax_fwd: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)need);
        return MH_ERR_WORKSPACE;
    }
    hipStream_t s = mh_stream(stream);
    float* ws = static_cast<float*>(workspace);
    const float invT = 1.f / temperature;
    if (E > 128) {
        const Plan p = make_plan(B, Nn);
        float* pos = ws;
        float* part_m = pos + B;
        float* part_s = part_m + (int64_t)p.nsplit * B;
        MH_LAUNCH(rowdot_kernel, dim3((unsigned)mh_ceil_div(B, 16)), dim3(256), 0, s, q, item, B, E, pos, pos_logq);
        launch_scorer<0>(p, q, neg_item, pos_ids, neg_ids, ids_dtype, B, Nn, E, invT, false_neg_score, logits, ld_logits,
                         part_m, part_s, nullptr, 0.f, nullptr, neg_logq, logq_after_mask, s);
        MH_LAUNCH(scorer_finalize_kernel, dim3((unsigned)mh_ceil_div(B, 256)), dim3(256), 0, s, pos, B, p.nsplit,
                           part_m, part_s, invT, logits, ld_logits, loss, lse);
        MH_CHECK_LAUNCH("mh_inbatch_softmax_fwd");
        return MH_OK;
    }
    const StreamWs w = stream_ws(0, B, Nn, E, 8);
    const MhStreamPlan& plan = w.row;
    float* pos = ws + w.pos;
    MH_LAUNCH(rowdot_kernel, dim3((unsigned)mh_ceil_div(B, 16)), dim3(256), 0, s, q, item, B, E, pos, pos_logq);
    {
        // MERLIN_HIP_SCORER_FWD=tiled: the forward-only pass on the tiled kernel of mh_scorer_tiled.hip (second-generation GEMM
        // core, transposed product).  OPT-IN: warmed up on one box it runs 32768 x 32768 x 128 in 2.52-2.53 ms against 2.45-2.46 ms
        // for the row-stationary stream kernel below (0.697 vs 0.714 of the fp32 MFMA peak; profiles/r4_notes.md) -- the 0.70-0.71
        // of the round-3 lab was real, what it was compared with (0.675-0.69) was an under-warmed measurement of the stream kernel.
        const char* fenv = getenv("MERLIN_HIP_SCORER_FWD");
        const int forced = !fenv ? 0 : (!strcmp(fenv, "tiled") ? 1 : (!strcmp(fenv, "stream") ? 2 : 0));
        const bool big = false;
        if (!logits && forced != 2 && (forced == 1 || big) && mh_scorer_tiled_supported(q, neg_item, B, Nn, E)) {
            const int32_t st = mh_scorer_tiled_fwd(q, neg_item, pos_ids, neg_ids, ids_dtype, B, Nn, E, invT, false_neg_score, neg_logq,
                                                   logq_after_mask, ws + w.part_m, ws + w.part_s, s);
            if (st != MH_OK) return st;
            mh_stream_fwd_finalize(pos, B, mh_scorer_tiled_nsplit(Nn), ws + w.part_m, ws + w.part_s, invT, nullptr, 0, loss, lse, s);
            MH_CHECK_LAUNCH("mh_inbatch_softmax_fwd");
            return MH_OK;
        }
    }
    if (!logits && split_ok(B, Nn, E, nullptr, neg_logq, q, neg_item, plan.nsplit)) {  // split-bf16 arithmetic, loss / lse only
        const MhSplitMatrix sq = mh_split_prepare(q, B, E, ws + w.split_q, split_images(), s);
        const MhSplitMatrix sn = mh_split_prepare(neg_item, Nn, E, ws + w.split_n, split_images(), s);
        int tps = 0;
        const int ns = mh_split_plan(B, Nn, split_images(), &tps);
        const int32_t st = mh_stream_split_launch(SM_FWD, 0, sq, B, sn, Nn, pos_ids, neg_ids, ids_dtype, nullptr, pos, invT, false_neg_score,
                                                  1.f, ws + w.part_m, ws + w.part_s, nullptr, nullptr, neg_logq, logq_after_mask, s);
        if (st != MH_OK) return st;
        mh_stream_fwd_finalize(pos, B, ns, ws + w.part_m, ws + w.part_s, invT, nullptr, 0, loss, lse, s);
        MH_CHECK_LAUNCH("mh_inbatch_softmax_fwd");
        return MH_OK;
    }
    const float* qx = q;
    const float* nx = neg_item;
    if (w.pad) {
        mh_stream_pad_rows(q, B, E, w.Ep, ws + w.qp, s);
        mh_stream_pad_rows(neg_item, Nn, E, w.Ep, ws + w.negp, s);
        qx = ws + w.qp;
        nx = ws + w.negp;
    }
    int32_t st = mh_stream_launch(SM_FWD, 0, plan, qx, B, nx, Nn, w.Ep, pos_ids, neg_ids, ids_dtype, nullptr, nullptr, invT,
                                  false_neg_score, 0.f, logits, ld_logits, ws + w.part_m, ws + w.part_s, nullptr, nullptr, neg_logq,
                                  logq_after_mask, s);
    if (st != MH_OK) return st;
    mh_stream_fwd_finalize(pos, B, plan.nsplit, ws + w.part_m, ws + w.part_s, invT, logits, ld_logits, loss, lse, s);
    MH_CHECK_LAUNCH("mh_inbatch_softmax_fwd");
    return MH_OK;
}

int32_t mh_inbatch_softmax_fwd_dq(const float* q, const float* item, const float* neg_item, const void* pos_ids,
                                  const void* neg_ids, int32_t ids_dtype, int64_t B, int64_t Nn, int32_t E,
                                  float temperature, float false_neg_score, const float* pos_logq,
                               const float* neg_logq, int32_t logq_after_mask, float grad_scale, float* loss, float* lse,
                                  float* dq, float* ditem, void* workspace, int64_t workspace_bytes,
                                  mh_stream_t stream) {
    MH_REQUIRE(q && item && neg_item && lse && dq, "mh_inbatch_softmax_fwd_dq: null argument");
    MH_REQUIRE(B >= 1 && Nn >= 1 && E >= 4 && E % 4 == 0, "mh_inbatch_softmax_fwd_dq: bad shape");
    if (E > 128) {
        mh_set_error("mh_inbatch_softmax_fwd_dq: E > 128 is not supported (call _fwd then _bwd)");
        return MH_ERR_UNSUPPORTED;
    }
    MH_REQUIRE((pos_ids == nullptr) == (neg_ids == nullptr), "mh_inbatch_softmax_fwd_dq: pass both id arrays or neither");
    MH_REQUIRE(!pos_ids || ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_inbatch_softmax_fwd_dq: bad ids_dtype");
    MH_REQUIRE(temperature > 0.f, "mh_inbatch_softmax_fwd_dq: temperature must be > 0");
    const int64_t need = mh_inbatch_softmax_workspace_bytes(B, Nn, E, 2);
    if (!workspace || workspace_bytes < need) {
        mh_set_error("mh_inbatch_softmax_fwd_dq: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)need);
        return MH_ERR_WORKSPACE;
    }
    hipStream_t s = mh_stream(stream);
    float* ws = static_cast<float*>(workspace);
    const float invT = 1.f / temperature;
    const float g = grad_scale * invT;  // d loss / d score = d loss / d z * (1/T)
    const StreamWs w = stream_ws(2, B, Nn, E, 8);
    const MhStreamPlan& plan = w.row;
    float* pos = ws + w.pos;
    MH_LAUNCH(rowdot_kernel, dim3((unsigned)mh_ceil_div(B, 16)), dim3(256), 0, s, q, item, B, E, pos, pos_logq);
    const float *qx = q, *ix = item, *nx = neg_item;
    if (w.pad) {
        mh_stream_pad_rows(q, B, E, w.Ep, ws + w.qp, s);
        mh_stream_pad_rows(item, B, E, w.Ep, ws + w.itemp, s);
        mh_stream_pad_rows(neg_item, Nn, E, w.Ep, ws + w.negp, s);
        qx = ws + w.qp;
        ix = ws + w.itemp;
        nx = ws + w.negp;
    }
    int32_t st;
    int nsplit = plan.nsplit;
    if (split_ok(B, Nn, E, nullptr, neg_logq, q, neg_item, plan.nsplit)) {
        const MhSplitMatrix sq = mh_split_prepare(q, B, E, ws + w.split_q, split_images(), s);
        const MhSplitMatrix sn = mh_split_prepare(neg_item, Nn, E, ws + w.split_n, split_images(), s);
        int tps = 0;
        nsplit = mh_split_plan(B, Nn, split_images(), &tps);
        st = mh_stream_split_launch(SM_FWD_GRAD, 0, sq, B, sn, Nn, pos_ids, neg_ids, ids_dtype, nullptr, pos, invT, false_neg_score,
                                    1.f, ws + w.part_m, ws + w.part_s, ws + w.opart_row, nullptr, neg_logq, logq_after_mask, s);
    } else {
        st = mh_stream_launch(SM_FWD_GRAD, 0, plan, qx, B, nx, Nn, w.Ep, pos_ids, neg_ids, ids_dtype, nullptr, pos, invT,
                              false_neg_score, 1.f, nullptr, 0, ws + w.part_m, ws + w.part_s, ws + w.opart_row, nullptr, neg_logq,
                              logq_after_mask, s);
    }
    if (st != MH_OK) return st;
    float* dq_o = w.pad ? ws + w.outp_row : dq;
    float* di_o = ditem ? (w.pad ? ws + w.outp_item : ditem) : nullptr;
    mh_stream_fwd_grad_combine(qx, ix, pos, B, w.Ep, nsplit, ws + w.part_m, ws + w.part_s, ws + w.opart_row, invT, g,
                               loss, lse, dq_o, di_o, s);
    if (w.pad) {
        mh_stream_unpad_rows(dq_o, B, E, w.Ep, dq, s);
        if (ditem) mh_stream_unpad_rows(di_o, B, E, w.Ep, ditem, s);
    }
    MH_CHECK_LAUNCH("mh_inbatch_softmax_fwd_dq");
    return MH_OK;
}

int32_t mh_inbatch_softmax_bwd(const float* q, const float* item, const float* neg_item, const void* pos_ids,
                               const void* neg_ids, int32_t ids_dtype, int64_t B, int64_t Nn, int32_t E,
                               float temperature, float false_neg_score, const float* pos_logq,
                               const float* neg_logq, int32_t logq_after_mask, const float* lse, float grad_scale,
                               float* dq, float* ditem, float* dneg_item, void* workspace, int64_t workspace_bytes,
                               mh_stream_t stream) {
    MH_REQUIRE(q && item && neg_item && lse && dneg_item, "mh_inbatch_softmax_bwd: null argument");
    MH_REQUIRE(B >= 1 && Nn >= 1 && E >= 8 && E % 4 == 0 && E <= 1024, "mh_inbatch_softmax_bwd: bad shape");
    MH_REQUIRE((pos_ids == nullptr) == (neg_ids == nullptr), "mh_inbatch_softmax_bwd: pass both id arrays or neither");
    MH_REQUIRE(temperature > 0.f, "mh_inbatch_softmax_bwd: temperature must be > 0");
    const int64_t need = mh_inbatch_softmax_workspace_bytes(B, Nn, E, 1);
    if (!workspace || workspace_bytes < need) {
        mh_set_error("mh_inbatch_softmax_bwd: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)need);
        return MH_ERR_WORKSPACE;
    }
    hipStream_t s = mh_stream(stream);
    float* ws = static_cast<float*>(workspace);
    const float invT = 1.f / temperature;
    const float gscale = grad_scale * invT;  // d loss / d score = d loss / d z * (1/T)
    if (E > 128) {
        MH_REQUIRE(dq != nullptr, "mh_inbatch_softmax_bwd: dq is required for E > 128");
        float* pos = ws;
        float* ds = pos + B;
        Plan p = make_plan(B, Nn);
        MH_LAUNCH(rowdot_kernel, dim3((unsigned)mh_ceil_div(B, 16)), dim3(256), 0, s, q, item, B, E, pos, pos_logq);
        launch_scorer<1>(p, q, neg_item, pos_ids, neg_ids, ids_dtype, B, Nn, E, invT, false_neg_score, nullptr, 0, nullptr,
                         nullptr, lse, gscale, ds, neg_logq, logq_after_mask, s);
        int32_t st = mh_internal_linear(ds, Nn, neg_item, nullptr, B, (int)Nn, E, MH_ACT_NONE, dq, E, nullptr, nullptr, s);
        if (st != MH_OK) return st;
        st = mh_internal_gemm_tn(ds, Nn, q, E, B, (int)Nn, E, dneg_item, s);
        if (st != MH_OK) return st;
        MH_LAUNCH(scorer_pos_grad_kernel, dim3((unsigned)mh_ceil_div(B * E, 256)), dim3(256), 0, s, q, item, pos, lse,
                           B, E, invT, gscale, dq, ditem);
        MH_CHECK_LAUNCH("mh_inbatch_softmax_bwd");
        return MH_OK;
    }
    // flash-style: the probability tiles are recomputed inside the MFMA kernels.
    //   row pass (skipped when dq == NULL, e.g. after mh_inbatch_softmax_fwd_dq): X = q,   Y = neg -> dq
    //   column pass:                                                              X = neg, Y = q   -> dneg_item
    const StreamWs w = stream_ws(1, B, Nn, E, 8);
    float* pos = ws + w.pos;
    MH_LAUNCH(rowdot_kernel, dim3((unsigned)mh_ceil_div(B, 16)), dim3(256), 0, s, q, item, B, E, pos, pos_logq);
    const float *qx = q, *ix = item, *nx = neg_item;
    if (w.pad) {
        mh_stream_pad_rows(q, B, E, w.Ep, ws + w.qp, s);
        mh_stream_pad_rows(item, B, E, w.Ep, ws + w.itemp, s);
        mh_stream_pad_rows(neg_item, Nn, E, w.Ep, ws + w.negp, s);
        qx = ws + w.qp;
        ix = ws + w.itemp;
        nx = ws + w.negp;
    }
    int32_t st;
    // bf16x3 arithmetic: both matrices split once for the two passes of this call
    const bool use_split = split_ok(B, Nn, E, nullptr, neg_logq, q, neg_item, w.row.nsplit) &&
                           split_ok(Nn, B, E, neg_logq, nullptr, q, neg_item, w.col.nsplit) && B >= 64;
    MhSplitMatrix sq{}, sn{};
    if (use_split) {
        sq = mh_split_prepare(q, B, E, ws + w.split_q, split_images(), s);
        sn = mh_split_prepare(neg_item, Nn, E, ws + w.split_n, split_images(), s);
    }
    if (dq) {
        const MhStreamPlan& pr = w.row;
        int ns = pr.nsplit;
        if (use_split) {
            int tps = 0;
            ns = mh_split_plan(B, Nn, split_images(), &tps);
            st = mh_stream_split_launch(SM_GRAD, 0, sq, B, sn, Nn, pos_ids, neg_ids, ids_dtype, lse, nullptr, invT, false_neg_score,
                                        gscale, nullptr, nullptr, ws + w.opart_row, nullptr, neg_logq, logq_after_mask, s);
        } else {
            st = mh_stream_launch(SM_GRAD, 0, pr, qx, B, nx, Nn, w.Ep, pos_ids, neg_ids, ids_dtype, lse, nullptr, invT,
                                  false_neg_score, gscale, nullptr, 0, nullptr, nullptr, ws + w.opart_row, nullptr, neg_logq, logq_after_mask, s);
        }
        if (st != MH_OK) return st;
        float* dq_o = w.pad ? ws + w.outp_row : dq;
        float* di_o = ditem ? (w.pad ? ws + w.outp_item : ditem) : nullptr;
        mh_stream_grad_combine(ws + w.opart_row, B, w.Ep, ns, pos, lse, ix, qx, invT, gscale, dq_o, di_o, s);
        if (w.pad) {
            mh_stream_unpad_rows(dq_o, B, E, w.Ep, dq, s);
            if (ditem) mh_stream_unpad_rows(di_o, B, E, w.Ep, ditem, s);
        }
    } else {
        MH_REQUIRE(ditem == nullptr, "mh_inbatch_softmax_bwd: ditem needs dq (both come from the row pass)");
    }
    const MhStreamPlan& pc = w.col;
    int nsc = pc.nsplit;
    if (use_split) {
        int tps = 0;
        nsc = mh_split_plan(Nn, B, split_images(), &tps);
        st = mh_stream_split_launch(SM_GRAD, 1, sn, Nn, sq, B, neg_ids, pos_ids, ids_dtype, lse, nullptr, invT, false_neg_score, gscale,
                                    nullptr, nullptr, ws + w.opart_col, neg_logq, nullptr, logq_after_mask, s);
    } else {
        st = mh_stream_launch(SM_GRAD, 1, pc, nx, Nn, qx, B, w.Ep, neg_ids, pos_ids, ids_dtype, lse, nullptr, invT,
                              false_neg_score, gscale, nullptr, 0, nullptr, nullptr, ws + w.opart_col, neg_logq, nullptr, logq_after_mask, s);
    }
    if (st != MH_OK) return st;
    float* dn_o = w.pad ? ws + w.outp_col : dneg_item;
    mh_stream_grad_combine(ws + w.opart_col, Nn, w.Ep, nsc, nullptr, nullptr, nullptr, nullptr, invT, gscale, dn_o,
                           nullptr, s);
    if (w.pad) mh_stream_unpad_rows(dn_o, Nn, E, w.Ep, dneg_item, s);
    MH_CHECK_LAUNCH("mh_inbatch_softmax_bwd");
    return MH_OK;
}

}  // extern "C"
