// In-batch sampled-softmax scorer for gfx950 (fp32 MFMA + fused epilogue).
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
__global__ __launch_bounds__(WM * WN * 64, (WM * WN >= 16) ? 8 : WM * WN / 2) void scorer_kernel(const float* __restrict__ q, const float* __restrict__ neg,
                                                    const IdT* __restrict__ pos_ids,
                                                    const IdT* __restrict__ neg_ids, int64_t B, int64_t Nn, int E,
                                                    float invT, float fns, float* __restrict__ logits,
                                                    int64_t ld_logits, float* __restrict__ part_m,
                                                    float* __restrict__ part_s, int tiles_per_split,
                                                    const float* __restrict__ lse, float gscale,
                                                    float* __restrict__ ds, int vec_q, int vec_n) {
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
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int64_t col = c0 + tn * 32 + acc_col(lane);
                cvalid[tn] = col < Nn;
                nid[tn] = (HAS_IDS && cvalid[tn]) ? neg_ids[col] : (IdT)0;
            }
            const float fns_z = fns * invT;
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
                        z[tn] = masked[tn] ? fns_z : acc[tm][tn][r] * invT;
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
                                                    int64_t M, int N, float* __restrict__ out) {
    const int sub = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    float s = 0.f;
    if (row < M)
        for (int k = sub; k < N; k += 16) s = fmaf(a[row * N + k], b[row * N + k], s);
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (row < M && sub == 0) out[row] = s;
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
                   hipStream_t s) {
    const int vec_q = ((reinterpret_cast<uintptr_t>(q) & 15) == 0) && (E % 4 == 0);
    const int vec_n = ((reinterpret_cast<uintptr_t>(neg) & 15) == 0) && (E % 4 == 0);
    dim3 grid((unsigned)p.row_tiles, (unsigned)p.nsplit);
    if (!pos_ids) {
        hipLaunchKernelGGL((scorer_kernel<MODE, false, int32_t, SWM, SWN>), grid, dim3(SWM * SWN * 64), 0, s, q, neg, (const int32_t*)nullptr,
                           (const int32_t*)nullptr, B, Nn, E, invT, fns, logits, ld_logits, part_m, part_s, p.tps, lse,
                           gscale, ds, vec_q, vec_n);
    } else if (ids_dtype == MH_I32) {
        hipLaunchKernelGGL((scorer_kernel<MODE, true, int32_t, SWM, SWN>), grid, dim3(SWM * SWN * 64), 0, s, q, neg, (const int32_t*)pos_ids,
                           (const int32_t*)neg_ids, B, Nn, E, invT, fns, logits, ld_logits, part_m, part_s, p.tps, lse,
                           gscale, ds, vec_q, vec_n);
    } else {
        hipLaunchKernelGGL((scorer_kernel<MODE, true, int64_t, SWM, SWN>), grid, dim3(SWM * SWN * 64), 0, s, q, neg, (const int64_t*)pos_ids,
                           (const int64_t*)neg_ids, B, Nn, E, invT, fns, logits, ld_logits, part_m, part_s, p.tps, lse,
                           gscale, ds, vec_q, vec_n);
    }
}

}  // namespace

extern "C" {

int64_t mh_inbatch_softmax_workspace_bytes(int64_t B, int64_t Nn, int32_t backward) {
    if (B <= 0) return 0;
    const Plan p = make_plan(B, Nn > 0 ? Nn : 1);
    int64_t floats = B + 2 * (int64_t)p.nsplit * B;
    if (backward) floats = B + B * (Nn > 0 ? Nn : 0);
    return floats * (int64_t)sizeof(float);
}

int32_t mh_inbatch_softmax_fwd(const float* q, const float* item, const float* neg_item, const void* pos_ids,
                               const void* neg_ids, int32_t ids_dtype, int64_t B, int64_t Nn, int32_t E,
                               float temperature, float false_neg_score, float* logits, int64_t ld_logits,
                               float* loss, float* lse, void* workspace, int64_t workspace_bytes,
                               mh_stream_t stream) {
    MH_REQUIRE(q && item && neg_item, "mh_inbatch_softmax_fwd: null argument");
    MH_REQUIRE(B >= 1 && Nn >= 1 && E >= 4 && E % 4 == 0 && E <= 1024, "mh_inbatch_softmax_fwd: bad shape B=%lld Nn=%lld E=%d",
               (long long)B, (long long)Nn, E);
    MH_REQUIRE((pos_ids == nullptr) == (neg_ids == nullptr), "mh_inbatch_softmax_fwd: pass both id arrays or neither");
    MH_REQUIRE(!pos_ids || ids_dtype == MH_I32 || ids_dtype == MH_I64, "mh_inbatch_softmax_fwd: bad ids_dtype");
    MH_REQUIRE(temperature > 0.f, "mh_inbatch_softmax_fwd: temperature must be > 0");
    MH_REQUIRE(!logits || ld_logits >= Nn + 1, "mh_inbatch_softmax_fwd: ld_logits < 1 + Nn");
    const Plan p = make_plan(B, Nn);
    const int64_t need = (B + 2 * (int64_t)p.nsplit * B) * (int64_t)sizeof(float);
    if (!workspace || workspace_bytes < need) {
        mh_set_error("mh_inbatch_softmax_fwd: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)need);
        return MH_ERR_WORKSPACE;
    }
    hipStream_t s = mh_stream(stream);
    float* pos = static_cast<float*>(workspace);
    float* part_m = pos + B;
    float* part_s = part_m + (int64_t)p.nsplit * B;
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)mh_ceil_div(B, 16)), dim3(256), 0, s, q, item, B, E, pos);
    launch_scorer<0>(p, q, neg_item, pos_ids, neg_ids, ids_dtype, B, Nn, E, 1.f / temperature, false_neg_score, logits,
                     ld_logits, part_m, part_s, nullptr, 0.f, nullptr, s);
    hipLaunchKernelGGL(scorer_finalize_kernel, dim3((unsigned)mh_ceil_div(B, 256)), dim3(256), 0, s, pos, B, p.nsplit,
                       part_m, part_s, 1.f / temperature, logits, ld_logits, loss, lse);
    MH_CHECK_LAUNCH("mh_inbatch_softmax_fwd");
    return MH_OK;
}

int32_t mh_inbatch_softmax_bwd(const float* q, const float* item, const float* neg_item, const void* pos_ids,
                               const void* neg_ids, int32_t ids_dtype, int64_t B, int64_t Nn, int32_t E,
                               float temperature, float false_neg_score, const float* lse, float grad_scale,
                               float* dq, float* ditem, float* dneg_item, void* workspace, int64_t workspace_bytes,
                               mh_stream_t stream) {
    MH_REQUIRE(q && item && neg_item && lse && dq && dneg_item, "mh_inbatch_softmax_bwd: null argument");
    MH_REQUIRE(B >= 1 && Nn >= 1 && E >= 8 && E % 4 == 0 && E <= 1024, "mh_inbatch_softmax_bwd: bad shape");
    MH_REQUIRE((pos_ids == nullptr) == (neg_ids == nullptr), "mh_inbatch_softmax_bwd: pass both id arrays or neither");
    MH_REQUIRE(temperature > 0.f, "mh_inbatch_softmax_bwd: temperature must be > 0");
    const int64_t need = (B + B * Nn) * (int64_t)sizeof(float);
    if (!workspace || workspace_bytes < need) {
        mh_set_error("mh_inbatch_softmax_bwd: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)need);
        return MH_ERR_WORKSPACE;
    }
    hipStream_t s = mh_stream(stream);
    const float invT = 1.f / temperature;
    const float gscale = grad_scale * invT;  // d loss / d score = d loss / d z * (1/T)
    float* pos = static_cast<float*>(workspace);
    float* ds = pos + B;
    Plan p = make_plan(B, Nn);
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)mh_ceil_div(B, 16)), dim3(256), 0, s, q, item, B, E, pos);
    launch_scorer<1>(p, q, neg_item, pos_ids, neg_ids, ids_dtype, B, Nn, E, invT, false_neg_score, nullptr, 0, nullptr,
                     nullptr, lse, gscale, ds, s);
    // dq = ds neg   (NN GEMM, contraction over the Nn negatives)
    int32_t st = mh_internal_linear(ds, Nn, neg_item, nullptr, B, (int)Nn, E, MH_ACT_NONE, dq, E, nullptr, nullptr, s);
    if (st != MH_OK) return st;
    // dneg = ds^T q (TN GEMM, contraction over the batch)
    st = mh_internal_gemm_tn(ds, Nn, q, E, B, (int)Nn, E, dneg_item, s);
    if (st != MH_OK) return st;
    hipLaunchKernelGGL(scorer_pos_grad_kernel, dim3((unsigned)mh_ceil_div(B * E, 256)), dim3(256), 0, s, q, item, pos, lse,
                       B, E, invT, gscale, dq, ditem);
    MH_CHECK_LAUNCH("mh_inbatch_softmax_bwd");
    return MH_OK;
}

}  // extern "C"
