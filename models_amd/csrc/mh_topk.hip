// Brute-force top-k retrieval for gfx950.
// Reference: BruteForce.call (merlin/models/tf/outputs/topk.py:182-237): scores = q C^T (:113-115),
// tf.math.top_k(scores, k) -- values descending, ties -> LOWER candidate index first (pinned by
// tests/unit/tf/utils/test_tf_utils.py:42-75) --, ids = tf.gather(identifiers, idx);
// V1 twin TopKIndexBlock.call (tf/core/index.py:219-240).
//
// The [Bq, N] score matrix (16 GB at Bq = 4096, N = 1 M) is never held: candidates are processed
// in chunks sized so that the fp32 score chunk (<= 128 MiB) stays inside the 256 MiB Infinity
// Cache between the producing fp32-MFMA GEMM and the consuming select kernel.
// Select: one wavefront per query row keeps the running top-k as a sorted list in LDS.  Because
// candidates arrive in ascending index order, "ties -> lower index" is exactly "a later
// candidate must be STRICTLY greater than the current k-th value to enter", so each 64-wide
// batch costs one compare + ballot; insertions (expected ~k ln(N/k) per row) shift the list.
// Scores are k-ascending fmaf chains, so indices are bit-exact vs oracle/oracle_c.c.
#include "mh_gemm_core.h"

#include <math.h>

using namespace mhgemm;

// top-k threshold filter on the row-stationary streaming core (mh_scorer_stream.hip)
int32_t mh_stream_filter(const float* q, int64_t Bq, const float* cand, int64_t n_cand, int E, const float* tau, int* cnt,
                         float* cs, int32_t* ci, int cap, int64_t idx0, hipStream_t s);

namespace {

constexpr int TOPK_MAX = 1024;

// block = 256 threads = 4 wavefronts = 4 query rows.  dynamic LDS: 4 x 2 x k x (float + int)
__global__ __launch_bounds__(256) void topk_select_kernel(const float* __restrict__ scores, int64_t ld, int64_t Bq,
                                                         int Nc, int64_t base, int k, int seen_before,
                                                         float* __restrict__ best_s, int32_t* __restrict__ best_i,
                                                         const int32_t* __restrict__ cand_ids, int last,
                                                         int32_t* __restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= Bq) return;
    float* Ls0 = smem + wave * 4 * k;
    float* Ls1 = Ls0 + k;
    int* Li0 = reinterpret_cast<int*>(Ls1 + k);
    int* Li1 = Li0 + k;
    float* Ls = Ls0;
    int* Li = Li0;
    float* Lsn = Ls1;
    int* Lin = Li1;

    int cnt = seen_before < k ? seen_before : k;  // filled entries (wave-uniform)
    for (int e = lane; e < k; e += 64) {
        if (e < cnt) {
            Ls[e] = best_s[row * k + e];
            Li[e] = best_i[row * k + e];
        } else {
            Ls[e] = NAN;  // unfilled sentinel: compares false
            Li[e] = 0x7fffffff;
        }
    }
    float tau = (cnt == k) ? Ls[k - 1] : -INFINITY;  // LDS write->read by the same wave: in order
    // (all lanes read the same address after the loop; the compiler orders DS ops of one wave)
    tau = __shfl(tau, 0);

    const float* srow = scores + row * ld;
    for (int j0 = 0; j0 < Nc; j0 += 256) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * 64 + lane;
            v[u] = (j < Nc) ? srow[j] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool pass = (v[u] > tau) || (cnt < k && (j0 + u * 64 + lane) < Nc);
            unsigned long long mask = __ballot(pass);
            while (mask) {
                const int l = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const float sc = __shfl(v[u], l);
                if (!((sc > tau) || cnt < k)) continue;  // tau may have risen inside this batch
                const int idx = (int)(base + j0 + u * 64 + l);
                // entries that stay ahead of the newcomer: score >= sc (their index is lower)
                int pos = 0;
                for (int e0 = 0; e0 < k; e0 += 64) {
                    const int e = e0 + lane;
                    const bool ahead = (e < k) && (Ls[e] >= sc);
                    pos += __popcll(__ballot(ahead));
                }
                if (pos >= k) continue;  // only possible for NaN scores
                for (int e = lane; e < k; e += 64) {
                    float s_new;
                    int i_new;
                    if (e < pos) {
                        s_new = Ls[e];
                        i_new = Li[e];
                    } else if (e == pos) {
                        s_new = sc;
                        i_new = idx;
                    } else {
                        s_new = Ls[e - 1];
                        i_new = Li[e - 1];
                    }
                    Lsn[e] = s_new;
                    Lin[e] = i_new;
                }
                float* ts = Ls; Ls = Lsn; Lsn = ts;
                int* ti = Li; Li = Lin; Lin = ti;
                if (cnt < k) ++cnt;
                if (cnt == k) tau = __shfl(Ls[k - 1], 0);
            }
        }
    }
    for (int e = lane; e < k; e += 64) {
        best_s[row * k + e] = Ls[e];
        best_i[row * k + e] = Li[e];
        if (last && out_ids) {
            const int i = Li[e];
            out_ids[row * k + e] = cand_ids ? cand_ids[i] : i;
        }
    }
}


// ---- fused filter: scores that cannot enter the top-k never leave the registers ---------------------
// After a dense bootstrap over the first candidates, every query row has a lower bound tau (its current
// k-th best score).  The remaining candidates are scored by a persistent 8-wave MFMA kernel (one 128-row
// query tile per workgroup, 128-candidate tiles streamed through LDS like the scorer) whose epilogue
// compares each score with tau[row] and appends the few survivors (score, index) to a per-row compact
// list with one atomic each.  The [Bq, N] score matrix is neither written nor read: HBM traffic drops
// from 8 B per score to ~8 B per SURVIVOR (k/n_seen of the scores).
constexpr int FBM = 128, FBN = 128, FWM = 4, FWN = 2;

__global__ __launch_bounds__(FWM * FWN * 64, 2) void topk_filter_gemm_kernel(
    const float* __restrict__ q, const float* __restrict__ cand, int64_t Bq, int64_t n_beg, int64_t n_end, int E,
    const float* __restrict__ tau, int* __restrict__ cnt, float* __restrict__ cs, int32_t* __restrict__ ci, int cap,
    int tiles_per_split, int vec_q, int vec_c) {
    constexpr int TM = FBM / FWM / 32, TN = FBN / FWN / 32, NTH = FWM * FWN * 64;
    __shared__ __attribute__((aligned(16))) float smem[2 * FBM * LDK + 2 * FBN * LDK + FBM];
    float* As0 = smem;
    float* As1 = smem + FBM * LDK;
    float* Bs0 = smem + 2 * FBM * LDK;
    float* Bs1 = Bs0 + FBN * LDK;
    float* tau_s = Bs1 + FBN * LDK;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int WROWS = TM * 32, WCOLS = TN * 32;
    const int64_t row0 = (int64_t)blockIdx.x * FBM;
    const int64_t ncols = n_end - n_beg;
    const int nct_all = (int)((ncols + FBN - 1) / FBN);
    const int ct_beg = blockIdx.y * tiles_per_split;
    const int ct_end = (ct_beg + tiles_per_split < nct_all) ? ct_beg + tiles_per_split : nct_all;
    const int nk = (E + BK - 1) / BK;
    const int total = (ct_end - ct_beg) * nk;
    if (threadIdx.x < FBM) tau_s[threadIdx.x] = (row0 + threadIdx.x < Bq) ? tau[row0 + threadIdx.x] : INFINITY;
    const float* cbase = cand + n_beg * E;
    KMajorTile<FBM, NTH> ta;
    KMajorTile<FBN, NTH> tb;
    f32x16 acc[TM][TN];
    zero_acc<TM, TN>(acc);
    ta.init(q, E, row0, Bq);
    tb.init(cbase, E, (int64_t)ct_beg * FBN, ncols);
    if (total > 0) {
        ta.load(0, E, vec_q);
        tb.load(0, E, vec_c);
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
            if (kt2 == 0) tb.init(cbase, E, (int64_t)ct2 * FBN, ncols);
            ta.load(kt2 * BK, E, vec_q);
            tb.load(kt2 * BK, E, vec_c);
        }
        mma_ktile<TM, TN, true>(Ac, wm * WROWS, Bc, wn * WCOLS, 0, acc);
        if (kt == nk - 1) {
            const int64_t c0 = (int64_t)ct * FBN + wn * WCOLS;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * WROWS + tm * 32 + acc_row(r, lane);
                    const float t = tau_s[rl];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const float v = acc[tm][tn][r];
                        const int64_t col = c0 + tn * 32 + acc_col(lane);
                        if (v >= t && col < ncols) {  // rare: k / n_seen of the scores
                            const int64_t row = row0 + rl;
                            const int pos = atomicAdd(&cnt[row], 1);
                            if (pos < cap) {
                                cs[row * cap + pos] = v;
                                ci[row * cap + pos] = (int32_t)(n_beg + col);
                            }
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
}

// Merge a compact (score, index) list into the running sorted top-k of each row; order = (score desc,
// index asc) -- the total order of tf.math.top_k -- so survivors may arrive in any order.
__global__ __launch_bounds__(256) void topk_stage_init_kernel(const float* __restrict__ out_scores, int k, int64_t Bq,
                                                              float* __restrict__ tau, int* __restrict__ cnt) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= Bq) return;
    tau[r] = out_scores[r * k + (k - 1)];
    cnt[r] = 0;
    cnt[Bq + r] = 0;
}

__global__ __launch_bounds__(256) void topk_merge_compact_kernel(const float* __restrict__ cs, const int32_t* __restrict__ ci,
                                                                int* __restrict__ cnt, int cap, int64_t Bq, int k,
                                                                float* __restrict__ best_s, int32_t* __restrict__ best_i,
                                                                float* __restrict__ tau, int* __restrict__ overflow,
                                                                const int32_t* __restrict__ cand_ids, int last,
                                                                int32_t* __restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= Bq) return;
    float* Ls = smem + wave * 4 * k;
    float* Lsn = Ls + k;
    int* Li = reinterpret_cast<int*>(Lsn + k);
    int* Lin = Li + k;
    for (int e = lane; e < k; e += 64) {
        Ls[e] = best_s[row * k + e];
        Li[e] = best_i[row * k + e];
    }
    int n = cnt[row];
    if (n > cap) {
        if (lane == 0) overflow[row] = 1;  // this row's list lost survivors: topk_redo_rows_kernel recomputes it
        n = cap;
    }
    float tl = __shfl(Ls[k - 1], 0);
    int il = __shfl(Li[k - 1], 0);
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        const float v = (j < n) ? cs[row * cap + j] : -INFINITY;
        const int vi = (j < n) ? ci[row * cap + j] : 0x7fffffff;
        const bool pass = (j < n) && ((v > tl) || (v == tl && vi < il));
        unsigned long long mask = __ballot(pass);
        while (mask) {
            const int l = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const float sc = __shfl(v, l);
            const int idx = __shfl(vi, l);
            if (!((sc > tl) || (sc == tl && idx < il))) continue;
            int pos = 0;
            for (int e0 = 0; e0 < k; e0 += 64) {
                const int e = e0 + lane;
                const bool ahead = (e < k) && ((Ls[e] > sc) || (Ls[e] == sc && Li[e] < idx));
                pos += __popcll(__ballot(ahead));
            }
            if (pos >= k) continue;
            for (int e = lane; e < k; e += 64) {
                float s_new;
                int i_new;
                if (e < pos) { s_new = Ls[e]; i_new = Li[e]; }
                else if (e == pos) { s_new = sc; i_new = idx; }
                else { s_new = Ls[e - 1]; i_new = Li[e - 1]; }
                Lsn[e] = s_new;
                Lin[e] = i_new;
            }
            float* ts = Ls; Ls = Lsn; Lsn = ts;
            int* ti = Li; Li = Lin; Lin = ti;
            tl = __shfl(Ls[k - 1], 0);
            il = __shfl(Li[k - 1], 0);
        }
    }
    for (int e = lane; e < k; e += 64) {
        best_s[row * k + e] = Ls[e];
        best_i[row * k + e] = Li[e];
        if (last && out_ids) {
            const int i = Li[e];
            out_ids[row * k + e] = cand_ids ? cand_ids[i] : i;
        }
    }
    if (lane == 0) {
        tau[row] = tl;
        cnt[row] = 0;
    }
}

// Exact recomputation of the rows whose compact list overflowed (adversarially ordered candidates, e.g. scores
// ascending with the index: every candidate passes the threshold): one wavefront per dirty row scores ALL candidates
// itself -- each lane one candidate, one k-ascending fmaf chain (bitwise the MFMA result) -- and runs the streaming
// select of topk_select_kernel.  Launched unconditionally (clean rows exit at once), so the call needs no host
// read-back and can be captured into a hipGraph.
__global__ __launch_bounds__(256) void topk_redo_rows_kernel(const float* __restrict__ q, const float* __restrict__ cand,
                                                            int64_t Bq, int64_t N, int E, int k,
                                                            const int* __restrict__ dirty, float* __restrict__ best_s,
                                                            int32_t* __restrict__ best_i,
                                                            const int32_t* __restrict__ cand_ids,
                                                            int32_t* __restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= Bq || !dirty[row]) return;
    float* Ls = smem + wave * 4 * k;
    float* Lsn = Ls + k;
    int* Li = reinterpret_cast<int*>(Lsn + k);
    int* Lin = Li + k;
    for (int e = lane; e < k; e += 64) {
        Ls[e] = NAN;  // unfilled sentinel: compares false
        Li[e] = 0x7fffffff;
    }
    int cnt = 0;
    float tau = -INFINITY;
    const float* qr = q + row * E;
    for (int64_t j0 = 0; j0 < N; j0 += 64) {
        const int64_t j = j0 + lane;
        float v = -INFINITY;
        if (j < N) {
            const float* cr = cand + j * E;
            float acc = 0.f;
            for (int e = 0; e < E; ++e) acc = fmaf(qr[e], cr[e], acc);
            v = acc;
        }
        const bool pass = (v > tau) || (cnt < k && j < N);
        unsigned long long mask = __ballot(pass);
        while (mask) {
            const int l = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const float sc = __shfl(v, l);
            if (!((sc > tau) || cnt < k)) continue;  // tau may have risen inside this batch
            const int idx = (int)(j0 + l);
            int pos = 0;
            for (int e0 = 0; e0 < k; e0 += 64) {
                const int e = e0 + lane;
                const bool ahead = (e < k) && (Ls[e] >= sc);
                pos += __popcll(__ballot(ahead));
            }
            if (pos >= k) continue;  // only possible for NaN scores
            for (int e = lane; e < k; e += 64) {
                float s_new;
                int i_new;
                if (e < pos) { s_new = Ls[e]; i_new = Li[e]; }
                else if (e == pos) { s_new = sc; i_new = idx; }
                else { s_new = Ls[e - 1]; i_new = Li[e - 1]; }
                Lsn[e] = s_new;
                Lin[e] = i_new;
            }
            float* ts = Ls; Ls = Lsn; Lsn = ts;
            int* ti = Li; Li = Lin; Lin = ti;
            if (cnt < k) ++cnt;
            if (cnt == k) tau = __shfl(Ls[k - 1], 0);
        }
    }
    for (int e = lane; e < k; e += 64) {
        best_s[row * k + e] = Ls[e];
        best_i[row * k + e] = Li[e];
        if (out_ids) {
            const int i = Li[e];
            out_ids[row * k + e] = cand_ids ? cand_ids[i] : i;
        }
    }
}

int64_t chunk_cols(int64_t Bq, int64_t N) {
    int64_t nc = (32ll << 20) / (Bq > 0 ? Bq : 1);  // 128 MiB of fp32 scores
    if (nc > 65536) nc = 65536;
    if (nc < 1024) nc = 1024;
    nc = nc / 128 * 128;
    if (nc > N) nc = (N + 3) / 4 * 4;
    return nc;
}

struct FusedPlan {
    bool fused;
    int64_t n0;     // candidates covered by the dense bootstrap
    int cap;        // compact-list capacity per row
    int64_t dense_floats, off_tau, off_cnt, off_cs, off_ci, total;
};

FusedPlan make_fused_plan(int64_t Bq, int64_t N, int k) {
    FusedPlan p;
    const int64_t nc = chunk_cols(Bq, N);
    p.n0 = nc;
    if (p.n0 < 16 * (int64_t)k) p.n0 = ((16 * (int64_t)k + nc - 1) / nc) * nc;  // tau from >= 16k candidates
    p.fused = N > 4 * p.n0;
    if (!p.fused) p.n0 = N;
    p.cap = 8192;
    p.dense_floats = Bq * nc;
    int64_t o = p.dense_floats * 4;
    auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
    o = al(o); p.off_tau = o; o += Bq * 4;
    o = al(o); p.off_cnt = o; o += 2 * Bq * 4;  // counts, then the per-row overflow (dirty) flags
    o = al(o); p.off_cs = o;  o += p.fused ? Bq * (int64_t)p.cap * 4 : 0;
    o = al(o); p.off_ci = o;  o += p.fused ? Bq * (int64_t)p.cap * 4 : 0;
    p.total = al(o);
    return p;
}

}  // namespace

extern "C" {

int64_t mh_topk_workspace_bytes(int64_t Bq, int64_t N, int32_t k) {
    if (Bq <= 0 || N <= 0) return 0;
    return make_fused_plan(Bq, N, k).total;
}

int32_t mh_topk_dot(const float* q, const float* cand, const int32_t* cand_ids, int64_t Bq, int64_t N, int32_t E,
                    int32_t k, float* out_scores, int32_t* out_ids, int32_t* out_idx, void* workspace,
                    int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(q && cand && out_scores && out_idx, "mh_topk_dot: null argument (out_scores and out_idx are required)");
    MH_REQUIRE(Bq >= 0 && N >= 1 && E >= 1, "mh_topk_dot: bad shape");
    MH_REQUIRE(k >= 1 && k <= TOPK_MAX && k <= N, "mh_topk_dot: k=%d must be in [1, min(%d, N=%lld)]", k, TOPK_MAX, (long long)N);
    MH_REQUIRE(N < (1ll << 31), "mh_topk_dot: N must fit int32 indices");
    if (Bq == 0) return MH_OK;
    const int64_t nc = chunk_cols(Bq, N);
    const FusedPlan p = make_fused_plan(Bq, N, k);
    if (!workspace || workspace_bytes < p.total) {
        mh_set_error("mh_topk_dot: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)p.total);
        return MH_ERR_WORKSPACE;
    }
    hipStream_t s = mh_stream(stream);
    char* ws = static_cast<char*>(workspace);
    float* sc = reinterpret_cast<float*>(ws);
    const size_t lds = (size_t)4 * 4 * k * sizeof(float);
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_merge_compact_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_redo_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    // ---- dense bootstrap over [0, n0): chunked score GEMM + streaming select ----
    for (int64_t c0 = 0; c0 < p.n0; c0 += nc) {
        const int64_t ncur = (c0 + nc < p.n0) ? nc : p.n0 - c0;
        const int32_t st = mh_internal_gemm_nt(q, E, cand + c0 * E, E, Bq, (int)ncur, E, sc, nc, s);
        if (st != MH_OK) return st;
        const int last = (!p.fused) && (c0 + nc >= p.n0);
        const int seen = (int)(c0 < k ? c0 : k);
        hipLaunchKernelGGL(topk_select_kernel, dim3((unsigned)mh_ceil_div(Bq, 4)), dim3(256), lds, s, sc, nc, Bq,
                           (int)ncur, c0, k, seen, out_scores, out_idx, cand_ids, last, out_ids);
    }
    if (p.fused) {
        float* tau = reinterpret_cast<float*>(ws + p.off_tau);
        int* cnt = reinterpret_cast<int*>(ws + p.off_cnt);
        int* overflow = cnt + Bq;  // [Bq] per-row dirty flags
        float* cs = reinterpret_cast<float*>(ws + p.off_cs);
        int32_t* ci = reinterpret_cast<int32_t*>(ws + p.off_ci);
        // tau[row] = current k-th best (strided read of the running list); survivor counts and dirty flags cleared.
        // One kernel node, not memset / memcpy nodes: see mh_fill_words in mh_common.h
        hipLaunchKernelGGL(topk_stage_init_kernel, dim3((unsigned)mh_ceil_div(Bq, 256)), dim3(256), 0, s, out_scores, k, Bq, tau, cnt);
        const int vec_q = ((reinterpret_cast<uintptr_t>(q) & 15) == 0) && (E % 4 == 0);
        const int vec_c = ((reinterpret_cast<uintptr_t>(cand) & 15) == 0) && (E % 4 == 0);
        const int row_tiles = (int)mh_ceil_div(Bq, FBM);
        // stages grow 8x: tau tightens between stages, so the survivor rate stays ~ k / n_seen
        int64_t beg = p.n0;
        while (beg < N) {
            int64_t end = beg * 8;
            if (end > N || N - end < beg) end = N;
            if (E == 32 || E == 64 || E == 128) {
                // row-stationary streaming core (mh_scorer_stream.hip): queries in registers, candidates by LDS DMA
                const int32_t st = mh_stream_filter(q, Bq, cand + beg * E, end - beg, E, tau, cnt, cs, ci, p.cap, beg, s);
                if (st != MH_OK) return st;
            } else {
                const int nct = (int)mh_ceil_div(end - beg, FBN);
                int want = (int)mh_ceil_div(2 * mh_num_cus(), row_tiles);
                if (want < 1) want = 1;
                if (want > nct) want = nct;
                const int tps = (int)mh_ceil_div(nct, want);
                const int nsplit = (int)mh_ceil_div(nct, tps);
                hipLaunchKernelGGL(topk_filter_gemm_kernel, dim3((unsigned)row_tiles, (unsigned)nsplit), dim3(FWM * FWN * 64), 0,
                                   s, q, cand, Bq, beg, end, E, tau, cnt, cs, ci, p.cap, tps, vec_q, vec_c);
            }
            hipLaunchKernelGGL(topk_merge_compact_kernel, dim3((unsigned)mh_ceil_div(Bq, 4)), dim3(256), lds, s, cs, ci, cnt,
                               p.cap, Bq, k, out_scores, out_idx, tau, overflow, cand_ids, end >= N ? 1 : 0, out_ids);
            beg = end;
        }
        // a compact list can only overflow on adversarial (e.g. ascending-sorted) data: those rows are recomputed
        // exactly on the device (clean rows exit at once) -- no host read-back, the call stays graph-capturable
        hipLaunchKernelGGL(topk_redo_rows_kernel, dim3((unsigned)mh_ceil_div(Bq, 4)), dim3(256), lds, s, q, cand, Bq, N, E, k,
                           overflow, out_scores, out_idx, cand_ids, out_ids);
    }
    MH_CHECK_LAUNCH("mh_topk_dot");
    return MH_OK;
}

}  // extern "C"
