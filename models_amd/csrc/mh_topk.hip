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
#include "mh_gemm2.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

using namespace mhgemm;

// top-k threshold filter on the row-stationary streaming core (mh_scorer_stream.hip)
int32_t mh_stream_filter(const float* q, int64_t Bq, const float* cand, int64_t n_cand, int E, const float* tau, int* cnt,
                         float* cs, int32_t* ci, int cap, int64_t idx0, hipStream_t s);

namespace {

constexpr int TOPK_MAX = 1024;

// The running top-k of one query row, sorted by (score desc, index asc), held in the REGISTERS of the row's wavefront:
// position p lives in lane p & 63, register p >> 6 (R = ceil(k / 64) registers per lane).  An insertion is a ballot count of
// the entries ahead of the newcomer plus one shuffle-shift of the tail -- ~10 instructions per register.  The first version
// kept the list in LDS and rewrote all k entries into a second buffer per insertion (two LDS round trips per 64 entries):
// with ~k ln(N / k) insertions per row the select / merge kernels cost 0.88 ms of a 10 ms top-100 call over 1 M candidates.
template <int R>
struct RegList {
    float s[R];
    int i[R];
    __device__ __forceinline__ void fill(int lane) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            s[r] = -INFINITY;  // unfilled: behind every real entry (index INT_MAX loses every tie)
            i[r] = 0x7fffffff;
        }
    }
    __device__ __forceinline__ void load(const float* bs, const int32_t* bi, int n, int lane) {  // first n entries from memory
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = r * 64 + lane;
            s[r] = (p < n) ? bs[p] : -INFINITY;
            i[r] = (p < n) ? bi[p] : 0x7fffffff;
        }
    }
    __device__ __forceinline__ void at(int p, float* sc, int* idx) const {  // wave-uniform p
        float a = 0.f;
        int b = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r == (p >> 6)) {
                a = __shfl(s[r], p & 63);
                b = __shfl(i[r], p & 63);
            }
        *sc = a;
        *idx = b;
    }
    // insert (sc, idx) (wave-uniform) into the first k positions; returns false when it does not make the list
    __device__ __forceinline__ bool insert(float sc, int idx, int k, int lane) {
        int pos = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool ahead = (r * 64 + lane < k) && ((s[r] > sc) || (s[r] == sc && i[r] < idx));
            pos += __popcll(__ballot(ahead));
        }
        if (pos >= k) return false;
#pragma unroll
        for (int r = R - 1; r >= 0; --r) {  // descending: register r - 1 is still the old one when r takes its lane 63
            float us = __shfl_up(s[r], 1);
            int ui = __shfl_up(i[r], 1);
            if (r > 0) {
                const float ws = __shfl(s[r - 1], 63);
                const int wi = __shfl(i[r - 1], 63);
                if (lane == 0) {
                    us = ws;
                    ui = wi;
                }
            }
            const int p = r * 64 + lane;
            if (p > pos) {
                s[r] = us;
                i[r] = ui;
            } else if (p == pos) {
                s[r] = sc;
                i[r] = idx;
            }
        }
        return true;
    }
    __device__ __forceinline__ void store(float* bs, int32_t* bi, int k, int lane) const {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = r * 64 + lane;
            if (p < k) {
                bs[p] = s[r];
                bi[p] = i[r];
            }
        }
    }
};

// block = 256 threads = 4 wavefronts = 4 query rows; the running list of a row lives in its wavefront's registers (RegList)
template <int R>
__global__ __launch_bounds__(256) void topk_select_kernel(const float* __restrict__ scores, int64_t ld, int64_t Bq,
                                                         int Nc, int64_t base, int k, int seen_before,
                                                         float* __restrict__ best_s, int32_t* __restrict__ best_i,
                                                         const int32_t* __restrict__ cand_ids, int last,
                                                         int32_t* __restrict__ out_ids) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= Bq) return;
    int cnt = seen_before < k ? seen_before : k;  // filled entries (wave-uniform)
    RegList<R> L;
    L.load(best_s + row * k, best_i + row * k, cnt, lane);
    float tau = -INFINITY;
    int ti = 0;
    if (cnt == k) L.at(k - 1, &tau, &ti);
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
            // candidates arrive in ascending index order: "ties -> lower index" == a later candidate must be STRICTLY greater
            // than the current k-th value to enter
            const bool pass = (v[u] > tau) || (cnt < k && (j0 + u * 64 + lane) < Nc);
            unsigned long long mask = __ballot(pass);
            while (mask) {
                const int l = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const float sc = __shfl(v[u], l);
                if (!((sc > tau) || cnt < k)) continue;  // tau may have risen inside this batch
                if (!L.insert(sc, (int)(base + j0 + u * 64 + l), k, lane)) continue;  // only possible for NaN scores
                if (cnt < k) ++cnt;
                if (cnt == k) L.at(k - 1, &tau, &ti);
            }
        }
    }
    L.store(best_s + row * k, best_i + row * k, k, lane);
    if (last && out_ids) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = r * 64 + lane;
            if (p < k) out_ids[row * k + p] = cand_ids ? cand_ids[L.i[r]] : L.i[r];
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

// ---- the same filter on the second-generation GEMM core (mh_gemm2.h: DMA tiles, 3-deep ring, 64 x 64 per wavefront) ----------
// One workgroup = one 256-candidate x 128-query score tile, the product computed TRANSPOSED (A = candidates, B = queries, NT):
// in the MFMA C layout a lane then holds 32 candidates of TWO queries, so the thresholds tau[query] are two registers per lane
// and the epilogue is 64 compares with nothing else in the common case (no survivor).  Column (query) tiles are the fastest
// grid dimension: the workgroups resident at one time sweep the query panel (2 MB at 4096 x 128: L2-resident) against a few
// candidate panels that are read from HBM once.  Scores are the same k-ascending fmaf chains as everywhere else (a product
// commutes bit for bit), so survivors carry the bits the select stage compares.  Measured against the row-stationary stream
// filter (mh_scorer_stream.hip) by MERLIN_HIP_TOPK_FILTER=stream|tiled: profiles/r4_notes.md.
constexpr int TFBM = 256, TFBN = 128, TFWM = 4, TFWN = 2, TFST = 3;

__global__ __launch_bounds__(TFWM* TFWN * 64) void topk_filter_tiled_kernel(const float* __restrict__ cand, const float* __restrict__ q,
                                                                           int64_t n_cand, int Bq, int E, const float* __restrict__ tau,
                                                                           int* __restrict__ cnt, float* __restrict__ cs,
                                                                           int32_t* __restrict__ ci, int cap, int64_t idx0,
                                                                           int ncol_tiles) {
    constexpr int TM = TFBM / TFWM / 32, TN = TFBN / TFWN / 32;
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int64_t row0 = (int64_t)(blockIdx.x / ncol_tiles) * TFBM;  // candidates (relative to this stage)
    const int n0 = (int)(blockIdx.x % ncol_tiles) * TFBN;            // queries
    f32x16 acc[TM][TN];
    mhgemm2::gemm2_tile<TFBM, TFBN, TFWM, TFWN, true, TFST, true, 16, 0>(cand, E, q, E, n_cand, Bq, E, row0, n0, smem, acc);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wm = wave / TFWN, wn = wave % TFWN;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int qi = n0 + wn * TN * 32 + tn * 32 + l31;
        const float t = (qi < Bq) ? tau[qi] : INFINITY;
        bool any = false;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) any |= acc[tm][tn][r] >= t;
        if (!any) continue;  // per lane: survivors are k / n_seen of the scores
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[tm][tn][r];
                const int64_t c = row0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (v >= t && c < n_cand) {
                    const int pos = atomicAdd(&cnt[qi], 1);
                    if (pos < cap) {
                        cs[(int64_t)qi * cap + pos] = v;
                        ci[(int64_t)qi * cap + pos] = (int32_t)(idx0 + c);
                    }
                }
            }
    }
}

int32_t tiled_filter(const float* q, int64_t Bq, const float* cand, int64_t n_cand, int E, const float* tau, int* cnt, float* cs,
                     int32_t* ci, int cap, int64_t idx0, hipStream_t s) {
    auto kern = topk_filter_tiled_kernel;
    const size_t lds = (size_t)TFST * (TFBM + TFBN) * 16 * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int ncol = (int)mh_ceil_div(Bq, TFBN);
    const int64_t nrow = mh_ceil_div(n_cand, TFBM);
    MH_REQUIRE(nrow * ncol < (1ll << 31), "top-k tiled filter: grid too large");
    MH_LAUNCH(kern, dim3((unsigned)(nrow * ncol)), dim3(TFWM * TFWN * 64), lds, s, cand, q, n_cand, (int)Bq, E, tau, cnt, cs,
                       ci, cap, idx0, ncol);
    return MH_OK;
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

template <int R>
__global__ __launch_bounds__(256) void topk_merge_compact_kernel(const float* __restrict__ cs, const int32_t* __restrict__ ci,
                                                                int* __restrict__ cnt, int cap, int64_t Bq, int k,
                                                                float* __restrict__ best_s, int32_t* __restrict__ best_i,
                                                                float* __restrict__ tau, int* __restrict__ overflow,
                                                                const int32_t* __restrict__ cand_ids, int last,
                                                                int32_t* __restrict__ out_ids) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= Bq) return;
    RegList<R> L;
    L.load(best_s + row * k, best_i + row * k, k, lane);
    int n = cnt[row];
    if (n > cap) {
        if (lane == 0) overflow[row] = 1;  // this row's list lost survivors: topk_redo_rows_kernel recomputes it
        n = cap;
    }
    float tl;
    int il;
    L.at(k - 1, &tl, &il);
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
            if (!L.insert(sc, idx, k, lane)) continue;
            L.at(k - 1, &tl, &il);
        }
    }
    L.store(best_s + row * k, best_i + row * k, k, lane);
    if (last && out_ids) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int p = r * 64 + lane;
            if (p < k) out_ids[row * k + p] = cand_ids ? cand_ids[L.i[r]] : L.i[r];
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
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_redo_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    // ---- dense bootstrap over [0, n0): chunked score GEMM + streaming select ----
    for (int64_t c0 = 0; c0 < p.n0; c0 += nc) {
        const int64_t ncur = (c0 + nc < p.n0) ? nc : p.n0 - c0;
        const int32_t st = mh_internal_gemm_nt(q, E, cand + c0 * E, E, Bq, (int)ncur, E, sc, nc, s);
        if (st != MH_OK) return st;
        const int last = (!p.fused) && (c0 + nc >= p.n0);
        const int seen = (int)(c0 < k ? c0 : k);
#define MH_TOPK_SELECT(R_)                                                                                          \
    MH_LAUNCH(topk_select_kernel<R_>, dim3((unsigned)mh_ceil_div(Bq, 4)), dim3(256), 0, s, sc, nc, Bq, (int)ncur, c0, \
                       k, seen, out_scores, out_idx, cand_ids, last, out_ids)
        if (k <= 64) MH_TOPK_SELECT(1);
        else if (k <= 128) MH_TOPK_SELECT(2);
        else if (k <= 256) MH_TOPK_SELECT(4);
        else if (k <= 512) MH_TOPK_SELECT(8);
        else MH_TOPK_SELECT(16);
#undef MH_TOPK_SELECT
    }
    if (p.fused) {
        float* tau = reinterpret_cast<float*>(ws + p.off_tau);
        int* cnt = reinterpret_cast<int*>(ws + p.off_cnt);
        int* overflow = cnt + Bq;  // [Bq] per-row dirty flags
        float* cs = reinterpret_cast<float*>(ws + p.off_cs);
        int32_t* ci = reinterpret_cast<int32_t*>(ws + p.off_ci);
        // tau[row] = current k-th best (strided read of the running list); survivor counts and dirty flags cleared.
        // One kernel node, not memset / memcpy nodes: see mh_fill_words in mh_common.h
        MH_LAUNCH(topk_stage_init_kernel, dim3((unsigned)mh_ceil_div(Bq, 256)), dim3(256), 0, s, out_scores, k, Bq, tau, cnt);
        const int vec_q = ((reinterpret_cast<uintptr_t>(q) & 15) == 0) && (E % 4 == 0);
        const int vec_c = ((reinterpret_cast<uintptr_t>(cand) & 15) == 0) && (E % 4 == 0);
        const int row_tiles = (int)mh_ceil_div(Bq, FBM);
        // stages grow 8x: tau tightens between stages, so the survivor rate stays ~ k / n_seen
        int64_t beg = p.n0;
        while (beg < N) {
            int64_t end = beg * 8;
            if (end > N || N - end < beg) end = N;
            // MERLIN_HIP_TOPK_FILTER = tiled | stream forces one of the two MFMA filters (A/B); default: tiled for E >= 96
            const char* fenv = getenv("MERLIN_HIP_TOPK_FILTER");  // read per call (a host-side string test per stage)
            const int forced = !fenv ? 0 : (!strcmp(fenv, "tiled") ? 1 : (!strcmp(fenv, "stream") ? 2 : 0));
            const bool can_tiled = vec_q && vec_c && E % 4 == 0 && E >= 16 && Bq < (1ll << 31);
            const bool can_stream = (E == 32 || E == 64 || E == 128);
            const bool use_tiled = can_tiled && (forced == 1 || (forced == 0 && (E >= 96 || !can_stream)));
            if (use_tiled) {
                const int32_t st = tiled_filter(q, Bq, cand + beg * E, end - beg, E, tau, cnt, cs, ci, p.cap, beg, s);
                if (st != MH_OK) return st;
            } else if (can_stream) {
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
                MH_LAUNCH(topk_filter_gemm_kernel, dim3((unsigned)row_tiles, (unsigned)nsplit), dim3(FWM * FWN * 64), 0,
                                   s, q, cand, Bq, beg, end, E, tau, cnt, cs, ci, p.cap, tps, vec_q, vec_c);
            }
#define MH_TOPK_MERGE(R_)                                                                                           \
    MH_LAUNCH(topk_merge_compact_kernel<R_>, dim3((unsigned)mh_ceil_div(Bq, 4)), dim3(256), 0, s, cs, ci, cnt, p.cap,  \
                       Bq, k, out_scores, out_idx, tau, overflow, cand_ids, end >= N ? 1 : 0, out_ids)
            if (k <= 64) MH_TOPK_MERGE(1);
            else if (k <= 128) MH_TOPK_MERGE(2);
            else if (k <= 256) MH_TOPK_MERGE(4);
            else if (k <= 512) MH_TOPK_MERGE(8);
            else MH_TOPK_MERGE(16);
#undef MH_TOPK_MERGE
            beg = end;
        }
        // a compact list can only overflow on adversarial (e.g. ascending-sorted) data: those rows are recomputed
        // exactly on the device (clean rows exit at once) -- no host read-back, the call stays graph-capturable
        MH_LAUNCH(topk_redo_rows_kernel, dim3((unsigned)mh_ceil_div(Bq, 4)), dim3(256), lds, s, q, cand, Bq, N, E, k,
                           overflow, out_scores, out_idx, cand_ids, out_ids);
    }
    MH_CHECK_LAUNCH("mh_topk_dot");
    return MH_OK;
}

}  // extern "C"
