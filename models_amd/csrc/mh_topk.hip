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


// =====================================================================================================================
// Split-bf16 ("bf16x3") filter: the top-k FILTER on the bf16 matrix pipe, the RESULT still bit-exact
// =====================================================================================================================
// fp32 MFMA runs at 1/16 of the bf16 rate on gfx950, and the filter stages above are MFMA-bound (0.71 of the fp32 peak).  A
// score does not have to be exact to be REJECTED.  Every fp32 value is split once, x = hi + lo + r with hi = bf16(x),
// lo = bf16(x - hi), and the filter multiplies   q . c  ~=  sum_e  hi_q hi_c + hi_q lo_c + lo_q hi_c   on
// v_mfma_f32_32x32x16_bf16 (three bf16 MFMAs per fp32-equivalent one: 16 / 3 of the fp32 rate).
//
// Error bound.  With u = 2^-9 (bf16 round-to-nearest): |x - hi| <= u |x| and |x - hi - lo| <= u^2 |x|.  The dropped terms
// lo_q lo_c + r_q c + (hi_q + lo_q) r_c are <= 3.0001 u^2 |q_e c_e| = 1.15e-5 |q_e c_e| per element; bf16 x bf16 products are
// exact in fp32; the 3 E = 384 fp32 additions of the accumulation cost <= 384 * 2^-23 = 4.6e-5 sum_e |q_e c_e| even if the
// matrix pipe truncated instead of rounding; the exact k-ascending fp32 fmaf chain itself is within 128 * 2^-24 = 0.8e-5 of
// the real product.  Together |A - X| <= 6.6e-5 sum_e |q_e c_e| <= 6.6e-5 |q|_2 |c|_2 (A approximate, X the exact fmaf chain).
// The code uses  m(row) = 2^-13 |q_row|_2 max_j |c_j|_2  (1.22e-4: almost twice the bound; measured: 2.3e-6, r3_bf16x3_lab.txt).
//
// Selection.  The whole staged pipeline (bootstrap -> filter -> merge) runs on approximate scores and keeps the best k' = k +
// slack of them per row (k' = the next multiple of 64 above k + 16).  Let a_k be the k-th and a_k' the k'-th best approximate
// score of a row.  k candidates have A >= a_k, hence X >= a_k - m, so the exact k-th best score is t >= a_k - m; every member of
// the exact top-k has X >= t, hence A >= a_k - 2 m.  If  a_k' < a_k - 2 m  the k' kept candidates therefore CONTAIN the exact
// top-k: they are re-scored with the exact fmaf chain (k' * Bq dot products: nothing), sorted by (score desc, index asc), and the
// first k are the answer -- scores and indices bit-identical to the fp32 pipeline and to oracle/oracle_c.c.  A row whose band
// [a_k - 2 m, a_k] holds more than `slack` candidates (piles of near-duplicate items) is marked dirty and recomputed exactly by
// topk_redo_rows_kernel, like a row whose survivor list overflowed.
//
// Kernel (structure of tools/exp/bf16x3_lab.hip): a workgroup of 4 wavefronts owns 256 queries and one split of the stage's
// candidates.  The query fragments of a wavefront (64 queries x 128 x {hi, lo} = 128 VGPRs) stay in registers; candidate tiles
// (32 rows x 128 x {hi, lo} = 16 KB) stream through a 4-deep LDS ring by DMA (global_load_lds_dwordx4, chunk-swizzled source
// addresses: ds_read_b128 of 16 consecutive rows hits 16 distinct 4-bank groups).  The product is transposed (rows = candidates,
// columns = queries): a lane holds 16 candidates of ONE query per 32-wide block, the threshold is one register per block, and
// the epilogue is 32 compares and nothing else unless a candidate survives.  Workgroup -> (split, query block) is XCD-aware: the
// 16 query blocks of one split run on ONE XCD at the same time, so a candidate tile enters that XCD's L2 once and is read 16 times.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

#ifndef MH_SX_NWV
#define MH_SX_NWV 4
#endif
#ifndef MH_SX_STAGES
#define MH_SX_STAGES 4
#endif
constexpr int SX_NWV = MH_SX_NWV, SX_QW = 64, SX_QB = SX_QW * SX_NWV, SX_CT = 32, SX_STAGES = MH_SX_STAGES;
// geometry of the filter kernel by the embedding width EW (128 or 64): k-steps, bytes of one image of a tile, of a tile (hi + lo), 16-byte
// DMA chunks per thread and tile
template <int EW>
struct SXG {
    static constexpr int KS = EW / 16, CR = EW / 8, ARR = SX_CT * EW * 2, TILE = 2 * ARR, DMA = TILE / (SX_NWV * 64 * 16);
    // chunk swizzle of row r of a tile image (conflict-free b128 fragment reads): 256-byte rows r mod 16, 128-byte rows (r / 2) mod 8
    static __device__ __forceinline__ int swz(int r) { return EW == 128 ? (r & 15) : ((r >> 1) & 7); }
};
constexpr int SX_MAX_SPLITS = 256;
constexpr float SX_MREL = 1.0f / 8192.0f;  // 2^-13: margin = SX_MREL |q| max|c|

__device__ __forceinline__ uint16_t sx_bf16_rne(float x) {
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);  // inf / nan: truncate (keeps the class)
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float sx_bf16_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// x[rows, E] -> hi, lo (bf16) and, per row, |x_row|^2: stored (norm2 != null) and / or folded into a device-wide maximum
// (max_bits: the float's bit pattern, non-negative, so an unsigned atomicMax orders it).  LPR = E / 4 lanes own one row.
template <int LPR>
__global__ __launch_bounds__(256) void topk_split_kernel(const float4* __restrict__ x, uint2* __restrict__ hi, uint2* __restrict__ lo,
                                                        int64_t n4, float* __restrict__ norm2, unsigned* __restrict__ max_bits) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float ss = 0.f;
    if (i < n4) {
        const float4 v = x[i];
        const float a[4] = {v.x, v.y, v.z, v.w};
        uint16_t h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = sx_bf16_rne(a[j]);
            l[j] = sx_bf16_rne(a[j] - sx_bf16_f32(h[j]));
            ss = fmaf(a[j], a[j], ss);
        }
        hi[i] = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
        lo[i] = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
    }
#pragma unroll
    for (int o = LPR / 2; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
    if (i < n4 && (threadIdx.x & (LPR - 1)) == 0) {
        if (norm2) norm2[i / LPR] = ss;
        if (max_bits) atomicMax(max_bits, __float_as_uint(ss));
    }
}

__device__ __forceinline__ void sx_dma16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void sx_wait_vm_and_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int EW>
__global__ __launch_bounds__(SX_NWV * 64, SX_NWV == 4 ? 2 : 1) void topk_filter_bf16x3_kernel(
    const uint16_t* __restrict__ chi, const uint16_t* __restrict__ clo, const uint16_t* __restrict__ qhi,
    const uint16_t* __restrict__ qlo, int64_t c_beg, int64_t c_end, int Bq, const float* __restrict__ tau, int* __restrict__ cnt,
    float* __restrict__ cs, int32_t* __restrict__ ci, int segcap, int* __restrict__ dirty, int nqb, int nsplit,
    int tiles_per_split, int xcd_map, const float* __restrict__ qn2, const unsigned* __restrict__ cmax2_bits) {
    // Survivors go to a segment PRIVATE to this workgroup: list[(row nsplit + split) segcap ..], its fill count kept in LDS (one
    // counter per query of the workgroup) and written out once at the end.  No global atomic: the first version reserved slots with
    // atomicAdd on a per-row counter -- a returning device-scope atomic is ~2 us of latency in the middle of the MFMA loop, and the
    // s_waitcnt it forces also drains the candidate DMA ring (first stage: ~16 survivors per wavefront and tile, 645 us for 6 % of
    // the catalogue).  A lane reserves the slots of ALL its survivors of a tile with ONE ds_add_rtn.
    constexpr int SX_E = EW, SX_KS = SXG<EW>::KS, SX_ARR = SXG<EW>::ARR, SX_TILE = SXG<EW>::TILE, SX_DMA = SXG<EW>::DMA;
    extern __shared__ __attribute__((aligned(1024))) unsigned char sx_smem[];
    __shared__ int sx_cnt[SX_QB];
    sx_cnt[threadIdx.x] = 0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    // XCD-aware work mapping: workgroup b runs on XCD b % 8; the j-th workgroup of an XCD takes query block j % nqb of split
    // xcd + 8 (j / nqb) -- the nqb query blocks of a split are neighbours in time on one XCD (nsplit is a multiple of 8)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int sp = xcd_map ? xcd + 8 * (j / nqb) : (int)(blockIdx.x / nqb);
    const int qb = xcd_map ? j % nqb : (int)(blockIdx.x % nqb);
    if (sp >= nsplit) return;
    const int64_t n_stage = c_end - c_beg;
    const int64_t tiles_all = (n_stage + SX_CT - 1) / SX_CT;
    const int64_t t_beg = (int64_t)sp * tiles_per_split;
    int T = (int)((t_beg + tiles_per_split <= tiles_all) ? tiles_per_split : (tiles_all - t_beg));
    if (T <= 0) {  // an empty split of a small stage: its segments are empty
        const int qi = qb * SX_QB + threadIdx.x;
        if (qi < Bq) cnt[(int64_t)qi * nsplit + sp] = 0;
        return;
    }
    const int64_t c0 = c_beg + t_beg * SX_CT;  // first candidate row of this split
    const int c0i = (int)c0;
    const int n_split = (int)((c_end - c0 < (int64_t)T * SX_CT) ? (c_end - c0) : (int64_t)T * SX_CT);  // valid rows of this split

    // query fragments (B operand: lane = column l31, k = 16 ks + 8 h .. + 7); rows past Bq repeat the last query (masked below)
    bf16x8_t qh[2][SX_KS], ql[2][SX_KS];
    float thr[2], thr_hi[2];
    int qrow[2];
    const float cmax2 = __uint_as_float(*cmax2_bits);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int qi = qb * SX_QB + wave * SX_QW + tn * 32 + l31;
        const int qc = qi < Bq ? qi : Bq - 1;
        qrow[tn] = qi < Bq ? qi : -1;
        thr[tn] = qi < Bq ? tau[qi] : INFINITY;
        // Coarse test on the hi hi term alone: the two small terms of a score are bounded by
        //   sum_e |cl_e qh_e| + |ch_e ql_e| <= 2 * 2^-9 (1 + 2^-8)^2 sum_e |c_e q_e| <= 2^-8 (1 + 2^-6) |c| |q|
        // (|x - bf16(x)| <= 2^-9 |x|, Cauchy-Schwarz), so a candidate whose hi hi product is below thr - 1.05 * 2^-8 |q| max|c|
        // cannot reach thr with them (1.05 also covers the fp32 additions of the 16 small MFMAs).
        thr_hi[tn] = thr[tn] - 1.05f * (1.0f / 256.0f) * sqrtf(qn2[qc] * cmax2);
#pragma unroll
        for (int ks = 0; ks < SX_KS; ++ks) {
            qh[tn][ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(qhi + (int64_t)qc * SX_E + ks * 16 + h * 8));
            ql[tn][ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(qlo + (int64_t)qc * SX_E + ks * 16 + h * 8));
        }
    }
    // DMA sources of this thread inside a tile: chunk position L = (jj NWV + wave) 64 + lane of the 1024-chunk tile image;
    // rows past the end of the stage re-read its last row (their scores are masked in the epilogue)
    int rsrc[SX_DMA], csrc[SX_DMA], asrc[SX_DMA];
#pragma unroll
    for (int jj = 0; jj < SX_DMA; ++jj) {
        const int L = (jj * SX_NWV + wave) * 64 + lane;
        constexpr int CPA = SX_CT * SXG<EW>::CR;  // chunks per image of a tile
        const int Lp = L % CPA, r = Lp / SXG<EW>::CR, p = Lp % SXG<EW>::CR;
        asrc[jj] = L / CPA;
        rsrc[jj] = r;
        csrc[jj] = (p ^ SXG<EW>::swz(r)) * 8;
    }
    const int64_t last_row = c_end - 1;
    auto issue = [&](int t) {
        unsigned char* st = sx_smem + (t % SX_STAGES) * SX_TILE;
#pragma unroll
        for (int jj = 0; jj < SX_DMA; ++jj) {
            int64_t row = c0 + (int64_t)t * SX_CT + rsrc[jj];
            row = row < last_row ? row : last_row;
            sx_dma16((asrc[jj] ? clo : chi) + row * SX_E + csrc[jj], st + (jj * SX_NWV + wave) * 1024);
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the query fragments: keeps the ring's vmcnt arithmetic exact
#pragma unroll
    for (int t = 0; t < SX_STAGES - 1; ++t)
        if (t < T) issue(t);
    const int rd = l31 * (EW * 2);  // byte offset of this lane's candidate row inside a tile array
    for (int t = 0; t < T; ++t) {
        // tile t is complete when at most the loads of tiles t+1 .. t+STAGES-2 are outstanding
        if (t + SX_STAGES - 2 < T) sx_wait_vm_and_barrier<(SX_STAGES - 2) * SX_DMA>();
        else sx_wait_vm_and_barrier<0>();
        if (t + SX_STAGES - 1 < T) issue(t + SX_STAGES - 1);
        const unsigned char* st = sx_smem + (t % SX_STAGES) * SX_TILE;
        // Two levels.  (1) The hi hi term of all 32 x 64 scores of the wavefront's tile: 16 MFMAs, a third of the three-term product.
        // (2) Only for a 32-query block in which SOME lane holds a score that the two small terms could still lift to its threshold
        // (thr_hi above; wave-uniform test) the remaining 16 MFMAs of that block run on top of the same accumulators -- the scores of a
        // skipped block are all below their thresholds whatever the small terms are, so it has no survivors either way, and a block that
        // is not skipped gets exactly the three-term scores: the survivor lists, the proof and the result are those of the full product.
        // Survivors are ~k' / n_seen of the scores: in the late stages (most of the catalogue) most blocks are skipped.
        auto lds_frag = [&](int ks, int arr) {
            const int pos = ((2 * ks + h) ^ SXG<EW>::swz(l31)) * 16;
            return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(st + arr * SX_ARR + rd + pos));
        };
        bf16x8_t ah[SX_KS];
#pragma unroll
        for (int ks = 0; ks < SX_KS; ++ks) ah[ks] = lds_frag(ks, 0);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 acc[2];
#pragma unroll
        for (int ks = 0; ks < SX_KS; ++ks) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], qh[0][ks], ks ? acc[0] : zero, 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], qh[1][ks], ks ? acc[1] : zero, 0, 0, 0);
        }
        bool live[2];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            float mx = __builtin_fmaxf(acc[tn][0], acc[tn][1]);
#pragma unroll
            for (int i = 2; i < 16; ++i) mx = __builtin_fmaxf(mx, acc[tn][i]);
            live[tn] = __any(mx >= thr_hi[tn]);
        }
        if (live[0] || live[1]) {
#pragma unroll
            for (int ks = 0; ks < SX_KS; ++ks) {
                const bf16x8_t al = lds_frag(ks, 1);
                if (live[0]) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[0][ks], acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], ql[0][ks], acc[0], 0, 0, 0);
                }
                if (live[1]) {
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[1][ks], acc[1], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], ql[1][ks], acc[1], 0, 0, 0);
                }
            }
        }
        const int cb = t * SX_CT + h * 4;  // candidate of accumulator entry i, relative to c0: cb + (i >> 2) * 8 + (i & 3)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            if (!live[tn]) continue;  // (wave-uniform) the block was rejected by its hi hi products
            // common case first: the largest of the 16 scores against the threshold (v_max3: 8 instructions), the mask only behind it
            float mx = __builtin_fmaxf(acc[tn][0], acc[tn][1]);
#pragma unroll
            for (int i = 2; i < 16; ++i) mx = __builtin_fmaxf(mx, acc[tn][i]);
            if (!(mx >= thr[tn])) continue;  // per lane; survivors are ~k' / n_seen of the scores
            if (qrow[tn] < 0) continue;
            unsigned m = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) m |= (acc[tn][i] >= thr[tn] ? 1u : 0u) << i;
            if (t * SX_CT + SX_CT > n_split) {  // the stage's last tile: rows past c_end repeat its last row
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (cb + (i >> 2) * 8 + (i & 3) >= n_split) m &= ~(1u << i);
                if (m == 0) continue;
            }
            const int base = atomicAdd(&sx_cnt[wave * SX_QW + tn * 32 + l31], __popc(m));  // LDS: both half-wavefronts hold this query
            // entry r of split sp of a row sits at [(row segcap + r) nsplit + sp]: the merge reads one entry of 64 splits per load
            const int64_t seg = (int64_t)qrow[tn] * segcap * nsplit + sp;
            float* csr = cs + seg;
            int32_t* cir = ci + seg;
            int pos = base;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if ((m >> i) & 1u) {
                    if (pos < segcap) {
                        csr[(int64_t)pos * nsplit] = acc[tn][i];
                        cir[(int64_t)pos * nsplit] = c0i + cb + (i >> 2) * 8 + (i & 3);
                    }
                    ++pos;
                }
            }
        }
    }
    __syncthreads();
    {
        const int qi = qb * SX_QB + threadIdx.x;
        if (qi < Bq) {
            const int n = sx_cnt[threadIdx.x];
            cnt[(int64_t)qi * nsplit + sp] = n < segcap ? n : segcap;
            if (n > segcap) dirty[qi] = 1;  // survivors were lost: topk_redo_rows_kernel recomputes the row exactly
        }
    }
}

// Merge the survivor segments of a row (one per split of the stage, written by topk_filter_bf16x3_kernel) into its running list.
// Lane s of the wavefront walks segment s0 + s: a round hands the wavefront one entry of each of 64 segments.  Order =
// (score desc, index asc), so the arrival order does not matter.
template <int R>
__global__ __launch_bounds__(256) void topk_merge_segments_kernel(const float* __restrict__ cs, const int32_t* __restrict__ ci,
                                                                 const int* __restrict__ cnt, int segcap, int nsplit, int64_t Bq, int k,
                                                                 float* __restrict__ best_s, int32_t* __restrict__ best_i,
                                                                 float* __restrict__ tau, const int* __restrict__ only = nullptr) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= Bq) return;
    if (only && !only[row]) return;  // the rows topk_sort_merge_kernel left over (more entries than its LDS holds)
    RegList<R> L;
    L.load(best_s + row * k, best_i + row * k, k, lane);
    float tl;
    int il;
    L.at(k - 1, &tl, &il);
    for (int s0 = 0; s0 < nsplit; s0 += 64) {
        const int sg = s0 + lane;
        const int n = sg < nsplit ? cnt[row * nsplit + sg] : 0;
        const int64_t seg = row * (int64_t)segcap * nsplit + sg;
        int nmax = n;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
        for (int r = 0; r < nmax; ++r) {
            const float v = (r < n) ? cs[seg + (int64_t)r * nsplit] : -INFINITY;
            const int vi = (r < n) ? ci[seg + (int64_t)r * nsplit] : 0x7fffffff;
            const bool pass = (r < n) && ((v > tl) || (v == tl && vi < il));
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
    }
    L.store(best_s + row * k, best_i + row * k, k, lane);
    if (lane == 0) tau[row] = tl;
}

// The merge of a stage (and the bootstrap's selection) as a SORT: a workgroup per row gathers the row's running list and its survivor segments
// (or a dense block of scores) into LDS as 64-bit keys -- (score descending, index ascending) as ONE ascending integer order: the float's bits
// mapped monotonically and complemented in the high word, the candidate index in the low word --, sorts them with a bitonic network and
// keeps the first k'.  The insertion list of topk_merge_segments_kernel / topk_select_kernel costs ~0.5 us per insertion and a stage inserts
// ~k' ln(growth) entries per row (115 us per stage at k' = 128, 276 us for the bootstrap over 2048 candidates); the network is 45-66 steps of
// <= 4 compare-exchanges per thread whatever the data.  Same total order, so the same list: the result of a call stays bit-identical.
// Rows with more entries than SORT_MAX are left to the insertion kernel (`big[row]` = 1).
constexpr int SORT_MAX = 1024;  // entries a wavefront sorts in its 8 KB of LDS

__device__ __forceinline__ uint64_t topk_key(float s, int32_t idx) {
    uint32_t u = __float_as_uint(s + 0.f);  // -0 -> +0: the float comparisons of the insertion list hold them equal
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending unsigned order = ascending float order (-inf lowest; no NaNs reach here)
    return ((uint64_t)(~u) << 32) | (uint32_t)idx;
}
__device__ __forceinline__ float topk_key_score(uint64_t key) {
    uint32_t u = ~(uint32_t)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

// Bitonic network over EPL x 64 keys held in the registers of one wavefront (element e = r * 64 + lane: register r, lane `lane`), ascending.
// Strides below 64 exchange between lanes (two 32-bit shuffles per key), strides from 64 up between registers of the same lane (static
// indices: the loops are fully unrolled).  No LDS, no barrier: ~8 instructions per key and step.
template <int EPL>
__device__ __forceinline__ void topk_bitonic_regs(uint64_t (&x)[EPL], int lane) {
    constexpr int P = EPL * 64;
#pragma unroll
    for (int size = 2; size <= P; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride < 64) {
#pragma unroll
                for (int r = 0; r < EPL; ++r) {
                    const uint32_t plo = (uint32_t)__shfl_xor((int)(uint32_t)x[r], stride);
                    const uint32_t phi = (uint32_t)__shfl_xor((int)(uint32_t)(x[r] >> 32), stride);
                    const uint64_t other = ((uint64_t)phi << 32) | plo;
                    const bool is_lo = (lane & stride) == 0;
                    const bool up = (((r * 64 + lane) & size) == 0) || size == P;  // size == P: the last merge is ascending everywhere
                    const bool keep_min = is_lo == up;
                    const uint64_t mn = x[r] < other ? x[r] : other, mx = x[r] < other ? other : x[r];
                    x[r] = keep_min ? mn : mx;
                }
            } else {
                constexpr int dummy = 0;
                (void)dummy;
#pragma unroll
                for (int r = 0; r < EPL; ++r) {
                    const int rs = stride >> 6;
                    if ((r & rs) == 0) {
                        const int q = r | rs;
                        const bool up = (((r * 64) & size) == 0) || size == P;
                        const uint64_t a = x[r], b = x[q];
                        const uint64_t mn = a < b ? a : b, mx = a < b ? b : a;
                        x[r] = up ? mn : mx;
                        x[q] = up ? mx : mn;
                    }
                }
            }
        }
    }
}

// One WAVEFRONT per row (four rows per workgroup), everything wave-synchronous -- no workgroup barrier: LDS operations of a wavefront complete
// in order.  (First version: a workgroup per row with a barrier per network step -- 90 us per launch, all of it the latency of ~5 dependent
// global round trips per workgroup with eight workgroups on a CU.)  Lane s holds the count of segment s (+ 64, ...) and fetches entry r of
// its segments for all r at once: nmax rounds of independent, coalesced loads.
// dense == nullptr: entries = the row's list (k) + its survivor segments (cnt / cs / ci as topk_filter_bf16x3_kernel leaves them);
// dense != nullptr: entries = (seen > 0 ? the row's list : nothing) + the ncur scores dense[row * ld + j] of candidates c0 + j
__global__ __launch_bounds__(256) void topk_sort_merge_kernel(const float* __restrict__ cs, const int32_t* __restrict__ ci,
                                                             const int* __restrict__ cnt, int segcap, int nsplit, int64_t Bq, int k,
                                                             float* __restrict__ best_s, int32_t* __restrict__ best_i,
                                                             float* __restrict__ tau, int* __restrict__ big, const float* __restrict__ dense,
                                                             int64_t ld, int ncur, int64_t c0, int seen) {
    __shared__ uint64_t keys_all[4][SORT_MAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= Bq) return;
    uint64_t* keys = keys_all[wave];
    int M;
    int n_s[4] = {0, 0, 0, 0}, off_s[4] = {0, 0, 0, 0};  // nsplit <= 256: lane s owns segments s, s + 64, s + 128, s + 192
    if (dense) {
        M = seen + ncur;
    } else {
        int run = k;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g * 64 < nsplit) {  // wave-uniform
                const int sg = g * 64 + lane;
                n_s[g] = sg < nsplit ? cnt[row * nsplit + sg] : 0;
                int incl = n_s[g];  // inclusive scan over the lanes
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_up(incl, o);
                    if (lane >= o) incl += up;
                }
                off_s[g] = run + incl - n_s[g];
                run += __shfl(incl, 63);
            }
        }
        M = run;
    }
    if (M > SORT_MAX) {  // wave-uniform
        if (lane == 0) big[row] = 1;
        return;
    }
    if (lane == 0) big[row] = 0;
    int P = 64;
    while (P < M) P <<= 1;
    const int nlist = dense ? seen : k;
    for (int i = lane; i < nlist; i += 64) keys[i] = topk_key(best_s[row * k + i], best_i[row * k + i]);
    if (dense) {
        for (int j = lane; j < ncur; j += 64) keys[nlist + j] = topk_key(dense[row * ld + j], (int32_t)(c0 + j));
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g * 64 < nsplit) {
                int nmax = n_s[g];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
                const int64_t seg = row * (int64_t)segcap * nsplit + g * 64 + lane;
                for (int r0 = 0; r0 < nmax; r0 += 8) {  // eight rounds of loads in flight
                    float v[8];
                    int32_t vi[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int r = (r0 + u < n_s[g]) ? r0 + u : 0;  // clamped: branch-free loads (entry 0 exists wherever the result is used)
                        const bool ok = n_s[g] > 0 && g * 64 + lane < nsplit;
                        v[u] = ok ? cs[seg + (int64_t)r * nsplit] : 0.f;
                        vi[u] = ok ? ci[seg + (int64_t)r * nsplit] : 0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (r0 + u < n_s[g]) keys[off_s[g] + r0 + u] = topk_key(v[u], vi[u]);
                }
            }
        }
    }
    for (int i = M + lane; i < P; i += 64) keys[i] = ~0ull;  // worst key: behind every entry
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the network in registers (LDS only served the gather): P / 64 keys per lane
    auto finish = [&](auto epl_tag) {
        constexpr int EPL = decltype(epl_tag)::value;
        uint64_t x[EPL];
#pragma unroll
        for (int r = 0; r < EPL; ++r) x[r] = keys[r * 64 + lane];
        topk_bitonic_regs<EPL>(x, lane);
        const int nout = M < k ? M : k;
#pragma unroll
        for (int r = 0; r < EPL; ++r) {
            const int i = r * 64 + lane;
            if (i < k) {
                // fewer entries than the list holds: the sentinel of the insertion list (topk_select_kernel)
                best_s[row * k + i] = i < nout ? topk_key_score(x[r]) : -INFINITY;
                best_i[row * k + i] = i < nout ? (int32_t)(uint32_t)x[r] : 0x7fffffff;
            }
            if (r == ((k - 1) >> 6)) {  // wave-uniform
                const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(x[r] >> 32), (k - 1) & 63);
                if (lane == 0) tau[row] = M >= k ? topk_key_score((uint64_t)hi << 32) : -INFINITY;
            }
        }
    };
    if (P <= 64) finish(std::integral_constant<int, 1>{});
    else if (P <= 128) finish(std::integral_constant<int, 2>{});
    else if (P <= 256) finish(std::integral_constant<int, 4>{});
    else if (P <= 512) finish(std::integral_constant<int, 8>{});
    else finish(std::integral_constant<int, 16>{});
}

// The last stage of the split pipeline: decide per row whether the k' kept candidates provably contain the exact top-k (see the
// block comment above), re-score them with the exact k-ascending fmaf chain, sort by (score desc, index asc), write the first k.
// One wavefront per row; the query row sits in LDS (read as a broadcast), every lane walks the candidate rows of its entries.
template <int R>
__global__ __launch_bounds__(256) void topk_finalize_exact_kernel(const float* __restrict__ q, const float* __restrict__ cand,
                                                                 int64_t Bq, int E, int k, int kp, const float* __restrict__ list_s,
                                                                 const int32_t* __restrict__ list_i, const float* __restrict__ qnorm2,
                                                                 const unsigned* __restrict__ cmax_bits, int* __restrict__ dirty,
                                                                 float* __restrict__ out_scores, int32_t* __restrict__ out_idx,
                                                                 const int32_t* __restrict__ cand_ids, int32_t* __restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) float fz_smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= Bq) return;
    if (dirty[row]) return;  // survivor list overflowed in some stage: topk_redo_rows_kernel recomputes the row
    RegList<R> L;
    L.load(list_s + row * kp, list_i + row * kp, kp, lane);
    float ak, alast;
    int tmp;
    L.at(k - 1, &ak, &tmp);
    L.at(kp - 1, &alast, &tmp);
    const float m = SX_MREL * sqrtf(qnorm2[row]) * sqrtf(__uint_as_float(*cmax_bits)) * 1.0001f;
    if (alast > -INFINITY && !(alast < ak - 2.f * m)) {  // the band [a_k - 2 m, a_k] may reach beyond the kept candidates
        if (lane == 0) dirty[row] = 1;
        return;
    }
    float* qs = fz_smem + wave * E;
    for (int e = lane; e < E; e += 64) qs[e] = q[row * E + e];
    __builtin_amdgcn_wave_barrier();  // the slice is private to this wavefront (LDS operations of a wavefront execute in order)
    float ex[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int idx = L.i[r];
        float acc = -INFINITY;
        if (idx != 0x7fffffff && L.s[r] > -INFINITY) {
            const float* cr = cand + (int64_t)idx * E;
            acc = 0.f;
            if ((E & 3) == 0 && (reinterpret_cast<uintptr_t>(cand) & 15) == 0) {
                for (int e = 0; e < E; e += 4) {
                    const float4 c4 = *reinterpret_cast<const float4*>(cr + e);
                    acc = fmaf(qs[e], c4.x, acc);
                    acc = fmaf(qs[e + 1], c4.y, acc);
                    acc = fmaf(qs[e + 2], c4.z, acc);
                    acc = fmaf(qs[e + 3], c4.w, acc);
                }
            } else {
                for (int e = 0; e < E; ++e) acc = fmaf(qs[e], cr[e], acc);
            }
        }
        ex[r] = acc;
    }
    RegList<R> M;
    M.fill(lane);
    for (int p = 0; p < kp; ++p) {
        float sc = 0.f;
        int idx = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r == (p >> 6)) {
                sc = __shfl(ex[r], p & 63);
                idx = __shfl(L.i[r], p & 63);
            }
        if (idx == 0x7fffffff || !(sc > -INFINITY)) continue;
        M.insert(sc, idx, kp, lane);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = r * 64 + lane;
        if (p < k) {
            out_scores[row * k + p] = M.s[r];
            out_idx[row * k + p] = M.i[r];
            if (out_ids) out_ids[row * k + p] = cand_ids ? cand_ids[M.i[r]] : M.i[r];
        }
    }
}

int split_kprime(int k) { return (k + 16 + 63) / 64 * 64; }

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


struct SplitPlan {
    FusedPlan f;       // the fp32 plan's geometry with k' in place of k (dense chunk, tau, counts, survivor lists)
    int kp;
    int64_t off_ls, off_li, off_qhi, off_qlo, off_qn, off_seg, total;
};

SplitPlan make_split_plan(int64_t Bq, int64_t N, int k, int E) {
    SplitPlan p;
    p.kp = split_kprime(k);
    p.f = make_fused_plan(Bq, N, p.kp);
    if (p.f.fused) {
        // The dense bootstrap only has to hand the filter stages a first threshold: 16 k' candidates (2048 at k = 100) instead of
        // the fp32 pipeline's 128 MiB chunk -- the streaming select costs ~0.5 us per list insertion, the filter stages that take
        // over the difference run at the bf16 rate.  MERLIN_HIP_TOPK_N0 overrides (experiments).
        const char* e = MH_LAB_ENV("MERLIN_HIP_TOPK_N0");
        int64_t n0 = e ? atoll(e) : 16 * (int64_t)p.kp;
        if (n0 < 16 * (int64_t)p.kp) n0 = 16 * (int64_t)p.kp;
        n0 = (n0 + 127) / 128 * 128;
        if (n0 < p.f.n0) p.f.n0 = n0;
    }
    auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
    int64_t o = p.f.total;
    p.off_ls = o;  o = al(o + Bq * (int64_t)p.kp * 4);
    p.off_li = o;  o = al(o + Bq * (int64_t)p.kp * 4);
    p.off_qhi = o; o = al(o + Bq * (int64_t)E * 2);
    p.off_qlo = o; o = al(o + Bq * (int64_t)E * 2);
    p.off_qn = o;  o = al(o + Bq * 4);
    p.off_seg = o; o = al(o + Bq * (int64_t)SX_MAX_SPLITS * 4);  // survivor counts per (row, split)
    p.total = o;
    return p;
}

int32_t launch_split(const float* x, int64_t n, int E, uint16_t* hi, uint16_t* lo, float* norm2, float* norm2_max, hipStream_t s) {
    const int64_t n4 = n * E / 4;
    const unsigned grid = (unsigned)mh_ceil_div(n4, 256);
    auto* x4 = reinterpret_cast<const float4*>(x);
    auto* h2 = reinterpret_cast<uint2*>(hi);
    auto* l2 = reinterpret_cast<uint2*>(lo);
    auto* mb = reinterpret_cast<unsigned*>(norm2_max);
    if (E == 128) MH_LAUNCH(topk_split_kernel<32>, dim3(grid), dim3(256), 0, s, x4, h2, l2, n4, norm2, mb);
    else if (E == 64) MH_LAUNCH(topk_split_kernel<16>, dim3(grid), dim3(256), 0, s, x4, h2, l2, n4, norm2, mb);
    else if (E == 32) MH_LAUNCH(topk_split_kernel<8>, dim3(grid), dim3(256), 0, s, x4, h2, l2, n4, norm2, mb);
    else if (E == 256) MH_LAUNCH(topk_split_kernel<64>, dim3(grid), dim3(256), 0, s, x4, h2, l2, n4, norm2, mb);
    else {
        mh_set_error("mh_topk_split: E=%d (supported: 32, 64, 128, 256)", E);
        return MH_ERR_INVALID_ARGUMENT;
    }
    return MH_OK;
}

}  // namespace

extern "C" {

int64_t mh_topk_split_workspace_bytes(int64_t Bq, int64_t N, int32_t k, int32_t E) {
    if (Bq <= 0 || N <= 0) return 0;
    return make_split_plan(Bq, N, k, E).total;
}

int32_t mh_topk_split(const float* x, int64_t n, int32_t E, uint16_t* hi, uint16_t* lo, float* norm2, float* norm2_max,
                      mh_stream_t stream) {
    MH_REQUIRE(x && hi && lo, "mh_topk_split: null argument");
    MH_REQUIRE(n >= 0 && E >= 4, "mh_topk_split: bad shape");
    MH_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(hi) & 7) == 0 &&
                   (reinterpret_cast<uintptr_t>(lo) & 7) == 0, "mh_topk_split: x must be 16-byte, hi / lo 8-byte aligned");
    if (n == 0) return MH_OK;
    const int32_t st = launch_split(x, n, E, hi, lo, norm2, norm2_max, mh_stream(stream));
    if (st != MH_OK) return st;
    MH_CHECK_LAUNCH("mh_topk_split");
    return MH_OK;
}

int32_t mh_topk_dot_split(const float* q, const float* cand, const uint16_t* cand_hi, const uint16_t* cand_lo,
                          const float* cand_norm2_max, const int32_t* cand_ids, int64_t Bq, int64_t N, int32_t E, int32_t k,
                          float* out_scores, int32_t* out_ids, int32_t* out_idx, void* workspace, int64_t workspace_bytes,
                          mh_stream_t stream) {
    MH_REQUIRE(q && cand && cand_hi && cand_lo && cand_norm2_max && out_scores && out_idx, "mh_topk_dot_split: null argument");
    MH_REQUIRE(Bq >= 0 && N >= 1 && E >= 1, "mh_topk_dot_split: bad shape");
    MH_REQUIRE(k >= 1 && k <= TOPK_MAX && k <= N, "mh_topk_dot_split: k=%d must be in [1, min(%d, N=%lld)]", k, TOPK_MAX, (long long)N);
    MH_REQUIRE(N < (1ll << 31), "mh_topk_dot_split: N must fit int32 indices");
    if (Bq == 0) return MH_OK;
    const SplitPlan p = make_split_plan(Bq, N, k, E);
    if (!workspace || workspace_bytes < p.total) {
        mh_set_error("mh_topk_dot_split: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)p.total);
        return MH_ERR_WORKSPACE;
    }
    const bool aligned = ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(cand) | reinterpret_cast<uintptr_t>(cand_hi) |
                           reinterpret_cast<uintptr_t>(cand_lo)) & 15) == 0;
    if (!(p.f.fused && (E == 128 || E == 64) && aligned && Bq < (1 << 30))) {
        // small catalogues (the dense path scores everything exactly anyway), other widths: the fp32 pipeline
        return mh_topk_dot(q, cand, cand_ids, Bq, N, E, k, out_scores, out_ids, out_idx, workspace, workspace_bytes, stream);
    }
    hipStream_t s = mh_stream(stream);
    char* ws = static_cast<char*>(workspace);
    float* sc = reinterpret_cast<float*>(ws);
    float* tau = reinterpret_cast<float*>(ws + p.f.off_tau);
    int* cnt = reinterpret_cast<int*>(ws + p.f.off_cnt);
    int* dirty = cnt + Bq;
    float* cs = reinterpret_cast<float*>(ws + p.f.off_cs);
    int32_t* ci = reinterpret_cast<int32_t*>(ws + p.f.off_ci);
    float* ls = reinterpret_cast<float*>(ws + p.off_ls);
    int32_t* li = reinterpret_cast<int32_t*>(ws + p.off_li);
    uint16_t* qhi = reinterpret_cast<uint16_t*>(ws + p.off_qhi);
    uint16_t* qlo = reinterpret_cast<uint16_t*>(ws + p.off_qlo);
    float* qn = reinterpret_cast<float*>(ws + p.off_qn);
    int* segcnt = reinterpret_cast<int*>(ws + p.off_seg);
    const int kp = p.kp;
    const int64_t nc = chunk_cols(Bq, N);
    {
        const int32_t st = launch_split(q, Bq, E, qhi, qlo, qn, nullptr, s);
        if (st != MH_OK) return st;
    }
#define MH_TOPK_BY_R(MACRO)            \
    do {                               \
        if (kp <= 64) MACRO(1);        \
        else if (kp <= 128) MACRO(2);  \
        else if (kp <= 256) MACRO(4);  \
        else if (kp <= 512) MACRO(8);  \
        else MACRO(17);                \
    } while (0)
    // ---- dense bootstrap over [0, n0) in exact fp32 (0.8 % of the candidates): the first k' entries of every row's list ----
    for (int64_t c0 = 0; c0 < p.f.n0; c0 += nc) {
        const int64_t ncur = (c0 + nc < p.f.n0) ? nc : p.f.n0 - c0;
        const int32_t st = mh_internal_gemm_nt(q, E, cand + c0 * E, E, Bq, (int)ncur, E, sc, nc, s);
        if (st != MH_OK) return st;
        const int seen = (int)(c0 < kp ? c0 : kp);
        if (kp <= SORT_MAX / 4 && ncur <= 8 * SORT_MAX) {
            // the bootstrap's selection as sorts (kernel comment): blocks of SORT_MAX - (entries kept so far) dense scores at a time
            for (int64_t j0 = 0; j0 < ncur;) {
                const int have = (int)(c0 + j0 < kp ? c0 + j0 : kp);
                const int take = (int)((ncur - j0 < SORT_MAX - have) ? ncur - j0 : SORT_MAX - have);
                MH_LAUNCH(topk_sort_merge_kernel, dim3((unsigned)mh_ceil_div(Bq, 4)), dim3(256), 0, s, (const float*)nullptr,
                          (const int32_t*)nullptr, (const int*)nullptr, 0, 0, Bq, kp, ls, li, tau, cnt, (const float*)(sc + j0), nc, take, c0 + j0,
                          have);
                j0 += take;
            }
            continue;
        }
#define MH_SEL(R_)                                                                                                          \
    MH_LAUNCH(topk_select_kernel<R_>, dim3((unsigned)mh_ceil_div(Bq, 4)), dim3(256), 0, s, sc, nc, Bq, (int)ncur, c0, kp, seen, \
              ls, li, (const int32_t*)nullptr, 0, (int32_t*)nullptr)
        MH_TOPK_BY_R(MH_SEL);
#undef MH_SEL
    }
    MH_LAUNCH(topk_stage_init_kernel, dim3((unsigned)mh_ceil_div(Bq, 256)), dim3(256), 0, s, ls, kp, Bq, tau, cnt);
    // ---- filter stages on the bf16 pipe (3-term split product), survivors merged by their approximate scores ----
    static bool attr_done = false;
    const size_t lds = (size_t)SX_STAGES * (E == 64 ? SXG<64>::TILE : SXG<128>::TILE);
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_filter_bf16x3_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SX_STAGES * SXG<128>::TILE);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_filter_bf16x3_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SX_STAGES * SXG<64>::TILE);
        attr_done = true;
    }
    const int nqb = (int)mh_ceil_div(Bq, SX_QB);
    const char* senv = MH_LAB_ENV("MERLIN_HIP_TOPK_SPLITS");
    const char* genv = MH_LAB_ENV("MERLIN_HIP_TOPK_GROWTH");
    int growth = genv ? atoi(genv) : 4;
    const char* menv = MH_LAB_ENV("MERLIN_HIP_TOPK_XCD_MAP");
    const int xcd_map = menv ? atoi(menv) : 0;  // measured: the plain order (query block fastest) is 3 % faster than one-split-per-XCD
    if (growth < 2) growth = 2;
    int64_t beg = p.f.n0;
    while (beg < N) {
        int64_t end = beg * growth;
        if (end > N || N - end < beg) end = N;
        const int64_t tiles_all = mh_ceil_div(end - beg, SX_CT);
        // enough workgroups to fill 256 CUs x 2 twice, a multiple of 8 splits (one XCD per split), >= 8 tiles per split
        int nsplit = senv ? atoi(senv) : (int)mh_ceil_div(4 * mh_num_cus(), nqb);
        nsplit = (nsplit + 7) / 8 * 8;
        if (nsplit < 8) nsplit = 8;
        if (nsplit > SX_MAX_SPLITS) nsplit = SX_MAX_SPLITS;
        while (nsplit > 8 && tiles_all / nsplit < 8) nsplit -= 8;
        const int tps = (int)mh_ceil_div(tiles_all, nsplit);
        const int segcap = p.f.cap / nsplit;  // the row's survivor capacity divided among the splits' private segments
        if (E == 64)
            MH_LAUNCH(topk_filter_bf16x3_kernel<64>, dim3((unsigned)(nsplit * nqb)), dim3(SX_NWV * 64), lds, s, cand_hi, cand_lo,
                      (const uint16_t*)qhi, (const uint16_t*)qlo, beg, end, (int)Bq, (const float*)tau, segcnt, cs, ci, segcap, dirty, nqb,
                      nsplit, tps, xcd_map, (const float*)qn, reinterpret_cast<const unsigned*>(cand_norm2_max));
        else
            MH_LAUNCH(topk_filter_bf16x3_kernel<128>, dim3((unsigned)(nsplit * nqb)), dim3(SX_NWV * 64), lds, s, cand_hi, cand_lo,
                      (const uint16_t*)qhi, (const uint16_t*)qlo, beg, end, (int)Bq, (const float*)tau, segcnt, cs, ci, segcap, dirty, nqb,
                      nsplit, tps, xcd_map, (const float*)qn, reinterpret_cast<const unsigned*>(cand_norm2_max));
        // the stage's merge as a sort (a workgroup per row); rows holding more than SORT_MAX entries go through the insertion list
        MH_LAUNCH(topk_sort_merge_kernel, dim3((unsigned)mh_ceil_div(Bq, 4)), dim3(256), 0, s, (const float*)cs, (const int32_t*)ci, (const int*)segcnt, segcap,
                  nsplit, Bq, kp, ls, li, tau, cnt, (const float*)nullptr, (int64_t)0, 0, (int64_t)0, 0);
#define MH_MRG(R_)                                                                                                             \
    MH_LAUNCH(topk_merge_segments_kernel<R_>, dim3((unsigned)mh_ceil_div(Bq, 4)), dim3(256), 0, s, (const float*)cs,              \
              (const int32_t*)ci, (const int*)segcnt, segcap, nsplit, Bq, kp, ls, li, tau, (const int*)cnt)
        MH_TOPK_BY_R(MH_MRG);
#undef MH_MRG
        beg = end;
    }
    // ---- exact re-score of the k' kept candidates, (score desc, index asc) order, first k out; rows that cannot be decided
    //      (or whose survivor list overflowed) are recomputed exactly on the device: no host read-back, graph-capturable ----
    const unsigned* cmax_bits = reinterpret_cast<const unsigned*>(cand_norm2_max);
#define MH_FIN(R_)                                                                                                          \
    MH_LAUNCH(topk_finalize_exact_kernel<R_>, dim3((unsigned)mh_ceil_div(Bq, 4)), dim3(256), (size_t)4 * E * sizeof(float), s, q, \
              cand, Bq, (int)E, (int)k, kp, (const float*)ls, (const int32_t*)li, (const float*)qn, cmax_bits, dirty, out_scores, \
              out_idx, cand_ids, out_ids)
    MH_TOPK_BY_R(MH_FIN);
#undef MH_FIN
#undef MH_TOPK_BY_R
    const size_t rlds = (size_t)4 * 4 * k * sizeof(float);
    if (rlds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_redo_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds);
    }
    MH_LAUNCH(topk_redo_rows_kernel, dim3((unsigned)mh_ceil_div(Bq, 4)), dim3(256), rlds, s, q, cand, Bq, N, (int)E, (int)k,
              (const int*)dirty, out_scores, out_idx, cand_ids, out_ids);
    MH_CHECK_LAUNCH("mh_topk_dot_split");
    return MH_OK;
}

}  // extern "C"

namespace {
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
