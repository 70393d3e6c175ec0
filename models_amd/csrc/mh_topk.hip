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
#include "mh_common.h"

#include <math.h>

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

int64_t chunk_cols(int64_t Bq, int64_t N) {
    int64_t nc = (32ll << 20) / (Bq > 0 ? Bq : 1);  // 128 MiB of fp32 scores
    if (nc > 65536) nc = 65536;
    if (nc < 1024) nc = 1024;
    nc = nc / 128 * 128;
    if (nc > N) nc = (N + 3) / 4 * 4;
    return nc;
}

}  // namespace

extern "C" {

int64_t mh_topk_workspace_bytes(int64_t Bq, int64_t N, int32_t k) {
    if (Bq <= 0 || N <= 0) return 0;
    return Bq * chunk_cols(Bq, N) * (int64_t)sizeof(float);
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
    if (!workspace || workspace_bytes < Bq * nc * (int64_t)sizeof(float)) {
        mh_set_error("mh_topk_dot: workspace too small (%lld bytes given)", (long long)workspace_bytes);
        return MH_ERR_WORKSPACE;
    }
    hipStream_t s = mh_stream(stream);
    float* sc = static_cast<float*>(workspace);
    const size_t lds = (size_t)4 * 4 * k * sizeof(float);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(topk_select_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            mh_set_error("mh_topk_dot: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
            return MH_ERR_LAUNCH;
        }
    }
    for (int64_t c0 = 0; c0 < N; c0 += nc) {
        const int64_t ncur = (c0 + nc < N) ? nc : N - c0;
        const int32_t st = mh_internal_gemm_nt(q, E, cand + c0 * E, E, Bq, (int)ncur, E, sc, nc, s);
        if (st != MH_OK) return st;
        const int last = (c0 + nc >= N);
        const int seen = (int)(c0 < k ? c0 : k);
        hipLaunchKernelGGL(topk_select_kernel, dim3((unsigned)mh_ceil_div(Bq, 4)), dim3(256), lds, s, sc, nc, Bq,
                           (int)ncur, c0, k, seen, out_scores, out_idx, cand_ids, last, out_ids);
    }
    MH_CHECK_LAUNCH("mh_topk_dot");
    return MH_OK;
}

}  // extern "C"
