/*
 * oracle_c.c -- plain-C restatement of the contraction kernels of the Merlin Models hot path,
 * bit-exact with the HIP kernels' documented accumulation order.
 *
 * TEST INFRASTRUCTURE ONLY (checker + reported CPU baseline); never linked into the product.
 *
 * Why C: the fp32 MFMA GEMMs of libmerlin_hip.so produce each output as ONE k-ascending
 * fp32 fmaf chain.  numpy/BLAS cannot restate that order; this file does, so that scores -- and
 * therefore tf.math.top_k indices (merlin/models/tf/outputs/topk.py:221-223) -- can be compared
 * bit for bit.  Parity pin: top-k tie rule pinned by the reference's
 * tests/unit/tf/utils/test_tf_utils.py:42-75; the fmaf-chain order itself is OUR contract
 * (TensorFlow's Eigen order is unspecified), so score VALUES are "parity unpinned" beyond the
 * reference's 1e-4 tolerance (merlin/models/tf/utils/testing_utils.py:113-119).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; fmaf() is the correctly rounded libm one).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* C[m,n] = sum_k A[m,k] * B[n,k]   (tf.matmul(q, c, transpose_b=True): outputs/topk.py:113-115,
 * blocks/retrieval/base.py:373-375, outputs/contrastive.py:303) */
void oc_gemm_nt_fmaf(const float* A, const float* B, int64_t M, int64_t N, int64_t K, float* C) {
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n) {
            float acc = 0.0f;
            const float* a = A + m * K;
            const float* b = B + n * K;
            for (int64_t k = 0; k < K; ++k) acc = fmaf(a[k], b[k], acc);
            C[m * N + n] = acc;
        }
}

/* Y[m,n] = sum_k X[m,k] * W[k,n]   (keras Dense kernel layout, blocks/mlp.py:275-280), no bias */
void oc_gemm_nn_fmaf(const float* X, const float* W, int64_t M, int64_t N, int64_t K, float* Y) {
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n) {
            float acc = 0.0f;
            for (int64_t k = 0; k < K; ++k) acc = fmaf(X[m * K + k], W[k * N + n], acc);
            Y[m * N + n] = acc;
        }
}

/* tf.math.top_k on one row-major score matrix: values descending, ties -> lower index.
 * Simple O(N k) selection by repeated insertion; fine for oracle-sized inputs. */
void oc_topk_rows(const float* S, int64_t M, int64_t N, int32_t k, float* vals, int32_t* idx) {
    for (int64_t m = 0; m < M; ++m) {
        const float* s = S + m * N;
        float* v = vals + m * (int64_t)k;
        int32_t* ix = idx + m * (int64_t)k;
        int32_t cnt = 0;
        for (int64_t n = 0; n < N; ++n) {
            const float x = s[n];
            if (cnt == k && !(x > v[k - 1])) continue; /* equal score: the earlier index stays */
            int32_t p = cnt < k ? cnt : k - 1;
            while (p > 0 && x > v[p - 1]) {
                v[p] = v[p - 1];
                ix[p] = ix[p - 1];
                --p;
            }
            v[p] = x;
            ix[p] = (int32_t)n;
            if (cnt < k) ++cnt;
        }
    }
}

/* BruteForce.call end to end for one query block (outputs/topk.py:182-237). */
void oc_bruteforce_topk(const float* Q, const float* Cand, const int32_t* ids, int64_t Bq, int64_t N,
                        int64_t E, int32_t k, float* vals, int32_t* out_ids, int32_t* out_idx) {
    float* row = (float*)malloc(sizeof(float) * (size_t)N);
    for (int64_t b = 0; b < Bq; ++b) {
        oc_gemm_nt_fmaf(Q + b * E, Cand, 1, N, E, row);
        oc_topk_rows(row, 1, N, k, vals + b * (int64_t)k, out_idx + b * (int64_t)k);
        for (int32_t j = 0; j < k; ++j) {
            const int32_t i = out_idx[b * (int64_t)k + j];
            out_ids[b * (int64_t)k + j] = ids ? ids[i] : i;
        }
    }
    free(row);
}
