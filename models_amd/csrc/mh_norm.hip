// Dropout and BatchNormalization of MLPBlock (gfx950): the optional layers merlin/models/tf/blocks/mlp.py:108-137 puts behind
// each Dense layer -- tf.keras.layers.Dropout(rate) and tf.keras.layers.BatchNormalization() (axis -1, momentum 0.99,
// epsilon 1e-3, non-fused 2-D path: biased batch variance, the same variance updates the moving average).
// Both are HBM-bound element-wise passes over [M, N] activations plus column reductions over the batch:
//   * dropout: the keep mask of element i of call c is a pure function of (seed, c, i) (Philox4x32-10, one call per four
//     elements), so nothing is stored for the backward -- it regenerates the mask from the call number the forward used.  The
//     call counter is device state, advanced by a one-thread kernel behind the forward: a replayed hipGraph drops out anew;
//   * batch norm: column sums in two stages (a workgroup reduces a slab of rows to per-column partials in registers / LDS, a
//     second kernel adds the partials in a FIXED order in double): deterministic, no atomics.
#include "mh_common.h"

namespace {

__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// state: [seed, calls, last].  use_last = 0: forward (call = calls); 1: backward (call = last, the forward's).
// y[i] = keep(i) ? x[i] / (1 - rate) : 0 with keep(i) = (word (i % 4) of Philox(i / 4, call; seed) >= rate * 2^32)
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                     uint32_t thresh, float scale, const uint64_t* __restrict__ state, int use_last) {
    const uint64_t seed = state[0], call = use_last ? state[2] : state[1];
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;  // group of four elements
    const int64_t i = q * 4;
    if (i >= n) return;
    uint32_t w[4];
    philox4((uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)call, (uint32_t)(call >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), w);
    if (i + 4 <= n && ((reinterpret_cast<uintptr_t>(x + i) | reinterpret_cast<uintptr_t>(y + i)) & 15) == 0) {
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + i));
        f32x4 o;
        o.x = w[0] >= thresh ? v.x * scale : 0.f;
        o.y = w[1] >= thresh ? v.y * scale : 0.f;
        o.z = w[2] >= thresh ? v.z * scale : 0.f;
        o.w = w[3] >= thresh ? v.w * scale : 0.f;
        *reinterpret_cast<f32x4*>(y + i) = o;
    } else {
        for (int j = 0; j < 4 && i + j < n; ++j) y[i + j] = w[j] >= thresh ? x[i + j] * scale : 0.f;
    }
}

__global__ void dropout_tick_kernel(uint64_t* state) {
    state[2] = state[1];
    state[1] = state[1] + 1;
}

// ---- column sums over the batch ---------------------------------------------------------------------------------------
// MODE 0: s0 = sum x, s1 = sum x^2 (forward statistics); MODE 1: s0 = sum dy, s1 = sum dy * xhat, xhat = (x - mean) * invstd.
// A workgroup owns SLAB consecutive rows; thread t owns columns t, t + 256, ... (coalesced rows); partials [nblk][2][N].
constexpr int SLAB = 128;

template <int MODE>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ x,
                                                    int64_t ldx, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                    int64_t M, int N, float* __restrict__ part) {
    const int64_t r0 = (int64_t)blockIdx.x * SLAB;
    const int64_t r1 = r0 + SLAB < M ? r0 + SLAB : M;
    for (int c = threadIdx.x; c < N; c += 256) {
        float s0 = 0.f, s1 = 0.f;
        const float mu = MODE ? mean[c] : 0.f, is = MODE ? invstd[c] : 0.f;
        int64_t r = r0;
        for (; r + 4 <= r1; r += 4) {  // four independent loads in flight
            float v[4], u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = a[(r + j) * lda + c];
                u[j] = MODE ? x[(r + j) * ldx + c] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0 += v[j];
                s1 += MODE ? v[j] * ((u[j] - mu) * is) : v[j] * v[j];
            }
        }
        for (; r < r1; ++r) {
            const float v = a[r * lda + c];
            s0 += v;
            s1 += MODE ? v * ((x[r * ldx + c] - mu) * is) : v * v;
        }
        part[((int64_t)blockIdx.x * 2 + 0) * N + c] = s0;
        part[((int64_t)blockIdx.x * 2 + 1) * N + c] = s1;
    }
}

// forward: partials -> batch mean / biased variance -> saved mean, invstd; moving stats updated in place
__global__ __launch_bounds__(256) void bn_stats_finish_kernel(const float* __restrict__ part, int nblk, int64_t M, int N, float eps,
                                                             float momentum, float* __restrict__ save_mean,
                                                             float* __restrict__ save_invstd, float* __restrict__ moving_mean,
                                                             float* __restrict__ moving_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    double s0 = 0.0, s1 = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s0 += (double)part[((int64_t)b * 2 + 0) * N + c];
        s1 += (double)part[((int64_t)b * 2 + 1) * N + c];
    }
    const double mu = s0 / (double)M;
    double var = s1 / (double)M - mu * mu;
    if (var < 0.0) var = 0.0;
    save_mean[c] = (float)mu;
    save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (moving_mean) {
        moving_mean[c] = moving_mean[c] * momentum + (float)mu * (1.f - momentum);
        moving_var[c] = moving_var[c] * momentum + (float)var * (1.f - momentum);
    }
}

// backward: partials -> dbeta = sum dy, dgamma = sum dy xhat
__global__ __launch_bounds__(256) void bn_grad_finish_kernel(const float* __restrict__ part, int nblk, int N, float* __restrict__ dbeta,
                                                            float* __restrict__ dgamma) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    double s0 = 0.0, s1 = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s0 += (double)part[((int64_t)b * 2 + 0) * N + c];
        s1 += (double)part[((int64_t)b * 2 + 1) * N + c];
    }
    dbeta[c] = (float)s0;
    dgamma[c] = (float)s1;
}

// MODE 0: y = (x - mean) invstd gamma + beta.  MODE 1: dx = gamma invstd (dy - dbeta / M - xhat dgamma / M) (training backward).
// MODE 2: dx = dy gamma invstd (inference-mode statistics are constants).
template <int MODE>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                      int64_t lddy, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ dbeta, const float* __restrict__ dgamma, int64_t M,
                                                      int N, float* __restrict__ out, int64_t ldo) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * (int64_t)N) return;
    const int64_t r = idx / N;
    const int c = (int)(idx - r * N);
    const float g = gamma ? gamma[c] : 1.f;
    if (MODE == 0) {
        out[r * ldo + c] = (x[r * ldx + c] - mean[c]) * invstd[c] * g + (beta ? beta[c] : 0.f);
    } else if (MODE == 1) {
        const float xh = (x[r * ldx + c] - mean[c]) * invstd[c];
        const float inv_m = 1.f / (float)M;
        out[r * ldo + c] = g * invstd[c] * (dy[r * lddy + c] - dbeta[c] * inv_m - xh * dgamma[c] * inv_m);
    } else {
        out[r * ldo + c] = dy[r * lddy + c] * g * invstd[c];
    }
}

__global__ __launch_bounds__(256) void bn_invstd_kernel(const float* __restrict__ var, int N, float eps, float* __restrict__ invstd) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N) invstd[c] = 1.f / sqrtf(var[c] + eps);
}

}  // namespace

extern "C" {

int32_t mh_dropout(const float* x, float* y, int64_t n, float rate, uint64_t* rng_state, int32_t backward, mh_stream_t stream) {
    MH_REQUIRE(rate >= 0.f && rate < 1.f, "mh_dropout: rate must be in [0, 1)");
    if (n <= 0) return MH_OK;
    MH_REQUIRE(x && y && rng_state, "mh_dropout: null argument");
    const uint32_t thresh = (uint32_t)fmin(4294967295.0, (double)rate * 4294967296.0);
    const int64_t nb = mh_ceil_div(mh_ceil_div(n, 4), 256);
    hipStream_t s = mh_stream(stream);
    MH_LAUNCH(dropout_kernel, dim3((unsigned)nb), dim3(256), 0, s, x, y, n, thresh, 1.f / (1.f - rate), rng_state, (int)backward);
    if (!backward) MH_LAUNCH(dropout_tick_kernel, dim3(1), dim3(1), 0, s, rng_state);
    MH_CHECK_LAUNCH("mh_dropout");
    return MH_OK;
}

int64_t mh_batchnorm_workspace_bytes(int64_t M, int32_t N) {
    if (M <= 0 || N <= 0) return 0;
    return mh_ceil_div(M, SLAB) * 2 * (int64_t)N * 4 + 256;
}

int32_t mh_batchnorm_fwd(const float* x, int64_t ldx, int64_t M, int32_t N, const float* gamma, const float* beta, float eps,
                         float momentum, int32_t training, float* moving_mean, float* moving_var, float* save_mean,
                         float* save_invstd, float* y, int64_t ldy, void* workspace, int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(x && y && moving_mean && moving_var && save_mean && save_invstd, "mh_batchnorm_fwd: null argument");
    MH_REQUIRE(N >= 1 && ldx >= N && ldy >= N, "mh_batchnorm_fwd: bad shape");
    if (M <= 0) return MH_OK;
    hipStream_t s = mh_stream(stream);
    const int nc = (int)mh_ceil_div(N, 256);
    if (training) {
        const int64_t need = mh_batchnorm_workspace_bytes(M, N);
        if (!workspace || workspace_bytes < need) {
            mh_set_error("mh_batchnorm_fwd: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)need);
            return MH_ERR_WORKSPACE;
        }
        float* part = static_cast<float*>(workspace);
        const int nblk = (int)mh_ceil_div(M, SLAB);
        MH_LAUNCH(colsum_kernel<0>, dim3(nblk), dim3(256), 0, s, x, ldx, (const float*)nullptr, (int64_t)0,
                           (const float*)nullptr, (const float*)nullptr, M, (int)N, part);
        MH_LAUNCH(bn_stats_finish_kernel, dim3(nc), dim3(256), 0, s, part, nblk, M, (int)N, eps, momentum, save_mean,
                           save_invstd, moving_mean, moving_var);
        MH_LAUNCH(bn_apply_kernel<0>, dim3((unsigned)mh_ceil_div(M * N, 256)), dim3(256), 0, s, x, ldx, (const float*)nullptr,
                           (int64_t)0, save_mean, save_invstd, gamma, beta, (const float*)nullptr, (const float*)nullptr, M, (int)N, y, ldy);
    } else {
        MH_LAUNCH(bn_invstd_kernel, dim3(nc), dim3(256), 0, s, moving_var, (int)N, eps, save_invstd);
        MH_LAUNCH(bn_apply_kernel<0>, dim3((unsigned)mh_ceil_div(M * N, 256)), dim3(256), 0, s, x, ldx, (const float*)nullptr,
                           (int64_t)0, moving_mean, save_invstd, gamma, beta, (const float*)nullptr, (const float*)nullptr, M, (int)N, y, ldy);
    }
    MH_CHECK_LAUNCH("mh_batchnorm_fwd");
    return MH_OK;
}

int32_t mh_batchnorm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t M, int32_t N, const float* gamma,
                         const float* save_mean, const float* save_invstd, int32_t training, float* dx, int64_t lddx,
                         float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(x && dy && save_mean && save_invstd && dx && dgamma && dbeta, "mh_batchnorm_bwd: null argument");
    MH_REQUIRE(N >= 1 && ldx >= N && lddy >= N && lddx >= N, "mh_batchnorm_bwd: bad shape");
    if (M <= 0) return MH_OK;
    const int64_t need = mh_batchnorm_workspace_bytes(M, N);
    if (!workspace || workspace_bytes < need) {
        mh_set_error("mh_batchnorm_bwd: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)need);
        return MH_ERR_WORKSPACE;
    }
    hipStream_t s = mh_stream(stream);
    float* part = static_cast<float*>(workspace);
    const int nblk = (int)mh_ceil_div(M, SLAB);
    const int nc = (int)mh_ceil_div(N, 256);
    MH_LAUNCH(colsum_kernel<1>, dim3(nblk), dim3(256), 0, s, dy, lddy, x, ldx, save_mean, save_invstd, M, (int)N, part);
    MH_LAUNCH(bn_grad_finish_kernel, dim3(nc), dim3(256), 0, s, part, nblk, (int)N, dbeta, dgamma);
    if (training)
        MH_LAUNCH(bn_apply_kernel<1>, dim3((unsigned)mh_ceil_div(M * N, 256)), dim3(256), 0, s, x, ldx, dy, lddy, save_mean,
                           save_invstd, gamma, (const float*)nullptr, dbeta, dgamma, M, (int)N, dx, lddx);
    else
        MH_LAUNCH(bn_apply_kernel<2>, dim3((unsigned)mh_ceil_div(M * N, 256)), dim3(256), 0, s, x, ldx, dy, lddy, save_mean,
                           save_invstd, gamma, (const float*)nullptr, dbeta, dgamma, M, (int)N, dx, lddx);
    MH_CHECK_LAUNCH("mh_batchnorm_bwd");
    return MH_OK;
}

}  // extern "C"
