// Small HBM-bound kernels around the GEMMs: BCE head loss, L2 row normalisation, row-wise dot,
// dense optimizer step.
#include "mh_common.h"

namespace {

// keras binary_crossentropy on probabilities (BinaryOutput, outputs/classification.py:72-123):
// clip to [1e-7, 1-1e-7]; dlogit = (p - y) * scale is the gradient w.r.t. the pre-sigmoid logit.
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ p, const float* __restrict__ label,
                                                 int64_t M, float scale, float* __restrict__ loss,
                                                 float* __restrict__ dlogit) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float pi = p[i], y = label[i];
    if (loss) {
        const float eps = 1e-7f;
        const float pc = fminf(fmaxf(pi, eps), 1.f - eps);
        loss[i] = -(y * logf(pc) + (1.f - y) * logf(1.f - pc));
    }
    if (dlogit) dlogit[i] = (pi - y) * scale;
}

// Same loss with the batch mean formed on the device in a fixed order: <= 256 workgroups grid-stride over the
// samples and write one partial each, bce_mean_finish_kernel adds the partials in a fixed order -> deterministic.
__global__ __launch_bounds__(256) void bce_mean_kernel(const float* __restrict__ p, const float* __restrict__ label,
                                                      int64_t M, float scale, float* __restrict__ partial,
                                                      float* __restrict__ dlogit) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) {
        const float pi = p[i], y = label[i];
        const float eps = 1e-7f;
        const float pc = fminf(fmaxf(pi, eps), 1.f - eps);
        acc += -(y * logf(pc) + (1.f - y) * logf(1.f - pc));
        if (dlogit) dlogit[i] = (pi - y) * scale;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// one wavefront: lane l adds partials l, l + 64, l + 128, l + 192 in that order, then a fixed xor-shuffle tree
__global__ __launch_bounds__(64) void bce_mean_finish_kernel(const float* __restrict__ partial, int nb, int64_t M,
                                                            float* __restrict__ mean) {
    const int lane = threadIdx.x;
    float t = 0.f;
    for (int i = lane; i < nb; i += 64) t += partial[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
    if (lane == 0) *mean = t / (float)M;
}

// mean of n floats, deterministic: <= 256 workgroups each write one partial (fixed assignment of elements to threads, fixed
// LDS tree), bce_mean_finish_kernel adds the partials in a fixed order and divides
__global__ __launch_bounds__(256) void mean_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ partial) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += x[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// one 64-lane wavefront per row: tf.linalg.l2_normalize(x, axis=-1) = x * rsqrt(max(sum x^2, eps^2))
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, int64_t M, int N, float eps,
                                                    float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * N;
    float s = 0.f;
    for (int k = lane; k < N; k += 64) s = fmaf(xr[k], xr[k], s);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float inv = rsqrtf(fmaxf(s, eps * eps));
    for (int k = lane; k < N; k += 64) y[row * N + k] = xr[k] * inv;
}

// backward of the row normalisation: with s = sum x^2, inv = rsqrt(max(s, eps^2)) and y = x * inv,
//   s >= eps^2:  dx = inv * (dy - y * (y . dy))        (the Jacobian of x / ||x||)
//   s <  eps^2:  dx = inv * dy                          (the clamp makes inv a constant)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        int64_t M, int N, float eps, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * N;
    const float* gr = dy + row * N;
    float s = 0.f, xg = 0.f;
    for (int k = lane; k < N; k += 64) {
        s = fmaf(xr[k], xr[k], s);
        xg = fmaf(xr[k], gr[k], xg);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s += __shfl_xor(s, off);
        xg += __shfl_xor(xg, off);
    }
    const float inv = rsqrtf(fmaxf(s, eps * eps));
    const float c = (s >= eps * eps) ? xg * inv * inv : 0.f;  // (y . dy) * inv, folded: dx = inv * (dy - x * c)
    for (int k = lane; k < N; k += 64) dx[row * N + k] = inv * (gr[k] - xr[k] * c);
}

// DotProduct.call (outputs/base.py:307-310): out[b] = sum_d a[b,d] * b[b,d], k-ascending per
// 16-lane partial chains + shuffle tree.
__global__ __launch_bounds__(256) void rowwise_dot_kernel(const float* __restrict__ a, int64_t lda,
                                                         const float* __restrict__ b, int64_t ldb, int64_t M,
                                                         int N, float* __restrict__ out) {
    const int sub = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    float s = 0.f;
    if (row < M)
        for (int k = sub; k < N; k += 16) s = fmaf(a[row * lda + k], b[row * ldb + k], s);
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (row < M && sub == 0) out[row] = s;
}

// keras SGD: w -= lr*g ; keras Adagrad: acc += g^2 ; w -= lr * g / (sqrt(acc) + eps)
__device__ __forceinline__ void opt_elem(float* w, float gi, float* acc, float* acc2, int64_t i, int opt, float lr,
                                         float eps, float b1, float b2) {
    if (opt == MH_OPT_ADAGRAD) {
        const float a = acc[i] + gi * gi;
        acc[i] = a;
        w[i] -= lr * gi / (sqrtf(a) + eps);
    } else if (opt == MH_OPT_ADAM) {  // keras Adam: lr already carries sqrt(1-b2^t)/(1-b1^t)
        const float m = acc[i] * b1 + gi * (1.f - b1);
        const float v = acc2[i] * b2 + gi * gi * (1.f - b2);
        acc[i] = m;
        acc2[i] = v;
        w[i] -= lr * m / (sqrtf(v) + eps);
    } else {
        w[i] -= lr * gi;
    }
}

__global__ __launch_bounds__(256) void dense_opt_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                       float* __restrict__ acc, float* __restrict__ acc2, int64_t n,
                                                       int opt, float lr, float eps, float b1, float b2,
                                                       const float* __restrict__ lr_dev) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    opt_elem(w, g[i], acc, acc2, i, opt, lr_dev ? *lr_dev : lr, eps, b1, b2);
}

// Adam bias correction on the device so that a captured graph replays with the right step count:
// step[0] += 1;  lr_t[0] = lr * sqrt(1 - b2^t) / (1 - b1^t)   (keras Adam / LazyAdam)
__global__ void adam_tick_kernel(float* __restrict__ step, float lr, float b1, float b2, float* __restrict__ lr_t) {
    const float t = step[0] + 1.f;
    step[0] = t;
    lr_t[0] = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
}

struct MultiOptArgs {
    float* w[MH_MAX_FEATURES];
    const float* g[MH_MAX_FEATURES];
    float* st[MH_MAX_FEATURES];
    float* st2[MH_MAX_FEATURES];
    int64_t n[MH_MAX_FEATURES];
};

// one launch for every dense parameter of the model: blockIdx.y = tensor, grid-stride over its elements
__global__ __launch_bounds__(256) void dense_opt_multi_kernel(const MultiOptArgs a, int opt, float lr, float eps,
                                                             float b1, float b2, const float* __restrict__ lr_dev) {
    const int t = blockIdx.y;
    float* __restrict__ w = a.w[t];
    const float* __restrict__ g = a.g[t];
    const int64_t n = a.n[t];
    const float lr_ = lr_dev ? *lr_dev : lr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        opt_elem(w, g[i], a.st[t], a.st2[t], i, opt, lr_, eps, b1, b2);
}

// Ranking metrics on PRE-SORTED labels (tf/metrics/topk.py:48-195): one thread per query row walks its k
// labels once: hits, precision-weighted hits (AP), discounted gain, ideal gain, first hit.
__global__ __launch_bounds__(256) void topk_metrics_kernel(const float* __restrict__ y, int64_t ld,
                                                          const float* __restrict__ counts, int64_t B, int k,
                                                          float* __restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= B) return;
    const float* yr = y + row * ld;
    const float cnt = counts ? counts[row] : 1.f;
    float hits = 0.f, ap = 0.f, dcg = 0.f, idcg = 0.f, mrr = 0.f;
    for (int j = 0; j < k; ++j) {
        const float rel = yr[j];
        const float disc = 1.f / log2f((float)j + 2.f);
        hits += rel;
        ap += rel * (hits / (float)(j + 1));   // precision@(j+1) at relevant positions
        dcg += rel * disc;
        if ((float)j < cnt) idcg += disc;
        if (mrr == 0.f && rel > 0.f) mrr = 1.f / (float)(j + 1);
    }
    const float denom = fminf(fmaxf(cnt, 1.f), (float)k);
    float* o = out + row * 6;
    o[0] = hits / denom;
    o[1] = hits / (float)k;
    o[2] = ap / denom;
    o[3] = dcg;
    o[4] = idcg > 0.f ? dcg / idcg : 0.f;
    o[5] = mrr;
}

// op 0: a*b   1: a+b   2: a*b + c
__global__ __launch_bounds__(256) void eltwise_kernel(int op, const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ c, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v;
    if (op == 0) v = a[i] * b[i];
    else if (op == 1) v = a[i] + b[i];
    else v = fmaf(a[i], b[i], c[i]);
    out[i] = v;
}

// the same on float4s, OP a template parameter (no per-element branch), one vector per thread: 16-byte aligned operands and
// n % 4 == 0 (every [M, d4] operand of the cross backward).  The scalar kernel moved 2.63 GB in 2.09 ms (0.16 of the HBM peak:
// one dword per thread and a runtime `op` test, round-3 profile).
template <int OP>
__global__ __launch_bounds__(256) void eltwise_vec_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                                         const f32x4* __restrict__ c, f32x4* __restrict__ out, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 x = a[i], y = b[i];
    f32x4 v;
    if (OP == 0) v = x * y;
    else if (OP == 1) v = x + y;
    else {
        const f32x4 z = c[i];
        v.x = fmaf(x.x, y.x, z.x);
        v.y = fmaf(x.y, y.y, z.y);
        v.z = fmaf(x.z, y.z, z.z);
        v.w = fmaf(x.w, y.w, z.w);
    }
    out[i] = v;
}

// ---- Keras activations that are not fused into the GEMM epilogues (MLPBlock(activation=...), blocks/mlp.py:35-139 accepts any
// Keras activation name; relu / sigmoid / linear ride in the epilogues, these run as their own element-wise layer, like
// tf.keras.layers.Activation).  Forward y = f(x); backward dx = dy * f'(x) from the SAVED INPUT (swish / gelu need it).
__device__ __forceinline__ float actx_fwd(float x, int act) {
    switch (act) {
        case MH_ACTX_TANH: return tanhf(x);
        case MH_ACTX_ELU: return x > 0.f ? x : expm1f(x);
        case MH_ACTX_SELU: return 1.0507009873554805f * (x > 0.f ? x : 1.6732632423543772f * expm1f(x));
        case MH_ACTX_SOFTPLUS: return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
        case MH_ACTX_SWISH: return x / (1.f + expf(-x));
        case MH_ACTX_GELU: return 0.5f * x * (1.f + erff(x * 0.7071067811865476f));
        case MH_ACTX_LEAKY_RELU: return x > 0.f ? x : 0.2f * x;
        case MH_ACTX_RELU6: return fminf(fmaxf(x, 0.f), 6.f);
        default: return x;
    }
}
__device__ __forceinline__ float actx_grad(float x, int act) {
    switch (act) {
        case MH_ACTX_TANH: {
            const float t = tanhf(x);
            return 1.f - t * t;
        }
        case MH_ACTX_ELU: return x > 0.f ? 1.f : expf(x);
        case MH_ACTX_SELU: return 1.0507009873554805f * (x > 0.f ? 1.f : 1.6732632423543772f * expf(x));
        case MH_ACTX_SOFTPLUS: return 1.f / (1.f + expf(-x));
        case MH_ACTX_SWISH: {
            const float sg = 1.f / (1.f + expf(-x));
            return sg + x * sg * (1.f - sg);
        }
        case MH_ACTX_GELU:
            return 0.5f * (1.f + erff(x * 0.7071067811865476f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
        case MH_ACTX_LEAKY_RELU: return x > 0.f ? 1.f : 0.2f;
        case MH_ACTX_RELU6: return (x > 0.f && x < 6.f) ? 1.f : 0.f;
        default: return 1.f;
    }
}
// dy == NULL: forward (out = f(x)); else backward (out = dy * f'(x)).  [M, N] operands with their own leading dimensions.
__global__ __launch_bounds__(256) void activation_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                        int64_t lddy, float* __restrict__ out, int64_t ldo, int64_t M, int N,
                                                        int act) {
    const int64_t n = M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / N;
        const int c = (int)(i - r * N);
        const float xv = x[r * ldx + c];
        out[r * ldo + c] = dy ? dy[r * lddy + c] * actx_grad(xv, act) : actx_fwd(xv, act);
    }
}

// l2_batch_regularization (inputs/embedding.py:463-464): loss += factor * sum(out^2), d loss / d out = 2 factor out.
// Pass 1: grad[b, d] += 2 factor out[b, d] over the [B, D] view (row strides ld_out / ld_grad), per-workgroup partial of
// sum(out^2); pass 2 adds the partials in a fixed order into loss_accum (deterministic).
__global__ __launch_bounds__(256) void l2_batch_reg_kernel(const float* __restrict__ out, int64_t ld_out,
                                                          float* __restrict__ grad, int64_t ld_grad, int64_t B, int D4,
                                                          float factor, float* __restrict__ partial) {
    __shared__ float red[256];
    float acc = 0.f;
    const int64_t n = B * D4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / D4;
        const int c = (int)(i - b * D4);
        const f32x4 o = *reinterpret_cast<const f32x4*>(out + b * ld_out + c * 4);
        acc += o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
        if (grad) {
            f32x4* g = reinterpret_cast<f32x4*>(grad + b * ld_grad + c * 4);
            *g = *g + o * (2.f * factor);
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(64) void l2_batch_reg_finish_kernel(const float* __restrict__ partial, int nb, float factor,
                                                                float* __restrict__ loss_accum) {
    const int lane = threadIdx.x;
    float t = 0.f;
    for (int i = lane; i < nb; i += 64) t += partial[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
    if (lane == 0) *loss_accum += factor * t;
}

}  // namespace

__global__ __launch_bounds__(256) void fill_words_kernel(uint32_t* __restrict__ p, uint32_t v, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = v;
}

int32_t mh_fill_words(void* dst, uint32_t value, int64_t words, hipStream_t s) {
    if (words <= 0) return MH_OK;
    int64_t nb = mh_ceil_div(words, 256 * 4);
    const int64_t cap = (int64_t)mh_num_cus() * 8;
    if (nb > cap) nb = cap;
    MH_LAUNCH(fill_words_kernel, dim3((unsigned)nb), dim3(256), 0, s, static_cast<uint32_t*>(dst), value, words);
    MH_CHECK_LAUNCH("mh_fill_words");
    return MH_OK;
}

// float4 copy, ONE 16-byte element per thread, nontemporal on both sides, one workgroup per 4 KB: the streaming rate a
// hand-written kernel reaches on this box -- the "achievable" figure the HBM rooflines of the library are read against,
// beside the 8 TB/s spec.  tools/exp/copy_lab.hip measured the alternatives on a 1 GiB copy: this form 6.6 TB/s (6.25 without
// the nontemporal hint), a workgroup-contiguous grid-stride loop with 4 loads in flight 5.9 TB/s, a whole-grid-stride loop
// 4.3-5.2 TB/s, hipMemcpy D2D 5.1 TB/s -- on gfx950 the hardware workgroup dispatcher beats a persistent loop for streaming.
__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// up to MH_MAX_FEATURES small buffers copied by ONE launch (blockIdx.y = buffer): the batch columns of a step replayed from a
// captured graph are refreshed in place -- 40 separate copies cost ~0.2 ms of host time per 1 ms step
struct CopyManyArgs {
    const uint32_t* src[MH_MAX_FEATURES];
    uint32_t* dst[MH_MAX_FEATURES];
    int64_t words[MH_MAX_FEATURES];
};

__global__ __launch_bounds__(256) void copy_many_kernel(const CopyManyArgs a) {
    const int b = blockIdx.y;
    const uint32_t* __restrict__ s = a.src[b];
    uint32_t* __restrict__ d = a.dst[b];
    const int64_t n = a.words[b];
    const bool vec = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (vec) {
        const int64_t n4 = n / 4;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
            reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
        for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d[i] = s[i];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d[i] = s[i];
    }
}


namespace {

// ---- ConcatFeatures of narrow columns: out[b, off_i + c] = src_i[b, c] -------------------------------------------------------
// (the 13 continuous features of a DLRM batch arrive as 13 [B] / [B, 1] tensors and feed the bottom MLP as one [B, 13] matrix.)
// A workgroup takes 256 rows: every source column is read down the batch (coalesced 1 KB per wavefront for width-1 sources),
// transposed through LDS, and the [256, W] block leaves as contiguous rows; columns [W, pad_to) are written as zeros, so a
// consumer may treat the row as pad_to wide (16-byte aligned rows for the vector paths of the Dense kernels).
struct ConcatArgs {
    const float* src[MH_MAX_FEATURES];
    int64_t ld[MH_MAX_FEATURES];
    int32_t width[MH_MAX_FEATURES];
    int32_t off[MH_MAX_FEATURES];
    int32_t n, W, Wp, all_width_one;
};

__global__ __launch_bounds__(256) void concat_columns_kernel(const ConcatArgs a, int64_t B, float* __restrict__ out, int64_t ldo) {
    extern __shared__ float tile[];  // [256][Wp + 1]
    const int LT = a.Wp + 1;
    const int64_t b0 = (int64_t)blockIdx.x * 256;
    const int64_t b = b0 + threadIdx.x;
    if (b < B) {
        if (a.all_width_one) {
            // the usual case ([B] / [B, 1] columns): 8 loads in flight per thread before the first LDS write (a load -> store
            // loop pays one memory round trip per column: 9.9 us for 13 columns of 64 K rows)
            for (int i0 = 0; i0 < a.n; i0 += 8) {  // uniform loop: the table reads are scalar loads
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (i0 + j < a.n) v[j] = a.src[i0 + j][b * a.ld[i0 + j]];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (i0 + j < a.n) tile[threadIdx.x * LT + i0 + j] = v[j];
            }
        } else {
            for (int i = 0; i < a.n; ++i) {
                const float* __restrict__ s = a.src[i] + b * a.ld[i];
                const int w = a.width[i], o = a.off[i];
                for (int c = 0; c < w; ++c) tile[threadIdx.x * LT + o + c] = s[c];
            }
        }
        for (int c = a.W; c < a.Wp; ++c) tile[threadIdx.x * LT + c] = 0.f;
    }
    __syncthreads();
    const int rows = (int)((B - b0) < 256 ? (B - b0) : 256);
    const int total = rows * a.Wp;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int r = e / a.Wp, c = e - r * a.Wp;
        out[(b0 + r) * ldo + c] = tile[r * LT + c];
    }
}

}  // namespace

extern "C" {

int32_t mh_copy_many(const void* const* src, void* const* dst, const int64_t* bytes, int32_t count, mh_stream_t stream) {
    MH_REQUIRE(count >= 0 && (count == 0 || (src && dst && bytes)), "mh_copy_many: null argument");
    hipStream_t s = mh_stream(stream);
    for (int c0 = 0; c0 < count; c0 += MH_MAX_FEATURES) {
        const int n = count - c0 < MH_MAX_FEATURES ? count - c0 : MH_MAX_FEATURES;
        CopyManyArgs a;
        int64_t big = 0;
        for (int i = 0; i < n; ++i) {
            MH_REQUIRE(bytes[c0 + i] >= 0 && bytes[c0 + i] % 4 == 0, "mh_copy_many: buffer %d is not a whole number of 4-byte words", c0 + i);
            MH_REQUIRE(bytes[c0 + i] == 0 || (src[c0 + i] && dst[c0 + i]), "mh_copy_many: null buffer %d", c0 + i);
            MH_REQUIRE(((reinterpret_cast<uintptr_t>(src[c0 + i]) | reinterpret_cast<uintptr_t>(dst[c0 + i])) & 3) == 0,
                       "mh_copy_many: buffer %d is not 4-byte aligned", c0 + i);
            a.src[i] = static_cast<const uint32_t*>(src[c0 + i]);
            a.dst[i] = static_cast<uint32_t*>(dst[c0 + i]);
            a.words[i] = bytes[c0 + i] / 4;
            if (a.words[i] > big) big = a.words[i];
        }
        for (int i = n; i < MH_MAX_FEATURES; ++i) {
            a.src[i] = nullptr;
            a.dst[i] = nullptr;
            a.words[i] = 0;
        }
        if (big == 0) continue;
        int64_t gx = mh_ceil_div(big, 256 * 16);  // 16 words (64 bytes) per thread of the largest buffer
        if (gx > 64) gx = 64;
        MH_LAUNCH(copy_many_kernel, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, s, a);
    }
    MH_CHECK_LAUNCH("mh_copy_many");
    return MH_OK;
}

int32_t mh_concat_columns(const float* const* src, const int64_t* ld, const int32_t* width, int32_t count, int64_t B, float* out,
                          int64_t ldo, int32_t pad_to, mh_stream_t stream) {
    MH_REQUIRE(src && ld && width && out, "mh_concat_columns: null argument");
    MH_REQUIRE(count >= 1 && count <= MH_MAX_FEATURES, "mh_concat_columns: count=%d outside [1,%d]", count, MH_MAX_FEATURES);
    ConcatArgs a;
    int W = 0;
    for (int i = 0; i < count; ++i) {
        MH_REQUIRE(src[i] && width[i] >= 1 && ld[i] >= width[i], "mh_concat_columns: source %d: null, empty or ld < width", i);
        a.src[i] = src[i];
        a.ld[i] = ld[i];
        a.width[i] = width[i];
        a.off[i] = W;
        W += width[i];
    }
    for (int i = count; i < MH_MAX_FEATURES; ++i) {
        a.src[i] = nullptr;
        a.ld[i] = 0;
        a.width[i] = 0;
        a.off[i] = 0;
    }
    const int Wp = pad_to > W ? pad_to : W;
    MH_REQUIRE(Wp <= 150, "mh_concat_columns: at most 150 output columns (got %d): wider concatenations are plain row copies", Wp);
    MH_REQUIRE(ldo >= Wp, "mh_concat_columns: ldo=%lld < %d", (long long)ldo, Wp);
    if (B <= 0) return MH_OK;
    a.n = count;
    a.W = W;
    a.Wp = Wp;
    a.all_width_one = W == count;
    const size_t lds = (size_t)256 * (Wp + 1) * sizeof(float);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(concat_columns_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    MH_LAUNCH(concat_columns_kernel, dim3((unsigned)mh_ceil_div(B, 256)), dim3(256), lds, mh_stream(stream), a, B, out, ldo);
    MH_CHECK_LAUNCH("mh_concat_columns");
    return MH_OK;
}

int32_t mh_stream_copy(const void* src, void* dst, int64_t bytes, mh_stream_t stream) {
    MH_REQUIRE(src && dst && bytes >= 0 && bytes % 16 == 0, "mh_stream_copy: null argument or size not a multiple of 16");
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0, "mh_stream_copy: 16-byte alignment required");
    if (bytes == 0) return MH_OK;
    const int64_t n4 = bytes / 16;
    const int64_t nb = mh_ceil_div(n4, 256);
    MH_REQUIRE(nb < (1ll << 31), "mh_stream_copy: at most 2^31 workgroups of 4 KB");
    MH_LAUNCH(stream_copy_kernel, dim3((unsigned)nb), dim3(256), 0, mh_stream(stream), static_cast<const f32x4*>(src),
                       static_cast<f32x4*>(dst), n4);
    MH_CHECK_LAUNCH("mh_stream_copy");
    return MH_OK;
}

int32_t mh_l2_batch_reg(const float* out, int64_t ld_out, float* grad, int64_t ld_grad, int64_t B, int32_t D, float factor,
                        float* loss_accum, float* workspace, mh_stream_t stream) {
    MH_REQUIRE(out && loss_accum && workspace, "mh_l2_batch_reg: null argument");
    MH_REQUIRE(D >= 4 && D % 4 == 0 && ld_out % 4 == 0 && (!grad || ld_grad % 4 == 0), "mh_l2_batch_reg: D and strides must be multiples of 4");
    if (B <= 0) return MH_OK;
    int64_t nb = mh_ceil_div(B * (D / 4), 256);
    if (nb > 256) nb = 256;
    MH_LAUNCH(l2_batch_reg_kernel, dim3((unsigned)nb), dim3(256), 0, mh_stream(stream), out, ld_out, grad, ld_grad, B,
                       D / 4, factor, workspace);
    MH_LAUNCH(l2_batch_reg_finish_kernel, dim3(1), dim3(64), 0, mh_stream(stream), workspace, (int)nb, factor, loss_accum);
    MH_CHECK_LAUNCH("mh_l2_batch_reg");
    return MH_OK;
}

int32_t mh_dense_optimizer_step_multi(float* const* w, const float* const* grad, float* const* state,
                                      const int64_t* n, int32_t count, int32_t optimizer, float lr, float eps,
                                      float* const* state2, float beta1, float beta2, const float* lr_device,
                                      mh_stream_t stream) {
    MH_REQUIRE(w && grad && n, "mh_dense_optimizer_step_multi: null argument");
    MH_REQUIRE(count >= 0 && count <= MH_MAX_FEATURES, "mh_dense_optimizer_step_multi: count=%d outside [0,%d]", count, MH_MAX_FEATURES);
    MH_REQUIRE(optimizer == MH_OPT_SGD || (optimizer == MH_OPT_ADAGRAD && state) || (optimizer == MH_OPT_ADAM && state && state2),
               "mh_dense_optimizer_step_multi: bad optimizer/state");
    if (count == 0) return MH_OK;
    MultiOptArgs a;
    int64_t nmax = 0;
    for (int i = 0; i < count; ++i) {
        MH_REQUIRE(w[i] && grad[i] && (optimizer == MH_OPT_SGD || state[i]), "mh_dense_optimizer_step_multi: null tensor %d", i);
        a.w[i] = w[i];
        a.g[i] = grad[i];
        a.st[i] = state ? state[i] : nullptr;
        a.st2[i] = state2 ? state2[i] : nullptr;
        a.n[i] = n[i];
        if (n[i] > nmax) nmax = n[i];
    }
    int64_t bx = mh_ceil_div(nmax, 256);
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    MH_LAUNCH(dense_opt_multi_kernel, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, mh_stream(stream), a,
                       optimizer, lr, eps, beta1, beta2, lr_device);
    MH_CHECK_LAUNCH("mh_dense_optimizer_step_multi");
    return MH_OK;
}

int32_t mh_topk_metrics(const float* labels_sorted, int64_t ld, const float* relevant_counts, int64_t B, int32_t k,
                        float* out, mh_stream_t stream) {
    MH_REQUIRE(labels_sorted && out && k >= 1 && ld >= k, "mh_topk_metrics: bad argument");
    if (B <= 0) return MH_OK;
    MH_LAUNCH(topk_metrics_kernel, dim3((unsigned)mh_ceil_div(B, 256)), dim3(256), 0, mh_stream(stream),
                       labels_sorted, ld, relevant_counts, B, k, out);
    MH_CHECK_LAUNCH("mh_topk_metrics");
    return MH_OK;
}

int32_t mh_eltwise(int32_t op, const float* a, const float* b, const float* c, float* out, int64_t n,
                   mh_stream_t stream) {
    MH_REQUIRE(a && b && out && op >= 0 && op <= 2 && (op != 2 || c), "mh_eltwise: bad argument");
    if (n <= 0) return MH_OK;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out) |
                           (c ? reinterpret_cast<uintptr_t>(c) : 0);
    if (n % 4 == 0 && (bits & 15) == 0) {
        const int64_t n4 = n / 4;
        const dim3 grid((unsigned)mh_ceil_div(n4, 256));
        const f32x4 *a4 = reinterpret_cast<const f32x4*>(a), *b4 = reinterpret_cast<const f32x4*>(b),
                    *c4 = reinterpret_cast<const f32x4*>(c);
        f32x4* o4 = reinterpret_cast<f32x4*>(out);
        if (op == 0) MH_LAUNCH(eltwise_vec_kernel<0>, grid, dim3(256), 0, mh_stream(stream), a4, b4, c4, o4, n4);
        else if (op == 1) MH_LAUNCH(eltwise_vec_kernel<1>, grid, dim3(256), 0, mh_stream(stream), a4, b4, c4, o4, n4);
        else MH_LAUNCH(eltwise_vec_kernel<2>, grid, dim3(256), 0, mh_stream(stream), a4, b4, c4, o4, n4);
    } else {
        MH_LAUNCH(eltwise_kernel, dim3((unsigned)mh_ceil_div(n, 256)), dim3(256), 0, mh_stream(stream), op, a, b,
                           c, out, n);
    }
    MH_CHECK_LAUNCH("mh_eltwise");
    return MH_OK;
}

int32_t mh_bce_fwd_bwd(const float* p, const float* label, int64_t M, float grad_scale, float* loss,
                       float* dlogit, mh_stream_t stream) {
    MH_REQUIRE(p && label, "mh_bce_fwd_bwd: null argument");
    if (M <= 0) return MH_OK;
    MH_LAUNCH(bce_kernel, dim3((unsigned)mh_ceil_div(M, 256)), dim3(256), 0, mh_stream(stream), p, label,
                       M, grad_scale, loss, dlogit);
    MH_CHECK_LAUNCH("mh_bce_fwd_bwd");
    return MH_OK;
}

static int64_t bce_partials(int64_t M) {
    int64_t nb = mh_ceil_div(M, 256);
    return nb > 256 ? 256 : nb;
}

int32_t mh_bce_mean_partial(const float* p, const float* label, int64_t M, float grad_scale, float* dlogit, float* workspace,
                            mh_stream_t stream) {
    MH_REQUIRE(p && label && workspace, "mh_bce_mean_partial: null argument");
    MH_REQUIRE(M >= 1, "mh_bce_mean_partial: empty batch has no mean");
    MH_LAUNCH(bce_mean_kernel, dim3((unsigned)bce_partials(M)), dim3(256), 0, mh_stream(stream), p, label, M, grad_scale,
                       workspace, dlogit);
    MH_CHECK_LAUNCH("mh_bce_mean_partial");
    return MH_OK;
}

int32_t mh_bce_mean_finish(const float* workspace, int64_t M, float* loss_mean, mh_stream_t stream) {
    MH_REQUIRE(workspace && loss_mean && M >= 1, "mh_bce_mean_finish: null argument or empty batch");
    MH_LAUNCH(bce_mean_finish_kernel, dim3(1), dim3(64), 0, mh_stream(stream), workspace, (int)bce_partials(M), M, loss_mean);
    MH_CHECK_LAUNCH("mh_bce_mean_finish");
    return MH_OK;
}

int32_t mh_bce_mean_fwd_bwd(const float* p, const float* label, int64_t M, float grad_scale, float* loss_mean,
                            float* dlogit, float* workspace, mh_stream_t stream) {
    MH_REQUIRE(loss_mean, "mh_bce_mean_fwd_bwd: null argument");
    const int32_t st = mh_bce_mean_partial(p, label, M, grad_scale, dlogit, workspace, stream);
    return st != MH_OK ? st : mh_bce_mean_finish(workspace, M, loss_mean, stream);
}

int32_t mh_activation(int32_t act, const float* x, int64_t ldx, const float* dy, int64_t lddy, float* out, int64_t ldo, int64_t M,
                      int32_t N, mh_stream_t stream) {
    MH_REQUIRE(x && out && N >= 1 && ldx >= N && ldo >= N && (!dy || lddy >= N), "mh_activation: bad argument");
    MH_REQUIRE(act >= MH_ACTX_TANH && act <= MH_ACTX_RELU6, "mh_activation: unknown activation %d", act);
    if (M <= 0) return MH_OK;
    int64_t nb = mh_ceil_div(M * N, 256);
    const int64_t cap = (int64_t)mh_num_cus() * 32;
    if (nb > cap) nb = cap;
    MH_LAUNCH(activation_kernel, dim3((unsigned)nb), dim3(256), 0, mh_stream(stream), x, ldx, dy, lddy, out, ldo, M, N, act);
    MH_CHECK_LAUNCH("mh_activation");
    return MH_OK;
}

// buf[m, col0 : col0 + ncols] = value: the pad columns between a row's width and its 16-byte aligned pitch
__global__ __launch_bounds__(256) void fill_columns_kernel(float* __restrict__ buf, int64_t M, int64_t ld, int32_t col0, int32_t ncols,
                                                           float value) {
    const int64_t n = M * ncols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        buf[(i / ncols) * ld + col0 + (int32_t)(i % ncols)] = value;
}

int32_t mh_fill_columns(float* buf, int64_t M, int64_t ld, int32_t col0, int32_t ncols, float value, mh_stream_t stream) {
    MH_REQUIRE(buf && col0 >= 0 && ncols >= 0 && (int64_t)col0 + ncols <= ld, "mh_fill_columns: bad argument");
    if (M <= 0 || ncols == 0) return MH_OK;
    int64_t nb = mh_ceil_div(M * ncols, 256);
    const int64_t cap = (int64_t)mh_num_cus() * 8;
    if (nb > cap) nb = cap;
    MH_LAUNCH(fill_columns_kernel, dim3((unsigned)nb), dim3(256), 0, mh_stream(stream), buf, M, ld, col0, ncols, value);
    MH_CHECK_LAUNCH("mh_fill_columns");
    return MH_OK;
}

int32_t mh_mean(const float* x, int64_t n, float* mean, float* workspace, mh_stream_t stream) {
    MH_REQUIRE(x && mean && workspace, "mh_mean: null argument");
    MH_REQUIRE(n >= 1, "mh_mean: an empty vector has no mean");
    int64_t nb = mh_ceil_div(n, 256);
    if (nb > 256) nb = 256;
    MH_LAUNCH(mean_partial_kernel, dim3((unsigned)nb), dim3(256), 0, mh_stream(stream), x, n, workspace);
    MH_LAUNCH(bce_mean_finish_kernel, dim3(1), dim3(64), 0, mh_stream(stream), workspace, (int)nb, n, mean);
    MH_CHECK_LAUNCH("mh_mean");
    return MH_OK;
}

int32_t mh_l2norm_rows(const float* x, int64_t M, int32_t N, float eps, float* y, mh_stream_t stream) {
    MH_REQUIRE(x && y && N >= 1, "mh_l2norm_rows: bad argument");
    if (M <= 0) return MH_OK;
    MH_LAUNCH(l2norm_kernel, dim3((unsigned)mh_ceil_div(M, 4)), dim3(256), 0, mh_stream(stream), x, M, N,
                       eps, y);
    MH_CHECK_LAUNCH("mh_l2norm_rows");
    return MH_OK;
}

int32_t mh_l2norm_rows_bwd(const float* x, const float* dy, int64_t M, int32_t N, float eps, float* dx,
                           mh_stream_t stream) {
    MH_REQUIRE(x && dy && dx && N >= 1, "mh_l2norm_rows_bwd: bad argument");
    if (M <= 0) return MH_OK;
    MH_LAUNCH(l2norm_bwd_kernel, dim3((unsigned)mh_ceil_div(M, 4)), dim3(256), 0, mh_stream(stream), x, dy, M,
                       N, eps, dx);
    MH_CHECK_LAUNCH("mh_l2norm_rows_bwd");
    return MH_OK;
}

int32_t mh_rowwise_dot(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t M, int32_t N,
                       float* out, mh_stream_t stream) {
    MH_REQUIRE(a && b && out && N >= 1 && lda >= N && ldb >= N, "mh_rowwise_dot: bad argument");
    if (M <= 0) return MH_OK;
    MH_LAUNCH(rowwise_dot_kernel, dim3((unsigned)mh_ceil_div(M, 16)), dim3(256), 0, mh_stream(stream), a,
                       lda, b, ldb, M, N, out);
    MH_CHECK_LAUNCH("mh_rowwise_dot");
    return MH_OK;
}

int32_t mh_adam_tick(float* step, float lr, float beta1, float beta2, float* lr_t, mh_stream_t stream) {
    MH_REQUIRE(step && lr_t, "mh_adam_tick: null argument");
    MH_LAUNCH(adam_tick_kernel, dim3(1), dim3(1), 0, mh_stream(stream), step, lr, beta1, beta2, lr_t);
    MH_CHECK_LAUNCH("mh_adam_tick");
    return MH_OK;
}

int32_t mh_dense_optimizer_step(float* w, const float* grad, float* state, int64_t n, int32_t optimizer,
                                float lr, float eps, float* state2, float beta1, float beta2,
                                const float* lr_device, mh_stream_t stream) {
    MH_REQUIRE(w && grad, "mh_dense_optimizer_step: null argument");
    MH_REQUIRE(optimizer == MH_OPT_SGD || (optimizer == MH_OPT_ADAGRAD && state) || (optimizer == MH_OPT_ADAM && state && state2),
               "mh_dense_optimizer_step: bad optimizer/state");
    if (n <= 0) return MH_OK;
    MH_LAUNCH(dense_opt_kernel, dim3((unsigned)mh_ceil_div(n, 256)), dim3(256), 0, mh_stream(stream), w,
                       grad, state, state2, n, optimizer, lr, eps, beta1, beta2, lr_device);
    MH_CHECK_LAUNCH("mh_dense_optimizer_step");
    return MH_OK;
}

}  // extern "C"
