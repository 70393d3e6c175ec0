// Chains of SMALL Dense layers in one kernel (gfx950): y_l = act_l(y_{l-1} W_l + b_l), l = 1..L, every width <= 128.
// Reference: MLPBlock = SequentialBlock of _Dense (merlin/models/tf/blocks/mlp.py:35-139, _Dense.call :275-280), the
// BinaryOutput head Dense(1, sigmoid) (tf/outputs/classification.py:114) and their gradients under the GradientTape of
// BaseModel.train_step (tf/models/base.py:1121-1174).
//
// Why: at batch 64 K the layers 13->128->64 (bottom MLP) and 128->64->32->1 (tail of the top MLP + head) of the DLRM
// config are 9-40 us launches that neither roofline reaches (each re-reads / re-writes [M, <=128] activations and fills
// the chip for a few microseconds).  Here a wavefront owns a 16-row strip and walks the whole chain on it:
//
//   forward   x strip -> LDS slab (A operand) -> v_mfma_f32_16x16x4_f32 against the layer's weights, which sit in LDS in
//             FRAGMENT order (one 256-byte row per (k-step, n-tile): conflict-free, immediate offsets) -> bias / act in
//             the C layout -> store y_l AND write it back to the slab as the next layer's A operand.  No workgroup
//             barrier after the weights are staged.  Each output is ONE k-ascending fmaf chain from zero, exactly like
//             linear_fwd_kernel: results are bit-identical to the layer-by-layer path (N > 4 layers).
//   backward  per strip, per layer (last to first): dz_l = g * act'(y_l) in the C layout; dW_l += y_{l-1}^T dz_l straight
//             from REGISTERS (the C layout of a [16 x 16] tile is both the A operand of y^T and the B operand of dz when
//             the contraction index is numbered (step r, k-slot q) <-> row 4 q + r); db_l += colsum; g = dz_l W_l^T
//             through the slab (k-ascending over n: bit-identical to gemm_nt_kernel).  dz never goes to HBM; dW / db
//             accumulate in registers over all strips of a wavefront, are summed across the workgroup through LDS (every
//             wavefront sums and stores a quarter of the tiles) and leave as ONE partial slab per workgroup; a second small
//             kernel sums the slabs in a fixed order.  LDS operands run two MFMA steps ahead of their use (chain_mfma_loop);
//             loop bodies are specialised on the activation; full strips take pointer-increment load / store paths.
//
// Widths are padded to multiples of 16 at compile time (template signature <P0, P1, P2, P3>); the host picks the
// cheapest signature that covers a chain and falls back to the layer-by-layer kernels when none does.
#include "mh_common.h"

namespace {

constexpr int SA = 132;  // slab row stride in floats (== 4 mod 64: the 16 rows x 4 k-slots of an A fragment hit 64 banks)
constexpr int CW = 4;    // wavefronts per workgroup
constexpr int MAXL = 3;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// The activation is a launch-time value: dispatch ONCE per layer / strip to a loop body specialised on it (a runtime
// `act` inside the element loops costs a handful of scalar branches per element, more than the MFMAs of a strip).
template <int ACT>
__device__ __forceinline__ float chain_act(float v) {
    if (ACT == MH_ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == MH_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}
// FOLDED = false: the form of act_grad_*_kernel (dz = dy * act'(y) of the layer that received dy);
// FOLDED = true: the form of gemm_nt_kernel's epilogue (the producer's derivative folded into dX).  Same value, but the
// sigmoid products associate differently: keeping both forms keeps dX bit-identical to the layer-by-layer path.
template <int ACT, bool FOLDED>
__device__ __forceinline__ float chain_act_grad(float g, float y) {
    if (ACT == MH_ACT_RELU) return y > 0.f ? g : 0.f;
    if (ACT == MH_ACT_SIGMOID) return FOLDED ? g * (y * (1.f - y)) : g * y * (1.f - y);
    return g;
}
template <int A>
struct ActTag {
    static constexpr int value = A;
};
template <typename F>
__device__ __forceinline__ void dispatch_act(int act, F&& f) {
    if (act == MH_ACT_RELU) f(ActTag<MH_ACT_RELU>{});
    else if (act == MH_ACT_SIGMOID) f(ActTag<MH_ACT_SIGMOID>{});
    else f(ActTag<MH_ACT_NONE>{});
}

struct ChainArgs {
    const float* x;  // [M, dims[0]]
    int64_t ldx;
    int64_t M;
    int dims[MAXL + 1];  // actual widths
    const float* W[MAXL];
    const float* b[MAXL];
    int act[MAXL];
    float* y[MAXL];  // forward: outputs; backward: the forward outputs (read)
    int64_t ldy[MAXL];
    // backward only
    const float* g;  // gradient w.r.t. the last layer's output (or its dz when pre_masked)
    int64_t ldg;
    int pre_masked;
    int x_act;
    float* dx;  // [M, dims[0]] or NULL
    int64_t lddx;
    float* slabs;  // [gridDim.x, ptotal] partial dW / db
    int ptotal;
    int poff_w[MAXL], poff_b[MAXL];
};

template <int P0, int P1, int P2, int P3>
struct Sig {
    static constexpr int L = (P3 > 0) ? 3 : ((P2 > 0) ? 2 : 1);
    static constexpr int wfloats = P0 * P1 + P1 * P2 + P2 * P3;
    static constexpr int maxp01 = P0 > P1 ? P0 : P1;
    static constexpr int maxp23 = P2 > P3 ? P2 : P3;
    static constexpr int maxp = maxp01 > maxp23 ? maxp01 : maxp23;
    static_assert(P0 % 16 == 0 && P1 % 16 == 0 && P2 % 16 == 0 && P3 % 16 == 0, "padded widths are multiples of 16");
    static_assert(maxp <= 128, "slab stride");
};

// ---- weights -> LDS, fragment order -------------------------------------------------------------------------------
// forward:  frag[(s * TN + tn) * 64 + lane] = W[k = 4 s + (lane >> 4)][n = 16 tn + (lane & 15)]      (B operand of x W)
// backward: frag[(s * TJ + tj) * 64 + lane] = W[kk = 16 tj + (lane & 15)][n = 4 s + (lane >> 4)]     (B operand of dz W^T)
template <int PIN, int POUT, bool TRANSPOSED>
__device__ __forceinline__ void stage_weights(const float* __restrict__ W, int K, int N, float* __restrict__ frag) {
    constexpr int TT = TRANSPOSED ? PIN / 16 : POUT / 16;
    constexpr int NTH = CW * 64;
    // every load of a batch is issued before the first LDS store (a load -> wait -> store loop pays one L2 round trip per
    // element: measured 30 us of a 60 us kernel); out-of-range elements read W[0] and are zeroed by a select, no branches
    if ((N & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0) {
        constexpr int NCH = PIN * POUT / 4;               // 16-byte chunks: 4 consecutive n of one k
        constexpr int PER = (NCH + NTH - 1) / NTH;
        constexpr int BATCH = PER < 8 ? PER : 8;
        for (int b0 = 0; b0 < PER; b0 += BATCH) {
            f32x4 v[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int c = threadIdx.x + (b0 + j) * NTH;
                int k, n;
                if (TRANSPOSED) {  // chunk = W[kk = 16 tj + l15][4 s .. 4 s + 3]
                    k = 16 * ((c >> 4) % TT) + (c & 15);
                    n = 4 * ((c >> 4) / TT);
                } else {  // chunk = 4 consecutive fragment entries: W[4 s + q][16 tn + 4 j4 ..]
                    const int idx = 4 * c, ln = idx & 63;
                    k = 4 * ((idx >> 6) / TT) + (ln >> 4);
                    n = 16 * ((idx >> 6) % TT) + (ln & 15);
                }
                const bool in = c < NCH && k < K && n < N;
                const f32x4 x = *reinterpret_cast<const f32x4*>(in ? W + (int64_t)k * N + n : W);
                v[j] = in ? x : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int c = threadIdx.x + (b0 + j) * NTH;
                if (c < NCH) {
                    if (TRANSPOSED) {
                        float* d = frag + (c >> 4) * 64 + (c & 15);
                        d[0] = v[j][0];
                        d[16] = v[j][1];
                        d[32] = v[j][2];
                        d[48] = v[j][3];
                    } else {
                        *reinterpret_cast<f32x4*>(frag + 4 * c) = v[j];
                    }
                }
            }
        }
    } else {
        constexpr int PER = (PIN * POUT + NTH - 1) / NTH;
        constexpr int BATCH = PER < 8 ? PER : 8;
        for (int b0 = 0; b0 < PER; b0 += BATCH) {
            float v[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int idx = threadIdx.x + (b0 + j) * NTH;
                const int ln = idx & 63, t = (idx >> 6) % TT, sidx = (idx >> 6) / TT;
                const int k = TRANSPOSED ? 16 * t + (ln & 15) : 4 * sidx + (ln >> 4);
                const int n = TRANSPOSED ? 4 * sidx + (ln >> 4) : 16 * t + (ln & 15);
                const bool in = idx < PIN * POUT && k < K && n < N;
                const float x = *(in ? W + (int64_t)k * N + n : W);
                v[j] = in ? x : 0.f;
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int idx = threadIdx.x + (b0 + j) * NTH;
                if (idx < PIN * POUT) frag[idx] = v[j];
            }
        }
    }
}

// acc[t] += sum_s A[s] * B[s][t] with the LDS operands of step s + DIST already on their way while the MFMAs of step s
// issue: a wavefront issues in order, so reads placed right before their MFMAs (what hipcc schedules when left alone) expose
// the LDS latency once per step -- with one or two wavefronts per SIMD nothing else fills that gap (measured: MFMA busy
// 0.41 / 0.31 of the SIMD cycles in the first version of these kernels).  The scheduling barriers pin the order.
template <int STEPS, int TT>
__device__ __forceinline__ void chain_mfma_loop(const float* __restrict__ ap, const float* __restrict__ bp, f32x4 (&acc)[TT]) {
    constexpr int DIST = STEPS >= 2 ? 2 : 1;  // prefetch distance in steps
    float a[DIST + 1];
    float b[DIST + 1][TT];
#pragma unroll
    for (int s = 0; s < DIST && s < STEPS; ++s) {
        a[s] = ap[4 * s];
#pragma unroll
        for (int t = 0; t < TT; ++t) b[s][t] = bp[(s * TT + t) * 64];
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s + DIST < STEPS) {
            a[(s + DIST) % (DIST + 1)] = ap[4 * (s + DIST)];
#pragma unroll
            for (int t = 0; t < TT; ++t) b[(s + DIST) % (DIST + 1)][t] = bp[((s + DIST) * TT + t) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TT; ++t) acc[t] = mfma16(a[s % (DIST + 1)], b[s % (DIST + 1)][t], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- one forward layer on a wave-private strip -------------------------------------------------------------------
template <int PIN, int POUT>
__device__ __forceinline__ void fwd_layer(float* __restrict__ slab, const float* __restrict__ frag,
                                          const float* __restrict__ bias, int N, int act, float* __restrict__ y,
                                          int64_t ldy, int64_t row0, int64_t M, int lane) {
    constexpr int TN = POUT / 16;
    const int l15 = lane & 15, q = lane >> 4;
    f32x4 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    chain_mfma_loop<PIN / 4, TN>(slab + l15 * SA + q, frag + lane, acc);
    // the bias sits in LDS next to the weights (zero where there is none / past N): read from global memory here, each
    // tile's value was its own dependent L2 round trip per strip (8 + 4 serialized waits: most of a strip's time)
    float bv[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bv[tn] = bias[16 * tn + l15];
    const bool full = row0 + 16 <= M && N == POUT;  // uniform: no bounds tests, no pad columns
    dispatch_act(act, [&](auto tag) {
        constexpr int ACT = decltype(tag)::value;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const bool live = full || 16 * tn + l15 < N;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = chain_act<ACT>(acc[tn][r] + bv[tn]);
                if (!live) v = 0.f;  // pad columns feed the next layer's (zero) pad weights: keep them finite
                acc[tn][r] = v;
                slab[(4 * q + r) * SA + 16 * tn + l15] = v;
            }
        }
    });
    if (full) {
        float* p0 = y + (row0 + 4 * q) * ldy + l15;
        float* p1 = p0 + ldy;
        float* p2 = p1 + ldy;
        float* p3 = p2 + ldy;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            p0[16 * tn] = acc[tn][0];
            p1[16 * tn] = acc[tn][1];
            p2[16 * tn] = acc[tn][2];
            p3[16 * tn] = acc[tn][3];
        }
    } else {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * tn + l15 < N && row0 + 4 * q + r < M) y[(row0 + 4 * q + r) * ldy + 16 * tn + l15] = acc[tn][r];
    }
}

// x strip [16, K0] -> registers (prefetch) -> slab.  VEC: 16-byte loads (K0 % 4 == 0, ld % 4 == 0, aligned base).
template <int P0, bool VEC>
struct XStrip {
    static constexpr int NV = VEC ? (16 * P0 / 4 + 63) / 64 : (16 * P0 + 63) / 64;
    f32x4 v[VEC ? NV : 1];
    float s[VEC ? 1 : NV];
    __device__ __forceinline__ void load(const float* __restrict__ x, int64_t ldx, int K0, int64_t row0, int64_t M, int lane) {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                const int row = c / (P0 / 4), c4 = c % (P0 / 4);
                const bool in = c < 16 * P0 / 4 && row0 + row < M && 4 * c4 < K0;
                const f32x4 t = *reinterpret_cast<const f32x4*>(in ? x + (row0 + row) * ldx + 4 * c4 : x);
                v[i] = in ? t : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = lane + 64 * i;
                const int row = e / P0, col = e % P0;
                const bool in = e < 16 * P0 && row0 + row < M && col < K0;
                const float t = *(in ? x + (row0 + row) * ldx + col : x);
                s[i] = in ? t : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ slab, int lane) const {
        if (VEC) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + 64 * i;
                const int row = c / (P0 / 4), c4 = c % (P0 / 4);
                if (c < 16 * P0 / 4) *reinterpret_cast<f32x4*>(slab + row * SA + 4 * c4) = v[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = lane + 64 * i;
                const int row = e / P0, col = e % P0;
                if (e < 16 * P0) slab[row * SA + col] = s[i];
            }
        }
    }
};

template <int P0, int P1, int P2, int P3, bool VEC>
__global__ __launch_bounds__(CW * 64) void chain_fwd_kernel(const ChainArgs a) {
    using S = Sig<P0, P1, P2, P3>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const w1 = smem;
    float* const w2 = w1 + P0 * P1;
    float* const w3 = w2 + P1 * P2;
    float* const slabs = smem + S::wfloats;
    float* const b1 = slabs + CW * 16 * SA;  // biases, zero-padded to the tile widths
    float* const b2 = b1 + P1;
    float* const b3 = b2 + P2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* const slab = slabs + wave * 16 * SA;
    for (int i = threadIdx.x; i < P1 + P2 + P3; i += CW * 64) {
        const int l = i < P1 ? 0 : (i < P1 + P2 ? 1 : 2);
        const int n = i - (l == 0 ? 0 : (l == 1 ? P1 : P1 + P2));
        b1[i] = (a.b[l] && n < a.dims[l + 1]) ? a.b[l][n] : 0.f;
    }

    // the first strip's loads (HBM) are in flight while the weights (L2) are staged
    const int64_t nstrips = (a.M + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * CW;
    int64_t strip = (int64_t)blockIdx.x * CW + wave;
    XStrip<P0, VEC> xs;
    if (strip < nstrips) xs.load(a.x, a.ldx, a.dims[0], strip * 16, a.M, lane);

    stage_weights<P0, P1, false>(a.W[0], a.dims[0], a.dims[1], w1);
    if constexpr (S::L >= 2) stage_weights<P1, P2, false>(a.W[1], a.dims[1], a.dims[2], w2);
    if constexpr (S::L >= 3) stage_weights<P2, P3, false>(a.W[2], a.dims[2], a.dims[3], w3);
    __syncthreads();

    for (; strip < nstrips; strip += stride) {
        const int64_t row0 = strip * 16;
        xs.store(slab, lane);
        if (strip + stride < nstrips) xs.load(a.x, a.ldx, a.dims[0], (strip + stride) * 16, a.M, lane);
        fwd_layer<P0, P1>(slab, w1, b1, a.dims[1], a.act[0], a.y[0], a.ldy[0], row0, a.M, lane);
        if constexpr (S::L >= 2) fwd_layer<P1, P2>(slab, w2, b2, a.dims[2], a.act[1], a.y[1], a.ldy[1], row0, a.M, lane);
        if constexpr (S::L >= 3) fwd_layer<P2, P3>(slab, w3, b3, a.dims[3], a.act[2], a.y[2], a.ldy[2], row0, a.M, lane);
    }
}

// ---- backward ------------------------------------------------------------------------------------------------------
// [16 x P] strip of a row-major matrix in the C layout of the MFMA: tile t, register r <-> row 4 q + r, column 16 t + l15
template <int P>
struct CTiles {
    static constexpr int T = P / 16;
    f32x4 t[T > 0 ? T : 1];
    __device__ __forceinline__ void load(const float* __restrict__ src, int64_t ld, int width, int64_t row0, int64_t M, int lane) {
        const int l15 = lane & 15, q = lane >> 4;
        if (row0 + 16 <= M && width == P) {
            // full strip, no pad columns (uniform test): four row pointers, every tile an immediate offset -- no per-element
            // address arithmetic or bounds selects (they were ~8 VALU instructions per loaded element)
            const float* p0 = src + (row0 + 4 * q) * ld + l15;
            const float* p1 = p0 + ld;
            const float* p2 = p1 + ld;
            const float* p3 = p2 + ld;
#pragma unroll
            for (int i = 0; i < T; ++i) {
                t[i][0] = p0[16 * i];
                t[i][1] = p1[16 * i];
                t[i][2] = p2[16 * i];
                t[i][3] = p3[16 * i];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int col = 16 * i + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + 4 * q + r;
                const bool in = col < width && row < M;
                const float v = *(in ? src + row * ld + col : src);  // branch-free: all loads of a strip in flight together
                t[i][r] = in ? v : 0.f;
            }
        }
    }
    // store to a row-major matrix (same fast path)
    __device__ __forceinline__ void store(float* __restrict__ dst, int64_t ld, int width, int64_t row0, int64_t M, int lane) const {
        const int l15 = lane & 15, q = lane >> 4;
        if (row0 + 16 <= M && width == P) {
            float* p0 = dst + (row0 + 4 * q) * ld + l15;
            float* p1 = p0 + ld;
            float* p2 = p1 + ld;
            float* p3 = p2 + ld;
#pragma unroll
            for (int i = 0; i < T; ++i) {
                p0[16 * i] = t[i][0];
                p1[16 * i] = t[i][1];
                p2[16 * i] = t[i][2];
                p3[16 * i] = t[i][3];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + 4 * q + r;
                const int col = 16 * i + l15;
                if (col < width && row < M) dst[row * ld + col] = t[i][r];
            }
    }
};

// g_in[tj] = sum_n dz[row][n] W[kk][n]: A = dz through the slab, B = transposed weight fragments
template <int PIN, int POUT>
__device__ __forceinline__ void bwd_dx(const CTiles<POUT>& dz, float* __restrict__ slab, const float* __restrict__ fragT,
                                       CTiles<PIN>& gin, int lane) {
    constexpr int TJ = PIN / 16;
    const int l15 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int tn = 0; tn < POUT / 16; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(4 * q + r) * SA + 16 * tn + l15] = dz.t[tn][r];
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) gin.t[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    chain_mfma_loop<POUT / 4, TJ>(slab + l15 * SA + q, fragT + lane, gin.t);
}

// dW[ti][tn] += y^T dz over the strip's 16 rows, from registers; db[tn] += column sums (per-lane partial over its 4 rows)
template <int PIN, int POUT>
__device__ __forceinline__ void bwd_dw(const CTiles<PIN>& yin, const CTiles<POUT>& dz, f32x4* __restrict__ dw,
                                       float* __restrict__ db) {
#pragma unroll
    for (int ti = 0; ti < PIN / 16; ++ti)
#pragma unroll
        for (int tn = 0; tn < POUT / 16; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                dw[ti * (POUT / 16) + tn] = mfma16(yin.t[ti][r], dz.t[tn][r], dw[ti * (POUT / 16) + tn]);
#pragma unroll
    for (int tn = 0; tn < POUT / 16; ++tn) db[tn] += (dz.t[tn][0] + dz.t[tn][1]) + (dz.t[tn][2] + dz.t[tn][3]);
}

template <int P, bool FOLDED = true>
__device__ __forceinline__ void mask_tiles(CTiles<P>& g, const CTiles<P>& y, int act) {
    if (act == MH_ACT_NONE) return;
    dispatch_act(act, [&](auto tag) {
        constexpr int ACT = decltype(tag)::value;
#pragma unroll
        for (int i = 0; i < P / 16; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) g.t[i][r] = chain_act_grad<ACT, FOLDED>(g.t[i][r], y.t[i][r]);
    });
}

// dx = (dz_1 W_1^T) * x_act'(x), stored from the C layout
template <int P0, int P1>
__device__ __forceinline__ void chain_dx_out(const ChainArgs& a, const CTiles<P1>& g1, const CTiles<P0>& y0,
                                             float* __restrict__ slab, const float* __restrict__ w1, int64_t row0, int lane) {
    CTiles<P0> g0;
    bwd_dx<P0, P1>(g1, slab, w1, g0, lane);
    mask_tiles<P0>(g0, y0, a.x_act);
    g0.store(a.dx, a.lddx, a.dims[0], row0, a.M, lane);
}

// everything a strip reads from HBM (prefetched one strip ahead)
template <int P0, int P1, int P2, int P3>
struct StripIn {
    using S = Sig<P0, P1, P2, P3>;
    static constexpr int PL = (S::L == 3) ? P3 : ((S::L == 2) ? P2 : P1);
    CTiles<P0> y0;
    CTiles<P1> y1;
    CTiles<(P2 > 0 ? P2 : 16)> y2;
    CTiles<(P3 > 0 ? P3 : 16)> y3;
    CTiles<PL> g;
    __device__ __forceinline__ void load(const ChainArgs& a, int64_t row0, int lane) {
        g.load(a.g, a.ldg, a.dims[S::L], row0, a.M, lane);
        y0.load(a.x, a.ldx, a.dims[0], row0, a.M, lane);
        // y_L is only needed for its activation derivative; y_l (l < L) also feed dW_{l+1}
        const bool need_last = !a.pre_masked && a.act[S::L - 1] != MH_ACT_NONE;
        if (S::L > 1 || need_last) y1.load(a.y[0], a.ldy[0], a.dims[1], row0, a.M, lane);
        if (S::L >= 2 && (S::L > 2 || need_last)) y2.load(a.y[1], a.ldy[1], a.dims[2], row0, a.M, lane);
        if (S::L >= 3 && need_last) y3.load(a.y[2], a.ldy[2], a.dims[3], row0, a.M, lane);
    }
};

// sum a register tile array over the CW wavefronts of the workgroup (result in wave 0), NB tiles per round
template <int NT>
__device__ __forceinline__ void wg_reduce_tiles(f32x4* __restrict__ v, float* __restrict__ scratch, int wave, int lane) {
    constexpr int NB = 8;
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += NB) {
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (t0 + j < NT) *reinterpret_cast<f32x4*>(scratch + (((wave - 1) * NB + j) * 64 + lane) * 4) = v[t0 + j];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (t0 + j < NT) {
#pragma unroll
                    for (int w = 0; w < CW - 1; ++w)
                        v[t0 + j] += *reinterpret_cast<const f32x4*>(scratch + ((w * NB + j) * 64 + lane) * 4);
                }
        }
    }
}

// dW tiles of one layer: summed over the CW wavefronts (fixed order) and written to the workgroup's slab by ALL wavefronts --
// every wavefront parks its tiles in LDS (PART at a time), then each sums and stores a quarter of them.  (Summing on wave 0
// alone, eight tiles a round, was ~6 us of a 55 us kernel.)
template <int PIN, int POUT>
__device__ __forceinline__ void wg_reduce_store_dw(const f32x4* __restrict__ dw, float* __restrict__ scratch,
                                                   float* __restrict__ out, int K, int N, int wave, int lane) {
    constexpr int TN_ = POUT / 16, NT = (PIN / 16) * TN_;
    constexpr int PART = 16;
    const int l15 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += PART) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PART; ++j)
            if (t0 + j < NT) *reinterpret_cast<f32x4*>(scratch + ((wave * PART + j) * 64 + lane) * 4) = dw[t0 + j];
        __syncthreads();
        for (int j = wave; j < PART && t0 + j < NT; j += CW) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(scratch + ((0 * PART + j) * 64 + lane) * 4);
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(scratch + ((1 * PART + j) * 64 + lane) * 4);
            const f32x4 s2 = *reinterpret_cast<const f32x4*>(scratch + ((2 * PART + j) * 64 + lane) * 4);
            const f32x4 s3 = *reinterpret_cast<const f32x4*>(scratch + ((3 * PART + j) * 64 + lane) * 4);
            const f32x4 v = (s0 + s1) + (s2 + s3);
            const int t = t0 + j, ti = t / TN_, tn = t % TN_;
            const int n = 16 * tn + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = 16 * ti + 4 * q + r;
                if (kk < K && n < N) out[kk * N + n] = v[r];
            }
        }
    }
}
static_assert(CW == 4, "wg_reduce_store_dw sums four wavefronts");

template <int P0, int P1, int P2, int P3, bool NEED_DX>
__global__ __launch_bounds__(CW * 64) void chain_bwd_kernel(const ChainArgs a) {
    using S = Sig<P0, P1, P2, P3>;
    constexpr int Q2 = P2 > 0 ? P2 : 16, Q3 = P3 > 0 ? P3 : 16;
    constexpr int NT1 = (P0 / 16) * (P1 / 16), NT2 = S::L >= 2 ? (P1 / 16) * (P2 / 16) : 0,
                  NT3 = S::L >= 3 ? (P2 / 16) * (P3 / 16) : 0;
    constexpr int NTW = NT1 + NT2 + NT3;
    constexpr int NB1 = P1 / 16, NB2 = S::L >= 2 ? P2 / 16 : 0, NB3 = S::L >= 3 ? P3 / 16 : 0;
    constexpr int NTB = (NB1 + NB2 + NB3 + 3) / 4;  // db partials packed four to a register tile for the reduction
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // transposed weight fragments (layer 1 only when dx is wanted), then the wave slabs; the reduction scratch at the
    // end of the kernel overlays everything
    float* const w1 = smem;
    float* const w2 = w1 + (NEED_DX ? P0 * P1 : 0);
    float* const w3 = w2 + P1 * P2;
    float* const slabs = w3 + P2 * P3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* const slab = slabs + wave * 16 * SA;

    const int64_t nstrips = (a.M + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * CW;
    int64_t strip = (int64_t)blockIdx.x * CW + wave;
    StripIn<P0, P1, P2, P3> nxt;
    if (strip < nstrips) nxt.load(a, strip * 16, lane);  // in flight while the weights are staged

    if constexpr (NEED_DX) stage_weights<P0, P1, true>(a.W[0], a.dims[0], a.dims[1], w1);
    if constexpr (S::L >= 2) stage_weights<P1, P2, true>(a.W[1], a.dims[1], a.dims[2], w2);
    if constexpr (S::L >= 3) stage_weights<P2, P3, true>(a.W[2], a.dims[2], a.dims[3], w3);
    __syncthreads();

    f32x4 dw[NTW];
    float db[NB1 + NB2 + NB3];
#pragma unroll
    for (int i = 0; i < NTW; ++i) dw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NB1 + NB2 + NB3; ++i) db[i] = 0.f;

    for (; strip < nstrips; strip += stride) {
        const int64_t row0 = strip * 16;
        StripIn<P0, P1, P2, P3> cur = nxt;
        // the next strip's loads are issued after the last layer (its g / y_L tiles are dead by then: fewer live registers)
        const bool more = strip + stride < nstrips;

        if constexpr (S::L == 3) {
            if (!a.pre_masked) mask_tiles<Q3, false>(cur.g, cur.y3, a.act[2]);
            bwd_dw<Q2, Q3>(cur.y2, cur.g, dw + NT1 + NT2, db + NB1 + NB2);
            CTiles<Q2> g2;
            bwd_dx<Q2, Q3>(cur.g, slab, w3, g2, lane);
            if (more) nxt.load(a, (strip + stride) * 16, lane);
            mask_tiles<Q2>(g2, cur.y2, a.act[1]);
            bwd_dw<P1, Q2>(cur.y1, g2, dw + NT1, db + NB1);
            CTiles<P1> g1;
            bwd_dx<P1, Q2>(g2, slab, w2, g1, lane);
            mask_tiles<P1>(g1, cur.y1, a.act[0]);
            bwd_dw<P0, P1>(cur.y0, g1, dw, db);
            if constexpr (NEED_DX) chain_dx_out<P0, P1>(a, g1, cur.y0, slab, w1, row0, lane);
        } else {
            if (!a.pre_masked) mask_tiles<Q2, false>(cur.g, cur.y2, a.act[1]);
            bwd_dw<P1, Q2>(cur.y1, cur.g, dw + NT1, db + NB1);
            CTiles<P1> g1;
            bwd_dx<P1, Q2>(cur.g, slab, w2, g1, lane);
            if (more) nxt.load(a, (strip + stride) * 16, lane);
            mask_tiles<P1>(g1, cur.y1, a.act[0]);
            bwd_dw<P0, P1>(cur.y0, g1, dw, db);
            if constexpr (NEED_DX) chain_dx_out<P0, P1>(a, g1, cur.y0, slab, w1, row0, lane);
        }
    }

    // ---- workgroup reduction (fixed wave order) and the partial slab ------------------------------------------------
    float* const out = a.slabs + (int64_t)blockIdx.x * a.ptotal;
    wg_reduce_store_dw<P0, P1>(dw, smem, out + a.poff_w[0], a.dims[0], a.dims[1], wave, lane);
    if constexpr (S::L >= 2) wg_reduce_store_dw<P1, P2>(dw + NT1, smem, out + a.poff_w[1], a.dims[1], a.dims[2], wave, lane);
    if constexpr (S::L >= 3) wg_reduce_store_dw<P2, P3>(dw + NT1 + NT2, smem, out + a.poff_w[2], a.dims[2], a.dims[3], wave, lane);
    f32x4 dbt[NTB];
#pragma unroll
    for (int i = 0; i < NTB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) dbt[i][r] = (4 * i + r < NB1 + NB2 + NB3) ? db[4 * i + r] : 0.f;
    wg_reduce_tiles<NTB>(dbt, smem, wave, lane);
    if (wave != 0) return;
    // db: sum the four k-slot partials of every column (lanes l15, l15+16, l15+32, l15+48)
#pragma unroll
    for (int i = 0; i < NB1 + NB2 + NB3; ++i) {
        float v = dbt[i / 4][i % 4];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        const int layer = (i < NB1) ? 0 : ((i < NB1 + NB2) ? 1 : 2);
        const int tn = (i < NB1) ? i : ((i < NB1 + NB2) ? i - NB1 : i - NB1 - NB2);
        const int n = 16 * tn + (lane & 15);
        if (lane < 16 && n < a.dims[layer + 1]) out[a.poff_b[layer] + n] = v;
    }
}

struct ReduceArgs {
    const float* slabs;
    int G, ptotal, L;
    int poff_w[MAXL], poff_b[MAXL], nw[MAXL], nb[MAXL];
    float* dW[MAXL];
    float* db[MAXL];
};

// out[p] = sum_g slabs[g][p] in a fixed order: a workgroup owns 64 parameters; 16 slab classes (slab k belongs to class k % 16)
// with 4 independent chains each, then a fixed-order tree through LDS.  A workgroup is 4 wavefronts, each walking 4 of the 16
// classes: 16 independent load chains per thread.  (First version: 1024-thread workgroups, one class per wavefront -- the same
// sums, but a 16-wavefront workgroup needs 16 free wave slots on ONE compute unit: beside the persistent sparse-apply kernel,
// which holds 5 of 5 wave slots' worth of registers on every SIMD until it ends, the launch waited for that kernel to finish --
// 5 us alone, 155 us in the step, on the launch stream's critical path.  A 4-wavefront workgroup of 32 registers fits beside it.)
// The order of every addition is the one of the first version: results are bit-identical.
__global__ __launch_bounds__(256) void chain_reduce_kernel(const ReduceArgs a) {
    __shared__ float red[1024];
    const int o = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + o;
#pragma unroll 1  // 32 registers: what the sparse-apply kernel leaves free on a SIMD (see above); the 4 chains of a class are in flight
    for (int c = 0; c < 4; ++c) {
        const int grp = w * 4 + c;  // the class (the first version's wavefront index)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (p < a.ptotal) {
            const int64_t step = (int64_t)a.ptotal;  // floats per slab
            const float* q = a.slabs + (int64_t)grp * step + p;
            int k = grp;
            for (; k + 48 < a.G; k += 64, q += 64 * step) {
                s0 += q[0];
                s1 += q[16 * step];
                s2 += q[32 * step];
                s3 += q[48 * step];
            }
            for (; k < a.G; k += 16, q += 16 * step) s0 += q[0];
        }
        red[grp * 64 + o] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (w != 0 || p >= a.ptotal) return;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 16; g += 4) v += (red[g * 64 + o] + red[(g + 1) * 64 + o]) + (red[(g + 2) * 64 + o] + red[(g + 3) * 64 + o]);
    for (int l = 0; l < a.L; ++l) {
        if (p >= a.poff_w[l] && p < a.poff_w[l] + a.nw[l]) {
            a.dW[l][p - a.poff_w[l]] = v;
            return;
        }
        if (p >= a.poff_b[l] && p < a.poff_b[l] + a.nb[l]) {
            if (a.db[l]) a.db[l][p - a.poff_b[l]] = v;
            return;
        }
    }
}

// ---- host side: signature table ----------------------------------------------------------------------------------
struct SigEntry {
    int p[4];
    void (*fwd_vec)(const ChainArgs);
    void (*fwd_sca)(const ChainArgs);
    void (*bwd_dx)(const ChainArgs);
    void (*bwd_nodx)(const ChainArgs);
};

#define MH_CHAIN_SIG(P0, P1, P2, P3)                                                                        \
    {                                                                                                       \
        {P0, P1, P2, P3}, chain_fwd_kernel<P0, P1, P2, P3, true>, chain_fwd_kernel<P0, P1, P2, P3, false>, \
            chain_bwd_kernel<P0, P1, P2, P3, true>, chain_bwd_kernel<P0, P1, P2, P3, false>                \
    }

// ordered by cost: the first entry that covers a chain is used.  Register budget of the backward: sum of
// P_{l-1} P_l / 64 accumulator registers per lane must stay below ~200 (one wavefront per SIMD, 512 registers).
const SigEntry SIGS[] = {
    MH_CHAIN_SIG(16, 32, 16, 0),   MH_CHAIN_SIG(16, 64, 32, 0),  MH_CHAIN_SIG(16, 128, 64, 0),
    MH_CHAIN_SIG(64, 32, 16, 0),   MH_CHAIN_SIG(64, 64, 32, 0),  MH_CHAIN_SIG(128, 64, 32, 0),
    MH_CHAIN_SIG(64, 32, 16, 16),  MH_CHAIN_SIG(64, 64, 32, 16), MH_CHAIN_SIG(128, 64, 32, 16),
};
constexpr int NSIGS = sizeof(SIGS) / sizeof(SIGS[0]);

int pad16(int v) { return (v + 15) / 16 * 16; }

const SigEntry* find_sig(int L, const int32_t* dims) {
    if (L < 2 || L > MAXL) return nullptr;  // a single layer gains nothing over linear_fwd_kernel
    for (int i = 0; i < NSIGS; ++i) {
        const int* p = SIGS[i].p;
        const int sl = p[3] > 0 ? 3 : 2;
        if (sl != L) continue;
        bool ok = true;
        for (int j = 0; j <= L; ++j) ok = ok && dims[j] >= 1 && pad16(dims[j]) <= p[j];
        if (ok) return &SIGS[i];
    }
    return nullptr;
}

int wfloats(const int* p) { return p[0] * p[1] + p[1] * p[2] + p[2] * p[3]; }

int chain_ptotal(int L, const int32_t* dims, int* poff_w, int* poff_b) {
    int off = 0;
    for (int l = 0; l < L; ++l) {
        poff_w[l] = off;
        off += dims[l] * dims[l + 1];
        poff_b[l] = off;
        off += dims[l + 1];
    }
    return off;
}

int chain_grid(int64_t M) {
    const int64_t tiles = mh_ceil_div(mh_ceil_div(M, 16), CW);
    const int64_t cap = (int64_t)mh_num_cus();
    return (int)(tiles < cap ? tiles : cap);
}

}  // namespace

extern "C" {

int32_t mh_mlp_chain_supported(int32_t L, const int32_t* dims) {
    if (!dims) return 0;
    return find_sig(L, dims) != nullptr ? 1 : 0;
}

int64_t mh_mlp_chain_bwd_workspace_bytes(int64_t M, int32_t L, const int32_t* dims) {
    if (!dims || !find_sig(L, dims) || M < 0) return -1;
    int pw[MAXL], pb[MAXL];
    const int pt = chain_ptotal(L, dims, pw, pb);
    return (int64_t)mh_num_cus() * pt * (int64_t)sizeof(float) + 256;
}

int32_t mh_mlp_chain_fwd(const float* x, int64_t ldx, int64_t M, int32_t L, const int32_t* dims,
                         const float* const* W, const float* const* b, const int32_t* act, float* const* y,
                         const int64_t* ldy, mh_stream_t stream) {
    MH_REQUIRE(x && dims && W && act && y && ldy, "mh_mlp_chain_fwd: null argument");
    const SigEntry* sig = find_sig(L, dims);
    if (!sig) {
        mh_set_error("mh_mlp_chain_fwd: no fused kernel covers this chain (L=%d): use mh_linear_bias_act_fwd per layer", L);
        return MH_ERR_UNSUPPORTED;
    }
    MH_REQUIRE(M >= 0 && ldx >= dims[0], "mh_mlp_chain_fwd: bad shape M=%lld ldx=%lld", (long long)M, (long long)ldx);
    ChainArgs a{};
    a.x = x;
    a.ldx = ldx;
    a.M = M;
    for (int l = 0; l <= L; ++l) a.dims[l] = dims[l];
    for (int l = 0; l < L; ++l) {
        MH_REQUIRE(W[l] && y[l] && ldy[l] >= dims[l + 1], "mh_mlp_chain_fwd: layer %d: null W / y or ldy < N", l);
        MH_REQUIRE(act[l] >= MH_ACT_NONE && act[l] <= MH_ACT_SIGMOID, "mh_mlp_chain_fwd: bad activation %d", act[l]);
        a.W[l] = W[l];
        a.b[l] = b ? b[l] : nullptr;
        a.act[l] = act[l];
        a.y[l] = y[l];
        a.ldy[l] = ldy[l];
    }
    if (M == 0) return MH_OK;
    const bool vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && ldx % 4 == 0 && dims[0] % 4 == 0;
    const size_t lds = (size_t)(wfloats(sig->p) + CW * 16 * SA + sig->p[1] + sig->p[2] + sig->p[3]) * sizeof(float);
    // two workgroups per CU when the LDS allows it: the strips of a workgroup's four wavefronts are independent
    const int64_t tiles = mh_ceil_div(mh_ceil_div(M, 16), CW);
    const int64_t cap = (int64_t)mh_num_cus() * (lds <= 80 * 1024 ? 2 : 1);
    const int grid = (int)(tiles < cap ? tiles : cap);
    auto kern = vec ? sig->fwd_vec : sig->fwd_sca;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    MH_LAUNCH(kern, dim3(grid), dim3(CW * 64), lds, mh_stream(stream), a);
    MH_CHECK_LAUNCH("mh_mlp_chain_fwd");
    return MH_OK;
}

// phases: 1 = the strip kernel (dx, one partial dW / db slab per workgroup in the workspace), 2 = the slab reduction (dW, db), 3 = both
static int32_t chain_bwd_phases(int phases, const float* x, int64_t ldx, int64_t M, int32_t L, const int32_t* dims,
                                const float* const* W, const int32_t* act, const float* const* y, const int64_t* ldy,
                                const float* g, int64_t ldg, int32_t pre_masked, int32_t x_act, float* dx, int64_t lddx,
                                float* const* dW, float* const* db, void* workspace, int64_t workspace_bytes,
                                mh_stream_t stream);

int32_t mh_mlp_chain_bwd(const float* x, int64_t ldx, int64_t M, int32_t L, const int32_t* dims,
                         const float* const* W, const int32_t* act, const float* const* y, const int64_t* ldy,
                         const float* g, int64_t ldg, int32_t pre_masked, int32_t x_act, float* dx, int64_t lddx,
                         float* const* dW, float* const* db, void* workspace, int64_t workspace_bytes,
                         mh_stream_t stream) {
    return chain_bwd_phases(3, x, ldx, M, L, dims, W, act, y, ldy, g, ldg, pre_masked, x_act, dx, lddx, dW, db, workspace,
                            workspace_bytes, stream);
}

int32_t mh_mlp_chain_bwd_partial(const float* x, int64_t ldx, int64_t M, int32_t L, const int32_t* dims,
                                 const float* const* W, const int32_t* act, const float* const* y, const int64_t* ldy,
                                 const float* g, int64_t ldg, int32_t pre_masked, int32_t x_act, float* dx, int64_t lddx,
                                 void* workspace, int64_t workspace_bytes, mh_stream_t stream) {
    return chain_bwd_phases(1, x, ldx, M, L, dims, W, act, y, ldy, g, ldg, pre_masked, x_act, dx, lddx, nullptr, nullptr, workspace,
                            workspace_bytes, stream);
}

int32_t mh_mlp_chain_bwd_reduce(int64_t M, int32_t L, const int32_t* dims, float* const* dW, float* const* db, void* workspace,
                                int64_t workspace_bytes, mh_stream_t stream) {
    return chain_bwd_phases(2, nullptr, 0, M, L, dims, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, 0, dW, db,
                            workspace, workspace_bytes, stream);
}

static int32_t chain_bwd_phases(int phases, const float* x, int64_t ldx, int64_t M, int32_t L, const int32_t* dims,
                                const float* const* W, const int32_t* act, const float* const* y, const int64_t* ldy,
                                const float* g, int64_t ldg, int32_t pre_masked, int32_t x_act, float* dx, int64_t lddx,
                                float* const* dW, float* const* db, void* workspace, int64_t workspace_bytes,
                                mh_stream_t stream) {
    const bool do_main = phases & 1, do_red = phases & 2;
    MH_REQUIRE(dims && (!do_main || (x && W && act && y && ldy && g)) && (!do_red || dW), "mh_mlp_chain_bwd: null argument");
    const SigEntry* sig = find_sig(L, dims);
    if (!sig) {
        mh_set_error("mh_mlp_chain_bwd: no fused kernel covers this chain (L=%d): use mh_linear_bias_act_bwd per layer", L);
        return MH_ERR_UNSUPPORTED;
    }
    MH_REQUIRE(M >= 1 && (!do_main || (ldx >= dims[0] && ldg >= dims[L])), "mh_mlp_chain_bwd: bad shape M=%lld", (long long)M);
    MH_REQUIRE(x_act >= MH_ACT_NONE && x_act <= MH_ACT_SIGMOID, "mh_mlp_chain_bwd: bad x_act");
    MH_REQUIRE(!dx || lddx >= dims[0], "mh_mlp_chain_bwd: lddx < K");
    const int64_t need = mh_mlp_chain_bwd_workspace_bytes(M, L, dims);
    if (!workspace || workspace_bytes < need) {
        mh_set_error("mh_mlp_chain_bwd: workspace too small (%lld < %lld bytes)", (long long)workspace_bytes, (long long)need);
        return MH_ERR_WORKSPACE;
    }
    ChainArgs a{};
    a.x = x;
    a.ldx = ldx;
    a.M = M;
    for (int l = 0; l <= L; ++l) a.dims[l] = dims[l];
    ReduceArgs r{};
    for (int l = 0; l < L; ++l) {
        if (do_main) {
            MH_REQUIRE(W[l] && y[l] && ldy[l] >= dims[l + 1], "mh_mlp_chain_bwd: layer %d: null W / y or ldy < N", l);
            MH_REQUIRE(act[l] >= MH_ACT_NONE && act[l] <= MH_ACT_SIGMOID, "mh_mlp_chain_bwd: bad activation %d", act[l]);
            a.W[l] = W[l];
            a.act[l] = act[l];
            a.y[l] = const_cast<float*>(y[l]);
            a.ldy[l] = ldy[l];
        }
        if (do_red) {
            MH_REQUIRE(dW[l], "mh_mlp_chain_bwd: layer %d: null dW", l);
            r.dW[l] = dW[l];
            r.db[l] = db ? db[l] : nullptr;
        }
        r.nw[l] = dims[l] * dims[l + 1];
        r.nb[l] = dims[l + 1];
    }
    a.g = g;
    a.ldg = ldg;
    a.pre_masked = pre_masked ? 1 : 0;
    a.x_act = x_act;
    a.dx = dx;
    a.lddx = lddx;
    a.slabs = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    a.ptotal = chain_ptotal(L, dims, a.poff_w, a.poff_b);
    const int grid = chain_grid(M);
    const int* p = sig->p;
    const int wf = (dx ? p[0] * p[1] : 0) + p[1] * p[2] + p[2] * p[3];
    size_t lds = (size_t)(wf + CW * 16 * SA) * sizeof(float);
    const size_t scratch = (size_t)CW * 16 * 64 * 4 * sizeof(float);  // wg_reduce_store_dw: 16 tiles of every wavefront
    if (lds < scratch) lds = scratch;
    hipStream_t s = mh_stream(stream);
    auto kern = dx ? sig->bwd_dx : sig->bwd_nodx;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (do_main) {
        MH_LAUNCH(kern, dim3(grid), dim3(CW * 64), lds, s, a);
        MH_CHECK_LAUNCH("mh_mlp_chain_bwd");
    }
    if (!do_red) return MH_OK;
    r.slabs = a.slabs;
    r.G = grid;
    r.ptotal = a.ptotal;
    r.L = L;
    for (int l = 0; l < L; ++l) {
        r.poff_w[l] = a.poff_w[l];
        r.poff_b[l] = a.poff_b[l];
    }
    MH_LAUNCH(chain_reduce_kernel, dim3((unsigned)mh_ceil_div(a.ptotal, 64)), dim3(256), 0, s, r);
    MH_CHECK_LAUNCH("mh_mlp_chain_bwd(reduce)");
    return MH_OK;
}

}  // extern "C"
