// Dense layer y = act(x W + b) on the fp32 MFMA pipe (gfx950).
// Reference: _Dense.call -> keras Dense (merlin/models/tf/blocks/mlp.py:275-280), MLPBlock (:35-139).
//
// Shapes on this path are tall-and-skinny (M = batch = 64 K rows, N <= 512, K <= 3341): the
// workgroup tile is 128 rows x {128,64,32} columns, W streams through LDS k-tile by k-tile
// (it is L2-resident: <= 212 KB on the DLRM config), x is read exactly once, y written once.
// Roofline: bytes = 4(MK + KN + MN), flops = 2MKN; layers with K,N >= 128 sit above the fp32
// ridge (157 TF / 8 TB/s ~ 20 flop/B), K = 13 / N <= 64 layers are HBM-bound.
#include <stdlib.h>

#include "mh_gemm_core.h"
#include "mh_gemm2.h"

using namespace mhgemm;

namespace {

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == MH_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MH_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void linear_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ W,
                                                        const float* __restrict__ bias, int64_t M,
                                                        int K, int N, int act, float* __restrict__ y,
                                                        int64_t ldy, int vec_x, int vec_w,
                                                        const float* __restrict__ x0,
                                                        const float* __restrict__ xres) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDK + 2 * BK * BN];
    float* As0 = smem;
    float* As1 = smem + BM * LDK;
    float* Bs0 = smem + 2 * BM * LDK;
    float* Bs1 = Bs0 + BK * BN;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    constexpr int NTH = WM * WN * 64;
    KMajorTile<BM, NTH> ta;
    NMajorTile<BN, NTH> tb;
    f32x16 acc[TM][TN];
    zero_acc<TM, TN>(acc);

    const int nk = (K + BK - 1) / BK;
    ta.init(x, ldx, row0, M);
    tb.init(W, N, n0, N);
    ta.load(0, K, vec_x);
    tb.load(W, N, 0, K, N, vec_w);
    ta.store(As0);
    tb.store(Bs0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        float* Ac = (kt & 1) ? As1 : As0;
        float* Bc = (kt & 1) ? Bs1 : Bs0;
        float* An = (kt & 1) ? As0 : As1;
        float* Bn = (kt & 1) ? Bs0 : Bs1;
        if (more) {
            ta.load((kt + 1) * BK, K, vec_x);
            tb.load(W, N, (kt + 1) * BK, K, N, vec_w);
        }
        mma_ktile<TM, TN, false>(Ac, wm * TM * 32, Bc, wn * TN * 32, BN, acc);
        if (more) {
            ta.store(An);
            tb.store(Bn);
        }
        __syncthreads();
    }
    EpiArgs ep{};
    ep.bias = bias;
    ep.act = act;
    ep.x0 = x0;
    ep.xres = xres;
    ep.ld_x0 = N;
    store_tile<TM, TN>(acc, y, ldy, row0 + wm * TM * 32, n0 + wn * TN * 32, M, N, lane, ep);
}

// N <= 4 heads (e.g. BinaryOutput's Dense(1, sigmoid), outputs/classification.py:114):
// HBM-bound GEMV, 16 lanes per row, k-ascending per-lane partial chains + shuffle tree.
template <int NMAX>
__global__ __launch_bounds__(256) void linear_small_n_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ W,
                                                            const float* __restrict__ bias, int64_t M,
                                                            int K, int N, int act, float* __restrict__ y,
                                                            int64_t ldy) {
    const int t = threadIdx.x;
    const int sub = t & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (t >> 4);
    float acc[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) acc[n] = 0.f;
    if (row < M) {
        const float* xr = x + row * ldx;
        for (int k = sub; k < K; k += 16) {
            const float xv = xr[k];
#pragma unroll
            for (int n = 0; n < NMAX; ++n)
                if (n < N) acc[n] = fmaf(xv, W[(int64_t)k * N + n], acc[n]);
        }
    }
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) acc[n] += __shfl_xor(acc[n], off);
    }
    if (row < M && sub == 0) {
#pragma unroll
        for (int n = 0; n < NMAX; ++n)
            if (n < N) y[row * ldy + n] = apply_act(acc[n] + (bias ? bias[n] : 0.f), act);
    }
}

}  // namespace

bool mh_internal_linear_v2(const float* x, int64_t ldx, const float* W, const float* b, int64_t M, int K, int N, int act,
                           float* y, int64_t ldy, const float* x0, const float* xres, float* p_out, hipStream_t s);

extern "C" {

int32_t mh_linear_bias_act_fwd(const float* x, int64_t ldx, const float* W, const float* b,
                               int64_t M, int32_t K, int32_t N, int32_t act, float* y,
                               int64_t ldy, mh_stream_t stream) {
    MH_REQUIRE(x && W && y, "mh_linear_bias_act_fwd: null argument");
    MH_REQUIRE(M >= 0 && K >= 1 && N >= 1, "mh_linear_bias_act_fwd: bad shape M=%lld K=%d N=%d", (long long)M, K, N);
    MH_REQUIRE(ldx >= K && ldy >= N, "mh_linear_bias_act_fwd: leading dimension smaller than row");
    MH_REQUIRE(act >= MH_ACT_NONE && act <= MH_ACT_SIGMOID, "mh_linear_bias_act_fwd: bad activation %d", act);
    if (M == 0) return MH_OK;
    hipStream_t s = mh_stream(stream);
    if (N <= 4) {
        dim3 grid((unsigned)mh_ceil_div(M, 16));
        MH_LAUNCH((linear_small_n_kernel<4>), grid, dim3(256), 0, s, x, ldx, W, b, M, K, N, act, y, ldy);
        MH_CHECK_LAUNCH("mh_linear_bias_act_fwd");
        return MH_OK;
    }
    return mh_internal_linear(x, ldx, W, b, M, K, N, act, y, ldy, nullptr, nullptr, s);
}

int32_t mh_cross_layer_fwd(const float* x0, const float* x, const float* W, const float* b, int64_t M,
                           int32_t d, float* out, mh_stream_t stream) {
    MH_REQUIRE(x0 && x && W && out, "mh_cross_layer_fwd: null argument");
    MH_REQUIRE(M >= 0 && d >= 5, "mh_cross_layer_fwd: bad shape M=%lld d=%d (d must be > 4)", (long long)M, d);
    if (M == 0) return MH_OK;
    return mh_internal_linear(x, d, W, b, M, d, d, MH_ACT_NONE, out, d, x0, x, mh_stream(stream));
}

int32_t mh_cross_layer_fwd_save(const float* x0, const float* x, const float* W, const float* b, int64_t M, int32_t d,
                                float* out, float* p_out, mh_stream_t stream) {
    MH_REQUIRE(x0 && x && W && out && p_out, "mh_cross_layer_fwd_save: null argument");
    MH_REQUIRE(M >= 0 && d >= 5, "mh_cross_layer_fwd_save: bad shape M=%lld d=%d (d must be > 4)", (long long)M, d);
    if (M == 0) return MH_OK;
    hipStream_t s = mh_stream(stream);
    if (mh_internal_linear_v2(x, d, W, b, M, d, d, MH_ACT_NONE, out, d, x0, x, p_out, s)) return MH_OK;
    // first-generation core: p = x W + b as a plain product, then the cross as one elementwise pass
    const int32_t st = mh_internal_linear(x, d, W, b, M, d, d, MH_ACT_NONE, p_out, d, nullptr, nullptr, s);
    if (st != MH_OK) return st;
    return mh_eltwise(2, x0, p_out, x, out, M * (int64_t)d, stream);
}

int32_t mh_cross_layer_lowrank_fwd(const float* x0, const float* x, const float* h, const float* V, const float* b,
                                   int64_t M, int32_t d, int32_t r, float* out, mh_stream_t stream) {
    MH_REQUIRE(x0 && x && h && V && out, "mh_cross_layer_lowrank_fwd: null argument");
    MH_REQUIRE(M >= 0 && d >= 5 && r >= 1, "mh_cross_layer_lowrank_fwd: bad shape M=%lld d=%d r=%d", (long long)M, d, r);
    if (M == 0) return MH_OK;
    return mh_internal_linear(h, r, V, b, M, r, d, MH_ACT_NONE, out, d, x0, x, mh_stream(stream));
}

}  // extern "C"

// Second-generation core (mh_gemm2.h: DMA tiles, 3-deep ring, 64x64 per wavefront, column tiles fastest) for the wide
// layers: DCN cross 3344 x 3344 112 -> 131 TF, 3344 -> 512 105 -> 116 TF, 512 -> 256 91 -> 95 TF (tools/exp/gemm_lab).
// One column tile (N <= 128) gains nothing: those layers are bound per workgroup, not by the loop (profiles/r2_notes.md);
// and the 256 x 128 tiles must still give every CU two workgroups (M = 32 K x N = 256 does not: first generation).
// Returns false when the shape / alignment is not eligible (the caller then takes the first-generation kernels).
bool mh_internal_linear_v2(const float* x, int64_t ldx, const float* W, const float* b, int64_t M, int K, int N, int act,
                           float* y, int64_t ldy, const float* x0, const float* xres, float* p_out, hipStream_t s) {
    static const bool no_v2 = getenv("MERLIN_HIP_GEMM_V1") != nullptr;
    const bool vec_x = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (ldx % 4 == 0);
    const bool vec_w = ((reinterpret_cast<uintptr_t>(W) & 15) == 0) && (N % 4 == 0);
    const bool fills = mh_ceil_div(M, 256) * mh_ceil_div(N, 128) >= 2 * (int64_t)mh_num_cus();
    if (no_v2 || !fills || !vec_x || !vec_w || N < 256 || K < 64 || K % 4 != 0) return false;
    mhgemm2::Epilogue ep{};
    ep.bias = b;
    ep.act = act;
    ep.x0 = x0;
    ep.xres = xres;
    ep.ld_x0 = N;
    ep.p_out = p_out;
    ep.ldp = N;
    return mhgemm2::launch<256, 128, 4, 2, false, 3>(x, ldx, W, N, M, N, K, y, ldy, ep, s) == hipSuccess;
}

int32_t mh_internal_linear(const float* x, int64_t ldx, const float* W, const float* b, int64_t M, int K, int N,
                           int act, float* y, int64_t ldy, const float* x0, const float* xres, hipStream_t s) {
    const int vec_x = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (ldx % 4 == 0);
    const int vec_w = ((reinterpret_cast<uintptr_t>(W) & 15) == 0) && (N % 4 == 0);
    if (mh_internal_linear_v2(x, ldx, W, b, M, K, N, act, y, ldy, x0, xres, nullptr, s)) return MH_OK;
    if (N > 64) {
        dim3 grid((unsigned)mh_ceil_div(M, 128), (unsigned)mh_ceil_div(N, 128));
        MH_LAUNCH((linear_fwd_kernel<128, 128, 4, 2>), grid, dim3(512), 0, s, x, ldx, W, b, M, K, N, act, y, ldy, vec_x, vec_w, x0, xres);
    } else if (N > 32) {
        dim3 grid((unsigned)mh_ceil_div(M, 128), 1);
        MH_LAUNCH((linear_fwd_kernel<128, 64, 4, 1>), grid, dim3(256), 0, s, x, ldx, W, b, M, K, N, act, y, ldy, vec_x, vec_w, x0, xres);
    } else {
        dim3 grid((unsigned)mh_ceil_div(M, 128), 1);
        MH_LAUNCH((linear_fwd_kernel<128, 32, 4, 1>), grid, dim3(256), 0, s, x, ldx, W, b, M, K, N, act, y, ldy, vec_x, vec_w, x0, xres);
    }
    MH_CHECK_LAUNCH("mh_linear_bias_act_fwd");
    return MH_OK;
}
