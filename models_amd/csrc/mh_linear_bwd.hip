// Backward of the Dense layer (gfx950, fp32 MFMA).
// Reference: the GradientTape backward of keras Dense inside BaseModel.train_step
// (merlin/models/tf/models/base.py:1121-1174); forward at blocks/mlp.py:275-280.
//
//   dz = dy * act'(y)                      (in place; per-block column sums -> db partials)
//   dx[M,K] = dz[M,N] W[K,N]^T             "NT" GEMM, contraction over N
//   dW[K,N] = x[M,K]^T dz[M,N]             "TN" GEMM, contraction over the batch: split over M
//                                          into S slices, partial [S,K,N] slabs, fixed-order reduce
// Every reduction has a fixed order (no float atomics): results are run-to-run deterministic.
#include <stdlib.h>

#include "mh_gemm_core.h"
#include "mh_gemm2.h"

using namespace mhgemm;

namespace {

constexpr int ACT_ROWS = 256;  // rows per act-grad block

// dz in place + column partial sums.  block = 256 threads = RG row groups x CW columns
// (CW = 32/64/128/256 >= min(N,256)); row-group partials are combined through LDS in fixed order.
__global__ __launch_bounds__(256) void act_grad_colsum_kernel(const float* __restrict__ y, int64_t ldy,
                                                             float* __restrict__ dy, int64_t lddy,
                                                             int64_t M, int N, int act, int CW,
                                                             float* __restrict__ db_partial) {
    __shared__ float red[256];
    const int RG = 256 / CW;
    const int rg = threadIdx.x / CW, c = threadIdx.x - rg * CW;
    const int64_t r0 = (int64_t)blockIdx.x * ACT_ROWS;
    const int64_t r1 = (r0 + ACT_ROWS < M) ? r0 + ACT_ROWS : M;
    for (int n0 = 0; n0 < N; n0 += CW) {
        const int n = n0 + c;
        float s = 0.f;
        if (n < N) {
            for (int64_t r = r0 + rg; r < r1; r += RG) {
                float g = dy[r * lddy + n];
                if (act == MH_ACT_RELU) {
                    g = (y[r * ldy + n] > 0.f) ? g : 0.f;
                    dy[r * lddy + n] = g;
                } else if (act == MH_ACT_SIGMOID) {
                    const float yy = y[r * ldy + n];
                    g = g * yy * (1.f - yy);
                    dy[r * lddy + n] = g;
                }
                s += g;
            }
        }
        if (db_partial) {
            red[threadIdx.x] = s;
            __syncthreads();
            if (rg == 0 && n < N) {
                float t = 0.f;
                for (int k = 0; k < RG; ++k) t += red[k * CW + c];
                db_partial[(int64_t)blockIdx.x * N + n] = t;
            }
            __syncthreads();
        }
    }
}

// out[i] = sum_s part[s*len + i].  block = 64 outputs x 16 slice groups (1024 threads); group g sums slices g, g+16, ... with 4
// independent loads in flight, then the 16 group sums are combined in fixed order through LDS -> deterministic, and S
// sequential HBM round trips become S/64 (with 4 groups a thread walked S/4 slices four at a time: 16 dependent rounds of HBM
// latency for the 256 splits of the 415 x 128 layer, 37 us for 54 MB).
// Few slabs (S <= 8) of a LONG vector -- dW of a wide layer, e.g. the 3344 x 3344 cross kernel split in two over the batch: the
// general kernel below spends a 1024-thread workgroup on 64 outputs (14 of its 16 groups idle at S = 2: 2.85 ms for 134 MB, round-3
// profile).  Here: one float4 of the output per thread, the slabs summed in ascending order (deterministic), streaming loads.
__global__ __launch_bounds__(256) void reduce_slabs_vec_kernel(const float* __restrict__ part, int S, int64_t len4,
                                                              float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= len4) return;
    f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k < S) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(part) + (int64_t)k * len4 + i);
    f32x4 t = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (k < S) t += v[k];
    reinterpret_cast<f32x4*>(out)[i] = t;
}

__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ part, int S,
                                                              int64_t len, float* __restrict__ out,
                                                              const float* __restrict__ part2, int64_t len2,
                                                              float* __restrict__ out2, int blocks1) {
    constexpr int NG = 16;
    __shared__ float red[NG * 64];
    // second (optional) reduction rides in the same launch: blocks >= blocks1 sum part2 (db next to dW)
    int bx = blockIdx.x;
    if (bx >= blocks1) {
        bx -= blocks1;
        part = part2;
        len = len2;
        out = out2;
    }
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t i = (int64_t)bx * 64 + o;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < len) {
        int k = g;
        for (; k + 3 * NG < S; k += 4 * NG) {
            s0 += part[(int64_t)k * len + i];
            s1 += part[(int64_t)(k + NG) * len + i];
            s2 += part[(int64_t)(k + 2 * NG) * len + i];
            s3 += part[(int64_t)(k + 3 * NG) * len + i];
        }
        for (; k < S; k += NG) s0 += part[(int64_t)k * len + i];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && i < len) {  // fixed order: deterministic
        float t[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) t[j] = red[64 * j + o];
#pragma unroll
        for (int w = NG / 2; w >= 1; w >>= 1)
#pragma unroll
            for (int j = 0; j < w; ++j) t[j] = t[j] + t[j + w];
        out[i] = t[0];
    }
}

// dz = dy * act'(y) in place, 16 bytes per lane (N % 4 == 0, 16-byte aligned rows): the streaming form of
// act_grad_colsum_kernel for the common case where the column sums ride on the dW GEMM.
__global__ __launch_bounds__(256) void act_grad_vec_kernel(const float* __restrict__ y, int64_t ldy,
                                                          float* __restrict__ dy, int64_t lddy, int64_t M, int N4,
                                                          int act) {
    const int64_t total = M * N4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / N4;
        const int c = (int)(i - r * N4) * 4;
        const f32x4 yy = *reinterpret_cast<const f32x4*>(y + r * ldy + c);
        f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * lddy + c);
        if (act == MH_ACT_RELU) {
            g.x = yy.x > 0.f ? g.x : 0.f;
            g.y = yy.y > 0.f ? g.y : 0.f;
            g.z = yy.z > 0.f ? g.z : 0.f;
            g.w = yy.w > 0.f ? g.w : 0.f;
        } else {
            g = g * yy * (f32x4{1.f, 1.f, 1.f, 1.f} - yy);
        }
        *reinterpret_cast<f32x4*>(dy + r * lddy + c) = g;
    }
}

// C[M, Nout] = A[M, Kc] * B[Nout, Kc]^T   (both operands contraction-contiguous)
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_kernel(const float* __restrict__ A, int64_t lda,
                                                     const float* __restrict__ Bm, int64_t ldb, int64_t M,
                                                     int Nout, int Kc, float* __restrict__ Cm, int64_t ldc,
                                                     int vec_a, int vec_b, const float* __restrict__ maskx,
                                                     int64_t ldm, int x_act, const float* __restrict__ addend,
                                                     int64_t ld_add) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDK + 2 * BN * LDK];
    float* As0 = smem;
    float* As1 = smem + BM * LDK;
    float* Bs0 = smem + 2 * BM * LDK;
    float* Bs1 = Bs0 + BN * LDK;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    constexpr int NTH = WM * WN * 64;
    KMajorTile<BM, NTH> ta;
    KMajorTile<BN, NTH> tb;
    f32x16 acc[TM][TN];
    zero_acc<TM, TN>(acc);
    const int nk = (Kc + BK - 1) / BK;
    ta.init(A, lda, row0, M);
    tb.init(Bm, ldb, n0, Nout);
    ta.load(0, Kc, vec_a);
    tb.load(0, Kc, vec_b);
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
            ta.load((kt + 1) * BK, Kc, vec_a);
            tb.load((kt + 1) * BK, Kc, vec_b);
        }
        mma_ktile<TM, TN, true>(Ac, wm * TM * 32, Bc, wn * TN * 32, 0, acc);
        if (more) {
            ta.store(An);
            tb.store(Bn);
        }
        __syncthreads();
    }
    // the producer's activation derivative is folded into dX: x = act(z_prev) is at hand (store_tile, mh_gemm_core.h)
    EpiArgs ep{};
    ep.maskx = maskx;
    ep.ldm = ldm;
    ep.x_act = maskx ? x_act : MH_ACT_NONE;
    ep.addend = addend;
    ep.ld_add = ld_add;
    store_tile<TM, TN>(acc, Cm, ldc, row0 + wm * TM * 32, n0 + wn * TN * 32, M, Nout, lane, ep);
}

// (An A-rows-stationary form of this product for the DLRM's first top-MLP layer -- gemm_nt_astat_kernel, rounds 3-5 -- was measured
// faster alone and slower in every multi-stream step, and is gone: profiles/r3_notes.md, r3_astat_timeline_v*.txt.)

// Split-M "TN" GEMM: part[s][K, N] = x[m in slice s][K]^T dz[m in slice s][N].
// Both operands are staged n-major ([BK contraction rows][cols]); fragments are ds_read_b32.
template <int BMO, int BNO>
__global__ __launch_bounds__(256) void gemm_tn_splitm_kernel(const float* __restrict__ X, int64_t ldx,
                                                            const float* __restrict__ Z, int64_t ldz,
                                                            int64_t M, int K, int N, int64_t rows_per_split,
                                                            float* __restrict__ part, int vec_x, int vec_z,
                                                            float* __restrict__ db_part) {
    constexpr int TM = BMO / 2 / 32, TN = BNO / 2 / 32;  // waves 2 x 2
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * BMO + 2 * BK * BNO];
    float* As0 = smem;
    float* As1 = smem + BK * BMO;
    float* Bs0 = smem + 2 * BK * BMO;
    float* Bs1 = Bs0 + BK * BNO;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int k0 = blockIdx.x * BMO;  // output row tile (over K)
    const int n0 = blockIdx.y * BNO;  // output col tile (over N)
    const int s = blockIdx.z;
    const int64_t m_beg = (int64_t)s * rows_per_split;
    const int64_t m_end = (m_beg + rows_per_split < M) ? m_beg + rows_per_split : M;
    const int nk = (int)((m_end - m_beg + BK - 1) / BK);

    NMajorTile<BMO> ta;
    NMajorTile<BNO> tb;
    f32x16 acc[TM][TN];
    zero_acc<TM, TN>(acc);
    float dbacc = 0.f;
    // NMajorTile::load(W, ldw, k0(contraction row start), K(contraction end), n0, N, vec)
    ta.init(X + m_beg * ldx, ldx, k0, K);
    tb.init(Z + m_beg * ldz, ldz, n0, N);
    const int vx = vec_x, vz = vec_z;  // ld-padded operands take the 16-byte path (NMajorTile::init)
    auto lda_ = [&](int kt) { ta.load(X + m_beg * ldx, ldx, kt * BK, (int)(m_end - m_beg), K, vx); };
    auto ldb_ = [&](int kt) { tb.load(Z + m_beg * ldz, ldz, kt * BK, (int)(m_end - m_beg), N, vz); };
    if (nk > 0) {
        lda_(0);
        ldb_(0);
        ta.store(As0);
        tb.store(Bs0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        const float* Ac = (kt & 1) ? As1 : As0;
        const float* Bc = (kt & 1) ? Bs1 : Bs0;
        float* An = (kt & 1) ? As0 : As1;
        float* Bn = (kt & 1) ? Bs0 : Bs1;
        if (more) {
            lda_(kt + 1);
            ldb_(kt + 1);
        }
        if (db_part && blockIdx.x == 0 && threadIdx.x < BNO) {
            // column sums of this slice of dz (db = colsum(dz)) ride along on the first row-tile's blocks
            float cs = 0.f;
#pragma unroll 8
            for (int r = 0; r < BK; ++r) cs += Bc[r * BNO + threadIdx.x];
            dbacc += cs;
        }
#pragma unroll
        for (int st = 0; st < BK / 2; ++st) {
            float a[TM], b[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[tm] = Ac[(2 * st + h) * BMO + wm * TM * 32 + tm * 32 + l31];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[tn] = Bc[(2 * st + h) * BNO + wn * TN * 32 + tn * 32 + l31];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(a[tm], b[tn], acc[tm][tn]);
        }
        if (more) {
            ta.store(An);
            tb.store(Bn);
        }
        __syncthreads();
    }
    if (db_part && blockIdx.x == 0 && threadIdx.x < BNO && n0 + (int)threadIdx.x < N)
        db_part[(int64_t)s * N + n0 + threadIdx.x] = dbacc;
    float* P = part + (int64_t)s * K * N;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + wn * TN * 32 + tn * 32 + acc_col(lane);
        if (col >= N) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = k0 + wm * TM * 32 + tm * 32 + acc_row(r, lane);
                if (row < K) P[(int64_t)row * N + col] = acc[tm][tn][r];
            }
    }
}

struct BwdPlan {
    int act_blocks;
    int splits;
    int64_t rows_per_split;
    int64_t dw_floats, db_floats;
};

bool big_tiles(int K, int N) { return K > 64 && N > 64; }

BwdPlan make_plan(int64_t M, int K, int N) {
    BwdPlan p;
    p.act_blocks = (int)mh_ceil_div(M, ACT_ROWS);
    const int t = big_tiles(K, N) ? 128 : 64;
    const int64_t tiles = mh_ceil_div(K, t) * mh_ceil_div(N, t);
    int64_t want = mh_ceil_div(1024, tiles);             // ~4 blocks per CU
    const int64_t max_by_rows = mh_ceil_div(M, 256);       // >= 256 rows (8 k-tiles) per split
    if (want > max_by_rows) want = max_by_rows;
    if (want < 1) want = 1;
    int64_t rps = mh_ceil_div(M, want);
    rps = mh_ceil_div(rps, BK) * BK;
    p.rows_per_split = rps;
    p.splits = (int)mh_ceil_div(M, rps);
    p.dw_floats = (int64_t)p.splits * K * N;
    p.db_floats = (int64_t)p.splits * N;
    return p;
}

}  // namespace

int32_t mh_internal_gemm_nt_mask(const float* A, int64_t lda, const float* Bm, int64_t ldb, int64_t M, int Nout,
                                 int Kc, float* Cm, int64_t ldc, const float* maskx, int64_t ldm, int x_act,
                                 hipStream_t s);
int32_t mh_internal_gemm_nt_ep(const float* A, int64_t lda, const float* Bm, int64_t ldb, int64_t M, int Nout,
                               int Kc, float* Cm, int64_t ldc, const float* maskx, int64_t ldm, int x_act,
                               const float* addend, int64_t ld_add, hipStream_t s);

int32_t mh_internal_gemm_nt(const float* A, int64_t lda, const float* Bm, int64_t ldb, int64_t M, int Nout, int Kc,
                            float* Cm, int64_t ldc, hipStream_t s) {
    return mh_internal_gemm_nt_mask(A, lda, Bm, ldb, M, Nout, Kc, Cm, ldc, nullptr, 0, MH_ACT_NONE, s);
}

int32_t mh_internal_gemm_nt_mask(const float* A, int64_t lda, const float* Bm, int64_t ldb, int64_t M, int Nout,
                                 int Kc, float* Cm, int64_t ldc, const float* maskx, int64_t ldm, int x_act,
                                 hipStream_t s) {
    return mh_internal_gemm_nt_ep(A, lda, Bm, ldb, M, Nout, Kc, Cm, ldc, maskx, ldm, x_act, nullptr, 0, s);
}

// C = A B^T, optionally with the producer's activation derivative folded in (maskx, x_act) and / or `+ addend` at the end
int32_t mh_internal_gemm_nt_ep(const float* A, int64_t lda, const float* Bm, int64_t ldb, int64_t M, int Nout,
                               int Kc, float* Cm, int64_t ldc, const float* maskx, int64_t ldm, int x_act,
                               const float* addend, int64_t ld_add, hipStream_t s) {
    const int vec_a = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (lda % 4 == 0);
    const int vec_b = ((reinterpret_cast<uintptr_t>(Bm) & 15) == 0) && (ldb % 4 == 0);
    static const bool no_v2 = getenv("MERLIN_HIP_GEMM_V1") != nullptr;
    const bool fills = mh_ceil_div(M, 256) * mh_ceil_div(Nout, 128) >= 2 * (int64_t)mh_num_cus();
    if (!no_v2 && fills && vec_a && vec_b && Nout >= 256 && Kc >= 256 && Kc % 4 == 0) {  // wide layers: second-generation core
        mhgemm2::Epilogue ep{};
        ep.maskx = maskx;
        ep.ldm = ldm;
        ep.x_act = maskx ? x_act : MH_ACT_NONE;
        ep.addend = addend;
        ep.ld_add = ld_add;
        const hipError_t e = mhgemm2::launch<256, 128, 4, 2, true, 3>(A, lda, Bm, ldb, M, Nout, Kc, Cm, ldc, ep, s);
        if (e != hipSuccess) {
            mh_set_error("gemm_nt: launch failed: %s", hipGetErrorString(e));
            return MH_ERR_LAUNCH;
        }
        return MH_OK;
    }
    if (Nout > 64) {
        dim3 grid((unsigned)mh_ceil_div(M, 128), (unsigned)mh_ceil_div(Nout, 128));
        MH_LAUNCH((gemm_nt_kernel<128, 128, 4, 2>), grid, dim3(512), 0, s, A, lda, Bm, ldb, M, Nout, Kc, Cm, ldc, vec_a, vec_b, maskx, ldm, x_act, addend, ld_add);
    } else if (Nout > 32) {
        dim3 grid((unsigned)mh_ceil_div(M, 128), 1);
        MH_LAUNCH((gemm_nt_kernel<128, 64, 4, 1>), grid, dim3(256), 0, s, A, lda, Bm, ldb, M, Nout, Kc, Cm, ldc, vec_a, vec_b, maskx, ldm, x_act, addend, ld_add);
    } else {
        dim3 grid((unsigned)mh_ceil_div(M, 128), 1);
        MH_LAUNCH((gemm_nt_kernel<128, 32, 4, 1>), grid, dim3(256), 0, s, A, lda, Bm, ldb, M, Nout, Kc, Cm, ldc, vec_a, vec_b, maskx, ldm, x_act, addend, ld_add);
    }
    MH_CHECK_LAUNCH("gemm_nt");
    return MH_OK;
}

int32_t mh_internal_gemm_tn(const float* X, int64_t ldx, const float* Z, int64_t ldz, int64_t M, int K, int N,
                            float* out, hipStream_t s) {
    const int vec_x = ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && (ldx % 4 == 0);
    const int vec_z = ((reinterpret_cast<uintptr_t>(Z) & 15) == 0) && (ldz % 4 == 0);
    const int64_t rps = mh_ceil_div(M, BK) * BK;
    dim3 grid((unsigned)mh_ceil_div(K, 64), (unsigned)mh_ceil_div(N, 64), 1);
    MH_LAUNCH((gemm_tn_splitm_kernel<64, 64>), grid, dim3(256), 0, s, X, ldx, Z, ldz, M, K, N, rps, out, vec_x, vec_z, (float*)nullptr);
    MH_CHECK_LAUNCH("gemm_tn");
    return MH_OK;
}

extern "C" {

int64_t mh_linear_bwd_workspace_bytes(int64_t M, int32_t K, int32_t N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    const BwdPlan p = make_plan(M, K, N);
    return (p.dw_floats + p.db_floats) * (int64_t)sizeof(float);
}

int32_t mh_linear_bias_act_bwd(const float* x, int64_t ldx, const float* W, const float* y, int64_t ldy,
                               float* dy, int64_t lddy, int64_t M, int32_t K, int32_t N, int32_t act,
                               int32_t x_act, float* dx, int64_t lddx, float* dW, float* db, void* workspace,
                               int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(x && dy, "mh_linear_bias_act_bwd: null argument");
    MH_REQUIRE(!dx || W, "mh_linear_bias_act_bwd: W is required for dx");
    MH_REQUIRE(M >= 1 && K >= 1 && N >= 1, "mh_linear_bias_act_bwd: bad shape");
    MH_REQUIRE(act == MH_ACT_NONE || y, "mh_linear_bias_act_bwd: y is required for the activation derivative");
    MH_REQUIRE(x_act >= MH_ACT_NONE && x_act <= MH_ACT_SIGMOID, "mh_linear_bias_act_bwd: bad x_act");
    MH_REQUIRE(ldx >= K && lddy >= N && (!dx || lddx >= K), "mh_linear_bias_act_bwd: bad leading dimension");
    const BwdPlan p = make_plan(M, K, N);
    MH_REQUIRE(!dW || (workspace && workspace_bytes >= (p.dw_floats + p.db_floats) * (int64_t)sizeof(float)),
               "mh_linear_bias_act_bwd: workspace too small (%lld bytes given)", (long long)workspace_bytes);
    MH_REQUIRE(dW || !db, "mh_linear_bias_act_bwd: db rides on the dW pass (pass dW too)");
    hipStream_t s = mh_stream(stream);
    float* ws_dw = static_cast<float*>(workspace);
    float* ws_db = ws_dw + p.dw_floats;

    if (act != MH_ACT_NONE) {
        const bool vec = (N % 4 == 0) && (ldy % 4 == 0) && (lddy % 4 == 0) &&
                         ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
        if (vec) {
            int64_t nb = mh_ceil_div(M * (N / 4), 256);
            const int64_t cap = (int64_t)mh_num_cus() * 16;
            if (nb > cap) nb = cap;
            MH_LAUNCH(act_grad_vec_kernel, dim3((unsigned)nb), dim3(256), 0, s, y, ldy, dy, lddy, M, N / 4, act);
        } else {
            const int CW = N <= 32 ? 32 : (N <= 64 ? 64 : (N <= 128 ? 128 : 256));
            MH_LAUNCH(act_grad_colsum_kernel, dim3(p.act_blocks), dim3(256), 0, s, y, ldy, dy, lddy, M, N,
                               act, CW, (float*)nullptr);
        }
    }
    if (dx) {
        const int32_t st = mh_internal_gemm_nt_mask(dy, lddy, W, (int64_t)N, M, K, N, dx, lddx, x, ldx, x_act, s);
        if (st != MH_OK) return st;
    }
    if (dW) {
        const int vec_x = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (ldx % 4 == 0);
        const int vec_dy = ((reinterpret_cast<uintptr_t>(dy) & 15) == 0) && (lddy % 4 == 0);
        static const bool no_v2 = getenv("MERLIN_HIP_GEMM_V1") != nullptr;
        // second-generation core (DMA tiles, 3-deep ring): 415 x 128 211 -> 199 us with dX, 3344 x 3344 27.0 -> 23.7 ms; the
        // 256 x 128 layer of the two-tower config is faster on the first generation (measured, profiles/r2_notes.md)
        // one slab: the GEMM writes dW / db themselves, nothing to reduce
        float* out_dw = (p.splits == 1) ? dW : ws_dw;
        float* out_db = (p.splits == 1) ? db : ws_db;
        if (!no_v2 && vec_x && vec_dy && K >= 256 && N >= 128 && (int64_t)K * N >= 49152) {
            const hipError_t e = mhgemm2::launch_tn<256, 128, 4, 2, 3>(x, ldx, dy, lddy, M, K, N, p.rows_per_split, p.splits,
                                                                       out_dw, db ? out_db : nullptr, s);
            if (e != hipSuccess) {
                mh_set_error("mh_linear_bias_act_bwd: launch failed: %s", hipGetErrorString(e));
                return MH_ERR_LAUNCH;
            }
        } else if (big_tiles(K, N)) {
            dim3 grid((unsigned)mh_ceil_div(K, 128), (unsigned)mh_ceil_div(N, 128), (unsigned)p.splits);
            MH_LAUNCH((gemm_tn_splitm_kernel<128, 128>), grid, dim3(256), 0, s, x, ldx, dy, lddy, M, K, N,
                               p.rows_per_split, out_dw, vec_x, vec_dy, db ? out_db : nullptr);
        } else {
            dim3 grid((unsigned)mh_ceil_div(K, 64), (unsigned)mh_ceil_div(N, 64), (unsigned)p.splits);
            MH_LAUNCH((gemm_tn_splitm_kernel<64, 64>), grid, dim3(256), 0, s, x, ldx, dy, lddy, M, K, N,
                               p.rows_per_split, out_dw, vec_x, vec_dy, db ? out_db : nullptr);
        }
        const int64_t len = (int64_t)K * N;
        if (p.splits == 1) {
            // nothing to reduce
        } else if (p.splits <= 8 && len % 4 == 0 && len >= (1 << 16) && (reinterpret_cast<uintptr_t>(dW) & 15) == 0) {
            // few slabs of a long vector (wide layers): float4 per thread; db (N floats x S slabs) keeps the small general kernel
            MH_LAUNCH(reduce_slabs_vec_kernel, dim3((unsigned)mh_ceil_div(len / 4, 256)), dim3(256), 0, s, ws_dw, p.splits,
                               len / 4, dW);
            if (db)
                MH_LAUNCH(reduce_partials_kernel, dim3((unsigned)mh_ceil_div(N, 64)), dim3(1024), 0, s, ws_db, p.splits,
                                   (int64_t)N, db, (const float*)nullptr, (int64_t)0, (float*)nullptr, (int)mh_ceil_div(N, 64));
        } else {
            const int b1 = (int)mh_ceil_div(len, 64), b2 = db ? (int)mh_ceil_div(N, 64) : 0;
            MH_LAUNCH(reduce_partials_kernel, dim3((unsigned)(b1 + b2)), dim3(1024), 0, s, ws_dw, p.splits, len, dW,
                               ws_db, (int64_t)N, db, b1);  // dW and db slabs in ONE launch
        }
    }
    MH_CHECK_LAUNCH("mh_linear_bias_act_bwd");
    return MH_OK;
}

// ---- backward of a DCN-v2 cross layer  out = x0 * p + x,  p = x W + b  (blocks/cross.py:188-202 under the tape) ---------
//   d loss / d p  = g   = dout * x0
//   d loss / d x0 (this layer's share) = dout * p            -> accumulated over the layers of a CrossBlock
//   d loss / d x  = g W^T + dout                              -> the residual add rides in the GEMM epilogue
//   dW = x^T g,  db = column sums of g
// Three phases selected by the non-NULL outputs (the caller may run dX on its launch stream and dW beside it on another):
//   dx0_acc != NULL : ONE pass over [M, d]: g = dout * x0 and dx0_acc = (accumulate ? dx0_acc : 0) + dout * p
//   dx      != NULL : dx = a W^T + dout on the MFMA GEMM (a = g; for a low-rank layer the caller passes a = d loss / d h and
//                     W = U [d, r]: `r` is the contraction width)
//   dW      != NULL : dW [d, d] = x^T g, db [d]  (workspace: mh_linear_bwd_workspace_bytes(M, d, d))
namespace {
__global__ __launch_bounds__(256) void cross_bwd_pre_kernel(const f32x4* __restrict__ dout, const f32x4* __restrict__ x0,
                                                           const f32x4* __restrict__ p, f32x4* __restrict__ g,
                                                           f32x4* __restrict__ dx0_acc, int accumulate, int64_t n4) {
    // one float4 per thread, one pass: the hardware workgroup dispatcher beats a persistent loop for pure streaming on this
    // chip (profiles/r3_notes.md, copy lab); dout / x0 / p are read once here -> streaming loads
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 d = __builtin_nontemporal_load(dout + i);
    const f32x4 a = __builtin_nontemporal_load(x0 + i);
    const f32x4 q = __builtin_nontemporal_load(p + i);
    f32x4 acc = d * q;
    if (accumulate) acc += dx0_acc[i];
    g[i] = d * a;
    dx0_acc[i] = acc;
}
}  // namespace

int32_t mh_cross_layer_bwd(const float* x0, const float* x, const float* p, const float* dout, const float* W, int64_t M,
                           int32_t d, int32_t r, float* g, float* dx0_acc, int32_t accumulate_dx0, float* dx, float* dW,
                           float* db, void* workspace, int64_t workspace_bytes, mh_stream_t stream) {
    MH_REQUIRE(M >= 1 && d >= 4 && d % 4 == 0, "mh_cross_layer_bwd: d=%d must be a positive multiple of 4 (zero-padded layers)", d);
    MH_REQUIRE(g && dout, "mh_cross_layer_bwd: g and dout are required");
    MH_REQUIRE(((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dout)) & 15) == 0,
               "mh_cross_layer_bwd: operands must be 16-byte aligned");
    hipStream_t s = mh_stream(stream);
    if (dx0_acc) {
        MH_REQUIRE(x0 && p, "mh_cross_layer_bwd: x0 and p are required for the element-wise phase");
        MH_REQUIRE(((reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(dx0_acc)) & 15) == 0,
                   "mh_cross_layer_bwd: operands must be 16-byte aligned");
        const int64_t n4 = M * (int64_t)(d / 4);
        MH_LAUNCH(cross_bwd_pre_kernel, dim3((unsigned)mh_ceil_div(n4, 256)), dim3(256), 0, s,
                           reinterpret_cast<const f32x4*>(dout), reinterpret_cast<const f32x4*>(x0),
                           reinterpret_cast<const f32x4*>(p), reinterpret_cast<f32x4*>(g), reinterpret_cast<f32x4*>(dx0_acc),
                           accumulate_dx0 ? 1 : 0, n4);
        MH_CHECK_LAUNCH("mh_cross_layer_bwd(pre)");
    }
    if (dx) {
        MH_REQUIRE(W && r >= 4 && r % 4 == 0, "mh_cross_layer_bwd: W [d, r] with r a positive multiple of 4 is required for dx");
        const int32_t st = mh_internal_gemm_nt_ep(g, (int64_t)r, W, (int64_t)r, M, d, r, dx, (int64_t)d, nullptr, 0, MH_ACT_NONE,
                                                  dout, (int64_t)d, s);
        if (st != MH_OK) return st;
    }
    if (dW) {
        MH_REQUIRE(x, "mh_cross_layer_bwd: x is required for dW");
        return mh_linear_bias_act_bwd(x, (int64_t)d, nullptr, nullptr, 0, g, (int64_t)d, M, d, d, MH_ACT_NONE, MH_ACT_NONE,
                                      nullptr, 0, dW, db, workspace, workspace_bytes, stream);
    }
    return MH_OK;
}

}  // extern "C"
